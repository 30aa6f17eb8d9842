#!/bin/bash
# usage: bash tools/ab_lib.sh <ab.py args...>   -- runs tools/ab.py alternately with the current and the old library
cp lcgp_amd/liblcgp_hip.so /tmp/cur.so
for rep in 1 2; do
  for lib in cur old; do
    if [ $lib = old ]; then cp build/liblcgp_old.so lcgp_amd/liblcgp_hip.so; else cp /tmp/cur.so lcgp_amd/liblcgp_hip.so; fi
    echo "== $lib"; python tools/ab.py "$@"
  done
done
cp /tmp/cur.so lcgp_amd/liblcgp_hip.so
