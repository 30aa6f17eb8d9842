#!/bin/bash
# Same-box A/B of two or more BUILDS of the library (timing differs by +-1.5 % between boxes of the pool, so builds are
# compared by alternating them in one gpurun call).
#   usage: bash tools/ab_builds.sh "<tools/ab.py arguments>" name1 name2 ...      with the builds at build/libv_<name>.so
# Each build is copied over lcgp_amd/liblcgp_hip.so in turn (the product library is restored at the end); results
# should be checked for equality first (tools/dump_eval.py writes NLL + gradient of three configurations to an .npy).
ARGS=$1; shift
cp lcgp_amd/liblcgp_hip.so /tmp/cur.so
for rep in 1 2; do
  for v in "$@"; do
    cp build/libv_$v.so lcgp_amd/liblcgp_hip.so
    echo "== $v"; python tools/ab.py $ARGS "d:" || { cp /tmp/cur.so lcgp_amd/liblcgp_hip.so; exit 1; }
  done
done
cp /tmp/cur.so lcgp_amd/liblcgp_hip.so
