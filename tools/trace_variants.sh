#!/bin/bash
# per-kernel sums of one evaluation for several builds (build/libv_<name>.so): bash tools/trace_variants.sh <tag> <q> name1 name2 ...
TAG=$1; Q=$2; shift; shift
cp lcgp_amd/liblcgp_hip.so /tmp/cur.so
export TMPDIR=/tmp
for v in "$@"; do
  cp build/libv_$v.so lcgp_amd/liblcgp_hip.so
  rm -rf /tmp/tr_$v
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$v -- python3 $GRAFT_REPO_ROOT/tools/run_enqueue.py $Q 4 > /tmp/tr_$v.log 2>&1)
  F=$(find /tmp/tr_$v -name '*kernel_trace.csv' | head -1)
  python tools/trace_view.py $F > gpurun_out/trace_${TAG}_$v.txt
  echo "== $v"; sed -n '/--- last evaluation/,$p' gpurun_out/trace_${TAG}_$v.txt | head -14
done
cp /tmp/cur.so lcgp_amd/liblcgp_hip.so
