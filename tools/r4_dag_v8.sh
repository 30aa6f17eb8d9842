#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r4_dag_v8
mkdir -p $OUT
cd $ROOT
for q in 8 4 2 1; do
  timeout -k 10 300 python tools/ab.py --q $q --reps 3 --steps 10 "launches:dag=0" "dag2:dag=2" "dag2_potrf_only:dag=2,dag_flags=256" > $OUT/ab_q$q.txt 2>&1 || { tail -5 $OUT/ab_q$q.txt; exit 1; }
  grep -v amdgpu.ids $OUT/ab_q$q.txt
done
LCGP_HIP_LIB=$ROOT/lcgp_amd/liblcgp_hip_trace.so timeout -k 10 300 python tools/dag_trace.py --q 8 --dag 2 --bucket 250 dag_flags=256 2>&1 | grep -v amdgpu.ids > $OUT/trace_q8.txt; head -32 $OUT/trace_q8.txt
