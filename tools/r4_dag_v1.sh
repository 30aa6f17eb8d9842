#!/bin/bash
# persistent factorisation launch: tests, A/B of the executors, task trace
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r4_dag}
mkdir -p $OUT
cd $ROOT
echo "[native]"; LCGP_TEST_DAG_REPS=10 timeout -k 10 300 tests/native/test_kernels 64 200 512 700 > $OUT/native.txt 2>&1; rc=$?; tail -1 $OUT/native.txt; [ $rc -eq 0 ] || { grep -n FAIL $OUT/native.txt | head; exit 1; }
echo "[pytest dag]"; timeout -k 10 900 python -m pytest tests/test_gpu_dag.py -x -q > $OUT/pytest_dag.txt 2>&1; rc=$?; tail -4 $OUT/pytest_dag.txt; [ $rc -eq 0 ] || { tail -30 $OUT/pytest_dag.txt; exit 1; }
for q in 8 4 2 1; do
  timeout -k 10 300 python tools/ab.py --q $q --reps 3 --steps 10 "launches:dag=0" "dag2:dag=2" "dag2hand:dag=2,dag_flags=128" > $OUT/ab_q$q.txt 2>&1 || { tail -5 $OUT/ab_q$q.txt; exit 1; }
  grep -v amdgpu.ids $OUT/ab_q$q.txt
done
if [ -f lcgp_amd/liblcgp_hip_trace.so ]; then
  for q in 8; do
    LCGP_HIP_LIB=$ROOT/lcgp_amd/liblcgp_hip_trace.so timeout -k 10 300 python tools/dag_trace.py --q $q --dag 2 --bucket 250 2>&1 | grep -v amdgpu.ids > $OUT/trace_q$q.txt || { tail -5 $OUT/trace_q$q.txt; exit 1; }
    head -50 $OUT/trace_q$q.txt
  done
fi
