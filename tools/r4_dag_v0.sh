#!/bin/bash
# first GPU contact of the persistent factorisation launch: native driver, the DAG tests, then timings
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r4_dag_v0
mkdir -p $OUT
cd $ROOT
echo "[native]"; timeout -k 10 300 tests/native/test_kernels 64 200 512 > $OUT/native.txt 2>&1; rc=$?; tail -5 $OUT/native.txt; [ $rc -eq 0 ] || { grep -n FAIL $OUT/native.txt | head; exit 1; }
echo "[pytest dag]"; timeout -k 10 900 python -m pytest tests/test_gpu_dag.py -x -q > $OUT/pytest_dag.txt 2>&1; rc=$?; tail -15 $OUT/pytest_dag.txt; [ $rc -eq 0 ] || exit 1
for q in 8 4 2 1; do
  echo "[ab q=$q]"; timeout -k 10 300 python tools/ab.py --q $q --reps 3 --steps 10 "launches:dag=0" "dag:dag=1" > $OUT/ab_q$q.txt 2>&1 || { tail -5 $OUT/ab_q$q.txt; exit 1; }
  cat $OUT/ab_q$q.txt
done
