#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r4_dag_diag
mkdir -p $OUT
cd $ROOT
for f in 0 16; do
  LCGP_TEST_DAG_FLAGS=$f LCGP_TEST_DAG_REPS=30 timeout -k 10 200 tests/native/test_kernels 512 700 > $OUT/native_f$f.txt 2>&1
  echo "flags=$f rc=$?"; grep "persistent" $OUT/native_f$f.txt | grep -v " 0 bad\|30 bad" | cut -c1-150
done
