#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
run() { timeout -k 5 300 python tools/dag_diff.py "$@" 2>&1 | grep "reps with\|Error\|error" | cut -c1-200; }
LCGP_TEST_DAG_REPS=40 timeout -k 10 200 tests/native/test_kernels 512 700 > gpurun_out/native_fix.txt 2>&1; echo "native rc=$?"; grep "persistent" gpurun_out/native_fix.txt | grep -v " 0 bad\|40 bad" | cut -c1-150
run 2048 6 12 1 0 progressive_tiles=0
run 512 3 60 1 0
run 1500 4 20 1 0 syrk_small_tiles=16
run 2048 2 20 1 0
run 4096 8 4 1 0
