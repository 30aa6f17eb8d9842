#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r4_prog_dag
mkdir -p $OUT
cd $ROOT
for q in 4 2 1; do
  timeout -k 10 400 python tools/ab.py --q $q --reps 3 --steps 10 "launches:" "launches_prog:progressive_tiles=1000000" "dag1_prog:dag=1,progressive_tiles=1000000" "dag1_prog_cap2k:dag=1,progressive_tiles=1000000,fill_leaf=2000,fill_step=2000" "dag1_prog_cap8k:dag=1,progressive_tiles=1000000,fill_leaf=8000,fill_step=8000" "dag1_prog_cap8k_lauum:dag=1,progressive_tiles=1000000,fill_leaf=8000,fill_step=8000,progressive_lauum=64" > $OUT/ab_q$q.txt 2>&1
  grep -v amdgpu.ids $OUT/ab_q$q.txt
done
