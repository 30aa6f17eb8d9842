#!/bin/bash
# Runs on the GPU box (gpurun): the measurements DESIGN.md / profiles/ quote for one round.
#   tools/profile_round.sh <tag>        -> gpurun_out/<tag>/...
# rocprofv3 passes: --kernel-trace --stats on its own; every --pmc pass on its own with --kernel-trace only
# (never together with the hip/hsa/memory trace domains); the program itself follows `--`.
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
step() { echo "[profile_round] $*"; }
PART=${2:-all}        # A = bench lines + counter passes, B = traces, fits (a gpurun call is limited to 20 minutes)
if [ "$PART" != B ]; then

step "bench (default: N=1, headline configuration)"
python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
step "bench, other configurations"
for c in 2 4 5; do python3 $ROOT/bench.py --config $c --steps 6 --warmup 2 > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err || exit 1; done
step "bench, one rank's share of the headline configuration (q_local = 4, 2, 1)"
for q in 4 2 1; do python3 $ROOT/bench.py --q $q --steps 12 --warmup 3 --no-cpu-baseline > $OUT/bench_q$q.json 2> $OUT/bench_q$q.err || exit 1; done
step "bench, the slowest rank's share of configs[4] on 4 GPUs (q = 6 -> 2 / 2 / 1 / 1) and the 1-component share"
for q in 2 1; do python3 $ROOT/bench.py --config 5 --q $q --steps 12 --warmup 3 --no-cpu-baseline > $OUT/bench_cfg5_q$q.json 2> $OUT/bench_cfg5_q$q.err || exit 1; done
step "host overhead (caller-owned plan vs plan per call; q = 8 and one rank's share)"
python3 $ROOT/tools/host_overhead.py 3 > $OUT/host_overhead.txt 2>&1 || exit 1
python3 $ROOT/tools/host_overhead.py 3 1 >> $OUT/host_overhead.txt 2>&1 || exit 1
step "kernel stats of the bench command"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1 || exit 1
step "PMC: VALU / occupancy counters (cfg3, 3 evaluations)"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_valu -- python3 $ROOT/tools/run_evals.py 3 0 3 > $OUT/pmc_valu.log 2>&1 || exit 1
step "PMC: the same counters, cfg4 (float32, n=16384), 1 evaluation"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_valu_cfg4 -- python3 $ROOT/tools/run_evals.py 4 0 1 > $OUT/pmc_valu_cfg4.log 2>&1 || exit 1
step "PMC: FETCH_SIZE"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/run_evals.py 3 0 3 > $OUT/pmc_fetch.log 2>&1 || exit 1
step "PMC: WRITE_SIZE"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/run_evals.py 3 0 3 > $OUT/pmc_write.log 2>&1 || exit 1
step "PMC: MFMA busy"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $ROOT/tools/run_evals.py 3 0 3 > $OUT/pmc_mfma.log 2>&1 || exit 1
step "PMC: MFMA busy, one component (the 8-GPU share: inverse formed behind the chain)"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma_q1 -- python3 $ROOT/tools/run_evals.py 3 1 3 > $OUT/pmc_mfma_q1.log 2>&1 || exit 1
fi
if [ "$PART" = A ]; then step done; exit 0; fi
step "kernel trace, one component, progressive (default) and classic schedule"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_q1 -- python3 $ROOT/tools/run_evals.py 3 1 4 > $OUT/trace_q1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_q1_classic -- python3 $ROOT/tools/run_evals.py 3 1 4 progressive_tiles=0 > $OUT/trace_q1_classic.log 2>&1 || exit 1
python3 $ROOT/tools/trace_view.py $(find $OUT/trace_q1 -name '*kernel_trace.csv' | head -1) > $OUT/timeline_q1_progressive.txt
python3 $ROOT/tools/trace_view.py $(find $OUT/trace_q1_classic -name '*kernel_trace.csv' | head -1) > $OUT/timeline_q1_classic.txt
step "kernel trace, q = 8 (the headline configuration)"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_q8 -- python3 $ROOT/tools/run_evals.py 3 8 4 > $OUT/trace_q8.log 2>&1 || exit 1
python3 $ROOT/tools/trace_view.py $(find $OUT/trace_q8 -name '*kernel_trace.csv' | head -1) > $OUT/timeline_q8.txt
step "fit() wall-clock of every configuration"
for c in 3 2 5 4; do python3 $ROOT/tools/fit_wallclock.py $c >> $OUT/fit_wallclock.txt 2>&1 || true; done
python3 $ROOT/tools/fit_wallclock.py 4 float64 >> $OUT/fit_wallclock.txt 2>&1 || true      # the float32 configuration fitted in float64: the comparison
step "two ranks on this one GPU over gloo (rehearsal of the N > 1 launch line; RCCL needs one GPU per rank)"
cd $ROOT && python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --backend gloo --no-cpu-baseline > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err || exit 1
step done
