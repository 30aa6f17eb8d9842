#!/bin/bash
# Hosted panels (left-looking at the outer level) against the default schedule: same-process A/B at q = 8, 4, 2, 1, the
# kernel timeline and the HBM / MFMA counters of the hosted factorisation.  Runs on the GPU box:  bash tools/r05_left_looking.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/ll
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for q in 8 4 2 1; do
  python3 $ROOT/tools/ab.py --q $q --reps 3 --steps 10 "default:" "hosted panels, defer 1:hosted=1,hosted_defer=1" "hosted panels, defer 2:hosted=1" "hosted panels, defer 4:hosted=1,hosted_defer=4" >> $OUT/ab.txt 2>&1 || exit 1
done
python3 $ROOT/tools/host_stamps.py run 8 > $OUT/stamps_q8.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/tools/run_evals.py 3 8 4 hosted=1 > $OUT/trace.log 2>&1 || exit 1
python3 $ROOT/tools/trace_view.py $(find $OUT/trace -name '*kernel_trace.csv' | head -1) > $OUT/timeline_q8_hosted.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/run_evals.py 3 0 3 hosted=1 > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/run_evals.py 3 0 3 hosted=1 > $OUT/pmc_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $ROOT/tools/run_evals.py 3 0 3 hosted=1 > $OUT/pmc_mfma.log 2>&1 || exit 1
for d in pmc_fetch pmc_write pmc_mfma; do python3 $ROOT/tools/pmc_summary.py $OUT/$d > $OUT/$d.txt; done
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma
echo done
