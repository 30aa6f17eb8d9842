cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r02e; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/run_evals.py 3 0 3 > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/run_evals.py 3 0 3 > $OUT/pmc_write.log 2>&1 || exit 1
python3 $ROOT/bench.py > $OUT/bench_pre.json 2> $OUT/bench_pre.err
