#!/bin/bash
# kernel trace of a few evaluations of the headline configuration at q components: bash tools/trace_q.sh <q> <tag> [sched field=value ...]
Q=$1; TAG=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$TAG -- python3 $GRAFT_REPO_ROOT/tools/run_evals.py 3 $Q 4 "$@" > /tmp/tr_$TAG.log 2>&1
F=$(find /tmp/tr_$TAG -name '*kernel_trace.csv' | head -1)
cd $GRAFT_REPO_ROOT
python tools/trace_view.py $F > gpurun_out/trace_$TAG.txt
tail -25 gpurun_out/trace_$TAG.txt
