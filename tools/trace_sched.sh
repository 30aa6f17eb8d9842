#!/bin/bash
# kernel timeline of one evaluation of the headline configuration under a schedule override: bash tools/trace_sched.sh <tag> <q> field=value ...
TAG=$1; Q=$2; shift; shift
export TMPDIR=/tmp
rm -rf /tmp/trs_$TAG
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/trs_$TAG -- python3 $GRAFT_REPO_ROOT/tools/run_evals.py 3 $Q 4 "$@" > /tmp/trs_$TAG.log 2>&1)
F=$(find /tmp/trs_$TAG -name '*kernel_trace.csv' | head -1)
python tools/trace_view.py $F > gpurun_out/timeline_$TAG.txt
