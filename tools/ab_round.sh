#!/bin/bash
# One gpurun call: results of the builds build/libv_<name>.so compared bit for bit, then timed alternately on this box.
#   usage: bash tools/ab_round.sh <tag> name1 name2 ...      (writes gpurun_out/<tag>_*.txt)
TAG=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  python tools/dump_eval.py build/libv_$v.so gpurun_out/${TAG}_eval_$v.npy || exit 1
done
python - "$TAG" "$@" <<'PY' | tee gpurun_out/${TAG}_equal.txt
import sys, numpy as np
tag, names = sys.argv[1], sys.argv[2:]
ref = np.load('gpurun_out/%s_eval_%s.npy' % (tag, names[0]))
for v in names[1:]:
    a = np.load('gpurun_out/%s_eval_%s.npy' % (tag, v))
    print(v, 'vs', names[0], 'bitwise equal:', bool(np.array_equal(a, ref)), 'max rel diff %.3e' % float(np.max(np.abs(a - ref) / (np.abs(ref) + 1e-300))))
PY
for q in ${ABQ:-8 1}; do
  bash tools/ab_builds.sh "--q $q --reps 3 --steps 8 ${ABEXTRA:-}" "$@" 2>&1 | tee gpurun_out/${TAG}_ab_q$q.txt
done
