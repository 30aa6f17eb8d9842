#!/bin/bash
# quick check of a library change on the GPU box: correctness on three sizes (native test + goldens), then timing at q = 8, 4, 2, 1
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_configs.py tests/test_gpu_edge_cases.py tests/test_gpu_api.py::test_c_abi_from_plain_cpp_without_torch -x -q 2>&1 | tail -3 || exit 1
for q in 8 4 2 1; do timeout -k 10 200 python tools/ab.py --q $q --reps 3 --steps 10 "default:" | grep default; done
for c in 2 5; do timeout -k 10 200 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-fit --no-stages 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg$c', round(d['ms_per_step'],4), 'ms')"; done
