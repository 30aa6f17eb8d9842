cd $GRAFT_REPO_ROOT
python tools/host_overhead.py 3 2>&1 | grep -v amdgpu.ids
python tools/host_overhead.py 3 1 2>&1 | grep -v amdgpu.ids
for q in 1 2 4; do python bench.py --q $q --steps 20 --warmup 3 --no-cpu-baseline --no-fit 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('q', o['config'].get('q_local', '?'), o['ms_per_step'], o.get('projected_from_q_local'))"; done
