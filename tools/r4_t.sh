cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_dag.py -x -q --durations=8 2>&1 | grep -v amdgpu.ids | tail -16
