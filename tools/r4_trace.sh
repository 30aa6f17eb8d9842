cd $GRAFT_REPO_ROOT
LCGP_HIP_LIB=$GRAFT_REPO_ROOT/lcgp_amd/liblcgp_hip_trace.so timeout -k 10 300 python tools/dag_trace.py --q 8 --dag 2 --bucket 500 --chain 80 dag_flags=256 2>&1 | grep -v amdgpu.ids > gpurun_out/trace_q8_chain.txt; grep "leaf  taken" gpurun_out/trace_q8_chain.txt
