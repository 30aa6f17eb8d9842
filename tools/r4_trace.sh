cd $GRAFT_REPO_ROOT
for q in 8; do LCGP_HIP_LIB=$GRAFT_REPO_ROOT/lcgp_amd/liblcgp_hip_trace.so timeout -k 10 300 python tools/dag_trace.py --q $q --dag 2 --bucket 200 2>&1 | grep -v amdgpu.ids > gpurun_out/trace_q$q.txt; head -45 gpurun_out/trace_q$q.txt; done
