#!/bin/bash
# the GPU suite on the production library, then the experiment-build variants (the tests the production library skips) on libldweaver_amd_exp.so
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r06u_gputest.log 2>&1; echo "tests rc $?"; tail -9 gpurun_out/r06s_gputest2.log
LDW_AMD_LIB=$PWD/ldweaver_amd/libldweaver_amd_exp.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or popcount or fp32_screen or edge_cases or default_library or apx_path_is_exact or spans_equal" > gpurun_out/r06u_gputest_exp.log 2>&1; echo "exp tests rc $?"; tail -3 gpurun_out/r06u_gputest_exp.log
for k in lds pipe; do
LDW_APX_KERNEL=$k LDW_AMD_LIB=$PWD/ldweaver_amd/libldweaver_amd_exp.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "apx_path_is_exact or apx_path_on_irregular" > gpurun_out/r06u_gputest_exp_$k.log 2>&1; echo "exp $k tests rc $?"; tail -2 gpurun_out/r06u_gputest_exp_$k.log
done
