#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_wire.py tests/test_gpu_ddp_single_gpu.py tests/test_gpu_rccl_world1.py tests/test_gpu_bench_contract.py tests/test_gpu_fused_adam.py -q > gpurun_out/b_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/b_tests.log
tail -15 gpurun_out/b_tests.log
