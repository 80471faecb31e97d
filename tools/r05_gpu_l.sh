#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_engine.py tests/test_gpu_fullsize.py tests/test_gpu_parity_full.py tests/test_gpu_sizes.py -q > gpurun_out/l_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/l_tests.log
grep -v "Gloo\|amdgpu.ids" gpurun_out/l_tests.log | tail -12
