#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_band_crop.py tests/test_gpu_engine.py tests/test_gpu_fullsize.py tests/test_gpu_fcn8s.py tests/test_gpu_lowprec.py -q > gpurun_out/p_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/p_tests.log
grep -v "Gloo\|amdgpu.ids" gpurun_out/p_tests.log | tail -30
