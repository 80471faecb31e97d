#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_headline_pin.py -q -s > gpurun_out/c_pin.log 2>&1
echo "pin rc $?" >> gpurun_out/c_pin.log
grep -v "^\[Gloo\]\|amdgpu.ids" gpurun_out/c_pin.log | tail -50
timeout 2400 python -m pytest tests/test_gpu_wire.py tests/test_gpu_ddp_single_gpu.py tests/test_gpu_bench_contract.py -q > gpurun_out/c_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/c_tests.log
tail -8 gpurun_out/c_tests.log
