#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_wire.py tests/test_gpu_engine.py tests/test_gpu_parity_full.py -q > gpurun_out/o_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/o_tests.log
grep -v "Gloo\|amdgpu.ids" gpurun_out/o_tests.log | tail -5
python bench.py --sub-record fp32 --steps 9 2>/dev/null | grep SUBRECORD | cut -c1-600
