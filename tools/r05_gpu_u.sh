#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_band_crop.py tests/test_gpu_engine.py tests/test_gpu_fcn8s.py tests/test_gpu_model.py -q 2>&1 | grep -v "Gloo\|amdgpu" | tail -6
for v in 0 1 0 1; do
  SZN_BAND_FUSE=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fuse=$v', d['value'], d['ms_per_step'])"
done
for v in 0 1; do SZN_BAND_FUSE=$v python bench.py --sub-record fp32 --steps 9 2>/dev/null | grep SUBRECORD | cut -c1-100; done
