#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_band_crop.py tests/test_gpu_model.py tests/test_gpu_engine.py -q 2>&1 | grep -v "Gloo\|amdgpu" | tail -12
for v in 0 1 0 1; do
  SZN_BAND_C11=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c11=$v', d['value'], d['ms_per_step'], d['config']['final_loss'])"
done
