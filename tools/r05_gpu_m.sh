#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/m_b1.log
for v in 0 auto 0 auto; do
  SZN_WGRAD_STREAM=$v python bench.py --sub-record b1 --steps 20 2>/dev/null | grep SUBRECORD | python -c "
import json,sys
d=json.loads(sys.stdin.read()[len('SUBRECORD '):])
print('B=1 wgrad_stream=$v bf16 eager %.3f graph %s %.3f | fp32 eager %.3f graph %.3f' % (d['bf16']['eager_ms_per_step'], d['bf16'].get('graph'), d['bf16']['ms_per_step'], d['fp32']['eager_ms_per_step'], d['fp32']['ms_per_step']))" >> gpurun_out/m_b1.log
done
cat gpurun_out/m_b1.log
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_headline_pin.py > gpurun_out/m_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/m_tests.log
grep -v "Gloo\|amdgpu.ids" gpurun_out/m_tests.log | tail -8
