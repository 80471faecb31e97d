#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py -q 2>&1 | tail -2
for i in 1 2; do python bench.py --sub-record b1 --steps 20 2>/dev/null | grep SUBRECORD | python -c "
import json,sys
d=json.loads(sys.stdin.read()[len('SUBRECORD '):])
print('B=1 bf16 eager %.3f graph %.3f | fp32 eager %.3f' % (d['bf16']['eager_ms_per_step'], d['bf16']['graph_ms_per_step'], d['fp32']['eager_ms_per_step']))"; done
