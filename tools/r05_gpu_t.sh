#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_wire.py tests/test_gpu_band_crop.py -q 2>&1 | tail -3
PREV=$GRAFT_REPO_ROOT/zeroshotsemanticsegmentation_amd/lib_prev/libszn_hip.so
LAYERS=conv2_2,conv3_2,conv4_1,conv4_2,conv5_1
for rep in 1 2; do
  echo "== prev"; SZN_LIB_PATH=$PREV python tools/bench_conv.py --layers $LAYERS --what wgrad --iters 20 2>/dev/null
  echo "== new"; python tools/bench_conv.py --layers $LAYERS --what wgrad --iters 20 2>/dev/null
done
for v in prev new prev new; do
  unset SZN_LIB_PATH; if [ $v = prev ]; then export SZN_LIB_PATH=$PREV; fi
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lib=$v', d['value'], d['ms_per_step'])"
done
