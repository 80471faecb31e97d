#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_wire.py -q -x > gpurun_out/f_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/f_tests.log
tail -4 gpurun_out/f_tests.log
PREV=$GRAFT_REPO_ROOT/zeroshotsemanticsegmentation_amd/lib_prev/libszn_hip.so
ABL=$GRAFT_REPO_ROOT/zeroshotsemanticsegmentation_amd/lib_ablate/libszn_hip.so
LAYERS=conv1_2,conv2_1,conv2_2,conv3_1,conv3_2,conv4_1,conv4_2,conv5_1
: > gpurun_out/f_conv.log
for rep in 1 2; do
  echo "== lib=prev rep $rep" >> gpurun_out/f_conv.log
  SZN_LIB_PATH=$PREV python tools/bench_conv.py --layers $LAYERS --what wgrad --iters 20 2>/dev/null >> gpurun_out/f_conv.log
  echo "== lib=new rep $rep" >> gpurun_out/f_conv.log
  python tools/bench_conv.py --layers $LAYERS --what wgrad --iters 20 2>/dev/null >> gpurun_out/f_conv.log
done
echo "== new, SZN_WGT_ABLATE=1 (no LDS-DMA in the loop)" >> gpurun_out/f_conv.log
SZN_LIB_PATH=$ABL SZN_WGT_ABLATE=1 python tools/bench_conv.py --layers $LAYERS --what wgrad --iters 20 2>/dev/null >> gpurun_out/f_conv.log
cat gpurun_out/f_conv.log
: > gpurun_out/f_bench.log
for v in prev new prev new; do
  unset SZN_LIB_PATH
  if [ $v = prev ]; then export SZN_LIB_PATH=$PREV; fi
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/f_bench_$v.json 2>> gpurun_out/f_bench.err
  python - <<PY >> gpurun_out/f_bench.log
import json
d=json.load(open("gpurun_out/f_bench_$v.json"))
r=d["roofline"]
print("lib=$v", d["value"], d["ms_per_step"], r["kernel"], r["frac"], r.get("step_mfma_frac"))
PY
done
cat gpurun_out/f_bench.log
