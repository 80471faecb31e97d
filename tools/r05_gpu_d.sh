#!/bin/bash
# round 5, GPU call D: conv_wgrad_taps with K steps running across tile boundaries -- parity + A/B against the previous build
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_wire.py -q -x > gpurun_out/d_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/d_tests.log
tail -4 gpurun_out/d_tests.log
PREV=$GRAFT_REPO_ROOT/zeroshotsemanticsegmentation_amd/lib_prev/libszn_hip.so
LAYERS=conv1_2,conv2_1,conv2_2,conv3_1,conv3_2,conv4_1,conv4_2,conv5_1
: > gpurun_out/d_conv.log
for rep in 1 2; do
  echo "== lib=prev rep $rep" >> gpurun_out/d_conv.log
  SZN_LIB_PATH=$PREV python tools/bench_conv.py --layers $LAYERS --what wgrad --iters 20 2>/dev/null >> gpurun_out/d_conv.log
  for fm in 0 1; do
    echo "== lib=new fill=$fm rep $rep" >> gpurun_out/d_conv.log
    SZN_WGT_FILL=$fm python tools/bench_conv.py --layers $LAYERS --what wgrad --iters 20 2>/dev/null >> gpurun_out/d_conv.log
  done
done
cat gpurun_out/d_conv.log
: > gpurun_out/d_bench.log
for v in prev new0 new1 prev new0 new1; do
  unset SZN_LIB_PATH SZN_WGT_FILL
  if [ $v = prev ]; then export SZN_LIB_PATH=$PREV; fi
  if [ $v = new1 ]; then export SZN_WGT_FILL=1; fi
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/d_bench_$v.json 2>> gpurun_out/d_bench.err
  python - <<PY >> gpurun_out/d_bench.log
import json
d=json.load(open("gpurun_out/d_bench_$v.json"))
r=d["roofline"]
print("lib=$v", d["value"], d["ms_per_step"], r["kernel"], r["frac"], r.get("step_mfma_frac"))
PY
done
cat gpurun_out/d_bench.log
