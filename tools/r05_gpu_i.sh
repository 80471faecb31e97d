#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused_adam.py -q -x > gpurun_out/i_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/i_tests.log
tail -15 gpurun_out/i_tests.log
: > gpurun_out/i_wa.log
for b in 8 1; do
for cfg in "0 -" "1 0" "1 1" "1 2" "1 3" "1 4" "1 6"; do
  set -- $cfg
  export SZN_WGW_HALF=$1
  if [ "$2" = "-" ]; then unset SZN_WGH_STAGGER; else export SZN_WGH_STAGGER=$2; fi
  python tools/bench_wgrad_adam.py --batch $b 2>/dev/null >> gpurun_out/i_wa.log
done; done
python - <<PY
import json
for l in open("gpurun_out/i_wa.log"):
    d=json.loads(l)
    f=d["layers"]["fc6"]
    print("B=%d half=%s stagger=%s | fc6 wgrad %.0f adam %.0f fused %.0f fused+grads %.0f us (%.0f GB/s) | fc7 fused %.0f" % (d["batch"], d["half"], d["half_stagger"], f["wgrad_us"], f["adam_us"], f["fused_us"], f["fused_keep_grads_us"], f["fused_GBps"], d["layers"]["fc7"]["fused_us"]))
PY
