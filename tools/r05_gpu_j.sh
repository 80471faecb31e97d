#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
: > gpurun_out/j_wa.log
for b in 8; do
for cfg in "0 0" "0 1" "1 0" "1 1"; do
  set -- $cfg
  SZN_WGW_HALF=$1 SZN_WGW_XCD2=$2 python tools/bench_wgrad_adam.py --batch $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['layers']['fc6']
print('B=%d half=$1 xcd2=$2 | fc6 wgrad %.0f adam %.0f fused %.0f fused+grads %.0f us' % (d['batch'], f['wgrad_us'], f['adam_us'], f['fused_us'], f['fused_keep_grads_us']))" >> gpurun_out/j_wa.log
done; done
cat gpurun_out/j_wa.log
: > gpurun_out/j_bench.log
for cfg in "0 0" "1 0" "0 1" "1 1" "0 0" "1 0"; do
  set -- $cfg
  SZN_WGW_HALF=$1 SZN_WGW_XCD2=$2 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/j_bench.json 2>> gpurun_out/j_bench.err
  python - <<PY >> gpurun_out/j_bench.log
import json
d=json.load(open("gpurun_out/j_bench.json"))
print("B=8 half=$1 xcd2=$2", d["value"], d["ms_per_step"])
PY
  SZN_WGW_HALF=$1 SZN_WGW_XCD2=$2 python bench.py --sub-record b1 --steps 20 2>/dev/null | grep SUBRECORD | python -c "
import json,sys
d=json.loads(sys.stdin.read()[len('SUBRECORD '):])
print('B=1 half=$1 xcd2=$2 bf16 eager %.3f graph %.3f' % (d['bf16']['eager_ms_per_step'], d['bf16']['ms_per_step']))" >> gpurun_out/j_bench.log
done
cat gpurun_out/j_bench.log
