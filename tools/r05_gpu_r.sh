#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_band_crop.py tests/test_gpu_engine.py -q 2>&1 | tail -3
: > gpurun_out/r_bench.log
for v in a b; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r_bench.json 2>> gpurun_out/r_bench.err
  python - <<PY >> gpurun_out/r_bench.log
import json
d=json.load(open("gpurun_out/r_bench.json"))
print("blocks=$v", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["final_loss"])
PY
done
cat gpurun_out/r_bench.log
