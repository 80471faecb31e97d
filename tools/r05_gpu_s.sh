#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python bench.py --precision fp32 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/s_fp32.json 2> gpurun_out/s_fp32.err
python - <<PY
import json
d=json.load(open("gpurun_out/s_fp32.json"))
print("fp32", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("step_mfma_frac"), d["roofline"].get("step_gflop_executed"))
for row in d["kernels"]["rows"][:24]:
    print("   %-52s %-28s %5.1f %8.3f ms  %s" % (row["kernel"][:52], row["entry"], row["calls_per_step"], row["ms_per_step"], row.get("frac")))
PY
