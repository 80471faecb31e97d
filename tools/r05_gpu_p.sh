#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_band_crop.py -q -x > gpurun_out/p_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/p_tests.log
grep -v "Gloo\|amdgpu.ids" gpurun_out/p_tests.log | tail -30
: > gpurun_out/p_bench.log
for v in 0 1 0 1; do
  SZN_BAND_CROP=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/p_bench.json 2>> gpurun_out/p_bench.err
  python - <<PY >> gpurun_out/p_bench.log
import json
d=json.load(open("gpurun_out/p_bench.json"))
print("band_crop=$v", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["final_loss"])
PY
done
cat gpurun_out/p_bench.log
