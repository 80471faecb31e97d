#!/bin/bash
# round 5: the whole GPU suite, the driver-style bench line, and the profile set (tools/r05_profile.sh <tag>)
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-a}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/full_${T}_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/full_${T}_tests.log
tail -6 gpurun_out/full_${T}_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full_${T}_smoke.log 2>&1; tail -2 gpurun_out/full_${T}_smoke.log
timeout 1200 python bench.py --steps 20 --warmup 3 > gpurun_out/full_${T}_bench.json 2> gpurun_out/full_${T}_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/full_${T}_bench.json"))
r=d["roofline"]
print("bench", d["value"], d["ms_per_step"], r["kernel"], r["frac"], r.get("step_mfma_frac"))
print("phase2", d["phase2"].get("ms_per_step"), "fp32", d["fp32"].get("ms_per_step"), "b1", d["b1"].get("bf16",{}).get("ms_per_step"))
print("comm", d["comm"].get("ms_per_step"))
print("proj", d["projection"]["nominal_shape"]["frac"], d["projection"]["nominal_shape_randn"]["frac"], [x["frac"] for x in d["projection"]["true_shape"]])
PY
bash tools/r05_profile.sh $T > gpurun_out/full_${T}_profile.log 2>&1
tail -30 gpurun_out/full_${T}_profile.log
