#!/bin/bash
# round 5, GPU call H: conv5_x on conv_igemm_8ph<1,1> inside the step (SZN_IGEMM_8PH=1) vs conv_igemm_v2: per-kernel rows + step time
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
: > gpurun_out/h_bench.log
for v in 0 1 0 1; do
  SZN_IGEMM_8PH=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/h_bench_$v.json 2>> gpurun_out/h_bench.err
  python - <<PY >> gpurun_out/h_bench.log
import json
d=json.load(open("gpurun_out/h_bench_$v.json"))
r=d["roofline"]
print("SZN_IGEMM_8PH=$v", d["value"], d["ms_per_step"], r["kernel"], r["frac"], r.get("step_mfma_frac"))
for row in d["kernels"]["rows"]:
    if row.get("bound")=="mfma" and ("igemm_v2" in row["kernel"] or "n128" in row["kernel"] or "wide" in row["kernel"]):
        print("   ", row["kernel"][:50], row["entry"], row["calls_per_step"], row["ms_per_step"], row["frac"])
PY
done
cat gpurun_out/h_bench.log
# fp32 per-kernel table (which kernel families of the reference-arithmetic step are furthest from the 157.3 TF roof)
python bench.py --precision fp32 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/h_fp32.json 2>> gpurun_out/h_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/h_fp32.json"))
print("fp32", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
for row in d["kernels"]["rows"][:28]:
    print("   %-52s %-28s %5.1f %8.3f ms  %s" % (row["kernel"][:52], row["entry"], row["calls_per_step"], row["ms_per_step"], row.get("frac")))
PY
