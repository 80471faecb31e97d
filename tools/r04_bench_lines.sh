# the bench lines of the other BASELINE configurations, final round-4 state -> gpurun_out/r04_bench_lines.json
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_lines; mkdir -p $O
A="--steps 20 --warmup 3 --no-cpu-baseline --no-extras"
python bench.py $A > $O/default_k59_bf16.json 2>/dev/null
python bench.py $A --classes 21 > $O/configs1_k21_bf16.json 2>/dev/null
python bench.py $A --precision fp16 > $O/k59_fp16.json 2>/dev/null
python bench.py $A --size 768 --precision fp16 --head-fp8 > $O/configs4_768_fp16_fp8head.json 2>/dev/null
python bench.py $A --arch fcn8s > $O/fcn8s_k59_bf16.json 2>/dev/null
python bench.py $A --phase seenmask > $O/phase2_k59_bf16.json 2>/dev/null
python bench.py $A --unfused-head > $O/unfused_head_k59_bf16.json 2>/dev/null
python - <<PY
import json, glob, os
out = {}
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        out[os.path.basename(f)[:-5]] = {"error": repr(e)}; continue
    r = d.get("roofline", {})
    out[os.path.basename(f)[:-5]] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"],
                                      "workload": d["config"]["workload"][:120], "dominant": r.get("kernel"), "frac": r.get("frac"),
                                      "family_frac": (r.get("conv_fwd_dgrad_family") or {}).get("frac"), "step_mfma_frac": r.get("step_mfma_frac")}
json.dump(out, open("gpurun_out/r04_bench_lines.json", "w"), indent=1)
for k, v in out.items(): print(k, v.get("value"), v.get("ms_per_step"), v.get("frac"), v.get("step_mfma_frac"))
PY
# ... and the driver-style default line (every sub-record) beside them
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_lines/driver_style_full.json 2> gpurun_out/r04_lines/driver_style_full.err
python - <<PY
import json
d = json.load(open("gpurun_out/r04_lines/driver_style_full.json"))
print("driver-style:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("step_mfma_frac"))
print("projection nominal:", d["projection"]["nominal_shape"]["frac"], d["projection"]["nominal_shape"].get("frac_sustained"), d["projection"]["nominal_shape_randn"]["frac"])
print("phase2", d["phase2"]["ms_per_step"], "fp32", d["fp32"]["ms_per_step"], "b1", d["b1"]["bf16"]["ms_per_step"], "comm", d["comm"]["ms_per_step"])
PY
