# A/B of the phase-shifted wave groups (SZN_WIDE_SHIFT) on one box: conv micro-benchmark, then the bench step
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-ab}
mkdir -p $OUT
python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -5 > $OUT/tests.log
python -m pytest tests/test_gpu_parity_full.py -x -q -k "bf16_layer or fullsize_train_step" 2>&1 | tail -5 >> $OUT/tests.log
python -m pytest tests/test_gpu_lowprec.py -x -q -k "dynamic" 2>&1 | tail -5 >> $OUT/tests.log
for rep in 1 2; do
for sh in 0 1; do
  echo "== SZN_WIDE_SHIFT=$sh rep $rep" >> $OUT/conv.log
  SZN_WIDE_SHIFT=$sh python tools/bench_conv.py --layers conv3_1,conv3_2,conv4_1,conv4_2,fc6,fc7 --what fwd,dgrad --iters 20 >> $OUT/conv.log 2>&1
done; done
for sh in 0 1 0 1; do
  SZN_WIDE_SHIFT=$sh python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_shift$sh.json 2>> $OUT/bench.err
  python - <<PY >> $OUT/bench.log
import json
d=json.load(open("$OUT/bench_shift$sh.json"))
print("shift $sh", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["conv_fwd_dgrad_family"]["frac"], d["roofline"].get("step_mfma_frac"))
PY
done
cat $OUT/tests.log $OUT/conv.log $OUT/bench.log
