cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_band_crop.py tests/test_gpu_model.py tests/test_gpu_headline_pin.py tests/test_gpu_fcn8s.py -q -x 2>&1 | tail -4
mkdir -p gpurun_out/ab_gather
for rep in 1 2; do for v in 0 1; do
  SZN_POOL_GATHER=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/ab_gather/bench_$v.json 2>> gpurun_out/ab_gather/bench.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_gather/bench_$v.json"))
r=d["roofline"]
print("pool_gather=$v", d["value"], d["ms_per_step"], r["kernel"], r["frac"], r.get("step_mfma_frac"))
PY
done; done
SZN_POOL_GATHER=1 python bench.py --sub-record fp32 2>/dev/null | grep SUBRECORD | cut -c1-120
SZN_POOL_GATHER=0 python bench.py --sub-record fp32 2>/dev/null | grep SUBRECORD | cut -c1-120
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_gather
rocprofv3 --kernel-trace --stats -d $O/p1 -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-events > $O/p1.log 2>&1
DB=$(find $O/p1 -name "*_results.db" | head -1)
python $R/tools/prof_timeline.py $DB $O/timeline_1.md > /dev/null
rm -rf $O/p1
head -1 $O/timeline_1.md; grep -n "maxpool_bwd\|band_" $O/timeline_1.md
