# A/B of one environment switch on one box: tools/ab_env.sh <tag> <VAR> <layers> <what> [bench]
# conv micro-benchmark with VAR=0 / default interleaved twice, then (optional) the bench step twice each
cd $GRAFT_REPO_ROOT
TAG=$1; VAR=$2; LAYERS=$3; WHAT=$4; A=${6:-0}; B=${7:-}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for rep in 1 2; do
for v in A B; do
  if [ $v = A ]; then export $VAR=$A; else if [ -n "$B" ]; then export $VAR=$B; else unset $VAR; fi; fi
  echo "== $VAR=${!VAR:-default} rep $rep" >> $OUT/conv.log
  python tools/bench_conv.py --layers $LAYERS --what $WHAT --iters 20 2>/dev/null >> $OUT/conv.log
done; done
if [ "$5" = "bench" ]; then
for v in A B A B; do
  if [ $v = A ]; then export $VAR=$A; else if [ -n "$B" ]; then export $VAR=$B; else unset $VAR; fi; fi
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_$v.json 2>> $OUT/bench.err
  python - <<PY >> $OUT/bench.log
import json
d=json.load(open("$OUT/bench_$v.json"))
r=d["roofline"]
print("$VAR=${!VAR:-default}", d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["conv_fwd_dgrad_family"]["frac"], r.get("step_mfma_frac"))
PY
done
fi
unset $VAR
cat $OUT/conv.log $OUT/bench.log 2>/dev/null
