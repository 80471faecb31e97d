# A/B of two builds of the library on one box: tools/ab_lib.sh <tag> <layers> <what> [bench]
#   arm A = zeroshotsemanticsegmentation_amd/lib_prev/libszn_hip.so (an older build), arm B = the in-tree lib/
cd $GRAFT_REPO_ROOT
TAG=$1; LAYERS=$2; WHAT=$3
OUT=gpurun_out/$TAG
mkdir -p $OUT
PREV=$GRAFT_REPO_ROOT/zeroshotsemanticsegmentation_amd/lib_prev/libszn_hip.so
for rep in 1 2; do
for v in A B; do
  if [ $v = A ]; then export SZN_LIB_PATH=$PREV; else unset SZN_LIB_PATH; fi
  echo "== lib=$v rep $rep" >> $OUT/conv.log
  python tools/bench_conv.py --layers $LAYERS --what $WHAT --iters 20 2>/dev/null >> $OUT/conv.log
done; done
if [ "$4" = "bench" ]; then
for v in A B A B; do
  if [ $v = A ]; then export SZN_LIB_PATH=$PREV; else unset SZN_LIB_PATH; fi
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_$v.json 2>> $OUT/bench.err
  python - <<PY >> $OUT/bench.log
import json
d=json.load(open("$OUT/bench_$v.json"))
r=d["roofline"]
print("lib=$v", d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["conv_fwd_dgrad_family"]["frac"], r.get("step_mfma_frac"))
PY
done
fi
unset SZN_LIB_PATH
cat $OUT/conv.log $OUT/bench.log 2>/dev/null
