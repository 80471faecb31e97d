# same-box A/B of conv_igemm_8ph builds: A = lib_prev (an earlier build), B = in-tree with SZN_8PH_MODE=0, C = in-tree with SZN_8PH_MODE=2
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r04_ab8ph}; mkdir -p $OUT
PREV=$GRAFT_REPO_ROOT/zeroshotsemanticsegmentation_amd/lib_prev/libszn_hip.so
for rep in 1 2; do
for v in A B C; do
  unset SZN_LIB_PATH SZN_8PH_MODE
  if [ $v = A ]; then export SZN_LIB_PATH=$PREV; fi
  if [ $v = B ]; then export SZN_8PH_MODE=0; fi
  if [ $v = C ]; then export SZN_8PH_MODE=2; fi
  echo "== $v rep $rep" >> $OUT/conv.log
  python tools/bench_conv.py --layers conv3_1,conv3_2,conv4_1,conv4_2,fc6 --what fwd,dgrad --iters 20 2>/dev/null >> $OUT/conv.log
done; done
for v in A B C A B C; do
  unset SZN_LIB_PATH SZN_8PH_MODE
  if [ $v = A ]; then export SZN_LIB_PATH=$PREV; fi
  if [ $v = B ]; then export SZN_8PH_MODE=0; fi
  if [ $v = C ]; then export SZN_8PH_MODE=2; fi
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['conv_fwd_dgrad_family']['frac'])" >> $OUT/bench.log
done
cat $OUT/conv.log $OUT/bench.log
