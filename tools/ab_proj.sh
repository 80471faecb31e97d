# projection-kernel experiments on one box: tools/ab_proj.sh <tag>   (nominal 262,144 x 4096 x 300 shape, interleaved arms, 2 reps)
cd $GRAFT_REPO_ROOT
TAG=${1:-abproj}
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() {  # label, env assignments...
  local label=$1; shift
  ( for kv in "$@"; do export $kv; done
    python tools/bench_proj.py --nominal-only --iters 20 2>>$OUT/err.log | python -c "
import json,sys
d=json.load(sys.stdin)
for r in d['nominal_shape']: print('$label', r['ms'], r['TF'], r['frac'], r['activation_stream_TBps'], r['kernel'])
" >> $OUT/proj.log )
}
for rep in 1 2; do
  run base
  run rot1 SZN_PROJ_ROT=1
  run rot7 SZN_PROJ_ROT=7
  run rot13_pf0 SZN_PROJ_ROT=13
  run pf1 SZN_PROJ_PF=1
  run pf2 SZN_PROJ_PF=2
  run pf3 SZN_PROJ_PF=3
  run pf4 SZN_PROJ_PF=4
  run rot7_pf2 SZN_PROJ_ROT=7 SZN_PROJ_PF=2
  run nt0 SZN_PROJ_NT=0
  run nt0_pf2 SZN_PROJ_NT=0 SZN_PROJ_PF=2
  run abl_samerows SZN_LIB_PATH=$GRAFT_REPO_ROOT/zeroshotsemanticsegmentation_amd/lib_ablate/libszn_hip.so SZN_PROJ_ABLATE=1
done
cat $OUT/proj.log; tail -5 $OUT/err.log
