# epilogue accounting of the 256-wide tile kernels (ablation build: WRONG results): full / no global stores + gate loads / no epilogue
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-epabl}
mkdir -p $OUT
export SZN_LIB_PATH=$GRAFT_REPO_ROOT/zeroshotsemanticsegmentation_amd/lib_ablate/libszn_hip.so
for rep in 1 2; do
for v in 0 1 2; do
  echo "== SZN_WIDE_EPABL=$v rep $rep" >> $OUT/conv.log
  SZN_WIDE_EPABL=$v python tools/bench_conv.py --layers conv3_1,conv3_2,conv4_1,conv4_2,fc7 --what fwd,dgrad --iters 20 2>/dev/null >> $OUT/conv.log
done; done
cat $OUT/conv.log
