# conv3x3_regw phase accounting (make ABLATE=1 build in lib_ablate/: SZN_REGW_ABLATE bits 1 = no MFMA phase, 2 = no global stores, 4 = no patch DMA)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/regw; mkdir -p $OUT; rm -f $OUT/log.txt
python -m pytest tests/test_gpu_fused_adam.py -q -x 2>&1 | tail -3 > $OUT/tests.log
cat $OUT/tests.log
export SZN_LIB_PATH=$GRAFT_REPO_ROOT/zeroshotsemanticsegmentation_amd/lib_ablate/libszn_hip.so
for ab in 0 1 2 4 6 5 7; do
  echo "== SZN_REGW_ABLATE=$ab" >> $OUT/log.txt
  SZN_REGW_ABLATE=$ab python tools/bench_conv.py --layers conv1_2,conv2_1,conv2_2 --what fwd,dgrad --iters 20 2>/dev/null >> $OUT/log.txt
done
echo "== zeros, ablate 0" >> $OUT/log.txt
SZN_REGW_ABLATE=0 python tools/bench_conv.py --layers conv1_2,conv2_1,conv2_2 --what fwd,dgrad --iters 20 --zeros 2>/dev/null >> $OUT/log.txt
cat $OUT/log.txt
