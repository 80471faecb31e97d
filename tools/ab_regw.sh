# conv3x3_regw: clock64 split of the tile loop (make ABLATE=1 build in lib_ablate/; SZN_REGW_ABLATE bit 8 = probe, 2 = no stores, 4 = no patch DMA)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/regw; mkdir -p $OUT; rm -f $OUT/probe.txt
python -m pytest tests/test_gpu_conv.py -q -x 2>&1 | tail -2
export SZN_LIB_PATH=$GRAFT_REPO_ROOT/zeroshotsemanticsegmentation_amd/lib_ablate/libszn_hip.so
for sp in 0 1; do
  echo "== SZN_REGW_ABLATE=8 SZN_REGW_SPREAD=$sp" >> $OUT/probe.txt
  SZN_REGW_SPREAD=$sp SZN_REGW_ABLATE=8 python tools/probe_regw_cycles.py 2>&1 | grep -v amdgpu.ids >> $OUT/probe.txt
done
unset SZN_LIB_PATH
cat $OUT/probe.txt
for rep in 1 2; do for sp in 0 1; do
  echo "== spread=$sp rep $rep"
  SZN_REGW_SPREAD=$sp python tools/bench_conv.py --layers conv1_2,conv2_1,conv2_2 --what fwd,dgrad --iters 20 2>/dev/null
done; done
