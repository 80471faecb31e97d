# conv3x3_regw: clock64 split of the tile loop (make ABLATE=1 build in lib_ablate/; SZN_REGW_ABLATE bit 8 = probe, 2 = no stores, 4 = no patch DMA)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/regw; mkdir -p $OUT; rm -f $OUT/probe.txt
export SZN_LIB_PATH=$GRAFT_REPO_ROOT/zeroshotsemanticsegmentation_amd/lib_ablate/libszn_hip.so
for ab in 8 14 10 12; do
  echo "== SZN_REGW_ABLATE=$ab" >> $OUT/probe.txt
  SZN_REGW_ABLATE=$ab python tools/probe_regw_cycles.py 2>&1 | grep -v amdgpu.ids >> $OUT/probe.txt
done
echo "== SZN_REGW_ABLATE=8 SZN_REGW_SHIFT=0" >> $OUT/probe.txt
SZN_REGW_SHIFT=0 SZN_REGW_ABLATE=8 python tools/probe_regw_cycles.py conv1_2 conv2_1 2>&1 | grep -v amdgpu.ids >> $OUT/probe.txt
cat $OUT/probe.txt
