#!/bin/bash
# round 5, GPU call E: where conv_wgrad_taps spends its time (ablation build: 1 = no LDS-DMA in the loop, 2 = no fragment reads / MFMA)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
ABL=$GRAFT_REPO_ROOT/zeroshotsemanticsegmentation_amd/lib_ablate/libszn_hip.so
LAYERS=conv1_2,conv2_2,conv3_2,conv4_2,conv5_1
: > gpurun_out/e_abl.log
for abl in 0 1 2 0 1 2; do
  echo "== SZN_WGT_ABLATE=$abl" >> gpurun_out/e_abl.log
  SZN_LIB_PATH=$ABL SZN_WGT_ABLATE=$abl python tools/bench_conv.py --layers $LAYERS --what wgrad --iters 20 2>/dev/null >> gpurun_out/e_abl.log
done
echo "== zeros" >> gpurun_out/e_abl.log
python tools/bench_conv.py --layers $LAYERS --what wgrad --iters 20 --zeros 2>/dev/null >> gpurun_out/e_abl.log
cat gpurun_out/e_abl.log
