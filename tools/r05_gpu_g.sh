#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
ABL=$GRAFT_REPO_ROOT/zeroshotsemanticsegmentation_amd/lib_ablate/libszn_hip.so
cd tools
SZN_LIB_PATH=$ABL SZN_WGT_ABLATE=9 python probe_wgt_cycles.py > ../gpurun_out/g_cycles.log 2>&1
cat ../gpurun_out/g_cycles.log | grep -v amdgpu.ids
