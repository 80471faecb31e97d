#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python tools/diag_band.py 96 2 2>&1 | grep -v amdgpu.ids | head -80
python tools/diag_band.py 512 8 2>&1 | grep -v amdgpu.ids | grep -A8 "mode conv2_1,conv3_1"
