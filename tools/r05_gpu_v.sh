#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_band_crop.py tests/test_gpu_sizes.py tests/test_gpu_seenmask_step.py -q 2>&1 | grep -v "Gloo\|amdgpu" | tail -5
