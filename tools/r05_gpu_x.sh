#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_headline_pin.py -q -s 2>&1 | grep -v "Gloo\|amdgpu" | tail -44
