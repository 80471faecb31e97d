cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_conv.py -q -x 2>&1 | tail -2
bash tools/ab_lib.sh ab_regw conv1_2,conv2_1,conv2_2 fwd,dgrad $1
