#!/bin/bash
# round 5, GPU call K: (1) B = 1 with the relaxed pixel-split rule of conv_wgrad_taps; (2) the clock the fp32 kernels run at
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_conv.py -q -x -k "wgrad or conv_fwd_dgrad" > gpurun_out/k_tests.log 2>&1; tail -2 gpurun_out/k_tests.log
: > gpurun_out/k_b1.log
for v in 0 1 0 1; do
  SZN_WGT_SMALLSPLIT=$v python bench.py --sub-record b1 --steps 20 2>/dev/null | grep SUBRECORD | python -c "
import json,sys
d=json.loads(sys.stdin.read()[len('SUBRECORD '):])
print('B=1 smallsplit=$v bf16 eager %.3f graph %.3f' % (d['bf16']['eager_ms_per_step'], d['bf16']['ms_per_step']))" >> gpurun_out/k_b1.log
done
cat gpurun_out/k_b1.log
cd /tmp
O=$GRAFT_REPO_ROOT/gpurun_out/k_fp32
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $O/raw2 -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --precision fp32 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-events > $O/pmc2.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_mfma_step.py $O/raw2 $O/r05_fp32_mfma_util.md > /dev/null
rm -rf $O/raw2
head -24 $O/r05_fp32_mfma_util.md
