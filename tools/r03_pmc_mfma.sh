# MFMA utilisation per kernel of the bench step, one --pmc pass: tools/r03_pmc_mfma.sh <tag>  ->  gpurun_out/pmc_mfma_<tag>/r03_<tag>_mfma_util.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-f}; O=$R/gpurun_out/pmc_mfma_$T; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $O/raw -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-events > $O/log 2>&1
python $R/tools/pmc_mfma_step.py $O/raw $O/r03_${T}_mfma_util.md
rm -rf $O/raw
head -16 $O/r03_${T}_mfma_util.md
