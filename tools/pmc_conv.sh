cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/bench_conv.py > $R/gpurun_out/bench_conv_v1.log 2>&1
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  n=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass -d $R/gpurun_out/pmc_v1_$n -o pmc --output-format csv -- python $R/tools/bench_conv.py --layers conv3_2,conv1_2,fc6 --what fwd,wgrad --iters 2 > $R/gpurun_out/pmc_v1_$n.log 2>&1
done
ls -R $R/gpurun_out/ | head -50
