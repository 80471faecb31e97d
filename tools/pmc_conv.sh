# PMC passes over the conv micro-benchmark (separate passes: counter-slot limits; never combined with tracing domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-v5}
LAYERS=${2:-conv4_2,conv1_2,conv3_2}
WHAT=${3:-fwd,wgrad}
i=0
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_TA_BUSY_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass -d $R/gpurun_out/pmc_${TAG}_$i -o pmc --output-format csv -- python $R/tools/bench_conv.py --layers $LAYERS --what $WHAT --iters 2 > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1
done
ls $R/gpurun_out/ | grep pmc_${TAG}
