# round 6: tools/r06_profile.sh <tag>  ->  gpurun_out/r6prof_<tag>/: kernel stats + time line of the headline step and of a B = 1 step,
# the stall-breakdown PMC pass, MFMA utilisation, and the forced one-rank RCCL exchange time line (1b)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-a}
O=$R/gpurun_out/r6prof_$T
mkdir -p $O
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-events"
rocprofv3 --kernel-trace --stats -d $O/p1 -o p1 -- python $R/bench.py $ARGS > $O/p1.log 2>&1
DB=$(find $O/p1 -name "*_results.db" | head -1)
python $R/tools/prof_summary.py $DB $O/r06_${T}_bf16_b8_k59_kernel_stats.md "round 6 state $T: python bench.py $ARGS (bf16, B=8, K=59; 7 steps traced)"
python $R/tools/prof_timeline.py $DB $O/r06_${T}_timeline.md > /dev/null
rm -rf $O/p1
rocprofv3 --kernel-trace --stats -d $O/p2 -o p2 -- python $R/bench.py --batch 1 $ARGS > $O/p2.log 2>&1
DB=$(find $O/p2 -name "*_results.db" | head -1)
python $R/tools/prof_summary.py $DB $O/r06_${T}_bf16_b1_kernel_stats.md "round 6 state $T, B = 1 (the reference's batch size): python bench.py --batch 1 $ARGS (7 steps traced)"
python $R/tools/prof_timeline.py $DB $O/r06_${T}_b1_timeline.md > /dev/null
rm -rf $O/p2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $O/raw -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-events > $O/pmc.log 2>&1
python $R/tools/pmc_stalls.py $O/raw $O/r06_${T}_stalls.md "round 6 state $T: stall breakdown per kernel of the bench step (bf16, B=8, K=59; 3 steps)" > /dev/null
rm -rf $O/raw
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $O/raw2 -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-events > $O/pmc2.log 2>&1
python $R/tools/pmc_mfma_step.py $O/raw2 $O/r06_${T}_mfma_util.md > /dev/null
rm -rf $O/raw2
bash $R/tools/pmc_bench.sh $O/r06_${T}_traffic.json > $O/pmc3.log 2>&1
rm -rf $R/gpurun_out/pmc_bench_fetch $R/gpurun_out/pmc_bench_write
rocprofv3 --kernel-trace --stats -d $O/p3 -o p3 -- python $R/bench.py --sub-record comm --batch 8 --size 512 --embed-dim 300 --classes 59 --steps 6 > $O/p3.log 2>&1
DB=$(find $O/p3 -name "*_results.db" | head -1)
python $R/tools/prof_comm_overlap.py $DB $O/r06_${T}_comm_overlap.md > /dev/null
rm -rf $O/p3
head -14 $O/r06_${T}_bf16_b8_k59_kernel_stats.md; head -24 $O/r06_${T}_bf16_b1_kernel_stats.md; head -20 $O/r06_${T}_stalls.md; head -12 $O/r06_${T}_mfma_util.md; tail -3 $O/pmc3.log; head -16 $O/r06_${T}_comm_overlap.md
