# round-3 profiles on one box: rocprofv3 kernel stats of the headline step and of the phase-2 step, then the PMC traffic passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3prof
mkdir -p $O
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-events"
rocprofv3 --kernel-trace --stats -d $O/p1 -o p1 -- python $R/bench.py $ARGS > $O/p1.log 2>&1
DB=$(find $O/p1 -name "*_results.db" | head -1)
python $R/tools/prof_summary.py $DB $O/r03_a_bf16_b8_k59_kernel_stats.md "round 3 state a: python bench.py $ARGS (bf16, B=8, K=59; 7 steps traced)"
rocprofv3 --kernel-trace --stats -d $O/p2 -o p2 -- python $R/bench.py --phase seenmask $ARGS > $O/p2.log 2>&1
DB=$(find $O/p2 -name "*_results.db" | head -1)
python $R/tools/prof_summary.py $DB $O/r03_a_phase2_kernel_stats.md "round 3 phase 2 (engine.SeenmaskStep): python bench.py --phase seenmask $ARGS (bf16, B=8, K=59; 7 steps traced)"
bash $R/tools/pmc_bench.sh $O/r03_a_traffic.json > $O/pmc.log 2>&1
tail -3 $O/p1.log $O/p2.log; head -30 $O/r03_a_phase2_kernel_stats.md; tail -15 $O/pmc.log
