# round-3 state e (final): kernel stats + time line of the headline step, phase-2 kernel stats, PMC traffic passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3prof_f
mkdir -p $O
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-events"
rocprofv3 --kernel-trace --stats -d $O/p1 -o p1 -- python $R/bench.py $ARGS > $O/p1.log 2>&1
DB=$(find $O/p1 -name "*_results.db" | head -1)
python $R/tools/prof_summary.py $DB $O/r03_f_bf16_b8_k59_kernel_stats.md "round 3 state f = final (state e + parallel taps reduce, XCD-aware tile order of the fc layers): python bench.py $ARGS (bf16, B=8, K=59; 7 steps traced)"
python $R/tools/prof_timeline.py $DB $O/r03_f_timeline.md > /dev/null
rm -rf $O/p1
rocprofv3 --kernel-trace --stats -d $O/p2 -o p2 -- python $R/bench.py --phase seenmask $ARGS > $O/p2.log 2>&1
DB=$(find $O/p2 -name "*_results.db" | head -1)
python $R/tools/prof_summary.py $DB $O/r03_f_phase2_kernel_stats.md "round 3 state f, phase 2 (engine.SeenmaskStep): python bench.py --phase seenmask $ARGS (bf16, B=8, K=59; 7 steps traced)"
rm -rf $O/p2
bash $R/tools/pmc_bench.sh $O/r03_f_traffic.json > $O/pmc.log 2>&1
rm -rf $R/gpurun_out/pmc_bench_fetch $R/gpurun_out/pmc_bench_write
head -12 $O/r03_f_bf16_b8_k59_kernel_stats.md; head -3 $O/r03_f_timeline.md; tail -5 $O/pmc.log
