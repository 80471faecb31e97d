# kernel trace of the headline step -> per-kernel table + the time line of one step (gaps, torch-launched kernels)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-tl}
O=$R/gpurun_out/$TAG
mkdir -p $O
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-events"
rocprofv3 --kernel-trace --stats -d $O/p1 -o p1 -- python $R/bench.py $ARGS > $O/p1.log 2>&1
DB=$(find $O/p1 -name "*_results.db" | head -1)
python $R/tools/prof_summary.py $DB $O/kernel_stats.md "python bench.py $ARGS (bf16, B=8, K=59; 7 steps traced)"
python $R/tools/prof_timeline.py $DB $O/timeline.md
rm -rf $O/p1
tail -2 $O/p1.log | cut -c1-300
