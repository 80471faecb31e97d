cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04p; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/p2 -o p2 -- python $R/bench.py --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-events > $O/p2.log 2>&1
DB=$(find $O/p2 -name "*_results.db" | head -1)
python $R/tools/prof_timeline.py $DB $O/b1_timeline.md > /dev/null
rm -rf $O/p2
cat $O/b1_timeline.md
