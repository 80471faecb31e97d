cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04n; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/p1 -o p1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-events > $O/p1.log 2>&1
DB=$(find $O/p1 -name "*_results.db" | head -1)
python $R/tools/prof_timeline.py $DB $O/timeline_fused.md > /dev/null
rm -rf $O/p1
grep -n "wgrad_wide\|adam\|col2im" $O/timeline_fused.md | head; head -3 $O/timeline_fused.md
