# fc6's weight gradient + Adam on a CU-masked stream in a one-image step: does a profiled run exit cleanly, and the caller-stream forms
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/cumask; mkdir -p $OUT; rm -f $OUT/log.txt
for m in 0 128:low; do for s in default side; do
  SZN_FC6_CUMASK=$m python tools/probe_cumask.py --stream $s 2>$OUT/err.log >> $OUT/log.txt; echo "rc $?" >> $OUT/log.txt
done; done
cat $OUT/log.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT/p2 -o p2 -- python $R/tools/probe_cumask.py --steps 5 > $OUT/p2.log 2>&1
echo "rocprof rc $?"
rm -rf $OUT/p2
grep -c SIGSEGV $OUT/p2.log
