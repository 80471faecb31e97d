# HBM traffic of the bench step from the PMC counters: two separate rocprofv3 --pmc passes (FETCH_SIZE uses 3 of the 4
# TCC slots, WRITE_SIZE 2), kernel-trace only (never combined with other tracing domains).
# usage (GPU box): bash tools/pmc_bench.sh [batch]   -> gpurun_out/pmc_bench_{fetch,write}/pmc_counter_collection.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-8}
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_bench_fetch -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline --no-kernel-events > $R/gpurun_out/pmc_bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_bench_write -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline --no-kernel-events > $R/gpurun_out/pmc_bench_write.log 2>&1
ls $R/gpurun_out/pmc_bench_fetch $R/gpurun_out/pmc_bench_write
