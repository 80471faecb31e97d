# HBM traffic of the bench step from the PMC counters: two separate rocprofv3 --pmc passes (FETCH_SIZE uses 3 of the 4
# TCC slots, WRITE_SIZE 2), kernel-trace only (never combined with other tracing domains), then the per-kernel summary.
# usage (GPU box): bash tools/pmc_bench.sh [out.json]   -> gpurun_out/pmc_bench_{fetch,write}/ + the json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=${1:-$R/gpurun_out/r02_traffic.json}
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-events"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_bench_fetch -o pmc --output-format csv -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_bench_write -o pmc --output-format csv -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_bench_write.log 2>&1
python $R/tools/pmc_traffic.py $R/gpurun_out/pmc_bench_fetch $R/gpurun_out/pmc_bench_write $OUT 8 bf16 512 59
