# Evidence for the pixel-projection GEMM (north star: "rocprof HBM GB/s and MFMA utilisation against gfx950 peak"): the nominal
# 262,144 x 4096 x 300 shape through szn_conv2d_fwd -> proj_gemm_stream.  One --kernel-trace --stats pass, then separate --pmc
# passes (counter-slot limits; never combined with other tracing domains), then tools/pmc_proj.py -> gpurun_out/r02_proj.json.
# usage (GPU box): bash tools/pmc_proj.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_proj
mkdir -p $O
python $R/tools/bench_proj.py --iters 10 > $O/bench_proj.json 2> $O/bench_proj.err
rocprofv3 --kernel-trace --stats -d $O/stats -o proj --output-format csv -- python $R/tools/bench_proj.py --nominal-only --iters 10 > $O/stats.log 2>&1
i=0
for pass in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass -d $O/pmc_$i -o pmc --output-format csv -- python $R/tools/bench_proj.py --nominal-only --iters 5 > $O/pmc_$i.log 2>&1
done
python $R/tools/pmc_proj.py $O $R/gpurun_out/${1:-r04}_proj.json
