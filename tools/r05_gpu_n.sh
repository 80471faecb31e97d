#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/n_fp32
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $O/raw -o pmc --output-format csv -- python $R/bench.py --precision fp32 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-events > $O/pmc.log 2>&1
python $R/tools/pmc_stalls.py $O/raw $O/r05_fp32_stalls.md "round 5: stall breakdown per kernel of the fp32 bench step (B=8, K=59; 3 steps)" > /dev/null
rm -rf $O/raw
head -16 $O/r05_fp32_stalls.md | cut -c1-170
rocprofv3 --kernel-trace --stats -d $O/p1 -o p1 -- python $R/bench.py --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-events > $O/p1.log 2>&1
DB=$(find $O/p1 -name "*_results.db" | head -1)
python $R/tools/prof_timeline.py $DB $O/r05_fp32_timeline.md > /dev/null
rm -rf $O/p1
head -120 $O/r05_fp32_timeline.md | cut -c1-120
