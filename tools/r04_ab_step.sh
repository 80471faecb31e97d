# same-box A/B of one environment switch on the whole step only: tools/r04_ab_step.sh <tag> <VAR> [b1]   (VAR=0 vs VAR unset, interleaved)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; VAR=$2; mkdir -p $OUT; rm -f $OUT/bench.log
for v in 0 1 0 1 0 1; do
  if [ $v = 0 ]; then export $VAR=0; else unset $VAR; fi
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$VAR=$v', d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['step_mfma_frac'])" >> $OUT/bench.log
  if [ "$3" = "b1" ]; then python bench.py --sub-record b1 --batch 8 --size 512 --embed-dim 300 --classes 59 --steps 20 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('SUBRECORD '):
        d=json.loads(l[10:]); b=d['bf16']; print('$VAR=$v b1', b['eager_ms_per_step'], b.get('ms_per_step'))" >> $OUT/bench.log; fi
done
unset $VAR
cat $OUT/bench.log
