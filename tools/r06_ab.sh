#!/bin/bash
# same-box alternating A/B of environment settings: tools/r06_ab.sh <out> <rounds> "<env A>" "<env B>" ... -- runs bench.py --no-extras per setting
out=$1; rounds=$2; shift 2
mkdir -p $(dirname $out)
for r in $(seq 1 $rounds); do
  for e in "$@"; do
    v=$(env $e python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "round $r  [$e]  $v ms" >> $out
  done
done
