for i in 1 2 3 4 5 6; do
python bench.py --phase seenmask --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | cut -c40-150
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['phase2'])"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['phase2'])"
