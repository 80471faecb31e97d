python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 3
python bench.py --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['phase2'], d['cpu_baseline']['value'])"
