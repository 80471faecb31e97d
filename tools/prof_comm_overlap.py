#!/usr/bin/env python
"""Which compute kernels ran WHILE librccl's kernels were running?  From a rocprofv3 (rocpd sqlite) kernel trace of
`bench.py --sub-record comm` (the forced one-rank gradient exchange): every RCCL dispatch of the last traced step of the first
forced variant with its duration and the kernels whose execution overlapped it.
usage: tools/prof_comm_overlap.py <results.db> <out.md>"""
import re
import sqlite3
import sys


def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*?>)?)', n)
    return (m.group(1) if m else n)[:70]


def main():
    db, out = sys.argv[1:3]
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    is_comm = lambda n: re.search(r"nccl|rccl|oneRank", n, re.I) is not None
    comm = [r for r in rows if is_comm(r[0])]
    if not comm:
        open(out, "w").write("# no RCCL kernels in the trace\n")
        print("no RCCL kernels found; kernel names:", sorted({short(r[0]) for r in rows})[:40])
        return
    # group RCCL dispatches into steps: a gap of more than 3 ms between consecutive RCCL kernels starts a new step
    steps, curstep = [], [comm[0]]
    for r in comm[1:]:
        if r[1] - curstep[-1][2] > 3e6:
            steps.append(curstep)
            curstep = []
        curstep.append(r)
    steps.append(curstep)
    full = [s for s in steps if len(s) == max(len(x) for x in steps)]
    st = full[len(full) // 2]                                  # a steady-state step
    t0 = st[0][1]
    with open(out, "w") as f:
        f.write("# RCCL kernels of one forced one-rank gradient exchange and the compute kernels they overlapped\n\n")
        f.write("rocprofv3 --kernel-trace of `bench.py --sub-record comm` (B = 8, 512 x 512, bf16; engine.GradBuckets(force=True): the step's real "
                "buckets through librccl on ProcessGroupNCCL's stream).  %d RCCL dispatches per step (%d steps traced); times relative to the first.\n\n"
                % (len(st), len(steps)))
        f.write("| RCCL kernel | start us | duration us | compute kernels running at the same time (overlap us) |\n|---|---|---|---|\n")
        tot_c, tot_o = 0.0, 0.0
        for n, s, e in st:
            ov = {}
            for n2, s2, e2 in rows:
                if e2 <= s or s2 >= e or is_comm(n2):
                    continue
                ov[short(n2)] = ov.get(short(n2), 0.0) + (min(e, e2) - max(s, s2)) / 1e3
            tot_c += (e - s) / 1e3
            # union of overlapped time
            iv = sorted((max(s, s2), min(e, e2)) for n2, s2, e2 in rows if not (e2 <= s or s2 >= e or is_comm(n2)))
            u, last = 0.0, s
            for a, b in iv:
                a = max(a, last)
                if b > a:
                    u += b - a
                    last = b
            tot_o += u / 1e3
            f.write("| %s | %.0f | %.0f | %s |\n" % (short(n), (s - t0) / 1e3, (e - s) / 1e3,
                                               ", ".join("%s %.0f" % kv for kv in sorted(ov.items(), key=lambda kv: -kv[1])[:6]) or "(none)"))
        f.write("\nRCCL busy %.0f us per step, %.0f us of it (%.0f %%) with a compute kernel running beside it.\n" % (tot_c, tot_o, 100 * tot_o / max(tot_c, 1e-9)))
    print(open(out).read())


if __name__ == "__main__":
    main()
