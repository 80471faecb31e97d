#!/usr/bin/env python
"""Time line of ONE train step from a rocprofv3 (rocpd sqlite) kernel trace: every dispatch in start order with the idle gap in
front of it, the busy / idle totals of the step and the kernels launched by torch itself (at::native / rocclr) with their neighbours.
usage: tools/prof_timeline.py <results.db> <out.md> [anchor-kernel-substring = conv1_1_fwd]"""
import re
import sqlite3
import sys


def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*?>)?)', n)
    return (m.group(1) if m else n)[:80]


def main():
    db, out = sys.argv[1:3]
    anchor = sys.argv[3] if len(sys.argv) > 3 else "conv1_1_fwd"
    con = sqlite3.connect(db)
    cur = con.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    rows = None
    for view in ("kernels",) + tuple(n for n in names if n.startswith("rocpd_kernel_dispatch")):
        if view not in names:
            continue
        cols = [r[1] for r in cur.execute("pragma table_info(%s)" % view)]
        if view == "kernels" and {"name", "start", "end"} <= set(cols):
            rows = list(cur.execute("select name, start, end from kernels order by start"))
            break
    if rows is None:
        print("no usable view; tables:", names)
        for n in names:
            print(n, [r[1] for r in cur.execute("pragma table_info(%s)" % n)])
        sys.exit(1)
    starts = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(starts) < 3:
        print("anchor %s seen %d times" % (anchor, len(starts)))
        sys.exit(1)
    a, b = starts[-2], starts[-1]                      # the last complete step
    step = rows[a:b]
    t0, t1 = step[0][1], rows[b][1]
    busy = sum(e - s for _, s, e in step)
    with open(out, "w") as f:
        f.write("# time line of one step (%d dispatches, %.3f ms wall, %.3f ms busy, %.3f ms idle between kernels)\n\n"
                % (len(step), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
        tor = [(n, e - s) for n, s, e in step if ("at::native" in n or "rocclr" in n)]
        f.write("kernels launched by torch / the runtime: %d, %.3f ms\n\n" % (len(tor), sum(d for _, d in tor) / 1e6))
        f.write("| # | kernel | us | gap before (us) |\n|---|---|---|---|\n")
        prev_end = None
        for i, (n, s, e) in enumerate(step):
            gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
            f.write("| %d | %s | %.1f | %.1f |\n" % (i, short(n), (e - s) / 1e3, gap))
            prev_end = max(e, prev_end or e)
    print(open(out).read()[:400])


if __name__ == "__main__":
    main()
