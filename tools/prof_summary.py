#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a markdown table for profiles/.
usage: tools/prof_summary.py <results.db> <out.md> "<title / command line>" """
import re
import sqlite3
import sys


def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*?>)?)', n)
    return (m.group(1) if m else n)[:90]


def main():
    db, out, title = sys.argv[1:4]
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(out, 'w') as f:
        f.write("# %s\n\nrocprofv3 --kernel-trace --stats; durations in microseconds (top_kernels view of %s).\n\n" % (title, db))
        f.write("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|\n")
        for n, c, t, a, p in rows:
            f.write("| %s | %d | %.0f | %.1f | %.2f |\n" % (short(n), c, t, a, p))


if __name__ == "__main__":
    main()
