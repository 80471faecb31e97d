#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output (pmc_counter_collection.csv) for this library's kernels.
usage: tools/pmc_summary.py <dir> [<dir> ...]   -> table: kernel, grid, counter averages"""
import csv
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*?>)?)', n)
    return (m.group(1) if m else n)[:60]


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for d in sys.argv[1:]:
        with open(d.rstrip('/') + '/pmc_counter_collection.csv') as f:
            for row in csv.DictReader(f):
                k = short(row['Kernel_Name'])
                if not any(t in k for t in ('conv_', 'conv3x3', 'conv1_1', 'col2im', 'bias_grad', 'maxpool', 'adam', 'fh_', 'im2col', 'pack_dgrad')):
                    continue
                key = (k, int(row['Grid_Size']))
                acc[key][row['Counter_Name']].append(float(row['Counter_Value']))
                acc[key]['_dur_us'].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
    for key in sorted(acc):
        print("%s grid=%d" % key)
        for c in sorted(acc[key]):
            v = acc[key][c]
            print("    %-28s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main()
