#!/usr/bin/env python
"""MFMA utilisation per kernel of the bench step from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_MFMA, GRBM_GUI_ACTIVE):
utilisation = MFMA-busy cycles / (1024 SIMDs x launch duration x 2.4 GHz) -- the same formula as tools/pmc_proj.py -- and the clock the
launch actually ran at (GRBM_GUI_ACTIVE / duration).  usage: tools/pmc_mfma_step.py <pmc dir> <out.md>"""
import csv, re, sys
from collections import defaultdict
csv.field_size_limit(1 << 30)


def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*?>)?)', n)
    return (m.group(1) if m else n)[:70]


def main():
    d, out = sys.argv[1:3]
    rows = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    import glob
    for f in glob.glob(d.rstrip('/') + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            rows[k][r['Counter_Name']] += float(r['Counter_Value'])
            key = (r.get('Dispatch_Id'), r['Start_Timestamp'])
            if key not in cnt[k]:
                cnt[k].add(key)
                rows[k]['_us'] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tab = []
    for k, c in rows.items():
        if c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) <= 0:
            continue
        us = c['_us']
        util = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * us * 2400.0)
        clk = c.get('GRBM_GUI_ACTIVE', 0) / 8.0 / us / 1e3 if us else 0      # the counter is summed over the 8 XCDs
        tab.append((us, k, len(cnt[k]), util, clk, c.get('SQ_INSTS_MFMA', 0)))
    tab.sort(reverse=True)
    with open(out, 'w') as f:
        f.write("# MFMA utilisation per kernel (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE on the bench step)\n\n")
        f.write("utilisation = MFMA-busy cycles / (1024 SIMDs x launch time x 2.4 GHz); clock = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / launch time (profiled passes run ~5 % below un-profiled ones)\n\n")
        f.write("| kernel | launches | total us | MFMA utilisation | clock GHz |\n|---|---|---|---|---|\n")
        for us, k, n, util, clk, _ in tab:
            f.write("| %s | %d | %.0f | %.3f | %.2f |\n" % (k, n, us, util, clk))
    print(open(out).read())


if __name__ == "__main__":
    main()
