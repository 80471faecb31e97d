#!/usr/bin/env python
"""Stall breakdown per kernel of the bench step from one rocprofv3 --pmc pass:
    SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
Per MI355X_MICROARCH.md (rocprofv3 PMC slots): SQ_WAIT_ANY = wave parked at s_waitcnt / s_barrier; SQ_WAIT_INST_ANY = issue stall
(MFMA dependency / pipe busy); SQ_WAIT_INST_LDS = LDS issue stall (a sub-bucket of WAIT_INST_ANY); WAIT_ANY + WAIT_INST_ANY +
ACTIVE_INST_ANY ~= WAVE_CYCLES (quad-cycles).  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = fraction of LDS-array cycles lost to conflicts.
usage: tools/pmc_stalls.py <pmc dir> <out.md> "<title>" """
import csv, glob, re, sys
from collections import defaultdict
csv.field_size_limit(1 << 30)


def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*?>)?)', n)
    return (m.group(1) if m else n)[:70]


def main():
    d, out, title = sys.argv[1:4]
    rows = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    for f in glob.glob(d.rstrip('/') + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            rows[k][r['Counter_Name']] += float(r['Counter_Value'])
            key = (r.get('Dispatch_Id'), r['Start_Timestamp'])
            if key not in cnt[k]:
                cnt[k].add(key)
                rows[k]['_us'] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tab = sorted(((c['_us'], k, c) for k, c in rows.items() if c.get('SQ_WAVE_CYCLES', 0) > 0), reverse=True)
    with open(out, 'w') as f:
        f.write("# %s\n\n" % title)
        f.write("rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS "
                "(one pass, counters summed over the launches of a kernel).  Fractions of SQ_WAVE_CYCLES: parked = SQ_WAIT_ANY (s_waitcnt / s_barrier), "
                "issue-stall = SQ_WAIT_INST_ANY (MFMA dependency / pipe busy; `of which LDS` = SQ_WAIT_INST_LDS), active = SQ_ACTIVE_INST_ANY.  "
                "LDS conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.\n\n")
        f.write("| kernel | launches | total us | parked | issue-stall | of which LDS | active | sum | LDS conflict | LDS insts / wave-kcycle |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for us, k, c in tab[:24]:
            wc = c['SQ_WAVE_CYCLES']
            p, s, l, a = c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc, c.get('SQ_WAIT_INST_LDS', 0) / wc, c.get('SQ_ACTIVE_INST_ANY', 0) / wc
            idx = c.get('SQ_LDS_IDX_ACTIVE', 0)
            conf = c.get('SQ_LDS_BANK_CONFLICT', 0) / idx if idx else 0.0
            f.write("| %s | %d | %.0f | %.3f | %.3f | %.3f | %.3f | %.2f | %.3f | %.1f |\n"
                    % (k, len(cnt[k]), us, p, s, l, a, p + s + a, conf, 1e3 * c.get('SQ_INSTS_LDS', 0) / wc))
    print(open(out).read())


if __name__ == "__main__":
    main()
