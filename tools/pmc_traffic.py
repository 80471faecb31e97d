#!/usr/bin/env python
"""HBM bytes per launch of every kernel of the bench step from two rocprofv3 PMC passes (tools/pmc_bench.sh).
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-counts wide (16 B/lane) streaming reads by exactly 2x
(MI355X_MICROARCH.md, section HBM), so it is doubled; WRITE_SIZE is uncalibrated there and used as is.  Infinity-cache hits
are counted too (memory-side of L2), so "traffic" is an upper bound of true HBM bytes.
usage: tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [batch] [precision] [size] [classes]"""
import csv
import json
import re
import sys

csv.field_size_limit(1 << 30)


def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    m = re.match(r'([A-Za-z0-9_:]+)', n)
    base = m.group(1) if m else n
    if base == "conv_wgrad_wide":              # the Adam form <T, true> (fc6's weight gradient + update) apart from <T, false> (fc7 / fc6's dgrad GEMM)
        t = re.search(r'<[^,>]*,\s*(true|false|1|0)\s*>', n)
        if t:
            base += "<T,%s>" % ("true" if t.group(1) in ("true", "1") else "false")
    return base


def collect(d, counter):
    tot, cnt = {}, {}
    with open(d.rstrip('/') + '/pmc_counter_collection.csv') as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] != counter:
                continue
            k = short(row['Kernel_Name'])
            tot[k] = tot.get(k, 0.0) + float(row['Counter_Value'])
            cnt[k] = cnt.get(k, 0) + 1
    return tot, cnt


def main():
    fd, wd, out = sys.argv[1:4]
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    prec = sys.argv[5] if len(sys.argv) > 5 else "bf16"
    size = int(sys.argv[6]) if len(sys.argv) > 6 else 512
    classes = int(sys.argv[7]) if len(sys.argv) > 7 else 59
    f, nf = collect(fd, "FETCH_SIZE")
    w, nw = collect(wd, "WRITE_SIZE")
    kernels = {}
    for k in sorted(f, key=lambda k: -(2 * f[k] + w.get(k, 0.0))):
        if k.startswith("at::") or k.startswith("__amd"):
            continue
        fetch = 2.0 * f[k] * 1024.0 / nf[k]
        write = w.get(k, 0.0) * 1024.0 / max(nw.get(k, 1), 1)
        kernels[k] = {"launches_sampled": nf[k], "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write),
                      "hbm_bytes_per_launch": round(fetch + write)}
    json.dump({"command": "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-events", "per_gpu_batch": batch,
               "precision": prec, "size": size, "classes": classes,
               "note": "FETCH_SIZE x2 (gfx950 wide-read under-count) + WRITE_SIZE, KB -> bytes; separate --pmc passes; memory-side "
                       "of L2 (infinity-cache hits included)",
               "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in list(kernels.items())[:12]:
        print("%-28s %5d launches  %8.1f MB/launch" % (k, v["launches_sampled"], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
