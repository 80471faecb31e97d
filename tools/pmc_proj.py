#!/usr/bin/env python
"""Summary of tools/pmc_proj.sh: kernel-trace duration, HBM bytes (FETCH_SIZE x2 on gfx950 for wide streaming reads, WRITE_SIZE as
is: MI355X_MICROARCH.md, section HBM) and MFMA utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch duration x 2.4 GHz))
of proj_gemm_stream on the nominal 262,144 x 4096 x 300 projection.
usage: tools/pmc_proj.py <dir with stats/ and pmc_*/> <out.json>"""
import csv
import glob
import json
import os
import sys

csv.field_size_limit(1 << 30)
KERN = "proj_gemm_stream"


def main():
    d, out = sys.argv[1:3]
    res = {"kernel": KERN, "shape": {"M": 262144, "K": 4096, "N": 300}, "peak_TF": 2500.0, "peak_hbm_TBps": 8.0}
    bj = os.path.join(d, "bench_proj.json")
    if os.path.exists(bj):
        for line in open(bj):
            if line.startswith("{"):
                res["bench_proj"] = json.loads(line)
    durs = []
    for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if KERN in row["Kernel_Name"]:
                durs.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    if durs:
        durs = durs[1:] if len(durs) > 1 else durs           # the first launch is the warm-up
        avg = sum(durs) / len(durs)
        flop = 2.0 * 262144 * 4096 * 300
        res["kernel_trace"] = {"launches": len(durs), "avg_us": round(avg, 1), "min_us": round(min(durs), 1),
                               "TFLOPs": round(flop / avg / 1e6, 1), "frac_of_peak": round(flop / avg / 1e6 / 2500.0, 4),
                               "activation_stream_TBps": round(262144 * 4096 * 2 / avg / 1e6, 2)}
    ctr, dur_pmc = {}, {}
    for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if KERN in row["Kernel_Name"]:
                ctr.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                dur_pmc.setdefault(row["Counter_Name"], []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    avgc = {k: sum(v) / len(v) for k, v in ctr.items()}
    res["counters_per_launch"] = {k: round(v, 1) for k, v in sorted(avgc.items())}
    if "FETCH_SIZE" in avgc:
        fetch = 2.0 * avgc["FETCH_SIZE"] * 1024.0
        write = avgc.get("WRITE_SIZE", 0.0) * 1024.0
        alg = 262144 * 4096 * 2 + 300 * 4096 * 2 + 262144 * 304 * 2
        res["hbm"] = {"fetch_bytes": round(fetch), "write_bytes": round(write), "traffic_bytes": round(fetch + write),
                      "algorithmic_bytes": alg, "traffic_over_algorithmic": round((fetch + write) / alg, 3),
                      "note": "FETCH_SIZE x2 (gfx950 wide-read under-count), WRITE_SIZE as is; memory-side of L2"}
        if "kernel_trace" in res:
            res["hbm"]["GBps_at_trace_duration"] = round((fetch + write) / res["kernel_trace"]["avg_us"] / 1e3, 1)
            res["hbm"]["frac_of_8TBps"] = round((fetch + write) / res["kernel_trace"]["avg_us"] / 1e6 / 8.0, 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avgc:
        # as profiles/r01_pmc_kernels.txt: busy cycles / (1024 SIMDs x duration of the launches of THAT pass x 2.4 GHz)
        du = dur_pmc["SQ_VALU_MFMA_BUSY_CYCLES"]
        us = sum(du) / len(du)
        res["mfma"] = {"mfma_busy_cycles": round(avgc["SQ_VALU_MFMA_BUSY_CYCLES"]), "pass_avg_us": round(us, 1),
                       "utilisation": round(avgc["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * us * 2400.0), 4),
                       "note": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch duration in that pass x 2.4 GHz)"}
        if "SQ_WAIT_INST_ANY" in avgc and avgc.get("SQ_WAVE_CYCLES"):
            res["mfma"]["wait_inst_any_over_wave_cycles"] = round(avgc["SQ_WAIT_INST_ANY"] / avgc["SQ_WAVE_CYCLES"], 4)
    if "SQ_LDS_BANK_CONFLICT" in avgc and avgc.get("SQ_LDS_IDX_ACTIVE"):
        res["lds_conflict_rate"] = round(avgc["SQ_LDS_BANK_CONFLICT"] / avgc["SQ_LDS_IDX_ACTIVE"], 4)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in res if k not in ("bench_proj", "counters_per_launch")}, indent=1))


if __name__ == "__main__":
    main()
