#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into small CSVs for profiles/.

    pmc_sum.py counters OUT.csv DIR [DIR...]     mean counter value per dispatch per kernel (from *counter_collection.csv)
    pmc_sum.py stats OUT.csv DIR                 per-kernel calls / total / average duration (from *kernel_stats.csv)
    pmc_sum.py json OUT.json SQ.csv FETCH.csv WRITE.csv [NOTE]   derived per-kernel figures (HBM bytes per launch, MFMA busy) from the counter CSVs
Kernel names are shortened to the template head so the files stay readable."""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:90]


def counters(out, dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(out, "w") as f:
        f.write("kernel,counter,mean_per_launch,launches\n")
        for k in sorted(acc):
            for c in sorted(acc[k]):
                v = acc[k][c]
                f.write(f'"{k}",{c},{sum(v) / len(v):.0f},{len(v)}\n')


def stats(out, d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["Percentage"])))
    rows.sort(key=lambda r: -r[2])
    with open(out, "w") as f:
        f.write("kernel,calls,total_ms,avg_us,percent\n")
        for n, c, t, a, p in rows:
            f.write(f'"{n}",{c},{t / 1e6:.3f},{a / 1e3:.2f},{p:.2f}\n')


def pmc_json(out, sq_csv, fetch_csv, write_csv, note):
    """The per-kernel derived figures bench.py quotes (`roofline.traffic`) and DESIGN.md tabulates, from the three counter CSVs above:
    HBM bytes per launch = FETCH_SIZE x 2 (gfx950 wide-read correction, MI355X_MICROARCH.md) + WRITE_SIZE, KB -> bytes;
    MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)."""
    import json

    def load(path):
        acc = collections.defaultdict(dict)
        for r in csv.DictReader(open(path)):
            acc[r["kernel"]][r["counter"]] = (float(r["mean_per_launch"]), int(r["launches"]))
        return acc
    sq, fe, wr = load(sq_csv), load(fetch_csv), load(write_csv)

    def one(k):
        d = {}
        if k in fe and "FETCH_SIZE" in fe[k]:
            d["fetch_size_kb"] = fe[k]["FETCH_SIZE"][0]
        if k in wr and "WRITE_SIZE" in wr[k]:
            d["write_size_kb"] = wr[k]["WRITE_SIZE"][0]
        if "fetch_size_kb" in d:
            d["hbm_bytes_per_launch"] = (2.0 * d["fetch_size_kb"] + d.get("write_size_kb", 0.0)) * 1024.0
        c = sq.get(k, {})
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"][0] > 0:
            d["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / 1024.0 / (c["GRBM_GUI_ACTIVE"][0] / 8.0)
        if "SQ_LDS_BANK_CONFLICT" in c:
            d["lds_bank_conflict_cycles"] = c["SQ_LDS_BANK_CONFLICT"][0]
        if k in fe and "FETCH_SIZE" in fe[k]:
            d["launches"] = fe[k]["FETCH_SIZE"][1]
        return d
    conv = [k for k in fe if k.startswith("conv3x3_halo4_kernel<3")]
    conv = max(conv, key=lambda k: fe[k]["FETCH_SIZE"][1]) if conv else None
    res = {"kernel": conv, "note": note}
    if conv:
        res.update(one(conv))
    res["linear"] = {k: one(k) for k in sorted(fe) if k.startswith("lin4_kernel")}
    res["flash_attention"] = {k: one(k) for k in sorted(fe) if k.startswith("flash_d32")}
    res["groupnorm"] = {k: one(k) for k in sorted(fe) if k.startswith("gn_")}
    knn = [k for k in fe if k.startswith("knn_scan")]
    if knn:
        res["knn_scan"] = dict(kernel=knn[0], **one(knn[0]))
    with open(out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "counters":
        counters(sys.argv[2], sys.argv[3:])
    elif sys.argv[1] == "json":
        pmc_json(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6] if len(sys.argv) > 6 else "")
    else:
        stats(sys.argv[2], sys.argv[3])
