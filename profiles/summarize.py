#!/usr/bin/env python3
"""Turn rocprofv3 output directories (under gpurun_out/) into the summaries kept in profiles/.

  python profiles/summarize.py stats  <dir>                              profiles/rNN_x_kernel_stats.csv
  python profiles/summarize.py pmc    <fetch-dir> <write-dir> <config>   profiles/rNN_x_pmc_summary.csv
  python profiles/summarize.py sq     <dir>                              profiles/rNN_x_sq_summary.csv

<dir> holds either rocprofv3's CSV output (--output-format csv) or its default rocpd SQLite
database (*_results.db); both are read without the GPU.

`pmc` also rewrites profiles/pmc_traffic.json (read by bench.py for roofline.traffic): HBM bytes
per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- FETCH_SIZE/WRITE_SIZE are in KiB and FETCH_SIZE
under-reports by 2x on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section).  The two counters
come from SEPARATE --pmc passes of the same command.  Launches from the first (untimed, cold)
step are included; they change the mean by < 2 %.
"""
import csv
import glob
import json
import os
import re
import sqlite3
import statistics
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.abspath(__file__))

# kernels whose traffic bench.py can report, keyed by the short name it uses
# (a short name sums every kernel its pattern matches: both renderers, both overflow stages ...)
SHORT = {
    "k_sweep": re.compile(r"k_sweep<(true|0)"),
    "k_sweep_overflow": re.compile(r"k_sweep<(false|1|2)"),
    "k_sweep_orient": re.compile(r"k_sweep_orient|k_hyp_stats"),
    "k_normals": re.compile(r"k_normals"),
    "k_frames": re.compile(r"k_frames"),
    "k_render": re.compile(r"k_render"),
    "k_lenet_conv": re.compile(r"k_lenet_conv"),
    "k_lenet_fc": re.compile(r"k_lenet_fc1"),
}


def find(d, suffix, required=True):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not hits and required:
        sys.exit(f"no *{suffix} under {d}")
    return hits[0] if hits else None


def dispatch_durations(d):
    """kernel name -> list of dispatch durations in ns"""
    out = defaultdict(list)
    db = find(d, "_results.db", required=False)
    if db:
        con = sqlite3.connect(db)
        for name, dur in con.execute("select name, duration from kernels"):
            out[name].append(int(dur))
        return out
    with open(find(d, "_kernel_trace.csv"), newline="") as f:
        for row in csv.DictReader(f):
            out[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    return out


def counter_means(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    db = find(d, "_results.db", required=False)
    if db:
        con = sqlite3.connect(db)
        rows = con.execute("select kernel_name, value from counters_collection where counter_name = ?",
                           (counter,))
    else:
        with open(find(d, "_counter_collection.csv"), newline="") as f:
            rows = [(r["Kernel_Name"], r["Counter_Value"]) for r in csv.DictReader(f)
                    if r["Counter_Name"] == counter]
    for name, value in rows:
        a = acc[name]
        a[0] += 1
        a[1] += float(value)
    return {k: (n, s / n) for k, (n, s) in acc.items()}


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else ""
    if mode == "stats":
        src, dst = sys.argv[2], sys.argv[3]
        durs = dispatch_durations(src)
        total = sum(sum(v) for v in durs.values())
        with open(dst, "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([k, len(v), sum(v), round(sum(v) / len(v), 3), round(100.0 * sum(v) / total, 2),
                            min(v), max(v), round(statistics.pstdev(v), 3)])
        print("wrote", dst)
        return
    if mode == "pmc":
        fdir, wdir, config, dst = sys.argv[2:6]
        fetch = counter_means(fdir, "FETCH_SIZE")
        write = counter_means(wdir, "WRITE_SIZE")
        rows = []
        for k in sorted(fetch):
            if not k.startswith(("ag2::", "void ag2::")):
                continue
            n, fk = fetch[k]
            wk = write.get(k, (0, 0.0))[1]
            rows.append((k, n, round(fk, 1), round(wk, 1), int((2 * fk + wk) * 1024)))
        with open(dst, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "FETCH_SIZE_KB_mean", "WRITE_SIZE_KB_mean",
                        "hbm_bytes_per_launch=(2*FETCH+WRITE)*1024"])
            w.writerows(rows)
        tpath = os.path.join(ROOT, "pmc_traffic.json")
        traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
        traffic[config] = {s: sum(r[4] for r in rows if rx.search(r[0])) for s, rx in SHORT.items()
                           if any(rx.search(r[0]) for r in rows)}
        # HBM write bytes per launch (WRITE_SIZE x 1024) on their own: bench.py divides them by the
        # bytes the kernel has to write (roofline.write_amplification: register spills show up here)
        traffic.setdefault("_writes", {})[config] = {
            s: int(sum(r[3] for r in rows if rx.search(r[0])) * 1024) for s, rx in SHORT.items()
            if any(rx.search(r[0]) for r in rows)}
        traffic.setdefault("_profile", {})[config] = os.path.basename(dst)
        traffic["_source"] = ("per config: profiles/<_profile[config]> (rocprofv3 --kernel-trace --pmc FETCH_SIZE / "
                              "WRITE_SIZE, separate passes of python3 bench.py --config <config> --steps 3 --warmup 1 "
                              "--no-cpu); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch")
        json.dump(traffic, open(tpath, "w"), indent=1)
        print("wrote", dst, "and", tpath)
        return
    if mode == "sq":
        # per-kernel means of the SQ counters of one --pmc pass; MFMA-busy fraction as the guide
        # defines the units: SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs of the busy
        # CUs, SQ_BUSY_CU_CYCLES counts cycles summed over busy CUs (4 SIMDs each)
        src, dst = sys.argv[2], sys.argv[3]
        config = sys.argv[4] if len(sys.argv) > 4 else "cfg2"
        names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU",
                 "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"]
        cols = {n: counter_means(src, n) for n in names}
        kernels = sorted(k for k in cols["SQ_BUSY_CU_CYCLES"] if k.startswith(("ag2::", "void ag2::")))
        with open(dst, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls"] + [n + "_mean" for n in names] +
                       ["mfma_busy_frac=MFMA_BUSY/(4*BUSY_CU)", "wave_wait_frac=WAIT_INST_ANY/WAVE_CYCLES",
                        "wave_valu_frac=ACTIVE_INST_VALU/WAVE_CYCLES"])
            for k in kernels:
                v = {n: cols[n].get(k, (0, 0.0))[1] for n in names}
                busy = v["SQ_BUSY_CU_CYCLES"] or 1.0
                wc = v["SQ_WAVE_CYCLES"] or 1.0
                w.writerow([k, cols["SQ_BUSY_CU_CYCLES"][k][0]] + [round(v[n], 1) for n in names] +
                           [round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * busy), 4),
                            round(v["SQ_WAIT_INST_ANY"] / wc, 4), round(v["SQ_ACTIVE_INST_VALU"] / wc, 4)])
        # MFMA-busy fractions of the matrix-core kernels, read by bench.py next to the traffic
        mpath = os.path.join(ROOT, "pmc_traffic.json")
        tr = json.load(open(mpath)) if os.path.exists(mpath) else {}
        busy = {}
        for short in ("k_lenet_conv", "k_lenet_fc"):
            for k in kernels:
                if SHORT[short].search(k):
                    b = cols["SQ_BUSY_CU_CYCLES"][k][1] or 1.0
                    busy[short] = round(cols["SQ_VALU_MFMA_BUSY_CYCLES"].get(k, (0, 0.0))[1] / (4.0 * b), 4)
        tr["mfma_busy" if config == "cfg2" else "mfma_busy_" + config] = busy
        if config == "cfg2":
            tr["_mfma_source"] = f"profiles/{os.path.basename(dst)}: SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)"
        json.dump(tr, open(mpath, "w"), indent=1)
        print("wrote", dst, "and", mpath)
        return
    sys.exit(__doc__)


if __name__ == "__main__":
    main()
