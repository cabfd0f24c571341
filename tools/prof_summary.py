#!/usr/bin/env python3
"""Summarise the rocprofv3 (rocpd sqlite) outputs of tools/profile.sh into CSV files under profiles/.

  python tools/prof_summary.py gpurun_out/prof_<tag> profiles/<prefix>

Writes <prefix>_kernel_stats.csv (kernel trace: calls, total/avg/min/max ns) and <prefix>_pmc.csv (per kernel,
per launch averages of every collected counter; FETCH_SIZE additionally x2 as MI355X_MICROARCH.md prescribes
for gfx950 wide streaming reads; FETCH/WRITE in bytes).
"""
import csv
import os
import re
import sqlite3
import sys


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


# FETCH_SIZE -> bytes. gfx950 tallies a 128-B request at 64 B (MI355X_MICROARCH.md, HBM section): x2 for reads that arrive as
# 128-B requests. Calibrated on this repository's own access patterns with tools/micro/gather_copy.hip under
# rocprofv3 --pmc FETCH_SIZE (512 MiB read once): 256 B per wave instruction (4 B/lane) -> 262 320 KiB = 1/2; 128-B pieces
# (8 lanes x 16 B) -> 262 152 KiB = 1/2; 64-B pieces (4 lanes x 16 B, the 16-row tiles) -> 524 298 KiB = exact.
# The last case no longer occurs in the product: with the XCD-aware slice order the pieces of a line are requested by
# workgroups of one XCD at about the same time and leave its L2 as whole 128-B requests -- k_conv_unit<true> reads
# 537 MB and the counter shows 277 MB (profiles/r2_e_pmc.csv), i.e. 1/2 again. (k_conv_tile, the fallback for
# SOT_NO_TOP, keeps the old order.)
FETCH_FACTOR_EXACT = ("k_conv_tile",)
QC_PREFIXES = ("k_qc_load", "k_conv_", "k_top", "k_mid", "k_fft_low<1", "k_fft_low_p", "k_tile_fft", "k_qc_finish", "k_clear_lsb",
               "k_sparse_choice", "k_set_points")
GGM_PREFIXES = ("k_roots", "k_ggm_", "k_sender_treegrp", "k_recv_")


def fetch_bytes(kernel, fetch_kib):
    return fetch_kib * 1024 * (1 if kernel.startswith(FETCH_FACTOR_EXACT) else 2)


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    tr = os.path.join(src, "trace", "trace_results.db")
    if os.path.exists(tr):
        c = sqlite3.connect(tr)
        rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                         "from kernels group by name order by sum(duration) desc").fetchall()
        tot = sum(r[2] for r in rows)
        with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.2f" % (100.0 * r[2] / tot), r[4], r[5]])
    pmc = {}
    for sub in ("fetch", "write", "sq", "sq2"):
        p = os.path.join(src, sub, sub + "_results.db")
        if not os.path.exists(p):
            continue
        c = sqlite3.connect(p)
        for name, ctr, n, v, d in c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) "
                                            "from counters_collection group by kernel_name, counter_name"):
            e = pmc.setdefault(short(name), {})
            e[ctr] = v
            e["launches"] = n
            e.setdefault("avg_ns_under_pmc", d)
    ctrs = sorted({k for e in pmc.values() for k in e} - {"launches", "avg_ns_under_pmc"})
    with open(prefix + "_pmc.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "avg_ns_under_pmc"] + ctrs + ["fetch_bytes_corrected", "write_bytes"])
        for k, e in sorted(pmc.items(), key=lambda kv: -kv[1].get("avg_ns_under_pmc", 0) * kv[1]["launches"]):
            fb = e.get("FETCH_SIZE")
            wb = e.get("WRITE_SIZE")
            w.writerow([k, e["launches"], "%.0f" % e["avg_ns_under_pmc"]] + ["%.1f" % e[c] if c in e else "" for c in ctrs]
                       + ["%.0f" % fetch_bytes(k, fb) if fb is not None else "", "%.0f" % (wb * 1024) if wb is not None else ""])


    # profiles/pmc_latest.json: what bench.py reports as roofline.traffic (dominant kernel, per launch)
    # the dominant kernel: the persistent form of the low kernel (n >= 2^24), else the plain one
    dom = next((k for k in pmc if k.startswith("k_fft_low_p")), None) or next((k for k in pmc if k.startswith("k_fft_low<1")), "k_fft_low<1>")
    if dom in pmc and "FETCH_SIZE" in pmc[dom] and "WRITE_SIZE" in pmc[dom]:
        import json
        e = pmc[dom]
        out = {
            "kernel": dom, "source": os.path.basename(prefix) + "_pmc.csv",
            "fetch_size_kib_raw": e["FETCH_SIZE"], "write_size_kib": e["WRITE_SIZE"],
            "dominant_traffic_bytes_per_launch": fetch_bytes(dom, e["FETCH_SIZE"]) + e["WRITE_SIZE"] * 1024,
            "valu_insts_per_launch": e.get("SQ_INSTS_VALU"), "salu_insts_per_launch": e.get("SQ_INSTS_SALU"),
            "lds_insts_per_launch": e.get("SQ_INSTS_LDS"),
            "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950 tallies 128-B "
                    "requests at 64 B, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, per launch of " + dom + ": 512 MiB of bit rows + the per-value-word mask tables (LM, Amat) streamed once per row half",
        }
        # stage totals. The profiled command is bench.py: every QC encode ends in one k_qc_finish launch, so
        # launches // encodes = launches of a kernel per encode (the transforms of the public polynomial at context
        # creation add fewer than one per encode); the GGM kernels run once per expansion = per encode.
        enc = next((e["launches"] for k, e in pmc.items() if k.startswith("k_qc_finish")), 0)
        if enc:
            tot = {"fetch": 0.0, "write": 0.0}
            per_kernel = {}
            for k, ee in pmc.items():
                if not k.startswith(QC_PREFIXES) or "FETCH_SIZE" not in ee or "WRITE_SIZE" not in ee:
                    continue
                per = ee["launches"] // enc
                if not per:
                    continue
                fb, wb = fetch_bytes(k, ee["FETCH_SIZE"]) * per, ee["WRITE_SIZE"] * 1024 * per
                per_kernel[k] = {"launches_per_encode": per, "bytes": fb + wb}
                tot["fetch"] += fb
                tot["write"] += wb
            out["qc_stage_traffic_bytes_per_direction"] = tot["fetch"] + tot["write"]
            out["qc_stage_fetch_bytes_per_direction"] = tot["fetch"]
            out["qc_stage_write_bytes_per_direction"] = tot["write"]
            out["qc_stage_traffic_by_kernel"] = per_kernel
            g = {"valu": 0.0, "lds": 0.0, "ns": 0.0}
            for k, ee in pmc.items():
                if not k.startswith(GGM_PREFIXES) or "SQ_INSTS_VALU" not in ee:
                    continue
                g["valu"] += ee["SQ_INSTS_VALU"] * ee["launches"]
                g["lds"] += ee.get("SQ_INSTS_LDS", 0.0) * ee["launches"]
                g["ns"] += ee["avg_ns_under_pmc"] * ee["launches"]
            if g["ns"]:
                # per second of GGM kernel time, summed over the stage's launches of the profiled run
                out["ggm_valu_insts_per_s"] = g["valu"] / (g["ns"] * 1e-9)
                out["ggm_lds_insts_per_s"] = g["lds"] / (g["ns"] * 1e-9)
        # what the counters were collected on: bench.py compares these hashes with the sources it runs and drops the
        # counter-derived fields when a kernel file has changed since (the numbers would describe other kernels)
        import hashlib
        import subprocess
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        out["source_sha256"] = {}
        for fn in ("qc.hip", "ggm.hip", "aes_dev.h"):
            with open(os.path.join(root, "flute_amd", "csrc", fn), "rb") as fh:
                out["source_sha256"][fn] = hashlib.sha256(fh.read()).hexdigest()
        try:
            out["source_commit"] = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
        except Exception:
            out["source_commit"] = None
        with open(os.path.join(os.path.dirname(prefix), "pmc_latest.json"), "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
