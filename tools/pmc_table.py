#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 counter_collection CSVs (any number of passes): one row per kernel, one column
per counter, plus the dispatch count and the average duration of the pass that carried GRBM_GUI_ACTIVE (or the first).
Derived columns, where their inputs are present (MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are
quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES cycles per SIMD; GRBM_GUI_ACTIVE summed over the 8 XCDs):
  clock_GHz = GRBM_GUI_ACTIVE / 8 / duration;  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)
  wait_any / wait_inst_any / active_inst_any / wait_inst_lds: shares of SQ_WAVE_CYCLES
  lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
usage: python tools/pmc_table.py pass0.csv pass1.csv ..."""
import collections
import csv
import re
import sys

vals = collections.defaultdict(lambda: collections.defaultdict(list))
durs = collections.defaultdict(dict)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        m = re.search(r"(k_\w+(?:<.*>)?)", name)
        k = m.group(1) if m else name.split("(")[0]
        k = re.sub(r"\s+", "", k)
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        durs[k][(path, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
counters = sorted({c for k in vals for c in vals[k]})
derived = ["clock_GHz", "mfma_busy", "wait_any", "wait_inst_any", "active_inst_any", "wait_inst_lds", "lds_conflict_share",
           "lds_idx_active_per_cu_cycle"]
w = csv.writer(sys.stdout)
w.writerow(["kernel", "dispatches", "avg_us"] + counters + derived)
for k in sorted(vals):
    c = {n: sum(v) / len(v) for n, v in vals[k].items()}
    n = max(len(v) for v in vals[k].values())
    t_ns = sum(durs[k].values()) / len(durs[k])
    if t_ns < 20000:          # counters of the small kernels say little: keep the table to the GEMMs and friends
        continue
    d = {}
    g = c.get("GRBM_GUI_ACTIVE")
    if g:
        d["clock_GHz"] = g / 8 / t_ns
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (g / 8 * 1024)
        if "SQ_LDS_IDX_ACTIVE" in c:
            d["lds_idx_active_per_cu_cycle"] = c["SQ_LDS_IDX_ACTIVE"] / (g / 8 * 256)
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for a, b in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"),
                     ("active_inst_any", "SQ_ACTIVE_INST_ANY"), ("wait_inst_lds", "SQ_WAIT_INST_LDS")):
            if b in c:
                d[a] = c[b] / wc
    if c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_share"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    w.writerow([k, n, "%.1f" % (t_ns / 1e3)] + ["%.6g" % c[x] if x in c else "" for x in counters]
               + ["%.4f" % d[x] if x in d else "" for x in derived])
