#!/usr/bin/env python3
"""Summarise a rocprofv3 run directory (kernel stats + separate PMC passes) into
the text committed under profiles/.  Usage: tools/summarize_prof.py gpurun_out/prof_r01b > profiles/...txt"""
import collections
import csv
import sys
from pathlib import Path

root = Path(sys.argv[1])
out = []
ks = root / "kt" / "kt_kernel_stats.csv"
if ks.exists():
    out.append("== rocprofv3 --kernel-trace --stats (kernel_stats.csv) ==")
    rows = list(csv.DictReader(open(ks)))
    out.append("%-78s %6s %12s %12s %8s" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
    for r in rows:
        out.append("%-78s %6s %12.3f %12.4f %8s" % (r["Name"][:78], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                    float(r["AverageNs"]) / 1e6, r["Percentage"]))
log = root / "kt.log"
iters = None  # VBEM iterations of the profiled run (timed + warm-up): launches per step = calls / iters
if log.exists():
    for line in open(log):
        if line.startswith("{"):
            out.append("\n== bench.py line of the profiled run ==\n" + line.strip())
            try:
                import json as _json

                _d = _json.loads(line)
                iters = int(_d["steps"]) + int(_d["warmup"])
            except Exception:  # noqa: BLE001
                pass


def short(n):
    """Kernel family + template arguments (every instantiation is reported on its own)."""
    for k in ("fused_small_kernel", "estep_diag_mfma_kernel", "estep_diag_kernel", "suffstat_diag_kernel", "estep_wide_kernel", "estep_kernel", "suffstat_feat_kernel", "suffstat_quad_kernel",
              "suffstat_kernel"):
        i = n.find(k)
        if i >= 0:
            j = n.find("(", i)
            return n[i:j if j > 0 else None].strip()
    return None


agg = collections.defaultdict(lambda: collections.defaultdict(list))
for tag in ("p1", "p2", "p3"):
    cc = root / tag / f"{tag}_counter_collection.csv"
    if not cc.exists():
        continue
    kt = {r["Dispatch_Id"]: r for r in csv.DictReader(open(root / tag / f"{tag}_kernel_trace.csv"))}
    seen = set()
    for r in csv.DictReader(open(cc)):
        s = short(r["Kernel_Name"])
        if not s:
            continue
        agg[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
        k = kt.get(r["Dispatch_Id"])
        if k and (tag, r["Dispatch_Id"]) not in seen:
            seen.add((tag, r["Dispatch_Id"]))
            agg[s]["dur_ns_" + tag].append(float(k["End_Timestamp"]) - float(k["Start_Timestamp"]))
out.append("\n== PMC (separate --pmc passes; per-dispatch means) ==")
for s, d in agg.items():
    m = {k: sum(v) / len(v) for k, v in d.items()}
    out.append(f"[{s}]")
    for k in sorted(m):
        out.append(f"  {k:34s} {m[k]:.6g}")
    if "GRBM_GUI_ACTIVE" in m and "dur_ns_p1" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8  # summed over the 8 XCDs
        out.append(f"  -> effective clock               {cyc / m['dur_ns_p1']:.3f} GHz")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            out.append(f"  -> MFMA pipe busy                {m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc * 100:.1f} % "
                       f"(SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / GRBM cycles per XCD)")
        if "SQ_INSTS_VALU_MFMA_MOPS_F64" in m:
            out.append(f"  -> executed MFMA flops           {m['SQ_INSTS_VALU_MFMA_MOPS_F64'] * 512 / m['dur_ns_p1'] / 1e3:.2f} TFLOP/s "
                       f"(v_mfma_f64_4x4x4_4b = 512 flop)")
    if "FETCH_SIZE" in m:
        # FETCH_SIZE is in KiB and reads 1/2 of a wide coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM): x2
        out.append(f"  -> HBM read  (FETCH_SIZE*1024*2)  {m['FETCH_SIZE'] * 1024 * 2 / 1e9:.3f} GB per launch")
    if "WRITE_SIZE" in m:
        out.append(f"  -> HBM write (WRITE_SIZE*1024)    {m['WRITE_SIZE'] * 1024 / 1e9:.3f} GB per launch (uncalibrated)")
# machine-readable: HBM bytes per launch (read = FETCH_SIZE KiB x 1024 x 2 on gfx950, write = WRITE_SIZE KiB x 1024, per
# /opt/skills/guides/MI355X_MICROARCH.md "HBM") and launches per VBEM iteration, per kernel instance.  tools/pmc_traffic.py
# collects these sections into profiles/rNN_pmc_traffic.json (what bench.py prints as roofline.traffic); a CPU test
# asserts that the JSON says what the summaries say.
import json

calls = {}
if ks.exists():
    for r in rows:
        sname = short(r["Name"])
        if sname:
            calls[sname] = calls.get(sname, 0) + int(r["Calls"])
traffic = {}
for sname, d in agg.items():
    m = {k: sum(v) / len(v) for k, v in d.items()}
    if "FETCH_SIZE" not in m and "WRITE_SIZE" not in m:
        continue
    rd = m.get("FETCH_SIZE", 0.0) * 1024 * 2
    wr = m.get("WRITE_SIZE", 0.0) * 1024
    e = {"read_bytes": round(rd), "write_bytes": round(wr), "bytes_per_launch": round(rd + wr)}
    if iters and sname in calls:
        e["launches_per_step"] = calls[sname] / iters
        e["total_ms_in_run"] = next((float(r["TotalDurationNs"]) / 1e6 for r in rows if short(r["Name"]) == sname), None)
    traffic[sname] = e
if traffic:
    out.append("\n== traffic (machine-readable; tools/pmc_traffic.py) ==\nTRAFFIC " + json.dumps(traffic, sort_keys=True))
print("\n".join(out))
