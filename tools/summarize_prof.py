#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory into the small files that get committed under
profiles/: <tag>_kernel_stats.csv (rocprofv3 --stats, per kernel), <tag>_summary.json (dominant kernel's
average duration, PMC bytes per launch with the guide's gfx950 correction) and the bench JSON lines."""
import csv
import glob
import json
import os
import sys


def find(root, pattern):
    hits = glob.glob(os.path.join(root, "**", pattern), recursive=True)
    return hits[0] if hits else None


def kernel_stats(root):
    path = find(root, "*kernel_stats.csv")
    rows = []
    if path:
        with open(path) as fh:
            rows = list(csv.DictReader(fh))
    return path, rows


def pmc_per_kernel(root, counter):
    """average counter value per dispatch, per kernel name (counter=None: every counter, keyed (kernel, counter))"""
    path = find(root, "*counter_collection.csv")
    acc = {}
    if not path:
        return acc
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if counter is None:
                k = (row.get("Kernel_Name", "?"), row.get("Counter_Name"))
                v = float(row.get("Counter_Value", "0") or 0)
                s = acc.setdefault(k, [0.0, 0])
                s[0] += v
                s[1] += 1
                continue
            if row.get("Counter_Name") != counter:
                continue
            k = row.get("Kernel_Name", "?")
            v = float(row.get("Counter_Value", "0") or 0)
            s = acc.setdefault(k, [0.0, 0])
            s[0] += v
            s[1] += 1
    return {k: (s[0] / s[1], s[1]) for k, s in acc.items()}


def timed_region(root, steps):
    """mean duration of the LAST `steps` dispatches of the residual + Jacobian kernel in a --kernel-trace run of
    bench.py: the timed region (what comes before it -- the first-allocation timing, the warm-up -- runs the same kernel
    in other allocations, so the whole-run average of --stats is not the number the bench line reports)"""
    path = find(root, "*kernel_trace.csv")
    if not path or not steps:
        return None
    rows = []
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if "k_residual_jacobian" in row.get("Kernel_Name", ""):
                rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
    rows.sort()
    rows = rows[-int(steps):]
    if not rows:
        return None
    d = [e - s for s, e in rows]
    return {"calls": len(d), "avg_ns": sum(d) / len(d), "min_ns": min(d), "max_ns": max(d)}


def bench_line(log):
    try:
        for line in open(log):
            if line.startswith("{") and '"metric"' in line:
                return json.loads(line)
    except OSError:
        pass
    return None


def main():
    out, tag = sys.argv[1], sys.argv[2]
    spath, rows = kernel_stats(os.path.join(out, "trace"))
    summary = {"tag": tag, "kernel_stats_csv": os.path.basename(spath) if spath else None, "kernels": []}
    if rows:
        with open(os.path.join(out, "%s_kernel_stats.csv" % tag), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
        for r in rows[:12]:
            summary["kernels"].append({"name": r.get("Name"), "calls": int(float(r.get("Calls", 0))),
                                       "avg_ns": float(r.get("AverageNs", 0)), "total_ns": float(r.get("TotalDurationNs", 0)),
                                       "pct": float(r.get("Percentage", 0))})
    fetch = pmc_per_kernel(os.path.join(out, "pmc_fetch"), "FETCH_SIZE")
    write = pmc_per_kernel(os.path.join(out, "pmc_write"), "WRITE_SIZE")
    dom = next((k for k in summary["kernels"] if "residual_jacobian" in (k["name"] or "")), None)
    pmc = {}
    for name in set(fetch) | set(write):
        f = fetch.get(name, (None, 0))[0]
        w = write.get(name, (None, 0))[0]
        pmc[name] = {"FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w}
    summary["pmc"] = pmc
    if dom:
        name = dom["name"]
        inst = name.split("c2b::")[-1].split("(")[0]                 # the instance the timed region ran (r05: the launch shape follows the set's store rate)
        key = next((k for k in pmc if inst in k), None) or next((k for k in pmc if "residual_jacobian" in k), None)
        if key and pmc[key]["FETCH_SIZE_KB_avg"] is not None and pmc[key]["WRITE_SIZE_KB_avg"] is not None:
            f, w = pmc[key]["FETCH_SIZE_KB_avg"], pmc[key]["WRITE_SIZE_KB_avg"]
            # MI355X_MICROARCH.md section HBM: counters are KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a
            # wide coalesced read -> doubled; WRITE_SIZE reads exactly for 16-B-per-lane streaming stores.
            summary["traffic_bytes_per_launch"] = int((2.0 * f + w) * 1024)
            summary["traffic_uncorrected_bytes_per_launch"] = int((f + w) * 1024)
            summary["traffic_note"] = "(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch of %s; separate --pmc passes" % name
        summary["dominant_kernel"] = dom
    sq = pmc_per_kernel(os.path.join(out, "pmc_sq"), None)
    inst = dom["name"].split("c2b::")[-1].split("(")[0] if dom else "residual_jacobian"
    sq_dom = {c: round(v[0], 1) for (k, c), v in sq.items() if inst in k} or {c: round(v[0], 1) for (k, c), v in sq.items() if "residual_jacobian" in k}
    if sq_dom:
        wc = sq_dom.get("SQ_WAVE_CYCLES") or 0
        summary["sq_counters_dominant_kernel"] = sq_dom
        if wc:
            summary["sq_derived"] = {
                "lds_bank_conflict_frac_of_lds_active": round(sq_dom.get("SQ_LDS_BANK_CONFLICT", 0) / max(sq_dom.get("SQ_LDS_IDX_ACTIVE", 1), 1), 4),
                "wait_any_frac_of_wave_cycles": round(sq_dom.get("SQ_WAIT_ANY", 0) / wc, 4),
                "active_inst_any_frac": round(sq_dom.get("SQ_ACTIVE_INST_ANY", 0) / wc, 4),
                "active_inst_valu_frac": round(sq_dom.get("SQ_ACTIVE_INST_VALU", 0) / wc, 4),
            }
    summary["bench_trace"] = bench_line(os.path.join(out, "trace.log"))
    if summary["bench_trace"]:
        tr = timed_region(os.path.join(out, "trace"), summary["bench_trace"].get("steps"))
        summary["timed_region_in_kernel_trace"] = tr
        if tr:
            ev = summary["bench_trace"]["roofline"]["kernel_avg_us"]
            summary["timed_region_vs_bench_events"] = {"rocprofv3_avg_us": round(tr["avg_ns"] / 1e3, 2), "bench_events_avg_us": ev,
                                                        "ratio": round(tr["avg_ns"] / 1e3 / ev, 4)}
    summary["bench_pmc_fetch"] = bench_line(os.path.join(out, "pmc_fetch.log"))
    with open(os.path.join(out, "%s_summary.json" % tag), "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps({k: summary.get(k) for k in ("dominant_kernel", "timed_region_vs_bench_events", "traffic_bytes_per_launch",
                                                  "sq_counters_dominant_kernel", "sq_derived")}, indent=1))
    for k in summary["kernels"][:8]:
        print("%-90s calls=%d avg=%.1f us  %.1f%%" % (k["name"][:90], k["calls"], k["avg_ns"] / 1e3, k["pct"]))


if __name__ == "__main__":
    main()
