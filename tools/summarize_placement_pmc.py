#!/usr/bin/env python3
"""Read the passes of tools/probe_placement_pmc.sh: per pass, the mean of every counter over the store-pattern
dispatches into the slow and into the fast allocation (told apart by grid size), their mean durations, and the
slow / fast ratio of each."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
out = {}
for meta in sorted(glob.glob(os.path.join(root, "*.json"))):
    name = os.path.basename(meta)[:-5]
    if name in ("summary",):
        continue
    try:
        info = json.loads(open(meta).read().strip().split("\n")[-1])
    except Exception as exc:
        out[name] = {"error": "no probe line: %s" % exc}
        continue
    entry = {"store_GBs_per_set": info["store_GBs_per_set"], "rates_tagged": info["rates_tagged"]}
    hits = glob.glob(os.path.join(root, name, "**", "*counter_collection.csv"), recursive=True)
    if hits:
        acc = {}
        with open(hits[0]) as fh:
            for row in csv.DictReader(fh):
                if "k_store_pattern" not in row.get("Kernel_Name", ""):
                    continue
                g = int(row["Grid_Size"])
                tag = "slow" if g == info["grid_slow"] else ("fast" if g == info["grid_fast"] else None)
                if tag is None:
                    continue
                d = acc.setdefault((tag, row["Counter_Name"]), {})
                d.setdefault(row["Dispatch_Id"], []).append(float(row["Counter_Value"] or 0))
                t = acc.setdefault((tag, "_duration_us"), {})
                t[row["Dispatch_Id"]] = [(int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3]
        counters = sorted({c for _, c in acc})
        for c in counters:
            rec = {}
            for tag in ("slow", "fast"):
                per = acc.get((tag, c), {})
                if not per:
                    continue
                sums = [sum(v) for v in per.values()]
                rec[tag] = round(sum(sums) / len(sums), 2)
                rec[tag + "_dispatches"] = len(sums)
                width = {len(v) for v in per.values()}
                if width != {1}:
                    rec[tag + "_values_per_dispatch"] = sorted(width)
                    # per-instance spread (max / mean over the instances), averaged over the dispatches
                    rec[tag + "_instance_max_over_mean"] = round(
                        sum(max(v) / (sum(v) / len(v)) for v in per.values() if sum(v) > 0) / max(1, len(per)), 4)
            if "slow" in rec and "fast" in rec and rec["fast"]:
                rec["slow_over_fast"] = round(rec["slow"] / rec["fast"], 4)
            entry[c] = rec
    else:
        entry["note"] = "no counter CSV (plain run or the pass failed)"
    out[name] = entry
print(json.dumps(out, indent=1))
