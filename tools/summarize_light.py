#!/usr/bin/env python3
"""Per-kernel table from the passes of tools/profile_light.sh: average duration (kernel trace), HBM bytes past the L2 per
launch (2 x FETCH_SIZE + WRITE_SIZE KiB, the guide's gfx950 correction), vector instructions per launch, and two fractions
of what the chip offers over the kernel's duration:
  valu_issue_frac = SQ_INSTS_VALU x 4 cycles / (1 024 SIMDs x 2.4 GHz x duration)   -- every vector instruction holds its
                    SIMD's issue port for at least four cycles (f64 add / mul / fma / min / max: exactly four; divides'
                    and square roots' reciprocal seeds and 64-bit integer multiplies: more), so this is a LOWER bound of
                    the share of issue slots taken, against the nominal clock (the chip runs these kernels below it);
  valu_busy_frac  = SQ_ACTIVE_INST_VALU x 4 / (1 024 x 2.4 GHz x duration)           -- SQ_ACTIVE_INST_* count quad-cycles
                    (MI355X_MICROARCH.md), so this one includes the multi-pass instructions at their real cost;
  hbm_frac        = HBM bytes / duration / 8 TB/s.
`bound`: "valu" when valu_busy_frac >= hbm_frac and >= 0.5, "hbm" when hbm_frac > valu_busy_frac and >= 0.5, otherwise
"latency" (neither resource half used: the kernel waits on dependent round trips).  binding_frac = the larger of the two."""
import collections
import csv
import glob
import json
import sys

out, tag = sys.argv[1], sys.argv[2]
CLOCK, SIMDS, PEAK = 2.4e9, 1024, 8.0e12


def find(sub, pat):
    h = glob.glob(out + "/" + sub + "/**/" + pat, recursive=True)
    return h[0] if h else None


def short(name):
    return name.split("c2b::")[1].split("(")[0] if "c2b::" in name else None


stats = {}
f = find("trace", "*kernel_stats.csv")
if f:
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k:
            stats[k] = {"calls": int(float(r["Calls"])), "avg_us": round(float(r["AverageNs"]) / 1e3, 2)}


def pmc(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    f = find(sub, "*counter_collection.csv")
    if f:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


sq = pmc("pmc_sq1")
for k, d in pmc("pmc_sq2").items():
    sq.setdefault(k, {}).update(d)
fe, wr = pmc("pmc_fetch"), pmc("pmc_write")
for k, d in stats.items():
    t = d["avg_us"] * 1e-6
    c = sq.get(k, {})
    if "FETCH_SIZE" in fe.get(k, {}) and "WRITE_SIZE" in wr.get(k, {}):
        d["fetch_MB"] = round(2 * fe[k]["FETCH_SIZE"] * 1024 / 1e6, 1)
        d["write_MB"] = round(wr[k]["WRITE_SIZE"] * 1024 / 1e6, 1)
        d["hbm_MB_per_launch"] = round(d["fetch_MB"] + d["write_MB"], 1)
        d["hbm_frac"] = round(d["hbm_MB_per_launch"] * 1e6 / t / PEAK, 3)
    if "SQ_INSTS_VALU" in c:
        d["insts_valu"] = int(c["SQ_INSTS_VALU"])
        d["valu_issue_frac"] = round(c["SQ_INSTS_VALU"] * 4 / (SIMDS * CLOCK * t), 3)
    if "SQ_ACTIVE_INST_VALU" in c:
        d["valu_busy_frac"] = round(c["SQ_ACTIVE_INST_VALU"] * 4 / (SIMDS * CLOCK * t), 3)
    if c.get("SQ_WAVE_CYCLES"):
        for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            if name in c:
                d[name.lower() + "_of_wave_cycles"] = round(c[name] / c["SQ_WAVE_CYCLES"], 3)
    for name in ("SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"):
        if name in c:
            d[name.lower()] = int(c[name])
    v, h = d.get("valu_busy_frac"), d.get("hbm_frac")
    if v is not None and h is not None:
        d["bound"] = "valu" if (v >= h and v >= 0.5) else ("hbm" if (h > v and h >= 0.5) else "latency")
        d["binding_frac"] = max(v, h)
json.dump(stats, open(out + "/../%s_light_sq.json" % tag, "w"), indent=1)
for k, d in sorted(stats.items(), key=lambda x: -x[1]["avg_us"] * x[1]["calls"]):
    print("%-58s calls %3d avg %8.2f us  HBM %7s MB (frac %5s)  VALU insts %10s issue %5s busy %5s  wait_any %5s  -> %s" % (
        k[:58], d["calls"], d["avg_us"], d.get("hbm_MB_per_launch"), d.get("hbm_frac"), d.get("insts_valu"), d.get("valu_issue_frac"),
        d.get("valu_busy_frac"), d.get("sq_wait_any_of_wave_cycles"), d.get("bound")))
