#!/usr/bin/env python3
"""Register / LDS / scratch use of the gfx950 kernels and, for one kernel, the skeleton of its vector-memory
operations, waits and branches (what the software-pipelined kernels are checked against: no s_waitcnt vmcnt(0)
and no scratch inside the loop).  Works on the CPU box: hipcc cross-compiles.

    python tools/isa_report.py [--tune] [--filter k_residual_jacobian_p] [--dump <mangled-name-substring>]
"""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "city2ba_amd", "csrc")

ap = argparse.ArgumentParser()
ap.add_argument("--tune", action="store_true", help="compile with -DC2B_TUNE (all variants)")
ap.add_argument("--filter", default="", help="only kernels whose demangled name contains this")
ap.add_argument("--dump", default="", help="print the VM-op / wait skeleton of the first kernel whose name contains this")
ap.add_argument("--keep", default="", help="directory to keep the temporaries in")
a = ap.parse_args()

tmp = a.keep or tempfile.mkdtemp(prefix="c2b_isa_")
os.makedirs(tmp, exist_ok=True)
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
       "-pthread", "--save-temps=obj", "-o", os.path.join(tmp, "lib.so"), os.path.join(CSRC, "capi.hip")]
if a.tune:
    cmd.insert(1, "-DC2B_TUNE")
subprocess.check_call(cmd, cwd=CSRC)
asm = open(glob.glob(os.path.join(tmp, "*gfx950*.s"))[0]).read()


def demangle(n):
    try:
        return subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], text=True).strip()
    except Exception:
        return n


rows = []
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", asm, re.S):
    name, blk = m.group(1), m.group(2)
    if not name.startswith("_Z"):
        continue
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) if re.search(r"\.%s:\s+(\d+)" % k, blk) else 0
    rows.append((demangle(name).split("(")[0].replace("void c2b::", ""), name, g("vgpr_count"), g("sgpr_count"),
                 g("group_segment_fixed_size"), g("private_segment_fixed_size")))
for dn, name, v, s, l, sc in rows:
    if a.filter in dn:
        waves = 512 // ((v + 7) // 8 * 8) if v else 8
        print("%-70s vgpr %3d (%d waves/SIMD)  sgpr %3d  lds %6d  scratch %d" % (dn[:70], v, min(waves, 8), s, l, sc))

if a.dump:
    for dn, name, *_ in rows:
        if a.dump in dn or a.dump in name:
            i = asm.index(name + ":")
            j = asm.index("s_endpgm", i)
            print("\n== %s ==" % dn)
            for k, line in enumerate(asm[i:j].split("\n")):
                t = line.strip()
                if re.match(r"(global_|flat_|buffer_|scratch_|s_waitcnt|s_cbranch|\.LBB|s_barrier|s_branch)", t):
                    print("%5d  %s" % (k, t[:110]))
            break
    else:
        sys.exit("no kernel matches --dump")
