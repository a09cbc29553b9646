#!/usr/bin/env python3
"""Register / LDS / scratch use of the gfx950 kernels and, for one kernel, the skeleton of its vector-memory
operations, waits and branches (what the software-pipelined kernels are checked against: no s_waitcnt vmcnt(0)
and no scratch inside the loop).  Works on the CPU box: hipcc cross-compiles.

    python tools/isa_report.py [--filter k_residual_jacobian_l] [--dump <mangled-name-substring>]

Also a module: compile_asm() / kernel_table() / kernel_body() are what tests/test_isa_pins.py asserts on.
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
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-pthread", "-ldl"]


def compile_asm(keep=""):
    """Compile capi.hip exactly like __graft_entry__.build_hip() plus --save-temps; returns the gfx950 assembly text."""
    tmp = keep or tempfile.mkdtemp(prefix="c2b_isa_")
    os.makedirs(tmp, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc"] + HIPCC_FLAGS + [
        "--save-temps=obj", "-o", os.path.join(tmp, "lib.so"), os.path.join(CSRC, "capi.hip")]
    subprocess.check_call(cmd, cwd=CSRC)
    return open(glob.glob(os.path.join(tmp, "*gfx950*.s"))[0]).read()


def demangle(names):
    names = list(names)
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
        try:
            out = subprocess.run([tool], input="\n".join(names) + "\n", text=True, capture_output=True, check=True).stdout
            got = out.rstrip("\n").split("\n")
            if len(got) == len(names):
                return got
        except Exception:
            continue
    return names


def kernel_table(asm):
    """[{name, mangled, vgpr, sgpr, lds, scratch, vgpr_spill, sgpr_spill}] from the .amdgpu_metadata note.  Every
    kernel is one YAML list item whose keys come in alphabetical order (.group_segment_fixed_size BEFORE .name), so the
    items are split first and every key is looked up inside its own item."""
    meta = asm[asm.index("amdhsa.kernels:"):]
    items = re.split(r"\n  - (?=\.)", meta)[1:]
    rows = []
    for it in items:
        def g(key, it=it):
            m = re.search(r"\.%s:\s+(\d+)" % key, it)
            return int(m.group(1)) if m else 0
        m = re.search(r"\.name:\s+(\S+)", it)
        if not m or not m.group(1).startswith("_Z"):
            continue
        rows.append({"mangled": m.group(1), "vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"),
                     "lds": g("group_segment_fixed_size"), "scratch": g("private_segment_fixed_size"),
                     "vgpr_spill": g("vgpr_spill_count"), "sgpr_spill": g("sgpr_spill_count")})
    for r, dn in zip(rows, demangle([r["mangled"] for r in rows])):
        dn = dn.split("(")[0]
        dn = dn[5:] if dn.startswith("void ") else dn
        r["name"] = dn[5:] if dn.startswith("c2b::") else dn
    return rows


def kernel_body(asm, mangled):
    """the instruction lines of one kernel (label to s_endpgm), stripped"""
    i = asm.index("\n" + mangled + ":")
    j = asm.index(".amdhsa_kernel", i) if ".amdhsa_kernel" in asm[i:] else len(asm)
    return [ln.strip() for ln in asm[i:j].split("\n") if ln.strip() and not ln.strip().startswith(";")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filter", default="", help="only kernels whose demangled name contains this")
    ap.add_argument("--dump", default="", help="print the VM-op / wait skeleton of the first kernel whose name contains this")
    ap.add_argument("--keep", default="", help="directory to keep the temporaries in")
    a = ap.parse_args()
    asm = compile_asm(a.keep)
    rows = kernel_table(asm)
    for r in rows:
        if a.filter in r["name"]:
            v = r["vgpr"]
            waves = 512 // ((v + 7) // 8 * 8) if v else 8
            print("%-70s vgpr %3d (%d waves/SIMD)  sgpr %3d  lds %6d  scratch %d" % (
                r["name"][:70], v, min(waves, 8), r["sgpr"], r["lds"], r["scratch"]))
    if a.dump:
        for r in rows:
            if a.dump in r["name"] or a.dump in r["mangled"]:
                print("\n== %s ==" % r["name"])
                for k, t in enumerate(kernel_body(asm, r["mangled"])):
                    if re.match(r"(global_|flat_|buffer_|scratch_|s_waitcnt|s_cbranch|\.LBB|s_barrier|s_branch)", t):
                        print("%5d  %s" % (k, t[:110]))
                break
        else:
            sys.exit("no kernel matches --dump")


if __name__ == "__main__":
    main()
