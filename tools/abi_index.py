#!/usr/bin/env python3
"""The index of the boundary's entry points by header and level (include/city2ba_hip.h + city2ba_hip_host.h +
city2ba_hip_experimental.h), as the main header's top comment carries it.
    python tools/abi_index.py            # print the index block
    python tools/abi_index.py --write    # replace the block between the two marker lines in the header
tests/test_abi.py checks that the block in the header is what this prints (every exported symbol listed exactly once)."""
import os
import re
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "city2ba_hip.h")
HEADERS = [HEADER, os.path.join(ROOT, "include", "city2ba_hip_host.h"), os.path.join(ROOT, "include", "city2ba_hip_experimental.h")]


def all_headers_text():
    """the three headers, one after the other (what `declared in the header` means to the tests)"""
    return "\n".join(open(h).read() for h in HEADERS)
BEGIN, END = " * ---- index of entry points by level (tools/abi_index.py) ----", " * ---- end of index ----"

# (title, first line of the header that belongs to the group) in header order; a group runs to the next one's start
GROUPS = [
    ("library", r"^const char \*c2b_version"),
    ("Level 0: workspace, camera records, points, rows", r"^int64_t c2b_workspace_bytes"),
    ("Level 0: per-observation passes (cam_idx and row-structure forms)", r"^int c2b_project\("),
    ("Level 0: Jacobian output sets and calibration", r"^typedef struct c2b_jacobian_outputs"),
    ("Level 0: visibility sweeps and occlusion", r"^int c2b_visibility_pairs"),
    ("Level 0: statistics", r"^int c2b_stats\("),
    ("collectives (RCCL) and sharded Level-0 forms", r"^typedef struct c2b_comm"),
    ("Level 0: noise", r"^int c2b_add_drift\("),
    ("the shard map", r"^int c2b_partition_cameras"),
    ("Level 1: a resident BAProblem", r"^typedef struct c2b_problem c2b_problem"),
    ("Level 1: one shard of a larger problem", r"^int c2b_problem_set_shard"),
]


def protos(text):
    body = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)
    out = []
    for m in re.finditer(r"(?m)^\s*(?:const\s+)?\w+\s*\**\s*(c2b_\w+)\s*\(", body):
        out.append((body.count("\n", 0, m.start(1)), m.group(1)))
    return out


def index_block():
    text = open(HEADER).read()
    lines = text.split("\n")
    starts = []
    for title, pat in GROUPS:
        hit = next((i for i, ln in enumerate(lines) if re.search(pat, ln)), None)
        if hit is not None:
            starts.append((hit, title))
    starts.sort()
    groups = {t: [] for _, t in starts}
    for ln, name in protos(text):
        owner = None
        for s, t in starts:
            if s <= ln:
                owner = t
        groups[owner or starts[0][1]].append(name)
    out = [BEGIN]

    def emit(title, names):
        short = [n[4:] for n in names]                      # without the c2b_ prefix
        out.append(" *   %s (%d):" % (title, len(names)))
        out.extend(" *     " + b for b in textwrap.wrap(", ".join(short), width=112))

    total = 0
    for _, t in starts:
        if groups[t]:
            total += len(groups[t])
            emit(t, groups[t])
    out.append(" *   -- city2ba_hip.h: %d entry points --" % total)
    for h, title in ((HEADERS[1], "city2ba_hip_host.h: host-side rows (CPU; never touch the GPU)"),
                     (HEADERS[2], "city2ba_hip_experimental.h: diagnostics, calibration, f32 extension")):
        names = [n for _, n in protos(open(h).read())]
        total += len(names)
        emit(title, names)
    out.append(" *   (%d entry points in all; names above without their c2b_ prefix)" % total)
    out.append(END)
    return "\n".join(out)


if __name__ == "__main__":
    blk = index_block()
    if "--write" in sys.argv:
        text = open(HEADER).read()
        if BEGIN in text:
            a, b = text.index(BEGIN), text.index(END) + len(END)
            text = text[:a] + blk + text[b:]
        else:
            anchor = " * Every function returns C2B_OK or a negative status"
            text = text.replace(anchor, blk + "\n *\n" + anchor, 1)
        open(HEADER, "w").write(text)
    print(blk)
