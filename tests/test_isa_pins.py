"""What the in-kernel folds rely on is gfx950 ISA behaviour, not the HIP memory model (kernels.hpp: ticket_fold):
the partial is published by an agent-scope write-through store (`sc1`) that is DRAINED (`s_waitcnt vmcnt(0)`) before
the arrival counter's `global_atomic_add` is issued, and the last arriver acquires with `buffer_inv sc1`.  Nothing in
the language pins that code generation, so this test does: it cross-compiles the product library with --save-temps
(hipcc needs no GPU) and asserts, for every shipped kernel, no scratch, no register spills and no FLAT memory
instruction, and for every kernel that folds, exactly that instruction order."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# the one kernel allowed to use scratch: its per-lane traversal stack (64 node ids) is indexed dynamically
SCRATCH_ALLOWED = {"k_occlusion_bvh": 272}
# r06: the generators' two one-camera-per-wave kernels keep the camera, the cell grid and their many arguments in scalar
# registers; with glibc's pow inlined (csrc/pow4_libm.hpp: nested special-case branches = live lane masks) the allocator
# parks two scalars in VGPR lanes (v_writelane / v_readlane: no memory, no scratch).  Everything else: none.
SGPR_SPILL_ALLOWED = {"k_visibility_dense<true>": 2, "k_cells_visibility<false>": 2}
FOLDING = ("k_observations<1,", "k_observations<3,", "k_observations<4,", "k_residual_jacobian_l<0, true", "k_residual_jacobian_l<1, true",
           "k_residual_jacobian_l<2, true", "k_stats_pass1<", "k_stats_pass2<")


@pytest.fixture(scope="module")
def isa():
    import isa_report as I
    asm = I.compile_asm()
    return I, asm, I.kernel_table(asm)


def test_every_shipped_kernel_has_no_scratch_no_spills_no_flat(isa):
    I, asm, rows = isa
    assert len(rows) > 60
    assert any(r["lds"] > 0 for r in rows), "the metadata parse lost the LDS sizes"
    for r in rows:
        short = r["name"].split("<")[0]
        assert r["scratch"] == SCRATCH_ALLOWED.get(short, 0), (r["name"], r["scratch"])
        assert r["vgpr_spill"] == 0 and r["sgpr_spill"] <= SGPR_SPILL_ALLOWED.get(r["name"], 0), r
        body = I.kernel_body(asm, r["mangled"])
        flat = [t for t in body if t.startswith("flat_")]
        assert not flat, (r["name"], flat[:3])
        if short not in SCRATCH_ALLOWED:
            assert not [t for t in body if t.startswith("scratch_")], r["name"]


def test_headline_kernel_resources(isa):
    _, _, rows = isa
    jac = [r for r in rows if r["name"].startswith("k_residual_jacobian_l<2, true, 8, true, 2, 4, true, true, 3>")]
    assert len(jac) == 1
    assert jac[0]["vgpr"] <= 128 and jac[0]["lds"] <= 80 * 1024       # 4 waves per SIMD, two workgroups per CU


def test_fold_publish_and_acquire_instruction_order(isa):
    I, asm, rows = isa
    folding = [r for r in rows if r["name"].startswith(FOLDING)]
    assert len(folding) >= 20
    for r in folding:
        body = I.kernel_body(asm, r["mangled"])
        adds = [k for k, t in enumerate(body) if t.startswith("global_atomic_add")]
        assert len(adds) == 2, (r["name"], "leaf and top arrival counters")
        # publish: sc1 (write-through) stores of the partial, then a full drain, then the first arrival atomic
        drain = max(k for k, t in enumerate(body[:adds[0]]) if re.match(r"s_waitcnt vmcnt\(0\)", t))
        stores = [k for k, t in enumerate(body[:drain]) if re.match(r"global_store_dwordx2 .* sc1$", t)]
        assert stores, (r["name"], "the workgroup's partial must leave through an sc1 store before the drain")
        between = body[stores[-1] + 1:adds[0]]
        assert not [t for t in between if t.startswith(("global_store", "global_load", "buffer_"))], (r["name"], between)
        # acquire: the last arriver invalidates before it reads the other workgroups' partials
        inv = [k for k, t in enumerate(body) if t == "buffer_inv sc1"]
        assert inv and inv[0] > adds[1], r["name"]
        loads_after = [k for k, t in enumerate(body) if k > inv[0] and t.startswith("global_load")]
        assert loads_after, r["name"]
        # counters are reset with sc1 stores by whoever completes them
        assert len([t for t in body if re.match(r"global_store_dword .* sc1$", t)]) >= 2, r["name"]


def test_other_offload_targets_are_refused_at_compile_time(tmp_path):
    """the publish of the in-kernel folds is validated for gfx942 / gfx950 only (kernels.hpp): building for anything else
    must stop with that message instead of producing a library whose sums may be stale"""
    import subprocess
    src = os.path.join(ROOT, "city2ba_amd", "csrc", "capi.hip")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "--offload-arch=gfx90a", "-ffp-contract=off", "-fPIC", "-std=c++17", "-c", src,
                        "-o", str(tmp_path / "x.o")], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "validated on gfx942 / gfx950 only" in r.stderr
