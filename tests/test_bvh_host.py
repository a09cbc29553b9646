"""Host-built hierarchy for the occlusion rays (csrc/host_bvh.hpp) through the C ABI, no GPU: structural invariants
(every triangle in exactly one leaf, child boxes enclose their subtrees, depth below the traversal stack) and a
pure-Python traversal that must find the same first-hit answers as the all-triangles loop."""
import ctypes as C

import numpy as np
import pytest

f32 = np.float32
EMPTY = -2 ** 31


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as entry
    entry.build_hip()
    from city2ba_amd import _lib as L
    return L


def build(lib, tri):
    tri = np.ascontiguousarray(tri, dtype=f32).reshape(-1, 9)
    h = C.c_void_p()
    lib.check(lib.lib().c2b_bvh_build(tri.ctypes.data_as(C.c_void_p), len(tri), C.byref(h)))
    nn, ns, depth = C.c_int64(), C.c_int64(), C.c_int()
    lib.check(lib.lib().c2b_bvh_sizes(h, C.byref(nn), C.byref(ns), C.byref(depth)))
    nodes = np.zeros((nn.value, 16), dtype=f32)
    tris = np.zeros((max(ns.value, 1), 12), dtype=f32)
    order = np.zeros(ns.value, dtype=np.uint32)
    lib.check(lib.lib().c2b_bvh_copy(h, nodes.ctypes.data_as(C.c_void_p), tris.ctypes.data_as(C.c_void_p),
                                     order.ctypes.data_as(C.c_void_p)))
    lib.lib().c2b_bvh_free(h)
    return nodes, tris[:ns.value], order, depth.value


def children(nodes, i):
    c = nodes[i].view(np.int32)[12:14]
    return [(nodes[i, 0:3], nodes[i, 3:6], int(c[0])), (nodes[i, 6:9], nodes[i, 9:12], int(c[1]))]


def leaf_range(code):
    v = ~code & 0xFFFFFFFF
    return v >> 3, (v & 7) + 1


def random_mesh(seed, n):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-50, 50, (n, 3))
    size = rng.choice([0.2, 1.0, 8.0], n, p=[0.6, 0.3, 0.1])[:, None]      # mixed scales, like a city mesh
    return np.concatenate([a, a + rng.normal(0, 1, (n, 3)) * size, a + rng.normal(0, 1, (n, 3)) * size], axis=1).astype(f32)


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 37, 1000, 20000])
def test_structure(lib, n):
    tri = random_mesh(n + 1, n)
    nodes, slots, order, depth = build(lib, tri)
    assert sorted(order.tolist()) == list(range(n))                        # a permutation: each triangle once
    assert depth < 64 and len(nodes) >= 1
    if n:
        t = tri[order]
        assert np.array_equal(slots[:, 0:3], t[:, 0:3])
        assert np.array_equal(slots[:, 3:6], t[:, 3:6] - t[:, 0:3])        # float32 edges, as the kernel forms them
        assert np.array_equal(slots[:, 6:9], t[:, 6:9] - t[:, 0:3])
    seen = np.zeros(n, dtype=int)

    def walk(i, d):
        """returns (lo, hi) of everything under node i"""
        lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
        for blo, bhi, c in children(nodes, i):
            if c == EMPTY:
                continue
            if c >= 0:
                slo, shi = walk(c, d + 1)
            else:
                first, cnt = leaf_range(c)
                assert 1 <= cnt <= 4 and first + cnt <= n
                seen[first:first + cnt] += 1
                v = tri[order[first:first + cnt]].reshape(-1, 3)
                slo, shi = v.min(0), v.max(0)
            assert np.all(blo <= slo) and np.all(bhi >= shi)               # child box encloses its subtree
            assert np.all(slo - blo < 1e-3) and np.all(bhi - shi < 1e-3)   # ... tightly (margin = a few ulps)
            lo, hi = np.minimum(lo, slo), np.maximum(hi, shi)
        assert d <= depth
        return lo, hi

    walk(0, 0)
    assert np.all(seen == 1)
    if n > 100:
        assert len(nodes) < n                                              # leaves hold several triangles
        assert depth <= 3 * int(np.log2(n))                                # SAH on a well-spread mesh stays shallow


def test_degenerate_meshes(lib):
    # all centroids identical / all triangles identical / collinear centroids: median splits must still terminate
    one = np.tile(np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], dtype=f32), (5000, 1))
    nodes, slots, order, depth = build(lib, one)
    assert sorted(order.tolist()) == list(range(5000)) and depth <= 12
    line = one.copy()
    line[:, [0, 3, 6]] += np.arange(5000, dtype=f32)[:, None]
    nodes, slots, order, depth = build(lib, line)
    assert depth < 40
    # very unbalanced scales: SAH depth cap, then medians
    rng = np.random.default_rng(3)
    a = (rng.uniform(0, 1, (4000, 3)) ** 12 * 1e4).astype(f32)
    tri = np.concatenate([a, a + f32(1e-3), a - f32(1e-3)], axis=1)
    assert build(lib, tri)[3] < 64
    bad = one[:3].copy()
    bad[1, 4] = np.nan
    with pytest.raises(lib.City2baError, match="not finite"):
        build(lib, bad)


def mt_hit(o, d, tfar, slot):
    """float32 Moeller-Trumbore of the kernels on a (v0, e1, e2) row"""
    v0, e1, e2 = slot[0:3], slot[3:6], slot[6:9]
    with np.errstate(all="ignore"):
        px = d[1] * e2[2] - d[2] * e2[1]
        py = d[2] * e2[0] - d[0] * e2[2]
        pz = d[0] * e2[1] - d[1] * e2[0]
        det = (e1[0] * px + e1[1] * py) + e1[2] * pz
        if det == 0:
            return False
        idet = f32(1.0) / det
        tx, ty, tz = o[0] - v0[0], o[1] - v0[1], o[2] - v0[2]
        u = ((tx * px + ty * py) + tz * pz) * idet
        if u < 0 or u > 1:
            return False
        qx = ty * e1[2] - tz * e1[1]
        qy = tz * e1[0] - tx * e1[2]
        qz = tx * e1[1] - ty * e1[0]
        w = ((d[0] * qx + d[1] * qy) + d[2] * qz) * idet
        if w < 0 or u + w > 1:
            return False
        th = ((e2[0] * qx + e2[1] * qy) + e2[2] * qz) * idet
        return bool(th > 0 and th <= tfar)


def box_hit(lo, hi, o, inv, tfar):
    with np.errstate(all="ignore"):
        a, b = (lo - o) * inv, (hi - o) * inv
        tn = np.nanmax(np.fmin(a, b)) if not np.all(np.isnan(np.fmin(a, b))) else -np.inf
        tf = np.nanmin(np.fmax(a, b)) if not np.all(np.isnan(np.fmax(a, b))) else np.inf
    return tn <= tf * f32(1.00001) + f32(1e-30) and tf >= 0 and tn <= tfar


def test_python_traversal_equals_all_triangles(lib):
    tri = random_mesh(11, 3000)
    tri[:50, [1, 4, 7]] = 0.0                                              # some axis-aligned triangles (zero-extent boxes)
    nodes, slots, order, depth = build(lib, tri)
    rng = np.random.default_rng(5)
    n_rays, hits = 400, 0
    for r in range(n_rays):
        o = rng.uniform(-50, 50, 3).astype(f32)
        d = rng.normal(0, 1, 3)
        if r % 7 == 0:
            d[rng.integers(3)] = 0.0                                       # axis-parallel rays: 1/d = inf in the slab test
        d = (d / np.linalg.norm(d)).astype(f32)
        tfar = f32(rng.uniform(5, 80))
        brute = any(mt_hit(o, d, tfar, s) for s in slots)
        with np.errstate(all="ignore"):
            inv = f32(1.0) / d
        stack, found = [0], False
        while stack and not found:
            i = stack.pop()
            for blo, bhi, c in children(nodes, i):
                if c == EMPTY or not box_hit(blo, bhi, o, inv, tfar):
                    continue
                if c >= 0:
                    stack.append(c)
                else:
                    first, cnt = leaf_range(c)
                    if any(mt_hit(o, d, tfar, slots[s]) for s in range(first, first + cnt)):
                        found = True
        assert found == brute
        hits += brute
    assert 20 < hits < n_rays - 20
