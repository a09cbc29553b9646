"""BASELINE.json configs[4] at FULL size: "generate from .obj 10k cams / 1M pts then noise.rs drift+rotation kernels,
f32 path, 1 MI355X" -- 10 000 cameras x 1 000 000 points = 1e10 (camera, point) pairs through the mesh generator's
predicate sweep (src/generate.rs:446-469, distortion on: k1, k2 != 0), and the entity-noise kernels in f32 over the
same 1 010 000 entities.  The oracle cannot sweep 1e10 pairs, so:
  * the dense sweep's CSR output is checked against the independent pair-list kernel on 48 sampled cameras x all
    points (indices and uv bits), and against the CPU oracle (libm pow, like the device) on 3 cameras x all points;
  * structural properties over everything: row_ptr monotone, per-camera point indices strictly ascending (the
    reference's push order), every kept uv inside [-1, 1]^2;
  * f32 drift / Gaussian noise of all 1.01 M entities against the f64 oracle at an f32 tolerance."""
import ctypes as C

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

N_CAM, N_PTS, EXTENT, MAX_DIST = 10_000, 1_000_000, 300.0, 10.0


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as entry
    entry.build()
    import torch
    from city2ba_amd import _lib as L
    from city2ba_amd import device as D
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(20245)
    w = rng.uniform(-np.pi, np.pi, (N_CAM, 3)) * rng.uniform(0, 1, (N_CAM, 1))
    centre = np.column_stack([rng.uniform(0, EXTENT, N_CAM), rng.uniform(1, 3, N_CAM), rng.uniform(0, EXTENT, N_CAM)])
    bal9 = np.column_stack([w, np.zeros((N_CAM, 3)), rng.uniform(0.8, 1.2, N_CAM), rng.uniform(-1e-2, 1e-2, (N_CAM, 2))])
    cams = np.stack([np.asarray(O.camera_from_bal(b)).reshape(15) for b in bal9])
    R = cams[:, :9].reshape(-1, 3, 3).transpose(0, 2, 1)                       # col-major -> row-major
    cams[:, 9:12] = -np.einsum("nij,nj->ni", R, centre)                        # t = -R c
    pts = np.column_stack([rng.uniform(0, EXTENT, N_PTS), rng.uniform(0, 6, N_PTS), rng.uniform(0, EXTENT, N_PTS)])
    cam15 = torch.from_numpy(cams).to(dev)
    camblk = D.cameras_prepare_state(cam15)
    pts4 = D.points_pad(torch.from_numpy(np.ascontiguousarray(pts)).to(dev))
    return dict(torch=torch, L=L, D=D, dev=dev, cams=cams, pts=pts, cam15=cam15, camblk=camblk, pts4=pts4)


def test_dense_sweep_1e10_pairs(env):
    torch, L, D, dev = env["torch"], env["L"], env["D"], env["dev"]
    lib = L.lib()
    camblk, pts4 = env["camblk"], env["pts4"]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
    n_tiles = lib.c2b_visibility_dense_tiles(N_PTS)
    counts = torch.empty(N_CAM * n_tiles, dtype=torch.int32, device=dev)
    tot = torch.empty(N_CAM + 1, dtype=torch.int64, device=dev)
    row = torch.empty(N_CAM + 1, dtype=torch.int64, device=dev)
    L.check(lib.c2b_visibility_dense_count(p(camblk), N_CAM, p(pts4), N_PTS, MAX_DIST, p(counts), p(tot), p(row), st))
    torch.cuda.synchronize()
    n_obs = int(row[-1].item())
    assert 100_000 < n_obs < 50_000_000
    pt_idx = torch.empty(n_obs, dtype=torch.int32, device=dev)
    uv = torch.full((n_obs, 2), float("nan"), dtype=torch.float64, device=dev)
    L.check(lib.c2b_visibility_dense_fill(p(camblk), N_CAM, p(pts4), N_PTS, MAX_DIST, p(counts), p(row), p(pt_idx), p(uv), st))
    torch.cuda.synchronize()

    # structure, over everything
    r = row.cpu().numpy()
    assert r[0] == 0 and np.all(np.diff(r) >= 0)
    pi_h = pt_idx.cpu().numpy().astype(np.int64)
    starts = np.zeros(n_obs, dtype=bool)
    starts[r[:-1][np.diff(r) > 0]] = True
    asc = np.diff(pi_h) > 0
    assert np.all(asc | starts[1:]), "per-camera point indices must be strictly ascending (push order of the reference)"
    assert bool(torch.isfinite(uv).all()) and float(uv.abs().max()) <= 1.0

    # the pair-list kernel on sampled cameras x ALL points: same kept indices, same uv bits
    rng = np.random.default_rng(5)
    sample = np.unique(np.concatenate([[0, N_CAM - 1], rng.integers(0, N_CAM, 46)]))
    all_pts = torch.arange(N_PTS, dtype=torch.int32, device=dev)
    uv_p = torch.empty((N_PTS, 2), dtype=torch.float64, device=dev)
    keep = torch.empty(N_PTS, dtype=torch.uint8, device=dev)
    kept_total = 0
    for c in sample:
        ci = torch.full((N_PTS,), int(c), dtype=torch.int32, device=dev)
        D.visibility_pairs(camblk, pts4, ci, all_pts, MAX_DIST, uv_p, keep)
        k = keep.bool()
        a, b = int(r[c]), int(r[c + 1])
        assert torch.equal(all_pts[k], pt_idx[a:b]), c
        assert torch.equal(uv_p[k].view(torch.int64), uv[a:b].view(torch.int64)), c
        kept_total += b - a
    assert kept_total > 100

    # the CPU oracle (|p|^4 by libm's pow, like the reference and the device) on 3 cameras x all points
    pi_all = np.arange(N_PTS, dtype=np.uint32)
    for c in sample[:3]:
        uv_o, keep_o = O.visibility_pairs(env["cams"][c:c + 1], env["pts"], np.zeros(N_PTS, dtype=np.uint32), pi_all, MAX_DIST)
        a, b = int(r[c]), int(r[c + 1])
        assert np.array_equal(pi_all[keep_o == 1], pi_h[a:b].astype(np.uint32))
        assert np.array_equal(uv_o[keep_o == 1].view(np.uint64), uv[a:b].cpu().numpy().view(np.uint64))


def test_f32_noise_kernels_on_a_million_entities(env):
    torch, D, dev = env["torch"], env["D"], env["dev"]
    cams, pts = env["cams"], env["pts"]
    ws = D.workspace(0, dev)
    c32, p32 = D.to_f32(env["cam15"]), D.to_f32(env["pts4"])
    st = D.stats_f32(c32, p32, ws)
    st_h = st.cpu().numpy()
    assert np.allclose(st_h[0:3], O.mean(cams, pts), rtol=1e-5, atol=1e-4)
    assert np.allclose(st_h[3:6], O.std(cams, pts), rtol=1e-5)
    _, idx = O.drift_origin(cams, pts)
    assert int(st_h[18]) == idx

    def back(c, p):
        return D.to_f64(c).cpu().numpy(), D.to_f64(p).cpu().numpy()[:, :3]

    # drift (src/noise.rs:68-116): displacement ~ strength * gamma * d^2, d up to ~400 here
    d = np.array([0.3, -0.5, 0.8])
    D.add_drift_f32(c32, p32, st, 2e-6, 1e-5, 0.2, d, seed=42)
    got_c, got_p = back(c32, p32)
    want_c, want_p = O.add_drift(cams, pts, 2e-6, 1e-5, 0.2, d, seed=42)
    assert np.max(np.abs(want_p - pts)) > 0.05                                  # it moved things
    assert np.max(np.abs(got_p - want_p)) < 4e-6 * np.abs(want_p).max()
    assert np.max(np.abs(got_c - want_c)) < 1e-5 * max(1.0, np.abs(want_c).max())

    # Gaussian camera / point noise (src/noise.rs:129-150) on the same million entities
    c32, p32 = D.to_f32(env["cam15"]), D.to_f32(env["pts4"])
    D.add_noise_entities_f32(c32, p32, st, 1e-3, 0.05, 0.1, seed=99)
    got_c, got_p = back(c32, p32)
    want_c, want_p, _ = O.add_noise(cams, pts, np.zeros((0, 2)), 1e-3, 0.05, 0.1, 0.0, seed=99)
    assert np.max(np.abs(got_p - want_p)) < 4e-6 * np.abs(want_p).max()
    assert np.max(np.abs(got_c - want_c)) < 2e-5 * max(1.0, np.abs(want_c).max())
    R = got_c[:, :9].reshape(-1, 3, 3)
    assert np.max(np.abs(np.einsum("nij,nkj->nik", R, R) - np.eye(3))) < 1e-5    # still rotations
    assert np.std(got_p - pts) > 0.03
