// obs_pipeline.hpp -- persistent, software-pipelined per-observation kernels for gfx950 (included by capi.hip).
//
// Why: a wave of the one-shot kernels (kernels.hpp) lives ~10 us and spends the first part of that in a chain of two
// dependent memory round trips (index -> point gather / camera rows) during which it has nothing to store; the
// kernel's 208 B/observation of stores only flow while enough *other* waves happen to be in their store phase.  On
// gfx950 every vector-memory operation of a wave -- loads AND stores -- retires through one in-order counter
// (vmcnt), so a wave that simply loops over tiles would make each tile's loads queue behind the previous tile's 13
// stores.  The cure is issue ORDER: inside one iteration the loads of the *following* tiles are issued BEFORE the
// current tile's stores, so waiting for them (vmcnt(N) with N >= the stores issued since) never waits for a store:
//
//     iteration i :  issue index/uv loads of tile i+2          (3 VM loads)
//                    issue point gather + camera rows of i+1   (3 VM loads; needs the indices of i+1, issued a
//                                                               whole iteration ago)
//                    camera rows of tile i: registers -> wave-private LDS tile
//                    arithmetic of tile i, transposition through the wave-private slab, 13 x 1-KiB stores
//
// The three register sets rotate by a 3x manual unroll (A,B,C -> B,C,A -> C,A,B): no register copies, static
// vmcnt counts.  All tiles in the loop are full (64 observations, unconditional stores); the at most one ragged
// tile of the whole launch is done by one wave after its loop with predicated stores.  The grid is persistent
// (occupancy x CUs workgroups); XCD x streams the x-th eighth of the tile list, its waves walk it with a stride of
// (waves per XCD), so at any time an XCD writes one moving window of the output.
//
// Error reduce: every lane accumulates its |r|^norm over its tiles, then ticket_fold (kernels.hpp).
//
// MEASURED (round 2, tools/tune_jac.py / tune_obs.py, --blocks 128): this form LOSES to the one-shot kernels of
// kernels.hpp -- 780 us against 718 us for the Jacobian (no error reduce), 145 us against 127 us for project -- and is
// therefore compiled into the tuning library only (-DC2B_TUNE).  Why: vmcnt retires in order, so the data of a
// gather issued after tile i-1's stores is only *visible* once those stores are acknowledged; a one-shot wave never
// waits for a store acknowledgement at all (it ends), a persistent one does every iteration.  And with one tile of
// look-ahead per dependent level a wave completes one tile per memory latency, which 12-24 resident waves per CU
// do not turn into more bytes in flight than 32 short-lived ones.
#pragma once
#include "kernels.hpp"

namespace c2b {

// Cameras staged per 64-observation tile: as many as one 16-byte chunk per lane covers (HOT doubles per camera):
// 5 for the Jacobian's 24 hot doubles, 8 for the 16 of project / error, 4 for the 28 of the visibility predicate.
// On the synthetic grid (~29 observations per camera) a tile spans 3-5 cameras; a tile that spans more takes the
// slower global-read path (still correct).
constexpr int pipe_cam_w(int hot) { return 64 / (hot / 2); }

struct PipeSet {
    uint32_t ci, pi;
    double2 ob;
    double4 X;
    d2_t cs;                           // this lane's 16-byte chunk of the tile's camera rows
    uint32_t c_first, n_staged;        // wave-uniform
};

// ---- pipeline stages ----------------------------------------------------------------------------------------------
template <bool WITH_UV>
C2B_DEV void pipe_issue_idx(PipeSet &S, int tile, int lane, int n, const uint32_t *__restrict__ cam_idx,
                            const uint32_t *__restrict__ pt_idx, const double2 *__restrict__ uv_obs) {
    int o = tile * 64 + lane;                          // 32-bit: a launch holds < 2^31 observations (launchers check)
    o = o < n ? o : n - 1;                             // prefetches past the end read the last observation
    S.ci = cam_idx[o];
    S.pi = pt_idx[o];
    if (WITH_UV) S.ob = uv_obs[o];
}

template <int HOT>
C2B_DEV void pipe_issue_gather(PipeSet &S, int lane, const double *__restrict__ camblk,
                               const double4 *__restrict__ pts4) {
    S.X = pts4[S.pi];
    const uint32_t c_first = __builtin_amdgcn_readfirstlane(S.ci);
    const uint32_t c_last = __builtin_amdgcn_readlane(S.ci, 63);
    uint32_t ns = c_last >= c_first ? c_last - c_first + 1 : 1;
    if (ns > (uint32_t)pipe_cam_w(HOT)) ns = pipe_cam_w(HOT);
    S.c_first = c_first;
    S.n_staged = ns;
    const int nch = (int)ns * (HOT / 2);
    const int ch = lane < nch ? lane : nch - 1;        // always load (static VM count); the LDS write is predicated
    const int k = ch / (HOT / 2), j = ch % (HOT / 2);
    S.cs = *reinterpret_cast<const d2_t *>(camblk + cam_chunk_at((int64_t)(c_first + k), j));
}

template <int HOT>
C2B_DEV void pipe_stage_cams(const PipeSet &S, int lane, double *sCam) {
    const int nch = (int)S.n_staged * (HOT / 2);
    if (lane < nch) *reinterpret_cast<d2_t *>(sCam + 2 * lane) = S.cs;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// One observation's projection + Jacobian with the 2x9 block written straight into this lane's 144-byte row of the
// wave's LDS slab as its entries become available (nothing but the 2x3 block and the residual stays in registers).
// Same arithmetic, operation for operation, as jacobian_obs (kernels.hpp).
typedef __attribute__((address_space(3))) d2_t *lds_d2ptr;
template <typename P>
C2B_DEV void jacobian_emit(P cam, const double4 X, const double2 ob, double &r0, double &r1, lds_d2ptr row,
                           double jp[6]) {
    const Proj p = project_obs(cam, X.x, X.y, X.z);
    r0 = p.u - ob.x; r1 = p.v - ob.y;
    const double f = cam[12], k1 = cam[13], k2 = cam[14];
    double iz = __builtin_amdgcn_rcp(p.qz);
    iz = fma(fma(-p.qz, iz, 1.0), iz, iz);
    iz = fma(fma(-p.qz, iz, 1.0), iz, iz);
    const double s = -f * iz;
    const double c = fma(4.0 * k2, p.n, 2.0 * k1);
    const double cpx = c * p.px;
    const double B00 = fma(cpx, p.px, p.rad), B01 = cpx * p.py, B11 = fma(c * p.py, p.py, p.rad);
    const double g = fma(c, p.n, p.rad);
    const double a00 = s * B00, a01 = s * B01, a02 = s * p.px * g;
    const double a10 = s * B01, a11 = s * B11, a12 = s * p.py * g;
    const double fn = f * p.n, fnn = fn * p.n;
    d2_t v;
    v.x = a01; v.y = a02; row[2] = v;                                   // jc[4], jc[5]
    v.x = a10; v.y = a11; row[6] = v;                                   // jc[12], jc[13]
    v.x = p.rad * p.px; v.y = fn * p.px; row[3] = v;                    // jc[6], jc[7]
    v.x = fn * p.py; v.y = fnn * p.py; row[8] = v;                      // jc[16], jc[17]
    const double yx = p.qx - cam[9], yy = p.qy - cam[10], yz = p.qz - cam[11];
    const double v0x = fma(yy, a02, -yz * a01), v0y = fma(yz, a00, -yx * a02), v0z = fma(yx, a01, -yy * a00);
    const double v1x = fma(yy, a12, -yz * a11), v1y = fma(yz, a10, -yx * a12), v1z = fma(yx, a11, -yy * a10);
    const P Jl = cam + kJl;
    double w0[3], w1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        w0[j] = fma(v0z, Jl[6 + j], fma(v0y, Jl[3 + j], v0x * Jl[j]));
        w1[j] = fma(v1z, Jl[6 + j], fma(v1y, Jl[3 + j], v1x * Jl[j]));
    }
    v.x = w0[0]; v.y = w0[1]; row[0] = v;                               // jc[0], jc[1]
    v.x = w0[2]; v.y = a00; row[1] = v;                                 // jc[2], jc[3]
    v.x = fnn * p.px; v.y = w1[0]; row[4] = v;                          // jc[8], jc[9]
    v.x = w1[1]; v.y = w1[2]; row[5] = v;                               // jc[10], jc[11]
    v.x = a12; v.y = p.rad * p.py; row[7] = v;                          // jc[14], jc[15]
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        jp[j] = fma(a02, cam[6 + j], fma(a01, cam[3 + j], a00 * cam[j]));
        jp[3 + j] = fma(a12, cam[6 + j], fma(a11, cam[3 + j], a10 * cam[j]));
    }
}

// arithmetic + transposition + stores of one 64-observation tile whose operands are in S / sCam.  The slab holds
// the whole tile's 2x9 blocks (64 x 144 B = 9 KiB, lane stride 144 B: conflict-free for ds_write_b128), is read
// back linearly and leaves as nine 1-KiB stores of whole 128-B lines; the 2x3 blocks then reuse its first 3 KiB.
template <int NK, bool WITH_ERR, bool NT, bool PARTIAL>
C2B_DEV void pipe_jacobian_tile(const PipeSet &S, int tile, int n_wave, int lane,
                                const double *__restrict__ camblk, const double *sCam, char *slab, double norm,
                                double2 *__restrict__ r_out, double *__restrict__ Jc, double *__restrict__ Jp,
                                double &eacc) {
    const int64_t wave0 = (int64_t)tile * 64;
    const bool valid = !PARTIAL || lane < n_wave;
    double r0 = 0.0, r1 = 0.0, jp[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const lds_d2ptr row = (lds_d2ptr)(slab + lane * 144);
    // One pass serves every lane whose camera is among the staged ones -- on camera-major input that is all of them,
    // and the loop below runs once.  Lanes left over (unsorted input, or more cameras in the tile than were staged)
    // are served one camera at a time: restage that camera, run the same LDS-only arithmetic for its lanes.  Only
    // ds_read code exists for the arithmetic, so nothing here can turn into FLAT loads.
    uint32_t c_first = S.c_first, n_staged = S.n_staged;
    uint64_t todo = __builtin_amdgcn_ballot_w64(valid);
    for (;;) {
        const uint32_t local = S.ci - c_first;
        const bool in = ((todo >> lane) & 1ull) != 0 && local < n_staged;
        if (in) jacobian_emit((lds_cptr)sCam + local * kCamHot, S.X, S.ob, r0, r1, row, jp);
        todo &= ~__builtin_amdgcn_ballot_w64(in);
        if (todo == 0) break;
        c_first = __builtin_amdgcn_readlane(S.ci, (int)__builtin_ctzll(todo));
        n_staged = 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < kCamHot / 2)
            *reinterpret_cast<d2_t *>(const_cast<double *>(sCam) + 2 * lane) =
                *reinterpret_cast<const d2_t *>(camblk + cam_chunk_at((int64_t)c_first, lane));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (valid) store16<NT>(reinterpret_cast<char *>(r_out + wave0 + lane), make_double2(r0, r1));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
        char *dst = reinterpret_cast<char *>(Jc) + wave0 * 144;
        const int bytes = n_wave * 144;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int off = (k * 64 + lane) * 16;
            if (!PARTIAL || off < bytes) store16<NT>(dst + off, *reinterpret_cast<const double2 *>(slab + off));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
        double2 *w = reinterpret_cast<double2 *>(slab + lane * 48);
#pragma unroll
        for (int k = 0; k < 3; ++k) w[k] = make_double2(jp[2 * k], jp[2 * k + 1]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        char *dst = reinterpret_cast<char *>(Jp) + wave0 * 48;
        const int bytes = n_wave * 48;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int off = (k * 64 + lane) * 16;
            if (!PARTIAL || off < bytes) store16<NT>(dst + off, *reinterpret_cast<const double2 *>(slab + off));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (WITH_ERR) eacc += valid ? abs_pow_k<NK>(r0, norm) + abs_pow_k<NK>(r1, norm) : 0.0;
}

// this wave's share of the full tiles: XCD (blockIdx & 7) owns a contiguous eighth, its waves stride through it
struct PipeRange { int first, stride, count; };
C2B_DEV PipeRange pipe_range(int full_tiles, int wpb, int wave) {
    const int q = full_tiles >> 3, rr = full_tiles & 7;
    const int xcd = blockIdx.x & 7;
    const int xs = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
    const int xlen = q + (xcd < rr ? 1 : 0);
    const int wq = (int)(blockIdx.x >> 3) * wpb + wave;
    PipeRange R;
    R.stride = (int)(gridDim.x >> 3) * wpb;
    R.first = xs + wq;
    R.count = wq < xlen ? (xlen - wq + R.stride - 1) / R.stride : 0;
    return R;
}

// ---- residual + 2x(9+3) Jacobian (+ fused error sum), persistent pipelined form -----------------------------------
// gridDim.x must be a multiple of 8.  block_part: >= gridDim.x doubles; out_sum may be NULL iff !WITH_ERR.
// MINW = waves per SIMD the register allocation must leave room for (HIP's second __launch_bounds__ argument).
template <int NK, bool WITH_ERR, int WPB, int MINW, bool NT>
__global__ __launch_bounds__(WPB * 64, MINW) void k_residual_jacobian_p(
    const double *__restrict__ camblk, const double4 *__restrict__ pts4, const uint32_t *__restrict__ cam_idx,
    const uint32_t *__restrict__ pt_idx, const double2 *__restrict__ uv_obs, int n, double norm,
    double2 *__restrict__ r_out, double *__restrict__ Jc, double *__restrict__ Jp, double *__restrict__ block_part,
    unsigned *__restrict__ ticket, double *__restrict__ out_sum) {
    constexpr int kSlab = 64 * 144;                       // the whole tile's 2x9 blocks
    constexpr int kCamBytes = 64 * 16;                    // one 16-byte chunk per lane
    __shared__ __attribute__((aligned(16))) char smem[WPB * (kSlab + kCamBytes)];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *slab = smem + wave * (kSlab + kCamBytes);
    double *sCam = reinterpret_cast<double *>(slab + kSlab);
    const PipeRange R = pipe_range(n >> 6, WPB, wave);
    double eacc = 0.0;

#define C2B_PIPE_BODY(CUR, NXT, NN)                                                                          \
    pipe_issue_idx<true>(NN, t + 2 * R.stride, lane, n, cam_idx, pt_idx, uv_obs);                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    pipe_issue_gather<kCamHot>(NXT, lane, camblk, pts4);                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    pipe_stage_cams<kCamHot>(CUR, lane, sCam);                                                                \
    pipe_jacobian_tile<NK, WITH_ERR, NT, false>(CUR, t, 64, lane, camblk, sCam, slab, norm, r_out, Jc, Jp, eacc); \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    t += R.stride;                                                                                            \
    if (++i >= R.count) goto pipe_done;

    if (R.count > 0) {
        PipeSet A, B, C;
        int t = R.first, i = 0;
        pipe_issue_idx<true>(A, t, lane, n, cam_idx, pt_idx, uv_obs);
        pipe_issue_idx<true>(B, t + R.stride, lane, n, cam_idx, pt_idx, uv_obs);
        __builtin_amdgcn_sched_barrier(0);
        pipe_issue_gather<kCamHot>(A, lane, camblk, pts4);
        __builtin_amdgcn_sched_barrier(0);
        // The first body is peeled so that the loop is entered in its steady state: the compiler's wait counts at
        // the loop head merge the entry edge with the back edge, and an entry straight from the prologue (nothing
        // issued after A's gather) would turn the head's waits into "everything older", i.e. the previous tile's
        // stores.
        do {
            C2B_PIPE_BODY(A, B, C)
            for (;;) {
                C2B_PIPE_BODY(B, C, A)
                C2B_PIPE_BODY(C, A, B)
                C2B_PIPE_BODY(A, B, C)
            }
        } while (false);
    pipe_done:;
    }
#undef C2B_PIPE_BODY

    // the ragged last tile of the launch (n % 64 observations): one wave, plain loads, predicated stores
    if ((n & 63) != 0 && blockIdx.x == 0 && wave == 0) {
        PipeSet D;
        const int tile = n >> 6;
        pipe_issue_idx<true>(D, tile, lane, n, cam_idx, pt_idx, uv_obs);
        pipe_issue_gather<kCamHot>(D, lane, camblk, pts4);
        // lanes past the end hold copies of the last observation, so lane 63's camera is the tile's last camera
        pipe_stage_cams<kCamHot>(D, lane, sCam);
        pipe_jacobian_tile<NK, WITH_ERR, NT, true>(D, tile, (int)(n & 63), lane, camblk, sCam, slab, norm, r_out, Jc, Jp,
                                               eacc);
    }

    if (WITH_ERR) {
        const double w = wave_sum(eacc);
        ticket_fold(w, reinterpret_cast<double *>(smem), block_part, ticket, out_sum);
    }
}

// ---- project / error / visibility, persistent pipelined form ------------------------------------------------------
// Same skeleton as the Jacobian kernel with a lighter tile body: 16 doubles of each camera (R, t, intrinsics) staged
// per tile (all 28 in visibility mode, which also needs the centre), one 1-KiB store of uv per tile.
template <int MODE, int NK, bool PARTIAL>
C2B_DEV void pipe_light_tile(const PipeSet &S, int tile, int n_wave, int lane, const double *__restrict__ camblk,
                             const double *sCam, int hot, double norm, double max_dist, double2 *__restrict__ uv_out,
                             uint8_t *__restrict__ keep, double &eacc) {
    const int64_t o = (int64_t)tile * 64 + lane;
    const bool valid = !PARTIAL || lane < n_wave;
    Proj p;
    p.qz = 1.0; p.u = 0.0; p.v = 0.0;
    double gx = 0.0, gy = 0.0, gz = 0.0;
    // one pass on camera-major input; leftover lanes are served one restaged camera at a time (pipe_jacobian_tile)
    uint32_t c_first = S.c_first, n_staged = S.n_staged;
    uint64_t todo = __builtin_amdgcn_ballot_w64(valid);
    for (;;) {
        const uint32_t local = S.ci - c_first;
        const bool in = ((todo >> lane) & 1ull) != 0 && local < n_staged;
        if (in) {
            const lds_cptr cam = (lds_cptr)sCam + local * hot;
            p = project_obs(cam, S.X.x, S.X.y, S.X.z);
            if (MODE == MODE_VISIBILITY) { gx = cam[kCenter]; gy = cam[kCenter + 1]; gz = cam[kCenter + 2]; }
        }
        todo &= ~__builtin_amdgcn_ballot_w64(in);
        if (todo == 0) break;
        c_first = __builtin_amdgcn_readlane(S.ci, (int)__builtin_ctzll(todo));
        n_staged = 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < hot / 2)
            *reinterpret_cast<d2_t *>(const_cast<double *>(sCam) + 2 * lane) =
                *reinterpret_cast<const d2_t *>(camblk + cam_chunk_at((int64_t)c_first, lane));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (MODE == MODE_VISIBILITY) {
        // keep = |center - p| < max_dist && q.z <= 0 && -1 <= u,v <= 1   (src/synthetic.rs:285-291, src/generate.rs:448-454)
        const double dx = gx - S.X.x, dy = gy - S.X.y, dz = gz - S.X.z;
        const double dist = sqrt(dot3(dx, dy, dz, dx, dy, dz));
        const bool front = dist < max_dist && p.qz <= 0.0;
        const bool k = front && p.u >= -1.0 && p.u <= 1.0 && p.v >= -1.0 && p.v <= 1.0;
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        if (valid) {
            uv_out[o] = front ? make_double2(p.u, p.v) : make_double2(nan, nan);
            keep[o] = k ? 1 : 0;
        }
    } else if (MODE == MODE_PROJECT) {
        if (valid) uv_out[o] = make_double2(p.u, p.v);
    } else {
        eacc += valid ? abs_pow_k<NK>(p.u - S.ob.x, norm) + abs_pow_k<NK>(p.v - S.ob.y, norm) : 0.0;
    }
}

template <int MODE, int NK, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_observations_p(
    const double *__restrict__ camblk, const double4 *__restrict__ pts4, const uint32_t *__restrict__ cam_idx,
    const uint32_t *__restrict__ pt_idx, const double2 *__restrict__ uv_obs, int n, double norm, double max_dist,
    double2 *__restrict__ uv_out, uint8_t *__restrict__ keep, double *__restrict__ block_part,
    unsigned *__restrict__ ticket, double *__restrict__ out_sum) {
    constexpr int HOT = MODE == MODE_VISIBILITY ? 28 : kCamLight;     // visibility also needs the centre at [24..26]
    constexpr bool UV = MODE == MODE_ERROR;
    __shared__ __attribute__((aligned(16))) double sCamAll[WPB * 128];   // 64 x 16-byte chunks per wave

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *sCam = sCamAll + wave * 128;
    const PipeRange R = pipe_range(n >> 6, WPB, wave);
    double eacc = 0.0;

#define C2B_PIPE_BODY(CUR, NXT, NN)                                                                          \
    pipe_issue_idx<UV>(NN, t + 2 * R.stride, lane, n, cam_idx, pt_idx, uv_obs);                               \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    pipe_issue_gather<HOT>(NXT, lane, camblk, pts4);                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    pipe_stage_cams<HOT>(CUR, lane, sCam);                                                                    \
    pipe_light_tile<MODE, NK, false>(CUR, t, 64, lane, camblk, sCam, HOT, norm, max_dist, uv_out, keep, eacc);    \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    t += R.stride;                                                                                            \
    if (++i >= R.count) goto pipe_done;

    if (R.count > 0) {
        PipeSet A, B, C;
        int t = R.first, i = 0;
        pipe_issue_idx<UV>(A, t, lane, n, cam_idx, pt_idx, uv_obs);
        pipe_issue_idx<UV>(B, t + R.stride, lane, n, cam_idx, pt_idx, uv_obs);
        __builtin_amdgcn_sched_barrier(0);
        pipe_issue_gather<HOT>(A, lane, camblk, pts4);
        __builtin_amdgcn_sched_barrier(0);
        // The first body is peeled so that the loop is entered in its steady state: the compiler's wait counts at
        // the loop head merge the entry edge with the back edge, and an entry straight from the prologue (nothing
        // issued after A's gather) would turn the head's waits into "everything older", i.e. the previous tile's
        // stores.
        do {
            C2B_PIPE_BODY(A, B, C)
            for (;;) {
                C2B_PIPE_BODY(B, C, A)
                C2B_PIPE_BODY(C, A, B)
                C2B_PIPE_BODY(A, B, C)
            }
        } while (false);
    pipe_done:;
    }
#undef C2B_PIPE_BODY

    if ((n & 63) != 0 && blockIdx.x == 0 && wave == 0) {
        PipeSet D;
        const int tile = n >> 6;
        pipe_issue_idx<UV>(D, tile, lane, n, cam_idx, pt_idx, uv_obs);
        pipe_issue_gather<HOT>(D, lane, camblk, pts4);
        pipe_stage_cams<HOT>(D, lane, sCam);
        pipe_light_tile<MODE, NK, true>(D, tile, (int)(n & 63), lane, camblk, sCam, HOT, norm, max_dist, uv_out, keep, eacc);
    }

    if (MODE == MODE_ERROR) {
        const double w = wave_sum(eacc);
        ticket_fold(w, sCamAll, block_part, ticket, out_sum);
    }
}

}  // namespace c2b
