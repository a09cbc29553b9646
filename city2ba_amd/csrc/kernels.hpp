// kernels.hpp -- gfx950 kernels of the city2ba hot path (included by capi.hip only).
//
// Design (MI355X: 256 CUs / 8 XCDs, wave64, 160 KB LDS per CU, HBM-bound stores):
//   * per-observation kernels are wave-centric: a wave owns two consecutive tiles of 64 observations
//     (two per lane) of the camera-major observation list and never synchronises with another wave;
//     all loads of both tiles are issued before any arithmetic;
//   * a wave's observations touch a short run of consecutive cameras: their hot records are staged
//     once in a wave-private LDS tile and every lane reads its camera by ds_read broadcast;
//   * point gather = two aligned 16-B loads from the padded [n][4] point table (L2/MALL
//     resident: 63 MB at --blocks 128);
//   * the 2x9 / 2x3 Jacobian blocks are transposed through wave-private LDS (144-B and 48-B
//     lane strides are bank-conflict-free for ds_write_b128) and leave as 1-KiB-per-instruction
//     contiguous non-temporal stores -- the kernel's 208 B/observation of output binds it to HBM;
//   * tile -> workgroup map is XCD-aware: the 8 XCDs each stream a contiguous eighth of the
//     observation list so that a camera's / point's neighbours hit the same 4 MiB L2;
//   * reductions are wave shuffles -> per-tile partial -> fixed-order two-stage fold
//     (no float atomics: run-to-run reproducible).
#pragma once
#include "camera_math.hpp"

namespace c2b {

constexpr int kBlock = 256;          // lanes per workgroup of the per-entity kernels (4 waves)
constexpr int kWaves = kBlock / 64;
constexpr int kRedBlocks = 1024;     // record slots of the entity reductions in the workspace (the largest grid a variant may use)
constexpr int kStatGrid = 512;       // largest grid of the entity reductions (one record per workgroup in the workspace): 2 per CU.  A/B at 2.6 M entities (r03): 256 / 512 / 768 / 1024 / 2048 / 4096 workgroups -> 35 / 37 / 38 / 40 / 49 / 68 us
constexpr int kStatRec = 20;         // doubles per stats partial record (18 used)
constexpr int kStatBatch = 4;        // entities a thread of the statistics passes loads before it uses any
constexpr int kStatBlock = 256;      // threads per workgroup of the one-launch statistics pass (k_stats_pass1)
constexpr bool kStatPipe = false;    // next batch's loads in flight while the current one is accumulated
constexpr bool kStatChunk = false;   // contiguous entities per workgroup instead of grid-strided slabs

typedef double d2_t __attribute__((ext_vector_type(2)));

// ---- XCD-aware tile map (bijective for any n_tiles; cdna guide T1) -----------------------
C2B_DEV int64_t xcd_tile(int64_t bid, int64_t n_tiles) {
    const int64_t q = n_tiles >> 3, r = n_tiles & 7;
    const int64_t xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// ---- wave / block reductions ---------------------------------------------------------------
C2B_DEV double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
C2B_DEV double wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
    return v;
}
C2B_DEV double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}

// ---- one-launch deterministic sum: workgroup partials + last-arriver fold ---------------------------------------
// Every wave hands in one value (valid on lane 0).  LDS -> one partial per workgroup, written through to memory
// (agent-scope store, drained) -> fetch_add on the launch's ticket words -> the workgroup that arrives last
// reads all gridDim.x partials (plain loads behind ONE agent-scope acquire) and sums them in a fixed order:
// thread t takes partials t, t + blockDim, ...; wave shuffle tree; waves in order.  Same grid => the same bits,
// run to run; no float atomics; no second launch.  The ticket words live in the CALLER'S WORKSPACE, next to the
// partials they count (capi.hip: workspace layout): a workspace serves one launch at a time by contract, so no two
// launches can ever meet on the same counters, whatever streams, threads or graphs they come from.
// c2b_workspace_init zeroes them once and writes a magic word; the last arriver resets them, so every launch -- and
// every replay of a captured launch -- starts clean.  A workspace that was never initialised has no magic: every
// workgroup then skips the counters and writes NaN, so the sum is loudly NaN instead of silently stale.
// Deliberately NO release fence per workgroup: its L2 write-back costs microseconds and serialises chip-wide
// (measured +2.1 ms over 768 workgroups); a CAS loop on the ticket serialises the same way -- one fetch_add does not.
// What replaces the release is gfx942 / gfx950 ISA behaviour, not the language memory model: the partial is an
// agent-scope atomic store (global_store ... sc1: written through to memory past the non-coherent per-XCD L2) that
// is drained (s_waitcnt vmcnt(0)) before the ticket's fetch_add is issued, and the last arriver's acquire is
// buffer_inv sc1.  tests/test_isa_pins.py asserts exactly these instructions in the compiled kernels.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "ticket_fold's publish (sc1 write-through store + s_waitcnt instead of a release fence) is validated on gfx942 / gfx950 only"
#endif
// sRed: >= blockDim.x/64 + 1 doubles of LDS.  Every thread of every workgroup must call this.
constexpr unsigned kTicketLine = 32;                 // words per 128-B line
constexpr unsigned kTicketLeaves = 64;
constexpr unsigned kTicketMagicAt = (1 + kTicketLeaves) * kTicketLine;   // its own line, after the top word and the leaves
constexpr unsigned kTicketWords = (2 + kTicketLeaves) * kTicketLine;     // one workspace's ticket block (8 448 bytes)
constexpr unsigned kTicketMagic = 0xC2B71C3Eu;

// thread 0 of a workgroup: count this workgroup in; true for the ONE workgroup of the launch that arrives last (it has
// then reset the counters and acquired every other workgroup's published data).  `magic` = the word at kTicketMagicAt
// (a plain load the caller issues early: written by an earlier launch, so any cache level may serve it); without it
// the workspace was never initialised and the counters are not touched.
C2B_DEV bool ticket_arrive(unsigned *__restrict__ ticket, unsigned magic) {
    if (magic != kTicketMagic) return false;
    // Two-level arrival count: workgroup b arrives at leaf word b % 64 (each leaf in its own 128-B line), the
    // workgroup that completes a leaf arrives at the top word.  18 849 workgroups adding to ONE word cost the
    // light kernels +115 us (one same-address atomic per ~13 ns); 64 leaves spread them over 64 lines.
    const unsigned leaf = blockIdx.x & (kTicketLeaves - 1u), leaf_n = (gridDim.x - leaf + kTicketLeaves - 1u) / kTicketLeaves;
    unsigned *lw = ticket + kTicketLine * (1u + leaf);
    bool last = false;
    if (__hip_atomic_fetch_add(lw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == leaf_n - 1) {
        __hip_atomic_store(lw, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned top_n = gridDim.x < kTicketLeaves ? gridDim.x : kTicketLeaves;
        if (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == top_n - 1) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = true;
        }
    }
    if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __builtin_amdgcn_s_waitcnt(0);
    }
    return last;
}

// N sums through ONE arrival count (N = 2: the L1 and L2 errors run_noise prints back to back, src/bin/city2ba.rs:283-287,
// 350-354).  wave_value[k] is valid on lane 0; set k's partials live at block_part[k * gridDim.x ...]; out_sum[k] gets
// set k's total.  Each set is folded exactly like the single sum (same thread -> partial map, same trees), so out_sum[k]
// carries the bits a one-sum launch of the same grid would produce.  sRed: >= N * blockDim.x/64 + 1 doubles of LDS.
template <int N>
C2B_DEV void ticket_fold_n(const double (&wave_value)[N], double *sRed, double *__restrict__ block_part,
                           unsigned *__restrict__ ticket, double *__restrict__ out_sum) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    unsigned magic = 0u;
    if (threadIdx.x == 0) magic = ticket[kTicketMagicAt];          // in flight across the barriers below
    __syncthreads();                                   // sRed may alias LDS other waves were still using
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) sRed[k * n_waves + wave] = wave_value[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            double b = 0.0;
            for (int w = 0; w < n_waves; ++w) b += sRed[k * n_waves + w];
            __hip_atomic_store(block_part + (size_t)k * gridDim.x + blockIdx.x, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_s_waitcnt(0);                 // (a builtin, not inline asm: asm would halve the VGPR budget)
        const bool last = ticket_arrive(ticket, magic);
        if (magic != kTicketMagic) {
#pragma unroll
            for (int k = 0; k < N; ++k) out_sum[k] = __longlong_as_double(0x7ff8000000000000LL);
        }
        sRed[N * n_waves] = last ? 1.0 : 0.0;
    }
    __syncthreads();
    if (sRed[N * n_waves] == 0.0) return;              // workgroup-uniform
    const unsigned n = gridDim.x, step = blockDim.x;
#pragma unroll
    for (int s = 0; s < N; ++s) {
        const double *part = block_part + (size_t)s * n;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0; // four independent chains keep the loads in flight
        unsigned k = threadIdx.x;
        for (; k + 3 * step < n; k += 4 * step) {
            a0 += part[k]; a1 += part[k + step]; a2 += part[k + 2 * step]; a3 += part[k + 3 * step];
        }
        for (; k < n; k += step) a0 += part[k];
        const double w = wave_sum((a0 + a1) + (a2 + a3));
        __syncthreads();
        if (lane == 0) sRed[wave] = w;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int i = 0; i < n_waves; ++i) t += sRed[i];
            out_sum[s] = t;
        }
    }
}
C2B_DEV void ticket_fold(double wave_value, double *sRed, double *__restrict__ block_part,
                         unsigned *__restrict__ ticket, double *__restrict__ out_sum) {
    const double v[1] = {wave_value};
    ticket_fold_n<1>(v, sRed, block_part, ticket, out_sum);
}

// c2b_workspace_init: zero the arrival counters, write the magic (one workgroup)
__global__ __launch_bounds__(256) void k_workspace_init(unsigned *__restrict__ ticket) {
    for (unsigned i = threadIdx.x; i < kTicketWords; i += 256) ticket[i] = i == kTicketMagicAt ? kTicketMagic : 0u;
}

// ---- a wave's 64 consecutive records of an array-of-structures table, through LDS ---------------------------------
// The per-camera kernels work on one record per lane (cam15: 15 scalars, bal9: 9).  Read or written straight from
// registers, record field k of 64 lanes is one instruction touching ~60 different 128-byte lines (lane stride 120 B), W
// instructions per record: ~900 line visits per wave for 60 lines of data.  These helpers move the wave's 64 records
// between global memory and a wave-private slab with one scalar per lane per instruction -- consecutive addresses, four
// whole lines per instruction for doubles -- and the lanes then pick their fields out of LDS (stride W scalars: at most
// two lanes per bank).  n_rows < 64 only in a table's last wave.
template <int W, typename T>
C2B_DEV void wave_rows_load(const T *__restrict__ g, int n_rows, T *slab, int lane) {
    const int total = n_rows * W;
#pragma unroll
    for (int it = 0; it < W; ++it) {
        const int e = it * 64 + lane;
        if (e < total) slab[e] = g[e];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <int W, typename T>
C2B_DEV void wave_rows_store(T *__restrict__ g, int n_rows, const T *slab, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int total = n_rows * W;
#pragma unroll
    for (int it = 0; it < W; ++it) {
        const int e = it * 64 + lane;
        if (e < total) g[e] = slab[e];
    }
}

// ---- per-camera kernels ---------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_cameras_from_bal(const double *__restrict__ bal9, int64_t n, double *__restrict__ cam15) {
    __shared__ double sRows[kWaves][64 * 15];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * blockDim.x + wave * 64;
    if (wave0 >= n) return;                                              // wave-uniform; no workgroup barrier below
    const int n_rows = n - wave0 < 64 ? (int)(n - wave0) : 64;
    double *slab = sRows[wave];
    const int64_t ic = wave0 + (lane < n_rows ? lane : n_rows - 1);
    double b[9], R[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) b[k] = bal9[9 * ic + k];               // strided reads re-use their lines in L1
    from_rodrigues(b[0], b[1], b[2], R);
    if (lane < n_rows) {
#pragma unroll
        for (int k = 0; k < 9; ++k) slab[15 * lane + k] = R[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) slab[15 * lane + 9 + k] = b[3 + k];
    }
    wave_rows_store<15>(cam15 + 15 * wave0, n_rows, slab, lane);
}

__global__ __launch_bounds__(kBlock) void k_cameras_to_bal(const double *__restrict__ cam15, int64_t n, double *__restrict__ bal9) {
    __shared__ double sRows[kWaves][64 * 9];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * blockDim.x + wave * 64;
    if (wave0 >= n) return;                                              // wave-uniform
    const int n_rows = n - wave0 < 64 ? (int)(n - wave0) : 64;
    double *slab = sRows[wave];
    const int64_t ic = wave0 + (lane < n_rows ? lane : n_rows - 1);
    double c[15], w[3];
#pragma unroll
    for (int k = 0; k < 15; ++k) c[k] = cam15[15 * ic + k];            // strided reads re-use their lines in L1 (A/B: no gain from LDS)
    to_rodrigues(c, w);
    if (lane < n_rows) {
        slab[9 * lane] = w[0]; slab[9 * lane + 1] = w[1]; slab[9 * lane + 2] = w[2];
#pragma unroll
        for (int k = 0; k < 6; ++k) slab[9 * lane + 3 + k] = c[9 + k];
    }
    wave_rows_store<9>(bal9 + 9 * wave0, n_rows, slab, lane);
}

// One camera per lane.  The 256-byte records leave through a wave-private LDS slab (two half-wave rounds of 32 records,
// lane stride 33 doubles so that the 8-byte LDS writes spread over the banks) as 16-byte-per-lane stores of whole
// lines: written straight from registers every one of the 32 store instructions touched 64 different lines (lane stride
// 256 B) -- 2 048 line visits per wave for 128 lines of output (59 us for 660 480 cameras, r02).
template <bool FROM_BAL>
__global__ __launch_bounds__(kBlock) void k_cameras_prepare(const double *__restrict__ in, int64_t n, double *__restrict__ camblk,
                                                           double *__restrict__ cen4) {
    constexpr int kStride = kCamBlk + 1;                                 // doubles per staged record
    __shared__ __attribute__((aligned(16))) double sOut[kWaves][32 * kStride + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * blockDim.x + wave * 64;  // this wave's first camera
    if (wave0 >= n) return;                                              // wave-uniform; no workgroup barrier below
    const int64_t i = wave0 + lane;
    const bool valid = i < n;
    const int64_t ic = valid ? i : n - 1;                                // lanes past the end recompute the last camera
    double c[15], w[3], blk[kCamBlk];
    double *slab = sOut[wave];
    const int n_wave = n - wave0 < 64 ? (int)(n - wave0) : 64;
    constexpr int W = FROM_BAL ? 9 : 15;
    {
        const double *b = in + W * ic;                                   // strided reads re-use their lines in L1
        if (FROM_BAL) {
            w[0] = b[0]; w[1] = b[1]; w[2] = b[2];
            from_rodrigues(w[0], w[1], w[2], c);
#pragma unroll
            for (int k = 0; k < 6; ++k) c[9 + k] = b[3 + k];
        } else {
#pragma unroll
            for (int k = 0; k < 15; ++k) c[k] = b[k];
            to_rodrigues(c, w);
        }
    }
    fill_camblk(c, w[0], w[1], w[2], blk);
    // the compact centre table (one 32-byte row per camera, the same bits as the record's centre field): what the
    // statistics passes and the generators' cell list read instead of a 128-byte line of the record per camera
    if (cen4 != nullptr && valid) {
        double2 *o = reinterpret_cast<double2 *>(cen4 + 4 * i);
        o[0] = make_double2(blk[kCenter], blk[kCenter + 1]);
        o[1] = make_double2(blk[kCenter + 2], 0.0);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if ((lane >> 5) == h) {
            double *o = slab + (lane & 31) * kStride;
#pragma unroll
            for (int k = 0; k < kCamBlk; ++k) o[k] = blk[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int nv = n_wave - h * 32;
        nv = nv < 0 ? 0 : (nv > 32 ? 32 : nv);
        // the blocked table (camera_math.hpp: groups of 8 cameras -- light lines, J_l tails, centres, pad) is written front to back:
        // chunk w of a group's 128 is chunk `part` of its camera `r8`
        double *dst = camblk + (wave0 + h * 32) * kCamBlk;               // wave0 + h * 32 is a multiple of 8: a group's start
#pragma unroll
        for (int it = 0; it < 8; ++it) {                                 // 32 records x 16 chunks of 16 bytes = 512 chunks
            const int ch = it * 64 + lane, w = ch & 127;
            const int r8 = w < 64 ? w >> 3 : (w < 96 ? (w - 64) >> 2 : ((w & 15) >> 1));
            const int part = w < 64 ? w & 7 : (w < 96 ? 8 + (w & 3) : (w < 112 ? 12 : 14) + (w & 1));
            const int rec = (ch >> 7) * 8 + r8;
            if (rec < nv) {
                const double *q = slab + rec * kStride + 2 * part;
                *reinterpret_cast<double2 *>(dst + 2 * ch) = make_double2(q[0], q[1]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// The centre table alone, from the in-memory state (r06): what the statistics need of a camera after something moved it.  The same
// cm_center on the same operands as fill_camblk -- the same bits as the table k_cameras_prepare<false> would write -- for 152 bytes of
// traffic per camera instead of 408: Level 1's statistics between add_drift and add_noise_entities (src/noise.rs:133 calls std() on
// the drifted problem) no longer derive a whole camera table that the entity noise invalidates before any pass reads it.
__global__ __launch_bounds__(kBlock) void k_cameras_centers(const double *__restrict__ cam15, int64_t n, double *__restrict__ cen4) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double c[12], ctr[3];
#pragma unroll
    for (int k = 0; k < 12; ++k) c[k] = cam15[15 * i + k];
    cm_center(c, c[9], c[10], c[11], ctr);
    double2 *o = reinterpret_cast<double2 *>(cen4 + 4 * i);
    o[0] = make_double2(ctr[0], ctr[1]);
    o[1] = make_double2(ctr[2], 0.0);
}

// Camera::from_position_direction, src/baproblem.rs:153-159: loc = -1.0 * dir.rotate_point(position)
__global__ void k_cameras_from_position_direction(const double *__restrict__ pos, const double *__restrict__ dir,
                                                  int64_t n, double *__restrict__ cam15) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double R[9], v[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = dir[9 * i + k];
    cm_mat_vec(R, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], v);
    double *o = cam15 + 15 * i;
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = R[k];
    o[9] = -1.0 * v[0]; o[10] = -1.0 * v[1]; o[11] = -1.0 * v[2];
    o[12] = 1.0; o[13] = 0.0; o[14] = 0.0;
}

__global__ void k_points_pad(const double *__restrict__ p3, int64_t n, double4 *__restrict__ p4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    p4[i] = make_double4(p3[3 * i], p3[3 * i + 1], p3[3 * i + 2], 0.0);
}
__global__ void k_points_unpad(const double4 *__restrict__ p4, int64_t n, double *__restrict__ p3) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double4 v = p4[i];
    p3[3 * i] = v.x; p3[3 * i + 1] = v.y; p3[3 * i + 2] = v.z;
}

// cam_idx[o] = (first c with row_ptr[c] > o + base) - 1
__global__ void k_expand_rows(const uint64_t *__restrict__ row_ptr, int64_t n_cam, int64_t base,
                              int64_t n_obs, uint32_t *__restrict__ cam_idx) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_obs) return;
    const uint64_t key = (uint64_t)(o + base);
    int64_t lo = 0, hi = n_cam + 1;          // search in row_ptr[0..n_cam]
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (row_ptr[mid] > key) hi = mid; else lo = mid + 1;
    }
    cam_idx[o] = (uint32_t)(lo - 1);
}

// the camera whose list holds observation o: (first c with row_ptr[c] > o) - 1, as k_expand_rows -- clamped into
// [0, n_cam) so that a row_ptr that does not cover o (row_ptr[0] > o, row_ptr[n_cam] <= o: a caller's bug) yields a
// wrong but valid camera, never an out-of-range record address
C2B_DEV uint32_t csr_search(const uint64_t *__restrict__ row_ptr, int n_cam, uint64_t o) {
    int lo = 0, hi = n_cam + 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (row_ptr[mid] > o) hi = mid; else lo = mid + 1;
    }
    lo = lo < 1 ? 1 : (lo > n_cam ? n_cam : lo);
    return (uint32_t)(lo - 1);
}

// The row structure as the per-observation kernels read it (k_observations<..., CSR = true>): one 16-byte record per
// 64 observations = { mask of the lanes whose observation opens a new camera's list (bit 0 unused), the camera of the
// tile's first observation, 0 }.  Bit 31 of the camera marks a tile with an EMPTY list inside it (two boundaries on
// one observation cannot be one mask bit); the kernels search row_ptr for those.
// Two launches (r03; one wave per tile with a binary search PER OBSERVATION took 218 us at 19.3 M observations):
// k_rows_pack_tiles -- one thread per tile: the camera of its first observation (the only search), mask cleared;
// k_rows_pack_marks -- one thread per camera: its list's first observation sets its lane's bit in its tile (unless that
// is lane 0, whose camera the record already names), and flags the tile if the list before it is empty (then the step
// from the previous observation's camera is larger than one, which the mask cannot say).
__global__ __launch_bounds__(256) void k_rows_pack_tiles(const uint64_t *__restrict__ row_ptr, int n_cam, int n,
                                                         uint4 *__restrict__ tiles) {
    const int tile = blockIdx.x * 256 + threadIdx.x;
    if ((int64_t)tile * 64 >= n) return;
    tiles[tile] = make_uint4(0u, 0u, csr_search(row_ptr, n_cam, (uint64_t)tile * 64), 0u);
}
__global__ __launch_bounds__(256) void k_rows_pack_marks(const uint64_t *__restrict__ row_ptr, int n_cam, int n,
                                                         uint32_t *__restrict__ tiles) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cam) return;
    const uint64_t b = row_ptr[c], e = row_ptr[c + 1];
    if (e <= b || b >= (uint64_t)n || (b & 63u) == 0) return;        // empty list, outside the list, or a tile's first lane
    uint32_t *rec = tiles + 4 * (b >> 6);
    const unsigned lane = (unsigned)(b & 63u);
    atomicOr(rec + (lane >> 5), 1u << (lane & 31u));
    if (c > 0 && row_ptr[c - 1] == b) atomicOr(rec + 2, 0x80000000u);  // the list before this one is empty
}

// The cameras of a wave's OPL tiles from the tile records (wave-uniform scalar loads): a lane's camera is the tile's
// first camera plus the number of mask bits at or below the lane -- two v_mbcnt per tile, nothing per observation.
// `tiles` / `base` / `n` are relative to the launch's first observation, which is observation obs_base of the list
// row_ptr describes (only the search for flagged tiles needs that).  Tiles past the end repeat the last one.
template <int OPL>
C2B_DEV void rows_cameras(const uint4 *__restrict__ tiles, const uint64_t *__restrict__ row_ptr, int n_cam, int base,
                          int n, int64_t obs_base, int lane, uint32_t (&ci)[OPL]) {
    const int last_tile = (n - 1) >> 6;
    // The records are wave-uniform, so they arrive by scalar loads -- ALL of them requested before any is looked at, each as
    // one 16-byte load (r05: the compiler had split every record into its flag word and, behind the branch on it, its mask:
    // 2 x OPL dependent scalar round trips at the head of every wave, in front of the gathers that need the camera ids).
    uint4 rec[OPL];
    int tis[OPL];
#pragma unroll
    for (int t = 0; t < OPL; ++t) {
        const int ti = (base >> 6) + t;
        tis[t] = ti < last_tile ? ti : last_tile;
        rec[t] = tiles[tis[t]];
    }
#pragma unroll
    for (int t = 0; t < OPL; ++t) {
        // bits 1..63 of the mask, moved down one place: mbcnt counts the set bits BELOW a lane (evaluated unconditionally, so
        // that the whole record is needed before the branch)
        const uint32_t lo = (rec[t].x >> 1) | (rec[t].y << 31), hi = rec[t].y >> 1;
        uint32_t c = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, rec[t].z));
        if (rec[t].z & 0x80000000u) {                                    // wave-uniform: an empty list inside this tile
            int o = tis[t] * 64 + lane;
            o = o < n ? o : n - 1;
            c = csr_search(row_ptr, n_cam, (uint64_t)(obs_base + o));
        }
        ci[t] = c;
    }
}

// ---- wave-private camera tile ------------------------------------------------------------------------
// A wave's 64 consecutive observations touch a short run of consecutive cameras (3-4 on the grid).  The
// wave copies the first HOT doubles of those records into its own LDS tile and every lane then reads its
// camera by ds_read broadcast.  Lanes whose camera is outside the staged run (unsorted cam_idx, or more
// than kCamW cameras in 64 observations) read global memory instead: correct, just slower.
constexpr int kCamW = 12;                      // cameras staged per wave


template <bool NT>
C2B_DEV void store16(char *dst, const double2 v) {
    if (NT) {
        d2_t t; t.x = v.x; t.y = v.y;
        __builtin_nontemporal_store(t, reinterpret_cast<d2_t *>(dst));
    } else {
        *reinterpret_cast<double2 *>(dst) = v;
    }
}

// ---- project / error / visibility: the light per-observation kernels ----------------------------
// Wave-centric like the Jacobian kernel; they need R, t, intrinsics only (15 doubles, staged as 16).
// MODE_ERROR12: the L1 and the L2 error sum in one pass (run_noise evaluates them back to back on the same data,
// src/bin/city2ba.rs:283-287, 350-354); MODE_NOISE_ERROR12: add_noise's observation pass (src/noise.rs:152-170) -- draw,
// perturb and store uv -- fused with the two error sums of the perturbed observations that follow it in run_noise.
// MODE_VISIBILITY_BITS: the predicate with the keep mask as ONE 64-bit ballot word per tile of 64 pairs (bit l = pair 64 t + l)
// instead of a byte per pair -- 1 of the pass's 26 bytes per pair less, and what a compaction wants anyway
enum { MODE_PROJECT = 0, MODE_ERROR = 1, MODE_VISIBILITY = 2, MODE_ERROR12 = 3, MODE_NOISE_ERROR12 = 4, MODE_VISIBILITY_BITS = 5 };
constexpr int kObsWPB = 8;                     // waves per workgroup
constexpr int kCamLight = 16;

constexpr int kObsOPL = 3;                     // observations per lane (all tiles' loads issued up front)

// ---- the slow path of the per-observation kernels: lanes whose camera was not staged ---------------------------
// (unsorted input, or more than kCamW cameras among a wave's observations).  One round serves up to kCamW DISTINCT
// cameras: they are picked from the leftover lanes (lowest lane first), their ids go to a small LDS table, their rows
// are staged into kCamW spare slots with the ordinary cooperative copy, and every lane whose camera was picked runs the
// same LDS-only arithmetic once.  A randomly ordered list (64 different cameras per tile) takes 6 rounds, not 64.
// Returns the number of cameras picked; `my` = this lane's slot or -1.
C2B_DEV int pick_cameras(uint64_t todo, uint32_t ci, int lane, uint32_t *sIdx, int &my) {
    my = -1;
    int ns = 0;
    uint64_t rest = todo;
    while (rest != 0 && ns < kCamW) {
        const uint32_t cf = __builtin_amdgcn_readlane(ci, (int)__builtin_ctzll(rest));
        const bool mine = ((rest >> lane) & 1ull) != 0 && ci == cf;
        if (mine) my = ns;
        if (lane == 0) sIdx[ns] = cf;
        rest &= ~__builtin_amdgcn_ballot_w64(mine);
        ++ns;
    }
    return ns;
}

// NK (camera_math.hpp: NORM_1 / NORM_2 / NORM_ANY) fixes the error norm at compile time; MODE_ERROR folds
// sum |du|^norm + |dv|^norm over ALL observations into out_sum[0] in this one launch (ticket_fold).
//
// These kernels are issue-bound as much as memory-bound (SQ counters, profiles/r02a_light_sq.json: ~240 vector
// instructions per wave of 128 observations, ~70 % of the SIMD's issue slots with 8 waves resident), so the code
// around the ~69 f64 operations of a projection is kept lean: 32-bit observation indices (a launch holds < 2^31),
// loads of lanes past the end clamped to the last observation instead of predicated (only stores are), ONE inlined
// copy of the arithmetic per tile reading its camera from LDS (lanes whose camera was not staged -- unsorted input,
// or > kCamW cameras in 128 observations -- are served in extra rounds, one restaged camera at a time).
C2B_DEV int xcd_tile32(int bid, int n_tiles) {
    const int q = n_tiles >> 3, r = n_tiles & 7;
    const int xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// doubles from the table's start to chunk j of a camera's staged row: chunks 0..7 = its light line, 8.. = its centre (the
// visibility predicate's; heavy line, record doubles 24..)
C2B_DEV int64_t cam_row(int64_t c, int j) {
    if (j >= kCamLight / 2) return cam_center_at(c) + 2 * (j - kCamLight / 2);
    return cam_light_at(c) + 2 * j;
}

// MINW = waves per SIMD the register allocation must leave room for (HIP's second __launch_bounds__ argument)
//
// CSR: the camera of an observation comes from the reference's own structure -- one list per camera, i.e. row_ptr
// (src/baproblem.rs:256-260) -- instead of a 4-byte index per observation: `cam_idx` then points at the tile records
// k_csr_pack derives from row_ptr (16 bytes per 64 observations).  SURVEY 8(d)'s algorithmic bytes assume exactly
// this: 4 B of point index per observation and the row structure once.
template <int MODE, int NK = NORM_2, int OPL = kObsOPL, int WPB = kObsWPB, int MINW = 1,
          bool CSR = false, bool NTS = false, int NTL = 0>
__global__ __launch_bounds__(WPB * 64, MINW) void k_observations(
    const double *__restrict__ camblk, const double4 *__restrict__ pts4,
    const uint32_t *__restrict__ cam_idx, const uint32_t *__restrict__ pt_idx,
    const double2 *__restrict__ uv_obs, int n, int n_btiles, double norm, double max_dist,
    double2 *__restrict__ uv_out, uint8_t *__restrict__ keep, double *__restrict__ block_part,
    unsigned *__restrict__ ticket, double *__restrict__ out_sum, const uint64_t *__restrict__ row_ptr, int n_cam,
    int64_t obs_base, uint64_t seed) {
    // MODE_NOISE_ERROR12: `norm` carries observations_std, uv_out is the observation array (read, perturbed, written
    // back), obs_base the global index of this launch's first observation (the draws' counter), uv_obs is unused
    // per staged camera: R, t, intrinsics (16 doubles) and, for the visibility predicate, the centre (camblk 24..27)
    constexpr bool VIS = MODE == MODE_VISIBILITY || MODE == MODE_VISIBILITY_BITS;
    constexpr int HOT = VIS ? 20 : kCamLight;
    constexpr int CH = HOT / 2;                                           // 16-byte chunks per camera
    constexpr int kPerWave = 2 * kCamW * HOT + 8;                        // staged cameras | slow-path slots | picked ids
    // MODE_NOISE_ERROR12: the draw's tables (camera_math.hpp: g_noise_tab, 10 KB) behind the camera tiles; MODE_ERROR with a norm
    // other than 1 and 2: the same tables + the exponential's (pow_tab, 11 KB)
    constexpr bool POWTAB = MODE == MODE_ERROR && NK == NORM_ANY;
    constexpr int kTabN = MODE == MODE_NOISE_ERROR12 ? kNoiseTab : (POWTAB ? kPowTab : 0);
    constexpr int kTabDoubles = 2 * kTabN;
    __shared__ __attribute__((aligned(16))) double sCamAll[WPB * kPerWave + kTabDoubles];
    tab2_t *sTab = reinterpret_cast<tab2_t *>(sCamAll + WPB * kPerWave);
    // The table's entries are REQUESTED first and written to LDS only after this wave's own index / point / camera loads are on
    // their way (vector-memory results return in order, so the table costs the workgroup no round trip of its own); the
    // workgroup barrier sits right before the arithmetic -- once per wave, in either branch of the wave-uniform `base < n`.
    constexpr int kTabTrips = (kTabN + WPB * 64 - 1) / (WPB * 64);
    tab2_t tabv[kTabTrips > 0 ? kTabTrips : 1];
    if (kTabN > 0) {
        static_assert((WPB * kPerWave) % 2 == 0, "the table's 16-byte entries start on a 16-byte boundary");
#pragma unroll
        for (int k = 0; k < kTabTrips; ++k) {
            const int i = (int)threadIdx.x + k * WPB * 64;
            tabv[k] = g_noise_tab[i < kTabN ? i : 0];
        }
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int base = (xcd_tile32(blockIdx.x, n_btiles) * WPB + wave) * (OPL * 64);
    double eacc = 0.0, eacc1 = 0.0;                                      // MODE_*ERROR12: eacc1 = the L1 sum, eacc = the L2 sum
    if (base < n) {                                                      // wave-uniform; waves past the end only fold
        uint32_t ci[OPL], pi[OPL];
        double4 X[OPL];
#pragma unroll
        for (int t = 0; t < OPL; ++t) {
            int o = base + t * 64 + lane;
            o = o < n ? o : n - 1;                                       // clamped, not predicated
            if (!CSR) ci[t] = (NTL & 1) ? __builtin_nontemporal_load(cam_idx + o) : cam_idx[o];
            pi[t] = (NTL & 1) ? __builtin_nontemporal_load(pt_idx + o) : pt_idx[o];
        }
        if (CSR) rows_cameras<OPL>(reinterpret_cast<const uint4 *>(cam_idx), row_ptr, n_cam, base, n, 0, lane, ci);

        // wave-private camera tile covering all OPL tiles: cameras ci[0](lane 0) .. ci[OPL-1](lane 63) on sorted input.
        // r05: its (at most kStageTrips x 64) 16-byte chunks are REQUESTED here, all at once and -- in the row-structure form,
        // where the camera ids come from scalar records -- before this wave waits for its point indices; they go to LDS after
        // the point gathers are on their way.  (Before: a load -> wait -> LDS-write loop behind the gathers, whose second trip,
        // taken whenever the wave's tiles touch more than 64 / CH cameras, was a round trip of its own.)
        double *sCam = sCamAll + wave * kPerWave;
        double *sSlow = sCam + kCamW * HOT;
        uint32_t *sIdx = reinterpret_cast<uint32_t *>(sSlow + kCamW * HOT);
        const uint32_t c_first = __builtin_amdgcn_readfirstlane(ci[0]);
        const uint32_t c_last = __builtin_amdgcn_readlane(ci[OPL - 1], 63);
        uint32_t n_staged = c_last >= c_first ? c_last - c_first + 1 : 1;
        if (n_staged > (uint32_t)kCamW) n_staged = kCamW;
        constexpr int kStageTrips = (kCamW * CH + 63) / 64;
        d2_t camv[kStageTrips];
#pragma unroll
        for (int q = 0; q < kStageTrips; ++q) {
            const int ch = lane + q * 64;
            if (ch < (int)n_staged * CH) {
                const int k = ch / CH, j = ch % CH;
                camv[q] = *reinterpret_cast<const d2_t *>(camblk + cam_row((int64_t)(c_first + k), j));
            }
        }
        if (kTabN > 0) {                                                 // the table's entries were requested first: they are here first
#pragma unroll
            for (int k = 0; k < kTabTrips; ++k) {
                const int i = (int)threadIdx.x + k * WPB * 64;
                if (i < kTabN) sTab[i] = tabv[k];
            }
        }
#pragma unroll
        for (int t = 0; t < OPL; ++t) X[t] = pts4[pi[t]];
#pragma unroll
        for (int q = 0; q < kStageTrips; ++q) {
            const int ch = lane + q * 64;
            if (ch < (int)n_staged * CH) *reinterpret_cast<d2_t *>(sCam + (ch / CH) * HOT + 2 * (ch % CH)) = camv[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (kTabN > 0) __syncthreads();

#pragma unroll
        for (int t = 0; t < OPL; ++t) {
            const int tile0 = base + t * 64;
            if (tile0 >= n) break;                                       // wave-uniform
            const int o = tile0 + lane;
            const bool valid = o < n;
            // the observed uv of THIS tile is requested only now: holding all OPL of them from the start costs 12
            // VGPRs and the eighth wave per SIMD (measured: 119 us held, 111 us requested per tile)
            double2 ob = make_double2(0.0, 0.0);
            if (MODE == MODE_ERROR || MODE == MODE_ERROR12 || MODE == MODE_NOISE_ERROR12) {
                const double2 *src = (MODE == MODE_NOISE_ERROR12 ? uv_out : uv_obs) + (valid ? o : n - 1);
                if (NTL & 2) {
                    const d2_t t2 = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(src));
                    ob = make_double2(t2.x, t2.y);
                } else {
                    ob = *src;
                }
            }
            // First pass, unmasked: every lane projects through a staged camera -- its own if that is staged, camera 0
            // of the tile otherwise (valid operands, result discarded).  On camera-major input that serves every lane
            // and nothing below runs.  Leftover lanes (unsorted input, or more cameras than were staged) are then
            // served one camera at a time through the extra slot by the same LDS-only arithmetic, merged under a mask.
            uint32_t local = ci[t] - c_first;
            bool in = local < n_staged;
            lds_cptr cam = (lds_cptr)sCam + (in ? local : 0u) * HOT;
            // every lane: the projection with |p|^4 = n * n (final for k2 == 0); which lanes' cameras have k2 != 0 is kept as a lane
            // mask in scalar registers -- the pow() route for those runs behind one wave-uniform branch below the loop.  What that
            // route needs of the head is |p| = (px, py): a lane with k2 != 0 carries THEM in the pixel's two registers (its n * n
            // pixel is not the answer anyway), so nothing is projected twice and nothing more is live across the loop than before.
            Proj p = project_obs_k0(cam, X[t].x, X[t].y, X[t].z);
            uint64_t k2nz = __builtin_amdgcn_ballot_w64(cam[14] != 0.0);
            if (k2nz != 0 && ((k2nz >> lane) & 1ull)) { p.u = p.px; p.v = p.py; }       // behind a wave-uniform test: free when k2 = 0
            double gx = 0.0, gy = 0.0, gz = 0.0;
            if (VIS) { gx = cam[16]; gy = cam[17]; gz = cam[18]; }
            uint64_t todo = __builtin_amdgcn_ballot_w64(valid && !in);
            while (todo != 0) {                                          // wave-uniform; never taken on sorted input
                int my;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int ns = pick_cameras(todo, ci[t], lane, sIdx, my);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int ch = lane; ch < ns * CH; ch += 64) {
                    const int k = ch / CH, j = ch % CH;
                    *reinterpret_cast<d2_t *>(sSlow + k * HOT + 2 * j) =
                        *reinterpret_cast<const d2_t *>(camblk + cam_row((int64_t)sIdx[k], j));
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const bool mine = my >= 0;
                lds_cptr cam2 = (lds_cptr)sSlow + (mine ? my : 0) * HOT;
                const Proj q = project_obs_k0(cam2, X[t].x, X[t].y, X[t].z);
                const bool k2mine = mine && cam2[14] != 0.0;
                const uint64_t served = __builtin_amdgcn_ballot_w64(mine);
                k2nz = (k2nz & ~served) | __builtin_amdgcn_ballot_w64(k2mine);
                if (mine) {
                    p = q;
                    if (k2mine) { p.u = q.px; p.v = q.py; }
                    if (VIS) { gx = cam2[16]; gy = cam2[17]; gz = cam2[18]; }
                }
                todo &= ~served;
            }
            if (k2nz != 0) {                                             // wave-uniform; cameras with k2 != 0 only (src/baproblem.rs:147-149)
                // |p|^4 = p.magnitude().powf(4.0) as the reference computes it -- libm's pow -- for the lanes that need it: |p|^2 again
                // from the (px, py) they carried (the same expression, the same bits), the intrinsics from the lane's staged camera
                // or, for a lane the loop above served (its slot was restaged every round), from the table itself, then the tail
                if ((k2nz >> lane) & 1ull) {
                    Proj q;
                    q.px = p.u; q.py = p.v;
                    q.n = q.px * q.px + q.py * q.py;
                    double f, k1, k2;
                    if (valid && !in) {
                        glb_cptr g = (glb_cptr)camblk + cam_row((int64_t)ci[t], 6);      // record doubles 12 ..
                        f = g[0]; k1 = g[1]; k2 = g[2];
                    } else {
                        f = cam[12]; k1 = cam[13]; k2 = cam[14];
                    }
                    project_tail(q, f, k1, k2, pow4_libm(sqrt(q.n)));
                    p.u = q.u; p.v = q.v;
                }
            }
            // keep = |center - p| < max_dist && q.z <= 0 && -1 <= u,v <= 1   (src/synthetic.rs:285-291, src/generate.rs:448-454)
            bool front = false;
            if (VIS) {
                const double dx = gx - X[t].x, dy = gy - X[t].y, dz = gz - X[t].z;
                const double dist = sqrt(dot3(dx, dy, dz, dx, dy, dz));
                front = dist < max_dist && p.qz <= 0.0;
            }
            if (VIS) {
                const bool k = front && p.u >= -1.0 && p.u <= 1.0 && p.v >= -1.0 && p.v <= 1.0;
                const double nan = __longlong_as_double(0x7ff8000000000000LL);
                if (valid) store16<NTS>(reinterpret_cast<char *>(uv_out + o), front ? make_double2(p.u, p.v) : make_double2(nan, nan));
                if (MODE == MODE_VISIBILITY_BITS) {
                    const uint64_t word = __builtin_amdgcn_ballot_w64(valid && k);       // tile0 is a multiple of 64
                    if (lane == 0) __builtin_nontemporal_store(word, reinterpret_cast<uint64_t *>(keep) + (tile0 >> 6));
                } else if (valid) {
                    if (NTS) __builtin_nontemporal_store((uint8_t)(k ? 1 : 0), keep + o);
                    else keep[o] = k ? 1 : 0;
                }
            } else if (MODE == MODE_PROJECT) {
                if (valid) store16<NTS>(reinterpret_cast<char *>(uv_out + o), make_double2(p.u, p.v));
            } else if (MODE == MODE_ERROR) {
                if (POWTAB) eacc += valid ? abs_pow_tab(p.u - ob.x, norm, (lds_tab)sTab) + abs_pow_tab(p.v - ob.y, norm, (lds_tab)sTab) : 0.0;
                else eacc += valid ? abs_pow_k<NK>(p.u - ob.x, norm) + abs_pow_k<NK>(p.v - ob.y, norm) : 0.0;
            } else {
                if (MODE == MODE_NOISE_ERROR12) {
                    // k_add_noise_observations' arithmetic, operation for operation (the stored uv is bit-identical)
                    double c, s;
                    const double z = obs_noise_draw(seed, (uint64_t)(obs_base + (valid ? o : n - 1)), (lds_tab)sTab, c, s);
                    const double r = 0.0 + norm * z;
                    ob.x = ob.x + c * r;
                    ob.y = ob.y + s * r;
                    if (valid) store16<NTS>(reinterpret_cast<char *>(uv_out + o), ob);
                }
                const double du = p.u - ob.x, dv = p.v - ob.y;
                eacc1 += valid ? abs_pow_k<NORM_1>(du, 1.0) + abs_pow_k<NORM_1>(dv, 1.0) : 0.0;
                eacc += valid ? abs_pow_k<NORM_2>(du, 2.0) + abs_pow_k<NORM_2>(dv, 2.0) : 0.0;
            }
        }
    }
    else if (kTabN > 0) {                                                // a wave past the end: its share of the table, the same barrier
#pragma unroll
        for (int k = 0; k < kTabTrips; ++k) {
            const int i = (int)threadIdx.x + k * WPB * 64;
            if (i < kTabN) sTab[i] = tabv[k];
        }
        __syncthreads();
    }
    if (MODE == MODE_ERROR) ticket_fold(wave_sum(eacc), sCamAll, block_part, ticket, out_sum);
    if (MODE == MODE_ERROR12 || MODE == MODE_NOISE_ERROR12) {
        const double w[2] = {wave_sum(eacc1), wave_sum(eacc)};            // out_sum[0] = L1, out_sum[1] = L2
        ticket_fold_n<2>(w, sCamAll, block_part, ticket, out_sum);
    }
}

// ---- residual + Jacobian, wave-centric form -------------------------------------------------------
// Every wave owns OPL consecutive tiles of 64 observations and never synchronises with another wave: it
// stages the (typically 3-7) cameras its observations touch in a wave-private LDS tile, transposes its
// Jacobian blocks through a wave-private slab (half a tile at a time) and leaves one error
// partial per tile.  With OPL = 2 each lane carries two observations: both tiles' index / uv loads, then
// both point gathers, are issued up front, so the second tile's memory latency hides behind the first
// tile's arithmetic and stores (the index -> gather chain is two dependent round trips per tile otherwise).

// one observation: projection (reference order) + the 2x9 / 2x3 blocks (explicit FMAs)
template <typename P>
C2B_DEV void jacobian_obs(P cam, const double4 X, const double2 ob, double &r0, double &r1,
                          double jc[18], double jp[6]) {
    const Proj p = project_obs(cam, X.x, X.y, X.z);
    r0 = p.u - ob.x; r1 = p.v - ob.y;
    const double f = cam[12], k1 = cam[13], k2 = cam[14];
    // -1/z by v_rcp_f64 + two Newton steps (~full precision, cheaper than an IEEE divide)
    double iz = __builtin_amdgcn_rcp(p.qz);
    iz = fma(fma(-p.qz, iz, 1.0), iz, iz);
    iz = fma(fma(-p.qz, iz, 1.0), iz, iz);
    const double s = -f * iz;                                   // f * (-1/z)
    const double c = fma(4.0 * k2, p.n, 2.0 * k1);              // 2 * d rad / d n
    const double cpx = c * p.px;
    const double B00 = fma(cpx, p.px, p.rad), B01 = cpx * p.py, B11 = fma(c * p.py, p.py, p.rad);
    const double g = fma(c, p.n, p.rad);
    const double a00 = s * B00, a01 = s * B01, a02 = s * p.px * g;
    const double a10 = s * B01, a11 = s * B11, a12 = s * p.py * g;
    // y = R X (= q - t);  v_i = y x a_i;  Jw_i = v_i^T J_l
    const double yx = p.qx - cam[9], yy = p.qy - cam[10], yz = p.qz - cam[11];
    const double v0x = fma(yy, a02, -yz * a01), v0y = fma(yz, a00, -yx * a02), v0z = fma(yx, a01, -yy * a00);
    const double v1x = fma(yy, a12, -yz * a11), v1y = fma(yz, a10, -yx * a12), v1z = fma(yx, a11, -yy * a10);
    const P Jl = cam + kJl;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        jc[j] = fma(v0z, Jl[6 + j], fma(v0y, Jl[3 + j], v0x * Jl[j]));
        jc[9 + j] = fma(v1z, Jl[6 + j], fma(v1y, Jl[3 + j], v1x * Jl[j]));
        jp[j] = fma(a02, cam[6 + j], fma(a01, cam[3 + j], a00 * cam[j]));
        jp[3 + j] = fma(a12, cam[6 + j], fma(a11, cam[3 + j], a10 * cam[j]));
    }
    jc[3] = a00; jc[4] = a01; jc[5] = a02;
    jc[12] = a10; jc[13] = a11; jc[14] = a12;
    const double fn = f * p.n, fnn = fn * p.n;
    jc[6] = p.rad * p.px;  jc[15] = p.rad * p.py;
    jc[7] = fn * p.px;     jc[16] = fn * p.py;
    jc[8] = fnn * p.px;    jc[17] = fnn * p.py;
}

// ---- residual + Jacobian: the kernel ---------------------------------------------------------------------
// WITH_ERR folds sum |r|^norm over all observations into out_sum[0] in the same launch (ticket_fold).  The wave-centric
// structure described above with the light kernels' economies: 32-bit observation indices, clamped (not predicated) loads, ONE inlined copy of the arithmetic per tile reading its camera through an
// LDS-typed pointer (first pass unmasked; lanes whose camera was not staged are served in extra rounds through a
// spare LDS slot and merged under a mask -- never taken on camera-major input), the observed uv requested per tile.
// OBUP: request every tile's observed uv up front with the indices (true) or when that tile's arithmetic starts (false)
// CSR: cam_idx points at the tile records of k_rows_pack (for this launch's first observation, which is observation
//      obs_base of the list row_ptr describes) instead of one camera index per observation
// NTL: bit 0 = non-temporal loads of the index streams, bit 1 = of the observed uv (streams read once per launch)
template <int NK, bool WITH_ERR, int WPB, bool NT, int OPL, int MINW, bool OBUP = true, bool CSR = false, int NTL = 0>
__global__ __launch_bounds__(WPB * 64, MINW) void k_residual_jacobian_l(
    const double *__restrict__ camblk, const double4 *__restrict__ pts4,
    const uint32_t *__restrict__ cam_idx, const uint32_t *__restrict__ pt_idx,
    const double2 *__restrict__ uv_obs, int n, int n_btiles, double norm,
    double2 *__restrict__ r_out, double *__restrict__ Jc, double *__restrict__ Jp,
    double *__restrict__ block_part, unsigned *__restrict__ ticket, double *__restrict__ out_sum,
    const uint64_t *__restrict__ row_ptr, int n_cam, int64_t obs_base) {
    constexpr int kSlab = 64 * 144 / 2;                                  // half a tile's 2x9 blocks (two rounds)
    constexpr int kCamBytes = 2 * kCamW * kCamHot * 8 + 64;              // staged cameras | slow-path slots | picked ids
    __shared__ __attribute__((aligned(16))) char smem[WPB * (kSlab + kCamBytes)];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int base = (xcd_tile32(blockIdx.x, n_btiles) * WPB + wave) * (OPL * 64);
    double eacc = 0.0;
    if (base < n) {                                                      // wave-uniform; waves past the end only fold
        uint32_t ci[OPL], pi[OPL];
        double2 obs_up[OPL];
        double4 X[OPL];
#pragma unroll
        for (int t = 0; t < OPL; ++t) {
            int o = base + t * 64 + lane;
            o = o < n ? o : n - 1;
            if (!CSR) ci[t] = (NTL & 1) ? __builtin_nontemporal_load(cam_idx + o) : cam_idx[o];
            pi[t] = (NTL & 1) ? __builtin_nontemporal_load(pt_idx + o) : pt_idx[o];
            if (OBUP) {
                if (NTL & 2) {
                    const d2_t t2 = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(uv_obs + o));
                    obs_up[t] = make_double2(t2.x, t2.y);
                } else {
                    obs_up[t] = uv_obs[o];
                }
            }
        }
        if (CSR) rows_cameras<OPL>(reinterpret_cast<const uint4 *>(cam_idx), row_ptr, n_cam, base, n, obs_base, lane, ci);
#pragma unroll
        for (int t = 0; t < OPL; ++t) X[t] = pts4[pi[t]];

        char *slab = smem + wave * (kSlab + kCamBytes);
        double *sCam = reinterpret_cast<double *>(slab + kSlab);
        double *sSlow = sCam + kCamW * kCamHot;
        uint32_t *sIdx = reinterpret_cast<uint32_t *>(sSlow + kCamW * kCamHot);
        const uint32_t c_first = __builtin_amdgcn_readfirstlane(ci[0]);
        const uint32_t c_last = __builtin_amdgcn_readlane(ci[OPL - 1], 63);
        uint32_t n_staged = c_last >= c_first ? c_last - c_first + 1 : 1;
        if (n_staged > (uint32_t)kCamW) n_staged = kCamW;
        // (addresses as a wave-uniform base -- the group of the first staged camera -- plus a 32-bit lane offset: the blocked table's
        // two-level index costs no 64-bit vector arithmetic, which this kernel has no registers for)
        const double *cam_base = camblk + (int64_t)(c_first >> 3) * (kCamGroup * kCamBlk);
        const int c_in_group = (int)(c_first & 7u);
        for (int ch = lane; ch < (int)n_staged * (kCamHot / 2); ch += 64) {
            const int k = ch / (kCamHot / 2), j = ch % (kCamHot / 2), cg = c_in_group + k;
            const int off = (cg >> 3) * (kCamGroup * kCamBlk) + cam_in_group_at(cg & 7, 2 * j);
            *reinterpret_cast<d2_t *>(sCam + k * kCamHot + 2 * j) = *reinterpret_cast<const d2_t *>(cam_base + off);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

#pragma unroll
        for (int t = 0; t < OPL; ++t) {
            const int tile0 = base + t * 64;
            if (tile0 >= n) break;                                       // wave-uniform
            const int o = tile0 + lane;
            const bool valid = o < n;
            const int n_wave = n - tile0 < 64 ? n - tile0 : 64;
            const double2 ob = OBUP ? obs_up[t] : uv_obs[valid ? o : n - 1];
            uint32_t local = ci[t] - c_first;
            bool in = local < n_staged;
            double r0, r1, jc[18], jp[6];
            jacobian_obs((lds_cptr)sCam + (in ? local : 0u) * kCamHot, X[t], ob, r0, r1, jc, jp);
            uint64_t todo = __builtin_amdgcn_ballot_w64(valid && !in);
            while (todo != 0) {                                          // wave-uniform; never taken on sorted input
                int my;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int ns = pick_cameras(todo, ci[t], lane, sIdx, my);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int ch = lane; ch < ns * (kCamHot / 2); ch += 64) {
                    const int k = ch / (kCamHot / 2), j = ch % (kCamHot / 2);
                    *reinterpret_cast<d2_t *>(sSlow + k * kCamHot + 2 * j) =
                        *reinterpret_cast<const d2_t *>(camblk + cam_chunk_at((int64_t)sIdx[k], j));
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                in = my >= 0;
                double q0, q1, qc[18], qp[6];
                jacobian_obs((lds_cptr)sSlow + (in ? my : 0) * kCamHot, X[t], ob, q0, q1, qc, qp);
                if (in) {
                    r0 = q0; r1 = q1;
#pragma unroll
                    for (int k = 0; k < 18; ++k) jc[k] = qc[k];
#pragma unroll
                    for (int k = 0; k < 6; ++k) jp[k] = qp[k];
                }
                todo &= ~__builtin_amdgcn_ballot_w64(in);
            }

            if (valid) store16<NT>(reinterpret_cast<char *>(r_out + o), make_double2(r0, r1));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if ((lane >> 5) == h) {
                    double2 *w = reinterpret_cast<double2 *>(slab + (lane & 31) * 144);
#pragma unroll
                    for (int k = 0; k < 9; ++k) w[k] = make_double2(jc[2 * k], jc[2 * k + 1]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                char *dst = reinterpret_cast<char *>(Jc) + ((int64_t)tile0 + h * 32) * 144;
                int nv = n_wave - h * 32;
                nv = nv < 0 ? 0 : (nv > 32 ? 32 : nv);
                const int bytes = nv * 144;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const int off = (k * 64 + lane) * 16;
                    if (off < bytes) store16<NT>(dst + off, *reinterpret_cast<const double2 *>(slab + off));
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            {
                double2 *w = reinterpret_cast<double2 *>(slab + lane * 48);
#pragma unroll
                for (int k = 0; k < 3; ++k) w[k] = make_double2(jp[2 * k], jp[2 * k + 1]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                char *dst = reinterpret_cast<char *>(Jp) + (int64_t)tile0 * 48;
                const int bytes = n_wave * 48;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int off = (k * 64 + lane) * 16;
                    if (off < bytes) store16<NT>(dst + off, *reinterpret_cast<const double2 *>(slab + off));
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (WITH_ERR) eacc += valid ? abs_pow_k<NK>(r0, norm) + abs_pow_k<NK>(r1, norm) : 0.0;
        }
    }
    if (WITH_ERR) ticket_fold(wave_sum(eacc), reinterpret_cast<double *>(smem), block_part, ticket, out_sum);
}

// ---- stats over camera centers ++ points ------------------------------------------------------------
// record: [0..2] sum(x/num) | [3..5] min | [6..8] max | [9] best dist | [10] best index
//
// Entity sources.  f64 problems read the camera centre from the derived camblk record; the f32 extension
// (BASELINE config 5) has no camblk and derives the centre from its f32 state on the fly.  Accumulation is
// always f64.
template <typename T> struct V4;
template <> struct V4<double> { typedef double4 type; };
template <> struct V4<float> { typedef float4 type; };

struct SrcBlk {                                  // f64: camera centres + pts4
    // Where camera i's centre lives: cen + i * cen_stride.  The compact centre table the camera-record kernels write
    // next to camblk (cen4[n_cam][4], 32-byte rows like pts4: stride 4) -- or, for a caller that kept none, the centre
    // field inside the 256-byte camblk records (stride kCamBlk: a whole 128-byte line fetched per camera for 24 useful
    // bytes -- 148.7 MB per pass over the --blocks 128 entities where the compact table moves 84.5, r04f / r05).
    const double *cen; int64_t cen_stride; const double4 *pts; int64_t n_cam;      // cen_stride 4: the centre table; 0: camblk itself
    static SrcBlk make(const double *camblk, const double *cen4, const double *pts4, int64_t n_cam) {
        return SrcBlk{cen4 ? cen4 : camblk, cen4 ? 4 : 0, reinterpret_cast<const double4 *>(pts4), n_cam};
    }
    // branch-free: both kinds of entity are three consecutive doubles at a 16-byte aligned address, so the address is
    // selected and the loads are unconditional -- a thread's batch of loads then issues back to back (with a branch per
    // entity every arm ended in s_waitcnt vmcnt(0) and the batch was serial again)
    C2B_DEV void get(int64_t i, double &x, double &y, double &z) const {
        const double *c = i < n_cam ? (cen_stride ? cen + i * cen_stride : cen + cam_center_at(i))
                                    : reinterpret_cast<const double *>(pts + (i - n_cam));
        const double2 xy = *reinterpret_cast<const double2 *>(c);
        x = xy.x; y = xy.y; z = c[2];
    }
};
struct SrcState32 {                              // f32: cam15 (float) + pts4 (float)
    const float *cam15; const float4 *pts; int64_t n_cam;
    C2B_DEV void get(int64_t i, double &x, double &y, double &z) const {
        if (i < n_cam) {
            const float *c = cam15 + 15 * i;
            float m[9], ctr[3];
#pragma unroll
            for (int k = 0; k < 9; ++k) m[k] = c[k];
            cm_center(m, c[9], c[10], c[11], ctr);
            x = ctr[0]; y = ctr[1]; z = ctr[2];
        } else {
            const float4 p = pts[i - n_cam];
            x = p.x; y = p.y; z = p.z;
        }
    }
};

// Sharded statistics: a rank holds cameras [cam_base, cam_base + n_cam_local) of n_cam_global and
// reduces points [pt_base, ...) of the replicated table.  Entity order of the reference = cameras, then points
// (src/baproblem.rs:282-289), so the GLOBAL index of local entity i is:
struct ShardMap {
    int64_t n_cam_local, cam_base, n_cam_global, pt_base;
    C2B_DEV double global_index(double local) const {
        const int64_t i = (int64_t)local;
        return (double)(i < n_cam_local ? cam_base + i : n_cam_global + pt_base + (i - n_cam_local));
    }
};

// closest to origin with fold1's semantics (src/noise.rs:80-86): strict <, ties -> later index.
// Only (distance, index) travel through the reduction; the winner's coordinates are re-read by
// index afterwards.  (Carrying xyz through a branchy merge was miscompiled by hipcc 7.2 -O3: the
// coordinate moves were emitted on the "no current best" path only.)  Index -1 = no element.
struct Best { double d, i; };
C2B_DEV Best best_merge(Best a, Best b) {
    const bool take = (a.i < 0.0) || (b.i >= 0.0 && (b.d < a.d || (b.d == a.d && b.i > a.i)));
    Best r;
    r.d = take ? b.d : a.d;
    r.i = take ? b.i : a.i;
    return r;
}

// Statistics (src/baproblem.rs:282-337, src/noise.rs:75-87).  Each pass leaves one record per workgroup, published like
// ticket_fold's partials (agent-scope write-through stores, drained, then the arrival count in the workspace's ticket
// block), and the workgroup that arrives last folds the records in a fixed order -- thread t takes records t, t + 256,
// ...; wave shuffle tree; waves in order -- and writes the result.  Rounds 1-2 used two passes and two more
// single-workgroup fold launches (73 us at --blocks 128; the 1024-thread fold kernel spilled: 128 VGPRs, 724-788 B of
// scratch).  A pass over the 2.6 M entities of that problem is bound by its bytes (148 MB of 128-byte lines for 24 useful
// bytes per camera: 26 us for the lightest possible pass, profiles/r03d), so the unsharded call now makes ONE pass:
// the standard deviation comes from per-thread (count, mean, M2) triples merged pairwise by Chan's update
//     delta = mean_b - mean_a;  mean = mean_a + delta n_b / n;  M2 = M2_a + M2_b + delta^2 n_a n_b / n
// (Chan, Golub, LeVeque 1979) -- no subtraction of large sums anywhere, so it is as well conditioned as the reference's
// second pass around the finished mean and agrees with it to rounding (tested at 1e-12).  The mean itself is still
// accumulated the reference's way, element by element scaled by 1/num.  Sharded cameras keep the two-pass form: the
// ranks exchange their shares between the passes anyway.
struct StatRec {
    double s[3], mn[3], mx[3];
    Best best;
    double cnt, mu[3], m2[3];                     // Chan triple per axis (shared count)
};
C2B_DEV StatRec stat_empty() {
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    StatRec r;
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.s[k] = 0.0; r.mn[k] = inf; r.mx[k] = -inf; r.mu[k] = 0.0; r.m2[k] = 0.0; }
    r.best.d = 0.0; r.best.i = -1.0;
    r.cnt = 0.0;
    return r;
}
template <bool STD>
C2B_DEV void stat_merge(StatRec &a, const StatRec &b) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { a.s[k] += b.s[k]; a.mn[k] = fmin(a.mn[k], b.mn[k]); a.mx[k] = fmax(a.mx[k], b.mx[k]); }
    a.best = best_merge(a.best, b.best);
    if (STD) {
        const double tot = a.cnt + b.cnt;
        const double f = tot > 0.0 ? b.cnt / tot : 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double delta = b.mu[k] - a.mu[k];
            a.mu[k] = a.mu[k] + delta * f;
            a.m2[k] = (a.m2[k] + b.m2[k]) + (delta * delta) * (a.cnt * f);
        }
        a.cnt = tot;
    }
}
// all 64 lanes -> lane 0 (shuffle trees, fixed order), field by field.
// The moments (r05): NOT Chan's update level by level -- six dependent IEEE divisions per tree, and the workgroup's waves
// then merged one after the other by thread 0 with one more each: measured, these reductions were ~11 of the pass's 29 us
// (512-thread workgroups, four more serial merges at both levels, cost 7 us more than 256-thread ones whatever the
// loop's shape, profiles/r05a_ab_stats_shapes.txt).  Instead the textbook combination of partial (count, mean, M2)
// triples in one step:  N = sum n_i;  mean = sum n_i mean_i / N;  M2 = sum (M2_i + n_i (mean_i - mean)^2)  -- the
// deviations are taken from the COMBINED mean, so it is as well conditioned as Chan's pairwise form (both equal the
// reference's second pass around the finished mean up to rounding; tested at 1e-12) and costs seven wave sums, one
// reciprocal and two broadcasts.
C2B_DEV double wave_bcast0(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
// LANES (a power of two <= 64): how many leading lanes hold records; the tree starts at LANES / 2.
// LEVEL-MAJOR: at every level all the fields' partner values are requested first and combined afterwards.  A lane
// shuffle on gfx9 is a ds_bpermute -- a round trip through the LDS crossbar of ~100 cycles -- so sixteen field-by-field
// trees of six dependent shuffles each cost ~7 us per workgroup (measured with the tuning build's time-stamp probe, r05:
// the two block reductions were 14 of the pass's 33 us); a level's shuffles issued back to back overlap their latencies.
// The value `OFF` lanes further down the wave (OFF a power of two), as far as a reduction TO LANE 0 needs it: lanes
// whose partner lies in another row of 16 get their own value back for OFF < 16 (their sums are never read).  OFF < 16
// is a DPP row shift -- register to register at vector-ALU rate -- where __shfl_down is a ds_bpermute through the LDS
// crossbar, ~100 cycles of latency and contended by every wave of the CU that reduces at the same moment (the block
// reduction of 4 waves: 3.9 us with bpermutes at every level, time-stamp probe, r05).
template <int OFF>
C2B_DEV double lane_down(double v) {
    if constexpr (OFF >= 16) {
        return __shfl_down(v, OFF, 64);
    } else {
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x100 + OFF, 0xf, 0xf, false);     // row_shl:OFF -- lane i reads lane i + OFF of its row
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x100 + OFF, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    }
}
// one level of the reduction: all the fields' partner values first, then the combinations (level-major: the shuffles of
// a level overlap their latencies; field by field, sixteen trees of six dependent shuffles cost ~7 us per workgroup)
template <bool STD, int OFF>
C2B_DEV void stat_level1(StatRec &r, double &tn, double (&cm)[3]) {
    double s[3], mn[3], mx[3], oc = 0.0, ocm[3] = {0.0, 0.0, 0.0};
    Best o;
#pragma unroll
    for (int k = 0; k < 3; ++k) { s[k] = lane_down<OFF>(r.s[k]); mn[k] = lane_down<OFF>(r.mn[k]); mx[k] = lane_down<OFF>(r.mx[k]); }
    o.d = lane_down<OFF>(r.best.d);
    o.i = lane_down<OFF>(r.best.i);
    if (STD) {
        oc = lane_down<OFF>(tn);
#pragma unroll
        for (int k = 0; k < 3; ++k) ocm[k] = lane_down<OFF>(cm[k]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.s[k] += s[k]; r.mn[k] = fmin(r.mn[k], mn[k]); r.mx[k] = fmax(r.mx[k], mx[k]); }
    r.best = best_merge(r.best, o);
    if (STD) {
        tn += oc;
#pragma unroll
        for (int k = 0; k < 3; ++k) cm[k] += ocm[k];
    }
    if constexpr (OFF > 1) stat_level1<STD, OFF / 2>(r, tn, cm);
}
template <int OFF>
C2B_DEV void stat_level2(double (&q)[3]) {
    double oq[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) oq[k] = lane_down<OFF>(q[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) q[k] += oq[k];
    if constexpr (OFF > 1) stat_level2<OFF / 2>(q);
}
// LANES (a power of two, 2 ... 64): how many leading lanes hold records; the tree starts at LANES / 2 -> lane 0.
template <bool STD, int LANES = 64>
C2B_DEV void stat_wave_reduce(StatRec &r) {
    static_assert(LANES >= 2 && LANES <= 64 && (LANES & (LANES - 1)) == 0, "a power of two");
    // phase 1: sums, extremes, the origin candidate, the count and the count-weighted means
    double tn = r.cnt, cm[3] = {r.cnt * r.mu[0], r.cnt * r.mu[1], r.cnt * r.mu[2]};
    stat_level1<STD, LANES / 2>(r, tn, cm);
    if (STD) {
        // phase 2: every lane's M2 moved to the combined mean (lane 0 holds the totals), then summed
        const double n_all = wave_bcast0(tn);
        const double inv = n_all > 0.0 ? 1.0 / n_all : 0.0;
        double mean[3], q[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            mean[k] = wave_bcast0(cm[k]) * inv;
            const double d = r.mu[k] - mean[k];
            q[k] = r.m2[k] + r.cnt * (d * d);
        }
        stat_level2<LANES / 2>(q);
        r.cnt = n_all;
#pragma unroll
        for (int k = 0; k < 3; ++k) { r.mu[k] = mean[k]; r.m2[k] = q[k]; }
    }
}
C2B_DEV void stat_to_lds(const StatRec &r, double *o) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { o[k] = r.s[k]; o[3 + k] = r.mn[k]; o[6 + k] = r.mx[k]; o[12 + k] = r.mu[k]; o[15 + k] = r.m2[k]; }
    o[9] = r.best.d; o[10] = r.best.i; o[11] = r.cnt;
}
C2B_DEV StatRec stat_from(const double *o) {
    StatRec r;
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.s[k] = o[k]; r.mn[k] = o[3 + k]; r.mx[k] = o[6 + k]; r.mu[k] = o[12 + k]; r.m2[k] = o[15 + k]; }
    r.best.d = o[9]; r.best.i = o[10]; r.cnt = o[11];
    return r;
}

// the workgroup's WAVES waves -> one record on thread 0: every wave reduces to its lane 0, the WAVES wave records go
// through LDS to the first lanes of wave 0, which reduces them the same way (lanes beyond WAVES hold the identity)
template <bool STD, int WAVES = kWaves>
C2B_DEV StatRec stat_block_reduce(StatRec r, double (*sh)[kStatRec]) {
    static_assert(WAVES <= 64, "one lane of wave 0 per wave");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    stat_wave_reduce<STD>(r);
    __syncthreads();
    if (lane == 0) stat_to_lds(r, sh[wave]);
    __syncthreads();
    if (wave == 0) {                                             // wave-uniform
        r = stat_empty();
        if (lane < WAVES) r = stat_from(sh[lane]);
        stat_wave_reduce<STD, WAVES>(r);                          // log2(WAVES) levels
    }
    return r;
}

// One batch of a thread's entities into the running record: mean sums, min / max, the origin search, and (STD) the
// batch's own (count, mean, M2) triple merged by Chan's update.  j0 + u * step = the entity index, valid while < lim.
template <bool STD, int BATCH, bool FULL>
C2B_DEV void stat_accumulate_impl(StatRec &a, double &best_thr, const double (&x)[BATCH], const double (&y)[BATCH],
                                  const double (&z)[BATCH], int64_t j0, int64_t step, int64_t lim, double inv_num) {
    double c = 0.0, bs[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
        const int64_t j = j0 + u * step;
        if (FULL || j < lim) {
            a.s[0] += x[u] * inv_num; a.s[1] += y[u] * inv_num; a.s[2] += z[u] * inv_num;
            a.mn[0] = fmin(a.mn[0], x[u]); a.mn[1] = fmin(a.mn[1], y[u]); a.mn[2] = fmin(a.mn[2], z[u]);
            a.mx[0] = fmax(a.mx[0], x[u]); a.mx[1] = fmax(a.mx[1], y[u]); a.mx[2] = fmax(a.mx[2], z[u]);
            const double d2 = dot3(x[u], y[u], z[u], x[u], y[u], z[u]);
            if (d2 <= best_thr) {                            // rare after the first few entities of a thread
                const double d = sqrt(d2);
                if (a.best.i < 0.0 || d <= a.best.d) {       // j grows within a thread: an equal distance is the later entity
                    a.best.d = d; a.best.i = (double)j;
                    best_thr = (d * d) * (1.0 + 0x1.0p-49);
                }
            }
            if (STD) { c += 1.0; bs[0] += x[u]; bs[1] += y[u]; bs[2] += z[u]; }
        }
    }
    if (STD && (FULL || c > 0.0)) {                          // the batch's own triple (two passes over registers), then Chan
        StatRec bt = stat_empty();
        const double ic = 1.0 / c;
        bt.cnt = c;
        bt.mu[0] = bs[0] * ic; bt.mu[1] = bs[1] * ic; bt.mu[2] = bs[2] * ic;
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            if (FULL || j0 + u * step < lim) {
                const double dx = x[u] - bt.mu[0], dy = y[u] - bt.mu[1], dz = z[u] - bt.mu[2];
                bt.m2[0] += dx * dx; bt.m2[1] += dy * dy; bt.m2[2] += dz * dz;
            }
        }
        const double tot = a.cnt + c, f = c / tot;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double delta = bt.mu[k] - a.mu[k];
            a.mu[k] = a.mu[k] + delta * f;
            a.m2[k] = (a.m2[k] + bt.m2[k]) + (delta * delta) * (a.cnt * f);
        }
        a.cnt = tot;
    }
}
// a batch whose last entity is in range (all but a thread's last trip) skips the per-entity range checks
template <bool STD, int BATCH>
C2B_DEV void stat_accumulate(StatRec &a, double &best_thr, const double (&x)[BATCH], const double (&y)[BATCH],
                             const double (&z)[BATCH], int64_t j0, int64_t step, int64_t lim, double inv_num) {
    if (j0 + (BATCH - 1) * step < lim) stat_accumulate_impl<STD, BATCH, true>(a, best_thr, x, y, z, j0, step, lim, inv_num);
    else stat_accumulate_impl<STD, BATCH, false>(a, best_thr, x, y, z, j0, step, lim, inv_num);
}

// stats[0..2]=mean, [6..8]=min, [9..11]=max, [12..14]=dim, [15..17]=origin (re-read by index), [18]=origin's GLOBAL
// index (map).  STD (the unsharded call): [3..5]=std, [19]=|std| from the (count, mean, M2) triples, all in this one
// launch.  !STD (a shard's share, c2b_stats_partial_pass1): [19]=the origin's distance (compared across ranks), no moments.
// rec: gridDim.x records of kStatRec doubles in the workspace.
//
// Where its time goes (r05, the tuning build's time-stamp probe at --blocks 128, 2.64 M entities, profiles/r05g_*): the
// entity loop moves its 84.5 MB (32 bytes per entity since the compact centre table; 148.7 MB before) in ~11 us --
// 7.7 TB/s, and no faster with twice the workgroups, 512-thread workgroups, a software pipeline or contiguous chunks per
// workgroup: a bandwidth figure, not a latency one -- and everything else is a chain of fixed latencies: ~1 us until
// the last workgroup has started, the workgroup's reduction (3.3 us), the record's publication and the arrival count
// (1 us), and in the workgroup that arrives last the records (1.8 us), their reduction (3.1 us) and the results (0.8 us).
// Those reductions were 7.4 + 6.6 us while every level of every field's shuffle tree was a dependent ds_bpermute with an
// IEEE division in Chan's update (stat_wave_reduce): 33 -> 25 us for the pass.  512-thread workgroups (4 waves per SIMD
// instead of 2) make the loop no faster and both reductions slower (8 wave records through LDS): 256 threads ship.
//
// The origin search takes no square root per entity: fold1 (src/noise.rs:80-86) compares the ROUNDED distances
// sqrt(x.x), so the exact rule "the latest entity among those whose rounded distance is smallest" is kept by comparing
// the squared distance with a threshold just above the square of the thread's current best distance -- 8 ulps above,
// where 1 would do: sqrt is monotone, so a squared distance beyond it cannot round to a distance <= the best -- and only
// an entity under the threshold (a handful per thread: the running minimum of a sequence improves O(log n) times) takes
// the square root and the exact comparison.  NaN coordinates never become the origin (the reference's fold would let
// the LAST NaN win; not reproduced, like every NaN rule of the reductions).
//
// PIPE: the loads of a thread's NEXT batch are issued before the current batch is accumulated (clamped, so that every
// trip issues the same number of loads and the wait in front of the arithmetic is a count, not "everything") -- a trip
// then costs the longer of a memory round trip and its ~400 vector instructions instead of their sum.
// CHUNK: workgroup b owns the contiguous entities [b * per, (b + 1) * per) (a thread's batch is BLOCK apart) instead of
// every gridDim.x-th slab of the whole table (a thread's batch gridDim.x * BLOCK apart).
template <typename Src, bool STD, int BLOCK = kStatBlock, int BATCH = kStatBatch, bool PIPE = kStatPipe, bool CHUNK = kStatChunk>
__global__ __launch_bounds__(BLOCK, (BLOCK >= 512 ? 4 : 1)) void k_stats_pass1(Src src, int64_t n, double num, double *__restrict__ rec,
                                                       unsigned *__restrict__ ticket, ShardMap map,
                                                       double *__restrict__ stats) {
    constexpr int WAVES = BLOCK / 64;
    __shared__ double sh[WAVES + 1][kStatRec];
    unsigned magic = 0u;
    if (threadIdx.x == 0) magic = ticket[kTicketMagicAt];
    StatRec a = stat_empty();
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    double best_thr = inf;                                  // squared distances above it cannot round to <= a.best.d
    // BATCH entities per thread and trip, all loads issued before the first use (indices past the end re-read the
    // last entity and are not accumulated).
    // mean(), src/baproblem.rs:282-289, folds `a + b / num` over the entities; here every element is scaled by the
    // one reciprocal instead of divided (three IEEE divides per entity were a third of this pass's instructions).  The
    // sums are tree-ordered already, so the last bits differ from the sequential fold either way (tested at 1e-12).
    const double inv_num = 1.0 / num;
    int64_t first, step, lim;                               // this thread's entities: first, first + step, ... while < lim
    if (CHUNK) {
        const int64_t per = ((n + gridDim.x - 1) / gridDim.x + BLOCK - 1) / BLOCK * BLOCK;
        first = (int64_t)blockIdx.x * per + threadIdx.x;
        step = BLOCK;
        lim = (int64_t)(blockIdx.x + 1) * per < n ? (int64_t)(blockIdx.x + 1) * per : n;
    } else {
        first = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
        step = (int64_t)gridDim.x * BLOCK;
        lim = n;
    }
    auto load = [&](int64_t j0, double (&x)[BATCH], double (&y)[BATCH], double (&z)[BATCH]) {
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int64_t j = j0 + u * step;
            src.get(j < lim ? j : n - 1, x[u], y[u], z[u]);
        }
    };
    if (PIPE) {
        double x0[BATCH], y0[BATCH], z0[BATCH], x1[BATCH], y1[BATCH], z1[BATCH];
        if (first < lim) load(first, x0, y0, z0);
        for (int64_t i = first; i < lim; i += 2 * step * BATCH) {          // two trips per iteration: the register sets swap roles
            load(i + step * BATCH, x1, y1, z1);                            // past the end: clamped re-reads, not accumulated
            stat_accumulate<STD, BATCH>(a, best_thr, x0, y0, z0, i, step, lim, inv_num);
            load(i + 2 * step * BATCH, x0, y0, z0);
            stat_accumulate<STD, BATCH>(a, best_thr, x1, y1, z1, i + step * BATCH, step, lim, inv_num);
        }
    } else {
        for (int64_t i = first; i < lim; i += step * BATCH) {
            double x[BATCH], y[BATCH], z[BATCH];
            load(i, x, y, z);
            stat_accumulate<STD, BATCH>(a, best_thr, x, y, z, i, step, lim, inv_num);
        }
    }
    a = stat_block_reduce<STD, WAVES>(a, sh);
    if (threadIdx.x == 0) {
        double t[kStatRec];
        stat_to_lds(a, t);
        double *o = rec + (int64_t)blockIdx.x * kStatRec;
#pragma unroll
        for (int k = 0; k < (STD ? 18 : 11); ++k) __hip_atomic_store(o + k, t[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        const bool last = ticket_arrive(ticket, magic);
        if (magic != kTicketMagic) {
            const double nan = __longlong_as_double(0x7ff8000000000000LL);
            for (int k = 0; k < 20; ++k) stats[k] = nan;
        }
        sh[WAVES][0] = last ? 1.0 : 0.0;
    }
    __syncthreads();
    if (sh[WAVES][0] == 0.0) return;                   // workgroup-uniform
    // the last workgroup: thread t takes records t, t + BLOCK, ...  The first two (all there are at the shipped grid of
    // 2 x BLOCK workgroups) are loaded together -- one round trip to the L2, not two -- and the first is taken as it
    // is instead of being merged into an empty record (a division saved on the critical path).
    auto load_rec = [&](unsigned r) {
        const double *q = rec + (int64_t)r * kStatRec;
        StatRec g = stat_empty();
#pragma unroll
        for (int k = 0; k < 3; ++k) { g.s[k] = q[k]; g.mn[k] = q[3 + k]; g.mx[k] = q[6 + k]; }
        g.best.d = q[9]; g.best.i = q[10];
        if (STD) {
            g.cnt = q[11];
#pragma unroll
            for (int k = 0; k < 3; ++k) { g.mu[k] = q[12 + k]; g.m2[k] = q[15 + k]; }
        }
        return g;
    };
    StatRec f = stat_empty();
    if (threadIdx.x < gridDim.x) {
        const unsigned r1 = threadIdx.x + BLOCK;
        const StatRec g0 = load_rec(threadIdx.x);
        const StatRec g1 = load_rec(r1 < gridDim.x ? r1 : threadIdx.x);      // clamped: both loads always issue
        f = g0;
        if (r1 < gridDim.x) stat_merge<STD>(f, g1);
        for (unsigned r = r1 + BLOCK; r < gridDim.x; r += BLOCK) stat_merge<STD>(f, load_rec(r));
    }
    f = stat_block_reduce<STD, WAVES>(f, sh);
    if (threadIdx.x != 0) return;
    double x = 0, y = 0, z = 0;
    if (f.best.i >= 0.0) src.get((int64_t)f.best.i, x, y, z);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        stats[k] = f.s[k];
        stats[6 + k] = f.mn[k];
        stats[9 + k] = f.mx[k];
        stats[12 + k] = f.mx[k] - f.mn[k];
    }
    stats[15] = x; stats[16] = y; stats[17] = z;
    stats[18] = f.best.i >= 0.0 ? map.global_index(f.best.i) : -1.0;
    if (STD) {
        const double sa = sqrt(f.m2[0] / num), sb = sqrt(f.m2[1] / num), sc = sqrt(f.m2[2] / num);
        stats[3] = sa; stats[4] = sb; stats[5] = sc;
        stats[19] = sqrt(dot3(sa, sb, sc, sa, sb, sc));      // |std()|  (InnerSpace::magnitude)
    } else {
        stats[19] = f.best.i >= 0.0 ? f.best.d : inf;
    }
}

// pass 2 (sharded statistics only since r03): sums of squared deviations from mean3.  RAW: leave the three sums in
// out[0..2] (a shard's share; the ranks' sums are gathered and finished on the host) instead of finishing std in
// out[3..5] and |std| in out[19].
template <typename Src, bool RAW>
__global__ __launch_bounds__(kBlock) void k_stats_pass2(Src src, int64_t n, const double *__restrict__ mean3,
                                                       double *__restrict__ rec, unsigned *__restrict__ ticket,
                                                       double *__restrict__ out) {
    __shared__ double sh[kWaves + 1][4];
    unsigned magic = 0u;
    if (threadIdx.x == 0) magic = ticket[kTicketMagicAt];
    const double m0 = mean3[0], m1 = mean3[1], m2 = mean3[2];
    double s0 = 0, s1 = 0, s2 = 0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride * kStatBatch) {
        double x[kStatBatch], y[kStatBatch], z[kStatBatch];
#pragma unroll
        for (int u = 0; u < kStatBatch; ++u) {
            const int64_t j = i + u * stride;
            src.get(j < n ? j : n - 1, x[u], y[u], z[u]);
        }
#pragma unroll
        for (int u = 0; u < kStatBatch; ++u) {
            if (i + u * stride < n) {
                s0 += (x[u] - m0) * (x[u] - m0); s1 += (y[u] - m1) * (y[u] - m1); s2 += (z[u] - m2) * (z[u] - m2);
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto block_sum = [&](double &a, double &b, double &c) {          // -> thread 0
        a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
        __syncthreads();
        if (lane == 0) { sh[wave][0] = a; sh[wave][1] = b; sh[wave][2] = c; }
        __syncthreads();
        if (threadIdx.x == 0) {
            a = (sh[0][0] + sh[1][0]) + (sh[2][0] + sh[3][0]);
            b = (sh[0][1] + sh[1][1]) + (sh[2][1] + sh[3][1]);
            c = (sh[0][2] + sh[1][2]) + (sh[2][2] + sh[3][2]);
        }
    };
    block_sum(s0, s1, s2);
    if (threadIdx.x == 0) {
        double *o = rec + (int64_t)blockIdx.x * kStatRec;
        __hip_atomic_store(o, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 1, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 2, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        const bool last = ticket_arrive(ticket, magic);
        if (magic != kTicketMagic) {
            const double nan = __longlong_as_double(0x7ff8000000000000LL);
            if (RAW) { out[0] = nan; out[1] = nan; out[2] = nan; }
            else { out[3] = nan; out[4] = nan; out[5] = nan; out[19] = nan; }
        }
        sh[kWaves][0] = last ? 1.0 : 0.0;
    }
    __syncthreads();
    if (sh[kWaves][0] == 0.0) return;                  // workgroup-uniform
    double t0 = 0, t1 = 0, t2 = 0;
    for (unsigned r = threadIdx.x; r < gridDim.x; r += kBlock) {
        const double *q = rec + (int64_t)r * kStatRec;
        t0 += q[0]; t1 += q[1]; t2 += q[2];
    }
    block_sum(t0, t1, t2);
    if (threadIdx.x != 0) return;
    if (RAW) { out[0] = t0; out[1] = t1; out[2] = t2; return; }
    const double num = (double)n;
    const double a = sqrt(t0 / num), b = sqrt(t1 / num), c = sqrt(t2 / num);
    out[3] = a; out[4] = b; out[5] = c;
    out[19] = sqrt(dot3(a, b, c, a, b, c));            // |std()|  (InnerSpace::magnitude)
}

// ---- noise kernels -------------------------------------------------------------------------------------
// Templated on the state scalar T (double = the parity path; float = config-5 extension).  Draws are always
// the f64 Philox/Box-Muller normals, rounded to T, so the f32 results track the f64 ones to f32 accuracy.

// add_drift, src/noise.rs:68-116.  One lane per entity (cameras first, then points).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_add_drift(T *__restrict__ cam15, int64_t n_cam,
                                                     typename V4<T>::type *__restrict__ pts4, int64_t n_pts,
                                                     const double *__restrict__ origin, double strength_d,
                                                     double angle_strength_d, double std_d, double dx_d, double dy_d,
                                                     double dz_d, const double *__restrict__ stats_norm,
                                                     uint64_t seed, int64_t cam_base) {
    // waves that hold cameras only write their 64 records back through LDS (wave_rows_store: whole lines); the one wave
    // that straddles the camera / point boundary stores per lane.  Reads stay per lane: a record's 15 strided loads re-use
    // their lines in L1, and staging them through LDS measured slower (r03q)
    __shared__ T sRows[kWaves][64 * 15];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * kBlock + wave * 64;
    const bool coop = wave0 + 64 <= n_cam;                                // wave-uniform
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_cam + n_pts) return;
    if (stats_norm) {
        // add_drift_normalized, src/noise.rs:47-56: dir = std().normalize(), strength *= |std()|
        const double s0 = stats_norm[3], s1 = stats_norm[4], s2 = stats_norm[5];
        const double mag = sqrt(dot3(s0, s1, s2, s0, s1, s2));
        const double inv = 1.0 / mag;
        dx_d = s0 * inv; dy_d = s1 * inv; dz_d = s2 * inv;
        strength_d = strength_d * mag;
    }
    const T dx = (T)dx_d, dy = (T)dy_d, dz = (T)dz_d, strength = (T)strength_d, angle_strength = (T)angle_strength_d;
    const T ox = (T)origin[0], oy = (T)origin[1], oz = (T)origin[2];
    if (i < n_cam) {
        T c[15], ctr[3], dR[9];
        double z0, z1;
#pragma unroll
        for (int k = 0; k < 15; ++k) c[k] = cam15[15 * i + k];         // strided reads re-use their lines in L1
        cm_center(c, c[9], c[10], c[11], ctr);
        const T ex = ctr[0] - ox, ey = ctr[1] - oy, ez = ctr[2] - oz;
        const T distance = sqrt(dot3(ex, ey, ez, ex, ey, ez));
        normal_pair(seed, kStreamDriftCam, (uint64_t)(cam_base + i), 0, z0, z1);     // draws keyed by the GLOBAL camera index
        const T va = (T)(1.0 + std_d * z0);             // angle draw first (src/noise.rs:104-107)
        const T vt = (T)(1.0 + std_d * z1);
        const T angle = angle_strength * va * pow_t(distance, (T)1.2);
        T sn, cs;
        sincos_t(angle, &sn, &cs);
        dR[0] = 1; dR[1] = 0; dR[2] = 0; dR[3] = 0; dR[4] = cs; dR[5] = sn; dR[6] = 0; dR[7] = -sn; dR[8] = cs;
        transform_cam15(c, dR, dx * strength * vt * distance * distance,
                        dy * strength * vt * distance * distance, dz * strength * vt * distance * distance);
        if (coop) {                                  // whole records (the three intrinsics unchanged) leave as whole lines
#pragma unroll
            for (int k = 0; k < 15; ++k) sRows[wave][15 * lane + k] = c[k];
            wave_rows_store<15>(cam15 + 15 * wave0, 64, sRows[wave], lane);
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) cam15[15 * i + k] = c[k];
        }
    } else {
        const int64_t j = i - n_cam;
        typename V4<T>::type p = pts4[j];
        const T ex = p.x - ox, ey = p.y - oy, ez = p.z - oz;
        const T distance = sqrt(dot3(ex, ey, ez, ex, ey, ez));
        double z0, z1;
        normal_pair(seed, kStreamDriftPt, (uint64_t)j, 0, z0, z1);
        const T v = (T)(1.0 + std_d * z0);
        p.x = p.x + dx * strength * v * distance * distance;
        p.y = p.y + dy * strength * v * distance * distance;
        p.z = p.z + dz * strength * v * distance * distance;
        pts4[j] = p;
    }
}

// add_noise cameras + points, src/noise.rs:129-150
template <typename T>
__global__ __launch_bounds__(kBlock) void k_add_noise_entities(T *__restrict__ cam15, int64_t n_cam,
                                                              typename V4<T>::type *__restrict__ pts4, int64_t n_pts,
                                                              const double *__restrict__ stats,
                                                              double translation_std, double rotation_std,
                                                              double point_std, uint64_t seed, int64_t cam_base) {
    __shared__ T sRows[kWaves][64 * 15];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * kBlock + wave * 64;
    const bool coop = wave0 + 64 <= n_cam;                                // wave-uniform (see k_add_drift)
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_cam + n_pts) return;
    if (i < n_cam) {
        const T bal_std = (T)stats[19];
        T c[15], dR[9];
        double a0, a1, a2, rot, b0, b1, b2, tr;
#pragma unroll
        for (int k = 0; k < 15; ++k) c[k] = cam15[15 * i + k];         // strided reads re-use their lines in L1
        normal_pair(seed, kStreamNoiseCam, (uint64_t)(cam_base + i), 0, a0, a1);
        normal_pair(seed, kStreamNoiseCam, (uint64_t)(cam_base + i), 1, a2, rot);
        normal_pair(seed, kStreamNoiseCam, (uint64_t)(cam_base + i), 2, b0, b1);
        normal_pair(seed, kStreamNoiseCam, (uint64_t)(cam_base + i), 3, b2, tr);
        const T A0 = (T)a0, A1 = (T)a1, A2 = (T)a2, B0 = (T)b0, B1 = (T)b1, B2 = (T)b2;
        const T ia = (T)1.0 / sqrt(dot3(A0, A1, A2, A0, A1, A2));
        const T ib = (T)1.0 / sqrt(dot3(B0, B1, B2, B0, B1, B2));
        const T ang = (T)(0.0 + rotation_std * rot);
        const T t = (T)(0.0 + translation_std * tr);
        cm_from_axis_angle(A0 * ia, A1 * ia, A2 * ia, ang, dR);
        transform_cam15(c, dR, (B0 * ib) * bal_std * t, (B1 * ib) * bal_std * t, (B2 * ib) * bal_std * t);
        if (coop) {                                  // whole records (the three intrinsics unchanged) leave as whole lines
#pragma unroll
            for (int k = 0; k < 15; ++k) sRows[wave][15 * lane + k] = c[k];
            wave_rows_store<15>(cam15 + 15 * wave0, 64, sRows[wave], lane);
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) cam15[15 * i + k] = c[k];
        }
    } else {
        const int64_t j = i - n_cam;
        double a0, a1, a2, m;
        normal_pair(seed, kStreamNoisePt, (uint64_t)j, 0, a0, a1);
        normal_pair(seed, kStreamNoisePt, (uint64_t)j, 1, a2, m);
        const T A0 = (T)a0, A1 = (T)a1, A2 = (T)a2;
        const T ia = (T)1.0 / sqrt(dot3(A0, A1, A2, A0, A1, A2));
        const T mm = (T)(0.0 + point_std * m);
        typename V4<T>::type p = pts4[j];
        p.x = p.x + (A0 * ia) * mm; p.y = p.y + (A1 * ia) * mm; p.z = p.z + (A2 * ia) * mm;
        pts4[j] = p;
    }
}

// add_noise observations, src/noise.rs:152-170
// One Philox block per observation (camera_math.hpp: obs_noise_draw): the direction (cos, sin)(2 pi u) of unit_random's
// pair -- its radius cancels in the normalisation, so neither it nor the normalisation (sqrt + two divides) is evaluated --
// and the Box-Muller magnitude.  r05: the draw reads its logarithm and its angles from tables a workgroup stages in LDS once
// (10 KB), so a workgroup walks many tiles of 64 observations (grid-stride, at most kNoiseGrid workgroups) instead of one
// thread per observation: 95 vector instructions per tile (SQ_INSTS_VALU) where r04 had 186.
constexpr int kNoiseGrid = 2048;               // 8 workgroups of 4 waves per CU
__global__ __launch_bounds__(kBlock) void k_add_noise_observations(double2 *__restrict__ uv, int64_t n, int64_t n_tiles,
                                                                  int64_t obs_base, double observations_std,
                                                                  uint64_t seed) {
    __shared__ __attribute__((aligned(16))) tab2_t sTab[kNoiseTab];
    noise_tab_stage(sTab, threadIdx.x, kBlock);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kWaves = kBlock / 64;
    for (int64_t t = (int64_t)blockIdx.x * kWaves + wave; t < n_tiles; t += (int64_t)gridDim.x * kWaves) {
        const int64_t o = t * 64 + lane;
        if (o < n) {
            double2 v = uv[o];
            double c, s;
            const double z = obs_noise_draw(seed, (uint64_t)(o + obs_base), (lds_tab)sTab, c, s);
            const double r = 0.0 + observations_std * z;
            v.x = v.x + c * r;
            v.y = v.y + s * r;
            uv[o] = v;
        }
    }
}

// add_sin_noise, src/noise.rs:388-416
template <typename T>
__global__ __launch_bounds__(kBlock) void k_add_sin_noise(T *__restrict__ cam15, int64_t n_cam,
                                                         typename V4<T>::type *__restrict__ pts4, int64_t n_pts,
                                                         const double *__restrict__ stats, double dx_d, double dy_d,
                                                         double dz_d, double nx_d, double ny_d, double nz_d,
                                                         double strength_d, double frequency_d) {
    __shared__ T sRows[kWaves][64 * 15];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * kBlock + wave * 64;
    const bool coop = wave0 + 64 <= n_cam;                                // wave-uniform (see k_add_drift)
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_cam + n_pts) return;
    double e0 = stats[12], e1 = stats[13], e2 = stats[14];
    if (e0 == 0.0) e0 = 1e-8;
    if (e1 == 0.0) e1 = 1e-8;
    if (e2 == 0.0) e2 = 1e-8;
    const T d0 = (T)e0, d1 = (T)e1, d2 = (T)e2;
    T nx = (T)nx_d, ny = (T)ny_d, nz = (T)nz_d;
    const T dx = (T)dx_d, dy = (T)dy_d, dz = (T)dz_d, strength = (T)strength_d, frequency = (T)frequency_d;
    const T inv = (T)1.0 / sqrt(dot3(nx, ny, nz, nx, ny, nz));
    nx *= inv; ny *= inv; nz *= inv;
    if (i < n_cam) {
        T c[15], ctr[3];
#pragma unroll
        for (int k = 0; k < 15; ++k) c[k] = cam15[15 * i + k];         // strided reads re-use their lines in L1
        cm_center(c, c[9], c[10], c[11], ctr);
        const T s = sin(dot3(ctr[0] / d0, ctr[1] / d1, ctr[2] / d2, dx, dy, dz) * frequency * (T)kPi) * strength;
        const T I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        transform_cam15(c, I, nx * s, ny * s, nz * s);
        if (coop) {                                  // whole records (the three intrinsics unchanged) leave as whole lines
#pragma unroll
            for (int k = 0; k < 15; ++k) sRows[wave][15 * lane + k] = c[k];
            wave_rows_store<15>(cam15 + 15 * wave0, 64, sRows[wave], lane);
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) cam15[15 * i + k] = c[k];
        }
    } else {
        const int64_t j = i - n_cam;
        typename V4<T>::type p = pts4[j];
        const T s = sin(dot3(p.x / d0, p.y / d1, p.z / d2, dx, dy, dz) * frequency * (T)kPi) * strength;
        p.x = p.x + nx * s; p.y = p.y + ny * s; p.z = p.z + nz * s;
        pts4[j] = p;
    }
}

// device indices are u32, the host ABI speaks usize (u64): widened on the device so that a download is one copy
__global__ void k_widen_u32(const uint32_t *__restrict__ a, int64_t n, uint64_t *__restrict__ b) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}

// f64 <-> f32 state conversion for the config-5 extension (element-wise, n scalars)
__global__ void k_f64_to_f32(const double *__restrict__ a, int64_t n, float *__restrict__ b) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = (float)a[i];
}
__global__ void k_f32_to_f64(const float *__restrict__ a, int64_t n, double *__restrict__ b) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = (double)a[i];
}

}  // namespace c2b

// =====================================================================================================
// Dense visibility sweep: every camera x every point, the O(C*P) loop of the mesh generator
// (visibility_graph, src/generate.rs:446-469) without its Embree occlusion stream.  Kept observations
// come out per camera in ascending point order, exactly like the reference's `for (i, point) in
// points.iter().enumerate()` push order.
//
// Mapping: a wave owns a tile of 256 consecutive points (4 per lane, in registers for the whole kernel)
// and walks a chunk of cameras whose centres come in through scalar loads (the camera index is
// wave-uniform).  The common case -- all four points farther than max_dist -- costs 8 f64 operations per
// pair and one wave-uniform branch; only survivors run project_world / project.  Pass 1 counts survivors
// per (camera, tile) into a zeroed table; a row scan turns counts into offsets; pass 2 repeats the
// predicate and writes (point index, uv) at offset + wave-prefix rank.
// =====================================================================================================
namespace c2b {

constexpr int kDensePPL = 4;                       // points per lane
constexpr int kDenseTile = 64 * kDensePPL;         // points per wave tile
constexpr int kDenseWPB = 4;                       // waves per workgroup
constexpr int kDenseCamTile = 64;                  // granularity of the camera chunks of blockIdx.y

C2B_DEV int wave_excl_scan(int v, int lane, int &total) {
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    total = __shfl(inc, 63, 64);
    return inc - v;
}

template <bool FILL>
__global__ __launch_bounds__(kDenseWPB * 64) void k_visibility_dense(
    const double *__restrict__ camblk, int64_t n_cam, int64_t cams_per_chunk, const double4 *__restrict__ pts4,
    int64_t n_pts, int64_t n_tiles, double max_dist, uint32_t *__restrict__ tile_counts,
    const uint64_t *__restrict__ row_ptr, uint32_t *__restrict__ pt_out, double2 *__restrict__ uv_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * kDenseWPB + wave;
    const int64_t p0 = tile * kDenseTile + (int64_t)lane * kDensePPL;      // this lane's first point
    if (tile >= n_tiles) return;                                           // wave-uniform; no workgroup barriers below

    // points past the end sit at 1e300: their squared distance is +inf and never passes the test
    double X[kDensePPL], Y[kDensePPL], Z[kDensePPL];
    bool pv[kDensePPL];
#pragma unroll
    for (int j = 0; j < kDensePPL; ++j) {
        pv[j] = p0 + j < n_pts;
        const double4 p = pv[j] ? pts4[p0 + j] : make_double4(1e300, 1e300, 1e300, 0);
        X[j] = p.x; Y[j] = p.y; Z[j] = p.z;
    }
    const double m2 = max_dist > 0.0 ? max_dist * max_dist : 0.0;          // a negative or NaN max_dist passes nothing (`magnitude() < max_dist`)
    const double m2_lo = m2 * (1.0 - 0x1.0p-50), m2_hi = m2 * (1.0 + 0x1.0p-50);

    const int64_t c_begin = (int64_t)blockIdx.y * cams_per_chunk;
    const int64_t c_end = c_begin + cams_per_chunk < n_cam ? c_begin + cams_per_chunk : n_cam;
    // The camera index is wave-uniform, so its centre arrives by scalar loads (SGPR operands of the vector
    // arithmetic): no LDS staging, no barriers.  The far-reject of all four points is one wave-uniform branch; a wave
    // that has a candidate falls into the per-point path below.  tile_counts is zeroed by the launcher, so only
    // non-empty (camera, tile) cells are written.
    auto camera_step = [&](const int64_t c) {
        const CamRec cam(camblk, c);
        const double cx = cam[kCenter], cy = cam[kCenter + 1], cz = cam[kCenter + 2];
        double d2[kDensePPL];
        bool cand = false;
#pragma unroll
        for (int j = 0; j < kDensePPL; ++j) {
            // (camera.center() - point).magnitude() < max_dist, src/generate.rs:450
            const double dx = cx - X[j], dy = cy - Y[j], dz = cz - Z[j];
            d2[j] = (dx * dx + dy * dy) + dz * dz;
            cand |= d2[j] < m2_hi;
        }
        if (!__any(cand)) return;
        unsigned keep_bits = 0;
        double u[kDensePPL], v[kDensePPL];
#pragma unroll
        for (int j = 0; j < kDensePPL; ++j) {
            u[j] = 0.0; v[j] = 0.0;
            if (pv[j] && d2[j] < m2_hi) {
                bool near = d2[j] < m2_lo;
                if (!near) near = sqrt(d2[j]) < max_dist;                // exact decision on the boundary band
                // behind the camera (q.z > 0, about half of the candidates): skip the divides of the projection
                if (near && dot3(cam[6], cam[7], cam[8], X[j], Y[j], Z[j]) + cam[11] <= 0.0) {
                    const Proj p = project_obs(cam, X[j], Y[j], Z[j]);
                    if (p.qz <= 0.0 && p.u >= -1.0 && p.u <= 1.0 && p.v >= -1.0 && p.v <= 1.0) {
                        keep_bits |= 1u << j;
                        u[j] = p.u; v[j] = p.v;
                    }
                }
            }
        }
        const int mine = __popc(keep_bits);
        if (!__any(mine != 0)) return;
        int total;
        const int before = wave_excl_scan(mine, lane, total);
        if (!FILL) {
            if (lane == 0) tile_counts[c * n_tiles + tile] = (uint32_t)total;
        } else {
            int64_t dst = (int64_t)row_ptr[c] + tile_counts[c * n_tiles + tile] + before;
#pragma unroll
            for (int j = 0; j < kDensePPL; ++j)
                if (keep_bits & (1u << j)) {
                    pt_out[dst] = (uint32_t)(p0 + j);
                    uv_out[dst] = make_double2(u[j], v[j]);
                    ++dst;
                }
        }
    };
    if (!FILL) {
        for (int64_t c = c_begin; c < c_end; ++c) camera_step(c);
    } else {
        // Pass 1 left the survivors of every (camera, tile) cell; only ~15 % of the cells hold any.  Sixty-four
        // cameras at a time, each lane reads one camera's offset and its successor (= the cell's count), and the
        // wave revisits the non-empty cells only.
        for (int64_t c0 = c_begin; c0 < c_end; c0 += 64) {
            const int64_t ci = c0 + lane;
            bool nonempty = false;
            if (ci < c_end) {
                const uint32_t o0 = tile_counts[ci * n_tiles + tile];
                const uint32_t o1 = tile + 1 < n_tiles ? tile_counts[ci * n_tiles + tile + 1]
                                                       : (uint32_t)(row_ptr[ci + 1] - row_ptr[ci]);
                nonempty = o1 != o0;
            }
            unsigned long long m = __ballot(nonempty);
            while (m) {
                const int k = __builtin_ctzll(m);
                m &= m - 1;
                camera_step(c0 + k);
            }
        }
    }
}

// per camera: exclusive scan of its row of tile counts (in place) + row total
__global__ __launch_bounds__(256) void k_dense_row_scan(uint32_t *__restrict__ tile_counts, int64_t n_tiles,
                                                       uint64_t *__restrict__ cam_total) {
    __shared__ int sWave[4];
    __shared__ int sCarry;
    uint32_t *row = tile_counts + (int64_t)blockIdx.x * n_tiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) sCarry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_tiles; base += 256) {
        const int64_t i = base + threadIdx.x;
        const int v = i < n_tiles ? (int)row[i] : 0;
        int total;
        const int ex = wave_excl_scan(v, lane, total);
        if (lane == 0) sWave[wave] = total;
        __syncthreads();
        int off = sCarry;
        for (int w = 0; w < wave; ++w) off += sWave[w];
        if (i < n_tiles) row[i] = (uint32_t)(off + ex);
        __syncthreads();
        if (threadIdx.x == 0) sCarry += (sWave[0] + sWave[1]) + (sWave[2] + sWave[3]);
        __syncthreads();
    }
    if (threadIdx.x == 0) cam_total[blockIdx.x] = (uint64_t)sCarry;
}

// row_ptr[0..n_cam] of a non-decreasing camera index list: row_ptr[c] = first position whose camera is >= c
__global__ __launch_bounds__(kBlock) void k_rows_from_sorted(const uint32_t *__restrict__ cam_idx, int64_t n, int64_t n_cam,
                                                            uint64_t *__restrict__ row_ptr) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i > n) return;
    const int64_t prev = i == 0 ? -1 : (int64_t)cam_idx[i - 1];
    const int64_t cur = i == n ? n_cam : (int64_t)cam_idx[i];
    for (int64_t c = prev + 1; c <= cur && c <= n_cam; ++c) row_ptr[c] = (uint64_t)i;
}

// ---- stable compaction of a CSR observation list by a keep mask (the occlusion filter's output), on the device ----
// one wave per camera: kept observations per row ...
__global__ __launch_bounds__(256) void k_keep_row_counts(const uint64_t *__restrict__ row_ptr, const uint8_t *__restrict__ keep,
                                                        int64_t n_cam, uint64_t *__restrict__ cam_total) {
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_cam) return;
    const uint64_t b = row_ptr[c], e = row_ptr[c + 1];
    int count = 0;
    for (uint64_t i = b + lane; i < e; i += 64) count += keep[i] ? 1 : 0;
    const double s = wave_sum((double)count);                 // exact: counts are far below 2^53
    if (lane == 0) cam_total[c] = (uint64_t)s;
}

// ... and, after the row scan, the kept (point index, uv) pairs moved to their new rows in their old order
__global__ __launch_bounds__(256) void k_keep_row_scatter(const uint64_t *__restrict__ row_old, const uint64_t *__restrict__ row_new,
                                                         const uint8_t *__restrict__ keep, const uint32_t *__restrict__ pt_in,
                                                         const double2 *__restrict__ uv_in, int64_t n_cam,
                                                         uint32_t *__restrict__ pt_out, double2 *__restrict__ uv_out) {
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_cam) return;
    const uint64_t b = row_old[c], e = row_old[c + 1];
    uint64_t dst = row_new[c];
    for (uint64_t base = b; base < e; base += 64) {           // wave-uniform trip count
        const uint64_t i = base + lane;
        const bool k = i < e && keep[i] != 0;
        const unsigned long long m = __ballot(k);
        if (k) {
            const uint64_t at = dst + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
            pt_out[at] = pt_in[i];
            uv_out[at] = uv_in[i];
        }
        dst += (uint64_t)__popcll(m);
    }
}

// row_ptr[0..n_cam] = exclusive scan of cam_total (one workgroup; n_cam is small next to n_cam * n_tiles)
__global__ __launch_bounds__(256) void k_dense_cam_scan(const uint64_t *__restrict__ cam_total, int64_t n_cam,
                                                       uint64_t *__restrict__ row_ptr) {
    __shared__ long long sWave[4];
    __shared__ long long sCarry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) sCarry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_cam; base += 256) {
        const int64_t i = base + threadIdx.x;
        const long long v = i < n_cam ? (long long)cam_total[i] : 0;
        long long inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const long long t = __shfl_up(inc, off, 64);
            if (lane >= off) inc += t;
        }
        if (lane == 63) sWave[wave] = inc;
        __syncthreads();
        long long off = sCarry;
        for (int w = 0; w < wave; ++w) off += sWave[w];
        if (i < n_cam) row_ptr[i] = (uint64_t)(off + inc - v);
        __syncthreads();
        if (threadIdx.x == 0) sCarry += (sWave[0] + sWave[1]) + (sWave[2] + sWave[3]);
        __syncthreads();
    }
    if (threadIdx.x == 0) row_ptr[n_cam] = (uint64_t)sCarry;
}

}  // namespace c2b

// ---- calibration kernels (c2b_calib_*): what this device does for pure streams, measured in the same process as ----
// ---- the bench so that a slow box can be told from a slow kernel.  They read / write only the buffers handed to ----
// ---- them and are not part of any compute path.                                                                 ----
namespace c2b {
// the residual + Jacobian kernel's 208 B/observation store geometry (1-KiB non-temporal stores of whole lines) with no
// loads, no LDS and no arithmetic: the floor its stores alone would take
template <bool NT, int WPB, bool XCD = true>
__global__ __launch_bounds__(WPB * 64) void k_store_pattern(int64_t n, int64_t n_btiles, double2 *__restrict__ r_out,
                                                           double *__restrict__ Jc, double *__restrict__ Jp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wt = (XCD ? xcd_tile(blockIdx.x, n_btiles) : (int64_t)blockIdx.x) * WPB + wave;
    const int64_t wave0 = wt * 64;
    if (wave0 + 64 > n) return;
    const double2 v = make_double2((double)lane, (double)wave);
    store16<NT>(reinterpret_cast<char *>(r_out + wave0 + lane), v);
    char *dc = reinterpret_cast<char *>(Jc) + wave0 * 144;
#pragma unroll
    for (int k = 0; k < 9; ++k) store16<NT>(dc + (k * 64 + lane) * 16, v);
    char *dp = reinterpret_cast<char *>(Jp) + wave0 * 48;
#pragma unroll
    for (int k = 0; k < 3; ++k) store16<NT>(dp + (k * 64 + lane) * 16, v);
}


// 16 bytes per lane streaming copy, one element per thread (the "float4 copy" MI355X_MICROARCH.md quotes 6.29 TB/s for)
__global__ __launch_bounds__(256) void k_copy16(const double2 *__restrict__ src, double2 *__restrict__ dst, int64_t n16) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}
}  // namespace c2b

// ---- the remaining Camera trait methods as batched kernels (src/baproblem.rs:141-143, 165-175) ----------------
namespace c2b {

// rows of cgmath's Matrix3::invert (cross products / determinant), as in cm_center
template <typename T>
C2B_DEV void cm_inverse_rows(const T *m, T a[3], T b[3], T c[3]) {
    const T m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6], m21 = m[7], m22 = m[8];
    const T det = m00 * (m11 * m22 - m21 * m12) - m10 * (m01 * m22 - m21 * m02) + m20 * (m01 * m12 - m11 * m02);
    a[0] = (m11 * m22 - m12 * m21) / det; a[1] = (m12 * m20 - m10 * m22) / det; a[2] = (m10 * m21 - m11 * m20) / det;
    b[0] = (m21 * m02 - m22 * m01) / det; b[1] = (m22 * m00 - m20 * m02) / det; b[2] = (m20 * m01 - m21 * m00) / det;
    c[0] = (m01 * m12 - m02 * m11) / det; c[1] = (m02 * m10 - m00 * m12) / det; c[2] = (m00 * m11 - m01 * m10) / det;
}

// Camera::project_world (to_world = false) / Camera::to_world (true) for pair i = (camera cam_idx[i], point i)
template <bool TO_WORLD>
__global__ void k_camera_point_map(const double *__restrict__ cam15, const uint32_t *__restrict__ cam_idx,
                                   const double *__restrict__ p3, int64_t n, double *__restrict__ out3) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double c[15];
    const double *src = cam15 + 15 * (int64_t)cam_idx[i];
#pragma unroll
    for (int k = 0; k < 15; ++k) c[k] = src[k];
    const double x = p3[3 * i], y = p3[3 * i + 1], z = p3[3 * i + 2];
    if (TO_WORLD) {
        // dir.invert().rotate_point(p - loc), src/baproblem.rs:173-175
        double a[3], b[3], cc[3];
        cm_inverse_rows(c, a, b, cc);
        const double dx = x - c[9], dy = y - c[10], dz = z - c[11];
        out3[3 * i] = dot3(a[0], a[1], a[2], dx, dy, dz);
        out3[3 * i + 1] = dot3(b[0], b[1], b[2], dx, dy, dz);
        out3[3 * i + 2] = dot3(cc[0], cc[1], cc[2], dx, dy, dz);
    } else {
        // dir.rotate_point(p) + loc, src/baproblem.rs:141-143
        double v[3];
        cm_mat_vec(c, x, y, z, v);
        out3[3 * i] = v[0] + c[9]; out3[3 * i + 1] = v[1] + c[10]; out3[3 * i + 2] = v[2] + c[11];
    }
}

// Camera::transform (src/baproblem.rs:165-171) for every camera with its own delta_dir [n][9] / delta_loc [n][3]
__global__ void k_cameras_transform(double *__restrict__ cam15, const double *__restrict__ dR9,
                                    const double *__restrict__ dloc3, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double c[15], d[9];
#pragma unroll
    for (int k = 0; k < 15; ++k) c[k] = cam15[15 * i + k];
#pragma unroll
    for (int k = 0; k < 9; ++k) d[k] = dR9[9 * i + k];
    transform_cam15(c, d, dloc3[3 * i], dloc3[3 * i + 1], dloc3[3 * i + 2]);
#pragma unroll
    for (int k = 0; k < 12; ++k) cam15[15 * i + k] = c[k];
}

}  // namespace c2b

// ---- occlusion test of the mesh generator: brute-force stand-in for Embree's occluded_stream_aos ---------------
// (src/generate.rs:455-476).  Per kept observation a f32 ray from the camera centre towards the point, tfar =
// |dir| - 1e-6; occluded iff any triangle is hit with 0 < t <= tfar.  Triangles are staged 256 at a time in LDS and
// read by broadcast (every lane tests the same triangle).  Embree's own intersector (BVH traversal order, its
// watertight / SIMD arithmetic) is not reproducible; rays grazing an edge may be classified differently.
namespace c2b {

constexpr int kOccTile = 256;     // triangles per LDS round (9 KB)

__global__ __launch_bounds__(kBlock) void k_occlusion(const double *__restrict__ camblk, const double4 *__restrict__ pts4,
                                                     const uint32_t *__restrict__ cam_idx,
                                                     const uint32_t *__restrict__ pt_idx, int64_t n,
                                                     const float *__restrict__ tri9, int64_t n_tri,
                                                     uint8_t *__restrict__ keep) {
    __shared__ float sTri[kOccTile * 9];
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = i < n;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 1, tfar = -1.0f;
    if (valid) {
        const double *c = camblk + cam_center_at((int64_t)cam_idx[i]);
        const double4 p = pts4[pt_idx[i]];
        const double ex = p.x - c[0], ey = p.y - c[1], ez = p.z - c[2];        // point - camera.center()
        const double mag = sqrt(dot3(ex, ey, ez, ex, ey, ez));
        const double inv = 1.0 / mag;                                          // dir.normalize() = dir * (1/|dir|)
        ox = (float)c[0]; oy = (float)c[1]; oz = (float)c[2];
        dx = (float)(ex * inv); dy = (float)(ey * inv); dz = (float)(ez * inv);
        tfar = (float)mag - 1e-6f;
    }
    bool occluded = false;
    for (int64_t base = 0; base < n_tri; base += kOccTile) {
        const int nt = n_tri - base < kOccTile ? (int)(n_tri - base) : kOccTile;
        __syncthreads();
        for (int e = threadIdx.x; e < nt * 9; e += kBlock) sTri[e] = tri9[base * 9 + e];
        __syncthreads();
        if (!valid || occluded) continue;
        for (int t = 0; t < nt; ++t) {
            const float *q = sTri + 9 * t;
            const float e1x = q[3] - q[0], e1y = q[4] - q[1], e1z = q[5] - q[2];
            const float e2x = q[6] - q[0], e2y = q[7] - q[1], e2z = q[8] - q[2];
            const float px = dy * e2z - dz * e2y, py = dz * e2x - dx * e2z, pz = dx * e2y - dy * e2x;
            const float det = e1x * px + e1y * py + e1z * pz;
            if (det == 0.0f) continue;
            const float inv = 1.0f / det;
            const float tx = ox - q[0], ty = oy - q[1], tz = oz - q[2];
            const float u = (tx * px + ty * py + tz * pz) * inv;
            if (u < 0.0f || u > 1.0f) continue;
            const float qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;
            const float w = (dx * qx + dy * qy + dz * qz) * inv;
            if (w < 0.0f || u + w > 1.0f) continue;
            const float th = (e2x * qx + e2y * qy + e2z * qz) * inv;
            if (th > 0.0f && th <= tfar) { occluded = true; break; }
        }
    }
    if (valid) keep[i] = occluded ? 0 : 1;
}

// The same rays through the host-built hierarchy (csrc/host_bvh.hpp): one 64-byte node (both children's boxes) per
// step, per-lane traversal stack, first hit ends the ray.  Leaves run the triangle test of k_occlusion on
// (v0, e1, e2) rows whose edges were subtracted in float32 on the host, i.e. the same values; boxes are inflated
// and compared with slack, so the hierarchy prunes without changing the answer.
constexpr int kBvhStack = 64;
constexpr int kBvhMinTriangles = 64;     // Level 1 switches from the all-triangles loop to the hierarchy here
constexpr int kBvhEmptyChild = INT32_MIN;

__device__ __forceinline__ bool bvh_box_hit(const float lo[3], const float hi[3], float ox, float oy, float oz, float ix,
                                            float iy, float iz, float tfar, float &tnear) {
    // fminf / fmaxf drop a NaN operand ((lo - o) * inf with lo == o): that slab then does not constrain
    const float ax = (lo[0] - ox) * ix, bx = (hi[0] - ox) * ix;
    const float ay = (lo[1] - oy) * iy, by = (hi[1] - oy) * iy;
    const float az = (lo[2] - oz) * iz, bz = (hi[2] - oz) * iz;
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    tnear = tn;
    return tn <= tf * 1.00001f + 1e-30f && tf >= 0.0f && tn <= tfar;
}

__global__ __launch_bounds__(kBlock) void k_occlusion_bvh(const double *__restrict__ camblk, const double4 *__restrict__ pts4,
                                                         const uint32_t *__restrict__ cam_idx,
                                                         const uint32_t *__restrict__ pt_idx, int64_t n,
                                                         const float4 *__restrict__ nodes, const float4 *__restrict__ tris,
                                                         uint8_t *__restrict__ keep, uint32_t *__restrict__ overflow) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double *c = camblk + cam_center_at((int64_t)cam_idx[i]);
    const double4 p = pts4[pt_idx[i]];
    const double ex = p.x - c[0], ey = p.y - c[1], ez = p.z - c[2];
    const double mag = sqrt(dot3(ex, ey, ez, ex, ey, ez));
    const double inv = 1.0 / mag;
    const float ox = (float)c[0], oy = (float)c[1], oz = (float)c[2];
    const float dx = (float)(ex * inv), dy = (float)(ey * inv), dz = (float)(ez * inv);
    const float tfar = (float)mag - 1e-6f;
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    int stack[kBvhStack];
    int sp = 0;
    int node = 0;
    bool occluded = false;
    while (true) {
        const float4 n0 = nodes[4 * (int64_t)node], n1 = nodes[4 * (int64_t)node + 1], n2 = nodes[4 * (int64_t)node + 2],
                     n3 = nodes[4 * (int64_t)node + 3];
        const float lo0[3] = {n0.x, n0.y, n0.z}, hi0[3] = {n0.w, n1.x, n1.y};
        const float lo1[3] = {n1.z, n1.w, n2.x}, hi1[3] = {n2.y, n2.z, n2.w};
        const int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
        float t0, t1;
        const bool h0 = c0 != kBvhEmptyChild && bvh_box_hit(lo0, hi0, ox, oy, oz, ix, iy, iz, tfar, t0);
        const bool h1 = c1 != kBvhEmptyChild && bvh_box_hit(lo1, hi1, ox, oy, oz, ix, iy, iz, tfar, t1);
        int next = -1;                                   // inner node to descend into
        int other = -1;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const bool h = side ? h1 : h0;
            const int ch = side ? c1 : c0;
            if (!h) continue;
            if (ch >= 0) {
                if (next < 0) next = ch; else other = ch;
                continue;
            }
            const unsigned code = (unsigned)~ch;
            const int first = (int)(code >> 3), cnt = (int)(code & 7u) + 1;
            for (int s = first; s < first + cnt && !occluded; ++s) {
                const float4 q0 = tris[3 * (int64_t)s], q1 = tris[3 * (int64_t)s + 1], q2 = tris[3 * (int64_t)s + 2];
                const float v0x = q0.x, v0y = q0.y, v0z = q0.z, e1x = q0.w, e1y = q1.x, e1z = q1.y, e2x = q1.z, e2y = q1.w,
                            e2z = q2.x;
                const float px = dy * e2z - dz * e2y, py = dz * e2x - dx * e2z, pz = dx * e2y - dy * e2x;
                const float det = e1x * px + e1y * py + e1z * pz;
                if (det == 0.0f) continue;
                const float idet = 1.0f / det;
                const float tx = ox - v0x, ty = oy - v0y, tz = oz - v0z;
                const float u = (tx * px + ty * py + tz * pz) * idet;
                if (u < 0.0f || u > 1.0f) continue;
                const float qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;
                const float w = (dx * qx + dy * qy + dz * qz) * idet;
                if (w < 0.0f || u + w > 1.0f) continue;
                const float th = (e2x * qx + e2y * qy + e2z * qz) * idet;
                if (th > 0.0f && th <= tfar) occluded = true;
            }
        }
        if (occluded) break;
        if (next >= 0) {
            if (other >= 0) {
                // both children are inner nodes: nearer first
                if (t1 < t0) { const int tmp = next; next = other; other = tmp; }
                // a hierarchy deeper than the stack (never one built by c2b_bvh_build, which refuses them) would lose
                // this subtree: say so instead of answering wrongly -- the caller must treat the whole mask as invalid
                if (sp < kBvhStack) stack[sp++] = other; else *overflow = 1u;
            }
            node = next;
        } else {
            if (!sp) break;
            node = stack[--sp];
        }
    }
    keep[i] = occluded ? 0 : 1;
}

}  // namespace c2b
