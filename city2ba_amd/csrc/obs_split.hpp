// obs_split.hpp -- EXPERIMENT of round 5, tuning library only (-DC2B_TUNE; included by capi.hip): the projection pass with
// the waves of a workgroup SPECIALISED -- one loader wave, seven compute waves -- instead of every wave doing everything.
//
// Why.  The light per-observation passes (kernels.hpp: k_observations) keep the vector ALUs 41-60 % busy and HBM at
// 57-67 % (rocprofv3 counters, profiles/r05h_light_sq.json): neither resource is the limit, the dependent chain
// index -> point gather / camera rows is, and every remedy inside the one-shot structure has been measured flat or
// negative (tiles per wave 1 ... 8, workgroup size, cache policies, prefetch, stagger; docs/log_r01_r03.md, DESIGN.md
// section 3.1).  Round 2's persistent, software-pipelined form lost for a structural reason: on gfx950 a wave's loads and
// stores retire through ONE in-order counter (vmcnt), so a wave that loops over tiles waits for its own earlier stores
// whenever it waits for a later load.  Specialisation removes exactly that coupling: the loader wave only LOADS (its
// vmcnt waits never meet a store), the compute waves only STORE to global memory (they never wait on vmcnt at all), and the
// two meet in LDS:
//
//   loader wave    for every batch of G tiles: 4-byte point indices (one coalesced load per tile), then -- once they have
//                  arrived -- the points themselves and the tile's camera rows, gathered STRAIGHT INTO LDS
//                  (global_load_lds_dwordx4: no registers hold the data), into a ring of 2 G slots; the index loads of
//                  batch b + 1 are in flight together with the gathers of batch b, so a batch costs one round trip;
//   compute waves  tile j of the workgroup belongs to compute wave j mod 7: wait for the slot's "ready" word, read the
//                  point (2 x ds_read_b128) and the camera (broadcast reads) from LDS, set the slot's "done" word, do the
//                  projection's arithmetic (camera_math.hpp: project_obs -- the same instructions as k_observations, so
//                  the same bits), store.
//
// MEASURED (r05, tools/tune_obs.py --only "project rows", --blocks 128, profiles/r05o_ab_project_split_negative.txt): correct -- the
// outputs are bit-identical to k_observations' -- and TWICE AS SLOW: 147 us back to back / 171 us from swept caches against
// 78 / 106 us (G 6, K 6; G 4: 156 / 192, G 8: 167 / 195, K 12: 153 / 176).  Why: the one-shot kernel keeps 32 waves x 3
// tiles = 96 tiles in flight per CU, in registers, and every wave issues its own loads; here a workgroup's loads are one
// wave's instruction stream (~50 instructions per tile at one wave's issue rate is a quarter of a microsecond per tile by
// itself) and the tiles in flight are what its ring holds: 2 G = 12 per workgroup, 48 per CU at 37 KB of LDS each -- a
// batch of G tiles costs the loader a full round trip plus its own issue time, so a CU is fed ~170 tiles per ~17 us where
// it consumes them in ~8.  More loaders or a deeper ring run into the 160 KB of LDS (a tile in flight costs 3 KB there,
// 18 registers per lane in the one-shot form).  Kept, like obs_pipeline.hpp, as a built and measured negative.
//
// Flags are 32-bit words in LDS (generation numbers, never reset); every spin is bounded (a bug ends in a wrong answer
// and a raised error word, never in a hung device).  Lanes whose camera is not among the kSplitCams staged ones (more
// cameras in a tile than staged, or a tile with an empty list inside) read the record from global memory under a mask.
#pragma once
#include "kernels.hpp"

namespace c2b {

typedef __attribute__((address_space(3))) void *lds_vptr;
typedef const __attribute__((address_space(1))) void *glb_vptr;

constexpr int kSplitCams = 6;                                  // camera rows (128 B each) staged per tile
constexpr int kSplitRows = 2048;                               // slot layout, bytes: X lo [64][16] | X hi [64][16] |
constexpr int kSplitCi = kSplitRows + kSplitCams * 128;        //   camera rows | ci [64] u32 | c_first, n_staged
constexpr int kSplitHdr = kSplitCi + 256;
constexpr int kSplitSlot = kSplitHdr + 16;
constexpr unsigned kSplitSpinLimit = 1u << 20;                 // x 64 cycles: ~30 ms

C2B_DEV uint32_t flag_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
C2B_DEV void flag_store(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// waits until *p >= want; false (and *err raised) when the bound is hit
C2B_DEV bool flag_wait(const uint32_t *p, uint32_t want, unsigned *err) {
    unsigned spins = 0;
    while (flag_load(p) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kSplitSpinLimit) {
            if ((threadIdx.x & 63) == 0) atomicAdd(err, 1u);
            return false;
        }
    }
    return true;
}

// every lane's camera of tile `ti` from its record (held in registers: wave-uniform values), kernels.hpp: rows_cameras
C2B_DEV uint32_t split_lane_camera(uint32_t rx, uint32_t ry, uint32_t rz, const uint64_t *__restrict__ row_ptr, int n_cam, int ti,
                                   int n, int lane) {
    if (rz & 0x80000000u) {                                              // wave-uniform: an empty list inside this tile
        int o = ti * 64 + lane;
        o = o < n ? o : n - 1;
        return csr_search(row_ptr, n_cam, (uint64_t)o);
    }
    const uint32_t lo = (rx >> 1) | (ry << 31), hi = ry >> 1;
    return __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, rz));
}

// G: tiles per loader batch (the ring holds 2 G); K: tiles per compute wave and workgroup (a workgroup takes 7 K tiles)
template <int G, int K, bool NTS>
__global__ __launch_bounds__(512) void k_project_split(
    const double *__restrict__ camblk, const double4 *__restrict__ pts4, const uint4 *__restrict__ tiles,
    const uint32_t *__restrict__ pt_idx, int n, int n_wg, double2 *__restrict__ uv_out,
    const uint64_t *__restrict__ row_ptr, int n_cam, unsigned *__restrict__ err) {
    constexpr int S = 2 * G, TPW = 7 * K;
    static_assert(G <= 32, "one lane per tile of a batch loads its record");
    __shared__ __attribute__((aligned(16))) char ring[S * kSplitSlot];
    __shared__ uint32_t ready[S], done[S];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < S) { ready[threadIdx.x] = 0u; done[threadIdx.x] = 0u; }
    __syncthreads();
    const int n_tiles = (n + 63) >> 6;
    const int tile0 = xcd_tile32(blockIdx.x, n_wg) * TPW;
    const int T = n_tiles - tile0 < TPW ? n_tiles - tile0 : TPW;          // this workgroup's tiles (<= 0: none)
    if (T <= 0) return;

    if (wave == 0) {
        // ---------------------------------------------------------------- the loader ------------------------------
        // It only loads: its waits (vmcnt) never meet a store.  Per batch: [wait: indices + records of batch b have
        // arrived, the gathers of batch b - 1 have landed in LDS] publish batch b - 1; wait until batch b - 2's slots
        // were read; issue batch b's gathers (points and camera rows, straight into LDS) and its camera indices
        // (ds_write); issue the index + record loads of batch b + 1.  One round trip per batch.
        const int n_batches = (T + G - 1) / G;
        uint32_t pi_nxt[G], pi_cur[G];
        uint4 rec_nxt, rec_cur;
        auto issue_idx = [&](int b) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int j = b * G + g;
                int o = (tile0 + (j < T ? j : T - 1)) * 64 + lane;
                o = o < n ? o : n - 1;
                pi_nxt[g] = pt_idx[o];
            }
            const int jr = b * G + (lane < G ? lane : G - 1);
            rec_nxt = tiles[tile0 + (jr < T ? jr : T - 1)];               // lane g holds tile g's record
        };
        issue_idx(0);
        for (int b = 0; b < n_batches; ++b) {
            __builtin_amdgcn_s_waitcnt(0);                                 // indices / records of batch b here, gathers of batch b - 1 landed
#pragma unroll
            for (int g = 0; g < G; ++g) pi_cur[g] = pi_nxt[g];
            rec_cur = rec_nxt;
            if (b > 0) {                                                   // publish batch b - 1
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane < G && (b - 1) * G + lane < T) flag_store(&ready[((b - 1) & 1) * G + lane], (uint32_t)((b - 1) / 2 + 1));
            }
            if (b >= 2) {                                                  // the slots of batch b were batch b - 2's: consumed?
                bool ok = true;
#pragma unroll
                for (int g = 0; g < G; ++g)
                    if ((b - 2) * G + g < T) ok = flag_wait(&done[(b & 1) * G + g], (uint32_t)((b - 2) / 2 + 1), err) && ok;
                if (!ok) return;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int j = b * G + g;
                if (j >= T) break;                                         // wave-uniform
                char *slot = ring + ((b & 1) * G + g) * kSplitSlot;
                const char *src = reinterpret_cast<const char *>(pts4 + pi_cur[g]);
                __builtin_amdgcn_global_load_lds((glb_vptr)src, (lds_vptr)slot, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_vptr)(src + 16), (lds_vptr)(slot + 1024), 16, 0, 0);
                const uint32_t rx = __builtin_amdgcn_readlane(rec_cur.x, g), ry = __builtin_amdgcn_readlane(rec_cur.y, g),
                               rz = __builtin_amdgcn_readlane(rec_cur.z, g);
                const int ti = tile0 + j;
                const uint32_t ci = split_lane_camera(rx, ry, rz, row_ptr, n_cam, ti, n, lane);
                const int n_wave = n - ti * 64 < 64 ? n - ti * 64 : 64;
                const uint32_t c_first = __builtin_amdgcn_readfirstlane(ci);
                const uint32_t c_last = __builtin_amdgcn_readlane(ci, n_wave - 1);
                uint32_t n_staged = c_last >= c_first ? c_last - c_first + 1 : 1;
                if (n_staged > (uint32_t)kSplitCams) n_staged = kSplitCams;
                if (lane < (int)n_staged * 8) {
                    const char *row = reinterpret_cast<const char *>(camblk + cam_light_at((int64_t)(c_first + (lane >> 3)))) + (lane & 7) * 16;
                    __builtin_amdgcn_global_load_lds((glb_vptr)row, (lds_vptr)(slot + kSplitRows), 16, 0, 0);
                }
                reinterpret_cast<uint32_t *>(slot + kSplitCi)[lane] = ci;
                if (lane == 0) *reinterpret_cast<uint2 *>(slot + kSplitHdr) = make_uint2(c_first, n_staged);
            }
            if (b + 1 < n_batches) issue_idx(b + 1);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        const int lb = n_batches - 1;
        if (lane < G && lb * G + lane < T) flag_store(&ready[(lb & 1) * G + lane], (uint32_t)(lb / 2 + 1));
        return;
    }

    // -------------------------------------------------------------------- the compute waves ----------------------
    // They only store to global memory: no load of theirs ever waits behind one of their stores.
    for (int j = wave - 1; j < T; j += 7) {
        const int ti = tile0 + j, b = j / G, g = j % G;
        const int s = (b & 1) * G + g;
        const uint32_t gen = (uint32_t)(b / 2 + 1);
        if (!flag_wait(&ready[s], gen, err)) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const char *slot = ring + s * kSplitSlot;
        const double2 xy = *reinterpret_cast<const double2 *>(slot + lane * 16);
        const double2 zw = *reinterpret_cast<const double2 *>(slot + 1024 + lane * 16);
        const uint32_t ci = reinterpret_cast<const uint32_t *>(slot + kSplitCi)[lane];
        const uint2 hdr = *reinterpret_cast<const uint2 *>(slot + kSplitHdr);
        const uint32_t c_first = __builtin_amdgcn_readfirstlane(hdr.x), n_staged = __builtin_amdgcn_readfirstlane(hdr.y);
        const int o = ti * 64 + lane;
        const bool valid = o < n;
        const uint32_t local = ci - c_first;
        const bool in = local < n_staged;
        lds_cptr cam = (lds_cptr)reinterpret_cast<const double *>(slot + kSplitRows) + (in ? local : 0u) * 16;
        Proj p = project_obs(cam, xy.x, xy.y, zw.x);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");             // this wave's LDS reads are done
        if (lane == 0) flag_store(&done[s], gen);
        if (__builtin_amdgcn_ballot_w64(valid && !in) != 0) {              // rare: more cameras than staged / an empty list inside
            const Proj q = project_obs(CamRec(camblk, (int64_t)ci), xy.x, xy.y, zw.x);
            if (!in) p = q;
        }
        if (valid) store16<NTS>(reinterpret_cast<char *>(uv_out + o), make_double2(p.u, p.v));
    }
}

}  // namespace c2b
