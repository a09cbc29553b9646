// cull_kernels.hpp -- BAProblem::cull (src/baproblem.rs:392-550) on the device.
//
// Same fixed-point loop as the host version (csrc/host_baproblem.hpp: lcc_pass, singleton_pass, compact), on the same
// index-only representation: the current graph in COO form (cam_idx / pt_idx, camera-major) plus, for every camera,
// point and observation, where it came from.  Payloads are gathered once at the end.
//
//   largest component : lock-free union-find (larger root hooks under the smaller, so a component's root is its
//                       smallest member -- the tie-break of the host version), sizes by atomic counters, arg-max of
//                       (size, smallest root) packed in one 64-bit atomicMax;
//   singletons        : per-camera / per-point observation counts by atomic counters;
//   renumbering       : exclusive scans of the keep flags (cameras, points, observations) -- order preserving, so the
//                       result equals the sequential version's element for element.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace c2b {

constexpr int kScanBlock = 256;                 // threads
constexpr int kScanPer = 4;                     // elements per thread
constexpr int kScanTile = kScanBlock * kScanPer;

// Every link points at a SMALLER index (k_uf_union hooks the larger root under the smaller), so a walk always terminates,
// a component's root is its smallest member, and a stale view of `parent` is harmless: a node that looked like a root
// once is still in the right component, and the hook itself is a compare-and-swap that only succeeds on a current root.
// FRESH = false walks with ordinary cached loads (L2 hits); FRESH = true with agent-scope atomic loads (global_load ...
// sc1: past the per-XCD L2, which other XCDs' hooks do not update) -- ~4x the latency per hop, used only to retry after
// a failed hook, where a stale cached line could otherwise be re-read for ever.
template <bool FRESH>
__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t x) {
    while (true) {
        const uint32_t p = FRESH ? __hip_atomic_load(parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : parent[x];
        if (p == x) return x;
        const uint32_t gp = FRESH ? __hip_atomic_load(parent + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : parent[p];
        if (gp != p) parent[x] = gp;            // path halving; a stale write still points at an ancestor
        x = p;
    }
}

__global__ void k_uf_init(uint32_t *parent, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) parent[i] = (uint32_t)i;
}

// union (camera c, point n_cam + p) of every observation
__global__ void k_uf_union(uint32_t *parent, const uint32_t *__restrict__ cam_idx, const uint32_t *__restrict__ pt_idx,
                           int64_t n_obs, uint32_t n_cam) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_obs) return;
    uint32_t a = cam_idx[e], b = n_cam + pt_idx[e];
    bool fresh = false;
    while (true) {
        a = fresh ? uf_find<true>(parent, a) : uf_find<false>(parent, a);
        b = fresh ? uf_find<true>(parent, b) : uf_find<false>(parent, b);
        if (a == b) break;
        if (a < b) { const uint32_t t = a; a = b; b = t; }          // a > b: hook the larger root under the smaller
        if (atomicCAS(&parent[a], a, b) == a) break;
        fresh = true;                                               // the view was stale: look again past the L2
    }
}

// sets[i] = root of i; size[root] += 1.  The counts are aggregated per wave first: a generator's graph is one giant
// component, so one atomicAdd per element was 2.6 M adds to ONE word -- 30 ms of the 100-ms cull at --blocks 128
// (profiles/r03r).  Lanes with the same root elect a leader (lowest lane) that adds their number once; a wave whose
// lanes all share a root issues a single add.
__global__ void k_uf_flatten(uint32_t *parent, int64_t n, uint32_t *__restrict__ sets, uint32_t *size) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    uint32_t r = 0;
    if (valid) {
        r = uf_find<false>(parent, (uint32_t)i);                     // the unions are complete (previous launch): cached loads
        sets[i] = r;
    }
    const int lane = threadIdx.x & 63;
    uint64_t todo = __builtin_amdgcn_ballot_w64(valid);
    while (todo != 0) {                                                  // wave-uniform
        const uint32_t lead = __builtin_amdgcn_readlane(r, (int)__builtin_ctzll(todo));
        const uint64_t same = __builtin_amdgcn_ballot_w64(valid && ((todo >> lane) & 1ull) != 0 && r == lead);
        if (lane == (int)__builtin_ctzll(same)) atomicAdd(&size[lead], (uint32_t)__builtin_popcountll(same));
        todo &= ~same;
    }
}

// best = max over roots of (size << 32 | ~root): the largest component, the smallest root among equals
__global__ void k_uf_largest(const uint32_t *__restrict__ sets, const uint32_t *__restrict__ size, int64_t n,
                             unsigned long long *best) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || sets[i] != (uint32_t)i) return;                   // roots only
    atomicMax(best, ((unsigned long long)size[i] << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i));
}

// keep flags of the largest-component pass.  faithful: the reference's observation filter (src/baproblem.rs:523)
// looks up element `point index` of the camera-first array.
__global__ void k_lcc_flags(const uint32_t *__restrict__ sets, const unsigned long long *__restrict__ best, uint32_t n_cam,
                            uint32_t n_pts, const uint32_t *__restrict__ cam_idx, const uint32_t *__restrict__ pt_idx,
                            int64_t n_obs, int faithful, uint32_t *__restrict__ keep_cam, uint32_t *__restrict__ keep_pt,
                            uint32_t *__restrict__ keep_obs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lcc = 0xFFFFFFFFu - (uint32_t)(*best & 0xFFFFFFFFull);
    if (i < n_cam) keep_cam[i] = sets[i] == lcc;
    if (i < n_pts) keep_pt[i] = sets[n_cam + i] == lcc;
    if (i < n_obs) {
        const uint32_t c = cam_idx[i], p = pt_idx[i];
        const bool pt_ok = sets[n_cam + p] == lcc;
        const bool filt = faithful ? sets[p] == lcc : true;
        keep_obs[i] = (sets[c] == lcc && pt_ok && filt) ? 1u : 0u;
    }
}

// observation counts per camera and per point
__global__ void k_degree(const uint32_t *__restrict__ cam_idx, const uint32_t *__restrict__ pt_idx, int64_t n_obs,
                         uint32_t *deg_cam, uint32_t *cnt_pt) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = e < n_obs;
    uint32_t c = 0;
    if (valid) {
        c = cam_idx[e];
        atomicAdd(&cnt_pt[pt_idx[e]], 1u);
    }
    // the list is camera-major: a wave's 64 observations belong to 3-4 cameras, so the camera degrees are added once
    // per camera and wave (see k_uf_flatten) instead of once per observation
    const int lane = threadIdx.x & 63;
    uint64_t todo = __builtin_amdgcn_ballot_w64(valid);
    while (todo != 0) {                                                  // wave-uniform
        const uint32_t lead = __builtin_amdgcn_readlane(c, (int)__builtin_ctzll(todo));
        const uint64_t same = __builtin_amdgcn_ballot_w64(valid && ((todo >> lane) & 1ull) != 0 && c == lead);
        if (lane == (int)__builtin_ctzll(same)) atomicAdd(&deg_cam[lead], (uint32_t)__builtin_popcountll(same));
        todo &= ~same;
    }
}

// remove_singletons (src/baproblem.rs:426-453): cameras need > 3 observations, points > 1
__global__ void k_singleton_flags(const uint32_t *__restrict__ deg_cam, const uint32_t *__restrict__ cnt_pt, uint32_t n_cam,
                                  uint32_t n_pts, const uint32_t *__restrict__ cam_idx, const uint32_t *__restrict__ pt_idx,
                                  int64_t n_obs, uint32_t *__restrict__ keep_cam, uint32_t *__restrict__ keep_pt,
                                  uint32_t *__restrict__ keep_obs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_cam) keep_cam[i] = deg_cam[i] > 3u;
    if (i < n_pts) keep_pt[i] = cnt_pt[i] > 1u;
    if (i < n_obs) keep_obs[i] = (deg_cam[cam_idx[i]] > 3u && cnt_pt[pt_idx[i]] > 1u) ? 1u : 0u;
}

// ---- exclusive scan of 0/1 flags (uint32), three kernels: tile sums, scan of the tile sums, final offsets ----------
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *sWave, uint32_t &block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) sWave[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += sWave[w];
    block_total = (sWave[0] + sWave[1]) + (sWave[2] + sWave[3]);
    return base + inc - v;
}

__global__ __launch_bounds__(kScanBlock) void k_scan_tiles(const uint32_t *__restrict__ flags, int64_t n, uint32_t *__restrict__ out,
                                                          uint32_t *__restrict__ tile_sum) {
    __shared__ uint32_t sWave[kScanBlock / 64];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanPer;
    uint32_t f[kScanPer], s = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) { f[k] = base + k < n ? flags[base + k] : 0u; s += f[k]; }
    uint32_t total;
    uint32_t ex = block_excl_scan(s, sWave, total);
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) {
        if (base + k < n) out[base + k] = ex;
        ex += f[k];
    }
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}

// one workgroup: tile_sum -> exclusive offsets in place; total[0] = grand total
__global__ __launch_bounds__(kScanBlock) void k_scan_tile_sums(uint32_t *__restrict__ tile_sum, int64_t n_tiles, uint32_t *__restrict__ total) {
    __shared__ uint32_t sWave[kScanBlock / 64];
    __shared__ uint32_t sCarry;
    if (threadIdx.x == 0) sCarry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_tiles; base += kScanBlock) {
        const int64_t i = base + threadIdx.x;
        const uint32_t v = i < n_tiles ? tile_sum[i] : 0u;
        uint32_t chunk;
        const uint32_t ex = block_excl_scan(v, sWave, chunk);
        const uint32_t carry = sCarry;
        if (i < n_tiles) tile_sum[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) sCarry = carry + chunk;
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = sCarry;
}

__global__ __launch_bounds__(kScanBlock) void k_scan_add(uint32_t *__restrict__ out, int64_t n, const uint32_t *__restrict__ tile_off) {
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanPer;
    const uint32_t off = tile_off[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanPer; ++k)
        if (base + k < n) out[base + k] += off;
}

// ---- renumbering: survivors move to their scanned positions, remembering where they came from -----------------------
__global__ void k_cull_move_nodes(const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos, int64_t n,
                                  const uint32_t *__restrict__ orig_in, uint32_t *__restrict__ orig_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && keep[i]) orig_out[pos[i]] = orig_in[i];
}

__global__ void k_cull_move_edges(const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos, int64_t n_obs,
                                  const uint32_t *__restrict__ cam_in, const uint32_t *__restrict__ pt_in,
                                  const uint32_t *__restrict__ orig_in, const uint32_t *__restrict__ cam_pos,
                                  const uint32_t *__restrict__ pt_pos, uint32_t *__restrict__ cam_out,
                                  uint32_t *__restrict__ pt_out, uint32_t *__restrict__ orig_out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_obs || !keep[e]) return;
    const uint32_t d = pos[e];
    cam_out[d] = cam_pos[cam_in[e]];
    pt_out[d] = pt_pos[pt_in[e]];
    orig_out[d] = orig_in[e];
}

// ---- final gathers ---------------------------------------------------------------------------------------------------
__global__ void k_gather_rows(const double *__restrict__ in, const uint32_t *__restrict__ orig, int64_t n, int width,
                              double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * width) return;
    const int64_t r = i / width, k = i % width;
    out[i] = in[(int64_t)orig[r] * width + k];
}

}  // namespace c2b
