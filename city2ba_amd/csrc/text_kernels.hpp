// The text form of a problem (.bal: src/baproblem.rs:709-733 writes it, :580-629 reads it) assembled / taken apart on the
// device: decimal.hpp's shortest round-trip digits per value, then the file image byte by byte.
//
// Writing.  A line's position in the file is the sum of the lengths of every line before it, so the image is made in
// two passes over tiles of 256 units (an observation line, or one value of a camera / point line with the space or
// newline behind it): pass 1 leaves each tile's byte count, one small kernel turns the counts into 64-bit tile bases,
// pass 2 recomputes its tile's lengths, scans them inside the workgroup and writes the characters.  Nothing per line
// is stored between the passes; the digits are simply found twice (a few hundred integer operations per value).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "decimal.hpp"

constexpr int kTextTile = 256;

__device__ c2b_dec::Tables g_dec_tables;              // filled once per device (capi_problem.hpp: device_dec_tables)

// exclusive scan of one value per thread over a workgroup of kTextTile threads; *total = the tile's sum (every thread)
__device__ __forceinline__ uint32_t text_tile_scan(uint32_t v, uint32_t *sh, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kTextTile / 64; ++w) {
        const uint32_t s = sh[w];
        if (w < wave) base += s;
        all += s;
    }
    *total = all;
    return base + inc - v;
}

// observation lines: "camera point u v\n"
template <bool EMIT>
__global__ __launch_bounds__(kTextTile) void k_text_obs(const uint32_t *__restrict__ cam_idx, const uint32_t *__restrict__ pt_idx,
                                                        const double2 *__restrict__ uv, int64_t n,
                                                        const c2b_dec::Tables *__restrict__ T, uint32_t *__restrict__ tile_len,
                                                        const uint64_t *__restrict__ tile_base, char *__restrict__ img) {
    __shared__ uint32_t sh[kTextTile / 64];
    const int64_t i = (int64_t)blockIdx.x * kTextTile + threadIdx.x;
    uint32_t len = 0, lc = 0, lp = 0, c = 0, p = 0;
    c2b_dec::Text tu, tv;
    tu.len = 0; tv.len = 0;
    if (i < n) {
        c = cam_idx[i]; p = pt_idx[i];
        const double2 o = uv[i];
        tu = c2b_dec::describe(o.x, T);
        tv = c2b_dec::describe(o.y, T);
        lc = c2b_dec::uint_len(c); lp = c2b_dec::uint_len(p);
        len = lc + 1 + lp + 1 + tu.len + 1 + tv.len + 1;
    }
    uint32_t total;
    const uint32_t at = text_tile_scan(len, sh, &total);
    if (!EMIT) {
        if (threadIdx.x == 0) tile_len[blockIdx.x] = total;
        return;
    }
    if (i >= n) return;
    char *q = img + tile_base[blockIdx.x] + at;
    c2b_dec::uint_emit(c, lc, q); q += lc; *q++ = ' ';
    c2b_dec::uint_emit(p, lp, q); q += lp; *q++ = ' ';
    c2b_dec::emit(tu, q); q += tu.len; *q++ = ' ';
    c2b_dec::emit(tv, q); q += tv.len; *q = '\n';
}

// rows of `width` values (`stride` doubles apart) joined by spaces, one row per line: a unit = one value + what follows it
template <bool EMIT>
__global__ __launch_bounds__(kTextTile) void k_text_vals(const double *__restrict__ src, int64_t n_rows, int width, int stride,
                                                         const c2b_dec::Tables *__restrict__ T, uint32_t *__restrict__ tile_len,
                                                         const uint64_t *__restrict__ tile_base, char *__restrict__ img) {
    __shared__ uint32_t sh[kTextTile / 64];
    const int64_t j = (int64_t)blockIdx.x * kTextTile + threadIdx.x, n = n_rows * width;
    c2b_dec::Text t;
    t.len = 0;
    uint32_t len = 0;
    int col = 0;
    if (j < n) {
        const int64_t row = j / width;
        col = (int)(j - row * width);
        t = c2b_dec::describe(src[row * stride + col], T);
        len = t.len + 1;
    }
    uint32_t total;
    const uint32_t at = text_tile_scan(len, sh, &total);
    if (!EMIT) {
        if (threadIdx.x == 0) tile_len[blockIdx.x] = total;
        return;
    }
    if (j >= n) return;
    char *q = img + tile_base[blockIdx.x] + at;
    c2b_dec::emit(t, q);
    q[t.len] = col == width - 1 ? '\n' : ' ';
}

// tile byte counts -> 64-bit tile bases (exclusive, starting at `first`); total[0] = first + every tile.  One workgroup:
// thread t owns a contiguous run of tiles.
__global__ __launch_bounds__(1024) void k_text_tile_bases(const uint32_t *__restrict__ tile_len, int64_t n_tiles, uint64_t first,
                                                          uint64_t *__restrict__ tile_base, uint64_t *__restrict__ total) {
    __shared__ uint64_t part[1024];
    const int64_t per = (n_tiles + 1023) / 1024, a = (int64_t)threadIdx.x * per, b = a + per < n_tiles ? a + per : n_tiles;
    uint64_t s = 0;
    for (int64_t k = a; k < b; ++k) s += tile_len[k];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = first;
        for (int t = 0; t < 1024; ++t) { const uint64_t v = part[t]; part[t] = run; run += v; }
        total[0] = run;
    }
    __syncthreads();
    uint64_t run = part[threadIdx.x];
    for (int64_t k = a; k < b; ++k) { tile_base[k] = run; run += tile_len[k]; }
}

// ---- reading ---------------------------------------------------------------------------------------------------------
// from_file_text (src/baproblem.rs:580-629) is a positional grammar over whitespace-separated tokens: 3 counts, 4 per
// observation, 9 per camera, 3 per point.  A token's meaning is its index, and its index is the number of token starts
// before it: pass 1 counts the starts in every tile of kParseTile bytes, k_text_tile_bases turns the counts into tile
// bases, pass 2 finds its starts again, ranks them inside the workgroup, parses each token where it stands
// (decimal.hpp: correctly rounded, or `unsure`) and stores the value where that index belongs.  flags[0] counts tokens
// the device declines (another spelling, more than 19 digits, an undecided rounding) -- the caller then gives the file
// to the host parser; flags[1] / flags[2]: a camera / point index out of range (BAProblem::new's asserts, :344-345).
constexpr int kParseBytes = 16;                              // per thread
constexpr int kParseTile = kTextTile * kParseBytes;         // bytes per workgroup

__device__ __forceinline__ bool text_ws(uint32_t c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r'; }

// bit b of the result: byte b of this thread's 16 starts a token
__device__ __forceinline__ uint32_t text_starts(const char *__restrict__ raw, int64_t at, int64_t n) {
    if (at >= n) return 0u;
    const uint4 v = *reinterpret_cast<const uint4 *>(raw + at);          // the buffer is padded to a multiple of 16
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    bool prev_ws = at == 0 ? true : text_ws((uint32_t)(uint8_t)raw[at - 1]);
    uint32_t m = 0;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const uint32_t c = (w[b >> 2] >> (8 * (b & 3))) & 0xffu;
        const bool ws = text_ws(c) || at + b >= n;
        if (!ws && prev_ws) m |= 1u << b;
        prev_ws = ws;
    }
    return m;
}

__global__ __launch_bounds__(kTextTile) void k_text_count_tokens(const char *__restrict__ raw, int64_t n, uint32_t *__restrict__ tile_cnt) {
    __shared__ uint32_t sh[kTextTile / 64];
    const int64_t at = ((int64_t)blockIdx.x * kTextTile + threadIdx.x) * kParseBytes;
    uint32_t total;
    (void)text_tile_scan((uint32_t)__builtin_popcount(text_starts(raw, at, n)), sh, &total);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}

__global__ __launch_bounds__(kTextTile) void k_text_parse(const char *__restrict__ raw, int64_t n, const uint64_t *__restrict__ tile_base,
                                                          const c2b_dec::ParseTables *__restrict__ T, uint64_t n_cam, uint64_t n_pts, uint64_t n_obs,
                                                          uint32_t *__restrict__ cam_idx, uint32_t *__restrict__ pt_idx,
                                                          double *__restrict__ uv, double *__restrict__ bal9, double *__restrict__ pts4,
                                                          uint32_t *__restrict__ flags) {
    __shared__ uint32_t sh[kTextTile / 64];
    const int64_t at = ((int64_t)blockIdx.x * kTextTile + threadIdx.x) * kParseBytes;
    uint32_t starts = text_starts(raw, at, n);
    uint32_t total;
    uint64_t k = tile_base[blockIdx.x] + text_tile_scan((uint32_t)__builtin_popcount(starts), sh, &total);
    const uint64_t k_obs = 3, k_cam = k_obs + 4 * n_obs, k_pts = k_cam + 9 * n_cam, k_end = k_pts + 3 * n_pts;
    for (; starts != 0; starts &= starts - 1, ++k) {
        if (k < k_obs || k >= k_end) continue;                // the header is the host's; nom leaves what follows the last point unread
        const int64_t s = at + __builtin_ctz(starts);
        int32_t len = 1;
        while (s + len < n && len <= 400 && !text_ws((uint32_t)(uint8_t)raw[s + len])) ++len;
        int status = len > 400 ? (int)c2b_dec::PARSE_IRREGULAR : (int)c2b_dec::PARSE_OK;
        if (k < k_cam) {
            const uint64_t j = k - k_obs, i = j >> 2;
            const int f = (int)(j & 3);
            if (f < 2) {
                const uint64_t v = status ? 0 : c2b_dec::parse_u64(raw + s, len, &status);
                if (!status) {
                    if (v >= (f == 0 ? n_cam : n_pts)) atomicOr(flags + 1 + f, 1u);
                    else (f == 0 ? cam_idx : pt_idx)[i] = (uint32_t)v;
                }
            } else {
                const double v = status ? 0.0 : c2b_dec::parse_f64(raw + s, len, T, &status);
                if (!status) uv[2 * i + (f - 2)] = v;
            }
        } else {
            const double v = status ? 0.0 : c2b_dec::parse_f64(raw + s, len, T, &status);
            if (!status) {
                if (k < k_pts) bal9[k - k_cam] = v;
                else { const uint64_t j = k - k_pts, r = j / 3; pts4[4 * r + (j - 3 * r)] = v; }
            }
        }
        if (status) atomicAdd(flags, 1u);
    }
}

// flags[3] |= 1 if the camera indices are not in non-decreasing order (the per-camera push of BAProblem::new, :347-353,
// is then a stable sort, which the host path does)
__global__ __launch_bounds__(256) void k_text_check_sorted(const uint32_t *__restrict__ cam_idx, int64_t n, uint32_t *__restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i + 1 < n && cam_idx[i] > cam_idx[i + 1]) atomicOr(flags + 3, 1u);
}

__device__ c2b_dec::ParseTables g_parse_tables;
