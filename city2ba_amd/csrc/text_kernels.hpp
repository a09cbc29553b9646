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
