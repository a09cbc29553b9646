// The text form of a problem (.bal: src/baproblem.rs:709-733 writes it, :580-629 reads it) assembled / taken apart on the
// device: decimal.hpp's shortest round-trip digits per value, then the file image byte by byte.
//
// Writing.  A line's position in the file is the sum of the lengths of every line before it, so the image is made in
// two passes over tiles of 256 units (an observation line, or one value of a camera / point line with the space or
// newline behind it): pass 1 leaves each tile's byte count, one small kernel turns the counts into 64-bit tile bases,
// pass 2 recomputes its tile's lengths, scans them inside the workgroup and writes the characters.  Nothing per line
// is stored between the passes; the digits are simply found twice (a few hundred integer operations per value).
// The characters of a tile are one contiguous piece of the file: they are written into LDS first -- at the offset
// that gives every byte its global address modulo 16 -- and leave as whole 16-byte stores (a tile whose text does not
// fit the LDS window, i.e. one full of 300-character values, writes its bytes to global memory directly).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "decimal.hpp"

constexpr int kTextTile = 256;
constexpr int kEmitCap = 24 * 1024;                   // LDS window of one tile's characters

typedef __attribute__((address_space(3))) char *lds_char;
typedef const __attribute__((address_space(3))) char *lds_cchar;
typedef unsigned int text_u4 __attribute__((ext_vector_type(4)));      // a plain vector (uint4 is a class: no address spaces)
typedef __attribute__((address_space(3))) text_u4 *lds_u4;

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

// out[shift + k] -> dst[k] for k < total, where shift = (address of dst) mod 16: whole aligned 16-byte stores, the ragged
// ends byte by byte
__device__ __forceinline__ void text_copy_out(char *out_generic, uint32_t shift, uint32_t total, char *__restrict__ dst) {
    const lds_cchar out = (lds_cchar)out_generic;
    const uint32_t head = total < ((16u - shift) & 15u) ? total : ((16u - shift) & 15u);
    if (threadIdx.x < head) dst[threadIdx.x] = out[shift + threadIdx.x];
    const uint32_t nb = (total - head) >> 4;
    for (uint32_t c = threadIdx.x; c < nb; c += kTextTile)
        *reinterpret_cast<text_u4 *>(dst + head + 16 * c) = *(lds_u4)(out_generic + shift + head + 16 * c);
    const uint32_t t0 = head + 16 * nb;
    if (threadIdx.x < total - t0) dst[t0 + threadIdx.x] = out[shift + t0 + threadIdx.x];
}

// one observation line / one value with its separator, written at q (global memory or LDS)
template <typename P>
__device__ __forceinline__ void text_put_obs(P q, uint32_t c, uint32_t lc, uint32_t p, uint32_t lp, const c2b_dec::Text &tu, const c2b_dec::Text &tv) {
    c2b_dec::uint_emit(c, lc, q); q += lc; *q++ = ' ';
    c2b_dec::uint_emit(p, lp, q); q += lp; *q++ = ' ';
    c2b_dec::emit(tu, q); q += tu.len; *q++ = ' ';
    c2b_dec::emit(tv, q); q += tv.len; *q = '\n';
}

// observation lines: "camera point u v\n"
template <bool EMIT>
__global__ __launch_bounds__(kTextTile) void k_text_obs(const uint32_t *__restrict__ cam_idx, const uint32_t *__restrict__ pt_idx,
                                                        const double2 *__restrict__ uv, int64_t n,
                                                        const c2b_dec::Tables *__restrict__ T, uint32_t *__restrict__ tile_len,
                                                        const uint64_t *__restrict__ tile_base, char *__restrict__ img) {
    __shared__ uint32_t sh[kTextTile / 64];
    __shared__ __attribute__((aligned(16))) char out[EMIT ? kEmitCap + 16 : 16];
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
    char *dst = img + tile_base[blockIdx.x];
    const uint32_t shift = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u);
    if (total + shift <= (uint32_t)kEmitCap) {                // workgroup-uniform
        if (i < n) text_put_obs((lds_char)out + shift + at, c, lc, p, lp, tu, tv);
        __syncthreads();
        text_copy_out(out, shift, total, dst);
    } else if (i < n) {
        text_put_obs(dst + at, c, lc, p, lp, tu, tv);
    }
}

// rows of `width` values (`stride` doubles apart) joined by spaces, one row per line: a unit = one value + what follows it
template <bool EMIT>
__global__ __launch_bounds__(kTextTile) void k_text_vals(const double *__restrict__ src, int64_t n_rows, int width, int stride,
                                                         const c2b_dec::Tables *__restrict__ T, uint32_t *__restrict__ tile_len,
                                                         const uint64_t *__restrict__ tile_base, char *__restrict__ img) {
    __shared__ uint32_t sh[kTextTile / 64];
    __shared__ __attribute__((aligned(16))) char out[EMIT ? kEmitCap + 16 : 16];
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
    char *dst = img + tile_base[blockIdx.x];
    const uint32_t shift = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u);
    const char sep = col == width - 1 ? '\n' : ' ';
    if (total + shift <= (uint32_t)kEmitCap) {                // workgroup-uniform
        if (j < n) {
            const lds_char q = (lds_char)out + shift + at;
            c2b_dec::emit(t, q);
            q[t.len] = sep;
        }
        __syncthreads();
        text_copy_out(out, shift, total, dst);
    } else if (j < n) {
        char *q = dst + at;
        c2b_dec::emit(t, q);
        q[t.len] = sep;
    }
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
constexpr int kParseLong = 400;                              // a longer token is declined
constexpr int kParseOver = 448;                              // bytes of the next tile staged with this one (>= kParseLong + 1, x16)

__device__ __forceinline__ bool text_ws(uint32_t c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r'; }

// bit b of the result: byte b of these 16 (the file's bytes [at, at + 16)) starts a token
__device__ __forceinline__ uint32_t text_starts(const text_u4 v, bool prev_ws, int64_t at, int64_t n) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
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

// (the buffer is padded with zeros to a multiple of 16 and one chunk more)
__global__ __launch_bounds__(kTextTile) void k_text_count_tokens(const char *__restrict__ raw, int64_t n, uint32_t *__restrict__ tile_cnt) {
    __shared__ uint32_t sh[kTextTile / 64];
    const int64_t at = ((int64_t)blockIdx.x * kTextTile + threadIdx.x) * kParseBytes;
    uint32_t starts = 0;
    if (at < n) starts = text_starts(*reinterpret_cast<const text_u4 *>(raw + at), at == 0 ? true : text_ws((uint32_t)(uint8_t)raw[at - 1]), at, n);
    uint32_t total;
    (void)text_tile_scan((uint32_t)__builtin_popcount(starts), sh, &total);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}

// The tile's bytes and the first kParseOver of the next tile go to LDS with 16-byte loads; tokens are measured and
// parsed from there (a token's bytes are read one by one, twice: dependent loads, 64 cycles each from LDS instead of a
// trip through the vector memory path).
__global__ __launch_bounds__(kTextTile) void k_text_parse(const char *__restrict__ raw, int64_t n, const uint64_t *__restrict__ tile_base,
                                                          const c2b_dec::ParseTables *__restrict__ T, uint64_t n_cam, uint64_t n_pts, uint64_t n_obs,
                                                          uint32_t *__restrict__ cam_idx, uint32_t *__restrict__ pt_idx,
                                                          double *__restrict__ uv, double *__restrict__ bal9, double *__restrict__ pts4,
                                                          uint32_t *__restrict__ flags) {
    __shared__ uint32_t sh[kTextTile / 64];
    __shared__ __attribute__((aligned(16))) char tile[kParseTile + kParseOver];
    __shared__ uint16_t tok[kParseTile / 2];                  // a token and the whitespace behind it take two bytes at least
    const int64_t tile0 = (int64_t)blockIdx.x * kParseTile, at = tile0 + (int64_t)threadIdx.x * kParseBytes;
    const int64_t n16 = (n + 15) & ~(int64_t)15;
    text_u4 mine = {0u, 0u, 0u, 0u};
    if (at < n16) mine = *reinterpret_cast<const text_u4 *>(raw + at);
    *(lds_u4)(tile + threadIdx.x * kParseBytes) = mine;
    if (threadIdx.x < kParseOver / 16) {
        const int64_t o = tile0 + kParseTile + (int64_t)threadIdx.x * 16;
        text_u4 v = {0u, 0u, 0u, 0u};
        if (o < n16) v = *reinterpret_cast<const text_u4 *>(raw + o);
        *(lds_u4)(tile + kParseTile + threadIdx.x * 16) = v;
    }
    __syncthreads();
    const lds_cchar text = (lds_cchar)tile;
    const int32_t here = (int32_t)threadIdx.x * kParseBytes;
    const bool prev_ws = threadIdx.x > 0 ? text_ws((uint32_t)(uint8_t)text[here - 1]) : (at == 0 ? true : text_ws((uint32_t)(uint8_t)raw[at - 1]));
    uint32_t starts = at < n ? text_starts(mine, prev_ws, at, n) : 0u;
    uint32_t total;
    uint32_t rank = text_tile_scan((uint32_t)__builtin_popcount(starts), sh, &total);
    // Where the tile's tokens start, in token order: the parsing below takes ONE token per lane and trip, whatever the
    // spacing of the file (16 bytes per lane hold 0-8 tokens: parsing them where they were found left most lanes idle
    // behind the fullest one), and deals the tokens of a group of 256 so that each wave sees ONE kind: consecutive
    // tokens of the observation block are camera, point, u, v, so wave w takes the tokens whose index is w mod 4 past
    // the block's first -- an integer parse or a decimal parse per wave, not both (8.2 -> see DESIGN 3.2).
    for (; starts != 0; starts &= starts - 1) tok[rank++] = (uint16_t)(here + __builtin_ctz(starts));
    __syncthreads();
    const uint64_t k0 = tile_base[blockIdx.x];
    const uint64_t k_obs = 3, k_cam = k_obs + 4 * n_obs, k_pts = k_cam + 9 * n_cam, k_end = k_pts + 3 * n_pts;
    const int64_t left = n - tile0;                           // bytes of the file from this tile's first on
    const int32_t limit = left < (int64_t)(kParseTile + kParseOver) ? (int32_t)left : kParseTile + kParseOver;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t slot = 4u * lane + ((wave - (uint32_t)(k0 - k_obs)) & 3u);          // a bijection of 0..255
    for (uint32_t r = slot; r < total; r += kTextTile) {
        const uint64_t k = k0 + r;
        if (k < k_obs || k >= k_end) continue;                // the header is the host's; nom leaves what follows the last point unread
        const int32_t s = (int32_t)tok[r];
        int32_t len = 1;
        while (s + len < limit && len <= kParseLong && !text_ws((uint32_t)(uint8_t)text[s + len])) ++len;
        int status = len > kParseLong ? (int)c2b_dec::PARSE_IRREGULAR : (int)c2b_dec::PARSE_OK;
        const lds_cchar t = text + s;
        if (k < k_cam) {
            const uint64_t j = k - k_obs, i = j >> 2;
            const int f = (int)(j & 3);
            if (f < 2) {
                const uint64_t v = status ? 0 : c2b_dec::parse_u64(t, len, &status);
                if (!status) {
                    if (v >= (f == 0 ? n_cam : n_pts)) atomicOr(flags + 1 + f, 1u);
                    else (f == 0 ? cam_idx : pt_idx)[i] = (uint32_t)v;
                }
            } else {
                const double v = status ? 0.0 : c2b_dec::parse_f64(t, len, T, &status);
                if (!status) uv[2 * i + (f - 2)] = v;
            }
        } else {
            const double v = status ? 0.0 : c2b_dec::parse_f64(t, len, T, &status);
            if (!status) {
                if (k < k_pts) bal9[k - k_cam] = v;
                else { const uint64_t j = k - k_pts, q = j / 3; pts4[4 * q + (j - 3 * q)] = v; }
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

// ---- observations that are not in camera order ----------------------------------------------------------------------
// BAProblem::new (src/baproblem.rs:342-355) pushes every observation onto its camera's list in FILE order: a stable sort
// of the observations by camera.  The files of the generators are camera-major already; the Bundle Adjustment in the
// Large datasets list their observations point by point.  A least-significant-digit radix sort, 8 bits a pass, of
// (camera, index of the observation in the file): per pass a histogram per tile, one scan over [digit][tile] (which is
// exactly the order "smaller digit first, then earlier tile"), and a scatter in which one wave walks its tile 64 keys
// at a time -- the lanes holding the same digit find each other with eight ballots, their order among themselves is
// their lane order, and the tile's running count per digit lives in LDS -- so every pass is stable by construction.
constexpr int kSortTile = 2048;                              // keys per wave

__global__ __launch_bounds__(64) void k_sort_hist(const uint32_t *__restrict__ keys, int64_t n, int shift, int64_t n_tiles,
                                                  uint32_t *__restrict__ hist) {
    __shared__ uint32_t cnt[256];
    const int lane = threadIdx.x;
    for (int d = lane; d < 256; d += 64) cnt[d] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int64_t base = (int64_t)blockIdx.x * kSortTile;
    for (int c = 0; c < kSortTile; c += 64) {
        const int64_t i = base + c + lane;
        if (i < n) atomicAdd(&cnt[(keys[i] >> shift) & 255u], 1u);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < 256; d += 64) hist[(int64_t)d * n_tiles + blockIdx.x] = cnt[d];
}

// vals_in == nullptr: the value of key i is i (the first pass)
__global__ __launch_bounds__(64) void k_sort_scatter(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, int64_t n,
                                                     int shift, int64_t n_tiles, const uint32_t *__restrict__ offs,
                                                     uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out) {
    __shared__ uint32_t at[256];
    const int lane = threadIdx.x;
    for (int d = lane; d < 256; d += 64) at[d] = offs[(int64_t)d * n_tiles + blockIdx.x];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int64_t base = (int64_t)blockIdx.x * kSortTile;
    const uint64_t below = (1ull << lane) - 1ull;
    for (int c = 0; c < kSortTile; c += 64) {
        const int64_t i = base + c + lane;
        if (base + c >= n) break;                             // wave-uniform
        const bool valid = i < n;
        const uint32_t key = valid ? keys_in[i] : 0u;
        const uint32_t val = valid ? (vals_in ? vals_in[i] : (uint32_t)i) : 0u;
        const uint32_t d = (key >> shift) & 255u;
        uint64_t same = __builtin_amdgcn_ballot_w64(valid);  // the lanes of this chunk with my digit
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t has = __builtin_amdgcn_ballot_w64(((d >> b) & 1u) != 0);
            same &= ((d >> b) & 1u) ? has : ~has;
        }
        const uint32_t start = at[d];                         // every lane of `same` reads the same count ...
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            const uint32_t dst = start + (uint32_t)__builtin_popcountll(same & below);
            keys_out[dst] = key;
            vals_out[dst] = val;
            if ((same & below) == 0) at[d] = start + (uint32_t)__builtin_popcountll(same);      // ... its first lane moves it on
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// the observations in their sorted order: perm[j] = position in the file of the j-th observation of the camera-major list
__global__ __launch_bounds__(256) void k_text_gather_obs(const uint32_t *__restrict__ perm, int64_t n, const uint32_t *__restrict__ pt_in,
                                                         const double2 *__restrict__ uv_in, uint32_t *__restrict__ pt_out,
                                                         double2 *__restrict__ uv_out) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const uint32_t i = perm[j];
    pt_out[j] = pt_in[i];
    uv_out[j] = uv_in[i];
}
