// Host-side bounding-volume hierarchy over the mesh triangles for the generator's occlusion rays
// (src/generate.rs:455-476; the reference commits the mesh to an Embree scene, src/bin/city2ba.rs:515-521).  Built
// once per mesh on the CPU (binned SAH, median splits below a depth cap), traversed on the device by
// k_occlusion_bvh.  The hierarchy only prunes: a leaf runs the same float32 ray/triangle test as the brute-force
// kernel, and every box is inflated by a few ulps of the scene so that pruning does not change the answer.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace c2b_host {

constexpr int kBvhLeafMax = 4;          // triangles per leaf
constexpr int kBvhSahDepth = 30;        // below this depth: median splits (halving => total depth < 64)
constexpr int kBvhBins = 16;
constexpr int32_t kBvhEmpty = INT32_MIN;

// both children's boxes live in the parent: one 64-byte read per traversal step
struct BvhNode {
    float lo0[3], hi0[3], lo1[3], hi1[3];
    int32_t c0, c1;                     // >= 0: inner node index; < 0: ~((first_slot << 3) | (count - 1)); kBvhEmpty
    int32_t pad[2];
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");

struct Bvh {
    std::vector<BvhNode> nodes;
    std::vector<float> tris;            // [n_slots][12]: v0.xyz, e1.xyz, e2.xyz, 3 pad; slot order = leaf order
    std::vector<uint32_t> order;        // slot -> input triangle
    int depth = 0;
};

struct BvhBuilder {
    const float *tri;
    std::vector<float> plo, phi, cen;   // per input triangle
    std::vector<uint32_t> idx;
    Bvh *out;
    float margin;

    static void grow(float lo[3], float hi[3], const float *a, const float *b) {
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], a[k]); hi[k] = std::max(hi[k], b[k]); }
    }
    static float half_area(const float lo[3], const float hi[3]) {
        const float d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
    }

    int32_t leaf(size_t begin, size_t end) const { return ~(int32_t)(((uint32_t)begin << 3) | (uint32_t)(end - begin - 1)); }

    // returns the child reference of [begin, end) and its (un-inflated) bounds
    int32_t build(size_t begin, size_t end, int depth, float lo[3], float hi[3]) {
        out->depth = std::max(out->depth, depth);
        const float inf = std::numeric_limits<float>::infinity();
        float clo[3] = {inf, inf, inf}, chi[3] = {-inf, -inf, -inf};
        for (int k = 0; k < 3; ++k) { lo[k] = inf; hi[k] = -inf; }
        for (size_t i = begin; i < end; ++i) {
            const uint32_t t = idx[i];
            grow(lo, hi, &plo[3 * t], &phi[3 * t]);
            grow(clo, chi, &cen[3 * t], &cen[3 * t]);
        }
        const size_t n = end - begin;
        if (n <= (size_t)kBvhLeafMax) return leaf(begin, end);
        int axis = 0;
        for (int k = 1; k < 3; ++k) if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
        size_t mid = begin + n / 2;
        bool split_done = false;
        if (depth < kBvhSahDepth && chi[axis] > clo[axis]) {
            float best = inf;
            int best_axis = -1, best_bin = -1;
            for (int ax = 0; ax < 3; ++ax) {
                const float ext = chi[ax] - clo[ax];
                if (!(ext > 0.0f)) continue;
                const float scale = (float)kBvhBins / ext;
                int cnt[kBvhBins] = {0};
                float blo[kBvhBins][3], bhi[kBvhBins][3];
                for (int b = 0; b < kBvhBins; ++b) for (int k = 0; k < 3; ++k) { blo[b][k] = inf; bhi[b][k] = -inf; }
                for (size_t i = begin; i < end; ++i) {
                    const uint32_t t = idx[i];
                    const int b = std::min(kBvhBins - 1, std::max(0, (int)((cen[3 * t + ax] - clo[ax]) * scale)));
                    ++cnt[b];
                    grow(blo[b], bhi[b], &plo[3 * t], &phi[3 * t]);
                }
                float rarea[kBvhBins];
                int rcnt[kBvhBins];
                float alo[3] = {inf, inf, inf}, ahi[3] = {-inf, -inf, -inf};
                int c = 0;
                for (int b = kBvhBins - 1; b > 0; --b) {
                    if (cnt[b]) grow(alo, ahi, blo[b], bhi[b]);
                    c += cnt[b];
                    rcnt[b] = c;
                    rarea[b] = c ? half_area(alo, ahi) : 0.0f;
                }
                for (int k = 0; k < 3; ++k) { alo[k] = inf; ahi[k] = -inf; }
                c = 0;
                for (int b = 0; b < kBvhBins - 1; ++b) {         // split between bin b and b + 1
                    if (cnt[b]) grow(alo, ahi, blo[b], bhi[b]);
                    c += cnt[b];
                    if (!c || !rcnt[b + 1]) continue;
                    const float cost = (float)c * half_area(alo, ahi) + (float)rcnt[b + 1] * rarea[b + 1];
                    if (cost < best) { best = cost; best_axis = ax; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                const float scale = (float)kBvhBins / (chi[best_axis] - clo[best_axis]);
                auto it = std::partition(idx.begin() + begin, idx.begin() + end, [&](uint32_t t) {
                    const int b = std::min(kBvhBins - 1, std::max(0, (int)((cen[3 * t + best_axis] - clo[best_axis]) * scale)));
                    return b <= best_bin;
                });
                mid = (size_t)(it - idx.begin());
                split_done = mid > begin && mid < end;
            }
        }
        if (!split_done) {                                        // median split (also the depth-cap path)
            mid = begin + n / 2;
            std::nth_element(idx.begin() + begin, idx.begin() + mid, idx.begin() + end, [&](uint32_t a, uint32_t b) {
                const float ca = cen[3 * a + axis], cb = cen[3 * b + axis];
                return ca < cb || (ca == cb && a < b);
            });
        }
        const size_t ni = out->nodes.size();
        out->nodes.emplace_back();
        float l0[3], h0[3], l1[3], h1[3];
        const int32_t c0 = build(begin, mid, depth + 1, l0, h0);
        const int32_t c1 = build(mid, end, depth + 1, l1, h1);
        BvhNode &nd = out->nodes[ni];
        for (int k = 0; k < 3; ++k) {
            nd.lo0[k] = l0[k] - margin; nd.hi0[k] = h0[k] + margin;
            nd.lo1[k] = l1[k] - margin; nd.hi1[k] = h1[k] + margin;
        }
        nd.c0 = c0; nd.c1 = c1; nd.pad[0] = nd.pad[1] = 0;
        return (int32_t)ni;
    }
};

inline void bvh_build(const float *tri9, int64_t n_tri, Bvh &out) {
    out = Bvh();
    BvhBuilder b;
    b.tri = tri9;
    b.out = &out;
    b.plo.resize((size_t)n_tri * 3); b.phi.resize((size_t)n_tri * 3); b.cen.resize((size_t)n_tri * 3);
    b.idx.resize((size_t)n_tri);
    float max_abs = 0.0f;
    for (int64_t t = 0; t < n_tri; ++t) {
        const float *q = tri9 + 9 * t;
        for (int k = 0; k < 3; ++k) {
            const float lo = std::min(q[k], std::min(q[3 + k], q[6 + k])), hi = std::max(q[k], std::max(q[3 + k], q[6 + k]));
            b.plo[3 * (size_t)t + k] = lo; b.phi[3 * (size_t)t + k] = hi;
            b.cen[3 * (size_t)t + k] = 0.5f * lo + 0.5f * hi;
            if (std::isfinite(lo) && std::isfinite(hi)) max_abs = std::max(max_abs, std::max(std::fabs(lo), std::fabs(hi)));
        }
        b.idx[(size_t)t] = (uint32_t)t;
    }
    b.margin = max_abs * 4.8e-7f + 1e-30f;                        // 4 ulps of the largest coordinate
    out.nodes.reserve((size_t)n_tri / 2 + 2);
    float lo[3], hi[3];
    if (n_tri <= kBvhLeafMax) {                                    // a single leaf still needs a root node
        out.nodes.emplace_back();
        const int32_t c0 = n_tri ? b.build(0, (size_t)n_tri, 0, lo, hi) : kBvhEmpty;
        BvhNode &nd = out.nodes[0];
        std::memset(&nd, 0, sizeof nd);
        for (int k = 0; k < 3; ++k) { nd.lo0[k] = n_tri ? lo[k] - b.margin : 0.0f; nd.hi0[k] = n_tri ? hi[k] + b.margin : 0.0f; }
        nd.c0 = c0; nd.c1 = kBvhEmpty;
    } else {
        b.build(0, (size_t)n_tri, 0, lo, hi);                      // root = node 0
    }
    out.order = b.idx;
    out.tris.assign((size_t)n_tri * 12, 0.0f);
    for (int64_t s = 0; s < n_tri; ++s) {
        const float *q = tri9 + 9 * (size_t)out.order[(size_t)s];
        float *d = &out.tris[12 * (size_t)s];
        for (int k = 0; k < 3; ++k) { d[k] = q[k]; d[3 + k] = q[3 + k] - q[k]; d[6 + k] = q[6 + k] - q[k]; }
    }
}

}  // namespace c2b_host
