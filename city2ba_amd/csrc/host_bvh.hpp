// Host-side bounding-volume hierarchy over the mesh triangles for the generator's occlusion rays
// (src/generate.rs:455-476; the reference commits the mesh to an Embree scene, src/bin/city2ba.rs:515-521).  Built
// once per mesh on the CPU (binned SAH, median splits below a depth cap; large sub-ranges on separate threads, then
// renumbered depth-first so the layout does not depend on thread timing), traversed on the device by
// k_occlusion_bvh.  The hierarchy only prunes: a leaf runs the same float32 ray/triangle test as the brute-force
// kernel, and every box is inflated by a few ulps of the scene so that pruning does not change the answer.
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <future>
#include <limits>
#include <thread>
#include <vector>

namespace c2b_host {

constexpr int kBvhLeafMax = 4;          // triangles per leaf
constexpr int kBvhSahDepth = 30;        // below this depth: median splits (halving => total depth < 64)
constexpr int kBvhBins = 16;
constexpr size_t kBvhParMin = 1 << 14;  // sub-ranges at least this large may be built by another thread
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
    std::vector<BvhNode> nodes;         // preallocated; slots handed out by `next` (threads build disjoint sub-ranges)
    std::atomic<size_t> next{0};
    std::atomic<int> depth{0};
    int par_depth = 0;                  // above this depth the two halves of a large range are built concurrently
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
        for (int d = this->depth.load(); d < depth && !this->depth.compare_exchange_weak(d, depth);) {}
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
        const size_t ni = next.fetch_add(1);
        float l0[3], h0[3], l1[3], h1[3];
        int32_t c0, c1;
        if (depth < par_depth && n >= kBvhParMin) {
            auto left = std::async(std::launch::async, [&] { return build(begin, mid, depth + 1, l0, h0); });
            c1 = build(mid, end, depth + 1, l1, h1);
            c0 = left.get();
        } else {
            c0 = build(begin, mid, depth + 1, l0, h0);
            c1 = build(mid, end, depth + 1, l1, h1);
        }
        BvhNode &nd = nodes[ni];
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
    b.nodes.resize((size_t)std::max<int64_t>(n_tri, 1));          // inner nodes < triangles
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    while ((1u << b.par_depth) < hw && b.par_depth < 6) ++b.par_depth;
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
    float lo[3], hi[3];
    if (n_tri <= kBvhLeafMax) {                                    // a single leaf still needs a root node
        const int32_t c0 = n_tri ? b.build(0, (size_t)n_tri, 0, lo, hi) : kBvhEmpty;
        out.nodes.resize(1);
        BvhNode &nd = out.nodes[0];
        std::memset(&nd, 0, sizeof nd);
        for (int k = 0; k < 3; ++k) { nd.lo0[k] = n_tri ? lo[k] - b.margin : 0.0f; nd.hi0[k] = n_tri ? hi[k] + b.margin : 0.0f; }
        nd.c0 = c0; nd.c1 = kBvhEmpty;
    } else {
        b.build(0, (size_t)n_tri, 0, lo, hi);                      // root = slot 0 (handed out before any recursion)
        // slots were handed out in thread-arrival order: renumber in depth-first order (deterministic, and a
        // subtree's nodes end up next to each other)
        const size_t n_nodes = b.next.load();
        std::vector<int32_t> remap(n_nodes, -1), stack{0};
        std::vector<int32_t> order;
        order.reserve(n_nodes);
        while (!stack.empty()) {
            const int32_t i = stack.back();
            stack.pop_back();
            remap[(size_t)i] = (int32_t)order.size();
            order.push_back(i);
            const BvhNode &nd = b.nodes[(size_t)i];
            if (nd.c1 >= 0) stack.push_back(nd.c1);
            if (nd.c0 >= 0) stack.push_back(nd.c0);
        }
        out.nodes.resize(n_nodes);
        for (size_t k = 0; k < n_nodes; ++k) {
            BvhNode nd = b.nodes[(size_t)order[k]];
            if (nd.c0 >= 0) nd.c0 = remap[(size_t)nd.c0];
            if (nd.c1 >= 0) nd.c1 = remap[(size_t)nd.c1];
            out.nodes[k] = nd;
        }
    }
    out.depth = b.depth.load();
    out.order = b.idx;
    out.tris.assign((size_t)n_tri * 12, 0.0f);
    for (int64_t s = 0; s < n_tri; ++s) {
        const float *q = tri9 + 9 * (size_t)out.order[(size_t)s];
        float *d = &out.tris[12 * (size_t)s];
        for (int k = 0; k < 3; ++k) { d[k] = q[k]; d[3 + k] = q[3 + k] - q[k]; d[6 + k] = q[6 + k] - q[k]; }
    }
}

// Host traversal, nearest hit with t >= 0 (scene.intersect of the camera placement, src/generate.rs:247-262): the
// device kernel's box test and triangle test, keeping the smallest distance instead of stopping at the first hit.
inline bool bvh_box_hit(const float lo[3], const float hi[3], const float o[3], const float inv[3], float tfar) {
    float tn = -std::numeric_limits<float>::infinity(), tf = std::numeric_limits<float>::infinity();
    for (int k = 0; k < 3; ++k) {
        const float a = (lo[k] - o[k]) * inv[k], b = (hi[k] - o[k]) * inv[k];
        tn = std::fmax(tn, std::fmin(a, b));                       // fmin / fmax drop a NaN operand, like the device's
        tf = std::fmin(tf, std::fmax(a, b));
    }
    return tn <= tf * 1.00001f + 1e-30f && tf >= 0.0f && tn <= tfar;
}

inline bool bvh_cast_ray(const Bvh &bvh, const float o[3], const float d[3], float *t_hit) {
    const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    float best = std::numeric_limits<float>::infinity();
    int32_t stack[64];
    int sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const BvhNode &nd = bvh.nodes[(size_t)stack[--sp]];
        for (int side = 0; side < 2; ++side) {
            const int32_t ch = side ? nd.c1 : nd.c0;
            if (ch == kBvhEmpty || !bvh_box_hit(side ? nd.lo1 : nd.lo0, side ? nd.hi1 : nd.hi0, o, inv, best)) continue;
            if (ch >= 0) { if (sp < 64) stack[sp++] = ch; continue; }
            const uint32_t code = (uint32_t)~ch;
            const size_t first = code >> 3, cnt = (code & 7u) + 1;
            for (size_t s = first; s < first + cnt; ++s) {
                const float *q = &bvh.tris[12 * s];
                const float *e1 = q + 3, *e2 = q + 6;
                const float pv[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
                const float det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
                if (det == 0.0f) continue;
                const float idet = 1.0f / det;
                const float tv[3] = {o[0] - q[0], o[1] - q[1], o[2] - q[2]};
                const float u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * idet;
                if (u < 0.0f || u > 1.0f) continue;
                const float qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
                const float w = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) * idet;
                if (w < 0.0f || u + w > 1.0f) continue;
                const float th = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * idet;
                if (th >= 0.0f && th < best) best = th;
            }
        }
    }
    *t_hit = best;
    return best < std::numeric_limits<float>::infinity();
}

}  // namespace c2b_host
