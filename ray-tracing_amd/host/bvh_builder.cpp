// bvh_builder.cpp — host BVH construction (binned SAH BVH2).
//
// Takes the place of the Vulkan driver's acceleration-structure build that the
// reference triggers at /root/reference/src/scene.rs:56-64 (and refreshes at
// src/app.rs:61).  There is no reference code to follow; the contract is only
// "nearest hit over all triangles" (SURVEY §8 A4).
//
// Shape: top-down, 16 centroid bins per axis, SAH with Ctrav = Ctri = 1,
// leaves of <= max_leaf primitives.  Subtrees whose SAH recursion could exceed
// max_depth fall back to object-median splits, so depth <= max_depth always
// and the kernel can size its traversal stack from Bvh2::max_depth.
// Big inputs build subtrees on std::threads (one task per subtree above a size
// threshold); node pairs are handed out by an atomic cursor.
#include "rt_host.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <mutex>
#include <thread>

namespace rt_host {

namespace {

constexpr float kInf = std::numeric_limits<float>::infinity();

struct Box3 {
    float lo[3] = {kInf, kInf, kInf};
    float hi[3] = {-kInf, -kInf, -kInf};
    void grow(const float *plo, const float *phi) {
        for (int k = 0; k < 3; k++) {
            lo[k] = plo[k] < lo[k] ? plo[k] : lo[k];
            hi[k] = phi[k] > hi[k] ? phi[k] : hi[k];
        }
    }
    void grow(const Box3 &b) { grow(b.lo, b.hi); }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.0f) || !(dy >= 0.0f) || !(dz >= 0.0f)) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Builder {
    const Aabb *boxes;
    BvhBuildOptions opt;
    Bvh2 *out;
    std::atomic<uint32_t> next_node{1};
    std::atomic<uint32_t> max_depth{0};
    std::atomic<uint32_t> max_leaf{0};
    std::atomic<uint64_t> n_leaves{0};
    std::atomic<int> spare_threads{0};
    static constexpr uint32_t kParallelThreshold = 1u << 16;

    static inline float centroid(const Aabb &b, int ax) { return 0.5f * b.lo[ax] + 0.5f * b.hi[ax]; }

    void note_depth(uint32_t d) {
        uint32_t cur = max_depth.load(std::memory_order_relaxed);
        while (d > cur && !max_depth.compare_exchange_weak(cur, d, std::memory_order_relaxed)) {}
    }
    void make_leaf(uint32_t node, uint32_t first, uint32_t count, uint32_t depth) {
        Bvh2Node &n = out->nodes[node];
        n.left = first;
        n.count = count;
        note_depth(depth);
        n_leaves.fetch_add(1, std::memory_order_relaxed);
        uint32_t cur = max_leaf.load(std::memory_order_relaxed);
        while (count > cur && !max_leaf.compare_exchange_weak(cur, count, std::memory_order_relaxed)) {}
    }

    // smallest depth a balanced split of `count` primitives still needs
    static uint32_t levels_needed(uint32_t count, uint32_t max_leaf) {
        uint32_t l = 0;
        uint64_t cap = max_leaf;
        while (cap < count) { cap *= 2; ++l; }
        return l;
    }

    void build(uint32_t node, uint32_t first, uint32_t count, uint32_t depth) {
        uint32_t *order = out->order.data();
        Box3 bounds, cbounds;
        for (uint32_t i = first; i < first + count; i++) {
            const Aabb &b = boxes[order[i]];
            bounds.grow(b.lo, b.hi);
            float c[3] = {centroid(b, 0), centroid(b, 1), centroid(b, 2)};
            cbounds.grow(c, c);
        }
        Bvh2Node &n = out->nodes[node];
        memcpy(n.box.lo, bounds.lo, 12);
        memcpy(n.box.hi, bounds.hi, 12);

        if (count <= 1) { make_leaf(node, first, count, depth); return; }

        const uint32_t need = levels_needed(count, opt.max_leaf);
        const bool force_median = depth + need + 1 >= opt.max_depth;
        if (count <= opt.max_leaf && force_median) { make_leaf(node, first, count, depth); return; }

        uint32_t mid = 0;
        bool have_split = false;
        if (!force_median) {
            const uint32_t NB = opt.bins;
            int best_axis = -1;
            uint32_t best_bin = 0;
            float best_cost = kInf;
            // one pass bins all three axes (the gather of boxes[order[i]] is what costs on big scenes)
            Box3 bin_box[3][32];
            uint32_t bin_cnt[3][32];
            float k1[3], c0[3];
            bool live[3];
            for (int ax = 0; ax < 3; ax++) {
                const float ext = cbounds.hi[ax] - cbounds.lo[ax];
                live[ax] = ext > 0.0f;
                k1[ax] = live[ax] ? (float)NB / ext : 0.0f;
                c0[ax] = cbounds.lo[ax];
                for (uint32_t i = 0; i < NB; i++) { bin_box[ax][i] = Box3(); bin_cnt[ax][i] = 0; }
            }
            for (uint32_t i = first; i < first + count; i++) {
                const Aabb &b = boxes[order[i]];
                for (int ax = 0; ax < 3; ax++) {
                    if (!live[ax]) continue;
                    int bi = (int)((centroid(b, ax) - c0[ax]) * k1[ax]);
                    bi = bi < 0 ? 0 : (bi >= (int)NB ? (int)NB - 1 : bi);
                    bin_cnt[ax][bi]++;
                    bin_box[ax][bi].grow(b.lo, b.hi);
                }
            }
            float right_area[32];
            uint32_t right_cnt[32];
            for (int ax = 0; ax < 3; ax++) {
                if (!live[ax]) continue;
                Box3 acc;
                uint32_t cnt = 0;
                for (uint32_t i = NB - 1; i > 0; i--) {
                    acc.grow(bin_box[ax][i]);
                    cnt += bin_cnt[ax][i];
                    right_area[i] = acc.half_area();
                    right_cnt[i] = cnt;
                }
                acc = Box3();
                cnt = 0;
                for (uint32_t i = 1; i < NB; i++) {
                    acc.grow(bin_box[ax][i - 1]);
                    cnt += bin_cnt[ax][i - 1];
                    if (cnt == 0 || right_cnt[i] == 0) continue;
                    float cost = acc.half_area() * (float)cnt + right_area[i] * (float)right_cnt[i];
                    if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = i; }
                }
            }
            if (best_axis >= 0) {
                // SAH: split cost Ctrav + (Al*Nl + Ar*Nr)/A against leaf cost N
                float parent_area = bounds.half_area();
                float split_cost = opt.traversal_cost + (parent_area > 0.0f ? best_cost / parent_area : (float)count);
                if (count <= opt.max_leaf && (float)count <= split_cost) { make_leaf(node, first, count, depth); return; }
                const float kb = k1[best_axis], cb0 = c0[best_axis];
                uint32_t *b = order + first, *e = order + first + count;
                uint32_t *m = std::partition(b, e, [&](uint32_t id) {
                    int bi = (int)((centroid(boxes[id], best_axis) - cb0) * kb);
                    bi = bi < 0 ? 0 : (bi >= (int)NB ? (int)NB - 1 : bi);
                    return (uint32_t)bi < best_bin;
                });
                mid = (uint32_t)(m - order);
                have_split = mid > first && mid < first + count;
            } else if (count <= opt.max_leaf) {
                make_leaf(node, first, count, depth);  // identical centroids, small enough
                return;
            }
        }
        if (!have_split) {
            // object median on the widest centroid axis (also the depth-guard path)
            int ax = 0;
            float best = -1.0f;
            for (int k = 0; k < 3; k++) {
                float e = cbounds.hi[k] - cbounds.lo[k];
                if (e > best) { best = e; ax = k; }
            }
            mid = first + count / 2;
            std::nth_element(order + first, order + mid, order + first + count, [&](uint32_t a, uint32_t b) {
                float ca = centroid(boxes[a], ax), cb = centroid(boxes[b], ax);
                return ca < cb || (ca == cb && a < b);
            });
        }

        const uint32_t left = next_node.fetch_add(2, std::memory_order_relaxed);
        n.left = left;
        n.count = 0;
        const uint32_t lcount = mid - first, rcount = first + count - mid;
        if (count >= kParallelThreshold && spare_threads.fetch_sub(1, std::memory_order_relaxed) > 0) {
            std::thread t([=]() { build(left, first, lcount, depth + 1); });
            build(left + 1, mid, rcount, depth + 1);
            t.join();
            spare_threads.fetch_add(1, std::memory_order_relaxed);
        } else {
            if (count >= kParallelThreshold) spare_threads.fetch_add(1, std::memory_order_relaxed);
            build(left, first, lcount, depth + 1);
            build(left + 1, mid, rcount, depth + 1);
        }
    }
};

}  // namespace

void build_bvh2(const Aabb *boxes, uint64_t n, const BvhBuildOptions &opt_in, Bvh2 &out) {
    auto t0 = std::chrono::steady_clock::now();
    BvhBuildOptions opt = opt_in;
    if (opt.max_leaf == 0) opt.max_leaf = 4;
    if (opt.max_leaf > 7) opt.max_leaf = 7;      // a leaf reference keeps its count in 3 bits (rt_device.h dev_leaf_ref)
    if (opt.bins < 4) opt.bins = 4;
    if (opt.bins > 32) opt.bins = 32;
    // depth bound must leave room for a balanced tree over n primitives
    uint32_t need = Builder::levels_needed((uint32_t)std::min<uint64_t>(n, 0xFFFFFFFFull), opt.max_leaf) + 2;
    if (opt.max_depth < need) opt.max_depth = need;

    out = Bvh2();
    out.order.resize(n);
    for (uint64_t i = 0; i < n; i++) out.order[i] = (uint32_t)i;
    out.nodes.resize(n > 0 ? 2 * n + 1 : 1);

    Builder b;
    b.boxes = boxes;
    b.opt = opt;
    b.out = &out;
    uint32_t threads = opt.threads ? opt.threads : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    b.spare_threads.store((int)threads - 1);
    if (n == 0) {
        Bvh2Node &r = out.nodes[0];
        for (int k = 0; k < 3; k++) { r.box.lo[k] = kInf; r.box.hi[k] = -kInf; }
        r.left = 0;
        r.count = 0;
        out.nodes.resize(1);
        out.n_leaves = 0;
    } else {
        b.build(0, 0, (uint32_t)n, 0);
        out.nodes.resize(b.next_node.load());
        out.nodes.shrink_to_fit();
    }
    out.max_depth = b.max_depth.load();
    out.max_leaf = b.max_leaf.load();
    out.n_leaves = b.n_leaves.load();

    // SAH cost of the finished tree (for reports/tests)
    double cost = 0.0;
    if (n > 0) {
        auto ha = [](const Aabb &a) {
            double dx = (double)a.hi[0] - a.lo[0], dy = (double)a.hi[1] - a.lo[1], dz = (double)a.hi[2] - a.lo[2];
            return dx * dy + dy * dz + dz * dx;
        };
        double root = ha(out.nodes[0].box);
        if (root > 0.0)
            for (const Bvh2Node &nd : out.nodes) cost += ha(nd.box) / root * (nd.count ? (double)nd.count : 1.0);
    }
    out.sah_cost = cost;
    out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void refit_bvh2(const Aabb *boxes, Bvh2 &bvh) {
    auto t0 = std::chrono::steady_clock::now();
    for (size_t i = bvh.nodes.size(); i-- > 0;) {
        Bvh2Node &n = bvh.nodes[i];
        Box3 b;
        if (n.count) {
            for (uint32_t k = 0; k < n.count; k++) b.grow(boxes[bvh.order[n.left + k]].lo, boxes[bvh.order[n.left + k]].hi);
        } else if (bvh.nodes.size() > 1) {
            b.grow(bvh.nodes[n.left].box.lo, bvh.nodes[n.left].box.hi);
            b.grow(bvh.nodes[n.left + 1].box.lo, bvh.nodes[n.left + 1].box.hi);
        } else {
            continue;   // empty tree
        }
        memcpy(n.box.lo, b.lo, 12);
        memcpy(n.box.hi, b.hi, 12);
    }
    bvh.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

bool gather_triangles(const RtVertex *vertices, uint64_t n_vertices, const uint32_t *indices, uint64_t n_indices,
                      const RtGeometryPrimitive *prims, uint32_t n_prims, const float *transforms,
                      std::vector<float> &corners, std::vector<uint32_t> &ids, std::string &err) {
    uint64_t n_tris = 0;
    for (uint32_t g = 0; g < n_prims; g++) {
        const RtGeometryPrimitive &p = prims[g];
        if ((uint64_t)p.first_index + p.index_count > n_indices) { err = "geometry index range exceeds the index buffer"; return false; }
        if ((uint64_t)p.first_vertex + p.vertex_count > n_vertices) { err = "geometry vertex range exceeds the vertex buffer"; return false; }
        if (p.index_count % 3 != 0) { err = "geometry index count is not a multiple of 3"; return false; }
        n_tris += p.index_count / 3;
    }
    if (n_tris >= (1ull << 28)) { err = "more than 2^28 triangles"; return false; }
    corners.resize(n_tris * 9);
    ids.resize(n_tris * 2);
    uint64_t k = 0;
    for (uint32_t g = 0; g < n_prims; g++) {
        const RtGeometryPrimitive &p = prims[g];
        const float *m = transforms ? transforms + 12 * p.transformation_offset : nullptr;
        for (uint32_t t = 0; t < p.index_count / 3; t++, k++) {
            for (int j = 0; j < 3; j++) {
                uint32_t idx = indices[p.first_index + 3 * t + j];
                if (idx >= p.vertex_count) { err = "index outside its geometry's vertex range"; return false; }
                const float *v = vertices[p.first_vertex + idx].position;
                float *dst = &corners[9 * k + 3 * j];
                if (m) {
                    for (int r = 0; r < 3; r++) {
                        volatile float acc = (m[4 * r] * v[0] + m[4 * r + 1] * v[1]);
                        acc = acc + m[4 * r + 2] * v[2];
                        dst[r] = acc + m[4 * r + 3];
                    }
                } else {
                    dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
                }
            }
            ids[2 * k] = g;
            ids[2 * k + 1] = t;
        }
    }
    return true;
}

}  // namespace rt_host
