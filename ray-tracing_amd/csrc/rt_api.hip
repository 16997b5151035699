// rt_api.hip — implementation of the C ABI in include/rt_abi.h (device half).
//
// Owns the device-resident scene (repacked BVH + triangle/shading records +
// materials), the accumulation buffer and the dispatch of the render kernel.
// Mirrors the resources the reference binds at /root/reference/src/main.rs:44-54
// and the per-frame work of App::render, src/app.rs:74-102.
#include <hip/hip_runtime.h>

#include <chrono>
#include <deque>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <algorithm>
#include <vector>

#include "../../include/rt_abi.h"
#include "../host/rt_host.h"
#include "rt_device.h"
#include "rt_math.h"
#include "rt_pack.h"

hipError_t rt_launch_render(const RenderParams &P, int variant, bool count, bool aov, int ext, int persistent_blocks, hipStream_t stream);
int rt_persistent_blocks(int device, uint32_t stack_entries, bool count, bool aov, int ext, bool lds_materials, int fmt);
hipError_t rt_launch_capture(const float *accum, uint8_t *rgb, size_t n_pixels, hipStream_t stream);
hipError_t rt_launch_deinterleave(const float *gathered, float *image, const RtGatherLayout &L, hipStream_t stream);
hipError_t rt_launch_refit(DevTri *tris, const DevShade *shade, const float *xform, RtRefitBox *tribox, uint64_t n_slots, float4 *chunks,
                           uint32_t stride, const int32_t *parent, unsigned int *arrived, RtRefitBox *cbox, uint32_t n_nodes,
                           RtRefitBox *root_out, bool q8, float prim_pad, hipStream_t stream);

hipError_t rt_launch_refit8(DevTri *tris, const DevShade *shade, const float *xform, RtRefitBox *tribox, uint64_t n_slots, float4 *chunks,
                            uint32_t stride, const int32_t *parent, unsigned int *arrived, RtRefitBox *cbox, uint32_t n_nodes,
                            RtRefitBox *root_out, float prim_pad, const float *grid_lo_cell, hipStream_t stream);

static std::atomic<uint64_t> g_device_allocs{0};   // hipMalloc calls this library has made (rt_debug_alloc_count; tests/test_zz_no_alloc.py)

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define RT_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) return fail(RT_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// every device allocation of the library goes through here, so that a test can prove a call path makes none
hipError_t dev_malloc(void **p, size_t bytes) {
    g_device_allocs.fetch_add(1, std::memory_order_relaxed);
    return hipMalloc(p, bytes);
}

template <typename T>
void dev_free(T *&p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

}  // namespace

// ---- static scratch of rt_dispatch_frames_static (rt_abi.h): callers that cannot allocate ----
// One array per GPU (every device that loads this code object holds its own copy at its own address, until the
// process ends): [0, 256) the work-queue head, then the per-frame staging planes of a multi-frame launch.
// In host code the NAME is only a stand-in for the symbol; the device address comes from hipGetSymbolAddress,
// taken once per device in rt_create with that device current, and only that pointer is ever handed to
// kernels / hipMemsetAsync.
#define RT_STATIC_HEADER_BYTES 256u
static __device__ __align__(256) unsigned char scratch[RT_STATIC_HEADER_BYTES + RT_STATIC_STAGE_BYTES];
#define RT_MAX_DEVICES 64
// Per device, set up once (under g_scratch_mutex) by the first rt_create on it: the address of `scratch` there, and the event that
// orders the users of that one array.  Every static dispatch makes its stream wait for the event of the static dispatch before it
// on that GPU (whatever context and stream that was) and records its own behind its last kernel, so two contexts on one GPU
// cannot zero each other's live queue head or overwrite each other's staging planes.  Neither call allocates.
#include <mutex>
static std::mutex g_scratch_mutex;
static void *g_scratch_ptr[RT_MAX_DEVICES];
static hipEvent_t g_scratch_done[RT_MAX_DEVICES];
static bool g_scratch_used[RT_MAX_DEVICES];        // g_scratch_done[d] has been recorded at least once

// Host copy of the scene inputs (rt_models_update rebuilds from them) and the host BVH.  One per context, or ONE shared
// by all the contexts of an rt_group: the scene is loaded, built and repacked once, however many GPUs render it.
struct SceneHost {
    std::vector<RtVertex> vertices;
    std::vector<uint32_t> indices;
    std::vector<RtGeometryPrimitive> primitives;
    std::vector<float> transforms;   // n_primitives x 12, row-major 3x4
    std::vector<RtGpuModel> models;
    std::vector<RtSphere> spheres;   // build-defined primitive (BASELINE config C4)
    rt_host::Bvh2 bvh;               // kept for host refits (rt_models_update)
    uint64_t bvh_prims = 0;
};

struct rt_context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    std::shared_ptr<SceneHost> hs = std::make_shared<SceneHost>();
    uint32_t n_textures = 0;
    DevTexture *d_textures = nullptr;
    uint32_t *d_texels = nullptr;

    // device scene
    DevNode *d_nodes = nullptr;
    DevTri *d_tris = nullptr;
    DevShade *d_shade = nullptr;
    DevMaterial *d_materials = nullptr;
    size_t materials_capacity = 0;   // records d_materials holds
    uint64_t n_nodes = 0, n_tris = 0, node_bytes = 0;
    uint32_t stack_entries = 1;
    float scene_abs[3] = {0.0f, 0.0f, 0.0f};
    float prim_pad_scale = 0.0f;     // the scene scale the last host build widened the primitives' boxes by (rt_prim_pad)
    bool have_scene = false;
    bool last_update_on_device = false;   // rt_models_update: the last call was served by device_refit alone
    bool models_committed = false;   // the last build_and_upload / device_refit got as far as changing device state
    // device refit (RT_OPT_REFIT = 2): topology + scratch, uploaded with the scene
    int32_t *d_parent = nullptr;     // per node: parent * 2 + child slot, root -1
    unsigned int *d_arrived = nullptr;
    RtRefitBox *d_cbox = nullptr, *d_tribox = nullptr, *d_root = nullptr;
    float *d_xform = nullptr;        // n_geometries x 12
    uint32_t node_stride = RT_NODE_CHUNKS;
    uint64_t refits = 0, rebuilds = 0;
    double build_seconds = 0.0;

    // accumulation
    float *d_accum = nullptr;
    bool accum_owned = false;
    uint32_t width = 0, height = 0;
    uint32_t shard_index = 0, shard_count = 1;
    uint32_t local_bands = 0;
    bool accum_fresh = false;
    uint32_t *d_aov = nullptr;
    uint64_t aov_elems = 0;
    uint8_t *d_rgb = nullptr;        // rt_capture_rgb8's device buffer
    uint64_t rgb_bytes = 0;

    DevCounters *d_counters = nullptr;
    DevQueue *d_queue = nullptr;
    int persistent_blocks = 0;       // resident workgroups of the kernel instance `persistent_sel` (0 = not asked yet)
    int persistent_sel = -1;
    int cu_count = 0;
    uint64_t dispatches = 0, launches = 0;

    bool opt_count = false, opt_aov = false, opt_cw = false;
    uint32_t opt_sampler = 0;        // RT_OPT_SAMPLER
    int opt_refit = 2;               // RT_OPT_REFIT: 0 rebuild, 1 host refit, 2 device refit
    int opt_node_stride = 0;
    int opt_node_format = 0;         // RT_OPT_NODE_FORMAT: 0 by size, 1 48-byte nodes, 2 32-byte q8 nodes
    bool node_q8 = false;            // format of the nodes on the device
    bool node_wide = false;          // 8-ary nodes (RT_OPT_NODE_FORMAT 3 / 4)
    bool node_compact = false;       // ... in 64-byte records (DevNode8c, format 4) with their corner grid
    float grid_lo[3] = {0.0f, 0.0f, 0.0f}, grid_cell[3] = {1.0f, 1.0f, 1.0f};
    // the 8-ary format is traversed by every kernel variant
    bool opt_kernel_wide_ok() const { return true; }
    int opt_kernel = 2;
    int32_t tune[8] = {0, 0, 0, 0, 64, 0, 0, 0};   // keys 100..107: phase weights NODE/LEAF/SHADE (0 = by scene size, see launch), workgroups per CU (0 = all that fit), FETCH weight, children per 8-ary node (4..8, else 8), builder max_leaf, bins
    float4 *d_stage = nullptr;       // per-frame sample sums of a multi-frame launch (lazy, grows)
    uint64_t stage_elems = 0;
    int opt_pipeline = 1;            // RT_OPT_FRAME_PIPELINE
    float sah_traversal_cost = 1.0f; // key 110, in 1/16
    uint32_t bvh_max_depth = 0;      // key 111, 0 = builder default
    bool collapse_greedy = false;    // key 112: 8-ary collapse by the greedy rule instead of the cost-optimal cut (measurement)
    uint64_t stage_budget = 4ull << 30;   // bytes of staging a multi-frame launch may use (key 109, tests)
    uint32_t chunk = 0;              // queue grain in pixel slots; 0 = by the amount of work (see launch())
    bool timed = false;
};

namespace {

uint32_t bands_for(uint32_t h, uint32_t shard_index, uint32_t shard_count) {
    uint32_t bands = (h + 7u) / 8u;
    if (bands <= shard_index) return 0;
    return (bands - shard_index + shard_count - 1u) / shard_count;
}

// rows the shard actually holds (the last band may be partial)
uint64_t local_rows_for(uint32_t h, uint32_t shard_index, uint32_t shard_count) {
    uint32_t lb = bands_for(h, shard_index, shard_count);
    if (lb == 0) return 0;
    uint32_t last_band = (lb - 1) * shard_count + shard_index;
    uint32_t last_rows = h - last_band * 8u;
    if (last_rows > 8u) last_rows = 8u;
    return (uint64_t)(lb - 1) * 8u + last_rows;
}

// Repack the host BVH2 into DevNode/DevTri/DevShade order.  Child boxes are
// widened by a few ulp so that a hit the triangle test accepts a hair outside
// its triangle is not culled by the box (tie policy).
struct Repacked {
    std::vector<DevNode> nodes;      // node i; uploaded at 16-byte chunk i * node_stride
    uint32_t node_stride = RT_NODE_CHUNKS;
    bool q8 = false;                 // 32-byte records (rt_pack.h), node_stride 2
    bool wide = false;               // 80-byte 8-ary nodes (DevNode8) in `nodes8`, node_stride 5
    std::vector<DevNode8> nodes8;
    bool compact = false;            // wide: 64-byte records (DevNode8c) in `nodes8c`, node_stride 4
    std::vector<DevNode8c> nodes8c;
    float grid_lo[3] = {0.0f, 0.0f, 0.0f}, grid_cell[3] = {1.0f, 1.0f, 1.0f};   // compact: the corner grid (rt_pack.h rt_grid_for)
    bool collapse_dp = true;         // wide: cost-optimal cut of the BVH2 (false: greedy, largest area first)
    int width8 = 8;                  // wide: children a node may hold (developer knob 105: what a narrower tree would visit)
    std::vector<uint32_t> order8;    // wide: slot -> primitive id (a node's leaf children are consecutive slots)
    std::vector<int32_t> parent;     // node -> parent * 2 + child slot (device refit)
    bool refit_sources_agree = true; // shading-record corners == BLAS corners (see repack)
    std::vector<DevTri> tris;
    std::vector<DevShade> shade;
    uint32_t stack_entries = 1;
    float scene_abs[3] = {0.0f, 0.0f, 0.0f};   // max |coordinate| of the padded root box
};

inline float pad_lo(float v) { return rt_pad_lo(v); }
inline float pad_hi(float v) { return rt_pad_hi(v); }

// Both children of one node: padded boxes → lower corner + outward-rounded f16 offsets (rt_pack.h).
void set_children(DevNode &n, const rt_host::Aabb &box0, int32_t ref0, const rt_host::Aabb &box1, int32_t ref1, bool q8) {
    if (q8) {
        memset(&n, 0, sizeof n);
        rt_pack_children_q8(n, box0.lo, box0.hi, box1.lo, box1.hi);
        n.ref0 = ref0; n.ref1 = ref1;
        return;
    }
    rt_pack_children(n, box0.lo, box0.hi, box1.lo, box1.hi);
    n.ref0 = ref0; n.ref1 = ref1;
    n.pad = 0;
}

// The 8-ary tree of RT_OPT_NODE_FORMAT = 3 (DevNode8, rt_device.h), collapsed from the SAH BVH2: a node adopts its
// grandchildren while it has room (largest surface area first), primitives of BVH2 leaves become one-primitive children.
// Children get their slots by position (octant order), inner children consecutive node indices, leaf children
// consecutive primitive slots; big trees are collapsed on several threads (ranges handed out by atomic cursors).
void collapse8(const rt_host::Bvh2 &bvh, const rt_host::Aabb *prim_boxes, Repacked &out) {
    const uint64_t n_prims = bvh.order.size();
    out.order8.assign(n_prims, 0u);
    uint64_t n_inner2 = 0;
    for (const auto &n : bvh.nodes) if (n.count == 0) n_inner2++;
    auto half_area = [](const rt_host::Aabb &b) {
        const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
        return (dx >= 0.0f && dy >= 0.0f && dz >= 0.0f) ? dx * dy + dy * dz + dz * dx : 0.0f;
    };
    // Which BVH2 nodes become 8-ary nodes?  Every 8-ary node visited costs one node step whatever it holds and every primitive is
    // some node's child whatever the cut, so the expected work is the summed surface area of the BVH2 nodes that are KEPT as
    // 8-ary nodes.  Minimised exactly by dynamic programming over the BVH2 (Ylitie, Karras, Laine 2017, section 4.1):
    //   T(n, i) = least area below n when n's subtree may use i child slots of its parent (i = 1: n itself becomes a node),
    //   T(n, 1) = area(n) + min_k T(l, k) + T(r, W - k),   T(n, i) = min(T(n, i - 1), min_k T(l, k) + T(r, i - k)),
    //   a leaf of c primitives: 0 with c slots or more, area(n) with fewer (a node of its own).
    // Children follow their parents in bvh.nodes, so one backward sweep fills the table.  40 bytes per BVH2 node: trees beyond
    // 2^28 nodes (and developer knob 112 = 1) use the greedy rule below (largest area first).
    const int W = out.width8;
    const bool dp = out.collapse_dp && bvh.nodes.size() > 1 && bvh.nodes.size() <= (1u << 28);
    std::vector<float> T;
    std::vector<uint8_t> pick_k;                            // [n * 8 + i]: i = 2..W-1: 0 = as with i - 1 slots, else k; i = W: k of T(n, 1)
    if (dp) {
        T.assign(bvh.nodes.size() * 8, 0.0f);
        pick_k.assign(bvh.nodes.size() * 8, 0);
        for (size_t n = bvh.nodes.size(); n-- > 0;) {
            const rt_host::Bvh2Node &nd = bvh.nodes[n];
            float *t = &T[n * 8];
            const float area = half_area(nd.box);
            if (nd.count) {
                for (int i = 1; i < 8; i++) t[i] = i >= (int)nd.count ? 0.0f : area;
                continue;
            }
            const float *tl = &T[(size_t)nd.left * 8], *tr = tl + 8;
            auto distribute = [&](int j, uint8_t &k_out) {
                float best = 3.4e38f;
                for (int k = 1; k < j; k++) {
                    const float c = tl[k] + tr[j - k];
                    if (c < best) { best = c; k_out = (uint8_t)k; }
                }
                return best;
            };
            t[1] = area + distribute(W, pick_k[n * 8 + (W & 7)]);
            for (int i = 2; i < W; i++) {
                uint8_t k = 0;
                const float d = distribute(i, k);
                if (d < t[i - 1]) { t[i] = d; pick_k[n * 8 + i] = k; } else { t[i] = t[i - 1]; pick_k[n * 8 + i] = 0; }
            }
            for (int i = W; i < 8; i++) t[i] = t[W - 1];
        }
    }
    const uint64_t n_nodes_max = dp ? bvh.nodes.size() : n_inner2 + 1;
    if (out.compact) {
        out.nodes8c.assign(n_nodes_max, DevNode8c());
        memset(out.nodes8c.data(), 0, out.nodes8c.size() * sizeof(DevNode8c));
    } else {
        out.nodes8.assign(n_nodes_max, DevNode8());
        memset(out.nodes8.data(), 0, out.nodes8.size() * sizeof(DevNode8));
    }
    out.parent.assign(n_nodes_max, -1);                  // node -> parent * 8 + child slot (device refit)
    std::atomic<uint64_t> next_node{1}, next_slot{0};
    std::atomic<uint32_t> max_depth{0};
    struct Item { uint32_t src, dst, depth; };

    // one node: fills out.nodes8[it.dst], appends its inner children to `todo`
    auto process = [&](const Item &it, std::vector<Item> &todo) {
        uint32_t open[8];
        bool kept[8] = {false, false, false, false, false, false, false, false};      // this BVH2 node stays a node (an inner child)
        int n_open = 0, total = 0;
        auto size_of = [&](uint32_t i) { return bvh.nodes[i].count ? (int)bvh.nodes[i].count : 1; };
        const rt_host::Bvh2Node &src = bvh.nodes[it.src];
        if (src.count || bvh.nodes.size() == 1) { open[n_open++] = it.src; total = (int)src.count; }      // the whole scene is one leaf (or empty)
        else if (dp) {
            struct Grant { uint32_t node; int slots; } work[16];
            int n_work = 0;
            const int k = pick_k[(size_t)it.src * 8 + (W & 7)];
            work[n_work++] = {src.left + 1u, W - k};
            work[n_work++] = {src.left, k};
            while (n_work) {
                const Grant g = work[--n_work];
                const rt_host::Bvh2Node &c = bvh.nodes[g.node];
                int i = g.slots;
                if (c.count) { kept[n_open] = i < (int)c.count; open[n_open++] = g.node; continue; }
                while (i > 1 && pick_k[(size_t)g.node * 8 + i] == 0) i--;
                if (i == 1) { kept[n_open] = true; open[n_open++] = g.node; continue; }
                const int kk = pick_k[(size_t)g.node * 8 + i];
                work[n_work++] = {c.left + 1u, i - kk};
                work[n_work++] = {c.left, kk};
            }
        } else { open[0] = src.left; open[1] = src.left + 1; n_open = 2; total = size_of(open[0]) + size_of(open[1]); }
        for (; !dp;) {
            int pick = -1;
            float best = -1.0f;
            for (int i = 0; i < n_open; i++) {
                const rt_host::Bvh2Node &c = bvh.nodes[open[i]];
                if (c.count) continue;
                if (total - 1 + size_of(c.left) + size_of(c.left + 1) > out.width8) continue;
                const float a = half_area(c.box);
                if (a > best) { best = a; pick = i; }
            }
            if (pick < 0) break;
            const uint32_t l = bvh.nodes[open[pick]].left;
            total += size_of(l) + size_of(l + 1) - 1;
            open[pick] = l;
            open[n_open++] = l + 1;
        }
        // children: inner subtrees and single primitives
        struct Child { bool inner; uint32_t ref; rt_host::Aabb box; };
        Child ch[8];
        int n_ch = 0;
        for (int i = 0; i < n_open; i++) {
            const rt_host::Bvh2Node &c = bvh.nodes[open[i]];
            if (dp ? kept[i] : (!c.count && !(bvh.nodes.size() == 1))) { ch[n_ch++] = {true, open[i], c.box}; continue; }
            for (uint32_t k = 0; k < c.count && n_ch < 8; k++) {
                const uint32_t prim = bvh.order[c.left + k];
                ch[n_ch++] = {false, prim, prim_boxes[prim]};
            }
        }
        // octant slots: greedily give each child the free slot whose direction (+-1, +-1, +-1) its offset from the node's centre favours most
        float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
        for (int i = 0; i < n_ch; i++)
            for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], ch[i].box.lo[k]); hi[k] = fmaxf(hi[k], ch[i].box.hi[k]); }
        float cost[8][8];
        for (int i = 0; i < n_ch; i++)
            for (int s8 = 0; s8 < 8; s8++) {
                float c = 0.0f;
                for (int k = 0; k < 3; k++) {
                    const float off = (0.5f * ch[i].box.lo[k] + 0.5f * ch[i].box.hi[k]) - (0.5f * lo[k] + 0.5f * hi[k]);
                    c += ((s8 >> k) & 1) ? off : -off;
                }
                cost[i][s8] = c == c ? c : 0.0f;
            }
        int slot_of[8], child_at[8];
        for (int i = 0; i < 8; i++) { slot_of[i] = -1; child_at[i] = -1; }
        for (int round = 0; round < n_ch; round++) {
            int bi = -1, bs = -1;
            float bc = -3.4e38f;
            for (int i = 0; i < n_ch; i++) {
                if (slot_of[i] >= 0) continue;
                for (int s8 = 0; s8 < 8; s8++)
                    if (child_at[s8] < 0 && (bi < 0 || cost[i][s8] > bc)) { bc = cost[i][s8]; bi = i; bs = s8; }
            }
            slot_of[bi] = bs;
            child_at[bs] = bi;
        }
        uint32_t imask = 0, lmask = 0, n_inner = 0, n_leaf = 0;
        float blo[8][3], bhi[8][3];
        for (int s8 = 0; s8 < 8; s8++) {
            const int i = child_at[s8];
            if (i < 0) continue;
            if (ch[i].inner) { imask |= 1u << s8; n_inner++; } else { lmask |= 1u << s8; n_leaf++; }
            memcpy(blo[s8], ch[i].box.lo, 12);
            memcpy(bhi[s8], ch[i].box.hi, 12);
        }
        const uint64_t base = n_inner ? next_node.fetch_add(n_inner, std::memory_order_relaxed) : 0;
        const uint64_t slot0 = n_leaf ? next_slot.fetch_add(n_leaf, std::memory_order_relaxed) : 0;
        if (out.compact) {
            DevNode8c &d = out.nodes8c[it.dst];
            rt_node8c_set_links(d, imask, lmask, (uint32_t)base, (uint32_t)slot0);
            rt_pack_node8c(d, blo, bhi, imask | lmask, out.grid_lo, out.grid_cell);
        } else {
            DevNode8 &d = out.nodes8[it.dst];
            d.exps_imask = imask << 24;
            d.lmask = lmask;
            d.child_base = (uint32_t)(base * out.node_stride);
            d.tri_base = (uint32_t)slot0;
            rt_pack_node8(d, blo, bhi, imask | lmask);
        }
        uint32_t ki = 0, kl = 0;
        for (int s8 = 0; s8 < 8; s8++) {
            const int i = child_at[s8];
            if (i < 0) continue;
            if (ch[i].inner) {
                out.parent[base + ki] = (int32_t)(it.dst * 8u + (uint32_t)s8);
                todo.push_back({ch[i].ref, (uint32_t)(base + ki++), it.depth + 1});
            }
            else out.order8[slot0 + kl++] = ch[i].ref;
        }
        uint32_t cur = max_depth.load(std::memory_order_relaxed);
        while (it.depth > cur && !max_depth.compare_exchange_weak(cur, it.depth, std::memory_order_relaxed)) {}
    };

    // breadth-first until there is enough independent work, then depth-first per thread
    std::vector<Item> frontier{{0u, 0u, 0u}};
#ifdef RT_COLLAPSE_THREADS      // developer builds: the node / primitive numbering the collapse would give on that many threads
    const unsigned hw = RT_COLLAPSE_THREADS;
#else
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
#endif
    const unsigned want = n_prims >= (1u << 18) ? hw * 16u : 0u;
    size_t head = 0;
    while (head < frontier.size() && frontier.size() - head < want) { const Item it = frontier[head++]; process(it, frontier); }
    if (want == 0) { while (head < frontier.size()) { const Item it = frontier[head++]; process(it, frontier); } }
    if (head < frontier.size()) {
        std::atomic<size_t> cursor{head};
        auto worker = [&]() {
            std::vector<Item> stack;
            for (;;) {
                const size_t i = cursor.fetch_add(1, std::memory_order_relaxed);
                if (i >= frontier.size()) break;
                stack.push_back(frontier[i]);
                while (!stack.empty()) { const Item it = stack.back(); stack.pop_back(); process(it, stack); }
            }
        };
        std::vector<std::thread> th;
        for (unsigned k = 1; k < hw; k++) th.emplace_back(worker);
        worker();
        for (auto &t : th) t.join();
    }
    if (out.compact) out.nodes8c.resize(next_node.load()); else out.nodes8.resize(next_node.load());
    out.parent.resize(next_node.load());
    // When a node at depth d is visited the stack holds at most d groups (one may be pushed each time a child is taken from a
    // group with siblings left; the root is the only child of a virtual group and pushes nothing), and primitives push nothing:
    // 2 words x the depth of the deepest node, exactly.  (Every KiB counts: C5 fits 5 workgroups per CU at 26 words, 4 at 30.)
    out.stack_entries = std::max(2u, 2u * max_depth.load());
}

void repack(const rt_host::Bvh2 &bvh, const std::vector<float> &corners, const std::vector<uint32_t> &ids,
            const rt_context &ctx, Repacked &out, const rt_host::Aabb *prim_boxes) {
    const uint64_t n_prims = bvh.order.size();     // triangles, then spheres
    if (out.wide) collapse8(bvh, prim_boxes, out);
    const std::vector<uint32_t> &order = out.wide ? out.order8 : bvh.order;
    const uint64_t n_real_tris = ids.size() / 2;
    const uint64_t n_tris = n_prims;
    out.tris.resize(n_tris ? n_tris : 1);
    out.shade.resize(n_tris ? n_tris : 1);
    memset(out.tris.data(), 0, out.tris.size() * sizeof(DevTri));
    memset(out.shade.data(), 0, out.shade.size() * sizeof(DevShade));
    // The device refit (rt_refit.hip) rebuilds a triangle from the corner positions in its shading record, which come
    // from the MODEL's vertex/index ranges (main.wgsl:86-91), while the BLAS takes them from the GeometryPrimitive's
    // (src/scene.rs:141-147).  The reference makes the two the same; a caller that does not gets the host refit.
    std::atomic<bool> sources_agree{true};
    auto fill = [&](uint64_t slot_lo, uint64_t slot_hi) {
    for (uint64_t slot = slot_lo; slot < slot_hi; slot++) {
        const uint32_t gid = order[slot];
        DevTri &t = out.tris[slot];
        if (gid >= n_real_tris) {
            const uint32_t si = (uint32_t)(gid - n_real_tris);
            const RtSphere &sp = ctx.hs->spheres[si];
            t.ax = sp.center[0]; t.ay = sp.center[1]; t.az = sp.center[2];
            t.e1x = sp.radius;
            t.gid = gid;
            t.is_sphere = 1;
            DevShade &s = out.shade[slot];
            memcpy(s.p0, sp.center, 12);
            s.n0x = 1.0f / sp.radius;
            s.geometry = 0x80000000u | si;
            s.primitive = 0;
            s.pad0 = sp.model;
            continue;
        }
        const float *c = &corners[9 * (size_t)gid];
        t.ax = c[0]; t.ay = c[1]; t.az = c[2];
        t.e1x = c[3] - c[0]; t.e1y = c[4] - c[1]; t.e1z = c[5] - c[2];   // e1 = b - a
        t.e2x = c[6] - c[0]; t.e2y = c[7] - c[1]; t.e2z = c[8] - c[2];   // e2 = c - a
        t.gid = gid;
        // shading attributes come from the MODEL's ranges (main.wgsl:86-91),
        // untransformed (object_to_world is the identity instance transform)
        const uint32_t g = ids[2 * (size_t)gid], p = ids[2 * (size_t)gid + 1];
        const RtGpuModel &m = ctx.hs->models[g];
        const uint32_t is = m.index_start + p * 3u;
        const RtVertex *v[3];
        for (int j = 0; j < 3; j++) v[j] = &ctx.hs->vertices[m.vertex_start + ctx.hs->indices[is + j]];
        DevShade &s = out.shade[slot];
        memcpy(s.p0, v[0]->position, 12); s.n0x = v[0]->normal[0]; s.n0y = v[0]->normal[1]; s.n0z = v[0]->normal[2]; memcpy(s.uv0, v[0]->uv, 8);
        memcpy(s.p1, v[1]->position, 12); memcpy(s.n1, v[1]->normal, 12); memcpy(s.uv1, v[1]->uv, 8);
        memcpy(s.p2, v[2]->position, 12); memcpy(s.n2, v[2]->normal, 12); memcpy(s.uv2, v[2]->uv, 8);
        s.geometry = g;
        s.primitive = p;
        const RtGeometryPrimitive &gp = ctx.hs->primitives[g];
        for (int j = 0; j < 3; j++)
            if (memcmp(ctx.hs->vertices[gp.first_vertex + ctx.hs->indices[gp.first_index + 3u * p + j]].position, v[j]->position, 12) != 0)
                sources_agree.store(false, std::memory_order_relaxed);
    }
    };
    {
        // records are independent: spread big scenes over the host threads
        const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        const unsigned parts = n_tris >= (1u << 18) ? hw : 1u;
        std::vector<std::thread> th;
        for (unsigned k = 1; k < parts; k++) th.emplace_back(fill, n_tris * k / parts, n_tris * (k + 1) / parts);
        fill(0, n_tris / parts);
        for (auto &t : th) t.join();
    }
    out.refit_sources_agree = sources_agree.load();

    auto leaf_ref = [&](const rt_host::Bvh2Node &n) { return dev_leaf_ref(n.left, n.count); };
    const rt_host::Bvh2Node &root = bvh.nodes[0];
    for (int k = 0; k < 3 && n_tris; k++) {
        const float m = fmaxf(fabsf(pad_lo(root.box.lo[k])), fabsf(pad_hi(root.box.hi[k])));
        out.scene_abs[k] = m <= 3.0e38f ? m : 3.0e38f;   // non-finite coordinates: every box test passes, still correct
    }
    if (out.wide) return;            // nodes8 / stack_entries are collapse8's
    if (root.count > 0 || bvh.nodes.size() == 1) {
        // the whole scene is one leaf (or empty): a root whose second child is empty
        out.nodes.resize(1);
        memset(&out.nodes[0], 0, sizeof(DevNode));
        rt_host::Aabb box = root.box;
        if (n_tris == 0) for (int k = 0; k < 3; k++) { box.lo[k] = 0.0f; box.hi[k] = 0.0f; }
        set_children(out.nodes[0], box, dev_leaf_ref(root.left, n_tris ? root.count : 0), box, dev_leaf_ref(0, 0), out.q8);
        out.stack_entries = 2;
        return;
    }
    // depth-first: a node's first child subtree follows it in memory
    uint64_t n_inner = 0;
    for (const auto &n : bvh.nodes) if (n.count == 0) n_inner++;
    out.nodes.resize(n_inner);
    memset(out.nodes.data(), 0, n_inner * sizeof(DevNode));
    out.parent.assign(n_inner, -1);
    struct Item { uint32_t src, dst; };
    std::deque<Item> todo;
    todo.push_back({0u, 0u});
    uint32_t next = 1;
    while (!todo.empty()) {
        // the first RT_TOP_NODES nodes are numbered breadth-first (the top of the tree sits together at the start of
        // the array), the rest depth-first
        const bool bfs = next <= RT_TOP_NODES;
        Item it = bfs ? todo.front() : todo.back();
        if (bfs) todo.pop_front(); else todo.pop_back();
        const rt_host::Bvh2Node &n = bvh.nodes[it.src];
        DevNode &d = out.nodes[it.dst];
        const rt_host::Bvh2Node &c0 = bvh.nodes[n.left], &c1 = bvh.nodes[n.left + 1];
        int32_t r0, r1;
        uint32_t d0 = 0, d1 = 0;
        if (c0.count) r0 = leaf_ref(c0); else { d0 = next++; r0 = (int32_t)(d0 * out.node_stride); out.parent[d0] = (int32_t)(it.dst * 2u); }
        if (c1.count) r1 = leaf_ref(c1); else { d1 = next++; r1 = (int32_t)(d1 * out.node_stride); out.parent[d1] = (int32_t)(it.dst * 2u + 1u); }
        set_children(d, c0.box, r0, c1.box, r1, out.q8);
        if (bfs) {
            if (!c0.count) todo.push_back({n.left, d0});
            if (!c1.count) todo.push_back({n.left + 1, d1});
        } else {
            if (!c1.count) todo.push_back({n.left + 1, d1});
            if (!c0.count) todo.push_back({n.left, d0});
        }
    }
    // a depth-first walk pushes at most one reference per inner node on the path: max_depth entries (+1 spare)
    out.stack_entries = bvh.max_depth + 1;
}

// models[] → the material records the kernel reads (main.wgsl:53-68)
void convert_materials(const rt_context &ctx, std::vector<DevMaterial> &mats) {
    mats.assign(ctx.hs->models.size() ? ctx.hs->models.size() : 1, DevMaterial());
    memset(mats.data(), 0, mats.size() * sizeof(DevMaterial));
    for (size_t i = 0; i < ctx.hs->models.size(); i++) {
        const RtMaterial &m = ctx.hs->models[i].material;
        DevMaterial &d = mats[i];
        memcpy(d.diffuse, m.metal.diffuse_color, 12);
        memcpy(d.specular, m.metal.specular_color, 12);
        d.specular_probability = m.metal.specular_probability;
        d.roughness = m.metal.roughness;
        for (int k = 0; k < 3; k++) d.emitted[k] = m.metal.emission_color[k] * m.metal.emission_strength;  // main.wgsl:56
        d.refractive_index = m.refractive_index;
        d.tag = m.tag;
        d.diffuse_texture = m.metal.diffuse_texture;
        d.normal_texture = m.metal.normal_texture;
    }
}

// What the host half of a scene build hands to the device half: built once, uploaded to one GPU or to every GPU of a group.
struct Built {
    Repacked rp;
    std::vector<DevMaterial> mats;
    std::vector<uint8_t> packed_nodes;   // the node array as it is uploaded (restrided when node_stride != RT_NODE_CHUNKS)
    uint64_t n_prims = 0;
    double build_seconds = 0.0;
    float prim_pad_scale = 0.0f;
    bool valid = false;                  // host_build filled it
};

// Host half: validate, gather triangles, build (or refit) the BVH2 in ctx->hs, repack it.  Touches no device state.
int host_build(rt_context *ctx, bool allow_refit, Built &out) {
    ctx->models_committed = false;
    std::string err;
    std::vector<float> corners;
    std::vector<uint32_t> ids;
    if (!rt_host::gather_triangles(ctx->hs->vertices.data(), ctx->hs->vertices.size(), ctx->hs->indices.data(), ctx->hs->indices.size(),
                                   ctx->hs->primitives.data(), (uint32_t)ctx->hs->primitives.size(),
                                   ctx->hs->transforms.empty() ? nullptr : ctx->hs->transforms.data(), corners, ids, err))
        return fail(RT_ERR_INVALID_ARGUMENT, "scene: %s", err.c_str());
    const uint64_t n_tris = ids.size() / 2;
    // models[] is indexed by geometry_index (main.wgsl:86)
    if (ctx->hs->models.size() < ctx->hs->primitives.size())
        return fail(RT_ERR_INVALID_ARGUMENT, "scene: %zu models for %zu geometries", ctx->hs->models.size(), ctx->hs->primitives.size());
    for (uint64_t t = 0; t < n_tris; t++) {
        const RtGpuModel &m = ctx->hs->models[ids[2 * t]];
        uint64_t is = (uint64_t)m.index_start + 3ull * ids[2 * t + 1];
        if (is + 3 > ctx->hs->indices.size())
            return fail(RT_ERR_INVALID_ARGUMENT, "scene: model index range exceeds the index buffer");
        for (int j = 0; j < 3; j++)
            if ((uint64_t)m.vertex_start + ctx->hs->indices[is + j] >= ctx->hs->vertices.size())
                return fail(RT_ERR_INVALID_ARGUMENT, "scene: model vertex range exceeds the vertex buffer");
    }
    const uint64_t n_prims = n_tris + ctx->hs->spheres.size();
    if (n_prims >= (1ull << 28)) return fail(RT_ERR_INVALID_ARGUMENT, "scene: more than 2^28 primitives");
    std::vector<rt_host::Aabb> boxes(n_prims);
    for (uint64_t t = 0; t < n_tris; t++) {
        const float *c = &corners[9 * t];
        const bool finite = rt_all_finite(c, 9);          // else: never hit, and kept out of the tree's boxes (rt_pack.h RT_EMPTY_BOX)
        for (int k = 0; k < 3; k++) {
            boxes[t].lo[k] = finite ? fminf(c[k], fminf(c[3 + k], c[6 + k])) : RT_EMPTY_BOX;
            boxes[t].hi[k] = finite ? fmaxf(c[k], fmaxf(c[3 + k], c[6 + k])) : -RT_EMPTY_BOX;
        }
    }
    for (uint64_t i = 0; i < ctx->hs->spheres.size(); i++) {
        const RtSphere &sp = ctx->hs->spheres[i];
        if (!(sp.radius > 0.0f)) return fail(RT_ERR_INVALID_ARGUMENT, "sphere %llu: radius must be positive", (unsigned long long)i);
        if (sp.model >= ctx->hs->models.size()) return fail(RT_ERR_INVALID_ARGUMENT, "sphere %llu: model index out of range", (unsigned long long)i);
        const float r = sp.radius * 1.0001f + 1e-6f;   // the quadratic's roots are not exact
        const bool finite = rt_all_finite(sp.center, 3) && rt_all_finite(&r, 1);
        for (int k = 0; k < 3; k++) {
            boxes[n_tris + i].lo[k] = finite ? sp.center[k] - r : RT_EMPTY_BOX;
            boxes[n_tris + i].hi[k] = finite ? sp.center[k] + r : -RT_EMPTY_BOX;
        }
    }
    // Node format (RT_OPT_NODE_FORMAT).  The compact formats (8-bit plane offsets) need the scene's extent to fit their scale range.
    rt_host::Aabb scene_box;
    for (int k = 0; k < 3; k++) { scene_box.lo[k] = 3.4e38f; scene_box.hi[k] = -3.4e38f; }
    bool scene_finite = n_prims > 0;
    for (uint64_t i = 0; i < n_prims; i++)
        for (int k = 0; k < 3; k++) {
            if (!(boxes[i].lo[k] >= -3.0e38f && boxes[i].hi[k] <= 3.0e38f)) scene_finite = false;
            scene_box.lo[k] = fminf(scene_box.lo[k], boxes[i].lo[k]);
            scene_box.hi[k] = fmaxf(scene_box.hi[k], boxes[i].hi[k]);
        }
    // Every primitive's box is widened by 2^-17 of the scene's scale (rt_pack.h rt_prim_pad): the triangle test's
    // own rounding grows with the distance between ray origin and triangle and with 1 / |cos| of the incidence, so a ray that
    // grazes a triangle within rounding of its edge can be ACCEPTED by the test while passing beside the triangle's exact box.
    // Found by tests/checkers/format_diff.py (one pixel-frame of config C4 at 1080p: the binary formats missed a hit that brute force,
    // the oracle's BVH and the 8-ary format find).  No finite pad is a proof (1 / |cos| is unbounded); this one makes the
    // case rarer by orders of magnitude for a fraction of a percent of extra box hits.
    {
        // scene scale = where the primitives ARE (4 standard deviations of their centres), not the largest coordinate: one
        // huge ground quad must not inflate ten thousand small triangles' boxes (C3: 2.2 % slower with the largest coordinate)
        float scene_r = 0.0f;
        for (int k = 0; k < 3 && scene_finite; k++) scene_r = fmaxf(scene_r, fmaxf(fabsf(scene_box.lo[k]), fabsf(scene_box.hi[k])));
        if (scene_finite) {
            double mean[3] = {0, 0, 0}, var = 0.0;
            for (uint64_t i = 0; i < n_prims; i++) for (int k = 0; k < 3; k++) mean[k] += 0.5 * ((double)boxes[i].lo[k] + boxes[i].hi[k]);
            for (int k = 0; k < 3; k++) mean[k] /= (double)n_prims;
            for (uint64_t i = 0; i < n_prims; i++)
                for (int k = 0; k < 3; k++) { const double d = 0.5 * ((double)boxes[i].lo[k] + boxes[i].hi[k]) - mean[k]; var += d * d; }
            const float spread = 4.0f * (float)sqrt(var / (double)n_prims);
            scene_r = fminf(scene_r, fmaxf(spread, scene_r * 0.015625f));
        }
        const float pad = rt_prim_pad(scene_r);
        out.prim_pad_scale = scene_r;
        for (uint64_t i = 0; i < n_prims && scene_finite; i++)
            for (int k = 0; k < 3; k++) { boxes[i].lo[k] -= pad; boxes[i].hi[k] += pad; }
        for (int k = 0; k < 3 && scene_finite; k++) { scene_box.lo[k] -= pad; scene_box.hi[k] += pad; }
    }
    const bool fits = scene_finite && rt_q8_fits(scene_box.lo, scene_box.hi);
    // Format by size (RT_OPT_NODE_FORMAT = 0): scenes whose nodes stay in the caches keep 48-byte binary nodes (fewest VALU
    // instructions per box; C3: 16.0 vs 18.0 ms per step); past ~12 MiB of nodes the fetches miss and the 8-ary nodes win
    // (five fetches for eight boxes, 30 visits per ray instead of 67 on C4: 100.6 vs 110.2 ms per step with 32-byte binary nodes).
    // A BVH2 over n primitives with leaves of ~2 has ~n inner nodes of 48 bytes.
    // (tools/format_threshold.py, lattice scenes at 1080p, 8-ary against binary nodes: +4 % at 95 k primitives, +1 % at 429 k, -2.7 % at 572 k,
    // -5.8 % at 858 k: the switch sits at ~12 MiB of would-be binary nodes, half a million primitives)
    const bool big = n_prims * sizeof(DevNode) / 2 >= (12ull << 20);
    const bool wide = fits && ctx->opt_kernel_wide_ok() && (ctx->opt_node_format == 3 || ctx->opt_node_format == 4 || (ctx->opt_node_format == 0 && big));
    // 64-byte records (format 4) unless the 80-byte ones are asked for; they address 2^26 nodes
    bool compact = wide && ctx->opt_node_format != 3;
    rt_host::Bvh2 &bvh = ctx->hs->bvh;
    // (a tree built with leaves of more than 4 primitives cannot be collapsed into 8-ary nodes: rebuild)
    if (allow_refit && ctx->opt_refit != 0 && ctx->hs->bvh_prims == n_prims && n_prims > 0 && !(wide && bvh.max_leaf > 4)) {
        // ≙ acceleration_structure.update(), src/app.rs:61: same primitives, moved
        rt_host::refit_bvh2(boxes.data(), bvh);
        ctx->refits++;
    } else {
        rt_host::BvhBuildOptions opt;
        if (ctx->tune[6] > 0) opt.max_leaf = (uint32_t)ctx->tune[6];
        // 8-ary nodes hold one primitive per leaf child: every primitive's own box is tested before the primitive is fetched,
        // so small BVH2 leaves cost nothing extra and two of them must fit one node
        if (wide) opt.max_leaf = ctx->tune[6] > 0 ? std::min<uint32_t>(opt.max_leaf, 4u) : 2u;
        if (ctx->tune[7] > 0) opt.bins = (uint32_t)ctx->tune[7];
        opt.traversal_cost = ctx->sah_traversal_cost;
        if (ctx->bvh_max_depth > 0) opt.max_depth = ctx->bvh_max_depth;
        rt_host::build_bvh2(boxes.data(), n_prims, opt, bvh);
        ctx->hs->bvh_prims = n_prims;
        ctx->rebuilds++;
    }
    if (bvh.nodes.size() * 4ull >= (1ull << 31)) return fail(RT_ERR_INVALID_ARGUMENT, "scene: BVH has too many nodes");
    Repacked &rp = out.rp;
    // Node stride: 48 B (default) packs the tree tightest; 64 B keeps every node inside one 64-B memory sector.
    // Measured: 48 B is 1.5 % faster on C3 (cache-resident), no difference on C4 (2 GB of records).
    rp.node_stride = (ctx->opt_node_stride == 3 || ctx->opt_node_stride == 4) ? (uint32_t)ctx->opt_node_stride : (uint32_t)RT_NODE_CHUNKS;
    // 8-ary nodes (format 3) wherever they fit; else binary nodes: 32-byte q8 records when the scene's extent fits AND the
    // nodes are too many to stay in the caches (measured: C4, 244 MB of 48-byte nodes, 8 % faster; C3, 0.26 MB, 2 % slower)
    rp.wide = wide;
    rp.compact = compact;
    rp.collapse_dp = !ctx->collapse_greedy;
    rp.width8 = (ctx->tune[5] >= 4 && ctx->tune[5] <= 8) ? ctx->tune[5] : 8;
    rp.q8 = !wide && fits && ctx->opt_node_format == 2;
    if (rp.q8) rp.node_stride = 2;
    if (rp.wide) rp.node_stride = ctx->opt_node_stride == 8 ? 8u : (uint32_t)RT_NODE8_CHUNKS;
    if (rp.compact) {
        rp.node_stride = RT_NODE8C_CHUNKS;
        rt_grid_for(scene_box.lo, scene_box.hi, rp.grid_lo, rp.grid_cell);
    }
    repack(bvh, corners, ids, *ctx, rp, boxes.data());
    if (rp.compact && rp.nodes8c.size() > RT_NODE8C_MAX_NODES) {
        // more nodes than a 26-bit child index addresses (beyond ~300 M primitives): the 80-byte records instead
        compact = false;
        rp = Repacked();
        rp.wide = true;
        rp.node_stride = RT_NODE8_CHUNKS;
        repack(bvh, corners, ids, *ctx, rp, boxes.data());
    }
    out.build_seconds = bvh.build_seconds;
    out.n_prims = n_prims;
    out.valid = true;
    convert_materials(*ctx, out.mats);
    if (rp.wide && !rp.compact && rp.node_stride != RT_NODE8_CHUNKS) {
        const size_t pitch = (size_t)rp.node_stride * 16u;
        out.packed_nodes.assign(rp.nodes8.size() * pitch, 0);
        for (size_t i = 0; i < rp.nodes8.size(); i++) memcpy(out.packed_nodes.data() + i * pitch, &rp.nodes8[i], sizeof(DevNode8));
    }
    if (!rp.wide && rp.node_stride != RT_NODE_CHUNKS) {
        // other strides (64-byte slots, 32-byte q8 records = the first 32 bytes of a DevNode): restride on the host and
        // upload in one piece (a pitched copy of 10^8 32-byte rows takes minutes)
        const size_t pitch = (size_t)rp.node_stride * 16u, row = pitch < sizeof(DevNode) ? pitch : sizeof(DevNode);
        out.packed_nodes.assign(rp.nodes.size() * pitch ? rp.nodes.size() * pitch : 1, 0);
        for (size_t i = 0; i < rp.nodes.size(); i++) memcpy(out.packed_nodes.data() + i * pitch, &rp.nodes[i], row);
    }
    return RT_OK;
}

// Device half: replace the context's device scene with `b`.
int device_upload(rt_context *ctx, const Built &b) {
    const Repacked &rp = b.rp;
    const std::vector<DevMaterial> &mats = b.mats;
    const uint64_t n_prims = b.n_prims;
    ctx->build_seconds = b.build_seconds;
    ctx->prim_pad_scale = b.prim_pad_scale;
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    ctx->models_committed = true;    // from here on the device scene is being replaced
    ctx->have_scene = false;
    dev_free(ctx->d_nodes); dev_free(ctx->d_tris); dev_free(ctx->d_shade); dev_free(ctx->d_materials);
    const size_t n_nodes = rp.compact ? rp.nodes8c.size() : (rp.wide ? rp.nodes8.size() : rp.nodes.size());
    const size_t node_bytes = n_nodes * (size_t)rp.node_stride * 16u;
    RT_HIP(dev_malloc((void **)&ctx->d_nodes, node_bytes));
    RT_HIP(dev_malloc((void **)&ctx->d_tris, rp.tris.size() * sizeof(DevTri)));
    RT_HIP(dev_malloc((void **)&ctx->d_shade, rp.shade.size() * sizeof(DevShade)));
    RT_HIP(dev_malloc((void **)&ctx->d_materials, mats.size() * sizeof(DevMaterial)));
    ctx->materials_capacity = mats.size();
    const void *node_src = rp.compact ? (const void *)rp.nodes8c.data() : rp.wide ? (rp.node_stride == RT_NODE8_CHUNKS ? (const void *)rp.nodes8.data() : (const void *)b.packed_nodes.data())
                                   : (rp.node_stride == RT_NODE_CHUNKS ? (const void *)rp.nodes.data() : (const void *)b.packed_nodes.data());
    RT_HIP(hipMemcpy(ctx->d_nodes, node_src, node_bytes, hipMemcpyHostToDevice));
    ctx->node_wide = rp.wide;
    ctx->node_compact = rp.compact;
    memcpy(ctx->grid_lo, rp.grid_lo, sizeof ctx->grid_lo);
    memcpy(ctx->grid_cell, rp.grid_cell, sizeof ctx->grid_cell);
    ctx->node_bytes = node_bytes;
    ctx->node_stride = rp.node_stride;
    ctx->node_q8 = rp.q8;
    dev_free(ctx->d_parent); dev_free(ctx->d_arrived); dev_free(ctx->d_cbox); dev_free(ctx->d_tribox); dev_free(ctx->d_root); dev_free(ctx->d_xform);
    if (ctx->opt_refit == 2 && rp.refit_sources_agree && rp.parent.size() == n_nodes && (rp.wide || n_nodes > 1)) {
        const size_t nn = n_nodes;
        RT_HIP(dev_malloc((void **)&ctx->d_parent, nn * sizeof(int32_t)));
        RT_HIP(dev_malloc((void **)&ctx->d_arrived, nn * sizeof(unsigned int)));
        RT_HIP(dev_malloc((void **)&ctx->d_cbox, (rp.wide ? 8 : 2) * nn * sizeof(RtRefitBox)));
        RT_HIP(dev_malloc((void **)&ctx->d_tribox, rp.tris.size() * sizeof(RtRefitBox)));
        RT_HIP(dev_malloc((void **)&ctx->d_root, sizeof(RtRefitBox)));
        RT_HIP(dev_malloc((void **)&ctx->d_xform, (ctx->hs->primitives.size() ? ctx->hs->primitives.size() : 1) * 12 * sizeof(float)));
        RT_HIP(hipMemcpy(ctx->d_parent, rp.parent.data(), nn * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    RT_HIP(hipMemcpy(ctx->d_tris, rp.tris.data(), rp.tris.size() * sizeof(DevTri), hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(ctx->d_shade, rp.shade.data(), rp.shade.size() * sizeof(DevShade), hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(ctx->d_materials, mats.data(), mats.size() * sizeof(DevMaterial), hipMemcpyHostToDevice));
    ctx->n_nodes = n_nodes;
    ctx->n_tris = n_prims;
    ctx->stack_entries = rp.stack_entries;
    memcpy(ctx->scene_abs, rp.scene_abs, sizeof(ctx->scene_abs));
    ctx->persistent_blocks = 0;
    ctx->have_scene = true;
    return RT_OK;
}

// `keep`: where an rt_group wants the host half's result left, so that its other devices upload the same build
int build_and_upload(rt_context *ctx, bool allow_refit = false, Built *keep = nullptr) {
    Built local;
    Built &b = keep ? *keep : local;
    b = Built();
    int r = host_build(ctx, allow_refit, b);
    return r ? r : device_upload(ctx, b);
}

int setup_accum(rt_context *ctx, uint32_t w, uint32_t h) {
    ctx->width = w;
    ctx->height = h;
    ctx->local_bands = bands_for(h, ctx->shard_index, ctx->shard_count);
    ctx->accum_fresh = true;
    dev_free(ctx->d_aov);
    ctx->aov_elems = 0;
    return RT_OK;
}

int ensure_aov(rt_context *ctx) {
    uint64_t elems = (uint64_t)ctx->local_bands * 8u * ctx->width;
    if (ctx->d_aov && ctx->aov_elems == elems) return RT_OK;
    dev_free(ctx->d_aov);
    RT_HIP(dev_malloc((void **)&ctx->d_aov, (elems ? elems : 1) * sizeof(RtHitId)));
    RT_HIP(hipMemsetAsync(ctx->d_aov, 0xFF, (elems ? elems : 1) * sizeof(RtHitId), ctx->stream));
    ctx->aov_elems = elems;
    return RT_OK;
}

// The work queue's head is a 32-bit counter that every wave bumps ONCE more after the queue has run dry (that is how it learns
// that it has): the slots of a launch must stay this far below 2^32 or the head would wrap and hand slots out again
// (8192 resident waves x four times the largest queue grain, 1024 slots).
constexpr uint64_t kQueueIndexMax = 0xFFFFFFFFull - (1ull << 26);

// frames a multi-frame launch may take: 32-bit work-queue index, staging budget, 16-bit frame field of the pixel state
uint32_t frame_group_max(const rt_context *ctx) {
    const uint64_t frame_slots = (uint64_t)((ctx->width + 7u) / 8u) * ctx->local_bands * 64u;
    if (!ctx->opt_pipeline || ctx->opt_kernel != 2 || ctx->opt_aov || ctx->width >= 65536u || frame_slots == 0) return 1;
    const uint64_t by_queue = kQueueIndexMax / frame_slots;
    const uint64_t by_stage = ctx->stage_budget / (frame_slots * sizeof(float4));
    const uint64_t cap = by_queue < by_stage ? by_queue : by_stage;
    return cap < 1 ? 1u : (uint32_t)(cap < 65535u ? cap : 65535u);
}

// staging planes for launches of up to n_frames frames (grows, never shrinks)
int ensure_stage(rt_context *ctx, uint32_t n_frames) {
    const uint64_t stride = local_rows_for(ctx->height, ctx->shard_index, ctx->shard_count) * ctx->width;
    const uint64_t need = (stride ? stride : 1) * n_frames;
    if (ctx->stage_elems >= need) return RT_OK;
    RT_HIP(hipStreamSynchronize(ctx->stream));
    dev_free(ctx->d_stage);
    ctx->stage_elems = 0;
    RT_HIP(dev_malloc((void **)&ctx->d_stage, need * sizeof(float4)));
    ctx->stage_elems = need;
    return RT_OK;
}

// static_scratch: the queue head and the staging planes live in this device's `scratch` array (rt_dispatch_frames_static,
// which has checked that they fit); nothing is allocated on that path.
// the kernel arguments of a launch of n_frames frames starting at uniform u
int make_params(rt_context *ctx, const RtUniform &u, uint32_t n_frames, unsigned char *static_scratch, RenderParams &P) {
    memset(&P, 0, sizeof P);
    P.n_frames = n_frames;
    P.local_rows = (uint32_t)local_rows_for(ctx->height, ctx->shard_index, ctx->shard_count);
    if (n_frames > 1) {
        if (static_scratch) {
            P.stage = reinterpret_cast<float4 *>(static_scratch + RT_STATIC_HEADER_BYTES);
        } else {
            int r = ensure_stage(ctx, n_frames);
            if (r) return r;
            P.stage = ctx->d_stage;
        }
        P.stage_stride = (uint64_t)P.local_rows * ctx->width;
    }
    P.node_q8 = ctx->node_q8 ? 1u : 0u;
    P.node_wide = ctx->node_wide ? (ctx->node_compact ? 2u : 1u) : 0u;
    for (int k = 0; k < 3; k++) {
        P.grid_lo[k] = ctx->grid_lo[k];
        const float c = ctx->grid_cell[k] * 0.00390625f;         // cell * 2^-8: the kernel adds the node's 4-bit exponent to its bit pattern
        memcpy(&P.grid_cell_bits[k], &c, 4);
    }
    P.nodes = ctx->d_nodes; P.tris = ctx->d_tris; P.shade = ctx->d_shade; P.materials = ctx->d_materials;
    P.textures = ctx->d_textures; P.texels = ctx->d_texels; P.n_textures = ctx->n_textures;
    P.n_materials = (uint32_t)ctx->materials_capacity;      // records actually uploaded (the LDS preload copies that many)
    P.accum = ctx->d_accum;
    P.aov = ctx->opt_aov ? ctx->d_aov : nullptr;
    P.counters = ctx->d_counters;
    P.width = u.window[0]; P.height = u.window[1];
    P.tiles_x = (P.width + 7u) / 8u;
    P.inv_tiles_x = P.tiles_x ? 1.0f / (float)P.tiles_x : 0.0f;
    P.local_bands = ctx->local_bands;
    P.shard_index = ctx->shard_index; P.shard_count = ctx->shard_count;
    P.frame = u.frame; P.accumulation_frame = u.accumulation_frame; P.flags = u.flags;
    P.max_bounces = u.max_bounces; P.samples = u.samples;
    P.stack_entries = ctx->stack_entries;
    P.front_face_cw = ctx->opt_cw ? 1u : 0u;
    P.sampler = ctx->opt_sampler;
    P.exposure = u.exposure; P.environment = u.environment;
    memcpy(P.cam_pos, u.camera.position, 12);
    // camera basis: camera_direction() + ray_direction(), shaders/ray.wgsl:1-10,51-60.
    // Uniform per dispatch, so evaluated once here with the same RTN-1 arithmetic.
    float sy, cy, sp, cp;
    rt_sincos(u.camera.yaw, &sy, &cy);
    rt_sincos(u.camera.pitch, &sp, &cp);
    rt_v3 forward = rt_normalize(v3(cy * cp, sp, sy * cp));
    rt_v3 right = rt_normalize(rt_cross(v3(0.0f, 1.0f, 0.0f), forward));
    rt_v3 up = rt_normalize(rt_cross(forward, right));
    float sh, ch;
    rt_sincos(u.camera.fov * 0.5f, &sh, &ch);
    P.fov_scale = sh / ch;
    memcpy(P.scene_abs, ctx->scene_abs, sizeof(P.scene_abs));
    P.aspect = u.camera.aspect;
    P.forward[0] = forward.x; P.forward[1] = forward.y; P.forward[2] = forward.z;
    P.right[0] = right.x; P.right[1] = right.y; P.right[2] = right.z;
    P.up[0] = up.x; P.up[1] = up.y; P.up[2] = up.z;
    P.n_node_chunks = (uint32_t)(ctx->node_bytes / 16u);
    P.wide_stride = ctx->node_stride;
    P.n_slots = ctx->n_tris;
    P.queue = static_scratch ? reinterpret_cast<DevQueue *>(static_scratch) : ctx->d_queue;
    memcpy(P.tune, ctx->tune, sizeof P.tune);
    {
        // Scheduler weights (which phase a wave runs next = the one with the most waiting lanes x weight).  Scenes whose nodes stay in the
        // caches: 3 / 4 / 2, a SHADE step is dear and runs best on full waves.  Scenes past the size at which the node format
        // switches too (host_build): 2 / 4 / 6 — there a node step is a trip to L2 or HBM, and rays that sit waiting to be shaded
        // are round trips not in flight (C4 91.0 -> 86.2, C5q 53.2 -> 49.8, C5 472 -> 436 ms per step; C3: 15.28 -> 15.9 the other way).
        const bool far_nodes = (uint64_t)ctx->n_tris * sizeof(DevNode) / 2 >= (12ull << 20);
        const int32_t by_size[3] = {far_nodes ? 2 : 3, 4, far_nodes ? 6 : 2};
        for (int k = 0; k < 3; k++) if (P.tune[k] <= 0) P.tune[k] = by_size[k];
    }
    // queue grain: 64 slots (one 8x8 tile) keeps the end of a launch short, which is what counts for a shard of
    // the image; a whole 4K image takes two tiles per claim (measured: 16.8 -> 16.5 ms/step at 8.3 M slots per
    // frame, but 8.46 -> 8.65 ms/step at 4.1 M, so the switch sits between the two)
    P.chunk = ctx->chunk ? ctx->chunk : ((uint64_t)P.tiles_x * P.local_bands * 64u >= (6u << 20) ? 128u : 64u);
    return RT_OK;
}

// which kernel instance a scene needs: 0 = triangles, untextured; 1 = + sphere records; 2 = + textures
int kernel_ext(const rt_context *ctx) { return ctx->n_textures > 0 ? 2 : (ctx->hs->spheres.empty() ? 0 : 1); }

// band_begin / band_end: rt_dispatch_bands — only the work items of those local bands (one frame, queue-fed kernels)
int launch(rt_context *ctx, const RtUniform &u, uint32_t n_frames = 1, unsigned char *static_scratch = nullptr, uint32_t band_begin = 0,
           uint32_t band_end = 0xFFFFFFFFu) {
    RenderParams P;
    { int r = make_params(ctx, u, n_frames, static_scratch, P); if (r) return r; }
    const bool last_piece = band_end >= P.local_bands;
    if (band_end < P.local_bands) P.local_bands = band_end;     // the queue is band-major: the kernel sees an image that ends there ...
    if (ctx->opt_kernel != 1) {
        // ... and a queue whose head starts at the first slot of band_begin (zero for a whole frame), whatever the last call left
        RT_HIP(hipMemsetAsync(P.queue, 0, sizeof(DevQueue), ctx->stream));
        if (band_begin) RT_HIP(hipMemsetD32Async((hipDeviceptr_t)&P.queue->head, (int)(band_begin * P.tiles_x * 64u), 1, ctx->stream));
        const int ext = kernel_ext(ctx);
        const bool ldsm = P.n_materials <= RT_LDS_MATERIALS;
        const int sel = (ctx->node_compact ? 64 : 0) | (ctx->node_wide ? 32 : 0) | (ctx->node_q8 ? 16 : 0) | (ldsm ? 8 : 0) | (ctx->opt_count ? 4 : 0) | (ctx->opt_aov ? 2 : 0) | (ext << 8);
        if (ctx->persistent_blocks == 0 || ctx->persistent_sel != sel) {
            ctx->persistent_blocks = rt_persistent_blocks(ctx->device, ctx->stack_entries, ctx->opt_count, ctx->opt_aov, ext, ldsm, ctx->node_wide ? (ctx->node_compact ? 3 : 2) : (ctx->node_q8 ? 1 : 0));
            ctx->persistent_sel = sel;
            // Resident workgroups per CU.  More waves hide more latency only while their rays' working set stays in the caches and the
            // TLB; measured (64-byte 8-ary nodes, ms per step in 4-step launches at 3 / 4 / 5 / 6 workgroups): C4, 2 GB of records:
            // - / 92.4 / 83.1 / 78.4; C5q, 9.4 GB: 47.3 / 40.0 / 38.3 / 41.6; C5, 38 GB: 401 / 354 / 437 / -.  Hence by size, unless knob 103
            // says otherwise.  (Round 2 measured this with the slice rays of slab_pad's comment still in every launch's tail and chose
            // 4 and 3; each step down then hid part of that tail.)
            const uint64_t record_bytes = ctx->node_bytes + ctx->n_tris * (sizeof(DevTri) + sizeof(DevShade));
            int per_cu = ctx->tune[3];
            if (per_cu <= 0) per_cu = record_bytes > (20ull << 30) ? 4 : (record_bytes > (4ull << 30) ? 5 : 0);
            if (per_cu > 0 && ctx->cu_count > 0 && ctx->persistent_blocks > per_cu * ctx->cu_count)
                ctx->persistent_blocks = per_cu * ctx->cu_count;
        }
    }
#ifdef RT_TAIL_PROBE
    const size_t probe_waves = (size_t)(ctx->persistent_blocks > 0 ? ctx->persistent_blocks : 0) * 4;
    if (ctx->opt_kernel == 2 && n_frames > 1 && probe_waves && !static_scratch) {
        RT_HIP(dev_malloc((void **)&P.probe, probe_waves * 24));
        RT_HIP(hipMemsetAsync(P.probe, 0, probe_waves * 24, ctx->stream));
    }
#endif
    RT_HIP(rt_launch_render(P, ctx->opt_kernel, ctx->opt_count, ctx->opt_aov, kernel_ext(ctx), ctx->persistent_blocks, ctx->stream));
#ifdef RT_TAIL_PROBE
    if (P.probe) {
        std::vector<unsigned long long> pr(3 * probe_waves), ends;
        RT_HIP(hipStreamSynchronize(ctx->stream));
        RT_HIP(hipMemcpy(pr.data(), P.probe, pr.size() * 8, hipMemcpyDeviceToHost));
        (void)hipFree(P.probe);
        unsigned long long t0 = ~0ull, d0 = ~0ull;
        for (size_t w = 0; w < probe_waves; w++) {
            if (!pr[3 * w]) continue;                     // wave not launched (grid capped by the amount of work)
            t0 = std::min(t0, pr[3 * w]);
            if (pr[3 * w + 1]) d0 = std::min(d0, pr[3 * w + 1]);
            ends.push_back(pr[3 * w + 2]);
        }
        if (!ends.empty()) {
            std::sort(ends.begin(), ends.end());
            auto ms = [&](unsigned long long t) { return (double)(t - t0) * 1e-5; };
            auto q = [&](double f) { return ms(ends[(size_t)(f * (double)(ends.size() - 1))]); };
            fprintf(stderr, "tail probe: %zu waves, queue empty at %.3f ms; waves exit: 10%% %.3f, 50%% %.3f, 90%% %.3f, 99%% %.3f, last %.3f ms\n",
                    ends.size(), ms(d0), q(0.10), q(0.50), q(0.90), q(0.99), q(1.0));
        }
    }
#endif
    if (last_piece) ctx->dispatches += n_frames;
    ctx->launches++;
    return RT_OK;
}

int check_dispatch(rt_context *ctx, const RtUniform *u) {
    if (!ctx || !u) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (!ctx->have_scene) return fail(RT_ERR_NO_SCENE, "dispatch before rt_scene_upload");
    if (!ctx->d_accum) return fail(RT_ERR_NO_ACCUM, "dispatch before rt_accum_resize/rt_accum_bind");
    if (u->window[0] != ctx->width || u->window[1] != ctx->height)
        return fail(RT_ERR_INVALID_ARGUMENT, "uniform.window %ux%u does not match the accumulation buffer %ux%u",
                    u->window[0], u->window[1], ctx->width, ctx->height);
    if ((uint64_t)u->window[0] * u->window[1] > 0xFFFFFFFFull)
        return fail(RT_ERR_INVALID_ARGUMENT, "window has more than 2^32 pixels");
    if ((uint64_t)((u->window[0] + 7u) / 8u) * ctx->local_bands * 64u > kQueueIndexMax)       // pixel slots of one frame (whole 8x8 tiles)
        return fail(RT_ERR_INVALID_ARGUMENT, "window has more pixel slots than the 32-bit work queue of a launch can index");
    if ((uint64_t)ctx->stack_entries * 256u * 4u + RT_LDS_MATERIALS * 64u + 2048u > 160u * 1024u)      // traversal stacks + the material copy + the 8-ary nodes' child-order table
        return fail(RT_ERR_INVALID_ARGUMENT, "BVH depth %u needs more LDS than a CU has", ctx->stack_entries);
    RT_HIP(hipSetDevice(ctx->device));
    if (ctx->opt_aov) { int r = ensure_aov(ctx); if (r) return r; }
    return RT_OK;
}

// rt_models_update on the device (RT_OPT_REFIT = 2): new triangle records and node boxes without leaving the GPU.
int device_refit(rt_context *ctx, Built *keep = nullptr) {
    auto t0 = std::chrono::steady_clock::now();
    RT_HIP(hipSetDevice(ctx->device));
    ctx->models_committed = true;
    std::vector<DevMaterial> mats;
    convert_materials(*ctx, mats);
    RT_HIP(hipMemcpyAsync(ctx->d_materials, mats.data(), mats.size() * sizeof(DevMaterial), hipMemcpyHostToDevice, ctx->stream));
    const float *xform = nullptr;
    std::vector<float> resolved;
    if (!ctx->hs->transforms.empty()) {
        resolved.resize(12 * ctx->hs->primitives.size());
        for (size_t g = 0; g < ctx->hs->primitives.size(); g++)      // src/scene.rs:141-147: a geometry names its transform by offset
            memcpy(&resolved[12 * g], &ctx->hs->transforms[12 * ctx->hs->primitives[g].transformation_offset], 48);
        RT_HIP(hipMemcpyAsync(ctx->d_xform, resolved.data(), resolved.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        xform = ctx->d_xform;
    }
    const float prim_pad = rt_prim_pad(ctx->prim_pad_scale);
    const float grid6[6] = {ctx->grid_lo[0], ctx->grid_lo[1], ctx->grid_lo[2], ctx->grid_cell[0], ctx->grid_cell[1], ctx->grid_cell[2]};
    if (ctx->node_wide)
        RT_HIP(rt_launch_refit8(ctx->d_tris, ctx->d_shade, xform, ctx->d_tribox, ctx->n_tris, reinterpret_cast<float4 *>(ctx->d_nodes), ctx->node_stride,
                                ctx->d_parent, ctx->d_arrived, ctx->d_cbox, (uint32_t)ctx->n_nodes, ctx->d_root, prim_pad,
                                ctx->node_compact ? grid6 : nullptr, ctx->stream));
    else
    RT_HIP(rt_launch_refit(ctx->d_tris, ctx->d_shade, xform, ctx->d_tribox, ctx->n_tris, reinterpret_cast<float4 *>(ctx->d_nodes), ctx->node_stride,
                           ctx->d_parent, ctx->d_arrived, ctx->d_cbox, (uint32_t)ctx->n_nodes, ctx->d_root, ctx->node_q8,
                           prim_pad, ctx->stream));
    RtRefitBox root;
    RT_HIP(hipMemcpyAsync(&root, ctx->d_root, sizeof root, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));                   // also keeps `mats` / `resolved` alive long enough
    for (int k = 0; k < 3; k++) {
        const float m = fmaxf(fabsf(rt_pad_lo(root.lo[k])), fabsf(rt_pad_hi(root.hi[k])));
        ctx->scene_abs[k] = m <= 3.0e38f ? m : 3.0e38f;
    }
    ctx->refits++;
    ctx->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (ctx->node_q8 || ctx->node_wide) {
        // the moved scene may have outgrown what 8-bit offsets can span (their outward rounding saturates there):
        // rebuild on the host, which chooses the node format again
        if (!rt_q8_fits(root.lo, root.hi)) return build_and_upload(ctx, true, keep);
        // compact 8-ary nodes: their corners are cells of the grid laid out at the last host build; a scene that moved out of it is rebuilt
        if (ctx->node_compact && !rt_grid_holds(root.lo, root.hi, ctx->grid_lo, ctx->grid_cell)) return build_and_upload(ctx, true, keep);
    }
    return RT_OK;
}

}  // namespace

extern "C" {

const char *rt_last_error(void) { return g_last_error.c_str(); }

int rt_create(int device_id, rt_context **out) {
    if (!out) return fail(RT_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(RT_ERR_NO_DEVICE, "no HIP device (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(RT_ERR_INVALID_ARGUMENT, "device %d out of range (0..%d)", device_id, n - 1);
    RT_HIP(hipSetDevice(device_id));
    if (device_id < RT_MAX_DEVICES) {
        // this device's copy of the static scratch array (rt_dispatch_frames_static) and the event that orders its users;
        // the device is current here
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[device_id]) {
            void *p = nullptr;
            hipEvent_t ev = nullptr;
            RT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
            RT_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            g_scratch_done[device_id] = ev;
            g_scratch_ptr[device_id] = p;
        }
    }
    rt_context *ctx = new rt_context();
    ctx->device = device_id;
    (void)hipDeviceGetAttribute(&ctx->cu_count, hipDeviceAttributeMultiprocessorCount, device_id);
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        dev_malloc((void **)&ctx->d_counters, sizeof(DevCounters)) != hipSuccess ||
        dev_malloc((void **)&ctx->d_queue, sizeof(DevQueue)) != hipSuccess ||
        hipMemset(ctx->d_counters, 0, sizeof(DevCounters)) != hipSuccess) {
        rt_destroy(ctx);
        return fail(RT_ERR_HIP, "context set-up failed: %s", hipGetErrorString(hipGetLastError()));
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return RT_OK;
}

int rt_destroy(rt_context *ctx) {
    if (!ctx) return RT_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    dev_free(ctx->d_nodes); dev_free(ctx->d_tris); dev_free(ctx->d_shade); dev_free(ctx->d_materials);
    if (ctx->accum_owned) dev_free(ctx->d_accum);
    dev_free(ctx->d_rgb);
    dev_free(ctx->d_aov); dev_free(ctx->d_counters); dev_free(ctx->d_queue); dev_free(ctx->d_stage);
    dev_free(ctx->d_parent); dev_free(ctx->d_arrived); dev_free(ctx->d_cbox); dev_free(ctx->d_tribox); dev_free(ctx->d_root); dev_free(ctx->d_xform);
    dev_free(ctx->d_textures); dev_free(ctx->d_texels);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return RT_OK;
}

int rt_set_stream(rt_context *ctx, void *hip_stream) {
    if (!ctx) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is null");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return RT_OK;
}

int rt_set_option(rt_context *ctx, int key, int64_t value) {
    if (!ctx) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is null");
    switch (key) {
        case RT_OPT_COUNT_TRAVERSAL: ctx->opt_count = value != 0; return RT_OK;
        case RT_OPT_AOV: ctx->opt_aov = value != 0; return RT_OK;
        case RT_OPT_KERNEL:
            if (value < 0 || value > 2) return fail(RT_ERR_INVALID_ARGUMENT, "RT_OPT_KERNEL: 0, 1 or 2");
            ctx->opt_kernel = (int)value;
            return RT_OK;
        case RT_OPT_FRONT_FACE_CW: ctx->opt_cw = value != 0; return RT_OK;
        case RT_OPT_SAMPLER:
            if (value < 0 || value > 3) return fail(RT_ERR_INVALID_ARGUMENT, "RT_OPT_SAMPLER: 0..3 (bit 0 linear, bit 1 repeat)");
            ctx->opt_sampler = (uint32_t)value; return RT_OK;
        case RT_OPT_REFIT:
            if (value < 0 || value > 2) return fail(RT_ERR_INVALID_ARGUMENT, "RT_OPT_REFIT: 0, 1 or 2");
            ctx->opt_refit = (int)value;
            return RT_OK;
        case RT_OPT_FRAME_PIPELINE: ctx->opt_pipeline = value != 0; return RT_OK;
        case RT_OPT_NODE_FORMAT:
            if (value < 0 || value > 4) return fail(RT_ERR_INVALID_ARGUMENT, "RT_OPT_NODE_FORMAT: 0 (automatic), 1 (48-byte binary), 2 (32-byte binary), 3 (80-byte 8-ary) or 4 (64-byte 8-ary)");
            ctx->opt_node_format = (int)value;
            return RT_OK;
        case RT_OPT_NODE_STRIDE:
            if (value != 0 && value != 3 && value != 4 && value != 8) return fail(RT_ERR_INVALID_ARGUMENT, "RT_OPT_NODE_STRIDE: 0 (auto), 3 or 4 (binary nodes), 8 (8-ary nodes)");
            ctx->opt_node_stride = value;
            return RT_OK;
        default:
            // developer knobs (tools/ab_bench.py, tools/shard_scaling.py): 100..107 scheduler weights (103: resident workgroups per CU, 0 = all that fit) and builder leaf size /
            // bins, 108 queue grain, 109 staging budget in KiB, 110 SAH traversal cost x16, 111 BVH depth cap, 112 greedy 8-ary collapse
            if (key == 106 && (value < 0 || value > 7)) return fail(RT_ERR_INVALID_ARGUMENT, "option 106 (triangles per leaf): 0..7");
            // a phase weight of 0 would let the scheduler pick a phase nobody waits in, for ever
            if ((key == 100 || key == 101 || key == 102 || key == 104) && (value < 1 || value > 65536))
                return fail(RT_ERR_INVALID_ARGUMENT, "option %d (scheduler weight): 1..65536", key);
            if (key >= 100 && key < 108) {
                ctx->tune[key - 100] = (int32_t)value;
                if (key == 103) ctx->persistent_blocks = 0;     // grid size is recomputed at the next launch
                return RT_OK;
            }
            if (key == 108 && value >= 0 && value <= 1024) { ctx->chunk = (uint32_t)value; return RT_OK; }
            if (key == 111 && value >= 0 && value <= 48) { ctx->bvh_max_depth = (uint32_t)value; return RT_OK; }
            if (key == 112 && (value == 0 || value == 1)) { ctx->collapse_greedy = value == 1; return RT_OK; }
            if (key == 110 && value >= 0) { ctx->sah_traversal_cost = (float)value / 16.0f; return RT_OK; }
            if (key == 109 && value >= 1) { ctx->stage_budget = (uint64_t)value << 10; return RT_OK; }   // KiB
            return fail(RT_ERR_INVALID_ARGUMENT, "unknown option %d", key);
    }
}

int rt_scene_upload(rt_context *ctx, const RtVertex *vertices, uint64_t n_vertices, const uint32_t *indices,
                    uint64_t n_indices, const RtGeometryPrimitive *primitives, uint32_t n_primitives,
                    const float *transforms, const RtGpuModel *models, uint32_t n_models) {
    if (!ctx) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is null");
    if ((n_vertices && !vertices) || (n_indices && !indices) || (n_primitives && !primitives) || (n_models && !models))
        return fail(RT_ERR_INVALID_ARGUMENT, "null array with non-zero count");
    for (uint32_t g = 0; g < n_primitives; g++)
        if (transforms && primitives[g].transformation_offset >= n_primitives)
            return fail(RT_ERR_INVALID_ARGUMENT, "geometry %u: transformation_offset out of range", g);
    ctx->hs->vertices.assign(vertices, vertices + n_vertices);
    ctx->hs->indices.assign(indices, indices + n_indices);
    ctx->hs->primitives.assign(primitives, primitives + n_primitives);
    ctx->hs->models.assign(models, models + n_models);
    if (transforms) ctx->hs->transforms.assign(transforms, transforms + 12ull * n_primitives);
    else ctx->hs->transforms.clear();
    ctx->hs->spheres.clear();
    ctx->n_textures = 0;
    ctx->have_scene = false;
    return build_and_upload(ctx);
}

int rt_spheres_upload(rt_context *ctx, const RtSphere *spheres, uint64_t n_spheres) {
    if (!ctx || (n_spheres && !spheres)) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (!ctx->have_scene) return fail(RT_ERR_NO_SCENE, "rt_spheres_upload before rt_scene_upload");
    std::vector<RtSphere> old(spheres, spheres + n_spheres);
    old.swap(ctx->hs->spheres);
    const int r = build_and_upload(ctx);
    if (r != RT_OK && !ctx->models_committed) old.swap(ctx->hs->spheres);    // refused before anything reached the device: the old spheres stay in force
    return r;
}

int rt_textures_upload(rt_context *ctx, const RtTexture *textures, uint32_t n_textures) {
    if (!ctx || (n_textures && !textures)) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (!ctx->have_scene) return fail(RT_ERR_NO_SCENE, "rt_textures_upload before rt_scene_upload");
    std::vector<DevTexture> table(n_textures ? n_textures : 1);
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_textures; i++) {
        if (!textures[i].rgba8 || textures[i].width == 0 || textures[i].height == 0)
            return fail(RT_ERR_INVALID_ARGUMENT, "texture %u is empty", i);
        if (total + (uint64_t)textures[i].width * textures[i].height > 0xFFFFFFFFull)
            return fail(RT_ERR_INVALID_ARGUMENT, "more than 2^32 texels in total");
        table[i].offset = (uint32_t)total;
        table[i].width = textures[i].width;
        table[i].height = textures[i].height;
        table[i].pad = 0;
        total += (uint64_t)textures[i].width * textures[i].height;
    }
    std::vector<uint32_t> texels(total ? total : 1);
    for (uint32_t i = 0; i < n_textures; i++) {
        const uint32_t w = textures[i].width, h = textures[i].height;
        for (uint32_t y = 0; y < h; y++)   // imageops::flip_vertical, src/scene.rs:77
            memcpy(&texels[table[i].offset + (size_t)y * w], textures[i].rgba8 + 4 * (size_t)(h - 1 - y) * w, 4 * (size_t)w);
    }
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    dev_free(ctx->d_textures); dev_free(ctx->d_texels);
    RT_HIP(dev_malloc((void **)&ctx->d_textures, table.size() * sizeof(DevTexture)));
    RT_HIP(dev_malloc((void **)&ctx->d_texels, texels.size() * sizeof(uint32_t)));
    RT_HIP(hipMemcpy(ctx->d_textures, table.data(), table.size() * sizeof(DevTexture), hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(ctx->d_texels, texels.data(), texels.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    ctx->n_textures = n_textures;
    return RT_OK;
}

}  // extern "C"

// rt_models_update; `keep` (rt_group_models_update): receives the host build when the update needed one
static int models_update_impl(rt_context *ctx, const RtGpuModel *models, uint32_t n_models, const float *transforms, Built *keep) {
    if (!ctx || (n_models && !models)) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (!ctx->have_scene) return fail(RT_ERR_NO_SCENE, "rt_models_update before rt_scene_upload");
    if (n_models < ctx->hs->primitives.size())
        return fail(RT_ERR_INVALID_ARGUMENT, "scene: %u models for %zu geometries", n_models, ctx->hs->primitives.size());
    if (transforms)
        for (size_t g = 0; g < ctx->hs->primitives.size(); g++)
            if (ctx->hs->primitives[g].transformation_offset >= ctx->hs->primitives.size())
                return fail(RT_ERR_INVALID_ARGUMENT, "geometry %zu: transformation_offset out of range", g);
    // The device refit re-derives triangles from the shading records baked at upload, which took their vertices from the
    // models' vertex_start / index_start (main.wgsl:86-91): it is only valid while those ranges are what was uploaded.
    bool ranges_kept = n_models >= ctx->hs->models.size();
    for (size_t g = 0; g < ctx->hs->models.size() && ranges_kept; g++)
        ranges_kept = models[g].vertex_start == ctx->hs->models[g].vertex_start && models[g].index_start == ctx->hs->models[g].index_start;
    bool spheres_ok = true;
    for (const RtSphere &sp : ctx->hs->spheres) spheres_ok = spheres_ok && sp.model < n_models;
    // commit the new models / transforms only once the path taken has accepted them
    std::vector<RtGpuModel> old_models(models, models + n_models);
    std::vector<float> old_transforms;
    old_models.swap(ctx->hs->models);
    if (transforms) {
        old_transforms.assign(transforms, transforms + 12ull * ctx->hs->primitives.size());
        old_transforms.swap(ctx->hs->transforms);
    }
    int r;
    // device refit: same primitives, same tree, same model ranges, the sphere -> model references still valid and the
    // material array as uploaded has room for the new models
    ctx->last_update_on_device = false;
    if (ctx->opt_refit == 2 && ctx->d_parent && (ctx->n_nodes > 1 || ctx->node_wide) && ranges_kept && spheres_ok && n_models <= ctx->materials_capacity) {
        const uint64_t rebuilds_before = ctx->rebuilds, refits_before = ctx->refits;
        r = device_refit(ctx, keep);
        ctx->last_update_on_device = r == RT_OK && ctx->rebuilds == rebuilds_before && ctx->refits == refits_before + 1;
    } else r = build_and_upload(ctx, /*allow_refit=*/true, keep);
    if (r != RT_OK && !ctx->models_committed) {       // rejected before anything reached the device: the old scene stays in force
        old_models.swap(ctx->hs->models);
        if (transforms) old_transforms.swap(ctx->hs->transforms);
    }
    return r;
}

extern "C" {

int rt_models_update(rt_context *ctx, const RtGpuModel *models, uint32_t n_models, const float *transforms) {
    return models_update_impl(ctx, models, n_models, transforms, nullptr);
}

int rt_set_shard(rt_context *ctx, uint32_t shard_index, uint32_t shard_count) {
    if (!ctx) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (shard_count == 0 || shard_index >= shard_count) return fail(RT_ERR_INVALID_ARGUMENT, "shard %u of %u", shard_index, shard_count);
    if (ctx->d_accum && (shard_index != ctx->shard_index || shard_count != ctx->shard_count)) {
        // the accumulation buffer was laid out for the old shard: drop it (a bound buffer is only un-bound)
        RT_HIP(hipSetDevice(ctx->device));
        RT_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->accum_owned) dev_free(ctx->d_accum);
        ctx->d_accum = nullptr;
        ctx->accum_owned = false;
        ctx->width = ctx->height = 0;
    }
    ctx->shard_index = shard_index;
    ctx->shard_count = shard_count;
    return RT_OK;
}

uint64_t rt_accum_local_rows(const rt_context *ctx, uint32_t w, uint32_t h) {
    (void)w;
    if (!ctx) return 0;
    return local_rows_for(h, ctx->shard_index, ctx->shard_count);
}

int rt_accum_resize(rt_context *ctx, uint32_t w, uint32_t h) {
    if (!ctx) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is null");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->accum_owned) dev_free(ctx->d_accum);
    ctx->d_accum = nullptr;
    // whole bands are allocated so a partial last band needs no special case
    uint64_t bytes = (uint64_t)bands_for(h, ctx->shard_index, ctx->shard_count) * 8u * w * RT_ACCUM_STRIDE_BYTES;
    RT_HIP(dev_malloc((void **)&ctx->d_accum, bytes ? bytes : 16));
    RT_HIP(hipMemsetAsync(ctx->d_accum, 0, bytes ? bytes : 16, ctx->stream));   // zeros, src/app.rs:90-92
    ctx->accum_owned = true;
    return setup_accum(ctx, w, h);
}

int rt_accum_bind(rt_context *ctx, void *device_ptr, uint64_t bytes, uint32_t w, uint32_t h) {
    if (!ctx || !device_ptr) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    uint64_t need = local_rows_for(h, ctx->shard_index, ctx->shard_count) * w * RT_ACCUM_STRIDE_BYTES;
    if (bytes < need) return fail(RT_ERR_INVALID_ARGUMENT, "accumulation buffer has %llu bytes, needs %llu",
                                  (unsigned long long)bytes, (unsigned long long)need);
    if (((uintptr_t)device_ptr & 15u) != 0) return fail(RT_ERR_INVALID_ARGUMENT, "accumulation buffer must be 16-byte aligned");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->accum_owned) dev_free(ctx->d_accum);
    ctx->d_accum = (float *)device_ptr;
    ctx->accum_owned = false;
    return setup_accum(ctx, w, h);
}

int rt_frames_reserve(rt_context *ctx, uint32_t n_frames) {
    if (!ctx) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (!ctx->d_accum) return fail(RT_ERR_NO_ACCUM, "rt_frames_reserve before rt_accum_resize/rt_accum_bind");
    RT_HIP(hipSetDevice(ctx->device));
    const uint32_t group = n_frames < frame_group_max(ctx) ? n_frames : frame_group_max(ctx);
    return group > 1 ? ensure_stage(ctx, group) : RT_OK;
}

int rt_dispatch(rt_context *ctx, const RtUniform *uniform) {
    int r = check_dispatch(ctx, uniform);
    if (r) return r;
    RT_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    r = launch(ctx, *uniform);
    if (r) return r;
    RT_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->timed = true;
    ctx->accum_fresh = false;
    return RT_OK;
}

int rt_dispatch_bands(rt_context *ctx, const RtUniform *uniform, uint32_t band_begin, uint32_t band_end) {
    int r = check_dispatch(ctx, uniform);
    if (r) return r;
    if (ctx->opt_kernel == 1) return fail(RT_ERR_INVALID_ARGUMENT, "rt_dispatch_bands needs a queue-fed kernel (RT_OPT_KERNEL 2 or 0)");
    if (band_begin > band_end) return fail(RT_ERR_INVALID_ARGUMENT, "rt_dispatch_bands: band range [%u, %u) of %u bands", band_begin, band_end, ctx->local_bands);
    if (band_end > ctx->local_bands) band_end = ctx->local_bands;       // "to the end"; a range that starts beyond it is empty
    if (band_begin > band_end) band_begin = band_end;
    RT_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    if (band_begin < band_end) {
        r = launch(ctx, *uniform, 1, nullptr, band_begin, band_end);
        if (r) return r;
    }
    ctx->accum_fresh = false;       // like rt_dispatch, whatever the range held (a shard of a group may own none of it): the cold start of src/app.rs:87-88 is over
    RT_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->timed = true;
    return RT_OK;
}

int rt_dispatch_frames(rt_context *ctx, RtUniform *uniform, uint32_t n_frames) {
    int r = check_dispatch(ctx, uniform);
    if (r) return r;
    RT_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    // The phased kernel takes a run of frames in ONE launch (frame-major work queue, colours staged per frame and
    // blended in frame order): the slow tail of a frame, where the last long pixels run on a nearly empty GPU,
    // overlaps the next frame.  If the staging planes cannot be allocated the run falls back to shorter launches.
    uint32_t group_max = frame_group_max(ctx);
    if (group_max > n_frames) group_max = n_frames;
    while (group_max > 1 && ensure_stage(ctx, group_max) != RT_OK) {
        (void)hipGetLastError();                           // out of memory is not sticky; try half
        group_max /= 2;
    }
    uint32_t i = 0;
    while (i < n_frames) {
        const uint32_t group = (n_frames - i < group_max) ? n_frames - i : group_max;
        uniform->accumulation_frame += 1;                  // src/app.rs:78
        if (ctx->accum_fresh) {                            // src/app.rs:87-88
            uniform->accumulation_frame = 0;
            ctx->accum_fresh = false;
        }
        uniform->frame += 1;                               // src/app.rs:96
        r = launch(ctx, *uniform, group);
        if (r) return r;
        uniform->accumulation_frame += group - 1;          // the launch covered frames frame .. frame + group - 1
        uniform->frame += group - 1;
        i += group;
    }
    RT_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->timed = true;
    return RT_OK;
}

int rt_dispatch_frames_static(rt_context *ctx, RtUniform *uniform, uint32_t n_frames) {
    if (!ctx || !uniform) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    // everything check_dispatch checks, minus the one thing there that allocates (the AOV buffer)
    if (ctx->opt_aov) return fail(RT_ERR_INVALID_ARGUMENT, "rt_dispatch_frames_static: RT_OPT_AOV needs a buffer this call may not allocate");
    int r = check_dispatch(ctx, uniform);
    if (r) return r;
    unsigned char *sc = ctx->device < RT_MAX_DEVICES ? static_cast<unsigned char *>(g_scratch_ptr[ctx->device]) : nullptr;
    if (!sc) return fail(RT_ERR_HIP, "rt_dispatch_frames_static: no static scratch on device %d", ctx->device);
    if (n_frames == 0) return RT_OK;
    // One launch for the whole run when the kernel can take it (as rt_dispatch_frames), with the staging planes in the
    // static array: n_frames x local pixels x 16 bytes must fit.  Other kernel variants / RT_OPT_FRAME_PIPELINE = 0
    // render frame by frame and stage nothing.
    const uint64_t frame_slots = (uint64_t)((ctx->width + 7u) / 8u) * ctx->local_bands * 64u;
    const bool pipelined = ctx->opt_pipeline && ctx->opt_kernel == 2 && ctx->width < 65536u && frame_slots != 0 && n_frames > 1;
    if (pipelined) {
        const uint64_t need = local_rows_for(ctx->height, ctx->shard_index, ctx->shard_count) * ctx->width * sizeof(float4) * n_frames;
        if (need > RT_STATIC_STAGE_BYTES)
            return fail(RT_ERR_INVALID_ARGUMENT, "rt_dispatch_frames_static: %u frames of this window need %llu bytes of staging, the static "
                        "scratch holds %llu (use rt_dispatch_frames, or fewer frames per call)", n_frames, (unsigned long long)need,
                        (unsigned long long)RT_STATIC_STAGE_BYTES);
        if (frame_slots * n_frames > kQueueIndexMax || n_frames > 65535u)
            return fail(RT_ERR_INVALID_ARGUMENT, "rt_dispatch_frames_static: %u frames exceed the 32-bit work queue of one launch", n_frames);
    }
    // Under stream capture (hipStreamBeginCapture on the context's stream) the call is recorded, not run: the timing events and
    // the cross-context ordering event are left out (an event recorded inside a capture cannot be waited on outside it); the
    // owner of the graph orders its launches against other users of this GPU's scratch.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(ctx->stream, &cap);
    const bool capturing = cap != hipStreamCaptureStatusNone;
    // held from the wait on the previous user's event to the record of this call's: two host threads cannot both queue behind the
    // same predecessor and then run side by side (launches are asynchronous, so the lock is held for microseconds)
    std::unique_lock<std::mutex> lock(g_scratch_mutex, std::defer_lock);
    if (!capturing) {
        lock.lock();
        if (g_scratch_used[ctx->device]) RT_HIP(hipStreamWaitEvent(ctx->stream, g_scratch_done[ctx->device], 0));
        RT_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    }
    uint32_t i = 0;
    while (i < n_frames) {
        const uint32_t group = pipelined ? n_frames : 1u;
        uniform->accumulation_frame += 1;                  // src/app.rs:78
        if (ctx->accum_fresh) {                            // src/app.rs:87-88
            uniform->accumulation_frame = 0;
            ctx->accum_fresh = false;
        }
        uniform->frame += 1;                               // src/app.rs:96
        r = launch(ctx, *uniform, group, sc);
        if (r) {
            // whatever part of the run did reach the stream still uses the scratch: the next user must queue behind it
            if (!capturing && hipEventRecord(g_scratch_done[ctx->device], ctx->stream) == hipSuccess) g_scratch_used[ctx->device] = true;
            return r;
        }
        uniform->accumulation_frame += group - 1;
        uniform->frame += group - 1;
        i += group;
    }
    if (!capturing) {
        RT_HIP(hipEventRecord(ctx->ev1, ctx->stream));
        RT_HIP(hipEventRecord(g_scratch_done[ctx->device], ctx->stream));
        g_scratch_used[ctx->device] = true;
        ctx->timed = true;
    }
    return RT_OK;
}

uint64_t rt_debug_alloc_count(void) { return g_device_allocs.load(); }

int rt_sync(rt_context *ctx) {
    if (!ctx) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is null");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_accum_download(rt_context *ctx, float *dst) {
    if (!ctx || !dst) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (!ctx->d_accum) return fail(RT_ERR_NO_ACCUM, "no accumulation buffer");
    RT_HIP(hipSetDevice(ctx->device));
    uint64_t bytes = local_rows_for(ctx->height, ctx->shard_index, ctx->shard_count) * ctx->width * RT_ACCUM_STRIDE_BYTES;
    RT_HIP(hipMemcpyAsync(dst, ctx->d_accum, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_capture_rgb8(rt_context *ctx, uint8_t *dst) {
    if (!ctx || !dst) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (!ctx->d_accum) return fail(RT_ERR_NO_ACCUM, "no accumulation buffer");
    RT_HIP(hipSetDevice(ctx->device));
    const uint64_t n_pixels = local_rows_for(ctx->height, ctx->shard_index, ctx->shard_count) * ctx->width;
    if (n_pixels == 0) return RT_OK;
    if (ctx->rgb_bytes < n_pixels * 3) {                   // kept between captures, grows with the window
        RT_HIP(hipStreamSynchronize(ctx->stream));
        dev_free(ctx->d_rgb);
        ctx->rgb_bytes = 0;
        RT_HIP(dev_malloc((void **)&ctx->d_rgb, n_pixels * 3));
        ctx->rgb_bytes = n_pixels * 3;
    }
    hipError_t e = rt_launch_capture(ctx->d_accum, ctx->d_rgb, n_pixels, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dst, ctx->d_rgb, n_pixels * 3, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "rt_capture_rgb8: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_aov_download(rt_context *ctx, RtHitId *dst) {
    if (!ctx || !dst) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (!ctx->d_aov) return fail(RT_ERR_INVALID_ARGUMENT, "no AOV buffer: set RT_OPT_AOV before dispatching");
    RT_HIP(hipSetDevice(ctx->device));
    uint64_t bytes = local_rows_for(ctx->height, ctx->shard_index, ctx->shard_count) * ctx->width * sizeof(RtHitId);
    RT_HIP(hipMemcpyAsync(dst, ctx->d_aov, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_stats(rt_context *ctx, RtStats *out) {
    if (!ctx || !out) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    RT_HIP(hipSetDevice(ctx->device));
    DevCounters c;
    RT_HIP(hipMemcpyAsync(&c, ctx->d_counters, sizeof c, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    memset(out, 0, sizeof *out);
    out->rays = c.rays;
    out->pixel_frames = c.pixel_frames;
    out->node_visits = c.node_visits;
    out->triangle_tests = c.triangle_tests;
    out->wave_node_iters = c.wave_node_iters;
    out->wave_leaf_iters = c.wave_leaf_iters;
    out->wave_steps = c.wave_steps;
    out->guard_trips = c.guard_trips;
    out->dispatches = ctx->dispatches;
    out->bvh_nodes = ctx->n_nodes;
    out->bvh_bytes = ctx->node_bytes + ctx->n_tris * (sizeof(DevTri) + sizeof(DevShade));
    out->bvh_build_seconds = ctx->build_seconds;
    out->bvh_refits = ctx->refits;
    out->bvh_rebuilds = ctx->rebuilds;
    out->launches = ctx->launches;
    out->stack_entries = ctx->stack_entries;
    out->resident_blocks = (uint32_t)ctx->persistent_blocks;
    out->node_format = !ctx->have_scene ? 0u : (ctx->node_wide ? (ctx->node_compact ? 4u : 3u) : (ctx->node_q8 ? 2u : 1u));
    return RT_OK;
}

int rt_stats_reset(rt_context *ctx) {
    if (!ctx) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is null");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipMemsetAsync(ctx->d_counters, 0, sizeof(DevCounters), ctx->stream));
    ctx->dispatches = 0; ctx->launches = 0;
    return RT_OK;
}

int rt_last_dispatch_ms(rt_context *ctx, float *ms) {
    if (!ctx || !ms) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (!ctx->timed) return fail(RT_ERR_INVALID_ARGUMENT, "no dispatch has been timed yet");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipEventSynchronize(ctx->ev1));
    RT_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return RT_OK;
}

int rt_app_render(rt_context *ctx, RtAppState *app, double now) {
    if (!ctx || !app) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    const int resized = rt_host_app_begin_frame(app, now);
    if (app->uniform.window[0] == 0 || app->uniform.window[1] == 0) return RT_OK;   // nothing to dispatch
    if (resized || !ctx->d_accum || ctx->width != app->uniform.window[0] || ctx->height != app->uniform.window[1]) {
        int r = rt_accum_resize(ctx, app->uniform.window[0], app->uniform.window[1]);   // src/app.rs:87-93
        if (r) return r;
    }
    return rt_dispatch(ctx, &app->uniform);                                            // src/app.rs:97-101
}

int rt_host_scene_finish(const rt_host_scene *s, rt_context *ctx) {
    if (!s || !ctx) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    const rt_host::Scene &sc = s->scene;
    int r = rt_scene_upload(ctx, sc.vertices.data(), sc.vertices.size(), sc.indices.data(), sc.indices.size(),
                            sc.primitives.data(), (uint32_t)sc.primitives.size(), nullptr, sc.models.data(),
                            (uint32_t)sc.models.size());
    if (r || sc.textures.empty()) return r;
    std::vector<RtTexture> tex(sc.textures.size());
    for (size_t i = 0; i < tex.size(); i++) {
        tex[i].rgba8 = sc.textures[i].rgba.data();
        tex[i].width = sc.textures[i].width;
        tex[i].height = sc.textures[i].height;
    }
    return rt_textures_upload(ctx, tex.data(), (uint32_t)tex.size());
}

}  // extern "C"


// ---------------------------------------------------------------------------------------------------------------
// rt_group: one host object that owns n GPUs (SURVEY §8(b) `create(device_ids[], n)`; the reference's single context
// owning all resources, src/main.rs:27-59).  The scene is validated, built and repacked ONCE and uploaded to every
// device; each device renders its bands (rt_set_shard) on its own stream; downloads gather the bands on the first
// device (peer copies over xGMI) and de-interleave them there.  Everything is driven from one host thread; launches
// are asynchronous, so the devices render concurrently.
struct rt_group {
    std::vector<rt_context *> ctx;
    std::vector<hipEvent_t> done;      // per shard: its stream's work up to the gather
    std::vector<int> peer;             // per shard: how its bands reach device 0 (rt_group_peer_access)
    float *d_gather = nullptr;         // on ctx[0]'s device: all shards' packed rows, one shard after the other
    float *d_image = nullptr;          // on ctx[0]'s device: the de-interleaved image
    uint64_t image_elems = 0;
    uint32_t width = 0, height = 0;
};

namespace {

// the full image, de-interleaved, in g->d_image on the first device (or that device's own buffer when there is one shard)
int group_gather(rt_group *g, const float **image) {
    rt_context *c0 = g->ctx[0];
    const size_t n = g->ctx.size();
    if (!c0->d_accum) return fail(RT_ERR_NO_ACCUM, "no accumulation buffer");
    if (n == 1) { *image = c0->d_accum; return RT_OK; }
    RtGatherLayout L;
    memset(&L, 0, sizeof L);
    L.width = g->width; L.height = g->height; L.n_shards = (uint32_t)n;
    uint32_t row = 0;
    for (size_t i = 0; i < n; i++) {
        L.first_row[i] = row;
        row += (uint32_t)local_rows_for(g->height, (uint32_t)i, (uint32_t)n);
    }
    for (size_t i = 1; i < n; i++) {
        RT_HIP(hipSetDevice(g->ctx[i]->device));
        RT_HIP(hipEventRecord(g->done[i], g->ctx[i]->stream));
    }
    RT_HIP(hipSetDevice(c0->device));
    for (size_t i = 0; i < n; i++) {
        rt_context *c = g->ctx[i];
        const size_t bytes = (size_t)local_rows_for(g->height, (uint32_t)i, (uint32_t)n) * g->width * RT_ACCUM_STRIDE_BYTES;
        if (bytes == 0) continue;
        if (i) RT_HIP(hipStreamWaitEvent(c0->stream, g->done[i], 0));
        float *dst = g->d_gather + (size_t)L.first_row[i] * g->width * 4u;
        if (c->device == c0->device) RT_HIP(hipMemcpyAsync(dst, c->d_accum, bytes, hipMemcpyDeviceToDevice, c0->stream));
        else RT_HIP(hipMemcpyPeerAsync(dst, c0->device, c->d_accum, c->device, bytes, c0->stream));
    }
    RT_HIP(rt_launch_deinterleave(g->d_gather, g->d_image, L, c0->stream));
    *image = g->d_image;
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_group_create(const int *device_ids, int n_devices, rt_group **out) {
    if (!out) return fail(RT_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > RT_GROUP_MAX) return fail(RT_ERR_INVALID_ARGUMENT, "1..%d devices", RT_GROUP_MAX);
    rt_group *g = new rt_group();
    for (int i = 0; i < n_devices; i++) {
        rt_context *c = nullptr;
        hipEvent_t ev = nullptr;
        int r = rt_create(device_ids[i], &c);
        if (r == RT_OK && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) r = fail(RT_ERR_HIP, "hipEventCreate failed");
        if (r == RT_OK) r = rt_set_shard(c, (uint32_t)i, (uint32_t)n_devices);
        if (r != RT_OK) {
            const std::string keep = g_last_error;
            if (ev) (void)hipEventDestroy(ev);
            if (c) rt_destroy(c);
            rt_group_destroy(g);
            g_last_error = keep;
            return r;
        }
        if (i) c->hs = g->ctx[0]->hs;              // ONE host copy of the scene and ONE host BVH for the whole group
        g->ctx.push_back(c);
        g->done.push_back(ev);
    }
    // peer access lets the gather's copies go GPU to GPU over xGMI; without it HIP stages them through the host.  The outcome
    // per shard is kept for rt_group_peer_access: a host-staged gather still gives the right image, only slower — the caller
    // (bench.py --mode group prints it) must be able to see which one it got.
    g->peer.assign((size_t)n_devices, RT_PEER_SAME_DEVICE);
    for (int i = 1; i < n_devices; i++) {
        if (device_ids[i] == device_ids[0]) continue;
        int can = 0;
        g->peer[(size_t)i] = RT_PEER_UNAVAILABLE;
        if (hipDeviceCanAccessPeer(&can, device_ids[0], device_ids[i]) == hipSuccess && can) {
            (void)hipSetDevice(device_ids[0]);
            const hipError_t e = hipDeviceEnablePeerAccess(device_ids[i], 0);
            (void)hipGetLastError();               // clears the sticky "already enabled"
            g->peer[(size_t)i] = (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) ? RT_PEER_ENABLED : RT_PEER_ENABLE_FAILED;
        }
    }
    *out = g;
    return RT_OK;
}

int rt_group_destroy(rt_group *g) {
    if (!g) return RT_OK;
    if (!g->ctx.empty()) {
        (void)hipSetDevice(g->ctx[0]->device);
        if (g->ctx[0]->stream) (void)hipStreamSynchronize(g->ctx[0]->stream);
        dev_free(g->d_gather); dev_free(g->d_image);
    }
    for (size_t i = 0; i < g->ctx.size(); i++) {
        (void)hipSetDevice(g->ctx[i]->device);
        if (g->done[i]) (void)hipEventDestroy(g->done[i]);
        rt_destroy(g->ctx[i]);
    }
    delete g;
    return RT_OK;
}

int rt_group_size(const rt_group *g) { return g ? (int)g->ctx.size() : 0; }

int rt_group_peer_access(const rt_group *g, int *status, int n) {
    if (!g || !status || n < (int)g->ctx.size()) return fail(RT_ERR_INVALID_ARGUMENT, "rt_group_peer_access: room for %d entries needed", g ? (int)g->ctx.size() : 0);
    for (size_t i = 0; i < g->ctx.size(); i++) status[i] = g->peer[i];
    return RT_OK;
}

int rt_group_context(rt_group *g, int shard, rt_context **out) {
    if (!g || !out || shard < 0 || shard >= (int)g->ctx.size()) return fail(RT_ERR_INVALID_ARGUMENT, "shard out of range");
    *out = g->ctx[(size_t)shard];
    return RT_OK;
}

int rt_group_set_option(rt_group *g, int key, int64_t value) {
    if (!g) return fail(RT_ERR_INVALID_ARGUMENT, "group is null");
    for (rt_context *c : g->ctx) { int r = rt_set_option(c, key, value); if (r) return r; }
    return RT_OK;
}

int rt_group_scene_upload(rt_group *g, const RtVertex *vertices, uint64_t n_vertices, const uint32_t *indices, uint64_t n_indices,
                          const RtGeometryPrimitive *primitives, uint32_t n_primitives, const float *transforms,
                          const RtGpuModel *models, uint32_t n_models) {
    if (!g) return fail(RT_ERR_INVALID_ARGUMENT, "group is null");
    if ((n_vertices && !vertices) || (n_indices && !indices) || (n_primitives && !primitives) || (n_models && !models))
        return fail(RT_ERR_INVALID_ARGUMENT, "null array with non-zero count");
    for (uint32_t p = 0; p < n_primitives; p++)
        if (transforms && primitives[p].transformation_offset >= n_primitives)
            return fail(RT_ERR_INVALID_ARGUMENT, "geometry %u: transformation_offset out of range", p);
    SceneHost &hs = *g->ctx[0]->hs;
    hs.vertices.assign(vertices, vertices + n_vertices);
    hs.indices.assign(indices, indices + n_indices);
    hs.primitives.assign(primitives, primitives + n_primitives);
    hs.models.assign(models, models + n_models);
    if (transforms) hs.transforms.assign(transforms, transforms + 12ull * n_primitives);
    else hs.transforms.clear();
    hs.spheres.clear();
    Built b;
    for (rt_context *c : g->ctx) { c->n_textures = 0; c->have_scene = false; }
    int r = host_build(g->ctx[0], false, b);                 // once
    for (size_t i = 0; i < g->ctx.size() && r == RT_OK; i++) r = device_upload(g->ctx[i], b);   // n times
    return r;
}

int rt_group_spheres_upload(rt_group *g, const RtSphere *spheres, uint64_t n_spheres) {
    if (!g || (n_spheres && !spheres)) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (!g->ctx[0]->have_scene) return fail(RT_ERR_NO_SCENE, "rt_group_spheres_upload before rt_group_scene_upload");
    std::vector<RtSphere> old(spheres, spheres + n_spheres);
    old.swap(g->ctx[0]->hs->spheres);
    Built b;
    int r = host_build(g->ctx[0], false, b);
    if (r != RT_OK) { old.swap(g->ctx[0]->hs->spheres); return r; }          // refused by the host half: the old spheres stay in force
    for (size_t i = 0; i < g->ctx.size() && r == RT_OK; i++) r = device_upload(g->ctx[i], b);
    return r;
}

int rt_group_textures_upload(rt_group *g, const RtTexture *textures, uint32_t n_textures) {
    if (!g) return fail(RT_ERR_INVALID_ARGUMENT, "group is null");
    for (rt_context *c : g->ctx) { int r = rt_textures_upload(c, textures, n_textures); if (r) return r; }
    return RT_OK;
}

int rt_group_models_update(rt_group *g, const RtGpuModel *models, uint32_t n_models, const float *transforms) {
    if (!g || (n_models && !models)) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    rt_context *c0 = g->ctx[0];
    if (!c0->have_scene) return fail(RT_ERR_NO_SCENE, "rt_group_models_update before rt_group_scene_upload");
    // the first context takes the update exactly as rt_models_update does (it validates, and commits the shared host copy);
    // the others then follow the path it took: a refit on their own device, or the upload of what the host rebuilt
    Built b;                                                 // filled when the update took a host build (refit or rebuild): ONE, shared
    int r = models_update_impl(c0, models, n_models, transforms, &b);
    if (r) return r;
    if (g->ctx.size() == 1) return RT_OK;
    if (!b.valid) {                                          // c0 refitted on its device: so does every other device
        for (size_t i = 1; i < g->ctx.size() && r == RT_OK; i++) {
            Built own;
            r = device_refit(g->ctx[i], &own);               // (same inputs, so the same path; a device that disagrees rebuilt for itself)
        }
        return r;
    }
    for (size_t i = 1; i < g->ctx.size() && r == RT_OK; i++) r = device_upload(g->ctx[i], b);
    return r;
}

int rt_group_accum_resize(rt_group *g, uint32_t w, uint32_t h) {
    if (!g) return fail(RT_ERR_INVALID_ARGUMENT, "group is null");
    for (rt_context *c : g->ctx) { int r = rt_accum_resize(c, w, h); if (r) return r; }
    g->width = w; g->height = h;
    if (g->ctx.size() > 1) {
        rt_context *c0 = g->ctx[0];
        RT_HIP(hipSetDevice(c0->device));
        RT_HIP(hipStreamSynchronize(c0->stream));
        const uint64_t elems = ((uint64_t)w * h ? (uint64_t)w * h : 1) * 4u;
        if (elems > g->image_elems) {
            dev_free(g->d_gather); dev_free(g->d_image);
            g->image_elems = 0;
            RT_HIP(dev_malloc((void **)&g->d_gather, elems * sizeof(float)));
            RT_HIP(dev_malloc((void **)&g->d_image, elems * sizeof(float)));
            g->image_elems = elems;
        }
    }
    return RT_OK;
}

int rt_group_dispatch(rt_group *g, const RtUniform *uniform) {
    if (!g) return fail(RT_ERR_INVALID_ARGUMENT, "group is null");
    for (rt_context *c : g->ctx) { int r = rt_dispatch(c, uniform); if (r) return r; }
    return RT_OK;
}

int rt_group_dispatch_bands(rt_group *g, const RtUniform *uniform, uint32_t band_begin, uint32_t band_end) {
    if (!g) return fail(RT_ERR_INVALID_ARGUMENT, "group is null");
    for (rt_context *c : g->ctx) { int r = rt_dispatch_bands(c, uniform, band_begin, band_end); if (r) return r; }   // each shard: its own bands of that range
    return RT_OK;
}

int rt_group_dispatch_frames(rt_group *g, RtUniform *uniform, uint32_t n_frames) {
    if (!g || !uniform) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    const RtUniform start = *uniform;
    RtUniform last = start;
    for (rt_context *c : g->ctx) {               // every device advances the same frame sequence from the same start
        RtUniform u = start;
        int r = rt_dispatch_frames(c, &u, n_frames);
        if (r) return r;
        last = u;
    }
    *uniform = last;
    return RT_OK;
}

int rt_group_sync(rt_group *g) {
    if (!g) return fail(RT_ERR_INVALID_ARGUMENT, "group is null");
    for (rt_context *c : g->ctx) { int r = rt_sync(c); if (r) return r; }
    return RT_OK;
}

int rt_group_accum_gather(rt_group *g, void **device_image) {
    if (!g || !device_image) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    const float *image = nullptr;
    int r = group_gather(g, &image);
    if (r) return r;
    rt_context *c0 = g->ctx[0];
    RT_HIP(hipSetDevice(c0->device));
    RT_HIP(hipStreamSynchronize(c0->stream));      // the gather waited for every shard's stream (events), so this is the whole group's work
    *device_image = const_cast<float *>(image);
    return RT_OK;
}

int rt_group_accum_download(rt_group *g, float *dst) {
    if (!g || !dst) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    const float *image = nullptr;
    int r = group_gather(g, &image);
    if (r) return r;
    rt_context *c0 = g->ctx[0];
    RT_HIP(hipSetDevice(c0->device));
    RT_HIP(hipMemcpyAsync(dst, image, (size_t)g->width * g->height * RT_ACCUM_STRIDE_BYTES, hipMemcpyDeviceToHost, c0->stream));
    RT_HIP(hipStreamSynchronize(c0->stream));
    return RT_OK;
}

int rt_group_capture_rgb8(rt_group *g, uint8_t *dst) {
    if (!g || !dst) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    const float *image = nullptr;
    int r = group_gather(g, &image);
    if (r) return r;
    rt_context *c0 = g->ctx[0];
    const uint64_t n_pixels = (uint64_t)g->width * g->height;
    if (n_pixels == 0) return RT_OK;
    RT_HIP(hipSetDevice(c0->device));
    if (c0->rgb_bytes < n_pixels * 3) {
        RT_HIP(hipStreamSynchronize(c0->stream));
        dev_free(c0->d_rgb);
        c0->rgb_bytes = 0;
        RT_HIP(dev_malloc((void **)&c0->d_rgb, n_pixels * 3));
        c0->rgb_bytes = n_pixels * 3;
    }
    RT_HIP(rt_launch_capture(image, c0->d_rgb, n_pixels, c0->stream));
    RT_HIP(hipMemcpyAsync(dst, c0->d_rgb, n_pixels * 3, hipMemcpyDeviceToHost, c0->stream));
    RT_HIP(hipStreamSynchronize(c0->stream));
    return RT_OK;
}

int rt_group_stats(rt_group *g, RtStats *out) {
    if (!g || !out) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    RtStats sum;
    memset(&sum, 0, sizeof sum);
    for (size_t i = 0; i < g->ctx.size(); i++) {
        RtStats s;
        int r = rt_stats(g->ctx[i], &s);
        if (r) return r;
        if (i == 0) sum = s;                     // static figures and host-side counts: the first device's
        else {
            sum.rays += s.rays; sum.pixel_frames += s.pixel_frames; sum.node_visits += s.node_visits; sum.triangle_tests += s.triangle_tests;
            sum.wave_node_iters += s.wave_node_iters; sum.wave_leaf_iters += s.wave_leaf_iters; sum.wave_steps += s.wave_steps;
            sum.launches += s.launches; sum.guard_trips += s.guard_trips;
        }
    }
    *out = sum;
    return RT_OK;
}

int rt_group_stats_reset(rt_group *g) {
    if (!g) return fail(RT_ERR_INVALID_ARGUMENT, "group is null");
    for (rt_context *c : g->ctx) { int r = rt_stats_reset(c); if (r) return r; }
    return RT_OK;
}

}  // extern "C"

// shared with host_api.cpp so host-only entry points report through the same channel
void rt_set_last_error(const std::string &s) { g_last_error = s; }
