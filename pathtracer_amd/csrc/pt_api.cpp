// pt_api.cpp — the C ABI of include/pt.h: scene upload, the wavefront render loop and the parity
// entry points.  Host-side orchestration only; all radiance arithmetic runs in pt_kernels.hip.
// There is deliberately no CPU fallback: without a usable HIP device every compute call fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "pt_device.h"
#include "pt_host.h"

using pth::set_error;

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return PT_ERR_DEVICE;                                                        \
        }                                                                                \
    } while (0)

namespace {

struct DeviceGuard {  // make the scene's device current for the duration of a call
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

struct DevBuf {  // owning device allocation
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n; else p = nullptr;
        return e;
    }
    template <class T>
    T* as() const { return (T*)p; }
};

struct Carver {  // carves 256-byte aligned arrays out of one allocation
    char* base;
    size_t off = 0;
    explicit Carver(void* b) : base((char*)b) {}
    template <class T>
    T* take(size_t n) {
        size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
        T* r = base ? (T*)(base + off) : nullptr;
        off += bytes;
        return r;
    }
};

constexpr uint32_t kMaxBounces = 61;
// counters[]: kSegs segment counters per bounce for the path queue, the same for the shadow queue,
// then the work cursors of the persistent extend and shadow launches (one set per bounce each)
constexpr uint32_t kCounterSets = kMaxBounces + 3;
constexpr uint32_t kSetWords = ptd::kSegs * ptd::kCtrStride;  // one set: kSegs counters, a 128-byte line each
constexpr uint32_t kCounterWords = 4 * kCounterSets * kSetWords;

// One batch of frames is traced as one or two "lanes" (sample sub-ranges), each with its own
// queues, hit records and counters.  Two lanes run on two streams, so that one lane's kernels fill
// the chip while the other's drain (the vote-scheduled extend has a long tail) and HBM-bound shade
// work overlaps FP64-bound extend work.  Measured: +6 % on the 10M-triangle scene, -2 % on
// CornellBox, hence on only for deep trees (pt_scene::pipeline).
struct Lane {
    ptd::Queue q[2];
    ptd::Hits hits;
    ptd::ShadowQueue sq;
    uint32_t* counters = nullptr;
    unsigned long long* lit = nullptr;  // Sample3::lit of this lane's slots
};

struct Workspace {
    DevBuf buf;
    uint32_t cap_slots = 0;   // Lsample entries (all lanes together)
    uint32_t lane_slots = 0;  // path slots per lane
    int n_lanes = 0;
    bool nee = false, diffs = false;
    Lane lane[2];
    ptd::Sample3 ls;
    DevBuf jitter;
    uint64_t jitter_cap = 0, tab_cap = 0;  // frames / pixel-table doubles (behind the jitters) the buffer holds
    uint32_t tab_w = 0, tab_h = 0;         // the image size the pixel table was last filled for
};

void carve_workspace(Workspace& w, void* base, uint32_t n_slots, uint32_t lane_slots, int n_lanes, bool nee, bool diffs,
                     size_t& total) {
    Carver c(base);
    // queues, hit records and the shadow queue are segmented (ptd::kSegs segments); Lsample is per slot
    const uint32_t slots = ptd::kSegs * ptd::seg_size_for(lane_slots);
    for (int l = 0; l < n_lanes; ++l) {
        Lane& ln = w.lane[l];
        for (int k = 0; k < 2; ++k) {
            for (int a = 0; a < 3; ++a) ln.q[k].o[a] = c.take<double>(slots);
            for (int a = 0; a < 3; ++a) ln.q[k].d[a] = c.take<double>(slots);
            for (int a = 0; a < 3; ++a) ln.q[k].T[a] = c.take<double>(slots);
            ln.q[k].prev_pdf = nee ? c.take<double>(slots) : nullptr;
            ln.q[k].slot = c.take<uint32_t>(slots);
            ln.q[k].diff = diffs ? c.take<double>((size_t)12 * slots) : nullptr;
            ln.q[k].diff_stride = slots;
            ln.q[k].stride9 = (uint32_t)((((size_t)slots * sizeof(double) + 255) & ~(size_t)255) / sizeof(double));  // Carver::take
        }
        ln.hits.tri = c.take<uint32_t>(slots);
        ln.hits.t = c.take<double>(slots);
        if (nee) {
            for (int a = 0; a < 3; ++a) ln.sq.o[a] = c.take<double>(slots);
            for (int a = 0; a < 3; ++a) ln.sq.d[a] = c.take<double>(slots);
            ln.sq.tmax = c.take<double>(slots);
            for (int a = 0; a < 3; ++a) ln.sq.c[a] = c.take<double>(slots);
            ln.sq.slot = c.take<uint32_t>(slots);
        } else {
            std::memset(&ln.sq, 0, sizeof ln.sq);
        }
        ln.counters = c.take<uint32_t>(kCounterWords);
    }
    w.ls.c = c.take<double>((size_t)3 * n_slots);
    for (int l = 0; l < n_lanes; ++l) w.lane[l].lit = c.take<unsigned long long>(((size_t)lane_slots + 63) / 64);
    w.ls.lit = w.lane[0].lit;
    total = c.off;
}

struct TimedLaunch {
    int kind;
    hipEvent_t a, b;
};

}  // namespace

struct pt_scene {
    int device = 0;
    uint32_t n_tris = 0, n_texs = 0, n_mats = 0;
    std::vector<pt_bvh_node> nodes;
    std::vector<uint32_t> order;
    uint32_t depth = 0;
    ptd::Scene dev{};
    std::vector<pt_bvh_node4> nodes4;  // deep trees: the four-wide device tree
    uint32_t stack_need = 0;           // entries the deepest path of the device tree needs
    DevBuf d_nodes4;
    DevBuf d_nodes, d_tris, d_uvs, d_leaf_of, d_mats, d_texs, d_lights, d_emissive, d_prims, d_tri_shade, d_tri_basis, d_eye_nodes, d_eye_tris, d_tri_edges, d_lds_image, d_ctr;
    std::vector<DevBuf*> d_texels;
    Workspace ws;
    std::mutex mu;
    std::mutex host_mu;   // pt_render: the cached device copy of the caller's host buffer
    DevBuf d_host_accum;
    int profiling = 0;
    pt_stats stats{};
    std::vector<TimedLaunch> timed;
    std::vector<hipEvent_t> event_pool;
    int pipeline = 0;               // 0: one lane; 1: two lanes on two streams for small batches (see Lane); 2: always two
    int persistent = 1;             // vote-scheduled launches claim their ranges from work cursors
    int screen_cull = 1;            // primary rays outside the scene's screen bounds are not generated
    int group_frames = 0;           // slot order: up to 2^group_frames frames of a few pixels per wave (Frame::grp_log2)
    hipStream_t side = nullptr;     // the second lane's stream
    hipEvent_t ws_free = nullptr;   // recorded at the end of every pt_render_device call: the workspace is free again
    bool ws_used = false;
    bool timers_off = false;        // set for the duration of a pt_render_device_noalloc call (no event creation)
    std::vector<hipEvent_t> sync;   // cross-stream ordering events (no timing)
    ~pt_scene() {
        if (side) (void)hipStreamDestroy(side);
        if (ws_free) (void)hipEventDestroy(ws_free);
        for (auto e : sync) (void)hipEventDestroy(e);
        for (auto* b : d_texels) delete b;
        for (auto e : event_pool) (void)hipEventDestroy(e);
        for (auto& t : timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    }
};

namespace {

// host-side f64 helpers for the camera basis (scene.rs:38-42), same operation order as vec3.rs
struct H3 {
    double x, y, z;
};
H3 hcross(H3 a, H3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
H3 hnorm(H3 a) {
    double l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return {a.x / l, a.y / l, a.z / l};
}
H3 hneg(H3 a) { return {-a.x, -a.y, -a.z}; }

// Traversal scheduling of a scene (pt_traverse.hip.h trace_range_vote / traverse).  Defaults from the
// sweeps recorded in DESIGN.md "Traversal scheduling": primary rays one ray per lane; secondary and
// shadow rays vote-scheduled over ranges of 512 rays when the tree is deeper than 8 levels (scenes
// from ~500 triangles up).  pt_scene_options overrides each.
inline bool opt(int8_t v, bool dflt) { return v < 0 ? dflt : v != 0; }
bool valid_wave_rays(uint32_t r) { return r >= 64 && r <= ptd::kSegAlign && (r & (r - 1)) == 0; }
void resolve_scheduling(ptd::Scene& dv, const pt_scene_options& o) {
    const bool deep = dv.bvh_depth > 8;
    dv.vote_primary = opt(o.vote_primary, false);
    dv.vote_secondary = opt(o.vote_secondary, deep);
    dv.vote_shadow = opt(o.vote_shadow, deep);
    dv.fuse = opt(o.fuse, true);
    dv.fuse_shadow = opt(o.fuse_shadow, true) && dv.fuse;
    dv.fuse01 = 0;  // set once the scene is on the device (scene_create_impl: needs lds_scene and a probe of the primary rays)
    // (128 / 256 / 512 / 1024 -> 8.73 / 9.17 / 9.29 / 9.3 Gsamples/s on the 10 M-triangle scene)
    dv.wave_rays = o.wave_rays ? o.wave_rays : 512;
}

template <class T>
int upload(DevBuf& b, const T* src, size_t n) {
    HIP_TRY(b.alloc(std::max<size_t>(n, 1) * sizeof(T)));
    if (n) HIP_TRY(hipMemcpy(b.p, src, n * sizeof(T), hipMemcpyHostToDevice));
    return PT_OK;
}

int ensure_workspace(pt_scene* s, uint32_t slots, uint32_t lane_slots, int n_lanes, bool nee, bool diffs) {
    Workspace& w = s->ws;
    if (w.buf.p && w.cap_slots >= slots && w.lane_slots >= lane_slots && w.n_lanes >= n_lanes && (w.nee || !nee) &&
        (w.diffs || !diffs))
        return PT_OK;
    HIP_TRY(hipDeviceSynchronize());
    w.buf.release();
    w.cap_slots = 0;
    size_t total = 0;
    Workspace probe;
    carve_workspace(probe, nullptr, slots, lane_slots, n_lanes, nee, diffs, total);
    hipError_t e = w.buf.alloc(total);
    if (e != hipSuccess) {
        set_error("workspace allocation of %zu bytes failed: %s", total, hipGetErrorString(e));
        return PT_ERR_NOMEM;
    }
    carve_workspace(w, w.buf.p, slots, lane_slots, n_lanes, nee, diffs, total);
    w.diffs = diffs;
    w.cap_slots = slots;
    w.lane_slots = lane_slots;
    w.n_lanes = n_lanes;
    w.nee = nee;
    return PT_OK;
}

constexpr uint64_t kMinLanePaths = 1u << 20;  // below this a lane cannot fill 256 CUs; keep one lane
constexpr uint64_t kMaxTwoLanePaths = 100000000;  // above this one lane is as fast (see pt_render_device)

int ensure_side_stream(pt_scene* s, size_t n_events) {
    if (!s->side) HIP_TRY(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
    while (s->sync.size() < n_events) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        s->sync.push_back(e);
    }
    return PT_OK;
}

struct Timer {  // optional HIP-event bracket around one kernel launch
    pt_scene* s;
    hipStream_t st;
    int kind;
    bool on;
    hipEvent_t a{}, b{};
    Timer(pt_scene* s_, hipStream_t st_, int kind_) : s(s_), st(st_), kind(kind_), on(s_->profiling >= 1 && !s_->timers_off) {
        if (!on) return;
        auto get = [&](hipEvent_t& e) {
            if (!s->event_pool.empty()) { e = s->event_pool.back(); s->event_pool.pop_back(); return true; }
            return hipEventCreate(&e) == hipSuccess;
        };
        on = get(a) && get(b);
        if (on) (void)hipEventRecord(a, st);
    }
    ~Timer() {
        if (!on) return;
        (void)hipEventRecord(b, st);
        s->timed.push_back({kind, a, b});
    }
};

// Reads the recorded event pairs into pt_stats.  Called where the host waits anyway (pt_get_stats,
// profiling switched off) and when many launches are pending — not per render call: a
// synchronisation per step would put a host round trip between steps of a timed loop.
constexpr size_t kMaxPendingTimers = 1u << 14;
int collect_timers(pt_scene* s) {
    if (s->timed.empty()) return PT_OK;
    HIP_TRY(hipDeviceSynchronize());
    for (auto& t : s->timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            s->stats.kernel[t.kind].launches += 1;
            s->stats.kernel[t.kind].ms += (double)ms;
        }
        s->event_pool.push_back(t.a);
        s->event_pool.push_back(t.b);
    }
    s->timed.clear();
    return PT_OK;
}

// rows of a tile of `th` rows that a call renders (pt_render_params::row_group / row_parts / row_index)
uint32_t owned_rows(const pt_render_params* p) {
    const uint32_t th = p->y1 - p->y0;
    if (p->row_parts <= 1) return th;
    // (check_params keeps row_group <= height; 64-bit all the same, so that no product can wrap)
    const uint64_t G = p->row_group, N = p->row_parts, r = p->row_index;
    const uint64_t groups = (th + G - 1) / G;           // the last one may be short
    if (r >= groups) return 0;
    const uint64_t mine = (groups - r + N - 1) / N;     // groups r, r + N, ...
    uint64_t rows = mine * G;
    if ((groups - 1) % N == r) rows -= groups * G - th;  // the short last group is this call's
    return (uint32_t)rows;
}

int check_params(const pt_render_params* p) {
    if (!p) { set_error("null params"); return PT_ERR_INVALID; }
    if (p->width == 0 || p->height == 0 || (uint64_t)p->width * p->height >= (1ull << 32)) {
        set_error("bad image size %ux%u", p->width, p->height);
        return PT_ERR_INVALID;
    }
    if (p->x0 >= p->x1 || p->y0 >= p->y1 || p->x1 > p->width || p->y1 > p->height) {
        set_error("bad tile [%u,%u)x[%u,%u) for %ux%u", p->x0, p->x1, p->y0, p->y1, p->width, p->height);
        return PT_ERR_INVALID;
    }
    if (p->max_bounces > kMaxBounces) { set_error("max_bounces > %u", kMaxBounces); return PT_ERR_UNSUPPORTED; }
    if ((uint64_t)(p->x1 - p->x0) * (p->y1 - p->y0) > (1ull << 30)) { set_error("tile too large"); return PT_ERR_UNSUPPORTED; }
    if (p->row_parts > 1 && (p->row_group == 0 || p->row_group > p->height || p->row_index >= p->row_parts)) {
        set_error("bad row partition: group %u (1..height), part %u of %u", p->row_group, p->row_index, p->row_parts);
        return PT_ERR_INVALID;
    }
    if (p->reserved != 0) {
        set_error("pt_render_params::reserved is %u, must be 0 (zero-initialise the struct)", p->reserved);
        return PT_ERR_INVALID;
    }
    if ((p->flags & PT_FLAG_REFERENCE_LIGHTS) && (p->flags & PT_FLAG_NEE)) {
        set_error("PT_FLAG_REFERENCE_LIGHTS and PT_FLAG_NEE exclude each other");
        return PT_ERR_INVALID;
    }
    if ((p->flags & PT_FLAG_PROPAGATE_DIFFS) && !std::isfinite(p->diff_size)) {
        set_error("diff_size is not finite");
        return PT_ERR_INVALID;
    }
    return PT_OK;
}

}  // namespace

namespace ptd {
void launch_jitter(uint64_t seed, uint64_t sample_begin, uint32_t n, double psx, double psy, double* out, hipStream_t st);
}

extern "C" {

int pt_bvh_build(const float* verts, uint32_t n_verts, const uint32_t* tris, uint32_t n_tris, uint32_t max_leaf,
                 pt_bvh_node* nodes_out, uint32_t cap_nodes, uint32_t* n_nodes_out, uint32_t* order_out,
                 uint32_t* max_depth_out, uint32_t n_threads) {
    if ((n_tris && (!verts || !tris)) || !nodes_out || !n_nodes_out || (n_tris && !order_out)) {
        set_error("null argument");
        return PT_ERR_INVALID;
    }
    std::vector<pt_bvh_node> nodes;
    std::vector<uint32_t> order;
    uint32_t depth = 0;
    int rc;
    try {
        rc = pth::bvh_build(verts, n_verts, tris, n_tris, max_leaf, nodes, order, depth, n_threads);
    } catch (const std::bad_alloc&) {
        set_error("out of host memory while building the BVH");
        return PT_ERR_NOMEM;
    }
    if (rc != PT_OK) { set_error("bvh build failed (bad index, non-finite vertex or max_leaf)"); return rc; }
    if (nodes.size() > cap_nodes) { set_error("nodes_out too small: need %zu", nodes.size()); return PT_ERR_INVALID; }
    std::memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(pt_bvh_node));
    if (n_tris) std::memcpy(order_out, order.data(), order.size() * sizeof(uint32_t));
    *n_nodes_out = (uint32_t)nodes.size();
    if (max_depth_out) *max_depth_out = depth;
    return PT_OK;
}

int pt_bvh_validate(const pt_bvh_node* nodes, uint32_t n_nodes, const uint32_t* tri_order, uint32_t n_tris,
                    uint32_t* max_depth_out) {
    if ((n_tris && (!nodes || !tri_order))) { set_error("null argument"); return PT_ERR_INVALID; }
    std::string why;
    uint32_t depth = 0;
    std::vector<uint8_t> seen;
    int rc;
    try {
        rc = pth::bvh_validate(nodes, n_nodes, tri_order, n_tris, (uint32_t)ptd::kStackDepth, depth, why);
        seen.assign(n_tris, 0);
    } catch (const std::bad_alloc&) {
        set_error("out of host memory while validating the BVH");
        return PT_ERR_NOMEM;
    }
    if (rc != PT_OK) { set_error("invalid BVH: %s", why.c_str()); return rc; }
    for (uint32_t i = 0; i < n_tris; ++i) {
        if (seen[tri_order[i]]) { set_error("invalid BVH: triangle %u listed twice", tri_order[i]); return PT_ERR_INVALID; }
        seen[tri_order[i]] = 1;
    }
    if (max_depth_out) *max_depth_out = depth;
    return PT_OK;
}

void pt_scene_options_init(pt_scene_options* o) {
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->struct_size = (uint32_t)sizeof *o;
    o->vote_primary = o->vote_secondary = o->vote_shadow = o->fuse = o->fuse_shadow = o->light_cull = o->screen_cull =
        o->pipeline = o->persistent = o->eye_tables = o->lds_scene = o->group_frames = o->lds_top = o->fuse01 = o->packet_primary = PT_OPT_DEFAULT;
}

void pt_scene_options_from_env(pt_scene_options* o) {
    if (!o) return;
    pt_scene_options_init(o);
    auto flag = [](const char* name, int8_t& v) {
        const char* e = getenv(name);
        if (e && *e) v = atoi(e) != 0;
    };
    auto count = [](const char* name, uint32_t& v) {
        const char* e = getenv(name);
        if (e && *e && atoi(e) > 0) v = (uint32_t)atoi(e);
    };
    flag("PT_TRAV_VOTE_PRIMARY", o->vote_primary);
    flag("PT_TRAV_VOTE", o->vote_secondary);
    flag("PT_TRAV_VOTE_SHADOW", o->vote_shadow);
    flag("PT_FUSE", o->fuse);
    flag("PT_FUSE_SHADOW", o->fuse_shadow);
    flag("PT_LIGHT_CULL", o->light_cull);
    flag("PT_SCREEN_CULL", o->screen_cull);
    flag("PT_PIPELINE", o->pipeline);
    flag("PT_PERSISTENT", o->persistent);
    flag("PT_EYE_TABLES", o->eye_tables);
    flag("PT_LDS_SCENE", o->lds_scene);
    flag("PT_LDS_TOP", o->lds_top);
    flag("PT_FUSE01", o->fuse01);
    flag("PT_PACKET", o->packet_primary);
    if (const char* e = getenv("PT_GROUP_FRAMES")) if (*e) o->group_frames = (int8_t)std::max(0, std::min(6, atoi(e)));
    count("PT_WAVE_RAYS", o->wave_rays);
    if (!valid_wave_rays(o->wave_rays)) o->wave_rays = 0;  // a sweep's bad value falls back to the default
    count("PT_BVH_LEAF", o->bvh_max_leaf);
    count("PT_BVH_THREADS", o->bvh_threads);
}

int pt_scene_create(const pt_scene_desc* d, int device_id, pt_scene** out) {
    pt_scene_options o;
    pt_scene_options_from_env(&o);
    return pt_scene_create_ex(d, device_id, &o, out);
}

static int scene_create_impl(const pt_scene_desc* d, int device_id, const pt_scene_options* options, pt_scene** out);
namespace { void screen_bounds(const pt_scene* s, uint32_t width, uint32_t height, ptd::Frame& fr); }
// No exception crosses the C ABI: a scene of 10 M triangles needs ~2 GB of host memory while it is built.
int pt_scene_create_ex(const pt_scene_desc* d, int device_id, const pt_scene_options* options, pt_scene** out) {
    try {
        return scene_create_impl(d, device_id, options, out);
    } catch (const std::bad_alloc&) {
        set_error("out of host memory while building the scene");
        return PT_ERR_NOMEM;
    } catch (const std::exception& e) {
        set_error("scene creation failed: %s", e.what());
        return PT_ERR_INVALID;
    }
}
static int scene_create_impl(const pt_scene_desc* d, int device_id, const pt_scene_options* options, pt_scene** out) {
    if (!d || !out) { set_error("null argument"); return PT_ERR_INVALID; }
    *out = nullptr;
    pt_scene_options o;
    pt_scene_options_init(&o);
    if (options) {
        if (options->struct_size != sizeof o) {
            set_error("pt_scene_options::struct_size is %u, this library's is %zu (call pt_scene_options_init first)",
                      options->struct_size, sizeof o);
            return PT_ERR_INVALID;
        }
        o = *options;
    }
    if (o.wave_rays && !valid_wave_rays(o.wave_rays)) { set_error("wave_rays must be a power of two in 64..%u", ptd::kSegAlign); return PT_ERR_INVALID; }
    if (o.bvh_max_leaf > 8) { set_error("bvh_max_leaf must be in 1..8"); return PT_ERR_INVALID; }
    if (o.bvh_threads > 64) { set_error("bvh_threads must be in 1..64"); return PT_ERR_INVALID; }
    if (d->n_tris && (!d->verts || !d->tris)) { set_error("verts/tris missing"); return PT_ERR_INVALID; }
    if (d->n_tris && d->n_mats == 0) { set_error("scene has triangles but no materials"); return PT_ERR_INVALID; }
    if ((d->n_mats && !d->mats) || (d->n_texs && !d->texs)) { set_error("mats/texs missing"); return PT_ERR_INVALID; }
    for (uint32_t i = 0; i < d->n_mats; ++i) {
        const pt_material& m = d->mats[i];
        if (m.kind < PT_MAT_LAMBERT || m.kind > PT_MAT_GLOSSY_GGX) { set_error("material %u: bad kind", i); return PT_ERR_INVALID; }
        if (m.kind == PT_MAT_TRANSPARENT && !(m.ior > 0.f)) { set_error("material %u: transparent needs ior > 0", i); return PT_ERR_INVALID; }
        if (m.tex >= (int32_t)d->n_texs) { set_error("material %u: texture index out of range", i); return PT_ERR_INVALID; }
    }
    for (uint32_t i = 0; i < d->n_tris; ++i) {
        if (d->tri_material && d->tri_material[i] >= d->n_mats) { set_error("triangle %u: material out of range", i); return PT_ERR_INVALID; }
        for (int k = 0; k < 3; ++k)
            if (d->tris[3 * i + k] >= d->n_verts) { set_error("triangle %u: vertex out of range", i); return PT_ERR_INVALID; }
    }
    if (d->n_prims && !d->prims) { set_error("prims missing"); return PT_ERR_INVALID; }
    if (d->n_prims > PT_MAX_PRIMS) { set_error("more than %u analytic primitives", PT_MAX_PRIMS); return PT_ERR_UNSUPPORTED; }
    for (uint32_t k = 0; k < d->n_prims; ++k) {
        const pt_primitive& q = d->prims[k];
        if (q.kind != PT_PRIM_SPHERE && q.kind != PT_PRIM_PLANE) { set_error("primitive %u: bad kind", k); return PT_ERR_INVALID; }
        if (q.material >= d->n_mats) { set_error("primitive %u: material out of range", k); return PT_ERR_INVALID; }
        const pt_material& m = d->mats[q.material];
        bool emissive = m.kind == PT_MAT_LAMBERT && (m.emission[0] != 0.f || m.emission[1] != 0.f || m.emission[2] != 0.f);
        if (emissive && q.kind == PT_PRIM_PLANE) { set_error("primitive %u: an emissive plane has infinite area", k); return PT_ERR_UNSUPPORTED; }
        if (q.kind == PT_PRIM_SPHERE && !(q.radius > 0.)) { set_error("primitive %u: sphere radius must be positive", k); return PT_ERR_INVALID; }
    }
    if (d->n_area_lights && !d->area_lights) { set_error("area_lights missing"); return PT_ERR_INVALID; }
    if (d->n_area_lights > 1) { set_error("more than one area light"); return PT_ERR_UNSUPPORTED; }
    for (uint32_t k = 0; k < d->n_area_lights; ++k) {
        const pt_area_light& al = d->area_lights[k];
        bool finite = std::isfinite(al.radius);
        for (int a = 0; a < 3; ++a) finite = finite && std::isfinite(al.center[a]) && std::isfinite(al.intensity[a]);
        if (!finite || !(al.radius >= 0.)) { set_error("area light %u: centre, radius >= 0 and intensity must be finite", k); return PT_ERR_INVALID; }
    }
    int n_dev = 0;
    hipError_t dev_err = hipGetDeviceCount(&n_dev);
    if (dev_err != hipSuccess || n_dev <= 0) {
        set_error("no HIP device available: hipGetDeviceCount -> %s, %d devices (the HIP path is the only implementation)",
                  hipGetErrorString(dev_err), n_dev);
        return PT_ERR_DEVICE;
    }
    if (device_id < 0 || device_id >= n_dev) { set_error("device %d out of range (%d devices)", device_id, n_dev); return PT_ERR_INVALID; }

    std::unique_ptr<pt_scene> s(new pt_scene());
    s->device = device_id;
    s->n_tris = d->n_tris;
    s->n_texs = d->n_texs;
    s->n_mats = d->n_mats;

    // ---- BVH: the caller's, or our own (bvh.rs:17-21)
    if (d->bvh_nodes) {
        if (!d->bvh_tri_order && d->n_tris) { set_error("bvh_nodes given without bvh_tri_order"); return PT_ERR_INVALID; }
        s->nodes.assign(d->bvh_nodes, d->bvh_nodes + d->n_bvh_nodes);
        s->order.assign(d->bvh_tri_order, d->bvh_tri_order + d->n_tris);
        // the caller's tree is checked as given, then widened by the margin the f32 box tests need
        // (pt_bvh.cpp; pt_scene_get_bvh hands back the widened boxes).  Nothing to widen without triangles.
        uint32_t depth = 0;
        int rc = pt_bvh_validate(s->nodes.data(), (uint32_t)s->nodes.size(), s->order.data(), d->n_tris, &depth);
        if (rc != PT_OK) return rc;
        if (d->n_tris && d->verts) pth::bvh_widen(s->nodes, d->verts, d->n_verts);
    } else {
        // leaves of at most 2 triangles (a quad): measured against 1/3/4/8 on both bench scenes
        // (CornellBox 13.3/13.6/11.6/11.5, 10 M triangles 3.03/3.42/3.16/2.84 Gsamples/s; 2: 13.6/3.43)
        const uint32_t max_leaf = o.bvh_max_leaf ? o.bvh_max_leaf : 2;
        int rc = pth::bvh_build(d->verts, d->n_verts, d->tris, d->n_tris, max_leaf, s->nodes, s->order, s->depth, o.bvh_threads);
        if (rc != PT_OK) { set_error("bvh build failed (non-finite vertex?)"); return rc; }
    }
    {   // the traversal kernels trust the tree (indices, leaf runs, depth <= their LDS stack): check it all
        uint32_t depth = 0;
        int rc = pt_bvh_validate(s->nodes.data(), (uint32_t)s->nodes.size(), s->order.data(), d->n_tris, &depth);
        if (rc != PT_OK) return rc;
        s->depth = depth;
    }

    DeviceGuard guard(device_id);
    if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return PT_ERR_DEVICE; }

    // ---- triangles in leaf order, 3 x float4 each
    std::vector<float> tri4((size_t)d->n_tris * 12);
    std::vector<float> uv6;
    std::vector<uint32_t> leaf_of(d->n_tris);
    if (d->tri_uvs) uv6.resize((size_t)d->n_tris * 6);
    for (uint32_t leaf = 0; leaf < d->n_tris; ++leaf) {
        uint32_t g = s->order[leaf];
        leaf_of[g] = leaf;
        uint32_t mat = d->tri_material ? d->tri_material[g] : 0;
        for (int k = 0; k < 3; ++k) {
            const float* v = d->verts + 3 * (size_t)d->tris[3 * g + k];
            float* dst = &tri4[(size_t)leaf * 12 + k * 4];
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
            uint32_t w = k == 0 ? g : (k == 1 ? mat : 0u);
            std::memcpy(&dst[3], &w, 4);
        }
        if (d->tri_uvs) std::memcpy(&uv6[(size_t)leaf * 6], d->tri_uvs + (size_t)g * 6, 6 * sizeof(float));
    }
    int rc;
    if (s->depth > 8) {  // deep: the kernels walk four-wide nodes (pt_bvh.cpp bvh_collapse4)
        rc = pth::bvh_collapse4(s->nodes, (uint32_t)ptd::kWideStack, s->nodes4, s->stack_need, 85u /* three levels breadth-first: more than any build holds in LDS */);
        if (rc != PT_OK) { set_error("four-wide collapse failed: stack need %u > %d", s->stack_need, ptd::kWideStack); return rc; }
        // the traversal addresses a node's planes as base + 32-bit byte offset (pt_traverse.hip.h node4_step)
        if (s->nodes4.size() >= ((size_t)1 << 25)) {
            set_error("%zu four-wide nodes: the traversal addresses at most 2^25", s->nodes4.size());
            return PT_ERR_UNSUPPORTED;
        }
        if ((rc = upload(s->d_nodes4, s->nodes4.data(), s->nodes4.size())) != PT_OK) return rc;
    } else {
        s->stack_need = s->depth;
        if ((rc = upload(s->d_nodes, s->nodes.data(), s->nodes.size())) != PT_OK) return rc;
    }
    if ((rc = upload(s->d_tris, tri4.data(), tri4.size())) != PT_OK) return rc;
    if ((rc = upload(s->d_leaf_of, leaf_of.data(), leaf_of.size())) != PT_OK) return rc;
    if (d->tri_uvs && (rc = upload(s->d_uvs, uv6.data(), uv6.size())) != PT_OK) return rc;

    // ---- materials (f32 -> f64 widening of obj.rs:66-69)
    std::vector<ptd::Material> mats(d->n_mats);
    for (uint32_t i = 0; i < d->n_mats; ++i) {
        const pt_material& m = d->mats[i];
        for (int c = 0; c < 3; ++c) { mats[i].color[c] = (double)m.color[c]; mats[i].emission[c] = (double)m.emission[c]; }
        mats[i].roughness = (double)m.roughness;
        mats[i].width = m.kind == PT_MAT_GLOSSY_GGX ? (double)m.roughness                 // GGX::new, microfacet.rs:197-199
                                                    : std::fmax((double)m.roughness, 1e-5);  // Beckmann::new, microfacet.rs:125
        mats[i].kind = m.kind;
        mats[i].tex = m.tex;
        mats[i].ior = (double)m.ior;
    }
    if ((rc = upload(s->d_mats, mats.data(), mats.size())) != PT_OK) return rc;

    // ---- textures: f64 pyramids (mipmap.rs:85-98)
    std::vector<ptd::Texture> texs(d->n_texs);
    for (uint32_t i = 0; i < d->n_texs; ++i) {
        std::vector<double> texels;
        std::vector<uint32_t> offset;
        rc = pth::build_pyramid(d->texs[i], texels, offset);
        if (rc != PT_OK) return rc;
        DevBuf* b = new DevBuf();
        s->d_texels.push_back(b);
        if ((rc = upload(*b, texels.data(), texels.size())) != PT_OK) return rc;
        std::memset(&texs[i], 0, sizeof texs[i]);
        texs[i].texels = b->as<double>();
        texs[i].n_levels = (uint32_t)offset.size();
        texs[i].width0 = d->texs[i].width;
        texs[i].height0 = d->texs[i].height;
        for (size_t l = 0; l < offset.size(); ++l) texs[i].offset[l] = offset[l];
    }
    if ((rc = upload(s->d_texs, texs.data(), texs.size())) != PT_OK) return rc;

    // ---- lights for NEE: emissive triangles in global id order, area CDF
    std::vector<ptd::Light> lights;
    std::vector<uint32_t> emissive;  // every triangle whose hit adds radiance (leaf order), for the last-bounce cull
    double acc = 0.;
    for (uint32_t g = 0; g < d->n_tris; ++g) {
        const ptd::Material& m = mats[d->tri_material ? d->tri_material[g] : 0];
        if (m.kind != PT_MAT_LAMBERT) continue;  // only Lambertian surfaces emit (microfacet.rs:33-35; legacy specular branches)
        if (m.emission[0] == 0. && m.emission[1] == 0. && m.emission[2] == 0.) continue;
        emissive.push_back(leaf_of[g]);
        H3 v[3];
        for (int k = 0; k < 3; ++k) {
            const float* p = d->verts + 3 * (size_t)d->tris[3 * g + k];
            v[k] = {(double)p[0], (double)p[1], (double)p[2]};
        }
        H3 e1{v[1].x - v[0].x, v[1].y - v[0].y, v[1].z - v[0].z}, e2{v[2].x - v[0].x, v[2].y - v[0].y, v[2].z - v[0].z};
        H3 c = hcross(e1, e2);
        double area = 0.5 * std::sqrt(c.x * c.x + c.y * c.y + c.z * c.z);
        if (!(area > 0.)) continue;
        acc += area;
        ptd::Light L;
        std::memset(&L, 0, sizeof L);
        const H3 n = hnorm(c);
        const double vv[3][3] = {{v[0].x, v[0].y, v[0].z}, {v[1].x, v[1].y, v[1].z}, {v[2].x, v[2].y, v[2].z}};
        for (int a = 0; a < 3; ++a) { L.v0[a] = vv[0][a]; L.v1[a] = vv[1][a]; L.v2[a] = vv[2][a]; L.emission[a] = m.emission[a]; }
        L.normal[0] = n.x; L.normal[1] = n.y; L.normal[2] = n.z;
        L.cdf = acc;
        L.leaf_index = leaf_of[g];
        lights.push_back(L);
    }
    // emissive spheres after the triangles (same order as the oracle's CDF)
    std::vector<ptd::Prim> prims(d->n_prims);
    for (uint32_t k = 0; k < d->n_prims; ++k) {
        const pt_primitive& q = d->prims[k];
        for (int a = 0; a < 3; ++a) { prims[k].p[a] = q.p[a]; prims[k].n[a] = q.n[a]; }
        prims[k].radius = q.radius;
        prims[k].kind = q.kind;
        prims[k].mat = q.material;
        const ptd::Material& m = mats[q.material];
        if (q.kind != PT_PRIM_SPHERE || m.kind != PT_MAT_LAMBERT) continue;
        if (m.emission[0] == 0. && m.emission[1] == 0. && m.emission[2] == 0.) continue;
        acc += 4. * 3.14159265358979323846 * q.radius * q.radius;
        ptd::Light L;
        std::memset(&L, 0, sizeof L);
        for (int a = 0; a < 3; ++a) { L.v0[a] = q.p[a]; L.emission[a] = m.emission[a]; }
        L.v1[0] = q.radius;
        L.cdf = acc;
        L.leaf_index = PT_PRIM_BASE + k;
        L.sphere = 1;
        lights.push_back(L);
    }
    bool emissive_prim = false;
    for (uint32_t k = 0; k < d->n_prims; ++k) {
        const ptd::Material& m = mats[d->prims[k].material];
        if (m.kind == PT_MAT_LAMBERT && (m.emission[0] != 0. || m.emission[1] != 0. || m.emission[2] != 0.)) emissive_prim = true;
    }
    DevBuf d_emissive_leaf;
    if ((rc = upload(d_emissive_leaf, emissive.data(), emissive.size())) != PT_OK) return rc;
    HIP_TRY(s->d_emissive.alloc(std::max<size_t>(emissive.size(), 1) * sizeof(ptd::EmitTri)));
    if ((rc = upload(s->d_lights, lights.data(), lights.size())) != PT_OK) return rc;
    if ((rc = upload(s->d_prims, prims.data(), prims.size())) != PT_OK) return rc;

    HIP_TRY(s->d_ctr.alloc(sizeof(ptd::Counters)));
    HIP_TRY(hipMemset(s->d_ctr.p, 0, sizeof(ptd::Counters)));
    HIP_TRY(hipEventCreateWithFlags(&s->ws_free, hipEventDisableTiming));

    ptd::Scene& dv = s->dev;
    dv.nodes = s->d_nodes.as<float4>();    // null for a deep tree
    dv.nodes4 = s->d_nodes4.as<float4>();  // null for a shallow one
    dv.tris = s->d_tris.as<float4>();
    dv.uvs = d->tri_uvs ? s->d_uvs.as<float2>() : nullptr;
    dv.leaf_of = s->d_leaf_of.as<uint32_t>();
    dv.mats = s->d_mats.as<ptd::Material>();
    dv.texs = s->d_texs.as<ptd::Texture>();
    dv.lights = s->d_lights.as<ptd::Light>();
    dv.emissive = s->d_emissive.as<ptd::EmitTri>();
    dv.n_emissive = (uint32_t)emissive.size();
    // last-bounce cull (pt_shade.hip.h may_reach_emitter): needs a black background and emitters that
    // are a few triangles; pt_scene_options::light_cull = 0 switches it off (tests render both ways)
    dv.cull_last = !emissive_prim && emissive.size() <= ptd::kMaxCullEmitters && d->background[0] == 0. &&
                   d->background[1] == 0. && d->background[2] == 0. && opt(o.light_cull, true);
    dv.prims = s->d_prims.as<ptd::Prim>();
    dv.n_prims = d->n_prims;
    dv.n_nodes = d->n_tris ? (uint32_t)s->nodes.size() : 0u;
    dv.n_tris = d->n_tris;
    dv.n_lights = (uint32_t)lights.size();
    dv.bvh_depth = s->depth;
    // (heightfield 10 M triangles, 0 / 5 / 21 / 28 / 32 / 48 nodes in LDS: 11.69 / 12.07 / 11.88-12.10 / 12.11 / 11.66 / 11.12
    // Gsamples/s in one call: the root and its children carry the gain, more LDS per wave costs occupancy)
    // (10 M-triangle scene: the primary kernel 150.4 -> 135.8 ms per 3 x 256 frames, 12.2 -> 12.6 Gsamples/s overall)
    dv.packet_primary = (dv.bvh_depth > 8 && opt(o.packet_primary, true)) ? 5 : 0;  // (2 pixels x 32 frames per wave: -3 % kernel time, 1 x 64: -9 %, 4 x 16: +10 %)
    if (const char* e = getenv("PT_PACKET_MIN")) if (dv.packet_primary && atoi(e) > 0) dv.packet_primary = (uint8_t)atoi(e);  // (sweep)
    dv.n_top = opt(o.lds_top, true) ? std::min<uint32_t>((uint32_t)ptd::kTopNodes, (uint32_t)s->nodes4.size()) : 0u;
    resolve_scheduling(dv, o);
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device_id));
        dv.n_cu = (uint32_t)prop.multiProcessorCount;
    }
    s->persistent = opt(o.persistent, s->persistent);
    s->screen_cull = opt(o.screen_cull, s->screen_cull);
    // (10 M-triangle scene at 4K, batches of 64 frames, 1 / 4 / 8 / 16 / 32 / 64 frames per wave: 10.13 / 10.74 / 11.06 /
    // 11.31 / 11.59 / 11.79 Gsamples/s.  A tree that lives in LDS gains nothing from coherent primary rays - its
    // bounce-0 waves are full already - and k_accumulate's radiance reads lose their coalescing: CornellBox -2 %,
    // glossy_planes -3.4 % at 64; so only deep trees by default - set below, once the materials are known.)
    s->pipeline = o.pipeline < 0 ? (dv.bvh_depth > 8 ? 1 : 0) : (o.pipeline ? 2 : 0);  // 2: asked for, at any batch size
    dv.mat_features = 0;
    for (uint32_t i = 0; i < d->n_mats; ++i) {
        if (d->mats[i].tex >= 0) dv.mat_features |= 1u;
        if (d->mats[i].kind != PT_MAT_LAMBERT) dv.mat_features |= 2u;
    }
    // ... a textured scene gains too (its waves then look one texel up: Test.obj 4K +3 %)
    s->group_frames = o.group_frames < 0 ? ((dv.bvh_depth > 8 || (dv.mat_features & 1u)) ? 6 : 0) : std::min<int>(o.group_frames, 6);
    dv.light_area = acc;
    dv.n_area_lights = d->n_area_lights;
    for (int a = 0; a < 7; ++a) dv.area_light[a] = 0.;
    if (d->n_area_lights) {
        const pt_area_light& al = d->area_lights[0];
        for (int a = 0; a < 3; ++a) { dv.area_light[a] = al.center[a]; dv.area_light[4 + a] = al.intensity[a]; }
        dv.area_light[3] = al.radius;
    }
    // scene.rs:38-42, hoisted out of Camera::ray
    H3 view{d->camera.view[0], d->camera.view[1], d->camera.view[2]};
    H3 up{d->camera.up[0], d->camera.up[1], d->camera.up[2]};
    H3 a0 = hneg(hnorm(hcross(view, up)));
    H3 a1 = hneg(hnorm(hcross(a0, view)));
    double cc = d->camera.camera_const;
    for (int a = 0; a < 3; ++a) dv.cam.eye[a] = d->camera.eye[a];
    dv.cam.view_dir[0] = view.x * cc; dv.cam.view_dir[1] = view.y * cc; dv.cam.view_dir[2] = view.z * cc;
    dv.cam.axes0[0] = a0.x; dv.cam.axes0[1] = a0.y; dv.cam.axes0[2] = a0.z;
    dv.cam.axes1[0] = a1.x; dv.cam.axes1[1] = a1.y; dv.cam.axes1[2] = a1.z;
    // ---- ray-independent tables, built on the device with the kernels' own arithmetic
    dv.n_tris = d->n_tris;
    HIP_TRY(s->d_tri_shade.alloc(std::max<size_t>(d->n_tris, 1) * sizeof(ptd::TriShade)));
    dv.tri_shade = s->d_tri_shade.as<ptd::TriShade>();
    dv.tri_basis = nullptr;
    if (d->n_tris && d->n_tris <= ptd::kEyeTableMaxTris) {
        HIP_TRY(s->d_tri_basis.alloc((size_t)d->n_tris * sizeof(ptd::TriBasis)));
        dv.tri_basis = s->d_tri_basis.as<ptd::TriBasis>();
    }
    ptd::launch_build_tri_shade(dv, s->d_tri_shade.as<ptd::TriShade>(), s->d_tri_basis.as<ptd::TriBasis>(), nullptr);
    ptd::launch_build_emitters(dv, d_emissive_leaf.as<uint32_t>(), dv.n_emissive, s->d_emissive.as<ptd::EmitTri>(), nullptr);
    HIP_TRY(hipDeviceSynchronize());  // d_emissive_leaf goes out of scope with this function
    dv.eye_nodes = nullptr;
    dv.eye_tris = nullptr;
    if (dv.n_nodes && s->depth <= 8 && d->n_tris <= ptd::kEyeTableMaxTris && opt(o.eye_tables, true)) {
        HIP_TRY(s->d_eye_nodes.alloc(s->nodes.size() * sizeof(ptd::EyeNode)));
        HIP_TRY(s->d_eye_tris.alloc((size_t)d->n_tris * sizeof(ptd::EyeTri)));
        ptd::launch_build_eye_tables(dv, s->d_eye_nodes.as<ptd::EyeNode>(), s->d_eye_tris.as<ptd::EyeTri>(), nullptr);
        dv.eye_nodes = s->d_eye_nodes.as<ptd::EyeNode>();
        dv.eye_tris = s->d_eye_tris.as<ptd::EyeTri>();
    }
    dv.n_mats = d->n_mats;
    dv.lds_scene = dv.eye_nodes && dv.n_nodes <= (uint32_t)ptd::kLdsScene && d->n_tris <= (uint32_t)ptd::kLdsScene &&
                   d->n_mats <= (uint32_t)ptd::kLdsMats && d->n_prims == 0 && opt(o.lds_scene, true);
    dv.tri_edges = nullptr;
    dv.lds_image = nullptr;
    if (dv.lds_scene) {
        HIP_TRY(s->d_tri_edges.alloc((size_t)d->n_tris * sizeof(ptd::EyeTri)));
        ptd::launch_build_tri_edges(dv, s->d_tri_edges.as<ptd::EyeTri>(), nullptr);
        dv.tri_edges = s->d_tri_edges.as<ptd::EyeTri>();
        // everything the path kernel keeps in LDS, in one array at fixed offsets (ptd::kImg...)
        static_assert(sizeof(ptd::EyeNode) == 64 && sizeof(pt_bvh_node) == 64 && sizeof(ptd::EyeTri) == 112 &&
                      sizeof(ptd::TriShade) == 32 && sizeof(ptd::TriBasis) == 48, "table strides of the LDS image");
        HIP_TRY(s->d_lds_image.alloc((size_t)ptd::kImgEnd * 16));
        HIP_TRY(hipMemset(s->d_lds_image.p, 0, (size_t)ptd::kImgEnd * 16));
        char* img = static_cast<char*>(s->d_lds_image.p);
        auto put = [&](int at, const void* src, size_t bytes) {
            return bytes ? hipMemcpy(img + (size_t)at * 16, src, bytes, hipMemcpyDeviceToDevice) : hipSuccess;
        };
        HIP_TRY(hipDeviceSynchronize());  // the tables above were built on the null stream
        HIP_TRY(put(ptd::kImgShade, dv.tri_shade, (size_t)d->n_tris * sizeof(ptd::TriShade)));
        HIP_TRY(put(ptd::kImgMats, dv.mats, (size_t)d->n_mats * sizeof(ptd::Material)));
        HIP_TRY(put(ptd::kImgBasis, dv.tri_basis, (size_t)d->n_tris * sizeof(ptd::TriBasis)));
        HIP_TRY(put(ptd::kImgEyeTris, dv.eye_tris, (size_t)d->n_tris * sizeof(ptd::EyeTri)));
        HIP_TRY(put(ptd::kImgEdges, dv.tri_edges, (size_t)d->n_tris * sizeof(ptd::EyeTri)));
        HIP_TRY(put(ptd::kImgTris, dv.tris, (size_t)d->n_tris * 48));
        ptd::launch_plane_major(reinterpret_cast<const float*>(dv.eye_nodes), reinterpret_cast<float*>(img + (size_t)ptd::kImgEyeNodes * 16), dv.n_nodes, nullptr);
        ptd::launch_plane_major(reinterpret_cast<const float*>(dv.nodes), reinterpret_cast<float*>(img + (size_t)ptd::kImgNodes * 16), dv.n_nodes, nullptr);
        dv.lds_image = reinterpret_cast<const float4*>(img);
    }
    HIP_TRY(hipDeviceSynchronize());
    for (int a = 0; a < 3; ++a) dv.background[a] = d->background[a];
    // Bounces 0 and 1 in one kernel (k_path F01) pays when nearly every primary ray hits something - a lane whose
    // path ended at its first vertex idles through the second.  Probe: the primary rays of a 64 x 64 image (frame
    // jitter 0) inside the scene's screen bounds, traced on the device; on from 50 % hits.  (Measured: CornellBox,
    // every primary ray inside the bounds hits, 31.8 -> 38.5 Gsamples/s; glossy_planes, 9.71 -> 12.02.)
    dv.fuse01 = 0;
    if (dv.lds_scene && o.fuse01 != 0) {
        dv.fuse01 = 1;
        if (o.fuse01 < 0) {
            ptd::Frame fr{};
            fr.width = fr.height = fr.tile_w = fr.tile_h = 64;
            fr.n_tile_pixels = 64 * 64;
            fr.n_samples = 1;
            DevBuf jit, tri, tt;
            HIP_TRY(jit.alloc(2 * sizeof(double)));
            HIP_TRY(hipMemset(jit.p, 0, 2 * sizeof(double)));
            HIP_TRY(tri.alloc(fr.n_tile_pixels * sizeof(uint32_t)));
            HIP_TRY(tt.alloc(fr.n_tile_pixels * sizeof(double)));
            fr.jitter = jit.as<double>();
            screen_bounds(s.get(), 64, 64, fr);
            ptd::finish_frame(fr);
            ptd::launch_extend_primary(dv, fr, ptd::Hits{tri.as<uint32_t>(), tt.as<double>()}, nullptr, false, nullptr);
            HIP_TRY(hipGetLastError());
            std::vector<uint32_t> hit(fr.n_tile_pixels);
            HIP_TRY(hipMemcpy(hit.data(), tri.p, hit.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
            size_t n_hit = 0;
            for (uint32_t v : hit) n_hit += v != PT_MISS;
            dv.fuse01 = fr.n_on_screen > 0 && n_hit * 2 >= (size_t)fr.n_on_screen;
        }
    }

    *out = s.release();
    return PT_OK;
}

void pt_scene_destroy(pt_scene* scene) {
    if (!scene) return;
    DeviceGuard guard(scene->device);
    (void)hipDeviceSynchronize();
    delete scene;
}

int pt_scene_get_bvh(const pt_scene* s, const pt_bvh_node** nodes, uint32_t* n_nodes, const uint32_t** tri_order,
                     uint32_t* max_depth) {
    if (!s) { set_error("null scene"); return PT_ERR_INVALID; }
    if (nodes) *nodes = s->nodes.data();
    if (n_nodes) *n_nodes = (uint32_t)s->nodes.size();
    if (tri_order) *tri_order = s->order.data();
    if (max_depth) *max_depth = s->depth;
    return PT_OK;
}

int pt_set_profiling(pt_scene* s, int level) {
    if (!s || level < 0 || level > 2) { set_error("bad argument"); return PT_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(s->mu);
    DeviceGuard guard(s->device);
    int rc = collect_timers(s);  // launches timed so far count, whatever the level becomes
    s->profiling = level;
    return rc;
}

int pt_reset_stats(pt_scene* s) {
    if (!s) { set_error("null scene"); return PT_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(s->mu);
    DeviceGuard guard(s->device);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(s->d_ctr.p, 0, sizeof(ptd::Counters)));
    int rc = collect_timers(s);  // returns the pending events to the pool
    if (rc != PT_OK) return rc;
    std::memset(&s->stats, 0, sizeof s->stats);
    return PT_OK;
}

int pt_get_stats(pt_scene* s, pt_stats* out) {
    if (!s || !out) { set_error("null argument"); return PT_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(s->mu);
    DeviceGuard guard(s->device);
    HIP_TRY(hipDeviceSynchronize());
    int rc = collect_timers(s);
    if (rc != PT_OK) return rc;
    ptd::Counters c;
    HIP_TRY(hipMemcpy(&c, s->d_ctr.p, sizeof c, hipMemcpyDeviceToHost));
    s->stats.rays = c.rays;
    s->stats.shadow_rays = c.shadow_rays;
    s->stats.nodes_visited = c.nodes;
    s->stats.tris_tested = c.tris;
    s->stats.lit_samples = c.lit;
    s->stats.unqueued_rays = c.unqueued;
    s->stats.scene_bytes = (uint64_t)s->n_tris * 48 + (s->nodes4.empty() ? s->nodes.size() * sizeof(pt_bvh_node) : s->nodes4.size() * sizeof(pt_bvh_node4));
    s->stats.tree_width = s->nodes4.empty() ? 2u : 4u;
    s->stats.stack_need = s->stack_need;
    s->stats.two_bounce = s->dev.fuse01 && s->dev.lds_scene && s->dev.fuse;
    s->stats.packet_primary = s->dev.packet_primary != 0;
    *out = s->stats;
    return PT_OK;
}

namespace {
// Frame::cx0..cy1: the pixels whose primary rays can reach the scene at all.  The root box's corners
// are projected through Camera::ray's own parametrisation (dir ~ view_dir + a0 x + a1 y, scene.rs:38-43;
// pixel <-> x, y as main.rs:412-424) and the rectangle grown by two pixels (the frame jitter moves a
// ray by a quarter pixel at most).  A corner beside or behind the eye, an analytic primitive or the
// environment PT_SCREEN_CULL=0 give the whole image.
void screen_bounds(const pt_scene* s, uint32_t width, uint32_t height, ptd::Frame& fr) {
    fr.cx0 = 0; fr.cy0 = 0; fr.cx1 = width; fr.cy1 = height;
    if (!s->screen_cull || s->dev.n_prims) return;
    if (s->nodes.empty() || s->n_tris == 0) { fr.cx1 = 0; fr.cy1 = 0; return; }   // nothing to hit
    const float* b = s->nodes[0].bounds;   // both children's boxes: lo xyz, hi xyz each
    double lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = b[a]; hi[a] = b[3 + a];
        if (s->nodes[0].child[1] != PT_BVH_EMPTY) { lo[a] = std::fmin(lo[a], (double)b[6 + a]); hi[a] = std::fmax(hi[a], (double)b[9 + a]); }
    }
    const ptd::Camera& c = s->dev.cam;
    const double vv = c.view_dir[0] * c.view_dir[0] + c.view_dir[1] * c.view_dir[1] + c.view_dir[2] * c.view_dir[2];
    const double aspect = (double)width / (double)height;
    double px0 = 1e300, px1 = -1e300, py0 = 1e300, py1 = -1e300;
    for (int k = 0; k < 8; ++k) {
        const double v[3] = {((k & 1) ? hi[0] : lo[0]) - c.eye[0], ((k & 2) ? hi[1] : lo[1]) - c.eye[1], ((k & 4) ? hi[2] : lo[2]) - c.eye[2]};
        const double vz = v[0] * c.view_dir[0] + v[1] * c.view_dir[1] + v[2] * c.view_dir[2];
        const double len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (!(vz > 1e-3 * len * std::sqrt(vv))) return;   // beside / behind the eye (or NaN): no bound
        const double x = (v[0] * c.axes0[0] + v[1] * c.axes0[1] + v[2] * c.axes0[2]) * vv / vz;
        const double y = (v[0] * c.axes1[0] + v[1] * c.axes1[1] + v[2] * c.axes1[2]) * vv / vz;
        const double px = (x / aspect + 1.) * 0.5 * (double)width - 0.5;
        const double py = (1. - y) * 0.5 * (double)height - 0.5;
        px0 = std::fmin(px0, px); px1 = std::fmax(px1, px);
        py0 = std::fmin(py0, py); py1 = std::fmax(py1, py);
    }
    if (!(px0 <= px1 && py0 <= py1)) return;
    auto clampi = [](double v, uint32_t n) { return (uint32_t)std::fmin(std::fmax(v, 0.), (double)n); };
    fr.cx0 = clampi(std::floor(px0) - 2., width);
    fr.cx1 = clampi(std::ceil(px1) + 3., width);
    fr.cy0 = clampi(std::floor(py0) - 2., height);
    fr.cy1 = clampi(std::ceil(py1) + 3., height);
}

// The tile a render call works on: the caller's, cut down to the scene's screen bounds.  A primary ray
// through a pixel outside them hits nothing and adds the background (main.rs:299) - with a black
// background +-0, which leaves the pixel's sum as it is - so such pixels are left out of the tile: no
// slots, no idle lanes in the bounce-0 launch, no share of k_accumulate for them (CornellBox at 16:9:
// 44 % of the columns).  Only with a black background.  Rows are cut only without a row partition (its
// groups are counted from y0).
pt_render_params clip_to_screen(const pt_scene* s, const pt_render_params* P) {
    pt_render_params Q = *P;
    if (s->dev.background[0] != 0. || s->dev.background[1] != 0. || s->dev.background[2] != 0.) return Q;
    ptd::Frame fr{};
    screen_bounds(s, P->width, P->height, fr);
    Q.x0 = std::max(P->x0, fr.cx0);
    Q.x1 = std::max(Q.x0, std::min(P->x1, fr.cx1));
    if (P->row_parts <= 1) {
        Q.y0 = std::max(P->y0, fr.cy0);
        Q.y1 = std::max(Q.y0, std::min(P->y1, fr.cy1));
    }
    return Q;
}
}  // namespace

}  // extern "C"

namespace {
// The wavefront loop of one render call: batches of `spp` frames over the workspace `w` (the scene's
// own, or the static one of pt_render_device_noalloc), `jitter` = room for spp frame jitters.
// Enqueues only: no allocation, no host synchronisation (with profiling on, event pairs come from
// the scene's pool or are created, which is why the no-alloc entry point switches timers off).
int render_batches(pt_scene* s, const pt_render_params* P, double* accum_dev, hipStream_t st, Workspace& w,
                   double* jitter, double* pix_tab, bool fill_tab, uint32_t spp, bool two, bool nee, bool diffs) {
    const uint32_t tw = P->x1 - P->x0, th = owned_rows(P);  // the rows this call renders
    const uint32_t npix = tw * th;
    int rc = PT_OK;
    (void)rc;
    const bool stats = s->profiling >= 2;
    ptd::Counters* ctr = s->d_ctr.as<ptd::Counters>();

    // a bounce runs as the fused path kernel where one exists; the node/triangle counters of
    // profiling level 2 live in the extend kernel, so that level keeps the two-kernel form
    // (and the differential-carrying shade kernel has no fused form)
    auto can_fuse = [&](uint32_t bounce) { return !stats && !diffs && ptd::path_fusable(s->dev, bounce); };
    struct LaneRun {  // one lane's share of the batch
        Lane* ln;
        hipStream_t st;
        ptd::Frame fr{};
        ptd::Sample3 ls;
        uint32_t n_max, seg, cap;
    };
    // one bounce of one lane, split in its two phases
    auto run_extend = [&](LaneRun& r, uint32_t bounce) {
        uint32_t* cursors = s->persistent ? r.ln->counters + (2 * kCounterSets + bounce) * kSetWords : nullptr;
        ptd::Live live = bounce == 0 ? ptd::Live{nullptr, 0, r.n_max, nullptr}
                                     : ptd::Live{r.ln->counters + bounce * kSetWords, r.seg, r.cap, cursors};
        Timer t(s, r.st, PT_K_EXTEND);
        // bounce 0 runs over the dense slot range; later bounces over the segmented queue
        if (bounce == 0) ptd::launch_extend_primary(s->dev, r.fr, r.ln->hits, ctr, stats, r.st);
        else ptd::launch_extend(s->dev, r.ln->q[bounce & 1], nullptr, nullptr, live, r.ln->hits, ctr, stats, r.st);
    };
    // the primary launch of a scene in LDS takes the second vertex along: bounces 0 and 1 in one kernel
    auto run_two = [&](LaneRun& r) {
        uint32_t* path_cnt = r.ln->counters;
        uint32_t* shad_cnt = r.ln->counters + kCounterSets * kSetWords;
        ptd::Live live{nullptr, 0, r.n_max, nullptr};
        ptd::Push mid{path_cnt + 1 * kSetWords, r.seg};    // counts the rays of bounce 1 (statistics)
        ptd::Push push{path_cnt + 2 * kSetWords, r.seg};   // the queue of bounce 2
        ptd::Push spush{shad_cnt, r.seg};
        {
            Timer t(s, r.st, PT_K_PATH);
            ptd::launch_path(s->dev, r.fr, 0, r.ln->q[0], live, r.ln->q[0], push, r.ln->sq, spush, r.ls, ctr, r.st, mid);
        }
        ptd::launch_count_rays(mid.counts, ctr, r.st);
        if (nee) ptd::launch_count_shadow_rays(spush.counts, ctr, r.st);
    };
    auto run_shade = [&](LaneRun& r, uint32_t bounce, bool fused) {
        uint32_t* path_cnt = r.ln->counters;
        uint32_t* shad_cnt = r.ln->counters + kCounterSets * kSetWords;
        ptd::Live live = bounce == 0 ? ptd::Live{nullptr, 0, r.n_max, nullptr}
                                     : ptd::Live{path_cnt + bounce * kSetWords, r.seg, r.cap, nullptr};
        ptd::Push push{path_cnt + (bounce + 1) * kSetWords, r.seg};
        ptd::Push spush{shad_cnt + bounce * kSetWords, r.seg};
        if (fused) {  // extend + shade in one kernel: nothing was launched by run_extend
            Timer t(s, r.st, PT_K_PATH);
            ptd::launch_path(s->dev, r.fr, bounce, r.ln->q[bounce & 1], live, r.ln->q[(bounce + 1) & 1], push, r.ln->sq,
                             spush, r.ls, ctr, r.st);
        } else {
            Timer t(s, r.st, PT_K_SHADE);
            ptd::launch_shade(s->dev, r.fr, bounce, r.ln->q[bounce & 1], r.ln->hits, live, r.ln->q[(bounce + 1) & 1],
                              push, r.ln->sq, spush, r.ls, r.st);
        }
        if (nee && bounce < P->max_bounces) {
            if (fused && s->dev.fuse_shadow) {  // traced inside the path kernel; only their count is left
                ptd::launch_count_shadow_rays(spush.counts, ctr, r.st);
            } else {
                Timer t(s, r.st, PT_K_SHADOW);
                uint32_t* cursors = s->persistent ? r.ln->counters + (3 * kCounterSets + bounce) * kSetWords : nullptr;
                ptd::launch_shadow(s->dev, r.ln->sq, ptd::Live{spush.counts, r.seg, r.cap, cursors}, r.ls, ctr, r.st);
            }
        }
    };

    // the screen-space coordinate of every column and row of the image (Frame::pix_ss): `pix_tab`, width + height
    // doubles, filled here unless the caller knows it to hold this image size already
    if (pix_tab && fill_tab) ptd::launch_pixel_table(P->width, P->height, pix_tab, st);
    for (uint32_t b0 = 0; b0 < P->sample_count; b0 += spp) {
        const uint32_t ns = std::min(spp, P->sample_count - b0);
        ptd::Frame fr{};
        fr.width = P->width; fr.height = P->height;
        fr.x0 = P->x0; fr.y0 = P->y0; fr.tile_w = tw; fr.tile_h = th;
        fr.row_group = fr.row_off = fr.row_skip = 0;
        if (P->row_parts > 1) {
            fr.row_group = P->row_group;
            fr.row_off = P->row_index * P->row_group;
            fr.row_skip = (P->row_parts - 1) * P->row_group;
        }
        fr.n_tile_pixels = npix;
        fr.grp_log2 = fr.grp1_log2 = 0;
        fr.n_samples = ns;
        fr.sample_base = P->sample_begin + b0;
        fr.seed = P->seed;
        fr.max_bounces = P->max_bounces;
        fr.flags = diffs ? P->flags : (P->flags & ~(uint32_t)PT_FLAG_PROPAGATE_DIFFS);
        fr.diff_size = P->diff_size;
        // main.rs:402-405: one jitter pair per frame, scaled by the pixel size (the batch's frames; the
        // previous batch's kernels are ahead of this launch on `st`, and a second lane joins `st` below)
        ptd::launch_jitter(P->seed, fr.sample_base, ns, 1. / (double)P->width, 1. / (double)P->height, jitter, st);
        fr.jitter = jitter;
        fr.pix_ss = pix_tab;
        screen_bounds(s, P->width, P->height, fr);
        ptd::finish_frame(fr);
        // the generate stage is fused into the bounce-0 extend and shade kernels (they rebuild the
        // camera ray from the pixel): no primary-ray queue is written or read.
        const int n_run = (two && ns >= 2) ? 2 : 1;
        LaneRun run[2];
        for (int l = 0; l < n_run; ++l) {
            LaneRun& r = run[l];
            const uint32_t s0 = l == 0 ? 0 : ns - ns / 2, s1 = (n_run == 2 && l == 0) ? ns - ns / 2 : ns;
            r.ln = &w.lane[l];
            r.st = l == 0 ? st : s->side;
            r.fr = fr;
            r.fr.n_samples = s1 - s0;
            r.fr.sample_base = fr.sample_base + s0;
            r.fr.jitter = fr.jitter + (size_t)s0 * 2;
            r.ls.c = w.ls.c + (size_t)3 * s0 * npix;
            r.ls.lit = w.lane[l].lit;
            // deep trees: the lane's frames eight (four, two) at a time, so that a wave of primary rays is a row of
            // eight pixels in eight frames instead of an 8x8 block in one (Frame::grp_log2)
            r.fr.grp_log2 = 0;
            while ((int)r.fr.grp_log2 < s->group_frames && (r.fr.n_samples & ((2u << r.fr.grp_log2) - 1u)) == 0) ++r.fr.grp_log2;
            r.n_max = (s1 - s0) * npix;
            r.seg = ptd::seg_size_for(r.n_max);  // <= the size the lane was carved for
            r.cap = ptd::kSegs * r.seg;
        }
        const bool two_bounces = !stats && !diffs && ptd::path_two_bounces(s->dev, run[0].fr);
        if (n_run == 1) {
            HIP_TRY(hipMemsetAsync(run[0].ln->counters, 0, kCounterWords * sizeof(uint32_t), st));
            if (two_bounces) run_two(run[0]);
            for (uint32_t bounce = two_bounces ? 2 : 0; bounce <= P->max_bounces; ++bounce) {
                const bool fused = can_fuse(bounce);
                if (!fused) run_extend(run[0], bounce);
                run_shade(run[0], bounce, fused);
            }
        } else {
            // The two lanes run free on their streams between a fork and a join.  Forcing them into
            // anti-phase with per-bounce events (extend of one lane beside shade of the other) was
            // measured slower than one lane on both bench scenes: the workgroup dispatcher already
            // interleaves whatever the two queues hold.
            hipStream_t sd = s->side;
            hipEvent_t* ev = s->sync.data();
            HIP_TRY(hipEventRecord(ev[0], st));  // fork: the side stream starts after everything queued so far
            HIP_TRY(hipStreamWaitEvent(sd, ev[0], 0));
            HIP_TRY(hipMemsetAsync(run[0].ln->counters, 0, kCounterWords * sizeof(uint32_t), st));
            HIP_TRY(hipMemsetAsync(run[1].ln->counters, 0, kCounterWords * sizeof(uint32_t), sd));
            if (two_bounces) { run_two(run[0]); run_two(run[1]); }
            for (uint32_t bounce = two_bounces ? 2 : 0; bounce <= P->max_bounces; ++bounce) {
                const bool fused = can_fuse(bounce);
                if (!fused) run_extend(run[0], bounce);
                if (!fused) run_extend(run[1], bounce);
                run_shade(run[0], bounce, fused);
                run_shade(run[1], bounce, fused);
            }
            HIP_TRY(hipEventRecord(ev[1], sd));  // join
            HIP_TRY(hipStreamWaitEvent(st, ev[1], 0));
        }
        fr.grp_log2 = run[0].fr.grp_log2;
        fr.grp1_log2 = n_run == 2 ? run[1].fr.grp_log2 : 0;
        {
            Timer t(s, st, PT_K_ACCUMULATE);
            ptd::launch_accumulate(fr, w.ls, n_run == 2 ? w.lane[1].lit : nullptr, run[0].fr.n_samples, accum_dev,
                                   stats ? ctr : nullptr, st);
        }
        HIP_TRY(hipGetLastError());
    }
    return PT_OK;
}
}  // namespace

extern "C" {

int pt_render_device(pt_scene* s, const pt_render_params* P_in, double* accum_dev, void* stream) {
    if (!s || !accum_dev) { set_error("null argument"); return PT_ERR_INVALID; }
    int rc = check_params(P_in);
    if (rc != PT_OK) return rc;
    if (P_in->sample_count == 0) return PT_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    DeviceGuard guard(s->device);
    if (!guard.ok) { set_error("hipSetDevice(%d) failed", s->device); return PT_ERR_DEVICE; }
    hipStream_t st = (hipStream_t)stream;

    const uint64_t n_asked = (uint64_t)(P_in->x1 - P_in->x0) * owned_rows(P_in) * P_in->sample_count;
    const pt_render_params clipped = clip_to_screen(s, P_in);
    const pt_render_params* P = &clipped;
    const uint32_t tw = P->x1 - P->x0, th = owned_rows(P);  // the rows this call renders
    const uint32_t npix = tw * th;
    if (npix == 0) { s->stats.samples += n_asked; return PT_OK; }  // nothing of the tile on screen, or a row partition with more parts than row groups
    const bool nee = ptd::samples_lights(s->dev, P->flags);
    // differentials only feed texture filtering: without a texture (or a second vertex) nothing reads them
    const bool diffs = (P->flags & PT_FLAG_PROPAGATE_DIFFS) && (P->flags & PT_FLAG_USE_DIFFS) && s->n_texs > 0 &&
                       P->max_bounces > 0;
    // Frames in flight per pass.  Every launch ends in a tail that does not shrink with the batch (the
    // longest rays of the 10M-triangle scene take ~0.4 ms on their own; measured 0.30 + 0.47 ms per
    // frame for its primary kernel at 4K), so batches are large: ~256M paths by default (~52 GB of
    // queues out of 288; 16 / 32 / 64 frames at 4K -> 9.17 / 9.43 / 9.65 Gsamples/s on that scene), halved
    // while the allocation does not fit.
    uint32_t spp = P->spp_per_batch;
    if (spp == 0) {
        spp = std::max<uint32_t>(1u, (1u << 28) / npix);
        // whole groups of frames (Frame::grp_log2): a multiple of 64, or the power of two below
        if (spp >= 64u) spp &= ~63u;
        else while (spp & (spp - 1u)) spp &= spp - 1u;
    }
    spp = std::min(spp, P->sample_count);
    spp = std::min<uint32_t>(spp, std::max<uint32_t>(1u, (1u << 30) / npix));
    // All renders of a scene share its workspace (queues, hit records, counters, per-sample radiance,
    // jitter).  A call on another stream than the previous one must not start while that one's kernels
    // still use it: every call ends by recording `ws_free` and begins by making its stream wait for it
    // (a no-op on the same stream).  s->mu only orders the host-side enqueueing.
    if (s->ws_used) HIP_TRY(hipStreamWaitEvent(st, s->ws_free, 0));
    bool two = false;
    for (;;) {
        // two lanes when a batch has frames to split, enough paths for each half to fill the chip, and is small
        // enough for a launch's tail to matter (10 M-triangle scene at 4K, one lane -> two: batches of 8 / 16 /
        // 32 / 64 frames +1.9 / +0.9 / -0.5 / -0.3 %; a forced pipeline - options, tests - keeps two at any size)
        two = s->pipeline != 0 && spp >= 2 && (uint64_t)(spp / 2) * npix >= kMinLanePaths &&
              (s->pipeline > 1 || (uint64_t)spp * npix < kMaxTwoLanePaths);
        const uint32_t lane_spp = two ? spp - spp / 2 : spp;
        rc = ensure_workspace(s, spp * npix, lane_spp * npix, two ? 2 : 1, nee, diffs);
        if (rc != PT_ERR_NOMEM || spp == 1 || P->spp_per_batch != 0) break;
        spp = (spp + 1) / 2;
    }
    if (rc != PT_OK) return rc;
    Workspace& w = s->ws;
    if (two && (rc = ensure_side_stream(s, 2)) != PT_OK) return rc;
    // the frame jitters, and behind them the pixel table (width + height doubles)
    const uint64_t tab = (uint64_t)P->width + P->height;
    if (w.jitter_cap < spp || w.tab_cap < tab) {
        HIP_TRY(hipDeviceSynchronize());  // an earlier call (on any stream) may still read the old buffer
        const uint64_t jc = std::max<uint64_t>(w.jitter_cap, spp), tc = std::max(w.tab_cap, tab);
        HIP_TRY(w.jitter.alloc((size_t)(jc * 2 + tc) * sizeof(double)));
        w.jitter_cap = jc;
        w.tab_cap = tc;
        w.tab_w = w.tab_h = 0;
    }
    // (a frame loop of one-frame calls would feel one more launch per call: the table is kept while the image size
    // stays; calls on other streams are ordered behind the one that filled it by ws_free, like the workspace itself)
    const bool fill_tab = w.tab_w != P->width || w.tab_h != P->height;
    rc = render_batches(s, P, accum_dev, st, w, w.jitter.as<double>(), w.jitter.as<double>() + w.jitter_cap * 2, fill_tab, spp, two, nee, diffs);
    if (rc == PT_OK) { w.tab_w = P->width; w.tab_h = P->height; }
    if (rc != PT_OK) return rc;
    s->stats.samples += n_asked;  // counted once the frames are enqueued, not when a workspace could not be had
    HIP_TRY(hipEventRecord(s->ws_free, st));
    s->ws_used = true;
    if (s->timed.size() >= kMaxPendingTimers) return collect_timers(s);
    return PT_OK;
}

}  // extern "C"

// ---- pt_render_device_noalloc: the same render over a fixed workspace the library never allocates -----
// `scratch` is the whole workspace of such a call (path queues, hit records, segment counters,
// per-sample radiance, frame jitters), carved exactly like the hipMalloc'ed one.  Every GPU the
// library runs on holds its own copy of the array, at its own address, from the moment the code
// object is loaded until the process ends.  In host code the array's NAME is only a handle for
// hipGetSymbolAddress: the device address it yields (once per device, with that device current) is
// what the kernels, hipMemsetAsync and the carving below are given, never the name itself.
static __device__ __align__(256) unsigned char scratch[PT_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
constexpr uint32_t kNoAllocJitterFrames = 1024;  // frames per batch of a no-alloc call at most (16 KiB of jitters)
std::mutex g_scratch_mu;
unsigned char* g_scratch_ptr[kMaxDevices];  // device address of `scratch` on device d (nullptr: not looked up yet)
// Every call on a GPU shares that GPU's array, whatever scene or stream it comes from.  Per device: a mutex
// held while a call enqueues its work, and an event recorded behind that work which the next call's stream
// waits for - so calls from two scenes, two streams or two host threads are ordered on the device instead of
// overwriting each other's queues.  Made with the symbol lookup (the first call on a device, outside capture).
std::mutex g_scratch_call_mu[kMaxDevices];
hipEvent_t g_scratch_done[kMaxDevices];     // recorded at the end of the last call on device d
bool g_scratch_used[kMaxDevices];

// device `dev` must be current
int scratch_address(int dev, unsigned char** out) {
    if (dev < 0 || dev >= kMaxDevices) { set_error("device %d: the static workspace supports devices 0..%d", dev, kMaxDevices - 1); return PT_ERR_UNSUPPORTED; }
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            set_error("hipGetSymbolAddress(scratch) on device %d failed: %s", dev, hipGetErrorString(e));
            return PT_ERR_DEVICE;
        }
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) {
            set_error("hipEventCreate for the static workspace of device %d failed: %s", dev, hipGetErrorString(e));
            return PT_ERR_DEVICE;
        }
        g_scratch_done[dev] = ev;
        g_scratch_ptr[dev] = (unsigned char*)p;
    }
    *out = g_scratch_ptr[dev];
    return PT_OK;
}
}  // namespace

extern "C" {

int pt_render_device_noalloc(pt_scene* s, const pt_render_params* P_in, double* accum_dev, void* stream) {
    if (!s || !accum_dev) { set_error("null argument"); return PT_ERR_INVALID; }
    int rc = check_params(P_in);
    if (rc != PT_OK) return rc;
    if (P_in->sample_count == 0) return PT_OK;
    const uint64_t n_asked = (uint64_t)(P_in->x1 - P_in->x0) * owned_rows(P_in) * P_in->sample_count;
    const pt_render_params clipped = clip_to_screen(s, P_in);
    const pt_render_params* P = &clipped;
    std::lock_guard<std::mutex> lk(s->mu);
    DeviceGuard guard(s->device);
    if (!guard.ok) { set_error("hipSetDevice(%d) failed", s->device); return PT_ERR_DEVICE; }
    unsigned char* base = nullptr;
    if ((rc = scratch_address(s->device, &base)) != PT_OK) return rc;
    hipStream_t st = (hipStream_t)stream;

    const uint32_t npix = (P->x1 - P->x0) * owned_rows(P);
    if (npix == 0) { s->stats.samples += n_asked; return PT_OK; }  // nothing on screen / an empty row partition
    const bool nee = ptd::samples_lights(s->dev, P->flags);
    const bool diffs = (P->flags & PT_FLAG_PROPAGATE_DIFFS) && (P->flags & PT_FLAG_USE_DIFFS) && s->n_texs > 0 &&
                       P->max_bounces > 0;
    // the frame jitters sit in front of the carved workspace
    // ... and behind them the pixel table (width + height doubles)
    const size_t jitter_only = ((size_t)kNoAllocJitterFrames * 2 * sizeof(double) + 255) & ~(size_t)255;
    const size_t jitter_bytes = jitter_only + ((((size_t)P->width + P->height) * sizeof(double) + 255) & ~(size_t)255);
    if (jitter_bytes >= (size_t)PT_NOALLOC_SCRATCH_BYTES) {
        set_error("an image of %ux%u is too large for the static workspace", P->width, P->height);
        return PT_ERR_UNSUPPORTED;
    }
    const size_t room = (size_t)PT_NOALLOC_SCRATCH_BYTES - jitter_bytes;
    auto needed = [&](uint32_t frames) {
        size_t total = 0;
        Workspace probe;
        carve_workspace(probe, nullptr, frames * npix, frames * npix, 1, nee, diffs, total);
        return total;
    };
    // frames per batch: what the caller asked for, what fits otherwise - one lane, one stream
    uint32_t spp = P->spp_per_batch ? P->spp_per_batch : P->sample_count;
    spp = std::min(std::min(spp, P->sample_count), kNoAllocJitterFrames);
    spp = std::min<uint32_t>(spp, std::max<uint32_t>(1u, (1u << 27) / npix));
    while (spp > 1 && needed(spp) > room) spp = (spp + 1) / 2;
    const size_t total = needed(spp);
    if (total > room) {
        set_error("one frame of a %ux%u tile%s needs a %zu-byte workspace; the static one holds %zu bytes "
                  "(PT_NOALLOC_SCRATCH_BYTES): render smaller tiles, or call pt_render_device",
                  P->x1 - P->x0, P->y1 - P->y0, nee ? " with NEE" : "", total, room);
        return PT_ERR_UNSUPPORTED;
    }
    Workspace w;  // pointers into `scratch` only: w.buf and w.jitter own nothing
    size_t carved = 0;
    carve_workspace(w, base + jitter_bytes, spp * npix, spp * npix, 1, nee, diffs, carved);
    // one call at a time per GPU enqueues over the shared array, behind the previous call's work.  (A stream
    // under capture can neither wait for an event recorded outside its capture nor lend one to later calls:
    // there the caller orders the calls, as the graph's replays are the caller's to order anyway.)
    std::lock_guard<std::mutex> dev_lk(g_scratch_call_mu[s->device]);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = st && hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    if (!capturing && g_scratch_used[s->device]) HIP_TRY(hipStreamWaitEvent(st, g_scratch_done[s->device], 0));
    // What must start at zero is zeroed per batch on `st` (the segment counters, render_batches); the
    // rest of the array keeps the previous call's bytes, which no kernel reads before writing: queue
    // entries and hit records are only read below the counters of the same batch, and the bounce-0
    // kernel stores every word of the radiance bitmap that the batch's accumulate reads.
    s->timers_off = true;   // a HIP-event pair per launch would have to be created: no kernel timing on this path
    // (the static array is every scene's on this GPU, and a captured call must stand on its own: the table is filled per call)
    rc = render_batches(s, P, accum_dev, st, w, (double*)base, (double*)(base + jitter_only), true, spp, false, nee, diffs);
    s->timers_off = false;
    if (rc != PT_OK) return rc;
    if (!capturing) {
        HIP_TRY(hipEventRecord(g_scratch_done[s->device], st));
        g_scratch_used[s->device] = true;
    }
    s->stats.samples += n_asked;
    return PT_OK;
}

int pt_render(pt_scene* s, const pt_render_params* P, double* accum_rgb) {
    if (!s || !accum_rgb) { set_error("null argument"); return PT_ERR_INVALID; }
    int rc = check_params(P);
    if (rc != PT_OK) return rc;
    DeviceGuard guard(s->device);
    if (!guard.ok) { set_error("hipSetDevice(%d) failed", s->device); return PT_ERR_DEVICE; }
    size_t n = (size_t)P->width * P->height * 3;
    std::lock_guard<std::mutex> lk(s->host_mu);  // one host-buffer render of a scene at a time: they share d_host_accum
    DevBuf& acc = s->d_host_accum;               // kept between calls (a hipMalloc + hipFree per frame otherwise)
    if (acc.bytes < n * sizeof(double)) {
        hipError_t e = acc.alloc(n * sizeof(double));
        if (e != hipSuccess) { set_error("accumulation buffer allocation failed: %s", hipGetErrorString(e)); return PT_ERR_NOMEM; }
    }
    HIP_TRY(hipMemcpy(acc.p, accum_rgb, n * sizeof(double), hipMemcpyHostToDevice));
    rc = pt_render_device(s, P, acc.as<double>(), nullptr);
    if (rc != PT_OK) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    HIP_TRY(hipMemcpy(accum_rgb, acc.p, n * sizeof(double), hipMemcpyDeviceToHost));
    return PT_OK;
}

}  // extern "C"

// ---- pt_accum: the frame loop's running sum (main.rs:396-400 `pixels`), resident on the device ---------
struct pt_accum {
    pt_scene* scene = nullptr;
    uint32_t width = 0, height = 0;
    DevBuf sums;           // width * height * 3 doubles
    DevBuf resolved;       // width * height u32 (or 3 u8) of the last resolve
    void* pinned = nullptr;  // page-locked staging for downloads / uploads (width * height * 3 doubles)
    hipStream_t stream = nullptr;
    std::mutex mu;
    ~pt_accum() {
        if (pinned) (void)hipHostFree(pinned);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace ptd {
void launch_resolve(const double* accum, size_t n_pixels, double denom, uint8_t* rgb8, uint32_t* packed, hipStream_t st);
}

extern "C" {

int pt_accum_create(pt_scene* s, uint32_t width, uint32_t height, pt_accum** out) {
    if (!s || !out) { set_error("null argument"); return PT_ERR_INVALID; }
    *out = nullptr;
    if (width == 0 || height == 0 || (uint64_t)width * height >= (1ull << 32)) { set_error("bad image size %ux%u", width, height); return PT_ERR_INVALID; }
    DeviceGuard guard(s->device);
    if (!guard.ok) { set_error("hipSetDevice(%d) failed", s->device); return PT_ERR_DEVICE; }
    std::unique_ptr<pt_accum> a(new (std::nothrow) pt_accum());
    if (!a) { set_error("out of memory"); return PT_ERR_NOMEM; }
    a->scene = s; a->width = width; a->height = height;
    const size_t bytes = (size_t)width * height * 3 * sizeof(double);
    if (a->sums.alloc(bytes) != hipSuccess || a->resolved.alloc((size_t)width * height * 4) != hipSuccess ||
        hipHostMalloc(&a->pinned, bytes, hipHostMallocDefault) != hipSuccess) {
        set_error("accumulation buffer of %ux%u: allocation failed", width, height);
        return PT_ERR_NOMEM;
    }
    HIP_TRY(hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking));
    HIP_TRY(hipMemsetAsync(a->sums.p, 0, bytes, a->stream));
    *out = a.release();
    return PT_OK;
}

void pt_accum_destroy(pt_accum* a) {
    if (!a) return;
    DeviceGuard guard(a->scene->device);
    if (a->stream) (void)hipStreamSynchronize(a->stream);
    delete a;
}

int pt_accum_render(pt_accum* a, const pt_render_params* P) {
    if (!a || !P) { set_error("null argument"); return PT_ERR_INVALID; }
    if (P->width != a->width || P->height != a->height) {
        set_error("params are for %ux%u, the accumulation buffer is %ux%u", P->width, P->height, a->width, a->height);
        return PT_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(a->mu);
    return pt_render_device(a->scene, P, a->sums.as<double>(), a->stream);
}

static int accum_resolve(pt_accum* a, uint64_t iterations, uint8_t* rgb8, uint32_t* packed) {
    if (!a || (!rgb8 && !packed) || iterations == 0) { set_error("bad argument"); return PT_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(a->mu);
    DeviceGuard guard(a->scene->device);
    if (!guard.ok) { set_error("hipSetDevice(%d) failed", a->scene->device); return PT_ERR_DEVICE; }
    const size_t npix = (size_t)a->width * a->height, bytes = rgb8 ? npix * 3 : npix * 4;
    ptd::launch_resolve(a->sums.as<double>(), npix, 1. / (double)iterations, rgb8 ? a->resolved.as<uint8_t>() : nullptr,
                        packed ? a->resolved.as<uint32_t>() : nullptr, a->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(a->pinned, a->resolved.p, bytes, hipMemcpyDeviceToHost, a->stream));
    HIP_TRY(hipStreamSynchronize(a->stream));
    std::memcpy(rgb8 ? (void*)rgb8 : (void*)packed, a->pinned, bytes);
    return PT_OK;
}
int pt_accum_resolve_rgb8(pt_accum* a, uint64_t iterations, uint8_t* rgb8) { return accum_resolve(a, iterations, rgb8, nullptr); }
int pt_accum_resolve_u32(pt_accum* a, uint64_t iterations, uint32_t* packed) { return accum_resolve(a, iterations, nullptr, packed); }

int pt_accum_download(pt_accum* a, double* accum_rgb) {
    if (!a || !accum_rgb) { set_error("null argument"); return PT_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(a->mu);
    DeviceGuard guard(a->scene->device);
    if (!guard.ok) { set_error("hipSetDevice(%d) failed", a->scene->device); return PT_ERR_DEVICE; }
    const size_t bytes = (size_t)a->width * a->height * 3 * sizeof(double);
    HIP_TRY(hipMemcpyAsync(a->pinned, a->sums.p, bytes, hipMemcpyDeviceToHost, a->stream));
    HIP_TRY(hipStreamSynchronize(a->stream));
    std::memcpy(accum_rgb, a->pinned, bytes);
    return PT_OK;
}

int pt_accum_upload(pt_accum* a, const double* accum_rgb) {
    if (!a) { set_error("null argument"); return PT_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(a->mu);
    DeviceGuard guard(a->scene->device);
    if (!guard.ok) { set_error("hipSetDevice(%d) failed", a->scene->device); return PT_ERR_DEVICE; }
    const size_t bytes = (size_t)a->width * a->height * 3 * sizeof(double);
    if (!accum_rgb) {  // start over
        HIP_TRY(hipMemsetAsync(a->sums.p, 0, bytes, a->stream));
        return PT_OK;
    }
    HIP_TRY(hipStreamSynchronize(a->stream));  // the staging buffer may still feed an earlier copy
    std::memcpy(a->pinned, accum_rgb, bytes);
    HIP_TRY(hipMemcpyAsync(a->sums.p, a->pinned, bytes, hipMemcpyHostToDevice, a->stream));
    HIP_TRY(hipStreamSynchronize(a->stream));
    return PT_OK;
}

int pt_accum_device(pt_accum* a, double** sums_dev, void** stream) {
    if (!a) { set_error("null argument"); return PT_ERR_INVALID; }
    if (sums_dev) *sums_dev = a->sums.as<double>();
    if (stream) *stream = (void*)a->stream;
    return PT_OK;
}

int pt_trace_closest(pt_scene* s, const pt_ray* rays, size_t n, uint32_t* tri_id, double* t) {
    if (!s || (n && (!rays || !tri_id || !t))) { set_error("null argument"); return PT_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(s->mu);
    DeviceGuard guard(s->device);
    if (!guard.ok) { set_error("hipSetDevice(%d) failed", s->device); return PT_ERR_DEVICE; }
    const size_t chunk = 1u << 22;
    DevBuf buf;
    std::vector<double> host;
    for (size_t c0 = 0; c0 < n; c0 += chunk) {
        uint32_t m = (uint32_t)std::min(chunk, n - c0);
        // SoA staging: ox oy oz dx dy dz tmin tmax | t_out | tri_out
        host.resize((size_t)m * 8);
        for (uint32_t i = 0; i < m; ++i) {
            const pt_ray& r = rays[c0 + i];
            for (int a = 0; a < 3; ++a) { host[(size_t)a * m + i] = r.o[a]; host[(size_t)(3 + a) * m + i] = r.d[a]; }
            host[(size_t)6 * m + i] = r.t_min;
            host[(size_t)7 * m + i] = r.t_max;
        }
        HIP_TRY(buf.alloc((size_t)m * (9 * sizeof(double) + sizeof(uint32_t))));
        double* db = buf.as<double>();
        HIP_TRY(hipMemcpy(db, host.data(), (size_t)m * 8 * sizeof(double), hipMemcpyHostToDevice));
        ptd::Queue q;
        std::memset(&q, 0, sizeof q);
        for (int a = 0; a < 3; ++a) { q.o[a] = db + (size_t)a * m; q.d[a] = db + (size_t)(3 + a) * m; }
        ptd::Hits hits;
        hits.t = db + (size_t)8 * m;
        hits.tri = (uint32_t*)(db + (size_t)9 * m);
        {
            Timer tm(s, nullptr, PT_K_EXTEND);
            ptd::launch_extend(s->dev, q, db + (size_t)6 * m, db + (size_t)7 * m, ptd::Live{nullptr, 0, m, nullptr}, hits,
                               s->d_ctr.as<ptd::Counters>(), s->profiling >= 2, nullptr);
        }
        ptd::launch_leaf_to_global(s->dev, hits.tri, m, nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(t + c0, hits.t, (size_t)m * sizeof(double), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(tri_id + c0, hits.tri, (size_t)m * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    if (s->timed.size() >= kMaxPendingTimers) return collect_timers(s);
    return PT_OK;
}

int pt_generate_rays(pt_scene* s, const pt_render_params* P, uint64_t sample, pt_ray* rays) {
    if (!s || !rays) { set_error("null argument"); return PT_ERR_INVALID; }
    int rc = check_params(P);
    if (rc != PT_OK) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    DeviceGuard guard(s->device);
    if (!guard.ok) { set_error("hipSetDevice(%d) failed", s->device); return PT_ERR_DEVICE; }
    const uint32_t tw = P->x1 - P->x0, th = P->y1 - P->y0, npix = tw * th;
    DevBuf out, jit;
    HIP_TRY(out.alloc((size_t)npix * sizeof(pt_ray)));
    HIP_TRY(jit.alloc(2 * sizeof(double)));
    ptd::launch_jitter(P->seed, sample, 1, 1. / (double)P->width, 1. / (double)P->height, jit.as<double>(), nullptr);
    ptd::Frame fr;
    std::memset(&fr, 0, sizeof fr);
    fr.width = P->width; fr.height = P->height; fr.x0 = P->x0; fr.y0 = P->y0; fr.tile_w = tw; fr.tile_h = th;
    fr.n_tile_pixels = npix; fr.n_samples = 1; fr.sample_base = sample; fr.seed = P->seed;
    fr.max_bounces = P->max_bounces; fr.flags = P->flags; fr.jitter = jit.as<double>();
    ptd::finish_frame(fr);
    ptd::launch_generate_rays(s->dev, fr, out.as<double>(), nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(rays, out.p, (size_t)npix * sizeof(pt_ray), hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_sample_brdf(pt_scene* s, const pt_ray* rays, const uint32_t* tri_id, const double* t, const uint32_t* pixel,
                   size_t n, uint64_t sample, uint32_t bounce, uint64_t seed, uint32_t flags, uint32_t width,
                   uint32_t height, double* f, double* wi, double* pdf, double* emission, double* position) {
    if (!s || (n && (!rays || !tri_id || !t || !pixel || !f || !wi || !pdf || !emission || !position))) {
        set_error("null argument");
        return PT_ERR_INVALID;
    }
    if (width == 0 || height == 0 || (uint64_t)width * height >= (1ull << 32) || n >= (1ull << 31)) {
        set_error("bad argument");
        return PT_ERR_INVALID;
    }
    for (size_t i = 0; i < n; ++i) {
        // a triangle id, or PT_PRIM_BASE + k for a hit on analytic primitive k (what pt_trace_closest returns)
        const bool prim = tri_id[i] >= PT_PRIM_BASE && tri_id[i] - PT_PRIM_BASE < s->dev.n_prims;
        if (tri_id[i] >= s->n_tris && !prim) { set_error("ray %zu: triangle id out of range", i); return PT_ERR_INVALID; }
        if (pixel[i] >= (uint64_t)width * height) { set_error("ray %zu: pixel out of range", i); return PT_ERR_INVALID; }
    }
    if (n == 0) return PT_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    DeviceGuard guard(s->device);
    if (!guard.ok) { set_error("hipSetDevice(%d) failed", s->device); return PT_ERR_DEVICE; }
    DevBuf d_rays, d_tri, d_t, d_pix, d_out, jit;
    int rc;
    if ((rc = upload(d_rays, rays, n)) != PT_OK) return rc;
    if ((rc = upload(d_tri, tri_id, n)) != PT_OK) return rc;
    if ((rc = upload(d_t, t, n)) != PT_OK) return rc;
    if ((rc = upload(d_pix, pixel, n)) != PT_OK) return rc;
    HIP_TRY(d_out.alloc(n * 13 * sizeof(double)));
    HIP_TRY(jit.alloc(2 * sizeof(double)));
    ptd::launch_jitter(seed, sample, 1, 1. / (double)width, 1. / (double)height, jit.as<double>(), nullptr);
    ptd::Frame fr;
    std::memset(&fr, 0, sizeof fr);
    fr.width = width; fr.height = height; fr.tile_w = width; fr.tile_h = height; fr.n_tile_pixels = width * height;
    fr.n_samples = 1; fr.sample_base = sample; fr.seed = seed; fr.flags = flags; fr.jitter = jit.as<double>();
    ptd::finish_frame(fr);
    double* o = d_out.as<double>();
    ptd::launch_sample_brdf(s->dev, fr, bounce, d_rays.as<double>(), d_tri.as<uint32_t>(), d_t.as<double>(),
                            d_pix.as<uint32_t>(), (uint32_t)n, o, o + 3 * n, o + 6 * n, o + 7 * n, o + 10 * n, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(f, o, 3 * n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(wi, o + 3 * n, 3 * n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pdf, o + 6 * n, n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(emission, o + 7 * n, 3 * n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(position, o + 10 * n, 3 * n * sizeof(double), hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_texture_lookup(pt_scene* s, uint32_t tex, const double* uv, const double* duvdx, const double* duvdy, size_t n,
                      double* rgb) {
    if (!s || (n && (!uv || !duvdx || !duvdy || !rgb))) { set_error("null argument"); return PT_ERR_INVALID; }
    if (tex >= s->n_texs) { set_error("texture %u out of range", tex); return PT_ERR_INVALID; }
    if (n == 0) return PT_OK;
    if (n >= (1ull << 31)) { set_error("too many lookups"); return PT_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(s->mu);
    DeviceGuard guard(s->device);
    if (!guard.ok) { set_error("hipSetDevice(%d) failed", s->device); return PT_ERR_DEVICE; }
    DevBuf d_uv, d_dx, d_dy, d_out;
    int rc;
    if ((rc = upload(d_uv, uv, 2 * n)) != PT_OK) return rc;
    if ((rc = upload(d_dx, duvdx, 2 * n)) != PT_OK) return rc;
    if ((rc = upload(d_dy, duvdy, 2 * n)) != PT_OK) return rc;
    HIP_TRY(d_out.alloc(3 * n * sizeof(double)));
    ptd::launch_texture_lookup(s->dev, tex, d_uv.as<double>(), d_dx.as<double>(), d_dy.as<double>(), (uint32_t)n,
                               d_out.as<double>(), nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(rgb, d_out.p, 3 * n * sizeof(double), hipMemcpyDeviceToHost));
    return PT_OK;
}

}  // extern "C"
