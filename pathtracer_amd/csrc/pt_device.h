// pt_device.h — device-side data layout and launcher declarations of the gfx950 wavefront path
// tracer.  Host code (pt_api.cpp) fills these structs; pt_kernels.hip consumes them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pt.h"

namespace ptd {

// ---- scene in HBM -------------------------------------------------------------------------
// nodes : 4 x float4 per pt_bvh_node  (64 B, one node = both children's boxes + child refs)
// tris  : 3 x float4 per triangle, in BVH leaf order:
//           [0] = v0.xyz , bits(global triangle id)
//           [1] = v1.xyz , bits(material id)
//           [2] = v2.xyz , 0
//         f32 vertices are what tobj delivers (obj.rs:25); widening to f64 in the kernel is exact.
// uvs   : 3 x float2 per triangle (leaf order) or nullptr = (0,0),(1,0),(1,1) (triangle_mesh.rs:93)
struct Material {
    double color[3];
    double emission[3];
    double width;  // Beckmann width = max(roughness, 1e-5) (microfacet.rs:125)
    double roughness;  // as given (main.rs:252 scales the differential spread by it)
    int32_t kind;
    int32_t tex;
    double ior;  // TRANSPARENT
};

struct Texture {
    const double* texels;  // all levels, 3 doubles per texel, level l starts at offset[l] texels
    uint32_t n_levels;
    uint32_t width0, height0;
    uint32_t offset[16];
};

struct Camera {  // scene.rs:37-43 with the per-ray invariant parts hoisted to the host
    double eye[3];
    double view_dir[3];  // view_direction * camera_const
    double axes0[3];     // -(view x up).normalize()
    double axes1[3];     // -(axes0 x view).normalize()
};

struct Prim {  // analytic sphere / plane (geometry/mod.rs:20-29), SURVEY §8f N4
    double p[3], n[3];
    double radius;
    int32_t kind;
    uint32_t mat;
};

// What shading needs of a hit triangle that does not depend on the ray: the unit geometric normal
// of triangle_mesh.rs:141 (a cross product, a square root and three divisions per hit otherwise)
// and the material id.  Built once on the device by the very function the kernels used to call, so
// the bits are the same.  Leaf order, 32 bytes.
struct TriShade {
    double n[3];
    uint32_t mat;
    uint32_t pad;
};
// The orthonormal basis Vec3::rotate_to (vec3.rs:154-166) builds around that normal, which every BRDF
// sample at the triangle would otherwise rebuild (a division and ~20 f64 operations): rotate_to(v, n) =
// t1 v.x + t2 v.y + n v.z.  Built by the expressions of rotate_to itself, so the sampled direction has the
// same bits.  Only for scenes of up to kEyeTableMaxTris triangles: for a large scene the 48-byte gather per
// shaded hit costs more than the arithmetic.
struct TriBasis {
    double t1[3], t2[3];
};

// Primary rays all start at the eye, so everything in the slab and triangle tests that depends on
// the origin but not on the direction is the same for every one of them: child boxes relative to
// the eye, and per triangle e0, e1, e0 x e1, v0 - o and (v0 - o).(e0 x e1).  These tables hold
// those values (triangles in f64, computed on the device by the expressions of the general test, so the
// specialised test gives the same bits with about a third fewer instructions; boxes in f32, rounded
// outward: the box test is only a cull).  Built for scenes of up to kEyeTableMaxTris triangles (64 B
// per node, 112 B per triangle).
struct EyeNode {
    float rel[12];  // child 0: lo - o (xyz), hi - o (xyz), rounded outward to f32 (pt_traverse.hip.h slab_rel); child 1 likewise
    int32_t child[2];
    uint32_t pad[2];
};
struct EyeTri {
    double e0[3], e1[3], normal[3], vo[3];
    double num;
    uint32_t gid;
    uint32_t pad;
};
constexpr uint32_t kEyeTableMaxTris = 1u << 16;
constexpr uint32_t kMaxCullEmitters = 8;
constexpr int kLdsScene = 32, kLdsMats = 16;  // nodes / triangles / materials of a scene that the fused kernel holds in LDS
// Scene::lds_image: everything such a kernel keeps in LDS as ONE array of float4, every table at a fixed offset and
// padded to kLdsScene / kLdsMats entries, built once per scene - the kernel's copy is one loop over a prefix (two
// ranges for a kernel that does not trace primary rays) instead of one loop per table.  Order: what every kernel
// needs, the primary rays' tables, the tables of rays from anywhere, the vertices (textured scenes only).
constexpr int kImgShade = 0;                                                        // TriShade[kLdsScene]
constexpr int kImgMats = kImgShade + kLdsScene * 2;                                 // Material[kLdsMats]
constexpr int kImgBasis = kImgMats + kLdsMats * (int)(sizeof(Material) / 16);       // TriBasis[kLdsScene]
constexpr int kImgEyeNodes = kImgBasis + kLdsScene * 3;                             // EyeNode[kLdsScene], plane-major
constexpr int kImgEyeTris = kImgEyeNodes + kLdsScene * 4;                           // EyeTri[kLdsScene]
constexpr int kImgNodes = kImgEyeTris + kLdsScene * 7;                              // nodes, plane-major
constexpr int kImgEdges = kImgNodes + kLdsScene * 4;                                // EyeTri[kLdsScene] of the origin (tri_edges)
constexpr int kImgTris = kImgEdges + kLdsScene * 7;                                 // float4 x 3 [kLdsScene]
constexpr int kImgEnd = kImgTris + kLdsScene * 3;
static_assert(sizeof(Material) % 16 == 0, "materials as float4s");
// An emissive triangle as the last-bounce cull tests it: the origin-independent part of the
// triangle test (v0, e0, e1, e0 x e1 in f64), computed once on the device by the test's own code.
struct EmitTri {
    double v0[3], e0[3], e1[3], normal[3];
};

// n / d for n < 2^31 as (n * mul) >> shift (exact: mul = ceil(2^shift / d), shift = 31 + ceil(log2 d)).
// The slot -> (frame, pixel) -> (x, y) mapping runs once per path vertex; a hardware-less u32
// division costs ~25 instructions, this one 2.
struct FastDiv {
    uint32_t mul, shift;
};
inline FastDiv make_fastdiv(uint32_t d) {
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    return {(uint32_t)(((1ull << (31 + l)) + d - 1) / d), 31 + l};
}  // more emitters than this and testing them all costs more than it saves

struct Light {  // one emissive triangle, or (sphere != 0) an emissive sphere: v0 = centre, v1[0] = radius (NEE)
    double v0[3], v1[3], v2[3];
    double normal[3];  // unit geometric normal
    double emission[3];
    double cdf;        // cumulative area up to and including this triangle
    uint32_t leaf_index;  // index of the triangle in leaf order
    uint32_t sphere;
};

struct Scene {
    const float4* nodes;   // two-child nodes (pt_bvh_node): the tree of a shallow scene (depth <= 8)
    const float4* nodes4;  // four-wide nodes (pt_bvh_node4, 8 x float4 each): the tree of a deep scene
    const float4* tris;
    const float2* uvs;
    const TriShade* tri_shade;  // [n_tris], leaf order
    const TriBasis* tri_basis;  // [n_tris], leaf order, or nullptr (large scenes)
    const EyeNode* eye_nodes;   // primary-ray tables (nullptr: not built for this scene)
    const EyeTri* eye_tris;
    const EyeTri* tri_edges;    // the same records for the origin (0, 0, 0): e0, e1, e0 x e1 and vo = v0 of every triangle, for
                                // rays from anywhere (scenes that live in LDS: a ds_read instead of 24 f64 operations per test)
    const float4* lds_image;    // scenes that live in LDS: all their tables as the path kernel keeps them (kImg..., kImgEnd float4s;
                                // the nodes and eye nodes plane-major, pt_traverse.hip.h node_step_lds)
    const uint32_t* leaf_of;  // global triangle id -> leaf-order index
    const Material* mats;
    const Texture* texs;
    const Light* lights;
    // Emissive triangles (leaf-order indices), for the last-bounce cull: a ray spawned at the
    // second-to-last vertex can only add radiance if it hits one of them (black background), so one
    // that passes none of their triangle tests is dropped instead of being queued and traced.
    const EmitTri* emissive;
    uint32_t n_emissive;
    uint32_t cull_last;  // the cull applies to this scene (black background, few emitters, no emissive sphere)
    const Prim* prims;  // tested against every ray before the tree is walked; hit id PT_PRIM_BASE + k
    uint32_t n_prims;
    uint32_t n_nodes, n_tris, n_lights, bvh_depth;
    uint32_t n_top;  // nodes4[0 .. n_top) are the top of the tree, breadth-first: held in LDS by the traversal kernels (<= kTopNodes; 0 = off)
    double light_area;
    uint32_t mat_features;  // bit 0: some material has a texture; bit 1: some is Glossy / Mirror / Transparent
                            // (selects the shade code compiled in: 0 = the lean untextured-Lambertian kernels)
    // traversal scheduling, resolved at scene creation (pt_api.cpp resolve_scheduling): vote-
    // scheduled ranges of wave_rays rays per wave, or one ray per lane
    uint8_t vote_primary, vote_secondary, vote_shadow;
    uint8_t fuse;  // extend + shade of a bounce in one kernel where its rays go one per lane
    uint8_t lds_scene;  // the scene is small enough to live in LDS inside the fused kernel (k_path LDS)
    uint32_t n_mats;
    uint8_t fuse_shadow;  // ... and the bounce's light sample traced in that kernel too
    uint8_t packet_primary;  // deep trees: the primary rays of a wave walk the four-wide tree together (traverse_packet4) where the
                             // wave holds at most 64 >> packet_primary pixels (Frame::grp_log2 >= packet_primary); 0 = never
    uint8_t fuse01;       // ... and the primary launch of a scene in LDS takes the path's second vertex along (k_path F01)
    uint32_t wave_rays;
    uint32_t n_cu;  // compute units of the device (size of a persistent launch)
    Camera cam;
    double background[3];
    // lights.rs:37-40 AreaLight of PT_FLAG_REFERENCE_LIGHTS: centre, radius, intensity (n_area_lights 0 or 1)
    double area_light[7];
    uint32_t n_area_lights;
};
// A vertex draws a light sample (the NEE instantiations of the shade kernels): the build's next-event
// estimation, or the reference's own direct-light path, which excludes it.
__host__ __device__ inline bool reference_lights(const Scene& sc, uint32_t flags) {
    return (flags & PT_FLAG_REFERENCE_LIGHTS) != 0 && sc.n_area_lights > 0;
}
__host__ __device__ inline bool samples_lights(const Scene& sc, uint32_t flags) {
    return reference_lights(sc, flags) || ((flags & PT_FLAG_NEE) != 0 && sc.n_lights > 0);
}

// ---- ray / path queue (SoA, one entry per path in flight) -----------------------------------
struct Queue {
    double* o[3];
    double* d[3];
    double* T[3];      // path throughput ("accumulated_pdf", main.rs:290)
    double* prev_pdf;  // pdf of the BRDF sample that spawned the ray (NEE/MIS only)
    uint32_t* slot;    // path slot = sample_in_batch * n_tile_pixels + tile_pixel
    // PT_FLAG_PROPAGATE_DIFFS only (else nullptr): the ray's differential rays, 12 arrays of
    // diff_stride doubles: rx_origin, ry_origin, rx_direction, ry_direction (ray.rs:9-12)
    double* diff;
    uint32_t diff_stride;
    // o[0..2], d[0..2], T[0..2] lie `stride9` doubles apart, in this order, from o[0] (carve_workspace); 0 = they do
    // not (the staging of pt_trace_closest).  A kernel that appends through o[0] + k * stride9 keeps 3 scalar
    // registers live for the queue instead of 18: the fused kernels spill scalar registers to vector lanes.
    uint32_t stride9;
};

// Per-slot radiance of the batch ("color_result", main.rs:287).  Most samples of a frame gather no
// radiance at all (95 % on CornellBox), so the array is sparse: c[slot] holds a value only where
// bit (slot & 63) of lit[slot >> 6] is set; elsewhere the sample counts as +0 and c is never
// written or read.  The bounce-0 kernels own whole words (lane i <-> slot i) and store them from a
// ballot, which also clears what the previous batch left; later contributions set their bit with an
// atomic OR.  rgb is interleaved so that a contribution touches one cache line.
struct Sample3 {
    double* c;                 // [slot][3]
    unsigned long long* lit;   // [ceil(slots / 64)]
};

struct Hits {
    uint32_t* tri;  // leaf-order triangle index or PT_MISS
    double* t;
};

struct ShadowQueue {
    double* o[3];
    double* d[3];
    double* tmax;
    double* c[3];  // contribution added to the sample if unoccluded
    uint32_t* slot;
};

// Queues are appended to by many workgroups at once.  One shared counter would serialise every
// append on a single L2 atomic (measured: ~90-140 appends/us, which bounded the shade kernel), so a
// queue is cut into kSegs segments of `seg_size` entries, each with its own counter; workgroup b
// appends to segment b % kSegs.  All segments fill evenly, entries stay dense inside a segment,
// and a consumer workgroup (seg_size is a multiple of the block size) lies inside one segment.
constexpr uint32_t kSegs = 64;
// Every counter sits in its own 128-byte line: device-scope atomics are served at the memory side
// and serialise per line, so 32 counters packed in one line share one counter's throughput.
constexpr uint32_t kCtrStride = 32;  // words between the counters of neighbouring segments
// extend / shadow: one wave works through a range of 64..kSegAlign consecutive entries (64 at a
// time, a finished lane takes the next one).  Segment sizes are multiples of kSegAlign, so such a
// range lies inside one segment.
constexpr uint32_t kSegAlign = 1024;
struct Live {                 // which entries of a queue are live
    const uint32_t* counts;   // nullptr: dense, entries [0, n) are live
    uint32_t seg_size;        // segmented: entry i is live iff i % seg_size < counts[i / seg_size]
    uint32_t n;               // dense: live count; segmented: kSegs * seg_size (index range)
    uint32_t* cursors;        // segmented, optional: kSegs work cursors (zeroed by the host) from which
                              // persistent waves claim ranges of entries; nullptr = one range per wave
    uint32_t sparse;          // segmented: few entries are live (the queue after the last-bounce cull), so the
                              // launch is a small grid whose workgroups enumerate the live blocks themselves
};
struct Push {                 // where survivors are appended
    uint32_t* counts;         // kSegs counters, zeroed by the host before the launch
    uint32_t seg_size;
};
inline uint32_t seg_size_for(uint32_t n_max) {  // entries per segment so that kSegs segments hold any output of n_max lanes
    uint32_t blocks = (n_max + 255u) / 256u;
    uint32_t seg = ((blocks + kSegs - 1) / kSegs) * 256u;
    return (seg + kSegAlign - 1) / kSegAlign * kSegAlign;
}

struct Frame {  // per-call constants of generate / shade
    uint32_t width, height;
    uint32_t x0, y0, tile_w, tile_h;
    uint32_t n_tile_pixels;
    uint32_t n_samples;    // frames in this batch
    uint64_t sample_base;  // frame index of sample 0 of the batch
    uint64_t seed;
    uint32_t max_bounces;
    uint32_t flags;
    const double* jitter;  // [n_samples][2] frame jitters, already scaled by the pixel size
    const double* pix_ss;  // [width + height] screen-space x of every column, y of every row (k_pixel_table: main.rs:412-419
                           // once per call instead of two f64 divisions per sample), or nullptr
    double diff_size;      // CONFIG.diff_size (PT_FLAG_PROPAGATE_DIFFS)
    // derived by finish_frame(): what every lane would otherwise recompute
    FastDiv div_pixels, div_tile_w;  // / n_tile_pixels, / tile_w
    double aspect;                   // (double)width / (double)height (main.rs:412-424 extended, F7)
    // Slot order inside a frame: 8x8-pixel blocks (a wave's 64 primary rays cover a square, not a
    // 64x1 strip: they walk the same part of the tree), then the strip right of the last whole block
    // column, then the rows below the last whole block row, both row by row.
    // Screen bounds of the scene: a primary ray through a pixel outside [cx0, cx1) x [cy0, cy1) cannot
    // hit anything (the projection of the root box, two pixels of margin; the whole image when the
    // eye is inside or beside the box), so such a lane skips the camera ray and the tree.
    uint32_t cx0, cy0, cx1, cy1;
    uint32_t n_on_screen;            // pixels of the tile inside those bounds (the primary rays actually traced, per frame)
    uint32_t w8, h8, n8, bw, rw;     // tile_w & ~7, tile_h & ~7, w8 * h8, w8 / 8, tile_w - w8 (0 -> 1 for the divisor)
    FastDiv div_bw, div_rw;
    // Slot order across the frames of a lane: G = 1 << grp_log2 frames at a time (G divides the lane's frame
    // count).  G = 1: frame after frame (slot = s * pixels + slot in frame).  G > 1: inside a group of G frames
    // a wave's 64 slots are 64 / G pixels of one 8x8 block - whole rows of it - in each of the G frames, so a
    // wave of primary rays covers G x fewer pixels and its rays differ by the frame jitter alone (pt_math.hip.h
    // slot_sp / sp_slot); the pixels outside whole blocks follow, frame after frame.  grp1_log2: the second
    // lane's, for k_accumulate, which reads both.
    uint32_t grp_log2, grp1_log2;
    FastDiv div_rest;                // / (n_tile_pixels - n8)
    // Interleaved row partition (pt_render_params::row_group / row_parts / row_index): tile_h counts only
    // the rows this call renders, and tile row ty is image row y0 + row_off + ty + (ty / row_group) *
    // row_skip, row_off = row_index * row_group, row_skip = (row_parts - 1) * row_group.  All zero = every
    // row (tile row ty is image row y0 + ty).
    uint32_t row_group, row_off, row_skip;
    FastDiv div_group;
};
inline void finish_frame(Frame& fr) {  // (cx0..cy1 are set by the caller; default: no culling)
    fr.div_pixels = make_fastdiv(fr.n_tile_pixels ? fr.n_tile_pixels : 1u);
    fr.div_tile_w = make_fastdiv(fr.tile_w ? fr.tile_w : 1u);
    fr.aspect = (double)fr.width / (double)fr.height;
    fr.w8 = fr.tile_w & ~7u;
    fr.h8 = fr.tile_h & ~7u;
    fr.n8 = fr.w8 * fr.h8;
    fr.bw = fr.w8 >> 3;
    fr.rw = fr.tile_w - fr.w8;
    fr.div_bw = make_fastdiv(fr.bw ? fr.bw : 1u);
    fr.div_rw = make_fastdiv(fr.rw ? fr.rw : 1u);
    fr.div_rest = make_fastdiv(fr.n_tile_pixels > fr.n8 ? fr.n_tile_pixels - fr.n8 : 1u);
    fr.div_group = make_fastdiv(fr.row_group ? fr.row_group : 1u);
    auto overlap = [](uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1) {
        uint32_t lo = a0 > b0 ? a0 : b0, hi = a1 < b1 ? a1 : b1;
        return hi > lo ? hi - lo : 0u;
    };
    uint32_t rows = 0;  // tile rows inside the screen bounds
    if (fr.row_skip == 0 && fr.row_off == 0) rows = overlap(fr.y0, fr.y0 + fr.tile_h, fr.cy0, fr.cy1);
    else
        for (uint32_t ty = 0; ty < fr.tile_h; ++ty) {
            const uint32_t gy = fr.y0 + fr.row_off + ty + ty / fr.row_group * fr.row_skip;
            rows += gy >= fr.cy0 && gy < fr.cy1;
        }
    fr.n_on_screen = overlap(fr.x0, fr.x0 + fr.tile_w, fr.cx0, fr.cx1) * rows;
}

struct Counters {  // device-side statistics (pt_stats)
    unsigned long long rays, shadow_rays, nodes, tris, lit, unqueued;
};

constexpr int kStackDepth = 32;  // deepest two-child tree accepted (bvh_validate); the builder stays below it
// Traversal stacks.  A shallow tree (depth <= 8) is walked as it is with 8 LDS entries per lane.  A deeper
// one is collapsed to four-wide nodes (pt_bvh.cpp bvh_collapse4), whose stack need the collapse bounds by
// kWideStack = kWideLds entries in LDS + kWideSpill in private memory (touched only by the rare ray that
// goes beyond 16).
constexpr int kWideLds = 16, kWideSpill = 48, kWideStack = kWideLds + kWideSpill;
// The top of a four-wide tree in LDS: every ray starts at the root and spends its first two or three node steps
// among the same few nodes, so the first kTopNodes nodes in breadth-first order (bvh_collapse4 numbers them first:
// the root, its children, their children = 1 + 4 + 16) are copied into LDS by every wave of a traversal kernel and
// read from there (ds_read_b128 instead of a trip through the texture addresser and the vector L1).  Same boxes,
// same order: hit ids and distances do not change.  2688 B per wave beside its 4 KB stack.
#ifndef PT_TOP_NODES
#define PT_TOP_NODES 21
#endif
constexpr int kTopNodes = PT_TOP_NODES;
#ifndef PT_PATH_TOP
#define PT_PATH_TOP 5
#endif
constexpr int kPathTopNodes = PT_PATH_TOP;  // ... and by the deep-tree path kernel (k_path), one walk per workgroup: two levels
constexpr int kBlock = 256;
constexpr int kTravBlock = 64;  // extend / shadow: one wave per workgroup, LDS and registers are released per wave

// ---- launchers (pt_kernels.hip) ---------------------------------------------------------------
void launch_generate_rays(const Scene& sc, const Frame& fr, double* rays_out /* pt_ray[n] */, hipStream_t st);
// count == nullptr -> n_max rays are live
void launch_extend(const Scene& sc, Queue q, const double* tmin, const double* tmax, Live live, Hits hits,
                   Counters* ctr, bool stats, hipStream_t st);
void launch_extend_primary(const Scene& sc, const Frame& fr, Hits hits, Counters* ctr, bool stats, hipStream_t st);
void launch_shade(const Scene& sc, const Frame& fr, uint32_t bounce, Queue in, Hits hits, Live live, Queue out,
                  Push out_push, ShadowQueue sq, Push sq_push, Sample3 lsample, hipStream_t st);
void launch_shadow(const Scene& sc, ShadowQueue sq, Live live, Sample3 lsample, Counters* ctr, hipStream_t st);
// fused extend + shade of one bounce; false if this scene / bounce has no fused form (caller then
// launches extend and shade)
bool path_fusable(const Scene& sc, uint32_t bounce);
// mid_push.counts != nullptr (bounce 0 only, path_two_bounces()): the launch traces bounces 0 and 1, `out` / `out_push`
// are the queue of bounce 2 and mid_push's counters receive the number of bounce-1 rays
bool path_two_bounces(const Scene& sc, const Frame& fr);
bool launch_path(const Scene& sc, const Frame& fr, uint32_t bounce, Queue in, Live live, Queue out, Push out_push,
                 ShadowQueue sq, Push sq_push, Sample3 lsample, Counters* ctr, hipStream_t st, Push mid_push = Push{nullptr, 0});
void launch_count_rays(const uint32_t* seg_counts, Counters* ctr, hipStream_t st);
// lsample.c covers every frame of the batch; frames [0, split) have their bits in lsample.lit, the
// rest (the second lane's) in lit1, both indexed from the lane's first frame
// the fused path kernel traced its own shadow rays (Scene::fuse_shadow) and left their number in the
// segment counters of the shadow queue it did not fill: add them to the statistics
void launch_count_shadow_rays(const uint32_t* seg_counts, Counters* ctr, hipStream_t st);
void launch_accumulate(const Frame& fr, Sample3 lsample, const unsigned long long* lit1, uint32_t split, double* accum,
                       Counters* ctr /* nullptr: no statistics */, hipStream_t st);
// parity entry points
void launch_build_eye_tables(const Scene& sc, EyeNode* nodes, EyeTri* tris, hipStream_t st);
void launch_pixel_table(uint32_t width, uint32_t height, double* out, hipStream_t st);  // Frame::pix_ss
void launch_build_tri_edges(const Scene& sc, EyeTri* tris, hipStream_t st);
void launch_plane_major(const float* src, float* dst, uint32_t n_nodes, hipStream_t st);  // two-child nodes / EyeNodes -> plane-major
void launch_build_tri_shade(const Scene& sc, TriShade* out, TriBasis* basis /* or nullptr */, hipStream_t st);
void launch_build_emitters(const Scene& sc, const uint32_t* leaf, uint32_t n, EmitTri* out, hipStream_t st);
void launch_leaf_to_global(const Scene& sc, uint32_t* tri, uint32_t n, hipStream_t st);
void launch_sample_brdf(const Scene& sc, const Frame& fr, uint32_t bounce, const double* rays /* pt_ray[n] */,
                        const uint32_t* tri_leaf, const double* t, const uint32_t* pixel, uint32_t n, double* f,
                        double* wi, double* pdf, double* emission, double* position, hipStream_t st);
void launch_texture_lookup(const Scene& sc, uint32_t tex, const double* uv, const double* duvdx,
                           const double* duvdy, uint32_t n, double* rgb, hipStream_t st);

}  // namespace ptd
