/*
 * pt.h — C ABI of the MI355X (gfx950) radiance inner loop.
 *
 * The reference (emimvi/pathtracer, Rust) has no FFI boundary; its seams are plain Rust
 * signatures inside one crate.  Each entry point below names the Rust item it
 * stands in for.  Everything here is plain C: pointers + sizes, no C++/torch types, so a Rust
 * `extern "C"` block can bind it directly.
 *
 * Conventions
 *   - return 0 (PT_OK) or a negative pt_status; never abort (the reference panics:
 *     obj.rs:16, mipmap.rs:61/88, vec3.rs:256).  pt_last_error() gives the thread-local message.
 *   - all input arrays are copied at create time; the caller keeps ownership.
 *   - outputs are caller-owned buffers (host pointers, or device pointers for the *_device forms).
 *   - a pt_scene is immutable after creation and may be used from several host threads.
 *   - the HIP path is the only implementation: if no gfx950 device/code object is usable the
 *     calls fail with PT_ERR_DEVICE (there is no CPU fallback).
 */
#ifndef PT_H
#define PT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_PRIM_BASE 0xF0000000u /* tri_id of a hit on analytic primitive k is PT_PRIM_BASE + k */
#define PT_MISS 0xFFFFFFFFu /* tri_id value for "no hit" (Option::None of scene.rs:61) */

typedef enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID = -1,     /* bad argument / malformed scene */
    PT_ERR_NOMEM = -2,       /* host or device allocation failed */
    PT_ERR_DEVICE = -3,      /* HIP runtime error, or no usable device */
    PT_ERR_IO = -4,          /* file could not be read / written */
    PT_ERR_UNSUPPORTED = -5  /* e.g. non power-of-two texture (mipmap.rs:86-89 panics) */
} pt_status;

/* material kinds: material.rs:31-58 (Lambertian), microfacet.rs:9-94 (Glossy<Beckmann>); Mirror and
 * Transparent are the two specular models of the reference's legacy integrator (commented
 * `shade()`, main.rs:117-142,178-185; material.rs:116-133), SURVEY.md §8f N3 */
enum { PT_MAT_LAMBERT = 0, PT_MAT_GLOSSY = 1, PT_MAT_MIRROR = 2, PT_MAT_TRANSPARENT = 3, PT_MAT_GLOSSY_GGX = 4 };

/* One material.  Values arrive as f32 from the MTL parser and are widened to f64 exactly as
 * obj.rs:66-69 does (Vec3::from([f32;3]), vec3.rs:187-191).
 *   LAMBERT: color = Kd, emission = Ka (obj.rs:66-69); tex >= 0 replaces color by the mip-mapped
 *            lookup of material.rs:100-111.
 *   GLOSSY : color = reflectance (Kd), roughness = Beckmann width (Ns; clamped to >= 1e-5 by
 *            microfacet.rs:125); emission is ignored (microfacet.rs:33-35 returns zero).
 *   GLOSSY_GGX: the same Glossy with the reference's second distribution, GGX (microfacet.rs:192-243,
 *            private and never instantiated there); roughness = alpha_g, not clamped.
 *   MIRROR : wi = d.reflect(n) (main.rs:179, vec3.rs:138-140), radiance passes unchanged; no
 *            emission (the legacy branch returns only the reflected radiance).
 *   TRANSPARENT: dielectric of index `ior` in air (FresnelDielectric::new(ior), microfacet.rs:
 *            306-358; material.rs:116-124 uses 1.5).  The legacy code traces both the reflected
 *            and the refracted ray and blends them R : 1-R (main.rs:117-142); in the path loop one
 *            of the two is followed, reflection with probability R (first draw of the bounce), so
 *            the expectation is the legacy value and the throughput is unchanged.  No emission. */
typedef struct pt_material {
    float color[3];
    float emission[3];
    float roughness;
    int32_t kind; /* PT_MAT_* */
    int32_t tex;  /* index into pt_scene_desc.texs, or -1 */
    float ior;    /* TRANSPARENT only */
} pt_material;

/* 8-bit RGB image, tightly packed, top row first (the order image::open(..).to_rgb() yields,
 * mipmap.rs:60-74).  Width and height must be powers of two (mipmap.rs:86-89). */
typedef struct pt_texture {
    uint32_t width, height;
    const uint8_t* rgb8;
} pt_texture;

/* scene.rs:20-35 Camera */
typedef struct pt_camera {
    double eye[3];
    double view[3];
    double up[3];
    double camera_const;
} pt_camera;

/* Flattened two-child BVH node (64 B = four 16-byte loads).  bounds = {lo0,hi0,lo1,hi1} of the
 * two children, f32 like the reference's external bvh crate (bvh.rs:37,98-101).  The kernels test
 * boxes in f32 as a pure cull (the f64 triangle test alone decides hits) and need every box to stand
 * 2^-21 of the scene's largest |coordinate| clear of its triangles: pt_bvh_build stores its boxes that
 * way, and pt_scene_create widens a caller's boxes by that margin itself (tight boxes are fine).
 * child >= 0: index of an inner node.  child < 0: leaf, ~child = (first << 3) | (count - 1),
 * a run of `count` (1..8) entries of bvh_tri_order.  PT_BVH_EMPTY: no child. */
#define PT_BVH_EMPTY 0x7FFFFFFF
typedef struct pt_bvh_node {
    float bounds[12];
    int32_t child[2];
    uint32_t pad[2];
} pt_bvh_node;

/* Flat scene = what scene.rs:13-18 + obj.rs:13-107 + triangle_mesh.rs:12-57 hold, with the
 * per-model TriMesh/Geometry boxes flattened into one triangle list in OBJ model/face order
 * (there is no instancing in the reference). */
/* Analytic primitives of geometry/mod.rs:20-158 (no reference scene holds one).
 *   SPHERE: Sphere::intersect, mod.rs:51-72: p = o - c, b = d.p, c = p.p - r^2, t = -b -/+ sqrt(b^2 - c)
 *           (the direction is taken as unit length), the nearer root with t_min < t < t_max;
 *           normal = (hit - c).normalize().
 *   PLANE : Plane::intersect, mod.rs:31-48: |n.d| >= 1e-6, t = (origin - o).n / (n.d), accepted for
 *           t_min <= t <= t_max; normal = n as given.
 * Untextured materials only.  An emissive sphere is also a light of the NEE extension (uniform
 * point on the sphere as lights.rs:7-15,70-78 draws it, proper solid-angle pdf + MIS instead of the
 * constant 1/(4 pi) of lights.rs:76); an emissive plane is refused (infinite area). */
enum { PT_PRIM_SPHERE = 0, PT_PRIM_PLANE = 1 };
typedef struct pt_primitive {
    int32_t kind;      /* PT_PRIM_* */
    uint32_t material; /* index into mats */
    double p[3];       /* sphere centre / plane origin */
    double n[3];       /* plane normal (unused for spheres) */
    double radius;     /* sphere */
} pt_primitive;

/* lights.rs:37-40 AreaLight { mesh: &Sphere, intensity }.  The sphere is the light's sampling domain only:
 * as in the reference, where Scene.lights and the scene's geometry are separate, it is not hit by rays. */
typedef struct pt_area_light {
    double center[3];
    double radius;
    double intensity[3];
} pt_area_light;

typedef struct pt_scene_desc {
    const float* verts;           /* [n_verts][3] positions, f32 as tobj delivers them (obj.rs:22-27) */
    uint32_t n_verts;
    const uint32_t* tris;         /* [n_tris][3] vertex ids */
    uint32_t n_tris;
    const float* tri_uvs;         /* [n_tris][3][2] or NULL = (0,0),(1,0),(1,1) for every triangle
                                     (triangle_mesh.rs:92-94) */
    const uint32_t* tri_material; /* [n_tris] index into mats */
    const pt_material* mats;
    uint32_t n_mats;
    const pt_texture* texs;
    uint32_t n_texs;
    pt_camera camera;
    double background[3];         /* scene.rs:16 */
    /* optional BVH built by the caller (the host side owns the build, bvh.rs:17-21);
     * NULL => the library runs its own binned-SAH builder (pt_bvh_build). */
    const pt_bvh_node* bvh_nodes;
    uint32_t n_bvh_nodes;
    const uint32_t* bvh_tri_order; /* [n_tris] triangle ids in leaf order */
    const pt_primitive* prims;     /* analytic primitives, tested against every ray (at most PT_MAX_PRIMS) */
    uint32_t n_prims;
    const pt_area_light* area_lights; /* Scene.lights (scene.rs:15) as the reference types them; used by
                                         PT_FLAG_REFERENCE_LIGHTS only.  At most one. */
    uint32_t n_area_lights;
} pt_scene_desc;
#define PT_MAX_PRIMS 256u

/* geometry/ray.rs:5-14 without the differential rays (those are rebuilt from the pixel, §DESIGN) */
typedef struct pt_ray {
    double o[3];
    double d[3];
    double t_min; /* Ray::new uses 1e-5 (ray.rs:21) */
    double t_max; /* Ray::new uses f64::MAX (ray.rs:22) */
} pt_ray;

enum {
    PT_FLAG_USE_DIFFS = 1u << 0, /* CONFIG.use_diffs, default true (main.rs:350) */
    PT_FLAG_NEE = 1u << 1,       /* next-event estimation against emissive triangles (+MIS) */
    PT_FLAG_RUSSIAN_ROULETTE = 1u << 3, /* legacy integrator main.rs:85-111: from the 5th path vertex
                                         * on a Lambertian vertex continues with probability
                                         * mean(albedo) and the throughput is divided by it */
    PT_FLAG_DROP_NONFINITE = 1u << 2, /* drop NaN/inf samples instead of accumulating them
                                         (deviation from main.rs:434-436; off by default) */
    PT_FLAG_PROPAGATE_DIFFS = 1u << 4, /* legacy reflect_diff / refract_diff (main.rs:192-281): the ray
                                        * spawned at a vertex carries differentials, so textures on
                                        * secondary hits are filtered (the live render() resets them,
                                        * main.rs:314).  Needs PT_FLAG_USE_DIFFS; ignored by scenes
                                        * without textures, where nothing reads a differential */
    PT_FLAG_ROUGHDIFF = 1u << 5, /* CONFIG.roughdiff (main.rs:351): a Glossy vertex spreads the
                                  * differentials by diff_size * roughness / 0.8 (main.rs:249-253) */
    PT_FLAG_REFERENCE_LIGHTS = 1u << 6 /* the reference's dormant direct-light path as written (lights.rs:42-62,
                                        * 70-78, 110-137), against pt_scene_desc::area_lights: a point drawn
                                        * uniformly on the light's sphere (sample_sphere), rejected when it faces
                                        * away, shadow ray to t_max = |diff|, irradiance = intensity / |diff|^2,
                                        * pdf the constant 1 / (4 pi) whatever the radius, Ld = brdf(0,0,0)
                                        * |wi . n| irradiance / pdf, no MIS (emission on a hit stays unweighted).
                                        * EstimateDirect has no caller in the reference; here it is evaluated at
                                        * every Lambertian vertex that is not the path's last, and
                                        * throughput * Ld is added to the sample (other materials: the
                                        * reference's zero-vector BRDF call is defined for Lambertian only).
                                        * Unoccluded it converges to what a Lambertian sphere of radiance
                                        * 2 intensity / radius^2 gives; under partial occlusion it does not
                                        * (tests/test_lights.py measures both).  Excludes PT_FLAG_NEE. */
};

/* CONFIG of main.rs:320-379 plus the partitioning knobs of SURVEY.md §8e. */
typedef struct pt_render_params {
    uint32_t width, height;   /* full image; reference is square dim x dim (main.rs:382-383) */
    uint32_t x0, y0, x1, y1;  /* tile [x0,x1) x [y0,y1) rendered by this call */
    uint64_t sample_begin;    /* first frame index (main.rs:400 loop iteration, 0-based) */
    uint32_t sample_count;    /* number of frames = samples per pixel added by this call */
    uint32_t max_bounces;     /* CONFIG.bounces, default 2 (main.rs:347) */
    uint32_t flags;           /* PT_FLAG_* */
    uint32_t spp_per_batch;   /* frames kept in flight per wavefront pass; 0 = auto */
    uint64_t seed;            /* key of the counter-based RNG that replaces rand::random (F8) */
    double diff_size;         /* CONFIG.diff_size, default 0 (main.rs:348): spread of the differential
                               * rays spawned at Lambertian / Glossy vertices (PT_FLAG_PROPAGATE_DIFFS) */
    /* Interleaved row partition (the frame closure of main.rs:409-432 cut across GPUs).
     * With row_parts > 1 the call renders, of the tile's rows, only the groups of `row_group`
     * consecutive rows (counted from y0) whose group number g has g % row_parts == row_index; the other
     * rows of accum are not touched.  N calls with row_index 0..N-1 - one per GPU - cover the tile
     * once, each GPU seeing every image region (a contiguous band per GPU leaves the ceiling's GPU
     * idle while the floor's works).  row_group = 8 keeps the kernels' 8x8-pixel ray blocks whole.
     * row_parts = 0 or 1: every row (the other two fields are ignored). */
    uint32_t row_group, row_parts, row_index;
    uint32_t reserved;        /* must be 0 (PT_ERR_INVALID otherwise: zero-initialise the struct) */
} pt_render_params;

typedef struct pt_kernel_stat {
    uint64_t launches;
    double ms; /* summed HIP-event time */
} pt_kernel_stat;

/* PT_K_GENERATE stays 0 in renders: the stage is fused into the bounce-0 kernels.  PT_K_PATH is
 * the fused extend+shade kernel of one bounce (scenes whose rays are traced one per lane); a render
 * reports each bounce under either PT_K_PATH or PT_K_EXTEND + PT_K_SHADE. */
enum { PT_K_GENERATE = 0, PT_K_EXTEND, PT_K_SHADE, PT_K_SHADOW, PT_K_ACCUMULATE, PT_K_PATH, PT_K_COUNT };

typedef struct pt_stats {
    uint64_t samples;        /* render() calls (main.rs:283) */
    uint64_t rays;           /* trace_closest calls (scene.rs:61) actually made: rays that provably add
                              * nothing (outside the scene's screen bounds; last-bounce rays that miss every
                              * emitter) are neither traced nor counted */
    uint64_t shadow_rays;    /* trace_any calls (scene.rs:66) */
    uint64_t nodes_visited;  /* child boxes slab-tested (32 B each); only with pt_set_profiling(..,2) */
    uint64_t tris_tested;    /* Triangle::intersect calls (triangle_mesh.rs:101) */
    pt_kernel_stat kernel[PT_K_COUNT];
    uint64_t lit_samples;    /* samples that gathered any radiance (the others are never stored or read:
                              * DESIGN.md "sparse radiance"); only with pt_set_profiling(..,2) */
    uint64_t scene_bytes;    /* device bytes of the tree and the triangles the traversal reads */
    uint32_t tree_width;     /* children per node of the device tree: 2 (depth <= 8) or 4 (deeper: collapsed) */
    uint32_t stack_need;     /* traversal stack entries the deepest path of the device tree needs */
    uint32_t two_bounce;     /* 1: renders of this scene with max_bounces >= 2 trace bounces 0 and 1 in one launch, so
                              * no path record exists between them (pt_scene_options::fuse01) */
    uint32_t packet_primary; /* 1: primary waves of this scene's deep tree walk it as packets where fully grouped */
    uint64_t unqueued_rays;  /* of `rays`, the secondary rays that never existed as a 76-byte path record: the rays of
                              * bounce 1 where bounces 0 and 1 run in one launch */
} pt_stats;

typedef struct pt_scene pt_scene;

/* How the library schedules the work of one scene.  No counterpart in the reference (its scheduling is rayon's,
 * main.rs:409-432); every combination renders the same image (the tests render them against each other).  Fields
 * are PT_OPT_DEFAULT (the library decides from the scene: what the measurements in DESIGN.md chose), 0 or 1;
 * the counts are 0 for the default. */
#define PT_OPT_DEFAULT (-1)
typedef struct pt_scene_options {
    uint32_t struct_size;    /* sizeof(pt_scene_options), set by pt_scene_options_init: lets the struct grow */
    int8_t vote_primary;     /* traverse primary / secondary / shadow rays as vote-scheduled ranges of wave_rays */
    int8_t vote_secondary;   /*   rays per wave instead of one ray per lane (default: deep trees, not primary) */
    int8_t vote_shadow;
    int8_t fuse;             /* extend + shade of a bounce in one kernel where its rays go one per lane */
    int8_t fuse_shadow;      /* ... and the bounce's light sample traced in that kernel too (needs fuse) */
    int8_t light_cull;       /* drop last-bounce rays that can reach no emitter (black background, few emitters) */
    int8_t screen_cull;      /* clip a tile to the scene's screen bounds (black background) */
    int8_t pipeline;         /* two batches in flight on two streams (default: deep trees) */
    int8_t persistent;       /* persistent launch of the vote-scheduled traversal kernels */
    int8_t eye_tables;       /* per-scene tables for rays from the eye (shallow trees of <= 65 536 triangles) */
    int8_t lds_scene;        /* a tiny scene lives in LDS inside the fused kernel (needs eye_tables) */
    int8_t group_frames;     /* k = 1..6: a wave of primary rays holds 64 / 2^k pixels of an 8x8 block in 2^k frames
                              * instead of the whole block in one frame; 0 = off (default: deep trees 6 - one pixel in 64
                              * frames, as far as the frame count of a batch divides -, others 0) */
    uint32_t wave_rays;      /* rays per vote-scheduled range: a power of two in 64..1024 (default 512) */
    uint32_t bvh_max_leaf;   /* triangles per leaf of the library's own BVH build, 1..8 (default 2) */
    uint32_t bvh_threads;    /* host threads of that build, 1..64 (default: the hardware's, at most 16) */
    int8_t lds_top;          /* deep trees: the top of the four-wide tree (its first 21 nodes, breadth-first) is held in
                              * LDS by every traversal wave (default on) */
    int8_t fuse01;           /* a scene that lives in LDS: the primary launch traces the path's second vertex too
                              * (default: on when a probe of the primary rays finds that at least half of them hit) */
    int8_t packet_primary;   /* deep trees: the 64 primary rays of a wave walk the four-wide tree together (one node
                              * index, one stack, per wave; default on) instead of one traversal per lane */
    int8_t reserved8[1];     /* 0 */
} pt_scene_options;
/* every field to its default */
void pt_scene_options_init(pt_scene_options* options);
/* ... then the overrides of the process environment, for tuning sweeps without a rebuild: PT_TRAV_VOTE_PRIMARY,
 * PT_TRAV_VOTE, PT_TRAV_VOTE_SHADOW, PT_FUSE, PT_FUSE_SHADOW, PT_LIGHT_CULL, PT_SCREEN_CULL, PT_PIPELINE,
 * PT_PERSISTENT, PT_EYE_TABLES, PT_LDS_SCENE, PT_LDS_TOP, PT_FUSE01, PT_PACKET = 0|1; PT_GROUP_FRAMES = 0..6; PT_WAVE_RAYS, PT_BVH_LEAF, PT_BVH_THREADS = counts.  This is the
 * only function of the library that reads the environment. */
void pt_scene_options_from_env(pt_scene_options* options);

/* ---- scene.rs:140-193 + bvh.rs:17-21: build the device-resident scene on HIP device `device_id`.
 * pt_scene_create_ex takes the scheduling options of this scene (NULL = all defaults) and never reads the
 * environment; pt_scene_create is pt_scene_create_ex with pt_scene_options_from_env's options.
 * Size limits (PT_ERR_UNSUPPORTED beyond): 2^28 triangles (leaf references), 2^25 four-wide nodes of a deep tree
 * (the traversal addresses a node through a 32-bit byte offset) - about 10^8 triangles at two per leaf. */
int pt_scene_create_ex(const pt_scene_desc* desc, int device_id, const pt_scene_options* options, pt_scene** out);
int pt_scene_create(const pt_scene_desc* desc, int device_id, pt_scene** out);
void pt_scene_destroy(pt_scene* scene);
/* introspection (host copies owned by the scene, valid until destroy).  The boxes are the ones the kernels
 * test: a caller-supplied tree comes back with its boxes widened by the margin of the f32 slab test. */
int pt_scene_get_bvh(const pt_scene* scene, const pt_bvh_node** nodes, uint32_t* n_nodes,
                     const uint32_t** tri_order, uint32_t* max_depth);

/* ---- bvh.rs:17-21 `e_BVH::unanimated(8, ..)`: host-side binned-SAH build (the external crate
 * is not vendored).  nodes_out has room for `cap_nodes` (>= max(1, n_tris)) nodes,
 * order_out for n_tris ids.  max_leaf in 1..8.  n_threads: host threads of the build, 0 = the hardware's (at most
 * 16); the tree is the same array whatever the count. */
int pt_bvh_build(const float* verts, uint32_t n_verts, const uint32_t* tris, uint32_t n_tris,
                 uint32_t max_leaf, pt_bvh_node* nodes_out, uint32_t cap_nodes,
                 uint32_t* n_nodes_out, uint32_t* order_out, uint32_t* max_depth_out, uint32_t n_threads);

/* ---- bvh.rs:17-21 when the host keeps its own builder (north_star: "BVH build in bvh.rs"): checks
 * everything the traversal kernels assume of a caller-supplied tree before pt_scene_create takes it
 * (child indices in range, each inner node reached once, leaf runs inside [0, n_tris) and covering
 * every triangle exactly once, tri_order a permutation, depth <= 32 levels: a shallow tree (depth <= 8) is
 * walked as given with an 8-entry LDS stack, a deeper one through its four-wide collapse, whose stack need
 * the collapse bounds by 16 LDS entries + 48 private ones — a tree deeper than 32 levels is
 * PT_ERR_UNSUPPORTED, never silently truncated).  Host only, no device. */
int pt_bvh_validate(const pt_bvh_node* nodes, uint32_t n_nodes, const uint32_t* tri_order, uint32_t n_tris,
                    uint32_t* max_depth_out);

/* ---- scene.rs:61 Scene::trace_closest on an explicit ray set (the extend kernel alone).
 * tri_id[i] = global triangle id (desc order), PT_PRIM_BASE + k for primitive k, or PT_MISS; t[i] = hit distance (unchanged t_max on a miss).
 * Tie rule for bit-identical t: lowest triangle id. */
int pt_trace_closest(pt_scene* scene, const pt_ray* rays, size_t n, uint32_t* tri_id, double* t);

/* ---- main.rs:402-427 + scene.rs:37-57: the generate kernel alone.  Writes the primary rays of
 * frame `sample` for the tile pixels in row-major tile order (rays[(y-y0)*(x1-x0) + (x-x0)]). */
int pt_generate_rays(pt_scene* scene, const pt_render_params* params, uint64_t sample,
                     pt_ray* rays);

/* ---- material.rs:60-66 `_Material::sample_brdf` + get_emission at given hits (the shade
 * kernel's sampling step alone).  For ray i that hit triangle tri_id[i] at distance t[i], with
 * RNG counters (pixel[i], sample, bounce): f[i], wi[i], pdf[i], emission[i], position[i]. */
int pt_sample_brdf(pt_scene* scene, const pt_ray* rays, const uint32_t* tri_id, const double* t,
                   const uint32_t* pixel, size_t n, uint64_t sample, uint32_t bounce,
                   uint64_t seed, uint32_t flags, uint32_t width, uint32_t height,
                   double* f /*[n][3]*/, double* wi /*[n][3]*/, double* pdf /*[n]*/,
                   double* emission /*[n][3]*/, double* position /*[n][3]*/);

/* ---- mipmap.rs:117-141 MipMap::sample_mipmap on texture `tex` of the scene. */
int pt_texture_lookup(pt_scene* scene, uint32_t tex, const double* uv /*[n][2]*/,
                      const double* duvdx /*[n][2]*/, const double* duvdy /*[n][2]*/, size_t n,
                      double* rgb /*[n][3]*/);

/* ---- main.rs:400-436: sample_count frames of the tile, accum[(y*width+x)*3+c] += radiance
 * (f64, the `pixels[i] += next[i]` of main.rs:434-436, frames added in index order).
 * pt_render: accum is a host buffer of width*height*3 doubles.
 * pt_render_device: accum is a device buffer on the scene's device; work is enqueued on `stream`
 * (a hipStream_t, NULL = default stream) and the call returns after enqueueing (with profiling on,
 * the event pairs are read back by pt_get_stats, not here).  */
int pt_render(pt_scene* scene, const pt_render_params* params, double* accum_rgb);
int pt_render_device(pt_scene* scene, const pt_render_params* params, double* accum_rgb_dev,
                     void* stream);

/* ---- main.rs:396-400, 434-467: the frame loop's running sum `pixels`, kept on the device.  The reference's
 * window loop adds one frame per iteration and shows clamp(sum / iterations) each time; with pt_render
 * that is a 50 MB upload and download of the f64 sums per frame at 1080p.  A pt_accum holds the sums in
 * HBM next to the scene: pt_accum_render adds frames (params->width/height must be the handle's; it only
 * enqueues, on the handle's own stream), pt_accum_resolve_* runs main.rs:456-467 on the device and
 * brings back the 8-bit image alone through page-locked staging (rgb8: 3 bytes per pixel, the PPM
 * writer's order; u32: 0x00RRGGBB as get_raw_pixels_u32 packs it for the window), pt_accum_download /
 * pt_accum_upload move the f64 sums for checkpoints (upload NULL = clear).  pt_accum_device exposes the
 * device pointer and stream for callers that mix in their own device work.  Destroy before the scene. */
typedef struct pt_accum pt_accum;
int pt_accum_create(pt_scene* scene, uint32_t width, uint32_t height, pt_accum** out);
void pt_accum_destroy(pt_accum* acc);
int pt_accum_render(pt_accum* acc, const pt_render_params* params);
int pt_accum_resolve_rgb8(pt_accum* acc, uint64_t iterations, uint8_t* rgb8 /* [w*h*3] host */);
int pt_accum_resolve_u32(pt_accum* acc, uint64_t iterations, uint32_t* packed /* [w*h] host */);
int pt_accum_download(pt_accum* acc, double* accum_rgb /* [w*h*3] host */);
int pt_accum_upload(pt_accum* acc, const double* accum_rgb /* [w*h*3] host, or NULL to clear */);
int pt_accum_device(pt_accum* acc, double** sums_dev, void** stream);

/* ---- No counterpart in the reference.  pt_render_device for callers that cannot allocate device
 * memory: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same
 * arguments, same result bit for bit; the whole workspace of the call (path queues, hit records,
 * counters, per-sample radiance, frame jitters, the table of pixel coordinates) is one fixed array of PT_NOALLOC_SCRATCH_BYTES that
 * lives in the library's code object - every GPU the library runs on holds it from load until the
 * process ends - so the call makes no hipMalloc / hipFree / event or stream creation.  Frames per
 * batch are cut to what fits; PT_ERR_UNSUPPORTED, with the sizes in pt_last_error(), if a single
 * frame of the tile does not fit: about 655 000 tile pixels inside the scene's screen bounds per call (e.g.
 * 1920 x 340; the table takes 8 bytes per column and row of the image off the array), 393 000 with
 * PT_FLAG_NEE, 327 680 with PT_FLAG_PROPAGATE_DIFFS on a textured scene (262 144 with both).  No kernel timing on this
 * path (pt_set_profiling level 1 is ignored; the counters of level 2 work).  Every call on a GPU
 * shares that GPU's array.  Calls from two streams, two scenes or two host threads are ORDERED by the
 * library (a per-device mutex around the enqueueing and a per-device event that every call's stream
 * waits for and re-records), so they run one after the other instead of racing; only a stream under
 * graph capture is left out of that ordering (it cannot wait for an outside event): the caller orders
 * captured calls.  The first call on a device looks the array's address up and creates that event (no
 * device memory is allocated, but make it outside graph capture). */
#define PT_NOALLOC_SCRATCH_BYTES (128u << 20)
int pt_render_device_noalloc(pt_scene* scene, const pt_render_params* params, double* accum_rgb_dev,
                             void* stream);

/* ---- main.rs:456-467 get_raw_pixels_u32 / mipmap.rs:21-23 f64_to_8bit_color:
 * rgb8[i] = clamp(accum[i] / iterations) * 255 + 0.5 truncated; no gamma. (host) */
int pt_resolve_rgb8(const double* accum_rgb, size_t n_pixels, uint64_t iterations, uint8_t* rgb8);
/* ---- mipmap.rs:25-42 Image::save_ppm: ASCII P3, one pixel per line. */
int pt_write_ppm(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb8);

/* ---- checkpoint / resume of a progressive render.  The reference keeps its
 * running sum `pixels` and the frame counter `iterations` only in memory (main.rs:396-400,434-436);
 * a 4096-frame run that dies starts over.  A checkpoint is that pair on disk: the f64 sums, the
 * number of frames they hold (= the sample_begin to resume from) and the seed they were drawn
 * with; because the RNG is keyed by (seed, frame), resuming continues the very same sequence and
 * the final image is bit-identical to an uninterrupted run.  File: 48-byte header (magic
 * "PTACCUM1", width, height, frames, seed, FNV-1a-64 of the payload) + width*height*3 doubles,
 * little endian.  pt_checkpoint_save writes to `path`.tmp and renames, so a crash mid-write leaves
 * the previous checkpoint intact.  pt_checkpoint_load: accum_rgb NULL = header only; otherwise it
 * must hold cap_pixels >= width*height pixels; a size, magic or checksum mismatch is PT_ERR_IO. */
typedef struct pt_checkpoint_info {
    uint32_t width, height;
    uint64_t frames; /* frames accumulated so far */
    uint64_t seed;
} pt_checkpoint_info;
int pt_checkpoint_save(const char* path, const pt_checkpoint_info* info, const double* accum_rgb);
int pt_checkpoint_load(const char* path, pt_checkpoint_info* info, double* accum_rgb, size_t cap_pixels);

/* profiling: level 0 = off, 1 = HIP-event time per kernel launch, 2 = also count BVH nodes
 * visited / triangles tested (instrumented extend kernel; slower). */
int pt_set_profiling(pt_scene* scene, int level);
int pt_get_stats(pt_scene* scene, pt_stats* out);
int pt_reset_stats(pt_scene* scene);

/* ---- obj.rs:13-107 load_obj with tobj 0.1.6 semantics, host side.
 * Returns an owned model; pt_model_desc() exposes it as a pt_scene_desc (camera/background are
 * zeroed: set them before pt_scene_create, scene.rs:184-186). */
typedef struct pt_model pt_model;
int pt_load_obj(const char* obj_path, uint32_t flags, pt_model** out);
enum {
    PT_LOAD_GLOSSY = 1u << 0,   /* wire illum 2 -> Glossy(Ns) as the commented obj.rs:30-53 intends */
    PT_LOAD_TEXTURES = 1u << 1, /* wire map_Kd -> MipMap (obj.rs:41-47) */
    PT_LOAD_SPECULAR = 1u << 2, /* wire illum 3 -> Mirror, illum 4 -> Transparent(Ni, 1.5 if absent)
                                 * (IlluminationModel::from, material.rs:157-170) */
    PT_LOAD_GGX = 1u << 3       /* with PT_LOAD_GLOSSY: illum 2 -> Glossy over GGX instead of Beckmann */
};
const pt_scene_desc* pt_model_desc(const pt_model* model);
uint32_t pt_model_count(const pt_model* model);               /* tobj models (= reference Geometry count) */
const char* pt_model_name(const pt_model* model, uint32_t i);
uint32_t pt_model_first_tri(const pt_model* model, uint32_t i); /* first global triangle of model i */
void pt_model_destroy(pt_model* model);

/* ---- mipmap.rs:59-83 MipMap::load decode step for PNG (8-bit RGB/RGBA/gray, non-interlaced). */
int pt_decode_png(const char* path, uint32_t* width, uint32_t* height, uint8_t** rgb8_out);
void pt_free(void* p);

/* counter-based RNG that replaces rand::random::<f64>(): Philox4x32-10,
 * key = seed, counter = (pixel, sample_lo, bounce | sample_hi << 8 .., stream). Exposed for tests. */
void pt_rng_uniform2(uint64_t seed, uint32_t pixel, uint64_t sample, uint32_t bounce,
                     uint32_t stream, double out[2]);

const char* pt_last_error(void);
const char* pt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PT_H */
