// pt_kernels.hip — gfx950 (CDNA4, wave64) wavefront path-tracing kernels.
//
//   generate   main.rs:409-427 pixel mapping + scene.rs:37-57 Camera::ray          -> ray queue
//   extend     scene.rs:61 trace_closest: BVH traversal with an LDS-resident per-lane stack and
//              the f64 ray/triangle test of triangle_mesh.rs:101-126                  -> hit records
//   shade      main.rs:303-314: emission, _Material::sample_brdf (material.rs:37-58,
//              microfacet.rs:52-93), mip-mapped texture lookup (mipmap.rs:100-141), throughput
//              update, next ray; survivors are compacted with wave64 ballot + mbcnt prefix sums
//   shadow     any-hit visibility for next-event estimation (lights.rs:51-60 shadow ray)
//   accumulate main.rs:434-436 pixels[i] += next[i], frames added in index order
//
// Everything is f64 like the reference (vec3.rs:6) and this file is compiled with
// -ffp-contract=off: the reference never fuses a*b+c, and primary-ray hit ids / distances have
// to be bit-identical to the CPU restatement.  Vertices and boxes are f32 in memory (the
// reference's own storage precision at those points: obj.rs:25, bvh.rs:98-101).
// The device code is one translation unit split over pt_math.hip.h (vectors, RNG, camera),
// pt_traverse.hip.h (extend, shadow) and pt_shade.hip.h (shade, path, accumulate); this file holds
// the parity-only kernels and the launchers.
#include "pt_math.hip.h"
#include "pt_traverse.hip.h"
#include "pt_shade.hip.h"

namespace ptd {

// ------------------------------------------------------------------------------------------------
// parity-only kernels
__global__ void k_leaf_to_global(Scene sc, uint32_t* tri, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t leaf = tri[i];
    if (leaf < PT_PRIM_BASE) tri[i] = (uint32_t)__float_as_int(sc.tris[(size_t)leaf * 3].w);  // primitives keep PT_PRIM_BASE + k
}

__global__ void k_sample_brdf(Scene sc, Frame fr, uint32_t bounce, const double* __restrict__ rays,
                              const uint32_t* __restrict__ tri_global, const double* __restrict__ tt,
                              const uint32_t* __restrict__ pixel_in, uint32_t n, double* f, double* wi, double* pdf,
                              double* emission, double* position_out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* r = rays + (size_t)i * 8;
    D3 o = mk(r[0], r[1], r[2]), d = mk(r[3], r[4], r[5]);
    uint32_t leaf = tri_global[i] >= PT_PRIM_BASE ? tri_global[i] : sc.leaf_of[tri_global[i]];
    uint32_t mat;
    D3 position, normal;
    hit_geometry(sc, leaf, o, d, tt[i], mat, position, normal);
    const Material& m = sc.mats[mat];
    uint32_t pixel = pixel_in[i];
    D3 diffuse = mk(0., 0., 0.);
    if (m.kind == PT_MAT_LAMBERT) diffuse = diffuse_of(sc, m, leaf, o, d, position, normal, fr, bounce == 0, pixel % fr.width, pixel / fr.width, 0);
    double xi1, xi2;
    uniform2(fr.seed, pixel, fr.sample_base, bounce, STREAM_BRDF, xi1, xi2);
    Brdf bs;
    int lobe;
    if (m.kind == PT_MAT_MIRROR || m.kind == PT_MAT_TRANSPARENT) bs = {mk(1., 1., 1.), specular_sample(m, d, normal, xi1, lobe), -1.};  // pdf < 0: specular
    else if (is_glossy(m.kind)) bs = glossy_sample(m.width, ld3(m.color), d, normal, xi1, xi2, m.kind == PT_MAT_GLOSSY_GGX);
    else {
        // the code the renders run (shade_vertex): the direction from lambertian_dir over the triangle's stored
        // basis; f = rho / pi and pdf = (n . wi) / pi as material.rs:37-58 (the render's throughput factor is the
        // closed form rho; the test checks f |wi . n| / pdf against it)
        const TriBasis* basis = (sc.tri_basis && leaf < PT_PRIM_BASE) ? sc.tri_basis + leaf : nullptr;
        const D3 w = lambertian_dir(normal, basis, xi1, xi2);
        bs = {diffuse * PT_FRAC_1_PI, w, dot(normal, w) * PT_FRAC_1_PI};
    }
    D3 le = m.kind != PT_MAT_LAMBERT ? mk(0., 0., 0.) : ld3(m.emission);
    f[3 * i] = bs.f.x; f[3 * i + 1] = bs.f.y; f[3 * i + 2] = bs.f.z;
    wi[3 * i] = bs.wi.x; wi[3 * i + 1] = bs.wi.y; wi[3 * i + 2] = bs.wi.z;
    pdf[i] = bs.pdf;
    emission[3 * i] = le.x; emission[3 * i + 1] = le.y; emission[3 * i + 2] = le.z;
    position_out[3 * i] = position.x; position_out[3 * i + 1] = position.y; position_out[3 * i + 2] = position.z;
}

__global__ void k_texture_lookup(Scene sc, uint32_t tex, const double* __restrict__ uv,
                                 const double* __restrict__ duvdx, const double* __restrict__ duvdy, uint32_t n,
                                 double* rgb) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    D3 c = sample_mipmap(sc.texs[tex], uv[2 * i], uv[2 * i + 1], duvdx[2 * i], duvdx[2 * i + 1], duvdy[2 * i],
                         duvdy[2 * i + 1]);
    rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
}

// main.rs:456-467 get_raw_pixels_u32 / mipmap.rs:21-23 f64_to_8bit_color on the device-resident sums: the
// arithmetic of the host's pt_resolve_rgb8, value for value (clamp(sum / iterations) * 255 + 0.5,
// truncated; `as u8` saturates and maps NaN to 0).  One lane per pixel; rgb8 (3 bytes per pixel, the
// PPM writer's order) or packed 0x00RRGGBB words (the window buffer), whichever pointer is given.
__global__ __launch_bounds__(kBlock) void k_resolve(const double* __restrict__ accum, size_t n_pixels, double denom,
                                                    uint8_t* __restrict__ rgb8, uint32_t* __restrict__ packed) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_pixels) return;
    uint32_t q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double x = accum[i * 3 + c] * denom;
        x = x < 0. ? 0. : (x > 1. ? 1. : x);
        const double v = x * 255. + 0.5;
        q[c] = !(v > 0.) ? 0u : (v >= 255. ? 255u : (uint32_t)v);
    }
    if (rgb8) { rgb8[i * 3] = (uint8_t)q[0]; rgb8[i * 3 + 1] = (uint8_t)q[1]; rgb8[i * 3 + 2] = (uint8_t)q[2]; }
    if (packed) packed[i] = (q[0] << 16) | (q[1] << 8) | q[2];
}

// ------------------------------------------------------------------------------------------------
// launchers
static inline uint32_t blocks_for(uint32_t n) { return (n + kBlock - 1) / kBlock; }

void launch_pixel_table(uint32_t width, uint32_t height, double* out, hipStream_t st) {
    hipLaunchKernelGGL(k_pixel_table, dim3(blocks_for(width + height)), dim3(kBlock), 0, st, width, height, out);
}
void launch_jitter(uint64_t seed, uint64_t sample_begin, uint32_t n, double psx, double psy, double* out,
                   hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_jitter, dim3(blocks_for(n)), dim3(kBlock), 0, st, seed, sample_begin, n, psx, psy, out);
}
void launch_resolve(const double* accum, size_t n_pixels, double denom, uint8_t* rgb8, uint32_t* packed, hipStream_t st) {
    if (!n_pixels) return;
    hipLaunchKernelGGL(k_resolve, dim3((uint32_t)((n_pixels + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, accum, n_pixels, denom, rgb8, packed);
}
void launch_generate_rays(const Scene& sc, const Frame& fr, double* rays_out, hipStream_t st) {
    if (!fr.n_tile_pixels) return;
    hipLaunchKernelGGL(k_generate_rays, dim3(blocks_for(fr.n_tile_pixels)), dim3(kBlock), 0, st, sc, fr, rays_out);
}
// Two traversal forms, by the scene's tree depth (pt_device.h "Traversal stacks"): the two-child tree with
// an 8-entry LDS stack up to depth 8, the four-wide tree with 16 LDS entries + private spill beyond.
// workgroups of a persistent launch: enough one-wave workgroups to fill every SIMD at the occupancy
// the traversal kernels reach (<= 6 waves), whatever the queue holds
static uint32_t persistent_blocks(const Scene& sc) {
    return std::max(1u, sc.n_cu) * 4u * 6u / (kTravBlock / 64);
}
template <bool STATS, bool PRIMARY, bool VOTE>
static void launch_extend_v(const Scene& sc, const Frame& fr, Queue q, const double* tmin, const double* tmax, Live live,
                            uint32_t wr, Hits hits, Counters* ctr, hipStream_t st) {
    const uint32_t br = wr * (kTravBlock / 64);  // rays per workgroup
    dim3 g(live.counts ? (live.n + br - 1) / br : trav_blocks(live.n, br)), b(kTravBlock);
    if (VOTE && live.cursors) g.x = min(g.x, persistent_blocks(sc));
    if (sc.bvh_depth <= 8) hipLaunchKernelGGL((k_extend<8, STATS, PRIMARY, VOTE>), g, b, 0, st, sc, fr, q, tmin, tmax, live, wr, hits, ctr);
    else hipLaunchKernelGGL((k_extend<kWideLds, STATS, PRIMARY, VOTE>), g, b, 0, st, sc, fr, q, tmin, tmax, live, wr, hits, ctr);
}
template <bool STATS, bool PRIMARY>
static void launch_extend_t(const Scene& sc, const Frame& fr, Queue q, const double* tmin, const double* tmax, Live live,
                            Hits hits, Counters* ctr, hipStream_t st) {
    if (PRIMARY ? sc.vote_primary : sc.vote_secondary)
        launch_extend_v<STATS, PRIMARY, true>(sc, fr, q, tmin, tmax, live, sc.wave_rays, hits, ctr, st);
    else launch_extend_v<STATS, PRIMARY, false>(sc, fr, q, tmin, tmax, live, 64, hits, ctr, st);
}
void launch_extend(const Scene& sc, Queue q, const double* tmin, const double* tmax, Live live, Hits hits,
                   Counters* ctr, bool stats, hipStream_t st) {
    if (!live.n) return;
    Frame fr{};
    if (stats) launch_extend_t<true, false>(sc, fr, q, tmin, tmax, live, hits, ctr, st);
    else launch_extend_t<false, false>(sc, fr, q, tmin, tmax, live, hits, ctr, st);
}
void launch_extend_primary(const Scene& sc, const Frame& fr, Hits hits, Counters* ctr, bool stats, hipStream_t st) {
    uint32_t n = fr.n_tile_pixels * fr.n_samples;
    if (!n) return;
    Queue q{};
    Live live{nullptr, 0, n};
    if (stats) launch_extend_t<true, true>(sc, fr, q, nullptr, nullptr, live, hits, ctr, st);
    else launch_extend_t<false, true>(sc, fr, q, nullptr, nullptr, live, hits, ctr, st);
}
template <bool FIRST, bool LAST>
static void launch_shade_t(bool nee, dim3 g, dim3 b, hipStream_t st, const Scene& sc, const Frame& fr, uint32_t bounce,
                           Queue in, Hits hits, Live live, Queue out, Push out_push, ShadowQueue sq, Push sq_push,
                           Sample3 ls) {
#define PT_SHADE(N, M, D) hipLaunchKernelGGL((k_shade<FIRST, LAST, N, M, D>), g, b, 0, st, sc, fr, bounce, in, hits, live, out, out_push, sq, sq_push, ls)
    const bool diffs = (fr.flags & PT_FLAG_PROPAGATE_DIFFS) && (sc.mat_features & 1u);  // the host sets the flag only where it has an effect
    switch (sc.mat_features & 3u) {
        case 0: if (nee) PT_SHADE(true, 0, false); else PT_SHADE(false, 0, false); break;
        case 1:
            if (diffs) { if (nee) PT_SHADE(true, 1, true); else PT_SHADE(false, 1, true); }
            else { if (nee) PT_SHADE(true, 1, false); else PT_SHADE(false, 1, false); }
            break;
        case 2: if (nee) PT_SHADE(true, 2, false); else PT_SHADE(false, 2, false); break;
        default:
            if (diffs) { if (nee) PT_SHADE(true, 3, true); else PT_SHADE(false, 3, true); }
            else { if (nee) PT_SHADE(true, 3, false); else PT_SHADE(false, 3, false); }
    }
#undef PT_SHADE
}
void launch_shade(const Scene& sc, const Frame& fr, uint32_t bounce, Queue in, Hits hits, Live live, Queue out,
                  Push out_push, ShadowQueue sq, Push sq_push, Sample3 ls, hipStream_t st) {
    if (!live.n) return;
    bool nee = samples_lights(sc, fr.flags);
    bool first = bounce == 0, last = bounce >= fr.max_bounces;
    dim3 g(blocks_for(live.n)), b(kBlock);
    if (last && !first && live.counts && sc.cull_last) {  // the queue after the cull: a few percent full
        live.sparse = 1;
        g.x = std::min(g.x, sc.n_cu * 8u);
    }
    if (first && last) launch_shade_t<true, true>(nee, g, b, st, sc, fr, bounce, in, hits, live, out, out_push, sq, sq_push, ls);
    else if (first) launch_shade_t<true, false>(nee, g, b, st, sc, fr, bounce, in, hits, live, out, out_push, sq, sq_push, ls);
    else if (last) launch_shade_t<false, true>(nee, g, b, st, sc, fr, bounce, in, hits, live, out, out_push, sq, sq_push, ls);
    else launch_shade_t<false, false>(nee, g, b, st, sc, fr, bounce, in, hits, live, out, out_push, sq, sq_push, ls);
}
template <int DEPTH, bool FIRST, bool LAST>
static void launch_path_t(bool nee, dim3 g, dim3 b, hipStream_t st, const Scene& sc, const Frame& fr, uint32_t bounce,
                          Queue in, Live live, Queue out, Push out_push, ShadowQueue sq, Push sq_push, Sample3 ls,
                          Counters* ctr, Push mid_push) {
#define PT_PATHK(N, M) hipLaunchKernelGGL((k_path<DEPTH, FIRST, LAST, N, M>), g, b, 0, st, sc, fr, bounce, in, live, out, out_push, sq, sq_push, ls, ctr, mid_push)
#define PT_PATHL(N, M) hipLaunchKernelGGL((k_path<8, FIRST, false, N, M, true>), g, b, 0, st, sc, fr, bounce, in, live, out, out_push, sq, sq_push, ls, ctr, mid_push)
#define PT_PATH2(N, M) hipLaunchKernelGGL((k_path<8, true, false, N, M, true, true>), g, b, 0, st, sc, fr, bounce, in, live, out, out_push, sq, sq_push, ls, ctr, mid_push)
    if (DEPTH == 8 && FIRST && !LAST && sc.lds_scene && mid_push.counts) {  // bounces 0 and 1 in one go
        switch (sc.mat_features & 3u) {
            case 0: if (nee) PT_PATH2(true, 0); else PT_PATH2(false, 0); break;
            case 1: if (nee) PT_PATH2(true, 1); else PT_PATH2(false, 1); break;
            case 2: if (nee) PT_PATH2(true, 2); else PT_PATH2(false, 2); break;
            default: if (nee) PT_PATH2(true, 3); else PT_PATH2(false, 3);
        }
        return;
    }
#undef PT_PATH2
    if (DEPTH == 8 && !LAST && sc.lds_scene) {  // tiny scene held in LDS
        switch (sc.mat_features & 3u) {
            case 0: if (nee) PT_PATHL(true, 0); else PT_PATHL(false, 0); break;
            case 1: if (nee) PT_PATHL(true, 1); else PT_PATHL(false, 1); break;
            case 2: if (nee) PT_PATHL(true, 2); else PT_PATHL(false, 2); break;
            default: if (nee) PT_PATHL(true, 3); else PT_PATHL(false, 3);
        }
        return;
    }
#undef PT_PATHL
    switch (sc.mat_features & 3u) {
        case 0: if (nee) PT_PATHK(true, 0); else PT_PATHK(false, 0); break;
        case 1: if (nee) PT_PATHK(true, 1); else PT_PATHK(false, 1); break;
        case 2: if (nee) PT_PATHK(true, 2); else PT_PATHK(false, 2); break;
        default: if (nee) PT_PATHK(true, 3); else PT_PATHK(false, 3);
    }
#undef PT_PATHK
}
// The fused form exists where a bounce is traced one ray per lane: every bounce of a shallow tree
// (the 8-entry stack instantiation) and the primary bounce of any tree.
bool path_fusable(const Scene& sc, uint32_t bounce) {
    if (!sc.fuse) return false;
    return bounce == 0 ? !sc.vote_primary : (sc.bvh_depth <= 8 && !sc.vote_secondary);
}
// The primary launch may take the path's second vertex along (k_path F01): a scene that lives in LDS, whose
// primary rays nearly all hit (Scene::fuse01), at least two bounces, light samples (if any) traced in the kernel.
bool path_two_bounces(const Scene& sc, const Frame& fr) {
    return sc.fuse01 && sc.lds_scene && sc.bvh_depth <= 8 && fr.max_bounces >= 2 && path_fusable(sc, 0) && path_fusable(sc, 1) &&
           (!samples_lights(sc, fr.flags) || sc.fuse_shadow);
}
bool launch_path(const Scene& sc, const Frame& fr, uint32_t bounce, Queue in, Live live, Queue out, Push out_push,
                 ShadowQueue sq, Push sq_push, Sample3 ls, Counters* ctr, hipStream_t st, Push mid_push) {
    if (!path_fusable(sc, bounce)) return false;
    if (!live.n) return true;
    bool nee = samples_lights(sc, fr.flags);
    bool first = bounce == 0, last = bounce >= fr.max_bounces;
    // the variant launch_path_t picks: the scene-in-LDS kernel has its own workgroup size
    const bool lds = sc.bvh_depth <= 8 && !last && sc.lds_scene;
    const uint32_t kPathBlock = lds ? PathBlock<8, true>::value : (sc.bvh_depth <= 8 ? PathBlock<8>::value : PathBlock<kWideLds>::value);
    dim3 g(live.counts ? (live.n + kPathBlock - 1) / kPathBlock : trav_blocks(live.n, kPathBlock)), b(kPathBlock);
    if (last && !first && live.counts && sc.cull_last) {  // the queue after the cull: a few percent full
        live.sparse = 1;
        g.x = std::min(g.x, sc.n_cu * 8u);
    }
    #define PT_PATH(D, F, L) launch_path_t<D, F, L>(nee, g, b, st, sc, fr, bounce, in, live, out, out_push, sq, sq_push, ls, ctr, mid_push)
    if (sc.bvh_depth <= 8) {
        if (first && last) PT_PATH(8, true, true);
        else if (first) PT_PATH(8, true, false);
        else if (last) PT_PATH(8, false, true);
        else PT_PATH(8, false, false);
    } else {  // deeper trees: primary bounce only (path_fusable)
        if (last) PT_PATH(kWideLds, true, true); else PT_PATH(kWideLds, true, false);
    }
    #undef PT_PATH
    return true;
}
template <bool VOTE>
static void launch_shadow_v(const Scene& sc, ShadowQueue sq, Live live, uint32_t wr, Sample3 ls, Counters* ctr, hipStream_t st) {
    const uint32_t br = wr * (kTravBlock / 64);
    dim3 g((live.n + br - 1) / br), b(kTravBlock);
    if (VOTE && live.cursors) g.x = min(g.x, persistent_blocks(sc));
    if (sc.bvh_depth <= 8) hipLaunchKernelGGL((k_shadow<8, VOTE>), g, b, 0, st, sc, sq, live, wr, ls, ctr);
    else hipLaunchKernelGGL((k_shadow<kWideLds, VOTE>), g, b, 0, st, sc, sq, live, wr, ls, ctr);
}
void launch_shadow(const Scene& sc, ShadowQueue sq, Live live, Sample3 ls, Counters* ctr, hipStream_t st) {
    if (!live.n) return;
    if (sc.vote_shadow) launch_shadow_v<true>(sc, sq, live, sc.wave_rays, ls, ctr, st);
    else launch_shadow_v<false>(sc, sq, live, 64, ls, ctr, st);
}
__global__ void k_count_shadow_rays(const uint32_t* seg_counts, Counters* ctr, bool rays) {
    unsigned long long n = seg_counts[threadIdx.x * kCtrStride];  // kSegs threads
#pragma unroll
    for (int d = 32; d; d >>= 1) n += __shfl_down(n, d);
    if (threadIdx.x == 0 && n) {
        atomicAdd(rays ? &ctr->rays : &ctr->shadow_rays, n);
        if (rays) atomicAdd(&ctr->unqueued, n);  // the rays of bounce 1 of a two-bounce launch: traced, never a queue record
    }
}
void launch_count_shadow_rays(const uint32_t* seg_counts, Counters* ctr, hipStream_t st) {
    hipLaunchKernelGGL(k_count_shadow_rays, dim3(1), dim3(kSegs), 0, st, seg_counts, ctr, false);
}
void launch_count_rays(const uint32_t* seg_counts, Counters* ctr, hipStream_t st) {
    hipLaunchKernelGGL(k_count_shadow_rays, dim3(1), dim3(kSegs), 0, st, seg_counts, ctr, true);
}
void launch_accumulate(const Frame& fr, Sample3 ls, const unsigned long long* lit1, uint32_t split, double* accum,
                       Counters* ctr, hipStream_t st) {
    if (!fr.n_tile_pixels) return;
    hipLaunchKernelGGL(k_accumulate, dim3(blocks_for(fr.n_tile_pixels)), dim3(kBlock), 0, st, fr, ls, lit1, split, accum, ctr);
}
void launch_build_eye_tables(const Scene& sc, EyeNode* nodes, EyeTri* tris, hipStream_t st) {
    uint32_t n = std::max(sc.n_nodes, sc.n_tris);
    if (!n) return;
    hipLaunchKernelGGL(k_build_eye_tables, dim3(blocks_for(n)), dim3(kBlock), 0, st, sc, nodes, tris);
}
void launch_build_tri_edges(const Scene& sc, EyeTri* tris, hipStream_t st) {
    if (!sc.n_tris) return;
    hipLaunchKernelGGL(k_build_tri_edges, dim3((sc.n_tris + 255) / 256), dim3(256), 0, st, sc, tris);
}
__global__ void k_plane_major(const float* __restrict__ src, float* __restrict__ dst, uint32_t n_floats) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_floats) dst[(k & ~15u) + lds_node_float(k & 15u)] = src[k];
}
void launch_plane_major(const float* src, float* dst, uint32_t n_nodes, hipStream_t st) {
    if (!n_nodes) return;
    hipLaunchKernelGGL(k_plane_major, dim3((n_nodes * 16 + 255) / 256), dim3(256), 0, st, src, dst, n_nodes * 16);
}
void launch_build_emitters(const Scene& sc, const uint32_t* leaf, uint32_t n, EmitTri* out, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_build_emitters, dim3(blocks_for(n)), dim3(kBlock), 0, st, sc, leaf, n, out);
}
void launch_build_tri_shade(const Scene& sc, TriShade* out, TriBasis* basis, hipStream_t st) {
    if (!sc.n_tris) return;
    hipLaunchKernelGGL(k_build_tri_shade, dim3(blocks_for(sc.n_tris)), dim3(kBlock), 0, st, sc, out, basis);
}
void launch_leaf_to_global(const Scene& sc, uint32_t* tri, uint32_t n, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_leaf_to_global, dim3(blocks_for(n)), dim3(kBlock), 0, st, sc, tri, n);
}
void launch_sample_brdf(const Scene& sc, const Frame& fr, uint32_t bounce, const double* rays, const uint32_t* tri,
                        const double* t, const uint32_t* pixel, uint32_t n, double* f, double* wi, double* pdf,
                        double* emission, double* position, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_sample_brdf, dim3(blocks_for(n)), dim3(kBlock), 0, st, sc, fr, bounce, rays, tri, t, pixel, n, f, wi, pdf, emission, position);
}
void launch_texture_lookup(const Scene& sc, uint32_t tex, const double* uv, const double* duvdx, const double* duvdy,
                           uint32_t n, double* rgb, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_texture_lookup, dim3(blocks_for(n)), dim3(kBlock), 0, st, sc, tex, uv, duvdx, duvdy, n, rgb);
}

}  // namespace ptd

#ifdef PT_PATH_PROFILE  // diagnostic build: scripts/diag/path_profile.py
extern "C" int pt_debug_path_cut(int cut) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(ptd::g_path_cut), &cut, sizeof(int)) == hipSuccess ? 0 : 2;
}
#endif
#ifdef PT_VOTE_PROFILE  // diagnostic build: scripts/diag/vote_profile.py
extern "C" int pt_debug_vote_profile(unsigned long long* out, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ptd::g_vote_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 2;
    if (reset) {
        const unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ptd::g_vote_prof), z, sizeof(z)) != hipSuccess) return 3;
    }
    return 0;
}
#endif
extern "C" void pt_rng_uniform2(uint64_t seed, uint32_t pixel, uint64_t sample, uint32_t bounce, uint32_t stream,
                                double out[2]) {
    ptd::uniform2(seed, pixel, sample, bounce, stream, out[0], out[1]);
}
