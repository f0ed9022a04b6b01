// pt_traverse.hip.h — extend: the triangle test of triangle_mesh.rs:101-126, slab test, BVH traversal (one ray per lane, vote-scheduled ranges, eye-relative tables), k_extend, k_shadow
// Part of the single translation unit pt_kernels.hip (device code is inlined across these files);
// included from there, in order: pt_math.hip.h, pt_traverse.hip.h, pt_shade.hip.h.
#pragma once
#include "pt_device.h"

#include "../../include/pt.h"

namespace ptd {

// ------------------------------------------------------------------------------------------------
// triangle_mesh.rs:101-126.  The conditions are evaluated in a different order than the reference
// (barycentrics before the distance, so most candidates cost one division instead of two); every
// condition is computed with the reference's arithmetic and they have no side effects, so the
// accept/reject decision and the accepted t are identical.
struct TriV {
    D3 v0, v1, v2;
};
__device__ __forceinline__ TriV widen(float4 a, float4 b, float4 c) {
    return {mk((double)a.x, (double)a.y, (double)a.z), mk((double)b.x, (double)b.y, (double)b.z),
            mk((double)c.x, (double)c.y, (double)c.z)};
}
__device__ __forceinline__ TriV load_tri(const Scene& sc, uint32_t leaf) {
    const float4* tp = sc.tris + (size_t)leaf * 3;
    return widen(tp[0], tp[1], tp[2]);
}
// the ray-direction-independent part of the test
struct TriPre {
    D3 e0, e1, normal, vo;
    double num;
};
__device__ __forceinline__ TriPre tri_pre(const TriV& tr, D3 o) {
    TriPre p;
    p.e0 = tr.v1 - tr.v0;
    p.e1 = tr.v0 - tr.v2;
    p.normal = cross(p.e0, p.e1);
    p.vo = tr.v0 - o;
    p.num = dot(p.vo, p.normal);
    return p;
}
__device__ __forceinline__ bool tri_test_core(const TriPre& p, D3 d, double t_min, double t_max, double& t_out,
                                              double& v_out, double& w_out) {
    double dn = dot(d, p.normal);
    if (fabs(dn) < 1.0e-12) return false;
    D3 tmp = cross(p.vo, d);
    double a = dot(tmp, p.e1), b = dot(tmp, p.e0), num = p.num;
    // Division-free exact rejects.  With |x| >= 2^-500 and 1e-12 <= |dn| <= 2^500 neither
    // x * (1/dn) nor x / dn can round to zero, so their sign is sign(x) * sign(dn): the reference's
    // `v < 0`, `w < 0` and (for t_min >= 0) `t < t_min` are decided without dividing.  NaNs and
    // out-of-range magnitudes fail the guards and take the full path below.
    if (fabs(dn) <= 0x1p500) {
        bool dneg = dn < 0.;
        if (fabs(a) >= 0x1p-500 && (a < 0.) != dneg) return false;                  // v < 0
        if (fabs(b) >= 0x1p-500 && (b < 0.) != dneg) return false;                  // w < 0
        if (t_min >= 0. && fabs(num) >= 0x1p-500 && (num < 0.) != dneg) return false;  // t < 0 <= t_min
    }
    double denom = 1. / dn;
    double v = a * denom;
    double w = b * denom;
    if (v < 0.0 || w < 0.0 || v + w > 1.0) return false;
    double t = num / dn;
    if (t < t_min || t > t_max) return false;
    t_out = t; v_out = v; w_out = w;
    return true;
}
__device__ __forceinline__ bool tri_test(const TriV& tr, D3 o, D3 d, double t_min, double t_max, double& t_out,
                                         double& v_out, double& w_out) {
    return tri_test_core(tri_pre(tr, o), d, t_min, t_max, t_out, v_out, w_out);
}

// The box test is a pure cull - it decides which boxes are opened, the f64 triangle test alone decides
// what is hit - so it runs in f32, at twice the f64 rate and without widening the f32 boxes
// (v_cvt_f64_f32 costs more than the subtraction it feeds).  It must never reject a box that holds a
// triangle the f64 test accepts.  Why it cannot:
//   * a hit point p = o + t d lies in the triangles' own box U; the stored box [lo, hi] is U widened by
//     pad = 2^-21 M per side (M = the scene's largest |coordinate|, pt_bvh.cpp / bvh_pad), so along every
//     axis with d != 0 the plane distances satisfy t_lo + pad |1/d| <= t <= t_hi - pad |1/d|;
//   * the ray's origin is taken twice, ol = fl(o + E) for the lo planes and oh = fl(o - E) for the hi
//     planes, E = 2^-21 max|o_k|: after rounding still >= 6.9 u max|o_k| beyond o (u = 2^-24), i.e. the box
//     is widened by a further ~7 u |o| on every side;
//   * a' = fl(fl(lo - ol) * fl(1/d)) carries three roundings: |a' - (lo - ol)/d| <= 3 u |lo - ol| / |d|
//     <= 3 u (M + 1.0001 |o|) / |d|, which is below the (8 u M + 6.9 u |o|) / |d| of slack.  So a' <= t
//     on the entry side and b' >= t on the exit side of every axis: max(a') <= t <= min(b');
//   * t_min is rounded down and t_max up to f32, so they still bracket t;
//   * d_k == 0, or 1/d_k beyond the f32 range: the factor is +-inf and the numerators of a box that holds
//     the hit are strictly negative (lo) and positive (hi) by the same slack, so the axis gives
//     [-inf, +inf].  A 0 * inf = NaN can only arise for a box the ray cannot hit; v_min/v_max_f32
//     return the non-NaN operand and either outcome is then right.
// A box opened without need costs time, never correctness.
struct RayF {
    float clx, cly, clz;  // fl(ol * i), ol = fl(o + E): subtracted from lo * i
    float chx, chy, chz;  // fl(oh * i), oh = fl(o - E): subtracted from hi * i
    float ix, iy, iz;     // i = fl(1 / d), magnitude capped at 2^100
    // Four-wide nodes (node4_step): which plane of an axis the ray meets first is known from the sign of i, so
    // the step loads "near" and "far" planes through these per-ray byte offsets into the node instead of loading lo
    // and hi and taking min / max of every pair (6 v_min / v_max per box).  fma is monotone in each operand and
    // lo <= hi, cl >= ch for i >= 0 (the reverse for i < 0), so the near value IS min(a, b) of slab32, bit for bit.
    uint32_t nx, ny, nz;  // near plane's float4: 0 / 16 / 32 (lo) or 48 / 64 / 80 (hi); far = near ^ 48 / 80 / 112
    uint32_t hx, hy, hz;  // the same for a two-child node in LDS (node_step_lds): nx / 2 ..; far = near ^ 24 / 40 / 56
    float cnx, cny, cnz;  // the near plane's c (cl or ch)
    float cfx, cfy, cfz;  // the far plane's
};
// The plane distances are formed as fma(plane, i, -c), c = fl(origin * i): one instruction per plane, and
// i itself in f32: i = rcp32(fl32(d)) = (1 / d)(1 + e), |e| <= 3 u (the conversion, and v_rcp_f32's 1 ulp =
// 2 u).  The factor (1 + e) multiplies both terms alike, so the computed distance is
// ((lo - ol) / d)(1 + e) + u |ol i| (rounding of c) + u |result| (the fma): against the sub-then-mul form
// analysed above the error terms change from 3 u |lo - ol| / |d| to (4 u |lo - ol| + u |ol|) / |d|
// <= (4 u M + 5.0004 u |o|) / |d|, still below the (8 u M + 6.9 u |o|) / |d| of slack.  An infinite i would give inf - inf, so |i| is capped at 2^100: on an
// axis with |d_k| < 2^-100 the computed slab is then [-s 2^100, +s 2^100] around the box, s >= 2^-21 M
// its slack, instead of [-inf, inf], which still holds every hit with t <= 2^79 M - all of them unless
// the whole direction vector is that tiny (|d| < 2^-60: such a ray opens every box, ix = NaN).
__device__ __forceinline__ float f32_up(double x) {  // smallest f32 >= x (x not NaN)
    float f = (float)x;
    if ((double)f < x) f = f >= 0.f ? __uint_as_float(__float_as_uint(f) + 1u) : __uint_as_float(__float_as_uint(f) - 1u);
    return f;
}
__device__ __forceinline__ float f32_down(double x) { return -f32_up(-x); }
__device__ __forceinline__ RayF make_rayf(D3 o, D3 d) {
    const double E = fmax(fmax(fabs(o.x), fabs(o.y)), fabs(o.z)) * 0x1p-21;
    const float olx = (float)(o.x + E), oly = (float)(o.y + E), olz = (float)(o.z + E);
    const float ohx = (float)(o.x - E), ohy = (float)(o.y - E), ohz = (float)(o.z - E);
    RayF r;
    // i = rcp32(fl32(d_k)): one v_cvt and one v_rcp_f32 (1 ulp) instead of an f64 division (~12 instructions)
    auto inv = [](double dk) { const float q = __builtin_amdgcn_rcpf((float)dk); return copysignf(fminf(fabsf(q), 0x1p100f), q); };
    r.ix = inv(d.x); r.iy = inv(d.y); r.iz = inv(d.z);
    if (!(fmax(fmax(fabs(d.x), fabs(d.y)), fabs(d.z)) >= 0x1p-60)) r.ix = r.iy = r.iz = __builtin_nanf("");  // no cull
    r.clx = olx * r.ix; r.cly = oly * r.iy; r.clz = olz * r.iz;
    r.chx = ohx * r.ix; r.chy = ohy * r.iy; r.chz = ohz * r.iz;
    const bool px = !(r.ix < 0.f), py = !(r.iy < 0.f), pz = !(r.iz < 0.f);  // (NaN i: no cull either way)
    r.nx = px ? 0u : 48u; r.ny = py ? 16u : 64u; r.nz = pz ? 32u : 80u;
    r.hx = px ? 0u : 24u; r.hy = py ? 8u : 32u; r.hz = pz ? 16u : 40u;
    r.cnx = px ? r.clx : r.chx; r.cny = py ? r.cly : r.chy; r.cnz = pz ? r.clz : r.chz;
    r.cfx = px ? r.chx : r.clx; r.cfy = py ? r.chy : r.cly; r.cfz = pz ? r.chz : r.clz;
    return r;
}
__device__ __forceinline__ bool slab32(float lox, float loy, float loz, float hix, float hiy, float hiz, const RayF& r,
                                       float t_min, float t_max, float& t_near) {
#ifdef PT_SLAB_MULSUB  // (tuning builds: the product and the difference as two instructions)
    float ax = lox * r.ix - r.clx, bx = hix * r.ix - r.chx;
    float ay = loy * r.iy - r.cly, by = hiy * r.iy - r.chy;
    float az = loz * r.iz - r.clz, bz = hiz * r.iz - r.chz;
#else
    float ax = __builtin_fmaf(lox, r.ix, -r.clx), bx = __builtin_fmaf(hix, r.ix, -r.chx);
    float ay = __builtin_fmaf(loy, r.iy, -r.cly), by = __builtin_fmaf(hiy, r.iy, -r.chy);
    float az = __builtin_fmaf(loz, r.iz, -r.clz), bz = __builtin_fmaf(hiz, r.iz, -r.chz);
#endif
    float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), t_min));
    float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), t_max));
    t_near = tn;
    return tn <= tf;
}

// BVH traversal, one ray per lane, stack in LDS laid out [entry][lane] (bank = lane: conflict-free).
// ANY = stop at the first accepted triangle (shadow rays).  Returns leaf-order index / PT_MISS.
// One node step: slab-test both children, pick the next node (near child first, far child pushed)
// or pop.  kDone marks an empty stack; it is negative like a leaf reference but decodes to a
// triangle offset >= 2^28, which no scene has.
// geometry/mod.rs:31-72 Plane::intersect / Sphere::intersect, the accept/reject part
__device__ __forceinline__ bool prim_test(const Prim& pr, D3 o, D3 d, double t_min, double t_max, double& t_out) {
    if (pr.kind == PT_PRIM_PLANE) {
        D3 n = ld3(pr.n);
        double denom = dot(n, d);
        if (fabs(denom) < 1e-6) return false;
        double t = dot(ld3(pr.p) - o, n) / denom;
        if (t > t_max || t < t_min) return false;
        t_out = t;
        return true;
    }
    D3 p = o - ld3(pr.p);
    double b = dot(d, p);
    double c = dot(p, p) - pr.radius * pr.radius;
    double dd = sqrt(b * b - c);  // NaN when the ray misses: both comparisons below fail
    double t1 = (-b - dd);
    double t2 = (-b + dd);
    if (t1 > t_min && t1 < t_max) { t_out = t1; return true; }
    if (t2 > t_min && t2 < t_max) { t_out = t2; return true; }
    return false;
}
// The analytic primitives are tested before the tree is walked: a hit shortens t_max for the
// traversal, and with ids above every triangle id a triangle at a bit-identical t wins the tie.
__device__ __forceinline__ void prims_closest(const Scene& sc, D3 o, D3 d, double t_min, double& t_max, uint32_t& best,
                                              uint32_t& best_id) {
    for (uint32_t k = 0; k < sc.n_prims; ++k) {
        double t;
        if (!prim_test(sc.prims[k], o, d, t_min, t_max, t)) continue;
        if (best != PT_MISS && !(t < t_max)) continue;
        best = best_id = PT_PRIM_BASE + k;
        t_max = t;
    }
}
#ifdef PT_DEBUG_TAIL  // diagnostic build only (scripts/tail_probe.py): start / end clock of every persistent extend wave
__device__ unsigned long long g_dbg_tail[16384];
extern "C" int pt_debug_tail(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_tail), sizeof(unsigned long long) * 16384);
}
#endif
constexpr int kDone = (int)0x80000000;
// The stack of a traversal can never be found full: a two-child tree of depth <= 8 gets 8 LDS entries,
// anything deeper is walked through its four-wide collapse, whose need bvh_collapse4 bounds by
// construction at kWideStack (kWideLds in LDS + kWideSpill private); bvh_validate refuses trees deeper
// than kStackDepth before either.  A build with -DPT_DEBUG_CHECKS counts any push that does, instead
// of trusting that argument: pt_debug_stack_overflows() must read 0 after any render.
#ifdef PT_DEBUG_CHECKS
__device__ unsigned int g_stack_overflows;
extern "C" int pt_debug_stack_overflows(unsigned int* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stack_overflows), sizeof(unsigned int));
}
#define PT_STACK_CHECK(sp, depth) do { if ((sp) >= (depth)) atomicAdd(&g_stack_overflows, 1u); } while (0)
#else
#define PT_STACK_CHECK(sp, depth) ((void)0)
#endif
template <int DEPTH, bool STATS, int STRIDE = kTravBlock>
__device__ __forceinline__ int node_step(const Scene& sc, int cur, const RayF& rf, float t_min, float t_max, int* stk,
                                         int& sp, uint32_t& n_nodes) {
    const float4* np = sc.nodes + (size_t)cur * 4;
    float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
    int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
    float tn0, tn1;
    bool h0 = slab32(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, rf, t_min, t_max, tn0) && c0 != PT_BVH_EMPTY;
    bool h1 = slab32(n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, rf, t_min, t_max, tn1) && c1 != PT_BVH_EMPTY;
    if (STATS) n_nodes += 2;
    if (h0 && h1) {
        int nearc = c0, farc = c1;
        if (tn1 < tn0) { nearc = c1; farc = c0; }
        PT_STACK_CHECK(sp, DEPTH);
        if (sp < DEPTH) { stk[sp * STRIDE] = farc; ++sp; }
        return nearc;
    }
    if (h0) return c0;
    if (h1) return c1;
    if (sp == 0) return kDone;
    --sp;
    return stk[sp * STRIDE];
}

// ---- two-child nodes of a scene held in LDS (k_path LDS) ---------------------------------------------------------
// The kernel's copy of a node (and of an EyeNode) is plane-major: float 2p + k = plane p (lo x y z, hi x y z) of child
// k, floats 12 and 13 the children; 64 bytes as before.  A ray then reads the near and the far plane of an axis for
// both children with one 8-byte load each, picked by the sign of its direction (RayF::hx.., as node4_step does):
// six loads, and no min / max of lo / hi pairs (12 instructions a step).  lds_node_float: where float j of the
// global layout (child-major: lo xyz, hi xyz of child 0, of child 1, the children) goes.
__device__ __forceinline__ uint32_t lds_node_float(uint32_t j) { return j < 6u ? 2u * j : (j < 12u ? 2u * (j - 6u) + 1u : j); }
typedef float V2f __attribute__((ext_vector_type(2)));
typedef int V2i __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const char* LdsB;
__device__ __forceinline__ V2f lds_f2(LdsB b, uint32_t off) { return *(const __attribute__((address_space(3))) V2f*)(b + off); }
__device__ __forceinline__ V2i lds_i2(LdsB b, uint32_t off) { return *(const __attribute__((address_space(3))) V2i*)(b + off); }
// what a step does once both children are tested (node_step, node_step_lds, traverse_eye)
template <int DEPTH, int STRIDE>
__device__ __forceinline__ int two_child_next(bool h0, bool h1, float tn0, float tn1, int c0, int c1, int* stk, int& sp) {
    if (h0 && h1) {
        int nearc = c0, farc = c1;
        if (tn1 < tn0) { nearc = c1; farc = c0; }
        PT_STACK_CHECK(sp, DEPTH);
        if (sp < DEPTH) { stk[sp * STRIDE] = farc; ++sp; }
        return nearc;
    }
    if (h0) return c0;
    if (h1) return c1;
    if (sp == 0) return kDone;
    --sp;
    return stk[sp * STRIDE];
}
template <int DEPTH, int STRIDE>
__device__ __forceinline__ int node_step_lds(const float4* nodes, int cur, const RayF& rf, float t_min, float t_max, int* stk, int& sp) {
    const LdsB b = (LdsB)nodes;
    const uint32_t nb = (uint32_t)cur << 6;
    const V2f px = lds_f2(b, nb | rf.hx), py = lds_f2(b, nb | rf.hy), pz = lds_f2(b, nb | rf.hz);
    const V2f qx = lds_f2(b, nb | (rf.hx ^ 24u)), qy = lds_f2(b, nb | (rf.hy ^ 40u)), qz = lds_f2(b, nb | (rf.hz ^ 56u));
    const V2i c = lds_i2(b, nb | 48u);
    // slab32 with the pairs' min / max resolved by the ray's signs (bit for bit its t_near and its verdict: RayF::nx)
    const float tn0 = fmaxf(fmaxf(__builtin_fmaf(px.x, rf.ix, -rf.cnx), __builtin_fmaf(py.x, rf.iy, -rf.cny)),
                            fmaxf(__builtin_fmaf(pz.x, rf.iz, -rf.cnz), t_min));
    const float tf0 = fminf(fminf(__builtin_fmaf(qx.x, rf.ix, -rf.cfx), __builtin_fmaf(qy.x, rf.iy, -rf.cfy)),
                            fminf(__builtin_fmaf(qz.x, rf.iz, -rf.cfz), t_max));
    const float tn1 = fmaxf(fmaxf(__builtin_fmaf(px.y, rf.ix, -rf.cnx), __builtin_fmaf(py.y, rf.iy, -rf.cny)),
                            fmaxf(__builtin_fmaf(pz.y, rf.iz, -rf.cnz), t_min));
    const float tf1 = fminf(fminf(__builtin_fmaf(qx.y, rf.ix, -rf.cfx), __builtin_fmaf(qy.y, rf.iy, -rf.cfy)),
                            fminf(__builtin_fmaf(qz.y, rf.iz, -rf.cfz), t_max));
    return two_child_next<DEPTH, STRIDE>(tn0 <= tf0 && c.x != PT_BVH_EMPTY, tn1 <= tf1 && c.y != PT_BVH_EMPTY, tn0, tn1, c.x, c.y, stk, sp);
}

// ---- four-wide nodes (deep trees) -----------------------------------------------------------------------
// The stack of a wide traversal: kWideLds entries in LDS ([entry][lane], as above), the rest in a private
// array that only a ray with more than kWideLds pending subtrees touches.  bvh_collapse4 bounds the need
// by kWideStack for every tree that passed bvh_validate.
typedef float V4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const V4f* LdsF4;  // a pointer the compiler knows to be LDS (ds_read_b128, not flat)
__device__ __forceinline__ float4 ldsf4(LdsF4 p, int k) { const V4f v = p[k]; return make_float4(v.x, v.y, v.z, v.w); }
struct TopNodes {   // Scene::nodes4[0 .. n) copied into LDS (pt_device.h kTopNodes)
    LdsF4 p;
    int n;
};
struct WideStack {
    int* lds;     // this lane's column: entry e at lds[e * STRIDE]
    int* spill;   // kWideSpill private entries
    TopNodes top;
};
// all threads of the workgroup, before any of them leaves; BLOCK threads
template <int BLOCK, int MAX = kTopNodes>
__device__ __forceinline__ TopNodes load_top_nodes(const Scene& sc, float4* s_top) {
    const int n = min((int)sc.n_top, MAX);
    for (int k = threadIdx.x; k < n * 8; k += BLOCK) s_top[k] = sc.nodes4[k];
    __syncthreads();
    return TopNodes{(LdsF4)(V4f*)s_top, n};
}
template <int STRIDE>
__device__ __forceinline__ void wide_push(const WideStack& st, int& sp, int ref) {
    PT_STACK_CHECK(sp, kWideStack);
    if (sp < kWideLds) st.lds[sp * STRIDE] = ref;
    else if (sp < kWideStack) st.spill[sp - kWideLds] = ref;
    ++sp;
}
template <int STRIDE>
__device__ __forceinline__ int wide_pop(const WideStack& st, int& sp) {
    if (sp == 0) return kDone;
    --sp;
    // (volatile: an LDS read on the common path, not one flat load through a select of the two addresses)
    int ref = *(volatile int*)(st.lds + min(sp, kWideLds - 1) * STRIDE);
    if (__builtin_expect(sp >= kWideLds, 0)) ref = st.spill[sp - kWideLds];
    return ref;
}
#ifdef PT_VOTE_PROFILE  // diagnostic build (scripts/diag/vote_profile.py): where the lanes of the vote loop go
__device__ unsigned long long g_vote_prof[16];
#define PT_VP(i, v) vp[i] += (v)
#else
#define PT_VP(i, v)
#endif
// One step at a four-wide node: the four boxes are tested together (one float4 per plane), the children
// that are hit are ordered by entry distance with a 5-comparator network on (distance bits, child) pairs
// (a miss has the largest key), the nearest becomes the next node and the others go on the stack,
// farthest first.
template <bool STATS, int STRIDE>
__device__ __forceinline__ int node4_step(const Scene& sc, int cur, const RayF& rf, float t_min, float t_max,
                                          const WideStack& st, int& sp, uint32_t& n_nodes) {
    // near / far planes of the three axes through the ray's offsets (RayF::nx..): the node's byte offset fits 32 bits
    // (scene_create refuses a tree of 2^25 four-wide nodes), so the loads are base + 32-bit offset
    float4 px, py, pz, qx, qy, qz, cf;
    const uint32_t nb = (uint32_t)cur << 7;
    const uint32_t ax = nb | rf.nx, ay = nb | rf.ny, az = nb | rf.nz;
    if (kTopNodes > 0 && cur < st.top.n) {  // the top of the tree lives in LDS
        typedef __attribute__((address_space(3))) const char* LdsB;
        LdsB b = (LdsB)st.top.p;
        auto ld = [&](uint32_t off) { const V4f v = *(LdsF4)(b + off); return make_float4(v.x, v.y, v.z, v.w); };
        px = ld(ax); py = ld(ay); pz = ld(az); qx = ld(ax ^ 48u); qy = ld(ay ^ 80u); qz = ld(az ^ 112u); cf = ld(nb | 96u);
    } else {
        const char* b = reinterpret_cast<const char*>(sc.nodes4);
        auto ld = [&](uint32_t off) { return *reinterpret_cast<const float4*>(b + off); };
        px = ld(ax); py = ld(ay); pz = ld(az); qx = ld(ax ^ 48u); qy = ld(ay ^ 80u); qz = ld(az ^ 112u); cf = ld(nb | 96u);
    }
    const int c[4] = {__float_as_int(cf.x), __float_as_int(cf.y), __float_as_int(cf.z), __float_as_int(cf.w)};
    const float nxp[4] = {px.x, px.y, px.z, px.w}, nyp[4] = {py.x, py.y, py.z, py.w}, nzp[4] = {pz.x, pz.y, pz.z, pz.w};
    const float fxp[4] = {qx.x, qx.y, qx.z, qx.w}, fyp[4] = {qy.x, qy.y, qy.z, qy.w}, fzp[4] = {qz.x, qz.y, qz.z, qz.w};
    uint32_t key[4];
    int ch[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // slab32 with the pairs' min / max resolved by the ray's signs
        const float tn = fmaxf(fmaxf(__builtin_fmaf(nxp[k], rf.ix, -rf.cnx), __builtin_fmaf(nyp[k], rf.iy, -rf.cny)),
                               fmaxf(__builtin_fmaf(nzp[k], rf.iz, -rf.cnz), t_min));
        const float tf = fminf(fminf(__builtin_fmaf(fxp[k], rf.ix, -rf.cfx), __builtin_fmaf(fyp[k], rf.iy, -rf.cfy)),
                               fminf(__builtin_fmaf(fzp[k], rf.iz, -rf.cfz), t_max));
        const bool hit = tn <= tf && c[k] != PT_BVH_EMPTY;
        if (STATS) n_nodes += c[k] != PT_BVH_EMPTY;
        // tn >= t_min; for the usual t_min >= 0 its bits order like unsigned integers (a negative t_min only
        // perturbs the visiting order)
        key[k] = hit ? __float_as_uint(tn) : 0xFFFFFFFFu;
        ch[k] = c[k];
    }
    // (key, child) pairs through the network: one compare and four selects per comparator, no branches
#define PT_CSWAP(a, b) { const bool sw_ = key[b] < key[a]; \
        const uint32_t ka_ = sw_ ? key[b] : key[a], kb_ = sw_ ? key[a] : key[b]; \
        const int ca_ = sw_ ? ch[b] : ch[a], cb_ = sw_ ? ch[a] : ch[b]; \
        key[a] = ka_; key[b] = kb_; ch[a] = ca_; ch[b] = cb_; }
    PT_CSWAP(0, 1) PT_CSWAP(2, 3) PT_CSWAP(0, 2) PT_CSWAP(1, 3) PT_CSWAP(1, 2)
#undef PT_CSWAP
#ifdef PT_VOTE_PROFILE
    {
        const unsigned long long m_ = __ballot(key[0] == 0xFFFFFFFFu), a_ = __ballot(true);
        const unsigned long long h2_ = __ballot(key[1] != 0xFFFFFFFFu), h3_ = __ballot(key[2] != 0xFFFFFFFFu), h4_ = __ballot(key[3] != 0xFFFFFFFFu);
        if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(a_)) {
            atomicAdd(&g_vote_prof[14], (unsigned long long)__popcll(m_));
            atomicAdd(&g_vote_prof[15], (unsigned long long)(__popcll(h2_) + __popcll(h3_) + __popcll(h4_)));
        }
    }
#endif
    if (key[0] == 0xFFFFFFFFu) return wide_pop<STRIDE>(st, sp);
    // the other hits (a prefix of key[1..3]) go on the stack, farthest first
    const int n = (key[1] != 0xFFFFFFFFu) + (key[2] != 0xFFFFFFFFu) + (key[3] != 0xFFFFFFFFu);
    const int c1 = ch[1], c2 = ch[2], c3 = ch[3];
    if (__builtin_expect(sp + n <= kWideLds, 1)) {  // all within the LDS part
        PT_STACK_CHECK(sp + n - 1, kWideStack);
        int* top = st.lds + (sp + n) * STRIDE;
        if (n >= 1) top[-1 * STRIDE] = c1;
        if (n >= 2) top[-2 * STRIDE] = c2;
        if (n >= 3) top[-3 * STRIDE] = c3;
        sp += n;
    } else {
        if (n >= 3) wide_push<STRIDE>(st, sp, c3);
        if (n >= 2) wide_push<STRIDE>(st, sp, c2);
        if (n >= 1) wide_push<STRIDE>(st, sp, c1);
    }
    return ch[0];
}

// BVH traversal of the rays [begin, end) by one wave, stack in LDS laid out [entry][lane]
// (bank = lane: conflict-free).  ANY = stop at the first accepted triangle (shadow rays).
//
// Every lane is a small state machine whose next unit of work is one of
//   node   (cur >= 0)             slab-test the two children of an inner node, descend / push / pop
//   tri    (cur < 0, != kDone)    test ONE triangle of the leaf range encoded in cur, advance / pop
//   retire (cur == kDone)         hand the finished ray to done() and claim the next unclaimed ray
//                                 of the wave's range (ballot + mbcnt rank: no atomics, no memory)
// and each trip of the loop the wave runs only the kind most of its lanes are waiting for (a
// ballot vote).  The chosen body therefore runs with at least a third — typically well over half —
// of the working lanes active, whatever mix of cheap and expensive rays the wave holds.  The
// one-ray-per-lane while-while form measured 16 % lane utilisation on incoherent secondary rays
// of the 10M-triangle scene: every lane waited for the longest descent, the largest leaf and the
// longest ray of its 64.
// fetch(idx, o, d, t_min, t_max) loads ray idx; done(idx, leaf_tri, t) consumes its result.
// Where a persistent wave gets its next range of queue entries from: the per-segment work cursors
// of a segmented queue.  A wave starts on segment (wave id mod kSegs) and moves on round-robin when
// a segment is used up; one atomic per claimed range, executed by one lane.  With cursors == nullptr
// the wave owns exactly the range it was started with.
struct RangePool {
    uint32_t* cursors;
    const uint32_t* counts;
    uint32_t seg_size, grab, seg, tried;
    __device__ __forceinline__ bool next(uint32_t& begin, uint32_t& end) {
        if (!cursors) return false;
        const uint32_t lane = threadIdx.x & 63u;
        static_assert(kSegs == 64, "one lane looks at one segment");
        // `tried` bounds the retries: a claim can lose a race, but every lost race means another wave
        // took a range, and once a segment is seen empty it stays empty
        while (tried < 4 * kSegs) {
            // one coherent load per lane: which segments still have unclaimed entries?
            const uint32_t live_k = min(counts[lane * kCtrStride], seg_size);
            const uint32_t cur_k = __hip_atomic_load(&cursors[lane * kCtrStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long open = __ballot(cur_k < live_k);
            if (!open) return false;
            // first open segment at or after `seg`, cyclically
            const unsigned long long rot = seg ? ((open >> seg) | (open << (64u - seg))) : open;
            seg = (seg + (uint32_t)__builtin_ctzll(rot)) % kSegs;
            const uint32_t live = (uint32_t)__builtin_amdgcn_readlane((int)live_k, (int)seg);
            // guided schedule: full ranges while the segment is well stocked, smaller ones towards
            // its end, so that the waves of a launch run dry at about the same time
            const uint32_t rem = live - (uint32_t)__builtin_amdgcn_readlane((int)cur_k, (int)seg);
            const uint32_t g = rem >= 32u * grab ? grab : (rem >= 8u * grab ? max(64u, grab / 2u) : 64u);
            uint32_t c = 0;
            if (lane == 0) c = atomicAdd(&cursors[seg * kCtrStride], g);
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
            if (c < live) {
                begin = seg * seg_size + c;
                end = seg * seg_size + min(c + g, live);
                return true;
            }
            ++tried;
        }
        return false;
    }
};

constexpr int kVoteDouble = 24;
template <int DEPTH, bool ANY, bool STATS, class Fetch, class Done>
__device__ __forceinline__ void trace_range_vote(const Scene& sc, uint32_t begin, uint32_t end, RangePool pool, int* stk,
                                                 TopNodes top, Fetch fetch, Done done, uint32_t& n_nodes, uint32_t& n_tris) {
    constexpr bool WIDE = DEPTH > 8;  // deep trees are walked through their four-wide nodes
    int spill[WIDE ? kWideSpill : 1];
    const WideStack wst{stk, spill, top};
    uint32_t cursor = begin;  // wave-uniform: first unclaimed ray
    bool more = pool.cursors != nullptr;  // wave-uniform: the pool may hold further ranges
    bool busy = false;
    uint32_t idx = 0, best = PT_MISS, best_id = PT_MISS;
    D3 o = mk(0., 0., 0.), d = o;
    RayF rf = make_rayf(o, o);
    double t_min = 0., t_max = 0.;
    float t_min32 = 0.f, t_max32 = 0.f;  // t_min rounded down, t_max rounded up (slab32)
    int sp = 0, cur = kDone;
#ifdef PT_VOTE_PROFILE
    unsigned long long vp[16] = {};
    auto vp_flush = [&]() { if ((threadIdx.x & 63u) == 0) for (int k = 0; k < 16; ++k) if (vp[k]) atomicAdd(&g_vote_prof[k], vp[k]); };
#endif
    // Two nested loops.  The inner one runs node and triangle phases, in which the ray itself (o, d, rf,
    // t_min) is loop-invariant: only the traversal state changes, so the compiler keeps the ray in place
    // (with the fetch in the same loop every trip paid ~25 register copies at the loop's merge point).
    // It is left when the vote goes to the retire phase, which the outer loop runs.
    for (;;) {
        if (more && cursor >= end) {  // the range is handed out: claim the next one (lanes still at work keep going)
            more = pool.next(begin, end);
            if (more) cursor = begin;
        }
        unsigned long long m_ret;
        int n_ret;
        bool is_ret;
        for (;;) {
            const bool is_node = cur >= 0;
            const bool is_tri = cur < 0 && cur != kDone;
            is_ret = cur == kDone && (busy || cursor < end);
            m_ret = __ballot(is_ret);
            const int n_node = __popcll(__ballot(is_node)), n_tri = __popcll(__ballot(is_tri));
            n_ret = __popcll(m_ret);
            // (plain majority: weighting the three counts 3:2:2, 2:1:2, 2:3:3 or 1:1:2 measured 0-3.5 % slower)
            if (n_node >= n_tri && n_node >= n_ret) {
                if (n_node == 0) {
#ifdef PT_VOTE_PROFILE
                    vp_flush();
#endif
                    return;  // nothing left in the range
                }
                PT_VP(0, 1); PT_VP(1, n_node); PT_VP(8, n_tri); PT_VP(9, n_ret);
                if (is_node) {
                    if (WIDE) cur = node4_step<STATS, kTravBlock>(sc, cur, rf, t_min32, t_max32, wst, sp, n_nodes);
                    else cur = node_step<DEPTH, STATS>(sc, cur, rf, t_min32, t_max32, stk, sp, n_nodes);
                }
                // a wave with at least kVoteDouble lanes at inner nodes takes a second node step before it votes again (the
                // vote costs ~15 instructions a trip; thresholds 16 / 24 / 32 / 48: extend -2.1 / -2.0 / -1.8 / -0.7 %)
                PT_VP(2, n_node >= kVoteDouble); PT_VP(3, n_node >= kVoteDouble ? __popcll(__ballot(cur >= 0)) : 0);
                if (n_node >= kVoteDouble && cur >= 0) {
                    if (WIDE) cur = node4_step<STATS, kTravBlock>(sc, cur, rf, t_min32, t_max32, wst, sp, n_nodes);
                    else cur = node_step<DEPTH, STATS>(sc, cur, rf, t_min32, t_max32, stk, sp, n_nodes);
                }
            } else if (n_tri >= n_ret) {
#ifdef PT_VOTE_PROFILE
                {
                    const uint32_t c_ = is_tri ? (((uint32_t)~cur) & 7u) + 1u : 0u;
                    uint32_t mx = 0, sm = 0;
                    for (int l = 0; l < 64; ++l) { const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)c_, l); mx = max(mx, v); sm += v; }
                    PT_VP(4, 1); PT_VP(5, n_tri); PT_VP(6, mx); PT_VP(7, sm); PT_VP(10, n_node); PT_VP(11, n_ret);
                }
#endif
                if (is_tri) {
                    const uint32_t code = (uint32_t)~cur, first = code >> 3, cnt = (code & 7u) + 1u;
                    bool stop = false;
                    for (uint32_t k = 0; k < cnt; ++k) {
                        const float4* tp = sc.tris + (size_t)(first + k) * 3;
                        float4 a = tp[0], b = tp[1], c = tp[2];
                        TriV tr = widen(a, b, c);
                        if (STATS) n_tris += 1;
                        double t, v, w;
                        if (!tri_test(tr, o, d, t_min, t_max, t, v, w)) continue;
                        uint32_t gid = (uint32_t)__float_as_int(a.w);
                        // the reference accepts t == t_max (triangle_mesh.rs:113) and keeps the candidate
                        // tested last; the build-defined rule keeps the lowest triangle id
                        if (best != PT_MISS && !(t < t_max || gid < best_id)) continue;
                        best = first + k; best_id = gid; t_max = t;
                        t_max32 = f32_up(t);
                        if (ANY) { stop = true; break; }
                    }
                    if (stop) cur = kDone;
                    else if (WIDE) cur = wide_pop<kTravBlock>(wst, sp);
                    else if (sp == 0) cur = kDone;
                    else { --sp; cur = stk[sp * kTravBlock]; }
                }
            } else {
                break;  // the retire phase has the most lanes
            }
        }
        PT_VP(12, 1); PT_VP(13, n_ret);
        if (is_ret) {
            if (busy) done(idx, best, t_max);
            busy = false;
            uint32_t cand = cursor + __builtin_amdgcn_mbcnt_hi((uint32_t)(m_ret >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((uint32_t)m_ret, 0u));
            if (cand < end) {
                idx = cand;
                fetch(idx, o, d, t_min, t_max);
                rf = make_rayf(o, d);
                best = PT_MISS; best_id = PT_MISS;
                sp = 0;
                cur = sc.n_nodes ? 0 : kDone;
                if (sc.n_prims) {
                    prims_closest(sc, o, d, t_min, t_max, best, best_id);
                    if (ANY && best != PT_MISS) cur = kDone;
                }
                t_min32 = f32_down(t_min);
                t_max32 = f32_up(t_max);
                busy = true;
            }
        }
        if (cursor < end) cursor += (uint32_t)n_ret;
    }
}

// The other scheduling of the same work, one ray per lane for its whole life: "while-while" —
// the wave keeps stepping through inner nodes until every lane holds a leaf (or is done), then all
// of them test their leaf together.  No bookkeeping per unit of work, so it wins where the rays of
// a wave are coherent or every ray is short (primary rays: 96 % lane utilisation on CornellBox;
// scenes under ~500 triangles), and loses by 15-40 % on incoherent rays in deeper trees (measured
// crossover: DESIGN.md "Traversal scheduling").  Returns leaf-order index / PT_MISS.
// EDGES: the triangle test starts from Scene::tri_edges instead of the vertices (the scene-in-LDS kernel).
template <int DEPTH, bool ANY, bool STATS, int STRIDE = kTravBlock, bool EDGES = false>
__device__ __forceinline__ uint32_t traverse(const Scene& sc, D3 o, D3 d, double t_min, double& t_max, int* stk,
                                             uint32_t& n_nodes, uint32_t& n_tris, TopNodes top = TopNodes{nullptr, 0}) {
    uint32_t best = PT_MISS, best_id = PT_MISS;
    if (sc.n_prims) {
        prims_closest(sc, o, d, t_min, t_max, best, best_id);
        if (ANY && best != PT_MISS) return best;
    }
    if (sc.n_nodes == 0) return best;
    constexpr bool WIDE = DEPTH > 8;
    int spill[WIDE ? kWideSpill : 1];
    const WideStack wst{stk, spill, top};
    const RayF rf = make_rayf(o, d);
    const float t_min32 = f32_down(t_min);
    float t_max32 = f32_up(t_max);
    int sp = 0;
    int cur = 0;
    while (cur != kDone) {
        while (cur >= 0) {
            if (WIDE) cur = node4_step<STATS, STRIDE>(sc, cur, rf, t_min32, t_max32, wst, sp, n_nodes);
            else if (EDGES) cur = node_step_lds<DEPTH, STRIDE>(sc.nodes, cur, rf, t_min32, t_max32, stk, sp);  // the scene-in-LDS kernel
            else cur = node_step<DEPTH, STATS, STRIDE>(sc, cur, rf, t_min32, t_max32, stk, sp, n_nodes);
        }
        if (cur == kDone) break;
        int code = ~cur;
        uint32_t first = (uint32_t)code >> 3, cnt = ((uint32_t)code & 7u) + 1u;
        for (uint32_t k = 0; k < cnt; ++k) {
            double t, v, w;
            uint32_t gid;
            if (STATS) n_tris += 1;
            if (EDGES) {
                const EyeTri& e = sc.tri_edges[first + k];
                TriPre p;
                p.e0 = ld3(e.e0); p.e1 = ld3(e.e1); p.normal = ld3(e.normal);
                p.vo = ld3(e.vo) - o;             // tri_pre: vo = v0 - o, num = vo . normal
                p.num = dot(p.vo, p.normal);
                if (!tri_test_core(p, d, t_min, t_max, t, v, w)) continue;
                gid = e.gid;
            } else {
                const float4* tp = sc.tris + (size_t)(first + k) * 3;
                float4 a = tp[0], b = tp[1], c = tp[2];
                TriV tr = widen(a, b, c);
                if (!tri_test(tr, o, d, t_min, t_max, t, v, w)) continue;
                gid = (uint32_t)__float_as_int(a.w);
            }
            // the reference accepts t == t_max (triangle_mesh.rs:113) and keeps the candidate
            // tested last; the build-defined rule keeps the lowest triangle id
            if (best != PT_MISS && !(t < t_max || gid < best_id)) continue;
            best = first + k; best_id = gid; t_max = t;
            t_max32 = f32_up(t);
            if (ANY) return best;
        }
        if (WIDE) { cur = wide_pop<STRIDE>(wst, sp); continue; }
        if (sp == 0) break;
        --sp;
        cur = stk[sp * STRIDE];
    }
    return best;
}

// ---- packet traversal of the four-wide tree: the 64 rays of a wave walk the tree TOGETHER ------------------------
// For waves whose rays are coherent - the primary rays of a deep tree, where a wave holds one pixel in 64 frames
// (Frame::grp_log2) and its rays differ by the frame jitter alone.  The node index, the stack and the visiting
// order are wave-uniform (scalar registers, one LDS column): a node is opened if ANY lane's slab test passes, a
// leaf's triangles are tested by every lane.  Against one traversal per lane there is no per-lane stack traffic, no
// per-lane sorting network, no divergence between lanes that are at different depths, and a node is fetched once
// per wave, through one address.  Each lane still runs the same slab tests against its own ray and the same f64
// triangle tests with its own t_max, and a superset of the leaves it would have opened alone is opened, so its
// closest hit (and the lowest-id tie rule) is what traverse() finds: ids and distances are bit-identical (tested).
// `ustk`: kWideStack ints of LDS for the wave.  Lanes with active == false hit nothing.
// The active lanes must agree on the sign of every direction component (packet_signs_agree; one pixel's jitter apart
// they nearly always do): the near and far planes of a node are then the same float4s for the whole wave.
__device__ __forceinline__ bool packet_signs_agree(const RayF& rf, bool active) {
    const unsigned long long act = __ballot(active);
    const unsigned long long sx = __ballot(active && rf.nx != 0u), sy = __ballot(active && rf.ny != 16u), sz = __ballot(active && rf.nz != 32u);
    return (sx == 0 || sx == act) && (sy == 0 || sy == act) && (sz == 0 || sz == act);
}
template <bool STATS>
__device__ __forceinline__ uint32_t traverse_packet4(const Scene& sc, D3 o, D3 d, const RayF& rf, double t_min, double& t_max, bool active,
                                                     int* ustk, TopNodes top, uint32_t& n_nodes, uint32_t& n_tris) {
    uint32_t best = PT_MISS, best_id = PT_MISS;
    if (sc.n_prims && active) prims_closest(sc, o, d, t_min, t_max, best, best_id);
    if (sc.n_nodes == 0 || !__ballot(active)) return best;
    const float t_min32 = f32_down(t_min);
    float t_max32 = active ? f32_up(t_max) : -1.f;   // below t_min: every slab test fails
    int sp = 0, cur = 0;                              // wave-uniform
    // the float4 of an axis' near plane within a node (lo: 0 / 1 / 2, hi: 3 / 4 / 5), the same for every active lane
    const int inx = __ballot(active && rf.nx != 0u) ? 3 : 0, iny = __ballot(active && rf.ny != 16u) ? 4 : 1,
              inz = __ballot(active && rf.nz != 32u) ? 5 : 2;
    for (;;) {
        while (cur >= 0) {
            // near and far planes by wave-uniform index (packet_signs_agree; RayF::nx): no min / max of lo / hi pairs
            float4 px, py, pz, qx, qy, qz, cf;
            if (kTopNodes > 0 && cur < top.n) {
                LdsF4 np = top.p + cur * 8;
                px = ldsf4(np, inx); py = ldsf4(np, iny); pz = ldsf4(np, inz);
                qx = ldsf4(np, inx ^ 3); qy = ldsf4(np, iny ^ 5); qz = ldsf4(np, inz ^ 7); cf = ldsf4(np, 6);
            } else {
                const float4* np = sc.nodes4 + (size_t)cur * 8;
                px = np[inx]; py = np[iny]; pz = np[inz]; qx = np[inx ^ 3]; qy = np[iny ^ 5]; qz = np[inz ^ 7]; cf = np[6];
            }
            const int c[4] = {__float_as_int(cf.x), __float_as_int(cf.y), __float_as_int(cf.z), __float_as_int(cf.w)};
            const float pxs[4] = {px.x, px.y, px.z, px.w}, pys[4] = {py.x, py.y, py.z, py.w}, pzs[4] = {pz.x, pz.y, pz.z, pz.w};
            const float qxs[4] = {qx.x, qx.y, qx.z, qx.w}, qys[4] = {qy.x, qy.y, qy.z, qy.w}, qzs[4] = {qz.x, qz.y, qz.z, qz.w};
            uint32_t key[4];
            int ch[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float tn = fmaxf(fmaxf(__builtin_fmaf(pxs[k], rf.ix, -rf.cnx), __builtin_fmaf(pys[k], rf.iy, -rf.cny)),
                                       fmaxf(__builtin_fmaf(pzs[k], rf.iz, -rf.cnz), t_min32));
                const float tf = fminf(fminf(__builtin_fmaf(qxs[k], rf.ix, -rf.cfx), __builtin_fmaf(qys[k], rf.iy, -rf.cfy)),
                                       fminf(__builtin_fmaf(qzs[k], rf.iz, -rf.cfz), t_max32));
                const bool hit = tn <= tf;
                if (STATS) n_nodes += active && c[k] != PT_BVH_EMPTY;
                const unsigned long long m = __ballot(hit);
                ch[k] = __builtin_amdgcn_readfirstlane(c[k]);
                // the child is opened if any lane hits its box; visiting order by the entry distance of the first such lane
                const bool any = m != 0 && ch[k] != PT_BVH_EMPTY;
                const uint32_t lead = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(tn), any ? (int)__builtin_ctzll(m) : 0);
                key[k] = any ? lead : 0xFFFFFFFFu;
            }
#define PT_CSWAP(a, b) { const bool sw_ = key[b] < key[a]; \
        const uint32_t ka_ = sw_ ? key[b] : key[a], kb_ = sw_ ? key[a] : key[b]; \
        const int ca_ = sw_ ? ch[b] : ch[a], cb_ = sw_ ? ch[a] : ch[b]; \
        key[a] = ka_; key[b] = kb_; ch[a] = ca_; ch[b] = cb_; }
            PT_CSWAP(0, 1) PT_CSWAP(2, 3) PT_CSWAP(0, 2) PT_CSWAP(1, 3) PT_CSWAP(1, 2)
#undef PT_CSWAP
            if (key[0] == 0xFFFFFFFFu) {
                if (sp == 0) { cur = kDone; break; }
                cur = __builtin_amdgcn_readfirstlane(ustk[--sp]);
                continue;
            }
            // (the collapse bounds the pending subtrees of ANY root-to-leaf walk by kWideStack, whichever children are opened)
            if (key[3] != 0xFFFFFFFFu) { PT_STACK_CHECK(sp, kWideStack); ustk[sp++] = ch[3]; }
            if (key[2] != 0xFFFFFFFFu) { PT_STACK_CHECK(sp, kWideStack); ustk[sp++] = ch[2]; }
            if (key[1] != 0xFFFFFFFFu) { PT_STACK_CHECK(sp, kWideStack); ustk[sp++] = ch[1]; }
            cur = ch[0];
        }
        if (cur == kDone) break;
        const uint32_t code = (uint32_t)~cur, first = code >> 3, cnt = (code & 7u) + 1u;   // uniform
        for (uint32_t k = 0; k < cnt; ++k) {
            const float4* tp = sc.tris + (size_t)(first + k) * 3;
            const float4 a = tp[0], b = tp[1], c = tp[2];
            const TriV tr = widen(a, b, c);
            if (STATS) n_tris += active;
            double t, v, w;
            if (!active || !tri_test(tr, o, d, t_min, t_max, t, v, w)) continue;
            const uint32_t gid = (uint32_t)__float_as_int(a.w);
            if (best != PT_MISS && !(t < t_max || gid < best_id)) continue;   // lowest id wins a tie, as in traverse()
            best = first + k; best_id = gid; t_max = t;
            t_max32 = f32_up(t);
        }
        if (sp == 0) break;
        cur = __builtin_amdgcn_readfirstlane(ustk[--sp]);
    }
    return best;
}

// ---- primary rays of a small scene: the eye-relative tables (pt_device.h EyeNode / EyeTri) -----
// slab32 with the subtraction done once per scene: r = (lo - o - E, hi - o + E) rounded outward to f32 by
// k_build_eye_tables (directed, so only the reciprocal's 3 u and the product's rounding remain: 4 u |r| of
// error against 8 u M + 8 u |o| of slack)
__device__ __forceinline__ bool slab_rel(const float* r, float ix, float iy, float iz, float t_min, float t_max, float& t_near) {
    float ax = r[0] * ix, bx = r[3] * ix;
    float ay = r[1] * iy, by = r[4] * iy;
    float az = r[2] * iz, bz = r[5] * iz;
    float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), t_min));
    float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), t_max));
    t_near = tn;
    return tn <= tf;
}
__global__ void k_build_eye_tables(Scene sc, EyeNode* nodes, EyeTri* tris) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const D3 o = ld3(sc.cam.eye);
    if (i < sc.n_nodes) {
        const float4* np = sc.nodes + (size_t)i * 4;
        float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
        const float b[12] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w};
        const double oo[3] = {o.x, o.y, o.z};
        const double E = fmax(fmax(fabs(o.x), fabs(o.y)), fabs(o.z)) * 0x1p-21;
        for (int k = 0; k < 12; ++k) {
            const bool is_lo = (k % 6) < 3;  // lo xyz, hi xyz per child
            const double rel = (double)b[k] - oo[k % 3];  // exact to ~1e-16: f32 box plane minus f64 eye
            nodes[i].rel[k] = is_lo ? f32_down(rel - E) : f32_up(rel + E);
        }
        nodes[i].child[0] = __float_as_int(n3.x);
        nodes[i].child[1] = __float_as_int(n3.y);
        nodes[i].pad[0] = nodes[i].pad[1] = 0;
    }
    if (i < sc.n_tris) {
        TriPre p = tri_pre(load_tri(sc, i), o);
        EyeTri& e = tris[i];
        e.e0[0] = p.e0.x; e.e0[1] = p.e0.y; e.e0[2] = p.e0.z;
        e.e1[0] = p.e1.x; e.e1[1] = p.e1.y; e.e1[2] = p.e1.z;
        e.normal[0] = p.normal.x; e.normal[1] = p.normal.y; e.normal[2] = p.normal.z;
        e.vo[0] = p.vo.x; e.vo[1] = p.vo.y; e.vo[2] = p.vo.z;
        e.num = p.num;
        e.gid = (uint32_t)__float_as_int(sc.tris[(size_t)i * 3].w);
        e.pad = 0;
    }
}
// Scene::tri_edges: the origin-independent part of tri_pre per triangle (the EyeTri of the origin (0, 0, 0):
// v0 - 0 is v0 exactly), by the same expressions, so a test that starts from the record gives the same bits
__global__ void k_build_tri_edges(Scene sc, EyeTri* tris) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sc.n_tris) return;
    TriPre p = tri_pre(load_tri(sc, i), mk(0., 0., 0.));
    EyeTri& e = tris[i];
    e.e0[0] = p.e0.x; e.e0[1] = p.e0.y; e.e0[2] = p.e0.z;
    e.e1[0] = p.e1.x; e.e1[1] = p.e1.y; e.e1[2] = p.e1.z;
    e.normal[0] = p.normal.x; e.normal[1] = p.normal.y; e.normal[2] = p.normal.z;
    e.vo[0] = p.vo.x; e.vo[1] = p.vo.y; e.vo[2] = p.vo.z;
    e.num = p.num;
    e.gid = (uint32_t)__float_as_int(sc.tris[(size_t)i * 3].w);
    e.pad = 0;
}
// traverse() for a ray that starts at the eye (closest hit only)
// LDSN: Scene::eye_nodes is the kernel's plane-major copy in LDS (node_step_lds)
template <int DEPTH, int STRIDE, bool LDSN = false>
__device__ __forceinline__ uint32_t traverse_eye(const Scene& sc, D3 d, double t_min, double& t_max, int* stk) {
    uint32_t best = PT_MISS, best_id = PT_MISS;
    if (sc.n_prims) prims_closest(sc, ld3(sc.cam.eye), d, t_min, t_max, best, best_id);
    if (sc.n_nodes == 0) return best;
    // (f32 reciprocals of the converted direction: |relative error| <= 3 u, slab_rel's bound becomes 4 u |r|)
    const float ix = __builtin_amdgcn_rcpf((float)d.x), iy = __builtin_amdgcn_rcpf((float)d.y), iz = __builtin_amdgcn_rcpf((float)d.z);
    const float t_min32 = f32_down(t_min);
    float t_max32 = f32_up(t_max);
    int sp = 0;
    int cur = 0;
    // LDSN: near planes by the sign of the direction (an infinite or NaN reciprocal only ever drops a NaN plane from a
    // max / min, which opens the box)
    const uint32_t hx = ix < 0.f ? 24u : 0u, hy = iy < 0.f ? 32u : 8u, hz = iz < 0.f ? 40u : 16u;
    while (cur != kDone) {
        while (LDSN && cur >= 0) {
            const LdsB b = (LdsB)sc.eye_nodes;
            const uint32_t nb = (uint32_t)cur << 6;
            const V2f px = lds_f2(b, nb | hx), py = lds_f2(b, nb | hy), pz = lds_f2(b, nb | hz);
            const V2f qx = lds_f2(b, nb | (hx ^ 24u)), qy = lds_f2(b, nb | (hy ^ 40u)), qz = lds_f2(b, nb | (hz ^ 56u));
            const V2i c = lds_i2(b, nb | 48u);
            const float tn0 = fmaxf(fmaxf(px.x * ix, py.x * iy), fmaxf(pz.x * iz, t_min32));
            const float tf0 = fminf(fminf(qx.x * ix, qy.x * iy), fminf(qz.x * iz, t_max32));
            const float tn1 = fmaxf(fmaxf(px.y * ix, py.y * iy), fmaxf(pz.y * iz, t_min32));
            const float tf1 = fminf(fminf(qx.y * ix, qy.y * iy), fminf(qz.y * iz, t_max32));
            cur = two_child_next<DEPTH, STRIDE>(tn0 <= tf0 && c.x != PT_BVH_EMPTY, tn1 <= tf1 && c.y != PT_BVH_EMPTY, tn0, tn1, c.x, c.y, stk, sp);
        }
        while (!LDSN && cur >= 0) {
            const EyeNode& n = sc.eye_nodes[cur];
            int c0 = n.child[0], c1 = n.child[1];
            float tn0, tn1;
            bool h0 = slab_rel(n.rel, ix, iy, iz, t_min32, t_max32, tn0) && c0 != PT_BVH_EMPTY;
            bool h1 = slab_rel(n.rel + 6, ix, iy, iz, t_min32, t_max32, tn1) && c1 != PT_BVH_EMPTY;
            if (h0 && h1) {
                int nearc = c0, farc = c1;
                if (tn1 < tn0) { nearc = c1; farc = c0; }
                PT_STACK_CHECK(sp, DEPTH);
                if (sp < DEPTH) { stk[sp * STRIDE] = farc; ++sp; }
                cur = nearc;
            } else if (h0) cur = c0;
            else if (h1) cur = c1;
            else if (sp == 0) cur = kDone;
            else { --sp; cur = stk[sp * STRIDE]; }
        }
        if (cur == kDone) break;
        int code = ~cur;
        uint32_t first = (uint32_t)code >> 3, cnt = ((uint32_t)code & 7u) + 1u;
        for (uint32_t k = 0; k < cnt; ++k) {
            const EyeTri& e = sc.eye_tris[first + k];
            TriPre p{ld3(e.e0), ld3(e.e1), ld3(e.normal), ld3(e.vo), e.num};
            double t, v, w;
            if (!tri_test_core(p, d, t_min, t_max, t, v, w)) continue;
            uint32_t gid = e.gid;
            if (best != PT_MISS && !(t < t_max || gid < best_id)) continue;  // lowest id wins a tie, as in traverse()
            best = first + k; best_id = gid; t_max = t;
            t_max32 = f32_up(t);
        }
        if (sp == 0) break;
        --sp;
        cur = stk[sp * STRIDE];
    }
    return best;
}

// Exclusive end of the live entries in the segment that holds entry i0 (i0 = first entry of a
// workgroup; a workgroup never straddles segments).  Dense queues have one segment [0, n).
__device__ __forceinline__ uint32_t live_end(const Live& lv, uint32_t i0) {
    if (!lv.counts) return lv.n;
    uint32_t seg = i0 / lv.seg_size;
    return seg * lv.seg_size + min(lv.counts[seg * kCtrStride], lv.seg_size);
}
// Logical block `blk` (of BLOCK entries) of a sparsely filled segmented queue -> its entry range;
// false past the last live block.  Every wave works it out for itself: lane k takes segment k's
// count (kSegs == 64), a wave prefix sum gives the first block of each segment, a ballot the segment.
template <int BLOCK>
__device__ __forceinline__ bool sparse_block(const Live& lv, uint32_t blk, uint32_t& begin, uint32_t& end) {
    static_assert(kSegs == 64, "one segment per lane");
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t cnt = min(lv.counts[lane * kCtrStride], lv.seg_size);
    const uint32_t nb = (cnt + BLOCK - 1) / BLOCK;
    uint32_t pre = nb;  // inclusive prefix
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t v = __shfl_up(pre, d);
        if (lane >= (uint32_t)d) pre += v;
    }
    const unsigned long long m = __ballot(pre > blk);
    if (!m) return false;
    const int seg = __ffsll((long long)m) - 1;
    const uint32_t first = __shfl(pre - nb, seg), seg_cnt = __shfl(cnt, seg);
    begin = seg * lv.seg_size + (blk - first) * BLOCK;
    end = min(begin + BLOCK, seg * lv.seg_size + seg_cnt);
    return true;
}
__device__ __forceinline__ unsigned long long live_total(const Live& lv) {
    if (!lv.counts) return lv.n;
    unsigned long long t = 0;
    for (uint32_t k = 0; k < kSegs; ++k) t += min(lv.counts[k * kCtrStride], lv.seg_size);
    return t;
}

// XCD-aware block remap.  Hardware deals consecutive workgroups round-robin over the 8 XCDs, so
// blocks b and b+8 share an L2.  Work is handed out in chunks of kChunk consecutive blocks
// (kChunk * 64 rays = a few image rows of primary rays): XCD x processes chunks x, x+8, x+16, ...
// Consecutive rays stay on one L2 (locality for node/triangle fetches) while every XCD sees the
// same mix of cheap and expensive regions (a contiguous eighth per XCD left XCDs idle: the ceiling
// band finishes long before the heightfield band).  Bijective on [0, nb) when nb % (8*kChunk) == 0.
constexpr uint32_t kChunkRays = 4096;  // wave_rays divides it
__device__ __forceinline__ uint32_t xcd_block(uint32_t b, uint32_t chunk) {
    uint32_t x = b & 7u, k = b >> 3;
    return ((k / chunk) * 8u + x) * chunk + (k % chunk);
}
__host__ __device__ __forceinline__ uint32_t trav_blocks(uint32_t n, uint32_t wave_rays) {  // blocks for n rays, padded to whole chunk rounds
    uint32_t nb = (n + wave_rays - 1) / wave_rays, r = 8u * (kChunkRays / wave_rays);
    return (nb + r - 1) / r * r;
}

// PRIMARY: the generate stage (main.rs:412-427 + scene.rs:37-45) runs in the fetch — ray i is path
// slot i of the batch and its camera ray is built in registers; nothing is read from the queue.
// One wave per workgroup; workgroup b owns the wave_rays queue entries [b, b+1) * wave_rays.
// Five waves per SIMD for the traversal kernels (at most 96 VGPRs; they would take 100-104 and run four):
// measured on the 10 M-triangle scene 4 / 5 / 6 waves -> 8.86 / 9.17 / 7.32 Gsamples/s (six spills).
#ifndef PT_TRAV_WAVES
#define PT_TRAV_WAVES 5
#endif
template <int DEPTH, bool STATS, bool PRIMARY, bool VOTE>
__global__ __launch_bounds__(kTravBlock, PT_TRAV_WAVES) void k_extend(Scene sc, Frame fr, Queue q, const double* __restrict__ tmin,
                                                   const double* __restrict__ tmax, Live live, uint32_t wave_rays,
                                                   Hits hits, Counters* ctr) {
    __shared__ int stack[DEPTH * kTravBlock];
    __shared__ float4 s_top[(DEPTH > 8 && kTopNodes > 0) ? kTopNodes * 8 : 1];
    TopNodes top{nullptr, 0};
    if (DEPTH > 8 && kTopNodes > 0) top = load_top_nodes<kTravBlock>(sc, s_top);
    if (ctr && blockIdx.x == 0 && threadIdx.x == 0)
        atomicAdd(&ctr->rays, PRIMARY ? (unsigned long long)fr.n_on_screen * fr.n_samples : live_total(live));
    // wave w of the launch owns the wave_rays queue entries [w, w + 1) * wave_rays
    const uint32_t wib = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t block_rays = wave_rays * (kTravBlock / 64);
    uint32_t begin, end;
    RangePool pool{nullptr, nullptr, 0, 0, 0, 0};
    if (VOTE && live.cursors) {
        // persistent waves: ranges of wave_rays entries are claimed from the segments' work cursors
        pool = RangePool{live.cursors, live.counts, live.seg_size, wave_rays,
                         (blockIdx.x * (kTravBlock / 64) + wib) % kSegs, 0};
        if (!pool.next(begin, end)) return;
    } else if (live.counts) {
        // segmented queue: live entries are already spread evenly over the index range, and the
        // hardware's round-robin block placement spreads them over the XCDs
        begin = blockIdx.x * block_rays + wib * wave_rays;
        if (begin >= live.n) return;
        end = min(begin + wave_rays, live_end(live, begin));
    } else {
        // dense queue [0, n): only the first trav_blocks(n) blocks take part, remapped in chunks
        if (blockIdx.x >= trav_blocks(live.n, block_rays)) return;
        begin = xcd_block(blockIdx.x, kChunkRays / block_rays) * block_rays + wib * wave_rays;
        end = min(begin + wave_rays, live.n);
    }
    if (begin >= end) return;
    uint32_t nn = 0, nt = 0;
    auto fetch = [&](uint32_t i, D3& o, D3& d, double& t_min, double& t_max) {
            if (PRIMARY) {
                uint32_t s, p;
                slot_sp(fr, fr.grp_log2, i, s, p);
                uint32_t ty, tx;
                slot_pixel(fr, p, tx, ty);
                const uint32_t gx = fr.x0 + tx, gy = frame_row(fr, ty);
                PixelArgs a = pixel_args(gx, gy, fr.width, fr.height, fr.jitter[2 * s], fr.jitter[2 * s + 1], fr.aspect);
                o = ld3(sc.cam.eye);
                d = camera_dir(sc.cam, a.x, a.y);
                t_min = 1e-5;
                // outside the scene's screen bounds (Frame::cx0..): an empty t range, nothing can be hit
                t_max = (gx >= fr.cx0 && gx < fr.cx1 && gy >= fr.cy0 && gy < fr.cy1) ? 1.7976931348623157e308 : -1.;
                return;
            } else {
                o = mk(q.o[0][i], q.o[1][i], q.o[2][i]);
                d = mk(q.d[0][i], q.d[1][i], q.d[2][i]);
            }
            t_min = tmin ? tmin[i] : 1e-5;                    // ray.rs:21
            t_max = tmax ? tmax[i] : 1.7976931348623157e308;  // ray.rs:22 f64::MAX
        };
    auto done = [&](uint32_t i, uint32_t tri, double t) {
            hits.tri[i] = tri;
            hits.t[i] = t;
        };
    if (VOTE) {
#ifdef PT_DEBUG_TAIL
        unsigned long long dbg_t0 = wall_clock64();
#endif
        trace_range_vote<DEPTH, false, STATS>(sc, begin, end, pool, stack + threadIdx.x, top, fetch, done, nn, nt);
#ifdef PT_DEBUG_TAIL
        if (!STATS && live.cursors && lane == 0 && blockIdx.x < 8192) { g_dbg_tail[2 * blockIdx.x] = dbg_t0; g_dbg_tail[2 * blockIdx.x + 1] = wall_clock64(); }
#endif
    } else if (begin + lane < end) {  // wave_rays == 64
        uint32_t i = begin + lane;
        D3 o, d;
        double t_min, t_max;
        fetch(i, o, d, t_min, t_max);
        if (PRIMARY && t_max < 0.) done(i, PT_MISS, t_max);  // off screen
        else done(i, traverse<DEPTH, false, STATS>(sc, o, d, t_min, t_max, stack + threadIdx.x, nn, nt, top), t_max);
    }
    if (STATS) {
        // wave-level reduction, one atomic pair per wave
        for (int off = 32; off > 0; off >>= 1) {
            nn += __shfl_down(nn, off);
            nt += __shfl_down(nt, off);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&ctr->nodes, (unsigned long long)nn);
            atomicAdd(&ctr->tris, (unsigned long long)nt);
        }
    }
}

// color_result += add for a slot that may not hold a value yet (Sample3).  One lane per slot and
// launch, so only this lane can set the slot's bit; other bits of the word change under it.
__device__ __forceinline__ void radiance_add(const Sample3& ls, uint32_t slot, D3 add) {
    const unsigned long long bit = 1ull << (slot & 63u);
    double* c = ls.c + (size_t)slot * 3;
    if (ls.lit[slot >> 6] & bit) {
        c[0] += add.x; c[1] += add.y; c[2] += add.z;
    } else {
        c[0] = 0. + add.x; c[1] = 0. + add.y; c[2] = 0. + add.z;
        atomicOr(ls.lit + (slot >> 6), bit);
    }
}

template <int DEPTH, bool VOTE>
__global__ __launch_bounds__(kTravBlock, PT_TRAV_WAVES) void k_shadow(Scene sc, ShadowQueue sq, Live live, uint32_t wave_rays, Sample3 ls,
                                                   Counters* ctr) {
    __shared__ int stack[DEPTH * kTravBlock];
    __shared__ float4 s_top[(DEPTH > 8 && kTopNodes > 0) ? kTopNodes * 8 : 1];
    TopNodes top{nullptr, 0};
    if (DEPTH > 8 && kTopNodes > 0) top = load_top_nodes<kTravBlock>(sc, s_top);
    if (ctr && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&ctr->shadow_rays, live_total(live));
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t begin, end;
    RangePool pool{nullptr, nullptr, 0, 0, 0, 0};
    if (VOTE && live.cursors) {
        pool = RangePool{live.cursors, live.counts, live.seg_size, wave_rays,
                         (blockIdx.x * (kTravBlock / 64) + (threadIdx.x >> 6)) % kSegs, 0};
        if (!pool.next(begin, end)) return;
    } else {
        begin = (blockIdx.x * (kTravBlock / 64) + (threadIdx.x >> 6)) * wave_rays;
        if (begin >= live.n) return;
        end = min(begin + wave_rays, live_end(live, begin));
    }
    if (begin >= end) return;
    uint32_t nn = 0, nt = 0;
    auto fetch = [&](uint32_t i, D3& o, D3& d, double& t_min, double& t_max) {
            o = mk(sq.o[0][i], sq.o[1][i], sq.o[2][i]);
            d = mk(sq.d[0][i], sq.d[1][i], sq.d[2][i]);
            t_min = 1e-5;
            t_max = sq.tmax[i];
        };
    auto done = [&](uint32_t i, uint32_t tri, double) {
            if (tri != PT_MISS) return;
            // unoccluded: the light sample counts (one shadow ray per path slot and launch)
            radiance_add(ls, sq.slot[i], mk(sq.c[0][i], sq.c[1][i], sq.c[2][i]));
        };
    if (VOTE) {
        trace_range_vote<DEPTH, true, false>(sc, begin, end, pool, stack + threadIdx.x, top, fetch, done, nn, nt);
    } else if (begin + lane < end) {  // wave_rays == 64
        uint32_t i = begin + lane;
        D3 o, d;
        double t_min, t_max;
        fetch(i, o, d, t_min, t_max);
        done(i, traverse<DEPTH, true, false>(sc, o, d, t_min, t_max, stack + threadIdx.x, nn, nt, top), t_max);
    }
}


}  // namespace ptd
