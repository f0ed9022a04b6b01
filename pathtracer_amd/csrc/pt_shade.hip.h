// pt_shade.hip.h — generate, texture lookup (mipmap.rs), BRDFs (material.rs, microfacet.rs), shade_vertex, queue compaction, k_shade, k_path (fused extend + shade), k_accumulate, k_jitter
// Part of the single translation unit pt_kernels.hip (device code is inlined across these files);
// included from there, in order: pt_math.hip.h, pt_traverse.hip.h, pt_shade.hip.h.
#pragma once
#include "pt_device.h"

#include "../../include/pt.h"

namespace ptd {

// ------------------------------------------------------------------------------------------------
// generate — in the render path this stage is fused into the prologue of the bounce-0 extend and
// shade kernels (PRIMARY / FIRST); standalone it exists as the parity entry below.
// parity entry: the same rays as AoS pt_ray records (frame fr.sample_base, n_samples == 1)
__global__ __launch_bounds__(kBlock) void k_generate_rays(Scene sc, Frame fr, double* __restrict__ rays) {
    uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= fr.n_tile_pixels) return;
    uint32_t ty = fdiv(p, fr.div_tile_w), tx = p - ty * fr.tile_w;
    PixelArgs a = pixel_args(fr.x0 + tx, frame_row(fr, ty), fr.width, fr.height, fr.jitter[0], fr.jitter[1], fr.aspect);
    D3 dir = camera_dir(sc.cam, a.x, a.y);
    double* r = rays + (size_t)p * 8;
    r[0] = sc.cam.eye[0]; r[1] = sc.cam.eye[1]; r[2] = sc.cam.eye[2];
    r[3] = dir.x; r[4] = dir.y; r[5] = dir.z;
    r[6] = 1e-5;
    r[7] = 1.7976931348623157e308;
}

// ------------------------------------------------------------------------------------------------
// mipmap.rs — texture lookup.  Texels are f64 RGB built on the host with the reference's
// arithmetic (u8/255, 2x2 means), rows stored in the reference's `data` order.
__device__ __forceinline__ uint32_t f64_as_usize_sat(double x) {  // Rust `as usize` (saturating, NaN -> 0)
    if (!(x > 0.)) return 0u;
    if (x >= 4294967295.0) return 0xFFFFFFFFu;  // any value past the clamp in get_pixel behaves alike
    return (uint32_t)x;
}
__device__ __forceinline__ D3 get_pixel(const Texture& tx, uint32_t level, uint32_t w, uint32_t h, uint32_t s,
                                        uint32_t t) {
    uint32_t tt = min(t, h - 1), ss = min(s, w - 1);                  // mipmap.rs:44-46
    const double* p = tx.texels + ((size_t)tx.offset[level] + (size_t)(h - tt - 1) * w + ss) * 3;  // mipmap.rs:52-55
    return mk(p[0], p[1], p[2]);
}
__device__ __forceinline__ uint32_t sat_inc(uint32_t a) { return a == 0xFFFFFFFFu ? a : a + 1u; }
__device__ D3 tex_triangle(const Texture& tx, uint32_t level, double u, double v) {  // mipmap.rs:100-115
    uint32_t w = max(tx.width0 >> level, 1u), h = max(tx.height0 >> level, 1u);
    double s = u * (double)w - 0.5;
    double t = v * (double)h - 0.5;
    uint32_t s0 = f64_as_usize_sat(s), t0 = f64_as_usize_sat(t);
    double ds = s - (double)s0;
    double dt = t - (double)t0;
    return (1. - ds) * (1. - dt) * get_pixel(tx, level, w, h, s0, t0) +
           (1. - ds) * dt * get_pixel(tx, level, w, h, s0, sat_inc(t0)) +
           ds * (1. - dt) * get_pixel(tx, level, w, h, sat_inc(s0), t0) +
           ds * dt * get_pixel(tx, level, w, h, sat_inc(s0), sat_inc(t0));
}
__device__ D3 sample_mipmap(const Texture& tx, double u, double v, double dudx, double dvdx, double dudy,
                            double dvdy) {  // mipmap.rs:117-141
    uint32_t n_levels = tx.n_levels;
    double width = fmax(fmax(fabs(dudx), fabs(dvdx)), fmax(fabs(dudy), fabs(dvdy)));
    double level = (double)n_levels - 1. + log2(fmax(width, 1e-8));
    if (level < 0.) {
        return tex_triangle(tx, 0, u, v);
    } else if (level >= (double)n_levels - 1.) {
        const double* p = tx.texels + (size_t)tx.offset[n_levels - 1] * 3;
        return mk(p[0], p[1], p[2]);
    } else {
        uint32_t ilevel = f64_as_usize_sat(floor(level));
        double delta = level - (double)ilevel;
        D3 a = tex_triangle(tx, ilevel, u, v), b = tex_triangle(tx, ilevel + 1, u, v);
        return (1. - delta) * a + delta * b;
    }
}

// surface.rs:112-123
__device__ __forceinline__ void solve2x2(double a00, double a01, double a10, double a11, double b0, double b1,
                                         double& x0, double& x1) {
    double det = a00 * a11 - a01 * a10;
    if (fabs(det) < 1e-10) { x0 = 0.; x1 = 0.; return; }
    double r0 = (a11 * b0 - a01 * b1) / det;
    double r1 = (a00 * b1 - a10 * b0) / det;
    if (r0 != r0 || r1 != r1) { x0 = 0.; x1 = 0.; return; }
    x0 = r0; x1 = r1;
}

// The differential rays of a ray (ray.rs:9-12).
struct RayDiffs {
    D3 rxo, ryo, rxd, ryd;
};
// surface.rs:65-84: where the differential rays meet the tangent plane of the hit.  false (and
// zeros) when either distance is infinite or NaN, which is what the all-zero differential rays of
// a Ray::new ray (main.rs:314) give.
__device__ __forceinline__ bool surface_dp(D3 normal, D3 position, const RayDiffs& rd, D3& dpdx, D3& dpdy) {
    double dd = dot(normal, position);
    double tx = -(dot(normal, rd.rxo) - dd) / dot(normal, rd.rxd);
    double ty = -(dot(normal, rd.ryo) - dd) / dot(normal, rd.ryd);
    dpdx = dpdy = mk(0., 0., 0.);
    if (isinf(tx) || tx != tx || isinf(ty) || ty != ty) return false;
    D3 px = rd.rxo + tx * rd.rxd;
    D3 py = rd.ryo + ty * rd.ryd;
    dpdx = px - position;
    dpdy = py - position;
    return true;
}
// surface.rs:86-108: the texture-space derivatives from dpdu / dpdv and the screen-space dpdx / dpdy
__device__ __forceinline__ void uv_derivs(D3 normal, D3 dpdu, D3 dpdv, D3 dpdx, D3 dpdy, double& dudx, double& dvdx,
                                          double& dudy, double& dvdy) {
    int d0, d1;
    if (fabs(normal.x) > fabs(normal.y) && fabs(normal.x) > fabs(normal.z)) { d0 = 1; d1 = 2; }
    else if (fabs(normal.y) > fabs(normal.z)) { d0 = 0; d1 = 2; }
    else { d0 = 0; d1 = 1; }
    double a00 = comp(dpdu, d0), a01 = comp(dpdv, d0), a10 = comp(dpdu, d1), a11 = comp(dpdv, d1);
    solve2x2(a00, a01, a10, a11, comp(dpdx, d0), comp(dpdx, d1), dudx, dvdx);
    solve2x2(a00, a01, a10, a11, comp(dpdy, d0), comp(dpdy, d1), dudy, dvdy);
}
// geometry/mod.rs:76-132: the derivatives of a sphere's (phi, theta) parametrisation at a hit and, from the
// fundamental forms, of its normal.  `hit_local` = position - centre.  At the poles the reference's own
// 0 * inf gives NaN; kept.
struct SphereDerivs {
    D3 dpdu, dpdv, dndu, dndv;
};
__device__ SphereDerivs sphere_derivs(D3 hit_local, double radius) {
    double inner = hit_local.z / radius;
    if (inner < -1. || inner >= 1.) inner = 1.;
    double theta = acos(inner);
    double z_radius = sqrt(hit_local.x * hit_local.x + hit_local.y * hit_local.y);
    double inv_z_radius = 1. / z_radius;
    double cos_phi = hit_local.x * inv_z_radius;
    double sin_phi = hit_local.y * inv_z_radius;
    SphereDerivs r;
    r.dpdu = mk(-2. * PT_PI * hit_local.y, 2. * PT_PI * hit_local.x, 0.);
    r.dpdv = mk(hit_local.z * cos_phi, hit_local.z * sin_phi, -radius * sin(theta)) * -PT_PI;
    D3 d2Pduu = (-2. * PT_PI * 2. * PT_PI) * mk(hit_local.x, hit_local.y, 0.);
    D3 d2Pduv = ((-PT_PI) * hit_local.z * 2. * PT_PI) * mk(-sin_phi, cos_phi, 0.);
    D3 d2Pdvv = (-(-PT_PI) * (-PT_PI)) * mk(hit_local.x, hit_local.y, hit_local.z);
    double E = dot(r.dpdu, r.dpdu), F = dot(r.dpdu, r.dpdv), G = dot(r.dpdv, r.dpdv);
    D3 N = normalize(cross(r.dpdu, r.dpdv));
    double e = dot(N, d2Pduu), f = dot(N, d2Pduv), g = dot(N, d2Pdvv);
    double invEGF2 = 1. / (E * G - F * F);
    r.dndu = ((f * F - e * G) * invEGF2) * r.dpdu + ((e * F - f * E) * invEGF2) * r.dpdv;
    r.dndv = ((g * F - f * G) * invEGF2) * r.dpdu + ((f * F - g * E) * invEGF2) * r.dpdv;
    return r;
}
// Diffuse colour of a textured Lambertian at a hit: uv interpolation + dpdu/dpdv
// (triangle_mesh.rs:128-138), ray differentials (surface.rs:86-109), material.rs:100-111.
// `has_dp`: dpdx / dpdy hold surface_dp's result for a ray that carries differentials (a primary
// ray with CONFIG.use_diffs, or any ray under PT_FLAG_PROPAGATE_DIFFS); otherwise the texture
// derivatives are zero (surface.rs:72-80).
__device__ D3 textured_color(const Scene& sc, const Material& m, const TriV& tr, uint32_t leaf, D3 o, D3 d,
                             D3 normal, bool has_dp, D3 dpdx, D3 dpdy) {
    double uv[6];
    if (sc.uvs) {
        const float2* up = sc.uvs + (size_t)leaf * 3;
        float2 a = up[0], b = up[1], c = up[2];
        uv[0] = (double)a.x; uv[1] = (double)a.y; uv[2] = (double)b.x; uv[3] = (double)b.y; uv[4] = (double)c.x; uv[5] = (double)c.y;
    } else {
        uv[0] = 0.; uv[1] = 0.; uv[2] = 1.0; uv[3] = 0.; uv[4] = 1.; uv[5] = 1.;
    }
    // barycentrics, triangle_mesh.rs:118-122
    D3 e0 = tr.v1 - tr.v0;
    D3 e1 = tr.v0 - tr.v2;
    D3 gn = cross(e0, e1);
    D3 tmp = cross(tr.v0 - o, d);
    double denom = 1. / dot(d, gn);
    double v = dot(tmp, e1) * denom;
    double w = dot(tmp, e0) * denom;
    double u = 1. - v - w;
    double tu = uv[0] * u + uv[2] * v + uv[4] * w;
    double tv = uv[1] * u + uv[3] * v + uv[5] * w;
    double dudx = 0., dvdx = 0., dudy = 0., dvdy = 0.;
    if (has_dp) {
        D3 dp02 = tr.v0 - tr.v2, dp12 = tr.v1 - tr.v2;
        double duv02x = uv[0] - uv[4], duv02y = uv[1] - uv[5], duv12x = uv[2] - uv[4], duv12y = uv[3] - uv[5];
        double determinant = duv02x * duv12y - duv02y * duv12x;
        double inverse_det = 1. / determinant;
        D3 dpdu = (duv12y * dp02 - duv02y * dp12) * inverse_det;
        D3 dpdv = (-duv12x * dp02 + duv02x * dp12) * inverse_det;
        uv_derivs(normal, dpdu, dpdv, dpdx, dpdy, dudx, dvdx, dudy, dvdy);
    }
    return sample_mipmap(sc.texs[m.tex], tu, tv, dudx, dvdx, dudy, dvdy);
}

// The same for an analytic primitive: its uv stays (0, 0) (Surface::default, surface.rs:35); a plane has no
// derivatives (Surface::new, geometry/mod.rs:45), a sphere those of sphere_derivs (mod.rs:136-145).
__device__ D3 prim_textured_color(const Scene& sc, const Material& m, const Prim& pr, D3 position, D3 normal,
                                  bool has_dp, D3 dpdx, D3 dpdy) {
    double dudx = 0., dvdx = 0., dudy = 0., dvdy = 0.;
    if (has_dp && pr.kind == PT_PRIM_SPHERE) {
        const SphereDerivs sd = sphere_derivs(position - ld3(pr.p), pr.radius);
        uv_derivs(normal, sd.dpdu, sd.dpdv, dpdx, dpdy, dudx, dvdx, dudy, dvdy);
    }
    return sample_mipmap(sc.texs[m.tex], 0., 0., dudx, dvdx, dudy, dvdy);
}

// ------------------------------------------------------------------------------------------------
// microfacet.rs — Beckmann + conductor Fresnel.  The Fresnel terms are fixed by microfacet.rs:45
// (ior 32, absorption 0.1, ior_t 1) and identical in the three channels, so one channel is computed.
__device__ __forceinline__ double conductor32(double cos_theta_i) {  // microfacet.rs:264-285
    const double eta = 1. / 32.;
    const double etak = 0.1 / 32.;
    double cosThetaI2 = cos_theta_i * cos_theta_i;
    double sinThetaI2 = 1. * (1. - cosThetaI2);
    double eta2 = eta * eta;
    double etak2 = etak * etak;
    double t0 = eta2 - etak2 - sinThetaI2;
    double a2plusb2 = sqrt(t0 * t0 + 4. * eta2 * etak2);
    double t1 = a2plusb2 + 1. * cosThetaI2;
    double a = sqrt(0.5 * (a2plusb2 + t0));
    double t2 = 2. * cos_theta_i * a;
    double Rs = (t1 - t2) / (t1 + t2);
    double t3 = cosThetaI2 * a2plusb2 + sinThetaI2 * sinThetaI2;
    double t4 = t2 * sinThetaI2;
    double Rp = Rs * (t3 - t4) / (t3 + t4);
    return 0.5 * (Rp + Rs);
}
__device__ __forceinline__ double beckmann_D(double width, D3 m, D3 n) {  // microfacet.rs:161-173
    double cos_theta_m = dot(m, n);
    if (cos_theta_m <= 0.) return 0.;
    double c2 = cos_theta_m * cos_theta_m;
    double cos4theta_m = c2 * c2;
    double theta_m = acos(cos_theta_m);
    double tan_m = tan(theta_m);
    double tan2theta_m = tan_m * tan_m;
    double alpha2 = width * width;
    return exp(-tan2theta_m / alpha2) / (PT_PI * alpha2 * cos4theta_m);
}
__device__ __forceinline__ double beckmann_G1(double width, D3 w, D3 m, D3 n) {  // microfacet.rs:175-189
    double cos_theta_v = dot(w, n);
    if (dot(w, m) / cos_theta_v <= 0.0) return 0.;
    double theta_v = acos(cos_theta_v);
    double a = 1. / (width * tan(theta_v));
    if (a >= 1.6) return 1.;
    return (3.535 * a + 2.181 * a * a) / (1. + 2.276 * a + 2.577 * a * a);
}
// microfacet.rs:192-243 GGX; tan2_theta = max(0, 1 - c^2) / c^2 (microfacet.rs:366-381)
__device__ __forceinline__ double ggx_D(double alpha, D3 h, D3 n) {
    double cos_theta = dot(h, n);
    if (cos_theta <= 0.0) return 0.;
    double alpha_g_sqr = alpha * alpha;
    double c2 = cos_theta * cos_theta;
    double tan_theta_sqr = fmax(0., 1. - c2) / c2;
    if (isinf(tan_theta_sqr)) return 0.;
    double alpha_tan2 = alpha_g_sqr + tan_theta_sqr;
    return alpha_g_sqr / (PT_PI * (c2 * c2) * (alpha_tan2 * alpha_tan2));
}
__device__ __forceinline__ double ggx_G1(double alpha, D3 w, D3 vh) {
    double cos_theta = dot(w, vh);
    double cos_theta_sqr = cos_theta * cos_theta;
    double tan_theta_sqr = (1. - cos_theta_sqr) / cos_theta_sqr;
    return 2. / (1. + sqrt(1. + alpha * alpha * tan_theta_sqr));
}
__device__ __forceinline__ double facet_D(bool ggx, double width, D3 m, D3 n) {
    return ggx ? ggx_D(width, m, n) : beckmann_D(width, m, n);
}
__device__ D3 glossy_f(double width, D3 reflectance, D3 wo, D3 wi, D3 n, bool ggx) {  // microfacet.rs:64-93
    D3 wh = normalize(wo + wi);
    double cos_theta_o = fabs(dot(wo, n));
    double cos_theta_i = fabs(dot(wi, n));
    if (cos_theta_i == 0. || cos_theta_o == 0.) return mk(0., 0., 0.);
    if (wh.x == 0. && wh.y == 0. && wh.z == 0.) return mk(0., 0., 0.);
    double fresnel = conductor32(dot(wo, wh));
    double dd = facet_D(ggx, width, wh, n);
    D3 hv = normalize(wi + wo);  // microfacet.rs:105
    double t = ggx ? ggx_G1(width, wo, hv) * ggx_G1(width, wi, hv)
                   : beckmann_G1(width, wo, hv, n) * beckmann_G1(width, wi, hv, n);
    double g = t < 0. ? 0. : (t > 1. ? 1. : t);
    return reflectance * dd * g * mk(fresnel, fresnel, fresnel) / (4. * cos_theta_i * cos_theta_o);
}
__device__ __forceinline__ double glossy_pdf_dir(double width, D3 wo, D3 wi, D3 n, bool ggx) {
    D3 h = normalize(wo + wi);
    return facet_D(ggx, width, h, n) * fabs(dot(h, n)) / (4. * dot(wo, h));
}
__device__ __forceinline__ bool is_glossy(int kind) { return kind == PT_MAT_GLOSSY || kind == PT_MAT_GLOSSY_GGX; }

struct Brdf {
    D3 f, wi;
    double pdf;
};
// vec3.rs:142-146
__device__ __forceinline__ D3 refract(D3 self, D3 normal, double eta) {
    double cos_i = -dot(self, normal);
    double sin_sqr = eta * eta * (1. - cos_i * cos_i);
    return self * eta + normal * (eta * cos_i - sqrt(1. - sin_sqr));
}
// FresnelDielectric::dielectric for an object of index `ior` in air (microfacet.rs:306-358)
__device__ __forceinline__ double dielectric(double ior, double cos_theta_i) {
    double ior_i = 1., ior_t = ior;
    if (cos_theta_i < 0.) {  // inside: the object's medium is the incident one
        ior_i = ior; ior_t = 1.;
        cos_theta_i = fabs(cos_theta_i);
    }
    double eta = ior_i / ior_t;
    double sin_theta_i = sqrt(fmax(0., 1. - cos_theta_i * cos_theta_i));
    double sin_theta_t = ior_i / ior_t * sin_theta_i;
    double sin_theta_t_sqr = eta * eta * (1. - cos_theta_i * cos_theta_i);
    if (sin_theta_t >= 1.) return 1.;  // total internal reflection
    double cos_theta_t = sqrt(fmax(0., 1. - sin_theta_t_sqr));
    double fresnel_par = ((ior_t * cos_theta_i) - (ior_i * cos_theta_t)) / ((ior_t * cos_theta_i) + (ior_i * cos_theta_t));
    double fresnel_perp = ((ior_i * cos_theta_i) - (ior_t * cos_theta_t)) / ((ior_i * cos_theta_i) + (ior_t * cos_theta_t));
    return (fresnel_par * fresnel_par + fresnel_perp * fresnel_perp) * 0.5;
}
// Mirror (main.rs:178-185) and Transparent (main.rs:117-142; one of its two branches, reflection
// with probability R from the bounce's first draw).  The radiance passes unchanged either way.
// `lobe`: 1 = reflected, 2 = refracted (which of reflect_diff / refract_diff spawns the ray).
__device__ __forceinline__ D3 specular_sample(const Material& m, D3 ray_dir, D3 n, double xi1, int& lobe) {
    lobe = 1;
    if (m.kind == PT_MAT_MIRROR) return reflect(ray_dir, n);
    double cos_theta = dot(-ray_dir, n);
    double eta = cos_theta < 0. ? m.ior / 1. : 1. / m.ior;  // FresnelDielectric::eta, microfacet.rs:318-326
    double fresnel_r = dielectric(m.ior, cos_theta);
    D3 normal = cos_theta < 0. ? -n : n;
    if (xi1 < fresnel_r) return reflect(ray_dir, normal);
    lobe = 2;
    return refract(ray_dir, normal, eta);
}
// The differential rays of the ray the legacy integrator spawns at a vertex (main.rs:192-281,
// commented out in the reference): reflect_diff for a Lambertian / Glossy sample (lobe 0) and a
// specular reflection (lobe 1), refract_diff for a refraction (lobe 2).  dndu = dndv = 0 on every
// surface the reference builds for a triangle or with Surface::new (surface.rs:40-41), so there the
// dndx / dndy terms, which add only signed zeros, are dropped; a sphere sets them (geometry/mod.rs:131-132):
// `curved` with dndx = dndu dudx + dndv dvdx, dndy likewise (main.rs:210-211, 268-269).
__device__ RayDiffs spawn_diffs(const Material& m, int lobe, D3 position, D3 surf_normal, D3 dpdx, D3 dpdy, D3 inc_dir,
                                const RayDiffs& inc, D3 wi, double diff_size, bool roughdiff, bool curved = false,
                                D3 dndx = D3{0., 0., 0.}, D3 dndy = D3{0., 0., 0.}) {
    RayDiffs nr;
    D3 wo = -inc_dir;
    nr.rxo = position + dpdx;
    nr.ryo = position + dpdy;
    if (lobe == 0) {  // main.rs:242-257
        if (roughdiff && is_glossy(m.kind)) diff_size = diff_size * m.roughness / 0.8;
        D3 v0;  // vec3.rs:175-184 create_tangent_vectors
        if (fabs(wi.x) > fabs(wi.y)) v0 = mk(-wi.z, 0., wi.x) / sqrt(wi.x * wi.x + wi.z * wi.z);
        else v0 = mk(0., wi.z, -wi.y) / sqrt(wi.y * wi.y + wi.z * wi.z);
        D3 v1 = cross(wi, v0);
        nr.rxd = normalize(wi + diff_size * v0);
        nr.ryd = normalize(wi + diff_size * v1);
        return nr;
    }
    double cos_theta = dot(wo, surf_normal);
    D3 normal = cos_theta < 0. ? -surf_normal : surf_normal;
    D3 dwodx = -inc.rxd - wo;
    D3 dwody = -inc.ryd - wo;
    if (curved) {  // main.rs:210-221, 268-275 in full
        double dDNdx = dot(dwodx, normal) + dot(wo, dndx);
        double dDNdy = dot(dwody, normal) + dot(wo, dndy);
        if (lobe == 1) {
            nr.rxd = wi - dwodx + 2. * (dot(wo, normal) * dndx + dDNdx * normal);
            nr.ryd = wi - dwody + 2. * (dot(wo, normal) * dndy + dDNdy * normal);
        } else {
            double eta = cos_theta < 0. ? m.ior / 1. : 1. / m.ior;
            double mu = eta * dot(-wo, normal) - dot(wi, normal);
            double dmudx = (eta - (eta * eta * dot(-wo, normal)) / dot(wi, normal)) * dDNdx;
            double dmudy = (eta - (eta * eta * dot(-wo, normal)) / dot(wi, normal)) * dDNdy;
            nr.rxd = wi + eta * dwodx - (mu * dndx + dmudx * normal);
            nr.ryd = wi + eta * dwody - (mu * dndy + dmudy * normal);
        }
        return nr;
    }
    double dDNdx = dot(dwodx, normal);
    double dDNdy = dot(dwody, normal);
    if (lobe == 1) {  // main.rs:258-276
        nr.rxd = wi - dwodx + 2. * (dDNdx * normal);
        nr.ryd = wi - dwody + 2. * (dDNdy * normal);
    } else {  // main.rs:198-221 as written (pbrt-v3's form; its signs are not those of d(refract), kept as the reference has them)
        double eta = cos_theta < 0. ? m.ior / 1. : 1. / m.ior;
        double dmudx = (eta - (eta * eta * dot(-wo, normal)) / dot(wi, normal)) * dDNdx;
        double dmudy = (eta - (eta * eta * dot(-wo, normal)) / dot(wi, normal)) * dDNdy;
        nr.rxd = wi + eta * dwodx - dmudx * normal;
        nr.ryd = wi + eta * dwody - dmudy * normal;
    }
    return nr;
}
// material.rs:9-23: the cosine-weighted direction, with the basis of the normal given (TriBasis) or built here.
// (f = rho / pi and pdf = (n . wi) / pi of material.rs:37-58 are one multiplication each, written at their uses.)
__device__ __forceinline__ D3 lambertian_dir(D3 n, const TriBasis* basis, double xi1, double xi2) {
    double cos_theta = sqrt(xi1);
    double sin_theta = sqrt(1.0 - xi1);
    double phi = 2.0 * PT_PI * xi2;
    double sp, cp;
    sincos(phi, &sp, &cp);
    const D3 local = mk(sin_theta * cp, sin_theta * sp, cos_theta);
    if (basis) return rotate_with(local, ld3(basis->t1), ld3(basis->t2), n);
    return rotate_to(local, n);
}
// microfacet.rs:52-62 + 147-159
__device__ Brdf glossy_sample(double width, D3 reflectance, D3 ray_dir, D3 n, double xi1, double xi2, bool ggx) {
    D3 wo = -ray_dir;
    double phi, cos_theta, sin_theta;
    if (ggx) {  // microfacet.rs:203-217
        double tan_theta_sqr = width * width * xi1 / (1. - xi1);
        phi = 2.0 * xi2 * PT_PI;
        double cos_theta_sqr = 1. / (1. + tan_theta_sqr);
        cos_theta = sqrt(cos_theta_sqr);
        sin_theta = sqrt(1. - cos_theta_sqr);
    } else {
        double tan_theta_sqr = -width * width * log(1. - xi1);
        phi = 2.0 * PT_PI * xi2;
        cos_theta = 1. / sqrt(1. + tan_theta_sqr);
        sin_theta = sqrt(fmax(0., 1. - cos_theta * cos_theta));
    }
    double sp, cp;
    sincos(phi, &sp, &cp);
    D3 h = normalize(rotate_to(mk(sin_theta * cp, sin_theta * sp, cos_theta), n));
    D3 wi = reflect(-wo, h);
    double pdf = facet_D(ggx, width, h, n) * fabs(dot(h, n)) / (4. * dot(wo, h));
    D3 f = glossy_f(width, reflectance, wo, wi, n, ggx);
    return {f, wi, pdf};
}

// Everything main.rs:303-311 needs at a hit.
struct ShadePoint {
    D3 position, normal, emission;
    Brdf bs;
    D3 diffuse;  // Lambertian colour after the texture lookup (NEE evaluates the BRDF with it)
};
// triangle_mesh.rs:141: normal = -(e0 x e1).normalize(), e0 = v1 - v0, e1 = v0 - v2
__device__ __forceinline__ D3 tri_unit_normal(const TriV& tr) { return -normalize(cross(tr.v1 - tr.v0, tr.v0 - tr.v2)); }
__global__ void k_build_tri_shade(Scene sc, TriShade* out, TriBasis* basis) {
    uint32_t leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= sc.n_tris) return;
    D3 n = tri_unit_normal(load_tri(sc, leaf));
    out[leaf].n[0] = n.x; out[leaf].n[1] = n.y; out[leaf].n[2] = n.z;
    out[leaf].mat = (uint32_t)__float_as_int(sc.tris[(size_t)leaf * 3 + 1].w);
    out[leaf].pad = 0;
    if (basis) {
        D3 t1, t2;
        rotate_basis(n, t1, t2);
        basis[leaf].t1[0] = t1.x; basis[leaf].t1[1] = t1.y; basis[leaf].t1[2] = t1.z;
        basis[leaf].t2[0] = t2.x; basis[leaf].t2[1] = t2.y; basis[leaf].t2[2] = t2.z;
    }
}
__device__ __forceinline__ uint32_t hit_material(const Scene& sc, uint32_t leaf) {
    return leaf >= PT_PRIM_BASE ? sc.prims[leaf - PT_PRIM_BASE].mat : sc.tri_shade[leaf].mat;
}
__device__ __forceinline__ void hit_geometry(const Scene& sc, uint32_t leaf, D3 o, D3 d, double t, uint32_t& mat,
                                             D3& position, D3& normal) {
    position = o + d * t;  // triangle_mesh.rs:140
    if (leaf >= PT_PRIM_BASE) {  // Surface::new(position, normal), geometry/mod.rs:46,71-72
        const Prim& pr = sc.prims[leaf - PT_PRIM_BASE];
        mat = pr.mat;
        normal = pr.kind == PT_PRIM_PLANE ? ld3(pr.n) : normalize(position - ld3(pr.p));
        return;
    }
    const TriShade& ts = sc.tri_shade[leaf];
    mat = ts.mat;
    normal = ld3(ts.n);
}
// scene.rs:47-54: the differential rays of the camera ray of `pixel` in frame `s` of the batch
// (gx, gy: the pixel's column and row in the image - the caller has them, a division of the pixel index by the width
// costs ~50 instructions; the pixel table where the call has one)
__device__ __forceinline__ RayDiffs camera_diffs(const Scene& sc, const Frame& fr, D3 eye, uint32_t gx, uint32_t gy, uint32_t s) {
    PixelArgs a = pixel_args(fr, gx, gy, fr.jitter[2 * s], fr.jitter[2 * s + 1]);
    return {eye, eye, camera_dir(sc.cam, a.x + a.ddx, a.y), camera_dir(sc.cam, a.x, a.y + a.ddy)};
}
__device__ __forceinline__ D3 diffuse_of(const Scene& sc, const Material& m, uint32_t leaf, D3 o,
                                         D3 d, D3 position, D3 normal, const Frame& fr, bool primary,
                                         uint32_t gx, uint32_t gy, uint32_t s) {
    if (m.tex < 0) return ld3(m.color);
    bool has_dp = false;
    D3 dpdx = mk(0., 0., 0.), dpdy = dpdx;
    if (leaf >= PT_PRIM_BASE) {
        const Prim& pr = sc.prims[leaf - PT_PRIM_BASE];
        if (pr.kind == PT_PRIM_SPHERE && primary && (fr.flags & PT_FLAG_USE_DIFFS))
            has_dp = surface_dp(normal, position, camera_diffs(sc, fr, o, gx, gy, s), dpdx, dpdy);
        return prim_textured_color(sc, m, pr, position, normal, has_dp, dpdx, dpdy);
    }
    const TriV tr = load_tri(sc, leaf);  // only a textured hit needs the vertices again (uv, dpdu, dpdv)
    if (primary && (fr.flags & PT_FLAG_USE_DIFFS))
        has_dp = surface_dp(normal, position, camera_diffs(sc, fr, o, gx, gy, s), dpdx, dpdy);
    return textured_color(sc, m, tr, leaf, o, d, normal, has_dp, dpdx, dpdy);
}

// ------------------------------------------------------------------------------------------------
// block-wide compaction: wave64 ballot + mbcnt prefix, one atomic per block
template <int BLOCK = kBlock>
__device__ __forceinline__ uint32_t compact_slot(bool alive, uint32_t* counter, uint32_t* lds /* BLOCK/64 + 1 */) {
    unsigned long long m = __ballot(alive);
    uint32_t lane_prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) {
            uint32_t c = lds[w];
            lds[w] = total;
            total += c;
        }
        lds[BLOCK / 64] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t pos = lds[BLOCK / 64] + lds[wave] + lane_prefix;
    __syncthreads();
    return pos;
}

// wave-wide compaction for one-wave workgroups: one atomic per wave
__device__ __forceinline__ uint32_t compact_wave(bool alive, uint32_t* counter) {
    unsigned long long m = __ballot(alive);
    uint32_t lane_prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(counter, (uint32_t)__popcll(m));
    return __shfl(base, 0) + lane_prefix;
}

// Last-bounce cull.  The ray spawned at the second-to-last vertex is traced only to add
// T * Le of its closest hit (main.rs:307) or T * background (main.rs:299).  With a black background
// that is +-0 - which the accumulation ignores - unless the closest hit is an emitter, and it cannot
// be one if the ray fails the triangle test of every emitter.  Such a ray is dropped here, at its
// producer: it is never queued, read back or traced, and the image is bit-identical.  Non-finite
// rays and throughputs are kept (NaN * 0 is not 0; the reference accumulates those).
__device__ __forceinline__ bool finite3(D3 a) { return __builtin_isfinite(a.x) && __builtin_isfinite(a.y) && __builtin_isfinite(a.z); }  // (v_cmp_class_f64)
// The cull may keep a ray that then misses (it is traced and adds nothing), never drop one that hits: so it runs
// tri_test_core's rejects without its three divisions (a wave of 64 incoherent rays nearly always holds a lane that
// would need them).  False only where tri_test_core(p, d, 1e-5, f64::MAX) is certain to reject:
__device__ __forceinline__ bool tri_may_hit(const TriPre& p, D3 d) {
    const double dn = dot(d, p.normal);
    if (fabs(dn) < 1.0e-12) return false;      // its own first reject
    if (!(fabs(dn) <= 0x1p500)) return true;   // outside the guard of the sign tests (or NaN): not decided here
    const D3 tmp = cross(p.vo, d);
    const double a = dot(tmp, p.e1), b = dot(tmp, p.e0);
    const bool dneg = dn < 0.;
    if (fabs(a) >= 0x1p-500 && (a < 0.) != dneg) return false;          // v < 0
    if (fabs(b) >= 0x1p-500 && (b < 0.) != dneg) return false;          // w < 0
    if (fabs(p.num) >= 0x1p-500 && (p.num < 0.) != dneg) return false;  // t < 0
    // v + w > 1: v = fl(a fl(1 / dn)), w likewise and their rounded sum are within 3 x 2^-53 (relative) of
    // (a + b) / dn, so a margin of 2^-40 decides it for every rounding (an infinite a or b has v + w = inf)
    if (fabs(a) + fabs(b) > fabs(dn) * (1. + 0x1p-40)) return false;
    // t < t_min = 1e-5, the same way (a ray that starts ON an emitter: every path whose vertex is a light)
    if (fabs(p.num) < fabs(dn) * (1e-5 * (1. - 0x1p-40))) return false;
    return true;  // (the bands of relative width 2^-40 around v + w = 1 and t = 1e-5 are left to the traversal)
}
__device__ __forceinline__ bool may_reach_emitter(const Scene& sc, D3 o, D3 d, D3 T) {
    if (!(finite3(o) && finite3(d) && finite3(T))) return true;
    for (uint32_t k = 0; k < sc.n_emissive; ++k) {
        const EmitTri& e = sc.emissive[k];  // uniform index: scalar loads
        TriPre p;  // tri_pre() with its origin-independent part read from the table
        p.e0 = ld3(e.e0); p.e1 = ld3(e.e1); p.normal = ld3(e.normal);
        p.vo = ld3(e.v0) - o;
        p.num = dot(p.vo, p.normal);
        if (tri_may_hit(p, d)) return true;
    }
    return false;
}
__global__ void k_build_emitters(Scene sc, const uint32_t* leaf, uint32_t n, EmitTri* out) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const TriV tr = load_tri(sc, leaf[k]);
    const TriPre p = tri_pre(tr, mk(0., 0., 0.));
    EmitTri& e = out[k];
    e.v0[0] = tr.v0.x; e.v0[1] = tr.v0.y; e.v0[2] = tr.v0.z;
    e.e0[0] = p.e0.x; e.e0[1] = p.e0.y; e.e0[2] = p.e0.z;
    e.e1[0] = p.e1.x; e.e1[1] = p.e1.y; e.e1[2] = p.e1.z;
    e.normal[0] = p.normal.x; e.normal[1] = p.normal.y; e.normal[2] = p.normal.z;
}

// The shading of one path vertex (main.rs:292-315), shared by the shade kernel and the fused path
// kernel.
//   FIRST = bounce 0: queue index == slot, throughput == 1, and the camera ray is recomputed from
//           the pixel (same arithmetic as the generate stage) instead of being read back from HBM.
//   LAST  = bounce == max_bounces: only main.rs:299/307 remain (the reference still samples the
//           BRDF there, main.rs:310, but nothing consumes it), so the ray and hit distance are not
//           even loaded unless NEE needs them for the MIS weight.
//   MATS  = which material code the scene needs (Scene::mat_features): bit 0 textures, bit 1
//           Glossy / Mirror / Transparent.  0 ("plain") =
//           the scene has only untextured Lambertian materials: the Glossy and texture code is
//           compiled out, which is what keeps this kernel's register count (and occupancy) sane.
//   FUSED = o, d hold the ray (the caller traced it); otherwise they are loaded here when needed.
__device__ __forceinline__ void own_add(const Sample3& ls, uint32_t slot, D3 add, bool& lit);
struct ShadeOut {
    bool alive = false, shadow = false;  // the path continues / a shadow ray was made
    uint32_t slot = 0;
    D3 n_o, n_d, n_T;  // next ray and throughput
    double n_pdf = 0.;
    D3 s_d, s_c;  // shadow ray direction and the radiance it carries if unoccluded
    double s_tmax = 0.;
    RayDiffs n_diff;  // DIFFS: the next ray's differential rays
    bool lit = false;  // FIRST: the sample's radiance was written (Sample3)
};
// the part of a path record the shading needs besides the ray; the fused kernel loads it before
// it walks the tree, so that the loads' latency is hidden behind the traversal
struct PathRec {
    uint32_t slot;
    D3 T;
    double prev_pdf;  // NEE: pdf of the BRDF sample that spawned the ray (the MIS weight of a hit on a light)
};
template <bool FIRST, bool NEE>
__device__ __forceinline__ PathRec load_rec(const Queue& in, uint32_t i) {
    if (FIRST) return {i, mk(1., 1., 1.), 0.};
    return {in.slot[i], mk(in.T[0][i], in.T[1][i], in.T[2][i]), NEE ? in.prev_pdf[i] : 0.};
}
//   DIFFS = PT_FLAG_PROPAGATE_DIFFS (with USE_DIFFS, on a scene with textures): rays carry their
//           differential rays through the queue (Queue::diff) and every vertex spawns them as the
//           legacy reflect_diff / refract_diff do.
//   OWN   = the lane is the bounce-0 lane of its slot and has not stored the slot's bit of Sample3::lit yet
//           (r.lit holds it): the second vertex of the kernel that traces bounces 0 and 1 in one go (k_path F01).
template <bool FIRST, bool LAST, bool NEE, int MATS, bool FUSED, bool DIFFS = false, bool OWN = false>
__device__ __forceinline__ void shade_vertex(const Scene& sc, const Frame& fr, uint32_t bounce, const Queue& in,
                                             uint32_t i, PathRec rec, const SlotId& id /* slot_id(fr, rec.slot) */, uint32_t leaf,
                                             double t_hit, D3 o, D3 d, const Sample3& ls, ShadeOut& r) {
    constexpr bool TEX = (MATS & 1) != 0, SPEC = (MATS & 2) != 0;
    const uint32_t slot = rec.slot;
    r.slot = slot;
    const uint32_t s = id.s, tx = id.tx, gy_ = id.gy, pixel = id.pixel;
    uint64_t sample = fr.sample_base + s;
    constexpr bool kNeedRay = !LAST || NEE;
    if (!FUSED) {  // FUSED: the caller traced the ray in this kernel and still holds it
        if (FIRST) {
            PixelArgs a = pixel_args(fr.x0 + tx, gy_, fr.width, fr.height, fr.jitter[2 * s], fr.jitter[2 * s + 1], fr.aspect);
            o = ld3(sc.cam.eye);
            d = camera_dir(sc.cam, a.x, a.y);
        } else if (kNeedRay) {
            o = mk(in.o[0][i], in.o[1][i], in.o[2][i]);
            d = mk(in.d[0][i], in.d[1][i], in.d[2][i]);
        }
    }
    D3 T = rec.T;
    D3 add;  // what main.rs:299 / :307 adds to color_result
    if (leaf == PT_MISS) {
        add = T * ld3(sc.background);  // main.rs:299
    } else if (!kNeedRay) {
        const Material& m = sc.mats[hit_material(sc, leaf)];
        D3 le = (SPEC && m.kind != PT_MAT_LAMBERT) ? mk(0., 0., 0.) : ld3(m.emission);  // microfacet.rs:33-35; specular: none
        add = T * le;  // main.rs:307
    } else {
        const double t = t_hit;
        uint32_t mat;
        D3 position, normal;
        hit_geometry(sc, leaf, o, d, t, mat, position, normal);
        const Material& m = sc.mats[mat];
        D3 le = (SPEC && m.kind != PT_MAT_LAMBERT) ? mk(0., 0., 0.) : ld3(m.emission);  // microfacet.rs:33-35; specular: none
        add = T * le;  // main.rs:307
        const bool ref_lights = NEE && reference_lights(sc, fr.flags);  // lights.rs as written: no MIS
        if (NEE && !FIRST && !ref_lights && (le.x != 0. || le.y != 0. || le.z != 0.)) {
            // MIS weight of a BRDF-sampled hit on a light (balance heuristic)
            // triangles emit from both faces (main.rs:307 adds Le on any hit), spheres from outside
            // only (lights.rs:47-49)
            // (the light's own normal (v1-v0) x (v2-v0), normalised, is -(e0 x e1) normalised = `normal`
            // bit for bit: negation commutes exactly with the cross product and the normalisation)
            double cos_l = leaf >= PT_PRIM_BASE ? dot(normal, -d) : fabs(dot(normal, d));
            double p_l = cos_l <= 0. ? 0. : t * t / (sc.light_area * cos_l);
            double pp = rec.prev_pdf;
            double wgt = pp > 0. ? pp / (pp + p_l) : (pp < 0. ? 1. : 0.);  // pp < 0: specular predecessor
            add = T * le * wgt;
        }
        if (!LAST) {
            D3 diffuse = mk(0., 0., 0.);
            RayDiffs inc_diff;
            D3 dpdx = mk(0., 0., 0.), dpdy = dpdx, dndx = dpdx, dndy = dpdx;
            bool curved = false;
            if (DIFFS) {  // surface.rs:59-84 for the ray's own differentials, at every vertex
                if (FIRST) {
                    inc_diff = camera_diffs(sc, fr, o, fr.x0 + tx, gy_, s);
                } else {
                    const double* q = in.diff + i;
                    const size_t st = in.diff_stride;
                    inc_diff.rxo = mk(q[0], q[st], q[2 * st]);
                    inc_diff.ryo = mk(q[3 * st], q[4 * st], q[5 * st]);
                    inc_diff.rxd = mk(q[6 * st], q[7 * st], q[8 * st]);
                    inc_diff.ryd = mk(q[9 * st], q[10 * st], q[11 * st]);
                }
                // a plane's surface is Surface::new (geometry/mod.rs:45): no differentials; a sphere's has them
                // and the derivatives of its normal (mod.rs:136-145)
                const Prim* pr = leaf >= PT_PRIM_BASE ? sc.prims + (leaf - PT_PRIM_BASE) : nullptr;
                curved = pr && pr->kind == PT_PRIM_SPHERE;
                bool has_dp = (!pr || curved) && surface_dp(normal, position, inc_diff, dpdx, dpdy);
                if (curved) {
                    const SphereDerivs sd = sphere_derivs(position - ld3(pr->p), pr->radius);
                    double dudx = 0., dvdx = 0., dudy = 0., dvdy = 0.;
                    if (has_dp) uv_derivs(normal, sd.dpdu, sd.dpdv, dpdx, dpdy, dudx, dvdx, dudy, dvdy);
                    dndx = sd.dndu * dudx + sd.dndv * dvdx;
                    dndy = sd.dndu * dudy + sd.dndv * dvdy;
                }
                if (m.kind == PT_MAT_LAMBERT) {
                    if (m.tex < 0) diffuse = ld3(m.color);
                    else if (pr) diffuse = prim_textured_color(sc, m, *pr, position, normal, has_dp, dpdx, dpdy);
                    else diffuse = textured_color(sc, m, load_tri(sc, leaf), leaf, o, d, normal, has_dp, dpdx, dpdy);
                }
            } else if (!SPEC || m.kind == PT_MAT_LAMBERT) {
                diffuse = TEX ? diffuse_of(sc, m, leaf, o, d, position, normal, fr, FIRST, fr.x0 + tx, gy_, s) : ld3(m.color);
            }
            if (ref_lights) {
                // EstimateDirect (lights.rs:110-137) over AreaLight::sample (lights.rs:42-62) and
                // Sphere::sample_surface (lights.rs:70-78), at a Lambertian vertex (pt.h, PT_FLAG_REFERENCE_LIGHTS)
                if (!SPEC || m.kind == PT_MAT_LAMBERT) {
                    double r0, r1;
                    uniform2(fr.seed, pixel, sample, bounce, STREAM_LIGHT, r0, r1);
                    double z = 1. - 2. * r0;  // sample_sphere, lights.rs:7-15
                    double rr = sqrt(fmax(0., 1. - z * z));
                    double phi = 2. * PT_PI * r1;
                    double sp, cp;
                    sincos(phi, &sp, &cp);
                    D3 unit_sample = mk(rr * cp, rr * sp, z);
                    D3 light_pos = unit_sample * sc.area_light[3] + mk(sc.area_light[0], sc.area_light[1], sc.area_light[2]);
                    const double light_pdf = 0.25 * PT_FRAC_1_PI;  // lights.rs:76, whatever the radius
                    D3 diff = light_pos - position;
                    D3 light_direction = normalize(diff);
                    if (!(dot(unit_sample, -light_direction) < 0.)) {  // lights.rs:48-50
                        D3 irradiance = mk(sc.area_light[4], sc.area_light[5], sc.area_light[6]) / dot(diff, diff);
                        D3 f = diffuse * PT_FRAC_1_PI * fabs(dot(light_direction, normal));
                        D3 Ld = f * irradiance / light_pdf;  // lights.rs:132-135
                        r.s_c = T * Ld;
                        r.s_d = light_direction;
                        r.s_tmax = length(diff);  // lights.rs:52
                        r.n_o = position;
                        r.shadow = true;
                    }
                }
            } else if (NEE && sc.n_lights) {
                double r0, r1;
                uniform2(fr.seed, pixel, sample, bounce, STREAM_LIGHT, r0, r1);
                double target = r0 * sc.light_area;
                uint32_t lo_i = 0, hi_i = sc.n_lights - 1;  // first light with cdf > target, else last
                while (lo_i < hi_i) {
                    uint32_t mid = (lo_i + hi_i) >> 1;
                    if (sc.lights[mid].cdf <= target) lo_i = mid + 1; else hi_i = mid;
                }
                const Light& L = sc.lights[lo_i];
                double cdf_lo = lo_i ? sc.lights[lo_i - 1].cdf : 0.;
                double u1 = (target - cdf_lo) / (L.cdf - cdf_lo);
                if (!(u1 >= 0.)) u1 = 0.;
                if (u1 > 1.) u1 = 1.;
                D3 lp, sphere_n = mk(0., 0., 0.);
                if (L.sphere) {  // lights.rs:7-15 sample_sphere + :70-78 Sphere::sample_surface
                    double z = 1. - 2. * u1;
                    double rr = sqrt(fmax(0., 1. - z * z));
                    double phi = 2. * PT_PI * r1;
                    double sp, cp;
                    sincos(phi, &sp, &cp);
                    sphere_n = mk(rr * cp, rr * sp, z);
                    lp = sphere_n * L.v1[0] + ld3(L.v0);
                } else {
                    double su = sqrt(u1);
                    double b0 = 1. - su, b1 = r1 * su;
                    lp = ld3(L.v0) * b0 + ld3(L.v1) * b1 + ld3(L.v2) * (1. - b0 - b1);
                }
                D3 diff = lp - position;
                double dist = length(diff);
                D3 wl = diff / dist;
                D3 wo = -d;
                double p_b;
                if ((SPEC && is_glossy(m.kind))) {
                    double pg = glossy_pdf_dir(m.width, wo, wl, normal, m.kind == PT_MAT_GLOSSY_GGX);
                    p_b = pg > 0. ? pg : 0.;
                } else if (SPEC && m.kind != PT_MAT_LAMBERT) {
                    p_b = 0.;  // specular: a light sample has zero weight
                } else {
                    double c = dot(normal, wl);
                    p_b = c > 0. ? c * PT_FRAC_1_PI : 0.;
                }
                double cos_l = L.sphere ? dot(sphere_n, -wl) : fabs(dot(ld3(L.normal), wl));
                double p_l = cos_l <= 0. ? 0. : dist * dist / (sc.light_area * cos_l);
                if (p_l > 0. && p_b > 0. && dist > 0.) {
                    D3 f = (SPEC && is_glossy(m.kind)) ? glossy_f(m.width, ld3(m.color), wo, wl, normal, m.kind == PT_MAT_GLOSSY_GGX)
                                                   : diffuse * PT_FRAC_1_PI;
                    double cosn = fabs(dot(wl, normal));
                    double wgt = p_l / (p_l + p_b);
                    r.s_c = T * f * ld3(L.emission) * (cosn * wgt / p_l);
                    r.s_d = wl;
                    r.s_tmax = dist * (1. - 1e-9);
                    r.n_o = position;  // the shadow ray starts here whether or not the path goes on (roulette)
                    r.shadow = true;
                }
            }
            // Russian roulette of the legacy integrator (main.rs:85-93,110): trace_depth = bounce + 1
            double luminance = 1.;
            bool absorbed = false;
            if ((fr.flags & PT_FLAG_RUSSIAN_ROULETTE) && bounce + 1 >= 5 && (!SPEC || m.kind == PT_MAT_LAMBERT)) {
                luminance = (diffuse.x + diffuse.y + diffuse.z) / 3.;
                double q0, q1;
                uniform2(fr.seed, pixel, sample, bounce, STREAM_ROULETTE, q0, q1);
                absorbed = luminance < q0;
            }
            if (!absorbed) {
                double xi1, xi2;
                uniform2(fr.seed, pixel, sample, bounce, STREAM_BRDF, xi1, xi2);
                int lobe = 0;
                if (SPEC && (m.kind == PT_MAT_MIRROR || m.kind == PT_MAT_TRANSPARENT)) {
                    r.n_T = T;  // specular: radiance passes unchanged
                    r.n_d = specular_sample(m, d, normal, xi1, lobe);
                    r.n_pdf = -1.;
                } else if (SPEC && is_glossy(m.kind)) {
                    Brdf bs = glossy_sample(m.width, ld3(m.color), d, normal, xi1, xi2, m.kind == PT_MAT_GLOSSY_GGX);
                    r.n_T = T * (bs.f * fabs(dot(bs.wi, normal)) / bs.pdf);  // main.rs:311
                    r.n_d = bs.wi;
                    r.n_pdf = bs.pdf > 0. ? bs.pdf : 0.;
                } else {
                    // Lambertian (material.rs:37-58): f = rho / pi, pdf = (n . wi) / pi, so main.rs:311's
                    // f |wi . n| / pdf is rho; the closed form is used (within 4 ulp of the quotient; they part
                    // only where xi1 < 2^-52 puts wi in the tangent plane and the quotient is -rho or NaN)
                    const TriBasis* basis = (sc.tri_basis && leaf < PT_PRIM_BASE) ? sc.tri_basis + leaf : nullptr;
                    r.n_d = lambertian_dir(normal, basis, xi1, xi2);
                    r.n_T = T * diffuse;
                    if (NEE) {  // the pdf feeds only the MIS weight of the next vertex
                        const double pdf = dot(normal, r.n_d) * PT_FRAC_1_PI;
                        r.n_pdf = pdf > 0. ? pdf : 0.;
                    }
                }
                if (luminance != 1.) r.n_T = r.n_T / luminance;
                r.n_o = position;  // main.rs:314
                if (DIFFS)
                    r.n_diff = spawn_diffs(m, lobe, position, normal, dpdx, dpdy, d, inc_diff, r.n_d, fr.diff_size,
                                           (fr.flags & PT_FLAG_ROUGHDIFF) != 0, curved, dndx, dndy);
                r.alive = true;
                if (sc.cull_last && bounce + 1 == fr.max_bounces) r.alive = may_reach_emitter(sc, r.n_o, r.n_d, r.n_T);
            }
        }
    }
    // color_result += add.  Adding +-0 never changes the sum (it starts at +0), NaN != 0 is true.
    if (add.x != 0. || add.y != 0. || add.z != 0.) {
        if (FIRST) {  // slot == lane index: the caller stores the wave's word of Sample3::lit from r.lit
            double* c = ls.c + (size_t)slot * 3;
            c[0] = 0. + add.x; c[1] = 0. + add.y; c[2] = 0. + add.z;
            r.lit = true;
        } else if (OWN) {  // radiance_add() with the slot's bit still in the register
            own_add(ls, slot, add, r.lit);
        } else {
            radiance_add(ls, slot, add);
        }
    }
}
// color_result += add for the lane that owns the slot's bit (bounce 0, before store_lit): the arithmetic of
// radiance_add (a first value is 0 + add, a later one is added), the bit kept in `lit`
__device__ __forceinline__ void own_add(const Sample3& ls, uint32_t slot, D3 add, bool& lit) {
    double* c = ls.c + (size_t)slot * 3;
    if (lit) { c[0] += add.x; c[1] += add.y; c[2] += add.z; }
    else { c[0] = 0. + add.x; c[1] = 0. + add.y; c[2] = 0. + add.z; lit = true; }
}
// bounce 0: the wave's 64 slots are one word of Sample3::lit (i0 = the wave's first slot, a multiple
// of 64); called by all lanes of the wave, converged
__device__ __forceinline__ void store_lit(const Sample3& ls, uint32_t i0, uint32_t end, bool lit) {
    unsigned long long m = __ballot(lit);
    if ((threadIdx.x & 63u) == 0 && i0 < end) ls.lit[i0 >> 6] = m;
}

template <bool NEE, bool DIFFS = false>
__device__ __forceinline__ void push_path(const Queue& out, uint32_t pos, const ShadeOut& r) {
    if (DIFFS) {
        double* q = out.diff + pos;
        const size_t st = out.diff_stride;
        const D3 v[4] = {r.n_diff.rxo, r.n_diff.ryo, r.n_diff.rxd, r.n_diff.ryd};
#pragma unroll
        for (int k = 0; k < 4; ++k) { q[3 * k * st] = v[k].x; q[(3 * k + 1) * st] = v[k].y; q[(3 * k + 2) * st] = v[k].z; }
    }
    {   // the nine arrays through one base pointer (Queue::stride9: every queue a kernel appends to is a carved one)
        double* b = out.o[0] + pos;
        const size_t st = out.stride9;
        b[0] = r.n_o.x; b[st] = r.n_o.y; b[2 * st] = r.n_o.z;
        b[3 * st] = r.n_d.x; b[4 * st] = r.n_d.y; b[5 * st] = r.n_d.z;
        b[6 * st] = r.n_T.x; b[7 * st] = r.n_T.y; b[8 * st] = r.n_T.z;
    }
    out.slot[pos] = r.slot;
    if (NEE) out.prev_pdf[pos] = r.n_pdf;
}
__device__ __forceinline__ void push_shadow(const ShadowQueue& sq, uint32_t spos, const ShadeOut& r) {
    sq.o[0][spos] = r.n_o.x; sq.o[1][spos] = r.n_o.y; sq.o[2][spos] = r.n_o.z;
    sq.d[0][spos] = r.s_d.x; sq.d[1][spos] = r.s_d.y; sq.d[2][spos] = r.s_d.z;
    sq.tmax[spos] = r.s_tmax;
    sq.c[0][spos] = r.s_c.x; sq.c[1][spos] = r.s_c.y; sq.c[2][spos] = r.s_c.z;
    sq.slot[spos] = r.slot;
}

// shade: one lane per live path.
template <bool FIRST, bool LAST, bool NEE, int MATS, bool DIFFS = false>
__global__ __launch_bounds__(kBlock) void k_shade(Scene sc, Frame fr, uint32_t bounce, Queue in, Hits hits,
                                                  Live live, Queue out, Push out_push, ShadowQueue sq, Push sq_push,
                                                  Sample3 ls) {
    __shared__ uint32_t lds_a[kBlock / 64 + 1];
    __shared__ uint32_t lds_b[kBlock / 64 + 1];
    if (LAST && !FIRST && live.sparse) {  // few live entries: this small grid enumerates the live blocks
        for (uint32_t blk = blockIdx.x;; blk += gridDim.x) {
            uint32_t begin, end;
            if (!sparse_block<kBlock>(live, blk, begin, end)) return;
            const uint32_t i = begin + threadIdx.x;
            if (i >= end) continue;
            ShadeOut r;
            uint32_t leaf = hits.tri[i];
            double t = (leaf != PT_MISS && NEE) ? hits.t[i] : 0.;
            const PathRec rec = load_rec<FIRST, NEE>(in, i);
            shade_vertex<FIRST, LAST, NEE, MATS, false, DIFFS>(sc, fr, bounce, in, i, rec, slot_id(fr, rec.slot), leaf, t,
                                                                mk(0., 0., 0.), mk(0., 0., 0.), ls, r);
        }
    }
    if (blockIdx.x * kBlock >= live.n) return;
    uint32_t n = live_end(live, blockIdx.x * kBlock);
    if (blockIdx.x * kBlock >= n) return;  // whole block idle
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t region = blockIdx.x % kSegs;  // the segment this workgroup appends to
    ShadeOut r;
    r.n_o = r.n_d = r.n_T = r.s_d = r.s_c = mk(0., 0., 0.);
    if (i < n) {
        uint32_t leaf = hits.tri[i];
        // the hit distance is only needed (and only loaded) where the vertex is shaded in full
        double t = (leaf != PT_MISS && (!LAST || NEE)) ? hits.t[i] : 0.;
        const PathRec rec = load_rec<FIRST, NEE>(in, i);
        shade_vertex<FIRST, LAST, NEE, MATS, false, DIFFS>(sc, fr, bounce, in, i, rec, slot_id(fr, rec.slot), leaf, t,
                                                            mk(0., 0., 0.), mk(0., 0., 0.), ls, r);
    }
    if (FIRST) store_lit(ls, i - (threadIdx.x & 63u), n, r.lit);
    if (LAST) return;  // nothing continues
    uint32_t pos = region * out_push.seg_size + compact_slot(r.alive, out_push.counts + region * kCtrStride, lds_a);
    if (r.alive) push_path<NEE, DIFFS>(out, pos, r);
    if (NEE) {
        uint32_t spos = region * sq_push.seg_size + compact_slot(r.shadow, sq_push.counts + region * kCtrStride, lds_b);
        if (r.shadow) push_shadow(sq, spos, r);
    }
}

// path: extend and shade of one bounce in one kernel, one ray per lane (the while-while traversal).
// The ray stays in registers between the two stages and the hit record never exists in memory: per
// secondary ray 76 B are read and at most 76 B written, against 48 + 12 (extend) + 12 + 76 (shade)
// + 76 for the two-kernel form, and the FP64-bound traversal of one wave overlaps the HBM-bound
// shading of its neighbours on the same SIMD.  Survivors are appended with one atomic per
// workgroup to segment blockIdx % kSegs.
// Workgroup size of the fused kernel: 128 threads for shallow trees (measured on CornellBox 1080p:
// 64 -> 9.38, 128 -> 11.09, 256 -> 10.93, 512 -> 9.90 Gsamples/s: one-wave workgroups are bound by
// workgroup dispatch when every wave is short), one wave for deep trees (waves differ 10x in length;
// LDS and registers are released per wave).
#ifndef PT_PATH_BLOCK_LDS
#define PT_PATH_BLOCK_LDS 256  // (512: -2 % on CornellBox, -12 % on glossy_planes; 1024: -32 %)
#endif
#ifndef PT_PATH_BLOCK_DEEP
#define PT_PATH_BLOCK_DEEP 64
#endif
// (with the scene in LDS - LDS below - 256 threads: 64 / 128 / 256 / 512 / 1024 -> 19.5 / 24.0 / 25.5 /
// 24.0 / 17.4 Gsamples/s on CornellBox; the copy is shared by twice the rays)
#ifdef PT_PATH_PROFILE  // diagnostic build (scripts/diag/path_profile.py): the path kernels stop after section g_path_cut
__device__ int g_path_cut;  // 0 = run everything (clock stamps proved useless: the compiler moves the work across them)
#define PT_CUT(k) (path_cut == (k))
#else
#define PT_CUT(k) false
#endif
template <int DEPTH, bool LDS = false> struct PathBlock {
    static constexpr int value = DEPTH <= 8 ? (LDS ? PT_PATH_BLOCK_LDS : 128) : PT_PATH_BLOCK_DEEP;
};
// LDS = the whole scene lives in LDS for the workgroup's lifetime (Scene::lds_scene: at most
// kLdsScene nodes, triangles and kLdsMats materials - every scene of the reference): nodes, triangles,
// shading records, materials and, for the primary bounce, the eye-relative tables are copied in by
// the first live block (~6 KB, 48 B per thread) and every dependent fetch of the traversal becomes a
// ds_read instead of a trip through the vector L1.
constexpr int kMatWords = (int)(sizeof(Material) / sizeof(double));
// F01 = bounces 0 and 1 in one go (FIRST && !LAST && LDS, max_bounces >= 2): a lane whose path survives its first
// vertex traces and shades the second one at once, in registers - no 76-byte record written and read back, no
// compaction, no second copy of the scene into LDS, one launch and one tail instead of two.  Lanes whose path
// ended at the first vertex idle meanwhile, so this is for scenes whose primary rays nearly all hit
// (Scene::fuse01: CornellBox 97 %).  Same arithmetic per path, additions in the same order: same bits.
// `mid_push` counts the paths that go on to the second vertex (the rays of bounce 1, for the statistics).
template <int DEPTH, bool FIRST, bool LAST, bool NEE, int MATS, bool LDS = false, bool F01 = false>
#ifndef PT_TEX_WAVES
#define PT_TEX_WAVES 5
#endif
// Waves per SIMD asked of the compiler: the texture code (MATS bit 0) lifts the kernels to 106-109 VGPRs, one
// step above the 96 that let a fifth wave in; bounded to 5 they spill 36-92 B per lane and still gain (textured
// Test.obj 4K 22.5 -> 24.4 Gsamples/s, with NEE 12.8 -> 13.6).  The untextured kernels fit already; bounding
// them too measured -0.4 ... -4 % on CornellBox and 0 on glossy_planes.
// The deep-tree kernels sit at 95-97 VGPRs without a bound: asked for five waves they stay at 96 (the NEE primary
// kernel took 97 after a change to the node step and lost its fifth wave: +6 % time).
__global__ __launch_bounds__((PathBlock<DEPTH, LDS>::value), (((MATS & 1) || DEPTH > 8) ? PT_TEX_WAVES : 1)) void k_path(Scene gsc, Frame fr, uint32_t bounce, Queue in, Live live, Queue out,
                                                 Push out_push, ShadowQueue sq, Push sq_push, Sample3 ls,
                                                 Counters* ctr, Push mid_push) {
    static_assert(!F01 || (FIRST && !LAST && LDS), "the two-bounce form is the primary launch of a scene in LDS");
    constexpr int kPathBlock = PathBlock<DEPTH, LDS>::value;
    __shared__ int stack[DEPTH * kPathBlock];
    // The deep-tree path kernel runs one 64-ray workgroup per wave, so every wave pays the copy of the tree's top for
    // the one walk it does: the root and its four children (kPathTopNodes = 5) instead of the extend kernel's three
    // levels (kTopNodes = 21, copied once per persistent wave): primary kernel -8 % (0 / 1 / 5 / 9 / 21 nodes:
    // 1418 / 1391 / 1358 / 1444 / 1477 ms per 128 launches), heightfield +2.7 %.
    constexpr int kPathTop = PT_PATH_TOP < kTopNodes ? PT_PATH_TOP : kTopNodes;
    __shared__ float4 s_top[(DEPTH > 8 && kPathTop > 0) ? kPathTop * 8 : 1];
    TopNodes top{nullptr, 0};
    if (DEPTH > 8 && kPathTop > 0) top = load_top_nodes<kPathBlock, kPathTop>(gsc, s_top);
    __shared__ uint32_t lds_a[kPathBlock / 64 + 1];
    __shared__ uint32_t lds_b[kPathBlock / 64 + 1];
    // the scene in LDS: one array, Scene::lds_image's layout (pt_device.h kImg...); a kernel allocates the prefix it needs
    constexpr bool kEdges = LDS && (!FIRST || NEE || F01);  // some ray of this kernel is traced by traverse()
    constexpr bool kVerts = LDS && (MATS & 1);              // a textured hit reads its vertices again
    constexpr int kImgUsed = !LDS ? 1 : (kVerts ? kImgEnd : (kEdges ? kImgTris : kImgNodes));
    __shared__ float4 s_img[kImgUsed];
    static_assert(sizeof(TriShade) == 32 && sizeof(TriBasis) == 48 && sizeof(EyeNode) == 64 && sizeof(EyeTri) == 112 && sizeof(Material) % 16 == 0, "LDS copies");
    Scene sc = gsc;
    if (LDS) {  // pointers first (compile-time: the address space is then known at every use), data below
        sc.nodes = s_img + kImgNodes;      // (plane-major: node_step_lds)
        sc.tris = s_img + (kVerts ? kImgTris : 0);
        sc.tri_shade = reinterpret_cast<const TriShade*>(s_img + kImgShade);
        if (!LAST) sc.tri_basis = reinterpret_cast<const TriBasis*>(s_img + kImgBasis);  // (an LDS scene always has the table)
        sc.mats = reinterpret_cast<const Material*>(s_img + kImgMats);
        if (FIRST) {
            sc.eye_nodes = reinterpret_cast<const EyeNode*>(s_img + kImgEyeNodes);
            sc.eye_tris = reinterpret_cast<const EyeTri*>(s_img + kImgEyeTris);
        }
        if (kEdges) sc.tri_edges = reinterpret_cast<const EyeTri*>(s_img + kImgEdges);
    }
    bool scene_copied = false;
    if (ctr && blockIdx.x == 0 && threadIdx.x == 0)
        atomicAdd(&ctr->rays, FIRST ? (unsigned long long)fr.n_on_screen * fr.n_samples : live_total(live));
    ShadeOut r;
    r.n_o = r.n_d = r.n_T = r.s_d = r.s_c = mk(0., 0., 0.);
    uint32_t n_mid = 0, n_shadow = 0;  // F01: this lane's path reached its second vertex; light samples it took
    const bool sparse = LAST && !FIRST && live.counts && live.sparse;
#ifdef PT_PATH_PROFILE
    const int path_cut = g_path_cut;
#endif
    for (uint32_t blk = blockIdx.x;; blk += gridDim.x) {  // one pass, except over a sparse queue
        uint32_t begin, end;
        if (sparse) {  // few live entries: this small grid enumerates the live blocks
            if (!sparse_block<kPathBlock>(live, blk, begin, end)) return;
        } else if (live.counts) {
            begin = blk * kPathBlock;
            if (begin >= live.n) return;
            end = min(begin + kPathBlock, live_end(live, begin));
        } else {
            if (blk >= trav_blocks(live.n, kPathBlock)) return;
            begin = xcd_block(blk, kChunkRays / kPathBlock) * kPathBlock;
            end = min(begin + kPathBlock, live.n);
        }
        if (begin >= end) return;
        if (LDS && !scene_copied) {  // block-uniform: every thread of a live block gets here
            // the prefix of the image this kernel reads; one that traces no primary rays skips their tables
            constexpr int kSkipFrom = FIRST ? kImgUsed : kImgEyeNodes, kSkipTo = FIRST ? kImgUsed : kImgNodes;
            for (int k = threadIdx.x; k < kSkipFrom; k += kPathBlock) s_img[k] = gsc.lds_image[k];
            for (int k = kSkipTo + threadIdx.x; k < kImgUsed; k += kPathBlock) s_img[k] = gsc.lds_image[k];
            __syncthreads();
            scene_copied = true;
        }
        if (PT_CUT(5)) return;  // (diagnostic build: the cost of getting here)
        const uint32_t i = begin + threadIdx.x;
        if (i < end) {
            D3 o = mk(0., 0., 0.), d = o;
            bool on_screen = true;  // FIRST: the pixel lies inside the scene's screen bounds (Frame::cx0..)
            SlotId id{0, 0, 0, 0};  // FIRST: of slot i, here; otherwise of the record's slot, once that is loaded
            if (FIRST) {
                id = slot_id(fr, i);
                const uint32_t gx = fr.x0 + id.tx, gy = id.gy, s = id.s;
                on_screen = gx >= fr.cx0 && gx < fr.cx1 && gy >= fr.cy0 && gy < fr.cy1;
                if (on_screen) {
                    PixelArgs a = pixel_args(fr, gx, gy, fr.jitter[2 * s], fr.jitter[2 * s + 1]);
                    o = ld3(sc.cam.eye);
                    d = camera_dir(sc.cam, a.x, a.y);
                }
            } else {
                o = mk(in.o[0][i], in.o[1][i], in.o[2][i]);
                d = mk(in.d[0][i], in.d[1][i], in.d[2][i]);
            }
            const PathRec rec = load_rec<FIRST, NEE>(in, i);  // in flight while the tree is walked
            if (PT_CUT(6)) { if (d.x == 0.5 && d.y == 0.25) r.lit = true; on_screen = false; }  // (diagnostic build: the ray exists)
            double t = 1.7976931348623157e308;  // ray.rs:22 f64::MAX; t_min = 1e-5 (ray.rs:21)
            uint32_t nn = 0, nt = 0;
            uint32_t leaf = PT_MISS;
            // Deep trees: a wave whose 64 slots are ONE pixel in 64 frames (Frame::grp_log2 == 6, inside the whole 8x8
            // blocks of the tile) walks the tree as a packet.  Wave-uniform: `begin` is the wave's first slot, groups
            // of frames and the blocks' share of them are multiples of 64 slots.  (Measured: 8 pixels x 8 frames per
            // wave is 38 % slower as a packet than lane by lane, an 8x8 block in one frame 2.9x.)
            bool packet = false;
            if (DEPTH > 8 && FIRST && kPathBlock == 64 && sc.packet_primary && fr.grp_log2 >= sc.packet_primary) {
                const uint32_t lg = fr.grp_log2, frame = fdiv(begin, fr.div_pixels);
                packet = begin - ((frame >> lg) << lg) * fr.n_tile_pixels < (fr.n8 << lg);
            }
            const RayF prf = make_rayf(o, d);
            packet = packet && packet_signs_agree(prf, on_screen);
            if (packet)
                leaf = traverse_packet4<false>(sc, o, d, prf, 1e-5, t, on_screen, stack, top, nn, nt);
            else if (on_screen)
                leaf = (FIRST && (LDS || sc.eye_nodes)) ? traverse_eye<DEPTH, kPathBlock, LDS>(sc, d, 1e-5, t, stack + threadIdx.x)
                                               : traverse<DEPTH, false, false, kPathBlock, LDS>(sc, o, d, 1e-5, t, stack + threadIdx.x, nn, nt, top);
            if (PT_CUT(1) || PT_CUT(6)) { if (leaf == 0x7ffffff0u && t == 0.5) r.lit = true; leaf = PT_MISS; on_screen = false; }
            if (!FIRST) id = slot_id(fr, rec.slot);
            if (!PT_CUT(1) && !PT_CUT(6))
            shade_vertex<FIRST, LAST, NEE, MATS, true>(sc, fr, bounce, in, i, rec, id, leaf, t, o, d, ls, r);
            if (PT_CUT(2)) r.alive = false;
            // the light sample's visibility, in place: the shadow ray never leaves the registers
            // (84 B written + read per shadow ray and one launch per bounce otherwise).  Same order
            // of additions as the shadow kernel's: the vertex's emission first, then its light sample.
            auto trace_shadow = [&](bool own) {
                double st_max = r.s_tmax;
                uint32_t nn2 = 0, nt2 = 0;
                if (traverse<DEPTH, true, false, kPathBlock, LDS>(sc, r.n_o, r.s_d, 1e-5, st_max, stack + threadIdx.x, nn2, nt2, top) == PT_MISS) {
                    if (own) own_add(ls, r.slot, r.s_c, r.lit);
                    else radiance_add(ls, r.slot, r.s_c);
                }
            };
            if (NEE && !LAST && sc.fuse_shadow && r.shadow) trace_shadow(FIRST);
            if (F01) {
                n_shadow += r.shadow;
                if (r.alive) {  // the second vertex of the path, at once
                    const D3 o2 = r.n_o, d2 = r.n_d;
                    const PathRec rec2{r.slot, r.n_T, r.n_pdf};
                    r.alive = false; r.shadow = false;
                    n_mid = 1;
                    double t2 = 1.7976931348623157e308;
                    const uint32_t leaf2 = traverse<DEPTH, false, false, kPathBlock, LDS>(sc, o2, d2, 1e-5, t2, stack + threadIdx.x, nn, nt, top);
                    if (PT_CUT(3)) { if (leaf2 == 0x7ffffff0u && t2 == 0.5) r.lit = true; }
                    else
                    shade_vertex<false, false, NEE, MATS, true, false, true>(sc, fr, bounce + 1, in, i, rec2, id, leaf2, t2, o2, d2, ls, r);
                    if (PT_CUT(4)) r.alive = false;
                    if (NEE && r.shadow) trace_shadow(true);
                    n_shadow += r.shadow;
                }
            }
        }
        if (FIRST) store_lit(ls, i - (threadIdx.x & 63u), end, r.lit);
        if (!sparse) break;
    }
    if (LAST) return;  // nothing continues
    const uint32_t region = blockIdx.x % kSegs;
    uint32_t pos = region * out_push.seg_size + (kPathBlock == 64 ? compact_wave(r.alive, out_push.counts + region * kCtrStride)
                                                                   : compact_slot<kPathBlock>(r.alive, out_push.counts + region * kCtrStride, lds_a));
    if (r.alive) push_path<NEE>(out, pos, r);
    if (F01) {
        // statistics only: the rays of bounce 1 and the light samples of both vertices, one atomic per wave each
        // (the counters of the queues this launch does not fill)
        const uint32_t mid = (uint32_t)__popcll(__ballot(n_mid != 0));
        uint32_t sh = n_shadow;
#pragma unroll
        for (int d = 32; d; d >>= 1) sh += __shfl_down(sh, d);
        if ((threadIdx.x & 63u) == 0) {
            if (mid) atomicAdd(mid_push.counts + region * kCtrStride, mid);
            if (NEE && sh) atomicAdd(sq_push.counts + region * kCtrStride, sh);
        }
        return;
    }
    if (NEE) {
        // (fuse_shadow: only the count is kept, for the statistics)
        uint32_t spos = region * sq_push.seg_size + (kPathBlock == 64 ? compact_wave(r.shadow, sq_push.counts + region * kCtrStride)
                                                                      : compact_slot<kPathBlock>(r.shadow, sq_push.counts + region * kCtrStride, lds_b));
        if (r.shadow && !sc.fuse_shadow) push_shadow(sq, spos, r);
    }
}

// accumulate: main.rs:434-436, frames of the batch added in index order (deterministic f64 sum)
__global__ __launch_bounds__(kBlock) void k_accumulate(Frame fr, Sample3 ls, const unsigned long long* __restrict__ lit1,
                                                       uint32_t split, double* __restrict__ accum, Counters* ctr) {
    // one thread per pixel, in slot order: a wave's 64 bits of a frame are then one or two words of
    // Sample3::lit and its radiance reads are contiguous
    const uint32_t slot_in_frame = blockIdx.x * kBlock + threadIdx.x;
    if (slot_in_frame >= fr.n_tile_pixels) return;
    uint32_t n_lit = 0;
    uint32_t ty, tx;
    slot_pixel(fr, slot_in_frame, tx, ty);
    size_t pix = (size_t)frame_row(fr, ty) * fr.width + fr.x0 + tx;
    double r = accum[pix * 3], g = accum[pix * 3 + 1], b = accum[pix * 3 + 2];
    bool drop = (fr.flags & PT_FLAG_DROP_NONFINITE) != 0;
    // frames in index order; the bitmap words of eight frames are fetched together (independent loads),
    // the rare radiance values behind them as they come
    constexpr uint32_t kGroup = 8;
    for (uint32_t s0 = 0; s0 < fr.n_samples; s0 += kGroup) {
        unsigned long long word[kGroup];
        size_t slot_of[kGroup];
#pragma unroll
        for (uint32_t j = 0; j < kGroup; ++j) {
            const uint32_t s = min(s0 + j, fr.n_samples - 1u);
            // slot within its lane (each lane orders its own frames: Frame::grp_log2)
            const size_t kl = s < split ? sp_slot(fr, fr.grp_log2, s, slot_in_frame) : sp_slot(fr, fr.grp1_log2, s - split, slot_in_frame);
            slot_of[j] = s < split ? kl : (size_t)split * fr.n_tile_pixels + kl;  // ... and within the batch's radiance array
            word[j] = (s < split ? ls.lit : lit1)[kl >> 6] >> (kl & 63u);
        }
#pragma unroll
        for (uint32_t j = 0; j < kGroup; ++j) {
            const uint32_t s = s0 + j;
            if (s >= fr.n_samples) break;
            double cr = 0., cg = 0., cb = 0.;  // a sample without a value is +0 (Sample3)
            if (word[j] & 1ull) {
                const double* c = ls.c + slot_of[j] * 3;
                cr = c[0]; cg = c[1]; cb = c[2];
                ++n_lit;
            }
            if (drop && !(isfinite(cr) && isfinite(cg) && isfinite(cb))) continue;
            r += cr; g += cg; b += cb;
        }
    }
    accum[pix * 3] = r; accum[pix * 3 + 1] = g; accum[pix * 3 + 2] = b;
    if (ctr && n_lit) atomicAdd(&ctr->lit, (unsigned long long)n_lit);
}

// main.rs:402-405 + 469-481 with samples_per_pixel = 1 (main.rs:386): one jitter pair per frame,
// shared by all pixels; step = 1, x drawn before y, scaled by the pixel size (not twice it).
// Frame::pix_ss: pixel_args' screen-space coordinate of every column and row, by its own expressions
__global__ void k_pixel_table(uint32_t width, uint32_t height, double* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < width) {
        const double pixel_ndc_x = ((double)i + 0.5) / (double)width;
        out[i] = 2. * pixel_ndc_x - 1.;
    } else if (i < width + height) {
        const double pixel_ndc_y = ((double)(i - width) + 0.5) / (double)height;
        out[i] = 1. - 2. * pixel_ndc_y;
    }
}
__global__ void k_jitter(uint64_t seed, uint64_t sample_begin, uint32_t n, double psx, double psy, double* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r0, r1;
    uniform2(seed, 0u, sample_begin + i, 0u, STREAM_JITTER, r0, r1);
    double step = 1. / 1.;
    double jx = step * (r0 + 0.) - 1. / 2.;
    double jy = step * (r1 + 0.) - 1. / 2.;
    out[2 * i] = jx * psx;
    out[2 * i + 1] = jy * psy;
}


}  // namespace ptd
