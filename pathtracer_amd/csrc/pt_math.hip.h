// pt_math.hip.h — f64 vector helpers (vec3.rs order of operations), Philox4x32-10, pixel -> camera ray (main.rs:412-424, scene.rs:37-57)
// Part of the single translation unit pt_kernels.hip (device code is inlined across these files);
// included from there, in order: pt_math.hip.h, pt_traverse.hip.h, pt_shade.hip.h.
#pragma once
#include "pt_device.h"

#include "../../include/pt.h"

namespace ptd {

// ------------------------------------------------------------------------------------------------
// f64 vector helpers — operation order follows algebra/vec3.rs
struct D3 {
    double x, y, z;
};
__device__ __forceinline__ D3 mk(double x, double y, double z) { return D3{x, y, z}; }
__device__ __forceinline__ D3 ld3(const double* p) { return D3{p[0], p[1], p[2]}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 operator-(D3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator*(D3 a, D3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ D3 operator*(D3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ D3 operator*(double s, D3 a) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ D3 operator/(D3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ D3 cross(D3 a, D3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ double length(D3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ D3 normalize(D3 a) { return a / length(a); }  // vec3.rs:110-112: division, not rsqrt
__device__ __forceinline__ double comp(D3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
// vec3.rs:138-140 (normalises)
__device__ __forceinline__ D3 reflect(D3 self, D3 other) { return normalize(self - 2. * dot(self, other) * other); }
// vec3.rs:154-166: rotate_to(self, n) = t1 self.x + t2 self.y + n self.z, (t1, t2) the Frisvad / Duff basis of n
__device__ __forceinline__ void rotate_basis(D3 n, D3& t1, D3& t2) {
    double sign = (n.z != n.z) ? n.z : copysign(1.0, n.z);  // f64::signum
    double a = -1. / (1. + fabs(n.z));
    double b = n.x * n.y * a;
    t1 = mk(1. + n.x * n.x * a, b, -sign * n.x);
    t2 = mk(sign * b, sign * (1. + n.y * n.y * a), -n.y);
}
__device__ __forceinline__ D3 rotate_with(D3 self, D3 t1, D3 t2, D3 n) { return t1 * self.x + t2 * self.y + n * self.z; }
__device__ __forceinline__ D3 rotate_to(D3 self, D3 n) {
    D3 t1, t2;
    rotate_basis(n, t1, t2);
    return rotate_with(self, t1, t2, n);
}

#define PT_PI 3.14159265358979323846264338327950288
#define PT_FRAC_1_PI 0.318309886183790671537767526745028724

// ------------------------------------------------------------------------------------------------
// Philox4x32-10, counter = (pixel, sample_lo, sample_hi, bounce | stream << 24), key = seed.
enum { STREAM_BRDF = 0, STREAM_LIGHT = 1, STREAM_JITTER = 2, STREAM_ROULETTE = 3 };
__host__ __device__ __forceinline__ void uniform2(uint64_t seed, uint32_t pixel, uint64_t sample, uint32_t bounce,
                                                  uint32_t stream, double& u0, double& u1) {
    uint32_t c0 = pixel, c1 = (uint32_t)sample, c2 = (uint32_t)(sample >> 32), c3 = bounce | (stream << 24);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;  // v_mul_hi_u32 + v_mul_lo_u32
        uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
        uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    uint64_t a = ((uint64_t)c1 << 32) | c0;
    uint64_t b = ((uint64_t)c3 << 32) | c2;
    u0 = (double)(a >> 11) * (1.0 / 9007199254740992.0);
    u1 = (double)(b >> 11) * (1.0 / 9007199254740992.0);
}

// ------------------------------------------------------------------------------------------------
// main.rs:412-424 pixel -> arguments of render(); scene.rs:37-57 Camera::ray
struct PixelArgs {
    double x, y, ddx, ddy;
};
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv f) { return (uint32_t)(((unsigned long long)n * f.mul) >> f.shift); }
// tile row -> image row (Frame: interleaved row partition; the identity + y0 for a plain tile)
__device__ __forceinline__ uint32_t frame_row(const Frame& fr, uint32_t ty) {
    return fr.y0 + fr.row_off + ty + fdiv(ty, fr.div_group) * fr.row_skip;
}
// slot within a frame -> pixel of the tile (Frame: 8x8 blocks, right strip, bottom rows)
__device__ __forceinline__ void slot_pixel(const Frame& fr, uint32_t p, uint32_t& tx, uint32_t& ty) {
    if (p < fr.n8) {
        uint32_t b = p >> 6, w = p & 63u;
        uint32_t by = fdiv(b, fr.div_bw), bx = b - by * fr.bw;
        tx = bx * 8 + (w & 7u);
        ty = by * 8 + (w >> 3);
        return;
    }
    uint32_t q = p - fr.n8;
    const uint32_t right = fr.rw * fr.h8;
    if (q < right) {
        ty = fdiv(q, fr.div_rw);
        tx = fr.w8 + (q - ty * fr.rw);
    } else {
        q -= right;
        uint32_t yy = fdiv(q, fr.div_tile_w);
        tx = q - yy * fr.tile_w;
        ty = fr.h8 + yy;
    }
}
// Slot of a lane <-> (frame of the lane s, slot in the frame p): Frame::grp_log2 = lg, G = 1 << lg <= 64.
// Inside a group of G frames wave (block b, part j) holds, for each of the G frames, the 64 / G pixels
// [j * 64 / G, (j + 1) * 64 / G) of the 8x8 block b in its row-major order: lane = f * (64 / G) + q.
__device__ __forceinline__ void slot_sp(const Frame& fr, uint32_t lg, uint32_t slot, uint32_t& s, uint32_t& p) {
    const uint32_t frame = fdiv(slot, fr.div_pixels);
    if (lg == 0) { s = frame; p = slot - frame * fr.n_tile_pixels; return; }
    const uint32_t g = frame >> lg;                              // floor(floor(slot / pixels) / G)
    const uint32_t r = slot - (g << lg) * fr.n_tile_pixels;      // within the group of G frames
    const uint32_t n8g = fr.n8 << lg;
    if (r < n8g) {
        const uint32_t l = r & 63u, wj = r >> 6;                 // lane of the wave; wave = block * G + j
        const uint32_t b = wj >> lg, j = wj & ((1u << lg) - 1u);
        const uint32_t f = l >> (6u - lg), q = l & ((64u >> lg) - 1u);
        s = (g << lg) + f;
        p = (b << 6) + (j << (6u - lg)) + q;
    } else {
        const uint32_t rr = r - n8g, rest = fr.n_tile_pixels - fr.n8;
        const uint32_t f = fdiv(rr, fr.div_rest);
        s = (g << lg) + f;
        p = fr.n8 + (rr - f * rest);
    }
}
// what the shading needs to know of a slot: its frame within the batch, its pixel (tile column, image row, index
// in the image).  ~50 integer instructions with two data-dependent branches, so a kernel computes it once per
// slot and hands it to every vertex it shades (the compiler does not merge the copies by itself).
struct SlotId {
    uint32_t s, tx, gy, pixel;
};
__device__ __forceinline__ SlotId slot_id(const Frame& fr, uint32_t slot) {
    uint32_t s, p, tx, ty;
    slot_sp(fr, fr.grp_log2, slot, s, p);
    slot_pixel(fr, p, tx, ty);
    const uint32_t gy = frame_row(fr, ty);
    return {s, tx, gy, gy * fr.width + fr.x0 + tx};
}
__device__ __forceinline__ size_t sp_slot(const Frame& fr, uint32_t lg, uint32_t s, uint32_t p) {
    if (lg == 0) return (size_t)s * fr.n_tile_pixels + p;
    const uint32_t g = s >> lg, f = s & ((1u << lg) - 1u);
    const size_t base = (size_t)(g << lg) * fr.n_tile_pixels;
    if (p < fr.n8) {
        const uint32_t b = p >> 6, w = p & 63u;
        const uint32_t j = w >> (6u - lg), q = w & ((64u >> lg) - 1u);
        return base + ((((size_t)b << lg) + j) << 6) + (f << (6u - lg)) + q;
    }
    return base + ((size_t)fr.n8 << lg) + (size_t)f * (fr.n_tile_pixels - fr.n8) + (p - fr.n8);
}
__device__ __forceinline__ PixelArgs pixel_args(uint32_t px, uint32_t py, uint32_t w, uint32_t h, double dx, double dy,
                                                double aspect /* (double)w / (double)h: 1.0 for the reference's square images (F7) */) {
    double pixel_size_x = 1. / (double)w, pixel_size_y = 1. / (double)h;
    double pixel_ndc_x = ((double)px + 0.5) / (double)w;
    double pixel_ndc_y = ((double)py + 0.5) / (double)h;
    double pixel_ss_x = 2. * pixel_ndc_x - 1.;
    double pixel_ss_y = 1. - 2. * pixel_ndc_y;
    return {(pixel_ss_x + dx) * aspect, pixel_ss_y + dy, 2. * pixel_size_x * aspect, 2. * pixel_size_y};
}
// pixel_args through Frame::pix_ss where the call has the table: the same operations, two of them (and both
// divisions) done once per column / row by k_pixel_table
__device__ __forceinline__ PixelArgs pixel_args(const Frame& fr, uint32_t px, uint32_t py, double dx, double dy) {
    if (!fr.pix_ss) return pixel_args(px, py, fr.width, fr.height, dx, dy, fr.aspect);
    const double pixel_size_x = 1. / (double)fr.width, pixel_size_y = 1. / (double)fr.height;  // (only ray differentials use them)
    return {(fr.pix_ss[px] + dx) * fr.aspect, fr.pix_ss[fr.width + py] + dy, 2. * pixel_size_x * fr.aspect, 2. * pixel_size_y};
}
__device__ __forceinline__ D3 camera_dir(const Camera& c, double x, double y) {
    return normalize(ld3(c.view_dir) + ld3(c.axes0) * x + ld3(c.axes1) * y);
}


}  // namespace ptd
