// pt_image.cpp — host-side image code around the hot path:
//   * PNG decode (the `image::open(..).to_rgb()` step of mipmap.rs:59-74) on top of zlib inflate,
//   * the MipMap pyramid of mipmap.rs:85-98 / 155-171 in the reference's f64 arithmetic,
//   * 8-bit resolve (main.rs:456-467, mipmap.rs:21-23) and the ASCII PPM writer (mipmap.rs:25-42).
#include "pt_host.h"

#include <zlib.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <cstdlib>
#include <cstring>
#include <new>

namespace pth {

namespace {
thread_local std::string g_error;
}

void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
}
const char* last_error() { return g_error.c_str(); }

static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int decode_png(const char* path, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgb8) {
    FILE* f = std::fopen(path, "rb");
    if (!f) { set_error("cannot open %s", path); return PT_ERR_IO; }
    std::vector<uint8_t> file;
    uint8_t buf[65536];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + got);
    std::fclose(f);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) { set_error("%s is not a PNG", path); return PT_ERR_UNSUPPORTED; }
    size_t pos = 8;
    uint32_t depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool have_ihdr = false;
    while (pos + 12 <= file.size()) {
        uint32_t len = be32(&file[pos]);
        const uint8_t* type = &file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) { set_error("%s: truncated chunk", path); return PT_ERR_IO; }
        const uint8_t* data = &file[pos + 8];
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, 4 + len) != be32(data + len)) { set_error("%s: chunk CRC mismatch", path); return PT_ERR_IO; }
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
            w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
            have_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || w == 0 || h == 0 || w > 65536 || h > 65536) { set_error("%s: bad IHDR", path); return PT_ERR_UNSUPPORTED; }
    if (depth != 8 || interlace != 0) { set_error("%s: only 8-bit non-interlaced PNG is supported", path); return PT_ERR_UNSUPPORTED; }
    uint32_t ch;
    switch (ctype) {
        case 0: ch = 1; break;
        case 2: ch = 3; break;
        case 3: ch = 1; break;
        case 4: ch = 2; break;
        case 6: ch = 4; break;
        default: set_error("%s: bad colour type", path); return PT_ERR_UNSUPPORTED;
    }
    size_t stride = (size_t)w * ch;
    // The IHDR is only a claim.  A deflate stream expands at most 1032:1, so the compressed data
    // bounds the pixels the file can hold; a header that promises more is refused before anything of
    // that size is allocated (a 100-byte file cannot make this allocate gigabytes).
    if ((double)(stride + 1) * (double)h > 1032.0 * (double)idat.size() + 1024.0) {
        set_error("%s: IHDR promises %ux%u pixels, the %zu compressed bytes cannot hold them", path, w, h, idat.size());
        return PT_ERR_IO;
    }
    if (ctype == 3 && (plte.empty() || plte.size() % 3 != 0)) { set_error("%s: palette image without a valid PLTE chunk", path); return PT_ERR_IO; }
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf raw_len = (uLongf)raw.size();
    int zr = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || raw_len != raw.size()) { set_error("%s: inflate failed (%d)", path, zr); return PT_ERR_IO; }
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* src = &raw[(stride + 1) * y];
        uint8_t ft = src[0];
        ++src;
        uint8_t* dst = &img[stride * y];
        const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t x = 0; x < stride; ++x) {
            int a = x >= ch ? dst[x - ch] : 0;
            int b = up ? up[x] : 0;
            int c = (up && x >= ch) ? up[x - ch] : 0;
            int v;
            switch (ft) {
                case 0: v = src[x]; break;
                case 1: v = src[x] + a; break;
                case 2: v = src[x] + b; break;
                case 3: v = src[x] + ((a + b) >> 1); break;
                case 4: {
                    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                    int pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    v = src[x] + pr;
                    break;
                }
                default: set_error("%s: bad filter type", path); return PT_ERR_IO;
            }
            dst[x] = (uint8_t)v;
        }
    }
    rgb8.resize((size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        const uint8_t* p = &img[i * ch];
        uint8_t r, g, b;
        if (ctype == 0 || ctype == 4) { r = g = b = p[0]; }
        else if (ctype == 3) {
            size_t k = (size_t)p[0] * 3;
            if (k + 2 >= plte.size()) { set_error("%s: palette index %u beyond the PLTE chunk", path, (unsigned)p[0]); return PT_ERR_IO; }
            r = plte[k]; g = plte[k + 1]; b = plte[k + 2];
        } else { r = p[0]; g = p[1]; b = p[2]; }  // to_rgb() drops alpha (mipmap.rs:62)
        rgb8[3 * i] = r; rgb8[3 * i + 1] = g; rgb8[3 * i + 2] = b;
    }
    return PT_OK;
}

int build_pyramid(const pt_texture& tex, std::vector<double>& texels, std::vector<uint32_t>& offset) {
    auto pow2 = [](uint32_t x) { return x != 0 && (x & (x - 1)) == 0; };
    if (!tex.rgb8 || !pow2(tex.width) || !pow2(tex.height)) {  // mipmap.rs:86-89 panics
        set_error("texture needs power-of-two dimensions (got %ux%u)", tex.width, tex.height);
        return PT_ERR_UNSUPPORTED;
    }
    if (tex.width < tex.height) {  // the reference's loop would run the width down to 0
        set_error("texture narrower than tall is not supported (%ux%u)", tex.width, tex.height);
        return PT_ERR_UNSUPPORTED;
    }
    texels.clear();
    offset.clear();
    uint32_t w = tex.width, h = tex.height;
    size_t n0 = (size_t)w * h;
    texels.resize(n0 * 3);
    for (size_t i = 0; i < n0 * 3; ++i) texels[i] = (double)tex.rgb8[i] / 255.;  // mipmap.rs:68-73
    offset.push_back(0);
    size_t cur = 0;
    while (h > 1) {  // mipmap.rs:92-95
        if (offset.size() >= 16) { set_error("texture too large"); return PT_ERR_UNSUPPORTED; }
        uint32_t nw = w / 2, nh = h / 2;
        size_t next = texels.size() / 3;
        texels.resize((next + (size_t)nw * nh) * 3);
        const double* src = &texels[cur * 3];
        double* dst = &texels[next * 3];
        // mipmap.rs:163-169: i runs h-2, h-4, .., 0 over *flipped* rows: image[i] is data row h-1-i.
        size_t k = 0;
        for (long i = (long)h - 2; i >= 0; i -= 2) {
            const double* r0 = src + (size_t)(h - 1 - (uint32_t)i) * w * 3;      // image[i]
            const double* r1 = src + (size_t)(h - 1 - ((uint32_t)i + 1)) * w * 3;  // image[i + 1]
            for (uint32_t j = 0; j < w; j += 2, ++k)
                for (int c = 0; c < 3; ++c)
                    dst[k * 3 + c] = (r0[j * 3 + c] + r1[j * 3 + c] + r0[(j + 1) * 3 + c] + r1[(j + 1) * 3 + c]) * 0.25;
        }
        offset.push_back((uint32_t)next);
        cur = next;
        w = nw;
        h = nh;
    }
    return PT_OK;
}

}  // namespace pth

extern "C" {

int pt_decode_png(const char* path, uint32_t* width, uint32_t* height, uint8_t** rgb8_out) {
    if (!path || !width || !height || !rgb8_out) { pth::set_error("null argument"); return PT_ERR_INVALID; }
    std::vector<uint8_t> px;
    int rc;
    try {
        rc = pth::decode_png(path, *width, *height, px);
    } catch (const std::bad_alloc&) {  // nothing may unwind across the C ABI
        pth::set_error("%s: out of memory while decoding", path);
        return PT_ERR_NOMEM;
    }
    if (rc != PT_OK) return rc;
    uint8_t* out = (uint8_t*)std::malloc(px.size());
    if (!out) { pth::set_error("out of memory"); return PT_ERR_NOMEM; }
    std::memcpy(out, px.data(), px.size());
    *rgb8_out = out;
    return PT_OK;
}
void pt_free(void* p) { std::free(p); }

int pt_resolve_rgb8(const double* accum_rgb, size_t n_pixels, uint64_t iterations, uint8_t* rgb8) {
    if (!accum_rgb || !rgb8 || iterations == 0) { pth::set_error("bad argument"); return PT_ERR_INVALID; }
    double denom = 1. / (double)iterations;  // main.rs:457
    for (size_t i = 0; i < n_pixels * 3; ++i) {
        double x = accum_rgb[i] * denom;
        x = x < 0. ? 0. : (x > 1. ? 1. : x);  // mipmap.rs:199-207
        double v = x * 255. + 0.5;           // mipmap.rs:22; `as u8` saturates, NaN -> 0
        rgb8[i] = !(v > 0.) ? 0 : (v >= 255. ? 255 : (uint8_t)v);
    }
    return PT_OK;
}

int pt_write_ppm(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb8) {
    if (!path || !rgb8) { pth::set_error("null argument"); return PT_ERR_INVALID; }
    FILE* f = std::fopen(path, "w");
    if (!f) { pth::set_error("cannot create %s", path); return PT_ERR_IO; }
    std::fprintf(f, "P3\n%u %u\n255\n", width, height);  // mipmap.rs:30
    for (size_t i = 0; i < (size_t)width * height; ++i)
        std::fprintf(f, "%u %u %u\n", rgb8[3 * i], rgb8[3 * i + 1], rgb8[3 * i + 2]);
    if (std::fclose(f) != 0) { pth::set_error("write to %s failed", path); return PT_ERR_IO; }
    return PT_OK;
}

namespace {
struct CkptHeader {
    char magic[8];
    uint32_t width, height;
    uint64_t frames, seed, fnv;
    uint64_t reserved;
};
static_assert(sizeof(CkptHeader) == 48, "checkpoint header layout");
const char kCkptMagic[8] = {'P', 'T', 'A', 'C', 'C', 'U', 'M', '1'};
uint64_t fnv1a64(const void* data, size_t n) {
    const unsigned char* p = (const unsigned char*)data;
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}
}  // namespace

int pt_checkpoint_save(const char* path, const pt_checkpoint_info* info, const double* accum_rgb) {
    if (!path || !info || !accum_rgb) { pth::set_error("null argument"); return PT_ERR_INVALID; }
    if (info->width == 0 || info->height == 0) { pth::set_error("empty image"); return PT_ERR_INVALID; }
    const size_t bytes = (size_t)info->width * info->height * 3 * sizeof(double);
    CkptHeader h{};
    std::memcpy(h.magic, kCkptMagic, 8);
    h.width = info->width; h.height = info->height; h.frames = info->frames; h.seed = info->seed;
    h.fnv = fnv1a64(accum_rgb, bytes);
    std::string tmp = std::string(path) + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) { pth::set_error("cannot create %s", tmp.c_str()); return PT_ERR_IO; }
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1 && std::fwrite(accum_rgb, 1, bytes, f) == bytes;
    ok = (std::fclose(f) == 0) && ok;
    if (!ok || std::rename(tmp.c_str(), path) != 0) {
        std::remove(tmp.c_str());
        pth::set_error("write to %s failed", path);
        return PT_ERR_IO;
    }
    return PT_OK;
}

int pt_checkpoint_load(const char* path, pt_checkpoint_info* info, double* accum_rgb, size_t cap_pixels) {
    if (!path || !info) { pth::set_error("null argument"); return PT_ERR_INVALID; }
    FILE* f = std::fopen(path, "rb");
    if (!f) { pth::set_error("cannot open %s", path); return PT_ERR_IO; }
    CkptHeader h{};
    int rc = PT_OK;
    if (std::fread(&h, sizeof h, 1, f) != 1 || std::memcmp(h.magic, kCkptMagic, 8) != 0 || h.width == 0 || h.height == 0) {
        pth::set_error("%s is not a checkpoint", path);
        rc = PT_ERR_IO;
    } else {
        info->width = h.width; info->height = h.height; info->frames = h.frames; info->seed = h.seed;
        const size_t px = (size_t)h.width * h.height, bytes = px * 3 * sizeof(double);
        if (accum_rgb) {
            if (cap_pixels < px) {
                pth::set_error("checkpoint is %ux%u, buffer holds %zu pixels", h.width, h.height, cap_pixels);
                rc = PT_ERR_INVALID;
            } else if (std::fread(accum_rgb, 1, bytes, f) != bytes || std::fgetc(f) != EOF) {
                pth::set_error("%s: truncated or oversized payload", path);
                rc = PT_ERR_IO;
            } else if (fnv1a64(accum_rgb, bytes) != h.fnv) {
                pth::set_error("%s: checksum mismatch", path);
                rc = PT_ERR_IO;
            }
        }
    }
    std::fclose(f);
    return rc;
}

const char* pt_last_error(void) { return pth::last_error(); }
const char* pt_version(void) { return "pathtracer_amd 0.1 (gfx950)"; }

}  // extern "C"
