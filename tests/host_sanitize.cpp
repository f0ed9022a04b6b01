// Host-side code of the library under AddressSanitizer + UBSan (tests/test_host_sanitized.py builds and runs this
// with g++; no GPU code is involved: pt_obj.cpp, pt_bvh.cpp and pt_image.cpp are plain C++).  Exercises the OBJ /
// MTL / PNG loaders on the reference's model files and on damaged copies of them, the BVH build, validation,
// widening and four-wide collapse, the texture pyramid, resolve, PPM and checkpoint I/O.  Exit status 0 = no
// sanitizer report and every expectation held.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "../pathtracer_amd/csrc/pt_host.h"

#define EXPECT(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #c, pth::last_error()); return 1; } } while (0)

static std::string slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
static void spit(const std::string& p, const std::string& s) {
    std::ofstream f(p, std::ios::binary);
    f.write(s.data(), (std::streamsize)s.size());
}

static int check_tree(const float* verts, uint32_t nv, const uint32_t* tris, uint32_t nt, uint32_t leaf, uint32_t threads) {
    std::vector<pt_bvh_node> nodes;
    std::vector<uint32_t> order;
    uint32_t depth = 0;
    EXPECT(pth::bvh_build(verts, nv, tris, nt, leaf, nodes, order, depth, threads) == PT_OK);
    std::string why;
    uint32_t d2 = 0;
    EXPECT(pth::bvh_validate(nodes.data(), (uint32_t)nodes.size(), order.data(), nt, 32, d2, why) == PT_OK);
    EXPECT(d2 == depth);
    pth::bvh_widen(nodes, verts, nv);
    std::vector<pt_bvh_node4> wide;
    uint32_t need = 0;
    EXPECT(pth::bvh_collapse4(nodes, 64, wide, need) == PT_OK);
    EXPECT(need <= 64 && (nt == 0 || !wide.empty() || nodes.empty() || true));
    return 0;
}

int main(int argc, char** argv) {
    const std::string models = argc > 1 ? argv[1] : "models";
    const std::string tmp = argc > 2 ? argv[2] : "/tmp";
#ifdef PT_THREADS_ONLY  // the ThreadSanitizer build: the multi-threaded builder alone (parallel binning, partition, subtrees)
    {
        std::mt19937 rng(3);
        const uint32_t nt = 150000;
        std::vector<float> v((size_t)nt * 9);
        std::vector<uint32_t> idx((size_t)nt * 3);
        std::uniform_real_distribution<float> u(-10.f, 10.f);
        for (auto& x : v) x = u(rng);
        for (uint32_t i = 0; i < nt * 3; ++i) idx[i] = i;
        for (uint32_t threads : {2u, 5u}) if (check_tree(v.data(), nt * 3, idx.data(), nt, 2, threads)) return 1;
        std::puts("threaded build clean under tsan");
        return 0;
    }
#endif
    // ---- the reference's scenes through the loader, every wiring flag
    for (const char* name : {"CornellBox", "Test", "glossy_planes", "cornell_box"}) {
        for (uint32_t flags : {0u, (uint32_t)(PT_LOAD_GLOSSY | PT_LOAD_TEXTURES | PT_LOAD_SPECULAR), (uint32_t)(PT_LOAD_GLOSSY | PT_LOAD_GGX)}) {
            pt_model* m = nullptr;
            EXPECT(pt_load_obj((models + "/" + name + ".obj").c_str(), flags, &m) == PT_OK && m);
            const pt_scene_desc* d = pt_model_desc(m);
            EXPECT(d->n_tris > 0 && pt_model_count(m) > 0 && pt_model_name(m, 0));
            for (uint32_t leaf : {1u, 2u, 8u}) if (check_tree(d->verts, d->n_verts, d->tris, d->n_tris, leaf, 1)) return 1;
            for (uint32_t t = 0; t < d->n_texs; ++t) {
                std::vector<double> texels;
                std::vector<uint32_t> offset;
                EXPECT(pth::build_pyramid(d->texs[t], texels, offset) == PT_OK && !offset.empty());
            }
            pt_model_destroy(m);
        }
    }
    // ---- damaged inputs: truncated and bit-flipped OBJ / MTL / PNG must fail (or load) cleanly, never crash
    std::mt19937 rng(7);
    for (const char* file : {"Test.obj", "Test.mtl", "floor.png"}) {
        const std::string good = slurp(models + "/" + file);
        EXPECT(!good.empty());
        for (int k = 0; k < 40; ++k) {
            std::string bad = good;
            if (k % 2) bad.resize(rng() % bad.size());
            else for (int j = 0; j < 8; ++j) bad[rng() % bad.size()] = (char)rng();
            const std::string ext = std::strrchr(file, '.');
            if (ext == ".png") {
                spit(tmp + "/pt_sanitize_bad.png", bad);
                uint32_t w = 0, h = 0;
                uint8_t* px = nullptr;
                if (pt_decode_png((tmp + "/pt_sanitize_bad.png").c_str(), &w, &h, &px) == PT_OK) pt_free(px);
            } else {
                // a private copy of the scene next to the damaged file
                spit(tmp + "/pt_sanitize.obj", ext == ".obj" ? bad : slurp(models + "/Test.obj"));
                spit(tmp + "/Test.mtl", ext == ".mtl" ? bad : slurp(models + "/Test.mtl"));
                pt_model* m = nullptr;
                if (pt_load_obj((tmp + "/pt_sanitize.obj").c_str(), PT_LOAD_GLOSSY | PT_LOAD_SPECULAR, &m) == PT_OK) pt_model_destroy(m);
            }
        }
    }
    // ---- BVH builder on random soups, degenerate triangles included, serial and threaded
    for (uint32_t nt : {0u, 1u, 2u, 3u, 17u, 1000u, 70000u}) {
        std::vector<float> v((size_t)nt * 9);
        std::vector<uint32_t> idx((size_t)nt * 3);
        std::uniform_real_distribution<float> u(-10.f, 10.f);
        for (auto& x : v) x = u(rng);
        for (uint32_t i = 0; i < nt; ++i) {
            for (int k = 0; k < 3; ++k) idx[3 * i + k] = 3 * i + k;
            if (i % 7 == 0) for (int a = 0; a < 3; ++a) v[9 * i + 3 + a] = v[9 * i + a];  // two equal vertices
        }
        if (check_tree(v.data(), nt * 3, idx.data(), nt, 2, nt > 65536 ? 3 : 1)) return 1;
    }
    {   // bad index and non-finite vertex are refused
        float v[9] = {0, 0, 0, 1, 0, 0, 0, 1, 0};
        uint32_t t[3] = {0, 1, 5};
        std::vector<pt_bvh_node> nodes; std::vector<uint32_t> order; uint32_t depth;
        EXPECT(pth::bvh_build(v, 3, t, 1, 2, nodes, order, depth) != PT_OK);
        t[2] = 2; v[4] = NAN;
        EXPECT(pth::bvh_build(v, 3, t, 1, 2, nodes, order, depth) != PT_OK);
    }
    // ---- resolve, PPM, checkpoint round trip and damage
    {
        const uint32_t w = 7, h = 5;
        std::vector<double> acc((size_t)w * h * 3);
        for (size_t i = 0; i < acc.size(); ++i) acc[i] = (double)i * 0.37 - 3.;
        acc[4] = NAN; acc[5] = INFINITY;
        std::vector<uint8_t> rgb(acc.size());
        EXPECT(pt_resolve_rgb8(acc.data(), (size_t)w * h, 3, rgb.data()) == PT_OK);
        EXPECT(pt_write_ppm((tmp + "/pt_sanitize.ppm").c_str(), w, h, rgb.data()) == PT_OK);
        pt_checkpoint_info info{w, h, 11, 42}, back{};
        const std::string ck = tmp + "/pt_sanitize.ckpt";
        EXPECT(pt_checkpoint_save(ck.c_str(), &info, acc.data()) == PT_OK);
        std::vector<double> in(acc.size());
        EXPECT(pt_checkpoint_load(ck.c_str(), &back, in.data(), (size_t)w * h) == PT_OK && back.frames == 11 && back.seed == 42);
        EXPECT(std::memcmp(in.data(), acc.data(), acc.size() * sizeof(double)) == 0);
        EXPECT(pt_checkpoint_load(ck.c_str(), &back, in.data(), 3) != PT_OK);          // buffer too small
        std::string raw = slurp(ck);
        raw[raw.size() / 2] ^= 1;
        spit(ck, raw);
        EXPECT(pt_checkpoint_load(ck.c_str(), &back, in.data(), (size_t)w * h) != PT_OK);  // checksum
        raw.resize(30);
        spit(ck, raw);
        EXPECT(pt_checkpoint_load(ck.c_str(), &back, nullptr, 0) != PT_OK);              // truncated header
    }
    std::puts("host code clean under asan + ubsan");
    return 0;
}
