// The reference's frame loop (main.rs:381-454) on the C ABI of include/pt.h, from compiled host code: what the
// Rust crate's main() would do through the extern "C" block of INTEGRATION.md, written in C++ because this image
// has no rustc.  Scene::cornell_box() (scene.rs:140-193): load ./models/Test.obj, eye (275, 275, -600) looking down
// +z, camera_const 2; then `iterations` passes of one frame each into a device-resident running sum, the 8-bit
// image resolved on the device whenever a frame is shown, and a P3 PPM at the end (mipmap.rs:25-42).
//
//   g++ -std=c++17 -O2 -Iinclude examples/frame_loop.cpp -Lpathtracer_amd/lib -lpt_hip -Wl,-rpath,pathtracer_amd/lib -o frame_loop
//   ./frame_loop models/Test.obj out.ppm [dim = 512] [iterations = 16] [bounces = 2] [seed = 1]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pt.h"

#define CHECK(call)                                                                          \
    do {                                                                                     \
        int rc_ = (call);                                                                    \
        if (rc_ != PT_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, pt_last_error()); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s scene.obj out.ppm [dim] [iterations] [bounces] [seed]\n", argv[0]); return 2; }
    const uint32_t dim = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 512u;      // CONFIG.dim (main.rs:343)
    const uint32_t iterations = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 16u;
    const uint32_t bounces = argc > 5 ? (uint32_t)std::atoi(argv[5]) : 2u;    // CONFIG.bounces (main.rs:347)
    const uint64_t seed = argc > 6 ? (uint64_t)std::atoll(argv[6]) : 1u;

    // obj.rs:13-107 load_obj
    pt_model* model = nullptr;
    CHECK(pt_load_obj(argv[1], 0, &model));
    pt_scene_desc desc = *pt_model_desc(model);
    const double eye[3] = {275., 275., -600.}, view[3] = {0., 0., 1.}, up[3] = {0., 1., 0.};  // scene.rs:184-186
    for (int a = 0; a < 3; ++a) { desc.camera.eye[a] = eye[a]; desc.camera.view[a] = view[a]; desc.camera.up[a] = up[a]; desc.background[a] = 0.; }
    desc.camera.camera_const = 2.;

    // scene.rs:140-193 + bvh.rs:17-21: the device-resident scene; the library's own defaults for every option
    pt_scene_options options;
    pt_scene_options_init(&options);
    pt_scene* scene = nullptr;
    CHECK(pt_scene_create_ex(&desc, 0, &options, &scene));

    // main.rs:396-400: `pixels` and `iterations`, resident next to the scene
    pt_accum* pixels = nullptr;
    CHECK(pt_accum_create(scene, dim, dim, &pixels));
    pt_render_params params{};
    params.width = params.height = dim;
    params.x1 = params.y1 = dim;
    params.max_bounces = bounces;
    params.flags = PT_FLAG_USE_DIFFS;  // CONFIG.use_diffs defaults to true (main.rs:350)
    params.seed = seed;
    std::vector<uint32_t> window((size_t)dim * dim);  // the minifb buffer (main.rs:438-467)
    for (uint32_t it = 0; it < iterations; ++it) {     // main.rs:400
        params.sample_begin = it;
        params.sample_count = 1;
        CHECK(pt_accum_render(pixels, &params));       // main.rs:409-436
        if ((it + 1) % 8 == 0 || it + 1 == iterations)  // a frame is shown: resolve on the device, 4 bytes per pixel come back
            CHECK(pt_accum_resolve_u32(pixels, it + 1, window.data()));
    }
    std::vector<uint8_t> rgb8((size_t)dim * dim * 3);
    CHECK(pt_accum_resolve_rgb8(pixels, iterations, rgb8.data()));
    CHECK(pt_write_ppm(argv[2], dim, dim, rgb8.data()));  // mipmap.rs:25-42
    pt_stats st{};
    CHECK(pt_get_stats(scene, &st));
    std::printf("%s: %ux%u, %u iterations, %llu samples, %llu rays traced -> %s\n", pt_version(), dim, dim, iterations,
                (unsigned long long)st.samples, (unsigned long long)st.rays, argv[2]);
    // the window's last frame and the PPM hold the same 8-bit image
    for (size_t i = 0; i < window.size(); ++i)
        if (window[i] != (((uint32_t)rgb8[3 * i] << 16) | ((uint32_t)rgb8[3 * i + 1] << 8) | (uint32_t)rgb8[3 * i + 2])) {
            std::fprintf(stderr, "window buffer and PPM differ at pixel %zu\n", i);
            return 1;
        }
    pt_accum_destroy(pixels);
    pt_scene_destroy(scene);
    pt_model_destroy(model);
    return 0;
}
