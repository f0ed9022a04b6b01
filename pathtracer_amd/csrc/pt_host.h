// pt_host.h — host-side internals shared by the C-ABI implementation files.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/pt.h"

// Four-wide node of the device tree of a deep scene (pt_bvh.cpp bvh_collapse4): the boxes of up to four
// children, one float4 per plane (lane k of each = child k), their references (as pt_bvh_node::child)
// and how many there are.  128 bytes = eight 16-byte loads.
struct pt_bvh_node4 {
    float lo[3][4];
    float hi[3][4];
    int32_t child[4];
    uint32_t count;
    uint32_t pad[3];
};
static_assert(sizeof(pt_bvh_node4) == 128, "device layout");

namespace pth {

// thread-local error text behind pt_last_error()
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
const char* last_error();

// pt_bvh.cpp
int bvh_build(const float* verts, uint32_t n_verts, const uint32_t* tris, uint32_t n_tris, uint32_t max_leaf,
              std::vector<pt_bvh_node>& out_nodes, std::vector<uint32_t>& out_order, uint32_t& out_depth, uint32_t n_threads = 0);
// `top`: the first `top` nodes of `out` are the top of the tree in breadth-first order (held in LDS by the kernels)
int bvh_collapse4(const std::vector<pt_bvh_node>& nodes2, uint32_t budget, std::vector<pt_bvh_node4>& out, uint32_t& need_out,
                  uint32_t top = 0);
void bvh_widen(std::vector<pt_bvh_node>& nodes, const float* verts, uint32_t n_verts);
int bvh_validate(const pt_bvh_node* nodes, uint32_t n_nodes, const uint32_t* order, uint32_t n_tris,
                 uint32_t stack_depth, uint32_t& depth_out, std::string& why);

// pt_image.cpp
int decode_png(const char* path, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgb8);
// mipmap.rs:85-98 + 155-171: f64 RGB pyramid, all levels concatenated, `offset[l]` in texels
int build_pyramid(const pt_texture& tex, std::vector<double>& texels, std::vector<uint32_t>& offset);

}  // namespace pth
