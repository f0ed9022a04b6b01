// pt_host.h — host-side internals shared by the C-ABI implementation files.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/pt.h"

namespace pth {

// thread-local error text behind pt_last_error()
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
const char* last_error();

// pt_bvh.cpp
int bvh_build(const float* verts, uint32_t n_verts, const uint32_t* tris, uint32_t n_tris, uint32_t max_leaf,
              std::vector<pt_bvh_node>& out_nodes, std::vector<uint32_t>& out_order, uint32_t& out_depth);
void bvh_widen(std::vector<pt_bvh_node>& nodes, const float* verts, uint32_t n_verts);
int bvh_validate(const pt_bvh_node* nodes, uint32_t n_nodes, const uint32_t* order, uint32_t n_tris,
                 uint32_t stack_depth, uint32_t& depth_out, std::string& why);

// pt_image.cpp
int decode_png(const char* path, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgb8);
// mipmap.rs:85-98 + 155-171: f64 RGB pyramid, all levels concatenated, `offset[l]` in texels
int build_pyramid(const pt_texture& tex, std::vector<double>& texels, std::vector<uint32_t>& offset);

}  // namespace pth
