// rtgpu_internal.h — declarations shared by the translation units of librtgpu.so.
#ifndef RTGPU_INTERNAL_H
#define RTGPU_INTERNAL_H

#include <string>

#include "rtgpu.h"
#include "rtgpu_format.h"

namespace rtgpu {

// bvh_build.cpp
int  flatten_scene(const rtgpu_scene_desc *scene, rtgpu_flat_scene *out, std::string &err);
void free_flat_scene(rtgpu_flat_scene *fs);

// host_util.cpp
void set_global_error(const std::string &msg);
const char *global_error();

// Tile bookkeeping shared by host code and kernels (host_util.cpp).
struct TileGrid {
  uint32_t tiles_x, tiles_y, n_tiles;
};
TileGrid tile_grid(uint32_t width, uint32_t height, uint32_t tile_size);
uint32_t rank_tiles(uint32_t n_tiles, uint32_t rank, uint32_t world);

}  // namespace rtgpu

#endif
