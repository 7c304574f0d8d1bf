// rtgpu_internal.h — declarations shared by the translation units of librtgpu.so.
#ifndef RTGPU_INTERNAL_H
#define RTGPU_INTERNAL_H

#include <string>

#include "rtgpu.h"
#include "rtgpu_format.h"

namespace rtgpu {

struct DevPool;  // dev_pool.h (device code only)

// bvh_build.cpp
int  flatten_scene(const rtgpu_scene_desc *scene, rtgpu_flat_scene *out, std::string &err);
void free_flat_scene(rtgpu_flat_scene *fs);
// input checks of flatten_scene (NULL arrays, counts, material indices, finite geometry)
int  validate_scene(const rtgpu_scene_desc *scene, std::string &err);
// the caller's materials + the precomputed dielectric constants; malloc'ed, 64-byte aligned
rtgpu_material *flatten_materials(const rtgpu_scene_desc *scene);
// 64-byte nodes -> packed 48-byte nodes (malloc'ed; NULL + err if a plane is not finite)
rtgpu_node48 *pack_nodes48(const rtgpu_node *nodes, uint64_t n, std::string &err);
// bvh4_build.cpp: 64-byte BVH2 nodes -> four-wide packed nodes (malloc'ed; NULL + err on failure)
rtgpu_node4 *collapse_nodes4(const rtgpu_node *nodes, uint64_t n, uint64_t *n_out, uint32_t *max_stack,
                             uint32_t **src_out, std::string &err);
// collapse4.hip: the same collapse on the device (numbering breadth first); exact-size hipMalloc'ed arrays
struct Nodes4Device {
  void *d_nodes4 = nullptr, *d_src = nullptr;
  uint64_t n_nodes4 = 0;
  uint32_t max_stack4 = 0;
  double gpu_ms = 0.0;
};
int  collapse_nodes4_device(const void *d_nodes, uint64_t n_nodes, DevPool *pool, Nodes4Device *out, std::string &err);
void free_nodes4_device(Nodes4Device *out, DevPool *pool);
// pack48.hip: four-wide nodes re-read from refitted 64-byte nodes through their source map, on the device
int  refresh_nodes4_device(const void *d_nodes, void *d_nodes4, const void *d_src, uint64_t n4, uint32_t *d_flag,
                           std::string &err);
// pack48.hip: pack_nodes48 on the device (d_flag: one u32 of device scratch, or NULL to allocate it there)
int  pack_nodes48_device(const void *d_nodes, uint64_t n, void *d_nodes48, uint32_t *d_flag, std::string &err);

// slot_tree.hip: the slot tree (rtgpu_format.h) of four-wide nodes + leaf-ordered primitives, on the device.
// d_slots is kept between calls (grow-only, the owner frees it); n_slots == 0: this tree has no slot form.
struct SlotTreeDevice {
  void *d_slots = nullptr;
  size_t cap_bytes = 0;
  uint64_t n_slots = 0;
  // the layout, kept for refreshes after refits: family base and slot of every four-wide node, slot of every primitive
  uint32_t *d_base = nullptr, *d_slot_of = nullptr, *d_tri_slot = nullptr, *d_sph_slot = nullptr;
  size_t cap_base = 0, cap_slot_of = 0, cap_tri_slot = 0, cap_sph_slot = 0;
  bool layout_valid = false;
  uint64_t layout_n4 = 0, layout_tris = 0, layout_sphs = 0, layout_slots = 0;
};
int  build_slot_tree_device(const void *d_nodes4, uint64_t n4, const void *d_tris, uint64_t n_tris, const void *d_sphs,
                            const void *d_sph_meta, uint64_t n_sphs, SlotTreeDevice *out, DevPool *pool,
                            void *stream /* hipStream_t */, std::string &err);
// the same tree after a refit (same topology, same counts): the slots rewritten in place, no scan
int  refresh_slot_tree_device(const void *d_nodes4, const void *d_tris, const void *d_sphs, const void *d_sph_meta,
                              SlotTreeDevice *out, DevPool *pool, void *stream /* hipStream_t */, std::string &err);
void free_slot_tree_device(SlotTreeDevice *out);

// lbvh_build.hip: the same flattened scene built on the GPU (device arrays, caller frees with free_lbvh)
struct LbvhDevice {
  void *d_nodes, *d_tris, *d_sphs, *d_sph_meta;
  uint64_t n_nodes;
  uint32_t max_depth;
  double gpu_ms;  // first kernel to last, HIP events
};
int  lbvh_build_device(const rtgpu_scene_desc *scene, LbvhDevice *out, std::string &err);
// the same over rtgpu_sphere / rtgpu_triangle arrays in device memory (scene gives the counts only)
// (pool: where outputs and temporaries come from and go back to; NULL = hipMalloc / hipFree)
int  lbvh_build_device_ptrs(const rtgpu_scene_desc *scene, const void *d_spheres, const void *d_triangles,
                            DevPool *pool, LbvhDevice *out, std::string &err);
void free_lbvh(LbvhDevice *out, DevPool *pool);

// refit.hip: new geometry under the uploaded tree's topology, in place on the device arrays
// (ws: device scratch of refit_workspace_words(n_nodes) u32 kept between refits, or NULL)
size_t refit_workspace_words(uint64_t n_nodes);
int  refit_device(const rtgpu_scene_desc *scene, void *d_nodes, uint64_t n_nodes, void *d_tris, void *d_sphs,
                  void *d_sph_meta, uint32_t *ws, double *gpu_ms, std::string &err);
// the same with the caller's rtgpu_sphere / rtgpu_triangle arrays already in device memory, and the
// input check that validate_scene does on the host (0 = fine, 1 = bad primitive, 2 = device error)
int  refit_device_ptrs(const void *d_spheres, uint32_t ns, const void *d_triangles, uint32_t nt, void *d_nodes,
                       uint64_t n_nodes, void *d_tris, void *d_sphs, void *d_sph_meta, uint32_t *ws, double *gpu_ms,
                       std::string &err);
int  check_prims_device(const void *d_spheres, uint32_t ns, const void *d_triangles, uint32_t nt, uint32_t n_materials,
                        uint32_t *d_flag, std::string &err, const char *who /* "refit" or "build": prefix of the message */);

}  // namespace rtgpu
// rtgpu_kernels.hip, for group.cpp: an already flattened scene of this process to one more GPU
extern "C" int rtgpu_internal_upload_flat(rtgpu_ctx *ctx, const rtgpu_flat_scene *fs, int take_ownership);
extern "C" int rtgpu_internal_counters(rtgpu_ctx *ctx, uint64_t **d_counters);
// group.cpp, after a failed exchange: drops the context WITHOUT waiting for the device (a stream of it may
// hold work that never completes) and without freeing device memory that such work might still name.
extern "C" void rtgpu_internal_abandon(rtgpu_ctx *ctx);
namespace rtgpu {

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
