// host_util.cpp — host-only helpers of the C ABI: error slot, tile bookkeeping
// (SURVEY.md §8e), camera reduction (row a1 set-up) and default settings.
#include "rtgpu_internal.h"

#include <cmath>
#include <cstring>
#include <mutex>

namespace rtgpu {

static std::mutex  g_err_mutex;
static std::string g_err;

void set_global_error(const std::string &msg) {
  std::lock_guard<std::mutex> lock(g_err_mutex);
  g_err = msg;
}

const char *global_error() { return g_err.c_str(); }

TileGrid tile_grid(uint32_t width, uint32_t height, uint32_t tile_size) {
  TileGrid g;
  g.tiles_x = (width + tile_size - 1) / tile_size;
  g.tiles_y = (height + tile_size - 1) / tile_size;
  g.n_tiles = g.tiles_x * g.tiles_y;
  return g;
}

// Tiles t with t % world == rank (interleaved for load balance).
uint32_t rank_tiles(uint32_t n_tiles, uint32_t rank, uint32_t world) {
  if (world == 0 || rank >= world) return 0;
  return n_tiles > rank ? (n_tiles - rank + world - 1) / world : 0;
}

}  // namespace rtgpu

extern "C" uint32_t rtgpu_abi_version(void) { return RTGPU_ABI_VERSION; }

extern "C" uint64_t rtgpu_compact_pixels(uint32_t width, uint32_t height, uint32_t tile_size,
                                         uint32_t rank, uint32_t world_size) {
  if (tile_size == 0 || width == 0 || height == 0) return 0;
  rtgpu::TileGrid g = rtgpu::tile_grid(width, height, tile_size);
  return (uint64_t)rtgpu::rank_tiles(g.n_tiles, rank, world_size) * tile_size * tile_size;
}

// RTIOW look-at thin-lens camera.  f64 throughout, one rounding to f32 at the
// end, so C++, Python and the oracle tests all see the same camera bits.
extern "C" void rtgpu_camera_look_at(rtgpu_camera *cam, const float from[3], const float at[3],
                                     const float up[3], float vfov_deg, float aspect,
                                     float aperture, float focus_dist) {
  double theta = (double)vfov_deg * 3.14159265358979323846 / 180.0;
  double h  = std::tan(theta * 0.5);
  double vh = 2.0 * h, vw = (double)aspect * vh;
  double w[3], u[3], v[3];
  double l = 0;
  for (int k = 0; k < 3; k++) { w[k] = (double)from[k] - (double)at[k]; l += w[k] * w[k]; }
  l = std::sqrt(l);
  for (int k = 0; k < 3; k++) w[k] /= l;
  u[0] = (double)up[1] * w[2] - (double)up[2] * w[1];
  u[1] = (double)up[2] * w[0] - (double)up[0] * w[2];
  u[2] = (double)up[0] * w[1] - (double)up[1] * w[0];
  l = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
  for (int k = 0; k < 3; k++) u[k] /= l;
  v[0] = w[1] * u[2] - w[2] * u[1];
  v[1] = w[2] * u[0] - w[0] * u[2];
  v[2] = w[0] * u[1] - w[1] * u[0];
  std::memset(cam, 0, sizeof(*cam));
  double fd = (double)focus_dist;
  for (int k = 0; k < 3; k++) {
    double hor = fd * vw * u[k], ver = fd * vh * v[k];
    cam->origin[k]     = from[k];
    cam->horizontal[k] = (float)hor;
    cam->vertical[k]   = (float)ver;
    cam->lower_left[k] = (float)((double)from[k] - hor * 0.5 - ver * 0.5 - fd * w[k]);
    cam->u[k]          = (float)u[k];
    cam->v[k]          = (float)v[k];
  }
  cam->lens_radius = aperture * 0.5f;
}

// The same camera from a camera-to-world matrix in the reference's convention
// (code/math/mat.hxx:2-4 `f32 v[4][4]`; code/math/mat.cxx:12-20: ROW vectors, world = p * M,
// so rows 0-2 are the camera's right / up / back axes in world space and row 3 - where
// translate2 puts a translation, mat.cxx:93-102 - is the eye).  The camera looks down its -z
// axis, y up (the y-up screen space of mat.cxx:104-129).  Axes need not be unit or exactly
// orthogonal: the basis is rebuilt as rtgpu_camera_look_at does from eye, eye - back and up.
extern "C" void rtgpu_camera_from_matrix(rtgpu_camera *cam, const float m[16], float vfov_deg, float aspect,
                                         float aperture, float focus_dist) {
  // (0,0,0,1) * M and (0,0,-1,1) * M, spelled as transformed_point spells them
  const float eye[3] = {m[12], m[13], m[14]};
  const float at[3] = {0.f * m[0] + 0.f * m[4] + -1.f * m[8] + 1.f * m[12], 0.f * m[1] + 0.f * m[5] + -1.f * m[9] + 1.f * m[13],
                       0.f * m[2] + 0.f * m[6] + -1.f * m[10] + 1.f * m[14]};
  const float up[3] = {m[4], m[5], m[6]};
  rtgpu_camera_look_at(cam, eye, at, up, vfov_deg, aspect, aperture, focus_dist);
}

extern "C" void rtgpu_default_settings(rtgpu_settings *s, uint32_t width, uint32_t height,
                                       uint32_t spp, uint32_t max_depth) {
  std::memset(s, 0, sizeof(*s));
  s->width = width;
  s->height = height;
  s->spp = spp;
  s->sample_offset = 0;
  s->max_depth = max_depth;
  s->seed = 1;  // render seed 1
  s->tmin = 1e-3f;
  s->sky_bottom[0] = s->sky_bottom[1] = s->sky_bottom[2] = 1.0f;
  s->sky_top[0] = 0.5f; s->sky_top[1] = 0.7f; s->sky_top[2] = 1.0f;
  s->tile_size = 64;
  s->rank = 0;
  s->world_size = 1;
  s->flags = 0;
  // Samples per work item (tools/gpu_chunk_sweep.py): 2 below 16 spp, where longer items
  // would leave too few of them per lane and lengthen the tail of the launch; 4 from 16
  // spp (half the queue traffic and chunk planes: C2 kernel 4.22 -> 4.17 ms, C3 at 64
  // spp 24.2 -> 21.7 ms); beyond 256 spp the chunks grow so that a frame never needs
  // more than 64 chunk planes (64 x 16 B per pixel of context memory)
  s->chunk_samples = spp > 256u ? (spp + 63u) / 64u : (spp >= 16u ? 4u : 2u);
}
