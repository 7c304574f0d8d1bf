// rt.hpp — C++ facade of the trace loop in the reference's own idiom, meant to
// be dropped into the reference tree as code/rt/rt.hpp (+ rt.cpp) and
// unity-included from code/first.cpp:10-23 like every other module.
//
// It adds files and ONE call in the frame loop (code/first.cpp:78-88, between
// win32_message_loop() and gfx_render()); it edits no existing module
// (SURVEY.md §0 item 3, §8b).  Everything here is a thin, allocation-free
// wrapper over the C ABI in include/rtgpu.h:
//   * namespace rt, free functions, snake_case with the module prefix,
//     `_or_panic` for calls that end in errf() on failure
//     (window_create_or_panic code/window/window.hpp:3, gfx_init_or_panic
//     code/gfx/gfx.hpp:21, errf code/first.cpp:27-43);
//   * PODs built from f32/u32/s64/Vec3/Vec4 (code/first.hpp:16-37,
//     code/math/vec.hxx:18-48); materials are a tagged POD, not a hierarchy;
//   * long-lived host buffers are the caller's (alloc_perm style,
//     code/base/memory.cxx:23-34) — nothing here allocates host memory;
//   * colours 0xRRGGBBAA (code/gfx/gfx.hpp:4-6), rows bottom-up
//     (code/math/mat.cxx:104-129).
//
// Inside the reference tree the scalar typedefs and Vec3/Vec4 come from
// first.hpp / math/vec.hxx; define RT_FACADE_STANDALONE (the headless Linux
// driver does) to get layout-identical restatements instead, because
// first.hpp refuses every compiler but 64-bit MSVC (code/first.hpp:1-11).
#pragma once

#include "rtgpu.h"

#ifdef RT_FACADE_STANDALONE
namespace rt {
using u8  = unsigned char;       // code/first.hpp:16-27
using u32 = unsigned int;
using s64 = signed long long;
using f32 = float;
struct Buffer { s64 count; u8 *bytes; };   // code/first.hpp:29-32
union Vec3 {                     // code/math/vec.hxx:18-28 (xy alias omitted)
  struct { f32 x, y, z; };
  f32 v[3] = {};
};
union Vec4 {                     // code/math/vec.hxx:30-48
  struct { f32 x, y, z, w; };
  struct { f32 r, g, b, a; };
  f32 v[4] = {};
};
struct Mat4x4 { f32 v[4][4]; };   // code/math/mat.hxx:2-4; row-vector convention, mat.cxx:12-20
}  // namespace rt
template <typename... TArgs>
[[noreturn]] void errf(char const *fmt, TArgs... args);  // code/first.hpp:95-97
#endif

namespace rt {
enum Material_Type : u32 {
  MATERIAL_LAMBERTIAN = RTGPU_MAT_LAMBERTIAN,
  MATERIAL_METAL      = RTGPU_MAT_METAL,      // param = fuzz
  MATERIAL_DIELECTRIC = RTGPU_MAT_DIELECTRIC, // param = index of refraction
  MATERIAL_EMISSIVE   = RTGPU_MAT_EMISSIVE    // a light ("Lights", notes/plan.md:16): emits albedo * param, ends the path
};

// Layout-identical to rtgpu_material / rtgpu_sphere / rtgpu_triangle, so a
// Scene is handed to the C layer without a copy.
struct Material { Vec3 albedo; f32 param; u32 type; u32 _unused[3]; };
struct Sphere   { Vec3 center; f32 radius; u32 material; };
struct Triangle { Vec3 v0, v1, v2; u32 material; };

struct Scene {
  Sphere   *spheres;    s64 spheres_count;
  Triangle *triangles;  s64 triangles_count;
  Material *materials;  s64 materials_count;
};

using Camera          = ::rtgpu_camera;    // fill with rt_camera_look_at
using Render_Settings = ::rtgpu_settings;  // fill with rt_default_settings
using Render_Stats    = ::rtgpu_stats;

// Render defaults in the style of code/config.hpp:4-12.
u32 constexpr static RT_SPP       = 16;
u32 constexpr static RT_MAX_DEPTH = 8;

void rt_init_or_panic(int device_index = 0);
void rt_shutdown();

void rt_scene_upload_or_panic(Scene const &scene);
// Moved / edited primitives under the uploaded tree (same counts, primitive i
// stays primitive i): boxes and leaf records are recomputed on the GPU, the
// topology is kept.  For per-frame animation; upload again when the tree no
// longer fits the geometry.
void rt_scene_refit_or_panic(Scene const &scene);
// the same with the moved spheres / triangles already in GPU memory (device pointers)
void rt_scene_refit_device_or_panic(Sphere const *d_spheres, s64 spheres_count, Triangle const *d_triangles,
                                    s64 triangles_count);
// A new tree (GPU-built LBVH) over spheres / triangles that already live in GPU memory
// (device pointers; counts and topology are free); the materials come from the host.
void rt_scene_build_device_or_panic(Sphere const *d_spheres, s64 spheres_count, Triangle const *d_triangles,
                                    s64 triangles_count, Material const *materials, s64 materials_count);

[[nodiscard]] Camera rt_camera_look_at(Vec3 from, Vec3 at, Vec3 up, f32 vfov_deg, f32 aspect,
                                       f32 aperture = 0.f, f32 focus_dist = 10.f);
// The same camera from a camera-to-world Mat4x4 in the reference's row-vector convention
// (world = p * M, code/math/mat.cxx:12-20): rows 0..2 = right / up / back axes, row 3 = eye
// (where translate2 puts a translation, mat.cxx:93-102); the camera looks down its -z axis.
[[nodiscard]] Camera rt_camera_from_basis(Mat4x4 const &camera_to_world, f32 vfov_deg, f32 aspect,
                                          f32 aperture = 0.f, f32 focus_dist = 10.f);
[[nodiscard]] Render_Settings rt_default_settings(u32 width, u32 height, u32 spp = RT_SPP,
                                                  u32 max_depth = RT_MAX_DEPTH);

// accum_out: width*height Vec4 {sum r, sum g, sum b, samples}, caller-owned
// (a 1080p frame is 33 MB: alloc_perm, not the 8 MB temp arena,
// code/config.hpp:11).  prim_id_out and stats_out may be null.
void rt_render_or_panic(Camera const &cam, Render_Settings const &settings, Vec4 *accum_out,
                        u32 *prim_id_out = nullptr, Render_Stats *stats_out = nullptr);

// The same frame without blocking the loop (the message pump of
// code/first.cpp:78-88 keeps running): begin, then once per loop iteration ask
// rt_render_done(); when it says true, rt_render_end_or_panic() fills the buffers
// given to begin.  One frame in flight; buffers stay valid until end.
void rt_render_begin_or_panic(Camera const &cam, Render_Settings const &settings, Vec4 *accum_out,
                              u32 *prim_id_out = nullptr);
[[nodiscard]] bool rt_render_done();
void rt_render_end_or_panic(Render_Stats *stats_out = nullptr);

// The progressive frame of an interactive loop: `settings.spp` more samples, the resolved
// 0xRRGGBBAA frame back.  The float sums stay on the GPU between calls (only 4 B per pixel
// cross PCIe); with RTGPU_FLAG_ACCUMULATE the frame adds to them, without it (camera moved,
// settings changed) the sum starts again.  rgba8_out: width*height u32, caller-owned.
void rt_render_resolved_or_panic(Camera const &cam, Render_Settings const &settings, u32 *rgba8_out,
                                 Render_Stats *stats_out = nullptr);

// ---- the control panel's state machine, platform-free --------------------
// What the ImGui panel that replaces ShowDemoWindow (code/imgui/imgui.cpp:25-26) drives, kept apart
// from ImGui and D3D11 so that it runs (and is tested) wherever the trace loop does: the widgets
// only FILL an Rt_Panel_Edit per UI frame (win32_patch/rt_present.cxx), rt_headless --script feeds
// the same struct from a text file.
struct Rt_Panel_State {
  u32    width, height;
  Vec3   cam_from, cam_at;
  f32    vfov_deg, aperture, focus_dist;
  Camera          camera;        // rebuilt from the fields above when the accumulation restarts
  Render_Settings settings;      // settings.spp = samples traced per displayed frame
  bool   restart;                // the next frame starts the accumulation again
  u32    samples_shown;          // samples in the frame on screen
  u32    frames_since_restart;
  f32    mrays_per_s;            // readout: rays of the last frame / its trace-kernel time
};
// One UI frame's worth of widget results; a field is read only if its `set_` flag is true.
struct Rt_Panel_Edit {
  bool set_spp;     u32 spp;
  bool set_depth;   u32 max_depth;
  bool set_camera;  Vec3 cam_from, cam_at; f32 vfov_deg, aperture, focus_dist;
  bool set_sky;     f32 sky_bottom[3], sky_top[3];
  bool restart_pressed;
};
// Defaults: the cover camera of scene S0, 1 spp per displayed frame (progressive), RT_MAX_DEPTH.
void rt_panel_init(Rt_Panel_State &panel, u32 width, u32 height);
// Anything that changes what a sample computes (depth, camera, lens, sky) restarts the accumulation;
// samples per frame does not (it only changes how fast the sum grows).  Values are clamped to what
// rtgpu_settings accepts (spp, depth >= 1; vfov in [1, 179]; aperture >= 0; focus > 0).
void rt_panel_apply(Rt_Panel_State &panel, Rt_Panel_Edit const &edit);
// One iteration of the frame loop (code/first.cpp:78-88): settings.spp more samples through
// rt_render_resolved_or_panic - a fresh sum with sample_offset 0 after a restart, otherwise
// RTGPU_FLAG_ACCUMULATE at the running offset - then the offset, sample count and Mrays/s readout
// advance.  rgba8_out: width*height u32 (0xRRGGBBAA, row 0 = bottom), caller-owned.
void rt_panel_step_or_panic(Rt_Panel_State &panel, u32 *rgba8_out);

// accum -> 0xRRGGBBAA (gamma 2), ready for a B8G8R8A8/R8G8B8A8 upload.
void rt_resolve_rgba8_or_panic(Vec4 const *accum, u32 *rgba8_out, s64 pixel_count);

[[nodiscard]] f32 rt_last_kernel_ms();

// Data formats either side of the loop, whole-file
// like os_read_entire_file / os_write_entire_file_or_panic
// (code/os/filesystem.hxx:4-9).  A loaded Scene owns malloc'ed arrays: release
// with rt_scene_free (or keep it for the process lifetime, alloc_perm style).
[[nodiscard]] Scene rt_scene_load_obj_or_panic(char const *path);
// the same from a file already in memory (what os_read_entire_file_or_panic returns, code/first.hpp:29-32)
[[nodiscard]] Scene rt_scene_parse_obj_or_panic(Buffer file);
void rt_scene_free(Scene &scene);
// rgba8: 0xRRGGBBAA, row 0 = bottom (what rt_resolve_rgba8_or_panic writes)
void rt_write_ppm_or_panic(char const *path, u32 const *rgba8, u32 width, u32 height);
void rt_write_png_or_panic(char const *path, u32 const *rgba8, u32 width, u32 height);

// ---- several GPUs of one node (include/rtgpu.h rtgpu_group_*) ----------------
// One worker thread per GPU inside the C layer; these calls come from the ONE thread that owns
// the window and the D3D11 device (code/gfx/gfx.cpp:36) - the workers never touch either.
// Tiles of settings.tile_size interleaved over the GPUs, scene replicated, one RCCL gather
// to the first device; the frame is the one-GPU frame bit for bit.
void rt_init_multi_or_panic(int const *devices, int devices_count);
void rt_shutdown_multi();
void rt_scene_upload_multi_or_panic(Scene const &scene);
// accum_out (width*height Vec4) and rgba8_out (width*height u32) may each be null
void rt_render_multi_or_panic(Camera const &cam, Render_Settings const &settings, Vec4 *accum_out,
                              u32 *rgba8_out, Render_Stats *stats_out = nullptr);
// two frames may be in flight: begin k+1 before ending k and the gather of k runs beside the trace of k+1
void rt_render_multi_begin_or_panic(Camera const &cam, Render_Settings const &settings);
void rt_render_multi_end_or_panic(Vec4 *accum_out, u32 *rgba8_out, Render_Stats *stats_out = nullptr);
[[nodiscard]] f32 rt_last_kernel_ms_multi(int index);
}  // namespace rt
