// rt.cpp — see rt.hpp.  One global context, like gD3d (code/gfx/gfx.cpp:4-11).
#include "rt.hpp"

namespace rt {
static_assert(sizeof(Material) == sizeof(::rtgpu_material), "Material layout");
static_assert(sizeof(Sphere) == sizeof(::rtgpu_sphere), "Sphere layout");
static_assert(sizeof(Triangle) == sizeof(::rtgpu_triangle), "Triangle layout");
static_assert(sizeof(Vec4) == 16 && sizeof(Vec3) == 12, "vector layout");

static_assert(sizeof(Mat4x4) == 64 && sizeof(Buffer) == 16, "Mat4x4 / Buffer layout");

static ::rtgpu_ctx *gRt_Ctx = nullptr;
static ::rtgpu_group *gRt_Group = nullptr;

void rt_init_or_panic(int device_index) {
  if (gRt_Ctx) return;
  if (::rtgpu_create(&gRt_Ctx, device_index) != 0)
    errf("rt: cannot create the GPU context: %s", ::rtgpu_last_error(nullptr));
}

void rt_shutdown() {
  ::rtgpu_destroy(gRt_Ctx);
  gRt_Ctx = nullptr;
}

static ::rtgpu_scene_desc desc_of(Scene const &scene) {
  ::rtgpu_scene_desc d = {};
  d.spheres   = (::rtgpu_sphere const *)scene.spheres;      d.n_spheres   = (uint64_t)scene.spheres_count;
  d.triangles = (::rtgpu_triangle const *)scene.triangles;  d.n_triangles = (uint64_t)scene.triangles_count;
  d.materials = (::rtgpu_material const *)scene.materials;  d.n_materials = (uint64_t)scene.materials_count;
  return d;
}

void rt_scene_upload_or_panic(Scene const &scene) {
  ::rtgpu_scene_desc d = desc_of(scene);
  if (::rtgpu_upload_scene(gRt_Ctx, &d) != 0)
    errf("rt: scene upload failed: %s", ::rtgpu_last_error(gRt_Ctx));
}

void rt_scene_refit_or_panic(Scene const &scene) {
  ::rtgpu_scene_desc d = desc_of(scene);
  if (::rtgpu_refit(gRt_Ctx, &d) != 0)
    errf("rt: scene refit failed: %s", ::rtgpu_last_error(gRt_Ctx));
}

void rt_scene_refit_device_or_panic(Sphere const *d_spheres, s64 spheres_count, Triangle const *d_triangles,
                                    s64 triangles_count) {
  if (::rtgpu_refit_device(gRt_Ctx, d_spheres, (uint64_t)spheres_count, d_triangles, (uint64_t)triangles_count) != 0)
    errf("rt: scene refit failed: %s", ::rtgpu_last_error(gRt_Ctx));
}

void rt_scene_build_device_or_panic(Sphere const *d_spheres, s64 spheres_count, Triangle const *d_triangles,
                                    s64 triangles_count, Material const *materials, s64 materials_count) {
  if (::rtgpu_build_device(gRt_Ctx, d_spheres, (uint64_t)spheres_count, d_triangles, (uint64_t)triangles_count,
                           (::rtgpu_material const *)materials, (uint64_t)materials_count) != 0)
    errf("rt: scene build failed: %s", ::rtgpu_last_error(gRt_Ctx));
}

Camera rt_camera_look_at(Vec3 from, Vec3 at, Vec3 up, f32 vfov_deg, f32 aspect, f32 aperture,
                         f32 focus_dist) {
  Camera cam;
  ::rtgpu_camera_look_at(&cam, from.v, at.v, up.v, vfov_deg, aspect, aperture, focus_dist);
  return cam;
}

Camera rt_camera_from_basis(Mat4x4 const &camera_to_world, f32 vfov_deg, f32 aspect, f32 aperture, f32 focus_dist) {
  Camera cam;
  ::rtgpu_camera_from_matrix(&cam, &camera_to_world.v[0][0], vfov_deg, aspect, aperture, focus_dist);
  return cam;
}

Render_Settings rt_default_settings(u32 width, u32 height, u32 spp, u32 max_depth) {
  Render_Settings s;
  ::rtgpu_default_settings(&s, width, height, spp, max_depth);
  return s;
}

void rt_render_or_panic(Camera const &cam, Render_Settings const &settings, Vec4 *accum_out,
                        u32 *prim_id_out, Render_Stats *stats_out) {
  if (::rtgpu_render(gRt_Ctx, &cam, &settings, (float *)accum_out, prim_id_out, stats_out) != 0)
    errf("rt: render failed: %s", ::rtgpu_last_error(gRt_Ctx));
}

void rt_render_begin_or_panic(Camera const &cam, Render_Settings const &settings, Vec4 *accum_out,
                              u32 *prim_id_out) {
  if (::rtgpu_render_begin(gRt_Ctx, &cam, &settings, (float *)accum_out, prim_id_out) != 0)
    errf("rt: render failed: %s", ::rtgpu_last_error(gRt_Ctx));
}

bool rt_render_done() {
  int done = 0;
  if (::rtgpu_render_poll(gRt_Ctx, &done) != 0) errf("rt: render failed: %s", ::rtgpu_last_error(gRt_Ctx));
  return done != 0;
}

void rt_render_end_or_panic(Render_Stats *stats_out) {
  if (::rtgpu_render_end(gRt_Ctx, stats_out) != 0)
    errf("rt: render failed: %s", ::rtgpu_last_error(gRt_Ctx));
}

void rt_render_resolved_or_panic(Camera const &cam, Render_Settings const &settings, u32 *rgba8_out,
                                 Render_Stats *stats_out) {
  if (::rtgpu_render_resolved(gRt_Ctx, &cam, &settings, rgba8_out, stats_out) != 0)
    errf("rt: render failed: %s", ::rtgpu_last_error(gRt_Ctx));
}

void rt_panel_init(Rt_Panel_State &panel, u32 width, u32 height) {
  panel = Rt_Panel_State{};
  panel.width  = width;
  panel.height = height;
  panel.cam_from.x = 13.f; panel.cam_from.y = 2.f; panel.cam_from.z = 3.f;   // S0's cover camera
  panel.vfov_deg   = 20.f;
  panel.aperture   = 0.f;
  panel.focus_dist = 10.f;
  panel.settings   = rt_default_settings(width, height, 1, RT_MAX_DEPTH);
  panel.restart    = true;
}

void rt_panel_apply(Rt_Panel_State &panel, Rt_Panel_Edit const &edit) {
  if (edit.set_spp) panel.settings.spp = edit.spp < 1u ? 1u : edit.spp;   // no restart: the sum just grows faster
  if (edit.set_depth) {
    u32 const d = edit.max_depth < 1u ? 1u : edit.max_depth;
    if (d != panel.settings.max_depth) panel.restart = true;
    panel.settings.max_depth = d;
  }
  if (edit.set_camera) {
    panel.cam_from   = edit.cam_from;
    panel.cam_at     = edit.cam_at;
    panel.vfov_deg   = edit.vfov_deg < 1.f ? 1.f : (edit.vfov_deg > 179.f ? 179.f : edit.vfov_deg);
    panel.aperture   = edit.aperture < 0.f ? 0.f : edit.aperture;
    panel.focus_dist = edit.focus_dist > 1e-3f ? edit.focus_dist : 1e-3f;
    panel.restart    = true;
  }
  if (edit.set_sky) {
    for (int k = 0; k < 3; k++) {
      panel.settings.sky_bottom[k] = edit.sky_bottom[k];
      panel.settings.sky_top[k]    = edit.sky_top[k];
    }
    panel.restart = true;
  }
  if (edit.restart_pressed) panel.restart = true;
}

void rt_panel_step_or_panic(Rt_Panel_State &panel, u32 *rgba8_out) {
  if (panel.restart) {
    Vec3 up; up.x = 0.f; up.y = 1.f; up.z = 0.f;
    panel.camera = rt_camera_look_at(panel.cam_from, panel.cam_at, up, panel.vfov_deg,
                                     (f32)panel.width / (f32)panel.height, panel.aperture, panel.focus_dist);
    panel.settings.sample_offset = 0;
    panel.settings.flags &= ~(u32)RTGPU_FLAG_ACCUMULATE;   // this frame starts the sum on the GPU
    panel.samples_shown = 0;
    panel.frames_since_restart = 0;
    panel.restart = false;
  }
  Render_Stats stats = {};
  rt_render_resolved_or_panic(panel.camera, panel.settings, rgba8_out, &stats);
  panel.settings.sample_offset += panel.settings.spp;
  panel.settings.flags |= RTGPU_FLAG_ACCUMULATE;
  panel.samples_shown += panel.settings.spp;
  panel.frames_since_restart++;
  f32 const ms = rt_last_kernel_ms();
  if (ms > 0.f) panel.mrays_per_s = (f32)((double)stats.rays / ((double)ms * 1e3));
}

void rt_resolve_rgba8_or_panic(Vec4 const *accum, u32 *rgba8_out, s64 pixel_count) {
  if (::rtgpu_resolve_rgba8(gRt_Ctx, (float const *)accum, rgba8_out, (uint64_t)pixel_count) != 0)
    errf("rt: resolve failed: %s", ::rtgpu_last_error(gRt_Ctx));
}

Scene rt_scene_load_obj_or_panic(char const *path) {
  ::rtgpu_scene_desc d = {};
  if (::rtgpu_scene_load_obj(path, &d) != 0) errf("rt: %s", ::rtgpu_last_error(nullptr));
  return Scene{(Sphere *)d.spheres,     (s64)d.n_spheres,
               (Triangle *)d.triangles, (s64)d.n_triangles,
               (Material *)d.materials, (s64)d.n_materials};
}

Scene rt_scene_parse_obj_or_panic(Buffer file) {
  ::rtgpu_scene_desc d = {};
  if (::rtgpu_scene_parse_obj((char const *)file.bytes, (uint64_t)file.count, &d) != 0)
    errf("rt: %s", ::rtgpu_last_error(nullptr));
  return Scene{(Sphere *)d.spheres,     (s64)d.n_spheres,
               (Triangle *)d.triangles, (s64)d.n_triangles,
               (Material *)d.materials, (s64)d.n_materials};
}

void rt_scene_free(Scene &scene) {
  ::rtgpu_scene_desc d = {};
  d.spheres = (::rtgpu_sphere const *)scene.spheres;
  d.triangles = (::rtgpu_triangle const *)scene.triangles;
  d.materials = (::rtgpu_material const *)scene.materials;
  ::rtgpu_scene_free(&d);
  scene = Scene{};
}

void rt_write_ppm_or_panic(char const *path, u32 const *rgba8, u32 width, u32 height) {
  if (::rtgpu_write_ppm(path, rgba8, width, height) != 0) errf("rt: %s", ::rtgpu_last_error(nullptr));
}

void rt_write_png_or_panic(char const *path, u32 const *rgba8, u32 width, u32 height) {
  if (::rtgpu_write_png(path, rgba8, width, height) != 0) errf("rt: %s", ::rtgpu_last_error(nullptr));
}

void rt_init_multi_or_panic(int const *devices, int devices_count) {
  if (gRt_Group) return;
  if (::rtgpu_group_create(&gRt_Group, devices, devices_count) != 0)
    errf("rt: cannot create the GPU group: %s", ::rtgpu_group_last_error(nullptr));
}

void rt_shutdown_multi() {
  ::rtgpu_group_destroy(gRt_Group);
  gRt_Group = nullptr;
}

void rt_scene_upload_multi_or_panic(Scene const &scene) {
  ::rtgpu_scene_desc d = desc_of(scene);
  if (::rtgpu_group_upload_scene(gRt_Group, &d, 0) != 0)
    errf("rt: scene upload failed: %s", ::rtgpu_group_last_error(gRt_Group));
}

void rt_render_multi_or_panic(Camera const &cam, Render_Settings const &settings, Vec4 *accum_out,
                              u32 *rgba8_out, Render_Stats *stats_out) {
  if (::rtgpu_group_render(gRt_Group, &cam, &settings, (float *)accum_out, rgba8_out, stats_out) != 0)
    errf("rt: render failed: %s", ::rtgpu_group_last_error(gRt_Group));
}

void rt_render_multi_begin_or_panic(Camera const &cam, Render_Settings const &settings) {
  if (::rtgpu_group_render_begin(gRt_Group, &cam, &settings) != 0)
    errf("rt: render failed: %s", ::rtgpu_group_last_error(gRt_Group));
}

void rt_render_multi_end_or_panic(Vec4 *accum_out, u32 *rgba8_out, Render_Stats *stats_out) {
  if (::rtgpu_group_render_end(gRt_Group, (float *)accum_out, rgba8_out, stats_out) != 0)
    errf("rt: render failed: %s", ::rtgpu_group_last_error(gRt_Group));
}

f32 rt_last_kernel_ms_multi(int index) {
  f32 ms = 0.f;
  if (::rtgpu_group_kernel_ms(gRt_Group, index, &ms) != 0) return -1.f;
  return ms;
}

f32 rt_last_kernel_ms() {
  f32 ms = 0.f;
  if (::rtgpu_last_kernel_ms(gRt_Ctx, &ms) != 0) return -1.f;
  return ms;
}
}  // namespace rt
