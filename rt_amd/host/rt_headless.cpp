// rt_headless.cpp — headless Linux stand-in for the reference's front-end
// (code/first.cpp cannot be built off Windows): the same
// init chain and "frame loop" shape, with a PPM file where Present would be.
//
//   rt_headless [--gpus N | --devices a,b,..] [--frames F] [scene 0|1|2 | mesh.obj] [width height spp depth level out.ppm]
// A scene argument that is not a number is read as a Wavefront OBJ file and
// framed from its bounding box.
//   --frames F        F progressive frames of `spp` samples each (sample offsets 0, spp, 2 spp ...);
//                     on one GPU through rt_render_resolved_or_panic: the float sums stay on the GPU
//   --gpus N          GPUs 0..N-1 of this node behind the group API (worker thread per GPU, RCCL
//                     tile gather to GPU 0), frames pipelined two deep
//   --devices a,b,..  the same with an explicit device list; a device listed twice rehearses
//                     several ranks on one GPU (tile buffers then move by device copies)
//   --camera-matrix m00,m01,..,m33   camera-to-world Mat4x4 in the reference's row-vector convention
//                     (rows right / up / back / eye, code/math/mat.cxx:12-20,93-102) through
//                     rt_camera_from_basis; --vfov DEG goes with it (default 20)
//   --obj-via buffer  an OBJ scene is read into a Buffer first (what os_read_entire_file_or_panic
//                     returns, code/first.hpp:29-32) and parsed with rt_scene_parse_obj_or_panic
//   --script FILE     the control panel's state machine (rt_panel_*, rt.hpp) fed from a text file in place
//                     of ImGui widgets, one command per line: `spp N`, `depth N`, `camera fx fy fz ax ay az
//                     vfov aperture focus`, `sky br bg bb tr tg tb`, `restart`, and `frame` = one iteration of
//                     the frame loop, whose resolved frame is written to <out>.<k>.ppm (k counts frames from 0)
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define RT_FACADE_STANDALONE 1
#include "rt.hpp"
#include "rt.cpp"

// errf as code/first.cpp:27-43 defines it, minus the Win32 last-error lookup.
template <typename... TArgs>
[[noreturn]] void errf(char const *fmt, TArgs... args) {
  std::fprintf(stderr, "!!! ERROR\n!!! \t");
  if constexpr (sizeof...(args) == 0) std::fprintf(stderr, "%s", fmt);
  else std::fprintf(stderr, fmt, args...);
  std::fprintf(stderr, "\n");
  std::fflush(stderr);
  std::exit(1);
}

using namespace rt;

int main(int argc, char **argv) {
  int devices[64], n_devices = 0, frames = 0;
  bool have_matrix = false, obj_via_buffer = false;
  Mat4x4 cam_matrix = {};
  f32 vfov = 20.f;
  char const *script_path = nullptr;
  while (argc > 1 && argv[1][0] == '-' && argv[1][1] == '-') {
    if (argc < 3) errf("%s needs a value", argv[1]);
    if (!std::strcmp(argv[1], "--gpus")) {
      n_devices = std::atoi(argv[2]);
      if (n_devices < 1 || n_devices > 64) errf("--gpus 1..64");
      for (int i = 0; i < n_devices; i++) devices[i] = i;
    } else if (!std::strcmp(argv[1], "--devices")) {
      for (char *tok = std::strtok(argv[2], ","); tok && n_devices < 64; tok = std::strtok(nullptr, ",")) devices[n_devices++] = std::atoi(tok);
      if (n_devices < 1) errf("--devices a,b,...");
    } else if (!std::strcmp(argv[1], "--camera-matrix")) {
      int k = 0;
      for (char *tok = std::strtok(argv[2], ","); tok && k < 16; tok = std::strtok(nullptr, ","), k++) cam_matrix.v[k / 4][k % 4] = std::strtof(tok, nullptr);
      if (k != 16) errf("--camera-matrix takes 16 comma-separated numbers, got %d", k);
      have_matrix = true;
    } else if (!std::strcmp(argv[1], "--vfov")) {
      vfov = std::strtof(argv[2], nullptr);
    } else if (!std::strcmp(argv[1], "--obj-via")) {
      if (!std::strcmp(argv[2], "buffer")) obj_via_buffer = true;
      else if (std::strcmp(argv[2], "path")) errf("--obj-via buffer|path");
    } else if (!std::strcmp(argv[1], "--script")) {
      script_path = argv[2];
    } else if (!std::strcmp(argv[1], "--frames")) {
      frames = std::atoi(argv[2]);
      if (frames < 1) errf("--frames >= 1");
    } else {
      errf("unknown option %s", argv[1]);
    }
    argv += 2;
    argc -= 2;
  }
  bool const multi = n_devices > 0;
  u32 kind   = argc > 1 ? (u32)std::atoi(argv[1]) : 0;
  u32 width  = argc > 2 ? (u32)std::atoi(argv[2]) : 1366;  // WIN_WIDTH/HEIGHT, code/config.hpp:7-8
  u32 height = argc > 3 ? (u32)std::atoi(argv[3]) : 768;
  u32 spp    = argc > 4 ? (u32)std::atoi(argv[4]) : RT_SPP;
  u32 depth  = argc > 5 ? (u32)std::atoi(argv[5]) : RT_MAX_DEPTH;
  u32 level  = argc > 6 ? (u32)std::atoi(argv[6]) : 6;
  char const *out_path = argc > 7 ? argv[7] : "rt_headless.ppm";

  bool from_file = argc > 1 && (argv[1][0] < '0' || argv[1][0] > '9');

  if (multi) rt_init_multi_or_panic(devices, n_devices);
  else rt_init_or_panic();

  ::rtgpu_scene_desc builtin = {};
  Camera cam;
  Scene scene;
  if (from_file) {
    if (obj_via_buffer) {
      // the reference's file idiom: the whole file in a Buffer, then the parser (os_read_entire_file_or_panic
      // is Win32-only, code/os/filesystem.cxx:2-60; plain stdio stands in for it here)
      std::FILE *f = std::fopen(argv[1], "rb");
      if (!f) errf("cannot open %s", argv[1]);
      std::fseek(f, 0, SEEK_END);
      Buffer file = {};
      file.count = (s64)std::ftell(f);
      std::fseek(f, 0, SEEK_SET);
      file.bytes = (u8 *)std::malloc(file.count > 0 ? (size_t)file.count : 1);
      if (!file.bytes || std::fread(file.bytes, 1, (size_t)file.count, f) != (size_t)file.count) errf("cannot read %s", argv[1]);
      std::fclose(f);
      scene = rt_scene_parse_obj_or_panic(file);
      std::free(file.bytes);   // the parser copies: the Buffer stays the caller's
    } else {
      scene = rt_scene_load_obj_or_panic(argv[1]);
    }
    if (scene.triangles_count == 0) errf("%s holds no faces", argv[1]);
    Vec3 lo = scene.triangles[0].v0, hi = lo;
    for (s64 i = 0; i < scene.triangles_count; i++) {
      Vec3 const *v[3] = {&scene.triangles[i].v0, &scene.triangles[i].v1, &scene.triangles[i].v2};
      for (auto p : v)
        for (int k = 0; k < 3; k++) {
          if (p->v[k] < lo.v[k]) lo.v[k] = p->v[k];
          if (p->v[k] > hi.v[k]) hi.v[k] = p->v[k];
        }
    }
    Vec3 mid, from;
    f32 r = 0.f;
    for (int k = 0; k < 3; k++) {
      mid.v[k] = 0.5f * (lo.v[k] + hi.v[k]);
      if (hi.v[k] - lo.v[k] > r) r = hi.v[k] - lo.v[k];
    }
    from.x = mid.x + 1.2f * r; from.y = mid.y + 0.8f * r; from.z = mid.z + 1.6f * r;
    Vec3 up; up.x = 0.f; up.y = 1.f; up.z = 0.f;
    cam = rt_camera_look_at(from, mid, up, 40.f, (f32)width / (f32)height);
    builtin.n_spheres = 0;
    builtin.n_triangles = (uint64_t)scene.triangles_count;
  } else {
    if (::rtgpu_scene_builtin(kind, 0x52545F31u, level, &builtin, &cam, (f32)width / (f32)height) != 0)
      errf("builtin scene: %s", ::rtgpu_last_error(nullptr));
    scene = Scene{(Sphere *)builtin.spheres,     (s64)builtin.n_spheres,
                  (Triangle *)builtin.triangles, (s64)builtin.n_triangles,
                  (Material *)builtin.materials, (s64)builtin.n_materials};
  }
  if (have_matrix) cam = rt_camera_from_basis(cam_matrix, vfov, (f32)width / (f32)height);
  if (multi) rt_scene_upload_multi_or_panic(scene);
  else rt_scene_upload_or_panic(scene);

  // alloc_perm-style: zeroed, lives until exit (code/base/memory.cxx:23-34)
  Vec4 *accum = (Vec4 *)std::calloc((size_t)width * height, sizeof(Vec4));
  u32  *rgba  = (u32 *)std::calloc((size_t)width * height, sizeof(u32));
  if (!accum || !rgba) errf("out of memory for a %ux%u frame", width, height);

  Render_Settings settings = rt_default_settings(width, height, spp, depth);
  Render_Stats stats = {};
  auto t0 = std::chrono::steady_clock::now();
  s64 pumped = 0;
  double kernel_ms = 0.0;
  unsigned long long rays = 0;
  int script_frames = 0;
  if (script_path) {
    // rows f4 of SURVEY.md §8: the panel's state machine, fed from a file where the ImGui widgets would feed it
    if (multi) errf("--script drives the one-GPU progressive loop");
    std::FILE *f = std::fopen(script_path, "r");
    if (!f) errf("cannot open %s", script_path);
    Rt_Panel_State panel;
    rt_panel_init(panel, width, height);
    {  // the command line's spp / depth are the panel's starting values
      Rt_Panel_Edit e = {};
      e.set_spp = true; e.spp = spp;
      e.set_depth = true; e.max_depth = depth;
      rt_panel_apply(panel, e);
    }
    char line[512];
    while (std::fgets(line, sizeof(line), f)) {
      Rt_Panel_Edit e = {};
      char cmd[32] = "";
      if (std::sscanf(line, "%31s", cmd) != 1 || cmd[0] == '#') continue;
      char const *rest = std::strstr(line, cmd) + std::strlen(cmd);
      if (!std::strcmp(cmd, "frame")) {
        rt_panel_step_or_panic(panel, rgba);
        char path[1024];
        std::snprintf(path, sizeof(path), "%s.%d.ppm", out_path, script_frames);
        rt_write_ppm_or_panic(path, rgba, width, height);
        std::printf("frame %d: %u samples shown, offset now %u, %.0f Mrays/s -> %s\n", script_frames, panel.samples_shown,
                    panel.settings.sample_offset, (double)panel.mrays_per_s, path);
        script_frames++;
        continue;
      } else if (!std::strcmp(cmd, "spp")) {
        e.set_spp = std::sscanf(rest, "%u", &e.spp) == 1;
        if (!e.set_spp) errf("script: spp N");
      } else if (!std::strcmp(cmd, "depth")) {
        e.set_depth = std::sscanf(rest, "%u", &e.max_depth) == 1;
        if (!e.set_depth) errf("script: depth N");
      } else if (!std::strcmp(cmd, "camera")) {
        e.set_camera = std::sscanf(rest, "%f %f %f %f %f %f %f %f %f", &e.cam_from.x, &e.cam_from.y, &e.cam_from.z, &e.cam_at.x,
                                   &e.cam_at.y, &e.cam_at.z, &e.vfov_deg, &e.aperture, &e.focus_dist) == 9;
        if (!e.set_camera) errf("script: camera fx fy fz ax ay az vfov aperture focus");
      } else if (!std::strcmp(cmd, "sky")) {
        e.set_sky = std::sscanf(rest, "%f %f %f %f %f %f", &e.sky_bottom[0], &e.sky_bottom[1], &e.sky_bottom[2], &e.sky_top[0],
                                &e.sky_top[1], &e.sky_top[2]) == 6;
        if (!e.set_sky) errf("script: sky br bg bb tr tg tb");
      } else if (!std::strcmp(cmd, "restart")) {
        e.restart_pressed = true;
      } else {
        errf("script: unknown command %s", cmd);
      }
      rt_panel_apply(panel, e);
    }
    std::fclose(f);
    kernel_ms = rt_last_kernel_ms();
  } else if (multi) {
    // frames pipelined two deep: the gather of frame k runs beside the trace of frame k+1
    int const n_frames = frames ? frames : 1;
    for (int f = 0; f <= n_frames; f++) {
      if (f < n_frames) {
        Render_Settings st = settings;
        st.sample_offset = (u32)f * spp;
        if (f) st.flags |= RTGPU_FLAG_ACCUMULATE;
        rt_render_multi_begin_or_panic(cam, st);
      }
      if (f > 0) {
        bool const last = f == n_frames;
        Render_Stats fs = {};
        rt_render_multi_end_or_panic(last ? accum : nullptr, last ? rgba : nullptr, &fs);
        rays += fs.rays;
      }
    }
    for (int i = 0; i < n_devices; i++) if ((double)rt_last_kernel_ms_multi(i) > kernel_ms) kernel_ms = rt_last_kernel_ms_multi(i);
  } else if (frames) {
    // the interactive loop: every iteration traces `spp` more samples and gets the resolved frame
    for (int f = 0; f < frames; f++) {
      Render_Settings st = settings;
      st.sample_offset = (u32)f * spp;
      if (f) st.flags |= RTGPU_FLAG_ACCUMULATE;
      rt_render_resolved_or_panic(cam, st, rgba, &stats);  // win32_message_loop(); ... gfx_render(); around it
      rays += stats.rays;
    }
    kernel_ms = rt_last_kernel_ms();
  } else {
    // The frame loop of code/first.cpp:78-88 with the trace in flight: begin once,
    // keep "pumping messages" until the frame is done, then collect it.
    rt_render_begin_or_panic(cam, settings, accum);
    while (!rt_render_done()) pumped++;  // win32_message_loop(); gfx_render(); would go here
    rt_render_end_or_panic(&stats);
    rays = stats.rays;
    kernel_ms = rt_last_kernel_ms();
    rt_resolve_rgba8_or_panic(accum, rgba, (s64)width * height);
  }
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

  std::printf("scene %u: %llu spheres, %llu triangles; %ux%u, %u spp x %d frame(s), depth %u, %d GPU(s)\n", kind,
              (unsigned long long)builtin.n_spheres, (unsigned long long)builtin.n_triangles, width,
              height, spp, frames ? frames : 1, depth, multi ? n_devices : 1);
  std::printf("%llu rays in %.3f ms wall (%.1f Mrays/s incl. host copies); last trace kernel %.3f ms; %lld loop iterations meanwhile\n",
              rays, ms, (double)rays / (ms * 1e3), kernel_ms, (long long)pumped);

  // where Present would be: PNG if the name says so, else PPM
  size_t const path_len = strlen(out_path);
  if (path_len > 4 && strcmp(out_path + path_len - 4, ".png") == 0) rt_write_png_or_panic(out_path, rgba, width, height);
  else rt_write_ppm_or_panic(out_path, rgba, width, height);
  std::printf("wrote %s\n", out_path);
  rt_scene_free(scene);
  if (multi) rt_shutdown_multi();
  else rt_shutdown();
  return 0;
}
