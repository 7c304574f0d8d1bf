// rt_headless.cpp — headless Linux stand-in for the reference's front-end
// (code/first.cpp cannot be built off Windows): the same
// init chain and "frame loop" shape, with a PPM file where Present would be.
//
//   rt_headless [scene 0|1|2] [width height spp depth level out.ppm]
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
  u32 kind   = argc > 1 ? (u32)std::atoi(argv[1]) : 0;
  u32 width  = argc > 2 ? (u32)std::atoi(argv[2]) : 1366;  // WIN_WIDTH/HEIGHT, code/config.hpp:7-8
  u32 height = argc > 3 ? (u32)std::atoi(argv[3]) : 768;
  u32 spp    = argc > 4 ? (u32)std::atoi(argv[4]) : RT_SPP;
  u32 depth  = argc > 5 ? (u32)std::atoi(argv[5]) : RT_MAX_DEPTH;
  u32 level  = argc > 6 ? (u32)std::atoi(argv[6]) : 6;
  char const *out_path = argc > 7 ? argv[7] : "rt_headless.ppm";

  rt_init_or_panic();

  ::rtgpu_scene_desc builtin;
  Camera cam;
  if (::rtgpu_scene_builtin(kind, 0x52545F31u, level, &builtin, &cam, (f32)width / (f32)height) != 0)
    errf("builtin scene: %s", ::rtgpu_last_error(nullptr));
  Scene scene = {(Sphere *)builtin.spheres,     (s64)builtin.n_spheres,
                 (Triangle *)builtin.triangles, (s64)builtin.n_triangles,
                 (Material *)builtin.materials, (s64)builtin.n_materials};
  rt_scene_upload_or_panic(scene);

  // alloc_perm-style: zeroed, lives until exit (code/base/memory.cxx:23-34)
  Vec4 *accum = (Vec4 *)std::calloc((size_t)width * height, sizeof(Vec4));
  u32  *rgba  = (u32 *)std::calloc((size_t)width * height, sizeof(u32));
  if (!accum || !rgba) errf("out of memory for a %ux%u frame", width, height);

  Render_Settings settings = rt_default_settings(width, height, spp, depth);
  Render_Stats stats;
  auto t0 = std::chrono::steady_clock::now();
  rt_render_or_panic(cam, settings, accum, nullptr, &stats);   // <- the one call of the frame loop
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  rt_resolve_rgba8_or_panic(accum, rgba, (s64)width * height);

  std::printf("scene %u: %llu spheres, %llu triangles; %ux%u, %u spp, depth %u\n", kind,
              (unsigned long long)builtin.n_spheres, (unsigned long long)builtin.n_triangles, width,
              height, spp, depth);
  std::printf("%llu rays, kernel %.3f ms (%.1f Mrays/s), call %.3f ms incl. PCIe copies\n",
              (unsigned long long)stats.rays, rt_last_kernel_ms(),
              (double)stats.rays / (rt_last_kernel_ms() * 1e3), ms);

  std::FILE *f = std::fopen(out_path, "wb");
  if (!f) errf("cannot open %s", out_path);
  std::fprintf(f, "P6 %u %u 255\n", width, height);
  for (u32 y = height; y-- > 0;)  // row 0 is the bottom row (code/math/mat.cxx:104-129)
    for (u32 x = 0; x < width; x++) {
      u32 c = rgba[(size_t)y * width + x];  // 0xRRGGBBAA (code/gfx/gfx.hpp:4-6)
      u8 px[3] = {(u8)(c >> 24), (u8)(c >> 16), (u8)(c >> 8)};
      std::fwrite(px, 1, 3, f);
    }
  std::fclose(f);
  std::printf("wrote %s\n", out_path);
  ::rtgpu_scene_free(&builtin);
  rt_shutdown();
  return 0;
}
