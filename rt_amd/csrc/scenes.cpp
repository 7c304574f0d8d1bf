// scenes.cpp — the procedural scenes of SURVEY.md §8d.  The reference ships no
// scene (run_tree/data holds one 2D shader) and no assets exist
// offline, so every benchmark/parity scene is generated here from a seed.
//
//   S0 "default": RTIOW book-1 cover layout (the literature the reference
//       cites, notes/materials.md:3-5) + a 2-triangle backdrop quad.
//   S1 "mesh":    displaced icosphere (level 8 = 1,310,720 tris) on a ground quad.
//   S2 "soup":    8 displaced icospheres at the corners of a cube, mixed materials.
//
// Host-only, deterministic for a given (kind, seed, param).
#include "rtgpu_internal.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

inline uint32_t pcg(uint32_t v) {
  uint32_t s = v * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
  return (w >> 22) ^ w;
}

struct Rng {
  uint32_t key, ctr;
  explicit Rng(uint32_t seed) : key(pcg(seed)), ctr(0) {}
  float u01() { return (float)(pcg(key + (ctr++) * 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f); }
  float range(float a, float b) { return a + (b - a) * u01(); }
};

struct V3 { float x, y, z; };
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 mul(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 unit(V3 a) {
  float l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
  return mul(a, 1.0f / l);
}

rtgpu_material make_mat(uint32_t type, float r, float g, float b, float param) {
  rtgpu_material m;
  std::memset(&m, 0, sizeof(m));
  m.albedo[0] = r; m.albedo[1] = g; m.albedo[2] = b;
  m.param = param;
  m.type = type;
  return m;
}

rtgpu_sphere make_sphere(float x, float y, float z, float r, uint32_t mat) {
  rtgpu_sphere s;
  s.center[0] = x; s.center[1] = y; s.center[2] = z;
  s.radius = r;
  s.material = mat;
  return s;
}

rtgpu_triangle make_tri(V3 a, V3 b, V3 c, uint32_t mat) {
  rtgpu_triangle t;
  t.v0[0] = a.x; t.v0[1] = a.y; t.v0[2] = a.z;
  t.v1[0] = b.x; t.v1[1] = b.y; t.v1[2] = b.z;
  t.v2[0] = c.x; t.v2[1] = c.y; t.v2[2] = c.z;
  t.material = mat;
  return t;
}

// ---- value-noise fbm (hash lattice, smoothstep interpolation) ---------------
inline float lattice(int x, int y, int z, uint32_t seed) {
  uint32_t h = pcg((uint32_t)x + pcg((uint32_t)y + pcg((uint32_t)z + pcg(seed))));
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}

float value_noise(V3 p, uint32_t seed) {
  float fx = std::floor(p.x), fy = std::floor(p.y), fz = std::floor(p.z);
  int ix = (int)fx, iy = (int)fy, iz = (int)fz;
  float tx = p.x - fx, ty = p.y - fy, tz = p.z - fz;
  tx = tx * tx * (3.f - 2.f * tx);
  ty = ty * ty * (3.f - 2.f * ty);
  tz = tz * tz * (3.f - 2.f * tz);
  float c[2][2][2];
  for (int a = 0; a < 2; a++)
    for (int b = 0; b < 2; b++)
      for (int d = 0; d < 2; d++) c[a][b][d] = lattice(ix + a, iy + b, iz + d, seed);
  float x00 = c[0][0][0] + (c[1][0][0] - c[0][0][0]) * tx;
  float x10 = c[0][1][0] + (c[1][1][0] - c[0][1][0]) * tx;
  float x01 = c[0][0][1] + (c[1][0][1] - c[0][0][1]) * tx;
  float x11 = c[0][1][1] + (c[1][1][1] - c[0][1][1]) * tx;
  float y0 = x00 + (x10 - x00) * ty;
  float y1 = x01 + (x11 - x01) * ty;
  return y0 + (y1 - y0) * tz;
}

float fbm(V3 p, uint32_t seed) {
  float sum = 0.f, amp = 0.5f, freq = 2.0f;
  for (int o = 0; o < 5; o++) {
    sum += amp * value_noise(mul(p, freq), seed + (uint32_t)o);
    amp *= 0.5f;
    freq *= 2.0f;
  }
  return sum;  // in [0, ~0.97)
}

// ---- icosphere -----------------------------------------------------------------
struct UTri { V3 a, b, c; };

std::vector<UTri> icosphere(uint32_t level) {
  const float t = (1.0f + std::sqrt(5.0f)) * 0.5f;
  V3 v[12] = {{-1, t, 0}, {1, t, 0}, {-1, -t, 0}, {1, -t, 0}, {0, -1, t}, {0, 1, t},
              {0, -1, -t}, {0, 1, -t}, {t, 0, -1}, {t, 0, 1}, {-t, 0, -1}, {-t, 0, 1}};
  for (auto &p : v) p = unit(p);
  static const int f[20][3] = {{0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11},
                               {1, 5, 9}, {5, 11, 4}, {11, 10, 2}, {10, 7, 6}, {7, 1, 8},
                               {3, 9, 4}, {3, 4, 2}, {3, 2, 6}, {3, 6, 8}, {3, 8, 9},
                               {4, 9, 5}, {2, 4, 11}, {6, 2, 10}, {8, 6, 7}, {9, 8, 1}};
  std::vector<UTri> cur;
  for (auto &ff : f) cur.push_back({v[ff[0]], v[ff[1]], v[ff[2]]});
  for (uint32_t l = 0; l < level; l++) {
    std::vector<UTri> next;
    next.reserve(cur.size() * 4);
    for (const UTri &tr : cur) {
      // a+b == b+a in IEEE arithmetic, so the midpoint of a shared edge is
      // bit-identical from both triangles: the mesh stays crack-free without
      // a vertex cache.
      V3 ab = unit(add(tr.a, tr.b)), bc = unit(add(tr.b, tr.c)), ca = unit(add(tr.c, tr.a));
      next.push_back({tr.a, ab, ca});
      next.push_back({tr.b, bc, ab});
      next.push_back({tr.c, ca, bc});
      next.push_back({ab, bc, ca});
    }
    cur.swap(next);
  }
  return cur;
}

void append_displaced_icosphere(std::vector<rtgpu_triangle> &out, const std::vector<UTri> &unit_tris,
                                V3 centre, float radius, float amount, uint32_t noise_seed,
                                uint32_t material) {
  size_t base = out.size();
  out.resize(base + unit_tris.size());
  auto displace = [=](V3 d) {
    float r = radius * (1.0f + amount * fbm(d, noise_seed));
    return add(centre, mul(d, r));
  };
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  if (nt > 16) nt = 16;
  if (unit_tris.size() < 4096) nt = 1;
  std::vector<std::thread> th;
  size_t n = unit_tris.size();
  for (unsigned t = 0; t < nt; t++) {
    size_t b = n * t / nt, e = n * (t + 1) / nt;
    auto work = [&, b, e]() {
      for (size_t i = b; i < e; i++) {
        const UTri &u = unit_tris[i];
        out[base + i] = make_tri(displace(u.a), displace(u.b), displace(u.c), material);
      }
    };
    if (nt == 1) work(); else th.emplace_back(work);
  }
  for (auto &x : th) x.join();
}

template <typename T>
T *to_malloc(const std::vector<T> &v) {
  T *p = (T *)std::calloc(v.size() ? v.size() : 1, sizeof(T));
  if (p && !v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
  return p;
}

}  // namespace

extern "C" int rtgpu_scene_builtin(uint32_t kind, uint32_t seed, uint32_t param,
                                   rtgpu_scene_desc *out, rtgpu_camera *cam_out, float aspect) {
  if (!out) return 1;
  std::memset(out, 0, sizeof(*out));
  std::vector<rtgpu_sphere>   sph;
  std::vector<rtgpu_triangle> tri;
  std::vector<rtgpu_material> mat;
  float from[3] = {13, 2, 3}, at[3] = {0, 0, 0}, up[3] = {0, 1, 0};
  float vfov = 20.f;
  Rng rng(seed);

  if (kind == 0) {
    // S0: RTIOW cover.  Ground first so that prim id 0 is the ground sphere.
    mat.push_back(make_mat(RTGPU_MAT_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 0.f));
    sph.push_back(make_sphere(0, -1000, 0, 1000, 0));
    for (int a = -11; a < 11; a++) {
      for (int b = -11; b < 11; b++) {
        float choose = rng.u01();
        float cx = (float)a + 0.9f * rng.u01(), cz = (float)b + 0.9f * rng.u01();
        float dx = cx - 4.f, dz = cz;
        // every branch draws the same number of variates so that the layout of
        // later spheres does not depend on earlier material choices
        float r0 = rng.u01(), r1 = rng.u01(), r2 = rng.u01(), r3 = rng.u01(), r4 = rng.u01(),
              r5 = rng.u01();
        if (std::sqrt(dx * dx + dz * dz) <= 0.9f) continue;
        uint32_t m = (uint32_t)mat.size();
        if (choose < 0.8f)
          mat.push_back(make_mat(RTGPU_MAT_LAMBERTIAN, r0 * r1, r2 * r3, r4 * r5, 0.f));
        else if (choose < 0.95f)
          mat.push_back(make_mat(RTGPU_MAT_METAL, 0.5f + 0.5f * r0, 0.5f + 0.5f * r1,
                                 0.5f + 0.5f * r2, 0.5f * r3));
        else
          mat.push_back(make_mat(RTGPU_MAT_DIELECTRIC, 1.f, 1.f, 1.f, 1.5f));
        sph.push_back(make_sphere(cx, 0.2f, cz, 0.2f, m));
      }
    }
    uint32_t m = (uint32_t)mat.size();
    mat.push_back(make_mat(RTGPU_MAT_DIELECTRIC, 1.f, 1.f, 1.f, 1.5f));
    mat.push_back(make_mat(RTGPU_MAT_LAMBERTIAN, 0.4f, 0.2f, 0.1f, 0.f));
    mat.push_back(make_mat(RTGPU_MAT_METAL, 0.7f, 0.6f, 0.5f, 0.f));
    mat.push_back(make_mat(RTGPU_MAT_LAMBERTIAN, 0.2f, 0.4f, 0.6f, 0.f));
    sph.push_back(make_sphere(0, 1, 0, 1, m));
    sph.push_back(make_sphere(-4, 1, 0, 1, m + 1));
    sph.push_back(make_sphere(4, 1, 0, 1, m + 2));
    // backdrop quad behind the spheres as seen from (13,2,3): exercises the
    // ray-triangle path on the default scene
    V3 q0 = {-14, 0, -12}, q1 = {-14, 0, 12}, q2 = {-14, 6, 12}, q3 = {-14, 6, -12};
    tri.push_back(make_tri(q0, q1, q2, m + 3));
    tri.push_back(make_tri(q0, q2, q3, m + 3));
  } else if (kind == 1 || kind == 2) {
    uint32_t level = param;
    if (level > 9) {
      rtgpu::set_global_error("icosphere level > 9 refused (would exceed 5.2M tris per sphere)");
      return 1;
    }
    std::vector<UTri> ut = icosphere(level);
    mat.push_back(make_mat(RTGPU_MAT_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 0.f));  // ground
    if (kind == 1) {
      mat.push_back(make_mat(RTGPU_MAT_LAMBERTIAN, 0.7f, 0.45f, 0.3f, 0.f));
      append_displaced_icosphere(tri, ut, {0, 0, 0}, 1.0f, 0.15f, pcg(seed ^ 0x5eedu), 1);
      float g = 20.f, y = -1.2f;
      tri.push_back(make_tri({-g, y, -g}, {-g, y, g}, {g, y, g}, 0));
      tri.push_back(make_tri({-g, y, -g}, {g, y, g}, {g, y, -g}, 0));
      from[0] = 0; from[1] = 1; from[2] = 4;
      vfov = 40.f;
    } else {
      mat.push_back(make_mat(RTGPU_MAT_LAMBERTIAN, 0.7f, 0.3f, 0.3f, 0.f));
      mat.push_back(make_mat(RTGPU_MAT_METAL, 0.8f, 0.8f, 0.7f, 0.3f));
      mat.push_back(make_mat(RTGPU_MAT_DIELECTRIC, 1.f, 1.f, 1.f, 1.5f));
      int i = 0;
      for (int sx = -1; sx <= 1; sx += 2)
        for (int sy = -1; sy <= 1; sy += 2)
          for (int sz = -1; sz <= 1; sz += 2, i++)
            // radius 1.75 leaves ~0.25-wide gaps to the sky between neighbours, so
            // most secondary rays stay inside the cluster (incoherent bounces)
            append_displaced_icosphere(tri, ut, {2.f * sx, 2.f * sy, 2.f * sz}, 1.75f, 0.15f,
                                       pcg(seed + (uint32_t)i * 977u), 1u + (uint32_t)(i % 3));
      from[0] = 0.1f; from[1] = 0.05f; from[2] = 0.2f;
      at[0] = 2.f; at[1] = 0.6f; at[2] = 1.2f;
      vfov = 75.f;
    }
  } else {
    rtgpu::set_global_error("unknown builtin scene kind");
    return 1;
  }

  out->spheres = to_malloc(sph);     out->n_spheres = sph.size();
  out->triangles = to_malloc(tri);   out->n_triangles = tri.size();
  out->materials = to_malloc(mat);   out->n_materials = mat.size();
  if (!out->spheres || !out->triangles || !out->materials) {
    rtgpu_scene_free(out);
    rtgpu::set_global_error("out of host memory generating the builtin scene");
    return 1;
  }
  if (cam_out) {
    // pinhole for parity runs; focus distance 10 as in RTIOW
    rtgpu_camera_look_at(cam_out, from, at, up, vfov, aspect, 0.0f, 10.0f);
  }
  return 0;
}

extern "C" void rtgpu_scene_free(rtgpu_scene_desc *scene) {
  if (!scene) return;
  std::free((void *)scene->spheres);
  std::free((void *)scene->triangles);
  std::free((void *)scene->materials);
  std::memset(scene, 0, sizeof(*scene));
}
