// bvh_build.cpp — host side of SURVEY.md §8a row a2: SAH-binned BVH2 build and
// flatten into the HBM layout of include/rtgpu_format.h.
//
// No reference counterpart exists (the reference has no BVH).
// Conventions followed: POD arrays, zero-initialised long-lived allocations as
// alloc_perm does (code/base/memory.cxx:23-34), f32 arithmetic only.
//
// Determinism: no data-dependent container ordering and a thread-count-independent
// task decomposition (struct Builder), so the same scene always flattens to the
// same bytes (the oracle walks these bytes).
#include "rtgpu_internal.h"
#include "pack48.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <system_error>
#include <thread>
#include <vector>

namespace rtgpu {

namespace {

struct Box {
  float lo[3], hi[3];
  void reset() {
    for (int k = 0; k < 3; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; }
  }
  void grow(const float *l, const float *h) {
    for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], l[k]); hi[k] = std::max(hi[k], h[k]); }
  }
  void grow(const Box &b) { grow(b.lo, b.hi); }
  float half_area() const {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;
    return dx * dy + dy * dz + dz * dx;
  }
};

struct BuildPrim {
  Box      box;       // padded primitive AABB
  float    c[3];      // centroid
  uint32_t index;     // index into the caller's sphere / triangle array
  uint32_t is_sphere;
};

constexpr int   kBins      = 16;
// SAH parameters.  Experiments may override them with RTGPU_SAH="trav,isect,maxleaf"
// (read per build); the defaults are part of what the committed fixtures pin.
int   kMaxLeaf   = 4;
float kTravCost  = 1.0f;
float kIsectCost = 1.0f;
void read_sah_env() {
  kMaxLeaf = 4; kTravCost = 1.0f; kIsectCost = 1.0f;
  if (const char *e = std::getenv("RTGPU_SAH")) {
    float t = 1.f, i = 1.f; int m = 4;
    if (std::sscanf(e, "%f,%f,%d", &t, &i, &m) == 3 && t > 0.f && i > 0.f && m >= 1 && m <= 4) {
      kTravCost = t; kIsectCost = i; kMaxLeaf = m;
    }
  }
}
constexpr uint32_t kMedianDepth = 40;  // below this depth: median splits only

// Pad so that the fma-form slab test of the trace loop (DESIGN.md "slab test")
// stays conservative: 2^-18 relative, with an absolute floor.
inline void pad_box(Box &b) {
  for (int k = 0; k < 3; k++) {
    float m = std::max(std::max(std::fabs(b.lo[k]), std::fabs(b.hi[k])), 1.0f);
    float e = m * (1.0f / 262144.0f);
    b.lo[k] -= e;
    b.hi[k] += e;
  }
}

// One builder per flatten call.  The leaf-ordered primitive arrays are sized up
// front and written by position: leaves are numbered left to right, so the slice
// a subtree writes is known from the number of spheres / triangles left of it.
// Nodes are numbered depth-first, which is NOT known up front, so every subtree
// is built into its own vector with subtree-local indices and stitched into its
// parent's ([parent][left subtree][right subtree]).  Large subtrees near the
// root run as parallel tasks; the bytes produced do not depend on the number of
// threads (each task only touches its own range of `prims` and its own slices).
struct Builder {
  const rtgpu_scene_desc *scene;
  std::vector<BuildPrim>  prims;
  std::vector<rtgpu_tri>  tris;      // sized n_triangles
  std::vector<rtgpu_sph>  sphs;      // sized n_spheres
  std::vector<uint32_t>   sph_meta;  // sized 2 * n_spheres

  struct Sub {                       // one subtree under construction
    std::vector<rtgpu_node> nodes;   // subtree-local indices; node 0 is its root (if it has inner nodes)
    uint32_t max_depth = 0;
    size_t   tri_at = 0, sph_at = 0; // next free slot of this subtree's leaf slices
  };

  uint32_t make_leaf(Sub &S, size_t b, size_t e, uint32_t depth) {
    S.max_depth = std::max(S.max_depth, depth);
    uint32_t n = (uint32_t)(e - b);
    bool sph = prims[b].is_sphere != 0;
    uint32_t first;
    if (sph) {
      first = (uint32_t)S.sph_at;
      for (size_t i = b; i < e; i++, S.sph_at++) {
        const rtgpu_sphere &s = scene->spheres[prims[i].index];
        rtgpu_sph r;
        r.c[0] = s.center[0]; r.c[1] = s.center[1]; r.c[2] = s.center[2]; r.r = s.radius;
        sphs[S.sph_at] = r;
        sph_meta[2 * S.sph_at]     = prims[i].index;  // prim id: spheres come first
        sph_meta[2 * S.sph_at + 1] = s.material;
      }
    } else {
      first = (uint32_t)S.tri_at;
      for (size_t i = b; i < e; i++, S.tri_at++) {
        const rtgpu_triangle &t = scene->triangles[prims[i].index];
        rtgpu_tri r;
        for (int k = 0; k < 3; k++) {
          r.v0[k] = t.v0[k];
          r.e1[k] = t.v1[k] - t.v0[k];
          r.e2[k] = t.v2[k] - t.v0[k];
        }
        r.prim_id  = (uint32_t)scene->n_spheres + prims[i].index;
        r.material = t.material;
        r._pad     = 0;
        tris[S.tri_at] = r;
      }
    }
    return RTGPU_REF_LEAF | (sph ? RTGPU_REF_SPHERE : 0u) | ((n - 1u) << 27) | first;
  }

  // fn(chunk_begin, chunk_end, t) over T equal chunks of [b, e), chunk t on its own thread
  static constexpr size_t kParScanMin = 1u << 18;
  static constexpr int    kParScanThreads = 8;
  template <typename F>
  static void chunked(size_t b, size_t e, int T, F fn) {
    if (T <= 1) { fn(b, e, 0); return; }
    std::vector<std::thread> th;
    size_t n = e - b;
    for (int t = 1; t < T; t++) {
      size_t cb = b + n * (size_t)t / (size_t)T, ce = b + n * (size_t)(t + 1) / (size_t)T;
      try {
        th.emplace_back(fn, cb, ce, t);
      } catch (const std::system_error &) {  // no thread to be had: this chunk runs here
        fn(cb, ce, t);
      }
    }
    fn(b, b + n / (size_t)T, 0);
    for (auto &x : th) x.join();
  }

  // Bounds of prims[b, e) and the split decision: false = make a leaf, true =
  // children are prims[b, mid) and prims[mid, e) (the range is partitioned in place).
  bool choose_split(size_t b, size_t e, uint32_t depth, Box &bounds, size_t &mid) {
    size_t n = e - b;
    Box cb;
    bounds.reset();
    cb.reset();
    size_t n_sph = 0;
    // ranges near the root are scanned by several threads; min/max and counts are
    // exact, so the merged result does not depend on the chunking
    const int T = n >= kParScanMin ? kParScanThreads : 1;
    {
      Box tb[kParScanThreads], tcb[kParScanThreads];
      size_t ts[kParScanThreads];
      chunked(b, e, T, [&](size_t cb_, size_t ce_, int t) {
        Box bb, cc;
        bb.reset(); cc.reset();
        size_t ns = 0;
        for (size_t i = cb_; i < ce_; i++) {
          bb.grow(prims[i].box);
          cc.grow(prims[i].c, prims[i].c);
          ns += prims[i].is_sphere;
        }
        tb[t] = bb; tcb[t] = cc; ts[t] = ns;
      });
      for (int t = 0; t < T; t++) { bounds.grow(tb[t]); cb.grow(tcb[t]); n_sph += ts[t]; }
    }
    bool mixed = n_sph != 0 && n_sph != n;
    if (n == 1) return false;

    mid = 0;
    bool have_split = false;

    if (mixed && n <= (size_t)kMaxLeaf) {
      // small mixed cluster: split by type so each leaf holds one type
      auto it = std::stable_partition(prims.begin() + b, prims.begin() + e,
                                      [](const BuildPrim &p) { return p.is_sphere != 0; });
      mid = (size_t)(it - prims.begin());
      have_split = true;
    }

    if (!have_split && depth < kMedianDepth) {
      // binned SAH over the three axes
      float best_cost = INFINITY;
      int   best_axis = -1, best_bin = -1;
      float parent_area = bounds.half_area();
      // one pass over the range fills the bins of all three axes
      struct Bins { Box box[3][kBins]; size_t cnt[3][kBins]; };
      std::vector<Bins> tbins((size_t)T);
      float scale3[3];
      for (int axis = 0; axis < 3; axis++) {
        float ext = cb.hi[axis] - cb.lo[axis];
        scale3[axis] = ext > 0.f ? (float)kBins / ext : 0.f;
      }
      chunked(b, e, T, [&](size_t cb_, size_t ce_, int t) {
        Bins &B = tbins[(size_t)t];
        for (int axis = 0; axis < 3; axis++)
          for (int i = 0; i < kBins; i++) { B.box[axis][i].reset(); B.cnt[axis][i] = 0; }
        for (size_t i = cb_; i < ce_; i++)
          for (int axis = 0; axis < 3; axis++) {
            int bi = (int)((prims[i].c[axis] - cb.lo[axis]) * scale3[axis]);
            bi = std::min(std::max(bi, 0), kBins - 1);
            B.box[axis][bi].grow(prims[i].box);
            B.cnt[axis][bi]++;
          }
      });
      for (int axis = 0; axis < 3; axis++) {
        float ext = cb.hi[axis] - cb.lo[axis];
        if (!(ext > 0.f)) continue;
        Box   bin_box[kBins];
        size_t bin_cnt[kBins];
        for (int i = 0; i < kBins; i++) {
          bin_box[i].reset(); bin_cnt[i] = 0;
          for (int t = 0; t < T; t++) { bin_box[i].grow(tbins[(size_t)t].box[axis][i]); bin_cnt[i] += tbins[(size_t)t].cnt[axis][i]; }
        }
        float  right_area[kBins];
        size_t right_cnt[kBins];
        Box    acc;
        acc.reset();
        size_t cnt = 0;
        for (int i = kBins - 1; i >= 1; i--) {
          acc.grow(bin_box[i]);
          cnt += bin_cnt[i];
          right_area[i] = acc.half_area();
          right_cnt[i]  = cnt;
        }
        acc.reset();
        cnt = 0;
        for (int i = 0; i < kBins - 1; i++) {
          acc.grow(bin_box[i]);
          cnt += bin_cnt[i];
          if (cnt == 0 || right_cnt[i + 1] == 0) continue;
          float cost = acc.half_area() * (float)cnt + right_area[i + 1] * (float)right_cnt[i + 1];
          if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = i; }
        }
      }
      if (best_axis >= 0) {
        float split_cost = kTravCost + kIsectCost * best_cost / std::max(parent_area, 1e-30f);
        float leaf_cost  = kIsectCost * (float)n;
        if (n <= (size_t)kMaxLeaf && !mixed && leaf_cost <= split_cost) return false;
        float lo = cb.lo[best_axis], ext = cb.hi[best_axis] - cb.lo[best_axis];
        float scale = (float)kBins / ext;
        auto it = std::stable_partition(
            prims.begin() + b, prims.begin() + e, [&](const BuildPrim &p) {
              int bi = (int)((p.c[best_axis] - lo) * scale);
              bi = std::min(std::max(bi, 0), kBins - 1);
              return bi <= best_bin;
            });
        mid = (size_t)(it - prims.begin());
        have_split = mid > b && mid < e;
      }
    }

    if (!have_split) {
      if (n <= (size_t)kMaxLeaf && !mixed) return false;
      // median split along the widest centroid axis (also the depth-bounded path)
      int axis = 0;
      float best = -1.f;
      for (int k = 0; k < 3; k++) {
        float ext = cb.hi[k] - cb.lo[k];
        if (ext > best) { best = ext; axis = k; }
      }
      mid = b + n / 2;
      if (mixed && !(best > 0.f)) {
        auto it = std::stable_partition(prims.begin() + b, prims.begin() + e,
                                        [](const BuildPrim &p) { return p.is_sphere != 0; });
        mid = (size_t)(it - prims.begin());
      } else {
        std::nth_element(prims.begin() + b, prims.begin() + mid, prims.begin() + e,
                         [axis](const BuildPrim &x, const BuildPrim &y) {
                           if (x.c[axis] != y.c[axis]) return x.c[axis] < y.c[axis];
                           if (x.is_sphere != y.is_sphere) return x.is_sphere > y.is_sphere;
                           return x.index < y.index;
                         });
      }
    }
    return true;
  }

  static void set_children(rtgpu_node &nd, const Box &b0, const Box &b1, uint32_t c0, uint32_t c1) {
    for (int k = 0; k < 3; k++) {
      nd.lo0[k] = b0.lo[k]; nd.hi0[k] = b0.hi[k];
      nd.lo1[k] = b1.lo[k]; nd.hi1[k] = b1.hi[k];
    }
    nd.child0 = c0;
    nd.child1 = c1;
    nd._pad[0] = nd._pad[1] = 0;
  }

  // Sequential depth-first build of prims[b, e) into S: returns the child reference
  // (an S-local node index, or a leaf reference) and the bounds.
  uint32_t build(Sub &S, size_t b, size_t e, uint32_t depth, Box &bounds) {
    size_t mid;
    if (!choose_split(b, e, depth, bounds, mid)) return make_leaf(S, b, e, depth);
    uint32_t me = (uint32_t)S.nodes.size();
    S.nodes.emplace_back();
    Box b0, b1;
    uint32_t c0 = build(S, b, mid, depth + 1, b0);
    uint32_t c1 = build(S, mid, e, depth + 1, b1);
    set_children(S.nodes[me], b0, b1, c0, c1);
    return me;
  }

  // Task-parallel build of a large range: the two children are built as separate
  // Subs (the left one on another thread) and stitched behind this node.
  static constexpr size_t   kTaskMinPrims = 1u << 16;
  uint32_t task_depth = 6;  // at most 64 leaf tasks; RTGPU_BUILD_TASK_DEPTH=0 builds on one thread
  uint32_t build_tasks(Sub &S, size_t b, size_t e, uint32_t depth, Box &bounds) {
    if (e - b < kTaskMinPrims || depth >= task_depth) return build(S, b, e, depth, bounds);
    size_t mid;
    if (!choose_split(b, e, depth, bounds, mid)) return make_leaf(S, b, e, depth);
    size_t left_sph = 0;
    for (size_t i = b; i < mid; i++) left_sph += prims[i].is_sphere;
    Sub L, R;
    L.sph_at = S.sph_at;             L.tri_at = S.tri_at;
    R.sph_at = S.sph_at + left_sph;  R.tri_at = S.tri_at + ((mid - b) - left_sph);
    Box b0, b1;
    uint32_t c0 = 0, c1 = 0;
    std::thread left;
    try {
      left = std::thread([&] { c0 = build_tasks(L, b, mid, depth + 1, b0); });
    } catch (const std::system_error &) {  // no thread to be had: build the left subtree here
      c0 = build_tasks(L, b, mid, depth + 1, b0);
    }
    c1 = build_tasks(R, mid, e, depth + 1, b1);
    if (left.joinable()) left.join();
    S.sph_at = R.sph_at;
    S.tri_at = R.tri_at;
    S.max_depth = std::max(S.max_depth, std::max(L.max_depth, R.max_depth));
    // stitch: [me][L.nodes][R.nodes], inner references shifted to S-local indices
    uint32_t me = (uint32_t)S.nodes.size();
    uint32_t l_at = me + 1u, r_at = l_at + (uint32_t)L.nodes.size();
    S.nodes.emplace_back();
    auto append = [&](const Sub &C, uint32_t at) {
      for (rtgpu_node nd : C.nodes) {
        if (!(nd.child0 & RTGPU_REF_LEAF)) nd.child0 += at;
        if (!(nd.child1 & RTGPU_REF_LEAF)) nd.child1 += at;
        S.nodes.push_back(nd);
      }
    };
    append(L, l_at);
    append(R, r_at);
    if (!(c0 & RTGPU_REF_LEAF)) c0 += l_at;
    if (!(c1 & RTGPU_REF_LEAF)) c1 += r_at;
    set_children(S.nodes[me], b0, b1, c0, c1);
    return me;
  }
};

template <typename T>
T *copy_out(const std::vector<T> &v, size_t min_elems = 1) {
  // zeroed, 64-byte aligned, never empty (callers may upload n+pad elements)
  size_t n = std::max(v.size(), min_elems);
  size_t bytes = ((n * sizeof(T) + 63) / 64) * 64;
  void *p = std::aligned_alloc(64, bytes);
  if (!p) return nullptr;
  std::memset(p, 0, bytes);
  if (!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
  return (T *)p;
}

}  // namespace

// a type the kernels do not know would be shaded as a dielectric without its
// precomputed constants: reject it here
static int check_materials(const rtgpu_scene_desc *scene, std::string &err) {
  for (uint64_t i = 0; i < scene->n_materials; i++)
    if (scene->materials[i].type > RTGPU_MAT_EMISSIVE) { err = "material type out of range"; return 1; }
  return 0;
}

int validate_scene(const rtgpu_scene_desc *scene, std::string &err) {
  if (!scene) { err = "scene is NULL"; return 1; }
  if ((scene->n_spheres && !scene->spheres) || (scene->n_triangles && !scene->triangles) ||
      (scene->n_materials && !scene->materials)) {
    err = "scene array pointer is NULL with a non-zero count";
    return 1;
  }
  if (scene->n_spheres > RTGPU_MAX_PRIMS_PER_TYPE || scene->n_triangles > RTGPU_MAX_PRIMS_PER_TYPE) {
    err = "too many primitives (limit 134,217,720 per type)";
    return 1;
  }
  if (scene->n_materials == 0 && (scene->n_spheres || scene->n_triangles)) {
    err = "scene has primitives but no materials";
    return 1;
  }
  if (check_materials(scene, err)) return 1;
  for (uint64_t i = 0; i < scene->n_spheres; i++) {
    const rtgpu_sphere &s = scene->spheres[i];
    if (s.material >= scene->n_materials) { err = "sphere material index out of range"; return 1; }
    if (!(s.radius > 0.f) || !std::isfinite(s.radius) || !std::isfinite(s.center[0]) ||
        !std::isfinite(s.center[1]) || !std::isfinite(s.center[2])) {
      err = "sphere with non-finite centre or non-positive radius";
      return 1;
    }
  }
  for (uint64_t i = 0; i < scene->n_triangles; i++) {
    const rtgpu_triangle &t = scene->triangles[i];
    if (t.material >= scene->n_materials) { err = "triangle material index out of range"; return 1; }
    for (int k = 0; k < 3; k++)
      if (!std::isfinite(t.v0[k]) || !std::isfinite(t.v1[k]) || !std::isfinite(t.v2[k])) {
        err = "triangle with a non-finite vertex";
        return 1;
      }
  }
  return 0;
}

// Flattened materials keep the caller's 32 bytes and use the three pad words of
// a dielectric for values the scatter step would otherwise divide for at every
// hit: 1/ior, and Schlick's r0 = ((1 - ri) / (1 + ri))^2 for ri = 1/ior (front
// face) and ri = ior (back face) - the same IEEE f32 operations, done once.
rtgpu_material *flatten_materials(const rtgpu_scene_desc *scene) {
  std::vector<rtgpu_material> mats(scene->materials, scene->materials + scene->n_materials);
  for (rtgpu_material &m : mats) {
    m._pad[0] = m._pad[1] = m._pad[2] = 0u;
    if (m.type != RTGPU_MAT_DIELECTRIC) continue;
    volatile float ior = m.param;  // volatile: keep each operation a separate rounded f32 op
    volatile float inv = 1.0f / ior;
    volatile float qf = (1.0f - inv) / (1.0f + inv);
    volatile float qb = (1.0f - ior) / (1.0f + ior);
    float v[3] = {inv, qf * qf, qb * qb};
    std::memcpy(m._pad, v, sizeof(v));
  }
  return copy_out(mats);
}

// nodes -> packed 48-byte nodes (include/rtgpu_format.h), malloc'ed and 64-byte
// aligned; NULL with err set if a plane is not finite (or on out-of-memory).
rtgpu_node48 *pack_nodes48(const rtgpu_node *nodes, uint64_t n, std::string &err) {
  size_t bytes = ((size_t)n * sizeof(rtgpu_node48) + 63) / 64 * 64;
  rtgpu_node48 *out = (rtgpu_node48 *)std::aligned_alloc(64, bytes ? bytes : 64);
  if (!out) { err = "out of host memory for the packed nodes"; return nullptr; }
  std::memset(out, 0, bytes ? bytes : 64);
  for (uint64_t i = 0; i < n; i++)
    if (!pack_node48(nodes[i], &out[i])) {
      std::free(out);
      err = "a box plane is not finite or too close to FLT_MAX for the packed node format";
      return nullptr;
    }
  return out;
}

int flatten_scene(const rtgpu_scene_desc *scene, rtgpu_flat_scene *out, std::string &err) {
  std::memset(out, 0, sizeof(*out));
  read_sah_env();
  if (!scene) { err = "scene is NULL"; return 1; }
  if ((scene->n_spheres && !scene->spheres) || (scene->n_triangles && !scene->triangles) ||
      (scene->n_materials && !scene->materials)) {
    err = "scene array pointer is NULL with a non-zero count";
    return 1;
  }
  if (scene->n_spheres > RTGPU_MAX_PRIMS_PER_TYPE || scene->n_triangles > RTGPU_MAX_PRIMS_PER_TYPE) {
    err = "too many primitives (limit 134,217,720 per type)";
    return 1;
  }
  if (scene->n_materials == 0 && (scene->n_spheres || scene->n_triangles)) {
    err = "scene has primitives but no materials";
    return 1;
  }
  if (check_materials(scene, err)) return 1;
  auto t0 = std::chrono::steady_clock::now();

  Builder B;
  B.scene = scene;
  if (const char *e = std::getenv("RTGPU_BUILD_TASK_DEPTH")) {
    int d = std::atoi(e);
    if (d >= 0 && d <= 8) B.task_depth = (uint32_t)d;
  }
  B.prims.reserve(scene->n_spheres + scene->n_triangles);
  for (uint64_t i = 0; i < scene->n_spheres; i++) {
    const rtgpu_sphere &s = scene->spheres[i];
    if (s.material >= scene->n_materials) { err = "sphere material index out of range"; return 1; }
    if (!(s.radius > 0.f) || !std::isfinite(s.radius) || !std::isfinite(s.center[0]) ||
        !std::isfinite(s.center[1]) || !std::isfinite(s.center[2])) {
      err = "sphere with non-finite centre or non-positive radius";
      return 1;
    }
    BuildPrim p;
    for (int k = 0; k < 3; k++) {
      p.box.lo[k] = s.center[k] - s.radius;
      p.box.hi[k] = s.center[k] + s.radius;
      p.c[k]      = s.center[k];
    }
    pad_box(p.box);
    p.index = (uint32_t)i;
    p.is_sphere = 1;
    B.prims.push_back(p);
  }
  for (uint64_t i = 0; i < scene->n_triangles; i++) {
    const rtgpu_triangle &t = scene->triangles[i];
    if (t.material >= scene->n_materials) { err = "triangle material index out of range"; return 1; }
    BuildPrim p;
    for (int k = 0; k < 3; k++) {
      if (!std::isfinite(t.v0[k]) || !std::isfinite(t.v1[k]) || !std::isfinite(t.v2[k])) {
        err = "triangle with a non-finite vertex";
        return 1;
      }
      p.box.lo[k] = std::min(t.v0[k], std::min(t.v1[k], t.v2[k]));
      p.box.hi[k] = std::max(t.v0[k], std::max(t.v1[k], t.v2[k]));
      p.c[k]      = (p.box.lo[k] + p.box.hi[k]) * 0.5f;
    }
    pad_box(p.box);
    p.index = (uint32_t)i;
    p.is_sphere = 0;
    B.prims.push_back(p);
  }

  size_t n = B.prims.size();
  B.tris.resize(scene->n_triangles);
  B.sphs.resize(scene->n_spheres);
  B.sph_meta.resize(scene->n_spheres * 2);
  Builder::Sub top;
  top.nodes.reserve(n ? n : 1);

  // Empty child: a box at +inf.  Its slab interval is [+inf,+inf] or [-inf,-inf]
  // on every axis, and the closest-hit distance starts at FLT_MAX, so it can
  // never pass tnear <= tfar (include/rtgpu_format.h).
  const float inv_lo = INFINITY, inv_hi = INFINITY;
  if (n <= 1) {
    rtgpu_node root;
    std::memset(&root, 0, sizeof(root));
    for (int k = 0; k < 3; k++) {
      root.lo0[k] = root.lo1[k] = inv_lo;
      root.hi0[k] = root.hi1[k] = inv_hi;
    }
    root.child0 = root.child1 = RTGPU_REF_EMPTY;
    if (n == 1) {
      for (int k = 0; k < 3; k++) { root.lo0[k] = B.prims[0].box.lo[k]; root.hi0[k] = B.prims[0].box.hi[k]; }
      root.child0 = B.make_leaf(top, 0, 1, 1);
    }
    top.nodes.push_back(root);
  } else {
    Box bounds;
    uint32_t r = B.build_tasks(top, 0, n, 0, bounds);
    if (r & RTGPU_REF_LEAF) {
      // 2..4 primitives of one type that SAH keeps together: the traversal starts at
      // a node, so the leaf hangs under a root with an empty sibling (as for n == 1)
      rtgpu_node root;
      std::memset(&root, 0, sizeof(root));
      for (int k = 0; k < 3; k++) {
        root.lo0[k] = bounds.lo[k];  root.hi0[k] = bounds.hi[k];
        root.lo1[k] = inv_lo;        root.hi1[k] = inv_hi;
      }
      root.child0 = r;
      root.child1 = RTGPU_REF_EMPTY;
      top.nodes.push_back(root);
      top.max_depth = 1;
    } else if (r != 0) {
      err = "internal: root is not node 0";
      return 1;
    }
  }
  if (top.tri_at != B.tris.size() || top.sph_at != B.sphs.size()) {
    err = "internal: leaf arrays not filled";
    return 1;
  }
  std::vector<rtgpu_node> &nodes = top.nodes;
  // Node order: the builder emits depth-first (a subtree is contiguous, which
  // keeps deep traversal steps within few cache lines); the first
  // RTGPU_TOP_NODES nodes are then re-numbered breadth-first, so that "index <
  // K" means "one of the K nodes nearest the root" - the part of the tree every
  // ray walks, which the kernel keeps in LDS.  Pure renumbering: boxes, child
  // order and therefore every traversal are unchanged.
  if (nodes.size() > 2) {
    const size_t nn = nodes.size();
    std::vector<uint32_t> new_of_old(nn, 0xFFFFFFFFu), order;
    order.reserve(nn);
    order.push_back(0);
    new_of_old[0] = 0;
    for (size_t head = 0; head < order.size() && order.size() < RTGPU_TOP_NODES; head++) {
      const rtgpu_node &nd = nodes[order[head]];
      const uint32_t ch[2] = {nd.child0, nd.child1};
      for (uint32_t c : ch)
        if (!(c & RTGPU_REF_LEAF) && order.size() < RTGPU_TOP_NODES) {
          new_of_old[c] = (uint32_t)order.size();
          order.push_back(c);
        }
    }
    for (size_t i = 0; i < nn; i++)  // the rest keeps its depth-first order
      if (new_of_old[i] == 0xFFFFFFFFu) { new_of_old[i] = (uint32_t)order.size(); order.push_back((uint32_t)i); }
    std::vector<rtgpu_node> re(nn);
    for (size_t k = 0; k < nn; k++) {
      rtgpu_node nd = nodes[order[k]];
      if (!(nd.child0 & RTGPU_REF_LEAF)) nd.child0 = new_of_old[nd.child0];
      if (!(nd.child1 & RTGPU_REF_LEAF)) nd.child1 = new_of_old[nd.child1];
      re[k] = nd;
    }
    nodes.swap(re);
  }

  if (nodes.size() * sizeof(rtgpu_node) >= (1ull << 32) || B.tris.size() * sizeof(rtgpu_tri) >= (1ull << 32)) {
    err = "scene too large: node or triangle buffer would reach 4 GiB (the kernel uses 32-bit offsets)";
    return 1;
  }
  if (top.max_depth > RTGPU_MAX_TREE_DEPTH) {
    err = "internal: BVH deeper than RTGPU_MAX_TREE_DEPTH";
    return 1;
  }

  // +8 zeroed records behind the leaf-ordered arrays: a leaf reference can
  // name at most 8 primitives, so even a corrupt count cannot read past them.
  out->nodes       = copy_out(nodes);
  out->n_nodes     = nodes.size();
  out->tris        = copy_out(B.tris, B.tris.size() + 8);
  out->n_tris      = B.tris.size();
  out->spheres     = copy_out(B.sphs, B.sphs.size() + 8);
  out->n_spheres   = B.sphs.size();
  out->sphere_meta = copy_out(B.sph_meta, B.sph_meta.size() + 16);
  out->materials   = flatten_materials(scene);
  out->n_materials = scene->n_materials;
  out->max_depth   = top.max_depth;
  if (!out->nodes || !out->tris || !out->spheres || !out->sphere_meta || !out->materials) {
    free_flat_scene(out);
    err = "out of host memory while flattening the scene";
    return 1;
  }
  uint64_t wide_min = RTGPU_NODE48_MIN_NODES;
  if (const char *env = std::getenv("RTGPU_WIDE_MIN_NODES")) wide_min = (uint64_t)std::atoll(env);  // experiments only: the oracle's validator holds the constant
  if (out->n_nodes >= wide_min) {
    out->nodes48 = pack_nodes48((const rtgpu_node *)out->nodes, out->n_nodes, err);
    if (!out->nodes48) { free_flat_scene(out); return 1; }
    uint32_t *src4 = nullptr;
    out->nodes4 = collapse_nodes4((const rtgpu_node *)out->nodes, out->n_nodes, &out->n_nodes4, &out->max_stack4, &src4, err);
    out->nodes4_src = src4;
    if (!out->nodes4) { free_flat_scene(out); return 1; }
  }
  out->build_ms =
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

void free_flat_scene(rtgpu_flat_scene *fs) {
  if (!fs) return;
  std::free((void *)fs->nodes);
  std::free((void *)fs->tris);
  std::free((void *)fs->spheres);
  std::free((void *)fs->sphere_meta);
  std::free((void *)fs->materials);
  std::free((void *)fs->nodes48);
  std::free((void *)fs->nodes4);
  std::free((void *)fs->nodes4_src);
  std::memset(fs, 0, sizeof(*fs));
}

}  // namespace rtgpu

extern "C" int rtgpu_flatten_scene_host(const rtgpu_scene_desc *scene, rtgpu_flat_scene *out) {
  std::string err;
  if (!out) return 1;
  int rc = rtgpu::flatten_scene(scene, out, err);
  if (rc) rtgpu::set_global_error(err);
  return rc;
}

extern "C" void rtgpu_flat_scene_free(rtgpu_flat_scene *fs) { rtgpu::free_flat_scene(fs); }

// Diagnostics (not in the public header): one node through the packer of pack48.h.
extern "C" int rtgpu_debug_pack_node48(const void *node64, void *node48) {
  rtgpu_node n;
  std::memcpy(&n, node64, sizeof(n));
  rtgpu_node48 p;
  if (!rtgpu::pack_node48(n, &p)) return 1;
  std::memcpy(node48, &p, sizeof(p));
  return 0;
}
