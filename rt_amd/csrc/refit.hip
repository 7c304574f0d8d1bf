// refit.hip — SURVEY.md §8f row 3 ("GPU LBVH/HLBVH builder + refit"): new vertex
// positions under an unchanged tree.  Animated or edited geometry keeps the
// topology of the uploaded BVH (child references, leaf order, primitive ids) and
// only the leaf-ordered primitive records and the boxes are recomputed, on the
// GPU, from the caller's arrays.  No reference counterpart exists (the reference
// has no BVH).
//
// Works on the traversal format itself (include/rtgpu_format.h), so it applies to
// trees of either builder.  A node stores its CHILDREN's boxes: the box of a leaf
// child is the union of the padded boxes of the leaf's primitives (padding as
// bvh_build.cpp pad_box), the box of an inner child is the union of the two boxes
// that child stores.  Boxes are fitted bottom-up in level passes over a work list,
// as lbvh_build.hip fits its boxes: a node is fitted in the first pass after its
// inner children, and kernel boundaries give the cross-XCD visibility.  Only exact
// min/max and the separately rounded pad arithmetic are involved (-ffp-contract=off),
// so the result is byte-identical to the CPU restatement under oracle/ and, for
// unchanged positions, to the boxes the builders wrote.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "rtgpu_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kFitThreads = 1024;

// (1) leaf-ordered primitive records from the primitives their ids name
__global__ __launch_bounds__(kThreads) void refit_prims(const rtgpu_sphere *sph, uint32_t ns,
                                                         const rtgpu_triangle *tri, uint32_t nt, rtgpu_tri *tris,
                                                         rtgpu_sph *sphs, uint32_t *meta, uint32_t *bad) {
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < ns + nt; i += gridDim.x * kThreads) {
    if (i < ns) {
      uint32_t p = meta[2 * (size_t)i];
      if (p >= ns) { atomicOr(bad, 1u); continue; }
      const rtgpu_sphere s = sph[p];
      rtgpu_sph r;
      r.c[0] = s.center[0]; r.c[1] = s.center[1]; r.c[2] = s.center[2]; r.r = s.radius;
      sphs[i] = r;
      meta[2 * (size_t)i + 1] = s.material;
    } else {
      uint32_t k = i - ns;
      uint32_t p = tris[k].prim_id;
      if (p < ns || p - ns >= nt) { atomicOr(bad, 1u); continue; }
      const rtgpu_triangle t = tri[p - ns];
      rtgpu_tri r;
      for (int c = 0; c < 3; c++) {
        r.v0[c] = t.v0[c];
        r.e1[c] = t.v1[c] - t.v0[c];
        r.e2[c] = t.v2[c] - t.v0[c];
      }
      r.prim_id = p; r.material = t.material; r._pad = 0;
      tris[k] = r;
    }
  }
}

struct Box6 { float lo[3], hi[3]; };

// padded box of primitive p (spheres first, then triangles), as the builders pad it
__device__ __forceinline__ Box6 prim_box(const rtgpu_sphere *sph, uint32_t ns, const rtgpu_triangle *tri, uint32_t p) {
  Box6 b;
  if (p < ns) {
    const rtgpu_sphere s = sph[p];
    for (int k = 0; k < 3; k++) { b.lo[k] = s.center[k] - s.radius; b.hi[k] = s.center[k] + s.radius; }
  } else {
    const rtgpu_triangle t = tri[p - ns];
    for (int k = 0; k < 3; k++) {
      b.lo[k] = fminf(t.v0[k], fminf(t.v1[k], t.v2[k]));
      b.hi[k] = fmaxf(t.v0[k], fmaxf(t.v1[k], t.v2[k]));
    }
  }
  for (int k = 0; k < 3; k++) {
    float m = fmaxf(fmaxf(fabsf(b.lo[k]), fabsf(b.hi[k])), 1.0f);
    float e = m * (1.0f / 262144.0f);
    b.lo[k] -= e;
    b.hi[k] += e;
  }
  return b;
}

__device__ Box6 child_box(uint32_t ref, const rtgpu_node *nodes, const rtgpu_tri *tris, const uint32_t *meta,
                          const rtgpu_sphere *sph, uint32_t ns, const rtgpu_triangle *tri) {
  Box6 o;
  if (ref == RTGPU_REF_EMPTY) {
    for (int k = 0; k < 3; k++) o.lo[k] = o.hi[k] = INFINITY;
  } else if (ref & RTGPU_REF_LEAF) {
    const uint32_t first = RTGPU_REF_FIRST(ref), count = RTGPU_REF_COUNT(ref);
    for (int k = 0; k < 3; k++) { o.lo[k] = INFINITY; o.hi[k] = -INFINITY; }
    for (uint32_t i = first; i < first + count; i++) {
      uint32_t p = (ref & RTGPU_REF_SPHERE) ? meta[2 * (size_t)i] : tris[i].prim_id;
      Box6 b = prim_box(sph, ns, tri, p);
      for (int k = 0; k < 3; k++) { o.lo[k] = fminf(o.lo[k], b.lo[k]); o.hi[k] = fmaxf(o.hi[k], b.hi[k]); }
    }
  } else {
    const rtgpu_node n = nodes[ref];
    for (int k = 0; k < 3; k++) { o.lo[k] = fminf(n.lo0[k], n.lo1[k]); o.hi[k] = fmaxf(n.hi0[k], n.hi1[k]); }
  }
  return o;
}

// (2) one level pass over the work list of nodes not refitted yet (pass 1: every
// node).  done[i] = number of the pass that refitted node i (0 = not yet); an inner
// child counts as ready only if an EARLIER launch refitted it.  Survivors are
// appended to the next list with one atomic per block (see lbvh_build.hip fit_pass).
__global__ __launch_bounds__(kFitThreads) void refit_pass(uint32_t n_nodes, uint32_t pass, rtgpu_node *nodes,
                                                           const rtgpu_tri *tris, const uint32_t *meta,
                                                           const rtgpu_sphere *sph, uint32_t ns,
                                                           const rtgpu_triangle *tri, uint32_t *done,
                                                           const uint32_t *list_in, uint32_t *list_out,
                                                           uint32_t *counts) {
  __shared__ uint32_t s_wave[kFitThreads / 64];
  __shared__ uint32_t s_base;
  const uint32_t n_in = pass == 1u ? n_nodes : counts[pass - 1u];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (uint32_t base = blockIdx.x * kFitThreads; base < n_in; base += gridDim.x * kFitThreads) {  // block-uniform
    const uint32_t idx = base + threadIdx.x;
    bool survive = false;
    uint32_t i = 0;
    if (idx < n_in) {
      i = pass == 1u ? idx : list_in[idx];
      const uint32_t a = nodes[i].child0, b = nodes[i].child1;
      const bool ia = !(a & RTGPU_REF_LEAF), ib = !(b & RTGPU_REF_LEAF);  // EMPTY carries the leaf bit
      const uint32_t da = ia ? done[a] : 1u, db = ib ? done[b] : 1u;
      const bool ra = !ia || (da != 0u && da < pass), rb = !ib || (db != 0u && db < pass);
      if (ra && rb) {
        Box6 ba = child_box(a, nodes, tris, meta, sph, ns, tri);
        Box6 bb = child_box(b, nodes, tris, meta, sph, ns, tri);
        for (int k = 0; k < 3; k++) {
          nodes[i].lo0[k] = ba.lo[k]; nodes[i].hi0[k] = ba.hi[k];
          nodes[i].lo1[k] = bb.lo[k]; nodes[i].hi1[k] = bb.hi[k];
        }
        done[i] = pass;
      } else {
        survive = true;
      }
    }
    const unsigned long long m = __ballot(survive);
    const uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t total = 0;
      for (int w = 0; w < kFitThreads / 64; w++) { uint32_t c = s_wave[w]; s_wave[w] = total; total += c; }
      s_base = total ? atomicAdd(&counts[pass], total) : 0u;
    }
    __syncthreads();
    if (survive) list_out[s_base + s_wave[wave] + before] = i;
    __syncthreads();  // s_wave / s_base are rewritten by the next round
  }
}

struct DevBufs {  // temporaries, freed on every exit
  std::vector<void *> ptrs;
  ~DevBufs() { for (void *p : ptrs) (void)hipFree(p); }
  template <typename T>
  hipError_t alloc(T **p, size_t n) {
    hipError_t e = hipMalloc((void **)p, (n ? n : 1) * sizeof(T));
    if (e == hipSuccess) ptrs.push_back(*p);
    return e;
  }
};

constexpr int kCounts = 2 * (int)RTGPU_MAX_TREE_DEPTH + 16;  // counts[pass], pass <= kMaxPasses

}  // namespace

namespace rtgpu {

#define RF_TRY(call)                                                   \
  do {                                                                 \
    hipError_t e_ = (call);                                            \
    if (e_ != hipSuccess) {                                            \
      err = std::string("refit: " #call ": ") + hipGetErrorString(e_); \
      if (ev0) (void)hipEventDestroy(ev0);                             \
      if (ev1) (void)hipEventDestroy(ev1);                             \
      return 2;                                                        \
    }                                                                  \
  } while (0)

// The current device must be the context's; `scene` must have passed validate_scene
// and hold the counts the device arrays were built from (the caller checks both).
// On an error return the device arrays may hold a mix of old and new records.
size_t refit_workspace_words(uint64_t n_nodes) { return 3 * (size_t)(n_nodes ? n_nodes : 1) + kCounts + 4; }

int refit_device(const rtgpu_scene_desc *scene, void *d_nodes, uint64_t n_nodes, void *d_tris, void *d_sphs,
                 void *d_sph_meta, uint32_t *ws, double *gpu_ms, std::string &err) {
  const uint32_t ns = (uint32_t)scene->n_spheres, nt = (uint32_t)scene->n_triangles;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;  // unused here: RF_TRY's clean-up names them
  DevBufs tmp;
  rtgpu_sphere *d_sph_in = nullptr;
  rtgpu_triangle *d_tri_in = nullptr;
  RF_TRY(tmp.alloc(&d_sph_in, ns));
  RF_TRY(tmp.alloc(&d_tri_in, nt));
  if (ns) RF_TRY(hipMemcpy(d_sph_in, scene->spheres, (size_t)ns * sizeof(rtgpu_sphere), hipMemcpyHostToDevice));
  if (nt) RF_TRY(hipMemcpy(d_tri_in, scene->triangles, (size_t)nt * sizeof(rtgpu_triangle), hipMemcpyHostToDevice));
  return refit_device_ptrs(d_sph_in, ns, d_tri_in, nt, d_nodes, n_nodes, d_tris, d_sphs, d_sph_meta, ws, gpu_ms, err);
}

// Input check for primitives that only exist on the device (rtgpu_refit_device): what
// validate_scene checks on the host.  Runs BEFORE anything is rewritten.
__global__ __launch_bounds__(kThreads) void check_prims(const rtgpu_sphere *sph, uint32_t ns, const rtgpu_triangle *tri,
                                                         uint32_t nt, uint32_t n_materials, uint32_t *bad) {
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < ns + nt; i += gridDim.x * kThreads) {
    bool ok;
    if (i < ns) {
      const rtgpu_sphere s = sph[i];
      ok = s.material < n_materials && s.radius > 0.f && isfinite(s.radius) && isfinite(s.center[0]) &&
           isfinite(s.center[1]) && isfinite(s.center[2]);
    } else {
      const rtgpu_triangle t = tri[i - ns];
      ok = t.material < n_materials;
      for (int k = 0; k < 3; k++) ok = ok && isfinite(t.v0[k]) && isfinite(t.v1[k]) && isfinite(t.v2[k]);
    }
    if (!ok) atomicOr(bad, 1u);
  }
}

int check_prims_device(const void *d_sph_in, uint32_t ns, const void *d_tri_in, uint32_t nt, uint32_t n_materials,
                       uint32_t *d_flag, std::string &err, const char *who) {
  if (ns + nt == 0) return 0;
  uint32_t *d_bad = d_flag, bad = 0;
  hipError_t e = d_flag ? hipSuccess : hipMalloc((void **)&d_bad, sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemset(d_bad, 0, sizeof(uint32_t));
  if (e == hipSuccess) {
    unsigned blocks = std::min<unsigned>((unsigned)(((size_t)ns + nt + kThreads - 1) / kThreads), 4096u);
    hipLaunchKernelGGL(check_prims, dim3(blocks), dim3(kThreads), 0, 0, (const rtgpu_sphere *)d_sph_in, ns,
                       (const rtgpu_triangle *)d_tri_in, nt, n_materials, d_bad);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost);
  if (!d_flag) (void)hipFree(d_bad);
  if (e != hipSuccess) { err = std::string(who) + ": input check: " + hipGetErrorString(e); return 2; }
  if (bad) { err = std::string(who) + ": a primitive has a non-finite coordinate, a non-positive radius or a material index out of range"; return 1; }
  return 0;
}

// The same with the caller's primitive arrays already on the device.  `ws`: scratch of
// refit_workspace_words(n_nodes) words that a caller keeps between refits (the context
// does, so a frame loop pays no hipMalloc / hipFree), or NULL to allocate it here.
int refit_device_ptrs(const void *d_sph_in_, uint32_t ns, const void *d_tri_in_, uint32_t nt, void *d_nodes,
                      uint64_t n_nodes, void *d_tris, void *d_sphs, void *d_sph_meta, uint32_t *ws, double *gpu_ms,
                      std::string &err) {
  const rtgpu_sphere *d_sph_in = (const rtgpu_sphere *)d_sph_in_;
  const rtgpu_triangle *d_tri_in = (const rtgpu_triangle *)d_tri_in_;
  const uint32_t P = ns + nt;
  const uint32_t nn = (uint32_t)n_nodes;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  DevBufs tmp;
  uint32_t *d_done = nullptr, *d_count = nullptr, *d_list_a = nullptr, *d_list_b = nullptr, *d_bad = nullptr;
  if (!ws) RF_TRY(tmp.alloc(&ws, refit_workspace_words(n_nodes)));
  const size_t stride = nn ? nn : 1;
  d_done = ws;
  d_list_a = ws + stride;
  d_list_b = ws + 2 * stride;
  d_count = ws + 3 * stride;
  d_bad = d_count + kCounts;
  RF_TRY(hipMemset(d_done, 0, (size_t)(nn ? nn : 1) * sizeof(uint32_t)));
  RF_TRY(hipMemset(d_count, 0, kCounts * sizeof(uint32_t)));
  RF_TRY(hipMemset(d_bad, 0, sizeof(uint32_t)));
  RF_TRY(hipEventCreate(&ev0));
  RF_TRY(hipEventCreate(&ev1));
  RF_TRY(hipEventRecord(ev0, 0));
  if (P) {
    unsigned blocks = std::min<unsigned>((unsigned)(((size_t)P + kThreads - 1) / kThreads), 4096u);
    hipLaunchKernelGGL(refit_prims, dim3(blocks), dim3(kThreads), 0, 0, d_sph_in, ns, d_tri_in, nt, (rtgpu_tri *)d_tris,
                       (rtgpu_sph *)d_sphs, (uint32_t *)d_sph_meta, d_bad);
    RF_TRY(hipGetLastError());
  }
  // level passes until the work list is empty; its length is read back every 8
  // passes.  A valid tree needs at most RTGPU_MAX_TREE_DEPTH + 1 of them.
  uint32_t left = nn, pass = 0;
  const unsigned fit_blocks = std::max(1u, std::min<unsigned>((unsigned)(((size_t)nn + kFitThreads - 1) / kFitThreads), 2048u));
  while (left != 0u) {
    if (pass + 8 >= (uint32_t)kCounts) {
      err = "refit: the tree did not close in the passes its depth bound allows (corrupt child references?)";
      (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
      return 3;
    }
    for (int k = 0; k < 8; k++) {
      pass++;
      const uint32_t *lin = (pass & 1u) ? d_list_b : d_list_a;  // pass 1 reads no list
      uint32_t *lout = (pass & 1u) ? d_list_a : d_list_b;
      hipLaunchKernelGGL(refit_pass, dim3(fit_blocks), dim3(kFitThreads), 0, 0, nn, pass, (rtgpu_node *)d_nodes,
                         (const rtgpu_tri *)d_tris, (const uint32_t *)d_sph_meta, d_sph_in, ns, d_tri_in, d_done, lin,
                         lout, d_count);
    }
    RF_TRY(hipGetLastError());
    RF_TRY(hipMemcpy(&left, d_count + pass, sizeof(left), hipMemcpyDeviceToHost));
  }
  RF_TRY(hipEventRecord(ev1, 0));
  RF_TRY(hipEventSynchronize(ev1));
  float ms = 0.f;
  RF_TRY(hipEventElapsedTime(&ms, ev0, ev1));
  if (gpu_ms) *gpu_ms = ms;
  uint32_t bad = 0;
  RF_TRY(hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost));
  (void)hipEventDestroy(ev0);
  (void)hipEventDestroy(ev1);
  if (bad) { err = "refit: a leaf record names a primitive id outside the scene"; return 3; }
  return 0;
}

}  // namespace rtgpu
