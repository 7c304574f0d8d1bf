// lbvh_build.hip — SURVEY.md §8f row 3: the BVH of row a2 built ON the GPU, so
// that scenes of C5's size do not wait for a host build (10.5 M triangles: 1.4 s
// of host SAH on 16 cores).  No reference counterpart exists (the reference has
// no BVH).
//
// Algorithm: T. Karras, "Maximizing Parallelism in the Construction of BVHs,
// Octrees, and k-d Trees" (HPG 2012): 30-bit Morton codes of the primitive
// centroids made unique by the primitive index (64-bit keys), one radix sort
// (rocPRIM: a plain library sort), then every internal node finds its key range
// and split from the longest common prefixes of neighbouring keys - all nodes in
// parallel, no recursion.  Boxes are fitted bottom-up in LEVEL PASSES (one launch
// per tree level, a node is fitted in the first pass after both children): kernel
// boundaries give the cross-XCD visibility that an in-kernel "second thread to
// arrive" scheme would need agent-scope fences per node for
// (MI355X_MICROARCH.md, "Inter-workgroup visibility").  The result is written
// straight into the traversal kernel's format (include/rtgpu_format.h), one
// primitive per leaf, nodes with large subtrees numbered first so the kernel's
// LDS node cache holds the top of the tree.
//
// The CPU restatement lives under oracle/ (test infrastructure); tests compare the two
// byte for byte.  Everything is integer arithmetic or exact float min/max except
// the quantisation (c - cmin) * (1024 / ext): separately rounded IEEE f32
// operations on both sides (this file is compiled with -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "dev_pool.h"
#include "rtgpu_internal.h"

namespace {

constexpr int kThreads = 256;

struct Box6 { float lo[3], hi[3]; };

__device__ __forceinline__ uint32_t enc_f32(float f) {  // order-preserving float -> uint
  uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(uint32_t e) {
  return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e);
}

__device__ __forceinline__ uint32_t expand10(uint32_t v) {
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// (1) padded box and centroid of every primitive; centroid bounds by a register /
// wave / block reduction and then six atomics per BLOCK on order-preserving
// encodings (min/max are exact, so the order does not matter).  Grid-stride with a
// bounded grid: one atomic per wave of a 10 M primitive scene is 10^6 atomics on
// six words, and a single word takes about 88 per microsecond chip-wide
// (MI355X_MICROARCH.md "dequeue") - that alone was 11 ms.
__global__ __launch_bounds__(kThreads) void prim_boxes(const rtgpu_sphere *sph, uint32_t ns,
                                                        const rtgpu_triangle *tri, uint32_t nt, Box6 *box,
                                                        float *cent, uint32_t *cbounds /* min xyz, max xyz */) {
  const uint32_t P = ns + nt;
  uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < P; i += gridDim.x * kThreads) {
    Box6 b;
    float c[3];
    if (i < ns) {
      const rtgpu_sphere s = sph[i];
      for (int k = 0; k < 3; k++) {
        b.lo[k] = s.center[k] - s.radius;
        b.hi[k] = s.center[k] + s.radius;
        c[k] = s.center[k];
      }
    } else {
      const rtgpu_triangle t = tri[i - ns];
      for (int k = 0; k < 3; k++) {
        b.lo[k] = fminf(t.v0[k], fminf(t.v1[k], t.v2[k]));
        b.hi[k] = fmaxf(t.v0[k], fmaxf(t.v1[k], t.v2[k]));
        c[k] = (b.lo[k] + b.hi[k]) * 0.5f;
      }
    }
    for (int k = 0; k < 3; k++) {  // pad_box of bvh_build.cpp
      float m = fmaxf(fmaxf(fabsf(b.lo[k]), fabsf(b.hi[k])), 1.0f);
      float e = m * (1.0f / 262144.0f);
      b.lo[k] -= e;
      b.hi[k] += e;
      lo[k] = min(lo[k], enc_f32(c[k]));
      hi[k] = max(hi[k], enc_f32(c[k]));
    }
    box[i] = b;
    cent[3 * (size_t)i + 0] = c[0];
    cent[3 * (size_t)i + 1] = c[1];
    cent[3 * (size_t)i + 2] = c[2];
  }
  __shared__ uint32_t s_lo[3][kThreads / 64], s_hi[3][kThreads / 64];
  for (int k = 0; k < 3; k++) {
    uint32_t l = lo[k], h = hi[k];
    for (int off = 32; off > 0; off >>= 1) {
      l = min(l, (uint32_t)__shfl_xor((int)l, off, 64));
      h = max(h, (uint32_t)__shfl_xor((int)h, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { s_lo[k][threadIdx.x >> 6] = l; s_hi[k][threadIdx.x >> 6] = h; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    uint32_t l = 0xFFFFFFFFu, h = 0u;
    for (int w = 0; w < kThreads / 64; w++) { l = min(l, s_lo[threadIdx.x][w]); h = max(h, s_hi[threadIdx.x][w]); }
    atomicMin(&cbounds[threadIdx.x], l);
    atomicMax(&cbounds[3 + threadIdx.x], h);
  }
}

// (2) key = morton30(centroid) << 32 | primitive index
__global__ __launch_bounds__(kThreads) void morton_keys(const float *cent, uint32_t P, const uint32_t *cbounds,
                                                         unsigned long long *keys) {
  uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= P) return;
  uint32_t q[3];
  for (int k = 0; k < 3; k++) {
    float cmin = dec_f32(cbounds[k]), cmax = dec_f32(cbounds[3 + k]);
    float ext = cmax - cmin;
    float scale = ext > 0.0f ? 1024.0f / ext : 0.0f;
    float f = (cent[3 * (size_t)i + k] - cmin) * scale;
    int v = (int)f;
    q[k] = (uint32_t)(v < 0 ? 0 : v > 1023 ? 1023 : v);
  }
  uint32_t code = (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
  keys[i] = ((unsigned long long)code << 32) | (unsigned long long)i;
}

// (3a) 1 where sorted position k holds a sphere (primitive index < ns)
__global__ __launch_bounds__(kThreads) void sphere_flags(const unsigned long long *keys, uint32_t P, uint32_t ns,
                                                          uint32_t *flag) {
  uint32_t k = blockIdx.x * kThreads + threadIdx.x;
  if (k < P) flag[k] = (uint32_t)keys[k] < ns ? 1u : 0u;
}

// (3b) leaf arrays in sorted order per type, and the leaf reference of every sorted position
__global__ __launch_bounds__(kThreads) void emit_leaves(const unsigned long long *keys, uint32_t P,
                                                         const uint32_t *sph_rank, const rtgpu_sphere *sph,
                                                         uint32_t ns, const rtgpu_triangle *tri, rtgpu_sph *sphs,
                                                         uint32_t *meta, rtgpu_tri *tris, uint32_t *leaf_ref) {
  uint32_t k = blockIdx.x * kThreads + threadIdx.x;
  if (k >= P) return;
  uint32_t p = (uint32_t)keys[k];
  uint32_t rs = sph_rank[k];  // spheres before position k
  if (p < ns) {
    const rtgpu_sphere s = sph[p];
    rtgpu_sph r;
    r.c[0] = s.center[0]; r.c[1] = s.center[1]; r.c[2] = s.center[2]; r.r = s.radius;
    sphs[rs] = r;
    meta[2 * (size_t)rs] = p;
    meta[2 * (size_t)rs + 1] = s.material;
    leaf_ref[k] = RTGPU_REF_LEAF | RTGPU_REF_SPHERE | rs;
  } else {
    uint32_t rt = k - rs;
    const rtgpu_triangle t = tri[p - ns];
    rtgpu_tri r;
    for (int c = 0; c < 3; c++) {
      r.v0[c] = t.v0[c];
      r.e1[c] = t.v1[c] - t.v0[c];
      r.e2[c] = t.v2[c] - t.v0[c];
    }
    r.prim_id = p; r.material = t.material; r._pad = 0;
    tris[rt] = r;
    leaf_ref[k] = RTGPU_REF_LEAF | rt;
  }
}

__device__ __forceinline__ int delta(const unsigned long long *keys, int n, int a, int b) {
  if (b < 0 || b >= n) return -1;
  return __clzll((long long)(keys[a] ^ keys[b]));  // keys are unique: the xor is never 0
}

// (4) Karras' hierarchy: children of internal node i; code >= 0 = internal node, < 0 = ~(sorted position)
__global__ __launch_bounds__(kThreads) void karras(const unsigned long long *keys, int P, int *c0, int *c1) {
  int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= P - 1) return;
  int d = delta(keys, P, i, i + 1) > delta(keys, P, i, i - 1) ? 1 : -1;
  int dmin = delta(keys, P, i, i - d);
  int lmax = 2;
  while (delta(keys, P, i, i + lmax * d) > dmin) lmax *= 2;  // ends: the index leaves [0, P) at the latest
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (delta(keys, P, i, i + (l + t) * d) > dmin) l += t;
  int j = i + l * d;
  int dnode = delta(keys, P, i, j);
  int s = 0;
  for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
    if (delta(keys, P, i, i + (s + t) * d) > dnode) s += t;
    if (t == 1) break;
  }
  int gamma = i + s * d + (d < 0 ? -1 : 0);
  int lo = i < j ? i : j, hi = i < j ? j : i;
  c0[i] = gamma == lo ? ~gamma : gamma;
  c1[i] = gamma + 1 == hi ? ~(gamma + 1) : gamma + 1;
}

// (5) one level pass of the bottom-up fit over a WORK LIST of the nodes not fitted
// yet (pass 1: every node).  done[i] = number of the pass that fitted node i (0 =
// not yet).  A child counts as ready only if an EARLIER launch fitted it, so
// everything read here was written before this kernel started.  Nodes that stay
// unfitted are appended to the next pass's list: the block counts its survivors
// (ballots + LDS) and takes its slice with ONE atomic, so the list head is not hit
// once per wave.  counts[pass] = length of the list this pass leaves behind.
constexpr int kFitThreads = 1024;
// kind[i]: 1 = only spheres below node i, 2 = only triangles, 0 = both; lo_pos[i]: lowest
// sorted position below it.  A node over triangles only with size <= RTGPU_LBVH_MAX_LEAF
// is MERGED: it becomes one leaf in its parent's child slot and counts as height 0.
// (Spheres are tested one per leaf step: merged sphere leaves measured 6 % slower on S0.)
__device__ __forceinline__ bool merged(uint32_t size, uint32_t kind) { return kind == 2u && size <= RTGPU_LBVH_MAX_LEAF; }

__global__ __launch_bounds__(kFitThreads) void fit_pass(int n_inner, uint32_t pass, const int *c0, const int *c1,
                                                         const unsigned long long *keys, uint32_t ns, const Box6 *pbox,
                                                         Box6 *nbox, uint32_t *size, uint32_t *height, uint32_t *kind,
                                                         uint32_t *lo_pos, uint32_t *done, const uint32_t *list_in,
                                                         uint32_t *list_out, uint32_t *counts) {
  __shared__ uint32_t s_wave[kFitThreads / 64];
  __shared__ uint32_t s_base;
  const uint32_t n_in = pass == 1u ? (uint32_t)n_inner : counts[pass - 1u];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (uint32_t base = blockIdx.x * kFitThreads; base < n_in; base += gridDim.x * kFitThreads) {  // block-uniform
    const uint32_t idx = base + threadIdx.x;
    bool survive = false;
    int i = -1;
    if (idx < n_in) {
      i = pass == 1u ? (int)idx : (int)list_in[idx];
      int a = c0[i], b = c1[i];
      uint32_t da = a < 0 ? 1u : done[a], db = b < 0 ? 1u : done[b];
      bool ra = a < 0 || (da != 0u && da < pass), rb = b < 0 || (db != 0u && db < pass);
      if (ra && rb) {
        Box6 ba = a < 0 ? pbox[(uint32_t)keys[~a]] : nbox[a];
        Box6 bb = b < 0 ? pbox[(uint32_t)keys[~b]] : nbox[b];
        Box6 o;
        for (int k = 0; k < 3; k++) {
          o.lo[k] = fminf(ba.lo[k], bb.lo[k]);
          o.hi[k] = fmaxf(ba.hi[k], bb.hi[k]);
        }
        nbox[i] = o;
        uint32_t sa = a < 0 ? 1u : size[a], sb = b < 0 ? 1u : size[b];
        uint32_t ha = a < 0 ? 0u : height[a], hb = b < 0 ? 0u : height[b];
        uint32_t ka = a < 0 ? ((uint32_t)keys[~a] < ns ? 1u : 2u) : kind[a];
        uint32_t kb = b < 0 ? ((uint32_t)keys[~b] < ns ? 1u : 2u) : kind[b];
        uint32_t kd = ka == kb ? ka : 0u;
        size[i] = sa + sb;
        kind[i] = kd;
        lo_pos[i] = a < 0 ? (uint32_t)~a : lo_pos[a];  // child 0 holds the lower key range
        // the root stays a node even when everything below it merges (one leaf + an EMPTY sibling)
        height[i] = merged(sa + sb, kd) ? (i == 0 ? 1u : 0u) : 1u + (ha > hb ? ha : hb);
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
    if (survive) list_out[s_base + s_wave[wave] + before] = (uint32_t)i;
    __syncthreads();  // s_wave / s_base are rewritten by the next round
  }
}

// (6) nodes that are emitted (the root, and every node that is not merged into a leaf), in two
// groups: flag[i] = 1 for the top of the tree (subtree of >= T primitives), rest[i] = 1 for the others
__global__ __launch_bounds__(kThreads) void top_flags(const uint32_t *size, const uint32_t *kind, int n_inner, uint32_t T,
                                                       uint32_t *flag, uint32_t *rest) {
  int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n_inner) return;
  const bool keep = i == 0 || !merged(size[i], kind[i]);
  const bool top = keep && size[i] >= T;
  flag[i] = top ? 1u : 0u;
  rest[i] = keep && !top ? 1u : 0u;
}

__device__ __forceinline__ uint32_t new_index(int i, const uint32_t *top_rank, const uint32_t *flag,
                                              const uint32_t *rest_rank, uint32_t n_top) {
  return flag[i] ? top_rank[i] : n_top + rest_rank[i];
}

// (7) traversal-format nodes at their final indices
__global__ __launch_bounds__(kThreads) void emit_nodes(int n_inner, const int *c0, const int *c1,
                                                        const unsigned long long *keys, const Box6 *pbox,
                                                        const Box6 *nbox, const uint32_t *leaf_ref,
                                                        const uint32_t *size, const uint32_t *kind, const uint32_t *lo_pos,
                                                        const uint32_t *top_rank, const uint32_t *flag,
                                                        const uint32_t *rest_rank, const uint32_t *rest,
                                                        const uint32_t *n_top_ptr, rtgpu_node *nodes) {
  int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n_inner || !(flag[i] || rest[i])) return;  // merged into a leaf of an ancestor
  uint32_t n_top = *n_top_ptr;
  rtgpu_node nd;
  nd._pad[0] = nd._pad[1] = 0;
  if (merged(size[i], kind[i])) {  // i == 0: everything under the root is one leaf
    const Box6 bx = nbox[i];
    for (int k = 0; k < 3; k++) {
      nd.lo0[k] = bx.lo[k]; nd.hi0[k] = bx.hi[k];
      nd.lo1[k] = INFINITY; nd.hi1[k] = INFINITY;
    }
    nd.child0 = leaf_ref[lo_pos[i]] | ((size[i] - 1u) << 27);
    nd.child1 = RTGPU_REF_EMPTY;
    nodes[0] = nd;
    return;
  }
  int a = c0[i], b = c1[i];
  Box6 ba = a < 0 ? pbox[(uint32_t)keys[~a]] : nbox[a];
  Box6 bb = b < 0 ? pbox[(uint32_t)keys[~b]] : nbox[b];
  for (int k = 0; k < 3; k++) {
    nd.lo0[k] = ba.lo[k]; nd.hi0[k] = ba.hi[k];
    nd.lo1[k] = bb.lo[k]; nd.hi1[k] = bb.hi[k];
  }
  // a merged child: the leaf of its first primitive widened to the subtree's count (its
  // primitives are adjacent in the leaf-ordered array of their type)
  auto ref_of = [&](int c) -> uint32_t {
    if (c < 0) return leaf_ref[~c];
    if (merged(size[c], kind[c])) return leaf_ref[lo_pos[c]] | ((size[c] - 1u) << 27);
    return new_index(c, top_rank, flag, rest_rank, n_top);
  };
  nd.child0 = ref_of(a);
  nd.child1 = ref_of(b);
  nodes[new_index(i, top_rank, flag, rest_rank, n_top)] = nd;
}

// n_top = rank of the last node + its flag (the exclusive scan does not hold the total)
__global__ void total_of_scan(const uint32_t *rank, const uint32_t *flag, int n, uint32_t *out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *out = rank[n - 1] + flag[n - 1];
}

struct DevBufs {  // everything the build allocates besides its outputs; given back on every exit
  rtgpu::DevPool *pool = nullptr;
  std::vector<void *> ptrs;
  ~DevBufs() { for (void *p : ptrs) rtgpu::dev_put(pool, p); }
  template <typename T>
  hipError_t alloc(T **p, size_t n) {
    hipError_t e = rtgpu::dev_get(pool, (void **)p, (n ? n : 1) * sizeof(T));
    if (e == hipSuccess) ptrs.push_back(*p);
    return e;
  }
};

constexpr int kCounts = 72;     // counts[1..64]: work-list length after each fit pass; [kNTopSlot]: top nodes
constexpr int kNTopSlot = 70;
constexpr int kNRestSlot = 71;  // counts[kNRestSlot]: emitted nodes outside the top

unsigned blocks_for(size_t n) { return (unsigned)((n + kThreads - 1) / kThreads); }

}  // namespace

namespace rtgpu {

#define LB_TRY(call)                                                                  \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) {                                                           \
      err = std::string("lbvh: " #call ": ") + hipGetErrorString(e_);                 \
      if (ev0) (void)hipEventDestroy(ev0);                                            \
      if (ev1) (void)hipEventDestroy(ev1);                                            \
      free_lbvh(out, pool);                                                                 \
      return 2;                                                                       \
    }                                                                                 \
  } while (0)

void free_lbvh(LbvhDevice *out, DevPool *pool) {
  if (!out) return;
  dev_put(pool, out->d_nodes); dev_put(pool, out->d_tris); dev_put(pool, out->d_sphs); dev_put(pool, out->d_sph_meta);
  std::memset(out, 0, sizeof(*out));
}

// The current device must be the context's.  The scene must have passed
// validate_scene.  Output arrays carry the zeroed tail records flatten_scene adds.
int lbvh_build_device(const rtgpu_scene_desc *scene, LbvhDevice *out, std::string &err) {
  return lbvh_build_device_ptrs(scene, nullptr, nullptr, nullptr, out, err);
}

// The same with the primitive arrays already in device memory when d_spheres_in /
// d_triangles_in are given (scene->spheres / ->triangles are then not read; the counts
// are).  The arrays must have passed check_prims_device.
// `pool`: where outputs and temporaries come from and go back to (NULL: hipMalloc / hipFree).
int lbvh_build_device_ptrs(const rtgpu_scene_desc *scene, const void *d_spheres_in, const void *d_triangles_in,
                           DevPool *pool, LbvhDevice *out, std::string &err) {
  std::memset(out, 0, sizeof(*out));
  const bool on_device = d_spheres_in || d_triangles_in;
  const uint64_t ns64 = scene->n_spheres, nt64 = scene->n_triangles, P64 = ns64 + nt64;
  if (P64 >= (1ull << 28)) { err = "lbvh: too many primitives (limit 2^28)"; return 1; }
  const uint32_t ns = (uint32_t)ns64, nt = (uint32_t)nt64, P = (uint32_t)P64;
  const int n_inner = P > 1 ? (int)P - 1 : 1;
  if ((uint64_t)n_inner * sizeof(rtgpu_node) >= (1ull << 32) || (uint64_t)nt * sizeof(rtgpu_tri) >= (1ull << 32)) {
    err = "scene too large: node or triangle buffer would reach 4 GiB (the kernel uses 32-bit offsets)";
    return 1;
  }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  DevBufs tmp;
  tmp.pool = pool;
  LB_TRY(dev_get(pool, &out->d_nodes, (size_t)n_inner * sizeof(rtgpu_node)));
  LB_TRY(dev_get(pool, &out->d_tris, ((size_t)nt + 8) * sizeof(rtgpu_tri)));
  LB_TRY(dev_get(pool, &out->d_sphs, ((size_t)ns + 8) * sizeof(rtgpu_sph)));
  LB_TRY(dev_get(pool, &out->d_sph_meta, ((size_t)ns * 2 + 16) * sizeof(uint32_t)));
  LB_TRY(hipMemset(out->d_tris, 0, ((size_t)nt + 8) * sizeof(rtgpu_tri)));
  LB_TRY(hipMemset(out->d_sphs, 0, ((size_t)ns + 8) * sizeof(rtgpu_sph)));
  LB_TRY(hipMemset(out->d_sph_meta, 0, ((size_t)ns * 2 + 16) * sizeof(uint32_t)));
  out->n_nodes = (uint64_t)n_inner;

  if (P <= 1) {  // same degenerate roots as the host builder
    rtgpu_node root;
    std::memset(&root, 0, sizeof(root));
    for (int k = 0; k < 3; k++) { root.lo0[k] = root.lo1[k] = INFINITY; root.hi0[k] = root.hi1[k] = INFINITY; }
    root.child0 = root.child1 = RTGPU_REF_EMPTY;
    if (P == 1) {
      float lo[3], hi[3];
      rtgpu_sphere s1;
      rtgpu_triangle t1;
      if (on_device && ns) LB_TRY(hipMemcpy(&s1, d_spheres_in, sizeof(s1), hipMemcpyDeviceToHost));
      if (on_device && nt) LB_TRY(hipMemcpy(&t1, d_triangles_in, sizeof(t1), hipMemcpyDeviceToHost));
      if (ns) {
        const rtgpu_sphere &s = on_device ? s1 : scene->spheres[0];
        for (int k = 0; k < 3; k++) { lo[k] = s.center[k] - s.radius; hi[k] = s.center[k] + s.radius; }
        rtgpu_sph r = {{s.center[0], s.center[1], s.center[2]}, s.radius};
        uint32_t meta[2] = {0u, s.material};
        LB_TRY(hipMemcpy(out->d_sphs, &r, sizeof(r), hipMemcpyHostToDevice));
        LB_TRY(hipMemcpy(out->d_sph_meta, meta, sizeof(meta), hipMemcpyHostToDevice));
        root.child0 = RTGPU_REF_LEAF | RTGPU_REF_SPHERE;
      } else {
        const rtgpu_triangle &t = on_device ? t1 : scene->triangles[0];
        rtgpu_tri r;
        std::memset(&r, 0, sizeof(r));
        for (int k = 0; k < 3; k++) {
          lo[k] = std::fmin(t.v0[k], std::fmin(t.v1[k], t.v2[k]));
          hi[k] = std::fmax(t.v0[k], std::fmax(t.v1[k], t.v2[k]));
          r.v0[k] = t.v0[k]; r.e1[k] = t.v1[k] - t.v0[k]; r.e2[k] = t.v2[k] - t.v0[k];
        }
        r.prim_id = 0; r.material = t.material;
        LB_TRY(hipMemcpy(out->d_tris, &r, sizeof(r), hipMemcpyHostToDevice));
        root.child0 = RTGPU_REF_LEAF;
      }
      for (int k = 0; k < 3; k++) {
        float m = std::fmax(std::fmax(std::fabs(lo[k]), std::fabs(hi[k])), 1.0f);
        float e = m * (1.0f / 262144.0f);
        root.lo0[k] = lo[k] - e; root.hi0[k] = hi[k] + e;
      }
      out->max_depth = 1;
    }
    LB_TRY(hipMemcpy(out->d_nodes, &root, sizeof(root), hipMemcpyHostToDevice));
    return 0;
  }

  rtgpu_sphere *d_sph_in = nullptr;
  rtgpu_triangle *d_tri_in = nullptr;
  Box6 *d_pbox = nullptr, *d_nbox = nullptr;
  float *d_cent = nullptr;
  uint32_t *d_cb = nullptr, *d_flag = nullptr, *d_rank = nullptr, *d_leaf_ref = nullptr, *d_size = nullptr,
           *d_height = nullptr, *d_done = nullptr, *d_count = nullptr, *d_list_a = nullptr, *d_list_b = nullptr,
           *d_kind = nullptr, *d_lo_pos = nullptr, *d_rest = nullptr, *d_rest_rank = nullptr;
  unsigned long long *d_keys_a = nullptr, *d_keys_b = nullptr;
  int *d_c0 = nullptr, *d_c1 = nullptr;
  if (on_device) {  // read in place
    d_sph_in = (rtgpu_sphere *)d_spheres_in;
    d_tri_in = (rtgpu_triangle *)d_triangles_in;
  } else {
    LB_TRY(tmp.alloc(&d_sph_in, ns));
    LB_TRY(tmp.alloc(&d_tri_in, nt));
  }
  LB_TRY(tmp.alloc(&d_pbox, P));
  LB_TRY(tmp.alloc(&d_nbox, (size_t)n_inner));
  LB_TRY(tmp.alloc(&d_cent, (size_t)P * 3));
  LB_TRY(tmp.alloc(&d_cb, 8));
  LB_TRY(tmp.alloc(&d_flag, P));
  LB_TRY(tmp.alloc(&d_rank, P));
  LB_TRY(tmp.alloc(&d_leaf_ref, P));
  LB_TRY(tmp.alloc(&d_size, (size_t)n_inner));
  LB_TRY(tmp.alloc(&d_height, (size_t)n_inner));
  LB_TRY(tmp.alloc(&d_kind, (size_t)n_inner));
  LB_TRY(tmp.alloc(&d_lo_pos, (size_t)n_inner));
  LB_TRY(tmp.alloc(&d_rest, (size_t)n_inner));
  LB_TRY(tmp.alloc(&d_rest_rank, (size_t)n_inner));
  LB_TRY(tmp.alloc(&d_done, (size_t)n_inner));
  LB_TRY(tmp.alloc(&d_count, kCounts));
  LB_TRY(tmp.alloc(&d_list_a, (size_t)n_inner));
  LB_TRY(tmp.alloc(&d_list_b, (size_t)n_inner));
  LB_TRY(tmp.alloc(&d_keys_a, P));
  LB_TRY(tmp.alloc(&d_keys_b, P));
  LB_TRY(tmp.alloc(&d_c0, (size_t)n_inner));
  LB_TRY(tmp.alloc(&d_c1, (size_t)n_inner));
  size_t sort_bytes = 0, scan_bytes = 0;
  LB_TRY(rocprim::radix_sort_keys(nullptr, sort_bytes, d_keys_a, d_keys_b, (size_t)P, 0u, 64u, (hipStream_t)0));
  LB_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, d_flag, d_rank, 0u, (size_t)P, rocprim::plus<uint32_t>(),
                                 (hipStream_t)0));
  char *d_tmp = nullptr;
  size_t tmp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  LB_TRY(tmp.alloc(&d_tmp, tmp_bytes));

  if (ns && !on_device) LB_TRY(hipMemcpy(d_sph_in, scene->spheres, (size_t)ns * sizeof(rtgpu_sphere), hipMemcpyHostToDevice));
  if (nt && !on_device) LB_TRY(hipMemcpy(d_tri_in, scene->triangles, (size_t)nt * sizeof(rtgpu_triangle), hipMemcpyHostToDevice));
  const uint32_t cb_init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
  LB_TRY(hipMemcpy(d_cb, cb_init, sizeof(cb_init), hipMemcpyHostToDevice));
  LB_TRY(hipMemset(d_done, 0, (size_t)n_inner * sizeof(uint32_t)));
  LB_TRY(hipMemset(d_count, 0, kCounts * sizeof(uint32_t)));

  LB_TRY(hipEventCreate(&ev0));
  LB_TRY(hipEventCreate(&ev1));
  LB_TRY(hipEventRecord(ev0, 0));
  hipLaunchKernelGGL(prim_boxes, dim3(std::min(blocks_for(P), 2048u)), dim3(kThreads), 0, 0, d_sph_in, ns, d_tri_in, nt, d_pbox,
                     d_cent, d_cb);
  hipLaunchKernelGGL(morton_keys, dim3(blocks_for(P)), dim3(kThreads), 0, 0, d_cent, P, d_cb, d_keys_a);
  LB_TRY(hipGetLastError());
  LB_TRY(rocprim::radix_sort_keys(d_tmp, sort_bytes, d_keys_a, d_keys_b, (size_t)P, 0u, 64u, (hipStream_t)0));
  const unsigned long long *keys = d_keys_b;
  hipLaunchKernelGGL(sphere_flags, dim3(blocks_for(P)), dim3(kThreads), 0, 0, keys, P, ns, d_flag);
  LB_TRY(hipGetLastError());
  LB_TRY(rocprim::exclusive_scan(d_tmp, scan_bytes, d_flag, d_rank, 0u, (size_t)P, rocprim::plus<uint32_t>(),
                                 (hipStream_t)0));
  hipLaunchKernelGGL(emit_leaves, dim3(blocks_for(P)), dim3(kThreads), 0, 0, keys, P, d_rank, d_sph_in, ns, d_tri_in,
                     (rtgpu_sph *)out->d_sphs, (uint32_t *)out->d_sph_meta, (rtgpu_tri *)out->d_tris, d_leaf_ref);
  hipLaunchKernelGGL(karras, dim3(blocks_for((size_t)n_inner)), dim3(kThreads), 0, 0, keys, (int)P, d_c0, d_c1);
  LB_TRY(hipGetLastError());

  // level passes until the work list is empty; its length is read back every 8
  // passes.  A tree over these keys is at most 30 + 28 levels deep: 64 is a hard stop.
  {
    uint32_t left = (uint32_t)n_inner, pass = 0;
    const unsigned fit_blocks = std::min<unsigned>((unsigned)(((size_t)n_inner + kFitThreads - 1) / kFitThreads), 2048u);
    while (left != 0u) {
      if (pass >= 64u) {
        err = "internal: lbvh fit did not finish in 64 level passes";
        (void)hipEventDestroy(ev0);
        (void)hipEventDestroy(ev1);
        free_lbvh(out, pool);
        return 3;
      }
      for (int k = 0; k < 8; k++) {
        pass++;
        const uint32_t *lin = (pass & 1u) ? d_list_b : d_list_a;   // pass 1 reads no list
        uint32_t *lout = (pass & 1u) ? d_list_a : d_list_b;
        hipLaunchKernelGGL(fit_pass, dim3(fit_blocks), dim3(kFitThreads), 0, 0, n_inner, pass, d_c0, d_c1, keys, ns, d_pbox,
                           d_nbox, d_size, d_height, d_kind, d_lo_pos, d_done, lin, lout, d_count);
      }
      LB_TRY(hipGetLastError());
      LB_TRY(hipMemcpy(&left, d_count + pass, sizeof(left), hipMemcpyDeviceToHost));
    }
  }

  uint32_t T = (uint32_t)((2ull * P + 4095ull) / 4096ull);
  if (T < 1u) T = 1u;
  hipLaunchKernelGGL(top_flags, dim3(blocks_for((size_t)n_inner)), dim3(kThreads), 0, 0, d_size, d_kind, n_inner, T, d_flag,
                     d_rest);
  LB_TRY(hipGetLastError());
  LB_TRY(rocprim::exclusive_scan(d_tmp, scan_bytes, d_flag, d_rank, 0u, (size_t)n_inner, rocprim::plus<uint32_t>(),
                                 (hipStream_t)0));
  LB_TRY(rocprim::exclusive_scan(d_tmp, scan_bytes, d_rest, d_rest_rank, 0u, (size_t)n_inner, rocprim::plus<uint32_t>(),
                                 (hipStream_t)0));
  hipLaunchKernelGGL(total_of_scan, dim3(1), dim3(64), 0, 0, d_rank, d_flag, n_inner, d_count + kNTopSlot);
  hipLaunchKernelGGL(total_of_scan, dim3(1), dim3(64), 0, 0, d_rest_rank, d_rest, n_inner, d_count + kNRestSlot);
  hipLaunchKernelGGL(emit_nodes, dim3(blocks_for((size_t)n_inner)), dim3(kThreads), 0, 0, n_inner, d_c0, d_c1, keys,
                     d_pbox, d_nbox, d_leaf_ref, d_size, d_kind, d_lo_pos, d_rank, d_flag, d_rest_rank, d_rest,
                     d_count + kNTopSlot, (rtgpu_node *)out->d_nodes);
  LB_TRY(hipGetLastError());
  LB_TRY(hipEventRecord(ev1, 0));
  LB_TRY(hipEventSynchronize(ev1));
  float ms = 0.f;
  LB_TRY(hipEventElapsedTime(&ms, ev0, ev1));
  out->gpu_ms = ms;
  uint32_t h0 = 0, kept[2] = {0u, 0u};
  LB_TRY(hipMemcpy(&h0, d_height, sizeof(h0), hipMemcpyDeviceToHost));
  LB_TRY(hipMemcpy(kept, d_count + kNTopSlot, sizeof(kept), hipMemcpyDeviceToHost));
  out->max_depth = h0;
  out->n_nodes = (uint64_t)kept[0] + kept[1];  // nodes merged into leaves are not emitted
  (void)hipEventDestroy(ev0);
  (void)hipEventDestroy(ev1);
  if (h0 > RTGPU_MAX_TREE_DEPTH) { err = "internal: LBVH deeper than RTGPU_MAX_TREE_DEPTH"; free_lbvh(out, pool); return 3; }
  return 0;
}

}  // namespace rtgpu
