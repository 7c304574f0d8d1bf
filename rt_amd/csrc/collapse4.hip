// collapse4.hip — the four-wide packed nodes (rtgpu_node4, include/rtgpu_format.h) made
// on the GPU from the 64-byte BVH2 nodes, for trees that are built there
// (lbvh_build.hip): nothing goes back to the host.  Same collapse rule and the same
// packed form as the host's bvh4_build.cpp (absorb the inner child of largest surface
// area while there are fewer than four children; child order = the BVH2's left to
// right order; exact planes moved to the conservative side to carry the reference
// bytes); what differs is the NUMBERING of the nodes: breadth first here (one level of
// four-wide nodes per pass; the inner children of a level are numbered in thread order,
// slot order within a thread, from an exclusive scan of the per-thread counts - so the
// array is the same on every run, context and rank), depth first there.  Node numbers DO
// reach the traversal: the child reference bytes ride in the low bytes of four planes per
// child (embed_byte), which moves those planes by up to 255 floats to the conservative side
// and with them near-equal entry distances, i.e. the visit order and the counters.  A
// run-dependent numbering (one atomic per wave, as this file first did it) therefore made
// counters and equal-t hits differ between runs and ranks; the scan removes that.  No
// reference counterpart (the reference has no BVH).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include <string>

#include "dev_pool.h"
#include "pack48.h"
#include "rtgpu_internal.h"

namespace {

struct Pend { uint32_t node2, above; };  // a four-wide node still to be made: its BVH2 node, stack entries above it

struct Child {
  uint32_t lo[3], hi[3];  // float bit patterns
  uint32_t ref, src;
};

__device__ __forceinline__ void take(const rtgpu_node *nodes, uint32_t at, int which, Child &c) {
  const rtgpu_node &nd = nodes[at];
  const float *lo = which ? nd.lo1 : nd.lo0, *hi = which ? nd.hi1 : nd.hi0;
  for (int k = 0; k < 3; k++) { c.lo[k] = __float_as_uint(lo[k]); c.hi[k] = __float_as_uint(hi[k]); }
  c.ref = which ? nd.child1 : nd.child0;
  c.src = (at << 1) | (uint32_t)which;
}

__device__ __forceinline__ float half_area(const Child &c) {
  // the host's expression, without contraction (the build sets -ffp-contract=off)
  float dx = __uint_as_float(c.hi[0]) - __uint_as_float(c.lo[0]), dy = __uint_as_float(c.hi[1]) - __uint_as_float(c.lo[1]),
        dz = __uint_as_float(c.hi[2]) - __uint_as_float(c.lo[2]);
  return dx * dy + dy * dz + dz * dx;
}

// The collapse of one pending four-wide node: up to four children of BVH2 node p.node2 (absorb
// the inner child of largest surface area while there are fewer than four).  Returns the child
// count, or -1 for a malformed tree.
__device__ __forceinline__ int gather4(const rtgpu_node *nodes, uint32_t n_nodes, uint32_t node2, Child ch[4]) {
  if (node2 >= n_nodes) return -1;
  take(nodes, node2, 0, ch[0]);
  take(nodes, node2, 1, ch[1]);
  int n = 2;
  while (n < 4) {
    int best = -1;
    float best_area = -1.0f;
    for (int k = 0; k < n; k++) {
      if (ch[k].ref & RTGPU_REF_LEAF) continue;  // leaves (and EMPTY) stay
      float a = half_area(ch[k]);
      if (a > best_area) { best_area = a; best = k; }
    }
    if (best < 0) break;
    const uint32_t r = ch[best].ref;
    if (r >= n_nodes) return -1;
    for (int k = n; k > best + 1; k--) ch[k] = ch[k - 1];  // open a slot after `best`
    take(nodes, r, 0, ch[best]);
    take(nodes, r, 1, ch[best + 1]);
    n++;
  }
  return n;
}

// pass 1 of a level: inner children per pending node (what the scan numbers them from)
__global__ __launch_bounds__(256) void count_level(const rtgpu_node *nodes, uint32_t n_nodes, uint32_t first,
                                                    uint32_t count, const Pend *pend, uint32_t *n_inner) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= count) return;
  Child ch[4];
  const int n = gather4(nodes, n_nodes, pend[first + t].node2, ch);
  uint32_t c = 0;
  for (int k = 0; k < n; k++) c += (ch[k].ref & RTGPU_REF_LEAF) ? 0u : 1u;
  n_inner[t] = c;
}

// counters: [0] four-wide nodes allocated so far, [1] largest stack bound, [2] error flag
// pass 2: make the nodes; inner child j (slot order) of thread t becomes node end + offs[t] + j
__global__ __launch_bounds__(256) void collapse_level(const rtgpu_node *nodes, uint32_t n_nodes, uint32_t first,
                                                      uint32_t count, uint32_t cap, Pend *pend, rtgpu_node4 *out,
                                                      uint32_t *src, uint32_t *counters, const uint32_t *offs) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  const bool live = t < count;
  Child ch[4];
  int n = 0;
  uint32_t below = 0;
  bool bad = false;
  if (live) {
    const Pend p = pend[first + t];
    n = gather4(nodes, n_nodes, p.node2, ch);
    if (n < 0) { bad = true; n = 0; }
    else below = p.above + (uint32_t)(n - 1);
  }
  // numbers for the inner children: thread order, slot order within the thread
  uint32_t my_ref[4];
  uint32_t next_id = live ? first + count + offs[t] : 0u;
  for (int k = 0; k < 4; k++) {
    my_ref[k] = k < n ? ch[k].ref : RTGPU_REF_EMPTY;
    if (live && !bad && k < n && !(ch[k].ref & RTGPU_REF_LEAF)) {
      if (next_id < cap) {
        pend[next_id] = Pend{ch[k].ref, below};
        my_ref[k] = next_id;
      } else {
        bad = true;  // more four-wide nodes than BVH2 nodes: a cycle in the child references
      }
      next_id++;
    }
  }
  if (live && t == count - 1u) counters[0] = next_id;  // nodes allocated so far (the level's last thread knows)
  if (live && !bad) {
    rtgpu_node4 nd;
    uint32_t s4[4];
    for (int k = 0; k < 4; k++) {
      uint32_t w[6];
      if (k < n && ch[k].ref != RTGPU_REF_EMPTY) {
        for (int j = 0; j < 3; j++) { w[j] = ch[k].lo[j]; w[3 + j] = ch[k].hi[j]; }
        s4[k] = ch[k].src;
      } else {
        for (int j = 0; j < 6; j++) w[j] = RTGPU_NODE4_FAR;
        my_ref[k] = RTGPU_REF_EMPTY;
        s4[k] = 0xFFFFFFFFu;
      }
      for (int j = 0; j < 4; j++)  // lo.x lo.y lo.z down, hi.x up
        if (!rtgpu::embed_byte(w[j], (my_ref[k] >> (8 * j)) & 0xFFu, j < 3, &w[j])) bad = true;
      if ((w[4] & 0x7F800000u) == 0x7F800000u || (w[5] & 0x7F800000u) == 0x7F800000u) bad = true;
      for (int j = 0; j < 6; j++) nd.w[6 * k + j] = w[j];
    }
    out[first + t] = nd;
    for (int k = 0; k < 4; k++) src[4 * (size_t)(first + t) + k] = s4[k];
  }
  // largest stack bound: wave maximum, one atomic per wave
  uint32_t m = live ? below : 0u;
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t o = __shfl_xor(m, d);
    m = o > m ? o : m;
  }
  if (lane == 0 && m) atomicMax(&counters[1], m);
  if (bad) atomicOr(&counters[2], 1u);
}

}  // namespace

namespace rtgpu {

void free_nodes4_device(Nodes4Device *out, DevPool *pool) {
  if (!out) return;
  dev_put(pool, out->d_nodes4);
  dev_put(pool, out->d_src);
  *out = Nodes4Device{};
}

#define C4_TRY(call)                                                        \
  do {                                                                      \
    hipError_t e_ = (call);                                                 \
    if (e_ != hipSuccess) {                                                 \
      err = std::string("four-wide nodes: " #call ": ") + hipGetErrorString(e_); \
      dev_put(pool, d_pend); dev_put(pool, d_tmp4); dev_put(pool, d_tmps); dev_put(pool, d_cnt); \
      dev_put(pool, d_ninner); dev_put(pool, d_offs); dev_put(pool, d_scan);            \
      free_nodes4_device(out, pool);                                        \
      return 2;                                                             \
    }                                                                       \
  } while (0)

// The current device must be the context's.  Synchronous.  On success out->d_nodes4 and
// out->d_src are exact-size device arrays (hipMalloc) the caller owns.
int collapse_nodes4_device(const void *d_nodes, uint64_t n_nodes, DevPool *pool, Nodes4Device *out, std::string &err) {
  *out = Nodes4Device{};
  Pend *d_pend = nullptr;
  rtgpu_node4 *d_tmp4 = nullptr;
  uint32_t *d_tmps = nullptr, *d_cnt = nullptr, *d_ninner = nullptr, *d_offs = nullptr;
  void *d_scan = nullptr;  // rocPRIM's scan scratch, sized for the widest level there can be
  size_t scan_bytes = 0;
  const char *fail = "four-wide nodes: malformed tree, or a box plane that is not finite / too close to FLT_MAX";
  if (n_nodes == 0 || n_nodes >= (1ull << 31)) { err = fail; return 1; }
  // a four-wide node consumes at least one BVH2 node: n_nodes bounds their number
  const uint32_t cap = (uint32_t)n_nodes;
  C4_TRY(dev_get(pool, (void **)&d_pend, (size_t)cap * sizeof(Pend)));
  C4_TRY(dev_get(pool, (void **)&d_tmp4, (size_t)cap * sizeof(rtgpu_node4)));
  C4_TRY(dev_get(pool, (void **)&d_tmps, (size_t)cap * 16));
  C4_TRY(dev_get(pool, (void **)&d_cnt, 4 * sizeof(uint32_t)));
  C4_TRY(dev_get(pool, (void **)&d_ninner, (size_t)cap * sizeof(uint32_t)));
  C4_TRY(dev_get(pool, (void **)&d_offs, (size_t)cap * sizeof(uint32_t)));
  C4_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, d_ninner, d_offs, 0u, (size_t)cap, rocprim::plus<uint32_t>(), 0));
  C4_TRY(dev_get(pool, &d_scan, scan_bytes ? scan_bytes : 64));
  const Pend root = {0u, 0u};
  const uint32_t init[4] = {1u, 0u, 0u, 0u};  // node 0 = the root's four-wide node
  C4_TRY(hipMemcpy(d_pend, &root, sizeof(root), hipMemcpyHostToDevice));
  C4_TRY(hipMemcpy(d_cnt, init, sizeof(init), hipMemcpyHostToDevice));
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  {
    hipError_t e = hipEventCreate(&ev0);
    if (e == hipSuccess && (e = hipEventCreate(&ev1)) != hipSuccess) (void)hipEventDestroy(ev0);
    C4_TRY(e);
  }
  (void)hipEventRecord(ev0, 0);
  uint32_t first = 0, end = 1, cnt[4] = {1u, 0u, 0u, 0u};
  int rc = 0;
  for (uint32_t level = 0; first < end; level++) {
    if (level > RTGPU_MAX_TREE_DEPTH + 1u) { rc = 1; break; }  // deeper than any valid tree: a cycle
    const uint32_t count = end - first;
    hipLaunchKernelGGL(count_level, dim3((count + 255u) / 256u), dim3(256), 0, 0, (const rtgpu_node *)d_nodes,
                       (uint32_t)n_nodes, first, count, d_pend, d_ninner);
    hipError_t e = hipGetLastError();
    size_t sb = scan_bytes;
    if (e == hipSuccess)
      e = rocprim::exclusive_scan(d_scan, sb, d_ninner, d_offs, 0u, (size_t)count, rocprim::plus<uint32_t>(), 0);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(collapse_level, dim3((count + 255u) / 256u), dim3(256), 0, 0, (const rtgpu_node *)d_nodes,
                         (uint32_t)n_nodes, first, count, cap, d_pend, d_tmp4, d_tmps, d_cnt, (const uint32_t *)d_offs);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1); C4_TRY(e); }
    if (cnt[2] || cnt[0] > cap) { rc = 1; break; }
    first = end;
    end = cnt[0];
  }
  (void)hipEventRecord(ev1, 0);
  (void)hipEventSynchronize(ev1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, ev0, ev1);
  (void)hipEventDestroy(ev0);
  (void)hipEventDestroy(ev1);
  const uint64_t total = end;
  if (!rc && total * sizeof(rtgpu_node4) >= (1ull << 32)) {
    err = "four-wide nodes: buffer would reach 4 GiB";
    rc = 1;
  } else if (rc) {
    err = fail;
  }
  dev_put(pool, d_ninner); dev_put(pool, d_offs); dev_put(pool, d_scan);
  d_ninner = d_offs = nullptr;
  d_scan = nullptr;
  if (rc) {
    dev_put(pool, d_pend); dev_put(pool, d_tmp4); dev_put(pool, d_tmps); dev_put(pool, d_cnt);
    return rc;
  }
  // exact-size copies (the bound above is about twice what a tree needs)
  const size_t nb = (total * sizeof(rtgpu_node4) + 63) / 64 * 64, sb = (total * 16 + 63) / 64 * 64;
  C4_TRY(dev_get(pool, &out->d_nodes4, nb));
  C4_TRY(dev_get(pool, &out->d_src, sb));
  C4_TRY(hipMemset(out->d_nodes4, 0, nb));
  C4_TRY(hipMemset(out->d_src, 0, sb));
  C4_TRY(hipMemcpy(out->d_nodes4, d_tmp4, total * sizeof(rtgpu_node4), hipMemcpyDeviceToDevice));
  C4_TRY(hipMemcpy(out->d_src, d_tmps, total * 16, hipMemcpyDeviceToDevice));
  dev_put(pool, d_pend); dev_put(pool, d_tmp4); dev_put(pool, d_tmps); dev_put(pool, d_cnt);
  out->n_nodes4 = total;
  out->max_stack4 = cnt[1];
  out->gpu_ms = ms;
  return 0;
}

}  // namespace rtgpu
