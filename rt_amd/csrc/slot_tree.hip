// slot_tree.hip — the slot tree (include/rtgpu_format.h) made on the GPU from the four-wide nodes and
// the leaf-ordered primitive arrays of a context, whoever built them (host or device) and after every
// refit.  Three passes over the four-wide nodes, one node per thread:
//   1. slots each node's family (its children) takes; an exclusive scan numbers the families in the
//      order of their parents, so a family is contiguous and every child lies after its parent;
//   2. every inner child learns its slot from its parent;
//   3. every node writes its own slot and copies the primitives of its leaf children into the family.
// The layout (which slot every node and every primitive has) depends on the topology only: it is kept, and after a REFIT -
// new boxes and primitives under the same tree - the slots are refreshed in place by kernels that need no scan and no
// look at the tree's shape: one thread per node for the node slots, four per triangle for the triangle slots.
// No reference counterpart (the reference has no BVH).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include <string>

#include "dev_pool.h"
#include "rtgpu_internal.h"
#include "slot_tree.h"

namespace {

using rtgpu::SlotChild;

__global__ __launch_bounds__(256) void family_sizes_kernel(const rtgpu_node4 *n4, uint32_t n, uint32_t *fam, uint32_t *bad) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const rtgpu_node4 nd = n4[i];
    uint32_t size = 0;
    for (int k = 0; k < 4; k++) {
      SlotChild c;
      if (!rtgpu::slot_child(nd, k, &c)) atomicOr(bad, 1u);
      size += c.size;
    }
    fam[i] = size;
  }
}

// family of node i = slots 1 + base[i] ...; slot_of[child] for its inner children (the root has slot 0)
__global__ __launch_bounds__(256) void child_slots_kernel(const rtgpu_node4 *n4, uint32_t n, const uint32_t *base, uint32_t *slot_of,
                                                          uint32_t *bad) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const rtgpu_node4 nd = n4[i];
    uint32_t at = 1u + base[i];
    if (i == 0) slot_of[0] = 0u;
    for (int k = 0; k < 4; k++) {
      SlotChild c;
      (void)rtgpu::slot_child(nd, k, &c);
      if (c.size && !(c.ref & RTGPU_REF_LEAF)) {
        if (c.ref < n && c.ref > i) slot_of[c.ref] = at;
        else atomicOr(bad, 2u);
      }
      at += c.size;
    }
  }
}

// every node writes its own slot and tells each primitive of its leaf children which slot is its (the primitives are copied by
// the kernels below, which read the leaf-ordered arrays in order)
__global__ __launch_bounds__(256) void write_slots_kernel(const rtgpu_node4 *n4, uint32_t n, const uint32_t *base, const uint32_t *slot_of,
                                                          uint4 *slots, uint64_t n_slots, uint32_t *tri_slot, uint64_t n_tris,
                                                          uint32_t *sph_slot, uint64_t n_sphs, uint32_t *bad) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const rtgpu_node4 nd = n4[i];
    const uint32_t family = 1u + base[i];
    uint32_t w[16];
    if (!rtgpu::make_node_slot(nd, family, w)) { atomicOr(bad, 4u); continue; }
    // (a well-formed tree fills exactly n_slots slots; anything else - a node no parent names, a family past the end - is reported)
    if (slot_of[i] >= n_slots || family > n_slots) { atomicOr(bad, 8u); continue; }
    uint4 *dst = slots + 4 * (size_t)slot_of[i];
    for (int q = 0; q < 4; q++) dst[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    uint32_t at = family;
    for (int k = 0; k < 4; k++) {
      SlotChild c;
      (void)rtgpu::slot_child(nd, k, &c);
      if (c.size && (c.ref & RTGPU_REF_LEAF)) {
        const uint32_t first = RTGPU_REF_FIRST(c.ref);
        const bool sphere = (c.ref & RTGPU_REF_SPHERE) != 0;
        if ((uint64_t)first + c.size > (sphere ? n_sphs : n_tris) || (uint64_t)at + c.size > n_slots) { atomicOr(bad, 8u); continue; }
        for (uint32_t j = 0; j < c.size; j++) (sphere ? sph_slot : tri_slot)[first + j] = at + j;
      }
      at += c.size;
    }
  }
}

// ---- refresh after a refit: the same layout, new boxes and primitives ----
__global__ __launch_bounds__(256) void refresh_node_slots_kernel(const rtgpu_node4 *n4, uint32_t n, const uint32_t *base, const uint32_t *slot_of,
                                                                 uint4 *slots, uint32_t *bad) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    uint32_t w[16];
    if (!rtgpu::make_node_slot(n4[i], 1u + base[i], w)) { atomicOr(bad, 4u); continue; }
    uint4 *dst = slots + 4 * (size_t)slot_of[i];
    for (int q = 0; q < 4; q++) dst[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
  }
}
// four threads per triangle: quarter q of the record to quarter q of its slot (the fourth quarter of a slot is zero)
__global__ __launch_bounds__(256) void refresh_tri_slots_kernel(const uint4 *tris, uint64_t n_tris, const uint32_t *tri_slot, uint4 *slots,
                                                                uint64_t n_slots, uint32_t *bad) {
  const uint64_t total = n_tris * 4;
  for (uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x; k < total; k += (uint64_t)gridDim.x * 256u) {
    const uint64_t t = k >> 2;
    const uint32_t q = (uint32_t)k & 3u, at = tri_slot[t];
    if (at >= n_slots) { if (q == 0) atomicOr(bad, 16u); continue; }  // a primitive no leaf names
    slots[4 * (size_t)at + q] = q < 3u ? tris[3 * t + q] : make_uint4(0u, 0u, 0u, 0u);
  }
}
__global__ __launch_bounds__(256) void refresh_sph_slots_kernel(const uint4 *sphs, const uint2 *sph_meta, uint64_t n_sphs,
                                                                const uint32_t *sph_slot, uint4 *slots, uint64_t n_slots, uint32_t *bad) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n_sphs; i += (uint64_t)gridDim.x * 256u) {
    const uint32_t at = sph_slot[i];
    if (at >= n_slots) { atomicOr(bad, 16u); continue; }
    uint4 *p = slots + 4 * (size_t)at;
    const uint2 m = sph_meta[i];
    p[0] = sphs[i];
    p[1] = make_uint4(m.x, m.y, 0u, 0u);
    p[2] = p[3] = make_uint4(0u, 0u, 0u, 0u);
  }
}

// the primitives into their slots (full build and refresh alike)
hipError_t copy_primitives(const void *d_tris, const void *d_sphs, const void *d_sph_meta, const rtgpu::SlotTreeDevice *t, uint64_t n_tris,
                           uint64_t n_sphs, uint64_t n_slots, uint32_t *d_bad, hipStream_t stream) {
  if (n_tris) {
    uint64_t b = (n_tris * 4 + 255) / 256;
    hipLaunchKernelGGL(refresh_tri_slots_kernel, dim3((unsigned)(b < 65536 ? b : 65536)), dim3(256), 0, stream, (const uint4 *)d_tris,
                       n_tris, t->d_tri_slot, (uint4 *)t->d_slots, n_slots, d_bad);
  }
  if (n_sphs) {
    uint64_t b = (n_sphs + 255) / 256;
    hipLaunchKernelGGL(refresh_sph_slots_kernel, dim3((unsigned)(b < 16384 ? b : 16384)), dim3(256), 0, stream, (const uint4 *)d_sphs,
                       (const uint2 *)d_sph_meta, n_sphs, t->d_sph_slot, (uint4 *)t->d_slots, n_slots, d_bad);
  }
  return hipGetLastError();
}

hipError_t grow(void **p, size_t *cap, size_t bytes) {
  if (*cap >= bytes) return hipSuccess;
  (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  hipError_t e = hipMalloc(p, bytes ? bytes : 4);
  if (e == hipSuccess) *cap = bytes ? bytes : 4;
  return e;
}

}  // namespace

namespace rtgpu {

// The current device must be the context's.  Synchronous on `stream`.  Returns 0 with out->n_slots > 0 when the
// slot tree was made, 0 with out->n_slots == 0 when this tree has no slot form (a leaf of more than four primitives,
// a box more than 65504 wide, 2^24 slots or more), 2 on a HIP error.  out->d_slots is kept and reused (grow-only);
// the temporaries come from `pool` when there is one (a loop that refits every frame asks for the same sizes each time).
int build_slot_tree_device(const void *d_nodes4, uint64_t n4, const void *d_tris, uint64_t n_tris, const void *d_sphs,
                           const void *d_sph_meta, uint64_t n_sphs, SlotTreeDevice *out, DevPool *pool, void *stream_, std::string &err) {
  hipStream_t stream = (hipStream_t)stream_;
  out->n_slots = 0;
  out->layout_valid = false;
  if (n4 == 0 || n4 >= (1ull << 24)) return 0;
  const uint32_t n = (uint32_t)n4;
  uint32_t *d_fam = nullptr, *d_bad = nullptr;
  void *d_tmp = nullptr;
  size_t tmp_bytes = 0;
  // the layout (kept in `out` for the refreshes after refits): family base and slot per node, slot per primitive
  hipError_t e = grow((void **)&out->d_base, &out->cap_base, (size_t)n * 4);
  if (e == hipSuccess) e = grow((void **)&out->d_slot_of, &out->cap_slot_of, (size_t)n * 4);
  if (e == hipSuccess) e = grow((void **)&out->d_tri_slot, &out->cap_tri_slot, (size_t)n_tris * 4);
  if (e == hipSuccess) e = grow((void **)&out->d_sph_slot, &out->cap_sph_slot, (size_t)n_sphs * 4);
  uint32_t *const d_base = out->d_base, *const d_slot_of = out->d_slot_of;
  if (e == hipSuccess) e = dev_get(pool, (void **)&d_fam, (size_t)n * 4);
  if (e == hipSuccess) e = dev_get(pool, (void **)&d_bad, 4);
  if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, 4, stream);
  if (e == hipSuccess) e = rocprim::exclusive_scan(nullptr, tmp_bytes, d_fam, d_base, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream);
  if (e == hipSuccess) e = dev_get(pool, &d_tmp, tmp_bytes ? tmp_bytes : 64);
  unsigned blocks = (unsigned)((n + 255u) / 256u);
  if (blocks > 16384u) blocks = 16384u;
  uint32_t bad = 0;
  // every node but the root is some node's inner child and every primitive sits in one leaf: the slots are known without a look
  const uint64_t n_slots = n4 + n_tris + n_sphs;
  bool usable = e == hipSuccess && n_slots < (1ull << 24);
  if (usable) {
    const size_t bytes = (size_t)n_slots * 64;
    if (out->cap_bytes < bytes) {
      (void)hipFree(out->d_slots);
      out->d_slots = nullptr;
      out->cap_bytes = 0;
      e = hipMalloc(&out->d_slots, bytes);
      if (e == hipSuccess) out->cap_bytes = bytes;
    }
    // (a primitive that no leaf names keeps an impossible slot and is reported)
    if (e == hipSuccess) e = hipMemsetAsync(out->d_slot_of, 0xFF, (size_t)n * 4, stream);
    if (e == hipSuccess && n_tris) e = hipMemsetAsync(out->d_tri_slot, 0xFF, (size_t)n_tris * 4, stream);
    if (e == hipSuccess && n_sphs) e = hipMemsetAsync(out->d_sph_slot, 0xFF, (size_t)n_sphs * 4, stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(family_sizes_kernel, dim3(blocks), dim3(256), 0, stream, (const rtgpu_node4 *)d_nodes4, n, d_fam, d_bad);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = rocprim::exclusive_scan(d_tmp, tmp_bytes, d_fam, d_base, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(child_slots_kernel, dim3(blocks), dim3(256), 0, stream, (const rtgpu_node4 *)d_nodes4, n, d_base, d_slot_of, d_bad);
      hipLaunchKernelGGL(write_slots_kernel, dim3(blocks), dim3(256), 0, stream, (const rtgpu_node4 *)d_nodes4, n, d_base, d_slot_of,
                         (uint4 *)out->d_slots, n_slots, out->d_tri_slot, n_tris, out->d_sph_slot, n_sphs, d_bad);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = copy_primitives(d_tris, d_sphs, d_sph_meta, out, n_tris, n_sphs, n_slots, d_bad, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    usable = e == hipSuccess && !bad;
  }
  dev_put(pool, d_fam); dev_put(pool, d_bad); dev_put(pool, d_tmp);
  if (e != hipSuccess) { err = std::string("slot tree: ") + hipGetErrorString(e); return 2; }
  if (usable) {
    out->n_slots = n_slots;
    out->layout_slots = n_slots;
    out->layout_valid = true;
    out->layout_n4 = n4; out->layout_tris = n_tris; out->layout_sphs = n_sphs;
  }
  return 0;
}

// After a refit (the same four-wide topology and primitive order: layout_valid, same counts): every slot rewritten in place.
// Returns like build_slot_tree_device; out->n_slots == 0 if a box no longer has a slot form (the layout stays valid).
int refresh_slot_tree_device(const void *d_nodes4, const void *d_tris, const void *d_sphs, const void *d_sph_meta,
                             SlotTreeDevice *out, DevPool *pool, void *stream_, std::string &err) {
  hipStream_t stream = (hipStream_t)stream_;
  const uint64_t n_slots = out->layout_slots;
  out->n_slots = 0;
  if (!out->layout_valid || !out->d_slots) { err = "internal: slot tree refresh without a layout"; return 2; }
  const uint32_t n = (uint32_t)out->layout_n4;
  uint32_t *d_bad = nullptr, bad = 0;
  hipError_t e = dev_get(pool, (void **)&d_bad, 4);
  if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, 4, stream);
  if (e == hipSuccess) {
    unsigned blocks = (unsigned)((n + 255u) / 256u);
    if (blocks > 16384u) blocks = 16384u;
    hipLaunchKernelGGL(refresh_node_slots_kernel, dim3(blocks), dim3(256), 0, stream, (const rtgpu_node4 *)d_nodes4, n, out->d_base,
                       out->d_slot_of, (uint4 *)out->d_slots, d_bad);
    e = copy_primitives(d_tris, d_sphs, d_sph_meta, out, out->layout_tris, out->layout_sphs, n_slots, d_bad, stream);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  dev_put(pool, d_bad);
  if (e != hipSuccess) { err = std::string("slot tree refresh: ") + hipGetErrorString(e); return 2; }
  if (!bad) out->n_slots = n_slots;
  return 0;
}

void free_slot_tree_device(SlotTreeDevice *out) {
  (void)hipFree(out->d_slots); (void)hipFree(out->d_base); (void)hipFree(out->d_slot_of);
  (void)hipFree(out->d_tri_slot); (void)hipFree(out->d_sph_slot);
  *out = SlotTreeDevice();
}

}  // namespace rtgpu
