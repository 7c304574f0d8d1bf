// pack48.hip — the packed 48-byte nodes (include/rtgpu_format.h) made on the GPU
// from the 64-byte nodes, for trees that are built (lbvh_build.hip) or refitted
// (refit.hip) there.  Same function as the host builder's (pack48.h), one node per
// thread.  No reference counterpart (the reference has no BVH).
#include <hip/hip_runtime.h>

#include <string>

#include "pack48.h"
#include "rtgpu_internal.h"

namespace {

__global__ __launch_bounds__(256) void pack48_kernel(const rtgpu_node *nodes, uint32_t n, rtgpu_node48 *out,
                                                     uint32_t *bad) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    rtgpu_node48 p;
    if (rtgpu::pack_node48(nodes[i], &p)) out[i] = p;
    else atomicOr(bad, 1u);
  }
}

// four-wide nodes after a refit: every child box re-read from the refitted 64-byte nodes
// through the source map of the collapse; references and topology stay
__global__ __launch_bounds__(256) void refresh4_kernel(const rtgpu_node *nodes, rtgpu_node4 *n4, const uint32_t *src,
                                                       uint32_t n, uint32_t *bad) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    rtgpu_node4 nd = n4[i];
    for (int k = 0; k < 4; k++) {
      const uint32_t s = src[4 * (size_t)i + k];
      if (s == 0xFFFFFFFFu) continue;  // missing child: its far box stays
      uint32_t *w = &nd.w[6 * k];
      const uint32_t ref = (w[0] & 0xFFu) | ((w[1] & 0xFFu) << 8) | ((w[2] & 0xFFu) << 16) | ((w[3] & 0xFFu) << 24);
      const rtgpu_node &b = nodes[s >> 1];
      const float *lo = (s & 1u) ? b.lo1 : b.lo0, *hi = (s & 1u) ? b.hi1 : b.hi0;
      for (int j = 0; j < 3; j++) { w[j] = __float_as_uint(lo[j]); w[3 + j] = __float_as_uint(hi[j]); }
      bool ok = (w[4] & 0x7F800000u) != 0x7F800000u && (w[5] & 0x7F800000u) != 0x7F800000u;
      for (int j = 0; j < 4; j++) ok = rtgpu::embed_byte(w[j], (ref >> (8 * j)) & 0xFFu, j < 3, &w[j]) && ok;
      if (!ok) atomicOr(bad, 1u);
    }
    n4[i] = nd;
  }
}

}  // namespace

namespace rtgpu {

// The current device must be the context's.  Synchronous.  d_flag: one u32 of device
// scratch the caller keeps, or NULL to allocate it here.
int refresh_nodes4_device(const void *d_nodes, void *d_nodes4, const void *d_src, uint64_t n4, uint32_t *d_flag,
                          std::string &err) {
  if (n4 == 0) return 0;
  uint32_t *d_bad = d_flag, bad = 0;
  hipError_t e = d_flag ? hipSuccess : hipMalloc((void **)&d_bad, sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemset(d_bad, 0, sizeof(uint32_t));
  if (e == hipSuccess) {
    unsigned blocks = (unsigned)((n4 + 255) / 256);
    if (blocks > 8192u) blocks = 8192u;
    hipLaunchKernelGGL(refresh4_kernel, dim3(blocks), dim3(256), 0, 0, (const rtgpu_node *)d_nodes, (rtgpu_node4 *)d_nodes4,
                       (const uint32_t *)d_src, (uint32_t)n4, d_bad);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost);
  if (!d_flag) (void)hipFree(d_bad);
  if (e != hipSuccess) { err = std::string("refresh4: ") + hipGetErrorString(e); return 2; }
  if (bad) { err = "a box plane is not finite or too close to FLT_MAX for the four-wide node format"; return 1; }
  return 0;
}

// The current device must be the context's.  Synchronous.
int pack_nodes48_device(const void *d_nodes, uint64_t n, void *d_nodes48, uint32_t *d_flag, std::string &err) {
  if (n == 0) return 0;
  uint32_t *d_bad = d_flag, bad = 0;
  hipError_t e = d_flag ? hipSuccess : hipMalloc((void **)&d_bad, sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemset(d_bad, 0, sizeof(uint32_t));
  if (e == hipSuccess) {
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (blocks > 8192u) blocks = 8192u;
    hipLaunchKernelGGL(pack48_kernel, dim3(blocks), dim3(256), 0, 0, (const rtgpu_node *)d_nodes, (uint32_t)n,
                       (rtgpu_node48 *)d_nodes48, d_bad);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost);
  if (!d_flag) (void)hipFree(d_bad);
  if (e != hipSuccess) { err = std::string("pack48: ") + hipGetErrorString(e); return 2; }
  if (bad) { err = "a box plane is not finite or too close to FLT_MAX for the packed node format"; return 1; }
  return 0;
}

}  // namespace rtgpu
