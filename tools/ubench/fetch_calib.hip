// Calibration of rocprofv3's FETCH_SIZE for THIS path's access pattern
// (MI355X_MICROARCH.md, HBM section: "other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Every lane
// reads 4 x 16 B of a random 64-B record of a table far larger than L2 and the
// Infinity Cache, exactly like a divergent BVH node fetch.  The program prints
// the byte count it asked for; run it under
//   rocprofv3 --pmc FETCH_SIZE -- tools/ubench/fetch_calib
// and divide.  (mode 1: wide coalesced 16 B/lane streaming read of the same
// number of bytes, the case the guide calibrated at "reports exactly half".)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint32_t pcg(uint32_t v) {
  uint32_t s = v * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
  return (w >> 22) ^ w;
}

__global__ __launch_bounds__(256) void calib_random(const float4 *tab, uint32_t n_rec, int iters, float *out) {
  uint32_t h = pcg(blockIdx.x * 256 + threadIdx.x);
  float acc = 0.f;
  for (int it = 0; it < iters; it++) {
    const float4 *p = tab + (size_t)(h % n_rec) * 4;
    float4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc += a.x + b.y + c.z + d.w;
    h = pcg(h + it);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void calib_stream(const float4 *tab, size_t n_vec, float *out) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  float acc = 0.f;
  for (; i < n_vec; i += stride) { float4 a = tab[i]; acc += a.x + a.w; }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void fill(float4 *tab, size_t n_vec) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  for (; i < n_vec; i += stride) tab[i] = make_float4((float)(i & 1023), 1.f, 2.f, 3.f);
}

int main() {
  CK(hipSetDevice(0));
  const uint32_t n_rec = 1u << 25;  // 32 Mi records x 64 B = 2 GiB
  const size_t n_vec = (size_t)n_rec * 4;
  float4 *tab; float *out;
  CK(hipMalloc(&tab, n_vec * 16));
  const int blocks = 256 * 8, iters = 64;
  CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, tab, n_vec);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(calib_random, dim3(blocks), dim3(256), 0, 0, tab, n_rec, iters, out);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(calib_stream, dim3(blocks), dim3(256), 0, 0, tab, n_vec, out);
  CK(hipDeviceSynchronize());
  printf("calib_random: requested %.0f bytes (%d lanes x %d records x 64 B)\n", (double)blocks * 256 * iters * 64,
         blocks * 256, iters);
  printf("calib_stream: requested %.0f bytes\n", (double)n_vec * 16);
  return 0;
}
