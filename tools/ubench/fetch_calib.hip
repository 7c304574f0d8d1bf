// Calibration of rocprofv3's FETCH_SIZE for THIS path's access patterns
// (MI355X_MICROARCH.md, HBM section: "other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Every lane reads
// L x 16 B of a random record of a table far larger than L2 and the Infinity Cache:
//   calib_random<4, 64>   a 64-byte BVH2 node          (4 loads)
//   calib_random<6, 96>   a 96-byte four-wide node     (6 loads; straddles a 128-byte line every other record)
//   calib_random<3, 48>   a 48-byte triangle record    (3 loads)
//   calib_stream          a wide coalesced 16 B/lane read of the whole table (the case the guide calibrated at
//                         "reports exactly half")
// The program prints the byte count each kernel asked for; tools/fetch_calibration.sh runs it under
//   rocprofv3 --pmc FETCH_SIZE -- tools/ubench/fetch_calib
// and tools/fetch_calibration.py divides, per kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint32_t pcg(uint32_t v) {
  uint32_t s = v * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
  return (w >> 22) ^ w;
}

template <int L, int REC>
__global__ __launch_bounds__(256) void calib_random(const char *tab, uint32_t n_rec, int iters, float *out) {
  uint32_t h = pcg(blockIdx.x * 256 + threadIdx.x);
  float acc = 0.f;
  for (int it = 0; it < iters; it++) {
    const float4 *p = (const float4 *)(tab + (size_t)(h % n_rec) * REC);
    float4 v[L];
#pragma unroll
    for (int j = 0; j < L; j++) v[j] = p[j];
#pragma unroll
    for (int j = 0; j < L; j++) acc += v[j].x * v[j].y + v[j].z * v[j].w;  // every component is used: the loads stay 16 bytes wide
    h = pcg(h + it);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void calib_stream(const float4 *tab, size_t n_vec, float *out) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  float acc = 0.f;
  for (; i < n_vec; i += stride) { float4 a = tab[i]; acc += a.x * a.y + a.z * a.w; }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void fill(float4 *tab, size_t n_vec) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  for (; i < n_vec; i += stride) tab[i] = make_float4((float)(i & 1023), 1.f, 2.f, 3.f);
}

int main() {
  CK(hipSetDevice(0));
  const size_t bytes = 3ull << 30;  // 3 GiB: 48 Mi records of 64 B, 32 Mi of 96 B, 64 Mi of 48 B
  const size_t n_vec = bytes / 16;
  char *tab; float *out;
  CK(hipMalloc(&tab, bytes));
  const int blocks = 256 * 8, iters = 64;
  CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (float4 *)tab, n_vec);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((calib_random<4, 64>), dim3(blocks), dim3(256), 0, 0, tab, (uint32_t)(bytes / 64), iters, out);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((calib_random<6, 96>), dim3(blocks), dim3(256), 0, 0, tab, (uint32_t)(bytes / 96), iters, out);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((calib_random<3, 48>), dim3(blocks), dim3(256), 0, 0, tab, (uint32_t)(bytes / 48), iters, out);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(calib_stream, dim3(blocks), dim3(256), 0, 0, (const float4 *)tab, n_vec, out);
  CK(hipDeviceSynchronize());
  const double lanes = (double)blocks * 256 * iters;
  printf("requested calib_random<4,64> %.0f bytes\n", lanes * 64);
  printf("requested calib_random<6,96> %.0f bytes\n", lanes * 96);
  printf("requested calib_random<3,48> %.0f bytes\n", lanes * 48);
  printf("requested calib_stream %.0f bytes\n", (double)bytes);
  return 0;
}
