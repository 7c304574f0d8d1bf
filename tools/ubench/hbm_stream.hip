// Micro-benchmark: the HBM bandwidth this box delivers to plain streaming
// kernels - the achievable peak the trace loop's memory-side traffic is priced against.
// read (sum of 16-byte loads), write (16-byte stores) and copy, 2 GiB per array (beyond the
// 256 MB Infinity Cache), 16 B per lane, grid-stride over 256 CUs x 8 blocks.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_read(const float4 *a, size_t n, float *out) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
  float s = 0.f;
  for (; i + 3 * step < n; i += 4 * step) {  // four independent loads in flight per lane
    float4 v0 = a[i], v1 = a[i + step], v2 = a[i + 2 * step], v3 = a[i + 3 * step];
    s += (v0.x + v1.y) + (v2.z + v3.w);
  }
  for (; i < n; i += step) s += a[i].x;
  if (s == 123.456f) out[0] = s;  // never true: keeps the loads
}
__global__ __launch_bounds__(256) void k_write(float4 *a, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
  for (; i < n; i += step) a[i] = make_float4(v, v, v, v);
}
__global__ __launch_bounds__(256) void k_copy(const float4 *a, float4 *b, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
  for (; i + 3 * step < n; i += 4 * step) {
    float4 v0 = a[i], v1 = a[i + step], v2 = a[i + 2 * step], v3 = a[i + 3 * step];
    b[i] = v0; b[i + step] = v1; b[i + 2 * step] = v2; b[i + 3 * step] = v3;
  }
  for (; i < n; i += step) b[i] = a[i];
}

int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const size_t bytes = 2ull << 30, n = bytes / 16;
  float4 *a, *b;
  float *out;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(a, 1, bytes));
  CK(hipMemset(b, 2, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%s, %d CUs; arrays of %zu MiB\n", prop.gcnArchName, prop.multiProcessorCount, bytes >> 20);
  for (int blocks_per_cu : {4, 8, 16})
    for (int kind = 0; kind < 3; kind++) {
      dim3 g(prop.multiProcessorCount * blocks_per_cu), blk(256);
      auto launch = [&]() {
        if (kind == 0) hipLaunchKernelGGL(k_read, g, blk, 0, 0, a, n, out);
        if (kind == 1) hipLaunchKernelGGL(k_write, g, blk, 0, 0, b, n, 3.0f);
        if (kind == 2) hipLaunchKernelGGL(k_copy, g, blk, 0, 0, a, b, n);
      };
      launch();
      CK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      const double moved = (kind == 2 ? 2.0 : 1.0) * (double)bytes;
      printf("%-5s %2d blocks/CU: %.3f ms  %.0f GB/s\n", kind == 0 ? "read" : kind == 1 ? "write" : "copy", blocks_per_cu, best,
             moved / (best * 1e-3) * 1e-9);
    }
  return 0;
}
