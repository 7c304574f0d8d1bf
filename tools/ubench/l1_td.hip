// Micro-benchmark: what does a divergent 64-byte node fetch cost per CU through
// the vector L1 (TA/TCP/TD) vs through LDS?  Each lane reads 4 x 16 B from a
// random 64-B record of a small (L1-resident) table; `active` lanes per wave
// take part.  Reports cycles per wave-level node fetch per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint32_t pcg(uint32_t v) {
  uint32_t s = v * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
  return (w >> 22) ^ w;
}

template <int MODE>  // 0: global 4 x dwordx4, 1: LDS 4 x b128 AoS, 2: global 1 x dwordx4, 3: LDS 4 planes [j][rec]
__global__ __launch_bounds__(256) void k(const float4 *tab, uint32_t n_rec, int iters, int active, int same,
                                          float *out) {
  extern __shared__ float4 lds[];
  if (MODE == 1) {
    for (uint32_t i = threadIdx.x; i < n_rec * 4; i += blockDim.x) lds[i] = tab[i];
    __syncthreads();
  }
  if (MODE == 3) {
    for (uint32_t i = threadIdx.x; i < n_rec * 4; i += blockDim.x) lds[(i & 3) * n_rec + (i >> 2)] = tab[i];
    __syncthreads();
  }
  unsigned lane = threadIdx.x & 63;
  uint32_t h = pcg(blockIdx.x * 256 + threadIdx.x);
  float acc = 0.f;
  if ((int)lane < active) {
    for (int it = 0; it < iters; it++) {
      uint32_t r = same ? (pcg(it) % n_rec) : (h % n_rec);
      float4 a, b, c, d;
      if (MODE == 0) {
        const float4 *p = tab + (size_t)r * 4;
        a = p[0]; b = p[1]; c = p[2]; d = p[3];
      } else if (MODE == 1) {
        const float4 *p = lds + r * 4;
        a = p[0]; b = p[1]; c = p[2]; d = p[3];
      } else if (MODE == 3) {
        a = lds[r]; b = lds[n_rec + r]; c = lds[2 * n_rec + r]; d = lds[3 * n_rec + r];
      } else {
        a = tab[r]; b = c = d = a;
      }
      acc += a.x + b.y + c.z + d.w;
      h = pcg(h + __float_as_uint(d.w));  // dependent: next address needs this fetch
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  int cus = prop.multiProcessorCount;
  const uint32_t n_rec = 460;
  std::vector<float> h(n_rec * 16);
  for (auto &x : h) x = (float)(rand() % 1000) * 1e-3f;
  float4 *tab; float *out;
  CK(hipMalloc(&tab, h.size() * 4)); CK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  int blocks_per_cu = 4;
  CK(hipMalloc(&out, (size_t)cus * blocks_per_cu * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int iters = 2000;
  double clk_ghz = prop.clockRate * 1e-6;
  printf("CUs %d clock %.2f GHz, %d blocks/CU of 256 (=%d waves/CU), table %u x 64 B\n", cus, clk_ghz, blocks_per_cu,
         blocks_per_cu * 4, n_rec);
  for (int mode = 0; mode < 4; mode++)
    for (int same = 0; same < 2; same++)
      for (int active : {64, 48, 32, 16, 8}) {
        auto launch = [&]() {
          dim3 g(cus * blocks_per_cu), b(256);
          if (mode == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, tab, n_rec, iters, active, same, out);
          if (mode == 1) hipLaunchKernelGGL(k<1>, g, b, n_rec * 64, 0, tab, n_rec, iters, active, same, out);
          if (mode == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, tab, n_rec * 4, iters, active, same, out);
          if (mode == 3) hipLaunchKernelGGL(k<3>, g, b, n_rec * 64, 0, tab, n_rec, iters, active, same, out);
        };
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double fetches_per_cu = (double)iters * blocks_per_cu * 4;  // wave-level fetches per CU
        double cyc = ms * 1e-3 * clk_ghz * 1e9 / fetches_per_cu;
        printf("mode %d (%s) same_addr=%d active=%2d: %.3f ms  %.1f cycles per wave-fetch per CU\n", mode,
               mode == 0 ? "global 4x16B" : mode == 1 ? "LDS 4xb128 AoS" : mode == 2 ? "global 1x16B" : "LDS 4 planes", same, active, ms, cyc);
      }
  return 0;
}
