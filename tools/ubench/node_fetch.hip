// Micro-benchmark for the trees in memory: how many dependent record fetches per cycle does one CU sustain when
//   G = 1 : every lane follows its own chain and reads L x 16 B of its record (trace_kernel: 64 records per wave-step)
//   G = 4 / 8 : G neighbouring lanes follow ONE chain; lane g reads the 16-byte pieces g, g + G, ... of the record
//               (each load instruction then covers G x 16 contiguous bytes per group: 64 / G records per wave-step)
// over tables that live in L1 / L2 / the Infinity Cache / HBM, with 4 ... 32 waves per CU.  The next record index
// depends on the loaded data (a dependent chain, like a traversal); FILL dependent fmas stand in for a step's arithmetic.
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

template <int G, int L, int REC, int FILL>
__global__ __launch_bounds__(256) void k(const char *tab, uint32_t rec_mask, int iters, float *out) {
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  const unsigned g = threadIdx.x & (G - 1);
  uint32_t h = pcg(tid / G);
  float acc = 0.f;
  for (int it = 0; it < iters; it++) {
    const uint32_t r = h & rec_mask;
    const float4 *p = (const float4 *)(tab + (size_t)r * REC) + g;
    float4 v[L];
#pragma unroll
    for (int j = 0; j < L; j++) v[j] = p[j * G];
    uint32_t w0 = 0;
#pragma unroll
    for (int j = 0; j < L; j++) { acc += v[j].x * v[j].y + v[j].z; w0 += __float_as_uint(v[j].w) ^ __float_as_uint(v[j].y) ^ __float_as_uint(v[j].x) ^ __float_as_uint(v[j].z); }
    if (G > 1) w0 = (uint32_t)__shfl((int)w0, (int)((threadIdx.x & 63u) & ~(unsigned)(G - 1)), 64);  // the group continues from what its first lane saw
    h = pcg(h + w0);
#pragma unroll
    for (int f = 0; f < FILL; f++) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
  }
  out[tid] = acc;
}

struct Case { const char *name; int G, L, rec; void (*fn0)(const char *, uint32_t, int, float *); void (*fn1)(const char *, uint32_t, int, float *); int fill; };

int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double clk = 2.4e9;  // nominal shader clock
  const size_t max_bytes = 1ull << 30;
  std::vector<uint32_t> host(max_bytes / 4);
  uint32_t s = 12345u;
  for (auto &x : host) { s = s * 1664525u + 1013904223u; x = s; }
  char *tab; float *out;
  CK(hipMalloc(&tab, max_bytes)); CK(hipMemcpy(tab, host.data(), max_bytes, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("CUs %d; records per cycle per CU at the nominal 2.4 GHz; cycles = one dependent step of a wave\n", cus);
  const Case cases[] = {
      {"lane/record  6x16B of  96B", 1, 6, 96, k<1, 6, 96, 0>, k<1, 6, 96, 100>, 100},
      {"lane/record  6x16B of 128B", 1, 6, 128, k<1, 6, 128, 0>, k<1, 6, 128, 100>, 100},
      {"lane/record  4x16B of  64B", 1, 4, 64, k<1, 4, 64, 0>, k<1, 4, 64, 40>, 40},
      {"lane/record  3x16B of  48B", 1, 3, 48, k<1, 3, 48, 0>, k<1, 3, 48, 40>, 40},
      {"lane/record  1x16B of  16B", 1, 1, 16, k<1, 1, 16, 0>, k<1, 1, 16, 20>, 20},
      {"quad/record  2x16B of 128B", 4, 2, 128, k<4, 2, 128, 0>, k<4, 2, 128, 30>, 30},
      {"quad/record  1x16B of  64B", 4, 1, 64, k<4, 1, 64, 0>, k<4, 1, 64, 20>, 20},
      {"oct/record   1x16B of 128B", 8, 1, 128, k<8, 1, 128, 0>, k<8, 1, 128, 20>, 20},
  };
  for (int fill = 1; fill < 2; fill++)
    for (int log2_bytes : {14, 21, 23, 25, 29}) {   // 16 KB (L1), 2 MB (one L2), 8 MB, 32 MB (the eight L2s / Infinity Cache), 512 MB (HBM)
      for (const Case &c : cases)
        for (int blocks_per_cu : {1, 4, 6}) {
          uint32_t n_rec = 1; while ((size_t)n_rec * 2 * c.rec <= (1ull << log2_bytes)) n_rec *= 2;
          const int iters = log2_bytes >= 25 ? 300 : 800;
          auto launch = [&]() { hipLaunchKernelGGL(fill ? c.fn1 : c.fn0, dim3(cus * blocks_per_cu), dim3(256), 0, 0, tab, n_rec - 1, iters, out); };
          launch(); CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          const double waves = blocks_per_cu * 4.0, rec_per_wave_step = 64.0 / c.G;
          const double cycles = ms * 1e-3 * clk;
          printf("%s fill %3d table %6d KB waves/CU %2.0f: %.3f ms  %.3f records/cycle/CU  %6.0f cycles/step  %5.1f cycles per wave-load\n", c.name,
                 fill ? c.fill : 0, (int)((size_t)n_rec * c.rec >> 10), waves, ms, iters * waves * rec_per_wave_step / cycles,
                 cycles / iters, cycles / iters / waves / c.L);
        }
      printf("\n");
    }
  return 0;
}
