// Micro-benchmark: do two workgroups of T threads become co-resident on one CU?
// grid = 2 x CUs blocks, each records when it started and then spins for ~1 ms.  If
// every block starts within microseconds, two are resident per CU; if half of them
// start a millisecond late, only one is.  Variants: threads per block, dynamic LDS per
// block, registers (a kernel padded to ~96 VGPRs by live values), scratch use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int T, int REGS, int SCRATCH>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(T * 2 / 256, T * 2 / 256)))
void spin(unsigned long long *start, float *sink, int spin_ticks) {
  extern __shared__ float lds[];
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  float r[REGS];
  for (int i = 0; i < REGS; i++) r[i] = (float)(threadIdx.x + i);
  volatile float scratch[SCRATCH ? 16 : 1];
  if (SCRATCH) for (int i = 0; i < 16; i++) scratch[i] = r[i % REGS];
  lds[threadIdx.x] = r[0];
  __syncthreads();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) {
    for (int i = 0; i < REGS; i++) r[i] = r[i] * 1.0001f + lds[(threadIdx.x + i) % T];
  }
  float acc = 0.f;
  for (int i = 0; i < REGS; i++) acc += r[i];
  if (SCRATCH) acc += scratch[threadIdx.x & 15];
  if (threadIdx.x == 0) start[blockIdx.x] = t0;
  sink[blockIdx.x * T + threadIdx.x] = acc;
}

template <int T, int REGS, int SCRATCH>
void run(int cus, size_t lds, unsigned long long *d_start, float *d_sink, const char *what) {
  int blocks = cus * 2;
  CK(hipFuncSetAttribute((const void *)spin<T, REGS, SCRATCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)spin<T, REGS, SCRATCH>, T, lds));
  hipLaunchKernelGGL((spin<T, REGS, SCRATCH>), dim3(blocks), dim3(T), lds, 0, d_start, d_sink, 100000);  // 1 ms at 100 MHz
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> s(blocks);
  CK(hipMemcpy(s.data(), d_start, blocks * 8, hipMemcpyDeviceToHost));
  unsigned long long t0 = *std::min_element(s.begin(), s.end());
  int late = 0;
  for (auto v : s) if (v - t0 > 50000) late++;  // started more than 0.5 ms after the first
  printf("%-44s T=%4d lds=%6zu B: occupancy API %d, %d of %d blocks started late -> %s\n", what, T, lds, occ, late, blocks,
         late == 0 ? "2 per CU" : "1 per CU");
}

int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  unsigned long long *d_start; float *d_sink;
  CK(hipMalloc(&d_start, cus * 2 * 8)); CK(hipMalloc(&d_sink, (size_t)cus * 2 * 1024 * 4));
  run<512, 8, 0>(cus, 70 << 10, d_start, d_sink, "512 threads, few registers");
  run<640, 8, 0>(cus, 70 << 10, d_start, d_sink, "640 threads, few registers");
  run<768, 8, 0>(cus, 70 << 10, d_start, d_sink, "768 threads, few registers");
  run<640, 8, 0>(cus, 78 << 10, d_start, d_sink, "640 threads, few registers, 78 KiB LDS");
  run<640, 64, 0>(cus, 70 << 10, d_start, d_sink, "640 threads, ~80-96 VGPRs");
  run<640, 64, 1>(cus, 70 << 10, d_start, d_sink, "640 threads, ~80-96 VGPRs, scratch");
  run<640, 8, 1>(cus, 70 << 10, d_start, d_sink, "640 threads, few registers, scratch");
  run<768, 48, 0>(cus, 70 << 10, d_start, d_sink, "768 threads, ~64-80 VGPRs");
  return 0;
}
