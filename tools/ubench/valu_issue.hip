// Micro-benchmark: the VALU ISSUE ceiling of gfx950, measured, for the
// roofline of the trace kernel.  Streams of independent wave64 instructions, every CU busy,
// 1 / 2 / 4 / 8 waves per SIMD:
//   fma     : v_fma_f32                       (2 flop per lane)
//   pk_fma  : v_pk_fma_f32                    (4 flop per lane)
//   mix     : the trace kernel's own blend - per 16 v_fma_f32: 7 s_* (SALU) and 1 v_cmp + 1 v_cndmask
//   dep     : one dependent v_fma_f32 chain per wave (latency, not throughput)
// Reports instructions per cycle per SIMD by the in-kernel shader clock (s_memtime) and by wall
// time at the nominal 2.4 GHz, the clock the chip held, and TFLOP/s.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define FMA16(OP)                                                                                         \
  OP " %0, %0, %16, %17\n" OP " %1, %1, %16, %17\n" OP " %2, %2, %16, %17\n" OP " %3, %3, %16, %17\n"       \
  OP " %4, %4, %16, %17\n" OP " %5, %5, %16, %17\n" OP " %6, %6, %16, %17\n" OP " %7, %7, %16, %17\n"       \
  OP " %8, %8, %16, %17\n" OP " %9, %9, %16, %17\n" OP " %10, %10, %16, %17\n" OP " %11, %11, %16, %17\n"   \
  OP " %12, %12, %16, %17\n" OP " %13, %13, %16, %17\n" OP " %14, %14, %16, %17\n" OP " %15, %15, %16, %17\n"

constexpr int kPerIter = 64;  // VALU instructions of the measured kind per loop iteration

template <int KIND>
__global__ __launch_bounds__(256) void k(int iters, float seed, float *out, unsigned long long *cyc) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = seed + (float)(threadIdx.x + i);
  float b = 0.999f, c = 1e-3f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[16];
#pragma unroll
  for (int i = 0; i < 16; i++) { p[i].x = a[i]; p[i].y = a[i] + 0.5f; }
  f2 pb = {0.999f, 0.998f}, pc = {1e-3f, 2e-3f};
  unsigned s0 = (unsigned)iters, s1 = 3u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    if (KIND == 0) {
#pragma unroll
      for (int r = 0; r < kPerIter / 16; r++)
        asm volatile(FMA16("v_fma_f32")
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                       "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
                     : "v"(b), "v"(c));
    } else if (KIND == 1) {
#pragma unroll
      for (int r = 0; r < kPerIter / 16; r++)
        asm volatile(FMA16("v_pk_fma_f32")
                     : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]),
                       "+v"(p[8]), "+v"(p[9]), "+v"(p[10]), "+v"(p[11]), "+v"(p[12]), "+v"(p[13]), "+v"(p[14]), "+v"(p[15])
                     : "v"(pb), "v"(pc));
    } else if (KIND == 2) {
      // 16 fma + 7 SALU + compare/select per group of 25: VALU : SALU = 18 : 7 (the C2 launch: 2.56 : 1.13)
#pragma unroll
      for (int r = 0; r < kPerIter / 16; r++) {
        asm volatile(FMA16("v_fma_f32")
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                       "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
                     : "v"(b), "v"(c));
        asm volatile("s_add_u32 %0, %0, %1\n s_lshl_b32 %1, %1, 1\n s_xor_b32 %0, %0, %1\n s_add_u32 %1, %1, 7\n"
                     "s_and_b32 %0, %0, 0xffff\n s_or_b32 %1, %1, 1\n s_sub_u32 %0, %0, %1\n"
                     "v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc\n"
                     : "+s"(s0), "+s"(s1), "+v"(a[0]) : "v"(b) : "vcc", "scc");
      }
    } else {
#pragma unroll
      for (int r = 0; r < kPerIter; r++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y;
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s + (float)(s0 + s1);
  if ((threadIdx.x & 63) == 0) cyc[((size_t)blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double nominal_ghz = 2.4;
  float *out;
  unsigned long long *cyc;
  const size_t max_threads = (size_t)cus * 8 * 256;
  CK(hipMalloc(&out, max_threads * 4));
  CK(hipMalloc(&cyc, max_threads / 64 * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%s, %d CUs, reported clock %.2f GHz; blocks of 256 threads = one wave per SIMD each\n", prop.gcnArchName, cus,
         prop.clockRate * 1e-6);
  printf("%-7s %5s %10s %14s %14s %10s %10s\n", "stream", "w/SIMD", "ms", "inst/cyc/SIMD", "inst/cyc/SIMD", "clock GHz",
         "TFLOP/s");
  printf("%-7s %5s %10s %14s %14s %10s %10s\n", "", "", "", "(s_memtime)", "(wall@2.4)", "(held)", "(wall)");
  const char *names[4] = {"fma", "pk_fma", "mix", "dep"};
  const double flop_per_lane[4] = {2, 4, 2, 2};
  for (int kind = 0; kind < 4; kind++)
    for (int wps : {1, 2, 4, 8}) {
      const int iters = kind == 3 ? 20000 : 40000;
      dim3 g(cus * wps), b(256);
      auto launch = [&]() {
        if (kind == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, iters, 1.0f, out, cyc);
        if (kind == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, iters, 1.0f, out, cyc);
        if (kind == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, iters, 1.0f, out, cyc);
        if (kind == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, iters, 1.0f, out, cyc);
      };
      launch();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> h((size_t)cus * wps * 4);
      CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
      double mean_cyc = 0;
      for (auto v : h) mean_cyc += (double)v;
      mean_cyc /= (double)h.size();
      // VALU instructions of the stream per wave (the mix adds 2 VALU per 16)
      const double valu_per_wave = (double)iters * kPerIter * (kind == 2 ? 18.0 / 16.0 : 1.0);
      const double per_simd = valu_per_wave * wps;  // one wave of each block per SIMD
      const double ipc_clock = per_simd / mean_cyc;
      const double ipc_wall = per_simd / (ms * 1e-3 * nominal_ghz * 1e9);
      const double held = mean_cyc / (ms * 1e-3) * 1e-9;
      const double tflops = (double)iters * kPerIter * 64.0 * flop_per_lane[kind] * wps * 4.0 * cus / (ms * 1e-3) * 1e-12;
      printf("%-7s %5d %10.3f %14.3f %14.3f %10.2f %10.1f\n", names[kind], wps, ms, ipc_clock, ipc_wall, held, tflops);
    }
  printf("peak VALU issue at the nominal clock: 1024 SIMDs x 2.4 GHz x (best inst/cyc/SIMD above)\n");
  return 0;
}
