// Micro-benchmark + exhaustive check: cheaper instruction sequences for the
// correctly rounded 1/x and sqrt(x) the trace kernel needs (the oracle computes IEEE
// division and sqrtf, so a replacement has to give the SAME BITS for every input it
// can see).  For each candidate: (1) all 2^32 bit patterns are compared with hipcc's
// IEEE sequence and mismatches are counted per input class, (2) throughput is timed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ float ieee_rcp(float x) { return 1.0f / x; }
__device__ __forceinline__ float ieee_sqrt(float x) { return __builtin_sqrtf(x); }

// 1/x: hardware estimate (1 ulp) + one Newton step in fma form
__device__ __forceinline__ float rcp_nr1(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  float e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
// two Newton steps
__device__ __forceinline__ float rcp_nr2(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  float e = __builtin_fmaf(-x, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
// sqrt: hardware estimate, then pick among s-1ulp, s, s+1ulp by the sign of the exact
// residuals (what hipcc emits, minus the rescaling of tiny inputs and the class fix-up)
__device__ __forceinline__ float sqrt_fix(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
  float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
  s = rm <= 0.0f ? sm : s;
  s = rp > 0.0f ? sp : s;
  return s;
}
// sqrt: Markstein / Goldschmidt from the reciprocal square root estimate
__device__ __forceinline__ float sqrt_rsq(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  float g = x * y, h = 0.5f * y;
  float r = __builtin_fmaf(-h, g, 0.5f);
  g = __builtin_fmaf(g, r, g);
  h = __builtin_fmaf(h, r, h);
  float d = __builtin_fmaf(-g, g, x);
  return __builtin_fmaf(d, h, g);
}

enum { N_CLASS = 256 };
// class of an input = its biased exponent field (0 = zero/denormal, 255 = inf/nan)
__device__ int classify(float x) { return (int)((__float_as_uint(x) >> 23) & 0xFFu); }

template <int WHICH>
__global__ void check(unsigned long long *bad /* [N_CLASS] */, uint32_t *example /* [N_CLASS] */) {
  unsigned long long i0 = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 256ull;
  // the 256 consecutive bit patterns of one thread share their exponent field
  unsigned long long cnt = 0; int cls = 0; uint32_t ex = 0;
  for (unsigned k = 0; k < 256; k++) {
    uint32_t bits = (uint32_t)(i0 + k);
    float x = __uint_as_float(bits);
    float want, got;
    if (WHICH < 2) { want = ieee_rcp(x); got = WHICH == 0 ? rcp_nr1(x) : rcp_nr2(x); }
    else {
      if (bits & 0x80000000u) continue;  // sqrt of negatives: NaN either way, not compared
      want = ieee_sqrt(x); got = WHICH == 2 ? sqrt_fix(x) : sqrt_rsq(x);
    }
    bool same = __float_as_uint(want) == __float_as_uint(got) || (want != want && got != got);
    if (!same) { cls = classify(x); cnt++; ex = bits; }
  }
  if (cnt) { atomicAdd(&bad[cls], cnt); example[cls] = ex; }
}

template <int WHICH>
__global__ void timeit(float *out, int iters) {
  float x = 1.0f + (float)threadIdx.x * 1e-3f, acc = 0.f;
  for (int i = 0; i < iters; i++) {
    float r;
    if (WHICH == 0) r = rcp_nr1(x); else if (WHICH == 1) r = rcp_nr2(x); else if (WHICH == 2) r = sqrt_fix(x);
    else if (WHICH == 3) r = sqrt_rsq(x); else if (WHICH == 4) r = ieee_rcp(x); else r = ieee_sqrt(x);
    acc += r;
    x = x * 1.0000001f + r * 1e-7f;  // dependent, so nothing is hoisted
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int WHICH>
void run_check(const char *name, unsigned long long *d_bad, uint32_t *d_ex) {
  CK(hipMemset(d_bad, 0, N_CLASS * 8)); CK(hipMemset(d_ex, 0, N_CLASS * 4));
  hipLaunchKernelGGL(check<WHICH>, dim3((1u << 24) / 256), dim3(256), 0, 0, d_bad, d_ex);  // 2^24 threads x 256 = 2^32
  CK(hipDeviceSynchronize());
  unsigned long long bad[N_CLASS]; uint32_t ex[N_CLASS];
  CK(hipMemcpy(bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost)); CK(hipMemcpy(ex, d_ex, sizeof(ex), hipMemcpyDeviceToHost));
  printf("%-10s inputs whose result differs from IEEE, by exponent field of the input (0 = zero/denormal, 255 = inf/nan):", name);
  unsigned long long total = 0;
  for (int c = 0; c < N_CLASS; c++) if (bad[c]) { printf("  [%d: %llu, e.g. %08x]", c, bad[c], ex[c]); total += bad[c]; }
  printf("  total %llu\n", total);
}

template <int WHICH>
void run_time(const char *name, float *d_out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4096, blocks = 256 * 16;
  hipLaunchKernelGGL(timeit<WHICH>, dim3(blocks), dim3(256), 0, 0, d_out, iters);
  CK(hipEventRecord(e0)); hipLaunchKernelGGL(timeit<WHICH>, dim3(blocks), dim3(256), 0, 0, d_out, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double per = ms * 1e-3 * 2.4e9 / ((double)iters * blocks * 4 / 1024.0);  // cycles per wave-level evaluation per SIMD
  printf("%-10s %.3f ms  ~%.1f SIMD cycles per evaluation (incl. 3 loop instructions)\n", name, ms, per);
}

int main() {
  CK(hipSetDevice(0));
  unsigned long long *d_bad; uint32_t *d_ex; float *d_out;
  CK(hipMalloc(&d_bad, N_CLASS * 8)); CK(hipMalloc(&d_ex, N_CLASS * 4)); CK(hipMalloc(&d_out, 256 * 16 * 256 * 4));
  run_check<0>("rcp_nr1", d_bad, d_ex);
  run_check<1>("rcp_nr2", d_bad, d_ex);
  run_check<2>("sqrt_fix", d_bad, d_ex);
  run_check<3>("sqrt_rsq", d_bad, d_ex);
  run_time<4>("ieee_rcp", d_out);
  run_time<0>("rcp_nr1", d_out);
  run_time<1>("rcp_nr2", d_out);
  run_time<5>("ieee_sqrt", d_out);
  run_time<2>("sqrt_fix", d_out);
  run_time<3>("sqrt_rsq", d_out);
  return 0;
}
