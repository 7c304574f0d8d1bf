// rtgpu_kernels.hip — the per-pixel trace loop on gfx950 (SURVEY.md §8a rows
// a1, a3-a8) and the C ABI around it (include/rtgpu.h).
//
// No reference counterpart exists for any of this; the written
// specification is the CPU oracle (oracle/rt_oracle.c), which this kernel has
// to match bit for bit: same f32 operations in the same order, no contraction
// (this file is compiled with -ffp-contract=off; the only fused operations are
// the explicit __builtin_fmaf of the slab test), IEEE division and sqrt
// (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt), the same
// counter-based integer RNG.  Vector arithmetic follows the reference's
// rt::Vec3 (code/math/vec.cxx:94-97 dot = mul then x+y+z; :66-69 normalize =
// v * (1/len); :174-211 component-wise operators).
//
// Structure: persistent threads, one 1024-thread block per CU that
// owns the CU's LDS: per-lane traversal stacks ([entry][thread], conflict-free)
// and a copy of the top of the BVH in four 16-byte planes.  A lane owns one work
// item = (pixel, chunk of samples) at a time and walks its samples in order; work
// items come from a global queue in groups of 64 (one atomic per group), chunk
// sums go to per-chunk planes and combine_kernel adds them in order, which keeps
// the frame sum deterministic.  Traversal runs in voted phases (node steps over
// 64-byte, packed 48-byte or four-wide 96-byte nodes / leaf steps of one primitive,
// or two triangles) and is left early when few lanes still traverse, so finished
// lanes shade and start their next ray.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "dev_pool.h"
#include "rtgpu_internal.h"

namespace {

// Launch shape: kBlocksPerCU blocks of kBlock threads per CU, each with an equal share of
// the CU's LDS.  (Macros so that tools/ can build variants for occupancy experiments.)
#ifndef RTGPU_KBLOCK
#define RTGPU_KBLOCK 1024
#endif
#ifndef RTGPU_BLOCKS_PER_CU
#define RTGPU_BLOCKS_PER_CU 1
#endif
#ifndef RTGPU_TRI_STEP
#define RTGPU_TRI_STEP 2
#endif
#ifndef RTGPU_WAVES_PER_EU
#define RTGPU_WAVES_PER_EU (RTGPU_KBLOCK * RTGPU_BLOCKS_PER_CU / 256)
#endif
#ifndef RTGPU_WAVES_PER_EU_MAX  // experiments: a register budget the compiler may aim below without being held to it
#define RTGPU_WAVES_PER_EU_MAX RTGPU_WAVES_PER_EU
#endif
constexpr int      kTriStep    = RTGPU_TRI_STEP;   // triangles a lane tests per leaf step (trees outside LDS)
constexpr int      kBlock      = RTGPU_KBLOCK;
#ifndef RTGPU_KBLOCK2
#define RTGPU_KBLOCK2 768
#endif
constexpr int      kBlock2     = RTGPU_KBLOCK2;    // threads per block of MODE 8 (two rays per lane): 12 waves, three per SIMD
// MODE 13: TWO blocks of kBlock6 threads per CU, each with its own copy of the scene in LDS: 24 waves per CU, six per SIMD
// (a twelve-wave block is three waves on every SIMD, so two of them are co-resident: tools/ubench/residency.hip)
constexpr int      kBlock6     = 768;
constexpr int      kLdsBytes6  = 160 * 1024 / 2;
constexpr int      kBlocksPerCU = RTGPU_BLOCKS_PER_CU; // one block per CU: 16 waves share one LDS copy of the top of the BVH
constexpr int      kStackLds   = 32;   // most entries per lane the LDS stack may get (4 KiB per entry per block)
constexpr int      kLdsBytes   = 160 * 1024 / kBlocksPerCU - (kBlocksPerCU > 1 ? 2048 : 0);  // LDS per CU on gfx950, shared equally by the CU's blocks
constexpr int      kStackOvf   = RTGPU_MAX_TREE_DEPTH + 2;  // global overflow entries per lane (covers stack_lds >= 2)
constexpr uint32_t kDone       = 0xFFFFFFFFu;  // == RTGPU_REF_EMPTY, has the leaf bit
// MODE 2 of the trace kernel: whole tree in LDS, planes this many nodes apart (the last
// plane's immediate ds_read offset, 3 * 16 * kSmallNodes, has to fit in 16 bits)
#ifndef RTGPU_SMALL_NODES
#define RTGPU_SMALL_NODES (RTGPU_BLOCKS_PER_CU == 1 ? 1365 : 512)
#endif
constexpr int      kSmallNodes = RTGPU_SMALL_NODES;
#ifndef RTGPU_NODE_UNROLL  // node steps per vote on the scenes in LDS: the vote is 4 of a step's ~45 instructions (1 / 2 / 3 / 4: C4 206.8 / 204.4 / 207.0 / 209.1 ms)
#define RTGPU_NODE_UNROLL 2
#endif
#ifndef RTGPU_PRIO_SMALL
#define RTGPU_PRIO_SMALL 1
#endif
// MODE 7 of the trace kernel: a scene small enough to sit in LDS WHOLE - tree (planes kTinyNodes apart), leaf-ordered
// spheres, triangles, and per PRIMITIVE what shading needs: the triangle's unit normal and id, the sphere's id and
// 1/r, and a copy of the primitive's material - at compile-time distances from the node planes, so that every read
// of the scene is a ds_read with an immediate offset, no read depends on another, and the kernel touches memory only
// for the work queue and the pixels.  Offsets in 16-byte units from the first plane; the last array ends below 64 KiB.
constexpr int      kTinyNodes = 512, kTinySphs = 496, kTinyTris = 16;
constexpr int      kTinySphAt = 4 * kTinyNodes, kTinyTriAt = kTinySphAt + kTinySphs, kTinyTriNAt = kTinyTriAt + 3 * kTinyTris,
                   kTinySphMatAt = kTinyTriNAt + kTinyTris, kTinyTriMatAt = kTinySphMatAt + 2 * kTinySphs,
                   kTinyMetaAt = kTinyTriMatAt + 2 * kTinyTris, kTinyEnd = kTinyMetaAt + kTinySphs / 2;
// MODE 13: three node planes (the child references, 8 bytes per node, sit at the start of the block's LDS: kTinyRefBytes)
constexpr int      kTinySphAt6 = 3 * kTinyNodes, kTinyEnd6 = kTinyEnd - kTinyNodes, kTinyRefBytes = kTinyNodes * 8;
static_assert(kTinyEnd * 16 <= 65536, "MODE 7: every array within reach of a 16-bit ds_read offset");
// MODE 5 of the trace kernel: the first kWideCached four-wide nodes (the top of the tree: both
// collapses number it breadth first) live in LDS as six 16-byte planes this many nodes apart
// (the last plane's immediate ds_read offset, 5 * 16 * kWideCached, has to fit in 16 bits),
// and the traversal stacks get the rest of the LDS, deeper entries going to global memory
constexpr int      kWideCached = 768;
constexpr int      kPathLdsWords = 6;  // slot-tree modes: words per lane kept in LDS behind the stacks (sum, throughput)
// Launch shape of the slot-tree modes (11, 12): kSlotBlocksPerCU blocks of kBlockSlots threads per CU - one wave per SIMD and
// block, so FIVE waves per SIMD where the 1024-thread block of the other modes gives four (two blocks of 640 threads never become
// co-resident: the hardware spreads a ten-wave block 3 + 3 + 2 + 2 over the SIMDs, twice the same way).  The product kernels of
// these modes live in 96 vector registers for it; their counting / profile variants do not and run four blocks per CU.
#ifndef RTGPU_KBLOCK_SLOTS
#define RTGPU_KBLOCK_SLOTS 256
#endif
#ifndef RTGPU_SLOT_BLOCKS_PER_CU
#define RTGPU_SLOT_BLOCKS_PER_CU 5
#endif
constexpr int      kBlockSlots = RTGPU_KBLOCK_SLOTS;
constexpr int      kSlotBlocksPerCU = RTGPU_SLOT_BLOCKS_PER_CU;
constexpr int      kSlotWaves = kBlockSlots * kSlotBlocksPerCU / 256;               // waves per SIMD of the product kernels
constexpr int      kLdsSlotBytes = 160 * 1024 / kSlotBlocksPerCU - 2048;           // LDS share of one such block
constexpr int      kLdsMats    = 64;   // trees in memory: material tables up to this size are copied to LDS (2 KiB)
constexpr unsigned kChunk      = 64;   // work items drawn per queue atomic (one 8x8 pixel block)
// The END of the queue is handed out in smaller draws from a second head: when the queue runs dry every wave still owns what is
// left of its last draw, which only its own lanes can take - half a draw on average, and the launch ends when the unluckiest wave
// has worked its share off.  How small depends on how fast the grid takes items: a returning atomic on one word saturates near 88
// per microsecond chip-wide, so a wave sizes its draws from the time its own draws have taken so far (KParams.tail_thr*: the host
// knows the waves of the grid).  Frames of slow items (S2: 0.6 ms per item and lane) end on draws of 16, frames of fast ones (S0,
// S1: 0.16 ms) keep 64 - for them small draws measured SLOWER (C2 +4 %, C3 +1 %), for C5's frame 3.9 % faster.
#ifndef RTGPU_TAIL_ATOMICS_PER_US   // the atomic rate the tail draws of the whole grid may add up to
#define RTGPU_TAIL_ATOMICS_PER_US 60
#endif
#ifndef RTGPU_TAIL_SHARE   // 0: no sample sharing at the end of the launch (experiments)
#define RTGPU_TAIL_SHARE 1
#endif
#ifndef RTGPU_TAIL_ITEMS_PER_LANE_X4   // length of that end of the queue: this many quarters of an item per lane of the grid
#define RTGPU_TAIL_ITEMS_PER_LANE_X4 4
#endif
constexpr unsigned kTailHead   = 31;   // counters[] word of the second head

constexpr float    kTMax       = 3.402823466e+38f;  // FLT_MAX: initial closest-hit distance

struct KParams {
  const float4 *nodes;
  const float4 *tris;
  const float4 *sphs;
  const uint2  *sph_meta;
  const float4 *mats;
  float4       *accum;
  uint32_t     *prim_id;
  uint32_t     *ovf_stack;  // [kStackOvf][grid threads], see struct Stack
  unsigned long long *wave_times;  // profile variant, per wave: {start, end, queue-dry} s_memrealtime (100 MHz), outer iterations after dry
  unsigned long long *counters;  // [0] queue head, [1] paths, [2] rays, [3] nodes, [4] tris, [5] sphs, [6] scatters, [7..20] profile
  float cam_origin[3], cam_ll[3], cam_hor[3], cam_ver[3], cam_u[3], cam_v[3];
  float lens_radius;
  float inv_w, inv_h, tmin;
  float sky_bottom[3], sky_top[3];
  uint32_t width, height, spp, sample_offset, max_depth, seed;
  uint32_t ts_log2, tiles_x, rank, world, n_items, flags;
  // items [0, n_big) are drawn kChunk at a time from counters[0], the rest from counters[kTailHead] 16, 32 or 64 at a time: 32 (16)
  // by a wave that has taken at least tail_thr32 (tail_thr16) ticks of 10 ns per draw so far
  uint32_t n_big, tail_thr32, tail_thr16;
  uint32_t stack_lds, stack_lds_only;
  uint32_t n_cached;  // nodes [0, n_cached) are read from the LDS copy
  uint32_t sph_cached; // MODE 2, 7: > 0 = the leaf-ordered spheres (this many: all of them) sit in LDS behind the node planes
  uint32_t tiny_tris;  // MODE 7: triangles to copy to LDS (all of them)
  uint32_t mats_cached;  // modes 0, 1, 3-6: > 0 = the material table (this many entries: all) sits in LDS behind the stacks
  uint32_t tiles_small;  // all tile indices < 2^22: the float tile-row estimate is usable
  float inv_tiles_x;
  // chunked frame sums (rtgpu_settings.chunk_samples): n_chunks work items per pixel; with
  // n_chunks > 1 chunk c of pixel slot i goes to planes[c * plane_stride + i] and
  // combine_kernel adds the planes in order.  (A frame of more than kRoundChunks chunks is traced in ROUNDS, each a launch
  // of its own over its share of the samples - sample_offset, spp and n_chunks are the round's - followed by a combine_kernel
  // that carries the running sum; the kernel does not know.)
  float4 *planes;
  uint32_t chunk_samples, n_chunks, plane_stride, groups_small;
  float inv_n_chunks;
};
constexpr size_t kCounterBytes = 256;
// Chunks of a frame traced per launch (KParams).  64 = the most rtgpu_default_settings ever asks for, so a frame with default
// settings is one round as before; the rounds bound the scratch of a frame whose caller asks for more (spp 4096 in chunks of one
// would be 4096 planes).  A round costs a launch tail: C4's frame (64 chunks) in four rounds of 16 measured 208.7 ms against 201.9.
#ifndef RTGPU_ROUND_CHUNKS
#define RTGPU_ROUND_CHUNKS 64
#endif
constexpr uint32_t kRoundChunks = RTGPU_ROUND_CHUNKS;

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float a, float b) { f2 r = {a, b}; return r; }
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { a = a * b; return a.x + a.y + a.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// 1/x and sqrt(x) with the BITS of IEEE division and sqrtf (what the oracle computes)
// in a third of the instructions hipcc emits for them.  hipcc's sequences rescale
// tiny/huge operands and fix up special values; on the ranges below none of that is
// needed, which tools/ubench/exact_rcp_sqrt.hip verifies EXHAUSTIVELY on gfx950 (all
// 2^32 inputs against hipcc's own sequences):
//   rcp_fast (v_rcp_f32 + one Newton step in fma form) == 1.0f / x  for 2^-126 <= |x| < 2^126
//   sqrt_fast (v_sqrt_f32, then s-1ulp / s / s+1ulp chosen by exact fma residuals)
//             == sqrtf(x)  for x == 0, x >= 2^-104, inf and NaN (x < 0: NaN either way)
// rcp_exact / sqrt_exact take any operand: if some lane of the wave is outside the
// verified range the whole wave takes hipcc's sequence (a wave-uniform branch).
__device__ __forceinline__ float rcp_fast(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  float e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float sqrt_fast(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
  float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
  s = rm <= 0.0f ? sm : s;
  s = rp > 0.0f ? sp : s;
  return s;
}
__device__ __forceinline__ float rcp_exact(float x) {
  const float a = __builtin_fabsf(x);
  const bool in_range = a >= 0x1p-126f && a < 0x1p126f;  // false for NaN
  if (__builtin_expect(__ballot(!in_range) == 0ull, 1)) return rcp_fast(x);
  return 1.0f / x;
}
__device__ __forceinline__ float sqrt_exact(float x) {
  const bool out_of_range = x > 0.0f && x < 0x1p-104f;
  if (__builtin_expect(__ballot(out_of_range) == 0ull, 1)) return sqrt_fast(x);
  return __builtin_sqrtf(x);
}
__device__ __forceinline__ V3 normalize(V3 a) { return a * rcp_exact(sqrt_exact(dot(a, a))); }
__device__ __forceinline__ float fmin_(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
// v_max_f32 / v_min_f32 as written (IEEE mode: same value as fmaxf/fminf for non-NaN
// operands); vmax_'s second operand is wave-uniform and stays in a scalar register
__device__ __forceinline__ float vmax_(float a, float uniform_b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(uniform_b), "v"(a));
  return r;
}
__device__ __forceinline__ float vmin_(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ uint32_t pcg(uint32_t v) {
  uint32_t s = v * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
  return (w >> 22) ^ w;
}

// Draw k of a path is pcg(key + k * 0x9E3779B9).  pcg(v) opens with s = v * A + C, and s is affine in k:
// the state carried is s itself, advanced by A * 0x9E3779B9 per draw (same values mod 2^32), which leaves
// ONE 32-bit multiply per draw (v_mul_lo_u32 is quarter rate) instead of three.
struct Rng {
  uint32_t s;
  __device__ __forceinline__ void start(uint32_t key) { s = key * 747796405u + 2891336453u; }
  __device__ __forceinline__ uint32_t u24() {  // the top 24 bits of the next draw
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    uint32_t h = (w >> 22) ^ w;
    s += 0xDE500ECDu;  // 747796405 * 0x9E3779B9 mod 2^32
    return h >> 8;
  }
  __device__ __forceinline__ float u01() { return (float)u24() * 0x1p-24f; }
};
static_assert((uint32_t)(747796405ull * 0x9E3779B9ull) == 0xDE500ECDu, "step of the carried pcg state");

__device__ __forceinline__ float clamp_dir(float d) {
  return __builtin_fabsf(d) < 1e-30f ? __builtin_copysignf(1e-30f, d) : d;
}

// Uniform direction on the unit sphere from exactly two draws (oracle/rt_oracle.c random_unit_vector, the same
// single f32 operations in the same order): z on a 24-bit grid, azimuth = quadrant + Taylor sine / cosine of an
// angle within [-pi/4, pi/4) in Horner form.  No rejection loop: a loop runs as long as the unluckiest of the
// wave's 64 lanes (3.3 rounds of ~42 instructions on average for the two-dimensional rejection used before).
__device__ __forceinline__ V3 random_unit_vector(Rng &rng) {
  const uint32_t k1 = rng.u24(), k2 = rng.u24();
  const float t = (float)k1 * 0x1p-23f;
  const float z = 1.0f - t;
  const float rad = sqrt_fast(t * (2.0f - t));  // 0 or >= 2^-23: inside sqrt_fast's verified range
  const uint32_t q = k2 >> 22;
  const float f = (float)(k2 & 0x3FFFFFu) * 0x1p-22f - 0.5f;
  const float x = f * 1.57079637f;
  const float x2 = x * x;
  float ps = __builtin_fmaf(x2, 2.75573192e-6f, -1.98412701e-4f);
  ps = __builtin_fmaf(x2, ps, 8.33333377e-3f);
  ps = __builtin_fmaf(x2, ps, -1.66666672e-1f);
  const float sn = __builtin_fmaf(x * x2, ps, x);
  float pc = __builtin_fmaf(x2, 2.48015876e-5f, -1.38888892e-3f);
  pc = __builtin_fmaf(x2, pc, 4.16666679e-2f);
  pc = __builtin_fmaf(x2, pc, -0.5f);
  const float cs = __builtin_fmaf(x2, pc, 1.0f);
  float ex = (q & 1u) ? sn : cs, ey = (q & 1u) ? cs : sn;
  ex = __uint_as_float(__float_as_uint(ex) ^ (((q + 1u) & 2u) << 30));
  ey = __uint_as_float(__float_as_uint(ey) ^ ((q & 2u) << 30));
  return mk(rad * ex, rad * ey, z);
}

__device__ __forceinline__ V3 reflect(V3 d, V3 n) {
  float k = 2.0f * dot(d, n);
  return d - n * k;
}

__device__ __forceinline__ unsigned lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Traversal stack: up to kStackLds entries per lane in LDS, laid out
// [entry][ray of the lane][thread] (lane-consecutive dwords: conflict-free ds_read/ds_write_b32);
// deeper entries go to a per-lane slice of a global overflow buffer laid out the
// same way.  Entry 0 holds the sentinel kDone, so a pop needs no emptiness test:
// popping the sentinel ends the traversal.  A stack never grows past the tree
// depth + 1, which the host knows: when that fits in LDS it launches the
// LDS_ONLY instantiation, whose node step has no branch at all - the far child
// is always written above the top (kept only if both children were hit) and the
// top is always read (used only if neither was).
// STRIDE: words between consecutive entries of one stack (the threads of the block x the rays a lane traverses).
// E: the type of an LDS entry.  MODE 13 keeps its references in 16 bits (int16_t: a ds_read_i16 gives back the sign-extended
// 32-bit form the kernel works on), which halves the stacks; only the LDS_ONLY form may.
template <bool LDS_ONLY, int STRIDE, typename E = uint32_t>
struct Stack {
  static_assert(LDS_ONLY || sizeof(E) == 4, "16-bit entries: LDS-only stacks");
  uint32_t *lds;        // &lds_stack[tid]
  uint32_t *ovf;        // &ovf[global thread]
  uint32_t ovf_stride;  // threads in the grid
  int lds_entries;      // entries of the LDS part in use (<= kStackLds)
  int sp;               // next free entry; >= 1 while a traversal is running (overflow-capable form)
  // LDS_ONLY form: the stack pointer IS the LDS address of the top entry (32-bit, explicit
  // address space), so a step needs no index -> address arithmetic
  typedef __attribute__((address_space(3))) E lds_u32;
  lds_u32 *top_at;      // &entry[sp - 1] of this lane
  lds_u32 *base_at;     // &entry[0] of this lane (the sentinel)
  __device__ __forceinline__ void reset() {
    sp = 1;
    top_at = base_at;
  }
  __device__ __forceinline__ void store(int at, uint32_t v) {
    if (LDS_ONLY || at < lds_entries) lds[at * STRIDE] = v;
    else ovf[(size_t)(at - lds_entries) * ovf_stride] = v;
  }
  __device__ __forceinline__ uint32_t load(int at) {
    if (LDS_ONLY || at < lds_entries) return lds[at * STRIDE];
    return ovf[(size_t)(at - lds_entries) * ovf_stride];
  }
  // the entry a pop would return (LDS form), read at the START of a node step so that the
  // LDS round trip overlaps the node fetch instead of following the slab test
  __device__ __forceinline__ uint32_t peek() const { return LDS_ONLY ? (uint32_t)(int32_t)top_at[0] : 0u; }
  // node step: push `far_c` if both children were hit, pop if none was
  __device__ __forceinline__ uint32_t step(bool both, bool none, uint32_t near_c, uint32_t far_c, uint32_t top) {
    if (LDS_ONLY) {
      // one address: top at [0] (read by peek()), free entry at [STRIDE] (immediate offset)
      top_at[STRIDE] = (E)far_c;
      top_at += both ? STRIDE : (none ? -STRIDE : 0);
      return none ? top : near_c;
    }
    if (both) { store(sp, far_c); sp++; }
    if (none) { sp--; near_c = load(sp); }
    return near_c;
  }
  // four-wide node step: n children hit, r0 the nearest, r1..r3 the others by distance;
  // the others are pushed farthest first (the LDS form writes three entries whatever n is:
  // what lies above the new top is never read)
  __device__ __forceinline__ uint32_t step4(int n, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, uint32_t top) {
    if (LDS_ONLY) {
      top_at[STRIDE] = n == 4 ? r3 : (n == 3 ? r2 : r1);
      top_at[2 * STRIDE] = n == 4 ? r2 : r1;
      top_at[3 * STRIDE] = r1;
      top_at += (n == 0 ? -1 : n - 1) * STRIDE;
      return n == 0 ? top : r0;
    }
    if (n == 0) { sp--; return load(sp); }
    if (n >= 4) { store(sp, r3); sp++; }
    if (n >= 3) { store(sp, r2); sp++; }
    if (n >= 2) { store(sp, r1); sp++; }
    return r0;
  }
  // Overflow-capable form with a fast path (MODE 5): while NO lane of the wave is within three
  // entries of the end of the LDS part, a step is the LDS form's straight-line code on the index
  // sp; otherwise entry by entry as above.  `fast` is wave-uniform; the entry a pop would return
  // is read at the start of the step like peek().
  __device__ __forceinline__ uint32_t peek_hybrid(bool &fast) const {
    fast = __ballot(sp + 3 > lds_entries) == 0ull;
    uint32_t top = 0u;
    if (fast) top = base_at[(sp - 1) * STRIDE];
    return top;
  }
  __device__ __forceinline__ uint32_t step4_hybrid(bool fast, int n, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, uint32_t top) {
    if (fast) {
      lds_u32 *t = base_at + (sp - 1) * STRIDE;
      t[STRIDE] = n == 4 ? r3 : (n == 3 ? r2 : r1);
      t[2 * STRIDE] = n == 4 ? r2 : r1;
      t[3 * STRIDE] = r1;
      sp += n == 0 ? -1 : n - 1;
      return n == 0 ? top : r0;
    }
    if (n == 0) { sp--; return load(sp); }
    if (n >= 4) { store(sp, r3); sp++; }
    if (n >= 3) { store(sp, r2); sp++; }
    if (n >= 2) { store(sp, r1); sp++; }
    return r0;
  }
  __device__ __forceinline__ uint32_t pop_if_hybrid(bool fast, bool doit, uint32_t top) {
    if (fast) {
      sp -= doit ? 1 : 0;
      return top;
    }
    uint32_t v = kDone;
    if (doit) { sp--; v = load(sp); }
    return v;
  }
  // leaf step: pop if the leaf is exhausted
  __device__ __forceinline__ uint32_t pop_if(bool doit, uint32_t top) {  // top = peek() at the start of the step
    if (LDS_ONLY) {
      top_at -= doit ? STRIDE : 0;
      return top;
    }
    uint32_t v = kDone;
    if (doit) { sp--; v = load(sp); }
    return v;
  }
};

// Lanes of the wave for which p holds, as a 32-bit scalar (s_bcnt1_i32_b64).
// Written as asm because the generic popcount is i64 and hipcc then carries the
// vote's comparisons as 64-bit VALU compares of scalars.
__device__ __forceinline__ int wave_count(bool p) {
  unsigned long long m = __ballot(p);
  int r;
  asm("s_bcnt1_i32_b64 %0, %1" : "=s"(r) : "s"(m) : "scc");
  return r;
}

// REFILL: leave the traversal loop as soon as fewer than REFILL lanes of the
// wave are still traversing (and at least one has finished), so that finished
// lanes shade and start their next ray instead of idling behind the longest
// traversal of the wave; unfinished lanes keep their traversal state in
// registers and resume.  REFILL = 0 waits for every lane (plain while-while).
// Profile build only (STATS == 2): count one per WAVE for the code it guards,
// charged to the lowest active lane, so that sum(lane counts) / (64 * wave count)
// is the SIMD lane utilisation of that section.
__device__ __forceinline__ void wave_tick(unsigned lane, unsigned long long &c) {
  unsigned long long m = __ballot(1);
  if (lane == (unsigned)__ffsll((long long)m) - 1u) c++;
}

// MODE 0: the stack may overflow into global memory; 1: the whole stack fits in LDS;
// 2: as 1, and the WHOLE tree (<= kSmallNodes nodes) is in the LDS node cache with its
// four planes a compile-time distance apart, so a node fetch is one address and four
// ds_read with immediate offsets, and no "cached or not" test.  3 and 4: as 0 and 1 with
// the packed 48-byte nodes (rtgpu_format.h): three loads per node step instead of four -
// every node fetch of such a tree goes to memory, and the packed form is a quarter smaller.
// 5 and 6: as 0 and 1 over the four-wide packed nodes (rtgpu_node4): a step tests up to four
// children with six loads, and a ray takes about half as many dependent steps.  6 keeps the whole
// stack in LDS; 5 gives the stack what is left beside an LDS copy of the first kWideCached nodes
// (six planes, kWideCached nodes apart) and sends deeper stack entries to global memory (the LDS
// copy measured slower and is off unless RTGPU_WIDE_CACHE=1; the hybrid stack is what trees with a
// stack bound above 40 entries run on).  7: as 2 for a scene that fits LDS WHOLE (kTiny* above):
// spheres, triangles and per-primitive shading records behind the tree; S0 ran this until mode 13.
// 8: as 7 with TWO rays per lane traversed in lock step (R = 2 below) in blocks of kBlock2 threads:
// every lane owns two work items, each with its own traversal stack; a traversal step requests the
// nodes (or primitives) of BOTH rays before it tests either, so that one ray's LDS round trip hides
// behind the other's arithmetic.  12 waves x 2 rays are 24 dependent chains per CU where 16 waves
// x 1 ray are 16, with ONE copy of the scene in LDS (24 waves of one ray each would need two blocks
// per CU and two copies, which do not fit).  Per ray the operations are exactly those of mode 7.
// 9 and 10: as 5 and 6 with every four-wide node in a 128-BYTE SLOT of the device copy (the 96 bytes of the format, 32 of
// padding): a node is then ONE line for the L2 and the fabric instead of 1.5 on average.  For trees far larger than the
// caches (S2 level 8: 0.4 GB of nodes, L2 hit rate 0.57, the fabric at 5.2 TB/s) the lines are what the launch waits for;
// for trees that live in L1 / L2 the aligned slots cost more than they save (every lane of a load reads the same 16 bytes of
// its line: bank conflicts), so the host chooses by the size of the tree.
// 11 and 12: as 5 and 6 over the SLOT TREE (rtgpu_format.h): four-wide nodes of 64 bytes - the child boxes as binary16 offsets
// from the node's origin, which v_fma_mix_f32 multiplies into the slab test as they are - in ONE array of 64-byte slots with
// the primitives, the children of a node adjacent.  Four loads per node step instead of six, no record across a line.
// 13: as 7 at SIX waves per SIMD: two blocks of kBlock6 threads per CU, each with its own copy of the scene.  What makes both
// fit - 80 vector registers, 80 KiB of LDS per block: the node planes of the LDS copy regrouped per axis, so that v_pk_fma_f32
// picks the component of 1/d and -o/d it needs with op_sel (six registers of slab constants instead of twelve); child
// references and stack entries of 16 bits (sign-extended when read), the reference pairs in a table at LDS offset 0 and an
// inner node's reference the byte offset of its pair; the work item's fields derived or packed.  S0 runs this (RTGPU_TINY6=0: 7).
// (The slot-exchange experiment of round 2 - modes 8-10 then, profiles/r02/experiments.txt H and J -
// was a measured negative and is gone from the source.)
template <int STATS, int REFILL, int MODE>
__global__ __launch_bounds__(MODE == 8 ? kBlock2 : MODE == 13 ? kBlock6 : (MODE == 11 || MODE == 12) ? kBlockSlots : kBlock)
__attribute__((amdgpu_waves_per_eu(MODE == 8 ? kBlock2 * RTGPU_BLOCKS_PER_CU / 256 : MODE == 13 ? (STATS == 0 ? 6 : 3)
                                   : (MODE == 11 || MODE == 12) ? (STATS == 0 ? kSlotWaves : kSlotWaves - 1) : RTGPU_WAVES_PER_EU,
                                   MODE == 8 ? kBlock2 * RTGPU_BLOCKS_PER_CU / 256 : MODE == 13 ? (STATS == 0 ? 6 : 3)
                                   : (MODE == 11 || MODE == 12) ? (STATS == 0 ? kSlotWaves : kSlotWaves - 1) : RTGPU_WAVES_PER_EU_MAX)))
void trace_kernel(const KParams P) {
  constexpr bool LDS_ONLY = MODE == 1 || MODE == 2 || MODE == 4 || MODE == 6 || MODE == 7 || MODE == 8 || MODE == 10 || MODE == 12 || MODE == 13,
                 SMALL = MODE == 2 || MODE == 7 || MODE == 8 || MODE == 13,
                 TINY = MODE == 7 || MODE == 8 || MODE == 13, PACKED = MODE == 3 || MODE == 4,
                 WIDE = MODE == 5 || MODE == 6 || MODE == 9 || MODE == 10, HYBRID = MODE == 5 || MODE == 9 || MODE == 11,
                 SLOT128 = MODE == 9 || MODE == 10, SLOTS = MODE == 11 || MODE == 12,
                 T6 = MODE == 13;  // the scene in LDS twice per CU, six waves per SIMD (see the mode list above)
  // leaf references: bit 31 leaf, bit 30 sphere, then count - 1 and the first primitive (rtgpu_format.h); the slot tree keeps
  // two bits of count and 28 of slot.  MODE 13 turns the references of its LDS copy into the sign-extended 16-bit form
  // (tiny_ref below): bit 15 leaf, bit 14 sphere, count - 1 in bits 11-13, the first primitive (< 2048) below
  constexpr uint32_t CNT_SHIFT = T6 ? 11u : SLOTS ? 28u : 27u, CNT_MASK = SLOTS ? 3u : 7u, FIRST_MASK = (1u << CNT_SHIFT) - 1u,
                     REF_SPH = T6 ? (1u << 14) : RTGPU_REF_SPHERE;
  constexpr int R = MODE == 8 ? 2 : 1;                  // rays a lane traverses in lock step
  // (2b) below: at the end of the launch idle lanes trace later samples of their wave's unfinished items.  The item's chunk
  // shares its register with that bookkeeping: chunk (of the round: below 64) | samples lent << 16 | helper flags
  // Trees in memory only: C5's frame -3.5 %, S2 level 8 at 4 spp -3.5 %, C3 unchanged; for the scenes in LDS the handful of
  // instructions it adds to every outer iteration cost more (C4's frame +1.5 %) than the shorter tail returns (C2 +-0).
  constexpr bool SHARE = R == 1 && !SMALL && RTGPU_TAIL_SHARE != 0;
  constexpr uint32_t kLentShift = 16u, kHelping = 1u << 24, kFresh = 1u << 25, kPending = 1u << 26;
  constexpr int BLOCK = MODE == 8 ? kBlock2 : T6 ? kBlock6 : SLOTS ? kBlockSlots : kBlock;   // threads per block
  constexpr int STRIDE = BLOCK * R;                     // words between consecutive entries of one stack
#ifdef RTGPU_NO_HIT_NORMAL  // experiments: shading fetches the triangle again
  constexpr bool KEEP_N = false;
#else
  constexpr bool KEEP_N = !SMALL;
#endif
#ifndef RTGPU_TINY_TRIS
#define RTGPU_TINY_TRIS 1
#endif
  constexpr int TRIS = TINY ? RTGPU_TINY_TRIS : SMALL ? 1 : kTriStep;  // triangles per leaf step (two in the LDS mode measured C2 1.4 % slower)
  // LDS: [traversal stacks: stack_lds x R x BLOCK u32][node cache: 4 planes x n_cached float4]
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint32_t *const lds_stack = smem;
  // explicit LDS address space: otherwise hipcc folds the "cached or not" choice
  // of the node fetch into ONE flat load of a selected generic pointer, and the
  // LDS reads go through the texture addresser like global ones
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) f4 lds_f4;
  // Slot-tree modes: the item's running sum and the path's throughput live in LDS behind the stacks (six words per lane, touched
  // by the shading pass and when the item ends), not in registers: six of the registers that stood between these modes and a
  // fifth wave per SIMD
  constexpr bool SUM_LDS = SLOTS, DIET = SLOTS || T6;
  typedef __attribute__((address_space(3))) float lds_f32;
  // MODE 13: [child references: kTinyNodes x 8 B][stacks of 16-bit entries][three node planes, spheres, ...].  The references sit
  // at the very start of the block's LDS and an inner node's reference IS the byte offset of its pair there (node x 8): the
  // ds_read of the pair needs no address arithmetic, the node planes one v_lshl_add (offset x 2 + base of the planes)
  constexpr uint32_t kRefWords = T6 ? (uint32_t)kTinyNodes * 2u : 0u;
  const uint32_t stack_words = kRefWords + (T6 ? P.stack_lds * (STRIDE / 2) : P.stack_lds * STRIDE);
  lds_f32 *const lds_sum = (lds_f32 *)(smem + stack_words) + threadIdx.x;
  lds_f32 *const lds_thr = lds_sum + 3 * BLOCK;
  lds_f4 *const node_cache = (lds_f4 *)(smem + stack_words + (SUM_LDS ? kPathLdsWords * BLOCK : 0));
  auto thr_get = [&](const V3 &reg) __attribute__((always_inline)) {
    return SUM_LDS ? mk(lds_thr[0], lds_thr[BLOCK], lds_thr[2 * BLOCK]) : reg;
  };
  auto thr_set = [&](V3 &reg, V3 v) __attribute__((always_inline)) {
    if (SUM_LDS) { lds_thr[0] = v.x; lds_thr[BLOCK] = v.y; lds_thr[2 * BLOCK] = v.z; }
    else reg = v;
  };
  const unsigned tid = threadIdx.x;
  const unsigned lane = lane_id();

  // The kernel is at the scalar-register limit (106), and what does not fit is spilled to lanes of a vector register:
  // a v_readlane (+ its wait states) wherever such a value is used - 48 of them in the loop of the scene-in-LDS kernel,
  // and the time follows their number (11 more cost C2 2.4 %).  So the parameters that only the COLD sections read -
  // camera, sky, frame and tile geometry, output pointers - are not kept in registers across the loop: those sections
  // read them from the kernel-argument segment where they use them (s_load through the scalar cache).  The empty asm
  // keeps the compiler from hoisting the loads back out of the loop.
  typedef const __attribute__((address_space(4))) KParams *KArgs;
  auto kargs = [&]() __attribute__((always_inline)) {
    KArgs k = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return k;
  };
  // The kernels of the scenes in LDS need fewer scalar registers (no scene pointers in the loop) and run their cold
  // sections more often per unit of time: camera and sky stay in registers there (74 + 18 scalar registers, no spill).
  constexpr bool HOIST_CAM = SMALL;
  const V3 cam_o_h = ld3(P.cam_origin), cam_ll_h = ld3(P.cam_ll), cam_hor_h = ld3(P.cam_hor), cam_ver_h = ld3(P.cam_ver);
  const V3 sky_b_h = ld3(P.sky_bottom), sky_t_h = ld3(P.sky_top);

  // state of one ray of the lane: the work item it belongs to, its path, and its traversal
  // (which survives an early exit from the traversal loop)
  typedef typename std::conditional<T6, int16_t, uint32_t>::type StackEntry;
  typedef Stack<LDS_ONLY, STRIDE, StackEntry> StackT;
  struct Ray {
    bool done, alive, have_pixel, pixel_valid;
    // the work item: pixel (DIET: x | y << 16 in px; both below 65536, checked at launch), its output slot, the sample being traced and the
    // end of the item's samples, the chunk, the segment of the path, the primary hit of the frame's first sample.
    // DIET (the slot-tree modes, which have to live in 96 registers for their fifth wave per SIMD): no sample_end (it follows
    // from the chunk where it is needed) and no first_id (written to its output word when it is found); sum and thr in LDS.
    uint32_t px, py, out_index, sample, sample_end, chunk, bounce, first_id;
    V3 sum, thr, ro, rd;  // (sum: in LDS for the slot-tree modes, SUM_LDS below)
    Rng rng;
    V3 new_dir;  // unnormalised direction of the segment about to start
    f2 inv_xy, inv_zx, inv_yz, noi_xy, noi_zx, noi_yz;
    // MODE 13 instead of those twelve: (1/d.x, 1/d.y), (-o.x/d.x, -o.y/d.y), (1/d.z, -o.z/d.z) - its LDS nodes are laid out so
    // that v_pk_fma_f32 picks the halves it needs with op_sel (node step below): six registers fewer across the loop
    f2 inv_a, noi_a, zz;
    float tbest;
    uint32_t cur;
    // closest hit so far: the leaf cursor at the time of the hit (type bit + slot of the
    // primitive in its leaf-ordered array), kDone = none
    uint32_t hit_ref;
    // material index of the closest TRIANGLE hit so far, taken from the record the test has in
    // registers: shading then fetches the material together with the triangle instead of after it
    uint32_t hit_mat;
    // trees in memory (not MODE 2, 7, 8): the closest triangle's e1 x e2 and id are kept too, so that shading does not
    // fetch the record again (a dependent round trip through the vector memory pipeline per shading pass)
    V3 hit_n;
    uint32_t hit_prim;
    StackT stk;
  };
  // (named objects, not an array: the compiler splits them into registers before it looks at any loop)
  Ray ray0, ray1;  // ray1: R == 2 only
  const uint32_t stack_slot = ((lane & 31u) << 1) | (lane >> 5);  // MODE 13: the lane's place among its wave's 64 stack columns
  auto init_ray = [&](Ray &S, int r) __attribute__((always_inline)) {
    S.done = S.alive = S.have_pixel = S.pixel_valid = false;
    S.px = S.py = S.out_index = S.sample = S.sample_end = S.chunk = S.bounce = 0u;
    S.first_id = RTGPU_NO_HIT;
    S.sum = mk(0.f, 0.f, 0.f); S.thr = mk(1.f, 1.f, 1.f); S.ro = mk(0.f, 0.f, 0.f); S.rd = mk(0.f, 0.f, 1.f);
    S.rng.s = 0u;
    S.new_dir = mk(0.f, 0.f, 1.f);
    S.inv_xy = mk2(1.f, 1.f); S.inv_zx = S.inv_xy; S.inv_yz = S.inv_xy;
    S.noi_xy = mk2(0.f, 0.f); S.noi_zx = S.noi_xy; S.noi_yz = S.noi_xy;
    S.inv_a = mk2(1.f, 1.f); S.noi_a = mk2(0.f, 0.f); S.zz = mk2(1.f, 0.f);
    S.tbest = kTMax;
    S.cur = kDone; S.hit_ref = kDone; S.hit_mat = 0u;
    S.hit_n = mk(0.f, 0.f, 1.f); S.hit_prim = 0u;
    S.stk.lds = lds_stack + r * BLOCK + tid;
    // MODE 13: 16-bit entries put two lanes into every bank; lanes l and l + 32 share a word (the LDS serves a wave's halves in
    // turn, so those two never meet), not lanes 2k and 2k + 1 (which conflict whenever their stacks differ in depth)
    S.stk.base_at = (typename StackT::lds_u32 *)(smem + kRefWords) + r * BLOCK + (T6 ? (tid & ~63u) + stack_slot : tid);
    S.stk.top_at = S.stk.base_at;
    S.stk.ovf_stride = gridDim.x * BLOCK;
    S.stk.ovf = P.ovf_stack + (size_t)blockIdx.x * BLOCK + tid;
    S.stk.sp = 1;
    S.stk.lds_entries = (int)P.stack_lds;
    S.stk.base_at[0] = (StackEntry)kDone;  // sentinel under every stack (never overwritten)
  };
  // MODE 13: LDS address of the first stack entry of this wave's lane 0 (wave-uniform)
  const uint32_t stack_wave_at = (uint32_t)__builtin_amdgcn_readfirstlane(
      (int)(uint32_t)(uintptr_t)((typename StackT::lds_u32 *)(smem + kRefWords) + (tid & ~63u)));
  init_ray(ray0, 0);
  init_ray(ray1, R - 1);  // (R == 1: the same words again; ray1 is then never used)
  unsigned long long n_paths = 0, n_rays = 0, n_nodes = 0, n_tris = 0, n_sphs = 0, n_scat = 0;
  unsigned long long w_outer = 0, w_shade = 0, l_shade = 0, w_gen = 0, w_node = 0, w_sph = 0, w_tri = 0,
                     w_trav = 0, l_trav = 0, l_node = 0, l_cached = 0, w_slow = 0, t_node = 0, t_leaf = 0;
  unsigned pool_next = 0, pool_end = 0;  // wave-local slice of the work queue (wave-uniform)
  // wave-uniform description of the group the pool was drawn from (see (2))
  uint32_t g_px = 0, g_py = 0, g_out = 0, g_row = 0, g_chunk = 0, g_sample = 0, g_sample_end = 0;
  bool queue_dry = false;                // wave-uniform: the global head has passed n_items
  unsigned tail_draw = 0;                // wave-uniform: 0 while the wave draws from the first head, then its draw size at the second
  unsigned g_end = 0;                    // wave-uniform: end of the part of the pool that g_* describes (a pool may cross into the next 8x8 block)
  unsigned n_draws = 0;                  // wave-uniform: draws from the first head so far
  const uint32_t t_begin = (uint32_t)__builtin_amdgcn_s_memrealtime();  // 100 MHz
  // LDS copy of the n_cached nodes nearest the root (the builder numbers them
  // breadth-first), as four planes [16-byte quarter][node]: lanes of a wave read
  // quarter j of DIFFERENT nodes, and in this layout consecutive nodes sit in
  // consecutive 16-byte bank groups (measured 56 cycles per 64-lane node fetch
  // vs 142 for node-major LDS and 181 through the vector L1, tools/ubench).
  const uint32_t n_cached = (SMALL || HYBRID) ? P.n_cached : 0u;
  constexpr uint32_t plane = (uint32_t)(TINY ? kTinyNodes : kSmallNodes);  // nodes per plane
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) u2 lds_u2;
  // MODE 13: offsets of the arrays behind the node planes (16-byte units): three planes (the child references are at the
  // start of the LDS), the other arrays follow as in kTiny*
  constexpr int SPH_AT = T6 ? kTinySphAt6 : kTinySphAt, TRI_AT = SPH_AT + kTinySphs, TRIN_AT = TRI_AT + 3 * kTinyTris,
                SPHMAT_AT = TRIN_AT + kTinyTris, TRIMAT_AT = SPHMAT_AT + 2 * kTinySphs, META_AT = TRIMAT_AT + 2 * kTinyTris;
  static_assert(!T6 || META_AT + kTinySphs / 2 == kTinyEnd6, "MODE 13: the host sizes the LDS by kTinyEnd6");
  lds_u2 *const ref_cache = (lds_u2 *)smem;  // MODE 13: LDS offset 0 (the kernel has no static LDS)
  if (SMALL && !T6)
    for (uint32_t i = tid; i < n_cached * 4u; i += BLOCK) {
      float4 v = P.nodes[i];
      f4 w = {v.x, v.y, v.z, v.w};
      node_cache[(i & 3u) * plane + (i >> 2)] = w;
    }
  if (T6) {
    // the node's twelve planes regrouped per axis: (lo0.x hi0.x lo0.y hi0.y) (lo0.z hi0.z lo1.x hi1.x) (lo1.y hi1.y lo1.z hi1.z),
    // so that a slab pair needs ONE component of 1/d and of -o/d in both halves of a v_pk_fma_f32 (op_sel), and the two
    // child references in the sign-extended 16-bit form the stacks of this mode hold
    auto tiny_ref = [](uint32_t r) {  // (inner node: its number x 8, the byte offset of its reference pair)
      return (int)r >= 0 ? r << 3 : (0xFFFF8000u | (((r >> 30) & 1u) << 14) | (((r >> 27) & 7u) << 11) | (r & 0x7FFu));
    };
    for (uint32_t i = tid; i < n_cached; i += BLOCK) {
      const float4 q0 = P.nodes[4u * i], q1 = P.nodes[4u * i + 1u], q2 = P.nodes[4u * i + 2u], q3 = P.nodes[4u * i + 3u];
      const f4 w0 = {q0.x, q0.w, q0.y, q1.x}, w1 = {q0.z, q1.y, q1.z, q2.y}, w2 = {q1.w, q2.z, q2.x, q2.w};
      const u2 c = {tiny_ref(__float_as_uint(q3.x)), tiny_ref(__float_as_uint(q3.y))};
      node_cache[i] = w0; node_cache[plane + i] = w1; node_cache[2u * plane + i] = w2;
      ref_cache[i] = c;
    }
  }
  // MODE 2: the leaf-ordered spheres behind the node planes when they fit: the sphere test is then an
  // LDS read like the node fetch instead of a round trip through the vector L1
  lds_f4 *const sph_cache = node_cache + (TINY ? (uint32_t)SPH_AT : 4 * plane);
  const bool sph_in_lds = TINY || (SMALL && P.sph_cached != 0u);
  if (sph_in_lds)
    for (uint32_t i = tid; i < P.sph_cached; i += BLOCK) {
      float4 v = P.sphs[i];
      f4 w = {v.x, v.y, v.z, TINY ? v.w * v.w : v.w};  // MODE 7: centre | r^2 (shading has 1/r in its own record)
      sph_cache[i] = w;
    }
  // MODE 7: and the triangles, and per primitive what shading reads (computed here once per block with the
  // operations shading would use: the same bits)
  lds_f4 *const tri_cache = node_cache + TRI_AT, *const trin_cache = node_cache + TRIN_AT,
         *const sphmat_cache = node_cache + SPHMAT_AT, *const trimat_cache = node_cache + TRIMAT_AT;
  lds_u2 *const meta_cache = (lds_u2 *)(node_cache + META_AT);
  if (TINY) {
    for (uint32_t i = tid; i < P.tiny_tris * 3u; i += BLOCK) {
      float4 v = P.tris[i];
      f4 w = {v.x, v.y, v.z, v.w};
      tri_cache[i] = w;
    }
    for (uint32_t i = tid; i < P.tiny_tris; i += BLOCK) {
      const float4 t0 = P.tris[3u * i], t1 = P.tris[3u * i + 1u], t2 = P.tris[3u * i + 2u];
      const V3 n = normalize(cross(mk(t1.x, t1.y, t1.z), mk(t2.x, t2.y, t2.z)));
      const f4 w = {n.x, n.y, n.z, t0.w};  // unit normal | id
      trin_cache[i] = w;
      const uint32_t m = __float_as_uint(t1.w);
      const float4 a = P.mats[2u * m], b = P.mats[2u * m + 1u];
      const f4 wa = {a.x, a.y, a.z, a.w}, wb = {b.x, b.y, b.z, b.w};
      trimat_cache[2u * i] = wa;
      trimat_cache[2u * i + 1u] = wb;
    }
    for (uint32_t i = tid; i < P.sph_cached; i += BLOCK) {
      const uint2 v = P.sph_meta[i];
      const u2 w = {v.x, __float_as_uint(rcp_exact(P.sphs[i].w))};  // id | 1/r
      meta_cache[i] = w;
      const float4 a = P.mats[2u * v.y], b = P.mats[2u * v.y + 1u];
      const f4 wa = {a.x, a.y, a.z, a.w}, wb = {b.x, b.y, b.z, b.w};
      sphmat_cache[2u * i] = wa;
      sphmat_cache[2u * i + 1u] = wb;
    }
  }
  // Trees in memory: a small material table (<= kLdsMats entries) behind the stacks, so that the shading pass waits
  // for no vector load at all (the triangle's normal and id are in registers, KEEP_N)
  lds_f4 *const mats_lds = node_cache + ((HYBRID && n_cached) ? 6 * kWideCached : 0);
  const bool mats_in_lds = !SMALL && P.mats_cached != 0u;
  if (mats_in_lds)
    for (uint32_t i = tid; i < P.mats_cached * 2u; i += BLOCK) {
      float4 v = P.mats[i];
      f4 w = {v.x, v.y, v.z, v.w};
      mats_lds[i] = w;
    }
  auto load_mat = [&](uint32_t m, float4 &m0, float4 &m1) __attribute__((always_inline)) {
    if (mats_in_lds) {
      const f4 a = mats_lds[m * 2u], b = mats_lds[m * 2u + 1u];
      m0 = make_float4(a.x, a.y, a.z, a.w); m1 = make_float4(b.x, b.y, b.z, b.w);
    } else {
      m0 = P.mats[m * 2u]; m1 = P.mats[m * 2u + 1u];
    }
  };
  if (HYBRID)  // quarter j of four-wide node k -> plane j, slot k
    for (uint32_t i = tid; i < n_cached * 6u; i += BLOCK) {
      float4 v = P.nodes[i];
      f4 w = {v.x, v.y, v.z, v.w};
      const uint32_t k = i / 6u;
      node_cache[(i - k * 6u) * (uint32_t)kWideCached + k] = w;
    }
  __syncthreads();
  const float tmin = P.tmin;
  const uint32_t seed_hash = pcg(P.seed);
  const char *const nodes_base = (const char *)P.nodes;
  const char *const tris_base = (const char *)P.tris;
  const char *const sphs_base = (const char *)P.sphs;

  unsigned long long t_start = 0, t_dry = 0;
  if (STATS == 2) t_start = __builtin_amdgcn_s_memrealtime();
  // profile build: shader-clock cycles this wave spent per section of the outer loop
  long long c_shade = 0, c_item = 0, c_gen = 0, c_setup = 0, c_trav = 0, c_mark = 0, c_nodes = 0, c_leaves = 0;
#define SECTION_END(acc) do { if (STATS == 2) { long long t_ = (long long)clock64(); acc += t_ - c_mark; c_mark = t_; } } while (0)
  if (STATS == 2) c_mark = (long long)clock64();

  unsigned long long n_after_dry = 0;
  for (;;) {
    if (STATS == 2) { wave_tick(lane, w_outer); if (queue_dry) n_after_dry++; }
    // sections (1)-(3) for one ray of the lane; true if the ray is in flight afterwards
    auto outer_sections = [&](Ray &S) __attribute__((always_inline)) {
    constexpr bool TSHARE = SHARE;
    // MODE 13 (LEFT): the samples the item still has to trace ride in the chunk's register above its eight bits (a round has at
    // most 64 chunks), so "item finished" is ONE compare; deriving the item's end from the chunk, as the other diet kernels do,
    // is two scalar loads, a wait and a quarter-rate multiply - twice per outer iteration
    constexpr bool LEFT = T6 && !SHARE;
    constexpr uint32_t kChunkMask = TSHARE ? 0xFFFFu : LEFT ? 0xFFu : 0xFFFFFFFFu;
    static_assert(kRoundChunks <= 256u, "LEFT: the chunk of a round in eight bits");
    // ---- (1) shade the rays whose traversal has finished (rows a6, a8) ----------
    if (S.alive && S.cur == kDone) {
      if (STATS == 2) { wave_tick(lane, w_shade); l_shade++; }
      V3 L = mk(0.f, 0.f, 0.f);
      bool path_end = false;
      const uint32_t hit_slot = S.hit_ref & FIRST_MASK;
      // the primary hit of the frame's first sample goes straight to its output word (once per pixel)
      auto write_first_id = [&](uint32_t id) __attribute__((always_inline)) {
        if (S.bounce == 0 && S.sample == 0) {
          if (DIET) {
            KArgs K = kargs();
            if (K->prim_id) K->prim_id[S.out_index] = id;
          } else {
            S.first_id = id;
          }
        }
      };
      if (S.hit_ref == kDone) {
        if (DIET) write_first_id(RTGPU_NO_HIT);
        float t = 0.5f * (S.rd.y + 1.0f);
        float a = 1.0f - t;
        V3 sb = sky_b_h, stp = sky_t_h;
        if (!HOIST_CAM) {
          KArgs K = kargs();
          sb = mk(K->sky_bottom[0], K->sky_bottom[1], K->sky_bottom[2]);
          stp = mk(K->sky_top[0], K->sky_top[1], K->sky_top[2]);
        }
        V3 sky = mk(sb.x * a + stp.x * t, sb.y * a + stp.y * t, sb.z * a + stp.z * t);
        L = thr_get(S.thr) * sky;
        path_end = true;
      } else {
        V3 p = S.ro + S.rd * S.tbest;
        V3 n;
        uint32_t prim;
        float4 m0, m1;
        if (S.hit_ref & REF_SPH) {
          float4 s;
          if (sph_in_lds) { const f4 w = sph_cache[hit_slot]; s = make_float4(w.x, w.y, w.z, w.w); }
          else if (SLOTS) s = *(const float4 *)(nodes_base + (hit_slot << 6));
          else s = P.sphs[hit_slot];
          if (TINY) {
            const u2 w = meta_cache[hit_slot];
            const f4 a = sphmat_cache[hit_slot * 2u], b = sphmat_cache[hit_slot * 2u + 1u];
            n = (p - mk(s.x, s.y, s.z)) * __uint_as_float(w.y);
            prim = w.x;
            m0 = make_float4(a.x, a.y, a.z, a.w); m1 = make_float4(b.x, b.y, b.z, b.w);
          } else {
            const uint2 meta = SLOTS ? *(const uint2 *)(nodes_base + (hit_slot << 6) + 16u) : P.sph_meta[hit_slot];
            float inv_r = rcp_exact(s.w);
            n = (p - mk(s.x, s.y, s.z)) * inv_r;
            prim = meta.x;
            load_mat(meta.y, m0, m1);
          }
        } else {
          // the material index came with the hit: its two loads go out with the triangle's
          if (TINY) {
            const f4 w = trin_cache[hit_slot];
            const f4 d = trimat_cache[hit_slot * 2u], e = trimat_cache[hit_slot * 2u + 1u];
            n = mk(w.x, w.y, w.z);
            prim = __float_as_uint(w.w);
            m0 = make_float4(d.x, d.y, d.z, d.w); m1 = make_float4(e.x, e.y, e.z, e.w);
          } else if (!KEEP_N) {
            const float4 *tp = SLOTS ? (const float4 *)(nodes_base + (hit_slot << 6)) : P.tris + (size_t)hit_slot * (sizeof(rtgpu_tri) / 16);
            const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
            load_mat(S.hit_mat, m0, m1);
            n = normalize(cross(mk(t1.x, t1.y, t1.z), mk(t2.x, t2.y, t2.z)));
            prim = __float_as_uint(t0.w);
          } else {
            load_mat(S.hit_mat, m0, m1);
            n = normalize(S.hit_n);
            prim = S.hit_prim;
          }
        }
        write_first_id(prim);
        float dn = dot(S.rd, n);
        bool front = dn < 0.0f;
        if (!front) n = neg(n);

        V3 albedo = mk(m0.x, m0.y, m0.z);
        float param = m0.w;
        uint32_t mtype = __float_as_uint(m1.x);
        if (STATS) n_scat++;
        // The three scatter functions share their expensive pieces, so a wave
        // that holds all three materials runs each piece once: one unit-vector
        // draw (lambertian, fuzzy metal), one final normalisation.  Per lane the
        // operations and the RNG draws are exactly the oracle's.
        const bool is_lamb = mtype == RTGPU_MAT_LAMBERTIAN, is_metal = mtype == RTGPU_MAT_METAL;
        V3 ruv = mk(0.f, 0.f, 0.f);
        if (is_lamb || (is_metal && param > 0.0f)) ruv = random_unit_vector(S.rng);
        V3 sd;
        bool absorbed = false;
        if (is_lamb) {
          sd = n + ruv;
          if (__builtin_fabsf(sd.x) < 1e-8f && __builtin_fabsf(sd.y) < 1e-8f && __builtin_fabsf(sd.z) < 1e-8f)
            sd = n;
        } else if (is_metal) {
          sd = reflect(S.rd, n);  // unit to rounding: rd and n are
          if (param > 0.0f) sd = sd + ruv * param;
          absorbed = !(dot(sd, n) > 0.0f);
        } else if (mtype == RTGPU_MAT_EMISSIVE) {
          L = thr_get(S.thr) * (albedo * param);  // a light ends the path with its radiance, from either face
          sd = n;
          absorbed = true;
        } else {
          // 1/ior and Schlick's r0 per face come precomputed with the flattened
          // material (bvh_build.cpp): the same f32 operations, done once on the host
          float ri = front ? m1.y : param;
          float cos_t = fmin_(-dot(S.rd, n), 1.0f);
          float sin_t = sqrt_exact(1.0f - cos_t * cos_t);
          bool cannot = ri * sin_t > 1.0f;
          float r0 = front ? m1.z : m1.w;
          float xx = 1.0f - cos_t;
          float x2 = xx * xx;
          float x5 = x2 * x2 * xx;
          float refl = r0 + (1.0f - r0) * x5;
          float u = S.rng.u01();
          if (cannot || refl > u) {
            sd = reflect(S.rd, n);
          } else {
            V3 perp = (S.rd + n * cos_t) * ri;
            float par = -sqrt_exact(__builtin_fabsf(1.0f - dot(perp, perp)));
            sd = perp + n * par;
          }
        }
        if (absorbed) {
          path_end = true;
        } else {
          thr_set(S.thr, thr_get(S.thr) * albedo);
          S.ro = p;
          S.new_dir = sd;  // normalised in (3), together with fresh camera rays
          S.bounce++;
          if (S.bounce >= P.max_depth) path_end = true;  // depth exhausted: L stays 0
        }
      }
      if (path_end) {
        if (SUM_LDS) { lds_sum[0] = lds_sum[0] + L.x; lds_sum[BLOCK] = lds_sum[BLOCK] + L.y; lds_sum[2 * BLOCK] = lds_sum[2 * BLOCK] + L.z; }
        else S.sum = S.sum + L;
        S.sample++;
        if (LEFT) S.chunk -= 256u;
        S.alive = false;
      }
    }

    SECTION_END(c_shade);
    // ---- (2) lanes without a path in flight: next sample, or next pixel ---------
    // last sample of the item + 1: the end of its chunk, or of the frame (an out-of-image pixel starts there)
    auto sample_end_of = [&](const Ray &S_) __attribute__((always_inline)) {
      if (!DIET) return S_.sample_end;
      // (the slot-tree kernels have the two scalar registers to keep these in: asked twice per outer iteration, a load from the
      // kernel-argument segment and its wait each time cost more than they)
      const uint32_t e = ((S_.chunk & kChunkMask) + 1u) * P.chunk_samples;
      return e < P.spp ? e : P.spp;
    };
    if (TSHARE && queue_dry) S.chunk &= ~kPending;
    auto item_finished = [&](const Ray &S_) __attribute__((always_inline)) {
      return LEFT ? S_.chunk < 256u : S_.sample >= sample_end_of(S_);
    };
    bool need_pixel = !S.alive && !S.done && (!S.have_pixel || item_finished(S));
    if (need_pixel && S.have_pixel) {
      if (S.pixel_valid) {
        KArgs K = kargs();
        const V3 sum = SUM_LDS ? mk(lds_sum[0], lds_sum[BLOCK], lds_sum[2 * BLOCK]) : S.sum;
        if (K->n_chunks > 1u) {
          K->planes[(size_t)(S.chunk & kChunkMask) * K->plane_stride + S.out_index] = make_float4(sum.x, sum.y, sum.z, 0.0f);
        } else {
          float4 o = make_float4(sum.x, sum.y, sum.z, (float)K->spp);
          if (K->flags & RTGPU_FLAG_ACCUMULATE) {
            float4 in = K->accum[S.out_index];
            o.x = in.x + o.x; o.y = in.y + o.y; o.z = in.z + o.z; o.w = in.w + o.w;
          }
          K->accum[S.out_index] = o;
        }
        if (!DIET && K->prim_id && (S.chunk & kChunkMask) == 0u) K->prim_id[S.out_index] = S.first_id;
      }
      S.have_pixel = false;
    }
    // Work queue: the global head is one word, and a returning atomic on one word
    // saturates near 88 per microsecond chip-wide (MI355X_MICROARCH.md, row
    // "dequeue") - one atomic per finished pixel made the queue, not the tracing,
    // the bottleneck of the 1080p frame.  So a wave draws a GROUP of kChunk items
    // per atomic: the 64 pixels of one 8x8 block for one chunk of samples.  All
    // that differs between the items of a group is the pixel's place in the block,
    // so the item -> (tile, block, chunk, sample range, output slot) arithmetic is
    // done once per group on wave-uniform values (g_*, scalar registers) and a lane
    // that takes an item only adds its (x, y) in the block.  Needy lanes take items
    // by their rank among the needy lanes (ballot + mbcnt prefix); when the group
    // runs out mid-wave the loop below goes round once more with a fresh group.
    unsigned long long need_mask = __ballot(need_pixel);
    while (need_mask != 0ull) {  // wave-uniform: every round hands out items or ends the loop
      if (pool_next == g_end) {  // wave-uniform: the next 8x8 block of the pool, after the next draw if the pool is empty
        KArgs K = kargs();
        const uint32_t ts = 1u << K->ts_log2;
        const uint32_t blocks_per_row_log2 = K->ts_log2 - 3u;  // 8x8 pixel blocks per tile row
        // once a wave has seen the queue dry it never asks again: thousands of
        // waves polling a dry head with one atomic per iteration saturate the
        // atomic unit and stretch the tail of the launch
        if (pool_next == pool_end) {
        unsigned nb = K->n_items, draw = kChunk;
        if (!queue_dry) {
          const unsigned leader = (unsigned)__ffsll((long long)need_mask) - 1u;
          unsigned long long fetched = ~0ull;
          if (tail_draw == 0u) {
            if (lane == leader) fetched = atomicAdd(&K->counters[0], (unsigned long long)kChunk);
            fetched = __shfl(fetched, (int)leader, 64);
            if (__builtin_amdgcn_readfirstlane((int)(fetched >= (unsigned long long)K->n_big)) != 0) {
              // the first head has passed n_big: from here on the second head, in draws sized by this wave's time per draw so far
              const float dt = (float)((uint32_t)__builtin_amdgcn_s_memrealtime() - t_begin), n = (float)n_draws;
              const bool d16 = dt >= (float)K->tail_thr16 * n, d32 = dt >= (float)K->tail_thr32 * n;
              tail_draw = (unsigned)__builtin_amdgcn_readfirstlane(d16 ? 16 : d32 ? 32 : 64);
            }
            n_draws++;
          }
          if (tail_draw != 0u) {  // (also in the iteration that found the first head past n_big)
            if (lane == leader) fetched = atomicAdd(&K->counters[kTailHead], (unsigned long long)tail_draw);
            fetched = __shfl(fetched, (int)leader, 64) + (unsigned long long)K->n_big;
            draw = tail_draw;
          }
          // a head beyond n_items means the queue is dry; clamp so u32 math cannot wrap
          nb = fetched >= (unsigned long long)K->n_items ? K->n_items : (unsigned)fetched;
          nb = (unsigned)__builtin_amdgcn_readfirstlane((int)nb);
        }
        if (nb >= K->n_items) {
          queue_dry = true;
          if (STATS == 2 && t_dry == 0) t_dry = __builtin_amdgcn_s_memrealtime();
          if (need_pixel) S.done = true;
          break;
        }
        pool_next = nb;
        // (n_items and n_big are multiples of kChunk; draws of different sizes share the second head, so a draw may start
        // anywhere on a multiple of 16, cross into the next 8x8 block and reach past the end of the queue)
        pool_end = nb + draw < K->n_items ? nb + draw : K->n_items;
        }
        g_end = (pool_next | 63u) + 1u < pool_end ? (pool_next | 63u) + 1u : pool_end;
        // group -> (64-pixel group, chunk of samples): groups x chunks, pixel-group-major
        uint32_t grp = pool_next >> 6, pix_grp = grp, chunk_u = 0u;
        if (K->n_chunks > 1u) {
          if (K->groups_small) {  // float estimate, exact below 2^22, corrected by at most one
            pix_grp = (uint32_t)((float)grp * K->inv_n_chunks);
            int rem = (int)(grp - pix_grp * K->n_chunks);
            pix_grp += (rem >= (int)K->n_chunks) ? 1u : 0u;
            pix_grp -= (rem < 0) ? 1u : 0u;
          } else {
            pix_grp = grp / K->n_chunks;
          }
          pix_grp = (uint32_t)__builtin_amdgcn_readfirstlane((int)pix_grp);
          chunk_u = grp - pix_grp * K->n_chunks;
        }
        // pixel group -> (local tile j, 8x8 block in the tile)
        const uint32_t it = pix_grp << 6;
        const uint32_t k = it & (ts * ts - 1u), j = it >> (2u * K->ts_log2);
        const uint32_t blk = k >> 6;
        const uint32_t tx = (blk & ((1u << blocks_per_row_log2) - 1u)) << 3;
        const uint32_t ty = (blk >> blocks_per_row_log2) << 3;
        const uint32_t tile = K->rank + j * K->world;
        // tile / tiles_x without the integer division: a float estimate (exact
        // operands below 2^22) corrected by at most one
        uint32_t tile_y;
        if (K->tiles_small) {
          tile_y = (uint32_t)((float)tile * K->inv_tiles_x);
          int rem = (int)(tile - tile_y * K->tiles_x);
          tile_y += (rem >= (int)K->tiles_x) ? 1u : 0u;
          tile_y -= (rem < 0) ? 1u : 0u;
        } else {
          tile_y = tile / K->tiles_x;
        }
        tile_y = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile_y);
        const uint32_t tile_x = tile - tile_y * K->tiles_x;
        g_px = (tile_x << K->ts_log2) + tx;
        g_py = (tile_y << K->ts_log2) + ty;
        g_row = (K->flags & RTGPU_FLAG_COMPACT) ? ts : K->width;  // output slots per image row
        g_out = (K->flags & RTGPU_FLAG_COMPACT) ? (j << (2u * K->ts_log2)) + (ty << K->ts_log2) + tx
                                                : g_py * K->width + g_px;
        g_chunk = chunk_u;
        g_sample = chunk_u * K->chunk_samples;
        g_sample_end = g_sample + K->chunk_samples < K->spp ? g_sample + K->chunk_samples : K->spp;
      }
      const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need_mask >> 32),
                                                      __builtin_amdgcn_mbcnt_lo((unsigned)need_mask, 0u));
      const unsigned cnt = (unsigned)__builtin_popcountll(need_mask);
      const unsigned avail = g_end - pool_next;  // wave-uniform, >= 1
      if (need_pixel && rank < avail) {
        const uint32_t within = (pool_next + rank) & 63u, wx = within & 7u, wy = within >> 3;
        const uint32_t px = g_px + wx, py = g_py + wy;
        S.px = DIET ? px | (py << 16) : px;
        S.py = py;
        S.pixel_valid = px < P.width && py < P.height;
        S.out_index = g_out + __umul24(wy, g_row) + wx;  // both below 2^24 (width <= 65536): v_mad_u32_u24 is full rate
        S.chunk = g_chunk;
        if (LEFT && S.pixel_valid) S.chunk = g_chunk + ((g_sample_end - g_sample) << 8);
        S.have_pixel = true;
        S.sample = S.pixel_valid ? g_sample : g_sample_end;  // out-of-image pixel: nothing to trace
        S.sample_end = g_sample_end;
        S.first_id = RTGPU_NO_HIT;
        if (SUM_LDS) { lds_sum[0] = 0.f; lds_sum[BLOCK] = 0.f; lds_sum[2 * BLOCK] = 0.f; }
        else S.sum = mk(0.f, 0.f, 0.f);
        need_pixel = false;
      }
      pool_next += cnt < avail ? cnt : avail;
      need_mask = __ballot(need_pixel);
    }
    // ---- (2b) the end of the launch: idle lanes trace LATER SAMPLES of their wave's unfinished items ----
    // Once the queue is dry a lane that finishes its item has nothing left to take, and the wave lasts as long as its slowest
    // item - up to chunk_samples paths one after the other (measured on C2: 54 outer iterations after the queue ran dry at
    // the median, of 15 per item on average; a fifth of the launch with ever fewer lanes at work).  The samples of an item are
    // independent paths; only their SUM has an order.  So an idle lane (the helper) takes one later sample of another lane's
    // item (the owner): pixel and sample index are all a path's random stream is keyed on.  It traces it into a sum of its own
    // that starts at zero, and the owner adds that sum where the sample's own radiance would have been added: when every
    // sample before it has been added.  0 + L == L bit for bit (and a sum is never -0), so the frame is the same frame.
    // Everything stays inside the wave: lanes find each other with ballots and v_readlane; no memory, no other wave.
    if (TSHARE && queue_dry) {
      const bool helping = (S.chunk & kHelping) != 0u;
      // (i) finished helpers hand over, in lane order; an owner takes a sample only when it is the next of its item
      unsigned long long hold = __ballot(helping && !S.alive && !(S.chunk & kFresh));
      while (hold != 0ull) {  // wave-uniform
        const int h = __builtin_ctzll(hold);
        hold &= hold - 1ull;
        const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)S.out_index, h);             // the owner's lane
        const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)S.sample, h) - 1u;            // the sample it traced
        V3 part;
        if (SUM_LDS) {
          const int d = h - (int)lane;  // the helper's words, seen from this lane's
          part = mk(lds_sum[d], lds_sum[BLOCK + d], lds_sum[2 * BLOCK + d]);
        } else {
          part = mk(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, S.sum.x), h)),
                    __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, S.sum.y), h)),
                    __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, S.sum.z), h)));
        }
        const bool take = lane == o && !S.alive && !helping && S.sample == j;
        if (take) {
          if (SUM_LDS) { lds_sum[0] = lds_sum[0] + part.x; lds_sum[BLOCK] = lds_sum[BLOCK] + part.y; lds_sum[2 * BLOCK] = lds_sum[2 * BLOCK] + part.z; }
          else S.sum = S.sum + part;
          S.sample++;
          S.chunk -= 1u << kLentShift;
          S.chunk |= kPending;  // (keeps the wave going until the top of (2) has seen the item, which may be complete now)
        }
        if (__ballot(take) != 0ull && lane == (unsigned)h) S.chunk = 0u;  // the helper is idle again
      }
      // (that may have completed an item: its lane writes it out in the next iteration - there is one: kPending)
      // (ii) idle lanes take the next samples nobody traces yet, one owner after the other so that they spread
      unsigned long long idle = __ballot(S.done && !(S.chunk & kHelping));
      if (idle != 0ull) {
        for (;;) {  // wave-uniform
          const uint32_t lent = (S.chunk >> kLentShift) & 0xFFu;
          const uint32_t next = S.sample + lent + (S.alive ? 1u : 0u);  // (an owner that waits has lent S.sample itself)
          const bool can_lend = !S.done && S.have_pixel && S.pixel_valid && (S.alive || lent != 0u) && lent < 255u &&
                                next < sample_end_of(S);
          unsigned long long own = __ballot(can_lend);
          if (own == 0ull) break;
          while (own != 0ull && idle != 0ull) {
            const int o = __builtin_ctzll(own), h = __builtin_ctzll(idle);
            own &= own - 1ull;
            idle &= idle - 1ull;
            const uint32_t px_o = (uint32_t)__builtin_amdgcn_readlane((int)S.px, o), py_o = (uint32_t)__builtin_amdgcn_readlane((int)S.py, o);
            const uint32_t smp = (uint32_t)__builtin_amdgcn_readlane((int)next, o);
            if (lane == (unsigned)h) {
              S.px = px_o; S.py = py_o; S.sample = smp;
              S.out_index = (uint32_t)o;
              S.chunk = kHelping | kFresh;
              S.have_pixel = false;
              if (SUM_LDS) { lds_sum[0] = 0.f; lds_sum[BLOCK] = 0.f; lds_sum[2 * BLOCK] = 0.f; }
              else S.sum = mk(0.f, 0.f, 0.f);
            }
            if (lane == (unsigned)o) S.chunk += 1u << kLentShift;
          }
          if (idle == 0ull) break;
        }
      }
    }
    SECTION_END(c_item);
    // (TSHARE: not a sample that a helper has - one compare: any of the bits above the chunk says no, and the only such bit a
    // lane with an item of its own can have besides its count of lent samples is kPending, cleared at the top of (2))
    const bool own_next = !S.alive && !S.done && S.have_pixel && !item_finished(S) &&
                          (!TSHARE || S.chunk <= kChunkMask);
    if (own_next || (TSHARE && queue_dry && !S.alive && (S.chunk & kFresh))) {
      if (TSHARE && queue_dry) S.chunk &= ~kFresh;
      // ---- camera ray (row a1) ----
      if (STATS == 2) wave_tick(lane, w_gen);
      KArgs K = kargs();
      V3 cam_o = cam_o_h, cam_ll = cam_ll_h, cam_hor = cam_hor_h, cam_ver = cam_ver_h;
      float inv_w = P.inv_w, inv_h = P.inv_h, lens_radius = P.lens_radius;
      uint32_t sample_offset = P.sample_offset, width = P.width;
      if (!HOIST_CAM) {
        cam_o = mk(K->cam_origin[0], K->cam_origin[1], K->cam_origin[2]); cam_ll = mk(K->cam_ll[0], K->cam_ll[1], K->cam_ll[2]);
        cam_hor = mk(K->cam_hor[0], K->cam_hor[1], K->cam_hor[2]); cam_ver = mk(K->cam_ver[0], K->cam_ver[1], K->cam_ver[2]);
        inv_w = K->inv_w; inv_h = K->inv_h; lens_radius = K->lens_radius; sample_offset = K->sample_offset; width = K->width;
      }
      // (the pixel's half of the key kept in a register per work item measured 0.8 % SLOWER on C2: a register more, a hash less)
      const uint32_t px = DIET ? S.px & 0xFFFFu : S.px, py = DIET ? S.px >> 16 : S.py;
      S.rng.start(pcg((sample_offset + S.sample) ^ pcg((__umul24(py, width) + px) ^ seed_hash)));
      float jx = S.rng.u01(), jy = S.rng.u01();
      float sx = ((float)px + jx) * inv_w;
      float sy = ((float)py + jy) * inv_h;
      V3 dir = ((cam_ll + cam_hor * sx) + cam_ver * sy) - cam_o;
      S.ro = cam_o;
      if (lens_radius > 0.0f) {
        float a = 0.0f, b = 0.0f;
        bool found = false;
#pragma unroll 1
        for (int i = 0; i < 8 && !found; i++) {
          float ta = S.rng.u01() * 2.0f - 1.0f;
          float tb = S.rng.u01() * 2.0f - 1.0f;
          if (ta * ta + tb * tb < 1.0f) { a = ta; b = tb; found = true; }
        }
        float ra = a * lens_radius, rb = b * lens_radius;
        V3 off = mk(K->cam_u[0], K->cam_u[1], K->cam_u[2]) * ra + mk(K->cam_v[0], K->cam_v[1], K->cam_v[2]) * rb;
        S.ro = cam_o + off;
        dir = dir - off;
      }
      S.new_dir = dir;  // normalised in (3)
      thr_set(S.thr, mk(1.f, 1.f, 1.f));
      S.bounce = 0;
      S.alive = true;
      n_paths++;
    }
    SECTION_END(c_gen);

    // ---- (3) start the traversal of every new segment --------------------------
    if (S.alive && S.cur == kDone) {
      S.rd = normalize(S.new_dir);
      // |clamp_dir(d)| lies in [1e-30, 1] for every finite direction: inside rcp_fast's verified range
      const V3 inv = mk(rcp_fast(clamp_dir(S.rd.x)), rcp_fast(clamp_dir(S.rd.y)), rcp_fast(clamp_dir(S.rd.z)));
      const V3 noi = mk(-(S.ro.x * inv.x), -(S.ro.y * inv.y), -(S.ro.z * inv.z));
      if (T6) {
        S.inv_a = mk2(inv.x, inv.y); S.noi_a = mk2(noi.x, noi.y); S.zz = mk2(inv.z, noi.z);
      } else {
        S.inv_xy = mk2(inv.x, inv.y); S.inv_zx = mk2(inv.z, inv.x); S.inv_yz = mk2(inv.y, inv.z);
        S.noi_xy = mk2(noi.x, noi.y); S.noi_zx = mk2(noi.z, noi.x); S.noi_yz = mk2(noi.y, noi.z);
      }
      S.tbest = kTMax;
      S.hit_ref = kDone;
      S.cur = 0;  // root
      if (T6) {
        // the stack's first entry from the wave's (uniform) offset and the lane, not from a register kept across the loop:
        // this kernel has none to spare (the empty asm keeps the sum from being hoisted out of the loop)
        uint32_t wave_at = stack_wave_at;
        asm volatile("" : "+s"(wave_at));
        S.stk.top_at = (typename StackT::lds_u32 *)(uintptr_t)(wave_at + (((lane & 31u) << 1) | (lane >> 5)) * (uint32_t)sizeof(StackEntry));
      } else {
        S.stk.reset();
      }
      n_rays++;
    }
    SECTION_END(c_setup);
    // (SHARE: a helper that holds a finished sample, an owner that waits for one, and an item that a handed-over sample has
    // just completed - it is written out at the top of (2) - keep the wave going)
    return S.alive || (TSHARE && S.chunk > kChunkMask);
    };
    bool any_alive = outer_sections(ray0);
    if (R == 2) any_alive = outer_sections(ray1) || any_alive;
    if (__ballot(any_alive) == 0ull) break;  // done implies !alive: the whole wave is dry

    // ---- (4) closest hit: ordered stack traversal (rows a3-a5) ------------------
    // Every lane walks its own ray(s) in the oracle's order; what the WAVE executes
    // next is put to a vote between the rays waiting at an inner node and the
    // rays waiting at a leaf: the node step runs while two thirds of the rays
    // wait for it, the leaf step otherwise (one popcount per step; measured best
    // of 1/3, 2/5, 1/2, 3/5, 2/3, 3/4, 4/5, 7/8), the other rays sit those steps out.  A plain while-while loop ran the node step for as few as
    // one lane (measured 13-27 % lane utilisation in the node step).
    // A leaf step tests ONE primitive and advances the reference in place
    // (first+1, count-1), so lanes with short leaves never wait for long ones.
    // The intersection tests are branch-free: a rejected candidate produces
    // values (possibly inf/NaN) that fail the final comparisons, exactly where
    // the oracle returns early.
    // With R = 2 a step first REQUESTS what both rays of the lane need and then tests them one after the
    // other: the second ray's LDS round trip runs under the first ray's arithmetic.
    // Scenes in LDS: waves in a traversal round issue ahead of waves in a shading pass (s_setprio 1 / 0).  The
    // round's steps are short dependent chains (LDS read -> slab test -> stack), the pass is long straight-line
    // code that fills any issue slot left: C2 3.74 -> 3.67 ms, C4's frame 212.2 -> 210.2 (2 and 3 measure the
    // same; the other way round 212.6).  Not on trees in memory: C3 18.22 -> 18.44 ms with it.
    if (SMALL) __builtin_amdgcn_s_setprio(RTGPU_PRIO_SMALL);
    // rays of the wave for which pred holds (summed over the rays of a lane)
#define RAYS_COUNT(expr) ([&]() __attribute__((always_inline)) { int n_; { const uint32_t c_ = ray0.cur; n_ = wave_count(expr); } \
                                                                 if (R == 2) { const uint32_t c_ = ray1.cur; n_ += wave_count(expr); } return n_; }())
    const int n_entry = RAYS_COUNT(c_ != kDone);
    for (;;) {
      // reference read as a signed int: inner node >= 0; leaf < -1; kDone == -1
      const int n_trav = RAYS_COUNT((int)c_ != -1);
      if (n_trav == 0 || (n_trav < REFILL * R && n_trav < n_entry)) break;  // wave-uniform
      int n_node = RAYS_COUNT((int)c_ >= 0);
      if (STATS == 2) { wave_tick(lane, w_trav); if ((int)ray0.cur != -1) l_trav++; if (R == 2 && (int)ray1.cur != -1) l_trav++; }
      long long c_phase = STATS == 2 ? (long long)clock64() : 0;
      // the two votes as ONE scalar compare per step: a node step needs n_node > 0 and 3 n_node >= 2 n_trav,
      // i.e. n_node >= ceil(2 n_trav / 3) (>= 1 since n_trav >= 1); a leaf step 3 n_leaf > n_trav, i.e.
      // n_leaf >= floor(n_trav / 3) + 1.
      const int node_quorum = (2 * n_trav + 2) / 3, leaf_quorum = n_trav / 3 + 1;
      if (n_node >= node_quorum) {
        do {
#pragma unroll
          for (int unroll_ = 0; unroll_ < (SMALL ? RTGPU_NODE_UNROLL : 1); unroll_++) {
          if (STATS == 2) { const int n_ = RAYS_COUNT((int)c_ != -1); if (lane == 0) t_node += (unsigned long long)n_; }
          if (WIDE && (int)ray0.cur >= 0) {
            Ray &S = ray0;
            uint32_t &cur = S.cur;
            const f2 inv_xy = S.inv_xy, inv_zx = S.inv_zx, inv_yz = S.inv_yz, noi_xy = S.noi_xy, noi_zx = S.noi_zx, noi_yz = S.noi_yz;
            const float tbest = S.tbest;
            StackT &stk = S.stk;
            if (STATS == 2) wave_tick(lane, w_node);
            // 96-byte node at cur * 96 = (cur * 3) << 5 (or in its 128-byte slot at cur << 7): six loads, four children
            uint32_t c3;
            asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(c3) : "v"(cur));
            const uint32_t node_off = SLOT128 ? cur << 7 : c3 << 5;
            // 128-byte slots: every lane of a load would read the SAME 16 bytes of its line (bank conflicts in the L1), so
            // odd nodes fetch their two halves (children 0,1 | children 2,3: identical code below) the other way round and
            // the child numbers in the sort keys are turned back (sw2): two bank positions per load instead of one
            // ... and nodes with bit 1 set sit 32 bytes into their slot (pad_nodes4_kernel): four positions in all, as many as the
            // 96-byte stride gives
            const uint32_t sw = SLOT128 ? (cur & 1u) : 0u, sw2 = sw << 1;
            const uint32_t slot_off = SLOT128 ? node_off + ((cur & 2u) << 4) : node_off;
            const uint32_t off_a = SLOT128 ? slot_off + sw * 48u : node_off, off_b = SLOT128 ? slot_off + 48u - sw * 48u : node_off + 48u;
            const float4 *n = (const float4 *)(nodes_base + off_a), *n2 = (const float4 *)(nodes_base + off_b);
            bool fast = true;
            const uint32_t top = HYBRID ? stk.peek_hybrid(fast) : stk.peek();
            float4 q0, q1, q2, q3, q4, q5;
            if (HYBRID) {
              // Lanes whose node is among the n_cached in LDS read it there (six ds_read_b128 with
              // immediate plane offsets), the others from memory, BOTH groups in flight together.
              // Written as one asm block because the two groups fill the same registers in
              // disjoint lanes: hipcc, which tracks pending loads per register and not per lane,
              // otherwise drains the global loads before it issues the LDS reads (a full memory
              // round trip added to every step of a wave that holds both kinds).
              f4 w0, w1, w2, w3, w4, w5;
              unsigned long long saved;
              const unsigned long long in_lds = __ballot(cur < n_cached);
              if (STATS == 2) { l_node++; if (cur < n_cached) l_cached++; if (!fast) wave_tick(lane, w_slow); }
              const uint32_t goff = off_a, goff2 = off_b;
              const uint32_t laddr = (uint32_t)(uintptr_t)(node_cache + cur);
              asm volatile(
                  "s_mov_b64 %[sv], exec\n\t"
                  "s_andn2_b64 exec, %[sv], %[m]\n\t"
                  "global_load_dwordx4 %[a0], %[go], %[nb]\n\t"
                  "global_load_dwordx4 %[a1], %[go], %[nb] offset:16\n\t"
                  "global_load_dwordx4 %[a2], %[go], %[nb] offset:32\n\t"
                  "global_load_dwordx4 %[a3], %[go2], %[nb]\n\t"
                  "global_load_dwordx4 %[a4], %[go2], %[nb] offset:16\n\t"
                  "global_load_dwordx4 %[a5], %[go2], %[nb] offset:32\n\t"
                  "s_and_b64 exec, %[sv], %[m]\n\t"
                  "ds_read_b128 %[a0], %[la]\n\t"
                  "ds_read_b128 %[a1], %[la] offset:12288\n\t"
                  "ds_read_b128 %[a2], %[la] offset:24576\n\t"
                  "ds_read_b128 %[a3], %[la] offset:36864\n\t"
                  "ds_read_b128 %[a4], %[la] offset:49152\n\t"
                  "ds_read_b128 %[a5], %[la] offset:61440\n\t"
                  "s_mov_b64 exec, %[sv]\n\t"
                  "s_waitcnt vmcnt(0) lgkmcnt(0)"
                  : [a0] "=&v"(w0), [a1] "=&v"(w1), [a2] "=&v"(w2), [a3] "=&v"(w3), [a4] "=&v"(w4), [a5] "=&v"(w5),
                    [sv] "=&s"(saved)
                  : [go] "v"(goff), [go2] "v"(goff2), [la] "v"(laddr), [nb] "s"(nodes_base), [m] "s"(in_lds)
                  : "memory");
              static_assert(kWideCached * 16 == 12288, "the plane offsets in the asm above are kWideCached * 16 apart");
              q0 = make_float4(w0.x, w0.y, w0.z, w0.w); q1 = make_float4(w1.x, w1.y, w1.z, w1.w);
              q2 = make_float4(w2.x, w2.y, w2.z, w2.w); q3 = make_float4(w3.x, w3.y, w3.z, w3.w);
              q4 = make_float4(w4.x, w4.y, w4.z, w4.w); q5 = make_float4(w5.x, w5.y, w5.z, w5.w);
            } else {
              q0 = n[0]; q1 = n[1]; q2 = n[2]; q3 = n2[0]; q4 = n2[1]; q5 = n2[2];
            }
            if (STATS) n_nodes++;
            // children 0,1 sit in q0..q2 and children 2,3 in q3..q5 exactly as the two
            // children of a BVH2 node sit in its first three quarters: the same six
            // v_pk_fma_f32 per pair of children
            float tn[4], tf[4];
#pragma unroll
            for (int h = 0; h < 2; h++) {
              const float4 a = h ? q3 : q0, b = h ? q4 : q1, c = h ? q5 : q2;
              f2 p0 = __builtin_elementwise_fma(mk2(a.x, a.y), inv_xy, noi_xy);
              f2 p1 = __builtin_elementwise_fma(mk2(a.z, a.w), inv_zx, noi_zx);
              f2 p2 = __builtin_elementwise_fma(mk2(b.x, b.y), inv_yz, noi_yz);
              f2 p3 = __builtin_elementwise_fma(mk2(b.z, b.w), inv_xy, noi_xy);
              f2 p4 = __builtin_elementwise_fma(mk2(c.x, c.y), inv_zx, noi_zx);
              f2 p5 = __builtin_elementwise_fma(mk2(c.z, c.w), inv_yz, noi_yz);
              tn[2 * h] = fmax_(fmax_(fmin_(p0.x, p1.y), fmin_(p0.y, p2.x)), vmax_(fmin_(p1.x, p2.y), tmin));
              tf[2 * h] = fmin_(fmin_(fmax_(p0.x, p1.y), fmax_(p0.y, p2.x)), vmin_(fmax_(p1.x, p2.y), tbest));
              tn[2 * h + 1] = fmax_(fmax_(fmin_(p3.x, p4.y), fmin_(p3.y, p5.x)), vmax_(fmin_(p4.x, p5.y), tmin));
              tf[2 * h + 1] = fmin_(fmin_(fmax_(p3.x, p4.y), fmax_(p3.y, p5.x)), vmin_(fmax_(p4.x, p5.y), tbest));
            }
            // references: byte j of child k = low byte of word 6k + j
            auto lowbytes = [](float w0, float w1, float w2, float w3) {
              const uint32_t lo = __builtin_amdgcn_perm(__float_as_uint(w1), __float_as_uint(w0), 0x0C0C0400u);
              const uint32_t hi = __builtin_amdgcn_perm(__float_as_uint(w3), __float_as_uint(w2), 0x0C0C0400u);
              return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
            };
            uint32_t r0 = lowbytes(q0.x, q0.y, q0.z, q0.w), r1 = lowbytes(q1.z, q1.w, q2.x, q2.y),
                     r2 = lowbytes(q3.x, q3.y, q3.z, q3.w), r3 = lowbytes(q4.z, q4.w, q5.x, q5.y);
            // keys: ((bits of tnear) & ~3) | child for a hit, all ones for a miss; tnear >= tmin >= 0
            const bool h0 = tn[0] <= tf[0], h1 = tn[1] <= tf[1], h2 = tn[2] <= tf[2], h3 = tn[3] <= tf[3];
            uint32_t k0 = h0 ? ((__float_as_uint(tn[0]) & 0x7FFFFFFCu) | (0u ^ sw2)) : 0xFFFFFFFFu;
            uint32_t k1 = h1 ? ((__float_as_uint(tn[1]) & 0x7FFFFFFCu) | (1u ^ sw2)) : 0xFFFFFFFFu;
            uint32_t k2 = h2 ? ((__float_as_uint(tn[2]) & 0x7FFFFFFCu) | (2u ^ sw2)) : 0xFFFFFFFFu;
            uint32_t k3 = h3 ? ((__float_as_uint(tn[3]) & 0x7FFFFFFCu) | (3u ^ sw2)) : 0xFFFFFFFFu;
            const int n_hit = (int)h0 + (int)h1 + (int)h2 + (int)h3;
            // five compare-exchanges on (key, reference) pairs: ascending keys
#define RTGPU_CX(ka, ra, kb, rb) do { const bool sw_ = kb < ka; const uint32_t tk_ = sw_ ? kb : ka, tr_ = sw_ ? rb : ra; \
                                       kb = sw_ ? ka : kb; rb = sw_ ? ra : rb; ka = tk_; ra = tr_; } while (0)
            RTGPU_CX(k0, r0, k1, r1);
            RTGPU_CX(k2, r2, k3, r3);
            RTGPU_CX(k0, r0, k2, r2);
            RTGPU_CX(k1, r1, k3, r3);
            RTGPU_CX(k1, r1, k2, r2);
#undef RTGPU_CX
            cur = HYBRID ? stk.step4_hybrid(fast, n_hit, r0, r1, r2, r3, top) : stk.step4(n_hit, r0, r1, r2, r3, top);
          }
          if (SLOTS && (int)ray0.cur >= 0) {
            Ray &S = ray0;
            uint32_t &cur = S.cur;
            StackT &stk = S.stk;
            if (STATS == 2) wave_tick(lane, w_node);
            // node slot at cur << 6: origin | family, then four children x three axes of (lo | hi) halves
            const uint4 *n = (const uint4 *)(nodes_base + (cur << 6));
            bool fast = true;
            const uint4 hd = n[0], pa = n[1], pb = n[2], pc = n[3];
            const uint32_t top = HYBRID ? stk.peek_hybrid(fast) : stk.peek();
            if (STATS == 2) { l_node++; if (!fast) wave_tick(lane, w_slow); }
            if (STATS) n_nodes++;
            // t = half * (1/d) + (origin * (1/d) - o/d): the second term once per node and axis
            const f2 adj_xy = __builtin_elementwise_fma(mk2(__uint_as_float(hd.x), __uint_as_float(hd.y)), S.inv_xy, S.noi_xy);
            const float adj_z = __builtin_fmaf(__uint_as_float(hd.z), S.inv_zx.x, S.noi_zx.x);
            const float ix = S.inv_xy.x, iy = S.inv_xy.y, iz = S.inv_zx.x, ax = adj_xy.x, ay = adj_xy.y, az = adj_z;
            const float tbest = S.tbest;
            auto lo_of = [](uint32_t w) { union { uint32_t u; _Float16 h[2]; } c; c.u = w; return (float)c.h[0]; };
            auto hi_of = [](uint32_t w) { union { uint32_t u; _Float16 h[2]; } c; c.u = w; return (float)c.h[1]; };
            const uint32_t pw[12] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w, pc.x, pc.y, pc.z, pc.w};
            float tn[4], tf[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const float x0 = __builtin_fmaf(lo_of(pw[3 * k]), ix, ax), x1 = __builtin_fmaf(hi_of(pw[3 * k]), ix, ax);
              const float y0 = __builtin_fmaf(lo_of(pw[3 * k + 1]), iy, ay), y1 = __builtin_fmaf(hi_of(pw[3 * k + 1]), iy, ay);
              const float z0 = __builtin_fmaf(lo_of(pw[3 * k + 2]), iz, az), z1 = __builtin_fmaf(hi_of(pw[3 * k + 2]), iz, az);
              tn[k] = fmax_(fmax_(fmin_(x0, x1), fmin_(y0, y1)), vmax_(fmin_(z0, z1), tmin));
              tf[k] = fmin_(fmin_(fmax_(x0, x1), fmax_(y0, y1)), vmin_(fmax_(z0, z1), tbest));
            }
            // slot references: family + (child byte turned so that its low nibble is the top of the word and its high nibble,
            // the child's place in the family, the bottom)
            const uint32_t family = hd.w & 0x00FFFFFFu;
            uint32_t r0 = family + (__builtin_amdgcn_alignbit(hd.x, hd.x, 4) & 0xF000000Fu);
            uint32_t r1 = family + (__builtin_amdgcn_alignbit(hd.y, hd.y, 4) & 0xF000000Fu);
            uint32_t r2 = family + (__builtin_amdgcn_alignbit(hd.z, hd.z, 4) & 0xF000000Fu);
            uint32_t r3 = family + (__builtin_amdgcn_alignbit(hd.w, hd.w, 28) & 0xF000000Fu);
            const bool h0 = tn[0] <= tf[0], h1 = tn[1] <= tf[1], h2 = tn[2] <= tf[2], h3 = tn[3] <= tf[3];
            uint32_t k0 = h0 ? ((__float_as_uint(tn[0]) & 0x7FFFFFFCu) | 0u) : 0xFFFFFFFFu;
            uint32_t k1 = h1 ? ((__float_as_uint(tn[1]) & 0x7FFFFFFCu) | 1u) : 0xFFFFFFFFu;
            uint32_t k2 = h2 ? ((__float_as_uint(tn[2]) & 0x7FFFFFFCu) | 2u) : 0xFFFFFFFFu;
            uint32_t k3 = h3 ? ((__float_as_uint(tn[3]) & 0x7FFFFFFCu) | 3u) : 0xFFFFFFFFu;
            const int n_hit = (int)h0 + (int)h1 + (int)h2 + (int)h3;
#define RTGPU_CX(ka, ra, kb, rb) do { const bool sw_ = kb < ka; const uint32_t tk_ = sw_ ? kb : ka, tr_ = sw_ ? rb : ra; \
                                       kb = sw_ ? ka : kb; rb = sw_ ? ra : rb; ka = tk_; ra = tr_; } while (0)
            RTGPU_CX(k0, r0, k1, r1);
            RTGPU_CX(k2, r2, k3, r3);
            RTGPU_CX(k0, r0, k2, r2);
            RTGPU_CX(k1, r1, k3, r3);
            RTGPU_CX(k1, r1, k2, r2);
#undef RTGPU_CX
            cur = HYBRID ? stk.step4_hybrid(fast, n_hit, r0, r1, r2, r3, top) : stk.step4(n_hit, r0, r1, r2, r3, top);
          }
          if (!WIDE && !SLOTS) {
            // BVH2 node step in two halves, so that with R = 2 both rays' fetches are requested before either is tested
            struct NodeData { float4 q0, q1, q2; uint32_t c0, c1, top; };
            NodeData nd0, nd1;
            auto node_fetch = [&](Ray &S, NodeData &D) __attribute__((always_inline)) {
              const uint32_t cur = S.cur;
              if (STATS == 2) wave_tick(lane, w_node);
              D.top = S.stk.peek();
              // MODE 0/1 (tree larger than the LDS cache) fetch every node through L1: a
              // partial cache measured no gain on the big scenes and costs a
              // test and two branches per step
              if (T6) {
                // cur = node x 8: the LDS address of the reference pair as it is, and half the node's offset in a plane
                typedef __attribute__((address_space(3))) char lds_char;
                lds_f4 *const nd = (lds_f4 *)((lds_char *)node_cache + (cur << 1));
                const f4 w0 = nd[0], w1 = nd[plane], w2 = nd[2u * plane];
                const u2 c = *(lds_u2 *)(uintptr_t)cur;
                D.q0 = make_float4(w0.x, w0.y, w0.z, w0.w);
                D.q1 = make_float4(w1.x, w1.y, w1.z, w1.w);
                D.q2 = make_float4(w2.x, w2.y, w2.z, w2.w);
                D.c0 = c.x; D.c1 = c.y;
              } else if (SMALL) {
                f4 w0 = node_cache[cur], w1 = node_cache[plane + cur], w2 = node_cache[2u * plane + cur],
                   w3 = node_cache[3u * plane + cur];
                D.q0 = make_float4(w0.x, w0.y, w0.z, w0.w);
                D.q1 = make_float4(w1.x, w1.y, w1.z, w1.w);
                D.q2 = make_float4(w2.x, w2.y, w2.z, w2.w);
                D.c0 = __float_as_uint(w3.x); D.c1 = __float_as_uint(w3.y);
              } else if (PACKED) {
                // 48-byte node at cur * 48 = (cur * 3) << 4; the child references are the low
                // bytes of q0's and q2's four planes, gathered with three v_perm_b32 each
                uint32_t c3;
                asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(c3) : "v"(cur));
                const float4 *n = (const float4 *)(nodes_base + (c3 << 4));
                D.q0 = n[0]; D.q1 = n[1]; D.q2 = n[2];
                const uint32_t a0 = __builtin_amdgcn_perm(__float_as_uint(D.q0.y), __float_as_uint(D.q0.x), 0x0C0C0400u);
                const uint32_t b0 = __builtin_amdgcn_perm(__float_as_uint(D.q0.w), __float_as_uint(D.q0.z), 0x0C0C0400u);
                D.c0 = __builtin_amdgcn_perm(b0, a0, 0x05040100u);
                const uint32_t a1 = __builtin_amdgcn_perm(__float_as_uint(D.q2.y), __float_as_uint(D.q2.x), 0x0C0C0400u);
                const uint32_t b1 = __builtin_amdgcn_perm(__float_as_uint(D.q2.w), __float_as_uint(D.q2.z), 0x0C0C0400u);
                D.c1 = __builtin_amdgcn_perm(b1, a1, 0x05040100u);
              } else {
                const float4 *n = (const float4 *)(nodes_base + (cur << 6));  // 32-bit offset: < 4 GiB of nodes
                D.q0 = n[0]; D.q1 = n[1]; D.q2 = n[2];
                const float4 q3 = n[3];
                D.c0 = __float_as_uint(q3.x); D.c1 = __float_as_uint(q3.y);
              }
              if (STATS) n_nodes++;
            };
            auto node_test = [&](Ray &S, const NodeData &D) __attribute__((always_inline)) {
              const float4 q0 = D.q0, q1 = D.q1, q2 = D.q2;
              // child 0: lo (q0.x q0.y q0.z) hi (q0.w q1.x q1.y); child 1: lo (q1.z q1.w q2.x) hi (q2.y q2.z q2.w)
              // the twelve slab fmas as six v_pk_fma_f32 (two IEEE fmas per instruction,
              // bit-identical to fmaf): the node's planes sit in memory in the pair order
              // (x,y)(z,x)(y,z)(x,y)(z,x)(y,z), matched by the pre-paired 1/d and -(o/d)
              float a0x, a0y, a0z, a1x, a1y, a1z, b0x, b0y, b0z, b1x, b1y, b1z;
              if (T6) {
                // planes per axis (see the LDS copy): (lo, hi) of one axis x that axis of 1/d in BOTH halves + that axis of
                // -o/d in both halves; the same twelve IEEE fmas
                // (one asm block: between separate asm statements hipcc puts an s_nop per dependence it cannot see into)
                f2 p0 = mk2(q0.x, q0.y), p1 = mk2(q0.z, q0.w), p2 = mk2(q1.x, q1.y), p3 = mk2(q1.z, q1.w), p4 = mk2(q2.x, q2.y),
                   p5 = mk2(q2.z, q2.w);
                asm("v_pk_fma_f32 %0, %0, %6, %7 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"   // x: the low halves of inv_a, noi_a
                    "v_pk_fma_f32 %1, %1, %6, %7 op_sel:[0,1,1] op_sel_hi:[1,1,1]\n\t"   // y: the high halves
                    "v_pk_fma_f32 %2, %2, %8, %8 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"   // z: 1/d.z low, -o.z/d.z high, of zz
                    "v_pk_fma_f32 %3, %3, %6, %7 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                    "v_pk_fma_f32 %4, %4, %6, %7 op_sel:[0,1,1] op_sel_hi:[1,1,1]\n\t"
                    "v_pk_fma_f32 %5, %5, %8, %8 op_sel:[0,0,1] op_sel_hi:[1,0,1]"
                    : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5) : "v"(S.inv_a), "v"(S.noi_a), "v"(S.zz));
                a0x = p0.x; a1x = p0.y; a0y = p1.x; a1y = p1.y; a0z = p2.x; a1z = p2.y;
                b0x = p3.x; b1x = p3.y; b0y = p4.x; b1y = p4.y; b0z = p5.x; b1z = p5.y;
              } else {
              f2 p0 = __builtin_elementwise_fma(mk2(q0.x, q0.y), S.inv_xy, S.noi_xy);  // lo0.x lo0.y
              f2 p1 = __builtin_elementwise_fma(mk2(q0.z, q0.w), S.inv_zx, S.noi_zx);  // lo0.z hi0.x
              f2 p2 = __builtin_elementwise_fma(mk2(q1.x, q1.y), S.inv_yz, S.noi_yz);  // hi0.y hi0.z
              f2 p3 = __builtin_elementwise_fma(mk2(q1.z, q1.w), S.inv_xy, S.noi_xy);  // lo1.x lo1.y
              f2 p4 = __builtin_elementwise_fma(mk2(q2.x, q2.y), S.inv_zx, S.noi_zx);  // lo1.z hi1.x
              f2 p5 = __builtin_elementwise_fma(mk2(q2.z, q2.w), S.inv_yz, S.noi_yz);  // hi1.y hi1.z
              a0x = p0.x; a0y = p0.y; a0z = p1.x; a1x = p1.y; a1y = p2.x; a1z = p2.y;
              b0x = p3.x; b0y = p3.y; b0z = p4.x; b1x = p4.y; b1y = p5.x; b1z = p5.y;
              }
              // vmax_/vmin_: the bare instruction for the two operands that hipcc cannot
              // prove canonical (tmin from the kernel arguments, tbest through the loop):
              // the builtin costs an extra v_max x,x per operand per step
              float tn0, tf0, tn1, tf1;
              if (T6) {
                // the asm's results are fma results (never signalling NaNs), which hipcc cannot know: its fminf / fmaxf would
                // canonicalise each of the twelve first (a v_max_f32 x, x apiece) - so the bare instructions, as vmax_ / vmin_
                float t1, t2, t3, t4;
                asm("v_min_f32 %0, %8, %9\n\t"
                    "v_min_f32 %4, %10, %11\n\t"
                    "v_min_f32 %5, %12, %13\n\t"
                    "v_min_f32 %2, %14, %15\n\t"
                    "v_min_f32 %6, %16, %17\n\t"
                    "v_min_f32 %7, %18, %19\n\t"
                    "v_max_f32 %5, %20, %5\n\t"
                    "v_max_f32 %7, %20, %7\n\t"
                    "v_max3_f32 %0, %0, %4, %5\n\t"
                    "v_max3_f32 %2, %2, %6, %7\n\t"
                    "v_max_f32 %1, %8, %9\n\t"
                    "v_max_f32 %4, %10, %11\n\t"
                    "v_max_f32 %5, %12, %13\n\t"
                    "v_max_f32 %3, %14, %15\n\t"
                    "v_max_f32 %6, %16, %17\n\t"
                    "v_max_f32 %7, %18, %19\n\t"
                    "v_min_f32 %5, %5, %21\n\t"
                    "v_min_f32 %7, %7, %21\n\t"
                    "v_min3_f32 %1, %1, %4, %5\n\t"
                    "v_min3_f32 %3, %3, %6, %7"
                    : "=&v"(tn0), "=&v"(tf0), "=&v"(tn1), "=&v"(tf1), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4)
                    : "v"(a0x), "v"(a1x), "v"(a0y), "v"(a1y), "v"(a0z), "v"(a1z), "v"(b0x), "v"(b1x), "v"(b0y), "v"(b1y),
                      "v"(b0z), "v"(b1z), "s"(tmin), "v"(S.tbest));
              } else {
                tn0 = fmax_(fmax_(fmin_(a0x, a1x), fmin_(a0y, a1y)), vmax_(fmin_(a0z, a1z), tmin));
                tf0 = fmin_(fmin_(fmax_(a0x, a1x), fmax_(a0y, a1y)), vmin_(fmax_(a0z, a1z), S.tbest));
                tn1 = fmax_(fmax_(fmin_(b0x, b1x), fmin_(b0y, b1y)), vmax_(fmin_(b0z, b1z), tmin));
                tf1 = fmin_(fmin_(fmax_(b0x, b1x), fmax_(b0y, b1y)), vmin_(fmax_(b0z, b1z), S.tbest));
              }
              // an empty child (scenes of < 2 primitives) has its box at +inf and can
              // never pass: tfar <= tbest <= FLT_MAX < tnear
              bool h0 = tn0 <= tf0;
              bool h1 = tn1 <= tf1;
              // both hit: nearer first (tie: child 0), the other is pushed
              bool swap = tn1 < tn0;
              uint32_t near_c = (h0 && !(h1 && swap)) ? D.c0 : D.c1;
              uint32_t far_c = swap ? D.c0 : D.c1;
              S.cur = S.stk.step(h0 && h1, !(h0 || h1), near_c, far_c, D.top);
            };
            if (R == 1) {
              if ((int)ray0.cur >= 0) { node_fetch(ray0, nd0); node_test(ray0, nd0); }
            } else {
              const bool at0 = (int)ray0.cur >= 0, at1 = (int)ray1.cur >= 0;
              if (at0) node_fetch(ray0, nd0);
              if (at1) node_fetch(ray1, nd1);
              if (at0) node_test(ray0, nd0);
              if (at1) node_test(ray1, nd1);
            }
          }
          }
          n_node = RAYS_COUNT((int)c_ >= 0);
        } while (n_node >= node_quorum);
        if (STATS == 2) c_nodes += (long long)clock64() - c_phase;
      } else {
        int n_leaf;
        do {
          if (STATS == 2) { const int n_ = RAYS_COUNT((int)c_ != -1); if (lane == 0) t_leaf += (unsigned long long)n_; }
          // leaf step in two halves like the node step
          struct LeafData { bool at_leaf, at_sph, at_tri, fast; uint32_t i, top, left, more; float4 s; float4 rec[TRIS][3]; };
          LeafData ld0, ld1;
          auto leaf_fetch = [&](Ray &S, LeafData &D) __attribute__((always_inline)) {
            const uint32_t cur = S.cur;
            D.at_leaf = (int)cur < -1;
            D.at_sph = D.at_leaf && (cur & REF_SPH);
            D.at_tri = D.at_leaf && !(cur & REF_SPH);
            D.i = cur & FIRST_MASK;
            D.fast = true;
            D.top = HYBRID ? S.stk.peek_hybrid(D.fast) : S.stk.peek();
            // a lane tests up to kTriStep of the triangles left in its leaf per step (all
            // records requested before the first test): same tests in the same order, fewer
            // dependent round trips
            D.left = (cur >> CNT_SHIFT) & CNT_MASK;                        // triangles after this one
            D.more = TRIS > 1 && D.at_tri ? (D.left < (uint32_t)(TRIS - 1) ? D.left : (uint32_t)(TRIS - 1)) : 0u;
          };
          auto sph_fetch = [&](Ray &S, LeafData &D) __attribute__((always_inline)) {
            if (D.at_sph) {
              if (sph_in_lds) { const f4 w = sph_cache[D.i]; D.s = make_float4(w.x, w.y, w.z, w.w); }
              else D.s = *(const float4 *)(SLOTS ? nodes_base + (D.i << 6) : sphs_base + (D.i << 4));
            }
          };
          auto sph_test = [&](Ray &S, LeafData &D) __attribute__((always_inline)) {
            if (D.at_sph) {
              const float4 s = D.s;
              if (STATS) n_sphs++;
              if (STATS == 2) wave_tick(lane, w_sph);
              V3 oc = S.ro - mk(s.x, s.y, s.z);
              float hb = dot(oc, S.rd);
              float cc = dot(oc, oc) - (TINY ? s.w : s.w * s.w);
              float disc = hb * hb - cc;
              float sq = sqrt_exact(disc);  // NaN for a negative discriminant: fails every test below
              float t = -hb - sq;
              bool ok = t > tmin && t < S.tbest;
              float t2 = -hb + sq;
              bool ok2 = t2 > tmin && t2 < S.tbest;
              t = ok ? t : t2;
              if (ok || ok2) { S.tbest = t; S.hit_ref = S.cur; }
            }
          };
          auto tri_fetch = [&](Ray &S, LeafData &D) __attribute__((always_inline)) {
            if (D.at_tri) {
              // i * 48 as (i * 3) << 4 with the * 3 spelled as one v_lshl_add_u32: hipcc turns any
              // C spelling back into v_mul_lo_u32, which is quarter rate.  32-bit offset: < 4 GiB of triangles
              uint32_t i3;
              asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(i3) : "v"(D.i));
              const float4 *tp = (const float4 *)(SLOTS ? nodes_base + (D.i << 6) : tris_base + (i3 << 4));  // (a slot per triangle)
#pragma unroll
              for (int k = 0; k < TRIS; k++) {
                if (TINY) {
                  // a lane with fewer triangles left re-reads its last one (never past the array)
                  const uint32_t j3 = i3 + 3u * ((uint32_t)k < D.more ? (uint32_t)k : D.more);
                  const f4 a = tri_cache[j3], b = tri_cache[j3 + 1u], c = tri_cache[j3 + 2u];
                  D.rec[k][0] = make_float4(a.x, a.y, a.z, a.w); D.rec[k][1] = make_float4(b.x, b.y, b.z, b.w);
                  D.rec[k][2] = make_float4(c.x, c.y, c.z, c.w);
                } else if (k == 0) {
                  D.rec[0][0] = tp[0]; D.rec[0][1] = tp[1]; D.rec[0][2] = tp[2];
                } else {
                  // Only lanes that HAVE a k-th triangle left request it: the vector memory pipeline charges a load by
                  // its active lanes (~0.7 cycle per lane and 16 bytes, tools/ubench/l1_td.hip), and on a fine mesh most
                  // leaves hold one triangle - re-reading the last record in the other lanes (what this did before)
                  // was a tenth of all the lane-loads of the S1 frame.  The loads of all k are still in flight together.
                  D.rec[k][0] = D.rec[k][1] = D.rec[k][2] = make_float4(0.f, 0.f, 0.f, 0.f);
                  if ((uint32_t)k <= D.more) {
                    const float4 *tq = tp + (SLOTS ? 4u : 3u) * (uint32_t)k;
                    D.rec[k][0] = tq[0]; D.rec[k][1] = tq[1]; D.rec[k][2] = tq[2];
                  }
                }
              }
            }
          };
          auto tri_test = [&](Ray &S, LeafData &D) __attribute__((always_inline)) {
            if (D.at_tri) {
              if (STATS) n_tris += 1u + D.more;
              if (STATS == 2) wave_tick(lane, w_tri);
#pragma unroll
              for (int k = 0; k < TRIS; k++) {
                if (k == 0 || (uint32_t)k <= D.more) {
                  const float4 a = D.rec[k][0], b = D.rec[k][1], c = D.rec[k][2];
                  V3 v0 = mk(a.x, a.y, a.z), e1 = mk(b.x, b.y, b.z), e2 = mk(c.x, c.y, c.z);
                  V3 pvec = cross(S.rd, e2);
                  float det = dot(e1, pvec);
                  bool valid = !(__builtin_fabsf(det) < 1e-30f);
                  float inv_det = rcp_exact(det);
                  V3 tvec = S.ro - v0;
                  float u = dot(tvec, pvec) * inv_det;
                  valid = valid && !(u < 0.0f || u > 1.0f);
                  V3 qvec = cross(tvec, e1);
                  float v = dot(S.rd, qvec) * inv_det;
                  valid = valid && !(v < 0.0f || u + v > 1.0f);
                  float t = dot(e2, qvec) * inv_det;
                  // the leaf cursor as it stands at this test: first + k, count - k
                  if (valid && t > tmin && t < S.tbest) {
                    S.tbest = t;
                    S.hit_ref = S.cur + (uint32_t)k - ((uint32_t)k << CNT_SHIFT);
                    S.hit_mat = __float_as_uint(b.w);
                    if (KEEP_N) { S.hit_n = cross(e1, e2); S.hit_prim = __float_as_uint(a.w); }
                  }
                }
              }
            }
          };
          auto leaf_advance = [&](Ray &S, LeafData &D) __attribute__((always_inline)) {
            if (D.at_leaf) {
              const bool last_now = D.left == D.more;
              uint32_t next = S.cur + (D.more + 1u) - ((D.more + 1u) << CNT_SHIFT);  // first+1, count-1 per test
              uint32_t popped = HYBRID ? S.stk.pop_if_hybrid(D.fast, last_now, D.top) : S.stk.pop_if(last_now, D.top);
              S.cur = last_now ? popped : next;
            }
          };
          if (R == 1) {
            leaf_fetch(ray0, ld0);
            sph_fetch(ray0, ld0); sph_test(ray0, ld0);
            tri_fetch(ray0, ld0); tri_test(ray0, ld0);
            leaf_advance(ray0, ld0);
          } else {
            leaf_fetch(ray0, ld0); sph_fetch(ray0, ld0); tri_fetch(ray0, ld0);
            leaf_fetch(ray1, ld1); sph_fetch(ray1, ld1); tri_fetch(ray1, ld1);
            sph_test(ray0, ld0); tri_test(ray0, ld0); leaf_advance(ray0, ld0);
            sph_test(ray1, ld1); tri_test(ray1, ld1); leaf_advance(ray1, ld1);
          }
          n_leaf = RAYS_COUNT((int)c_ < -1);
        } while (n_leaf >= leaf_quorum);
        if (STATS == 2) c_leaves += (long long)clock64() - c_phase;
      }
    }
#undef RAYS_COUNT
    if (SMALL) __builtin_amdgcn_s_setprio(0);
    SECTION_END(c_trav);
  }
#undef SECTION_END

  // ---- counters: one atomic per wave and counter --------------------------------
  n_paths = wave_sum(n_paths);
  n_rays = wave_sum(n_rays);
  if (lane == 0) {
    atomicAdd(&P.counters[1], n_paths);
    atomicAdd(&P.counters[2], n_rays);
  }
  if (STATS) {
    n_nodes = wave_sum(n_nodes);
    n_tris = wave_sum(n_tris);
    n_sphs = wave_sum(n_sphs);
    n_scat = wave_sum(n_scat);
    if (lane == 0) {
      atomicAdd(&P.counters[3], n_nodes);
      atomicAdd(&P.counters[4], n_tris);
      atomicAdd(&P.counters[5], n_sphs);
      atomicAdd(&P.counters[6], n_scat);
    }
  }
  if (STATS == 2) {
    unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      size_t w = ((size_t)blockIdx.x * BLOCK + tid) >> 6;
      P.wave_times[w * 4 + 0] = t_start;
      P.wave_times[w * 4 + 1] = t_end;
      P.wave_times[w * 4 + 2] = t_dry;
      P.wave_times[w * 4 + 3] = n_after_dry;
    }
    unsigned long long v[9] = {w_outer, w_shade, l_shade, w_gen, w_node, w_sph, w_tri, w_trav, l_trav};
#pragma unroll
    for (int i = 0; i < 9; i++) {
      unsigned long long t = wave_sum(v[i]);
      if (lane == 0) atomicAdd(&P.counters[7 + i], t);
    }
    if (lane == 0) {
      atomicAdd(&P.counters[16], (unsigned long long)c_shade);
      atomicAdd(&P.counters[17], (unsigned long long)c_item);
      atomicAdd(&P.counters[18], (unsigned long long)c_gen);
      atomicAdd(&P.counters[19], (unsigned long long)c_setup);
      atomicAdd(&P.counters[20], (unsigned long long)c_trav);
      atomicAdd(&P.counters[21], (unsigned long long)c_nodes);   // of c_trav: node phases
      atomicAdd(&P.counters[22], (unsigned long long)c_leaves);  // of c_trav: leaf phases
    }
    {  // MODE 5: lanes in node steps, of them served from the LDS copy of the top; steps on the slow stack path
      unsigned long long a = wave_sum(l_node), b = wave_sum(l_cached), c = wave_sum(w_slow);
      if (lane == 0) { atomicAdd(&P.counters[23], a); atomicAdd(&P.counters[24], b); atomicAdd(&P.counters[25], c); }
      // rays still traversing (at a node or a leaf), summed over node steps / leaf steps of the wave
      unsigned long long d = wave_sum(t_node), e = wave_sum(t_leaf);
      if (lane == 0) { atomicAdd(&P.counters[26], d); atomicAdd(&P.counters[27], e); }
    }
  }
}

// chunk planes -> accum: sum = C0, then += C1, += C2 ... in order (row a8).
// Only pixels this rank owns are touched (image layout: pixels of its tiles;
// compact layout: slots of in-image pixels), exactly the set the trace kernel wrote.
struct CombineGeom {
  uint32_t width, height, ts_log2, tiles_x, rank, world, compact;
};
// run_in: the running sum of the rounds before this one (NULL in the first round); run_out: where this round's running sum
// goes (NULL in the last round, which writes accum).  The additions are those of one pass over all chunks, in the same order.
__global__ __launch_bounds__(256) void combine_kernel(const float4 *planes, float4 *accum, uint64_t n,
                                                       uint32_t n_chunks, uint64_t stride, float spp,
                                                       uint32_t accumulate, CombineGeom g, const float4 *run_in, float4 *run_out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  const uint32_t mask = (1u << g.ts_log2) - 1u;
  for (; i < n; i += step) {
    bool mine;
    if (g.compact) {
      uint32_t j = (uint32_t)(i >> (2u * g.ts_log2)), ty = (uint32_t)(i >> g.ts_log2) & mask, tx = (uint32_t)i & mask;
      uint32_t tile = g.rank + j * g.world;
      uint32_t tile_y = tile / g.tiles_x, tile_x = tile - tile_y * g.tiles_x;
      mine = ((tile_x << g.ts_log2) + tx) < g.width && ((tile_y << g.ts_log2) + ty) < g.height;
    } else {
      uint32_t y = (uint32_t)(i / g.width), x = (uint32_t)(i - (uint64_t)y * g.width);
      uint32_t tile = (y >> g.ts_log2) * g.tiles_x + (x >> g.ts_log2);
      mine = tile % g.world == g.rank;
    }
    if (!mine) continue;
    float4 s = run_in ? run_in[i] : planes[i];
    for (uint32_t c = run_in ? 0u : 1u; c < n_chunks; c++) {
      float4 v = planes[(uint64_t)c * stride + i];
      s.x = s.x + v.x; s.y = s.y + v.y; s.z = s.z + v.z;
    }
    if (run_out) { run_out[i] = s; continue; }
    float4 o = make_float4(s.x, s.y, s.z, spp);
    if (accumulate) {
      float4 in = accum[i];
      o.x = in.x + o.x; o.y = in.y + o.y; o.z = in.z + o.z; o.w = in.w + o.w;
    }
    accum[i] = o;
  }
}

// accum (float4 sums) -> 0xRRGGBBAA (row a8; packing per shaders.hlsl:18-25)
__global__ __launch_bounds__(256) void resolve_kernel(const float4 *accum, uint32_t *out, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float4 a = accum[i];
    uint32_t px = 0xFFu;
    if (a.w > 0.0f) {
      float c[3] = {a.x, a.y, a.z};
      uint32_t ch[3];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        float v = __builtin_sqrtf(fmax_(c[k] / a.w, 0.0f));
        v = fmin_(v, 0.999f);
        ch[k] = (uint32_t)(256.0f * v);
      }
      px = (ch[0] << 24) | (ch[1] << 16) | (ch[2] << 8) | 0xFFu;
    }
    out[i] = px;
  }
}

// gathered compact tile buffers (rank-major, each `stride_px` pixels) -> image
template <typename T>
__global__ __launch_bounds__(256) void untile_kernel(const T *gathered, T *image, uint32_t width,
                                                      uint32_t height, uint32_t ts_log2,
                                                      uint32_t tiles_x, uint32_t world,
                                                      uint64_t stride_px) {
  uint64_t n = (uint64_t)width * height;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  uint32_t mask = (1u << ts_log2) - 1u;
  for (; i < n; i += step) {
    uint32_t y = (uint32_t)(i / width), x = (uint32_t)(i - (uint64_t)y * width);
    uint32_t tile = (y >> ts_log2) * tiles_x + (x >> ts_log2);
    uint32_t rank = tile % world, j = tile / world;
    uint64_t src = (uint64_t)rank * stride_px + ((uint64_t)j << (2u * ts_log2)) +
                   ((uint64_t)(y & mask) << ts_log2) + (x & mask);
    image[i] = gathered[src];
  }
}

// four-wide nodes, 96 bytes apart -> the same nodes in 128-byte slots (modes 9, 10): one thread per 16-byte piece
__global__ __launch_bounds__(256) void pad_nodes4_kernel(const float4 *src, float4 *dst, uint64_t n_nodes) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x, n = n_nodes * 8;
  for (; i < n; i += step) {
    // node k starts 0 or 32 bytes into its slot (bit 1 of k): with the half exchange of odd nodes in the node step, the lanes
    // of a load spread over four 16-byte positions of their lines
    const uint64_t node = i >> 3, piece = i & 7, first = (node & 2u) ? 2u : 0u;
    dst[i] = (piece >= first && piece < first + 6) ? src[node * 6 + (piece - first)] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// ---- static device scratch of rtgpu_render_device_static (the entry point that allocates
// nothing, include/rtgpu.h) ---------------------------------------------------------------
// One array per GPU the library runs on, held until the process ends:
//   [0, kStaticHeader)                 queue head + counters of the launch (the 256 B the
//                                      context's d_counters holds for the other entry points)
//   [kStaticHeader, kStaticScratch)    chunk planes: n_chunks x output pixels x 16 B
// In HOST code `g_static_scratch` is only a stand-in for the symbol, not a device address: the
// address comes from hipGetSymbolAddress, once per device with that device current
// (static_scratch_for, called by rtgpu_create), and that pointer is what kernels, memsets and
// copies get.
constexpr size_t kStaticHeader = 4096;
constexpr size_t kStaticPlanes = 64ull << 20;  // 1920x1080 in up to 2 chunks, 1280x720 in up to 4
constexpr size_t kStaticScratch = kStaticHeader + kStaticPlanes;
static __device__ __align__(256) unsigned char g_static_scratch[kStaticScratch];
constexpr int kMaxDevices = 64;
unsigned char *g_static_ptr[kMaxDevices] = {};  // per device: where ITS copy of the array lives

// The current device must be `device`.  NULL + message on failure.
unsigned char *static_scratch_for(int device, std::string &err) {
  if (device < 0 || device >= kMaxDevices) { err = "static scratch: device index beyond the table"; return nullptr; }
  if (!g_static_ptr[device]) {
    void *p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_static_scratch));
    if (e != hipSuccess || !p) {
      err = std::string("hipGetSymbolAddress(static scratch): ") + hipGetErrorString(e);
      return nullptr;
    }
    g_static_ptr[device] = (unsigned char *)p;
  }
  return g_static_ptr[device];
}

}  // namespace

// =============================== C ABI ===========================================

struct rtgpu_ctx {
  int device = 0;
  std::string err;
  bool have_scene = false;
  mutable rtgpu_flat_scene host{};
  mutable bool host_stale = false;  // rtgpu_refit changed the device arrays; host copy refreshed on demand
  void *d_nodes = nullptr, *d_tris = nullptr, *d_sphs = nullptr, *d_sph_meta = nullptr,
       *d_mats = nullptr;
  void *d_nodes48 = nullptr;    // packed nodes of a tree of >= RTGPU_NODE48_MIN_NODES nodes, else NULL
  void *d_nodes4 = nullptr;     // four-wide packed nodes of such a tree, else NULL
  void *d_nodes4_src = nullptr; // their source map (4 u32 per node), read by a refit
  // the same four-wide nodes in 128-byte slots (kernel modes 9, 10), for trees far larger than the caches: made by the first
  // frame after an upload, build or refit (pad_valid says whether it mirrors d_nodes4)
  void *d_nodes4_pad = nullptr; size_t nodes4_pad_bytes = 0;
  bool pad_valid = false;
  int pad_nodes4 = -1;          // RTGPU_NODE4_PAD=0/1 forces the choice; -1: by the size of the tree (kPadMinBytes)
  // the slot tree (rtgpu_format.h; modes 11, 12): made from d_nodes4 and the primitive arrays by the first frame after an
  // upload, build or refit.  slot_state: 0 not made yet, 1 valid, 2 this tree has no slot form (the four-wide nodes are used),
  // 3 made for this topology but the geometry has been refitted since: refreshed in place by the next frame
  rtgpu::SlotTreeDevice slot;
  int slot_state = 0;
  int slot_tree = -1;           // RTGPU_SLOT_TREE=0/1 forces the choice; -1: kSlotTreeDefault
  uint32_t *d_refit_ws = nullptr; // refit scratch (3 u32 per node), allocated by the first refit of a scene
  rtgpu::DevPool pool;            // device memory kept between rtgpu_build_device calls (dev_pool.h)
  bool no_wide = false;         // RTGPU_NO_WIDE=1: keep the BVH2 traversal on large trees (experiments)
  unsigned long long *d_counters = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  // Every launch shares the context's scratch (queue head and counters, overflow stacks, chunk
  // planes, ev0 / ev1), and rtgpu_render_device is asynchronous on any stream the caller
  // names: ev_last marks the end of the last launch's work (trace + combine) and a launch on
  // ANOTHER stream waits for it first, so two frames never run on the scratch at once.
  unsigned long long *last_counters = nullptr;  // where the last launch kept its counters (own buffer or static scratch)
  unsigned char *static_scratch = nullptr;  // this GPU's g_static_scratch (static_scratch_for)
  char kernel_name[48] = "";  // instantiation of the last launch, as rocprofv3 prints it
  hipEvent_t ev_last = nullptr;
  hipStream_t last_stream = nullptr;
  bool have_last = false;
  int num_cus = 0;
  int variant = 0;              // index into kVariants (refill threshold), see rtgpu_create
  int stack_lds = kStackLds;    // cap on LDS stack entries (RTGPU_STACK_LDS shrinks it for tests)
  bool no_node_cache = false;   // RTGPU_NO_NODE_CACHE=1: every node through the vector L1 (experiments)
  int tiny6 = -1;               // RTGPU_TINY6=0/1: MODE 13 (two blocks of 768 threads per CU, six waves per SIMD) instead of 7; -1: kTiny6Default
  int rays2 = -1;               // RTGPU_RAYS2=0/1: MODE 8 (two rays per lane in lock step) instead of 7; -1: the default (kRays2Default)
  bool no_mats_cache = false;   // RTGPU_NO_MATS_CACHE=1: trees in memory read materials through L1 (experiments)
  bool no_tiny = false;         // RTGPU_NO_TINY=1: a scene that fits LDS whole still runs MODE 2 (experiments)
  bool no_sphere_cache = false; // RTGPU_NO_SPHERE_CACHE=1: MODE 2 reads spheres through L1 as before (experiments)
  bool wide_cache = false;      // RTGPU_WIDE_CACHE=1: four-wide trees with an LDS copy of their top (measured slower)
  int blocks_per_cu[2] = {0, 0};
  uint32_t *d_ovf = nullptr;    // traversal-stack overflow, sized for the largest grid
  unsigned long long *d_wave_times = nullptr;  // profile variant: 4 x u64 per wave
  float4 *d_planes = nullptr;   // chunk planes (grow-only, zero-filled when (re)allocated)
  size_t planes_bytes = 0;
  size_t ovf_threads = 0;
  uint64_t last_waves = 0;      // waves of the last trace launch (rtgpu_debug_wave_times)
  // grow-only staging for the host-pointer entry points
  void *d_accum = nullptr;  size_t d_accum_bytes = 0;
  void *d_ids = nullptr;    size_t d_ids_bytes = 0;
  // rtgpu_render_resolved: the progressive accumulation lives HERE between calls (never crosses
  // PCIe); only the resolved 4 B/px frame is downloaded, through pinned staging
  void *d_res_accum = nullptr;  size_t d_res_accum_bytes = 0;
  void *d_res_rgba = nullptr;   size_t d_res_rgba_bytes = 0;
  uint32_t *h_res_rgba = nullptr; size_t h_res_rgba_bytes = 0;  // hipHostMalloc
  uint32_t res_w = 0, res_h = 0;
  bool res_valid = false;       // d_res_accum holds the running sum of a res_w x res_h frame
  // rtgpu_render_begin / _poll / _end: the frame in flight on the context's own stream
  hipStream_t stream = nullptr;
  hipEvent_t ev_done = nullptr;
  bool pending = false;
  float *pend_accum = nullptr;
  uint32_t *pend_ids = nullptr;
  uint64_t pend_px = 0;
};

namespace {

// Kernel variants: {refill threshold} x {plain, instrumented}.  One source, one
// code path; the threshold only changes WHEN finished lanes are refilled, never
// what a path computes.  RTGPU_REFILL (0/20/40) selects it for experiments.
using KernelFn = void (*)(const KParams);
constexpr int kModes = 14;
struct Variant { int refill; KernelFn fn[3][kModes]; };  // [plain, stats, profile][MODE 0..13 of trace_kernel]
#ifdef RTGPU_QUICK_MODE  // development: ONE instantiation (resource usage of a kernel under work in seconds, not a minute)
#define RTGPU_VARIANT_ROW(S, R) {trace_kernel<0, 20, RTGPU_QUICK_MODE>}
#define RTGPU_VARIANT(R) {R, {RTGPU_VARIANT_ROW(0, R)}}
#else
#define RTGPU_VARIANT_ROW(S, R)                                                                         \
  {trace_kernel<S, R, 0>, trace_kernel<S, R, 1>, trace_kernel<S, R, 2>, trace_kernel<S, R, 3>, trace_kernel<S, R, 4>, \
   trace_kernel<S, R, 5>, trace_kernel<S, R, 6>, trace_kernel<S, R, 7>, trace_kernel<S, R, 8>, trace_kernel<S, R, 9>, \
   trace_kernel<S, R, 10>, trace_kernel<S, R, 11>, trace_kernel<S, R, 12>, trace_kernel<S, R, 13>}
#define RTGPU_VARIANT(R) {R, {RTGPU_VARIANT_ROW(0, R), RTGPU_VARIANT_ROW(1, R), RTGPU_VARIANT_ROW(2, R)}}
#endif
#ifndef RTGPU_REFILL_SMALL
#define RTGPU_REFILL_SMALL 20
#endif
#ifndef RTGPU_REFILL_LARGE
#define RTGPU_REFILL_LARGE 40
#endif
const Variant kVariants[] = {RTGPU_VARIANT(0), RTGPU_VARIANT(RTGPU_REFILL_SMALL), RTGPU_VARIANT(RTGPU_REFILL_LARGE)};
// Default refill threshold, by what bounds the scene: 20 where the whole tree
// sits in LDS and the kernel is VALU-bound (an outer iteration is expensive, so leave
// the traversal loop late); 40 where nodes come from memory and the time follows the
// number of dependent steps a ray takes (so keep the lanes of a step together; measured 16 / 32 / 40 / 48 / 56).
constexpr int kVariantSmallTree = 1, kVariantLargeTree = 2;
#ifndef RTGPU_RAYS2_DEFAULT
#define RTGPU_RAYS2_DEFAULT 0
#endif
constexpr bool kRays2Default = RTGPU_RAYS2_DEFAULT != 0;
#ifndef RTGPU_TINY6_DEFAULT
#define RTGPU_TINY6_DEFAULT 1
#endif
constexpr bool kTiny6Default = RTGPU_TINY6_DEFAULT != 0;
#ifndef RTGPU_PAD_MIN_MB
#define RTGPU_PAD_MIN_MB 128
#endif
#ifndef RTGPU_SLOT_TREE_DEFAULT
#define RTGPU_SLOT_TREE_DEFAULT 1
#endif
constexpr bool kSlotTreeDefault = RTGPU_SLOT_TREE_DEFAULT != 0;
constexpr uint64_t kPadMinBytes = (uint64_t)RTGPU_PAD_MIN_MB << 20;  // four-wide node arrays from this size on are traversed in 128-byte slots  // MODE 8 for scenes that fit LDS whole (RTGPU_RAYS2 overrides)

bool hip_fail(rtgpu_ctx *ctx, hipError_t e, const char *what) {
  if (e == hipSuccess) return false;
  ctx->err = std::string(what) + ": " + hipGetErrorString(e);
  return true;
}
#define HIP_TRY(ctx, call)                         \
  do {                                             \
    if (hip_fail((ctx), (call), #call)) return 2;  \
  } while (0)

void free_scene_device(rtgpu_ctx *ctx) {
  // arrays that came from the context's pool go back to it, the others are hipFree'd
  for (void *p : {ctx->d_nodes, ctx->d_tris, ctx->d_sphs, ctx->d_sph_meta, ctx->d_mats, ctx->d_nodes48, ctx->d_nodes4,
                  ctx->d_nodes4_src, (void *)ctx->d_refit_ws})
    ctx->pool.put(p);
  ctx->d_refit_ws = nullptr;
  ctx->d_nodes = ctx->d_tris = ctx->d_sphs = ctx->d_sph_meta = ctx->d_mats = ctx->d_nodes48 = ctx->d_nodes4 = nullptr;
  ctx->d_nodes4_src = nullptr;
  ctx->pad_valid = false;  // (the padded copy's memory is kept for the next scene)
  ctx->slot_state = 0;     // (and the slot tree's)
  ctx->slot.layout_valid = false;  // (its layout belonged to the scene that goes)
  rtgpu::free_flat_scene(&ctx->host);
  ctx->have_scene = false;
  ctx->host_stale = false;
  ctx->res_valid = false;  // an accumulation of the old scene is not continued
}

// scratch the refits of this scene share: allocated once, freed with the scene
int refit_workspace(rtgpu_ctx *ctx) {
  if (ctx->d_refit_ws) return 0;
  HIP_TRY(ctx, hipMalloc((void **)&ctx->d_refit_ws, rtgpu::refit_workspace_words(ctx->host.n_nodes) * sizeof(uint32_t)));
  return 0;
}

int upload(rtgpu_ctx *ctx, void **dst, const void *src, size_t bytes) {
  bytes = ((bytes + 63) / 64) * 64;
  if (bytes == 0) bytes = 64;
  HIP_TRY(ctx, hipMalloc(dst, bytes));
  HIP_TRY(ctx, hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return 0;
}

int check_settings(rtgpu_ctx *ctx, const rtgpu_camera *cam, const rtgpu_settings *s,
                   uint32_t *ts_log2_out) {
  if (!cam || !s) { ctx->err = "camera or settings is NULL"; return 1; }
  if (ctx->pending) { ctx->err = "a frame begun with rtgpu_render_begin is pending (call rtgpu_render_end)"; return 1; }
  if (!ctx->have_scene) { ctx->err = "no scene uploaded (call rtgpu_upload_scene first)"; return 1; }
  if (s->width == 0 || s->height == 0 || s->width > 65536u || s->height > 65536u) {
    ctx->err = "width/height must be in 1..65536"; return 1;
  }
  if (s->spp == 0 || s->max_depth == 0) { ctx->err = "spp and max_depth must be >= 1"; return 1; }
  if (s->world_size == 0 || s->rank >= s->world_size) { ctx->err = "rank/world_size invalid"; return 1; }
  uint32_t ts = s->tile_size, l = 0;
  while ((1u << l) < ts && l < 31) l++;
  if (ts < 8 || ts > 1024 || (1u << l) != ts) { ctx->err = "tile_size must be a power of two in 8..1024"; return 1; }
  if (!(s->tmin >= 0.0f)) { ctx->err = "tmin must be >= 0"; return 1; }
  if (s->chunk_samples == 0) { ctx->err = "chunk_samples must be >= 1"; return 1; }
  *ts_log2_out = l;
  return 0;
}

}  // namespace

extern "C" const char *rtgpu_last_error(const rtgpu_ctx *ctx) {
  return ctx ? ctx->err.c_str() : rtgpu::global_error();
}

extern "C" int rtgpu_create(rtgpu_ctx **out, int device_index) {
  if (!out) return 1;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    rtgpu::set_global_error(std::string("no HIP device: ") + hipGetErrorString(e));
    return 2;
  }
  if (device_index < 0 || device_index >= n) {
    rtgpu::set_global_error("device_index out of range");
    return 1;
  }
  rtgpu_ctx *ctx = new rtgpu_ctx();
  ctx->device = device_index;
  hipDeviceProp_t prop;
  if ((e = hipSetDevice(device_index)) != hipSuccess ||
      (e = hipGetDeviceProperties(&prop, device_index)) != hipSuccess ||
      (e = hipMalloc((void **)&ctx->d_counters, kCounterBytes)) != hipSuccess ||
      (e = hipEventCreate(&ctx->ev0)) != hipSuccess || (e = hipEventCreate(&ctx->ev1)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&ctx->ev_last, hipEventDisableTiming)) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming)) != hipSuccess) {
    rtgpu::set_global_error(std::string("rtgpu_create: ") + hipGetErrorString(e));
    rtgpu_destroy(ctx);
    return 2;
  }
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    rtgpu::set_global_error(std::string("rtgpu is built for gfx950 only, device is ") + prop.gcnArchName);
    rtgpu_destroy(ctx);
    return 2;
  }
  {
    std::string serr;
    ctx->static_scratch = static_scratch_for(device_index, serr);
    if (!ctx->static_scratch) {
      rtgpu::set_global_error("rtgpu_create: " + serr);
      rtgpu_destroy(ctx);
      return 2;
    }
  }
  ctx->num_cus = prop.multiProcessorCount;
  ctx->variant = -1;  // by scene, unless RTGPU_REFILL names one
  if (const char *env = std::getenv("RTGPU_REFILL")) {
    int want = std::atoi(env);
    for (int i = 0; i < (int)(sizeof(kVariants) / sizeof(kVariants[0])); i++)
      if (kVariants[i].refill == want) ctx->variant = i;
  }
  if (const char *env = std::getenv("RTGPU_NO_NODE_CACHE")) ctx->no_node_cache = std::atoi(env) != 0;
  if (const char *env = std::getenv("RTGPU_NO_WIDE")) ctx->no_wide = std::atoi(env) != 0;
  if (const char *env = std::getenv("RTGPU_WIDE_CACHE")) ctx->wide_cache = std::atoi(env) != 0;
  if (const char *env = std::getenv("RTGPU_NO_SPHERE_CACHE")) ctx->no_sphere_cache = std::atoi(env) != 0;
  if (const char *env = std::getenv("RTGPU_NO_TINY")) ctx->no_tiny = std::atoi(env) != 0;
  if (const char *env = std::getenv("RTGPU_RAYS2")) ctx->rays2 = std::atoi(env) != 0 ? 1 : 0;
  if (const char *env = std::getenv("RTGPU_TINY6")) ctx->tiny6 = std::atoi(env) != 0 ? 1 : 0;
  if (const char *env = std::getenv("RTGPU_NODE4_PAD")) ctx->pad_nodes4 = std::atoi(env) != 0 ? 1 : 0;
  if (const char *env = std::getenv("RTGPU_SLOT_TREE")) ctx->slot_tree = std::atoi(env) != 0 ? 1 : 0;
  if (const char *env = std::getenv("RTGPU_NO_MATS_CACHE")) ctx->no_mats_cache = std::atoi(env) != 0;
  if (const char *env = std::getenv("RTGPU_STACK_LDS")) {
    int want = std::atoi(env);
    if (want >= 2 && want <= kStackLds) ctx->stack_lds = want;
  }
  // one 1024-thread block per CU, holding all 160 KiB of its LDS
  for (const Variant &v : kVariants)
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < kModes; b++)
        if ((e = hipFuncSetAttribute((const void *)v.fn[a][b], hipFuncAttributeMaxDynamicSharedMemorySize,
                                     kLdsBytes)) != hipSuccess) {
          rtgpu::set_global_error(std::string("rtgpu_create: LDS attribute: ") + hipGetErrorString(e));
          rtgpu_destroy(ctx);
          return 2;
        }
  ctx->blocks_per_cu[0] = ctx->blocks_per_cu[1] = kBlocksPerCU;
  // (threads per CU: the larger of the two launch shapes)
  size_t cu_threads = (size_t)kBlocksPerCU * kBlock > (size_t)kSlotBlocksPerCU * kBlockSlots ? (size_t)kBlocksPerCU * kBlock
                                                                                           : (size_t)kSlotBlocksPerCU * kBlockSlots;
  if (cu_threads < 2u * kBlock6) cu_threads = 2u * kBlock6;  // MODE 13
  ctx->ovf_threads = (size_t)ctx->num_cus * cu_threads;
  if ((e = hipMalloc((void **)&ctx->d_wave_times, (size_t)ctx->num_cus * (cu_threads / 64) * 4 * 8)) != hipSuccess) {
    rtgpu::set_global_error(std::string("rtgpu_create: ") + hipGetErrorString(e));
    rtgpu_destroy(ctx);
    return 2;
  }
  if (ctx->ovf_threads == 0) ctx->ovf_threads = kBlock;
  if ((e = hipMalloc((void **)&ctx->d_ovf, ctx->ovf_threads * kStackOvf * sizeof(uint32_t))) != hipSuccess) {
    rtgpu::set_global_error(std::string("rtgpu_create: overflow stack: ") + hipGetErrorString(e));
    rtgpu_destroy(ctx);
    return 2;
  }
  *out = ctx;
  return 0;
}

extern "C" void rtgpu_destroy(rtgpu_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();  // a frame begun and never ended may still be running
  free_scene_device(ctx);
  ctx->pool.trim();
  (void)hipFree(ctx->d_counters);
  (void)hipFree(ctx->d_ovf);
  (void)hipFree(ctx->d_wave_times);
  (void)hipFree(ctx->d_planes);
  (void)hipFree(ctx->d_nodes4_pad);
  rtgpu::free_slot_tree_device(&ctx->slot);
  (void)hipFree(ctx->d_accum);
  (void)hipFree(ctx->d_ids);
  (void)hipFree(ctx->d_res_accum);
  (void)hipFree(ctx->d_res_rgba);
  if (ctx->h_res_rgba) (void)hipHostFree(ctx->h_res_rgba);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->ev_done) (void)hipEventDestroy(ctx->ev_done);
  if (ctx->ev_last) (void)hipEventDestroy(ctx->ev_last);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" int rtgpu_upload_scene(rtgpu_ctx *ctx, const rtgpu_scene_desc *scene) {
  if (!ctx) return 1;
  if (ctx->pending) { ctx->err = "a frame begun with rtgpu_render_begin is pending (call rtgpu_render_end)"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  free_scene_device(ctx);
  ctx->pool.trim();
  if (rtgpu::flatten_scene(scene, &ctx->host, ctx->err)) return 1;
  if (ctx->no_wide && ctx->host.nodes4) {  // RTGPU_NO_WIDE: the oracle must see what the kernel traverses
    std::free((void *)ctx->host.nodes4);
    std::free((void *)ctx->host.nodes4_src);
    ctx->host.nodes4 = nullptr;
    ctx->host.nodes4_src = nullptr;
    ctx->host.n_nodes4 = 0;
    ctx->host.max_stack4 = 0;
  }
  const rtgpu_flat_scene &h = ctx->host;
  // the +8 (+16 meta words) zeroed tail records of flatten_scene are uploaded too
  if (upload(ctx, &ctx->d_nodes, h.nodes, h.n_nodes * sizeof(rtgpu_node)) ||
      upload(ctx, &ctx->d_tris, h.tris, (h.n_tris + 8) * sizeof(rtgpu_tri)) ||
      upload(ctx, &ctx->d_sphs, h.spheres, (h.n_spheres + 8) * sizeof(rtgpu_sph)) ||
      upload(ctx, &ctx->d_sph_meta, h.sphere_meta, (h.n_spheres * 2 + 16) * sizeof(uint32_t)) ||
      upload(ctx, &ctx->d_mats, h.materials, h.n_materials * sizeof(rtgpu_material)) ||
      (h.nodes48 && upload(ctx, &ctx->d_nodes48, h.nodes48, h.n_nodes * sizeof(rtgpu_node48))) ||
      (h.nodes4 && (upload(ctx, &ctx->d_nodes4, h.nodes4, h.n_nodes4 * sizeof(rtgpu_node4)) ||
                    upload(ctx, &ctx->d_nodes4_src, h.nodes4_src, h.n_nodes4 * 16)))) {
    std::string keep = ctx->err;
    free_scene_device(ctx);
    ctx->err = keep;
    return 2;
  }
  ctx->have_scene = true;
  return 0;
}

// Internal (rtgpu_internal.h, used by group.cpp): a scene this process flattened itself goes to
// one more GPU without being built again.  The context keeps counts and its own material table;
// the host arrays stay the caller's unless `take_ownership` (then they are the context's host
// mirror and freed with the scene); otherwise rtgpu_get_flat_scene copies them back on demand.
extern "C" int rtgpu_internal_upload_flat(rtgpu_ctx *ctx, const rtgpu_flat_scene *fs, int take_ownership) {
  if (!ctx || !fs) return 1;
  if (ctx->pending) { ctx->err = "a frame begun with rtgpu_render_begin is pending (call rtgpu_render_end)"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  free_scene_device(ctx);
  ctx->pool.trim();
  const bool wide = fs->nodes4 && !ctx->no_wide;
  if (upload(ctx, &ctx->d_nodes, fs->nodes, fs->n_nodes * sizeof(rtgpu_node)) ||
      upload(ctx, &ctx->d_tris, fs->tris, (fs->n_tris + 8) * sizeof(rtgpu_tri)) ||
      upload(ctx, &ctx->d_sphs, fs->spheres, (fs->n_spheres + 8) * sizeof(rtgpu_sph)) ||
      upload(ctx, &ctx->d_sph_meta, fs->sphere_meta, (fs->n_spheres * 2 + 16) * sizeof(uint32_t)) ||
      upload(ctx, &ctx->d_mats, fs->materials, fs->n_materials * sizeof(rtgpu_material)) ||
      (fs->nodes48 && upload(ctx, &ctx->d_nodes48, fs->nodes48, fs->n_nodes * sizeof(rtgpu_node48))) ||
      (wide && (upload(ctx, &ctx->d_nodes4, fs->nodes4, fs->n_nodes4 * sizeof(rtgpu_node4)) ||
                upload(ctx, &ctx->d_nodes4_src, fs->nodes4_src, fs->n_nodes4 * 16)))) {
    std::string keep = ctx->err;
    free_scene_device(ctx);
    ctx->err = keep;
    return 2;
  }
  rtgpu_flat_scene &h = ctx->host;
  if (take_ownership) {
    h = *fs;
    if (!wide) {
      std::free((void *)h.nodes4); std::free((void *)h.nodes4_src);
      h.nodes4 = nullptr; h.nodes4_src = nullptr; h.n_nodes4 = 0; h.max_stack4 = 0;
    }
  } else {
    std::memset(&h, 0, sizeof(h));
    h.n_nodes = fs->n_nodes; h.n_tris = fs->n_tris; h.n_spheres = fs->n_spheres; h.n_materials = fs->n_materials;
    h.max_depth = fs->max_depth; h.build_ms = fs->build_ms;
    if (wide) { h.n_nodes4 = fs->n_nodes4; h.max_stack4 = fs->max_stack4; }
    const size_t mb = fs->n_materials ? (size_t)fs->n_materials * sizeof(rtgpu_material) : 64;
    void *m = std::aligned_alloc(64, (mb + 63) / 64 * 64);
    if (!m) { free_scene_device(ctx); ctx->err = "out of host memory for the material table"; return 2; }
    std::memcpy(m, fs->materials, (size_t)fs->n_materials * sizeof(rtgpu_material));
    h.materials = m;
    ctx->host_stale = true;
  }
  ctx->have_scene = true;
  return 0;
}

namespace {
int upload_lbvh_common(rtgpu_ctx *ctx, const rtgpu_scene_desc *scene, const void *d_spheres, const void *d_triangles,
                       bool from_device);
}

// As rtgpu_upload_scene, with the BVH built on the GPU (lbvh_build.hip).
extern "C" int rtgpu_upload_scene_lbvh(rtgpu_ctx *ctx, const rtgpu_scene_desc *scene) {
  if (!ctx) return 1;
  if (ctx->pending) { ctx->err = "a frame begun with rtgpu_render_begin is pending (call rtgpu_render_end)"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  free_scene_device(ctx);
  ctx->pool.trim();
  if (rtgpu::validate_scene(scene, ctx->err)) return 1;
  return upload_lbvh_common(ctx, scene, nullptr, nullptr, false);
}

// The same for primitives that already live in GPU memory (a simulation or a loader wrote
// them there): rtgpu_sphere / rtgpu_triangle arrays on the context's device, the material
// table from the host.  Nothing but the materials crosses PCIe; the arrays are only read.
extern "C" int rtgpu_build_device(rtgpu_ctx *ctx, const void *d_spheres, uint64_t n_spheres, const void *d_triangles,
                                  uint64_t n_triangles, const rtgpu_material *materials, uint64_t n_materials) {
  if (!ctx) return 1;
  if (ctx->pending) { ctx->err = "a frame begun with rtgpu_render_begin is pending (call rtgpu_render_end)"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  free_scene_device(ctx);
  if ((n_spheres && !d_spheres) || (n_triangles && !d_triangles)) { ctx->err = "NULL device array with a non-zero count"; return 1; }
  if (n_spheres + n_triangles >= (1ull << 28)) { ctx->err = "lbvh: too many primitives (limit 2^28)"; return 1; }
  rtgpu_scene_desc scene;
  std::memset(&scene, 0, sizeof(scene));
  scene.materials = materials;
  scene.n_materials = n_materials;
  if (rtgpu::validate_scene(&scene, ctx->err)) return 1;  // the material table
  // what validate_scene checks for primitives on the host, on the device
  if (int rc = rtgpu::check_prims_device(d_spheres, (uint32_t)n_spheres, d_triangles, (uint32_t)n_triangles,
                                         (uint32_t)std::min<uint64_t>(n_materials, 0xFFFFFFFFull), nullptr, ctx->err, "build"))
    return rc;
  scene.n_spheres = n_spheres;
  scene.n_triangles = n_triangles;
  // outputs and temporaries come from the context's pool: a frame loop that builds every
  // frame asks for the same sizes each time (dev_pool.h)
  int rc = upload_lbvh_common(ctx, &scene, n_spheres ? d_spheres : nullptr, n_triangles ? d_triangles : nullptr, true);
  ctx->pool.end_of_build();
  return rc;
}

namespace {
// scene: validated; its primitive arrays are read from d_spheres / d_triangles when given
int upload_lbvh_common(rtgpu_ctx *ctx, const rtgpu_scene_desc *scene, const void *d_spheres, const void *d_triangles,
                       bool from_device) {
  rtgpu::LbvhDevice L;
  rtgpu::DevPool *pool = from_device ? &ctx->pool : nullptr;  // rtgpu_build_device works from the pool
  int rc = rtgpu::lbvh_build_device_ptrs(scene, d_spheres, d_triangles, pool, &L, ctx->err);
  if (rc) return rc;
  ctx->d_nodes = L.d_nodes; ctx->d_tris = L.d_tris; ctx->d_sphs = L.d_sphs; ctx->d_sph_meta = L.d_sph_meta;
  // The host copy of the flattened scene (what rtgpu_get_flat_scene hands out) is made on
  // demand: the tree, its packed copy and its four-wide nodes are all made on the GPU, and
  // an upload of 10.5 M triangles takes 0.1 s instead of 0.55 s without the copies back.
  rtgpu_flat_scene &h = ctx->host;
  std::memset(&h, 0, sizeof(h));
  h.n_nodes = L.n_nodes; h.n_tris = scene->n_triangles; h.n_spheres = scene->n_spheres;
  h.n_materials = scene->n_materials; h.max_depth = L.max_depth; h.build_ms = L.gpu_ms;
  h.materials = rtgpu::flatten_materials(scene);
  if (!h.materials) {
    ctx->err = "out of host memory for the material table";
    free_scene_device(ctx);
    return 2;
  }
  if (upload(ctx, &ctx->d_mats, h.materials, h.n_materials * sizeof(rtgpu_material))) {
    std::string keep = ctx->err;
    free_scene_device(ctx);
    ctx->err = keep;
    return 2;
  }
  if (h.n_nodes >= RTGPU_NODE48_MIN_NODES) {
    const size_t pb = (size_t)h.n_nodes * sizeof(rtgpu_node48);
    hipError_t e = rtgpu::dev_get(pool, &ctx->d_nodes48, (pb + 63) / 64 * 64);
    if (e != hipSuccess) {
      ctx->err = std::string("lbvh: packed nodes: ") + hipGetErrorString(e);
      free_scene_device(ctx);
      return 2;
    }
    std::string perr;
    if (int prc = rtgpu::pack_nodes48_device(ctx->d_nodes, h.n_nodes, ctx->d_nodes48, nullptr, perr)) {
      free_scene_device(ctx);
      ctx->err = perr;
      return prc;
    }
    if (!ctx->no_wide) {  // four-wide nodes: collapsed on the device (collapse4.hip)
      rtgpu::Nodes4Device w;
      if (int crc = rtgpu::collapse_nodes4_device(ctx->d_nodes, h.n_nodes, pool, &w, perr)) {
        free_scene_device(ctx);
        ctx->err = perr;
        return crc;
      }
      ctx->d_nodes4 = w.d_nodes4;
      ctx->d_nodes4_src = w.d_src;
      h.n_nodes4 = w.n_nodes4;
      h.max_stack4 = w.max_stack4;
      h.build_ms += w.gpu_ms;
    }
  }
  ctx->host_stale = true;
  ctx->have_scene = true;
  return 0;
}
}  // namespace

// New geometry (and material table) under the topology of the uploaded tree
// (refit.hip).  The host copy of the flattened scene is refreshed lazily, by the
// next rtgpu_get_flat_scene: a frame loop that refits every frame never pays for it.
extern "C" int rtgpu_refit(rtgpu_ctx *ctx, const rtgpu_scene_desc *scene) {
  if (!ctx) return 1;
  if (ctx->pending) { ctx->err = "a frame begun with rtgpu_render_begin is pending (call rtgpu_render_end)"; return 1; }
  if (!ctx->have_scene) { ctx->err = "no scene uploaded (call rtgpu_upload_scene first)"; return 1; }
  if (rtgpu::validate_scene(scene, ctx->err)) return 1;
  rtgpu_flat_scene &h = ctx->host;
  if (scene->n_spheres != h.n_spheres || scene->n_triangles != h.n_tris || scene->n_materials != h.n_materials) {
    ctx->err = "refit: the scene must hold the sphere, triangle and material counts of the uploaded one";
    return 1;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  rtgpu_material *mats = rtgpu::flatten_materials(scene);
  if (!mats) { ctx->err = "out of host memory for the material table"; return 2; }
  std::free((void *)h.materials);
  h.materials = mats;
  if (h.n_materials)
    HIP_TRY(ctx, hipMemcpy(ctx->d_mats, mats, h.n_materials * sizeof(rtgpu_material), hipMemcpyHostToDevice));
  if (refit_workspace(ctx)) return 2;
  uint32_t *flag = ctx->d_refit_ws + rtgpu::refit_workspace_words(h.n_nodes) - 1;  // last word: the packers' flag
  ctx->host_stale = true;
  double ms = 0.0;
  int rc = rtgpu::refit_device(scene, ctx->d_nodes, h.n_nodes, ctx->d_tris, ctx->d_sphs, ctx->d_sph_meta, ctx->d_refit_ws, &ms, ctx->err);
  if (!rc && ctx->d_nodes48) rc = rtgpu::pack_nodes48_device(ctx->d_nodes, h.n_nodes, ctx->d_nodes48, flag, ctx->err);
  // the four-wide nodes keep their collapse: every child box is re-read from the refitted BVH2
  if (!rc && ctx->d_nodes4) rc = rtgpu::refresh_nodes4_device(ctx->d_nodes, ctx->d_nodes4, ctx->d_nodes4_src, h.n_nodes4, flag, ctx->err);
  if (rc) {  // the device arrays may be half rewritten: the scene is gone
    std::string keep = ctx->err;
    free_scene_device(ctx);
    ctx->err = keep + " (scene dropped, upload again)";
    return rc;
  }
  h.build_ms = ms;
  ctx->res_valid = false;  // new geometry: rtgpu_render_resolved starts its sum again (include/rtgpu.h)
  ctx->pad_valid = false;  // ... and the padded copy of the four-wide nodes is remade by the next frame
  ctx->slot_state = ctx->slot.layout_valid ? 3 : 0;  // ... and the slot tree refreshed (its layout follows the topology, which stays)
  return 0;
}

// rtgpu_refit for geometry that already lives on the GPU (an animation or simulation
// step wrote it there): nothing crosses PCIe.  Materials stay as uploaded.
extern "C" int rtgpu_refit_device(rtgpu_ctx *ctx, const void *d_spheres, uint64_t n_spheres,
                                  const void *d_triangles, uint64_t n_triangles) {
  if (!ctx) return 1;
  if (ctx->pending) { ctx->err = "a frame begun with rtgpu_render_begin is pending (call rtgpu_render_end)"; return 1; }
  if (!ctx->have_scene) { ctx->err = "no scene uploaded (call rtgpu_upload_scene first)"; return 1; }
  rtgpu_flat_scene &h = ctx->host;
  if (n_spheres != h.n_spheres || n_triangles != h.n_tris) {
    ctx->err = "refit: the scene must hold the sphere and triangle counts of the uploaded one";
    return 1;
  }
  if ((n_spheres && !d_spheres) || (n_triangles && !d_triangles)) { ctx->err = "refit: NULL device array with a non-zero count"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  if (refit_workspace(ctx)) return 2;
  uint32_t *flag = ctx->d_refit_ws + rtgpu::refit_workspace_words(h.n_nodes) - 1;  // last word: the checks' and packers' flag
  // what validate_scene checks on the host, before anything is rewritten
  if (int rc = rtgpu::check_prims_device(d_spheres, (uint32_t)n_spheres, d_triangles, (uint32_t)n_triangles,
                                         (uint32_t)h.n_materials, flag, ctx->err, "refit"))
    return rc;
  ctx->host_stale = true;
  double ms = 0.0;
  int rc = rtgpu::refit_device_ptrs(d_spheres, (uint32_t)n_spheres, d_triangles, (uint32_t)n_triangles, ctx->d_nodes,
                                    h.n_nodes, ctx->d_tris, ctx->d_sphs, ctx->d_sph_meta, ctx->d_refit_ws, &ms, ctx->err);
  if (!rc && ctx->d_nodes48) rc = rtgpu::pack_nodes48_device(ctx->d_nodes, h.n_nodes, ctx->d_nodes48, flag, ctx->err);
  if (!rc && ctx->d_nodes4) rc = rtgpu::refresh_nodes4_device(ctx->d_nodes, ctx->d_nodes4, ctx->d_nodes4_src, h.n_nodes4, flag, ctx->err);
  if (rc) {  // the device arrays may be half rewritten: the scene is gone
    std::string keep = ctx->err;
    free_scene_device(ctx);
    ctx->err = keep + " (scene dropped, upload again)";
    return rc;
  }
  h.build_ms = ms;
  ctx->res_valid = false;  // new geometry: rtgpu_render_resolved starts its sum again (include/rtgpu.h)
  ctx->pad_valid = false;  // ... and the padded copy of the four-wide nodes is remade by the next frame
  ctx->slot_state = ctx->slot.layout_valid ? 3 : 0;  // ... and the slot tree refreshed (its layout follows the topology, which stays)
  return 0;
}

extern "C" int rtgpu_get_flat_scene(const rtgpu_ctx *ctx, rtgpu_flat_scene *out) {
  if (!ctx || !out) return 1;
  if (!ctx->have_scene) { const_cast<rtgpu_ctx *>(ctx)->err = "no scene uploaded (call rtgpu_upload_scene first)"; return 1; }
  if (ctx->host_stale) {
    // arrays a GPU build has not copied back yet are allocated here (with the zeroed tail
    // records of flatten_scene); the context stays logically const
    rtgpu_flat_scene &h = const_cast<rtgpu_ctx *>(ctx)->host;
    auto want = [](const void *cur, size_t bytes) -> const void * {
      if (cur) return cur;
      bytes = bytes ? (bytes + 63) / 64 * 64 : 64;
      void *m = std::aligned_alloc(64, bytes);
      if (m) std::memset(m, 0, bytes);
      return m;
    };
    h.nodes = want(h.nodes, (size_t)h.n_nodes * sizeof(rtgpu_node));
    h.tris = want(h.tris, ((size_t)h.n_tris + 8) * sizeof(rtgpu_tri));
    h.spheres = want(h.spheres, ((size_t)h.n_spheres + 8) * sizeof(rtgpu_sph));
    h.sphere_meta = (const uint32_t *)want(h.sphere_meta, ((size_t)h.n_spheres * 2 + 16) * sizeof(uint32_t));
    if (ctx->d_nodes48) h.nodes48 = want(h.nodes48, (size_t)h.n_nodes * sizeof(rtgpu_node48));
    if (ctx->d_nodes4) {
      h.nodes4 = want(h.nodes4, (size_t)h.n_nodes4 * sizeof(rtgpu_node4));
      h.nodes4_src = (const uint32_t *)want(h.nodes4_src, (size_t)h.n_nodes4 * 16);
    }
    std::string &msg = const_cast<rtgpu_ctx *>(ctx)->err;
    if (!h.nodes || !h.tris || !h.spheres || !h.sphere_meta || (ctx->d_nodes48 && !h.nodes48) ||
        (ctx->d_nodes4 && (!h.nodes4 || !h.nodes4_src))) {
      msg = "rtgpu_get_flat_scene: out of host memory for the copy of the flattened scene";
      return 2;
    }
    if (hipSetDevice(ctx->device) != hipSuccess ||
        hipMemcpy((void *)h.nodes, ctx->d_nodes, (size_t)h.n_nodes * sizeof(rtgpu_node), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy((void *)h.tris, ctx->d_tris, (size_t)h.n_tris * sizeof(rtgpu_tri), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy((void *)h.spheres, ctx->d_sphs, (size_t)h.n_spheres * sizeof(rtgpu_sph), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy((void *)h.sphere_meta, ctx->d_sph_meta, (size_t)h.n_spheres * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess ||
        (h.nodes48 && hipMemcpy((void *)h.nodes48, ctx->d_nodes48, (size_t)h.n_nodes * sizeof(rtgpu_node48), hipMemcpyDeviceToHost) != hipSuccess) ||
        (h.nodes4 && (hipMemcpy((void *)h.nodes4, ctx->d_nodes4, (size_t)h.n_nodes4 * sizeof(rtgpu_node4), hipMemcpyDeviceToHost) != hipSuccess ||
                      hipMemcpy((void *)h.nodes4_src, ctx->d_nodes4_src, (size_t)h.n_nodes4 * 16, hipMemcpyDeviceToHost) != hipSuccess))) {
      msg = std::string("rtgpu_get_flat_scene: copying the flattened scene back failed: ") + hipGetErrorString(hipGetLastError());
      return 2;
    }
    ctx->host_stale = false;
  }
  *out = ctx->host;
  return 0;
}

namespace {
// Device scratch of one launch.  The context's own buffers (grown on demand) for the ordinary
// entry points; the static array of this GPU for rtgpu_render_device_static, which may not
// allocate: then planes_cap is a hard limit and there is no overflow stack.
struct LaunchScratch {
  unsigned long long *counters;
  float4 *planes;          // NULL with planes_cap == 0: take (and grow) the context's
  size_t planes_cap;
  bool may_allocate;
};

int launch_frame(rtgpu_ctx *ctx, const rtgpu_camera *cam, const rtgpu_settings *s, void *d_accum_rgba, void *d_prim_id,
                 void *stream_, rtgpu_stats *stats, LaunchScratch sc);
}  // namespace

extern "C" int rtgpu_render_device(rtgpu_ctx *ctx, const rtgpu_camera *cam, const rtgpu_settings *s,
                                   void *d_accum_rgba, void *d_prim_id, void *stream_,
                                   rtgpu_stats *stats) {
  if (!ctx) return 1;
  return launch_frame(ctx, cam, s, d_accum_rgba, d_prim_id, stream_, stats, LaunchScratch{ctx->d_counters, nullptr, 0, true});
}

// The entry point that allocates nothing (include/rtgpu.h): the launch's scratch - queue head,
// counters, chunk planes - lives in this GPU's static array; no hipMalloc / hipFree or any
// allocating routine is reached from here.
extern "C" int rtgpu_render_device_static(rtgpu_ctx *ctx, const rtgpu_camera *cam, const rtgpu_settings *s,
                                          void *d_accum_rgba, void *d_prim_id, void *stream_) {
  if (!ctx) return 1;
  if (!ctx->static_scratch) { ctx->err = "static scratch of this GPU was not resolved at rtgpu_create"; return 2; }
  if (s && (s->flags & RTGPU_FLAG_PROFILE)) { ctx->err = "rtgpu_render_device_static: RTGPU_FLAG_PROFILE is not supported"; return 1; }
  LaunchScratch sc{(unsigned long long *)ctx->static_scratch, (float4 *)(ctx->static_scratch + kStaticHeader), kStaticPlanes, false};
  return launch_frame(ctx, cam, s, d_accum_rgba, d_prim_id, stream_, nullptr, sc);
}

extern "C" uint64_t rtgpu_static_scratch_bytes(void) { return (uint64_t)kStaticPlanes; }

namespace {
int launch_frame(rtgpu_ctx *ctx, const rtgpu_camera *cam, const rtgpu_settings *s, void *d_accum_rgba, void *d_prim_id,
                 void *stream_, rtgpu_stats *stats, LaunchScratch sc) {
  uint32_t ts_log2 = 0;
  if (check_settings(ctx, cam, s, &ts_log2)) return 1;
  if (!d_accum_rgba) { ctx->err = "accum output pointer is NULL"; return 1; }
  hipStream_t stream = (hipStream_t)stream_;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  ctx->last_counters = sc.counters;
  // the previous launch ran on another stream: order this one after it (see ev_last)
  if (ctx->have_last && ctx->last_stream != stream) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->ev_last, 0));

  rtgpu::TileGrid g = rtgpu::tile_grid(s->width, s->height, s->tile_size);
  uint64_t local_tiles = rtgpu::rank_tiles(g.n_tiles, s->rank, s->world_size);
  uint64_t n_px_items = local_tiles << (2 * ts_log2);
  uint32_t n_chunks = (s->spp + s->chunk_samples - 1) / s->chunk_samples;
  // A frame of more than kRoundChunks chunks is traced in rounds: the planes of one round plus a plane for the running sum -
  // 65 planes at most, whatever spp / chunk_samples the caller chose.
  const uint32_t round_chunks = n_chunks < kRoundChunks ? n_chunks : kRoundChunks;
  const uint32_t n_rounds = (n_chunks + kRoundChunks - 1) / kRoundChunks;
  const uint32_t n_planes = n_chunks > 1 ? round_chunks + (n_rounds > 1 ? 1u : 0u) : 0u;
  uint64_t n_items = n_px_items * round_chunks;  // of the largest round
  if (n_items >= 0xFFFFFF00ull) { ctx->err = "frame too large for one launch (>= 2^32 - 256 work items)"; return 1; }
  uint64_t n_px_out = (s->flags & RTGPU_FLAG_COMPACT) ? n_px_items : (uint64_t)s->width * s->height;
  float4 *planes = sc.planes;
  if (n_chunks > 1 && n_px_out > 0 && !sc.may_allocate) {
    if ((uint64_t)n_planes * n_px_out * sizeof(float4) > (uint64_t)sc.planes_cap) {
      char msg[256];
      std::snprintf(msg, sizeof(msg), "rtgpu_render_device_static: %u chunk planes x %llu pixels x 16 B exceed the "
                    "static scratch of %llu bytes (use chunk_samples >= spp for one chunk, fewer chunks, or a smaller frame)",
                    n_planes, (unsigned long long)n_px_out, (unsigned long long)sc.planes_cap);
      ctx->err = msg;
      return 1;
    }
  }
  if (n_chunks > 1 && n_px_out > 0 && sc.may_allocate) {
    size_t want = (size_t)n_planes * n_px_out * sizeof(float4);
    if (ctx->planes_bytes < want) {  // grow-only
      HIP_TRY(ctx, hipStreamSynchronize(stream));  // (which, after the wait above, covers the last launch on any stream)
      (void)hipFree(ctx->d_planes);
      ctx->d_planes = nullptr;
      ctx->planes_bytes = 0;
      HIP_TRY(ctx, hipMalloc((void **)&ctx->d_planes, want));
      HIP_TRY(ctx, hipMemset(ctx->d_planes, 0, want));
      HIP_TRY(ctx, hipDeviceSynchronize());  // the memset runs on the null stream; `stream` may be non-blocking
      ctx->planes_bytes = want;
    }
  }
  if (sc.may_allocate) planes = ctx->d_planes;

  KParams P;
  std::memset(&P, 0, sizeof(P));
  // four-wide nodes when the tree has them (and their stack fits), else packed BVH2 nodes, else the 64-byte ones
  const uint32_t stack_fit_all = (uint32_t)(kLdsBytes / (kBlock * sizeof(uint32_t)));
  const uint32_t need4 = ctx->host.max_stack4 + 4u;  // + sentinel + the three unconditional stores above the top
  const uint32_t cap4 = ctx->stack_lds == kStackLds ? stack_fit_all : (uint32_t)ctx->stack_lds;
  const bool wide = ctx->d_nodes4 != nullptr && need4 <= (cap4 < stack_fit_all ? cap4 : stack_fit_all) + (uint32_t)kStackOvf;
  // four-wide trees: the top kWideCached nodes in LDS, the stack in what is left, deeper entries
  // in the overflow buffer (MODE 5).  The bound need4 is a worst case few rays come near, and the
  // vector memory pipeline is what these scenes wait for.  Not for the entry point
  // that may not use the context's overflow buffer, nor with RTGPU_WIDE_CACHE=0 (experiments).
  const bool wide_cache = wide && ctx->wide_cache && sc.may_allocate && kBlocksPerCU == 1;
  // The slot tree (modes 11, 12) in preference to the four-wide nodes it is made from, when this tree has one.  Made here, by
  // the first frame after an upload / build / refit; the entry point that may not allocate uses it only if it exists already -
  // and only if the whole stack fits the smaller LDS share of these modes' blocks (it has no overflow stack).
  const uint32_t slot_stack_fit = (uint32_t)((kLdsSlotBytes - kPathLdsWords * kBlockSlots * (int)sizeof(uint32_t)) / (kBlockSlots * sizeof(uint32_t)));
  bool slots = false;
  if (wide && !wide_cache && kBlocksPerCU == 1) {
    const bool want_slots = ctx->slot_tree >= 0 ? ctx->slot_tree != 0 : kSlotTreeDefault;
    if (want_slots && ctx->slot_state == 3 && sc.may_allocate) {
      if (rtgpu::refresh_slot_tree_device(ctx->d_nodes4, ctx->d_tris, ctx->d_sphs, ctx->d_sph_meta, &ctx->slot, &ctx->pool, (void *)stream, ctx->err))
        return 2;
      ctx->slot_state = ctx->slot.n_slots > 0 ? 1 : 2;
    }
    if (want_slots && ctx->slot_state == 0 && sc.may_allocate) {
      if (rtgpu::build_slot_tree_device(ctx->d_nodes4, ctx->host.n_nodes4, ctx->d_tris, ctx->host.n_tris, ctx->d_sphs, ctx->d_sph_meta,
                                        ctx->host.n_spheres, &ctx->slot, &ctx->pool, (void *)stream, ctx->err))
        return 2;
      ctx->slot_state = ctx->slot.n_slots > 0 ? 1 : 2;
    }
    slots = want_slots && ctx->slot_state == 1;
    if (slots && !sc.may_allocate && need4 > (cap4 < slot_stack_fit ? cap4 : slot_stack_fit)) slots = false;
  }
  const unsigned block_threads = slots ? (unsigned)kBlockSlots : (unsigned)kBlock;      // (MODE 8 picks its own below)
  const size_t lds_budget = slots ? (size_t)kLdsSlotBytes : (size_t)kLdsBytes;
  const size_t path_bytes = slots ? (size_t)kPathLdsWords * kBlockSlots * sizeof(uint32_t) : 0;  // sum and throughput per lane
  P.nodes = (const float4 *)(slots ? ctx->slot.d_slots : wide ? ctx->d_nodes4 : ctx->d_nodes48 ? ctx->d_nodes48 : ctx->d_nodes);
  P.tris = (const float4 *)ctx->d_tris;
  P.sphs = (const float4 *)ctx->d_sphs;
  P.sph_meta = (const uint2 *)ctx->d_sph_meta;
  P.mats = (const float4 *)ctx->d_mats;
  P.accum = (float4 *)d_accum_rgba;
  P.prim_id = (uint32_t *)d_prim_id;
  P.counters = sc.counters;
  P.ovf_stack = ctx->d_ovf;
  P.wave_times = ctx->d_wave_times;
  for (int k = 0; k < 3; k++) {
    P.cam_origin[k] = cam->origin[k]; P.cam_ll[k] = cam->lower_left[k];
    P.cam_hor[k] = cam->horizontal[k]; P.cam_ver[k] = cam->vertical[k];
    P.cam_u[k] = cam->u[k]; P.cam_v[k] = cam->v[k];
    P.sky_bottom[k] = s->sky_bottom[k]; P.sky_top[k] = s->sky_top[k];
  }
  P.lens_radius = cam->lens_radius;
  P.inv_w = 1.0f / (float)s->width;   // IEEE division on the host, as the oracle does
  P.inv_h = 1.0f / (float)s->height;
  P.tmin = s->tmin;
  P.width = s->width; P.height = s->height; P.spp = s->spp; P.sample_offset = s->sample_offset;
  P.max_depth = s->max_depth; P.seed = s->seed;
  P.ts_log2 = ts_log2; P.tiles_x = g.tiles_x; P.rank = s->rank; P.world = s->world_size;
  P.n_items = (uint32_t)n_items; P.flags = s->flags;
  P.n_big = (uint32_t)n_items;  // (set below, once the grid is known)
  P.planes = planes;
  P.chunk_samples = s->chunk_samples; P.n_chunks = round_chunks; P.plane_stride = (uint32_t)n_px_out;
  P.groups_small = (n_items >> 6) < (1ull << 22) ? 1u : 0u;
  P.inv_n_chunks = 1.0f / (float)round_chunks;
  P.tiles_small = g.n_tiles < (1u << 22) ? 1u : 0u;
  P.inv_tiles_x = 1.0f / (float)g.tiles_x;
  // LDS budget of the one block per CU: the stack gets what the tree depth
  // needs (pushes never exceed the depth; +1 sentinel, +1 for the unconditional
  // store above the top).
  uint32_t need = wide ? need4 : ctx->host.max_depth + 2u;
  uint32_t stack_lds = need < 8u ? 8u : need;
  const uint32_t cap = wide ? cap4 : (uint32_t)ctx->stack_lds;  // kStackLds (the whole LDS for four-wide nodes), or RTGPU_STACK_LDS
  if (stack_lds > cap) stack_lds = cap;
  uint32_t stack_fit = (uint32_t)((lds_budget - path_bytes) / (block_threads * sizeof(uint32_t)));      // entries the block's LDS share holds
  if (wide_cache) stack_fit = (uint32_t)((kLdsBytes - kWideCached * (int)sizeof(rtgpu_node4)) / (kBlock * sizeof(uint32_t)));
  if (stack_lds > stack_fit) stack_lds = stack_fit;
  P.stack_lds = stack_lds;
  P.stack_lds_only = (need <= stack_lds && !wide_cache) ? 1u : 0u;
  if (!sc.may_allocate && !P.stack_lds_only) {
    ctx->err = "rtgpu_render_device_static: the tree is too deep for traversal stacks in LDS alone (the overflow stack is not part of the static scratch)";
    return 1;
  }
  // small scenes: the whole tree in LDS with planes kSmallNodes apart (MODE 2); larger
  // trees are fetched through L1 (a partial cache measured no gain)
  const size_t stack_bytes = (size_t)stack_lds * block_threads * sizeof(uint32_t);
  size_t lds_bytes = stack_bytes + path_bytes;
  int mode = P.stack_lds_only ? 1 : 0;
  P.n_cached = 0;
  if (mode == 1 && !ctx->no_node_cache && !wide && !ctx->d_nodes48 && ctx->host.n_nodes <= (uint64_t)kSmallNodes &&
      stack_bytes + (size_t)kSmallNodes * sizeof(rtgpu_node) <= (size_t)kLdsBytes) {
    mode = 2;
    P.n_cached = (uint32_t)ctx->host.n_nodes;
    lds_bytes = stack_bytes + (size_t)kSmallNodes * sizeof(rtgpu_node);
    // the spheres too, if what is left holds them all (S0: 485 x 16 B)
    const size_t sph_bytes = (size_t)ctx->host.n_spheres * sizeof(rtgpu_sph);
    if (!ctx->no_sphere_cache && ctx->host.n_spheres > 0 && lds_bytes + sph_bytes <= (size_t)kLdsBytes) {
      P.sph_cached = (uint32_t)ctx->host.n_spheres;   // the count: the kernel copies that many
      lds_bytes += (sph_bytes + 15) / 16 * 16;
    }
    // a scene that fits LDS whole (S0 does): MODE 7
    if (!ctx->no_sphere_cache && !ctx->no_tiny && kBlocksPerCU == 1 && ctx->host.n_nodes <= (uint64_t)kTinyNodes &&
        ctx->host.n_spheres <= (uint64_t)kTinySphs && ctx->host.n_tris <= (uint64_t)kTinyTris && stack_bytes + (size_t)kTinyEnd * 16 <= (size_t)kLdsBytes) {
      mode = 7;
      P.sph_cached = (uint32_t)ctx->host.n_spheres;
      P.tiny_tris = (uint32_t)ctx->host.n_tris;
      lds_bytes = stack_bytes + (size_t)kTinyEnd * 16;
      // two rays per lane in lock step (MODE 8): blocks of kBlock2 threads, two stacks per lane
      const bool rays2 = ctx->rays2 >= 0 ? ctx->rays2 != 0 : kRays2Default;
      const size_t stack_bytes2 = (size_t)stack_lds * 2u * kBlock2 * sizeof(uint32_t);
      if (rays2 && stack_bytes2 + (size_t)kTinyEnd * 16 <= (size_t)kLdsBytes) {
        mode = 8;
        lds_bytes = stack_bytes2 + (size_t)kTinyEnd * 16;
      }
      // the scene twice per CU, six waves per SIMD (MODE 13): two blocks of kBlock6 threads, 16-bit stack entries and
      // references (the first primitive of a leaf below 2048: kTinySphs and kTinyTris are)
      const bool tiny6 = ctx->tiny6 >= 0 ? ctx->tiny6 != 0 : kTiny6Default;
      const size_t stack_bytes6 = (size_t)stack_lds * kBlock6 * sizeof(uint16_t);
      static_assert(kTinySphs < 2048 && kTinyTris < 2048 && kTinyNodes * 8 < 32768, "MODE 13: 16-bit references");
      if (mode == 7 && tiny6 && (size_t)kTinyRefBytes + stack_bytes6 + (size_t)kTinyEnd6 * 16 <= (size_t)kLdsBytes6) {
        mode = 13;
        lds_bytes = (size_t)kTinyRefBytes + stack_bytes6 + (size_t)kTinyEnd6 * 16;
      }
    }
  }
  if (wide || ctx->d_nodes48) {  // never a MODE 2 tree: RTGPU_NODE48_MIN_NODES > kSmallNodes
    if (mode == 2 || mode == 7 || mode == 8 || mode == 13) { ctx->err = "internal: packed nodes on a tree that fits the LDS cache"; return 2; }
    mode += wide ? 5 : 3;
  }
  if (slots) mode += 6;  // 5 -> 11, 6 -> 12
  // Trees far larger than the caches: the four-wide nodes in 128-byte slots (modes 9, 10).  The padded copy is made here, by
  // the first frame after an upload / build / refit; the entry point that may not allocate uses it only if it exists already.
  if ((mode == 5 || mode == 6) && !wide_cache) {
    const uint64_t n4 = ctx->host.n_nodes4, bytes96 = n4 * sizeof(rtgpu_node4), bytes128 = n4 * 128;
    const bool want_pad = (ctx->pad_nodes4 >= 0 ? ctx->pad_nodes4 != 0 : bytes96 >= kPadMinBytes) && n4 > 0 && bytes128 < (1ull << 32);
    if (want_pad && (sc.may_allocate || (ctx->pad_valid && ctx->nodes4_pad_bytes >= bytes128))) {
      if (!ctx->pad_valid || ctx->nodes4_pad_bytes < bytes128) {
        if (ctx->nodes4_pad_bytes < bytes128) {
          HIP_TRY(ctx, hipStreamSynchronize(stream));
          (void)hipFree(ctx->d_nodes4_pad);
          ctx->d_nodes4_pad = nullptr;
          ctx->nodes4_pad_bytes = 0;
          HIP_TRY(ctx, hipMalloc(&ctx->d_nodes4_pad, bytes128));
          ctx->nodes4_pad_bytes = bytes128;
        }
        const uint64_t blocks = (n4 * 8 + 255) / 256;
        hipLaunchKernelGGL(pad_nodes4_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, stream,
                           (const float4 *)ctx->d_nodes4, (float4 *)ctx->d_nodes4_pad, n4);
        HIP_TRY(ctx, hipGetLastError());
        ctx->pad_valid = true;
      }
      P.nodes = (const float4 *)ctx->d_nodes4_pad;
      mode += 4;  // 5 -> 9, 6 -> 10
    }
  }
  if (wide_cache) {  // mode == 5
    P.n_cached = (uint32_t)(ctx->host.n_nodes4 < (uint64_t)kWideCached ? ctx->host.n_nodes4 : (uint64_t)kWideCached);
    lds_bytes = stack_bytes + (size_t)kWideCached * sizeof(rtgpu_node4);
  }
  if (mode != 2 && mode != 7 && mode != 8 && mode != 13 && !ctx->no_mats_cache && ctx->host.n_materials > 0 &&
      ctx->host.n_materials <= (uint64_t)kLdsMats && kBlocksPerCU == 1) {
    const size_t mb = (size_t)ctx->host.n_materials * sizeof(rtgpu_material);
    while (lds_bytes + mb > lds_budget && !P.stack_lds_only && sc.may_allocate && P.stack_lds > 8u) {
      // an overflow-capable stack gives up LDS entries for the table (one where a block has 1024 threads, two where it has 256)
      P.stack_lds -= 1u;
      lds_bytes -= (size_t)block_threads * sizeof(uint32_t);
    }
    if (lds_bytes + mb <= lds_budget) {
      P.mats_cached = (uint32_t)ctx->host.n_materials;
      lds_bytes += mb;
    }
  }

  bool want_stats = (s->flags & (RTGPU_FLAG_STATS | RTGPU_FLAG_PROFILE)) != 0;
  HIP_TRY(ctx, hipMemsetAsync(sc.counters, 0, kCounterBytes, stream));
  ctx->timed = false;
  if (n_items > 0) {
    const int inst = (s->flags & RTGPU_FLAG_PROFILE) ? 2 : want_stats ? 1 : 0;
    const unsigned block = mode == 8 ? (unsigned)kBlock2 : mode == 13 ? (unsigned)kBlock6 : block_threads;
    const uint64_t per_block = mode == 8 ? 2ull * kBlock2 : (uint64_t)block;   // work items a block holds at a time
    uint64_t want = (n_items + per_block - 1) / per_block;
    // (slot-tree modes: five blocks of 256 threads per CU; their counting and profile variants need more registers: four)
    // (MODE 13: two blocks of 768 threads per CU; its counting and profile variants have the registers of three waves per SIMD: one)
    const uint64_t per_cu = slots ? (uint64_t)(inst == 0 ? kSlotBlocksPerCU : kSlotBlocksPerCU - 1)
                            : mode == 13 ? (uint64_t)(inst == 0 ? 2 : 1) : (uint64_t)ctx->blocks_per_cu[want_stats ? 1 : 0];
    uint64_t cap = (uint64_t)ctx->num_cus * per_cu;
    unsigned grid = (unsigned)(want < cap ? want : cap);
    if ((size_t)grid * block > ctx->ovf_threads) { ctx->err = "internal: grid exceeds the overflow stack"; return 2; }
    ctx->last_waves = (uint64_t)grid * (block / 64u);
    // the end of the queue in small draws: a quarter of an item per lane of the grid x RTGPU_TAIL_ITEMS_PER_LANE_X4
    auto n_big_of = [&](uint64_t items, unsigned grid_) {
      const uint64_t tail = ((uint64_t)grid_ * block * RTGPU_TAIL_ITEMS_PER_LANE_X4 / 4u + 63u) & ~63ull;
      return (uint32_t)(items > tail ? items - tail : 0u);
    };
    P.n_big = n_big_of(n_items, grid);
    {  // ticks of 10 ns per draw of 64 from which the grid's draws of 32 (16) stay below RTGPU_TAIL_ATOMICS_PER_US
      const double waves = (double)grid * (block / 64u);
      P.tail_thr32 = (uint32_t)(100.0 * waves * 2.0 / RTGPU_TAIL_ATOMICS_PER_US);
      P.tail_thr16 = (uint32_t)(100.0 * waves * 4.0 / RTGPU_TAIL_ATOMICS_PER_US);
      if (const char *env = std::getenv("RTGPU_TAIL_DRAW")) {  // experiments: 16 / 32 / 64 for every wave
        const int d = std::atoi(env);
        P.tail_thr16 = d == 16 ? 0u : 0xFFFFFFFFu;
        P.tail_thr32 = d == 16 || d == 32 ? 0u : 0xFFFFFFFFu;
      }
    }
    const Variant &var = kVariants[ctx->variant >= 0 ? ctx->variant : (mode == 2 || mode == 7 || mode == 8 || mode == 13 ? kVariantSmallTree : kVariantLargeTree)];
    static_assert(RTGPU_NODE48_MIN_NODES > (unsigned)kSmallNodes, "a packed tree must not fit the LDS node cache");
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, stream));
    KernelFn fn = var.fn[inst][mode];
    std::snprintf(ctx->kernel_name, sizeof(ctx->kernel_name), "trace_kernel<%d,%d,%d>", inst, var.refill, mode);
    float4 *const run = planes ? planes + (size_t)round_chunks * n_px_out : nullptr;  // the running sum of the rounds so far
    uint32_t first_chunk = 0;
    for (uint32_t r = 0; r < n_rounds; r++) {
      // kRoundChunks chunks per round; the kernel writes a one-chunk launch straight to the frame, so a round never has ONE
      // chunk: where one would be left over, the round before gives it a second
      uint32_t chunks_r = n_chunks - first_chunk < kRoundChunks ? n_chunks - first_chunk : kRoundChunks;
      if (n_rounds > 1 && n_chunks - first_chunk - chunks_r == 1u) chunks_r -= 1u;
      unsigned grid_r = grid;
      if (n_rounds > 1) {
        // the round as a frame of its own: its share of the samples (the RNG is keyed on sample_offset + sample, the last chunk
        // is cut at spp), its number of work items, a fresh queue head (the counters go on counting); the first sample's
        // primary hit belongs to the first round
        const uint64_t items_r = n_px_items * chunks_r, want_r = (items_r + per_block - 1) / per_block;
        P.sample_offset = s->sample_offset + first_chunk * s->chunk_samples;
        P.spp = s->spp - first_chunk * s->chunk_samples;
        P.n_chunks = chunks_r; P.inv_n_chunks = 1.0f / (float)chunks_r;
        P.n_items = (uint32_t)items_r;
        P.n_big = n_big_of(items_r, (unsigned)(want_r < cap ? want_r : cap));
        P.groups_small = (items_r >> 6) < (1ull << 22) ? 1u : 0u;
        if (r > 0) P.prim_id = nullptr;
        grid_r = (unsigned)(want_r < cap ? want_r : cap);
        if (r > 0) {
          HIP_TRY(ctx, hipMemsetAsync(sc.counters, 0, sizeof(unsigned long long), stream));
          HIP_TRY(ctx, hipMemsetAsync(sc.counters + kTailHead, 0, sizeof(unsigned long long), stream));
        }
      }
      hipLaunchKernelGGL(fn, dim3(grid_r), dim3(block), lds_bytes, stream, P);
      HIP_TRY(ctx, hipGetLastError());
      if (r + 1 == n_rounds) HIP_TRY(ctx, hipEventRecord(ctx->ev1, stream));  // (of a frame in rounds: with the combines between them)
      if (n_chunks > 1) {
        uint64_t blocks = (n_px_out + 255) / 256;
        unsigned cgrid = (unsigned)(blocks < 4096 ? blocks : 4096);
        hipLaunchKernelGGL(combine_kernel, dim3(cgrid), dim3(256), 0, stream, (const float4 *)planes,
                           (float4 *)d_accum_rgba, n_px_out, chunks_r, n_px_out, (float)s->spp,
                           (uint32_t)((s->flags & RTGPU_FLAG_ACCUMULATE) ? 1u : 0u),
                           CombineGeom{s->width, s->height, ts_log2, g.tiles_x, s->rank, s->world_size,
                                       (uint32_t)((s->flags & RTGPU_FLAG_COMPACT) ? 1u : 0u)},
                           (const float4 *)(r > 0 ? run : nullptr), r + 1 < n_rounds ? run : (float4 *)nullptr);
        HIP_TRY(ctx, hipGetLastError());
      }
      first_chunk += chunks_r;
    }
    ctx->timed = true;
  }
  HIP_TRY(ctx, hipEventRecord(ctx->ev_last, stream));
  ctx->last_stream = stream;
  ctx->have_last = true;
  if (stats) {
    unsigned long long c[8];
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    HIP_TRY(ctx, hipMemcpy(c, sc.counters, sizeof(c), hipMemcpyDeviceToHost));
    stats->paths = c[1]; stats->rays = c[2]; stats->nodes = c[3];
    stats->tri_tests = c[4]; stats->sphere_tests = c[5]; stats->scatters = c[6];
  }
  return 0;
}
}  // namespace

extern "C" int rtgpu_profile_counters(rtgpu_ctx *ctx, uint64_t out[32]) {
  if (!ctx || !out) return 1;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(out, ctx->last_counters ? ctx->last_counters : ctx->d_counters, 256, hipMemcpyDeviceToHost));
  return 0;
}

// Exhaustive self-check of rcp_fast / sqrt_fast / rcp_exact / sqrt_exact against hipcc's
// IEEE sequences: every one of the 2^32 bit patterns.  out[0], out[1]: inputs INSIDE the
// verified ranges on which rcp_fast / sqrt_fast differ; out[2], out[3]: inputs (any) on
// which the guarded rcp_exact / sqrt_exact differ.  All four must be zero.
__global__ __launch_bounds__(256) void check_exact_math_kernel(unsigned long long *out) {
  const unsigned long long i0 = ((unsigned long long)blockIdx.x * 256ull + threadIdx.x) * 256ull;
  unsigned long long bad[4] = {0, 0, 0, 0};
  for (unsigned k = 0; k < 256; k++) {
    const uint32_t bits = (uint32_t)(i0 + k);
    const float x = __uint_as_float(bits), a = __builtin_fabsf(x);
    auto differ = [](float p, float q) { return __float_as_uint(p) != __float_as_uint(q) && !(p != p && q != q); };
    const float want_r = 1.0f / x;
    if (a >= 0x1p-126f && a < 0x1p126f && differ(rcp_fast(x), want_r)) bad[0]++;
    if (differ(rcp_exact(x), want_r)) bad[2]++;
    if (!(bits & 0x80000000u) || x != x) {  // sqrt: non-negative operands and NaNs
      const float want_s = __builtin_sqrtf(x);
      if ((x == 0.0f || x >= 0x1p-104f || x != x) && differ(sqrt_fast(x), want_s)) bad[1]++;
      if (differ(sqrt_exact(x), want_s)) bad[3]++;
    }
  }
  for (int c = 0; c < 4; c++)
    if (bad[c]) atomicAdd(&out[c], bad[c]);
}

// Diagnostics (not in the public header): see check_exact_math_kernel.
extern "C" int rtgpu_debug_check_exact_math(rtgpu_ctx *ctx, uint64_t out[4]) {
  if (!ctx || !out) return 1;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemset(ctx->d_counters, 0, 256));
  hipLaunchKernelGGL(check_exact_math_kernel, dim3((1u << 24) / 256), dim3(256), 0, 0, ctx->d_counters);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(out, ctx->d_counters, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return 0;
}

// Diagnostics (not in the public header): blocks of the production variant (MODE 1) the
// runtime says fit on one CU with `lds_bytes` of dynamic LDS each.
extern "C" int rtgpu_debug_occupancy(rtgpu_ctx *ctx, uint64_t lds_bytes, int *blocks_per_cu) {
  if (!ctx || !blocks_per_cu) return 1;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, (const void *)kVariants[ctx->variant >= 0 ? ctx->variant : kVariantLargeTree].fn[0][1],
                                                            kBlock, (size_t)lds_bytes));
  return 0;
}

// Diagnostics (not in the public header): the slot tree of the uploaded scene as the trace loop would traverse it (made now if
// the last upload / build / refit has not been rendered yet).  *n_slots = 0: the scene has none (no four-wide nodes, or no
// slot form).  Copies min(*n_slots, max_slots) slots of 16 words to out.
extern "C" int rtgpu_debug_slot_tree(rtgpu_ctx *ctx, uint32_t *out, uint64_t max_slots, uint64_t *n_slots) {
  if (!ctx || !n_slots) return 1;
  *n_slots = 0;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  if (!ctx->d_nodes4) return 0;
  if (ctx->slot_state == 3) {
    if (rtgpu::refresh_slot_tree_device(ctx->d_nodes4, ctx->d_tris, ctx->d_sphs, ctx->d_sph_meta, &ctx->slot, &ctx->pool, nullptr, ctx->err))
      return 2;
    ctx->slot_state = ctx->slot.n_slots > 0 ? 1 : 2;
  }
  if (ctx->slot_state == 0) {
    if (rtgpu::build_slot_tree_device(ctx->d_nodes4, ctx->host.n_nodes4, ctx->d_tris, ctx->host.n_tris, ctx->d_sphs, ctx->d_sph_meta,
                                      ctx->host.n_spheres, &ctx->slot, &ctx->pool, nullptr, ctx->err))
      return 2;
    ctx->slot_state = ctx->slot.n_slots > 0 ? 1 : 2;
  }
  if (ctx->slot_state != 1) return 0;
  *n_slots = ctx->slot.n_slots;
  const uint64_t n = ctx->slot.n_slots < max_slots ? ctx->slot.n_slots : max_slots;
  if (out && n) HIP_TRY(ctx, hipMemcpy(out, ctx->slot.d_slots, (size_t)n * 64, hipMemcpyDeviceToHost));
  return 0;
}

// Diagnostics (not in the public header): {start, end, queue-dry} 100 MHz timestamps and
// the outer iterations after queue-dry of every wave of the last RTGPU_FLAG_PROFILE launch
// (4 u64 per wave).
extern "C" int rtgpu_debug_wave_times(rtgpu_ctx *ctx, uint64_t *out, uint64_t max_waves, uint64_t *n_waves) {
  if (!ctx || !out || !n_waves) return 1;
  uint64_t n = ctx->last_waves ? ctx->last_waves : (uint64_t)ctx->num_cus * kBlocksPerCU * (kBlock / 64);
  if (n > max_waves) n = max_waves;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(out, ctx->d_wave_times, n * 4 * 8, hipMemcpyDeviceToHost));
  *n_waves = n;
  return 0;
}

// Internal (rtgpu_internal.h, used by group.cpp): where the context's launches keep their counters
// ([1..6] as rtgpu_stats), so that a pipelined caller can copy them out in stream order.
extern "C" int rtgpu_internal_counters(rtgpu_ctx *ctx, uint64_t **d_counters) {
  if (!ctx || !d_counters) return 1;
  *d_counters = (uint64_t *)ctx->d_counters;
  return 0;
}

extern "C" void rtgpu_internal_abandon(rtgpu_ctx *ctx) {
  if (!ctx) return;
  rtgpu::free_flat_scene(&ctx->host);  // host memory only; device buffers, events and streams are leaked on purpose
  delete ctx;
}

extern "C" const char *rtgpu_last_kernel_name(const rtgpu_ctx *ctx) { return ctx ? ctx->kernel_name : ""; }

extern "C" int rtgpu_last_kernel_ms(rtgpu_ctx *ctx, float *ms) {
  if (!ctx || !ms) return 1;
  if (!ctx->timed) { ctx->err = "no timed kernel launch on this context yet"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
  HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  return 0;
}

namespace {
int ensure(rtgpu_ctx *ctx, void **p, size_t *have, size_t want) {
  if (*have >= want) return 0;
  (void)hipFree(*p);
  *p = nullptr;
  *have = 0;
  HIP_TRY(ctx, hipMalloc(p, want));
  *have = want;
  return 0;
}
}  // namespace

extern "C" int rtgpu_render(rtgpu_ctx *ctx, const rtgpu_camera *cam, const rtgpu_settings *s,
                            float *accum_rgba, uint32_t *prim_id, rtgpu_stats *stats) {
  if (!ctx) return 1;
  uint32_t ts_log2 = 0;
  if (check_settings(ctx, cam, s, &ts_log2)) return 1;
  if (!accum_rgba) { ctx->err = "accum output pointer is NULL"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint64_t n_px = (s->flags & RTGPU_FLAG_COMPACT)
                      ? rtgpu_compact_pixels(s->width, s->height, s->tile_size, s->rank, s->world_size)
                      : (uint64_t)s->width * s->height;
  if (n_px == 0) return 0;
  if (ensure(ctx, &ctx->d_accum, &ctx->d_accum_bytes, n_px * 16)) return 2;
  if (prim_id && ensure(ctx, &ctx->d_ids, &ctx->d_ids_bytes, n_px * 4)) return 2;
  if (s->flags & RTGPU_FLAG_ACCUMULATE)
    HIP_TRY(ctx, hipMemcpy(ctx->d_accum, accum_rgba, n_px * 16, hipMemcpyHostToDevice));
  else
    HIP_TRY(ctx, hipMemset(ctx->d_accum, 0, n_px * 16));
  if (prim_id) HIP_TRY(ctx, hipMemset(ctx->d_ids, 0xFF, n_px * 4));
  int rc = rtgpu_render_device(ctx, cam, s, ctx->d_accum, prim_id ? ctx->d_ids : nullptr, nullptr, stats);
  if (rc) return rc;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(accum_rgba, ctx->d_accum, n_px * 16, hipMemcpyDeviceToHost));
  if (prim_id) HIP_TRY(ctx, hipMemcpy(prim_id, ctx->d_ids, n_px * 4, hipMemcpyDeviceToHost));
  return 0;
}

// Progressive frame loop without the PCIe round trip of the float frame (include/rtgpu.h): the
// accumulation stays in the context, the call returns the resolved 0xRRGGBBAA frame.
extern "C" int rtgpu_render_resolved(rtgpu_ctx *ctx, const rtgpu_camera *cam, const rtgpu_settings *s,
                                     uint32_t *rgba8, rtgpu_stats *stats) {
  if (!ctx) return 1;
  uint32_t ts_log2 = 0;
  if (check_settings(ctx, cam, s, &ts_log2)) return 1;
  if (s->flags & RTGPU_FLAG_COMPACT) { ctx->err = "rtgpu_render_resolved works on whole frames (no RTGPU_FLAG_COMPACT)"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint64_t n_px = (uint64_t)s->width * s->height;
  if (ensure(ctx, &ctx->d_res_rgba, &ctx->d_res_rgba_bytes, n_px * 4)) return 2;
  if (ctx->h_res_rgba_bytes < n_px * 4) {
    if (ctx->h_res_rgba) (void)hipHostFree(ctx->h_res_rgba);
    ctx->h_res_rgba = nullptr;
    ctx->h_res_rgba_bytes = 0;
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_res_rgba, n_px * 4, hipHostMallocDefault));
    ctx->h_res_rgba_bytes = n_px * 4;
  }
  const bool fresh = !(s->flags & RTGPU_FLAG_ACCUMULATE) || !ctx->res_valid || ctx->res_w != s->width || ctx->res_h != s->height ||
                     ctx->d_res_accum_bytes < n_px * 16;
  if (ctx->d_res_accum_bytes < n_px * 16) {
    ctx->res_valid = false;
    if (ensure(ctx, &ctx->d_res_accum, &ctx->d_res_accum_bytes, n_px * 16)) return 2;
  }
  rtgpu_settings st = *s;
  if (fresh) {  // nothing to continue: this frame starts the sum (a world > 1 rank leaves foreign pixels at 0)
    st.flags &= ~(uint32_t)RTGPU_FLAG_ACCUMULATE;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_res_accum, 0, n_px * 16, ctx->stream));
  }
  ctx->res_valid = false;
  int rc = rtgpu_render_device(ctx, cam, &st, ctx->d_res_accum, nullptr, ctx->stream, nullptr);
  if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
  rc = rtgpu_resolve_rgba8_device(ctx, ctx->d_res_accum, ctx->d_res_rgba, n_px, ctx->stream);
  if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
  HIP_TRY(ctx, hipMemcpyAsync(ctx->h_res_rgba, ctx->d_res_rgba, n_px * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->res_w = s->width; ctx->res_h = s->height;
  ctx->res_valid = true;
  if (rgba8) std::memcpy(rgba8, ctx->h_res_rgba, n_px * 4);
  if (stats) {
    unsigned long long c[8];
    HIP_TRY(ctx, hipMemcpy(c, ctx->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    stats->paths = c[1]; stats->rays = c[2]; stats->nodes = c[3];
    stats->tri_tests = c[4]; stats->sphere_tests = c[5]; stats->scatters = c[6];
  }
  return 0;
}

extern "C" const uint32_t *rtgpu_resolved_frame(const rtgpu_ctx *ctx) {
  return ctx && ctx->res_valid ? ctx->h_res_rgba : nullptr;
}

extern "C" int rtgpu_resident_accum(rtgpu_ctx *ctx, float *accum_rgba, uint64_t n_pixels) {
  if (!ctx) return 1;
  if (!ctx->res_valid) { ctx->err = "no accumulation is resident (call rtgpu_render_resolved first)"; return 1; }
  if (!accum_rgba || n_pixels != (uint64_t)ctx->res_w * ctx->res_h) { ctx->err = "rtgpu_resident_accum: buffer is NULL or not width*height pixels of the resident frame"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpy(accum_rgba, ctx->d_res_accum, n_pixels * 16, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" void rtgpu_resident_reset(rtgpu_ctx *ctx) { if (ctx) ctx->res_valid = false; }

// Non-blocking form of rtgpu_render: the frame runs on the context's own stream
// while the caller's loop keeps going (SURVEY.md §8b "blocking or begin/poll").
extern "C" int rtgpu_render_begin(rtgpu_ctx *ctx, const rtgpu_camera *cam, const rtgpu_settings *s,
                                  float *accum_rgba, uint32_t *prim_id) {
  if (!ctx) return 1;
  uint32_t ts_log2 = 0;
  if (check_settings(ctx, cam, s, &ts_log2)) return 1;
  if (!accum_rgba) { ctx->err = "accum output pointer is NULL"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint64_t n_px = (s->flags & RTGPU_FLAG_COMPACT)
                      ? rtgpu_compact_pixels(s->width, s->height, s->tile_size, s->rank, s->world_size)
                      : (uint64_t)s->width * s->height;
  if (n_px) {
    if (ensure(ctx, &ctx->d_accum, &ctx->d_accum_bytes, n_px * 16)) return 2;
    if (prim_id && ensure(ctx, &ctx->d_ids, &ctx->d_ids_bytes, n_px * 4)) return 2;
    if (s->flags & RTGPU_FLAG_ACCUMULATE)
      HIP_TRY(ctx, hipMemcpyAsync(ctx->d_accum, accum_rgba, n_px * 16, hipMemcpyHostToDevice, ctx->stream));
    else
      HIP_TRY(ctx, hipMemsetAsync(ctx->d_accum, 0, n_px * 16, ctx->stream));
    if (prim_id) HIP_TRY(ctx, hipMemsetAsync(ctx->d_ids, 0xFF, n_px * 4, ctx->stream));
    int rc = rtgpu_render_device(ctx, cam, s, ctx->d_accum, prim_id ? ctx->d_ids : nullptr, ctx->stream, nullptr);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
  }
  HIP_TRY(ctx, hipEventRecord(ctx->ev_done, ctx->stream));
  ctx->pending = true;
  ctx->pend_accum = accum_rgba;
  ctx->pend_ids = prim_id;
  ctx->pend_px = n_px;
  return 0;
}

extern "C" int rtgpu_render_poll(rtgpu_ctx *ctx, int *done) {
  if (!ctx || !done) return 1;
  *done = 0;
  if (!ctx->pending) { ctx->err = "no frame is pending (call rtgpu_render_begin first)"; return 1; }
  hipError_t e = hipEventQuery(ctx->ev_done);
  if (e == hipSuccess) { *done = 1; return 0; }
  if (e == hipErrorNotReady) return 0;
  hip_fail(ctx, e, "hipEventQuery");
  return 2;
}

extern "C" int rtgpu_render_end(rtgpu_ctx *ctx, rtgpu_stats *stats) {
  if (!ctx) return 1;
  if (!ctx->pending) { ctx->err = "no frame is pending (call rtgpu_render_begin first)"; return 1; }
  ctx->pending = false;  // whatever happens below, the frame is over
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipEventSynchronize(ctx->ev_done));
  if (ctx->pend_px) {
    HIP_TRY(ctx, hipMemcpy(ctx->pend_accum, ctx->d_accum, ctx->pend_px * 16, hipMemcpyDeviceToHost));
    if (ctx->pend_ids) HIP_TRY(ctx, hipMemcpy(ctx->pend_ids, ctx->d_ids, ctx->pend_px * 4, hipMemcpyDeviceToHost));
  }
  if (stats) {
    unsigned long long c[8];
    HIP_TRY(ctx, hipMemcpy(c, ctx->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    stats->paths = c[1]; stats->rays = c[2]; stats->nodes = c[3];
    stats->tri_tests = c[4]; stats->sphere_tests = c[5]; stats->scatters = c[6];
  }
  return 0;
}

extern "C" int rtgpu_resolve_rgba8_device(rtgpu_ctx *ctx, const void *d_accum_rgba, void *d_rgba8,
                                          uint64_t n_pixels, void *stream) {
  if (!ctx) return 1;
  if (!d_accum_rgba || !d_rgba8) { ctx->err = "resolve: NULL pointer"; return 1; }
  if (n_pixels == 0) return 0;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint64_t blocks = (n_pixels + 255) / 256;
  unsigned grid = (unsigned)(blocks < 2048 ? blocks : 2048);
  hipLaunchKernelGGL(resolve_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     (const float4 *)d_accum_rgba, (uint32_t *)d_rgba8, n_pixels);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

extern "C" int rtgpu_resolve_rgba8(rtgpu_ctx *ctx, const float *accum_rgba, uint32_t *rgba8,
                                   uint64_t n_pixels) {
  if (!ctx) return 1;
  if (ctx->pending) { ctx->err = "a frame begun with rtgpu_render_begin is pending (call rtgpu_render_end)"; return 1; }
  if (!accum_rgba || !rgba8) { ctx->err = "resolve: NULL pointer"; return 1; }
  if (n_pixels == 0) return 0;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ensure(ctx, &ctx->d_accum, &ctx->d_accum_bytes, n_pixels * 16)) return 2;
  if (ensure(ctx, &ctx->d_ids, &ctx->d_ids_bytes, n_pixels * 4)) return 2;
  HIP_TRY(ctx, hipMemcpy(ctx->d_accum, accum_rgba, n_pixels * 16, hipMemcpyHostToDevice));
  int rc = rtgpu_resolve_rgba8_device(ctx, ctx->d_accum, ctx->d_ids, n_pixels, nullptr);
  if (rc) return rc;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(rgba8, ctx->d_ids, n_pixels * 4, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int rtgpu_untile_device(rtgpu_ctx *ctx, const void *d_gathered, void *d_image,
                                   uint32_t width, uint32_t height, uint32_t tile_size,
                                   uint32_t world_size, uint32_t elem_bytes, void *stream) {
  if (!ctx) return 1;
  if (!d_gathered || !d_image) { ctx->err = "untile: NULL pointer"; return 1; }
  uint32_t l = 0;
  while ((1u << l) < tile_size && l < 31) l++;
  if (tile_size < 8 || (1u << l) != tile_size || world_size == 0 || width == 0 || height == 0) {
    ctx->err = "untile: bad tile_size/world_size/size"; return 1;
  }
  if (elem_bytes != 16 && elem_bytes != 4) { ctx->err = "untile: elem_bytes must be 16 or 4"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  rtgpu::TileGrid g = rtgpu::tile_grid(width, height, tile_size);
  uint64_t stride_px = rtgpu_compact_pixels(width, height, tile_size, 0, world_size);
  uint64_t n = (uint64_t)width * height;
  uint64_t blocks = (n + 255) / 256;
  unsigned grid = (unsigned)(blocks < 4096 ? blocks : 4096);
  if (elem_bytes == 16)
    hipLaunchKernelGGL(untile_kernel<float4>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)d_gathered, (float4 *)d_image, width, height, l, g.tiles_x,
                       world_size, stride_px);
  else
    hipLaunchKernelGGL(untile_kernel<uint32_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t *)d_gathered, (uint32_t *)d_image, width, height, l, g.tiles_x,
                       world_size, stride_px);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}
