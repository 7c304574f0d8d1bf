// pack48.h — rtgpu_node (64 B) -> rtgpu_node48 (48 B), include/rtgpu_format.h.
// One implementation for the host builder (bvh_build.cpp) and the device kernel
// (pack48.hip): pure integer work on the float bit patterns, so both produce the
// same bytes.  No reference counterpart (the reference has no BVH).
#ifndef RTGPU_PACK48_H
#define RTGPU_PACK48_H

#include <stdint.h>
#include <string.h>

#include "rtgpu_format.h"

#ifdef __HIPCC__
#define RTGPU_HD __host__ __device__
#else
#define RTGPU_HD
#endif

namespace rtgpu {

// order-preserving map of float bit patterns onto u32 (negative floats first)
RTGPU_HD inline uint32_t ord_enc(uint32_t b) { return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
RTGPU_HD inline uint32_t ord_dec(uint32_t e) { return (e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e; }

// The float nearest to `bits` on the conservative side (down: <=, else >=) whose
// low byte is `byte`.  False if `bits` is not finite or the result would not be.
RTGPU_HD inline bool embed_byte(uint32_t bits, uint32_t byte, bool down, uint32_t *out) {
  if ((bits & 0x7F800000u) == 0x7F800000u) return false;
  const uint32_t e0 = ord_enc(bits);
  for (int k = 0; k < 2; k++) {
    const int64_t block = (int64_t)(e0 >> 8) + (down ? -k : k);
    if (block < 0 || block > 0xFFFFFF) return false;
    const bool positive = ((block >> 23) & 1) != 0;  // top bit of the ordered value
    const uint32_t low = positive ? byte : (~byte & 0xFFu);  // negative floats are stored complemented
    const uint32_t e = ((uint32_t)block << 8) | low;
    if (down ? e <= e0 : e >= e0) {
      const uint32_t r = ord_dec(e);
      if ((r & 0x7F800000u) == 0x7F800000u) return false;
      *out = r;
      return true;
    }
  }
  return false;  // not reached: the neighbouring block always lies on the wanted side
}

RTGPU_HD inline bool pack_node48(const rtgpu_node &n, rtgpu_node48 *out) {
  uint32_t w[12];
  const float *f = n.lo0;  // lo0[3] hi0[3] lo1[3] hi1[3] are contiguous: 12 planes
  for (int i = 0; i < 12; i++) {
    union { float f; uint32_t u; } c;
    c.f = f[i];
    w[i] = c.u;
    if ((c.u & 0x7F800000u) == 0x7F800000u) return false;
  }
  for (int k = 0; k < 4; k++) {
    // word k: lo0.x lo0.y lo0.z (down) hi0.x (up); word 8 + k: lo1.z (down) hi1.x hi1.y hi1.z (up)
    if (!embed_byte(w[k], (n.child0 >> (8 * k)) & 0xFFu, k < 3, &w[k])) return false;
    if (!embed_byte(w[8 + k], (n.child1 >> (8 * k)) & 0xFFu, k < 1, &w[8 + k])) return false;
  }
  for (int i = 0; i < 12; i++) out->w[i] = w[i];
  return true;
}

}  // namespace rtgpu

#endif
