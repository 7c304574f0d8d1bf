// slot_tree.h — arithmetic of the slot tree (include/rtgpu_format.h: four-wide nodes with half-precision
// child boxes, tree and primitives in one array of 64-byte slots).  One implementation for the device
// builder (slot_tree.hip) and its host mirror: integer work on bit patterns and one IEEE double
// subtraction per plane, so both produce the same bytes.  No reference counterpart (the reference has no
// BVH).
#ifndef RTGPU_SLOT_TREE_H
#define RTGPU_SLOT_TREE_H

#include <stdint.h>
#include <string.h>

#include "pack48.h"
#include "rtgpu_format.h"

namespace rtgpu {

// The binary16 nearest to d on the wanted side (up: >= d, else <= d) among zero and the NORMAL halves
// (no subnormals: what v_fma_mix_f32 does with them depends on a mode register).  d must be >= 0 and
// finite.  False if no such half exists (up: d above 65504).
RTGPU_HD inline bool half_toward(double d, bool up, uint32_t *out) {
  if (!(d >= 0.0)) return false;
  uint64_t b;
  memcpy(&b, &d, sizeof(b));
  const int e = (int)((b >> 52) & 0x7FFu) - 1023;  // d = 1.m x 2^e (e = -1023: zero or a double subnormal)
  if (e < -14) {                                   // below the smallest normal half, 2^-14
    *out = (up && d > 0.0) ? 0x0400u : 0u;
    return true;
  }
  if (e > 15) {                                    // 65536 and above
    if (up) return false;
    *out = 0x7BFFu;                                // 65504, the largest half
    return true;
  }
  const uint64_t m = b & 0xFFFFFFFFFFFFFull;
  uint32_t h = ((uint32_t)(e + 15) << 10) | (uint32_t)(m >> 42);
  if (up && (m & ((1ull << 42) - 1ull)) != 0) h++;  // (a carry runs into the exponent as it should)
  if (h >= 0x7C00u) {
    if (up) return false;
    h = 0x7BFFu;
  }
  *out = h;
  return true;
}

// What one child of a four-wide node is in the slot tree
struct SlotChild {
  uint32_t ref;    // its reference in rtgpu_node4 (RTGPU_REF_EMPTY: none)
  uint32_t size;   // slots it takes in the family: 0 none, 1 inner node, n for a leaf of n primitives
  uint32_t nibble; // top four bits of its slot reference: leaf, sphere, count - 1 (two bits)
};
RTGPU_HD inline uint32_t node4_ref(const rtgpu_node4 &n, int k) {
  const uint32_t *w = &n.w[6 * k];
  return (w[0] & 0xFFu) | ((w[1] & 0xFFu) << 8) | ((w[2] & 0xFFu) << 16) | ((w[3] & 0xFFu) << 24);
}
// False if the child cannot be represented (a leaf of more than four primitives).
RTGPU_HD inline bool slot_child(const rtgpu_node4 &n, int k, SlotChild *c) {
  c->ref = node4_ref(n, k);
  c->size = 0; c->nibble = 0;
  if (c->ref == RTGPU_REF_EMPTY) return true;
  if (!(c->ref & RTGPU_REF_LEAF)) { c->size = 1; return true; }
  const uint32_t count = RTGPU_REF_COUNT(c->ref);
  if (count > 4u) return false;
  c->size = count;
  c->nibble = 8u | ((c->ref & RTGPU_REF_SPHERE) ? 4u : 0u) | (count - 1u);
  return true;
}

// The node slot of a four-wide node: family = slot of its first child, bytes as in rtgpu_format.h.
// False if a plane cannot be represented (not finite, or an offset above 65504).
RTGPU_HD inline bool make_node_slot(const rtgpu_node4 &n, uint32_t family, uint32_t out[16]) {
  SlotChild c[4];
  uint32_t off[4], o = 0;
  for (int k = 0; k < 4; k++) {
    if (!slot_child(n, k, &c[k])) return false;
    off[k] = o;
    o += c[k].size;
  }
  if (family > 0x00FFFFFFu || o > 16u) return false;
  // origin: per axis the smallest lo plane of the children that exist, then the nearest float below
  // that carries the child byte (children 0..2; child 3's byte rides in word 3)
  uint32_t org[3];
  for (int a = 0; a < 3; a++) {
    bool any = false;
    float lo = 0.f;
    for (int k = 0; k < 4; k++) {
      if (c[k].ref == RTGPU_REF_EMPTY) continue;
      float v;
      memcpy(&v, &n.w[6 * k + a], sizeof(v));
      if (!any || v < lo) lo = v;
      any = true;
    }
    if (!any) return false;
    uint32_t bits;
    memcpy(&bits, &lo, sizeof(bits));
    const uint32_t byte = (off[a] << 4) | c[a].nibble;
    if (!embed_byte(bits, byte, true, &org[a])) return false;
    out[a] = org[a];
  }
  out[3] = family | (((off[3] << 4) | c[3].nibble) << 24);
  for (int k = 0; k < 4; k++)
    for (int a = 0; a < 3; a++) {
      uint32_t word = 0x7C007C00u;  // missing child: +inf | +inf, which no ray reaches
      if (c[k].ref != RTGPU_REF_EMPTY) {
        float fo, flo, fhi;
        memcpy(&fo, &org[a], sizeof(fo));
        memcpy(&flo, &n.w[6 * k + a], sizeof(flo));
        memcpy(&fhi, &n.w[6 * k + 3 + a], sizeof(fhi));
        uint32_t hl, hh;
        if (!half_toward((double)flo - (double)fo, false, &hl)) return false;
        if (!half_toward((double)fhi - (double)fo, true, &hh)) return false;
        word = hl | (hh << 16);
      }
      out[4 + 3 * k + a] = word;
    }
  return true;
}

}  // namespace rtgpu

#endif
