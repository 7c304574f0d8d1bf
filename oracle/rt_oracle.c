/*
 * rt_oracle.c — CPU oracle for the per-pixel trace loop.  TEST INFRASTRUCTURE:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  The product (librtgpu.so) never links, loads or calls it.
 *
 * PARITY UNPINNED BY THE REFERENCE.  The reference snapshot (Condzi/rt @ v1)
 * contains no ray tracer, no RNG, no scene and no test or golden vector for
 * this path.  This file is therefore the written
 * specification of the trace loop, not a restatement of reference code, for
 * everything except the vector arithmetic, which IS present in the reference
 * and is followed operation for operation:
 *
 *   v3_dot        code/math/vec.cxx:94-97   multiply, then x + y + z left to right
 *   v3_len        code/math/vec.cxx:52-54   sqrt(dot(v, v))
 *   v3_normalize  code/math/vec.cxx:66-69   v *= 1/len(v)  (reciprocal, then multiply)
 *   v3 + - *      code/math/vec.cxx:174-211 component-wise, scalar on the right
 *   rgba8 packing run_tree/data/shaders/shaders.hlsl:18-25, code/gfx/gfx.hpp:4-6
 *   y-up rows     code/math/mat.cxx:104-129
 *
 * Those rows are pinned against the reference's own code compiled from
 * /root/reference (oracle/_ref, see oracle/Makefile) and against the vectors
 * committed in tests/golden/ref_math.json.
 *
 * Material semantics follow the literature the reference cites as its study
 * material (notes/materials.md:3-5): P. Shirley, "Ray Tracing in One Weekend"
 * — lambertian, metal with fuzz, dielectric with Schlick, sky gradient.
 *
 * ARITHMETIC CONTRACT (both this file and the HIP kernel):
 *   IEEE-754 binary32, round to nearest even; no contraction (compile with
 *   -ffp-contract=off) — the ONLY fused operations are the explicit fmaf()
 *   calls of the slab test; division and sqrt correctly rounded; fminf/fmaxf
 *   only ever see non-NaN operands; no transcendental functions.
 *   The random stream is a counter-based integer hash, so a path's result does
 *   not depend on scheduling, tile order or GPU count.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "rtgpu.h"        /* POD structs of the C ABI (data only) */
#include "rtgpu_format.h" /* flattened scene layout (data only)   */

#ifdef _OPENMP
#include <omp.h>
#endif

#define RTO_API __attribute__((visibility("default")))

/* ---- RNG (row a7) ---------------------------------------------------------- */

/* PCG-RXS-M-XS 32-bit permutation (M. O'Neill, "PCG", 2014): a bijection on u32. */
static inline uint32_t pcg(uint32_t v) {
  uint32_t s = v * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
  return (w >> 22) ^ w;
}

/* One key per (pixel, sample): bijective in `sample` for a fixed pixel, so two
 * samples of one pixel never share a stream. */
static inline uint32_t path_key(uint32_t pixel, uint32_t sample, uint32_t seed) {
  return pcg(sample ^ pcg(pixel ^ pcg(seed)));
}

typedef struct { uint32_t key, ctr; } rng_t;

static inline float u01(rng_t *r) {
  uint32_t h = pcg(r->key + r->ctr * 0x9E3779B9u);
  r->ctr++;
  return (float)(h >> 8) * 0x1p-24f;
}

RTO_API uint32_t rto_pcg(uint32_t v) { return pcg(v); }
RTO_API uint32_t rto_path_key(uint32_t pixel, uint32_t sample, uint32_t seed) {
  return path_key(pixel, sample, seed);
}
RTO_API float rto_u01(uint32_t key, uint32_t ctr) {
  rng_t r = {key, ctr};
  return u01(&r);
}

/* ---- vector arithmetic (row a9) ---------------------------------------------- */

typedef struct { float x, y, z; } v3;

static inline v3 v3_add(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static inline v3 v3_sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline v3 v3_mul(v3 a, v3 b) { v3 r = {a.x * b.x, a.y * b.y, a.z * b.z}; return r; }
static inline v3 v3_scale(v3 a, float s) { v3 r = {a.x * s, a.y * s, a.z * s}; return r; }
static inline float v3_dot(v3 a, v3 b) {
  a = v3_mul(a, b);
  return a.x + a.y + a.z;
}
static inline float v3_len(v3 a) { return sqrtf(v3_dot(a, a)); }
static inline v3 v3_normalize(v3 a) { return v3_scale(a, 1.0f / v3_len(a)); }
/* not in the reference (code/math/vec.hxx has no cross): mul, mul, sub per lane */
static inline v3 v3_cross(v3 a, v3 b) {
  v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
  return r;
}
static inline v3 v3_neg(v3 a) { v3 r = {-a.x, -a.y, -a.z}; return r; }
static inline v3 v3_from(const float *p) { v3 r = {p[0], p[1], p[2]}; return r; }

RTO_API float rto_dot3(const float *a, const float *b) { return v3_dot(v3_from(a), v3_from(b)); }
RTO_API float rto_len3(const float *a) { return v3_len(v3_from(a)); }
RTO_API void rto_normalize3(const float *a, float *out) {
  v3 r = v3_normalize(v3_from(a));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
RTO_API void rto_cross3(const float *a, const float *b, float *out) {
  v3 r = v3_cross(v3_from(a), v3_from(b));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
/* rt::transformed_point, row-vector convention (code/math/mat.cxx:12-20) */
RTO_API void rto_transform_point4(const float *p, const float *m /*4x4 row major*/, float *out) {
  for (int c = 0; c < 4; c++)
    out[c] = p[0] * m[0 * 4 + c] + p[1] * m[1 * 4 + c] + p[2] * m[2 * 4 + c] + p[3] * m[3 * 4 + c];
}

/* ---- intersection primitives (rows a3-a5) ------------------------------------ */

typedef struct {
  v3 o, d;        /* d is unit length (normalised at generation and after scatter) */
  v3 inv;         /* 1/d with |d| clamped away from 0                              */
  v3 noi;         /* -(o * inv), the addend of the fma-form slab test               */
} ray_t;

static inline float clamp_dir(float d) {
  return fabsf(d) < 1e-30f ? copysignf(1e-30f, d) : d;
}

static inline void ray_prepare(ray_t *r) {
  r->inv.x = 1.0f / clamp_dir(r->d.x);
  r->inv.y = 1.0f / clamp_dir(r->d.y);
  r->inv.z = 1.0f / clamp_dir(r->d.z);
  r->noi.x = -(r->o.x * r->inv.x);
  r->noi.y = -(r->o.y * r->inv.y);
  r->noi.z = -(r->o.z * r->inv.z);
}

/* Slab test.  Returns hit and the entry distance used for near-first ordering. */
static inline int slab(const ray_t *r, const float *lo, const float *hi, float tmin, float tbest,
                       float *tnear_out) {
  float t0x = fmaf(lo[0], r->inv.x, r->noi.x), t1x = fmaf(hi[0], r->inv.x, r->noi.x);
  float t0y = fmaf(lo[1], r->inv.y, r->noi.y), t1y = fmaf(hi[1], r->inv.y, r->noi.y);
  float t0z = fmaf(lo[2], r->inv.z, r->noi.z), t1z = fmaf(hi[2], r->inv.z, r->noi.z);
  float tnear = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
  float tfar  = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tbest));
  *tnear_out = tnear;
  return tnear <= tfar;
}

/* Ray-sphere, half-b form with a = 1 (unit direction); nearest root in (tmin, tbest). */
static inline int hit_sphere(const ray_t *r, const rtgpu_sph *s, float tmin, float tbest, float *t_out) {
  v3 c = {s->c[0], s->c[1], s->c[2]};
  v3 oc = v3_sub(r->o, c);
  float hb = v3_dot(oc, r->d);
  float cc = v3_dot(oc, oc) - s->r * s->r;
  float disc = hb * hb - cc;
  if (disc < 0.0f) return 0;
  float sq = sqrtf(disc);
  float t = -hb - sq;
  if (!(t > tmin && t < tbest)) {
    t = -hb + sq;
    if (!(t > tmin && t < tbest)) return 0;
  }
  *t_out = t;
  return 1;
}

/* Moller-Trumbore, two-sided. */
static inline int hit_tri(const ray_t *r, const rtgpu_tri *tr, float tmin, float tbest, float *t_out) {
  v3 v0 = v3_from(tr->v0), e1 = v3_from(tr->e1), e2 = v3_from(tr->e2);
  v3 pvec = v3_cross(r->d, e2);
  float det = v3_dot(e1, pvec);
  if (fabsf(det) < 1e-30f) return 0;
  float inv_det = 1.0f / det;
  v3 tvec = v3_sub(r->o, v0);
  float u = v3_dot(tvec, pvec) * inv_det;
  if (u < 0.0f || u > 1.0f) return 0;
  v3 qvec = v3_cross(tvec, e1);
  float v = v3_dot(r->d, qvec) * inv_det;
  if (v < 0.0f || u + v > 1.0f) return 0;
  float t = v3_dot(e2, qvec) * inv_det;
  if (!(t > tmin && t < tbest)) return 0;
  *t_out = t;
  return 1;
}

RTO_API int rto_kat_slab(const float *o, const float *d, const float *lo, const float *hi,
                         float tmin, float tbest, float *tnear) {
  ray_t r;
  r.o = v3_from(o);
  r.d = v3_from(d);
  ray_prepare(&r);
  return slab(&r, lo, hi, tmin, tbest, tnear);
}
RTO_API int rto_kat_sphere(const float *o, const float *d, const float *sph4, float tmin,
                           float tbest, float *t) {
  ray_t r;
  rtgpu_sph s = {{sph4[0], sph4[1], sph4[2]}, sph4[3]};
  r.o = v3_from(o);
  r.d = v3_from(d);
  *t = 0.0f;
  return hit_sphere(&r, &s, tmin, tbest, t);
}
RTO_API int rto_kat_tri(const float *o, const float *d, const float *v0, const float *v1,
                        const float *v2, float tmin, float tbest, float *t) {
  ray_t r;
  rtgpu_tri tr;
  memset(&tr, 0, sizeof(tr));
  for (int k = 0; k < 3; k++) {
    tr.v0[k] = v0[k];
    tr.e1[k] = v1[k] - v0[k];
    tr.e2[k] = v2[k] - v0[k];
  }
  r.o = v3_from(o);
  r.d = v3_from(d);
  *t = 0.0f;
  return hit_tri(&r, &tr, tmin, tbest, t);
}

/* ---- scene access ------------------------------------------------------------- */

typedef struct {
  const rtgpu_node *nodes;
  const rtgpu_node48 *nodes48; /* packed form (rtgpu_format.h); traversed instead of nodes when present */
  const rtgpu_node4 *nodes4;   /* four-wide packed form; traversed in preference when present */
  const uint32_t *slots;       /* slot tree (rtgpu_format.h), 16 words per slot; traversed in preference when present */
  const rtgpu_tri *tris;
  const rtgpu_sph *sphs;
  const uint32_t *sph_meta;
  const rtgpu_material *mats;
  uint64_t n_nodes, n_tris, n_sphs, n_mats;
} scene_t;

#define T_MAX 3.402823466e+38f /* FLT_MAX: initial closest-hit distance */

typedef struct {
  int hit;
  int is_sphere;
  uint32_t slot;  /* index into the leaf-ordered array of its type (slot tree: the primitive's slot) */
  float t;
} hit_t;

typedef struct {
  uint64_t paths, rays, nodes, tri_tests, sphere_tests, scatters;
} counters_t;

static inline void leaf_intersect(const scene_t *sc, const ray_t *r, uint32_t ref, float tmin,
                                  hit_t *best, counters_t *cn) {
  uint32_t first = RTGPU_REF_FIRST(ref), count = RTGPU_REF_COUNT(ref);
  if (ref & RTGPU_REF_SPHERE) {
    for (uint32_t i = first; i < first + count; i++) {
      float t;
      cn->sphere_tests++;
      if (hit_sphere(r, &sc->sphs[i], tmin, best->t, &t)) {
        best->hit = 1; best->is_sphere = 1; best->slot = i; best->t = t;
      }
    }
  } else {
    for (uint32_t i = first; i < first + count; i++) {
      float t;
      cn->tri_tests++;
      if (hit_tri(r, &sc->tris[i], tmin, best->t, &t)) {
        best->hit = 1; best->is_sphere = 0; best->slot = i; best->t = t;
      }
    }
  }
}

/* Closest hit by ordered stack traversal of the flattened BVH2 (row a3):
 * one node fetch tests both children; both hit -> nearer first (tie: child 0),
 * the other is pushed; a popped entry is visited without any re-test. */
static int g_max_stack; /* debug high-water mark (benign race under OpenMP) */
RTO_API int rto_debug_max_stack(int reset) {
  int v = g_max_stack;
  if (reset) g_max_stack = 0;
  return v;
}

/* Closest hit over the four-wide packed nodes (rtgpu_node4, include/rtgpu_format.h):
 * one node fetch tests up to four children; those that pass are visited by ascending
 * key ((bits of tnear) & 0x7FFFFFFC) | k - the nearest now, the others pushed farthest
 * first; a popped entry is visited without any re-test.  The planes are the words as
 * stored (sixteen of the twenty-four carry a byte of a child reference in their low
 * byte and were moved to the conservative side by the builder). */
static hit_t intersect_bvh4(const scene_t *sc, const ray_t *r, float tmin, counters_t *cn) {
  hit_t best = {0, 0, 0, T_MAX};
  uint32_t stack[3 * (RTGPU_MAX_TREE_DEPTH + 2) + 2];
  int sp = 0;
  uint32_t cur = 0; /* root */
  for (;;) {
    if (cur & RTGPU_REF_LEAF) {
      if (cur != RTGPU_REF_EMPTY) leaf_intersect(sc, r, cur, tmin, &best, cn);
    } else {
      const uint32_t *w = sc->nodes4[cur].w;
      float pl[24];
      memcpy(pl, w, sizeof(pl));
      cn->nodes++;
      uint32_t key[4], ref[4];
      int n = 0;
      for (int k = 0; k < 4; k++) {
        float tn;
        const uint32_t *cw = w + 6 * k;
        ref[k] = (cw[0] & 0xFFu) | ((cw[1] & 0xFFu) << 8) | ((cw[2] & 0xFFu) << 16) | ((cw[3] & 0xFFu) << 24);
        int h = slab(r, pl + 6 * k, pl + 6 * k + 3, tmin, best.t, &tn);
        uint32_t tb;
        memcpy(&tb, &tn, 4);
        key[k] = h ? ((tb & 0x7FFFFFFCu) | (uint32_t)k) : 0xFFFFFFFFu;
        n += h;
      }
      if (n > 0) {
        /* ascending keys (distinct among the hits); misses (all ones) sort last */
        uint32_t ord[4] = {key[0], key[1], key[2], key[3]};
        for (int a = 1; a < 4; a++) {
          uint32_t v = ord[a];
          int b = a - 1;
          while (b >= 0 && ord[b] > v) { ord[b + 1] = ord[b]; b--; }
          ord[b + 1] = v;
        }
        for (int a = n - 1; a >= 1; a--) stack[sp++] = ref[ord[a] & 3u];
        if (sp > g_max_stack) g_max_stack = sp;
        cur = ref[ord[0] & 3u];
        continue;
      }
    }
    if (sp == 0) break;
    cur = stack[--sp];
  }
  return best;
}

/* binary16 (zero, normal, +inf: all the slot tree holds) -> float, exactly */
static inline float half_to_float(uint32_t h) {
  uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu, bits;
  if (e == 0) bits = 0u;                       /* zero (the builder writes no subnormal) */
  else if (e == 31) bits = 0x7F800000u;        /* +inf: a missing child */
  else bits = ((e + 112u) << 23) | (m << 13);  /* rebias 15 -> 127 */
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

/* Closest hit over the slot tree (include/rtgpu_format.h): the traversal of intersect_bvh4 - up to four
 * children per node, visited by ascending key ((bits of tnear) & 0x7FFFFFFC) | k - over node slots whose
 * child boxes are binary16 offsets from the node's origin: per axis a = fmaf(origin, 1/d, -o/d) once, then
 * every plane t = fmaf(half, 1/d, a) (on gfx950 one v_fma_mix_f32 each).  The origin words are the floats AS
 * STORED (their low bytes carry the child bytes).  Slot references: bit 31 leaf, bit 30 sphere, bits 28..29
 * count - 1, bits 0..27 the first slot; a leaf's primitives sit in consecutive slots. */
static hit_t intersect_slots(const scene_t *sc, const ray_t *r, float tmin, counters_t *cn) {
  hit_t best = {0, 0, 0, T_MAX};
  uint32_t stack[3 * (RTGPU_MAX_TREE_DEPTH + 2) + 2];
  int sp = 0;
  uint32_t cur = 0; /* root */
  for (;;) {
    if (cur & RTGPU_REF_LEAF) {
      uint32_t first = cur & 0x0FFFFFFFu, count = ((cur >> 28) & 3u) + 1u;
      for (uint32_t i = first; i < first + count; i++) {
        float t;
        if (cur & RTGPU_REF_SPHERE) {
          cn->sphere_tests++;
          if (hit_sphere(r, (const rtgpu_sph *)(sc->slots + 16 * (size_t)i), tmin, best.t, &t)) {
            best.hit = 1; best.is_sphere = 1; best.slot = i; best.t = t;
          }
        } else {
          cn->tri_tests++;
          if (hit_tri(r, (const rtgpu_tri *)(sc->slots + 16 * (size_t)i), tmin, best.t, &t)) {
            best.hit = 1; best.is_sphere = 0; best.slot = i; best.t = t;
          }
        }
      }
    } else {
      const uint32_t *w = sc->slots + 16 * (size_t)cur;
      float org[3];
      memcpy(org, w, sizeof(org));
      cn->nodes++;
      const float ax = fmaf(org[0], r->inv.x, r->noi.x), ay = fmaf(org[1], r->inv.y, r->noi.y),
                  az = fmaf(org[2], r->inv.z, r->noi.z);
      const uint32_t family = w[3] & 0x00FFFFFFu;
      const uint32_t byte[4] = {w[0] & 0xFFu, w[1] & 0xFFu, w[2] & 0xFFu, w[3] >> 24};
      uint32_t key[4], ref[4];
      int n = 0;
      for (int k = 0; k < 4; k++) {
        const uint32_t *cw = w + 4 + 3 * k;
        float t0x = fmaf(half_to_float(cw[0] & 0xFFFFu), r->inv.x, ax), t1x = fmaf(half_to_float(cw[0] >> 16), r->inv.x, ax);
        float t0y = fmaf(half_to_float(cw[1] & 0xFFFFu), r->inv.y, ay), t1y = fmaf(half_to_float(cw[1] >> 16), r->inv.y, ay);
        float t0z = fmaf(half_to_float(cw[2] & 0xFFFFu), r->inv.z, az), t1z = fmaf(half_to_float(cw[2] >> 16), r->inv.z, az);
        float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
        float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), best.t));
        int h = tn <= tf;
        uint32_t tb;
        memcpy(&tb, &tn, 4);
        /* child byte: low nibble = top four bits of the reference, high nibble = place in the family */
        ref[k] = ((byte[k] & 0xFu) << 28) | (family + (byte[k] >> 4));
        key[k] = h ? ((tb & 0x7FFFFFFCu) | (uint32_t)k) : 0xFFFFFFFFu;
        n += h;
      }
      if (n > 0) {
        uint32_t ord[4] = {key[0], key[1], key[2], key[3]};
        for (int a = 1; a < 4; a++) {
          uint32_t v = ord[a];
          int b = a - 1;
          while (b >= 0 && ord[b] > v) { ord[b + 1] = ord[b]; b--; }
          ord[b + 1] = v;
        }
        for (int a = n - 1; a >= 1; a--) stack[sp++] = ref[ord[a] & 3u];
        if (sp > g_max_stack) g_max_stack = sp;
        cur = ref[ord[0] & 3u];
        continue;
      }
    }
    if (sp == 0) break;
    cur = stack[--sp];
  }
  return best;
}

static hit_t intersect_bvh(const scene_t *sc, const ray_t *r, float tmin, counters_t *cn) {
  if (sc->slots) return intersect_slots(sc, r, tmin, cn);
  if (sc->nodes4) return intersect_bvh4(sc, r, tmin, cn);
  hit_t best = {0, 0, 0, T_MAX};
  uint32_t stack[RTGPU_MAX_TREE_DEPTH + 2];
  int sp = 0;
  uint32_t cur = 0; /* root */
  for (;;) {
    if (cur & RTGPU_REF_LEAF) {
      leaf_intersect(sc, r, cur, tmin, &best, cn);
    } else {
      float tn0, tn1;
      int h0, h1;
      uint32_t c0, c1;
      cn->nodes++;
      if (sc->nodes48) {
        /* packed node: the twelve words ARE the planes (eight of them carry a byte of
         * a child reference in their low byte and were moved to the conservative side
         * by the packer); child k's reference = the low bytes of words 8k .. 8k+3 */
        const uint32_t *w = sc->nodes48[cur].w;
        float pl[12];
        memcpy(pl, w, sizeof(pl));
        c0 = (w[0] & 0xFFu) | ((w[1] & 0xFFu) << 8) | ((w[2] & 0xFFu) << 16) | ((w[3] & 0xFFu) << 24);
        c1 = (w[8] & 0xFFu) | ((w[9] & 0xFFu) << 8) | ((w[10] & 0xFFu) << 16) | ((w[11] & 0xFFu) << 24);
        h0 = slab(r, pl + 0, pl + 3, tmin, best.t, &tn0);
        h1 = slab(r, pl + 6, pl + 9, tmin, best.t, &tn1);
      } else {
        const rtgpu_node *n = &sc->nodes[cur];
        /* an empty child (scenes of < 2 primitives) has its box at +inf: its slab
         * interval is [+inf,+inf] or [-inf,-inf] per axis and best.t <= FLT_MAX,
         * so it never passes */
        c0 = n->child0; c1 = n->child1;
        h0 = slab(r, n->lo0, n->hi0, tmin, best.t, &tn0);
        h1 = slab(r, n->lo1, n->hi1, tmin, best.t, &tn1);
      }
      if (h0 && h1) {
        if (tn1 < tn0) { stack[sp++] = c0; cur = c1; }
        else           { stack[sp++] = c1; cur = c0; }
        if (sp > g_max_stack) g_max_stack = sp;
        continue;
      }
      if (h0) { cur = c0; continue; }
      if (h1) { cur = c1; continue; }
    }
    if (sp == 0) break;
    cur = stack[--sp];
  }
  return best;
}

/* Independent check of builder + traversal: test every primitive, no BVH.
 * Same per-primitive tests; visiting order = leaf-array order (spheres, then
 * triangles), so only exact-t ties between primitives can differ from the BVH. */
static hit_t intersect_brute(const scene_t *sc, const ray_t *r, float tmin, counters_t *cn) {
  hit_t best = {0, 0, 0, T_MAX};
  for (uint64_t i = 0; i < sc->n_sphs; i++) {
    float t;
    cn->sphere_tests++;
    if (hit_sphere(r, &sc->sphs[i], tmin, best.t, &t)) {
      best.hit = 1; best.is_sphere = 1; best.slot = (uint32_t)i; best.t = t;
    }
  }
  for (uint64_t i = 0; i < sc->n_tris; i++) {
    float t;
    cn->tri_tests++;
    if (hit_tri(r, &sc->tris[i], tmin, best.t, &t)) {
      best.hit = 1; best.is_sphere = 0; best.slot = (uint32_t)i; best.t = t;
    }
  }
  return best;
}

/* ---- sampling ------------------------------------------------------------------- */

/* Uniform direction on the unit sphere from exactly two draws, no rejection loop (on a 64-wide
 * wavefront a rejection loop runs as long as its unluckiest lane: 3.3 rounds on average, measured a
 * third of the shading pass).  Archimedes' hat box: z uniform in (-1, 1] and the azimuth uniform,
 * both on 24-bit grids.  The azimuth is quadrant q (two bits) plus an angle x in [-pi/4, pi/4);
 * sin x and cos x are the Taylor polynomials through x^9 / x^8 in Horner form with fmaf (truncation
 * below 2.4e-8 on that interval; every operation is a single IEEE f32 operation, so any conforming
 * implementation produces the same bits), turned by q * 90 degrees with sign flips and one exchange.
 * |result| = 1 within 2e-7; scatter() normalises what it builds from it. */
static inline uint32_t u24(rng_t *r) {
  uint32_t h = pcg(r->key + r->ctr * 0x9E3779B9u);
  r->ctr++;
  return h >> 8;
}
static inline float bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t float_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static inline v3 random_unit_vector(rng_t *rng) {
  uint32_t k1 = u24(rng), k2 = u24(rng);
  float t = (float)k1 * 0x1p-23f;                        /* 1 - z, exact, in [0, 2) */
  float z = 1.0f - t;                                    /* exact */
  float rad = sqrtf(t * (2.0f - t));                     /* sqrt((1 - z)(1 + z)): one rounding before the root */
  uint32_t q = k2 >> 22;
  float f = (float)(k2 & 0x3FFFFFu) * 0x1p-22f - 0.5f;   /* exact, in [-1/2, 1/2) */
  float x = f * 1.57079637f;                             /* pi/2 as f32 */
  float x2 = x * x;
  float ps = fmaf(x2, 2.75573192e-6f, -1.98412701e-4f);  /* 1/9!, -1/7! */
  ps = fmaf(x2, ps, 8.33333377e-3f);                     /* 1/5! */
  ps = fmaf(x2, ps, -1.66666672e-1f);                    /* -1/3! */
  float sn = fmaf(x * x2, ps, x);
  float pc = fmaf(x2, 2.48015876e-5f, -1.38888892e-3f);  /* 1/8!, -1/6! */
  pc = fmaf(x2, pc, 4.16666679e-2f);                     /* 1/4! */
  pc = fmaf(x2, pc, -0.5f);
  float cs = fmaf(x2, pc, 1.0f);
  /* q: 0 (c, s)  1 (-s, c)  2 (-c, -s)  3 (s, -c) */
  float ex = (q & 1u) ? sn : cs, ey = (q & 1u) ? cs : sn;
  ex = bits_to_float(float_to_bits(ex) ^ (((q + 1u) & 2u) << 30));
  ey = bits_to_float(float_to_bits(ey) ^ ((q & 2u) << 30));
  v3 r = {rad * ex, rad * ey, z};
  return r;
}

static inline v3 reflect(v3 d, v3 n) {
  float k = 2.0f * v3_dot(d, n);
  return v3_sub(d, v3_scale(n, k));
}

/* One scattering event (row a6; semantics of Shirley's "Ray Tracing in One
 * Weekend", which the reference cites as study material, notes/materials.md:3-5).
 * d = unit direction of the arriving ray, n = unit GEOMETRIC normal at the hit
 * (outward for a sphere).  Returns 0 if the path is absorbed, else 1 with the
 * unit direction of the next segment in *out.  Draws from rng exactly: lambertian
 * and fuzzy metal one random_unit_vector; dielectric one u01 (always). */
static inline int scatter(uint32_t type, float param, v3 d, v3 n, rng_t *rng, v3 *out) {
  float dn = v3_dot(d, n);
  int front = dn < 0.0f;
  if (!front) n = v3_neg(n);
  v3 nd;
  if (type == RTGPU_MAT_LAMBERTIAN) {
    v3 ruv = random_unit_vector(rng);
    v3 sd = v3_add(n, ruv);
    if (fabsf(sd.x) < 1e-8f && fabsf(sd.y) < 1e-8f && fabsf(sd.z) < 1e-8f) sd = n;
    nd = v3_normalize(sd);
  } else if (type == RTGPU_MAT_METAL) {
    /* d and n are unit vectors, so reflect(d, n) is one to rounding (RTIOW normalises here because its ray
     * directions are not unit; the next segment's direction is normalised below in any case) */
    v3 rf = reflect(d, n);
    if (param > 0.0f) {
      v3 ruv = random_unit_vector(rng);
      rf = v3_add(rf, v3_scale(ruv, param));
    }
    if (!(v3_dot(rf, n) > 0.0f)) return 0;
    nd = v3_normalize(rf);
  } else {
    float ri = front ? 1.0f / param : param;
    float cos_t = fminf(-v3_dot(d, n), 1.0f);
    float sin_t = sqrtf(1.0f - cos_t * cos_t);
    int cannot = ri * sin_t > 1.0f;
    float r0 = (1.0f - ri) / (1.0f + ri);
    r0 = r0 * r0;
    float xx = 1.0f - cos_t;
    float x2 = xx * xx;
    float x5 = x2 * x2 * xx;
    float refl = r0 + (1.0f - r0) * x5;
    float u = u01(rng); /* always drawn */
    v3 sd;
    if (cannot || refl > u) {
      sd = reflect(d, n);
    } else {
      v3 perp = v3_scale(v3_add(d, v3_scale(n, cos_t)), ri);
      float par = -sqrtf(fabsf(1.0f - v3_dot(perp, perp)));
      sd = v3_add(perp, v3_scale(n, par));
    }
    nd = v3_normalize(sd);
  }
  *out = nd;
  return 1;
}

/* Known-answer entry point for scatter(): the rng starts at draw `ctr` of `key`;
 * draws_out = number of u01 draws consumed. */
RTO_API void rto_kat_unit_vector(uint32_t key, uint32_t ctr, float *out) {
  rng_t rng = {key, ctr};
  v3 r = random_unit_vector(&rng);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

RTO_API int rto_kat_scatter(uint32_t type, float param, const float *d, const float *n, uint32_t key,
                            uint32_t ctr, float *out_dir, uint32_t *draws_out) {
  rng_t rng = {key, ctr};
  v3 nd = {0.0f, 0.0f, 0.0f};
  int ok = scatter(type, param, v3_from(d), v3_from(n), &rng, &nd);
  out_dir[0] = nd.x; out_dir[1] = nd.y; out_dir[2] = nd.z;
  if (draws_out) *draws_out = rng.ctr - ctr;
  return ok;
}

/* ---- one path (rows a1, a6, a8) ---------------------------------------------------- */

typedef struct {
  const scene_t *sc;
  const rtgpu_camera *cam;
  const rtgpu_settings *st;
  float inv_w, inv_h;
  int brute;
} job_t;

static v3 trace_path(const job_t *J, uint32_t x, uint32_t y, uint32_t sample, uint32_t *primary_id,
                     counters_t *cn) {
  const rtgpu_settings *st = J->st;
  const rtgpu_camera *cam = J->cam;
  rng_t rng = {path_key(y * st->width + x, sample, st->seed), 0};
  v3 L = {0.0f, 0.0f, 0.0f};
  cn->paths++;

  /* camera ray (row a1) */
  float jx = u01(&rng), jy = u01(&rng);
  float sx = ((float)x + jx) * J->inv_w;
  float sy = ((float)y + jy) * J->inv_h;
  ray_t r;
  v3 org = v3_from(cam->origin);
  v3 dir = v3_sub(v3_add(v3_add(v3_from(cam->lower_left), v3_scale(v3_from(cam->horizontal), sx)),
                         v3_scale(v3_from(cam->vertical), sy)),
                  org);
  r.o = org;
  if (cam->lens_radius > 0.0f) {
    float a = 0.0f, b = 0.0f;
    for (int i = 0; i < 8; i++) {
      float ta = u01(&rng) * 2.0f - 1.0f;
      float tb = u01(&rng) * 2.0f - 1.0f;
      if (ta * ta + tb * tb < 1.0f) { a = ta; b = tb; break; }
    }
    float ra = a * cam->lens_radius, rb = b * cam->lens_radius;
    v3 off = v3_add(v3_scale(v3_from(cam->u), ra), v3_scale(v3_from(cam->v), rb));
    r.o = v3_add(org, off);
    dir = v3_sub(dir, off);
  }
  r.d = v3_normalize(dir);

  v3 thr = {1.0f, 1.0f, 1.0f};
  *primary_id = RTGPU_NO_HIT;
  for (uint32_t bounce = 0; bounce < st->max_depth; bounce++) {
    ray_prepare(&r);
    cn->rays++;
    hit_t h = J->brute ? intersect_brute(J->sc, &r, st->tmin, cn)
                       : intersect_bvh(J->sc, &r, st->tmin, cn);
    if (!h.hit) {
      /* sky gradient miss shader */
      float t = 0.5f * (r.d.y + 1.0f);
      float a = 1.0f - t;
      v3 sky = {st->sky_bottom[0] * a + st->sky_top[0] * t, st->sky_bottom[1] * a + st->sky_top[1] * t,
                st->sky_bottom[2] * a + st->sky_top[2] * t};
      L = v3_mul(thr, sky);
      break;
    }
    /* surface point, geometric normal, ids */
    v3 p = v3_add(r.o, v3_scale(r.d, h.t));
    v3 n;
    uint32_t prim, mat;
    if (h.is_sphere && J->sc->slots && !J->brute) { /* sphere slot: {c, r}, then {prim_id, material} */
      const uint32_t *w = J->sc->slots + 16 * (size_t)h.slot;
      const rtgpu_sph *s = (const rtgpu_sph *)w;
      v3 c = {s->c[0], s->c[1], s->c[2]};
      float inv_r = 1.0f / s->r;
      n = v3_scale(v3_sub(p, c), inv_r);
      prim = w[4];
      mat = w[5];
    } else if (h.is_sphere) {
      const rtgpu_sph *s = &J->sc->sphs[h.slot];
      v3 c = {s->c[0], s->c[1], s->c[2]};
      float inv_r = 1.0f / s->r;
      n = v3_scale(v3_sub(p, c), inv_r);
      prim = J->sc->sph_meta[2 * h.slot];
      mat = J->sc->sph_meta[2 * h.slot + 1];
    } else {
      const rtgpu_tri *tr = (J->sc->slots && !J->brute) ? (const rtgpu_tri *)(J->sc->slots + 16 * (size_t)h.slot)
                                                        : &J->sc->tris[h.slot];
      n = v3_normalize(v3_cross(v3_from(tr->e1), v3_from(tr->e2)));
      prim = tr->prim_id;
      mat = tr->material;
    }
    if (bounce == 0) *primary_id = prim;
    /* scatter (row a6) */
    const rtgpu_material *m = &J->sc->mats[mat];
    v3 albedo = v3_from(m->albedo);
    v3 nd;
    cn->scatters++;
    if (m->type == RTGPU_MAT_EMISSIVE) { /* a light ends the path with its radiance, from either face */
      L = v3_mul(thr, v3_scale(albedo, m->param));
      break;
    }
    if (!scatter(m->type, m->param, r.d, n, &rng, &nd)) break; /* absorbed: L stays 0 */
    thr = v3_mul(thr, albedo);
    r.o = p;
    r.d = nd;
  }
  return L;
}

/* ---- frame ------------------------------------------------------------------------- */

/*
 * Renders pixels x in [x0,x1), y in [y0,y1) of the width*height image into
 * accum (float4 per pixel, image layout, row 0 = bottom) and prim_id (u32 per
 * pixel, primary hit of sample `sample_offset`).  Pixels outside the crop, or
 * in tiles another rank owns, are left untouched.  Per pixel the frame sum is
 * the in-order sum of chunk sums, each chunk the in-order sum of chunk_samples
 * consecutive samples (rtgpu_settings.chunk_samples), then
 *   out = (ACCUMULATE ? in : 0) + (sum.r, sum.g, sum.b, (f32)spp).
 */
RTO_API uint32_t *rto_build_slot_tree(const rtgpu_flat_scene *fs, uint64_t *n_slots);
/* What rto_render traverses when the scene has four-wide nodes: their slot tree if the tree has one (1, the
 * default: what the product's trace loop does) or the four-wide nodes themselves (0: RTGPU_SLOT_TREE=0 there). */
static int g_slot_tree = 1;
RTO_API void rto_set_slot_tree(int on) { g_slot_tree = on; }

/* slots: the scene's slot tree from rto_build_slot_tree (traversed), or NULL (the four-wide / packed / plain nodes) */
RTO_API int rto_render_slots(const rtgpu_flat_scene *fs, const rtgpu_camera *cam, const rtgpu_settings *st,
                             float *accum, uint32_t *prim_id, rtgpu_stats *stats, uint32_t x0, uint32_t y0,
                             uint32_t x1, uint32_t y1, int brute_force, int nthreads, const uint32_t *slots) {
  scene_t sc;
  sc.slots = slots;
  sc.nodes = (const rtgpu_node *)fs->nodes;
  sc.nodes48 = (const rtgpu_node48 *)fs->nodes48;
  sc.nodes4 = (const rtgpu_node4 *)fs->nodes4;
  sc.tris = (const rtgpu_tri *)fs->tris;
  sc.sphs = (const rtgpu_sph *)fs->spheres;
  sc.sph_meta = fs->sphere_meta;
  sc.mats = (const rtgpu_material *)fs->materials;
  sc.n_nodes = fs->n_nodes; sc.n_tris = fs->n_tris; sc.n_sphs = fs->n_spheres; sc.n_mats = fs->n_materials;
  if (!st->width || !st->height || !st->tile_size || !st->world_size || st->rank >= st->world_size ||
      !st->chunk_samples)
    return 1;
  if (x1 > st->width) x1 = st->width;
  if (y1 > st->height) y1 = st->height;

  job_t J;
  J.sc = &sc; J.cam = cam; J.st = st; J.brute = brute_force;
  J.inv_w = 1.0f / (float)st->width;
  J.inv_h = 1.0f / (float)st->height;
  uint32_t tiles_x = (st->width + st->tile_size - 1) / st->tile_size;

  counters_t total;
  memset(&total, 0, sizeof(total));
  if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
  {
    counters_t cn;
    memset(&cn, 0, sizeof(cn));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int64_t yy = (int64_t)y0; yy < (int64_t)y1; yy++) {
      uint32_t y = (uint32_t)yy;
      for (uint32_t x = x0; x < x1; x++) {
        uint32_t tile = (y / st->tile_size) * tiles_x + x / st->tile_size;
        if (tile % st->world_size != st->rank) continue;
        /* frame sum = (((C0 + C1) + C2) ...), chunk Ck = its samples summed in order */
        v3 sum = {0.0f, 0.0f, 0.0f};
        uint32_t first_id = RTGPU_NO_HIT;
        for (uint32_t s0 = 0; s0 < st->spp; s0 += st->chunk_samples) {
          v3 chunk = {0.0f, 0.0f, 0.0f};
          uint32_t s1 = s0 + st->chunk_samples < st->spp ? s0 + st->chunk_samples : st->spp;
          for (uint32_t s = s0; s < s1; s++) {
            uint32_t id;
            v3 L = trace_path(&J, x, y, st->sample_offset + s, &id, &cn);
            if (s == 0) first_id = id;
            chunk = v3_add(chunk, L);
          }
          sum = s0 == 0 ? chunk : v3_add(sum, chunk);
        }
        size_t p = (size_t)y * st->width + x;
        if (accum) {
          float *o = accum + 4 * p;
          if (st->flags & RTGPU_FLAG_ACCUMULATE) {
            o[0] = o[0] + sum.x; o[1] = o[1] + sum.y; o[2] = o[2] + sum.z; o[3] = o[3] + (float)st->spp;
          } else {
            o[0] = sum.x; o[1] = sum.y; o[2] = sum.z; o[3] = (float)st->spp;
          }
        }
        if (prim_id) prim_id[p] = first_id;
      }
    }
#ifdef _OPENMP
#pragma omp critical
#endif
    {
      total.paths += cn.paths; total.rays += cn.rays; total.nodes += cn.nodes;
      total.tri_tests += cn.tri_tests; total.sphere_tests += cn.sphere_tests;
      total.scatters += cn.scatters;
    }
  }
  if (stats) {
    stats->paths = total.paths; stats->rays = total.rays; stats->nodes = total.nodes;
    stats->tri_tests = total.tri_tests; stats->sphere_tests = total.sphere_tests;
    stats->scatters = total.scatters;
  }
  return 0;
}

RTO_API int rto_render(const rtgpu_flat_scene *fs, const rtgpu_camera *cam, const rtgpu_settings *st,
                       float *accum, uint32_t *prim_id, rtgpu_stats *stats, uint32_t x0, uint32_t y0,
                       uint32_t x1, uint32_t y1, int brute_force, int nthreads) {
  uint32_t *slots = NULL;
  if (g_slot_tree && fs->nodes4 && !brute_force) {
    uint64_t n_slots = 0;
    slots = rto_build_slot_tree(fs, &n_slots);
    if (!slots && n_slots) return 2; /* out of memory */
  }
  int rc = rto_render_slots(fs, cam, st, accum, prim_id, stats, x0, y0, x1, y1, brute_force, nthreads, slots);
  free(slots);
  return rc;
}

/* Resolve (row a8): mean, gamma 2, clamp, 8 bits per channel, 0xRRGGBBAA
 * (run_tree/data/shaders/shaders.hlsl:18-25 unpacks exactly this). */
RTO_API void rto_resolve_rgba8(const float *accum, uint32_t *rgba8, uint64_t n_pixels) {
  for (uint64_t i = 0; i < n_pixels; i++) {
    const float *a = accum + 4 * i;
    uint32_t px = 0xFFu;
    if (a[3] > 0.0f) {
      uint32_t ch[3];
      for (int k = 0; k < 3; k++) {
        float c = sqrtf(fmaxf(a[k] / a[3], 0.0f));
        c = fminf(c, 0.999f);
        ch[k] = (uint32_t)(256.0f * c);
      }
      px = (ch[0] << 24) | (ch[1] << 16) | (ch[2] << 8) | 0xFFu;
    }
    rgba8[i] = px;
  }
}

/* Structural validation of a flattened BVH (checks the builder, independent of
 * any ray): every reference in range, every primitive inside its leaf box,
 * every node's child boxes inside the box its parent stores for it, every
 * primitive referenced exactly once.  Returns 0 if valid, else a code. */
static int box_inside(const float *lo, const float *hi, const float *plo, const float *phi) {
  for (int k = 0; k < 3; k++)
    if (lo[k] < plo[k] || hi[k] > phi[k]) return 0;
  return 1;
}

/* a child box of a packed node may stick out of the box its parent stores for that
 * node by the packing move (each plane is moved on its own, < 512 floats) */
static int64_t ord_f32(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (u & 0x80000000u) ? -(int64_t)(u & 0x7FFFFFFFu) - 1 : (int64_t)u;
}
static int box_inside_packed(const float *lo, const float *hi, const float *plo, const float *phi) {
  for (int k = 0; k < 3; k++) {
    if (isinf(plo[k]) && isinf(phi[k])) continue; /* the root's "parent" */
    if (ord_f32(lo[k]) < ord_f32(plo[k]) - 512 || ord_f32(hi[k]) > ord_f32(phi[k]) + 512) return 0;
  }
  return 1;
}

RTO_API int rto_validate_bvh(const rtgpu_flat_scene *fs, uint32_t *max_depth_out) {
  const rtgpu_node *nodes = (const rtgpu_node *)fs->nodes;
  const rtgpu_tri *tris = (const rtgpu_tri *)fs->tris;
  const rtgpu_sph *sphs = (const rtgpu_sph *)fs->spheres;
  uint8_t *seen_t = (uint8_t *)calloc(fs->n_tris + 1, 1);
  uint8_t *seen_s = (uint8_t *)calloc(fs->n_spheres + 1, 1);
  typedef struct { uint32_t ref; uint32_t depth; float lo[3], hi[3]; } item_t;
  size_t cap = 2 * (size_t)fs->n_nodes + 16, sp = 0;
  item_t *stack = (item_t *)malloc(cap * sizeof(item_t));
  int rc = 0;
  uint32_t maxd = 0;
  item_t root = {0, 0, {-INFINITY, -INFINITY, -INFINITY}, {INFINITY, INFINITY, INFINITY}};
  stack[sp++] = root;
  while (sp && !rc) {
    item_t it = stack[--sp];
    if (it.ref == RTGPU_REF_EMPTY) continue;
    if (it.ref & RTGPU_REF_LEAF) {
      uint32_t first = RTGPU_REF_FIRST(it.ref), count = RTGPU_REF_COUNT(it.ref);
      if (it.depth > maxd) maxd = it.depth;
      for (uint32_t i = first; i < first + count && !rc; i++) {
        float lo[3], hi[3];
        if (it.ref & RTGPU_REF_SPHERE) {
          if (i >= fs->n_spheres) { rc = 2; break; }
          if (seen_s[i]++) { rc = 3; break; }
          for (int k = 0; k < 3; k++) { lo[k] = sphs[i].c[k] - sphs[i].r; hi[k] = sphs[i].c[k] + sphs[i].r; }
        } else {
          if (i >= fs->n_tris) { rc = 2; break; }
          if (seen_t[i]++) { rc = 3; break; }
          for (int k = 0; k < 3; k++) {
            float a = tris[i].v0[k], b = a + tris[i].e1[k], c = a + tris[i].e2[k];
            /* v0+e rounds; allow the builder's padding to absorb 1 ulp */
            lo[k] = fminf(a, fminf(b, c)); hi[k] = fmaxf(a, fmaxf(b, c));
            float e = fmaxf(fmaxf(fabsf(lo[k]), fabsf(hi[k])), 1.0f) * 0x1p-22f;
            lo[k] += e; hi[k] -= e;
            if (lo[k] > hi[k]) lo[k] = hi[k] = 0.5f * (lo[k] + hi[k]);
          }
        }
        if (!box_inside(lo, hi, it.lo, it.hi)) rc = 4;
      }
      continue;
    }
    if (it.ref >= fs->n_nodes) { rc = 5; break; }
    if (it.depth > RTGPU_MAX_TREE_DEPTH) { rc = 6; break; }
    const rtgpu_node *n = &nodes[it.ref];
    if (n->child0 != RTGPU_REF_EMPTY && !box_inside(n->lo0, n->hi0, it.lo, it.hi)) { rc = 7; break; }
    if (n->child1 != RTGPU_REF_EMPTY && !box_inside(n->lo1, n->hi1, it.lo, it.hi)) { rc = 7; break; }
    if (sp + 2 > cap) { rc = 8; break; }
    item_t a = {n->child0, it.depth + 1, {n->lo0[0], n->lo0[1], n->lo0[2]}, {n->hi0[0], n->hi0[1], n->hi0[2]}};
    item_t b = {n->child1, it.depth + 1, {n->lo1[0], n->lo1[1], n->lo1[2]}, {n->hi1[0], n->hi1[1], n->hi1[2]}};
    stack[sp++] = a;
    stack[sp++] = b;
  }
  if (!rc) {
    for (uint64_t i = 0; i < fs->n_tris; i++) if (seen_t[i] != 1) { rc = 9; break; }
    for (uint64_t i = 0; i < fs->n_spheres; i++) if (seen_s[i] != 1) { rc = 9; break; }
  }
  /* packed copy: present exactly from the threshold up; same references; every plane
   * on the conservative side of the exact one and within 255 floats of it (511 across zero) */
  if (!rc && ((fs->nodes48 != NULL) != (fs->n_nodes >= RTGPU_NODE48_MIN_NODES))) rc = 10;
  if (!rc && fs->nodes48) {
    const rtgpu_node48 *pk = (const rtgpu_node48 *)fs->nodes48;
    for (uint64_t i = 0; i < fs->n_nodes && !rc; i++) {
      const uint32_t *w = pk[i].w;
      uint32_t x[12];
      memcpy(x, nodes[i].lo0, 12);  memcpy(x + 3, nodes[i].hi0, 12);
      memcpy(x + 6, nodes[i].lo1, 12);  memcpy(x + 9, nodes[i].hi1, 12);
      uint32_t c0 = (w[0] & 0xFFu) | ((w[1] & 0xFFu) << 8) | ((w[2] & 0xFFu) << 16) | ((w[3] & 0xFFu) << 24);
      uint32_t c1 = (w[8] & 0xFFu) | ((w[9] & 0xFFu) << 8) | ((w[10] & 0xFFu) << 16) | ((w[11] & 0xFFu) << 24);
      if (c0 != nodes[i].child0 || c1 != nodes[i].child1) { rc = 11; break; }
      for (int k = 0; k < 12; k++) {
        int is_lo = (k % 6) < 3; /* words 0-2 and 6-8 are lo planes */
        int64_t a = (w[k] & 0x80000000u) ? -(int64_t)(w[k] & 0x7FFFFFFFu) - 1 : (int64_t)w[k];  /* ordered: -0 just below +0 */
        int64_t b = (x[k] & 0x80000000u) ? -(int64_t)(x[k] & 0x7FFFFFFFu) - 1 : (int64_t)x[k];
        int moved = k < 4 || k >= 8;
        int64_t lim = ((w[k] ^ x[k]) & 0x80000000u) ? 511 : 255; /* the low byte flips meaning across zero */
        if (!moved ? a != b : (is_lo ? (a > b || b - a > lim) : (a < b || a - b > lim))) { rc = 12; break; }
      }
    }
  }
  /* four-wide nodes: present from the same threshold; every node reached once from the
   * root, every primitive referenced once, every child box inside the box its parent
   * stores for that node, every primitive inside its leaf's box, stack bound honoured */
  if (!rc && fs->nodes4 && fs->n_nodes < RTGPU_NODE48_MIN_NODES) rc = 13;
  if (!rc && fs->nodes4) {
    const rtgpu_node4 *n4 = (const rtgpu_node4 *)fs->nodes4;
    memset(seen_t, 0, fs->n_tris + 1);
    memset(seen_s, 0, fs->n_spheres + 1);
    typedef struct { uint32_t ref; uint32_t on_stack; float lo[3], hi[3]; } item4_t;
    size_t cap4 = 4 * (size_t)fs->n_nodes4 + 16, sp4 = 0;
    item4_t *st4 = (item4_t *)malloc(cap4 * sizeof(item4_t));
    uint64_t visited = 0;
    item4_t root4 = {0, 0, {-INFINITY, -INFINITY, -INFINITY}, {INFINITY, INFINITY, INFINITY}};
    st4[sp4++] = root4;
    while (sp4 && !rc) {
      item4_t it = st4[--sp4];
      if (it.ref == RTGPU_REF_EMPTY) continue;
      if (it.ref & RTGPU_REF_LEAF) {
        uint32_t first = RTGPU_REF_FIRST(it.ref), count = RTGPU_REF_COUNT(it.ref);
        for (uint32_t i = first; i < first + count && !rc; i++) {
          float lo[3], hi[3];
          if (it.ref & RTGPU_REF_SPHERE) {
            if (i >= fs->n_spheres) { rc = 14; break; }
            if (seen_s[i]++) { rc = 15; break; }
            for (int k = 0; k < 3; k++) { lo[k] = sphs[i].c[k] - sphs[i].r; hi[k] = sphs[i].c[k] + sphs[i].r; }
          } else {
            if (i >= fs->n_tris) { rc = 14; break; }
            if (seen_t[i]++) { rc = 15; break; }
            for (int k = 0; k < 3; k++) {
              float a = tris[i].v0[k], b = a + tris[i].e1[k], c = a + tris[i].e2[k];
              lo[k] = fminf(a, fminf(b, c)); hi[k] = fmaxf(a, fmaxf(b, c));
              float e = fmaxf(fmaxf(fabsf(lo[k]), fabsf(hi[k])), 1.0f) * 0x1p-22f;
              lo[k] += e; hi[k] -= e;
              if (lo[k] > hi[k]) lo[k] = hi[k] = 0.5f * (lo[k] + hi[k]);
            }
          }
          if (!box_inside(lo, hi, it.lo, it.hi)) rc = 16;
        }
        continue;
      }
      if (it.ref >= fs->n_nodes4) { rc = 14; break; }
      visited++;
      const uint32_t *w = n4[it.ref].w;
      int nch = 0;
      for (int k = 0; k < 4; k++) {
        const uint32_t *cw = w + 6 * k;
        uint32_t ref = (cw[0] & 0xFFu) | ((cw[1] & 0xFFu) << 8) | ((cw[2] & 0xFFu) << 16) | ((cw[3] & 0xFFu) << 24);
        if (ref == RTGPU_REF_EMPTY) continue;
        nch++;
        item4_t c;
        c.ref = ref;
        memcpy(c.lo, cw, 12);
        memcpy(c.hi, cw + 3, 12);
        if (!box_inside_packed(c.lo, c.hi, it.lo, it.hi)) { rc = 17; break; }
        c.on_stack = 0;
        if (sp4 + 1 > cap4) { rc = 18; break; }
        st4[sp4++] = c;
      }
      if (!rc && nch < 2) rc = 19;
      /* siblings that can lie below this node's descendants on the traversal stack */
      for (size_t q = sp4 - (size_t)nch; !rc && q < sp4; q++) {
        st4[q].on_stack = it.on_stack + (uint32_t)nch - 1u;
        if (st4[q].on_stack > fs->max_stack4) rc = 20;
      }
    }
    if (!rc && visited != fs->n_nodes4) rc = 21;
    if (!rc) {
      for (uint64_t i = 0; i < fs->n_tris; i++) if (seen_t[i] != 1) { rc = 22; break; }
      for (uint64_t i = 0; i < fs->n_spheres; i++) if (seen_s[i] != 1) { rc = 22; break; }
    }
    free(st4);
  }
  if (max_depth_out) *max_depth_out = maxd;
  free(seen_t); free(seen_s); free(stack);
  return rc;
}

RTO_API int rto_has_openmp(void) {
#ifdef _OPENMP
  return 1;
#else
  return 0;
#endif
}

/* ==== Packed 48-byte nodes (include/rtgpu_format.h) =================================
 *
 * CPU restatement of rt_amd/csrc/pack48.h (no reference counterpart: the reference
 * has no BVH).  A child reference's four bytes ride in the low bytes
 * of four planes; each such plane moves to the nearest float on the conservative
 * side (lo planes down, hi planes up) whose low byte is the wanted one.  Written
 * with float comparisons and nextafter-free bit stepping on magnitudes, i.e. NOT
 * the ordered-integer formulation of the product, so that the two check each other. */
static int pack_plane(uint32_t bits, uint32_t byte, int down, uint32_t *out) {
  if ((bits & 0x7F800000u) == 0x7F800000u) return 0;
  float x;
  memcpy(&x, &bits, 4);
  /* candidates: the same 256-float block with the byte swapped in, and the two
   * neighbouring blocks in magnitude (same sign); across zero: the other sign's
   * block of smallest magnitude.  Pick the conservative one nearest to x. */
  uint32_t sign = bits & 0x80000000u, mag = bits & 0x7FFFFFFFu;
  uint32_t cand[4];
  int nc = 0;
  cand[nc++] = sign | (mag & 0x7FFFFF00u) | byte;
  if ((mag >> 8) < 0x7FFFFFu) cand[nc++] = sign | (((mag >> 8) + 1u) << 8) | byte;
  if ((mag >> 8) > 0u) cand[nc++] = sign | (((mag >> 8) - 1u) << 8) | byte;
  else cand[nc++] = (sign ^ 0x80000000u) | byte; /* block 0 of the other sign */
  int have = 0;
  float best = 0.0f;
  uint32_t best_bits = 0;
  for (int i = 0; i < nc; i++) {
    if ((cand[i] & 0x7F800000u) == 0x7F800000u) continue;
    float c;
    memcpy(&c, &cand[i], 4);
    /* total order on bit patterns where -0 < +0 (the product's ordered integers):
     * compare as floats first, break the +-0 tie by sign */
    int le = c < x || (c == x && ((cand[i] >> 31) >= (bits >> 31)));
    int ge = c > x || (c == x && ((cand[i] >> 31) <= (bits >> 31)));
    if (down ? !le : !ge) continue;
    int closer = !have || (down ? (c > best || (c == best && (cand[i] >> 31) < (best_bits >> 31)))
                                : (c < best || (c == best && (cand[i] >> 31) > (best_bits >> 31))));
    if (closer) { have = 1; best = c; best_bits = cand[i]; }
  }
  if (!have) return 0;
  *out = best_bits;
  return 1;
}

RTO_API int rto_pack_node48(const rtgpu_node *n, rtgpu_node48 *out) {
  uint32_t w[12];
  memcpy(w, n->lo0, 12);  memcpy(w + 3, n->hi0, 12);
  memcpy(w + 6, n->lo1, 12);  memcpy(w + 9, n->hi1, 12);
  for (int i = 0; i < 12; i++)
    if ((w[i] & 0x7F800000u) == 0x7F800000u) return 0;
  for (int k = 0; k < 4; k++) {
    if (!pack_plane(w[k], (n->child0 >> (8 * k)) & 0xFFu, k < 3, &w[k])) return 0;
    if (!pack_plane(w[8 + k], (n->child1 >> (8 * k)) & 0xFFu, k < 1, &w[8 + k])) return 0;
  }
  memcpy(out->w, w, sizeof(w));
  return 1;
}

/* ==== Slot tree (include/rtgpu_format.h) ============================================
 *
 * CPU restatement of rt_amd/csrc/slot_tree.hip (no reference counterpart: the reference has no BVH,
 * SURVEY.md §0): the four-wide nodes and the primitives they name as ONE array of 64-byte slots.
 *  - family of node i (its children, in child order: an inner child 1 slot, a leaf of n primitives n
 *    slots) starts at slot 1 + (slots of the families of nodes 0 .. i-1); the root is slot 0;
 *  - node slot: words 0..2 the origin (per axis the smallest lo plane of the children that exist,
 *    moved DOWN to the nearest float whose low byte is the child byte of child 0 / 1 / 2), word 3 =
 *    family | child byte of child 3 << 24, words 4 + 3k + a = half(lo - origin) | half(hi - origin) << 16
 *    for child k, axis a: the differences in double, rounded DOWN / UP to zero or a normal binary16
 *    (0x7C00 twice for a missing child);
 *  - child byte = place in the family << 4 | leaf << 3 | sphere << 2 | count - 1;
 *  - triangle slot: the 48 bytes of rtgpu_tri, sphere slot: rtgpu_sph then {prim_id, material}; zero after.
 * A tree with a leaf of more than four primitives, a child box more than 65504 from the origin or 2^24
 * slots or more has no slot form (the four-wide nodes are traversed).  The half rounding here works on
 * frexp / ldexp / floor / ceil, the product's on the bits of the double, so that the two check each other. */
static int half_round(double d, int up, uint32_t *out) {
  if (!(d >= 0.0)) return 0;
  if (d < 0x1p-14) { *out = (up && d > 0.0) ? 0x0400u : 0u; return 1; }
  if (d >= 65536.0) { if (up) return 0; *out = 0x7BFFu; return 1; }
  int ex;
  (void)frexp(d, &ex); /* d = m * 2^ex, m in [0.5, 1) */
  int e = ex - 1;
  double q = ldexp(d, 10 - e); /* in [1024, 2048), exact */
  double qi = up ? ceil(q) : floor(q);
  if (qi >= 2048.0) { e++; qi = 1024.0; }
  if (e > 15) { if (up) return 0; *out = 0x7BFFu; return 1; }
  *out = ((uint32_t)(e + 15) << 10) | ((uint32_t)qi - 1024u);
  return 1;
}

typedef struct { uint32_t ref, size, nibble; } slot_child_t;
static int slot_child_of(const rtgpu_node4 *n, int k, slot_child_t *c) {
  const uint32_t *cw = n->w + 6 * k;
  c->ref = (cw[0] & 0xFFu) | ((cw[1] & 0xFFu) << 8) | ((cw[2] & 0xFFu) << 16) | ((cw[3] & 0xFFu) << 24);
  c->size = 0; c->nibble = 0;
  if (c->ref == RTGPU_REF_EMPTY) return 1;
  if (!(c->ref & RTGPU_REF_LEAF)) { c->size = 1; return 1; }
  uint32_t count = RTGPU_REF_COUNT(c->ref);
  if (count > 4u) return 0;
  c->size = count;
  c->nibble = 8u | ((c->ref & RTGPU_REF_SPHERE) ? 4u : 0u) | (count - 1u);
  return 1;
}

/* The slot tree of a flattened scene that has four-wide nodes: malloc'ed, 16 words per slot, *n_slots of
 * them; NULL with *n_slots = 0 if the tree has no slot form (or no four-wide nodes), NULL with *n_slots = 1
 * if memory ran out. */
RTO_API uint32_t *rto_build_slot_tree(const rtgpu_flat_scene *fs, uint64_t *n_slots) {
  *n_slots = 0;
  const rtgpu_node4 *n4 = (const rtgpu_node4 *)fs->nodes4;
  const uint64_t n = fs->n_nodes4;
  if (!n4 || n == 0 || n >= (1ull << 24)) return NULL;
  uint32_t *family = malloc((size_t)n * sizeof(uint32_t)), *slot_of = malloc((size_t)n * sizeof(uint32_t));
  if (!family || !slot_of) { free(family); free(slot_of); *n_slots = 1; return NULL; }
  uint64_t total = 1;
  int ok = 1;
  for (uint64_t i = 0; i < n && ok; i++) {
    family[i] = (uint32_t)total;
    for (int k = 0; k < 4; k++) {
      slot_child_t c;
      if (!slot_child_of(&n4[i], k, &c)) ok = 0;
      total += c.size;
    }
    if (total >= (1ull << 24)) ok = 0;
  }
  uint32_t *slots = ok ? calloc((size_t)total * 16, sizeof(uint32_t)) : NULL;
  if (ok && !slots) { free(family); free(slot_of); *n_slots = 1; return NULL; }
  slot_of[0] = 0;
  for (uint64_t i = 0; i < n && ok; i++) {
    slot_child_t c[4];
    uint32_t off[4], at = 0;
    for (int k = 0; k < 4; k++) {
      (void)slot_child_of(&n4[i], k, &c[k]);
      off[k] = at;
      at += c[k].size;
      if (c[k].size && !(c[k].ref & RTGPU_REF_LEAF)) {
        if (c[k].ref <= i || c[k].ref >= n) { ok = 0; break; }
        slot_of[c[k].ref] = family[i] + off[k];
      }
    }
    if (!ok) break;
    uint32_t *w = slots + 16 * (size_t)slot_of[i];
    float org[3];
    for (int a = 0; a < 3 && ok; a++) {
      int any = 0;
      float lo = 0.0f;
      for (int k = 0; k < 4; k++) {
        if (c[k].ref == RTGPU_REF_EMPTY) continue;
        float v;
        memcpy(&v, &n4[i].w[6 * k + a], 4);
        if (!any || v < lo) lo = v;
        any = 1;
      }
      uint32_t bits;
      memcpy(&bits, &lo, 4);
      if (!any || !pack_plane(bits, (off[a] << 4) | c[a].nibble, 1, &w[a])) { ok = 0; break; }
      memcpy(&org[a], &w[a], 4);
    }
    if (!ok) break;
    w[3] = family[i] | (((off[3] << 4) | c[3].nibble) << 24);
    for (int k = 0; k < 4 && ok; k++)
      for (int a = 0; a < 3; a++) {
        uint32_t word = 0x7C007C00u;
        if (c[k].ref != RTGPU_REF_EMPTY) {
          float lo, hi;
          uint32_t hl, hh;
          memcpy(&lo, &n4[i].w[6 * k + a], 4);
          memcpy(&hi, &n4[i].w[6 * k + 3 + a], 4);
          if (!half_round((double)lo - (double)org[a], 0, &hl) || !half_round((double)hi - (double)org[a], 1, &hh)) { ok = 0; break; }
          word = hl | (hh << 16);
        }
        w[4 + 3 * k + a] = word;
      }
    for (int k = 0; k < 4 && ok; k++) {
      if (!c[k].size || !(c[k].ref & RTGPU_REF_LEAF)) continue;
      const uint32_t first = RTGPU_REF_FIRST(c[k].ref);
      for (uint32_t j = 0; j < c[k].size; j++) {
        uint32_t *p = slots + 16 * (size_t)(family[i] + off[k] + j);
        if (c[k].ref & RTGPU_REF_SPHERE) {
          memcpy(p, (const rtgpu_sph *)fs->spheres + (first + j), sizeof(rtgpu_sph));
          p[4] = fs->sphere_meta[2 * (size_t)(first + j)];
          p[5] = fs->sphere_meta[2 * (size_t)(first + j) + 1];
        } else {
          memcpy(p, (const rtgpu_tri *)fs->tris + (first + j), sizeof(rtgpu_tri));
        }
      }
    }
  }
  free(family); free(slot_of);
  if (!ok) { free(slots); return NULL; }
  *n_slots = total;
  return slots;
}
RTO_API void rto_free_words(uint32_t *p) { free(p); }

/* nodes48 of a flattened scene, as the flattener adds it: NULL below the threshold */
static rtgpu_node48 *pack_all(const rtgpu_node *nodes, uint64_t n, int *fail) {
  *fail = 0;
  if (n < RTGPU_NODE48_MIN_NODES) return NULL;
  rtgpu_node48 *out = calloc((size_t)n, sizeof(rtgpu_node48));
  if (!out) { *fail = 1; return NULL; }
  for (uint64_t i = 0; i < n; i++)
    if (!rto_pack_node48(&nodes[i], &out[i])) { free(out); *fail = 1; return NULL; }
  return out;
}

/* ==== LBVH builder restatement ================================
 *
 * CPU restatement of the GPU builder in rt_amd/csrc/lbvh_build.hip (no reference
 * counterpart: the reference has no BVH).  Published algorithm:
 * T. Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and
 * k-d Trees", HPG 2012 - Morton codes of the primitive centroids, one radix
 * sort, the hierarchy from the longest common prefixes of neighbouring keys.
 * Everything below is integer arithmetic or exact float min/max except the
 * quantisation (c - cmin) * (1024 / ext), which both sides compute as separately
 * rounded IEEE f32 operations.  The output is a flattened scene in the format of
 * include/rtgpu_format.h, byte for byte what the GPU builder produces:
 *
 *  - prims: spheres first (index = sphere index), then triangles; key =
 *    morton30(centroid) << 32 | index, sorted ascending (keys are unique);
 *  - one primitive per leaf, leaf arrays in sorted order per type;
 *  - internal node i of Karras' numbering (root = 0) covers a key range and splits
 *    it where the common prefix changes; its record holds both children's boxes
 *    (primitive boxes padded as the SAH builder pads them);
 *  - renumbering: nodes whose subtree holds >= T = max(1, ceil(2P/4096))
 *    primitives ("the top of the tree", fewer than 4096 of them) come first, in
 *    Karras order, the others follow in Karras order (root stays 0).
 */
static inline uint32_t expand10(uint32_t v) { /* 10 bits -> every third bit */
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

static inline int clz64(uint64_t x) { return x ? __builtin_clzll(x) : 64; }

static int cmp_u64(const void *a, const void *b) {
  uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x < y ? -1 : x > y;
}

typedef struct { float lo[3], hi[3]; } box_t;

static void prim_box(const rtgpu_scene_desc *sc, uint32_t i, box_t *b, float *cent) {
  if (i < sc->n_spheres) {
    const rtgpu_sphere *s = &sc->spheres[i];
    for (int k = 0; k < 3; k++) {
      b->lo[k] = s->center[k] - s->radius;
      b->hi[k] = s->center[k] + s->radius;
      cent[k] = s->center[k];
    }
  } else {
    const rtgpu_triangle *t = &sc->triangles[i - sc->n_spheres];
    for (int k = 0; k < 3; k++) {
      b->lo[k] = fminf(t->v0[k], fminf(t->v1[k], t->v2[k]));
      b->hi[k] = fmaxf(t->v0[k], fmaxf(t->v1[k], t->v2[k]));
      cent[k] = (b->lo[k] + b->hi[k]) * 0.5f;
    }
  }
}

static void pad_box(box_t *b) { /* as rt_amd/csrc/bvh_build.cpp pad_box */
  for (int k = 0; k < 3; k++) {
    float m = fmaxf(fmaxf(fabsf(b->lo[k]), fabsf(b->hi[k])), 1.0f);
    float e = m * (1.0f / 262144.0f);
    b->lo[k] -= e;
    b->hi[k] += e;
  }
}

static inline int lbvh_delta(const uint64_t *keys, int64_t n, int64_t a, int64_t b) {
  if (b < 0 || b >= n) return -1;
  return clz64(keys[a] ^ keys[b]);
}

RTO_API void rto_flat_free(rtgpu_flat_scene *fs) {
  if (!fs) return;
  free((void *)fs->nodes); free((void *)fs->tris); free((void *)fs->spheres);
  free((void *)fs->sphere_meta); free((void *)fs->materials); free((void *)fs->nodes48);
  free((void *)fs->nodes4); free((void *)fs->nodes4_src);
  memset(fs, 0, sizeof(*fs));
}

RTO_API int rto_build_lbvh(const rtgpu_scene_desc *sc, rtgpu_flat_scene *out) {
  memset(out, 0, sizeof(*out));
  const int64_t ns = (int64_t)sc->n_spheres, nt = (int64_t)sc->n_triangles, P = ns + nt;
  const int64_t n_inner = P > 1 ? P - 1 : 1;
  rtgpu_node *nodes = calloc((size_t)n_inner, sizeof(rtgpu_node));
  rtgpu_tri *tris = calloc((size_t)nt + 8, sizeof(rtgpu_tri));
  rtgpu_sph *sphs = calloc((size_t)ns + 8, sizeof(rtgpu_sph));
  uint32_t *meta = calloc((size_t)(2 * ns + 16), sizeof(uint32_t));
  rtgpu_material *mats = calloc(sc->n_materials ? sc->n_materials : 1, sizeof(rtgpu_material));
  uint64_t *keys = malloc(sizeof(uint64_t) * (size_t)(P ? P : 1));
  box_t *pbox = malloc(sizeof(box_t) * (size_t)(P ? P : 1));
  if (!nodes || !tris || !sphs || !meta || !mats || !keys || !pbox) return 2;

  /* materials: the flattener's dielectric constants (bvh_build.cpp) */
  for (uint64_t k = 0; k < sc->n_materials; k++) {
    mats[k] = sc->materials[k];
    mats[k]._pad[0] = mats[k]._pad[1] = mats[k]._pad[2] = 0;
    if (mats[k].type == RTGPU_MAT_DIELECTRIC) {
      float ior = mats[k].param, inv = 1.0f / ior;
      float qf = (1.0f - inv) / (1.0f + inv), qb = (1.0f - ior) / (1.0f + ior);
      float v[3] = {inv, qf * qf, qb * qb};
      memcpy(mats[k]._pad, v, sizeof(v));
    }
  }

  /* boxes, centroid bounds, keys */
  float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
  float *cent = malloc(sizeof(float) * 3 * (size_t)(P ? P : 1));
  for (int64_t i = 0; i < P; i++) {
    prim_box(sc, (uint32_t)i, &pbox[i], &cent[3 * i]);
    pad_box(&pbox[i]);
    for (int k = 0; k < 3; k++) {
      cmin[k] = fminf(cmin[k], cent[3 * i + k]);
      cmax[k] = fmaxf(cmax[k], cent[3 * i + k]);
    }
  }
  float scale[3];
  for (int k = 0; k < 3; k++) {
    float ext = cmax[k] - cmin[k];
    scale[k] = ext > 0.0f ? 1024.0f / ext : 0.0f;
  }
  for (int64_t i = 0; i < P; i++) {
    uint32_t q[3];
    for (int k = 0; k < 3; k++) {
      float f = (cent[3 * i + k] - cmin[k]) * scale[k];
      int v = (int)f;
      q[k] = (uint32_t)(v < 0 ? 0 : v > 1023 ? 1023 : v);
    }
    uint32_t code = (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
    keys[i] = ((uint64_t)code << 32) | (uint64_t)i;
  }
  free(cent);
  qsort(keys, (size_t)P, sizeof(uint64_t), cmp_u64);

  /* leaf arrays in sorted order, and each sorted position's leaf reference */
  uint32_t *leaf_ref = malloc(sizeof(uint32_t) * (size_t)(P ? P : 1));
  uint32_t n_s = 0, n_t = 0;
  for (int64_t k = 0; k < P; k++) {
    uint32_t p = (uint32_t)keys[k];
    if (p < (uint32_t)ns) {
      const rtgpu_sphere *s = &sc->spheres[p];
      sphs[n_s].c[0] = s->center[0]; sphs[n_s].c[1] = s->center[1]; sphs[n_s].c[2] = s->center[2];
      sphs[n_s].r = s->radius;
      meta[2 * n_s] = p; meta[2 * n_s + 1] = s->material;
      leaf_ref[k] = RTGPU_REF_LEAF | RTGPU_REF_SPHERE | n_s;
      n_s++;
    } else {
      const rtgpu_triangle *t = &sc->triangles[p - ns];
      for (int c = 0; c < 3; c++) {
        tris[n_t].v0[c] = t->v0[c];
        tris[n_t].e1[c] = t->v1[c] - t->v0[c];
        tris[n_t].e2[c] = t->v2[c] - t->v0[c];
      }
      tris[n_t].prim_id = p; tris[n_t].material = t->material; tris[n_t]._pad = 0;
      leaf_ref[k] = RTGPU_REF_LEAF | n_t;
      n_t++;
    }
  }

  uint32_t max_depth = 0;
  int64_t n_emitted = n_inner;  /* nodes merged into leaves are not emitted */
  if (P <= 1) {
    for (int k = 0; k < 3; k++) {
      nodes[0].lo0[k] = nodes[0].lo1[k] = INFINITY;
      nodes[0].hi0[k] = nodes[0].hi1[k] = INFINITY;
    }
    nodes[0].child0 = nodes[0].child1 = RTGPU_REF_EMPTY;
    if (P == 1) {
      for (int k = 0; k < 3; k++) { nodes[0].lo0[k] = pbox[0].lo[k]; nodes[0].hi0[k] = pbox[0].hi[k]; }
      nodes[0].child0 = leaf_ref[0];
      max_depth = 1;
    }
  } else {
    /* Karras: range and split of every internal node; child code: >= 0 internal index, < 0 = ~leaf position */
    int64_t *c0 = malloc(sizeof(int64_t) * (size_t)n_inner), *c1 = malloc(sizeof(int64_t) * (size_t)n_inner);
    for (int64_t i = 0; i < n_inner; i++) {
      int d = lbvh_delta(keys, P, i, i + 1) > lbvh_delta(keys, P, i, i - 1) ? 1 : -1;
      int dmin = lbvh_delta(keys, P, i, i - d);
      int64_t lmax = 2;
      while (lbvh_delta(keys, P, i, i + lmax * d) > dmin) lmax *= 2;
      int64_t l = 0;
      for (int64_t t = lmax / 2; t >= 1; t /= 2)
        if (lbvh_delta(keys, P, i, i + (l + t) * d) > dmin) l += t;
      int64_t j = i + l * d;
      int dnode = lbvh_delta(keys, P, i, j);
      int64_t s = 0;
      for (int64_t t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (lbvh_delta(keys, P, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
      }
      int64_t gamma = i + s * d + (d < 0 ? -1 : 0);
      int64_t lo = i < j ? i : j, hi = i < j ? j : i;
      c0[i] = gamma == lo ? ~gamma : gamma;
      c1[i] = gamma + 1 == hi ? ~(gamma + 1) : gamma + 1;
    }
    /* boxes, sizes, heights bottom-up (a child always has ... no index order: iterate until done) */
    box_t *nbox = malloc(sizeof(box_t) * (size_t)n_inner);
    uint32_t *size = calloc((size_t)n_inner, sizeof(uint32_t)), *height = calloc((size_t)n_inner, sizeof(uint32_t));
    /* kind: 1 = only spheres below the node, 2 = only triangles, 0 = both; lo_pos: lowest sorted
     * position below it.  Triangles only and size <= RTGPU_LBVH_MAX_LEAF: the node is MERGED into one
     * leaf of its parent (lbvh_build.hip fit_pass / emit_nodes). */
    uint32_t *kind = calloc((size_t)n_inner, sizeof(uint32_t)), *lo_pos = calloc((size_t)n_inner, sizeof(uint32_t));
#define LBVH_MERGED(i) (kind[i] == 2u && size[i] <= RTGPU_LBVH_MAX_LEAF)
    uint8_t *done = calloc((size_t)n_inner, 1);
    /* explicit stack post-order from the root */
    int64_t *stk = malloc(sizeof(int64_t) * 256);
    int sp = 0;
    stk[sp++] = 0;
    while (sp > 0) {
      int64_t i = stk[sp - 1];
      int64_t a = c0[i], b = c1[i];
      int ready = (a < 0 || done[a]) && (b < 0 || done[b]);
      if (!ready) {
        if (a >= 0 && !done[a]) stk[sp++] = a;
        if (b >= 0 && !done[b]) stk[sp++] = b;
        if (sp > 250) return 3;
        continue;
      }
      sp--;
      const box_t *ba = a < 0 ? &pbox[(uint32_t)keys[~a]] : &nbox[a];
      const box_t *bb = b < 0 ? &pbox[(uint32_t)keys[~b]] : &nbox[b];
      for (int k = 0; k < 3; k++) {
        nbox[i].lo[k] = fminf(ba->lo[k], bb->lo[k]);
        nbox[i].hi[k] = fmaxf(ba->hi[k], bb->hi[k]);
      }
      uint32_t sa = a < 0 ? 1 : size[a], sb = b < 0 ? 1 : size[b];
      uint32_t ha = a < 0 ? 0 : height[a], hb = b < 0 ? 0 : height[b];
      uint32_t ka = a < 0 ? ((uint32_t)keys[~a] < (uint32_t)ns ? 1u : 2u) : kind[a];
      uint32_t kb = b < 0 ? ((uint32_t)keys[~b] < (uint32_t)ns ? 1u : 2u) : kind[b];
      size[i] = sa + sb;
      kind[i] = ka == kb ? ka : 0u;
      lo_pos[i] = a < 0 ? (uint32_t)~a : lo_pos[a];
      /* the root stays a node even when everything below it merges */
      height[i] = LBVH_MERGED(i) ? (i == 0 ? 1u : 0u) : 1 + (ha > hb ? ha : hb);
      done[i] = 1;
    }
    max_depth = height[0];
    /* renumber: top nodes (subtree >= T prims) first, both groups in Karras order */
    uint32_t T = (uint32_t)((2 * P + 4095) / 4096);
    if (T < 1) T = 1;
    /* emitted: the root and every node that is not merged (a merged node's ancestors up to the
     * first unmerged one are merged too, so "not merged" = "not inside a leaf") */
    uint32_t *new_of = malloc(sizeof(uint32_t) * (size_t)n_inner);
    uint32_t n_top = 0, n_rest = 0;
#define LBVH_KEEP(i) ((i) == 0 || !LBVH_MERGED(i))
    for (int64_t i = 0; i < n_inner; i++) {
      if (!LBVH_KEEP(i)) continue;
      if (size[i] >= T) n_top++; else n_rest++;
    }
    uint32_t at_top = 0, at_rest = n_top;
    for (int64_t i = 0; i < n_inner; i++) {
      new_of[i] = 0xFFFFFFFFu;
      if (LBVH_KEEP(i)) new_of[i] = size[i] >= T ? at_top++ : at_rest++;
    }
    n_emitted = (int64_t)n_top + n_rest;
    for (int64_t i = 0; i < n_inner; i++) {
      if (!LBVH_KEEP(i)) continue;
      rtgpu_node *nd = &nodes[new_of[i]];
      nd->_pad[0] = nd->_pad[1] = 0;
      if (LBVH_MERGED(i)) {  /* i == 0: one leaf and an EMPTY sibling */
        for (int k = 0; k < 3; k++) {
          nd->lo0[k] = nbox[i].lo[k]; nd->hi0[k] = nbox[i].hi[k];
          nd->lo1[k] = INFINITY; nd->hi1[k] = INFINITY;
        }
        nd->child0 = leaf_ref[lo_pos[i]] | ((size[i] - 1u) << 27);
        nd->child1 = RTGPU_REF_EMPTY;
        continue;
      }
      int64_t a = c0[i], b = c1[i];
      const box_t *ba = a < 0 ? &pbox[(uint32_t)keys[~a]] : &nbox[a];
      const box_t *bb = b < 0 ? &pbox[(uint32_t)keys[~b]] : &nbox[b];
      for (int k = 0; k < 3; k++) {
        nd->lo0[k] = ba->lo[k]; nd->hi0[k] = ba->hi[k];
        nd->lo1[k] = bb->lo[k]; nd->hi1[k] = bb->hi[k];
      }
      nd->child0 = a < 0 ? leaf_ref[~a] : LBVH_MERGED(a) ? (leaf_ref[lo_pos[a]] | ((size[a] - 1u) << 27)) : new_of[a];
      nd->child1 = b < 0 ? leaf_ref[~b] : LBVH_MERGED(b) ? (leaf_ref[lo_pos[b]] | ((size[b] - 1u) << 27)) : new_of[b];
    }
#undef LBVH_KEEP
#undef LBVH_MERGED
    free(kind); free(lo_pos);
    free(c0); free(c1); free(nbox); free(size); free(height); free(done); free(stk); free(new_of);
  }
  free(keys); free(pbox); free(leaf_ref);
  out->nodes = nodes; out->n_nodes = (uint64_t)n_emitted;
  out->tris = tris; out->n_tris = (uint64_t)nt;
  out->spheres = sphs; out->n_spheres = (uint64_t)ns;
  out->sphere_meta = meta;
  out->materials = mats; out->n_materials = sc->n_materials;
  out->max_depth = max_depth;
  int pfail;
  out->nodes48 = pack_all(nodes, (uint64_t)n_emitted, &pfail);
  if (pfail) return 4;
  return 0;
}

/* ==== Refit restatement ("builder + refit") =====================
 *
 * CPU restatement of rt_amd/csrc/refit.hip (no reference counterpart: the
 * reference has no BVH).  Keeps the TOPOLOGY of a flattened scene -
 * child references, leaf order, primitive ids - and takes the geometry and
 * material indices from `sc`, which must hold the same number of spheres,
 * triangles and materials the tree was built from:
 *   - every leaf-ordered primitive record is rewritten from the primitive its id
 *     names (triangle edges = f32 differences, as the builders store them);
 *   - the box a node stores for a leaf child = union of the padded boxes of the
 *     leaf's primitives; for an inner child = union of the two boxes that child
 *     stores.  min/max are exact, so any evaluation order gives the same bytes.
 * Output: a new flattened scene (free with rto_flat_free). */
static void refit_child(const rtgpu_flat_scene *fs, const rtgpu_scene_desc *sc, const rtgpu_node *nodes,
                        const rtgpu_tri *tris, const uint32_t *meta, uint32_t ref, float *lo, float *hi) {
  (void)fs;
  if (ref == RTGPU_REF_EMPTY) { for (int k = 0; k < 3; k++) lo[k] = hi[k] = INFINITY; return; }
  if (ref & RTGPU_REF_LEAF) {
    uint32_t first = RTGPU_REF_FIRST(ref), count = RTGPU_REF_COUNT(ref);
    for (int k = 0; k < 3; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; }
    for (uint32_t i = first; i < first + count; i++) {
      uint32_t p = (ref & RTGPU_REF_SPHERE) ? meta[2 * (size_t)i] : tris[i].prim_id;
      box_t b; float c[3];
      prim_box(sc, p, &b, c);
      pad_box(&b);
      for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], b.lo[k]); hi[k] = fmaxf(hi[k], b.hi[k]); }
    }
    return;
  }
  const rtgpu_node *n = &nodes[ref];
  for (int k = 0; k < 3; k++) { lo[k] = fminf(n->lo0[k], n->lo1[k]); hi[k] = fmaxf(n->hi0[k], n->hi1[k]); }
}

RTO_API int rto_refit(const rtgpu_flat_scene *in, const rtgpu_scene_desc *sc, rtgpu_flat_scene *out) {
  memset(out, 0, sizeof(*out));
  if (in->n_spheres != sc->n_spheres || in->n_tris != sc->n_triangles || in->n_materials != sc->n_materials) return 1;
  const size_t nn = (size_t)in->n_nodes, nt = (size_t)in->n_tris, ns = (size_t)in->n_spheres;
  rtgpu_node *nodes = malloc((nn ? nn : 1) * sizeof(rtgpu_node));
  rtgpu_tri *tris = calloc(nt + 8, sizeof(rtgpu_tri));
  rtgpu_sph *sphs = calloc(ns + 8, sizeof(rtgpu_sph));
  uint32_t *meta = calloc(2 * ns + 16, sizeof(uint32_t));
  rtgpu_material *mats = calloc(sc->n_materials ? sc->n_materials : 1, sizeof(rtgpu_material));
  uint8_t *done = calloc(nn ? nn : 1, 1);
  uint32_t *stk = malloc(sizeof(uint32_t) * 2 * (RTGPU_MAX_TREE_DEPTH + 4));
  if (!nodes || !tris || !sphs || !meta || !mats || !done || !stk) return 2;
  memcpy(nodes, in->nodes, nn * sizeof(rtgpu_node));
  memcpy(tris, in->tris, nt * sizeof(rtgpu_tri));
  memcpy(meta, in->sphere_meta, 2 * ns * sizeof(uint32_t));
  for (uint64_t k = 0; k < sc->n_materials; k++) {  /* as rto_build_lbvh */
    mats[k] = sc->materials[k];
    mats[k]._pad[0] = mats[k]._pad[1] = mats[k]._pad[2] = 0;
    if (mats[k].type == RTGPU_MAT_DIELECTRIC) {
      float ior = mats[k].param, inv = 1.0f / ior;
      float qf = (1.0f - inv) / (1.0f + inv), qb = (1.0f - ior) / (1.0f + ior);
      float v[3] = {inv, qf * qf, qb * qb};
      memcpy(mats[k]._pad, v, sizeof(v));
    }
  }
  for (size_t i = 0; i < nt; i++) {
    uint32_t p = tris[i].prim_id;
    if (p < ns || p - ns >= nt) return 3;
    const rtgpu_triangle *t = &sc->triangles[p - ns];
    for (int c = 0; c < 3; c++) {
      tris[i].v0[c] = t->v0[c];
      tris[i].e1[c] = t->v1[c] - t->v0[c];
      tris[i].e2[c] = t->v2[c] - t->v0[c];
    }
    tris[i].material = t->material;
  }
  for (size_t i = 0; i < ns; i++) {
    uint32_t p = meta[2 * i];
    if (p >= ns) return 3;
    const rtgpu_sphere *s = &sc->spheres[p];
    sphs[i].c[0] = s->center[0]; sphs[i].c[1] = s->center[1]; sphs[i].c[2] = s->center[2];
    sphs[i].r = s->radius;
    meta[2 * i + 1] = s->material;
  }
  /* post-order from the root with an explicit stack */
  int sp = 0;
  if (nn) stk[sp++] = 0;
  while (sp > 0) {
    uint32_t i = stk[sp - 1];
    rtgpu_node *n = &nodes[i];
    uint32_t a = n->child0, b = n->child1;
    int ia = !(a & RTGPU_REF_LEAF), ib = !(b & RTGPU_REF_LEAF);  /* EMPTY has the leaf bit */
    if ((ia && a >= nn) || (ib && b >= nn)) return 3;
    int ready = (!ia || done[a]) && (!ib || done[b]);
    if (!ready) {
      if (sp + 2 > (int)(2 * (RTGPU_MAX_TREE_DEPTH + 4))) return 3;
      if (ia && !done[a]) stk[sp++] = a;
      if (ib && !done[b]) stk[sp++] = b;
      continue;
    }
    sp--;
    refit_child(in, sc, nodes, tris, meta, a, n->lo0, n->hi0);
    refit_child(in, sc, nodes, tris, meta, b, n->lo1, n->hi1);
    done[i] = 1;
  }
  free(done); free(stk);
  out->nodes = nodes; out->n_nodes = in->n_nodes;
  out->tris = tris; out->n_tris = in->n_tris;
  out->spheres = sphs; out->n_spheres = in->n_spheres;
  out->sphere_meta = meta;
  out->materials = mats; out->n_materials = sc->n_materials;
  out->max_depth = in->max_depth;
  int pfail;
  out->nodes48 = pack_all(nodes, in->n_nodes, &pfail);
  if (pfail) return 4;
  /* four-wide nodes: the collapse stays, every child box is re-read from the refitted
   * BVH2 through the source map and packed again around the same reference */
  if (in->nodes4 && in->nodes4_src) {
    const size_t n4 = (size_t)in->n_nodes4;
    rtgpu_node4 *w4 = malloc((n4 ? n4 : 1) * sizeof(rtgpu_node4));
    uint32_t *src = malloc((n4 ? n4 : 1) * 16);
    if (!w4 || !src) return 2;
    memcpy(w4, in->nodes4, n4 * sizeof(rtgpu_node4));
    memcpy(src, in->nodes4_src, n4 * 16);
    for (size_t i = 0; i < n4; i++)
      for (int k = 0; k < 4; k++) {
        uint32_t sidx = src[4 * i + k];
        if (sidx == 0xFFFFFFFFu) continue;
        if ((sidx >> 1) >= nn) return 3;
        uint32_t *w = &w4[i].w[6 * k];
        uint32_t ref = (w[0] & 0xFFu) | ((w[1] & 0xFFu) << 8) | ((w[2] & 0xFFu) << 16) | ((w[3] & 0xFFu) << 24);
        const rtgpu_node *b = &nodes[sidx >> 1];
        memcpy(w, (sidx & 1u) ? b->lo1 : b->lo0, 12);
        memcpy(w + 3, (sidx & 1u) ? b->hi1 : b->hi0, 12);
        if ((w[4] & 0x7F800000u) == 0x7F800000u || (w[5] & 0x7F800000u) == 0x7F800000u) return 4;
        for (int j = 0; j < 4; j++)
          if (!pack_plane(w[j], (ref >> (8 * j)) & 0xFFu, j < 3, &w[j])) return 4;
      }
    out->nodes4 = w4;
    out->nodes4_src = src;
    out->n_nodes4 = in->n_nodes4;
    out->max_stack4 = in->max_stack4;
  }
  return 0;
}
