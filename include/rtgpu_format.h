/*
 * rtgpu_format.h — byte layout of the flattened scene in HBM (SURVEY.md §8a
 * row a2).  This is a DATA FORMAT shared by the product (rt_amd/csrc) and the
 * CPU oracle (oracle/rt_oracle.c): both walk the same buffers, which is what
 * makes primary-hit ids comparable bit-for-bit, ties included.  It has no
 * counterpart in the reference (no BVH exists there).
 *
 * All records are arrays of 32-bit words, little endian; f32 = IEEE binary32
 * (rt::f32, code/first.hpp:26), u32 = rt::u32 (code/first.hpp:18).
 */
#ifndef RTGPU_FORMAT_H
#define RTGPU_FORMAT_H

#include <stdint.h>

/* BVH2 node, 64 bytes, 64-byte aligned: one node = both children's boxes, so a
 * traversal step costs exactly one 64-byte fetch (4 x dwordx4 per lane). */
typedef struct rtgpu_node {
  float    lo0[3], hi0[3];   /* AABB of child 0 */
  float    lo1[3], hi1[3];   /* AABB of child 1 */
  uint32_t child0, child1;   /* child references, see below */
  uint32_t _pad[2];
} rtgpu_node;

/* Packed BVH2 node, 48 bytes, 16-byte aligned: the form the trace loop reads for
 * trees of RTGPU_NODE48_MIN_NODES nodes or more (the ones it cannot keep in LDS).
 * Every node fetch of such a tree goes through the texture path, whose cost is per
 * load instruction: 48 bytes are three 16-byte loads instead of four.
 * The twelve planes keep the order of rtgpu_node (word w: 0-2 lo0, 3-5 hi0, 6-8 lo1,
 * 9-11 hi1); the two child references ride in the LOW BYTES of eight of them:
 *   byte k of child0 = low byte of word k      (k = 0..3: lo0.x lo0.y lo0.z hi0.x)
 *   byte k of child1 = low byte of word 8 + k  (k = 0..3: lo1.z hi1.x hi1.y hi1.z)
 * and each of those eight planes is moved to the NEAREST float that carries its
 * byte on the conservative side: a lo plane becomes the largest float <= the exact
 * plane, a hi plane the smallest float >= it (at most 255 floats away, 511 where
 * the move crosses zero among the denormals: the box grows by < 3.1e-5 of the
 * plane's magnitude on those eight planes).
 * The float AS STORED is the plane: nothing is masked before the slab test.  Words
 * 4-7 (hi0.y hi0.z lo1.x lo1.y) are the exact planes.  Both the kernels and the
 * oracle traverse this array when it is present, so hits and counters stay
 * comparable bit for bit.  A tree with a non-finite plane (the empty child of a
 * scene that fits one leaf) has no packed form; such trees are far below the
 * threshold. */
typedef struct rtgpu_node48 {
  uint32_t w[12];
} rtgpu_node48;
/* GPU LBVH builder (rt_amd/csrc/lbvh_build.hip, restated in oracle/rt_oracle.c): a subtree
 * of at most this many TRIANGLES (and nothing else) becomes a single leaf; its triangles
 * are adjacent in the leaf-ordered array.  The trace loop tests two triangles per leaf
 * step, so a leaf of two costs what a leaf of one does and saves a node.  Spheres stay
 * one per leaf. */
#ifndef RTGPU_LBVH_MAX_LEAF
#define RTGPU_LBVH_MAX_LEAF 2u
#endif
#define RTGPU_NODE48_MIN_NODES 1366u

/* Four-wide packed node, 96 bytes, 16-byte aligned: up to four children per node,
 * made by collapsing the BVH2 (a node absorbs the inner child of largest surface
 * area until it has four children or only leaves; child order = BVH2 order).  On
 * trees too large for LDS the time of a ray follows the number of DEPENDENT node
 * fetches it makes; a four-wide tree roughly halves that number.
 * Child k (0..3) owns words 6k .. 6k+5 = lo.x lo.y lo.z hi.x hi.y hi.z; byte j of
 * its reference is the low byte of word 6k + j, and those four planes sit on the
 * conservative side of the exact ones exactly as in rtgpu_node48 (lo planes moved
 * down, hi.x up, < 256 floats); hi.y and hi.z are exact.  A missing child has the
 * reference RTGPU_REF_EMPTY and a box collapsed at RTGPU_NODE4_FAR on every axis,
 * which no ray reaches.  Present from RTGPU_NODE48_MIN_NODES BVH2 nodes up, and
 * then the array the trace loop and the oracle traverse (a refit keeps the
 * collapse and re-reads every child box from the refitted BVH2).
 * Traversal order at a node: children whose slab test passes, by ascending key
 * ((bits of tnear) & 0x7FFFFFFC) | k - nearest first, the others pushed farthest
 * first. */
typedef struct rtgpu_node4 {
  uint32_t w[24];
} rtgpu_node4;
#define RTGPU_NODE4_FAR 0x7F000000u /* 2^127 as float bits */
/* Node order of the four-wide array: node 0 is the root and the first RTGPU_TOP_NODES4 nodes are
 * the ones nearest the root, breadth first (host collapse; the GPU collapse numbers the whole
 * tree level by level), so the top of the tree is a prefix of the array: the trace kernel keeps
 * as much of it in LDS as fits beside the traversal stacks.  Every inner child reference names a
 * later node. */
#define RTGPU_TOP_NODES4 1024u

/* Slot tree: the four-wide tree AND its primitives as one array of 64-byte slots (16 words), made on the
 * device from the four-wide nodes and the leaf-ordered arrays (rt_amd/csrc/slot_tree.hip; restated in
 * oracle/rt_oracle.c; rewritten in place after a refit, whose topology and therefore slot numbering stay)
 * and traversed in preference to them by the trace loop and the oracle.  On trees in
 * memory the trace loop's time follows the 16-byte loads a lane issues and the 128-byte lines they touch
 * (DESIGN.md §5.2): a node is four loads here instead of six, no record lies across a line, and the
 * children of a node are adjacent.
 *   node slot      w0 w1 w2  origin x y z (f32): per axis the smallest lo plane of the node's children, moved
 *                            DOWN to the nearest float whose LOW BYTE is the child byte of child 0 / 1 / 2;
 *                            the float AS STORED is the origin, nothing is masked
 *                  w3        bits 0..23 FAMILY = slot of the node's first child; bits 24..31 child byte of child 3
 *                  w[4+3k+a] child k (0..3), axis a (x y z): low half = binary16 of (lo - origin) rounded DOWN,
 *                            high half = binary16 of (hi - origin) rounded UP (differences taken in double;
 *                            zero and NORMAL halves only: a positive difference below 2^-14 becomes 0 / 2^-14);
 *                            a missing child has +inf (0x7C00) in all six halves, which no ray reaches
 *   child byte     bits 0..3 = the top four bits of the child's slot reference (leaf, sphere, count - 1);
 *                  bits 4..7 = where the child starts in the family, in slots
 *   family         the children of one node in child order: an inner child is 1 slot (its node), a leaf of n
 *                  primitives n slots.  Slot 0 is the root; the family of four-wide node i starts at
 *                  1 + (slots of the families of nodes 0 .. i-1), so every child lies after its parent
 *   triangle slot  words 0..11 = rtgpu_tri, 12..15 zero
 *   sphere slot    words 0..3 = rtgpu_sph, 4 = prim_id, 5 = material, 6..15 zero
 * Slot reference (the traversal-stack entry over this tree): bit 31 leaf, bit 30 sphere, bits 28..29
 * count - 1, bits 0..27 the first slot; inner node: its slot.  The slab test of child k, axis a is
 *   t = fmaf(half, 1/d, adj),  adj = fmaf(origin, 1/d, -o/d)   (adj once per node and axis)
 * with the half converted exactly (v_fma_mix_f32 on gfx950); traversal order as over rtgpu_node4.
 * The boxes contain the four-wide nodes' boxes (planes move outward by less than 2^-10 of their distance
 * from the origin), so the hits are the same and only the visit counts grow (by ~0.2 %).
 * A tree has NO slot form - and is traversed through rtgpu_node4 - if a leaf holds more than four
 * primitives, a plane lies more than 65504 above its node's origin, or it would take 2^24 slots (1 GiB)
 * or more. */
#define RTGPU_SLOT_WORDS 16u
#define RTGPU_SLOT_MAX   (1u << 24)

/* Child reference (also the traversal-stack entry):
 *   bit 31 = 0 : inner node, bits 0..30 = node index
 *   bit 31 = 1 : leaf; bit 30 = 1 sphere / 0 triangle;
 *                bits 27..29 = count-1 (1..8 primitives, one type per leaf);
 *                bits 0..26  = first index into the leaf-ordered array
 *   0xFFFFFFFF : empty child (only under the root of a scene whose primitives fit
 *                one leaf: fewer than 2, or 2..4 of one type); its box sits
 *                at +inf on every axis and the closest-hit distance starts at
 *                FLT_MAX, so the slab test can never pass for it              */
#define RTGPU_REF_LEAF    0x80000000u
#define RTGPU_REF_SPHERE  0x40000000u
#define RTGPU_REF_EMPTY   0xFFFFFFFFu
#define RTGPU_REF_COUNT(r) ((((r) >> 27) & 7u) + 1u)
#define RTGPU_REF_FIRST(r) ((r) & 0x07FFFFFFu)
#define RTGPU_MAX_PRIMS_PER_TYPE 0x07FFFFF8u

/* Triangle, 48 bytes, leaf order: v0, e1 = v1-v0, e2 = v2-v0 (f32 subtraction
 * done once on the host); the spare w lanes carry id and material. */
typedef struct rtgpu_tri {
  float v0[3]; uint32_t prim_id;
  float e1[3]; uint32_t material;
  float e2[3]; uint32_t _pad;
} rtgpu_tri;

/* Sphere, 16 bytes, leaf order; {prim_id, material} live in a parallel array
 * (8 bytes) fetched only for the closest hit. */
typedef struct rtgpu_sph {
  float c[3]; float r;
} rtgpu_sph;

/* Node order: node 0 is the root; the first RTGPU_TOP_NODES nodes are numbered
 * breadth-first (index < K  <=>  among the K nodes nearest the root: the kernel
 * keeps as many of them in LDS as fit), the remaining nodes depth-first. */
#define RTGPU_TOP_NODES 4096u

/* Traversal stack bound: the builder never produces a tree deeper than this
 * (root = depth 0), and both the kernel and the oracle size stacks from it. */
#define RTGPU_MAX_TREE_DEPTH 62u

#endif
