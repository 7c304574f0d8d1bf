// bvh4_build.cpp — the four-wide packed nodes (rtgpu_node4, include/rtgpu_format.h)
// made from the flattened BVH2 of either builder.  Host-only.  No reference
// counterpart (the reference has no BVH); the wide-node idea is
// SURVEY.md §8f row 3 ("wide (BVH4/8) compressed nodes").
//
// Collapse: a node starts with the two children of a BVH2 node and absorbs, while
// it has fewer than four, the INNER child of largest surface area (ties: the
// first), putting that child's two children in its place - so child order stays
// the BVH2's left-to-right order and every box is one the BVH2 already stores.
// Nodes are emitted parent first, children's subtrees in child order.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pack48.h"
#include "rtgpu_internal.h"

namespace rtgpu {

namespace {

struct Child {
  float lo[3], hi[3];
  uint32_t ref;
};

inline float half_area(const Child &c) {
  float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
  return dx * dy + dy * dz + dz * dx;
}

struct Collapser {
  const rtgpu_node *nodes;
  uint64_t n_nodes;
  std::vector<rtgpu_node4> out;
  uint32_t max_stack = 0;
  bool bad = false;

  // emits the node4 for the subtree under BVH2 node i; `above` = stack entries that can
  // lie below this node's own pushes (siblings of its ancestors)
  uint32_t emit(uint32_t i, uint32_t above, uint32_t depth) {
    if (i >= n_nodes || depth > RTGPU_MAX_TREE_DEPTH + 1) { bad = true; return 0; }
    Child ch[4];
    int n = 0;
    auto take = [&](const rtgpu_node &nd, int which, Child &c) {
      const float *lo = which ? nd.lo1 : nd.lo0, *hi = which ? nd.hi1 : nd.hi0;
      for (int k = 0; k < 3; k++) { c.lo[k] = lo[k]; c.hi[k] = hi[k]; }
      c.ref = which ? nd.child1 : nd.child0;
    };
    take(nodes[i], 0, ch[0]);
    take(nodes[i], 1, ch[1]);
    n = 2;
    while (n < 4) {
      int best = -1;
      float best_area = -1.0f;
      for (int k = 0; k < n; k++) {
        if (ch[k].ref & RTGPU_REF_LEAF) continue;  // leaves (and EMPTY) stay
        float a = half_area(ch[k]);
        if (a > best_area) { best_area = a; best = k; }
      }
      if (best < 0) break;
      const uint32_t r = ch[best].ref;
      if (r >= n_nodes) { bad = true; return 0; }
      for (int k = n; k > best + 1; k--) ch[k] = ch[k - 1];  // open a slot after `best`
      take(nodes[r], 0, ch[best]);
      take(nodes[r], 1, ch[best + 1]);
      n++;
    }
    const uint32_t me = (uint32_t)out.size();
    out.emplace_back();
    const uint32_t below = above + (uint32_t)(n - 1);
    if (below > max_stack) max_stack = below;
    uint32_t refs[4];
    for (int k = 0; k < n; k++)
      refs[k] = (ch[k].ref & RTGPU_REF_LEAF) ? ch[k].ref : emit(ch[k].ref, below, depth + 1);
    if (bad) return 0;
    rtgpu_node4 nd;
    for (int k = 0; k < 4; k++) {
      uint32_t w[6], ref = RTGPU_REF_EMPTY;
      if (k < n && ch[k].ref != RTGPU_REF_EMPTY) {
        std::memcpy(w, ch[k].lo, 12);
        std::memcpy(w + 3, ch[k].hi, 12);
        ref = refs[k];
      } else {
        for (int j = 0; j < 6; j++) w[j] = RTGPU_NODE4_FAR;
      }
      for (int j = 0; j < 4; j++)  // lo.x lo.y lo.z down, hi.x up
        if (!embed_byte(w[j], (ref >> (8 * j)) & 0xFFu, j < 3, &w[j])) { bad = true; return 0; }
      if ((w[4] & 0x7F800000u) == 0x7F800000u || (w[5] & 0x7F800000u) == 0x7F800000u) { bad = true; return 0; }
      std::memcpy(&nd.w[6 * k], w, sizeof(w));
    }
    out[me] = nd;
    return me;
  }
};

}  // namespace

// nodes (64-byte BVH2) -> four-wide packed nodes; malloc'ed, 64-byte aligned.  NULL with
// err set if a plane has no packed form, the tree is malformed or memory runs out.
rtgpu_node4 *collapse_nodes4(const rtgpu_node *nodes, uint64_t n, uint64_t *n_out, uint32_t *max_stack,
                             std::string &err) {
  Collapser C;
  C.nodes = nodes;
  C.n_nodes = n;
  C.out.reserve((size_t)(n / 2 + 16));
  uint32_t root = C.emit(0, 0, 0);
  if (C.bad || root != 0) {
    err = "four-wide nodes: malformed tree, or a box plane that is not finite / too close to FLT_MAX";
    return nullptr;
  }
  size_t bytes = (C.out.size() * sizeof(rtgpu_node4) + 63) / 64 * 64;
  rtgpu_node4 *p = (rtgpu_node4 *)std::aligned_alloc(64, bytes);
  if (!p) { err = "out of host memory for the four-wide nodes"; return nullptr; }
  std::memset(p, 0, bytes);
  std::memcpy(p, C.out.data(), C.out.size() * sizeof(rtgpu_node4));
  *n_out = C.out.size();
  *max_stack = C.max_stack;
  return p;
}

}  // namespace rtgpu
