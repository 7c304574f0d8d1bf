// bvh4_build.cpp — the four-wide packed nodes (rtgpu_node4, include/rtgpu_format.h)
// made from the flattened BVH2 of either builder.  Host-only.  No reference
// counterpart (the reference has no BVH); the wide-node idea is
// SURVEY.md §8f row 3 ("wide (BVH4/8) compressed nodes").
//
// Collapse: a node starts with the two children of a BVH2 node and absorbs, while
// it has fewer than four, the INNER child of largest surface area (ties: the
// first), putting that child's two children in its place - so child order stays
// the BVH2's left-to-right order and every box is one the BVH2 already stores.
// Nodes are emitted parent first, children's subtrees in child order, within the top
// of the tree and within every subtree below it (built as parallel tasks); the final
// numbering then puts the RTGPU_TOP_NODES4 nodes nearest the root first, breadth first, so
// that the top of the tree is a prefix of the array (the trace kernel keeps it in LDS),
// and leaves the others in emission order.
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "pack48.h"
#include "rtgpu_internal.h"

namespace rtgpu {

namespace {

struct Child {
  float lo[3], hi[3];
  uint32_t ref;
  uint32_t src;  // (BVH2 node << 1) | child slot: where this box is stored
};

inline float half_area(const Child &c) {
  float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
  return dx * dy + dy * dz + dz * dx;
}

// A node before packing: exact planes, references still to be finalised.  `ref` of an
// inner child is an index LOCAL to the vector the node sits in (top part or one task).
struct Tmp {
  uint32_t w[24];
  uint32_t ref[4];
  uint32_t src[4];    // Child::src, 0xFFFFFFFF for a missing child
  uint8_t  n;         // children (2..4)
  uint8_t  task[4];   // 1: ref[k] is a task id (top part only)
};

struct Collapser {
  const rtgpu_node *nodes;
  uint64_t n_nodes;
  std::vector<Tmp> out;
  uint32_t max_stack = 0;
  bool bad = false;
  // top part only: subtrees below `task_level` node4 levels become tasks
  int task_level = -1;
  struct Task { uint32_t node2, above; };
  std::vector<Task> *tasks = nullptr;

  // emits the node4 for the subtree under BVH2 node i; `above` = stack entries that can
  // lie below this node's own pushes (siblings of its ancestors)
  uint32_t emit(uint32_t i, uint32_t above, uint32_t depth, int level) {
    if (i >= n_nodes || depth > RTGPU_MAX_TREE_DEPTH + 1) { bad = true; return 0; }
    Child ch[4];
    int n = 0;
    auto take = [&](uint32_t at, int which, Child &c) {
      const rtgpu_node &nd = nodes[at];
      const float *lo = which ? nd.lo1 : nd.lo0, *hi = which ? nd.hi1 : nd.hi0;
      for (int k = 0; k < 3; k++) { c.lo[k] = lo[k]; c.hi[k] = hi[k]; }
      c.ref = which ? nd.child1 : nd.child0;
      c.src = (at << 1) | (uint32_t)which;
    };
    take(i, 0, ch[0]);
    take(i, 1, ch[1]);
    n = 2;
    uint32_t d2 = depth + 1;  // BVH2 depth of the deepest absorbed child
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
      take(r, 0, ch[best]);
      take(r, 1, ch[best + 1]);
      n++;
      d2 = depth + 2;
    }
    const uint32_t me = (uint32_t)out.size();
    out.emplace_back();
    const uint32_t below = above + (uint32_t)(n - 1);
    if (below > max_stack) max_stack = below;
    Tmp t;
    std::memset(&t, 0, sizeof(t));
    t.n = (uint8_t)n;
    for (int k = 0; k < 4; k++) {
      if (k < n && ch[k].ref != RTGPU_REF_EMPTY) {
        std::memcpy(&t.w[6 * k], ch[k].lo, 12);
        std::memcpy(&t.w[6 * k + 3], ch[k].hi, 12);
        t.src[k] = ch[k].src;
        if (ch[k].ref & RTGPU_REF_LEAF) {
          t.ref[k] = ch[k].ref;
        } else if (level == task_level) {
          t.ref[k] = (uint32_t)tasks->size();
          t.task[k] = 1;
          tasks->push_back(Task{ch[k].ref, below});
        } else {
          t.ref[k] = emit(ch[k].ref, below, d2, level + 1);
        }
      } else {
        for (int j = 0; j < 6; j++) t.w[6 * k + j] = RTGPU_NODE4_FAR;
        t.ref[k] = RTGPU_REF_EMPTY;
        t.src[k] = 0xFFFFFFFFu;
      }
    }
    if (bad) return 0;
    out[me] = t;
    return me;
  }
};

// planes + final references -> packed node; false if a plane has no packed form
bool pack4(const Tmp &t, rtgpu_node4 *out) {
  for (int k = 0; k < 4; k++) {
    uint32_t w[6];
    std::memcpy(w, &t.w[6 * k], sizeof(w));
    for (int j = 0; j < 4; j++)  // lo.x lo.y lo.z down, hi.x up
      if (!embed_byte(w[j], (t.ref[k] >> (8 * j)) & 0xFFu, j < 3, &w[j])) return false;
    if ((w[4] & 0x7F800000u) == 0x7F800000u || (w[5] & 0x7F800000u) == 0x7F800000u) return false;
    std::memcpy(&out->w[6 * k], w, sizeof(w));
  }
  return true;
}

// fn(t) for t in [0, n) on up to T threads (dynamic: one atomic counter)
template <typename F>
void parallel_for(size_t n, int T, F fn) {
  std::atomic<size_t> next{0};
  auto work = [&] { for (size_t t; (t = next.fetch_add(1)) < n;) fn(t); };
  std::vector<std::thread> th;
  for (int k = 1; k < T && (size_t)k < n; k++) {
    try { th.emplace_back(work); } catch (const std::system_error &) { break; }
  }
  work();
  for (auto &x : th) x.join();
}

}  // namespace

// nodes (64-byte BVH2) -> four-wide packed nodes; malloc'ed, 64-byte aligned.  NULL with
// err set if a plane has no packed form, the tree is malformed or memory runs out.
// The top kTaskLevel + 1 node4 levels are collapsed on the calling thread; every subtree
// below becomes a task (at most 4^(kTaskLevel+1) of them) built on its own into a private
// vector, and the pieces are laid out [top][task 0][task 1]...: the bytes do not depend on
// the number of threads.
rtgpu_node4 *collapse_nodes4(const rtgpu_node *nodes, uint64_t n, uint64_t *n_out, uint32_t *max_stack,
                             uint32_t **src_out, std::string &err) {
  constexpr int kTaskLevel = 3;
  constexpr int kThreads = 16;
  const char *fail = "four-wide nodes: malformed tree, or a box plane that is not finite / too close to FLT_MAX";
  Collapser top;
  std::vector<Collapser::Task> tasks;
  top.nodes = nodes;
  top.n_nodes = n;
  top.task_level = kTaskLevel;
  top.tasks = &tasks;
  uint32_t root = top.emit(0, 0, 0, 0);
  if (top.bad || root != 0) { err = fail; return nullptr; }
  std::vector<Collapser> sub(tasks.size());
  parallel_for(tasks.size(), kThreads, [&](size_t t) {
    sub[t].nodes = nodes;
    sub[t].n_nodes = n;
    sub[t].emit(tasks[t].node2, tasks[t].above, 0, 0);  // depth is re-counted from the task's root: the bound stays valid
  });
  std::vector<size_t> base(tasks.size() + 1);
  base[0] = top.out.size();
  uint32_t ms = top.max_stack;
  for (size_t t = 0; t < tasks.size(); t++) {
    if (sub[t].bad) { err = fail; return nullptr; }
    base[t + 1] = base[t] + sub[t].out.size();
    if (sub[t].max_stack > ms) ms = sub[t].max_stack;
  }
  const size_t total = base[tasks.size()];
  if (total >= (1ull << 31) || total * sizeof(rtgpu_node4) >= (1ull << 32)) {
    err = "four-wide nodes: buffer would reach 4 GiB";
    return nullptr;
  }
  size_t bytes = (total * sizeof(rtgpu_node4) + 63) / 64 * 64;
  rtgpu_node4 *p = (rtgpu_node4 *)std::aligned_alloc(64, bytes);
  size_t src_bytes = (total * 16 + 63) / 64 * 64;
  uint32_t *src = (uint32_t *)std::aligned_alloc(64, src_bytes);
  if (!p || !src) { std::free(p); std::free(src); err = "out of host memory for the four-wide nodes"; return nullptr; }
  std::memset((char *)p + total * sizeof(rtgpu_node4), 0, bytes - total * sizeof(rtgpu_node4));
  std::memset((char *)src + total * 16, 0, src_bytes - total * 16);
  std::atomic<bool> bad{false};
  // emission index of a node's inner child (pieces: 0 = top part at 0, 1 + t = task t at base[t])
  auto child_index = [&](size_t piece, const Tmp &t, int c) -> uint32_t {
    const size_t at = piece == 0 ? 0 : base[piece - 1];
    return t.task[c] ? (uint32_t)base[t.ref[c]] : (uint32_t)(at + t.ref[c]);
  };
  auto node_at = [&](uint32_t g, size_t *piece_out) -> const Tmp & {
    size_t piece = 0;
    if (g >= base[0]) {  // the task whose range holds g
      size_t lo = 0, hi = tasks.size();
      while (hi - lo > 1) { size_t mid = (lo + hi) / 2; (base[mid] <= g ? lo : hi) = mid; }
      piece = lo + 1;
    }
    *piece_out = piece;
    return piece == 0 ? top.out[g] : sub[piece - 1].out[g - base[piece - 1]];
  };
  // final numbering: emission index -> array index.  Breadth first from the root for the first
  // RTGPU_TOP_NODES4 nodes, emission order for the rest.
  std::vector<uint32_t> perm(total);
  {
    std::vector<uint32_t> order;  // emission indices of the top nodes, breadth first
    order.reserve(RTGPU_TOP_NODES4);
    order.push_back(0);
    for (size_t head = 0; head < order.size() && order.size() < RTGPU_TOP_NODES4; head++) {
      size_t piece;
      const Tmp &t = node_at(order[head], &piece);
      for (int c = 0; c < 4 && order.size() < RTGPU_TOP_NODES4; c++)
        if (!(t.ref[c] & RTGPU_REF_LEAF)) order.push_back(child_index(piece, t, c));
    }
    std::vector<uint8_t> in_top(total, 0);
    for (size_t k = 0; k < order.size(); k++) { perm[order[k]] = (uint32_t)k; in_top[order[k]] = 1; }
    uint32_t next = (uint32_t)order.size();
    for (size_t g = 0; g < total; g++)
      if (!in_top[g]) perm[g] = next++;
  }
  parallel_for(tasks.size() + 1, kThreads, [&](size_t piece) {
    const std::vector<Tmp> &v = piece == 0 ? top.out : sub[piece - 1].out;
    const size_t at = piece == 0 ? 0 : base[piece - 1];
    for (size_t k = 0; k < v.size(); k++) {
      Tmp t = v[k];
      for (int c = 0; c < 4; c++) {
        if (t.ref[c] & RTGPU_REF_LEAF) continue;  // leaves and EMPTY
        t.ref[c] = perm[child_index(piece, v[k], c)];
      }
      const size_t to = perm[at + k];
      if (!pack4(t, &p[to])) bad = true;
      std::memcpy(&src[4 * to], t.src, 16);
    }
  });
  if (bad) { std::free(p); std::free(src); err = fail; return nullptr; }
  *src_out = src;
  *n_out = total;
  *max_stack = ms;
  return p;
}

}  // namespace rtgpu

// Test hook: node count and stack bound of the host collapse of a BVH2 (what the device
// collapse of collapse4.hip must reproduce; its numbering differs).  0 on success.
extern "C" int rtgpu_debug_collapse4_counts(const void *nodes, uint64_t n, uint64_t *n4, uint32_t *max_stack) {
  std::string err;
  uint32_t *src = nullptr;
  rtgpu_node4 *p = rtgpu::collapse_nodes4((const rtgpu_node *)nodes, n, n4, max_stack, &src, err);
  if (!p) return 1;
  std::free(p);
  std::free(src);
  return 0;
}
