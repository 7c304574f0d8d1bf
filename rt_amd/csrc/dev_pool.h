// dev_pool.h — device memory kept between the GPU builds of one context.  A frame loop
// that calls rtgpu_build_device every frame asks for the same two dozen sizes each time;
// hipMalloc / hipFree of them cost as much as the build's kernels.  Blocks go back to an
// idle list instead of hipFree and are handed out again (best fit, at most twice the
// size asked for); a block that sat idle through two builds is freed.  One pool per
// context, used from the context's thread only.  Included by the .hip files only.
#ifndef RTGPU_DEV_POOL_H
#define RTGPU_DEV_POOL_H

#include <hip/hip_runtime.h>

#include <unordered_map>
#include <vector>

namespace rtgpu {

struct DevPool {
  struct Idle { size_t bytes; void *p; int age; };
  std::unordered_map<void *, size_t> live;  // blocks handed out
  std::vector<Idle> idle;                   // blocks waiting for reuse

  hipError_t get(void **p, size_t bytes) {
    bytes = bytes ? (bytes + 255) / 256 * 256 : 256;
    int best = -1;
    for (int i = 0; i < (int)idle.size(); i++)
      if (idle[i].bytes >= bytes && idle[i].bytes / 2 <= bytes && (best < 0 || idle[i].bytes < idle[best].bytes)) best = i;
    if (best >= 0) {
      *p = idle[best].p;
      live[*p] = idle[best].bytes;
      idle.erase(idle.begin() + best);
      return hipSuccess;
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {  // out of memory: give the idle blocks back and try once more
      (void)hipGetLastError();
      trim();
      e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess) live[*p] = bytes;
    return e;
  }
  // a block of this pool goes to the idle list; any other device pointer is hipFree'd
  void put(void *p) {
    if (!p) return;
    auto it = live.find(p);
    if (it == live.end()) { (void)hipFree(p); return; }
    idle.push_back(Idle{it->second, p, 0});
    live.erase(it);
  }
  // after a build: blocks that no build has asked for since the one before are freed
  void end_of_build() {
    size_t keep = 0;
    for (size_t i = 0; i < idle.size(); i++) {
      if (idle[i].age >= 1) { (void)hipFree(idle[i].p); continue; }
      idle[i].age++;
      idle[keep++] = idle[i];
    }
    idle.resize(keep);
  }
  void trim() {
    for (auto &b : idle) (void)hipFree(b.p);
    idle.clear();
  }
};

// pool-or-plain allocation for code that runs with and without a pool
inline hipError_t dev_get(DevPool *pool, void **p, size_t bytes) {
  return pool ? pool->get(p, bytes) : hipMalloc(p, bytes ? bytes : 1);
}
inline void dev_put(DevPool *pool, void *p) {
  if (pool) pool->put(p);
  else (void)hipFree(p);
}

}  // namespace rtgpu

#endif
