// group.cpp — several GPUs of one node behind one handle (include/rtgpu.h).
//
// No reference counterpart: the reference is one process, one thread, one display GPU
// (code/gfx/gfx.cpp:36).  Shape, as §8e specifies it: ONE process, one rtgpu context and one
// worker thread per GPU, one RCCL communicator made by ncclCommInitAll; image tiles
// interleaved over the GPUs, scene replicated, and one exchange step per frame - ncclGroupStart,
// N-1 ncclRecv on the root / one ncclSend per peer (direct: every peer uses its own xGMI link
// to the root at the same time), ncclGroupEnd - followed by the untile and resolve kernels on
// the root.  The exchange runs on a second stream per GPU from a copy of the tile buffer, so the
// trace of frame k+1 can start while frame k is still on the wire (two frames in flight).
//
// The workers touch HIP and RCCL only; they never call back into the caller, so a
// single-threaded display device stays with the calling thread (gfx.cpp:36).
//
// RCCL is bound at run time (dlopen of librccl.so.1): a process that already holds PyTorch's
// copy of the library then shares it instead of loading a second one, and single-GPU users of
// librtgpu.so do not need RCCL at all.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rtgpu_internal.h"

namespace {

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string &err) {
    if (handle) return true;
    for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
      handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (handle) break;
    }
    if (!handle) { err = std::string("cannot load RCCL (librccl.so.1): ") + dlerror(); return false; }
    auto sym = [&](const char *n) { return dlsym(handle, n); };
    CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    CommAbort = (decltype(CommAbort))sym("ncclCommAbort");
    GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    Send = (decltype(Send))sym("ncclSend");
    Recv = (decltype(Recv))sym("ncclRecv");
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    if (!CommInitAll || !CommDestroy || !CommAbort || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) {
      err = "librccl.so.1 lacks an entry point (ncclCommInitAll / ncclSend / ncclRecv ...)";
      return false;
    }
    return true;
  }
};
Rccl g_rccl;

struct Buf {  // grow-only device buffer
  void *p = nullptr;
  size_t bytes = 0;
  hipError_t want(size_t n) {
    if (bytes >= n) return hipSuccess;
    (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    hipError_t e = hipMalloc(&p, n);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  void release() { (void)hipFree(p); p = nullptr; bytes = 0; }
};

struct Worker {
  int index = 0, device = 0;
  rtgpu_ctx *ctx = nullptr;
  ncclComm_t comm = nullptr;
  hipStream_t trace_stream = nullptr, comm_stream = nullptr;
  Buf compact;            // this GPU's tile sums: resident between frames (progressive accumulation)
  Buf send[2];            // the copy a frame's exchange reads
  hipEvent_t ev_traced[2] = {nullptr, nullptr}, ev_sent[2] = {nullptr, nullptr};
  unsigned long long *h_counters[2] = {nullptr, nullptr};  // pinned: counters of the frame in slot b
  // root only
  Buf gathered[2], image[2], rgba[2];
  hipEvent_t ev_frame[2] = {nullptr, nullptr};
  std::thread thread;
  std::string err;
  int rc = 0;
};

struct Frame {
  uint32_t width = 0, height = 0;
  uint64_t n_pad = 0;
};

}  // namespace

struct rtgpu_group {
  std::vector<Worker> workers;
  bool use_rccl = true;   // false: a device is listed twice (rehearsal) - peer-to-peer copies instead
  bool self_rccl = false; // one device + RTGPU_GROUP_SELF_RCCL=1: a one-rank communicator, the root's own buffer goes
                          // through a grouped ncclSend / ncclRecv to itself (exercises the binding on a one-GPU box)
  int debug_fail_exchange = -1;  // RTGPU_GROUP_DEBUG_FAIL_EXCHANGE=rank (tests): that rank fails at the start of the exchange step
  bool broken = false;    // an exchange failed half way: the communicators were aborted, only destroy is left
  std::string err;
  // job hand-off: the caller posts one function, every worker runs it on its GPU
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  std::function<int(Worker &)> job;
  uint64_t job_seq = 0;
  int done = 0;
  bool quit = false;
  // frames in flight (at most two): slot = sequence number & 1
  uint64_t begun = 0, ended = 0;
  Frame frames[2];
  // what the resident tile sums hold (for RTGPU_FLAG_ACCUMULATE)
  bool resident = false;
  uint32_t res_w = 0, res_h = 0, res_tile = 0;
};

namespace {

std::string g_group_error;

void worker_main(rtgpu_group *g, Worker *w) {
  uint64_t seen = 0;
  for (;;) {
    std::function<int(Worker &)> job;
    {
      std::unique_lock<std::mutex> lk(g->m);
      g->cv_job.wait(lk, [&] { return g->quit || g->job_seq > seen; });
      if (g->quit) return;
      seen = g->job_seq;
      job = g->job;
    }
    w->err.clear();
    if (hipSetDevice(w->device) != hipSuccess) { w->rc = 2; w->err = "hipSetDevice failed"; }
    else w->rc = job(*w);
    {
      std::lock_guard<std::mutex> lk(g->m);
      g->done++;
    }
    g->cv_done.notify_one();
  }
}

// Runs `job` on every worker (each with its device current) and waits; first failure wins.
int run_all(rtgpu_group *g, std::function<int(Worker &)> job) {
  {
    std::lock_guard<std::mutex> lk(g->m);
    g->job = std::move(job);
    g->done = 0;
    g->job_seq++;
  }
  g->cv_job.notify_all();
  {
    std::unique_lock<std::mutex> lk(g->m);
    g->cv_done.wait(lk, [&] { return g->done == (int)g->workers.size(); });
  }
  for (Worker &w : g->workers)
    if (w.rc) {
      g->err = "GPU " + std::to_string(w.device) + " (rank " + std::to_string(w.index) + "): " + w.err;
      return w.rc;
    }
  return 0;
}

#define W_HIP(call)                                                        \
  do {                                                                     \
    hipError_t e_ = (call);                                                \
    if (e_ != hipSuccess) { w.err = std::string(#call ": ") + hipGetErrorString(e_); return 2; } \
  } while (0)
#define W_RT(call)                                                         \
  do {                                                                     \
    int r_ = (call);                                                       \
    if (r_) { w.err = rtgpu_last_error(w.ctx); return r_; }                \
  } while (0)

int fail(rtgpu_group *g, int rc, const std::string &msg) {
  g->err = msg;
  return rc;
}

}  // namespace

extern "C" const char *rtgpu_group_last_error(const rtgpu_group *g) { return g ? g->err.c_str() : g_group_error.c_str(); }
extern "C" int rtgpu_group_size(const rtgpu_group *g) { return g ? (int)g->workers.size() : 0; }
extern "C" int rtgpu_group_comm_ranks(const rtgpu_group *g) {
  if (!g || g->workers.empty() || !g->workers[0].comm) return 0;
  return (int)g->workers.size();
}

extern "C" void rtgpu_group_destroy(rtgpu_group *g) {
  if (!g) return;
  const bool broken = g->broken;  // aborted communicators: their streams may hold work that never completes
  if (!g->workers.empty() && g->workers[0].thread.joinable()) {
    if (!broken)
      run_all(g, [](Worker &w) {  // everything queued on this GPU has to be over before its buffers go
        (void)hipDeviceSynchronize();
        return 0;
      });
    {
      std::lock_guard<std::mutex> lk(g->m);
      g->quit = true;
    }
    g->cv_job.notify_all();
    for (Worker &w : g->workers)
      if (w.thread.joinable()) w.thread.join();
  }
  for (Worker &w : g->workers) {
    if (!w.ctx) continue;  // never set up (rtgpu_group_create failed before it)
    (void)hipSetDevice(w.device);
    if (broken) {
      // the trace stream's work is finite; the exchange stream is left alone (abort_comms ended its
      // collectives, and buffers it might still name are deliberately leaked rather than freed under it)
      if (w.trace_stream) (void)hipStreamSynchronize(w.trace_stream);
      rtgpu_internal_abandon(w.ctx);
      continue;
    }
    (void)hipDeviceSynchronize();
    if (w.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(w.comm);
    w.compact.release();
    for (int b = 0; b < 2; b++) {
      w.send[b].release(); w.gathered[b].release(); w.image[b].release(); w.rgba[b].release();
      if (w.ev_traced[b]) (void)hipEventDestroy(w.ev_traced[b]);
      if (w.ev_sent[b]) (void)hipEventDestroy(w.ev_sent[b]);
      if (w.ev_frame[b]) (void)hipEventDestroy(w.ev_frame[b]);
      if (w.h_counters[b]) (void)hipHostFree(w.h_counters[b]);
    }
    if (w.trace_stream) (void)hipStreamDestroy(w.trace_stream);
    if (w.comm_stream) (void)hipStreamDestroy(w.comm_stream);
    rtgpu_destroy(w.ctx);
  }
  (void)hipGetLastError();  // nothing of a failed set-up may surface later as some launch's error
  delete g;
}

extern "C" int rtgpu_group_create(rtgpu_group **out, const int *devices, int n) {
  if (!out) return 1;
  *out = nullptr;
  if (!devices || n < 1 || n > 64) { g_group_error = "rtgpu_group_create: 1..64 devices"; return 1; }
  rtgpu_group *g = new rtgpu_group();
  g->workers.resize((size_t)n);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++)
      if (devices[i] == devices[j]) g->use_rccl = false;
  auto bail = [&](int rc, const std::string &msg) {
    g_group_error = msg;
    rtgpu_group_destroy(g);
    return rc;
  };
  for (int i = 0; i < n; i++) {
    Worker &w = g->workers[(size_t)i];
    w.index = i;
    w.device = devices[i];
    if (rtgpu_create(&w.ctx, devices[i])) return bail(2, std::string("rtgpu_group_create: ") + rtgpu_last_error(nullptr));
    hipError_t e = hipSetDevice(devices[i]);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&w.trace_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&w.comm_stream, hipStreamNonBlocking);
    for (int b = 0; b < 2 && e == hipSuccess; b++) {
      e = hipEventCreateWithFlags(&w.ev_traced[b], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&w.ev_sent[b], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&w.ev_frame[b], hipEventDisableTiming);
      if (e == hipSuccess) e = hipHostMalloc((void **)&w.h_counters[b], 64, hipHostMallocDefault);
    }
    if (e != hipSuccess) return bail(2, std::string("rtgpu_group_create: ") + hipGetErrorString(e));
  }
  if (const char *env = std::getenv("RTGPU_GROUP_DEBUG_FAIL_EXCHANGE")) g->debug_fail_exchange = std::atoi(env);
  if (n == 1) {
    const char *env = std::getenv("RTGPU_GROUP_SELF_RCCL");
    g->self_rccl = env && std::atoi(env) != 0;
  }
  if (g->use_rccl && (n > 1 || g->self_rccl)) {
    std::string lerr;
    if (!g_rccl.load(lerr)) return bail(2, "rtgpu_group_create: " + lerr);
    std::vector<ncclComm_t> comms((size_t)n);
    ncclResult_t r = g_rccl.CommInitAll(comms.data(), n, devices);
    if (r != ncclSuccess) return bail(2, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
    for (int i = 0; i < n; i++) g->workers[(size_t)i].comm = comms[(size_t)i];
  } else if (!g->use_rccl) {
    // peer-to-peer copies: the peers write into the root's gather buffer
    for (int i = 1; i < n; i++)
      if (devices[i] != devices[0]) {
        (void)hipSetDevice(devices[i]);
        hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
          return bail(2, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
        (void)hipGetLastError();
      }
  }
  for (Worker &w : g->workers) w.thread = std::thread(worker_main, g, &w);
  *out = g;
  return 0;
}

extern "C" int rtgpu_group_upload_scene(rtgpu_group *g, const rtgpu_scene_desc *scene, int builder) {
  if (!g) return 1;
  if (g->broken) return fail(g, 2, "the group is unusable after a failed exchange (destroy it and create a new one)");
  if (g->begun != g->ended) return fail(g, 1, "a frame is in flight (call rtgpu_group_render_end first)");
  g->resident = false;
  if (builder == 1)  // every GPU builds its own LBVH from the primitives: same bytes everywhere (collapse4.hip)
    return run_all(g, [scene](Worker &w) {
      W_RT(rtgpu_upload_scene_lbvh(w.ctx, scene));
      return 0;
    });
  if (builder != 0) return fail(g, 1, "builder must be 0 (SAH on the host) or 1 (LBVH on the GPUs)");
  // one host build, N uploads in parallel
  rtgpu_flat_scene fs;
  std::memset(&fs, 0, sizeof(fs));
  if (rtgpu::flatten_scene(scene, &fs, g->err)) return 1;
  int rc = run_all(g, [&fs](Worker &w) {
    W_RT(rtgpu_internal_upload_flat(w.ctx, &fs, 0));
    return 0;
  });
  rtgpu::free_flat_scene(&fs);
  return rc;
}

namespace {
// An exchange failed on some GPU while others may already have queued their half of it: end every
// communicator's pending work (ncclCommAbort makes in-flight RCCL kernels exit) and leave the group
// usable for nothing but rtgpu_group_destroy, which then does not wait for the exchange streams.
int break_group(rtgpu_group *g, int rc, const std::string &msg) {
  if (g_rccl.CommAbort)
    for (Worker &w : g->workers)
      if (w.comm) {
        (void)hipSetDevice(w.device);
        (void)g_rccl.CommAbort(w.comm);
        w.comm = nullptr;
      }
  g->broken = true;
  g->resident = false;
  g->err = msg + " (the group is unusable now: destroy it and create a new one)";
  return rc;
}
}  // namespace

extern "C" int rtgpu_group_render_begin(rtgpu_group *g, const rtgpu_camera *cam, const rtgpu_settings *s) {
  if (!g) return 1;
  if (g->broken) return fail(g, 2, "the group is unusable after a failed exchange (destroy it and create a new one)");
  if (!cam || !s) return fail(g, 1, "camera or settings is NULL");
  if (g->begun - g->ended >= 2) return fail(g, 1, "two frames are in flight already (call rtgpu_group_render_end)");
  const int n = (int)g->workers.size();
  const int b = (int)(g->begun & 1);
  const uint64_t n_pad = rtgpu_compact_pixels(s->width, s->height, s->tile_size, 0, (uint32_t)n);  // rank 0 owns the most tiles
  if (n_pad == 0) return fail(g, 1, "bad frame or tile size");
  const uint64_t n_px = (uint64_t)s->width * s->height;
  const bool fresh = !(s->flags & RTGPU_FLAG_ACCUMULATE) || !g->resident || g->res_w != s->width || g->res_h != s->height ||
                     g->res_tile != s->tile_size;
  g->resident = false;
  rtgpu_group *gp = g;
  Worker *root = &g->workers[0];
  // phase 1a, every GPU: buffers and the trace of its tiles.  Nothing touches an exchange stream yet, so a failure
  // here (out of memory, a refused setting) leaves no half of an exchange queued anywhere: the frame is simply not begun.
  int rc = run_all(g, [=](Worker &w) {
    W_HIP(w.compact.want(n_pad * 16));
    W_HIP(w.send[b].want(n_pad * 16));
    if (w.index == 0) {
      W_HIP(w.gathered[b].want((uint64_t)n * n_pad * 16));
      W_HIP(w.image[b].want(n_px * 16));
      W_HIP(w.rgba[b].want(n_px * 4));
    }
    rtgpu_settings st = *s;
    st.rank = (uint32_t)w.index;
    st.world_size = (uint32_t)n;
    st.flags |= RTGPU_FLAG_COMPACT;
    if (fresh) {
      st.flags &= ~(uint32_t)RTGPU_FLAG_ACCUMULATE;
      W_HIP(hipMemsetAsync(w.compact.p, 0, n_pad * 16, w.trace_stream));  // padding and out-of-image slots stay 0
    }
    W_RT(rtgpu_render_device(w.ctx, cam, &st, w.compact.p, nullptr, w.trace_stream, nullptr));
    uint64_t *counters = nullptr;
    W_RT(rtgpu_internal_counters(w.ctx, &counters));
    W_HIP(hipMemcpyAsync(w.h_counters[b], counters, 64, hipMemcpyDeviceToHost, w.trace_stream));
    // the exchange reads a copy, so the next frame may overwrite the tile sums while this one is on the wire
    W_HIP(hipStreamWaitEvent(w.trace_stream, w.ev_sent[b], 0));
    W_HIP(hipMemcpyAsync(w.send[b].p, w.compact.p, n_pad * 16, hipMemcpyDeviceToDevice, w.trace_stream));
    W_HIP(hipEventRecord(w.ev_traced[b], w.trace_stream));
    return 0;
  });
  if (rc) return rc;  // (a frame with RTGPU_FLAG_ACCUMULATE after this starts a fresh sum: resident is false)
  // phase 1b, every GPU: the one exchange step of the frame - every peer straight to the root.  Reached only when
  // every GPU has traced, so each ncclRecv the root queues has its ncclSend.
  rc = run_all(g, [=](Worker &w) {
    if (w.index == gp->debug_fail_exchange) { w.err = "exchange failure injected (RTGPU_GROUP_DEBUG_FAIL_EXCHANGE)"; return 2; }
    W_HIP(hipStreamWaitEvent(w.comm_stream, w.ev_traced[b], 0));
    const size_t count = (size_t)n_pad * 4;  // floats
    const bool self = gp->self_rccl && w.comm;
    if (w.index == 0 && !self)
      W_HIP(hipMemcpyAsync(w.gathered[b].p, w.send[b].p, n_pad * 16, hipMemcpyDeviceToDevice, w.comm_stream));
    if ((n > 1 && gp->use_rccl) || self) {
      ncclResult_t r = g_rccl.GroupStart();
      if (r != ncclSuccess) { w.err = std::string("ncclGroupStart: ") + g_rccl.GetErrorString(r); return 2; }
      // a failed call inside the group still meets its ncclGroupEnd: the thread's RCCL group must not stay open
      if (w.index == 0) {
        for (int q = 1; q < n && r == ncclSuccess; q++)
          r = g_rccl.Recv((char *)w.gathered[b].p + (size_t)q * n_pad * 16, count, ncclFloat, q, w.comm, w.comm_stream);
        if (self && r == ncclSuccess) {  // one-rank communicator: the root's own part travels through RCCL too
          r = g_rccl.Send(w.send[b].p, count, ncclFloat, 0, w.comm, w.comm_stream);
          if (r == ncclSuccess) r = g_rccl.Recv(w.gathered[b].p, count, ncclFloat, 0, w.comm, w.comm_stream);
        }
      } else {
        r = g_rccl.Send(w.send[b].p, count, ncclFloat, 0, w.comm, w.comm_stream);
      }
      const ncclResult_t r_end = g_rccl.GroupEnd();
      if (r != ncclSuccess) { w.err = std::string("ncclSend / ncclRecv: ") + g_rccl.GetErrorString(r); return 2; }
      if (r_end != ncclSuccess) { w.err = std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(r_end); return 2; }
    } else if (n > 1 && w.index != 0) {
      // rehearsal transport (a device listed twice): the peer writes its part into the root's buffer.
      // root->gathered[b] was sized by the root's own job; both jobs run before the caller's phase 2.
      W_HIP(hipStreamWaitEvent(w.comm_stream, root->ev_frame[b], 0));  // the root is done with slot b's last frame
    }
    return 0;
  });
  if (rc) return break_group(g, rc, g->err);
  // phase 2, from the calling thread (the workers are idle again): what needs every GPU's phase 1
  if (n > 1 && !g->use_rccl) {
    for (int r = 1; r < n; r++) {
      Worker &w = g->workers[(size_t)r];
      if (hipSetDevice(w.device) != hipSuccess ||
          hipMemcpyAsync((char *)root->gathered[b].p + (size_t)r * n_pad * 16, w.send[b].p, n_pad * 16, hipMemcpyDeviceToDevice,
                         w.comm_stream) != hipSuccess)
        return break_group(g, 2, "peer copy of a tile buffer failed");
    }
  }
  for (Worker &w : g->workers) {
    hipError_t e = hipSetDevice(w.device);
    if (e == hipSuccess) e = hipEventRecord(w.ev_sent[b], w.comm_stream);
    if (e != hipSuccess) return break_group(g, 2, std::string("hipEventRecord: ") + hipGetErrorString(e));
  }
  {
    hipError_t e = hipSetDevice(root->device);
    if (e == hipSuccess && n > 1 && !g->use_rccl)
      for (int r = 1; r < n && e == hipSuccess; r++) e = hipStreamWaitEvent(root->comm_stream, g->workers[(size_t)r].ev_sent[b], 0);
    if (e != hipSuccess) return break_group(g, 2, std::string("group exchange: ") + hipGetErrorString(e));
    if (rtgpu_untile_device(root->ctx, root->gathered[b].p, root->image[b].p, s->width, s->height, s->tile_size, (uint32_t)n, 16,
                            root->comm_stream) ||
        rtgpu_resolve_rgba8_device(root->ctx, root->image[b].p, root->rgba[b].p, n_px, root->comm_stream))
      return break_group(g, 2, rtgpu_last_error(root->ctx));
    e = hipEventRecord(root->ev_frame[b], root->comm_stream);
    if (e != hipSuccess) return break_group(g, 2, std::string("hipEventRecord: ") + hipGetErrorString(e));
  }
  g->frames[b].width = s->width;
  g->frames[b].height = s->height;
  g->frames[b].n_pad = n_pad;
  g->begun++;
  g->resident = true;
  g->res_w = s->width; g->res_h = s->height; g->res_tile = s->tile_size;
  return 0;
}

extern "C" int rtgpu_group_render_end(rtgpu_group *g, float *accum_rgba, uint32_t *rgba8, rtgpu_stats *stats) {
  if (!g) return 1;
  if (g->broken) return fail(g, 2, "the group is unusable after a failed exchange (destroy it and create a new one)");
  if (g->begun == g->ended) return fail(g, 1, "no frame is in flight (call rtgpu_group_render_begin first)");
  const int b = (int)(g->ended & 1);
  g->ended++;
  Worker &root = g->workers[0];
  const uint64_t n_px = (uint64_t)g->frames[b].width * g->frames[b].height;
  hipError_t e = hipSetDevice(root.device);
  if (e == hipSuccess) e = hipEventSynchronize(root.ev_frame[b]);
  if (e == hipSuccess && accum_rgba) e = hipMemcpy(accum_rgba, root.image[b].p, n_px * 16, hipMemcpyDeviceToHost);
  if (e == hipSuccess && rgba8) e = hipMemcpy(rgba8, root.rgba[b].p, n_px * 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail(g, 2, std::string("rtgpu_group_render_end: ") + hipGetErrorString(e));
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    for (Worker &w : g->workers) {
      // the frame event on the root follows every GPU's exchange, which follows its counters' copy
      (void)hipSetDevice(w.device);
      if ((e = hipEventSynchronize(w.ev_traced[b])) != hipSuccess) return fail(g, 2, hipGetErrorString(e));
      const unsigned long long *c = w.h_counters[b];
      stats->paths += c[1]; stats->rays += c[2]; stats->nodes += c[3];
      stats->tri_tests += c[4]; stats->sphere_tests += c[5]; stats->scatters += c[6];
    }
  }
  return 0;
}

extern "C" int rtgpu_group_render(rtgpu_group *g, const rtgpu_camera *cam, const rtgpu_settings *s, float *accum_rgba,
                                  uint32_t *rgba8, rtgpu_stats *stats) {
  if (!g) return 1;
  if (g->begun != g->ended) return fail(g, 1, "frames are in flight (end them before the blocking call)");
  int rc = rtgpu_group_render_begin(g, cam, s);
  if (rc) return rc;
  return rtgpu_group_render_end(g, accum_rgba, rgba8, stats);
}

extern "C" int rtgpu_group_kernel_ms(rtgpu_group *g, int index, float *ms) {
  if (!g || !ms || index < 0 || index >= (int)g->workers.size()) return 1;
  if (rtgpu_last_kernel_ms(g->workers[(size_t)index].ctx, ms)) return fail(g, 1, rtgpu_last_error(g->workers[(size_t)index].ctx));
  return 0;
}
