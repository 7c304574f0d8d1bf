/*
 * rtgpu.h — C ABI of the MI355X trace-loop library (librtgpu.so).
 *
 * The reference snapshot (Condzi/rt @ v1) has NO renderer, no plugin/FFI
 * interface and no render entry point.  The seam this
 * library fills is the gap in the reference frame loop between
 * win32_message_loop() and gfx_render() (code/first.cpp:78-88).  Each entry
 * point below therefore cites the reference convention it follows rather than
 * a reference function it replaces.
 *
 * Conventions taken from the reference:
 *   - POD structs of f32/u32 only, designated-initialiser friendly, no STL,
 *     no virtuals (code/first.hpp:16-37, scripts/build_internal.bat:4).
 *   - Vectors are 3 consecutive f32 (rt::Vec3, code/math/vec.hxx:18-28);
 *     accumulation pixels are 4 consecutive f32 (rt::Vec4, vec.hxx:30-48).
 *   - Screen space is y-up, origin bottom-left (code/math/mat.cxx:104-129):
 *     pixel (x, y) lives at index y*width + x and row 0 is the BOTTOM row.
 *   - 8-bit colours are packed 0xRRGGBBAA (code/gfx/gfx.hpp:4-6,
 *     run_tree/data/shaders/shaders.hlsl:18-25).
 *   - The C layer never exits; it returns non-zero and keeps a message
 *     (rtgpu_last_error).  The C++ facade (rt_amd/host/rt.hpp) turns that into
 *     the reference's fatal errf() convention (code/first.cpp:27-43).
 *   - Caller owns host memory passed in, the library copies; device memory is
 *     owned by the context (arena ownership rule, code/base/memory.cxx:23-50).
 *
 * All functions are thread-compatible per context (the reference is single
 * threaded: code/gfx/gfx.cpp:36); calls on one context must not overlap.
 */
#ifndef RTGPU_H
#define RTGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTGPU_ABI_VERSION 5u

/* ---- scene description (host side, caller-owned) ------------------------ */

enum {
  RTGPU_MAT_LAMBERTIAN = 0u, /* param unused                                  */
  RTGPU_MAT_METAL      = 1u, /* param = fuzz in [0,1]                         */
  RTGPU_MAT_DIELECTRIC = 2u, /* param = index of refraction                   */
  RTGPU_MAT_EMISSIVE   = 3u  /* a light: emits albedo * param from both faces
                                and ends the path (nothing is scattered)       */
};

/* Material semantics follow the literature the reference cites
 * (notes/materials.md:3-5 -> "Ray Tracing in One Weekend"): a tagged POD, not a
 * class hierarchy.  32 bytes.  The emissive type is the
 * diffuse light of the same series ("The Next Week"); "Lights" is an item of the
 * reference's plan (notes/plan.md:16). */
typedef struct rtgpu_material {
  float    albedo[3];
  float    param;
  uint32_t type;
  uint32_t _pad[3];
} rtgpu_material;

typedef struct rtgpu_sphere {     /* 20 bytes */
  float    center[3];
  float    radius;
  uint32_t material;
} rtgpu_sphere;

typedef struct rtgpu_triangle {   /* 40 bytes */
  float    v0[3], v1[3], v2[3];
  uint32_t material;
} rtgpu_triangle;

/* Primitive ids reported by the renderer: sphere i -> i,
 * triangle j -> n_spheres + j, miss -> RTGPU_NO_HIT. */
#define RTGPU_NO_HIT 0xFFFFFFFFu

typedef struct rtgpu_scene_desc {
  const rtgpu_sphere   *spheres;   uint64_t n_spheres;
  const rtgpu_triangle *triangles; uint64_t n_triangles;
  const rtgpu_material *materials; uint64_t n_materials;
} rtgpu_scene_desc;

/* ---- camera -------------------------------------------------------------- */

/* Thin-lens camera already reduced to the vectors the per-ray code consumes.
 * Fill it with rtgpu_camera_look_at() so every caller (C++, Python, oracle
 * tests) gets the same f32 values.  80 bytes. */
typedef struct rtgpu_camera {
  float origin[3];
  float lower_left[3];
  float horizontal[3];
  float vertical[3];
  float u[3];          /* lens basis, unit */
  float v[3];
  float lens_radius;   /* 0 => pinhole, no lens RNG draws */
  float _pad;
} rtgpu_camera;

/* ---- render settings ----------------------------------------------------- */

enum {
  RTGPU_FLAG_ACCUMULATE = 1u, /* out = in + frame (progressive), else out = frame */
  RTGPU_FLAG_COMPACT    = 2u, /* write this rank's tiles as a compact tile list   */
  RTGPU_FLAG_STATS      = 4u, /* run the instrumented variant (node/prim counters) */
  RTGPU_FLAG_PROFILE    = 8u  /* STATS + per-wave trip counters (rtgpu_profile_counters) */
};

typedef struct rtgpu_settings {
  uint32_t width, height;
  uint32_t spp;            /* samples per pixel traced by this call            */
  uint32_t sample_offset;  /* index of the first sample (progressive frames)   */
  uint32_t max_depth;      /* max segments per path                            */
  uint32_t seed;
  float    tmin;           /* self-intersection offset, e.g. 1e-3              */
  float    sky_bottom[3];  /* miss shader: bottom*(1-t) + top*t, t=0.5(d.y+1)  */
  float    sky_top[3];
  uint32_t tile_size;      /* multi-GPU tile edge (pixels), power of two >= 8  */
  uint32_t rank;           /* this context renders tiles t with t % world == rank */
  uint32_t world_size;
  uint32_t flags;
  uint32_t chunk_samples;  /* samples per work item, >= 1: a pixel's frame sum is
                            * (((C0 + C1) + C2) ...) where chunk Ck sums its samples
                            * [k*chunk_samples, min((k+1)*chunk_samples, spp)) in
                            * order.  Finer chunks = shorter tail of the persistent
                            * kernel; the value is part of the arithmetic contract.
                            * rtgpu_default_settings: 2 below 16 spp, 4 up to 256
                            * spp, else ceil(spp / 64); n chunks > 1 cost n x 16 B per pixel
                            * of context memory (the chunk planes), 65 x 16 B at most: a
                            * frame of more than 64 chunks is traced in rounds of 64 */
} rtgpu_settings;

typedef struct rtgpu_stats {
  uint64_t paths;          /* (pixel, sample) pairs traced                     */
  uint64_t rays;           /* segments traced (primary + every bounce)         */
  uint64_t nodes;          /* 64-byte BVH nodes fetched      (STATS flag only) */
  uint64_t tri_tests;      /* ray-triangle tests             (STATS flag only) */
  uint64_t sphere_tests;   /* ray-sphere tests               (STATS flag only) */
  uint64_t scatters;       /* material fetch + scatter evals (STATS flag only) */
} rtgpu_stats;

/* Flattened scene as it lives in HBM (and in a host mirror).  Exposed so that
 * tests can hand the *same* buffers to the CPU oracle, and so a caller can
 * serialise them.  Pointers are owned by the context. */
typedef struct rtgpu_flat_scene {
  const void    *nodes;        uint64_t n_nodes;      /* 64 B each  */
  const void    *tris;         uint64_t n_tris;       /* 48 B each, leaf order */
  const void    *spheres;      uint64_t n_spheres;    /* 16 B each, leaf order */
  const uint32_t*sphere_meta;                         /* 2 u32 each: id, material */
  const void    *materials;    uint64_t n_materials;  /* 32 B each  */
  uint32_t       max_depth;    /* deepest leaf (root = 0)           */
  uint32_t       _pad;
  double         build_ms;
  const void    *nodes48;      /* n_nodes packed nodes of 48 B (rtgpu_format.h) when
                                  n_nodes >= RTGPU_NODE48_MIN_NODES, else NULL: the
                                  array the trace loop traverses when nodes4 is
                                  absent (RTGPU_NO_WIDE, or a four-wide stack that
                                  does not fit); a refit refreshes both           */
  const void    *nodes4;       /* n_nodes4 four-wide packed nodes of 96 B, same
                                  threshold; traversed in preference when present  */
  uint64_t       n_nodes4;
  uint32_t       max_stack4;   /* bound on the traversal stack over nodes4         */
  uint32_t       _pad2;
  const uint32_t*nodes4_src;   /* 4 u32 per four-wide node: where child k's box lives
                                  in `nodes` - (node index << 1) | child slot, or
                                  0xFFFFFFFF for a missing child; a refit re-reads
                                  the boxes through it                             */
} rtgpu_flat_scene;

typedef struct rtgpu_ctx rtgpu_ctx;

/* ---- entry points -------------------------------------------------------- */

/* Follows X_init_or_panic (code/gfx/gfx.hpp:21) minus the panic: binds one
 * context to one GPU (one process per GPU is the intended multi-GPU use). */
int  rtgpu_create(rtgpu_ctx **out, int device_index);
void rtgpu_destroy(rtgpu_ctx *ctx);
const char *rtgpu_last_error(const rtgpu_ctx *ctx);  /* ctx may be NULL: last create error */
uint32_t rtgpu_abi_version(void);

/* Builds the BVH on the host, flattens scene + BVH into SoA node/primitive
 * buffers and copies them to HBM. */
int  rtgpu_upload_scene(rtgpu_ctx *ctx, const rtgpu_scene_desc *scene);

/* SURVEY.md §8f row 3 (no reference counterpart): as rtgpu_upload_scene, but the
 * BVH is built on the GPU - Morton codes, one radix sort, Karras' parallel
 * hierarchy (HPG 2012), level-by-level box fit - straight into the same
 * flattened format, one primitive (or two triangles) per leaf.  Builds in milliseconds where the
 * host SAH build takes seconds, for a tree that costs more node visits per ray;
 * meant for scenes that change.  The packed copy and the four-wide nodes of the tree
 * are made on the GPU too; rtgpu_get_flat_scene copies the result back when it is
 * first asked for (build_ms = GPU time of the build). */
int  rtgpu_upload_scene_lbvh(rtgpu_ctx *ctx, const rtgpu_scene_desc *scene);
/* rtgpu_upload_scene_lbvh for primitives that already live in the memory of the
 * context's GPU (a simulation, a skinning pass or a loader wrote them there):
 * arrays of rtgpu_sphere / rtgpu_triangle (either may be NULL with a count of 0),
 * only read; the material table comes from the host.  Nothing else crosses PCIe
 * and the input check runs on the GPU.  Replaces the uploaded scene; counts and
 * topology are free, so this is what a frame loop calls when rtgpu_refit_device's
 * fixed tree no longer fits.  The context keeps the builder's temporaries between
 * such calls (rtgpu_upload_scene* and rtgpu_destroy release them).  No reference
 * counterpart. */
int  rtgpu_build_device(rtgpu_ctx *ctx, const void *d_spheres, uint64_t n_spheres,
                        const void *d_triangles, uint64_t n_triangles,
                        const rtgpu_material *materials, uint64_t n_materials);
/* New positions (and material table) under the UNCHANGED tree of the uploaded
 * scene: `scene` must hold the same number of spheres, triangles and materials,
 * primitive i being the moved / edited primitive i.  Leaf records and every box
 * are recomputed on the GPU (rt_amd/csrc/refit.hip); topology, leaf order and
 * primitive ids stay, so for unchanged positions the tree is byte-identical.  The
 * tree's quality degrades as the geometry moves away from what it was built for:
 * upload again when frames slow down.  No reference counterpart (the reference
 * has no BVH; §8f row 3 names builder + refit).  On a device error
 * the scene is dropped and must be uploaded again. */
int  rtgpu_refit(rtgpu_ctx *ctx, const rtgpu_scene_desc *scene);
/* The same for geometry that already lives in the memory of the context's GPU
 * (arrays of rtgpu_sphere / rtgpu_triangle in the caller's order and with the
 * uploaded counts; either may be NULL with a count of 0): nothing crosses PCIe.
 * Materials stay as uploaded.  The input check of rtgpu_refit runs on the GPU
 * before anything is rewritten; bad input leaves the scene intact. */
int  rtgpu_refit_device(rtgpu_ctx *ctx, const void *d_spheres, uint64_t n_spheres,
                        const void *d_triangles, uint64_t n_triangles);
int  rtgpu_get_flat_scene(const rtgpu_ctx *ctx, rtgpu_flat_scene *out);

/* Host-only variant of the two calls above: build + flatten without a GPU
 * (used by CPU-side tests and tools).  Free with rtgpu_flat_scene_free. */
int  rtgpu_flatten_scene_host(const rtgpu_scene_desc *scene, rtgpu_flat_scene *out);
void rtgpu_flat_scene_free(rtgpu_flat_scene *fs);

/* The trace loop (rows a1, a3-a8).  Blocking.  accum_rgba: host, width*height
 * float4 {sum r, sum g, sum b, sample count}; prim_id (optional): host,
 * width*height u32, primary-hit primitive of sample `sample_offset`.
 * With RTGPU_FLAG_COMPACT the outputs hold only this rank's tiles
 * (rtgpu_compact_pixels() pixels, tile-major). */
int  rtgpu_render(rtgpu_ctx *ctx, const rtgpu_camera *cam, const rtgpu_settings *s,
                  float *accum_rgba, uint32_t *prim_id, rtgpu_stats *stats);

/* The progressive frame loop of an interactive front-end (the reference's loop,
 * code/first.cpp:78-88, presents one image per iteration): trace `spp` more samples and get
 * the RESOLVED frame back.  The float accumulation never leaves the GPU - it lives in the
 * context between calls, owned by it like every device buffer (arena rule of
 * code/base/memory.cxx:23-34: allocated once, kept) - and only 4 bytes per pixel cross PCIe,
 * through a pinned staging buffer with an asynchronous copy.  With RTGPU_FLAG_ACCUMULATE the
 * frame adds to the resident sum (give each call the next sample_offset); without it, after a
 * size change, after rtgpu_resident_reset and after a scene upload, build or refit the sum
 * starts again from this frame.  rgba8: host, width*height 0xRRGGBBAA, row 0 = bottom; may be
 * NULL, then read the frame in place from rtgpu_resolved_frame() (context-owned pinned host
 * memory, valid until the next call on the context).  Whole frames only (no
 * RTGPU_FLAG_COMPACT).  Blocking.  rtgpu_resident_accum copies the float sums out when a
 * caller wants them (rtgpu_render's accum_rgba layout). */
int  rtgpu_render_resolved(rtgpu_ctx *ctx, const rtgpu_camera *cam, const rtgpu_settings *s,
                           uint32_t *rgba8 /* may be NULL */, rtgpu_stats *stats /* may be NULL */);
const uint32_t *rtgpu_resolved_frame(const rtgpu_ctx *ctx);
int  rtgpu_resident_accum(rtgpu_ctx *ctx, float *accum_rgba, uint64_t n_pixels);
void rtgpu_resident_reset(rtgpu_ctx *ctx);

/* Non-blocking form of rtgpu_render, for a frame loop that must keep running
 * while the frame is traced (the reference's loop pumps window messages every
 * iteration, code/first.cpp:78-88, and its author wants tracing to go on while
 * the window is minimised, code/window/wnd_proc.cxx:42-44; SURVEY.md §8b
 * "blocking or begin/poll").  _begin queues the frame on a stream the context
 * owns and returns; _poll sets *done without blocking; _end blocks until the
 * frame is finished and copies it into the buffers given to _begin (which must
 * stay valid until then; with RTGPU_FLAG_ACCUMULATE accum_rgba is also read by
 * _begin).  One frame at a time: while a frame is pending every other call that
 * renders, resolves through host buffers, uploads or refits returns an error.
 * The three calls come from one thread, like everything else on a context. */
int  rtgpu_render_begin(rtgpu_ctx *ctx, const rtgpu_camera *cam, const rtgpu_settings *s,
                        float *accum_rgba, uint32_t *prim_id /* may be NULL */);
int  rtgpu_render_poll(rtgpu_ctx *ctx, int *done);
int  rtgpu_render_end(rtgpu_ctx *ctx, rtgpu_stats *stats /* may be NULL */);

/* As rtgpu_render, outputs in device memory of the context's GPU, launched on `stream`
 * (a hipStream_t, NULL = default stream).  Asynchronous unless `stats` is
 * non-NULL (then it synchronises the stream to read the counters back).
 * Stream rule: every launch of a context uses the context's one set of scratch buffers
 * (work-queue head, counters, overflow stacks, chunk planes).  The library orders launches
 * itself: a call on a stream other than the previous call's first makes that stream wait
 * (hipStreamWaitEvent) for the end of the previous launch, so frames queued on different
 * streams run one after the other, never together.  What stays the caller's job is the
 * order between a frame and the caller's OWN work on the output buffers. */
int  rtgpu_render_device(rtgpu_ctx *ctx, const rtgpu_camera *cam, const rtgpu_settings *s,
                         void *d_accum_rgba, void *d_prim_id, void *stream,
                         rtgpu_stats *stats);

/* The entry point that ALLOCATES NOTHING.  It has no counterpart in the reference (nor has
 * anything else here); it exists for callers that cannot call hipMalloc: code
 * under HIP graph capture, tight serving / frame loops, harnesses that forbid allocation.
 * Same frame as rtgpu_render_device, bit for bit (same kernels); what differs is where the
 * launch keeps its scratch - the work-queue head, the counters and the chunk planes live in a
 * static device array of the library (one copy per GPU, held until the process ends) instead
 * of context buffers that grow on demand.  No hipMalloc, hipFree or other allocating routine
 * is reached from this call; it is asynchronous on `stream` like rtgpu_render_device.
 * Limits, each answered with a non-zero return and a message instead of an allocation:
 *   - chunk planes: min(ceil(spp / chunk_samples), 65) x output pixels x 16 B must fit
 *     rtgpu_static_scratch_bytes() (64 MiB: 1920x1080 in up to 2 chunks, 1280x720 in up to 4;
 *     one chunk - chunk_samples >= spp - needs no planes and fits any frame);
 *   - the traversal stack of the uploaded tree must fit LDS (true for every tree up to depth
 *     30, or 36 four-wide stack entries; deeper trees need the overflow stack of
 *     rtgpu_render_device);
 *   - no RTGPU_FLAG_PROFILE; counters of RTGPU_FLAG_STATS are read with rtgpu_profile_counters.
 * A tree that lives in memory is traversed through its slot tree (rtgpu_format.h), which rtgpu_render_device
 * makes on the first frame after an upload, build or refit and this call cannot; and the kernels of the slot
 * tree keep 24 stack entries per lane in LDS, the rest in the overflow stack this call does not have.  Before
 * such a frame, or for a tree whose stack bound is above 24, it traverses the four-wide nodes the slot tree is
 * made from - the same hits, 0.2 % fewer node visits in the counters, 4-15 % more time.
 * All contexts of a process on one GPU share that GPU's array and it keeps the last call's
 * contents (the call clears what has to start at zero): two calls that OVERLAP on one GPU -
 * two contexts, or one context from two threads - race.  Calls of one context on different
 * streams are ordered by the library as for rtgpu_render_device. */
int  rtgpu_render_device_static(rtgpu_ctx *ctx, const rtgpu_camera *cam, const rtgpu_settings *s,
                                void *d_accum_rgba, void *d_prim_id, void *stream);
uint64_t rtgpu_static_scratch_bytes(void);

/* Duration of the trace kernel of the last render call on this context,
 * measured with HIP events on the stream the kernel ran on.  Synchronises. */
int  rtgpu_last_kernel_ms(rtgpu_ctx *ctx, float *ms);
/* Name of the trace-kernel instantiation that launch used, as a profiler prints it:
 * "trace_kernel<STATS,REFILL,MODE>" (MODE: DESIGN.md §4.1); "" before the first launch.
 * The string lives in the context. */
const char *rtgpu_last_kernel_name(const rtgpu_ctx *ctx);

/* Diagnostics: raw counters of the last launch on this context.  out[0] queue
 * head, [1..6] as rtgpu_stats, and with RTGPU_FLAG_PROFILE [7..15] = per-WAVE
 * execution counts {outer iterations, shade blocks, lanes in shade blocks,
 * camera-ray blocks, node steps, sphere tests, triangle tests, traversal
 * rounds, lanes in traversal rounds}: lane totals / (64 x wave counts) is the
 * SIMD lane utilisation of each section; [16..20] = shader-clock cycles summed
 * over waves spent in {shading, work-item bookkeeping, camera rays, traversal
 * set-up, traversal}; [21], [22] = the node-phase and leaf-phase cycles inside the
 * traversal figure; [23..25] = lanes in node steps, of them served from the LDS copy of the
 * top of a four-wide tree, steps on the slow stack path (RTGPU_WIDE_CACHE=1 experiment);
 * [26], [27] = lanes that still had a ray in traversal, summed over the node steps / leaf
 * steps of every wave; [28], [29] = slot exchanges per wave and lanes that took part
 * (RTGPU_DUAL=1 experiment); [30], [31] zero.  Synchronises the device. */
int  rtgpu_profile_counters(rtgpu_ctx *ctx, uint64_t out[32]);

/* accum (float4 sums) -> 0xRRGGBBAA, gamma 2, like RTIOW's write_color.
 * Packing per run_tree/data/shaders/shaders.hlsl:18-25. Device pointers. */
int  rtgpu_resolve_rgba8_device(rtgpu_ctx *ctx, const void *d_accum_rgba, void *d_rgba8,
                                uint64_t n_pixels, void *stream);
int  rtgpu_resolve_rgba8(rtgpu_ctx *ctx, const float *accum_rgba, uint32_t *rgba8,
                         uint64_t n_pixels);

/* Multi-GPU helpers.  Number of pixels in the compact buffer
 * of `rank`, and the scatter of `world_size` gathered compact buffers
 * (concatenated, each padded to rtgpu_compact_pixels(.., rank 0) pixels) back
 * into a width*height image.  elem_bytes is 16 (accum) or 4 (prim ids). */
uint64_t rtgpu_compact_pixels(uint32_t width, uint32_t height, uint32_t tile_size,
                              uint32_t rank, uint32_t world_size);
int  rtgpu_untile_device(rtgpu_ctx *ctx, const void *d_gathered, void *d_image,
                         uint32_t width, uint32_t height, uint32_t tile_size,
                         uint32_t world_size, uint32_t elem_bytes, void *stream);

/* ---- several GPUs of one node behind one handle ------------------------------
 * No reference counterpart (the reference is one process, one thread, one GPU for display:
 * code/gfx/gfx.cpp:36).  One process, one context and one worker thread per GPU, one RCCL
 * communicator (ncclCommInitAll).  The scene is flattened once on the host and replicated;
 * a frame is cut into tile_size x tile_size tiles, tile t belongs to GPU t mod N; every GPU
 * traces its tiles into a compact buffer; ONE exchange step - ncclGroupStart, N-1 ncclRecv on
 * the root (GPU 0 of the list) and one ncclSend per peer, each peer straight to the root over
 * its own xGMI link - then the root de-interleaves (untile kernel) and resolves.  Frames are
 * pipelined two deep: _begin queues a frame on every GPU and returns, _end waits for the
 * OLDEST queued frame and copies it out, so the exchange of frame k runs beside the trace of
 * frame k+1 when the caller begins k+1 before ending k.  The image is bit-identical to the
 * one-GPU frame for any N and tile size (the RNG is keyed on the global pixel).
 * The worker threads never call back into the caller and touch nothing but HIP / RCCL: the
 * single-threaded D3D11 device of the reference (code/gfx/gfx.cpp:36) stays with the thread
 * that calls these functions, and all calls on a group come from that one thread.
 * A device may appear more than once in `devices` (rehearsal of N ranks on a one-GPU box):
 * RCCL does not accept that, and the exchange is then done with peer-to-peer copies. */
typedef struct rtgpu_group rtgpu_group;
int  rtgpu_group_create(rtgpu_group **out, const int *devices, int n_devices);
void rtgpu_group_destroy(rtgpu_group *g);
const char *rtgpu_group_last_error(const rtgpu_group *g);   /* g may be NULL: last create error */
int  rtgpu_group_size(const rtgpu_group *g);
/* Ranks of the RCCL communicator the group exchanges tiles over: n_devices for distinct devices,
 * 0 when a device is listed twice (rehearsal: device copies, no RCCL), 1 for a one-device group
 * created with RTGPU_GROUP_SELF_RCCL=1 (its buffer then goes through a grouped ncclSend / ncclRecv to
 * itself: the part of the wire a one-GPU box can run), else 0 for a one-device group. */
int  rtgpu_group_comm_ranks(const rtgpu_group *g);
/* builder: 0 = binned SAH on the host (built once, uploaded to every GPU), 1 = LBVH, built by
 * every GPU for itself from the uploaded primitives (same bytes on every GPU). */
int  rtgpu_group_upload_scene(rtgpu_group *g, const rtgpu_scene_desc *scene, int builder);
/* settings: rank, world_size and RTGPU_FLAG_COMPACT are set by the group; tile_size is used
 * as given.  RTGPU_FLAG_ACCUMULATE adds to the frame the group holds from the previous call
 * of the same size (resident per-GPU tile sums, as rtgpu_render_resolved). */
int  rtgpu_group_render_begin(rtgpu_group *g, const rtgpu_camera *cam, const rtgpu_settings *s);
/* Outputs of the oldest frame begun: accum_rgba (width*height float4, host) and rgba8
 * (width*height 0xRRGGBBAA, host), either may be NULL; stats summed over the GPUs. */
int  rtgpu_group_render_end(rtgpu_group *g, float *accum_rgba, uint32_t *rgba8, rtgpu_stats *stats);
/* _begin + _end */
int  rtgpu_group_render(rtgpu_group *g, const rtgpu_camera *cam, const rtgpu_settings *s,
                        float *accum_rgba, uint32_t *rgba8, rtgpu_stats *stats);
/* After a failure in the exchange step (an RCCL call failed on some GPU while others had queued their
 * half) the communicators are aborted (ncclCommAbort) and every call above answers with an error until
 * rtgpu_group_destroy, which then does not wait for the exchange streams.  A frame every GPU refuses
 * BEFORE its trace (bad settings, out of memory) queues nothing and leaves the group usable.
 * Trace-kernel time of the last ended frame on GPU `index` of the list (ms). */
int  rtgpu_group_kernel_ms(rtgpu_group *g, int index, float *ms);

/* Camera helper: RTIOW look-at thin-lens camera, computed in f64 and rounded
 * once to f32 so all callers agree bit-for-bit.  Row-vector convention of
 * code/math/mat.cxx:12-20 is not involved: the camera is stored as vectors. */
void rtgpu_camera_look_at(rtgpu_camera *cam, const float from[3], const float at[3],
                          const float up[3], float vfov_deg, float aspect,
                          float aperture, float focus_dist);

/* The same camera from a camera-to-world rt::Mat4x4 (code/math/mat.hxx:2-4, 16 floats row
 * by row) in the reference's ROW-VECTOR convention (code/math/mat.cxx:12-20: world = p * M):
 * rows 0..2 = the camera's right, up and back axes in world space, row 3 = the eye (where
 * translate2 puts a translation, mat.cxx:93-102).  The camera looks down its -z axis. */
void rtgpu_camera_from_matrix(rtgpu_camera *cam, const float m[16], float vfov_deg, float aspect,
                              float aperture, float focus_dist);

void rtgpu_default_settings(rtgpu_settings *s, uint32_t width, uint32_t height,
                            uint32_t spp, uint32_t max_depth);

/* Procedural scenes of SURVEY.md §8d (no assets exist offline).
 * kind: 0 = S0 "default" (RTIOW cover + quad), 1 = S1 "mesh" (displaced
 * icosphere, `param` = subdivision level, 8 => 1,310,720 tris),
 * 2 = S2 "soup" (8 displaced icospheres, mixed materials, `param` = level).
 * The arrays are malloc'ed; free with rtgpu_scene_free. */
int  rtgpu_scene_builtin(uint32_t kind, uint32_t seed, uint32_t param,
                         rtgpu_scene_desc *out, rtgpu_camera *cam_out, float aspect);
void rtgpu_scene_free(rtgpu_scene_desc *scene);

/* ---- data formats either side of the loop --------
 * No reference counterpart ("3D models" is a plan item, notes/plan.md:21);
 * whole-file I/O in the manner of code/os/filesystem.hxx:4-9.
 *
 * Wavefront OBJ text -> triangles.  `v x y z`, `f` with v, v/vt, v//vn or
 * v/vt/vn tokens, 1-based or negative (relative) indices, polygons fanned
 * around their first vertex, `usemtl NAME`; other records are skipped.
 * materials[0] = grey lambertian (faces before any usemtl), materials[1..] =
 * one grey lambertian per usemtl name in order of first appearance; the
 * arrays are malloc'ed and the caller's to edit; free with rtgpu_scene_free.
 * Errors (bad index, non-finite vertex, short face, I/O) -> non-zero +
 * rtgpu_last_error(NULL). */
int  rtgpu_scene_parse_obj(const char *text, uint64_t n_bytes, rtgpu_scene_desc *out);
int  rtgpu_scene_load_obj(const char *path, rtgpu_scene_desc *out);

/* Resolved frame (0xRRGGBBAA, row 0 = bottom, as rtgpu_resolve_rgba8 writes it)
 * -> binary PPM (P6), rows flipped to top-down, alpha dropped. */
int  rtgpu_write_ppm(const char *path, const uint32_t *rgba8, uint32_t width, uint32_t height);
/* The same image as an 8-bit RGBA PNG (rows top-down, alpha kept; stored deflate
 * blocks, so no compression library is involved). */
int  rtgpu_write_png(const char *path, const uint32_t *rgba8, uint32_t width, uint32_t height);

#ifdef __cplusplus
}
#endif
#endif /* RTGPU_H */
