#!/usr/bin/env python3
"""bench.py — the headline metric of BASELINE.json on MI355X.

A "step" is one pass of the trace loop over one frame.

  --config c2 (default at N=1)  BASELINE.json configs[1]: S0, 1920x1080, 16 spp, depth 8 on one
                                MI355X.  With N>1 ranks the per-GPU work is held fixed (16*N spp
                                per frame): weak scaling.
  --config c4 (default at N>1)  BASELINE.json configs[3]: S0, 3840x2160, 256 spp in total, depth 8,
                                the frame split into 16x16 tiles interleaved over the N ranks:
                                strong scaling.  (At N=1 it is the same frame on one GPU.)
  --config c5                   BASELINE.json configs[4]: S2 level 8 (10,485,760 triangles), 1920x1080,
                                16 spp per frame, depth 8, split over the N ranks: strong scaling.

Every rank traces its tiles from a replicated scene into a compact tile buffer, the buffers are
gathered to rank 0 (each peer straight to the root) and rank 0 de-interleaves them; the gather
of frame k overlaps the trace of frame k+1.  Two hosts for that path:
  * launched by torch.distributed.run (WORLD_SIZE set; what the driver does for N > 1): one
    process per GPU, `dist.gather` over backend nccl = RCCL;
  * plain `python3 bench.py --gpus N` (no launcher): --transport group (default) renders through
    the native rtgpu_group_* in THIS process (one worker thread per GPU, ncclCommInitAll,
    ncclSend / ncclRecv - SURVEY.md §8e as written); --transport procs starts N fresh child
    processes under torch.distributed.run and relays rank 0's line (the parent never touches a GPU).

Prints ONE JSON line (rank 0).  `value` = rays (segments: primary + every bounce, counted by the
kernel) traced by all ranks per second of wall time, inputs and outputs resident in HBM.

`roofline` names the bound SURVEY.md §8d gives the workload - FP32 VALU for S0, whose 29 KB
tree is served from LDS - as counted flops / kernel time against the 157.3 TFLOP/s vector peak;
`roofline.valu_issue` is the wave-instruction rate against the issue ceiling (nominal: one
wave64 VALU instruction per SIMD per 2 cycles at 2.4 GHz; measured: tools/ubench/valu_issue.hip,
profiles/r02/ubench_valu_issue.txt), `roofline.hbm` the memory side (measured traffic from the
committed PMC passes of the same workload; algorithmic bytes for information).  At N=1 the
`also` array carries C3 (S1, 1.31 M triangles, 1080p, 64 spp: HBM/L2-bound by §8d), C4's frame and
C5's frame (S2 level 8, 10.5 M triangles, 1080p, 16 spp) on one GPU, measured outside the timed
region, each beside its own `cpu_baseline`.  `cpu_baseline` is the CPU oracle (oracle/rt_oracle.c)
on a bounded sample of the same workload: the same scene, frame size and depth at a reduced
sample count - the oracle's rate is flat in spp (every sample is an independent path).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak (spec)
VALU_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: peak FP32 vector = 1024 SIMDs x 2.4 GHz x 32 lanes x 2 flop
VALU_ISSUE_PEAK_GINST = 1228.8  # the same peak in wave64 instructions: 1024 SIMDs x 2.4 GHz / 2 cycles (SIMD-32)


def algorithmic_bytes(st, n_pixels, node_bytes):
    """SURVEY.md §8d.  node_bytes: 64 (BVH2 nodes - the C2 tree, read from its LDS copy - or the node slots of the
    slot tree), 48 (packed BVH2 nodes) or 96 (four-wide nodes) - what one node step of the launched mode reads."""
    return node_bytes * st["nodes"] + 48 * st["tri_tests"] + 16 * st["sphere_tests"] + 32 * st["scatters"] + 16 * n_pixels


def algorithmic_flops(st):
    return 34 * st["nodes"] + 47 * st["tri_tests"] + 21 * st["sphere_tests"] + 60 * st["scatters"]


def committed_pmc(name, width, height, spp, depth):
    """PMC figures of THIS workload from the committed summary of the last profile
    (profiles/traffic.json, written by tools/update_profiles.py from rocprofv3 --pmc passes:
    FETCH_SIZE / WRITE_SIZE in runs of their own, KiB -> B, gfx950 correction calibrated with
    tools/ubench/fetch_calib.hip as MI355X_MICROARCH.md prescribes).  Counters cannot be read
    from inside this process; None when the summary is of another workload."""
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return None, {}
    wl = doc.get("workloads", {}).get(name)
    if not wl or [wl.get(k) for k in ("width", "height", "spp", "depth")] != [width, height, spp, depth]:
        return None, doc.get("ubench", {})
    return wl, doc.get("ubench", {})


def kernel_mode(kernel):
    """MODE of a 'trace_kernel<STATS,REFILL,MODE>' name, or None."""
    try:
        return int(kernel.rstrip(">").split(",")[-1]) if kernel.startswith("trace_kernel<") else None
    except ValueError:
        return None


def node_bytes_of(ctx):
    """Bytes one node step of the LAST launch read (call after a frame)."""
    mode = kernel_mode(ctx.kernel_name())
    if mode in (11, 12):          # the slot tree: 64-byte node slots
        return 64
    c = ctx.flat_scene().c
    return 96 if c.nodes4 else 48 if c.nodes48 else 64


def roofline_block(bound, counts, n_px, node_bytes, k_ms, kernel, pmc, ub, in_memory=None):
    """The roofline object of one launch: `bound` as SURVEY.md §8d names it for the workload.
    in_memory: the tree is fetched through the vector memory pipeline (default: by the kernel's mode)."""
    if in_memory is None:
        in_memory = kernel_mode(kernel) not in (None, 2, 7, 8)
    sec = k_ms * 1e-3
    abytes = algorithmic_bytes(counts, n_px, node_bytes)
    aflops = algorithmic_flops(counts)
    traffic = int(pmc["traffic_bytes_per_launch"]) if pmc else None
    valu = {"achieved_tflops": round(aflops / sec / 1e12, 3), "peak_tflops": VALU_PEAK_TFLOPS,
            "frac": round(aflops / sec / 1e12 / VALU_PEAK_TFLOPS, 4), "flops_per_ray": round(aflops / counts["rays"], 1)}
    hbm = {"measured_gbs": round(traffic / sec / 1e9, 1) if traffic else None, "peak_gbs": HBM_PEAK_GBS,
           "frac": round(traffic / sec / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
           "achievable_peak_gbs": ub.get("hbm_read_gbs"),
           "algorithmic_gbs": round(abytes / sec / 1e9, 1), "algorithmic_bytes_per_launch": int(abytes),
           "bytes_per_ray": round(abytes / counts["rays"], 1),
           "note": "measured = FETCH_SIZE + WRITE_SIZE of the committed PMC passes of this workload / this run's kernel "
                   "time; algorithmic = every node / primitive / material fetch a ray makes, most of them served by "
                   "LDS, L1 and L2 - informational, never a fraction of HBM"}
    vmem = None
    caps = ub.get("vmem_records_per_cycle_per_cu") or {}
    if in_memory and caps.get("node96_l2") and caps.get("node64_l2") and caps.get("tri48_l2"):
        # Trees in memory: what the vector memory pipeline (TA / L1) can deliver for THIS access pattern - every lane its own
        # record, 16 bytes per lane and load.  tools/ubench/node_fetch.hip measures records per cycle per CU by record shape;
        # the floor of the launch is its node visits and triangle tests at those rates (tables that fit one L2: an upper
        # bound on the rate, since part of the real fetches miss L2).
        cus, clk = 256, 2.4e9
        node_cap = caps["node96_l2"] if node_bytes == 96 else caps["node64_l2"] if node_bytes == 64 else caps["tri48_l2"]   # (packed 48-byte BVH2 nodes fetch like triangles)
        floor_s = (counts["nodes"] / node_cap + counts["tri_tests"] / caps["tri48_l2"]) / cus / clk
        vmem = {"records_per_launch": counts["nodes"] + counts["tri_tests"],
                "achieved_records_per_cycle_per_cu": round((counts["nodes"] + counts["tri_tests"]) / (sec * clk * cus), 4),
                "peak_node_records_per_cycle_per_cu": node_cap, "peak_triangle_records_per_cycle_per_cu": caps["tri48_l2"],
                "floor_ms": round(floor_s * 1e3, 3), "frac": round(floor_s / sec, 3), "source": ub.get("vmem_source"),
                "note": "the vector memory pipeline's ceiling for this access pattern: one random record per lane and step through the "
                        "vector L1 (16 B per lane and load; ~0.7 cycle per lane-load), at the rate of tables that fit one L2.  frac = time "
                        "the launch's record fetches need at that rate / kernel time.  The HBM fraction above is what the contract's "
                        "`bound` asks for; this and valu_issue are the two ceilings the kernel runs under, neither saturated: what is "
                        "left is latency that five waves per SIMD do not cover (DESIGN.md §5.2)."}
    out = {"bound": bound, "traffic": traffic, "traffic_source": pmc.get("source") if pmc else None,
           "kernel": kernel, "kernel_ms": round(k_ms, 4),
           "kernel_mrays_per_s": round(counts["rays"] / sec / 1e6, 1), "valu": valu, "hbm": hbm, "vmem": vmem}
    if bound == "valu":
        out.update({"achieved": valu["achieved_tflops"], "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": valu["frac"]})
    else:
        out.update({"achieved": hbm["measured_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm["frac"]})
    if pmc and "valu_insts_per_launch" in pmc:
        rate = pmc["valu_insts_per_launch"] / sec / 1e9
        out["valu_issue"] = {
            "achieved_ginst_per_s": round(rate, 1), "peak_ginst_per_s": VALU_ISSUE_PEAK_GINST,
            "frac": round(rate / VALU_ISSUE_PEAK_GINST, 3),
            "measured_peak_ginst_per_s": ub.get("valu_issue_peak_ginst_per_s"),
            "frac_of_measured_peak": round(rate / ub["valu_issue_peak_ginst_per_s"], 3) if ub.get("valu_issue_peak_ginst_per_s") else None,
            "lane_utilisation": pmc.get("valu_lane_utilisation"),
            "insts_per_ray": round(pmc["valu_insts_per_launch"] * 64 / counts["rays"], 1),
            "note": "wave64 VALU instructions per launch from the committed PMC pass (SQ_INSTS_VALU) / this run's kernel "
                    "time.  peak: 1024 SIMDs x 2.4 GHz / 2 cycles; measured_peak: the best sustained rate of an independent "
                    "v_fma_f32 stream on all CUs (the chip lowers its clock under that load), " + str(ub.get("source"))}
    else:
        out["valu_issue"] = None
    return out


def cpu_baseline(flat_c, cam, label, width, height, depth, sample_spp, frame_spp, one_core=True):
    """CPU oracle on the box's host cores over the flattened scene the GPU traverses: the SAME
    scene / frame size / depth at `sample_spp` of the frame's `frame_spp` samples per pixel (a
    bounded sample; the rate is flat in spp: samples are independent paths of equal expected cost)."""
    import rt_amd as ra
    from oracle import Oracle

    cores = min(len(os.sched_getaffinity(0)), 16)
    st = ra.default_settings(width, height, spp=sample_spp, max_depth=depth)
    o = Oracle()
    # what the GPU traverses: the slot tree when the scene has one - made once, outside the timed calls, like the GPU's
    slots = o.slot_tree(flat_c) if flat_c.nodes4 else None
    t0 = time.perf_counter()
    _, _, stats = o.render(flat_c, cam, st, threads=cores, want_ids=False, slots=slots, slot_tree=False)
    dt = time.perf_counter() - t0
    out = {
        "value": round(stats["rays"] / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": f"{label} {width}x{height}, {sample_spp} of the frame's {frame_spp} spp, depth {depth}: "
                  f"{stats['rays']} rays in {dt:.2f} s (OpenMP over rows; rate flat in spp)",
    }
    if one_core:
        # the reference program is single-threaded by design: one core, 1 spp
        st1 = ra.default_settings(width, height, spp=1, max_depth=depth)
        t0 = time.perf_counter()
        _, _, stats1 = o.render(flat_c, cam, st1, threads=1, want_ids=False, slots=slots, slot_tree=False)
        dt1 = time.perf_counter() - t0
        out["one_core"] = {"value": round(stats1["rays"] / dt1 / 1e6, 3), "unit": "Mrays/s", "cores": 1,
                           "sample": f"1 spp of the same frame: {stats1['rays']} rays in {dt1:.2f} s"}
    return out


def also_workload(ra, torch, ctx, dev, name, label, kind, param, w, h, spp, depth, bound, reps, cpu_spp):
    """One more workload on this GPU, outside the timed region: `reps` frames after one
    instrumented frame (counts) that doubles as the warm-up; then the CPU oracle on `cpu_spp`
    samples per pixel of the same frame (0 = skip)."""
    scene, cam = ra.Scene.builtin(kind, param=param, aspect=w / h)
    ctx.upload(scene)
    acc = torch.zeros((h * w, 4), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    st = ra.default_settings(w, h, spp=spp, max_depth=depth)
    st_s = st.copy()
    st_s.flags |= ra.FLAG_STATS
    stats = ra.Stats()
    ctx.render_device(cam, st_s, acc.data_ptr(), None, stream, stats)
    counts = stats.as_dict()
    ms, t0 = [], time.perf_counter()
    for _ in range(reps):
        ctx.render_device(cam, st, acc.data_ptr(), None, stream)
        ms.append(ctx.kernel_ms())
    torch.cuda.synchronize(dev)
    wall_ms = (time.perf_counter() - t0) * 1e3 / reps
    k_ms = sum(ms) / len(ms)
    pmc, ub = committed_pmc(name, w, h, spp, depth)
    del acc
    out = {"workload": label + f" ({scene.n_prims} prims), {w}x{h}, {spp} spp, depth {depth}, one GPU",
           "ms": round(wall_ms, 3), "frames": reps, "rays_per_frame": counts["rays"],
           "mrays_per_s": round(counts["rays"] / (wall_ms * 1e-3) / 1e6, 1),
           "roofline": roofline_block(bound, counts, w * h, node_bytes_of(ctx), k_ms, ctx.kernel_name(), pmc, ub)}
    out["cpu_baseline"] = (cpu_baseline(ctx.flat_scene().c, cam, label.split(":")[0], w, h, depth, cpu_spp, spp, one_core=False)
                           if cpu_spp > 0 else None)
    return out


CONFIGS = {
    # name: (scene kind, scene param, label, width, height, spp, scaling, bound, steps, warmup)
    "c2": (0, 0, "C2: procedural scene S0 (RTIOW cover", 1920, 1080, 16, "weak", "valu", 20, 3),
    "c4": (0, 0, "C4: procedural scene S0 (RTIOW cover", 3840, 2160, 256, "strong", "valu", 5, 1),
    "c5": (2, 8, "C5: procedural soup S2 level 8 (8 displaced icospheres", 1920, 1080, 16, "strong", "hbm", 10, 2),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=tuple(CONFIGS), default=None,
                    help="c2: 1080p, 16 spp per GPU (weak scaling over N); c4: 4K, 256 spp in total (strong scaling); "
                         "c5: 10.5 M triangles, 1080p, 16 spp in total (strong scaling); default c2 at --gpus 1, c4 otherwise")
    ap.add_argument("--transport", choices=("group", "procs"), default="group",
                    help="--gpus N > 1 WITHOUT a launcher: group = the native rtgpu_group in this process (one thread per GPU, "
                         "RCCL send/recv); procs = N fresh child processes under torch.distributed.run.  Ignored when "
                         "WORLD_SIZE is set (the launcher's per-process path)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None, help="c2: samples per pixel PER GPU; c4, c5: per frame")
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--builder", choices=("sah", "lbvh"), default="sah", help="BVH builder (lbvh: on the GPU)")
    ap.add_argument("--cpu-spp", type=int, default=16, help="spp of the cpu_baseline sample of the timed workload (0 = skip all CPU baselines)")
    ap.add_argument("--no-also", action="store_true", help="skip the C3 / C4-frame / C5-frame measurements of the `also` array")
    return ap.parse_args()


def resolve_config(args, world):
    config = args.config or ("c2" if world == 1 else "c4")
    kind, param, label, w0, h0, spp0, scaling, bound, steps0, warm0 = CONFIGS[config]
    W, H = args.width or w0, args.height or h0
    spp_arg = args.spp or spp0
    spp = spp_arg * world if config == "c2" else spp_arg   # c2: per-GPU work fixed; c4, c5: the frame is fixed
    steps = args.steps or steps0
    warmup = warm0 if args.warmup is None else args.warmup
    return config, kind, param, label, W, H, spp_arg, spp, scaling, bound, steps, warmup


def workload_text(config, label, n_prims, W, H, spp_arg, spp, depth, tile, world, how):
    return (f"{label}, {n_prims} prims), {W}x{H}, "
            + (f"{spp_arg} spp per GPU ({spp} spp total)" if config == "c2" else f"{spp} spp per frame (split over the ranks)")
            + f", depth {depth}, pinhole, {tile}x{tile} tiles interleaved over {world} GPU(s)" + how)


def estimate_for_rank(pmc, world):
    """N > 1: no PMC pass of a rank's compact launch is committed; what the single-GPU pass of the whole
    frame implies for one rank (1/N of its tiles) is reported as an ESTIMATE under its own key, never as
    `traffic` / `hbm.measured_gbs` / `valu_issue`."""
    if not pmc:
        return None
    return {"estimated": True, "model": f"1/{world} of the single-GPU launch of the whole frame",
            "traffic_bytes_per_launch": pmc["traffic_bytes_per_launch"] / world,
            "valu_insts_per_launch": pmc.get("valu_insts_per_launch", 0) / world or None,
            "source": pmc.get("source")}


def self_launch(args):
    """--gpus N > 1, no launcher, --transport procs: N fresh children under torch.distributed.run; this
    process never touches a GPU (a process that has initialised the GPU must not be replaced, and is not:
    the children are started, not exec'ed) and relays rank 0's JSON line and the children's exit code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if r.returncode != 0 or len(lines) != 1:
        sys.stdout.write(r.stdout)
        raise SystemExit(f"bench.py: the {args.gpus} child ranks failed (exit code {r.returncode}, {len(lines)} result lines)")
    print(lines[0], flush=True)


def need_gpus(torch, n, share_gpu):
    have = torch.cuda.device_count()   # counting devices does not initialise the GPU
    want = 1 if share_gpu else n
    if have < want:
        raise SystemExit(f"bench.py --gpus {n} needs {want} GPU(s), this machine has {have} (no CPU fallback exists)")


def run_group(args, torch, ra, share_gpu):
    """--gpus N > 1 in ONE process: the native group (rt_amd/csrc/group.cpp), frames pipelined two deep."""
    world = args.gpus
    config, kind, param, label, W, H, spp_arg, spp, scaling, bound, steps, warmup = resolve_config(args, world)
    depth, tile = args.depth, 16
    devices = [0] * world if share_gpu else list(range(world))
    g = ra.Group(devices)
    scene, cam = ra.Scene.builtin(kind, param=param, aspect=W / H)
    g.upload(scene, builder=args.builder)
    st = ra.default_settings(W, H, spp=spp, max_depth=depth, tile_size=tile)
    st_stats = st.copy()
    st_stats.flags |= ra.FLAG_STATS
    _, _, stats = g.render(cam, st_stats, want_accum=False, want_rgba8=False)   # counts for the roofline, untimed
    counts = stats.as_dict()

    def sync():
        for d in set(devices):
            torch.cuda.synchronize(d)

    def frames(n):
        rays = 0
        for k in range(n + 1):
            if k < n:
                g.render_begin(cam, st)          # frame k+1 is queued before frame k is collected
            if k > 0:
                rays += g.render_end(want_accum=False, want_rgba8=False)[2].rays
        return rays

    frames(warmup)
    sync()
    t0 = time.perf_counter()
    rays = frames(steps)
    sync()
    dt = time.perf_counter() - t0
    k_ms = max(g.kernel_ms(i) for i in range(world))
    ranks = g.comm_ranks
    g.close()
    pmc, ub = committed_pmc(config, W, H, spp_arg, depth)
    per_rank = {k: v // world for k, v in counts.items()}
    node_bytes = 64   # S0: the tree's LDS copy; trees in memory: the node slots of the slot tree
    roof = roofline_block(bound, per_rank, W * H // world, node_bytes, k_ms, "trace_kernel (slowest rank's last frame)", None, ub,
                          in_memory=kind != 0)
    roof["estimated_from_single_gpu_pass"] = estimate_for_rank(pmc, world)
    how = (f", native group: one process, {world} worker threads, "
           + (f"RCCL communicator of {ranks} ranks (ncclCommInitAll), ncclSend / ncclRecv to GPU {devices[0]}" if ranks
              else "tile buffers moved by device copies")
           + ", gather overlapped with the next frame's trace"
           + (" [REHEARSAL: all ranks on GPU 0, no RCCL]" if share_gpu else ""))
    ms_step = dt / steps * 1e3
    out = {
        "metric": "Mrays/s (whole job)", "value": round(rays / dt / 1e6, 2), "unit": "Mrays/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload_text(config, label, scene.n_prims, W, H, spp_arg, spp, depth, tile, world, how),
                   "transport": "group", "rccl_ranks": ranks, "rays_per_step": rays // steps, "paths_per_step": W * H * spp,
                   "mrays_per_s_per_gpu": round(rays / dt / 1e6 / world, 2), "ms_per_frame_1080p_16spp": None},
        "roofline": roof, "cpu_baseline": None,
    }
    print(json.dumps(out), flush=True)


def main():
    args = parse_args()
    launched = "WORLD_SIZE" in os.environ
    share_gpu = os.environ.get("RT_BENCH_SHARE_GPU") == "1"
    # RT_BENCH_SHARE_GPU=1 is a REHEARSAL mode for a one-GPU box: every rank uses GPU 0 and the gather
    # goes through gloo on host copies (procs) or device copies (group).  Never used by the driver; the
    # JSON says so in config.workload.
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    import torch
    if not launched and args.gpus > 1:
        need_gpus(torch, args.gpus, share_gpu)
        if args.transport == "procs":
            return self_launch(args)
        import rt_amd as ra
        return run_group(args, torch, ra, share_gpu)

    import torch.distributed as dist

    import rt_amd as ra

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    need_gpus(torch, world, share_gpu)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    config, kind, param, label, W, H, spp_arg, spp, scaling, bound, steps, warmup = resolve_config(args, world)
    depth = args.depth
    # 16x16 tiles: the slowest of 8 ranks is then 0.8 % above the mean (64x64: 4.2 %, 32x32:
    # 1.3 %; tools/gpu_rank_balance.py), and every rank owns the same number of tiles.  The
    # image does not depend on the tile size.
    tile = 16
    ctx = ra.Context(local_rank)
    scene, cam = ra.Scene.builtin(kind, param=param, aspect=W / H)
    ctx.upload(scene, builder=args.builder)

    compact = world > 1
    flags = ra.FLAG_COMPACT if compact else 0
    st = ra.default_settings(W, H, spp=spp, max_depth=depth, tile_size=tile, rank=rank,
                             world_size=world, flags=flags)
    n_out = ra.compact_pixels(W, H, tile, 0, world) if compact else W * H  # rank 0 has the most tiles
    # two compact buffers: frame k+1 is traced while frame k is gathered
    accum = [torch.zeros((n_out, 4), dtype=torch.float32, device=dev) for _ in range(2 if compact else 1)]
    image = torch.zeros((H * W, 4), dtype=torch.float32, device=dev) if compact else accum[0]
    gathered = ([torch.zeros((world * n_out, 4), dtype=torch.float32, device=dev) for _ in range(2)]
                if compact and rank == 0 else None)
    stream = torch.cuda.current_stream(dev)
    in_flight = [None, None]   # the gather of the frame last traced into accum[b]

    def finish(b):
        """Frame in buffer b: wait for its gather (stream-side, the host does not block) and de-interleave."""
        w = in_flight[b]
        if w is None:
            return
        in_flight[b] = None
        if w is not True:
            w.wait()
        if rank == 0:
            ctx.untile_device(gathered[b].data_ptr(), image.data_ptr(), W, H, tile, world, 16, stream.cuda_stream)

    def step(k):
        if not compact:
            ctx.render_device(cam, st, accum[0].data_ptr(), None, stream.cuda_stream)
            return ctx.kernel_ms()  # HIP events around the trace kernel on `stream`
        b = k & 1
        finish(b)                  # frame k-2 (same buffer): nothing left to do unless it was skipped
        ctx.render_device(cam, st, accum[b].data_ptr(), None, stream.cuda_stream)
        # the exchange step of the path: compact tiles -> rank 0
        if share_gpu:  # rehearsal only: gloo cannot move device tensors
            host = accum[b].cpu()
            parts = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
            dist.gather(host, parts, dst=0)
            if rank == 0:
                gathered[b].copy_(torch.cat(parts))
            in_flight[b] = True
        else:
            in_flight[b] = dist.gather(accum[b], list(gathered[b].chunk(world)) if rank == 0 else None, dst=0,
                                       async_op=True)
        finish(b ^ 1)              # frame k-1: its gather ran beside this frame's trace
        return None

    def drain():
        finish(0)
        finish(1)

    # counts for the roofline: instrumented variant of the same frame, untimed
    st_stats = st.copy()
    st_stats.flags = flags | ra.FLAG_STATS
    stats = ra.Stats()
    ctx.render_device(cam, st_stats, accum[0].data_ptr(), None, stream.cuda_stream, stats)
    counts = stats.as_dict()

    for k in range(warmup):
        step(k)
    drain()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    kernel_ms = []
    for k in range(steps):
        kernel_ms.append(step(k))
    drain()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if compact:
        kernel_ms = [ctx.kernel_ms()]  # the last frame's kernel (reading it every step would serialise the pipeline)

    t_all = torch.tensor([dt], dtype=torch.float64, device=dev)
    rays_all = torch.tensor([float(counts["rays"])], dtype=torch.float64, device=dev)
    if share_gpu:
        t_all, rays_all = t_all.cpu(), rays_all.cpu()
    if world > 1:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays_all, op=dist.ReduceOp.SUM)
    dt = float(t_all.item())
    rays_per_step = float(rays_all.item())

    if rank == 0:
        k_ms = sum(kernel_ms) / len(kernel_ms)
        n_px_rank = counts["paths"] // spp
        pmc, ub = committed_pmc(config, W, H, spp_arg, depth)
        ms_step = dt / steps * 1e3
        how = ""
        if compact:
            how = (f", one process per GPU, torch.distributed gather to rank 0 over "
                   + ("gloo via host copies [REHEARSAL: all ranks on one GPU]" if share_gpu
                      else f"backend nccl (RCCL communicator of {dist.get_world_size()} ranks)")
                   + ", overlapped with the next frame's trace")
        roof = roofline_block(bound, counts, n_px_rank, node_bytes_of(ctx), k_ms, ctx.kernel_name(), pmc if world == 1 else None, ub)
        if world > 1:
            roof["estimated_from_single_gpu_pass"] = estimate_for_rank(pmc, world)
        out = {
            "metric": "Mrays/s (whole job)", "value": round(rays_per_step * steps / dt / 1e6, 2),
            "unit": "Mrays/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": workload_text(config, label, scene.n_prims, W, H, spp_arg, spp, depth, tile, world, how),
                "transport": "procs" if compact else None,
                "rccl_ranks": (0 if share_gpu else dist.get_world_size()) if compact else None,
                "rays_per_step": int(rays_per_step), "paths_per_step": W * H * spp,
                "mrays_per_s_per_gpu": round(rays_per_step * steps / dt / 1e6 / world, 2),
                "ms_per_frame_1080p_16spp": round(ms_step, 4) if (world == 1 and config == "c2" and (W, H, spp) == (1920, 1080, 16)) else None,
            },
            "roofline": roof,
        }
        if args.cpu_spp > 0 and world == 1:
            sample = args.cpu_spp if config != "c5" else min(args.cpu_spp, 2)
            out["cpu_baseline"] = cpu_baseline(ctx.flat_scene().c, cam, config.upper(), W, H, depth, min(sample, spp), spp)
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_also:
            cpu = args.cpu_spp > 0
            also = []
            if config != "c5":  # (the timed workload itself otherwise)
                also.append(also_workload(ra, torch, ctx, dev, "c5", "C5: procedural soup S2 level 8", 2, 8, 1920, 1080, 16, 8, "hbm", 3, 2 if cpu else 0))
            also.append(also_workload(ra, torch, ctx, dev, "c3", "C3: procedural mesh S1 level 8", 1, 8, 1920, 1080, 64, 8, "hbm", 3, 4 if cpu else 0))
            if config == "c2":
                also.append(also_workload(ra, torch, ctx, dev, "c4", "C4's frame: S0", 0, 0, 3840, 2160, 256, 8, "valu", 2, 2 if cpu else 0))
            out["also"] = also
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
