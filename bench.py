#!/usr/bin/env python3
"""bench.py — the headline metric of BASELINE.json on MI355X.

A "step" is one pass of the trace loop over one frame of the default scene S0
(SURVEY.md §8d).  At N=1 the workload is BASELINE.json configs[1] ("C2"):
S0, 1920x1080, 16 spp, depth 8, one MI355X.  At N>1 the frame is split into
16x16 tiles interleaved over the ranks (one process per GPU), every rank traces
its tiles from a replicated scene, RCCL gathers the compact tile buffers to
rank 0 and rank 0 de-interleaves them; the per-GPU work is held fixed (weak
scaling) by tracing 16*N spp, so a step at N GPUs is a 16N-spp frame.

Prints ONE JSON line (rank 0).  `value` = rays (segments: primary + every
bounce, counted by the kernel) traced by all ranks per second of wall time,
inputs resident in HBM.  `roofline` is the trace kernel's algorithmic bytes
(64 B/node + 48 B/triangle test + 16 B/sphere test + 32 B/
material fetch + 16 B/pixel written, counted by the instrumented kernel on the
same frame and checked against the oracle in tests) divided by the kernel's
mean duration (HIP events on the stream it ran on) vs 8 TB/s HBM.
`cpu_baseline` is the CPU oracle (oracle/rt_oracle.c) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak (spec)
VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 vector (spec)
VALU_ISSUE_PEAK_GINST = 614.4  # 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction


def algorithmic_bytes(st, n_pixels, node_bytes=64):
    """node_bytes: 64 (the C2 tree, read from its LDS copy) or 48 (packed nodes of trees of
    1366 nodes or more, include/rtgpu_format.h)."""
    return node_bytes * st["nodes"] + 48 * st["tri_tests"] + 16 * st["sphere_tests"] + 32 * st["scatters"] + 16 * n_pixels


def algorithmic_flops(st):
    return 34 * st["nodes"] + 47 * st["tri_tests"] + 21 * st["sphere_tests"] + 60 * st["scatters"]


def measured_traffic(width, height, spp, depth, world):
    """HBM bytes per launch of the trace kernel from the PMC passes of THIS
    workload (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in runs of their own,
    tools/final_measure.sh; units and gfx950 correction as MI355X_MICROARCH.md
    prescribes, calibrated on the path's own access pattern with
    tools/ubench/fetch_calib.hip).  Counters cannot be read from inside this
    process, so the committed summary of the last profile is reported, and only
    when it was taken on the same workload; otherwise null."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        doc = json.load(open(path))
    except (OSError, ValueError):
        return None, None, None
    wl = doc.get("workload", {})
    if world != 1 or [wl.get(k) for k in ("width", "height", "spp", "depth")] != [width, height, spp, depth]:
        return None, None, None
    return int(doc["traffic_bytes_per_launch"]), doc.get("source"), doc


def cpu_baseline(width, height, depth, sample_spp):
    """CPU oracle on the box's host cores: the SAME scene/resolution/depth at
    `sample_spp` samples per pixel (a bounded sample of the 16-spp frame)."""
    import rt_amd as ra
    from oracle import Oracle

    cores = min(len(os.sched_getaffinity(0)), 16)
    scene, cam = ra.Scene.builtin(0, aspect=width / height)
    fs = ra.FlatScene.build(scene)
    st = ra.default_settings(width, height, spp=sample_spp, max_depth=depth)
    o = Oracle()
    t0 = time.perf_counter()
    _, _, stats = o.render(fs.c, cam, st, threads=cores, want_ids=False)
    dt = time.perf_counter() - t0
    # the reference program is single-threaded by design: one core, 1 spp
    st1 = ra.default_settings(width, height, spp=1, max_depth=depth)
    t0 = time.perf_counter()
    _, _, stats1 = o.render(fs.c, cam, st1, threads=1, want_ids=False)
    dt1 = time.perf_counter() - t0
    return {
        "value": round(stats["rays"] / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": f"S0 {width}x{height}, {sample_spp} of the frame's spp, depth {depth}: "
                  f"{stats['rays']} rays in {dt:.2f} s (OpenMP over rows)",
        "one_core": {"value": round(stats1["rays"] / dt1 / 1e6, 3), "unit": "Mrays/s", "cores": 1,
                     "sample": f"1 spp of the same frame: {stats1['rays']} rays in {dt1:.2f} s"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=16, help="samples per pixel PER GPU (weak scaling)")
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--cpu-spp", type=int, default=16, help="spp of the cpu_baseline sample (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import rt_amd as ra

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    # RT_BENCH_SHARE_GPU=1 is a REHEARSAL mode for a one-GPU box: every rank uses
    # GPU 0 and the gather goes through gloo on host copies.  Never used by the
    # driver; the JSON says so in config.workload.
    share_gpu = os.environ.get("RT_BENCH_SHARE_GPU") == "1"
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

    W, H, depth = args.width, args.height, args.depth
    spp = args.spp * world
    # 16x16 tiles: the slowest of 8 ranks is then 0.8 % above the mean (64x64: 4.2 %, 32x32:
    # 1.3 %; tools/gpu_rank_balance.py), and every rank owns the same number of tiles.  The
    # image does not depend on the tile size.
    tile = 16
    ctx = ra.Context(local_rank)
    scene, cam = ra.Scene.builtin(0, aspect=W / H)
    ctx.upload(scene)

    compact = world > 1
    flags = ra.FLAG_COMPACT if compact else 0
    st = ra.default_settings(W, H, spp=spp, max_depth=depth, tile_size=tile, rank=rank,
                             world_size=world, flags=flags)
    n_out = ra.compact_pixels(W, H, tile, 0, world) if compact else W * H  # rank 0 has the most tiles
    accum = torch.zeros((n_out, 4), dtype=torch.float32, device=dev)
    image = torch.zeros((H * W, 4), dtype=torch.float32, device=dev) if compact else accum
    gathered = (torch.zeros((world * n_out, 4), dtype=torch.float32, device=dev)
                if compact and rank == 0 else None)
    stream = torch.cuda.current_stream(dev)

    def step():
        ctx.render_device(cam, st, accum.data_ptr(), None, stream.cuda_stream)
        ms = None
        if compact:
            # real exchange step of the path: tiles -> rank 0
            if share_gpu:  # rehearsal only: gloo cannot move device tensors
                host = accum.cpu()
                parts = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
                dist.gather(host, parts, dst=0)
                if rank == 0:
                    gathered.copy_(torch.cat(parts))
            else:
                dist.gather(accum, list(gathered.chunk(world)) if rank == 0 else None, dst=0)
            if rank == 0:
                ctx.untile_device(gathered.data_ptr(), image.data_ptr(), W, H, tile, world, 16,
                                  stream.cuda_stream)
        ms = ctx.kernel_ms()  # HIP events around the trace kernel on `stream`
        return ms

    # counts for the roofline: instrumented variant of the same frame, untimed
    st_stats = st.copy()
    st_stats.flags = flags | ra.FLAG_STATS
    stats = ra.Stats()
    ctx.render_device(cam, st_stats, accum.data_ptr(), None, stream.cuda_stream, stats)
    counts = stats.as_dict()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        kernel_ms.append(step())
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0

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
        abytes = algorithmic_bytes(counts, n_px_rank, (96 if ctx.flat_scene().c.nodes4 else 48 if ctx.flat_scene().c.nodes48 else 64))
        aflops = algorithmic_flops(counts)
        achieved = abytes / (k_ms * 1e-3) / 1e9
        traffic, traffic_src, pmc = measured_traffic(W, H, args.spp, depth, world)
        out = {
            "metric": "Mrays/s (whole job)", "value": round(rays_per_step * args.steps / dt / 1e6, 2),
            "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"C2: procedural scene S0 (RTIOW cover, {scene.n_prims} prims), "
                            f"{W}x{H}, {args.spp} spp per GPU ({spp} spp total), depth {depth}, "
                            f"pinhole, {tile}x{tile} tiles interleaved over {world} GPU(s)"
                            + (", RCCL gather to rank 0" if compact and not share_gpu else "")
                            + (" [REHEARSAL: all ranks on one GPU, gloo gather via host]" if share_gpu else ""),
                "rays_per_step": int(rays_per_step), "paths_per_step": W * H * spp,
                "mrays_per_s_per_gpu": round(rays_per_step * args.steps / dt / 1e6 / world, 2),
                "ms_per_frame_1080p_16spp": round(dt / args.steps * 1e3, 4) if world == 1 else None,
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": traffic_src,
                "note": "achieved counts ALGORITHMIC bytes (every node/primitive/material fetch a ray makes); "
                        "most are served from the LDS node cache and the vector L1, so frac can exceed what "
                        "HBM delivers - `traffic` is what actually crossed the L2's memory side; the kernel "
                        "is bound by VALU issue and divergent-fetch latency, see `valu` and DESIGN.md",
                "kernel": "trace_kernel<STATS=0,REFILL=16>", "kernel_ms": round(k_ms, 4),
                "algorithmic_bytes_per_launch": int(abytes),
                "bytes_per_ray": round(abytes / counts["rays"], 1),
                "valu": {"achieved_tflops": round(aflops / (k_ms * 1e-3) / 1e12, 3),
                         "peak_tflops": VALU_PEAK_TFLOPS,
                         "frac": round(aflops / (k_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4),
                         "flops_per_ray": round(aflops / counts["rays"], 1)},
                "kernel_mrays_per_s": round(counts["rays"] / (k_ms * 1e-3) / 1e6, 1),
                # the bound that actually holds on this workload: VALU issue slots.  A wave64 VALU
                # instruction occupies its SIMD for 4 cycles: 1024 SIMDs x 2.4 GHz / 4 = 614.4 G
                # wave-instructions/s.  Instruction count from the committed PMC pass of this workload.
                "valu_issue": None if not pmc or "valu_insts_per_launch" not in pmc else {
                    "achieved_ginst_per_s": round(pmc["valu_insts_per_launch"] / (k_ms * 1e-3) / 1e9, 1),
                    "peak_ginst_per_s": VALU_ISSUE_PEAK_GINST,
                    "frac": round(pmc["valu_insts_per_launch"] / (k_ms * 1e-3) / 1e9 / VALU_ISSUE_PEAK_GINST, 3),
                    "lane_utilisation": pmc.get("valu_lane_utilisation"),
                    "insts_per_ray": round(pmc["valu_insts_per_launch"] * 64 / counts["rays"], 1),
                    "note": "instruction count from the committed PMC pass, kernel time from this run, "
                            "peak at the nominal 2.4 GHz (one wave64 VALU instruction per SIMD per 4 cycles): "
                            "0.99-1.03 across runs and counter sources, i.e. every VALU issue slot of the "
                            "launch is used"},
            },
        }
        if args.cpu_spp > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(W, H, depth, args.cpu_spp)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
