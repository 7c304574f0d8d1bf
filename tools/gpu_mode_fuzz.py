"""Random small scenes through every small-scene kernel mode against the oracle, bit for bit (frames, primary ids,
counters): sphere / triangle / material counts around the limits of the LDS modes, random cameras with and without a
lens, random depth, spp, chunking and tile partition.  usage: python tools/gpu_mode_fuzz.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import rt_amd as ra
from oracle import Oracle
from util import make_scene

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
oracle = Oracle()
ctx = ra.Context(0)
modes = {}
for case in range(n_cases):
    n_sph = int(rng.choice([0, 1, 2, 7, 60, 200, 495, 496, 497, 700, 1300]))
    n_tri = int(rng.choice([0, 1, 2, 15, 16, 17, 40, 300]))
    if n_sph + n_tri == 0:
        n_sph = 3
    n_mat = int(rng.choice([1, 3, 6, 600]))
    mats = []
    for i in range(n_mat):
        t = int(rng.integers(0, 4))
        p = {0: 0.0, 1: float(rng.choice([0.0, 0.3, 1.0])), 2: float(rng.choice([1.3, 1.5, 2.4])), 3: float(rng.uniform(1, 6))}[t]
        mats.append((t, *rng.uniform(0.2, 1.0, 3).tolist(), p))
    sph = [(*rng.uniform(-3, 3, 3).tolist(), float(rng.uniform(0.05, 0.8)), int(rng.integers(0, n_mat))) for _ in range(n_sph)]
    tri = []
    for _ in range(n_tri):
        c = rng.uniform(-3, 3, 3)
        tri.append((tuple(c + rng.normal(0, .6, 3)), tuple(c + rng.normal(0, .6, 3)), tuple(c + rng.normal(0, .6, 3)),
                    int(rng.integers(0, n_mat))))
    scene = make_scene(ra, spheres=sph, triangles=tri, materials=mats)
    w, h = int(rng.choice([33, 64, 96, 130])), int(rng.choice([17, 48, 64]))
    ap = float(rng.choice([0.0, 0.0, 0.2]))
    cam = ra.look_at(tuple(rng.uniform(-2, 2, 3)), tuple(rng.uniform(-3, 3, 3)), vfov=float(rng.uniform(30, 90)), aspect=w / h,
                     aperture=ap, focus_dist=3.0)
    spp = int(rng.choice([1, 3, 5, 16]))
    world = int(rng.choice([1, 1, 3]))
    st = ra.default_settings(w, h, spp=spp, max_depth=int(rng.choice([1, 2, 6, 12])), seed=int(rng.integers(0, 1 << 31)),
                             chunk_samples=int(rng.choice([1, 2, 4, 16])), tile_size=int(rng.choice([8, 16, 64])),
                             rank=int(rng.integers(0, world)), world_size=world, flags=ra.FLAG_STATS)
    ctx.upload(scene)
    a, ia, sa = ctx.render(cam, st)
    st_o = st.copy(); st_o.flags = 0
    b, ib, sb = oracle.render(ctx.flat_scene().c, cam, st_o, threads=16)
    mode = ctx.kernel_name().rstrip(">").split(",")[-1]
    modes[mode] = modes.get(mode, 0) + 1
    ok = np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(ia, ib) and sa.as_dict() == sb
    print(f"case {case}: {n_sph} sph {n_tri} tri {n_mat} mat {w}x{h}x{spp} depth {st.max_depth} chunk {st.chunk_samples} "
          f"rank {st.rank}/{world} lens {ap} mode {mode}: {'ok' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        sys.exit(1)
print("modes run:", modes)
