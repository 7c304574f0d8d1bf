"""SIMD lane utilisation per kernel section (profile variant, RTGPU_FLAG_PROFILE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rt_amd as ra
from gpu_perf import CONFIGS
ctx = ra.Context(0)
for name in sys.argv[1:] or ["c2", "c3", "c5"]:
    kind, level, w, h, spp, depth = CONFIGS[name]
    scene, cam = ra.Scene.builtin(kind, param=level, aspect=w / h)
    ctx.upload(scene)
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    st = ra.default_settings(w, h, spp=spp, max_depth=depth, flags=ra.FLAG_PROFILE)
    ctx.render_device(cam, st, acc.data_ptr(), None, None)
    c = ctx.profile_counters()
    _, paths, rays, nodes, tris, sphs, scat, w_outer, w_shade, l_shade, w_gen, w_node, w_sph, w_tri, w_trav, l_trav = c[:16]
    cyc = c[16:21]
    u = lambda l, wv: l / (64.0 * wv) if wv else float("nan")
    print(f"{name}: rays={rays} | wave-level counts per 64 rays: outer={w_outer*64/rays:.2f} shade={w_shade*64/rays:.2f} "
          f"gen={w_gen*64/rays:.2f} node={w_node*64/rays:.1f} sph={w_sph*64/rays:.2f} tri={w_tri*64/rays:.2f} trav_rounds={w_trav*64/rays:.2f}\n"
          f"   lane utilisation: shade={u(l_shade, w_shade):.3f} gen={u(paths, w_gen):.3f} node={u(nodes, w_node):.3f} "
          f"sph={u(sphs, w_sph):.3f} tri={u(tris, w_tri):.3f} trav_round={u(l_trav, w_trav):.3f}\n"
          f"   wave time share: " + " ".join(f"{n}={v / max(1, sum(cyc)):.3f}" for n, v in
                                           zip(("shade", "item", "gen", "setup", "traversal"), cyc))
          + f"  (of traversal: node phases {c[21] / max(1, c[20]):.3f}, leaf phases {c[22] / max(1, c[20]):.3f};"
          + f" cycles per node step {c[21] / max(1, w_node):.0f}, per leaf step {c[22] / max(1, w_sph + w_tri):.0f})"
          + f"\n   lanes still traversing per node step {c[26] / max(1, w_node):.1f} (of them at a node {nodes / max(1, c[26]):.3f}), "
            f"per leaf step {c[27] / max(1, w_sph + w_tri):.1f} (of them tested {(sphs + tris) / max(1, c[27]):.3f})"
          + f"\n   shader-clock ticks of a wave per outer iteration: " + " ".join(f"{n}={v / max(1, w_outer):.0f}" for n, v in
                                           zip(("shade", "item", "gen", "setup", "traversal"), cyc))
          + f"; per wave in all {sum(cyc) / 4096:.0f} ticks for a {ctx.kernel_ms():.3f} ms kernel = {sum(cyc) / 4096 / ctx.kernel_ms() / 1e3:.0f} ticks per microsecond"
          + (f"\n   {ctx.kernel_name()}: node-step lanes served from the LDS copy of the top {c[24] / max(1, c[23]):.3f}, "
             f"node steps on the slow stack path {c[25] / max(1, w_node):.4f}" if c[23] else "")
          + (f"\n   {ctx.kernel_name()}: slot exchanges per 64 rays {c[28] * 64 / rays:.2f}, lanes per exchange {c[29] / max(1, c[28]):.1f}"
             if c[28] else ""), flush=True)
ctx.close()
