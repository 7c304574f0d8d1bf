"""rtgpu_render_device_static, the entry point that allocates nothing: its launch scratch
(queue head, counters, chunk planes) is a static device array of the library.  Called through
the C ABI and compared bit for bit with rtgpu_render_device on two calls in a row, the second
with a smaller input (the array keeps the first call's contents), then against the oracle;
free device memory must not move across the calls; sizes beyond the array are refused.
(Named to be collected last: the round's end runs pytest -x.)"""
import numpy as np
import pytest

from util import assert_bit_equal, make_scene

pytestmark = pytest.mark.gpu


def _pair(ra, ctx, cam, st):
    """The same frame through rtgpu_render_device and rtgpu_render_device_static."""
    import torch
    n = st.width * st.height
    want = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    want_ids = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    got = torch.full((n, 4), 7.0, dtype=torch.float32, device="cuda")      # stale output: must be overwritten
    got_ids = torch.full((n,), 5, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ctx.render_device(cam, st, want.data_ptr(), want_ids.data_ptr(), stream)
    torch.cuda.synchronize()
    want_counters, want_kernel = ctx.profile_counters()[1:7], ctx.kernel_name()
    free_before = torch.cuda.mem_get_info()[0]
    ctx.render_device_static(cam, st, got.data_ptr(), got_ids.data_ptr(), stream)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before, "device memory moved across the no-alloc call"
    got_counters = ctx.profile_counters()[1:7]          # paths, rays, nodes, triangle tests, sphere tests, scatters
    if ctx.kernel_name() == want_kernel:
        assert got_counters == want_counters
    else:
        # a tree in memory whose stack bound is above what a slot-tree block keeps in LDS: this entry point (no overflow stack)
        # walks the four-wide nodes the slot tree is made from - the same hits through slightly tighter boxes
        assert want_kernel.endswith(",11>") and ctx.kernel_name().endswith(",6>")
        assert [got_counters[i] for i in (0, 1, 5)] == [want_counters[i] for i in (0, 1, 5)]
        assert 0.99 * want_counters[2] < got_counters[2] <= want_counters[2]
    assert torch.equal(got.view(torch.int32), want.view(torch.int32)), "radiance differs from rtgpu_render_device"
    assert torch.equal(got_ids, want_ids)
    return got, got_ids


def test_static_scratch_entry_point(ra, oracle):
    import torch
    ctx = ra.Context(0)
    try:
        assert ra.lib.rtgpu_static_scratch_bytes() == 64 << 20
        scene, cam = ra.Scene.builtin(0, aspect=16 / 9)
        ctx.upload(scene)
        # first call: 1280x720, 16 spp in 4 chunks = 59 MB of chunk planes, nearly the whole array
        big = ra.default_settings(1280, 720, spp=16, max_depth=8, flags=ra.FLAG_STATS)
        assert big.chunk_samples == 4
        _pair(ra, ctx, cam, big)
        # second call, smaller input, other chunking (3 chunks over 5 spp): stale planes must not leak
        small = ra.default_settings(200, 120, spp=5, max_depth=8, chunk_samples=2, seed=3, flags=ra.FLAG_STATS)
        cam_s = ra.look_at((13, 2, 3), (0, 0, 0), vfov=20.0, aspect=200 / 120)
        got, got_ids = _pair(ra, ctx, cam_s, small)
        st_o = small.copy()
        st_o.flags = 0
        b, ib, _ = oracle.render(ctx.flat_scene().c, cam_s, st_o, threads=8)
        assert_bit_equal(got.cpu().numpy().reshape(120, 200, 4), b, "no-alloc frame vs oracle")
        assert np.array_equal(got_ids.cpu().numpy().view(np.uint32).reshape(120, 200), ib)
        # one chunk needs no planes at all: a frame of any size
        one = ra.default_settings(1920, 1080, spp=2, max_depth=8, chunk_samples=2)
        _pair(ra, ctx, ra.look_at((13, 2, 3), (0, 0, 0), vfov=20.0, aspect=16 / 9), one)
        # a large tree, two chunks: the allocating call of the pair walks its slot tree with the deeper stack entries in memory (mode 11),
        # this entry point the four-wide nodes with the whole stack in LDS (mode 6)
        mesh, cam_m = ra.Scene.builtin(1, param=6, aspect=16 / 9)
        ctx.upload(mesh)
        _pair(ra, ctx, cam_m, ra.default_settings(640, 360, spp=4, max_depth=8, flags=ra.FLAG_STATS))
        # a shallow tree in memory: both walk the slot tree (mode 12), which the allocating call has made
        grid = make_scene(ra, spheres=[(float(x) - 24.0, float(y) - 24.0, 0.0, 0.3, 0) for x in range(48) for y in range(48)],
                          materials=[(0, .6, .6, .6, 0)])
        ctx.upload(grid)
        _pair(ra, ctx, ra.look_at((3.0, -30.0, 14.0), (0, 0, 0), vfov=60.0, aspect=16 / 9),
              ra.default_settings(320, 180, spp=4, max_depth=6, flags=ra.FLAG_STATS))
        assert ctx.kernel_name().endswith(",12>")
        # beyond the array: refused with a message, nothing allocated, nothing launched
        too_big = ra.default_settings(1920, 1080, spp=16, max_depth=8)     # 4 chunks x 33 MB
        acc = torch.zeros((1920 * 1080, 4), dtype=torch.float32, device="cuda")
        with pytest.raises(ra.RtgpuError, match="static scratch"):
            ctx.render_device_static(cam, too_big, acc.data_ptr())
        profile = ra.default_settings(64, 64, spp=1, max_depth=2, flags=ra.FLAG_PROFILE)
        with pytest.raises(ra.RtgpuError, match="PROFILE"):
            ctx.render_device_static(cam, profile, acc.data_ptr())
        # a deep tree whose stack still fits LDS; one that does not is refused (no overflow stack here)
        sph = [(float(3.0 ** i), 0.0, 0.0, float(0.25 * 3.0 ** i), 0) for i in range(-16, 40)]
        ctx.upload(make_scene(ra, spheres=sph, materials=[(0, .5, .5, .5, 0)]))
        st = ra.default_settings(32, 32, spp=2, max_depth=4)
        cam_d = ra.look_at((-1e-9, 0.0, 0.0), (1, 0, 0), vfov=30.0, aspect=1.0)
        if ctx.flat_scene().c.max_depth + 2 <= 32:
            _pair(ra, ctx, cam_d, st)
        else:
            with pytest.raises(ra.RtgpuError, match="too deep"):
                ctx.render_device_static(cam_d, st, acc.data_ptr())
    finally:
        ctx.close()


def test_static_entry_point_under_graph_capture(ra):
    """The use include/rtgpu.h names first: the call captured into a HIP graph (nothing in it allocates, so the
    capture is legal) and the graph replayed - every replay writes the frame rtgpu_render_device writes."""
    import torch
    ctx = ra.Context(0)
    try:
        scene, cam = ra.Scene.builtin(0, aspect=16 / 9)
        ctx.upload(scene)
        w, h = 640, 360
        st = ra.default_settings(w, h, spp=4, max_depth=8)              # two chunks of two samples: planes + combine_kernel
        want = torch.zeros((w * h, 4), dtype=torch.float32, device="cuda")
        ctx.render_device(cam, st, want.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out = torch.zeros((w * h, 4), dtype=torch.float32, device="cuda")
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            ctx.render_device_static(cam, st, out.data_ptr(), None, side.cuda_stream)     # warm-up on the capture stream
        side.synchronize()
        assert torch.equal(out.view(torch.int32), want.view(torch.int32))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            ctx.render_device_static(cam, st, out.data_ptr(), None, side.cuda_stream)
        for k in range(3):
            out.fill_(float(k + 5))
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out.view(torch.int32), want.view(torch.int32)), f"replay {k}"
    finally:
        ctx.close()
