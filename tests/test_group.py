"""The native multi-GPU path (rtgpu_group_*) and the resident progressive frame
(rtgpu_render_resolved), through the C ABI on the one GPU of the box.

A one-device group runs the real pipeline (worker thread, two streams, compact tiles, untile,
resolve); a group that lists device 0 several times rehearses N ranks on one GPU (RCCL refuses
a device twice, so the tile buffers then move by device copies; the RCCL exchange between DIFFERENT
GPUs needs N real GPUs and has run nowhere yet - what a one-GPU box can run of it is the one-rank
communicator of test_one_rank_communicator_moves_the_tiles_through_rccl).
Every frame must equal the one-context frame / the oracle bit for bit."""
import numpy as np
import pytest

from util import assert_bit_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices,tile", [([0], 64), ([0, 0], 16), ([0, 0, 0], 64), ([0] * 8, 16)])
def test_group_frame_equals_the_single_context_frame(ra, oracle, gpu_ctx, devices, tile):
    w, h = 400, 240
    scene, cam = ra.Scene.builtin(0, aspect=w / h)
    gpu_ctx.upload(scene)
    st = ra.default_settings(w, h, spp=4, max_depth=8, tile_size=tile, flags=ra.FLAG_STATS)
    want, _, want_stats = gpu_ctx.render(cam, st)
    g = ra.Group(devices)
    try:
        assert g.size == len(devices)
        g.upload(scene)
        acc, rgba, stats = g.render(cam, st)
        assert_bit_equal(acc, want, f"group of {len(devices)}")
        assert np.array_equal(rgba, oracle.resolve(want))
        assert stats.as_dict() == want_stats.as_dict()
        assert g.kernel_ms(0) > 0
        # progressive: the group keeps the per-GPU tile sums between calls
        st2 = ra.default_settings(w, h, spp=4, max_depth=8, tile_size=tile, sample_offset=4, flags=ra.FLAG_ACCUMULATE)
        acc2, _, _ = g.render(cam, st2)
        both, _, _ = gpu_ctx.render(cam, st2, accum=want.copy())
        assert_bit_equal(acc2, both, "group, accumulated")
        # two frames in flight: begin k+1 before ending k; a third is refused
        sa = ra.default_settings(w, h, spp=2, max_depth=8, tile_size=tile, seed=11)
        sb = ra.default_settings(w, h, spp=3, max_depth=8, tile_size=tile, seed=12)
        g.render_begin(cam, sa)
        g.render_begin(cam, sb)
        with pytest.raises(ra.RtgpuError, match="in flight"):
            g.render_begin(cam, sa)
        fa, _, _ = g.render_end()
        fb, _, _ = g.render_end()
        assert_bit_equal(fa, gpu_ctx.render(cam, sa)[0], "pipelined frame A")
        assert_bit_equal(fb, gpu_ctx.render(cam, sb)[0], "pipelined frame B")
        with pytest.raises(ra.RtgpuError, match="no frame"):
            g.render_end()
        # the GPU-built tree is the same on every GPU of the group
        g.upload(scene, builder="lbvh")
        gpu_ctx.upload(scene, builder="lbvh")
        acc3, _, st3 = g.render(cam, st)
        want3, _, want_st3 = gpu_ctx.render(cam, st)
        assert_bit_equal(acc3, want3, "group, LBVH")
        assert st3.as_dict() == want_st3.as_dict()
    finally:
        g.close()


def test_group_large_tree(ra, oracle, gpu_ctx):
    """A tree with four-wide nodes replicated from ONE host build to four ranks."""
    w, h = 320, 180
    scene, cam = ra.Scene.builtin(1, param=6, aspect=w / h)
    st = ra.default_settings(w, h, spp=2, max_depth=8, tile_size=32, flags=ra.FLAG_STATS)
    g = ra.Group([0, 0, 0, 0])
    try:
        g.upload(scene)
        acc, rgba, stats = g.render(cam, st)
    finally:
        g.close()
    gpu_ctx.upload(scene)
    st_o = st.copy()
    st_o.flags = 0
    b, _, sb = oracle.render(gpu_ctx.flat_scene().c, cam, st_o, threads=16)
    assert_bit_equal(acc, b, "group vs oracle")
    assert stats.as_dict() == sb and np.array_equal(rgba, oracle.resolve(b))


def test_group_errors(ra):
    with pytest.raises(ra.RtgpuError):
        ra.Group([99])
    g = ra.Group([0])
    try:
        scene, cam = ra.Scene.builtin(0)
        with pytest.raises(ra.RtgpuError, match="no scene"):
            g.render(cam, ra.default_settings(64, 64))
        g.upload(scene)
        with pytest.raises(ra.RtgpuError):
            g.render(cam, ra.default_settings(64, 64, tile_size=48))
        acc, _, _ = g.render(cam, ra.default_settings(64, 64))      # and the group still works
        assert np.all(acc[..., 3] == 1.0)
    finally:
        g.close()


def test_render_resolved_keeps_the_accumulation_on_the_gpu(ra, oracle, gpu_ctx):
    """rtgpu_render_resolved: progressive frames whose float sums never cross PCIe; every
    resolved frame and the resident sums equal the oracle's bit for bit."""
    w, h = 320, 180
    scene, cam = ra.Scene.builtin(0, aspect=w / h)
    gpu_ctx.upload(scene)
    fs = gpu_ctx.flat_scene()
    acc_o = np.zeros((h, w, 4), np.float32)
    for frame in range(4):
        st = ra.default_settings(w, h, spp=3, max_depth=8, sample_offset=3 * frame,
                                 flags=ra.FLAG_ACCUMULATE)      # the first call has nothing to continue: starts the sum
        rgba, stats = gpu_ctx.render_resolved(cam, st, want_stats=True)
        st_o = st.copy()
        st_o.flags = ra.FLAG_ACCUMULATE if frame else 0
        acc_o, _, so = oracle.render(fs.c, cam, st_o, accum=acc_o, threads=16)
        assert np.array_equal(rgba, oracle.resolve(acc_o)), f"frame {frame}"
        assert stats.rays == so["rays"]
    assert_bit_equal(gpu_ctx.resident_accum(w, h), acc_o, "resident sums")
    # without the flag, after a reset and after a size change the sum starts again
    st = ra.default_settings(w, h, spp=2, max_depth=8, seed=9)
    one, _, _ = oracle.render(fs.c, cam, st, threads=16)
    for how in ("no flag", "reset", "size"):
        if how == "reset":
            gpu_ctx.resident_reset()
        st_g = st.copy()
        st_g.flags = 0 if how == "no flag" else ra.FLAG_ACCUMULATE
        if how == "size":
            gpu_ctx.render_resolved(cam, ra.default_settings(64, 32, spp=1, flags=ra.FLAG_ACCUMULATE))
        rgba, _ = gpu_ctx.render_resolved(cam, st_g)
        assert np.array_equal(rgba, oracle.resolve(one)), how
    with pytest.raises(ra.RtgpuError, match="COMPACT"):
        gpu_ctx.render_resolved(cam, ra.default_settings(w, h, flags=ra.FLAG_COMPACT))
    gpu_ctx.upload(scene)                                        # an upload drops the resident sum
    with pytest.raises(ra.RtgpuError, match="resident"):
        gpu_ctx.resident_accum(w, h)


@pytest.mark.parametrize("device_side", [False, True])
def test_refit_restarts_the_resident_accumulation(ra, oracle, gpu_ctx, device_side):
    """include/rtgpu.h: after a refit the rtgpu_render_resolved sum starts again - a progressive loop
    that keeps RTGPU_FLAG_ACCUMULATE across a refit must not add samples of the new geometry onto the
    sum of the old one (ghosting)."""
    import torch
    w, h = 160, 96
    scene, cam = ra.Scene.builtin(0, aspect=w / h)
    gpu_ctx.upload(scene)
    st = ra.default_settings(w, h, spp=2, max_depth=6, flags=ra.FLAG_ACCUMULATE)
    gpu_ctx.render_resolved(cam, st)
    gpu_ctx.render_resolved(cam, st)                             # two frames of the old geometry in the sum
    moved = ra.Scene(scene.spheres.copy(), scene.triangles.copy(), scene.materials.copy())
    moved.spheres["center"][1:, 1] += 0.35                       # every small sphere hops
    if device_side:
        ds = torch.from_numpy(moved.spheres.view(np.uint8).copy()).cuda()
        dt = torch.from_numpy(moved.triangles.view(np.uint8).copy()).cuda()
        gpu_ctx.refit_device(ds.data_ptr(), len(moved.spheres), dt.data_ptr(), len(moved.triangles))
    else:
        gpu_ctx.refit(moved)
    with pytest.raises(ra.RtgpuError, match="resident"):
        gpu_ctx.resident_accum(w, h)
    rgba, _ = gpu_ctx.render_resolved(cam, st)                   # flag still set: must be a fresh frame
    st_o = st.copy()
    st_o.flags = 0
    fresh, _, _ = oracle.render(gpu_ctx.flat_scene().c, cam, st_o, threads=8)
    assert np.array_equal(rgba, oracle.resolve(fresh))
    assert_bit_equal(gpu_ctx.resident_accum(w, h), fresh, "sum after the refit")


def test_group_survives_a_refused_frame_and_breaks_cleanly(ra, gpu_ctx, monkeypatch):
    """A frame every GPU refuses before its trace leaves nothing queued: the next frame works.  A failure in
    the exchange step (injected) leaves the group unusable, says so, and rtgpu_group_destroy returns."""
    w, h = 128, 64
    scene, cam = ra.Scene.builtin(0, aspect=w / h)
    gpu_ctx.upload(scene)
    st = ra.default_settings(w, h, spp=2, max_depth=4, tile_size=16)
    want, _, _ = gpu_ctx.render(cam, st)
    g = ra.Group([0, 0, 0])
    try:
        g.upload(scene)
        bad = st.copy()
        bad.spp = 0
        for _ in range(3):                                       # more refusals than frames may be in flight
            with pytest.raises(ra.RtgpuError, match="spp"):
                g.render_begin(cam, bad)
        acc, _, _ = g.render(cam, st)
        assert_bit_equal(acc, want, "frame after refused frames")
    finally:
        g.close()
    monkeypatch.setenv("RTGPU_GROUP_DEBUG_FAIL_EXCHANGE", "1")
    g = ra.Group([0, 0, 0])
    try:
        g.upload(scene)
        with pytest.raises(ra.RtgpuError, match="injected.*unusable"):
            g.render_begin(cam, st)
        with pytest.raises(ra.RtgpuError, match="unusable"):
            g.render_begin(cam, st)
        with pytest.raises(ra.RtgpuError, match="unusable"):
            g.upload(scene)
    finally:
        g.close()                                                # must not hang
    monkeypatch.delenv("RTGPU_GROUP_DEBUG_FAIL_EXCHANGE")
    acc, _, _ = gpu_ctx.render(cam, st)                          # the GPU is fine afterwards
    assert_bit_equal(acc, want, "context after a broken group")


def test_one_rank_communicator_moves_the_tiles_through_rccl(ra, gpu_ctx, monkeypatch):
    """RTGPU_GROUP_SELF_RCCL=1: a one-device group makes a one-rank communicator (ncclCommInitAll) and its
    tile buffer travels by a grouped ncclSend / ncclRecv to itself on the exchange stream.  This is the part
    of the N-GPU wire a one-GPU box can run: dlopen of librccl, the symbol table, the call signatures and the
    stream ordering around the exchange.  Two frames in flight."""
    monkeypatch.setenv("RTGPU_GROUP_SELF_RCCL", "1")
    w, h = 200, 120
    scene, cam = ra.Scene.builtin(0, aspect=w / h)
    gpu_ctx.upload(scene)
    sa = ra.default_settings(w, h, spp=3, max_depth=8, tile_size=32, seed=5, flags=ra.FLAG_STATS)
    sb = ra.default_settings(w, h, spp=2, max_depth=8, tile_size=32, seed=6, flags=ra.FLAG_STATS)
    g = ra.Group([0])
    try:
        g.upload(scene)
        g.render_begin(cam, sa)
        g.render_begin(cam, sb)
        fa, _, st_a = g.render_end()
        fb, _, st_b = g.render_end()
        wa, _, wst_a = gpu_ctx.render(cam, sa)
        wb, _, wst_b = gpu_ctx.render(cam, sb)
        assert_bit_equal(fa, wa, "frame A through RCCL")
        assert_bit_equal(fb, wb, "frame B through RCCL")
        assert st_a.as_dict() == wst_a.as_dict() and st_b.as_dict() == wst_b.as_dict()
    finally:
        g.close()
