"""The kernel's experiment switches (profiles/r02/experiments.txt) change HOW the trace loop schedules its
work, never what a path computes: every switch gives the frame and the counters of the product kernel, bit for
bit.  The switches are read at rtgpu_create, so each case renders through a context of its own."""
import numpy as np
import pytest

from util import assert_bit_equal

pytestmark = pytest.mark.gpu

# switch -> kernel mode expected on (S0: the whole scene fits LDS, a mesh with a four-wide tree in memory)
CASES = [
    ({}, ((7, 8), 6)),                                              # (7 or 8: whichever the build makes the default)
    ({"RTGPU_NO_TINY": "1"}, (2, 6)),                               # the tree in LDS, spheres behind it
    ({"RTGPU_NO_SPHERE_CACHE": "1"}, (2, 6)),                       # the tree in LDS, spheres through L1
    ({"RTGPU_RAYS2": "1"}, (8, 6)),                                 # two rays per lane in lock step (scenes in LDS)
    ({"RTGPU_RAYS2": "0"}, (7, 6)),                                 # one ray per lane
    ({"RTGPU_NODE4_PAD": "1"}, (7, 10)),                            # four-wide nodes in 128-byte slots (what huge trees run on)
    ({"RTGPU_NODE4_PAD": "1", "RTGPU_STACK_LDS": "14"}, (7, 9)),    # ... with the deeper stack entries in memory
    ({"RTGPU_WIDE_CACHE": "1"}, (7, 5)),                            # experiment A: top of the four-wide tree in LDS
    ({"RTGPU_REFILL": "0"}, (7, 6)),                                # plain while-while traversal rounds
    ({"RTGPU_NO_MATS_CACHE": "1"}, (7, 6)),                         # trees in memory: materials through L1, not LDS
]


def _frames(ra, ctx):
    out = []
    for kind, param, w, h, spp in ((0, 0, 160, 90, 4), (1, 5, 128, 72, 3)):
        scene, cam = ra.Scene.builtin(kind, param=param, aspect=w / h)
        ctx.upload(scene)
        st = ra.default_settings(w, h, spp=spp, max_depth=8, seed=11, flags=ra.FLAG_STATS)
        a, ia, sa = ctx.render(cam, st)
        out.append((a, ia, sa.as_dict(), int(ctx.kernel_name().rstrip(">").split(",")[-1])))
    return out


@pytest.mark.parametrize("env,modes", CASES[1:], ids=lambda v: "+".join(v) if isinstance(v, dict) else None)
def test_experiment_switches_leave_the_frame_alone(ra, gpu_ctx, monkeypatch, env, modes):
    want = _frames(ra, gpu_ctx)
    assert want[0][3] in CASES[0][1][0] and want[1][3] == CASES[0][1][1]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ctx = ra.Context(0)
    try:
        got = _frames(ra, ctx)
    finally:
        ctx.close()
    if modes[0] == 7 and "RTGPU_RAYS2" not in env:
        modes = (want[0][3], modes[1])                           # the switch leaves the choice between 7 and 8 alone
    assert tuple(f[3] for f in got) == modes
    for (a, ia, sa, _), (b, ib, sb, _) in zip(got, want):
        assert_bit_equal(a, b, f"{env}")
        assert np.array_equal(ia, ib) and sa == sb
