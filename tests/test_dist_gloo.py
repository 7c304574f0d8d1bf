"""N>1 path on CPU: two processes, gloo.  Each rank's compact tile buffer (here
cut out of the ORACLE's render of that rank's tiles, standing in for the GPU
kernel) goes through the same gather + de-interleave plumbing bench.py uses
(rt_amd.dist) and must reproduce the single-process frame bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, ts, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import rt_amd as ra
    from oracle import Oracle
    from rt_amd import dist as rdist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene, cam = ra.Scene.builtin(0, aspect=w / h)
    fs = ra.FlatScene.build(scene)
    st = ra.default_settings(w, h, spp=2, max_depth=6, tile_size=ts, rank=rank, world_size=world)
    img, _, stats = Oracle().render(fs.c, cam, st)                   # only this rank's tiles are written
    local = torch.from_numpy(rdist.compact_from_image(img, w, h, ts, rank, world))
    assert local.shape[0] == rdist.padded_pixels(w, h, ts, world)
    gathered = rdist.gather_compact(local, rank, world, dst=0)
    rays = torch.tensor([stats["rays"]], dtype=torch.int64)
    dist.all_reduce(rays)
    if rank == 0:
        full = rdist.untile_host(gathered.numpy(), w, h, ts, world)
        np.save(out_path, full)
        np.save(out_path + ".rays.npy", rays.numpy())
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_two_rank_gather_reproduces_the_frame(tmp_path, world):
    import torch.multiprocessing as mp

    import rt_amd as ra
    from oracle import Oracle

    w, h, ts = 100, 60, 16
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, ts, out), nprocs=world, join=True)
    got = np.load(out)
    scene, cam = ra.Scene.builtin(0, aspect=w / h)
    fs = ra.FlatScene.build(scene)
    want, _, stats = Oracle().render(fs.c, cam, ra.default_settings(w, h, spp=2, max_depth=6, tile_size=ts))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert int(np.load(out + ".rays.npy")[0]) == stats["rays"]
