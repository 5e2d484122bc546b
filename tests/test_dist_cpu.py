"""world_size-2 (and 3) gloo tests of the multi-GPU plumbing on CPU: interleaved row-block sharding, one gather of
the per-rank slices to rank 0, de-interleave to scanline order.  The per-rank renderer is the oracle here (a stand-in
so the test runs without GPUs); on the GPU box tests/test_gpu_parity.py drives the same plan through the HIP path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, height_kw, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import rtmi_loader
    from oracle import binding as ob
    pkg = rtmi_loader.load()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cam = ob.camera_setup(ob.camera_params(**height_kw))
        objs, mats = ob.make_world_spheres(12345)
        W, H = cam.img_width, cam.img_height
        plan = pkg.RowShardPlan(H, 8, world)
        y_first, n_blocks, rows = plan.shard(rank)
        local = torch.zeros((plan.max_rows, W, 3), dtype=torch.float32)
        loc = 0
        for k in range(n_blocks):
            y0 = y_first + k * world * 8
            y1 = min(H, y0 + 8)
            rgb, _ = ob.render_rect_counter(cam, objs, mats, 5, 0, y0, W, y1)
            local[loc:loc + (y1 - y0)] = torch.from_numpy(rgb)
            loc += y1 - y0
        assert loc == rows
        frame = pkg.gather_frame(local, plan, rank)
        if rank == 0:
            np.save(out_path, frame.numpy())
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,width", [(2, 64), (3, 100)])
def test_sharded_render_gathers_to_the_full_frame(tmp_path, ob, world, width):
    kw = dict(image_width=width, samples_per_pixel=2, max_depth=6)
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), kw, out), nprocs=world, join=True)
    cam = ob.camera_setup(ob.camera_params(**kw))
    objs, mats = ob.make_world_spheres(12345)
    want, _ = ob.render_rect_counter(cam, objs, mats, 5, 0, 0, cam.img_width, cam.img_height, nthreads=4)
    got = np.load(out)
    assert got.shape == want.shape and got.tobytes() == want.tobytes()


def _plan_worker(rank, world, port, height, width, out_path):
    """As _worker, with a synthetic renderer (pixel value = its absolute row and column): the shard plan, the rank-major
    gather and the de-interleave at the driver's own geometry, where the oracle would take minutes per rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import rtmi_loader
    pkg = rtmi_loader.load()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = pkg.RowShardPlan(height, 8, world)
        y_first, n_blocks, rows = plan.shard(rank)
        local = torch.full((plan.max_rows, width, 3), -1.0)
        cols = torch.arange(width, dtype=torch.float32)
        loc = 0
        for k in range(n_blocks):
            y0 = y_first + k * world * 8
            for y in range(y0, min(height, y0 + 8)):
                local[loc, :, 0] = float(y)
                local[loc, :, 1] = cols
                local[loc, :, 2] = float(rank)
                loc += 1
        assert loc == rows
        frame = pkg.gather_frame(local, plan, rank)
        if rank == 0:
            np.save(out_path, frame.numpy())
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("height", [1080, 7, 64])
def test_world_8_shard_plan_gathers_to_scanline_order(tmp_path, height):
    """World 8 (the driver's scaling run): 1080 rows = 135 blocks, ragged (ranks 0-6: 17 blocks, rank 7: 16); 7 rows: ranks 1-7
    render nothing; 64 rows: one block each."""
    world, width = 8, 24
    out = str(tmp_path / "frame.npy")
    mp.spawn(_plan_worker, args=(world, _free_port(), height, width, out), nprocs=world, join=True)
    got = np.load(out)
    assert got.shape == (height, width, 3)
    assert np.array_equal(got[:, 0, 0], np.arange(height, dtype=np.float32))  # scanline order
    assert np.array_equal(got[0, :, 1], np.arange(width, dtype=np.float32))
    assert np.array_equal(got[:, 0, 2], (np.arange(height) // 8 % world).astype(np.float32))  # block b came from rank b mod 8
