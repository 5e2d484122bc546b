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


def _row_costs(height):
    """a frame whose rows differ in cost the way the S-RTOW frame's do: cheap sky above, a heavy band at the horizon"""
    y = np.arange((height + 7) // 8, dtype=np.float64)
    return (1000 + 9000 * (y > 0.45 * len(y)) + 40000 * np.exp(-((y - 0.5 * len(y)) / 3.0) ** 2)).astype(np.uint64)


def _cost_plan_worker(rank, world, port, height, width, out_path):
    """As _plan_worker with the blocks dealt out by cost (rtmi_shard_plan through CostShardPlan)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import rtmi_loader
    pkg = rtmi_loader.load()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bc = torch.from_numpy(_row_costs(height).astype(np.int64)) if rank == 0 else torch.zeros((height + 7) // 8, dtype=torch.int64)
        dist.broadcast(bc, src=0)  # (rank 0's costs travel, as in bench.py)
        plan = pkg.CostShardPlan(height, 8, world, bc.numpy().astype(np.uint64))
        local = torch.full((plan.max_rows, width, 3), -1.0)
        cols = torch.arange(width, dtype=torch.float32)
        loc = 0
        for b in plan.blocks(rank):
            for y in range(int(b) * 8, min(height, int(b) * 8 + 8)):
                local[loc, :, 0] = float(y)
                local[loc, :, 1] = cols
                local[loc, :, 2] = float(rank)
                loc += 1
        assert loc == plan.rows(rank)
        frame = pkg.gather_frame(local, plan, rank)
        if rank == 0:
            np.save(out_path, frame.numpy())
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("height", [1080, 7, 64])
def test_world_8_cost_balanced_plan_gathers_to_scanline_order(tmp_path, height):
    """World 8 with the row blocks dealt out by cost: every row comes back once, in scanline order, from the rank the plan names;
    the ranks hold the same number of blocks (+-1) and their costs differ by less than the costliest block."""
    import rtmi_loader
    pkg = rtmi_loader.load()
    world, width = 8, 24
    out = str(tmp_path / "frame.npy")
    mp.spawn(_cost_plan_worker, args=(world, _free_port(), height, width, out), nprocs=world, join=True)
    got = np.load(out)
    assert got.shape == (height, width, 3)
    assert np.array_equal(got[:, 0, 0], np.arange(height, dtype=np.float32))
    assert np.array_equal(got[0, :, 1], np.arange(width, dtype=np.float32))
    cost = _row_costs(height)
    plan = pkg.CostShardPlan(height, 8, world, cost)
    assert np.array_equal(got[:, 0, 2], plan.rank_of_block[np.arange(height) // 8].astype(np.float32))
    counts = [len(plan.blocks(r)) for r in range(world)]
    loads = [int(cost[plan.blocks(r)].sum()) for r in range(world)]
    assert max(counts) - min(counts) <= 1
    if height == 1080:
        mod = [int(cost[np.arange(len(cost)) % world == r].sum()) for r in range(world)]
        assert max(loads) - min(loads) < int(cost.max()) and max(loads) < max(mod)  # better than block b -> rank b mod 8
    # without costs the plan is block b -> rank b mod N, as RowShardPlan
    blind, old = pkg.CostShardPlan(height, 8, world), pkg.RowShardPlan(height, 8, world)
    assert np.array_equal(blind.index, old.index) and blind.max_rows == old.max_rows


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
