"""The N > 1 path on CPU: world_size-2 and -3 `gloo` process groups. Each rank renders ITS
row tile (with the oracle standing in for the GPU kernels — this is a test), the product's
`syzygy_amd.rowtile` does the partition + the single gather, rank 0 composes and compares with
the oracle's whole-frame render: bit-exact."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, block_rows, out_path, W, H):
    import torch.distributed as dist

    from oracle import binding as ob
    from syzygy_amd import rowtile
    from tests import util

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        inp = util.Inputs(W, H, elevation_degrees=25.0, spots=5)
        tile = rowtile.make_tile(H, rank, world, block_rows)
        srows = rowtile.stride_rows(H, world, block_rows)
        tl = ob.transmittance_lut(inp.atm, 64, 16)
        sl = ob.skyview_lut(inp.atm, inp.cam, tl, 64, 32)
        local = np.zeros((srows, W, 4), np.uint16)
        if tile.local_rows:
            f = ob.HostFrame(W, tile.local_rows)
            ob.gbuffer_fill(f, inp.rect, tile, inp.cam, inp.synthetic.fill)
            ob.lights(f, inp.rect, tile, None, inp.cam, inp.dirs, 2, 1, inp.spots, 5)
            ob.composite(f, inp.rect, tile, None, inp.atm, inp.cam, inp.dirs, 0, tl, sl)
            local[: tile.local_rows] = f.color
        gathered = rowtile.gather_tiles(torch.from_numpy(local.view(np.int16)), rank, world)
        if rank == 0:
            g = gathered.numpy().view(np.uint16)
            image = np.zeros((H, W, 4), np.uint16)
            seen = np.zeros(H, bool)
            for r in range(world):
                rows = rowtile.global_rows(H, r, world, block_rows)
                assert len(rows) == rowtile.local_rows(H, r, world, block_rows)
                image[rows] = g[r, : len(rows)]
                assert not seen[rows].any()
                seen[rows] = True
            assert seen.all()
            full = ob.HostFrame(W, H)
            ob.gbuffer_fill(full, inp.rect, None, inp.cam, inp.synthetic.fill)
            ob.lights(full, inp.rect, None, None, inp.cam, inp.dirs, 2, 1, inp.spots, 5)
            ob.composite(full, inp.rect, None, None, inp.atm, inp.cam, inp.dirs, 0, tl, sl)
            np.save(out_path, np.array([int((image == full.color).all()), int(image.any())]))
        else:
            assert gathered is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


# (8 ranks: the configuration north_star names; 17 blocks of 8 rows, so the shares are unequal - 3, 2, 2, ... blocks)
@pytest.mark.parametrize("world,block_rows,size", [(2, 8, (48, 40)), (3, 4, (40, 30)), (2, 16, (32, 20)), (8, 8, (40, 136))])
def test_rowtile_gather_compose_gloo(world, block_rows, size):
    W, H = size
    with tempfile.TemporaryDirectory() as tmp:
        out_path = os.path.join(tmp, "result.npy")
        mp.spawn(_worker, args=(world, _free_port(), block_rows, out_path, W, H), nprocs=world, join=True)
        ok, nonzero = np.load(out_path)
    assert ok == 1 and nonzero == 1


def _lut_worker(rank, world, port, out_path):
    import torch.distributed as dist

    from syzygy_amd import rowtile

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H, W = 16, 8
        want = torch.arange(H * W * 4, dtype=torch.float32).reshape(H, W, 4)
        lut = torch.full((H, W, 4), -1.0)
        b, e = rowtile.lut_rows(H, rank, world)
        lut[b:e] = want[b:e]  # this rank's slice
        rowtile.allgather_lut(lut, rank, world)
        np.save(out_path + f".{rank}.npy", np.array([int(torch.equal(lut, want))]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_lut_slices_allgather_gloo(world):
    with tempfile.TemporaryDirectory() as tmp:
        out_path = os.path.join(tmp, "r")
        mp.spawn(_lut_worker, args=(world, _free_port(), out_path), nprocs=world, join=True)
        for r in range(world):
            assert np.load(out_path + f".{r}.npy")[0] == 1


def test_lut_rows_partition():
    from syzygy_amd import rowtile

    assert [rowtile.lut_rows(1024, r, 8) for r in range(8)] == [(r * 128, (r + 1) * 128) for r in range(8)]
    assert rowtile.lut_rows(1024, 0, 1) == (0, 1024)
    with pytest.raises(ValueError):
        rowtile.lut_rows(1024, 0, 3)


def test_global_rows_cover_the_frame_once():
    from syzygy_amd import rowtile

    for H in (1, 17, 100, 4320):
        for n in (1, 2, 3, 8):
            for b in (1, 8, 16):
                rows = np.concatenate([rowtile.global_rows(H, r, n, b) for r in range(n)])
                assert sorted(rows.tolist()) == list(range(H))
                assert rowtile.stride_rows(H, n, b) == max(rowtile.local_rows(H, r, n, b) for r in range(n))


def test_compose_refuses_cpu_tensors():
    from syzygy_amd import rowtile

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rowtile.compose(torch.zeros((2, 4, 8, 4), dtype=torch.int16), 8, 2)
