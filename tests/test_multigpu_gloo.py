"""World-size-2 (and 4) CPU rehearsal of the strip partition + all-gather used for multi-GPU runs.
The oracle stands in for the per-rank renderer here (this is a test of the partition / gather logic,
not of the HIP path)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ar = importlib.import_module("ascii-renderer_amd")
    mg = importlib.import_module("ascii-renderer_amd.multigpu")
    from oracle import oracle as orc
    cols, rows = 96, 48
    sc = ar.scenes.icosphere_scene(cols=cols, rows=rows, subdiv=2)
    row0, row1 = mg.strip_bounds(rows, world, rank)
    rgba, _d, _i = orc.render(sc["vbo"], sc["idx"], sc["proj"], sc["view"], sc["lights"], cols, rows)
    full = orc.quantize(rgba, cols, rows, ar.scenes.DEFAULT_RAMP, 1, 2, 12)
    local = torch.from_numpy(full[row0:row1].copy())          # this rank's band only
    got = mg.all_gather_strips(local)
    ok = np.array_equal(got.numpy(), full)
    # batched variant: three different frames (rolled copies of the grid), one collective per two frames
    frames = [np.roll(full, 7 * k, axis=1) for k in range(3)]
    sb = mg.StripBatcher(row1 - row0, cols, 2, "cpu")
    sb.push(torch.from_numpy(frames[0][row0:row1].copy()))
    ok = ok and sb.gathered == 0
    sb.push(torch.from_numpy(frames[1][row0:row1].copy()))
    ok = ok and sb.gathered == 2 and np.array_equal(sb.frame(0).numpy(), frames[0]) and np.array_equal(sb.frame(1).numpy(), frames[1])
    sb.push(torch.from_numpy(frames[2][row0:row1].copy()))
    sb.flush()
    ok = ok and sb.gathered == 1 and np.array_equal(sb.frame(0).numpy(), frames[2])
    # ring rotation and the order of the hand-over callbacks: with two rings "frames wait for earlier gathers" comes
    # before the gather of the batch just finished, with one ring after it; gathered frames stay readable while the
    # other ring fills
    for rings, want in ((2, ["release", "acquire"]), (1, ["acquire", "release"])):
        log = []
        sb = mg.StripBatcher(row1 - row0, cols, 1, "cpu", before_gather=lambda: log.append("acquire"),
                             after_gather=lambda: log.append("release"), rings=rings)
        for k in range(3):
            ptr = sb.slot_ptr()
            ok = ok and ptr == sb.ring.data_ptr() + (k % rings) * (row1 - row0) * cols
            sb.ring[sb.cur, 0].copy_(torch.from_numpy(frames[k][row0:row1].copy()))
            mark = len(log)
            sb.advance()
            ok = ok and log[mark:] == want
            ok = ok and np.array_equal(sb.frame(0).numpy(), frames[k])
    # a frame whose rows do not divide over the ranks: bands differ by a row, slots are as tall as the tallest
    rows_u = 45 if world == 2 else 46
    grid = np.random.default_rng(11).integers(32, 127, size=(3, rows_u, cols), dtype=np.uint8)
    r0, r1 = mg.strip_bounds(rows_u, world, rank)
    sb = mg.StripBatcher(mg.band_height(rows_u, world), cols, 2, "cpu", rows=rows_u)
    for k in range(2):
        sb.push(torch.from_numpy(grid[k, r0:r1].copy()))
    ok = ok and sb.gathered == 2 and all(np.array_equal(sb.frame(k).numpy(), grid[k]) for k in range(2))
    sb.push(torch.from_numpy(grid[2, r0:r1].copy()))
    sb.flush()
    ok = ok and sb.gathered == 1 and np.array_equal(sb.frame(0).numpy(), grid[2])
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        with open(tmp, "w") as fh:
            fh.write(str(int(flag.item())))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_strip_all_gather_reassembles_the_frame(world, tmp_path):
    out = str(tmp_path / "ok.txt")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert open(out).read() == "1"


def test_strip_bounds(pkg):
    mg = importlib.import_module("ascii-renderer_amd.multigpu")
    assert [mg.strip_bounds(1080, 8, r) for r in range(8)] == [(135 * r, 135 * (r + 1)) for r in range(8)]
    assert mg.strip_bounds(2160, 1, 0) == (0, 2160)
    assert [mg.strip_bounds(1080, 7, r) for r in range(7)] == [(0, 155), (155, 310), (310, 464), (464, 618), (618, 772), (772, 926), (926, 1080)]
    with pytest.raises(ValueError):
        mg.strip_bounds(1080, 2, 2)


def test_strips_of_any_row_count_tile_the_frame():
    """The partition arithmetic, Python twin against the library's ar_strip_rows (no GPU involved): strips are contiguous,
    cover every row once, differ by at most one row - also where the rows do not divide (1080 over 7, 36 over 8) and
    where there are more strips than rows."""
    sys.path.insert(0, ROOT)
    ar = importlib.import_module("ascii-renderer_amd")
    mg = importlib.import_module("ascii-renderer_amd.multigpu")
    for rows, world in [(1080, 7), (36, 8), (1080, 8), (2160, 8), (45, 2), (46, 4), (3, 8), (1, 1), (0, 4), (1081, 64)]:
        edge = 0
        heights = []
        for rank in range(world):
            r0, r1 = mg.strip_bounds(rows, world, rank)
            assert (r0, r1) == ar.strip_rows(rows, world, rank), (rows, world, rank)
            assert r0 == edge and r1 >= r0
            heights.append(r1 - r0)
            edge = r1
        assert edge == rows and max(heights) - min(heights) <= 1 and max(heights) == mg.band_height(rows, world)
        assert heights == sorted(heights, reverse=True)          # the taller strips come first
    with pytest.raises(ValueError):
        ar.strip_rows(10, 0, 0)
    with pytest.raises(ValueError):
        mg.strip_bounds(10, 4, 4)
