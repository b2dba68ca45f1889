"""CPU: strip planning and the neighbour halo exchange (gloo, world size 3)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_plan_strips_cover_scene():
    from jspsr_amd.tiling import plan_strips
    for H, world, halo in ((4096, 8, 128), (1024, 4, 128), (512, 3, 64), (256, 2, 128)):
        s = plan_strips(H, world, halo)
        assert s[0].y0 == 0 and s[-1].y1 == H
        for a, b in zip(s[:-1], s[1:]):
            assert a.y1 == b.y0
        for t in s:
            assert 0 <= t.ty0 <= t.y0 and t.y1 <= t.ty1 <= H and (t.ty1 - t.ty0) % 8 == 0
            assert len({x.ty1 - x.ty0 for x in s}) == 1            # equal windows
            assert t.y0 == 0 or t.y0 - t.ty0 >= min(halo, t.y0)    # halo towards every interior edge
            assert t.y1 == H or t.ty1 - t.y1 >= min(halo, H - t.y1)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jspsr_amd.tiling import exchange_halo
    scene = torch.arange(2 * 48 * 5, dtype=torch.float32).reshape(1, 2, 48, 5)
    mine = scene[:, :, rank * 16:(rank + 1) * 16].clone()
    got = exchange_halo(mine, 8)
    lo, hi = max(0, rank * 16 - 8), min(48, (rank + 1) * 16 + 8)
    q.put((rank, torch.equal(got, scene[:, :, lo:hi])))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_halo_gloo_world3():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _worker_window(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jspsr_amd.tiling import exchange_window, plan_strips
    H, halo = 96, 16
    strips = plan_strips(H, world, halo)          # clamped equal windows: border ranks need 2*halo rows from one side
    scene = torch.arange(2 * H * 5, dtype=torch.float32).reshape(1, 2, H, 5)
    s = strips[rank]
    got = exchange_window(scene[:, :, s.y0:s.y1].clone(), strips)
    q.put((rank, bool(torch.equal(got, scene[:, :, s.ty0:s.ty1])), (s.y0 - s.ty0, s.ty1 - s.y1)))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_window_follows_the_strip_plan_gloo_world3():
    """The composition the sharded forward uses: each rank owns only rows [y0,y1) and assembles the window
    [ty0,ty1) of plan_strips (asymmetric at the scene borders) from its neighbours."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_window, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == (0, 32) and res[1][2] == (16, 16) and res[2][2] == (32, 0)   # 2*halo towards the only neighbour


def _worker_window8(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jspsr_amd.tiling import exchange_window, plan_strips
    H, halo, Wd = 4096, 128, 6                     # BASELINE config 5's scene height and halo over its 8 ranks (narrow: row logic only)
    strips = plan_strips(H, world, halo)
    s = strips[rank]
    rows = torch.arange(s.y0, s.y1, dtype=torch.float32).view(1, 1, -1, 1)
    mine = torch.cat([rows * 8 + c + torch.arange(Wd, dtype=torch.float32).view(1, 1, 1, Wd) / 16 for c in range(3)], 1)   # value names (row, channel, column)
    got = exchange_window(mine, strips)
    wr = torch.arange(s.ty0, s.ty1, dtype=torch.float32).view(1, 1, -1, 1)
    want = torch.cat([wr * 8 + c + torch.arange(Wd, dtype=torch.float32).view(1, 1, 1, Wd) / 16 for c in range(3)], 1)
    q.put((rank, bool(torch.equal(got, want)), (s.y0, s.y1, s.ty0, s.ty1)))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_window_world8_at_4096_rows():
    """the strip plan and the neighbour exchange at config 5's own geometry -- 4096 rows, 8 ranks,
    128-row halos: 512 interior rows each, 768-row windows, the two border ranks fetching 256 rows from their one neighbour."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_window8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    for r, _, (y0, y1, ty0, ty1) in res:
        assert (y0, y1) == (512 * r, 512 * (r + 1)) and ty1 - ty0 == 768
        assert (ty0, ty1) == ((0, 768) if r == 0 else (4096 - 768, 4096) if r == 7 else (y0 - 128, y1 + 128))
