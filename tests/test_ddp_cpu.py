"""CPU, world_size 2 over gloo: the bucketed gradient reducer equals single-process gradients
of the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jspsr_amd.ddp import GradReducer, broadcast_module
    torch.manual_seed(100 + rank)  # different init per rank: broadcast must fix it
    net = torch.nn.Sequential(torch.nn.Conv2d(2, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 1, 3, padding=1))
    broadcast_module(net)
    red = GradReducer(net.parameters(), bucket_bytes=64)  # tiny buckets -> several all-reduces
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 2, 8, 8, generator=g)
    y = torch.randn(4, 1, 8, 8, generator=g)
    for _ in range(2):  # second step checks that aliases survive zero_grad
        red.zero_grad()
        sl = slice(rank * 2, rank * 2 + 2)
        ((net(x[sl]) - y[sl]) ** 2).mean().backward()
        red.finish()
    q.put((rank, [p.grad.clone() for p in net.parameters()], [p.detach().clone() for p in net.parameters()], len(red.buckets)))
    dist.barrier()
    dist.destroy_process_group()


def test_gradreducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, g0, w0, nb), (_, g1, w1, _) = res
    assert nb > 1
    for a, b in zip(w0, w1):
        assert torch.equal(a, b)
    for a, b in zip(g0, g1):
        assert torch.allclose(a, b, atol=1e-7)
    # single-process reference on the full batch with rank 0's (broadcast) weights
    net = torch.nn.Sequential(torch.nn.Conv2d(2, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 1, 3, padding=1))
    with torch.no_grad():
        for p, w in zip(net.parameters(), w0):
            p.copy_(w)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 2, 8, 8, generator=g)
    y = torch.randn(4, 1, 8, 8, generator=g)
    ((net(x) - y) ** 2).mean().backward()
    for p, a in zip(net.parameters(), g0):
        assert torch.allclose(p.grad, a, atol=1e-6)
