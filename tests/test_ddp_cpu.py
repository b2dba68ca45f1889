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
    # numpy arrays travel by value (tensors travel as shared-memory handles that die with this process)
    q.put((rank, [p.grad.numpy().copy() for p in net.parameters()], [p.detach().numpy().copy() for p in net.parameters()],
           len(red.buckets)))
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
    (_, g0, w0, nb), (_, g1, w1, _) = [(r, [torch.from_numpy(a) for a in g], [torch.from_numpy(a) for a in w], n)
                                       for r, g, w, n in res]
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


class _Work:
    def wait(self):
        return True


def _stub_reducer(monkeypatch, world=2, bucket_bytes=64):
    from jspsr_amd.ddp import GradReducer
    calls = []
    monkeypatch.setattr(dist, "all_reduce", lambda t, op=None, group=None, async_op=False: (calls.append(t.numel()), _Work())[1])
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(2, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 1, 3, padding=1))
    red = GradReducer(net.parameters(), bucket_bytes=bucket_bytes, world=world)
    x = torch.randn(2, 2, 8, 8)
    return net, red, x, calls


def test_backward_before_first_zero_grad_is_counted(monkeypatch):
    """The bucket counters start from the bucket sizes: a backward pass issued before the first reducer.zero_grad()
    reduces every bucket exactly once (they used to start at zero and go negative: nothing was reduced, silently)."""
    net, red, x, calls = _stub_reducer(monkeypatch)
    net(x).mean().backward()
    assert len(calls) == len(red.buckets) and all(v == 0 for v in red._pending)
    red.finish()
    with pytest.raises(RuntimeError, match="exactly once per step"):
        red.finish()          # a second finish() would divide by the world size again


def test_dropped_gradient_aliases_are_an_error_not_a_silent_zero_step(monkeypatch):
    """The reference's loop calls model.zero_grad(set_to_none=True) (train/train_utils.py:210).  If that drops the
    aliases into the flat buffer, autograd allocates fresh .grad tensors and the flat buffer (what is all-reduced and
    what FlatAdamW reads) stays zero: finish() must refuse, and reducer.zero_grad() must restore the aliases."""
    net, red, x, calls = _stub_reducer(monkeypatch)
    red.zero_grad()
    net.zero_grad(set_to_none=True)
    net(x).mean().backward()
    with pytest.raises(RuntimeError, match="no longer aliases"):
        red.finish()
    red.zero_grad()           # re-aliases
    net(x).mean().backward()
    red.finish()
    assert all(p.grad.data_ptr() >= red.flat.data_ptr() for p in net.parameters()) and red.flat.abs().sum() > 0


def test_attached_module_zero_grad_keeps_the_aliases(monkeypatch):
    from jspsr_amd.blocks import HotPathModule
    from jspsr_amd.ddp import GradReducer

    class Net(HotPathModule):
        def __init__(self):
            super().__init__()
            self.c = torch.nn.Conv2d(2, 1, 3, padding=1)

        def forward(self, x):
            return self.c(x)

    net = Net()
    red = GradReducer(net.parameters(), world=1).attach(net)
    for _ in range(2):
        net.zero_grad(set_to_none=True)       # what the reference's loop does
        net(torch.randn(1, 2, 4, 4)).mean().backward()
        red.finish()
        assert red.flat.abs().sum() > 0 and net.c.weight.grad.data_ptr() == red.flat.data_ptr() + 4 * net.c.bias.numel()


def test_shared_layer_with_direct_gradients_is_refused(monkeypatch):
    net, red, x, calls = _stub_reducer(monkeypatch)
    p = next(net.parameters())
    red.zero_grad()
    p._jspsr_grad_ready(p)
    with pytest.raises(RuntimeError, match="more than once"):
        p._jspsr_grad_ready(p)


def _worker_buffers(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jspsr_amd.ddp import GradReducer, broadcast_module, sync_buffers
    torch.manual_seed(7)
    net = torch.nn.Sequential(torch.nn.Conv2d(2, 4, 3, padding=1), torch.nn.BatchNorm2d(4), torch.nn.ReLU(), torch.nn.Conv2d(4, 1, 1))
    broadcast_module(net)
    red = GradReducer(net.parameters())
    g = torch.Generator().manual_seed(rank)         # a different shard per rank: the running statistics drift apart
    net.train()
    for _ in range(3):
        red.zero_grad()
        net(torch.randn(4, 2, 8, 8, generator=g) * (1 + rank)).mean().backward()
        red.finish()
    bn = net[1]
    before = (bn.running_mean.clone(), bn.running_var.clone())
    n = red.sync_buffers(net)
    # a replica that ran a different number of steps is refused (num_batches_tracked must agree)
    if rank == 1:
        bn.num_batches_tracked += 1
    try:
        sync_buffers(net)
        refused = False
    except RuntimeError:
        refused = True
    q.put((rank, n, [t.numpy().copy() for t in before], [bn.running_mean.numpy().copy(), bn.running_var.numpy().copy()],
           int(bn.num_batches_tracked), refused))
    dist.barrier()
    dist.destroy_process_group()


def test_sync_buffers_averages_running_statistics_world2_gloo():
    """BatchNorm statistics are per replica during training; before a checkpoint the running statistics
    become the mean over the replicas (GradReducer.sync_buffers) so that rank 0 saves the job's, not its own."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_buffers, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, n0, b0, a0, t0, r0), (_, n1, b1, a1, t1, r1) = res
    assert n0 == n1 == 2 and t0 == 3 and r0 and r1
    for i in range(2):
        assert abs(b0[i] - b1[i]).max() > 1e-3                       # they had drifted
        assert abs(a0[i] - (b0[i] + b1[i]) / 2).max() < 1e-6         # and are now the mean, on both ranks
        assert abs(a0[i] - a1[i]).max() == 0
