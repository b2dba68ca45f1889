"""GPU, two ranks on ONE card over gloo (RCCL refuses two ranks per device, gloo moves CUDA tensors through the
host): the whole data-parallel step of bench.py -- broadcast, branch / weight-gradient streams, gradients written
straight into the flat buffer, bucketed asynchronous all-reduce issued from the hooks, finish(), FlatAdamW --
with real collectives.  Each rank works on its own tiles; the averaged gradient must equal the mean of the two
ranks' independently computed gradients, and both replicas must hold identical parameters afterwards."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

IC = {"lr_dem": 1, "image": 3, "mask": 15, "COP30": 1}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(rank):
    from oracle import jspsr_ref as R
    inputs, _ = R.synthetic_batch(2, 32, 64, True, seed=40 + rank, dtype=torch.float32)
    probe = R.probe_gradient((2, 1, 32, 64), 50 + rank, torch.float32)
    return [t.cuda() for t in inputs], probe.cuda()


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jspsr_amd.JSPSR import Model
    from jspsr_amd.ddp import GradReducer, broadcast_module
    from jspsr_amd.optim import FlatAdamW
    torch.manual_seed(7 + rank)                      # different init per rank: the broadcast must fix it
    m = Model(IC, num_feature=8).cuda()
    broadcast_module(m)
    start = {k: v.detach().clone().cpu() for k, v in m.state_dict().items()}
    red = GradReducer(m.parameters(), bucket_bytes=128 << 10)   # several buckets
    red.watch_streams(m.side_streams())
    opt = FlatAdamW(red, lr=1e-3, weight_decay=1e-6)
    inputs, probe = _batch(rank)
    red.zero_grad()
    (m(*inputs) * probe).mean().backward()
    red.finish()
    torch.cuda.synchronize()
    grads = [p.grad.detach().clone().cpu() for p in m.parameters()]
    opt.step()
    torch.cuda.synchronize()
    after = [p.detach().clone().cpu() for p in m.parameters()]
    torch.save((start, grads, after, len(red.buckets)), os.path.join(outdir, f"rank{rank}.pt"))   # files, not a Queue:
    dist.barrier()                                                   # hundreds of tensors through fd passing is fragile
    dist.destroy_process_group()


def test_two_ranks_one_card_full_step(tmp_path):
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
    hung = [p for p in procs if p.is_alive()]
    for p in hung:
        p.kill()
    assert not hung, "a rank did not finish"
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    (s0, g0, a0, nb), (s1, g1, a1, _) = (torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(2))
    assert nb > 3
    for k in s0:                                     # broadcast: same starting point
        assert torch.equal(s0[k], s1[k]), k
    for a, b in zip(g0, g1):                         # all-reduced: same gradient on both ranks
        assert torch.equal(a, b)
    for a, b in zip(a0, a1):                         # same optimizer step
        assert torch.equal(a, b)
    # reference: each rank's gradient computed alone (no reducer, one stream), averaged here
    from jspsr_amd import ops
    from jspsr_amd.JSPSR import Model
    keep = ops.wgrad_async
    ops.wgrad_async = False
    try:
        ref = []
        for rank in range(2):
            m = Model(IC, num_feature=8).cuda()
            m.load_state_dict(s0)
            m.branch_streams = False
            inputs, probe = _batch(rank)
            (m(*inputs) * probe).mean().backward()
            ref.append([p.grad.detach().clone().cpu() for p in m.parameters()])
    finally:
        ops.wgrad_async = keep
    for i, (a, b, got) in enumerate(zip(ref[0], ref[1], g0)):
        want = (a + b) / 2
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-9 + 1e-6 * want.abs().max().item()), i


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without torch.distributed.run must start its own two ranks (child processes created
    before anything touches the GPU) and print ONE JSON line for n_gpus 2.  One-GPU rehearsal: both ranks on cuda:0
    over gloo (JSPSR_BENCH_REHEARSAL=1); on a real node the same code path runs one rank per GPU over RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["JSPSR_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "1", "--no-roofline"], env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0 and out["config"]["global_tiles"] == 2
    assert out["cpu_baseline"] and out["cpu_baseline"]["value"] > 0        # rank 0 times the CPU oracle at every N (north_star: "in the same run")
