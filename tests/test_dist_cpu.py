"""N>1 path on CPU: world_size-2 gloo. The gradient exchange (cpcsv.dist.GradBucket) must turn per-rank
gradients into their mean on every rank, leave ranks bit-identical, and be a no-op without a process group."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "cpcstoryvisualization-pytorch_amd")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, payload="fp32"):
    sys.path.insert(0, PKG)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from cpcsv import dist as cdist
    r, w, _ = cdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and cdist.is_distributed()
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    if rank == 1:                                  # diverge, then re-sync from rank 0
        for p in net.parameters():
            p.data.add_(1.0)
    cdist.broadcast_module(net)
    bucket = cdist.GradBucket(net.parameters(), payload=payload).adopt()      # the product path: persistent flat gradient buffer
    bucket.zero()
    torch.manual_seed(100 + rank)                  # different shard per rank
    x = torch.randn(4, 7)
    net(x).pow(2).sum().backward()
    local = [p.grad.clone() for p in net.parameters()]
    assert all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in net.parameters())   # still views of the flat buffer
    bucket.allreduce_mean()
    tmax = cdist.all_reduce_max(torch.tensor([10.0 + rank], dtype=torch.float64))       # bench.py's max-over-ranks time
    assert float(tmax) == 10.0 + world - 1
    q.put((rank, [p.detach().tolist() for p in net.parameters()], [g.tolist() for g in local],
           [p.grad.tolist() for p in net.parameters()]))      # plain lists: no shared-memory hand-off to outlive us
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("payload", ["fp32", "bf16"])
def test_gradient_mean_allreduce_world2(payload):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, payload)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    T = lambda xs: [torch.tensor(x) for x in xs]
    (_, w0, l0, g0), (_, w1, l1, g1) = [(r, T(a), T(b), T(c)) for r, a, b, c in res]
    for a, b in zip(w0, w1):
        assert torch.equal(a, b)                                       # replicas identical after broadcast
    for a, b, c, d in zip(l0, l1, g0, g1):
        tol = 1e-6 if payload == "fp32" else 1e-2 * float((a.abs() + b.abs()).max())   # bf16 payload: 8 bits of mantissa
        assert torch.allclose(c, (a + b) / 2, atol=tol) and torch.equal(c, d)   # mean, same on both ranks


def test_bucket_is_noop_without_process_group():
    sys.path.insert(0, PKG)
    from cpcsv import dist as cdist
    assert not cdist.is_distributed()
    lin = torch.nn.Linear(3, 2)
    lin(torch.ones(1, 3)).sum().backward()
    before = lin.weight.grad.clone()
    cdist.GradBucket(lin.parameters()).allreduce_mean()
    assert torch.equal(before, lin.weight.grad)


def test_retired_parameters_leave_the_live_gradient_buffer():
    """GradBucket.adopt(retired=...): the deferred-update weights have no master-layout gradient (it lives in a layer
    accumulator); their .grad is a stride-0 view of ONE zero - outside the buffer that zero() clears and allreduce_mean()
    sends, costing no memory (632 MB at cfg/final.yml widths before) - and is marked so that nothing writes through it.
    Single process, no process group: allreduce_mean is a no-op and must not touch anything."""
    sys.path.insert(0, PKG)
    from cpcsv import dist as cdist
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.Linear(4, 3))
    w0, b0, w1, b1 = list(net.parameters())
    bucket = cdist.GradBucket(net.parameters()).adopt()
    assert bucket.flat.numel() == sum(p.numel() for p in net.parameters())
    bucket.adopt(retired=[w0])
    live = b0.numel() + w1.numel() + b1.numel()
    assert bucket.flat.numel() == live and bucket._storage.numel() == live       # no storage behind the live part any more
    lo, hi = bucket.flat.data_ptr(), bucket.flat.data_ptr() + 4 * live
    assert all(lo <= p.grad.data_ptr() < hi for p in (b0, w1, b1))
    assert not (lo <= w0.grad.data_ptr() < hi) and w0.grad.shape == w0.shape and set(w0.grad.stride()) == {0}
    assert getattr(w0, "_cpcsv_retired", False) and float(w0.grad.abs().sum()) == 0.0
    for p in (b0, w1, b1):
        p.grad.fill_(1.0)
    bucket.zero()
    assert all(float(p.grad.abs().sum()) == 0.0 for p in (b0, w1, b1))
    bucket.allreduce_mean()                                                   # no process group: nothing happens
    assert float(w0.grad.abs().sum()) == 0.0
    assert abs(bucket.norm()) == 0.0
