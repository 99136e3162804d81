"""Child process of tests/test_gpu_dist.py: ONE rank of a 2-rank data-parallel run of the real trainer on one GPU
(CPCSV_DIST_BACKEND=gloo, CPCSV_FORCE_DEVICE=0: RCCL refuses two ranks on one device). Started as a fresh process by
the test (never an exec from a GPU-initialised one). Usage: dist_worker.py <mode> <out.npz>
  mode=parity : one eager step on this rank's shard with its recorded noise; dumps the ALL-REDUCED gradients, losses
  mode=graphs : 6 steps with every captured piece on and live RNG; dumps the final weights and a finiteness flag
  mode=rccl1  : ONE rank (started by torch.distributed.run --nproc-per-node 1): 6 steps, captured pieces on, seeded RNG, the
                conv / dense weights on the deferred-update path; with CPCSV_FORCE_EXCHANGE=1 the process group is RCCL
                (backend nccl, world 1) and every optimiser step runs behind the chunked asynchronous exchange; dumps the
                per-step losses, final weights, which pieces were captured, and the backend that ran"""
import os
import sys

if "torch" not in sys.modules:
    os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    os.environ["CPCSV_PACKET_CAPTURE_EARLY"] = os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def shard(batch, rank, world):
    return {k: v.chunk(world, 0)[rank].contiguous() for k, v in batch.items()}


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    from tests import golden_util as gu
    from tests import parity_util as pu
    from cpcsv import runtime
    from oracle.cpcsv_oracle import synthetic_batch
    runtime.set_deterministic(True)
    fx = gu.load("step_plain.npz")
    oc = gu.cfg_of(fx).but(st_batch=2, im_batch=4)                     # per-rank batch; global = 4 stories / 8 images
    sds = {k: gu.group(fx, "before/" + k) for k in ("G", "D_im", "D_st", "D_se")}
    tr = pu.make_trainer(oc, sds, "fp32")                               # GANTrainer joins the process group itself
    assert tr.world == world and tr.rank == rank
    if mode == "rccl1":
        return rccl1(tr, oc, out_path)
    stb, imb = synthetic_batch(oc.but(st_batch=2 * world, im_batch=4 * world), seed=77)
    stb, imb = pu.to_dev(shard(stb, rank, world)), pu.to_dev(shard(imb, rank, world))
    res = {}
    if mode == "parity":
        tape = np.load(sys.argv[3])
        tape = [torch.from_numpy(tape["r%d_%03d" % (rank, i)]) for i in range(sum(k.startswith("r%d_" % rank) for k in tape.files))]
        pu.set_noise(tr.nets[0], pu.TapeSource(tape))
        grads = {}
        hooks = pu._capture_grads(tr, grads)          # taken right before each optimiser step = after the all-reduce
        out = tr.train_step(stb, imb)
        torch.cuda.synchronize()
        for net, g in grads.items():
            for name, t in g.items():
                res["grad/%s/%s" % (net, name)] = t.numpy()
        for k, v in out.items():
            res["loss/" + k] = np.float64(float(v))
        res["deferred_layers"] = np.array(sum(len(o._layers) for o in tr._opt_of.values() if o is not None))
        for key, net in zip(("G", "D_im", "D_st", "D_se"), tr.nets):
            res["w/" + key] = torch.cat([p.detach().flatten() for p in net.parameters()]).cpu().numpy()
    else:
        torch.manual_seed(1000 + rank)
        torch.cuda.manual_seed_all(1000 + rank)
        finite = True
        for _ in range(6):
            out = tr.train_step(stb, imb)
        torch.cuda.synchronize()
        for key, net in zip(("G", "D_im", "D_st", "D_se"), tr.nets):
            flat = torch.cat([p.detach().flatten() for p in net.parameters()])
            finite = finite and bool(torch.isfinite(flat).all())
            res["w/" + key] = flat.cpu().numpy()
            # spectral-norm u/v depend on weights and call count only: must stay identical without communication
            res["sn/" + key] = torch.cat([b.detach().flatten() for n, b in net.named_buffers() if n.endswith(("weight_u", "weight_v"))] or [torch.zeros(1)]).cpu().numpy()
        res["finite"] = np.array(finite)
        res["captured"] = np.array([getattr(tr.__dict__.get("_ng"), "captured", False), getattr(tr.__dict__.get("_gg"), "captured", False),
                                    all(g.captured for g in tr.__dict__.get("_cg", {}).values())])
    np.savez(out_path, **res)
    from cpcsv import dist as cdist
    cdist.shutdown()


def rccl1(tr, oc, out_path):
    import time
    import torch.distributed as dist
    from cpcsv import dist as cdist
    from oracle.cpcsv_oracle import synthetic_batch
    from tests import parity_util as pu
    stb, imb = synthetic_batch(oc, seed=77)
    stb, imb = pu.to_dev(stb), pu.to_dev(imb)
    torch.manual_seed(4242)
    torch.cuda.manual_seed_all(4242)
    res, losses = {}, []
    calls = {"n": 0}
    if dist.is_initialized():                                       # count the collectives the step really issues
        orig = dist.all_reduce

        def counted(*a, **k):
            calls["n"] += 1
            return orig(*a, **k)
        dist.all_reduce = counted
    t0 = None
    for i in range(8):
        if i == 5:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        out = tr.train_step(stb, imb)
        losses.append([float(out[k]) for k in ("G/loss", "img_D/loss", "st_D/loss", "seg_D/loss")])
    torch.cuda.synchronize()
    res["ms_per_step"] = np.float64(1e3 * (time.perf_counter() - t0) / 3)
    res["losses"] = np.array(losses)
    for key, net in zip(("G", "D_im", "D_st", "D_se"), tr.nets):
        res["w/" + key] = torch.cat([p.detach().flatten() for p in net.parameters()]).cpu().numpy()
    res["captured"] = np.array([getattr(tr.__dict__.get("_ng"), "captured", False), getattr(tr.__dict__.get("_gg"), "captured", False),
                                all(g.captured for g in tr.__dict__.get("_cg", {}).values())])
    res["exchange"] = np.array(bool(tr.exchange))
    res["backend"] = np.array(dist.get_backend() if dist.is_initialized() else "none")
    res["allreduce_calls"] = np.array(calls["n"])
    res["deferred_layers"] = np.array(sum(len(o._layers) for o in tr._opt_of.values() if o is not None))
    res["distributed"] = np.array(bool(cdist.is_distributed()))
    np.savez(out_path, **res)
    if os.environ.get("CPCSV_OLD_TEARDOWN") == "1" and dist.is_initialized():      # (tools/rccl_soak.sh: the round-4 order, to catch its abort)
        dist.barrier()
        dist.destroy_process_group()
        return
    cdist.shutdown()


if __name__ == "__main__":
    main()
