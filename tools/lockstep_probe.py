#!/usr/bin/env python3
"""The plain fp32 lock-step run of tests/parity_util.run_multistep_parity, one line per step with every net's gradient error
against the oracle - under switches that move or remove concurrency, to locate a timing-dependent deviation:
    SYNC=<points>  comma list of places where a torch.cuda.synchronize() is inserted into GANTrainer.train_step:
                   gfwd_before / gfwd_after (around the generator's differentiable forward), critic_before / critic_after
                   (around every critic update's enqueue), nograd_after
    plus any product environment (CPCSV_STREAMS=0, CPCSV_TEXT_FUSED=0, CPCSV_POISON=1, ...).
usage: SYNC=gfwd_before python tools/lockstep_probe.py"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd")):
    sys.path.insert(0, p)
if "torch" not in sys.modules:
    os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    os.environ["CPCSV_PACKET_CAPTURE_EARLY"] = os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"]
import torch  # noqa: E402


def main():
    import trainer as T
    from tests import parity_util as pu
    pts = set(x for x in os.environ.get("SYNC", "").split(",") if x)

    def wrap(name, before, after):
        orig = getattr(T.GANTrainer, name)

        def f(self, *a, **k):
            if before in pts:
                torch.cuda.synchronize()
            out = orig(self, *a, **k)
            if after in pts:
                torch.cuda.synchronize()
            return out
        setattr(T.GANTrainer, name, f)
    wrap("_generator_forward", "gfwd_before", "gfwd_after")
    wrap("_critic_backward", "critic_before", "critic_after")
    wrap("_nograd_fakes", "nograd_before", "nograd_after")
    reps = pu.run_multistep_parity("plain", "fp32", lockstep=True, check=False)
    tag = " ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("CPCSV_") and k not in ("CPCSV_PACKET_CAPTURE_EARLY",))
    for k, rep in enumerate(reps):
        print("[SYNC=%s %s] step %d  " % (",".join(sorted(pts)) or "-", tag, k)
              + "  ".join("%s %.2e" % (n, rep[n]) for n in ("loss_rel", "gradl2_G", "gradl2_D_im", "gradl2_D_st", "gradl2_D_se")), flush=True)


if __name__ == "__main__":
    main()
