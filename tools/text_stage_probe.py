"""cpcsv_text_stage launch by launch at the benchmark's dimensions (cfg/final.yml text widths, ST=12 / IM=60): per stage of the
forward and of the backward, its jobs (type, rows of the two calls, N, K, blocks) and its GPU time alone (events around a re-launch
of the same stage descriptor, 20 times). Where the ~0.7 ms of stage time per step goes.   python tools/text_stage_probe.py   (GPU box)"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

TYPES = {}


def main():
    from cpcsv import _lib as L, kernels as K, runtime, textpath as TP
    from oracle.cpcsv_oracle.config import pororo_cfg
    from tests.parity_util import apply_cfg
    for k in dir(L):
        if k.startswith("TXT_") and isinstance(getattr(L, k), int) and k not in ("TXT_MAX_JOBS", "TXT_MAX_ROWS"):
            TYPES[getattr(L, k)] = k[4:]
    runtime.set_compute_dtype("bf16")
    st, im = int(os.environ.get("ST", "12")), int(os.environ.get("IM", "60"))
    oc = pororo_cfg(st_batch=st, im_batch=im, gf_dim=4, gf_seg_dim=16, df_dim=8)       # text widths of cfg/final.yml, a tiny decoder
    apply_cfg(oc)
    import model as mod
    from miscc.utils import weights_init
    g = mod.StoryGAN(oc.video_len).apply(weights_init).cuda().train()
    t, md, td = oc.video_len, oc.text_dim + oc.label_num, oc.text_dim
    ins = (torch.randn(st, t, md, device="cuda"), torch.randn(st, t, td, device="cuda"),
           torch.randn(im, md, device="cuda"), torch.randn(im, t, td, device="cuda"))
    log = []
    orig_run = TP._Stages.run

    def run(self):
        s_ = K.stream()
        for i in sorted(self.st):
            sd = self.st[i]
            K._call("cpcsv_text_stage", C.byref(sd), s_)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):            # (idempotent except for BatchNorm running statistics and gradient accumulators: timing only)
                K._call("cpcsv_text_stage", C.byref(sd), s_)
            e1.record()
            torch.cuda.synchronize()
            jobs = ["%s[M=%d,%d N=%d K=%d blk=%d]" % (TYPES.get(sd.job[k].type, sd.job[k].type), sd.job[k].M[0], sd.job[k].M[1] if sd.job[k].npass > 1 else 0,
                                                       sd.job[k].N, sd.job[k].K, sd.job[k].nblk) for k in range(sd.njobs)]
            log.append((i, e0.elapsed_time(e1) / 20 * 1e3, sum(sd.job[k].nblk for k in range(sd.njobs)), jobs))
    TP._Stages.run = run
    captured = {}

    class Stop(Exception):
        pass

    def grab(zmc_all, nst, nim, bs_, vl, seg, temp_, im_m, r_mu, r_lv, c_mu, c_lv):
        captured.update(zmc=zmc_all, r_mu=r_mu, r_lv=r_lv, c_mu=c_mu, c_lv=c_lv)
        raise Stop()
    g._decode_both = grab
    try:
        g._sample_both(ins[0], ins[1], ins[2], ins[3], True, st, t, ins[1].reshape(-1, t * td), ins[0].reshape(-1, md), ins[3].reshape(-1, t * td))
    except Stop:
        pass
    nf = len(log)
    loss = sum(v.float().sum() for v in captured.values())
    loss.backward()
    torch.cuda.synchronize()
    for part, rows in (("forward", log[:nf]), ("backward", log[nf:])):
        print("## %s: %d stages, %.1f us" % (part, len(rows), sum(r[1] for r in rows)))
        for i, us, blocks, jobs in rows:
            print("  stage %2d  %6.1f us  %4d blocks  %s" % (i, us, blocks, "  ".join(jobs)))


if __name__ == "__main__":
    main()
