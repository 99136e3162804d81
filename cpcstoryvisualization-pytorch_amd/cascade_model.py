"""cascade_model.py — cascade variant of the generator (reference cascade_model.py:221-540).

The generated segmentation image is re-encoded (presample + 4 downBlocks) and THOSE features gate
the image branch; the critics are identical to model.py's (the reference duplicates them verbatim,
cascade_model.py:75-104,543-674), so they are re-exported.
"""
import torch.nn as nn

from cpcsv import functional as F
from cpcsv import modules as M
from model import (CA_NET, D_GET_LOGITS, STAGE1_D_IMG, STAGE1_D_SEG, STAGE1_D_STY_V2,  # noqa: F401
                   StoryGAN as _PlainStoryGAN, conv3x3, upBlock)


def downBlock(in_planes, out_planes):
    """conv3x3 stride 2 WITH bias -> BatchNorm2d -> ReLU (reference cascade_model.py:36-41)."""
    return M.FusedSequential(M.Conv2d(in_planes, out_planes, 3, 2, 1, bias=True), M.BatchNorm2d(out_planes), nn.ReLU(True))


class StoryGAN(_PlainStoryGAN):
    def _define_cascade(self, ngf_seg):
        self.presample = M.FusedSequential(conv3x3(1, ngf_seg // 16), M.BatchNorm2d(ngf_seg // 16), nn.ReLU())  # :312-316
        self.downsample1_seg = downBlock(ngf_seg // 16, ngf_seg // 8)
        self.downsample2_seg = downBlock(ngf_seg // 8, ngf_seg // 4)
        self.downsample3_seg = downBlock(ngf_seg // 4, ngf_seg // 2)
        self.downsample4_seg = downBlock(ngf_seg // 2, ngf_seg)

    def _encode_seg(self, seg_nhwc):
        z = self.presample(seg_nhwc)
        g4 = self.downsample1_seg(z)
        g3 = self.downsample2_seg(g4)
        g2 = self.downsample3_seg(g3)
        g1 = self.downsample4_seg(g2)
        return g1, g2, g3, g4

    def _decode(self, zmc_all):
        """reference cascade_model.py:401-438."""
        x = F.FeatToNhwcFn.apply(self.fc(zmc_all), self.gf_dim, 4, 4)
        s0 = F.FeatToNhwcFn.apply(self.fc_seg(zmc_all), self.gf_dim_seg, 4, 4)
        h1 = self.upsample1_seg(s0)
        h2 = self.upsample2_seg(h1)
        h3 = self.upsample3_seg(h2)
        h4 = self.upsample4_seg(h3)
        segm = self.img_seg(h4)
        g1, g2, g3, g4 = self._encode_seg(segm)
        x = F.GateFn.apply(self.seg_c(g1), x)                      # :419
        x = self.upsample1(x)
        x = F.GateFn.apply(self.seg_c1(g2), x)                     # :423
        x = self.upsample2(x)
        x = self.upsample3(x)
        x = self.upsample4(x)
        lat = lambda t, c: t[..., :c].permute(0, 3, 1, 2)          # NCHW-shaped views, like the reference returns
        latents = (tuple(lat(t, c) for t, c in zip((s0, h1, h2, h3), self._seg_widths())),
                   tuple(lat(t, c) for t, c in zip((g1, g2, g3, g4), self._seg_widths())))       # :441
        return latents, self.img(x), segm

    def _seg_widths(self):
        n = self.gf_dim_seg
        return (n, n // 2, n // 4, n // 8)

    def train_autoencoder(self, real_segments):
        """reference cascade_model.py:528-540. real_segments: (N,1,64,64) fp32."""
        from cpcsv.runtime import tdtype
        g1, _, _, _ = self._encode_seg(F.ToNhwcFn.apply(real_segments, tdtype()))
        h = g1
        for up in (self.upsample1_seg, self.upsample2_seg, self.upsample3_seg, self.upsample4_seg):
            h = up(h)
        return F.ToPlanarFn.apply(self.img_seg(h), 1)


STAGE1_G = StoryGAN
