"""The four differentiable image processors restated with stock torch ops
(oracle side; test infrastructure only).  Each `process` cites the reference lines it
follows and returns the same dict keys as the reference class.
"""
from dataclasses import dataclass, field
from typing import List

import torch
import torch.nn.functional as F

from . import geometry as G

CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]
CLIP_STD = [0.26862954, 0.26130258, 0.27577711]


def _aa(img, h, w):
    """`F.interpolate(..., mode='bilinear', antialias=True)` as the reference's GPU run evaluates it.

    ATen's CPU kernel (torch 2.10 here) resamples the width first and then the height, and its height
    pass goes wrong on an image that is ONE pixel wide: 594x1 -> 16x1 differs by 0.1-0.9 from the same
    column inside a two-pixel-wide image, from the explicit tap tables (oracle/resample.py) and from
    the GPU kernel (tests/test_gpu_processors.py::test_width_one_resize_follows_the_gpu_kernel).  Found
    by tools/fuzz_parity.py on a 594x24 image in one 16x16 Mllama tile.  The reference runs on the GPU,
    so a width-1 result takes the same two passes with the column duplicated for the second one."""
    x = img.unsqueeze(0)
    h, w = int(h), int(w)
    if w == 1 and h != x.shape[2]:
        x = F.interpolate(x, size=[x.shape[2], 1], mode="bilinear", align_corners=False, antialias=True)
        x = x.expand(-1, -1, -1, 2).contiguous()
        return F.interpolate(x, size=[h, 2], mode="bilinear", align_corners=False, antialias=True)[..., :1].squeeze(0)
    return F.interpolate(x, size=[h, w], mode="bilinear", align_corners=False, antialias=True).squeeze(0)


def _norm(img, mean, std):
    m = torch.tensor(mean, dtype=img.dtype, device=img.device).view(-1, 1, 1)
    s = torch.tensor(std, dtype=img.dtype, device=img.device).view(-1, 1, 1)
    return (img - m) / s


@dataclass
class LlavaOracle:
    """llavaprocessor.py:134-149."""
    crop_h: int = 336
    crop_w: int = 336
    mean: List[float] = field(default_factory=lambda: list(CLIP_MEAN))
    std: List[float] = field(default_factory=lambda: list(CLIP_STD))

    def process(self, image):
        img = _aa(image, self.crop_h, self.crop_w)              # :143
        return {"pixel_values": _norm(img, self.mean, self.std).unsqueeze(0)}   # :145-148


@dataclass
class MllamaOracle:
    """llama32processor.py:360-405 with its helpers :255-358."""
    tile: int = 560
    max_tiles: int = 4
    mean: List[float] = field(default_factory=lambda: list(CLIP_MEAN))
    std: List[float] = field(default_factory=lambda: list(CLIP_STD))

    def process(self, image):
        _, H, W = image.shape
        nh, nw, th, tw = G.mllama_geometry(H, W, self.max_tiles, self.tile)   # :255-279
        img = _aa(image, nh, nw)                                               # :284
        img = F.pad(img.unsqueeze(0), [0, tw * self.tile - nw, 0, th * self.tile - nh],
                    mode="constant", value=0.0).squeeze(0)                     # :288-306
        img = _norm(img, self.mean, self.std)                                  # :311-315 (pad first: Q12)
        C = img.shape[0]
        tiles = img.reshape(C, th, self.tile, tw, self.tile).permute(1, 3, 0, 2, 4)
        tiles = tiles.reshape(th * tw, C, self.tile, self.tile)                # :326-332
        out = torch.zeros(1, 1, self.max_tiles, C, self.tile, self.tile, dtype=img.dtype, device=img.device)
        out[0, 0, :th * tw] = tiles                                            # :344-355
        return {"pixel_values": out, "aspect_ratio_ids": None, "num_tiles": th * tw}


@dataclass
class Phi3Oracle:
    """phi3processor.py:239-250 with `_pad` :173-216 and `_process` :218-237."""
    num_crops: int = 6
    mean: List[float] = field(default_factory=lambda: list(CLIP_MEAN))
    std: List[float] = field(default_factory=lambda: list(CLIP_STD))

    def process(self, image):
        _, H, W = image.shape
        g = G.phi3_hd_geometry(H, W, self.num_crops)
        img = image.transpose(2, 1) if g["trans"] else image                       # :179-182
        img = F.interpolate(img.unsqueeze(0).float(), size=[g["new_h"], g["new_w"]],
                            mode="bilinear").squeeze(0)                            # :194
        img = F.pad(img.unsqueeze(0), [0, 0, g["pad_top"], g["pad_bottom"]],
                    mode="constant", value=1.0).squeeze(0)                         # :209
        if g["trans"]:
            img = img.transpose(2, 1)                                              # :213-214
        img = _norm(img, self.mean, self.std)                                      # :241
        _, h, w = img.shape
        glob = F.interpolate(img.unsqueeze(0).float(), size=(336, 336), mode="bicubic")   # :220
        local = img.reshape(1, 3, h // 336, 336, w // 336, 336).permute(0, 2, 4, 1, 3, 5)
        local = local.reshape(-1, 3, 336, 336)                                     # :227
        tiles = torch.cat([glob, local], dim=0)                                    # :229
        if tiles.shape[0] < self.num_crops + 1:                                    # :232-235
            pad = torch.zeros(self.num_crops + 1 - tiles.shape[0], 3, 336, 336, dtype=tiles.dtype, device=tiles.device)
            tiles = torch.cat([tiles, pad], dim=0)
        return {"pixel_values": tiles.unsqueeze(0), "image_sizes": [[h, w]],
                "num_img_tokens": [G.phi3_num_img_tokens(h, w)]}                   # :244-249


@dataclass
class Qwen2VLOracle:
    """qwen2VLprocessor.py:211-272."""
    patch: int = 14
    merge: int = 2
    temporal: int = 2
    min_pixels: int = 56 * 56
    max_pixels: int = 28 * 28 * 1280
    mean: List[float] = field(default_factory=lambda: list(CLIP_MEAN))
    std: List[float] = field(default_factory=lambda: list(CLIP_STD))

    def process(self, image):
        _, H, W = image.shape
        hb, wb = G.qwen_smart_resize(H, W, self.patch, self.merge, self.min_pixels, self.max_pixels)
        img = _norm(_aa(image, hb, wb), self.mean, self.std)                       # :233-234
        c = img.shape[0]
        gh, gw = hb // self.patch, wb // self.patch
        vid = img.unsqueeze(0).repeat(self.temporal, 1, 1, 1)                      # :242-243
        p = vid.reshape(1, self.temporal, c, gh // self.merge, self.merge, self.patch,
                        gw // self.merge, self.merge, self.patch)                  # :249-259
        p = p.permute(0, 3, 6, 4, 7, 2, 1, 5, 8)                                   # :262
        flat = p.reshape(gh * gw, c * self.temporal * self.patch * self.patch)     # :265-267
        return {"pixel_values": flat, "num_tiles": [gh * gw]}
