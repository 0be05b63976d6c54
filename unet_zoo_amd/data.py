"""Input pipeline on the GPU: what ``BoneDataset.__getitem__`` does per sample on the CPU
(unet_zoo/data/datasets.py:40-59) --

    image: PIL RGB -> transforms.Resize((512, 512)) -> ToTensor -> Normalize(ImageNet mean / std)
    mask:  PIL L   -> transforms.Resize((512, 512)) -> ToTensor -> (> 0.5).float()

-- as two kernel launches per tensor on decoded uint8 pixels, bit-exact with Pillow + torch on the CPU.

``transforms.Resize`` on a PIL image is ``Image.resize(size, BILINEAR)``: Pillow's antialiased two-pass resample,
8-bit pixels, 22-bit fixed-point coefficients, rounding to uint8 after each pass (src/libImaging/Resample.c).  The
coefficient tables depend only on (input length, output length); they are built here on the host in float64 with
Pillow's operation order (``precompute_coeffs`` + ``normalize_coeffs_8bpc``) and cached, the passes are integer
arithmetic in ``uz_pil_resample_h_u8`` / ``uz_pil_resample_v_f32`` (the second one fused with ToTensor + Normalize /
the mask threshold, writing the (3, H, W) / (1, H, W) fp32 planes the models take).

File decoding (PIL / libjpeg / libpng) stays on the host: only the pixels travel.
"""
from __future__ import annotations

import math
from ctypes import c_float
from typing import Dict, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L

PRECISION_BITS = 32 - 8 - 2
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def bilinear_coefficients(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """(bounds (out, 2) int32, coefficients (out, ksize) int32, ksize) of Pillow's BILINEAR filter for one axis"""
    scale = in_size / out_size                       # in1 - in0 over the output length, box = the whole image
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                      # BILINEAR.support = 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            if v < 0.0:
                v = -v
            wv = 1.0 - v if v < 1.0 else 0.0
            w[x] = wv
            ww += wv
        if ww != 0.0:
            for x in range(xmax):
                w[x] /= ww
        bounds[xx] = (xmin, xmax)
        for x in range(ksize):                       # normalize_coeffs_8bpc
            p = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + p) if w[x] < 0 else int(0.5 + p)
    return bounds, kk, ksize


class GpuPreprocessor:
    """Resize + ToTensor + Normalize (images) / Resize + ToTensor + threshold (masks) of decoded uint8 pixels."""

    def __init__(self, size: int = 512, mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD,
                 device="cuda"):
        self.size = int(size)
        self.device = torch.device(device)
        self._mean = (c_float * 3)(*mean)
        self._std = (c_float * 3)(*std)
        self._tables: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor, int]] = {}

    def _table(self, in_size: int, out_size: int):
        key = (in_size, out_size)
        t = self._tables.get(key)
        if t is None:
            b, k, ks = bilinear_coefficients(in_size, out_size)
            t = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device), ks)
            self._tables[key] = t
        return t

    def _one(self, pix: torch.Tensor, out: torch.Tensor, mode: int) -> None:
        if pix.dtype != torch.uint8 or pix.dim() not in (2, 3):
            raise TypeError(f"expected uint8 pixels (H, W, C) or (H, W), got {pix.dtype} {tuple(pix.shape)}")
        if pix.dim() == 2:
            pix = pix.unsqueeze(-1)
        H, W, C = pix.shape
        if C != (3 if mode == 0 else 1):
            raise ValueError(f"{'image' if mode == 0 else 'mask'} needs {3 if mode == 0 else 1} channel(s), got {C}")
        pix = pix.to(self.device).contiguous()
        L.require_cuda(pix, out)
        lib, s = L.load(), L.stream_ptr()
        S = self.size
        bh, kh, ksh = self._table(W, S)
        tmp = torch.empty((H, S, C), dtype=torch.uint8, device=self.device)
        L.check(lib.uz_pil_resample_h_u8(pix.data_ptr(), H, W, C, bh.data_ptr(), kh.data_ptr(), ksh, S, tmp.data_ptr(), s),
                "uz_pil_resample_h_u8")
        bv, kv, ksv = self._table(H, S)
        L.check(lib.uz_pil_resample_v_f32(tmp.data_ptr(), H, S, C, bv.data_ptr(), kv.data_ptr(), ksv, S, self._mean, self._std,
                                          mode, out.data_ptr(), s), "uz_pil_resample_v_f32")

    def images(self, pixels: Sequence[torch.Tensor]) -> torch.Tensor:
        """list of (H_i, W_i, 3) uint8 RGB tensors (host or device) -> (N, 3, size, size) fp32, normalised"""
        out = torch.empty((len(pixels), 3, self.size, self.size), dtype=torch.float32, device=self.device)
        for i, p in enumerate(pixels):
            self._one(p, out[i], 0)
        return out

    def masks(self, pixels: Sequence[torch.Tensor]) -> torch.Tensor:
        """list of (H_i, W_i) uint8 greyscale tensors -> (N, 1, size, size) fp32 in {0, 1}"""
        out = torch.empty((len(pixels), 1, self.size, self.size), dtype=torch.float32, device=self.device)
        for i, p in enumerate(pixels):
            self._one(p, out[i], 1)
        return out

    def __call__(self, images: Sequence[torch.Tensor], masks: Sequence[torch.Tensor]):
        return self.images(images), self.masks(masks)
