"""CPU oracle of the input pipeline (TEST INFRASTRUCTURE -- see oracle/__init__.py).

The reference's `BoneDataset` (unet_zoo/data/datasets.py:40-59) delegates to torchvision + Pillow, third-party
dependencies (pinned `torchvision==0.17.2`, `pillow==10.2.0` in the reference's requirements.txt; this image has
Pillow 12.2.0 and no torchvision).  `transforms.Resize` on a PIL image is `Image.resize(size[::-1], BILINEAR)`;
`ToTensor` is uint8 HWC -> float32 CHW / 255; `Normalize` is `(t - mean) / std` in float32.

Two checkers: `reference_pipeline_*` calls Pillow itself (the dependency, where it is installed) and restates the two
torchvision transforms with torch ops; `pil_bilinear_resize_u8` restates Pillow's 8-bit two-pass resample in numpy
(src/libImaging/Resample.c) so that the algorithm the kernels follow is written down independently of Pillow's
binary -- tests/test_data_pipeline.py pins it against Pillow on random images.
"""
import math

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def _coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR (triangle, support 1) filter, whole-image box"""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        ws = []
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            ws.append(1.0 - a if a < 1.0 else 0.0)
        tot = sum(ws)
        if tot != 0.0:
            ws = [w / tot for w in ws]
        ks = [int(0.5 + w * (1 << PRECISION_BITS)) if w >= 0 else int(-0.5 + w * (1 << PRECISION_BITS)) for w in ws]
        out.append((xmin, ks))
    return out


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255)


def pil_bilinear_resize_u8(a: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """Image.fromarray(a).resize((out_w, out_h), BILINEAR) for uint8 (H, W) or (H, W, C): horizontal pass to
    (H, out_w), rounded to uint8, then the vertical pass"""
    squeeze = a.ndim == 2
    if squeeze:
        a = a[:, :, None]
    H, W, C = a.shape
    src = a.astype(np.int64)
    tmp = np.zeros((H, out_w, C), dtype=np.int64)
    for x, (xmin, ks) in enumerate(_coeffs(W, out_w)):
        acc = np.full((H, C), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for t, k in enumerate(ks):
            acc += src[:, xmin + t, :] * k
        tmp[:, x, :] = _clip8(acc)
    out = np.zeros((out_h, out_w, C), dtype=np.int64)
    for y, (ymin, ks) in enumerate(_coeffs(H, out_h)):
        acc = np.full((out_w, C), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for t, k in enumerate(ks):
            acc += tmp[ymin + t, :, :] * k
        out[y] = _clip8(acc)
    out = out.astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def to_tensor_normalize(resized_u8: np.ndarray) -> torch.Tensor:
    """transforms.ToTensor() then transforms.Normalize(MEAN, STD) (datasets.py:42-43) on an (H, W, 3) uint8 array"""
    t = torch.from_numpy(np.ascontiguousarray(resized_u8)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    mean = torch.tensor(MEAN, dtype=torch.float32).view(-1, 1, 1)
    std = torch.tensor(STD, dtype=torch.float32).view(-1, 1, 1)
    return t.sub(mean).div(std)


def to_tensor_mask(resized_u8: np.ndarray) -> torch.Tensor:
    """transforms.ToTensor() then `(mask_tensor > 0.5).float()` (datasets.py:48, :59) on an (H, W) uint8 array"""
    t = torch.from_numpy(np.ascontiguousarray(resized_u8))[None].to(torch.float32).div(255)
    return (t > 0.5).float()


def reference_pipeline_image(rgb_u8: np.ndarray, size: int = 512) -> torch.Tensor:
    from PIL import Image
    return to_tensor_normalize(np.asarray(Image.fromarray(rgb_u8, "RGB").resize((size, size), Image.BILINEAR)))


def reference_pipeline_mask(grey_u8: np.ndarray, size: int = 512) -> torch.Tensor:
    from PIL import Image
    return to_tensor_mask(np.asarray(Image.fromarray(grey_u8, "L").resize((size, size), Image.BILINEAR)))
