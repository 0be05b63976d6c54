"""The input pipeline (unet_zoo/data/datasets.py:40-59): CPU part pins the numpy restatement of Pillow's
8-bit bilinear resample against Pillow itself and the product's coefficient tables against the restatement; the GPU
part runs the kernels through the C ABI and demands bit-equality with Pillow + torch."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import pil_resize
from unet_zoo_amd.data import GpuPreprocessor, bilinear_coefficients

PIL = pytest.importorskip("PIL.Image")
SIZES = [((37, 53), 64), ((300, 211), 128), ((1024, 768), 512), ((512, 512), 512), ((100, 700), 512), ((7, 5), 32)]


def _pixels(h, w, c, seed):
    rng = np.random.RandomState(seed)
    base = rng.rand(h, w, c) * 255
    yy, xx = np.mgrid[0:h, 0:w]
    base[..., 0] = (base[..., 0] + 40 * np.sin(xx / 7.0) + 40 * np.cos(yy / 5.0)).clip(0, 255)   # structure, not only noise
    a = base.astype(np.uint8)
    return a if c > 1 else a[..., 0]


@pytest.mark.parametrize("hw,size", SIZES)
def test_numpy_restatement_equals_pillow(hw, size):
    rgb, grey = _pixels(*hw, 3, 1), _pixels(*hw, 1, 2)
    want = np.asarray(PIL.fromarray(rgb, "RGB").resize((size, size), PIL.BILINEAR))
    assert np.array_equal(pil_resize.pil_bilinear_resize_u8(rgb, size, size), want)
    wantm = np.asarray(PIL.fromarray(grey, "L").resize((size, size), PIL.BILINEAR))
    assert np.array_equal(pil_resize.pil_bilinear_resize_u8(grey, size, size), wantm)
    # non-square target too
    want2 = np.asarray(PIL.fromarray(rgb, "RGB").resize((size, size // 2), PIL.BILINEAR))
    assert np.array_equal(pil_resize.pil_bilinear_resize_u8(rgb, size // 2, size), want2)


@pytest.mark.parametrize("n_in,n_out", [(53, 64), (1024, 512), (512, 512), (700, 512), (5, 32), (2048, 512)])
def test_product_coefficient_tables_equal_the_restatement(n_in, n_out):
    bounds, kk, ksize = bilinear_coefficients(n_in, n_out)
    ref = pil_resize._coeffs(n_in, n_out)
    assert len(ref) == n_out
    for x, (xmin, ks) in enumerate(ref):
        assert tuple(bounds[x]) == (xmin, len(ks))
        assert list(kk[x, :len(ks)]) == ks and not kk[x, len(ks):].any()
    assert abs(int(kk.sum(1).min()) - (1 << 22)) <= ksize and abs(int(kk.sum(1).max()) - (1 << 22)) <= ksize


def test_golden_fixture_digest(golden_dir):
    """a committed known answer (made by tests/golden/gen_pipeline_golden.py with Pillow): the restatement and the
    conversion reproduce it on any machine"""
    with open(os.path.join(golden_dir, "pipeline_golden.json")) as f:
        g = json.load(f)
    rgb, grey = _pixels(g["h"], g["w"], 3, g["seed"]), _pixels(g["h"], g["w"], 1, g["seed"] + 1)
    img = pil_resize.to_tensor_normalize(pil_resize.pil_bilinear_resize_u8(rgb, g["size"], g["size"]))
    msk = pil_resize.to_tensor_mask(pil_resize.pil_bilinear_resize_u8(grey, g["size"], g["size"]))
    assert hashlib.sha256(img.numpy().tobytes()).hexdigest() == g["image_sha256"]
    assert hashlib.sha256(msk.numpy().tobytes()).hexdigest() == g["mask_sha256"]
    assert int(msk.sum()) == g["mask_positive"]


@pytest.mark.gpu
@pytest.mark.parametrize("hw,size", SIZES)
def test_gpu_pipeline_is_bit_exact_with_pillow_and_torch(hw, size):
    rgb, grey = _pixels(*hw, 3, 3), _pixels(*hw, 1, 4)
    pre = GpuPreprocessor(size=size)
    img, msk = pre([torch.from_numpy(rgb), torch.from_numpy(rgb[::-1].copy())], [torch.from_numpy(grey), torch.from_numpy(grey.T.copy())])
    assert img.shape == (2, 3, size, size) and msk.shape == (2, 1, size, size) and img.dtype == torch.float32
    assert torch.equal(img[0].cpu(), pil_resize.reference_pipeline_image(rgb, size))
    assert torch.equal(img[1].cpu(), pil_resize.reference_pipeline_image(rgb[::-1].copy(), size))
    assert torch.equal(msk[0].cpu(), pil_resize.reference_pipeline_mask(grey, size))
    assert torch.equal(msk[1].cpu(), pil_resize.reference_pipeline_mask(grey.T.copy(), size))


@pytest.mark.gpu
def test_gpu_pipeline_feeds_a_model_and_rejects_bad_input():
    import unet_zoo_amd
    pre = GpuPreprocessor(size=64)
    imgs = [torch.from_numpy(_pixels(90, 70, 3, s)) for s in (5, 6)]
    masks = [torch.from_numpy(_pixels(90, 70, 1, s)) for s in (7, 8)]
    x, t = pre(imgs, masks)
    torch.manual_seed(0)
    m = unet_zoo_amd.create_model("unet").cuda().train()
    step = unet_zoo_amd.GraphedStep(m, "bce_dice", lr=1e-3)
    l0 = step(x, t).item()
    l1 = [step(x, t).item() for _ in range(5)][-1]
    assert np.isfinite(l0) and l1 < l0
    with pytest.raises(TypeError):
        pre.images([torch.zeros(4, 4, 3)])
    with pytest.raises(ValueError):
        pre.images([torch.zeros(4, 4, 1, dtype=torch.uint8)])
