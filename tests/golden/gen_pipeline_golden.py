"""Known answer for the input pipeline, made with Pillow (the reference's own dependency for transforms.Resize):
    python tests/golden/gen_pipeline_golden.py
writes tests/golden/pipeline_golden.json (digests of the normalised image and of the thresholded mask)."""
import hashlib
import json
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pil_resize  # noqa: E402


def _pixels(h, w, c, seed):      # the same synthetic pixels tests/test_data_pipeline.py builds
    rng = np.random.RandomState(seed)
    base = rng.rand(h, w, c) * 255
    yy, xx = np.mgrid[0:h, 0:w]
    base[..., 0] = (base[..., 0] + 40 * np.sin(xx / 7.0) + 40 * np.cos(yy / 5.0)).clip(0, 255)
    a = base.astype(np.uint8)
    return a if c > 1 else a[..., 0]


h, w, size, seed = 301, 457, 128, 11
rgb, grey = _pixels(h, w, 3, seed), _pixels(h, w, 1, seed + 1)
img = pil_resize.to_tensor_normalize(np.asarray(Image.fromarray(rgb, "RGB").resize((size, size), Image.BILINEAR)))
msk = pil_resize.to_tensor_mask(np.asarray(Image.fromarray(grey, "L").resize((size, size), Image.BILINEAR)))
out = {"h": h, "w": w, "size": size, "seed": seed, "pillow": Image.__version__ if hasattr(Image, "__version__") else "",
       "image_sha256": hashlib.sha256(img.numpy().tobytes()).hexdigest(),
       "mask_sha256": hashlib.sha256(msk.numpy().tobytes()).hexdigest(), "mask_positive": int(msk.sum())}
with open(os.path.join(HERE, "pipeline_golden.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(out)
