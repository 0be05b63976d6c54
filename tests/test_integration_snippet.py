"""INTEGRATION.md's code is executed, not just shown: the ctypes stub of section 2 (whose ConvDesc was 8 bytes short
in round 1) and the three-line GraphedStep change of section 1a."""
import ctypes
import os
import re

import pytest
import torch
import torch.nn.functional as F

import unet_zoo_amd
from unet_zoo_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.findall(r"```python\n(.*?)```", text, flags=re.S)


def _snippet(marker):
    found = [b for b in _blocks() if marker in b]
    assert len(found) == 1, f"{marker!r} appears in {len(found)} python blocks of INTEGRATION.md"
    return found[0]


def _header_struct_fields(name):
    text = open(os.path.join(ROOT, "include", "unetzoo_hip.h")).read()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        assert decl.startswith("int "), decl
        fields += [f.strip() for f in decl[4:].split(",")]
    return fields


def test_ctypes_stub_matches_the_header():
    ns = {}
    exec(_snippet("# snippet: conv3x3_nhwc"), ns)          # loads the library too (no GPU call)
    stub = ns["ConvDesc"]
    want = _header_struct_fields("uz_conv_desc")
    assert [n for n, _ in stub._fields_] == want == [n for n, _ in _lib.ConvDesc._fields_]
    assert ctypes.sizeof(stub) == ctypes.sizeof(_lib.ConvDesc) == 4 * len(want)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ctypes_stub_computes_conv3x3(dtype):
    ns = {}
    exec(_snippet("# snippet: conv3x3_nhwc"), ns)
    N, H, W, Cin, Cout = 2, 32, 48, 64, 128
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    x2d = x.permute(0, 2, 3, 1).reshape(N * H * W, Cin).to(dtype).contiguous()
    y = ns["conv3x3_nhwc"](x2d, ns["pack_conv3x3"](w, dtype), b, N, H, W)
    torch.cuda.synchronize()
    got = y.float().reshape(N, H, W, Cout).permute(0, 3, 1, 2)
    want = F.conv2d(x2d.float().reshape(N, H, W, Cin).permute(0, 3, 1, 2), w.to(dtype).float(), b, padding=1)
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert (got - want).abs().max().item() <= tol * want.abs().max().item()


@pytest.mark.gpu
def test_graphed_step_snippet_runs_as_written():
    class config:                       # the names the snippet uses from scripts/train.py
        LEARNING_RATE = 1e-4
    torch.manual_seed(0)
    model = unet_zoo_amd.create_model("unet", in_channels=3, num_classes=1).cuda().train()
    criterion = torch.nn.BCEWithLogitsLoss()
    g = torch.Generator().manual_seed(1)
    img = torch.randn(2, 3, 64, 64, generator=g).cuda()
    mask = (torch.rand(2, 1, 64, 64, generator=g) > 0.5).float().cuda()
    ns = {"unet_zoo_amd": unet_zoo_amd, "model": model, "criterion": criterion, "config": config, "img": img, "mask": mask}
    code = _snippet("unet_zoo_amd.GraphedStep(model, criterion")
    first = None
    for i in range(4):
        exec(code if i == 0 else "loss = step(img, mask)\noutputs = step.outputs", ns)
        torch.cuda.synchronize()
        first = float(ns["loss"]) if first is None else first
    assert ns["outputs"].shape == (2, 1, 64, 64) and torch.isfinite(ns["outputs"]).all()
    assert float(ns["loss"]) < first        # four optimizer steps on one batch reduce the loss
