"""CPU: checkpoint I/O in the reference's format (unet_zoo/utils/multi_gpu.py:39-87)."""
import os

import torch

import unet_zoo_amd
from unet_zoo_amd.checkpoint import load_model_state, save_model_state, strip_module_prefix


def _unet(seed):
    torch.manual_seed(seed)
    return unet_zoo_amd.create_model("unet", in_channels=3, num_classes=1)


def test_round_trip_and_module_prefix(tmp_path):
    a, b = _unet(0), _unet(1)
    path = os.path.join(tmp_path, "unet.pth")
    save_model_state(a, path)
    assert list(torch.load(path).keys()) == list(a.state_dict().keys())      # a plain state_dict, as the reference writes
    load_model_state(b, path, torch.device("cpu"))
    assert all(torch.equal(v, b.state_dict()[k]) for k, v in a.state_dict().items())
    # a file written from a DataParallel-wrapped reference model: every key carries 'module.'
    torch.save({"module." + k: v for k, v in a.state_dict().items()}, path)
    c = _unet(2)
    load_model_state(c, path, torch.device("cpu"))
    assert all(torch.equal(v, c.state_dict()[k]) for k, v in a.state_dict().items())
    assert strip_module_prefix({"module.x": 1, "y": 2}) == {"x": 1, "y": 2}


def test_missing_file_and_partial_match(tmp_path, capsys):
    m = _unet(3)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    assert load_model_state(m, os.path.join(tmp_path, "nope.pth"), torch.device("cpu")) is m
    assert "no checkpoint" in capsys.readouterr().out
    assert all(torch.equal(v, m.state_dict()[k]) for k, v in before.items())
    sd = _unet(4).state_dict()
    sd.pop(next(iter(sd)))                       # one tensor missing: strict load fails, strict=False loads the rest
    path = os.path.join(tmp_path, "partial.pth")
    torch.save(sd, path)
    load_model_state(m, path, torch.device("cpu"))
    assert "strict=False" in capsys.readouterr().out
    k = list(sd)[5]
    assert torch.equal(m.state_dict()[k], sd[k])


def test_only_data_parallel_wrappers_are_unwrapped(tmp_path):
    """a model that merely has an attribute called `module` is not a wrapper"""
    import torch.nn as nn
    from unet_zoo_amd.checkpoint import _unwrap

    class HasModule(nn.Module):
        def __init__(self):
            super().__init__()
            self.module = nn.Linear(2, 2)
            self.head = nn.Linear(2, 1)

    m = HasModule()
    assert _unwrap(m) is m
    path = os.path.join(tmp_path, "m.pth")
    save_model_state(m, path)
    assert sorted(torch.load(path)) == ["head.bias", "head.weight", "module.bias", "module.weight"]
    dp = nn.DataParallel(nn.Linear(2, 2))
    assert _unwrap(dp) is dp.module
    save_model_state(dp, path)
    assert sorted(torch.load(path)) == ["bias", "weight"]


def test_saved_tensors_do_not_share_a_flat_storage(tmp_path):
    """parameters that are views of one flat buffer (FlatClipAdamW / GraphedStep) are saved as tensors of their own"""
    m = _unet(5)
    flat = torch.zeros(sum(p.numel() for p in m.parameters()))
    off = 0
    for p in m.parameters():
        flat[off:off + p.numel()].copy_(p.detach().reshape(-1))
        p.data = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
    path = os.path.join(tmp_path, "flat.pth")
    save_model_state(m, path)
    sd = torch.load(path)
    k = next(iter(sd))
    assert sd[k].untyped_storage().nbytes() == sd[k].numel() * 4
