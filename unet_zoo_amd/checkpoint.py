"""Checkpoint I/O in the reference's file format (unet_zoo/utils/multi_gpu.py:39-87).

A checkpoint is ``torch.save`` of the bare ``state_dict`` of the UNWRAPPED model.  The HIP models register their
parameters under the reference's names, shapes and order (the seed-0 manifests in tests/golden pin that) and rebuild
their kernel-layout weight copies from the fp32 masters on the next forward, so a ``.pth`` written by the reference
loads here unmodified, and one written here loads in the reference.

Behaviour kept from the reference's loader: a missing file leaves the model untouched (a warning, no exception);
keys saved from a wrapper lose their ``module.`` prefix; a strict load is tried first and a non-strict one second; if
both fail the model keeps its current weights.  The return value is the model that was passed in.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, Mapping

import torch
import torch.nn as nn

_PREFIX = "module."


def _wrapper_types() -> tuple:
    from .parallel import RcclDataParallel
    return (nn.DataParallel, nn.parallel.DistributedDataParallel, RcclDataParallel)


def _unwrap(model: nn.Module) -> nn.Module:
    """the module inside a data-parallel wrapper (only the three wrapper classes: a model that merely HAS an
    attribute called `module` is left alone)"""
    return model.module if isinstance(model, _wrapper_types()) else model


def strip_module_prefix(state_dict: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out: Dict[str, torch.Tensor] = OrderedDict()
    for key, value in state_dict.items():
        out[key[len(_PREFIX):] if key.startswith(_PREFIX) else key] = value
    return out


def save_model_state(model: nn.Module, path: str) -> None:
    """Tensors are detached copies: after GraphedStep / FlatClipAdamW the parameters are views of one flat buffer,
    and a saved view would drag the whole buffer's storage into the file."""
    state = OrderedDict((k, v.detach().clone()) for k, v in _unwrap(model).state_dict().items())
    torch.save(state, path)


def _report(message: str) -> None:
    print(message)


def load_model_state(model: nn.Module, path: str, device) -> nn.Module:
    if not os.path.exists(path):
        _report(f"Warning: no checkpoint at {path}; the model keeps its current weights.")
        return model
    target = _unwrap(model)
    state = strip_module_prefix(torch.load(path, map_location=device))
    loaded = False
    for strict in (True, False):
        try:
            target.load_state_dict(state, strict=strict)
        except Exception as err:  # noqa: BLE001  (size mismatches raise RuntimeError under either mode)
            _report(f"load_state_dict(strict={strict}) failed for {path}: {err}")
            continue
        _report(f"Loaded {path} onto {device}" + ("." if strict else " with strict=False (partial match)."))
        loaded = True
        break
    if not loaded:
        _report(f"{path} could not be loaded; the model keeps its current weights.")
    cache = getattr(target, "_pack_cache", None)
    if cache is not None:
        cache.invalidate()      # kernel-layout copies are rebuilt from the loaded masters on the next forward
    return model
