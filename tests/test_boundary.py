"""CPU: the drop-in boundary — registry semantics of unet_zoo/models/__init__.py:59-238 — and the
C-ABI library: loads here (no GPU) and exports every symbol include/unetzoo_hip.h declares."""
import ctypes
import os
import re

import pytest
import torch

import unet_zoo_amd
from unet_zoo_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REFERENCE_NAMES = sorted([
    'unet', 'attention_unet', 'transatt_unet', 'raunet', 'da_transformer', 'unet_transformer',
    'uctransnet', 'multiresunet', 'nested_unet', 'missformer', 'vnet', 'u2net', 'u2netp',
    'swin_unet_v2', 'resunet', 'wranet', 'egeunet', 'unext', 'unext_s', 'mmunet', 'axialunet',
    'gated', 'medt', 'logo'])


def test_list_models_is_the_reference_list():
    assert unet_zoo_amd.list_models() == REFERENCE_NAMES
    assert set(unet_zoo_amd.hip_models()) <= set(REFERENCE_NAMES)


def test_unknown_model_raises_value_error():
    with pytest.raises(ValueError, match="Unknown model"):
        unet_zoo_amd.create_model("no_such_net")


def test_name_is_case_insensitive_and_depth_is_swallowed():
    m = unet_zoo_amd.create_model("UNet", in_channels=3, num_classes=2, depth=5)
    assert m.out.conv.out_channels == 2 and isinstance(m, torch.nn.Module)


def test_stray_kwarg_raises_type_error():
    with pytest.raises(TypeError):
        unet_zoo_amd.create_model("unet", bogus=1)


def test_swin_requires_image_size():
    with pytest.raises(ValueError, match="image_size"):
        unet_zoo_amd.create_model("swin_unet_v2")


def test_pretrained_only_warns(capsys):
    unet_zoo_amd.create_model("unet", pretrained=True)
    assert "not yet implemented" in capsys.readouterr().out


def test_get_model_config_default_empty():
    assert unet_zoo_amd.get_model_config("unet") == {}


def test_cpu_forward_fails_loudly():
    m = unet_zoo_amd.create_model("unet")
    with pytest.raises(_lib.HipLibraryError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 32, 32))


def test_state_dict_round_trip_with_module_prefix():
    """Checkpoints are bare state_dicts, possibly with DataParallel's 'module.' prefix
    (multi_gpu.py:39-74)."""
    torch.manual_seed(3)
    a = unet_zoo_amd.create_model("unet")
    b = unet_zoo_amd.create_model("unet")
    sd = {"module." + k: v for k, v in a.state_dict().items()}
    b.load_state_dict({k[len("module."):]: v for k, v in sd.items()})
    for (k1, v1), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_library_loads_and_exports_header_symbols():
    lib = _lib.load()
    assert lib.uz_abi_version() == 1
    header = open(os.path.join(ROOT, "include", "unetzoo_hip.h")).read()
    declared = set(re.findall(r"\b(uz_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} not exported by {_lib.LIB_NAME}"


def test_bad_descriptor_is_rejected_without_a_gpu():
    lib = _lib.load()
    d = _lib.ConvDesc(_lib.UZ_BF16, 1, 8, 8, 8, 8, 20, 24, 64, 64, 9, 0, 1, 0, 0)  # Cin % 8 != 0
    assert lib.uz_conv_igemm_grid_m(ctypes.byref(d)) == -1
    assert b"Cin" in lib.uz_last_error_string()


def test_library_is_the_build_of_the_sources_in_the_tree():
    """libunetzoo_hip.so is a build artefact (git-ignored; it travels to the GPU box as it is): uz_source_hash() must be
    the sha256 of the kernel sources as they are in the tree now, in the Makefile's order -- a stale library would
    otherwise be tested and timed in place of the code under review (it happened in round 3: a reverted experiment
    stayed in the .so for a dozen measurements)"""
    import hashlib
    import re
    from unet_zoo_amd import _lib as L
    csrc = os.path.join(ROOT, "unet_zoo_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    srcs = re.search(r"^SRCS = (.*)$", mk, re.M).group(1).split()
    files = [os.path.join(csrc, f) for f in srcs] + [os.path.join(csrc, "uz_common.h"),
                                                      os.path.join(ROOT, "include", "unetzoo_hip.h")]
    h = hashlib.sha256()
    for f in files:
        h.update(open(f, "rb").read())
    got = L.load().uz_source_hash().decode()
    assert got == h.hexdigest(), "libunetzoo_hip.so is stale: run `make -C unet_zoo_amd/csrc` (or __graft_entry__.build())"
