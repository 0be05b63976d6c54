"""Generate tests/golden/* by IMPORTING the reference's own model files (run in the build
container only: /root/reference does not exist on the GPU box).

    python oracle/gen_golden.py

TEST INFRASTRUCTURE — see oracle/__init__.py.  Nothing is copied from the reference: its
``unet_zoo/models/{common_layers,unet}.py`` are loaded with importlib under a synthetic parent
package (``import unet_zoo`` itself fails on the absent torchvision) and executed
on CPU in fp32; only inputs-by-formula digests and OUTPUT numbers are written.
"""
from __future__ import annotations

import hashlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference/unet_zoo/models"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle.torch_ref import synthetic_batch  # noqa: E402


def load_reference(*files: str):
    pkg = types.ModuleType("refzoo")
    pkg.__path__ = [os.path.dirname(REF)]
    sub = types.ModuleType("refzoo.models")
    sub.__path__ = [REF]
    sys.modules["refzoo"], sys.modules["refzoo.models"] = pkg, sub
    mods = {}
    for f in files:
        name = f"refzoo.models.{f}"
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, f + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        mods[f] = m
    return mods


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()


def sample_idx(n: int, k: int) -> np.ndarray:
    rng = np.random.RandomState(12345)
    return np.sort(rng.choice(n, size=min(k, n), replace=False))


def run_case(model, B, H, W, tag, full_logits: bool, name: str = "unet",
             bn_keys=("down_convolution_1.conv.conv_op.1", "bottle_neck.conv_op.4", "up_convolution_4.conv.conv_op.4")):
    x, mask = synthetic_batch(B, 3, H, W, seed=1)
    model.train()
    logits = model(x)
    loss = F.binary_cross_entropy_with_logits(logits, mask)
    model.zero_grad()
    loss.backward()
    named = [(n, p) for n, p in model.named_parameters() if p.grad is not None]   # missformer: unused norm2/norm3
    gnorm = torch.sqrt(sum((p.grad.double() ** 2).sum() for _, p in named)).item()
    arrays = {}
    meta = {
        "model": name, "B": B, "H": H, "W": W, "input_sha256": sha(x), "mask_sha256": sha(mask),
        "loss": loss.item(), "global_grad_norm": gnorm,
        "train_logits_mean": logits.mean().item(), "train_logits_std": logits.std().item(),
        "train_positive_pixels": int((logits > 0).sum().item()),
        "grad_l2": {n: p.grad.double().norm().item() for n, p in named},
    }
    flat = logits.detach().flatten()
    idx = sample_idx(flat.numel(), 4096)
    arrays["logit_idx"] = idx
    arrays["train_logits_sampled"] = flat[idx].numpy()
    if full_logits:
        arrays["train_logits"] = logits.detach().numpy()
        for n, p in named:  # 64 sampled gradient values per parameter
            gi = sample_idx(p.numel(), 64)
            arrays["gidx/" + n] = gi
            arrays["gval/" + n] = p.grad.flatten()[gi].numpy()
    # running statistics after the single train-mode forward
    sd = model.state_dict()
    for k in bn_keys:
        arrays["rm/" + k] = sd[k + ".running_mean"].numpy()
        arrays["rv/" + k] = sd[k + ".running_var"].numpy()
    model.eval()
    with torch.no_grad():
        ev = model(x)
    meta.update(eval_logits_mean=ev.mean().item(), eval_logits_std=ev.std().item(),
                eval_positive_pixels=int((ev > 0).sum().item()))
    arrays["eval_logits_sampled"] = ev.flatten()[idx].numpy()
    if full_logits:
        arrays["eval_logits"] = ev.numpy()
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **arrays)
    with open(os.path.join(OUT, f"{tag}.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(tag, "loss", meta["loss"], "gnorm", gnorm, "train mean", meta["train_logits_mean"])


def write_manifest(model, name):
    sd = model.state_dict()
    manifest = {
        "model": name, "seed": 0, "n_params": sum(p.numel() for p in model.parameters()),
        "entries": [[k, list(v.shape), str(v.dtype).replace("torch.", ""), sha(v)] for k, v in sd.items()],
    }
    with open(os.path.join(OUT, f"{name}_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0)
    print("manifest", name, manifest["n_params"], len(manifest["entries"]))


def write_attention_unet():
    """attention_unet: B=2 3x64x64 with every number kept (the BASELINE
    config's 512x512 is reduced as SURVEY §8c allows)."""
    mods = load_reference("common_layers", "attention_unet")
    Ref = mods["attention_unet"].AttentionUNet
    torch.manual_seed(0)
    model = Ref(in_channels=3, num_classes=1, depth=5)
    write_manifest(model, "attention_unet")
    run_case(model, 2, 64, 64, "attention_unet_b2_64", full_logits=True, name="attention_unet",
             bn_keys=("conv1.conv.1", "att5.psi.1", "att2.w_g.1", "up2.up.2", "upconv2.conv.4"))


def write_u2net():
    """u2net: seed-0 U2NET(3, 1), B=2 3x64x64 train step with the summed
    seven-head BCE loss of the reference's training loop (training_loop.py:24-32, 60-64), all seven
    maps kept; reduced from the BASELINE config's 512x512 as SURVEY §8c allows."""
    mods = load_reference("u2net")
    torch.manual_seed(0)
    model = mods["u2net"].U2NET(in_ch=3, out_ch=1)
    write_manifest(model, "u2net")
    B, H, W, tag = 2, 64, 64, "u2net_b2_64"
    x, mask = synthetic_batch(B, 3, H, W, seed=1)
    model.train()
    outs = model(x)
    loss = sum(F.binary_cross_entropy_with_logits(v, mask) for v in outs.values())
    model.zero_grad()
    loss.backward()
    named = [(n, p) for n, p in model.named_parameters() if p.grad is not None]   # missformer: unused norm2/norm3
    gnorm = torch.sqrt(sum((p.grad.double() ** 2).sum() for _, p in named)).item()
    arrays = {}
    meta = {"model": "u2net", "B": B, "H": H, "W": W, "input_sha256": sha(x), "mask_sha256": sha(mask),
            "loss": loss.item(), "global_grad_norm": gnorm, "keys": list(outs.keys()),
            "grad_l2": {n: p.grad.double().norm().item() for n, p in named},
            "train_positive_pixels": {k: int((v > 0).sum().item()) for k, v in outs.items()}}
    for k, v in outs.items():
        arrays["train/" + k] = v.detach().numpy()
    for n, p in named:
        gi = sample_idx(p.numel(), 64)
        arrays["gidx/" + n] = gi
        arrays["gval/" + n] = p.grad.flatten()[gi].numpy()
    sd = model.state_dict()
    bn_keys = ("stage1.rebnconvin.bn_s1", "stage1.rebnconv7.bn_s1", "stage5.rebnconv4.bn_s1", "stage6.rebnconv1d.bn_s1",
               "stage3d.rebnconv2d.bn_s1", "stage1d.rebnconv1d.bn_s1")
    meta["bn_keys"] = list(bn_keys)
    for k in bn_keys:
        arrays["rm/" + k] = sd[k + ".running_mean"].numpy()
        arrays["rv/" + k] = sd[k + ".running_var"].numpy()
    model.eval()
    with torch.no_grad():
        ev = model(x)
    for k, v in ev.items():
        arrays["eval/" + k] = v.numpy()
    meta["eval_positive_pixels"] = {k: int((v > 0).sum().item()) for k, v in ev.items()}
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **arrays)
    with open(os.path.join(OUT, f"{tag}.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(tag, "loss", meta["loss"], "gnorm", gnorm)


def write_nested_unet():
    """nested_unet (UNet++): seed-0 NestedUNet(num_classes=1, in_channels=3), B=2 3x64x64 with
    every number kept; and the deep-supervision variant's four maps (summed BCE) at the same size."""
    mods = load_reference("nested_unet")
    Ref = mods["nested_unet"].NestedUNet
    torch.manual_seed(0)
    model = Ref(num_classes=1, in_channels=3)
    write_manifest(model, "nested_unet")
    run_case(model, 2, 64, 64, "nested_unet_b2_64", full_logits=True, name="nested_unet",
             bn_keys=("conv0_0.bn1", "conv2_0.bn2", "conv4_0.bn1", "conv1_2.bn1", "conv0_4.bn2"))
    torch.manual_seed(0)
    model = Ref(num_classes=2, in_channels=3, deep_supervision=True)
    x, mask = synthetic_batch(2, 3, 32, 48, seed=3)
    mask2 = torch.cat([mask, 1.0 - mask], 1)
    model.train()
    outs = model(x)
    loss = sum(F.binary_cross_entropy_with_logits(o, mask2) for o in outs)
    model.zero_grad()
    loss.backward()
    named = list(model.named_parameters())
    meta = {"model": "nested_unet", "deep_supervision": True, "B": 2, "H": 32, "W": 48, "loss": loss.item(),
            "global_grad_norm": torch.sqrt(sum((p.grad.double() ** 2).sum() for _, p in named)).item(),
            "grad_l2": {n: p.grad.double().norm().item() for n, p in named}}
    arrays = {f"train/{i}": o.detach().numpy() for i, o in enumerate(outs)}
    np.savez_compressed(os.path.join(OUT, "nested_unet_ds_b2_32x48.npz"), **arrays)
    with open(os.path.join(OUT, "nested_unet_ds_b2_32x48.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("nested_unet_ds", meta["loss"], meta["global_grad_norm"])


def write_resunet():
    """resunet: seed-0 ResUnet(in_channels=3, num_classes=1), B=2 3x64x64, every number kept."""
    mods = load_reference("common_layers", "resunet")
    torch.manual_seed(0)
    model = mods["resunet"].ResUnet(in_channels=3, num_classes=1)
    write_manifest(model, "resunet")
    run_case(model, 2, 64, 64, "resunet_b2_64", full_logits=True, name="resunet",
             bn_keys=("input_layer.1", "residual_conv_1.conv_block.0", "residual_conv_2.conv_skip.1",
                      "bridge.conv_block.3", "up_residual_conv1.conv_block.0", "up_residual_conv3.conv_skip.1"))


def write_missformer():
    """missformer: seed-0 MISSFormer(num_classes=1, in_channels=3, image_size=128) — the class is built
    directly because create_model() drops image_size (models/__init__.py:145-148) — B=2 3x128x128, every number kept;
    the manifest pins the seed-0 construction at the default image_size=512 as well (same parameter shapes)."""
    mods = load_reference("missformer")
    torch.manual_seed(0)
    model = mods["missformer"].MISSFormer(num_classes=1, in_channels=3, image_size=128)
    write_manifest(model, "missformer")
    run_case(model, 2, 128, 128, "missformer_b2_128", full_logits=True, name="missformer", bn_keys=())


def write_transatt_unet():
    """transatt_unet: seed-0 TransAttUNet(in_channels=3, num_classes=1), B=2 3x64x64, every number
    kept.  The train-mode dropout on the channel-attention matrix (rate 0.1, transatt_unet.py:86-88) is switched
    off for the fixture: its mask comes from the RNG stream, which no second implementation can share."""
    mods = load_reference("common_layers", "transatt_unet")
    torch.manual_seed(0)
    model = mods["transatt_unet"].TransAttUNet(in_channels=3, num_classes=1)
    write_manifest(model, "transatt_unet")
    model.sdpa.dropout.p = 0.0
    with torch.no_grad():
        model.pam.gamma.fill_(0.5)      # the reference initialises it to 0, which hides the whole position-attention branch
    run_case(model, 2, 64, 64, "transatt_unet_b2_64", full_logits=True, name="transatt_unet",
             bn_keys=("inc.double_conv.1", "down4.maxpool_conv.1.double_conv.4", "up1.conv.double_conv.1", "up4.conv.double_conv.4"))


def write_unet_transformer():
    """unet_transformer: seed-0 U_Transformer(in_channels=3, num_classes=1) with the default 64x64
    attention grid, B=2 3x64x64, every number kept"""
    mods = load_reference("common_layers", "unet_transformer")
    torch.manual_seed(0)
    model = mods["unet_transformer"].U_Transformer(in_channels=3, num_classes=1)
    write_manifest(model, "unet_transformer")
    run_case(model, 2, 64, 64, "unet_transformer_b2_64", full_logits=True, name="unet_transformer",
             bn_keys=("inc.conv_op.1", "down3.maxpool_conv.1.double_conv.4", "up1.MHCA.Sconv_process.2",
                      "up2.MHCA.conv_after_attention.1", "up3.MHCA.Yconv2_process.3", "up3.conv.4"))


def write_multiresunet():
    """multiresunet: seed-0 MultiResUnet(in_channels=3, num_classes=1), B=2 3x64x64, every number kept"""
    mods = load_reference("multiresunet")
    torch.manual_seed(0)
    model = mods["multiresunet"].MultiResUnet(in_channels=3, num_classes=1)
    write_manifest(model, "multiresunet")
    run_case(model, 2, 64, 64, "multiresunet_b2_64", full_logits=True, name="multiresunet",
             bn_keys=("multiresblock1.conv2d_bn_1x1.batchnorm", "multiresblock1.batch_norm1", "respath1.blocks.2.2",
                      "multiresblock5.conv2d_bn_7x7.batchnorm", "multiresblock9.batch_norm1", "conv_final.batchnorm"))


def write_uctransnet():
    """uctransnet: seed-0 UCTransNet(get_uctransnet_config(), img_size=64), B=2 3x64x64, every number
    kept; the transformer's dropouts (embeddings 0.1, MLP 0.1, uctransnet.py:16-18) are switched off for the fixture
    (their masks come from the RNG stream); a second manifest pins the seed-0 construction at img_size=256."""
    mods = load_reference("common_layers", "uctransnet")
    U = mods["uctransnet"]
    torch.manual_seed(0)
    write_manifest(U.UCTransNet(U.get_uctransnet_config(), in_channels=3, num_classes=1, img_size=256), "uctransnet_256")
    torch.manual_seed(0)
    model = U.UCTransNet(U.get_uctransnet_config(), in_channels=3, num_classes=1, img_size=64)
    write_manifest(model, "uctransnet_64")
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    with torch.no_grad():       # zero position embeddings (the reference's initial value) would leave their path untested
        for i in range(4):
            getattr(model.mtc, f"embeddings_{i + 1}").position_embeddings.normal_(0.0, 0.5, generator=torch.Generator().manual_seed(10 + i))
    run_case(model, 2, 64, 64, "uctransnet_b2_64", full_logits=True, name="uctransnet",
             bn_keys=("inc.norm", "down4.nConvs.1.norm", "mtc.reconstruct_1.norm", "mtc.reconstruct_4.norm", "up4.nConvs.0.norm", "up1.nConvs.1.norm"))


def _timm_stand_in():
    """`swin_unet_v2.py:9` imports three helpers from timm, which this image lacks:
    to_2tuple, trunc_normal_ (= torch.nn.init.trunc_normal_) and DropPath (stochastic depth: per-sample
    Bernoulli(keep) / keep in training, identity otherwise).  Supplied in-process; the goldens below
    use drop_path_rate=0 (train) or eval mode, where DropPath is the identity either way."""
    import torch.nn as nn
    layers = types.ModuleType("timm.models.layers")

    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1.0 - self.drop_prob
            m = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
            return x * m / keep

    layers.DropPath = DropPath
    layers.to_2tuple = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    timm, models = types.ModuleType("timm"), types.ModuleType("timm.models")
    timm.models, models.layers = models, layers
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers})


def write_swin():
    """swin_unet_v2.  (A) img 64 / window 4, B=2: full train step (drop_path_rate 0)
    with every gradient norm + sampled values, eval logits; (B) img 256 / window 8 (the north-star shape)
    and (C) img 224 / window 7 (BASELINE configs[3] shape), B=1: sampled train/eval logits."""
    import contextlib
    import io
    _timm_stand_in()
    Ref = load_reference("swin_unet_v2")["swin_unet_v2"].SwinTransformerSys

    def build(img, ws):
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            return Ref(img_size=img, in_chans=3, num_classes=1, window_size=ws, drop_path_rate=0.0)

    model = build(64, 4)
    write_manifest(model, "swin_unet_v2_64_ws4")
    B, H, tag = 2, 64, "swin_unet_v2_b2_64_ws4"
    x, mask = synthetic_batch(B, 3, H, H, seed=1)
    model.train()
    logits = model(x)
    loss = F.binary_cross_entropy_with_logits(logits, mask)
    model.zero_grad()
    loss.backward()
    named = [(n, p) for n, p in model.named_parameters() if p.grad is not None]
    unused = [n for n, p in model.named_parameters() if p.grad is None]
    gnorm = torch.sqrt(sum((p.grad.double() ** 2).sum() for _, p in named)).item()
    arrays = {"train_logits": logits.detach().numpy()}
    meta = {"model": "swin_unet_v2", "B": B, "H": H, "W": H, "window_size": 4, "input_sha256": sha(x),
            "mask_sha256": sha(mask), "loss": loss.item(), "global_grad_norm": gnorm,
            "grad_l2": {n: p.grad.double().norm().item() for n, p in named}, "unused_parameters": unused,
            "train_positive_pixels": int((logits > 0).sum().item())}
    for n, p in named:
        gi = sample_idx(p.numel(), 64)
        arrays["gidx/" + n] = gi
        arrays["gval/" + n] = p.grad.flatten()[gi].numpy()
    model.eval()
    with torch.no_grad():
        ev = model(x)
    arrays["eval_logits"] = ev.numpy()
    meta["eval_positive_pixels"] = int((ev > 0).sum().item())
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **arrays)
    with open(os.path.join(OUT, f"{tag}.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(tag, "loss", meta["loss"], "gnorm", gnorm, "unused", len(unused))

    for img, ws in ((256, 8), (224, 7)):
        model = build(img, ws)
        if img == 256:
            write_manifest(model, "swin_unet_v2_256_ws8")
        tag = f"swin_unet_v2_b1_{img}_ws{ws}"
        x, mask = synthetic_batch(1, 3, img, img, seed=1)
        model.train()
        with torch.no_grad():
            tr = model(x)
        model.eval()
        with torch.no_grad():
            ev = model(x)
        idx = sample_idx(tr.numel(), 8192)
        np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), logit_idx=idx,
                            train_logits_sampled=tr.flatten()[idx].numpy(), eval_logits_sampled=ev.flatten()[idx].numpy())
        meta = {"model": "swin_unet_v2", "B": 1, "H": img, "W": img, "window_size": ws, "input_sha256": sha(x),
                "train_logits_mean": tr.mean().item(), "train_logits_std": tr.std().item(),
                "train_positive_pixels": int((tr > 0).sum().item()),
                "loss": F.binary_cross_entropy_with_logits(tr, mask).item()}
        with open(os.path.join(OUT, f"{tag}.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        print(tag, meta["train_logits_mean"], meta["train_positive_pixels"])


def main():
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ["swin"]:
        torch.set_num_threads(8)
        write_swin()
        return
    if sys.argv[1:] == ["u2net"]:
        torch.set_num_threads(8)
        write_u2net()
        return
    if sys.argv[1:] == ["resunet"]:
        torch.set_num_threads(8)
        write_resunet()
        return
    if sys.argv[1:] == ["missformer"]:
        torch.set_num_threads(8)
        write_missformer()
        return
    if sys.argv[1:] == ["nested_unet"]:
        torch.set_num_threads(8)
        write_nested_unet()
        return
    if sys.argv[1:] == ["unet_transformer"]:
        torch.set_num_threads(8)
        write_unet_transformer()
        return
    if sys.argv[1:] == ["uctransnet"]:
        torch.set_num_threads(8)
        write_uctransnet()
        return
    if sys.argv[1:] == ["multiresunet"]:
        torch.set_num_threads(8)
        write_multiresunet()
        return
    if sys.argv[1:] == ["transatt_unet"]:
        torch.set_num_threads(8)
        write_transatt_unet()
        return
    torch.set_num_threads(8)
    mods = load_reference("common_layers", "unet")
    RefUNet = mods["unet"].UNet

    torch.manual_seed(0)
    model = RefUNet(in_channels=3, num_classes=1)
    sd = model.state_dict()
    manifest = {
        "model": "unet", "seed": 0, "n_params": sum(p.numel() for p in model.parameters()),
        "entries": [[k, list(v.shape), str(v.dtype).replace("torch.", ""), sha(v)] for k, v in sd.items()],
    }
    allbytes = hashlib.sha256()
    for k, v in sd.items():
        allbytes.update(v.detach().contiguous().numpy().tobytes())
    manifest["state_digest"] = allbytes.hexdigest()
    with open(os.path.join(OUT, "unet_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0)
    print("manifest", manifest["n_params"], len(manifest["entries"]), manifest["state_digest"][:16])

    # case A: small, every number kept
    run_case(model, 2, 64, 64, "unet_b2_64", full_logits=True)
    write_attention_unet()
    # case B: BASELINE configs[0] (B=2, 256x256): fresh seed-0 model
    torch.manual_seed(0)
    model = RefUNet(in_channels=3, num_classes=1)
    run_case(model, 2, 256, 256, "unet_b2_256", full_logits=False)
    write_u2net()
    write_swin()
    write_nested_unet()
    write_resunet()
    write_missformer()
    write_transatt_unet()
    write_unet_transformer()
    write_multiresunet()
    write_uctransnet()


if __name__ == "__main__":
    main()
