"""GPU: the last decoder block's BatchNorm + ReLU folded into the 1x1 head (Engine.fold_bn_apply_head, round 5): the head's
forward reads the RAW convolution output through the map (uz_outconv_fwd_xf), its backward forms the activation from the raw
tensor (uz_outconv_bwd_bnred with x = NULL).  The activation it forms is the number the stand-alone pass would have stored,
and every sum runs in the same order, so a training step must be the same BIT FOR BIT with the switch on and off.
Reference: DoubleConv / ConvBlock / DoubleConvo second half -> OutConv (common_layers.py:31-33, :55-57, :125, :143-145)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.mark.parametrize("name", ["unet", "attention_unet", "transatt_unet"])
def test_a_training_step_is_the_same_with_and_without_the_folded_head(name, monkeypatch):
    """B=2 64x64 bf16: logits, loss, every parameter gradient and every BatchNorm buffer bit for bit; and the folded run
    really took the folded entries"""
    import unet_zoo_amd
    from unet_zoo_amd import ops
    from unet_zoo_amd.engine import Engine
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(1)).to(DEV)
    t = (torch.rand(2, 1, 64, 64, generator=torch.Generator().manual_seed(2)) > 0.5).float().to(DEV)
    calls = {"fwd_xf": 0, "bwd_lazy": 0}
    fwd0, bwd0 = ops.outconv_fwd, ops.outconv_bwd

    def fwd(xa, w, b, xform=None):
        calls["fwd_xf"] += xform is not None
        return fwd0(xa, w, b, xform=xform)

    def bwd(*a, **k):
        calls["bwd_lazy"] += bool(k.get("lazy"))
        return bwd0(*a, **k)

    monkeypatch.setattr(ops, "outconv_fwd", fwd)
    monkeypatch.setattr(ops, "outconv_bwd", bwd)
    runs = []
    for folded in (True, False):
        torch.manual_seed(0)
        m = unet_zoo_amd.create_model(name, in_channels=3, num_classes=1)
        m.run_dtype = torch.bfloat16
        m = m.cuda().train()
        monkeypatch.setattr(Engine, "fold_bn_apply_head", folded)
        before = dict(calls)
        out = m(x)
        logits = out[0] if isinstance(out, (list, tuple)) else out
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits.float(), t)
        loss.backward()
        torch.cuda.synchronize()
        took = (calls["fwd_xf"] - before["fwd_xf"], calls["bwd_lazy"] - before["bwd_lazy"])
        assert took == ((1, 1) if folded else (0, 0)), took
        runs.append((m, loss.detach().clone(), logits.detach().clone()))
    (m1, l1, o1), (m2, l2, o2) = runs
    assert torch.equal(o1, o2) and torch.equal(l1, l2)
    for (n1, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert (p1.grad is None) == (p2.grad is None), n1
        if p1.grad is not None:
            assert torch.equal(p1.grad, p2.grad), n1
    for (n1, b1), (_, b2) in zip(m1.named_buffers(), m2.named_buffers()):
        assert torch.equal(b1, b2), n1


def test_inference_reads_the_head_through_the_running_statistics():
    """model.eval() under no_grad: the folded head on running statistics equals the two-pass form bit for bit"""
    import unet_zoo_amd
    from unet_zoo_amd.engine import Engine
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(3)).to(DEV)
    outs = []
    for folded in (True, False):
        torch.manual_seed(0)
        m = unet_zoo_amd.create_model("unet", in_channels=3, num_classes=2)
        m.run_dtype = torch.bfloat16
        m = m.cuda().eval()
        old = Engine.fold_bn_apply_head
        Engine.fold_bn_apply_head = folded
        try:
            with torch.no_grad():
                out = m(x)
        finally:
            Engine.fold_bn_apply_head = old
        outs.append((out[0] if isinstance(out, (list, tuple)) else out).clone())
    assert torch.equal(outs[0], outs[1])
