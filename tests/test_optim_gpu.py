"""GPU: the flat clip_grad_norm_ + AdamW step against torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW
(the reference's step tail, unet_zoo/utils/training_loop.py:119-121)."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from unet_zoo_amd.optim import FlatClipAdamW

DEV = "cuda"


@pytest.mark.parametrize("max_norm,scale", [(1.0, 5.0), (1.0, 1e-3), (0.0, 1.0)])
def test_flat_clip_adamw_matches_torch(max_norm, scale):
    g = torch.Generator().manual_seed(51)
    shapes = [(64, 3, 3, 3), (64,), (7, 5), (1,), (129, 33), (2, 3, 5, 7)]
    ours = [nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in shapes]
    ref = [nn.Parameter(p.detach().clone()) for p in ours]
    opt = FlatClipAdamW(ours, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=max_norm)
    topt = torch.optim.AdamW(ref, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    for p, s in zip(ours, shapes):                    # parameters and gradients are views of the flat buffers
        assert p.shape == torch.Size(s) and p.grad.shape == p.shape
        assert opt.flat_p.data_ptr() <= p.data_ptr() < opt.flat_p.data_ptr() + opt.flat_p.numel() * 4
    for step in range(5):
        norms = []
        for p, r in zip(ours, ref):
            gr = (torch.randn(p.shape, generator=g) * scale).to(DEV)
            p.grad.copy_(gr)
            r.grad = gr.clone()
        total = torch.sqrt(sum((r.grad.double() ** 2).sum() for r in ref))
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_(ref, max_norm)
        topt.step()
        opt.step()
        assert abs(opt.last_grad_norm().item() - total.item()) < 1e-5 * total.item()
        for p, r in zip(ours, ref):
            assert torch.allclose(p.detach(), r.detach(), rtol=2e-5, atol=2e-6), (step, p.shape)
    assert opt.step_count.item() == 5.0
