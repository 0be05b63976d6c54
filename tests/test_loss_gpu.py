"""GPU: the step's loss and Dice metric on the device (uz_bce_dice; scripts/train.py:135, utils/metrics.py:7-24)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from unet_zoo_amd.loss import bce_dice_with_logits, loss_and_dice

DEV = "cuda"


def dice_reference(prediction, target, epsilon=1e-7, threshold=0.5):
    """the arithmetic of utils/metrics.py:7-24 (oracle restatement for this test)"""
    p = (torch.sigmoid(prediction) > threshold).float().flatten()
    t = target.flatten()
    union = p.sum() + t.sum()
    if union == 0:
        return torch.tensor(1.0)
    return (2.0 * (p * t).sum() + epsilon) / (union + epsilon)


@pytest.mark.parametrize("shape", [(2, 1, 64, 64), (3, 1, 37, 53), (16, 1, 256, 256)])
def test_bce_and_dice_match_torch(shape):
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(shape, generator=g) * 4).requires_grad_(True)
    t = (torch.rand(shape, generator=g) > 0.6).float()
    ref = F.binary_cross_entropy_with_logits(x, t)
    ref.backward()
    xd = x.detach().to(DEV).requires_grad_(True)
    loss, dice = bce_dice_with_logits(xd, t.to(DEV))
    (loss * 1.0).backward()
    assert abs(loss.item() - ref.item()) < 2e-6 * max(1.0, abs(ref.item()))
    assert abs(dice.item() - dice_reference(x.detach(), t).item()) < 1e-6
    assert (xd.grad.cpu() - x.grad).abs().max() <= 2e-6 * x.grad.abs().max()
    assert not dice.requires_grad


def test_empty_union_and_containers():
    x = torch.full((1, 1, 8, 8), -3.0, device=DEV)
    t = torch.zeros(1, 1, 8, 8, device=DEV)
    assert bce_dice_with_logits(x, t)[1].item() == 1.0
    g = torch.Generator().manual_seed(4)
    outs = {f"d{i}": torch.randn(2, 1, 16, 16, generator=g).to(DEV).requires_grad_(True) for i in range(3)}
    m = (torch.rand(2, 1, 16, 16, generator=g) > 0.5).float().to(DEV)
    loss, dice = loss_and_dice(outs, m)
    ref = sum(F.binary_cross_entropy_with_logits(v, m) for v in outs.values())
    assert abs(loss.item() - ref.item()) < 1e-5
    assert abs(dice.item() - dice_reference(outs["d0"].detach().cpu(), m.cpu()).item()) < 1e-6
    loss.backward()
    assert all(v.grad is not None for v in outs.values())
    lst = [outs["d0"].detach(), outs["d1"].detach()]
    _, dice_l = loss_and_dice(lst, m)
    assert abs(dice_l.item() - dice_reference(lst[-1].cpu(), m.cpu()).item()) < 1e-6
