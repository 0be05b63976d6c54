"""GPU: unet_zoo_amd.GraphedStep -- the reference's training step (training_loop.py:108-124) replayed from hipGraphs
-- against the same step launched eagerly: bit for bit over three optimizer steps, every model on the engine."""
import pytest
import torch
import torch.nn.functional as F

import unet_zoo_amd
from unet_zoo_amd.loss import loss_and_dice
from unet_zoo_amd.optim import FlatClipAdamW

pytestmark = pytest.mark.gpu

CASES = [
    ("unet", {}, 2, 64),
    ("attention_unet", {}, 2, 64),
    ("u2net", {}, 2, 64),
    ("swin_unet_v2", {"image_size": 64, "window_size": 4, "drop_path_rate": 0.0}, 2, 64),
    ("nested_unet", {}, 2, 64),
    ("resunet", {}, 2, 64),
    ("missformer", {"image_size": 128}, 2, 128),
    # the four compositions of round 2 (their attention cores run as torch ops inside the captured graphs: the
    # capture-time memset check of step.py applies to them first of all)
    ("transatt_unet", {}, 2, 64),
    ("unet_transformer", {}, 2, 64),
    ("multiresunet", {}, 2, 64),
    ("uctransnet", {"image_size": 64}, 2, 64),
]


def _make(name, kw, dtype):
    torch.manual_seed(0)
    if name == "missformer":
        from unet_zoo_amd.models import MISSFormer
        m = MISSFormer(num_classes=1, in_channels=3, **kw)     # the registry drops image_size (as the reference does)
    else:
        m = unet_zoo_amd.create_model(name, in_channels=3, num_classes=1, **kw)
    m.run_dtype = dtype
    if name == "uctransnet":          # dropout draws from the generator: the two runs would see different masks
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
    return m.cuda().train()


def _batch(b, size, seed=1):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(b, 3, size, size, generator=g).cuda(), (torch.rand(b, 1, size, size, generator=g) > 0.5).float().cuda())


@pytest.mark.parametrize("name,kw,b,size", CASES, ids=[c[0] for c in CASES])
def test_replayed_step_equals_eager_step_bitwise(name, kw, b, size):
    x, t = _batch(b, size)
    # graphed
    m1 = _make(name, kw, torch.bfloat16)
    gs = unet_zoo_amd.GraphedStep(m1, "bce_dice", lr=1e-3, weight_decay=1e-5, max_norm=1.0)
    g_losses, g_norms, g_dice = [], [], []
    for _ in range(3):
        loss = gs(x, t)
        torch.cuda.synchronize()
        g_losses.append(loss.item())
        g_dice.append(gs.dice.item())
        g_norms.append(gs.grad_norm.item())
    # eager: the same kernels through the ordinary autograd node, the same flat optimizer in the same order
    m2 = _make(name, kw, torch.bfloat16)
    n1 = {id(p): n for n, p in m1.named_parameters()}
    p2 = dict(m2.named_parameters())
    opt = FlatClipAdamW([p2[n1[id(p)]] for p in gs.opt.params], lr=1e-3, weight_decay=1e-5, max_norm=1.0)
    m2._pack_cache.repoint()
    m2.grads_in_place = True
    e_losses, e_norms, e_dice = [], [], []
    for _ in range(3):
        loss, dice = loss_and_dice(m2(x), t)
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        e_losses.append(loss.item())
        e_dice.append(dice.item())
        e_norms.append(opt.last_grad_norm().item())
    assert g_losses == e_losses, (g_losses, e_losses)
    assert g_dice == e_dice
    assert g_norms == e_norms, (g_norms, e_norms)
    assert torch.equal(gs.opt.flat_p, opt.flat_p)
    assert all(l == l and 0.0 < l < 20.0 for l in g_losses)
    # buffers (BatchNorm running statistics, counters): the set-up's dry run must not have left a trace
    for (k, a), (_, bb) in zip(m1.named_buffers(), m2.named_buffers()):
        assert torch.equal(a, bb), k


def test_callable_criterion_runs_eagerly_between_the_graphs():
    x, t = _batch(2, 64)
    m1, m2 = _make("unet", {}, torch.float32), _make("unet", {}, torch.float32)
    fused = unet_zoo_amd.GraphedStep(m1, "bce_dice", lr=1e-3)
    plain = unet_zoo_amd.GraphedStep(m2, torch.nn.BCEWithLogitsLoss(), lr=1e-3)
    for i in range(3):
        a, b = fused(x, t), plain(x, t)
        torch.cuda.synchronize()
        assert abs(a.item() - b.item()) < 2e-6 * max(1.0, abs(a.item())), (i, a.item(), b.item())
    assert plain.dice is None and fused.dice is not None
    rel = (fused.opt.flat_p - _reorder(plain, fused)).abs().max() / fused.opt.flat_p.abs().max()
    assert rel < 1e-5, rel
    assert "eager criterion" in plain.describe() and "eager" not in fused.describe()


def _reorder(src, like):
    """src's flat parameters in like's order (both models have identical names)"""
    names_like = {id(p): n for n, p in like.model.named_parameters()}
    by_name = dict(src.model.named_parameters())
    out = torch.zeros_like(like.opt.flat_p)
    for p, (a0, _) in zip(like.opt.params, like.opt.spans):
        q = by_name[names_like[id(p)]]
        out[a0:a0 + q.numel()] = q.detach().reshape(-1)
    return out


def test_new_shape_new_graphs_and_lr_change_and_host_inputs():
    m = _make("unet", {}, torch.bfloat16)
    gs = unet_zoo_amd.GraphedStep(m, "bce_dice", lr=1e-3)
    x, t = _batch(2, 64)
    l0 = gs(x.cpu(), t.cpu()).item()            # host tensors are copied into the static buffers
    x2, t2 = _batch(1, 32, seed=3)
    gs(x2, t2)                                  # a second (input, target) shape: captured on first use
    assert gs.outputs.shape == (1, 1, 32, 32) and len(gs._graphs) == 2
    l1 = gs(x, t).item()
    assert l1 < l0
    gs.set_lr(0.0)
    before = gs.opt.flat_p.clone()
    gs(x, t)
    torch.cuda.synchronize()
    # lr = 0: AdamW's update and its decoupled weight decay are both scaled by the rate
    assert torch.equal(before, gs.opt.flat_p)
    with torch.no_grad():                       # evaluation keeps using model(img) on the same parameters
        m.eval()
        y = m(x)
        m.train()
    assert y.shape == (2, 1, 64, 64) and torch.isfinite(y).all()
    m.eval()
    with pytest.raises(RuntimeError):
        unet_zoo_amd.GraphedStep(m)(x, t)


def test_loss_in_the_graph_is_the_loss_of_the_replayed_logits():
    """round-1 finding 'loss 0.0 in graph mode': with the library reduction gone from the captured region the
    number read from the graph equals the eager re-evaluation of the same replay's logits, replay after replay"""
    m = _make("swin_unet_v2", {"image_size": 128, "window_size": 4}, torch.bfloat16)   # stochastic depth on
    gs = unet_zoo_amd.GraphedStep(m, "bce_dice", lr=1e-4)
    x, t = _batch(4, 128)
    for i in range(8):
        loss = gs(x, t)
        torch.cuda.synchronize()
        want = F.binary_cross_entropy_with_logits(gs.outputs, t).item()
        assert abs(loss.item() - want) < 1e-5, (i, loss.item(), want)
        assert loss.item() > 0.1


def test_capture_check_sees_memset_nodes():
    """step._memset_nodes(): the guard GraphedStep applies to every graph it captures must really list the nodes -- a
    multi-block torch reduction resets its semaphores with hipMemsetAsync (the node that acts only in the first replay
    on this stack), an elementwise op does not"""
    from unet_zoo_amd import step as S
    big = torch.randn(1 << 24, device="cuda")
    out = torch.zeros((), device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        big.sum()                                   # warm-up outside capture (allocations, lazy init)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g1 = S._new_graph()
    with torch.cuda.graph(g1, capture_error_mode=S.CAPTURE_MODE):
        out.copy_(big.sum())
    n1 = S._memset_nodes(g1)
    g2 = S._new_graph()
    with torch.cuda.graph(g2, capture_error_mode=S.CAPTURE_MODE):
        big.mul_(1.0)
    n2 = S._memset_nodes(g2)
    assert n1 is not None and n2 is not None, "the runtime handle of a captured graph could not be inspected"
    assert n2 == 0
    assert n1 >= 1, "a split reduction without a memset node: the canary of DESIGN.md 5a no longer applies -- re-check"
    with pytest.raises(RuntimeError, match="memset node"):
        S._check_capture(g1, "test graph")
