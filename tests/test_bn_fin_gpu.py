"""GPU: the BatchNorm finalize riding in its consumer's launch (uz_bn_relu_add_apply_fin, uz_bn_relu_bwd_apply_fin, round 5)
against the launches it replaces (uz_bn_finalize + uz_bn_relu_add_apply; uz_bn_bwd_finalize + uz_bn_relu_bwd_apply): the
same row groups and pairing tree, so EVERY output must be equal bit for bit -- scale / shift / mean / invstd, the running
statistics, the activation and its pool, the totals, dgamma / dbeta and dy.  Reference: BatchNorm2d + ReLU [+ MaxPool2d]
in training mode and their autograd (common_layers.py:29-32, :90).  Cases: both row-group widths of the finalize (rows >= 128
with C <= 512, and the other), a grid smaller than the finalize (falls back to the two launches), pool / residual / no ReLU,
fp32, and 30 repeats of one case on re-zeroed flags (a lost release would show as a stale vector)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from unet_zoo_amd import _lib as L
from unet_zoo_amd import ops

DEV = "cuda"

CASES = [
    # dtype, N, H, W, C, rows, pool, residual, relu
    (torch.bfloat16, 4, 64, 64, 64, 256, False, False, True),     # EW 8
    (torch.bfloat16, 4, 64, 64, 64, 256, True, False, True),
    (torch.bfloat16, 2, 33, 47, 128, 130, True, True, True),      # odd sizes, ceil pool, residual
    (torch.bfloat16, 2, 16, 16, 1024, 64, False, False, True),    # EW 32: 32 finalizing workgroups
    (torch.bfloat16, 1, 2, 2, 1024, 16, False, False, True),      # grid (2 workgroups) < finalize (32): the two launches
    (torch.bfloat16, 3, 24, 40, 96, 200, False, False, False),    # BatchNorm without ReLU
    (torch.float32, 2, 32, 32, 64, 128, True, False, True),
]


def make(dt, N, H, W, C, rows, g):
    y = ops.new_act(N, H, W, C, dt, DEV)
    y.buf.copy_(torch.randn(N * H * W, C, generator=g).to(dt))
    stats = torch.randn(rows, 2, C, generator=g).to(DEV)
    stats[:, 1] = stats[:, 1].abs() * 40 + 60          # sum of squares large enough for a positive variance
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    return y, stats, gamma, beta


@pytest.mark.parametrize("dt,N,H,W,C,rows,pool,residual,relu", CASES)
def test_forward_finalize_inside_the_apply_launch(dt, N, H, W, C, rows, pool, residual, relu):
    g = torch.Generator().manual_seed(C + rows)
    y, stats, gamma, beta = make(dt, N, H, W, C, rows, g)
    count = rows * 17
    ceil_mode = bool(H % 2 or W % 2)
    res = None
    if residual:
        res = ops.new_act(N, H, W, C, dt, DEV)
        res.buf.copy_(torch.randn(N * H * W, C, generator=g).to(dt))

    def outputs():
        act = ops.new_act(N, H, W, C, dt, DEV)
        pooled = ops.new_act(N, (H + 1) // 2 if ceil_mode else H // 2, (W + 1) // 2 if ceil_mode else W // 2, C, dt, DEV) if pool else None
        return act, pooled

    rm0, rv0 = torch.randn(C, device=DEV), torch.rand(C, device=DEV) + 0.5
    # the two launches
    rm1, rv1 = rm0.clone(), rv0.clone()
    vec1 = ops.bn_finalize(stats, count, gamma, beta, 1e-5, 0.1, rm1, rv1)
    a1, p1 = outputs()
    ops.bn_relu_apply(y, vec1[0], vec1[1], a1, p1, res, ceil_mode, relu=relu)
    # one launch
    for rep in range(30 if (C, pool) == (64, False) else 1):
        rm2, rv2 = rm0.clone(), rv0.clone()
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        a2, p2 = outputs()
        vec2 = ops.bn_relu_apply_fin(y, stats, count, gamma, beta, 1e-5, 0.1, rm2, rv2, flag, a2, p2, res, ceil_mode, relu=relu)
        torch.cuda.synchronize()
        assert torch.equal(vec1, vec2), rep
        assert torch.equal(rm1, rm2) and torch.equal(rv1, rv2)
        assert torch.equal(a1.buf, a2.buf), rep
        if pool:
            assert torch.equal(p1.buf, p2.buf)
        ew = 8 if (rows >= 128 and C <= 512) else 32
        nfin = (C + ew - 1) // ew
        assert flag.item() in (nfin, 0)            # 0: the fallback did not touch it
        if (N, H, W) == (1, 2, 2):
            assert flag.item() == 0


@pytest.mark.parametrize("dt,N,H,W,C,rows,pool,residual,relu", CASES)
def test_backward_finalize_inside_the_apply_launch(dt, N, H, W, C, rows, pool, residual, relu):
    g = torch.Generator().manual_seed(C + rows + 1)
    y, _, gamma, beta = make(dt, N, H, W, C, rows, g)
    ceil_mode = bool(H % 2 or W % 2)
    vec = torch.stack([gamma, beta, torch.randn(C, generator=g).to(DEV) * 0.1, (torch.rand(C, generator=g) + 0.5).to(DEV)])
    g0 = ops.new_act(N, H, W, C, dt, DEV)
    g0.buf.copy_(torch.randn(N * H * W, C, generator=g).to(dt))
    gp = None
    if pool and relu:
        gp = ops.new_act(N, (H + 1) // 2 if ceil_mode else H // 2, (W + 1) // 2 if ceil_mode else W // 2, C, dt, DEV)
        gp.buf.copy_(torch.randn(gp.P, C, generator=g).to(dt))

    def run(flag, partials):
        sums = torch.zeros(2, C, dtype=torch.float64, device=DEV)
        dy = ops.new_act(N, H, W, C, dt, DEV)
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        ops.bn_relu_bwd(y, vec, g0, None, gp, sums, dy, dg, db, ceil_mode, relu=relu, partials=partials, fin_flag=flag)
        torch.cuda.synchronize()
        return sums, dy.buf, dg, db

    # (a) the rows of the reduce pass
    ref = run(None, None)
    for rep in range(30 if (C, pool) == (64, False) else 1):
        got = run(torch.zeros(1, dtype=torch.int32, device=DEV), None)
        for a, b in zip(ref, got):
            assert torch.equal(a, b), rep
    # (b) rows as a convolution's epilogue leaves them (uz_conv_igemm_bnred): only without pool / second gradient
    if gp is None and relu:
        parts = torch.randn(rows, 2, C, generator=g).to(DEV)
        ref = run(None, parts)
        got = run(torch.zeros(1, dtype=torch.int32, device=DEV), parts)
        for a, b in zip(ref, got):
            assert torch.equal(a, b)


@pytest.mark.parametrize("name", ["unet", "u2net", "resunet"])
def test_a_training_step_is_the_same_with_and_without_the_fused_finalize(name):
    """B=2 64x64 bf16: logits, loss, every parameter gradient and every BatchNorm buffer bit for bit"""
    import unet_zoo_amd
    from unet_zoo_amd.engine import Engine
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(1)).to(DEV)
    t = (torch.rand(2, 1, 64, 64, generator=torch.Generator().manual_seed(2)) > 0.5).float().to(DEV)
    runs = []
    for fused in (True, False):
        torch.manual_seed(0)
        m = unet_zoo_amd.create_model(name, in_channels=3, num_classes=1)
        m.run_dtype = torch.bfloat16
        m = m.cuda().train()
        old = Engine.fuse_bn_finalize
        Engine.fuse_bn_finalize = fused
        try:
            out = m(x)
            outs = list(out.values()) if isinstance(out, dict) else list(out) if isinstance(out, (list, tuple)) else [out]
            outs = [o for o in outs if torch.is_tensor(o)] if not all(torch.is_tensor(o) for o in outs) else outs
            flat = []
            for o in outs:
                flat += list(o) if isinstance(o, (list, tuple)) else [o]
            logits = flat[0]
            loss = sum(torch.nn.functional.binary_cross_entropy_with_logits(o.float(), t) for o in flat)
            loss.backward()
            torch.cuda.synchronize()
        finally:
            Engine.fuse_bn_finalize = old
        runs.append((m, loss.detach().clone(), logits.detach().clone()))
    (m1, l1, o1), (m2, l2, o2) = runs
    assert torch.equal(o1, o2) and torch.equal(l1, l2)
    for (n1, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert (p1.grad is None) == (p2.grad is None), n1
        if p1.grad is not None:
            assert torch.equal(p1.grad, p2.grad), n1
    for (n1, b1), (_, b2) in zip(m1.named_buffers(), m2.named_buffers()):
        assert torch.equal(b1, b2), n1


@pytest.mark.parametrize("dt,N,H,W,C,rows,pool,residual,relu", CASES)
def test_walking_from_the_end_gives_the_same_values(dt, N, H, W, C, rows, pool, residual, relu):
    """bit 2 of the passes' flag word (Engine.reverse_element_passes): the apply pass is a pure element pass -- identical bytes;
    the backward's totals are summed in other pixel groups -- equal to fp32 rounding, dy within one rounding of the tensor type"""
    g = torch.Generator().manual_seed(C + 7)
    y, _, gamma, beta = make(dt, N, H, W, C, rows, g)
    ceil_mode = bool(H % 2 or W % 2)
    res = None
    if residual:
        res = ops.new_act(N, H, W, C, dt, DEV)
        res.buf.copy_(torch.randn(N * H * W, C, generator=g).to(dt))
    outs = []
    for rev in (False, True):
        act = ops.new_act(N, H, W, C, dt, DEV)
        pooled = ops.new_act(N, (H + 1) // 2 if ceil_mode else H // 2, (W + 1) // 2 if ceil_mode else W // 2, C, dt, DEV) if pool else None
        ops.bn_relu_apply(y, gamma, beta, act, pooled, res, ceil_mode, relu=relu, reverse=rev)
        outs.append((act.buf.clone(), pooled.buf.clone() if pool else None))
    assert torch.equal(outs[0][0], outs[1][0])
    if pool:
        assert torch.equal(outs[0][1], outs[1][1])

    vec = torch.stack([gamma, beta, torch.randn(C, generator=g).to(DEV) * 0.1, (torch.rand(C, generator=g) + 0.5).to(DEV)])
    g0 = ops.new_act(N, H, W, C, dt, DEV)
    g0.buf.copy_(torch.randn(N * H * W, C, generator=g).to(dt))
    gp = None
    if pool and relu:
        gp = ops.new_act(N, (H + 1) // 2 if ceil_mode else H // 2, (W + 1) // 2 if ceil_mode else W // 2, C, dt, DEV)
        gp.buf.copy_(torch.randn(gp.P, C, generator=g).to(dt))
    res = []
    for rev in (False, True):
        sums = torch.zeros(2, C, dtype=torch.float64, device=DEV)
        dy = ops.new_act(N, H, W, C, dt, DEV)
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        ops.bn_relu_bwd(y, vec, g0, None, gp, sums, dy, dg, db, ceil_mode, relu=relu, reverse=rev)
        torch.cuda.synchronize()
        res.append((sums.clone(), dy.buf.float().clone(), dg.clone(), db.clone()))
    assert torch.allclose(res[0][0], res[1][0], rtol=1e-5, atol=1e-4)
    assert torch.allclose(res[0][2], res[1][2], rtol=1e-5, atol=1e-4) and torch.allclose(res[0][3], res[1][3], rtol=1e-5, atol=1e-4)
    tol = 1e-5 if dt == torch.float32 else 2.0 ** -7
    assert ((res[0][1] - res[1][1]).abs() <= tol * (res[0][1].abs() + 1e-2 * res[0][1].abs().max())).all()
