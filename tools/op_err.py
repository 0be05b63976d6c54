"""Debug helper: print actual relative errors of individual kernels at realistic sizes (fp32)."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from unet_zoo_amd import _lib as L, ops
from unet_zoo_amd.ops import act_from_nchw

dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
def rnd(t): return t.to(dt).float()
DEV = "cuda"
def relerr(a, b): return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()
def l2rel(a, b): return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()

for (N, H, W, Cin, Cout) in [(2, 64, 64, 64, 64), (2, 128, 128, 64, 64), (2, 64, 64, 128, 64), (2, 128, 128, 128, 64), (2, 16, 16, 256, 256)]:
    g = torch.Generator().manual_seed(1)
    x = rnd(torch.randn(N, Cin, H, W, generator=g).relu()).requires_grad_(True)
    w = rnd(torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).requires_grad_(True)
    dy = torch.randn(N, Cout, H, W, generator=g)
    dy = rnd(dy - dy.mean((0, 2, 3), keepdim=True))
    y = F.conv2d(x, w, None, padding=1)
    y.backward(dy)
    dya = act_from_nchw(dy.to(DEV), dt); xa = act_from_nchw(x.detach().to(DEV), dt)
    yy = ops.new_act(N, H, W, Cout, dt, DEV)
    ops.conv_igemm(xa, ops.pack_weights(w.detach().to(DEV), L.PACK_CONV_FWD, dt), None, yy, ntaps=9)
    dx = ops.new_act(N, H, W, Cin, dt, DEV)
    ops.conv_igemm(dya, ops.pack_weights(w.detach().to(DEV), L.PACK_CONV_DGRAD, dt), None, dx, ntaps=9)
    dw = ops.wgrad(dya, xa, (Cout, Cin, 3, 3), ntaps=9)
    print(N, H, W, Cin, Cout, "fwd", relerr(yy.dense().cpu(), y.detach()), "dgrad", relerr(dx.dense().cpu(), x.grad),
          "wgrad max", relerr(dw.cpu(), w.grad), "l2", l2rel(dw.cpu(), w.grad))

for (N, H, W, C) in [(2, 64, 64, 64), (2, 128, 128, 64)]:
    g = torch.Generator().manual_seed(6)
    y = rnd(torch.randn(N, C, H, W, generator=g) * 2 + 0.5).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.1).requires_grad_(True)
    act_ref = F.relu(F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5))
    g0 = rnd(torch.randn(N, C, H, W, generator=g))
    (act_ref * g0).sum().backward()
    ya = act_from_nchw(y.detach().to(DEV), dt)
    yd = ya.buf.double()
    stats = torch.stack([yd.sum(0), (yd ** 2).sum(0)]).float().reshape(1, 2, C)
    vec = ops.bn_finalize(stats, N * H * W, gamma.detach().to(DEV), beta.detach().to(DEV), 1e-5, 0.1, None, None)
    act = ops.new_act(N, H, W, C, dt, DEV)
    ops.bn_relu_apply(ya, vec[0], vec[1], act, None)
    sums = torch.zeros(2, C, dtype=torch.float64, device=DEV)
    dyo = ops.new_act(N, H, W, C, dt, DEV)
    dgb = torch.empty(2, C, device=DEV)
    ops.bn_relu_bwd(ya, vec, act_from_nchw(g0.to(DEV), dt), None, None, sums, dyo, dgb[0], dgb[1])
    print("bn", N, H, W, C, "act", relerr(act.dense().cpu(), act_ref.detach()), "dy max", relerr(dyo.dense().cpu(), y.grad),
          "l2", l2rel(dyo.dense().cpu(), y.grad), "dgamma", relerr(dgb[0].cpu(), gamma.grad), "dbeta", relerr(dgb[1].cpu(), beta.grad))
