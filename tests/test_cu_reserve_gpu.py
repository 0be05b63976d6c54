"""GPU: uz_set_cu_reserve() -- the persistent grids of the convolution / GEMM / weight-gradient kernels sized for
256 - n CUs, so that RCCL's all-reduce kernels find a CU beside the backward (reference seam
unet_zoo/utils/multi_gpu.py:20-31).  The tensors a kernel writes do not depend on its grid; only the partition of the
per-workgroup fp32 partial sums (BatchNorm statistics, weight-gradient slabs) does."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import unet_zoo_amd
from unet_zoo_amd import _lib as L
from unet_zoo_amd import ops
from oracle import torch_ref

DEV = "cuda"


@pytest.fixture(autouse=True)
def _restore():
    yield
    L.set_cu_reserve(0)


def test_reserve_is_validated_and_reported():
    L.set_cu_reserve(8)
    assert L.get_cu_reserve() == 8
    with pytest.raises(Exception):
        L.set_cu_reserve(-1)
    with pytest.raises(Exception):
        L.set_cu_reserve(200)
    assert L.get_cu_reserve() == 8


@pytest.mark.parametrize("N,S,Cin,Cout", [(16, 64, 256, 256), (16, 32, 512, 512), (4, 128, 64, 64)])
def test_conv_outputs_do_not_depend_on_the_reserve(N, S, Cin, Cout):
    dt = torch.bfloat16
    gen = torch.Generator(device=DEV).manual_seed(41)
    x = ops.new_act(N, S, S, Cin, dt, DEV)
    x.buf.copy_(torch.randn(x.buf.shape, generator=gen, device=DEV))
    dy = ops.new_act(N, S, S, Cout, dt, DEV)
    dy.buf.copy_(torch.randn(dy.buf.shape, generator=gen, device=DEV))
    w = torch.randn(Cout, Cin, 3, 3, generator=gen, device=DEV) * 0.05
    wp = ops.pack_weights(w, L.PACK_CONV_FWD, dt)
    res = []
    for r in (0, 8, 16):
        L.set_cu_reserve(r)
        y = ops.new_act(N, S, S, Cout, dt, DEV)
        st = ops.conv_igemm(x, wp, None, y, ntaps=9, want_stats=True)
        dw = ops.wgrad(dy, x, (Cout, Cin, 3, 3), ntaps=9)
        res.append((y.buf.clone(), st.double().sum(0), st.shape[0], dw.clone()))
    assert res[1][2] <= 248 and res[2][2] <= 240            # fewer partial rows = fewer workgroups
    for k in (1, 2):
        assert torch.equal(res[0][0], res[k][0])            # the convolution output: bit for bit
        assert ((res[0][1] - res[k][1]).abs().max() / res[0][1].abs().max()).item() < 1e-6
        assert ((res[0][3] - res[k][3]).abs().max() / res[0][3].abs().max()).item() < 1e-5


def test_train_step_with_a_reserve_matches_the_full_chip():
    x, mask = torch_ref.synthetic_batch(4, 3, 128, 128, seed=42)
    x, mask = x.to(DEV), mask.to(DEV)
    outs = []
    for r in (0, 16):
        L.set_cu_reserve(r)
        torch.manual_seed(0)
        m = unet_zoo_amd.create_model("unet", in_channels=3, num_classes=1)
        m.run_dtype = torch.bfloat16
        m = m.to(DEV).train()
        out = m(x)
        loss = F.binary_cross_entropy_with_logits(out, mask)
        loss.backward()
        torch.cuda.synchronize()
        outs.append((float(loss), out.detach().float().clone(),
                     torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None]).clone()))
    assert abs(outs[0][0] - outs[1][0]) < 1e-4
    assert ((outs[0][1] - outs[1][1]).abs().max() / outs[0][1].abs().max()).item() < 2e-2
    g0, g1 = outs[0][2].double(), outs[1][2].double()
    assert (torch.dot(g0, g1) / (g0.norm() * g1.norm())).item() > 0.999
