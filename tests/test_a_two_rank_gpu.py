"""GPU, two ranks: GraphedStep's data-parallel strategy (backward phases as hipGraphs, each phase's span of the
flat gradient buffer all-reduced while the next phase runs; SURVEY §8e, replacing multi_gpu.py:20-31) against the
single-rank gradients of the two shards.

The file sorts first on purpose: the rank processes are started by unet_zoo_amd.launch, which must happen before
THIS process has initialised the GPU.  With two GPUs the ranks use RCCL; on a one-GPU box both ranks share the card and
reduce through gloo, which still exercises the launcher, the rank wiring, the phase plan, the span layout, the
parameter broadcast and the optimizer on averaged gradients on hardware."""
import io
import os
import sys

import pytest
import torch

from unet_zoo_amd import launch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("model_name,size,batch,dtype", [("unet", 64, 2, "fp32")])
def test_two_rank_graphed_step_equals_mean_of_shard_gradients(tmp_path, model_name, size, batch, dtype):
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU; run this file first (it sorts first) or alone")
    # both rank pairs (fp32 all-reduce; the bf16 exchange option) run BEFORE this process touches the GPU
    results = {}
    for comm in ("fp32", "bf16"):
        out = os.path.join(tmp_path, f"rank0_{comm}.pt")
        serr = io.StringIO()
        rc = launch.spawn_ranks(2, [sys.executable, os.path.join(HERE, "_two_rank_step.py"), out, model_name, str(size),
                                    str(batch), dtype, comm], need_gpus=False, stdout=io.StringIO(), stderr=serr)
        assert rc == 0, serr.getvalue()[-4000:]
        results[comm] = torch.load(out)
    for got in results.values():
        assert got["n_phases"] >= 2 and len(got["spans"]) == got["n_phases"]
        # both ranks hold identical parameters after two steps (same start: rank 1's offset was overwritten by the broadcast)
        assert got["param_sums"][0] == got["param_sums"][1]
    assert "bf16" in results["bf16"]["launch"]

    # single rank: gradients of each shard with the ordinary autograd path, then their mean
    import unet_zoo_amd
    from unet_zoo_amd.loss import loss_and_dice
    sys.path.insert(0, HERE)
    from _two_rank_step import shard
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = unet_zoo_amd.create_model(model_name, in_channels=3, num_classes=1)
    model.run_dtype = torch.float32 if dtype == "fp32" else torch.bfloat16
    model = model.to(dev).train()
    per_rank, losses = [], []
    for r in range(2):
        for p in model.parameters():
            p.grad = None
        x, m = shard(r, batch, size)
        loss, _ = loss_and_dice(model(x.to(dev)), m.to(dev))
        loss.backward()
        per_rank.append({n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None})
        losses.append(float(loss))
    for comm, got in results.items():
        assert abs(got["loss"] - losses[0]) < 1e-6
        assert set(got["grads"]) == set(per_rank[0])
        worst = 0.0
        bf = torch.bfloat16
        for n, g in got["grads"].items():
            if comm == "bf16":      # every rank's span rounded to bf16, fp32 sum, mean, one rounding (GraphedStep(comm_dtype=bf16))
                want = ((per_rank[0][n].to(bf).float() + per_rank[1][n].to(bf).float()) / 2).to(bf).float()
            else:
                want = (per_rank[0][n] + per_rank[1][n]) / 2
            err = (g - want).abs().max().item() / (want.abs().max().item() + 1e-12)
            worst = max(worst, err)
        # (bf16 exchange: the ranks' own fp32 gradients differ from the single-process ones by summation order, ~1e-6, which can
        # move a bf16 rounding: one ulp = 2^-8 relative on a few elements)
        assert worst < (1e-5 if comm == "fp32" else 8e-3), worst

