"""GPU, one rank over RCCL: the `nccl` branch of GraphedStep's data-parallel strategy in a fresh child process (started
by unet_zoo_amd.launch before THIS process has touched the GPU -- the file sorts before every other GPU test on purpose;
a parent that has initialised the GPU must not start rank processes)."""
import io
import os
import sys

import pytest
import torch

from unet_zoo_amd import launch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("model_name,size,batch,dtype", [("unet", 64, 4, "bf16"), ("swin_unet_v2", 64, 4, "fp32")])
def test_rccl_branch_of_graphed_step_at_world_1(tmp_path, model_name, size, batch, dtype):
    """GraphedStep(data_parallel=True, phases=3) over the `nccl` backend (RCCL) in a fresh one-rank process: the
    phase graphs, the asynchronous all-reduce(AVG) of every span and their work handles, the broadcast at set-up --
    gradients, losses and the parameters after two optimizer steps must equal the single-graph step bit for bit
    (the reference seam: nn.DataParallel wrapping, unet_zoo/utils/multi_gpu.py:20-31)"""
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU; run this file first (it sorts first) or alone")
    out = os.path.join(tmp_path, "w1.pt")
    serr = io.StringIO()
    rc = launch.spawn_ranks(1, [sys.executable, os.path.join(HERE, "_nccl_world1_step.py"), out, model_name, str(size),
                                str(batch), dtype], need_gpus=False, stdout=io.StringIO(), stderr=serr)
    assert rc == 0, serr.getvalue()[-4000:]
    got = torch.load(out)
    assert got["backend"] == "nccl" and got["n_phases"] >= 2 and len(got["spans"]) == got["n_phases"]
    assert "RCCL" in got["describe"]
    assert got["n_params"] > 10
    assert got["loss"][0] == got["loss"][1] and got["loss2"][0] == got["loss2"][1]
    assert not got["bad_grads"], got["bad_grads"][:5]
    assert not got["bad_params"], got["bad_params"][:5]
    # comm_dtype = bf16: one rank's exchange is the local bf16 round trip of the fp32 gradients, bit for bit
    assert not got["bad_bf16"], got["bad_bf16"][:5]
