"""Rank process of tests/test_a_two_rank_gpu.py (started by unet_zoo_amd.launch): GraphedStep with the data-parallel
launch strategy (backward phases + one all-reduce per phase) on this rank's shard; rank 0 saves what the parent
compares with single-rank gradients of the two shards.  RCCL when every rank has its own GPU, otherwise both ranks
share cuda:0 and the collectives go through gloo (staged through the host) -- the phase / span / ordering logic
under test is the same."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import unet_zoo_amd  # noqa: E402
from unet_zoo_amd import launch  # noqa: E402


def shard(rank, batch, size):
    g = torch.Generator().manual_seed(4321 + rank)
    x = torch.randn(batch, 3, size, size, generator=g)
    m = (torch.rand(batch, 1, size, size, generator=g) > 0.5).float()
    return x, m


def main():
    out_path, model_name, size, batch, dtype = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    comm_dtype = torch.bfloat16 if (len(sys.argv) > 6 and sys.argv[6] == "bf16") else None
    rank, local_rank, world = launch.rank_info()
    own_gpu = torch.cuda.device_count() >= world
    dev = torch.device("cuda", local_rank if own_gpu else 0)
    torch.cuda.set_device(dev)
    if own_gpu:
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    torch.manual_seed(0)
    kw = {"image_size": size, "window_size": 4, "drop_path_rate": 0.0} if model_name == "swin_unet_v2" else {}
    model = unet_zoo_amd.create_model(model_name, in_channels=3, num_classes=1, **kw)
    model.run_dtype = torch.float32 if dtype == "fp32" else torch.bfloat16
    model = model.to(dev).train()
    if rank == 1:
        with torch.no_grad():              # rank 1 starts from different weights: set-up must broadcast rank 0's
            for p in model.parameters():
                p.add_(0.5)
    x, m = shard(rank, batch, size)
    gs = unet_zoo_amd.GraphedStep(model, "bce_dice", lr=1e-3, weight_decay=1e-5, phases=3, comm_dtype=comm_dtype)
    assert gs.distributed and gs.world == world
    loss = gs.forward_backward(x.to(dev), m.to(dev))
    torch.cuda.synchronize()
    loss = float(loss)                     # a static tensor of the graph: the next replay overwrites it
    names = {id(p): n for n, p in model.named_parameters()}
    grads = {names[id(p)]: p.grad.detach().cpu().clone() for p in gs.opt.params}
    gs.optimizer_step()
    loss2 = gs(x.to(dev), m.to(dev))       # a second full step through the replayed graphs
    torch.cuda.synchronize()
    params = {names[id(p)]: p.detach().cpu().clone() for p in gs.opt.params}
    gathered = [None] * world
    dist.all_gather_object(gathered, {n: float(v.double().sum()) for n, v in params.items()})
    if rank == 0:
        torch.save({"grads": grads, "loss": float(loss), "loss2": float(loss2), "params": params,
                    "param_sums": gathered, "spans": gs._spans, "n_phases": len(gs._cuts) - 1,
                    "backend": dist.get_backend(), "launch": gs.describe()}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
