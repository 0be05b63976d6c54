"""Child of tests/test_a_two_rank_gpu.py::test_rccl_branch_of_graphed_step_at_world_1 (started by unet_zoo_amd.launch with
WORLD_SIZE = 1): GraphedStep's data-parallel launch strategy over the `nccl` backend (= RCCL) -- backward phases as
hipGraphs, one asynchronous all-reduce(AVG) per phase span with its work handle, the parameter broadcast -- against the
single-graph step of the same model in the same process.  With one rank the average is the identity, so gradients,
losses and updated parameters must agree bit for bit; what runs is RCCL's real kernels on the real stream ordering."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import unet_zoo_amd  # noqa: E402
from unet_zoo_amd import launch  # noqa: E402


def build(name, size, dtype, dev):
    torch.manual_seed(0)
    kw = {"image_size": size, "window_size": 4, "drop_path_rate": 0.0} if name == "swin_unet_v2" else {}
    m = unet_zoo_amd.create_model(name, in_channels=3, num_classes=1, **kw)
    m.run_dtype = torch.float32 if dtype == "fp32" else torch.bfloat16
    return m.to(dev).train()


def run(gs, model, x, t):
    loss1 = float(gs.forward_backward(x, t))
    torch.cuda.synchronize()
    names = {id(p): n for n, p in model.named_parameters()}
    grads = {names[id(p)]: p.grad.detach().clone() for p in gs.opt.params}
    gs.optimizer_step()
    loss2 = float(gs(x, t))
    torch.cuda.synchronize()
    params = {names[id(p)]: p.detach().clone() for p in gs.opt.params}
    return loss1, loss2, grads, params


def main():
    out_path, name, size, batch, dtype = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    rank, local_rank, world = launch.rank_info()
    assert world == 1
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(batch, 3, size, size, generator=g).to(dev)
    t = (torch.rand(batch, 1, size, size, generator=g) > 0.5).float().to(dev)

    m0 = build(name, size, dtype, dev)
    ref = run(unet_zoo_amd.GraphedStep(m0, "bce_dice", lr=1e-3, weight_decay=1e-5), m0, x, t)
    m1 = build(name, size, dtype, dev)
    gs = unet_zoo_amd.GraphedStep(m1, "bce_dice", lr=1e-3, weight_decay=1e-5, data_parallel=True, phases=3)
    assert gs.distributed and gs._nccl and gs.world == 1
    got = run(gs, m1, x, t)
    n_phases = len(gs._cuts) - 1
    bad_g = [n for n in ref[2] if not torch.equal(ref[2][n], got[2][n])]
    bad_p = [n for n in ref[3] if not torch.equal(ref[3][n], got[3][n])]
    # the other collective schedule -- one all-reduce of the whole buffer after the last phase -- on the same graphs,
    # and the timing pass that chooses between the two
    m2 = build(name, size, dtype, dev)
    gs2 = unet_zoo_amd.GraphedStep(m2, "bce_dice", lr=1e-3, weight_decay=1e-5, data_parallel=True, phases=3, comm="tail")
    tail = run(gs2, m2, x, t)
    bad_g += [n + " (tail)" for n in ref[2] if not torch.equal(ref[2][n], tail[2][n])]
    bad_p += [n + " (tail)" for n in ref[3] if not torch.equal(ref[3][n], tail[3][n])]
    # the bf16 gradient exchange (comm_dtype = torch.bfloat16: all-to-all of bf16 shards, fp32 sum on arrival, all-gather) over
    # RCCL, both schedules: with one rank the result is the local round trip through bf16, bit for bit
    bad_bf16 = []
    for mode in ("overlap", "tail"):
        m3 = build(name, size, dtype, dev)
        gs3 = unet_zoo_amd.GraphedStep(m3, "bce_dice", lr=1e-3, weight_decay=1e-5, data_parallel=True, phases=3, comm=mode,
                                       comm_dtype=torch.bfloat16)
        float(gs3.forward_backward(x, t))
        torch.cuda.synchronize()
        names3 = {id(p): n for n, p in m3.named_parameters()}
        for p in gs3.opt.params:
            n = names3[id(p)]
            if not torch.equal(p.grad, ref[2][n].to(torch.bfloat16).float()):
                bad_bf16.append(f"{n} ({mode})")
        gs3.optimizer_step()
        float(gs3(x, t))            # a second step through the same buffers and streams
        torch.cuda.synchronize()
        assert "bf16" in gs3.describe()
    tuning = gs2.autotune_comm(x, t, steps=2)
    assert set(tuning) == {"overlap", "tail"} and gs2.comm in ("overlap", "tail") and "all-reduce" in gs2.describe()
    torch.save({"loss": (ref[0], got[0]), "loss2": (ref[1], got[1]), "bad_grads": bad_g, "bad_params": bad_p,
                "bad_bf16": bad_bf16, "n_phases": n_phases, "spans": gs._spans, "n_params": len(ref[2]), "describe": gs.describe(),
                "backend": dist.get_backend()}, out_path)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
