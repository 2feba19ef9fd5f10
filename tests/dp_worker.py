"""One rank of the data-parallel tests (tests/test_dp_gpu.py starts it under torch.distributed.run).  Also imported by the test module
for the inputs and the single-process reference, so that both sides build exactly the same model, streams and loader.

    python dp_worker.py <job.json>      job: backend (gloo | nccl), share_gpu (every rank on device 0), out (path prefix of the
                                        per-spec, per-rank .npz), specs: [ {mode (step | epoch | cascade_step | cascade_epoch),
                                        model fields, B, T, n} ... ] run one after the other on one process group
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def stream(n, seed, device="cuda"):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(n, 2, generator=g) - 0.5) * 1.4
    x = x + 0.05 * torch.sign(x)
    y = x * (1.0 - 0.2 * (x * x).sum(-1, keepdim=True)) + 0.05 * torch.roll(x, 1, 0)
    return x.contiguous().to(device), y.contiguous().to(device)


class Loader:
    """the attributes FusedAdamW.train_epoch reads from project.DeviceFrameLoader"""

    def __init__(self, x, y, T, batch, seed):
        self.x, self.y, self.frame_length, self.stride, self.batch_size = x, y, T, 1, batch
        self.n = x.shape[0] - T + 1
        self._order = torch.randperm(self.n, generator=torch.Generator().manual_seed(seed)).to(x.device)

    def epoch_order(self):
        return self._order


def build(spec, device):
    """(net, optimiser) of the spec: a single backbone, or a DPD in front of a frozen PA"""
    from opendpd_amd import CascadedModel, CoreModel
    from opendpd_amd.train_funcs import FusedAdamW
    torch.manual_seed(0)
    if spec["mode"].startswith("cascade"):
        net = CascadedModel(dpd_model=CoreModel(2, spec["H"], 1, spec["bb"], thx=0.01, thh=0.05), pa_model=CoreModel(2, spec["pa_H"], 1, spec["pa_bb"]))
        net.freeze_pa_model()
    else:
        net = CoreModel(2, spec["H"], 1, spec["bb"])
    net = net.to(device)
    return net, FusedAdamW(net, lr=1e-3)


def trained(net):
    return (net.dpd_model if hasattr(net, "dpd_model") else net).backbone


def run(spec, device, rank=0, world=1):
    """the spec's work on this rank: returns dict(losses, params, grad)"""
    from opendpd_amd import dist as odist
    from opendpd_amd.train_funcs import fused_train_step
    net, opt = build(spec, device)
    T, B = spec["T"], spec["B"]
    if spec["mode"] in ("step", "cascade_step"):
        g = torch.Generator().manual_seed(7)
        x = (torch.rand(B, T, 2, generator=g) - 0.5) * 1.4
        x = (x + 0.05 * torch.sign(x)).to(device)
        t = (torch.rand(B, T, 2, generator=g) - 0.5).to(device)
        lo, hi = odist.shard_range(B, rank, world)
        losses = []
        for _ in range(spec.get("steps", 1)):
            if hi > lo:
                losses.append(fused_train_step(opt, x[lo:hi].contiguous(), t[lo:hi].contiguous(), "l2", 200.0, global_count=B * T * 2))
            else:
                losses.append(opt.empty_step(200.0, B * T * 2))
        losses = torch.stack(losses)
    else:
        xs, ys = stream(spec["n"], 3, device)
        loader = Loader(xs, ys, T, B, seed=5)
        if spec["mode"] == "epoch":
            assert opt.can_run_epoch(loader), "no native epoch loop for this spec"
            losses = opt.train_epoch(loader, "l2", 200.0)
        else:
            assert opt.can_run_cascade_epoch(loader), "no native cascade epoch loop for this spec"
            losses = opt.train_epoch_cascade(loader, "l2", 200.0)
    torch.cuda.synchronize()
    comm = opt.native_comm()
    return dict(losses=losses.cpu().numpy(), params=trained(net).flat_params().cpu().numpy().copy(), grad=opt.grad.cpu().numpy().copy(),
                comm=np.array(comm.kind if comm is not None else "torch"), errors=np.array(comm.errors() if comm is not None else 0))


def main():
    job = json.load(open(sys.argv[1]))
    from opendpd_amd import dist as odist
    rank, local, world = odist.env_world()
    dev = torch.device("cuda", 0 if job.get("share_gpu") else local)
    torch.cuda.set_device(dev)
    odist.init(job["backend"], device=dev)
    for i, spec in enumerate(job["specs"]):       # one process group and one communicator for the whole list
        out = run(spec, dev, rank, world)
        np.savez(job["out"] + f"_{i}_{rank}.npz", **out)
    odist.reset_native_comm()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
