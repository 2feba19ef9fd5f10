import torch, sys
from opendpd_amd import CoreModel, _lib
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
lib = _lib.load()
for occ in (1, 2):
    lib.odpd_set_tuning(b"s16_min_batch", 0); lib.odpd_set_tuning(b"s16_occupancy", occ)
    for B in (16384, 32768, 65536):
        net = CoreModel(2, 13, 1, "dgru").cuda(); opt = FusedAdamW(net, lr=1e-4)
        x = torch.rand(B, 200, 2, device="cuda") * 0.8 + 0.05; t = torch.rand(B, 200, 2, device="cuda")
        for _ in range(2): fused_train_step(opt, x, t, "l2", 200.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fused_train_step(opt, x, t, "l2", 200.0)
        e1.record(); torch.cuda.synchronize()
        print(f"occ{occ} B={B} {e0.elapsed_time(e1)/5:.3f} ms")
