"""Lockstep sweeps: K independent `train_pa` runs — the seeds x hidden-size loops of the reference's bash_scripts/train_all_pa.sh:26-57, which
start one process per run — trained TOGETHER on one GPU.  At the reference's batch sizes one run occupies 256 (batch 64: 64) of the chip's
>= 8 192 wave slots and its step is a chain of three launches of ~0.1 ms; here every step of the epoch is ONE fused train launch that carries
all K runs of a model shape (run k owns workgroups [k G, (k + 1) G) and sees exactly the launch it would have had alone), one row reduction
and one clip + AdamW launch (csrc: odpd_train_epoch_sweep), and the evaluation passes of the K models are one launch per batch of segments
(odpd_backbone_fwd_sweep).  Each run is BIT-IDENTICAL to its solo run — same seeded initialisation, same epoch orders (each run keeps its
own copy of the global RNG state, swapped in around everything that draws from it), same kernels, same reduction order — and writes the same
`save/` and `log/` files (tests/test_sweep_gpu.py compares them with solo runs).  Runs whose shape the sweep kernels do not serve (other
families, batches beyond the one-sequence-per-wave regime) advance through the ordinary per-run epoch, in the same lockstep loop."""
import ctypes as C
import random
import time

import numpy as np
import torch

from . import _lib
from .models import CoreModel
from .project import Project, count_net_params
from .train_funcs import FusedAdamW, _loss_kind, _sample_format


class _Rng:
    """one run's private copy of the process-global RNG states (torch CPU, numpy, random): `with run.rng:` swaps it in and back out"""

    def __init__(self):
        self.state = None
        self._outer = None

    @staticmethod
    def _get():
        return torch.get_rng_state(), np.random.get_state(), random.getstate()

    @staticmethod
    def _set(st):
        torch.set_rng_state(st[0]); np.random.set_state(st[1]); random.setstate(st[2])

    def __enter__(self):
        self._outer = self._get()
        if self.state is not None:
            self._set(self.state)
        return self

    def __exit__(self, *exc):
        self.state = self._get()
        self._set(self._outer)
        return False


class _Run:
    def __init__(self, proj):
        self.proj, self.rng = proj, _Rng()
        self.net = self.opt = self.sched = self.loaders = self.criterion = None
        self.partials = self.losses = self.order = self.workspace = None


def _setup(run):
    """steps/train_pa.py:10-59 up to the training loop (project.run_train_pa), under the run's own RNG"""
    p = run.proj
    p.set_device()
    run.loaders, input_size = p.build_dataloaders()
    run.net = CoreModel(input_size, p.PA_hidden_size, p.PA_num_layers, p.PA_backbone, window_size=p.window_size,
                        num_dvr_units=p.num_dvr_units).to(p.device)
    p.build_logger(p.gen_pa_model_id(count_net_params(run.net)))
    run.opt, run.sched = p.build_optimizer(run.net)
    run.criterion = p.build_criterion()


def _group_key(run):
    p = run.proj
    return (p.PA_backbone, p.PA_hidden_size, p.PA_num_layers, p.batch_size, p.frame_length, p.frame_stride, p.loss_type, p.opt_type,
            p.dataset_label, p.frame_storage, p.grad_clip_val)


class _Group:
    """runs of one model shape on one dataset: what one sweep launch can carry"""

    def __init__(self, runs, exact=True):
        self.runs, self.exact = runs, exact
        r0 = runs[0]
        lib = _lib.load()
        train = r0.loaders[0]
        self.T, self.n, self.B = train.frame_length, train.n, min(train.batch_size, train.n)
        self.n_steps = (self.n + self.B - 1) // self.B
        last = self.n - (self.n_steps - 1) * self.B
        bb = r0.net.backbone
        self.desc = getattr(bb, "desc", None)
        fused = all(isinstance(r.opt, FusedAdamW) and r.opt.kind == "adamw" and r.opt.pa is None and r.opt.world_size() == 1
                    and getattr(r.net.backbone, "frozen_mask", None) is None for r in runs)
        kind = _loss_kind(r0.criterion)
        self.kind = kind
        ok = bool(fused and kind is not None and self.desc is not None and len(runs) > 1 and train.x.is_cuda)
        # throughput mode: the 16-sequences-per-wave kernel for every run (K x B / 16 waves fill the chip); exact mode: the kernel a solo run
        # takes at this batch size (its frame state fills a CU's LDS: no more than one run's worth of frames is ever resident)
        self.flags = _lib.SWEEP_S16 if (ok and not exact and train.x.dtype == torch.float32 and lib.odpd_sweep_s16_supported(C.byref(self.desc))) else 0
        self.train_sweep = bool(ok and (self.flags or (lib.odpd_sweep_train_supported(C.byref(self.desc), self.B, self.T)
                                                       and lib.odpd_sweep_train_supported(C.byref(self.desc), last, self.T))))
        self.dev = train.x.device if hasattr(train, "x") else None
        self.scratch = None
        self._tuning_gen = None
        self._size_buffers()

    def _size_buffers(self):
        """partial rows / workspace / scratch of every run, sized for the CURRENT kernel-selection knobs (odpd_set_tuning: gp_max_batch,
        s16_occupancy, s16_min_batch .. change the grids the kernels write); re-done by train_epoch when odpd_tuning_generation has moved
        (a stale buffer under a larger grid would be a device out-of-bounds write)"""
        lib = _lib.load()
        runs, bb = self.runs, self.runs[0].net.backbone
        last = self.n - (self.n_steps - 1) * self.B
        self._tuning_gen = int(lib.odpd_tuning_generation())
        if self.train_sweep:
            rows = max(int(lib.odpd_sweep_partial_rows(C.byref(self.desc), b, self.T, self.flags)) for b in {self.B, last})
            ws = max(int(lib.odpd_sweep_workspace_floats(C.byref(self.desc), b, self.T, self.flags)) for b in {self.B, last})
            P = bb.n_flat
            for r in runs:
                r.opt._ensure(self.dev)
                r.partials = torch.empty(rows, P + _lib.LOSS_COLS, dtype=torch.float32, device=self.dev)
                r.workspace = torch.empty(ws, dtype=torch.float32, device=self.dev) if ws > 0 else None
            self.scratch = torch.empty(int(lib.odpd_sweep_scratch_bytes(len(runs), self.n_steps)), dtype=torch.uint8, device=self.dev)

    # ---- one training epoch of every run of the group ----
    def train_epoch(self):
        from .train_funcs import _check_exchange_health, net_train
        if not self.train_sweep:
            for r in self.runs:
                with r.rng:
                    r.net = net_train(r.proj.log_train, r.net, r.loaders[0], r.opt, r.criterion, r.proj.grad_clip_val, r.proj.device)
            return
        lib = _lib.load()
        if int(lib.odpd_tuning_generation()) != self._tuning_gen:
            self._size_buffers()
        K = len(self.runs)
        table = (_lib.SweepRun * K)()
        train0 = self.runs[0].loaders[0]
        for k, r in enumerate(self.runs):
            r.net.train()
            with r.rng:
                r.order = r.loaders[0].epoch_order()           # the run's own shuffle, drawn from the run's own RNG stream
            r.losses = torch.empty(self.n_steps, dtype=torch.float32, device=self.dev)
            g = r.opt.param_groups[0]
            flat = r.net.backbone.flat_params(full_check=True)
            table[k] = _lib.SweepRun(flat.data_ptr(), r.opt.grad.data_ptr(), r.opt.exp_avg.data_ptr(), r.opt.exp_avg_sq.data_ptr(),
                                     r.partials.data_ptr(), r.losses.data_ptr(), None, r.workspace.data_ptr() if r.workspace is not None else None,
                                     r.order.data_ptr(), float(g["lr"]))
        g0 = self.runs[0].opt.param_groups[0]
        assert all(r.opt.step_count == self.runs[0].opt.step_count for r in self.runs)
        fr = _lib.Frames(train0.x.data_ptr(), train0.y.data_ptr(), None, self.n, self.T, train0.stride, _sample_format(train0.x, train0.y), 0)
        rc = lib.odpd_train_epoch_sweep(_lib.stream_ptr(), C.byref(self.desc), K, table, _lib.LOSS_IDS[self.kind], C.byref(fr), self.B,
                                        self.runs[0].opt.step_count + 1, float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"]),
                                        float(g0["weight_decay"]), float(self.runs[0].proj.grad_clip_val or 0.0), self.flags,
                                        C.c_void_p(self.scratch.data_ptr()))
        _lib.check(rc, "odpd_train_epoch_sweep")
        # the K epoch means: each the reduction its solo run makes (train_funcs.net_train), queued back to back and read with ONE synchronisation
        means = torch.stack([r.losses.double().mean() for r in self.runs]).cpu() if self.n_steps else None
        for k, r in enumerate(self.runs):
            r.opt.step_count += self.n_steps
            r.opt.last_epoch_losses = r.losses
            r.proj.log_train["loss"] = float(means[k].item()) if means is not None else float("nan")

    # ---- validation + test of every run of the group (train_funcs.net_eval_pair, the K forwards as one launch per batch pair) ----
    def eval_epoch(self):
        from .metrics import calculate_metrics
        from .train_funcs import net_eval, net_eval_pair
        lib = _lib.load()
        p0 = self.runs[0].proj
        batches, lists = None, None
        if p0.eval_val and p0.eval_test and self.desc is not None and len(self.runs) > 1:
            # (one pass over a torch DataLoader draws its base seed from the global RNG even without shuffling: every run makes the two passes
            # its solo epoch makes, under its own RNG copy — the NEXT epoch's shuffle depends on it)
            lists = []
            for r in self.runs:
                with r.rng:
                    lists.append((list(r.loaders[1]), list(r.loaders[2])))
            ba, bb_ = lists[0]
            if len(ba) == len(bb_) and all(fa.shape[1:] == fb.shape[1:] for (fa, _), (fb, _) in zip(ba, bb_)):
                xs = [torch.cat((fa, fb), dim=0).to(self.dev).float().contiguous() for (fa, _), (fb, _) in zip(ba, bb_)]
                if all(lib.odpd_sweep_fwd_supported(C.byref(self.desc), x.shape[0], x.shape[1]) for x in xs):
                    batches = (ba, bb_, xs)
        if batches is None:
            for r, pre in zip(self.runs, lists if lists is not None else [None] * len(self.runs)):
                p = r.proj
                if pre is not None:      # the passes over the loaders were made above (their RNG draws with them)
                    _, (pv, tv), (pt, tt) = net_eval_pair(p.log_val, p.log_test, r.net, pre[0], pre[1], r.criterion, p.device)
                    p.log_val = calculate_metrics(p.args, p.log_val, pv, tv)
                    p.log_test = calculate_metrics(p.args, p.log_test, pt, tt)
                    continue
                with r.rng:
                    self._eval_solo(r, calculate_metrics, net_eval, net_eval_pair)
            return
        ba, bb_, xs = batches
        self._eval_sweep(ba, bb_, xs, calculate_metrics)

    @staticmethod
    def _eval_solo(r, calculate_metrics, net_eval, net_eval_pair):
                p = r.proj
                if p.eval_val and p.eval_test:
                    _, (pv, tv), (pt, tt) = net_eval_pair(p.log_val, p.log_test, r.net, r.loaders[1], r.loaders[2], r.criterion, p.device)
                    p.log_val = calculate_metrics(p.args, p.log_val, pv, tv)
                    p.log_test = calculate_metrics(p.args, p.log_test, pt, tt)
                elif p.eval_val:
                    _, pred, truth = net_eval(p.log_val, r.net, r.loaders[1], r.criterion, p.device)
                    p.log_val = calculate_metrics(p.args, p.log_val, pred, truth)
                elif p.eval_test:
                    _, pred, truth = net_eval(p.log_test, r.net, r.loaders[2], r.criterion, p.device)
                    p.log_test = calculate_metrics(p.args, p.log_test, pred, truth)

    def _eval_sweep(self, ba, bb_, xs, calculate_metrics):
        lib = _lib.load()
        K = len(self.runs)
        scratch = torch.empty(int(lib.odpd_sweep_scratch_bytes(K, 0)), dtype=torch.uint8, device=self.dev)
        outs = [([], [], []) for _ in range(K)], [([], [], []) for _ in range(K)]
        with torch.no_grad():
            for (fa, ta), (fb, tb), x in zip(ba, bb_, xs):
                ta, tb = ta.to(self.dev), tb.to(self.dev)
                ys = [torch.empty(x.shape[0], x.shape[1], 2, dtype=torch.float32, device=self.dev) for _ in range(K)]
                table = (_lib.SweepRun * K)()
                for k, r in enumerate(self.runs):
                    r.net.eval()
                    table[k] = _lib.SweepRun(r.net.backbone.flat_params().data_ptr(), None, None, None, None, None, ys[k].data_ptr(), None, None, 0.0)
                _lib.check(lib.odpd_backbone_fwd_sweep(_lib.stream_ptr(), C.byref(self.desc), K, table, x.shape[0], x.shape[1], _lib.ptr(x),
                                                       C.c_void_p(scratch.data_ptr())), "odpd_backbone_fwd_sweep")
                for k, r in enumerate(self.runs):
                    for (losses, pred, truth), o, t in ((outs[0][k], ys[k][:fa.shape[0]], ta), (outs[1][k], ys[k][fa.shape[0]:], tb)):
                        losses.append(r.criterion(o, t)); pred.append(o); truth.append(t)
        # host side of the K x 2 evaluations at sweep level: one device-to-host copy per split for all runs' predictions, one for every loss,
        # the split's ground truth once (the runs of a group share their dataset), metrics over the (K, segments, nperseg) stack with one
        # FFT / Welch call per split (metrics.calculate_metrics_many: each run's numbers are those of its own calculate_metrics call)
        from .metrics import calculate_metrics_many
        for split, name in ((0, "log_val"), (1, "log_test")):
            per_run = outs[split]
            if not per_run[0][0]:
                for r in self.runs:
                    getattr(r.proj, name)["loss"] = float("nan")
                continue
            loss_m = torch.stack([torch.stack([l.float() for l in per_run[k][0]]) for k in range(K)]).cpu().numpy()      # (K, batches)
            preds = torch.stack([torch.cat(per_run[k][1], dim=0) for k in range(K)]).cpu().numpy()                       # (K, segments, nperseg, 2)
            truth = torch.cat(per_run[0][2], dim=0).cpu().numpy()
            stats = []
            for k, r in enumerate(self.runs):
                getattr(r.proj, name)["loss"] = float(np.mean(loss_m[k]))
                stats.append(getattr(r.proj, name))
            done = calculate_metrics_many(self.runs[0].proj.args, stats, [preds[k] for k in range(K)], [truth] * K)
            for r, st in zip(self.runs, done):
                setattr(r.proj, name, st)


def train_pa_sweep(dataset_name=None, seeds=(0,), hidden_sizes=None, PA_backbone="gru", PA_hidden_size=23, n_epochs=100, batch_size=256,
                   lr=5e-4, accelerator="cuda", frame_length=200, exact=True, **kwargs):
    """`opendpd.api.train_pa` (opendpd/api.py:27-104) for every (hidden size, seed) pair, the runs of one model shape advancing in lockstep
    on one GPU.  Returns one result dictionary per run ({'status', 'model_path', 'log_path', 'seed', 'PA_hidden_size', 'lockstep',
    'mode'}), in hidden-size-major order.
    exact=True: every run on the kernels its solo `train_pa` takes — the files are those the solo calls would have written, bit for bit;
    at the reference's batch sizes those kernels keep a frame's whole BPTT state in LDS (one frame per CU), so the GPU part of K runs costs K
    solo epochs and the gain is the launch count.  exact=False (throughput mode, GRU family with hidden <= 16): the training steps run on
    the 16-sequences-per-wave MFMA kernel, K x batch / 16 waves at once — each run then equals its solo run with that kernel forced
    (odpd_set_tuning("s16_min_batch", 0)) bit for bit, and the default solo run to float tolerance (a different summation order)."""
    if not dataset_name:
        raise ValueError("train_pa_sweep requires dataset_name. Create a dataset first with create_dataset().")
    from . import data as D
    runs = []
    from . import project as PJ
    D._share = {}      # one parse of the dataset's CSV files for all runs ...
    PJ._stream_share = {}      # ... and one upload of its train streams (read-only on the device)
    try:
        for H in (list(hidden_sizes) if hidden_sizes else [PA_hidden_size]):
            for seed in seeds:
                run = _Run(None)
                with run.rng:      # Project.__init__ seeds the global RNGs (project.py:108-112): from here on this run owns its own copy of them
                    run.proj = Project(step="train_pa", dataset_name=dataset_name, PA_backbone=PA_backbone, PA_hidden_size=int(H), n_epochs=n_epochs,
                                       batch_size=batch_size, lr=lr, accelerator=accelerator, frame_length=frame_length, seed=int(seed), **kwargs)
                    _setup(run)
                runs.append(run)
    finally:
        D._share = None
        PJ._stream_share = None
    groups = {}
    for r in runs:
        groups.setdefault(_group_key(r), []).append(r)
    groups = [_Group(v, exact=exact) for v in groups.values()]
    # sweep-level bookkeeping: history rows are appended (project.CsvLogger), an improved model is remembered on the device
    # and its checkpoint written once, when the sweep ends or is interrupted (`flush`), instead of one torch.save per run and improving epoch
    for r in runs:
        r.proj.logger.defer_checkpoints = True
    start = time.time()
    try:
        for epoch in range(n_epochs):
            for g in groups:
                g.train_epoch()
            for g in groups:
                g.eval_epoch()
            for r in runs:      # (log row, best-model bookkeeping, plateau scheduler: nothing here draws from an RNG — no state swap)
                r.proj.finish_epoch(r.net, r.opt, r.sched, epoch, start, "NMSE")
    finally:
        for r in runs:
            r.proj.logger.flush()
    return [{"status": "completed", "model_path": r.proj.path_save_file_best, "log_path": r.proj.path_log_file_best, "seed": r.proj.seed,
             "PA_hidden_size": r.proj.PA_hidden_size, "lockstep": bool(g_.train_sweep),
             "mode": ("s16" if g_.flags else "exact") if g_.train_sweep else "per-run"}
            for g_ in groups for r in g_.runs]
