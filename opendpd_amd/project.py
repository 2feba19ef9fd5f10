"""Run orchestration: a compact restatement of the reference's arguments.py / project.py / steps/*.py / modules/paths.py /
modules/loggers.py for the three steps train_pa, train_dpd, run_dpd — same hyper-parameter names and defaults, same
`save/ log/ dpd_out/` layout and model-ID strings (project.py:57-92, paths.py:75-117), same CSV log columns
(paths.py:9-72), same best-model rule (loggers.py:165-179: epoch 0 always, then strict <).

What differs is where the work runs: the frames never leave the GPU (`DeviceFrameLoader` gathers each batch from the
resident I/Q stream with the index order a reference DataLoader(shuffle=True) would draw), and the step is the fused
HIP train step."""
import argparse
import os
import warnings
import random
import time

import numpy as np
import pandas as pd
import torch
import torch.nn as nn
from torch.utils.data import DataLoader

from . import data as D
from .metrics import calculate_metrics
from .models import CascadedModel, CoreModel
from .quant import get_quant_model
from .train_funcs import FusedAdam, FusedAdamW, FusedRMSprop, FusedSGD, net_eval, net_eval_pair, net_train

DEFAULTS = dict(  # arguments.py:8-89
    dataset_name=None, dataset_path=None, filename="", log_precision=8, step="run_dpd", eval_val=1, eval_test=1,
    accelerator="cuda", devices=0, re_level="soft", use_segments=False, frame_length=200, frame_stride=1, seed=0,
    loss_type="l2", opt_type="adamw", batch_size=256, batch_size_eval=256, n_epochs=100, lr_schedule=0, lr=5e-4,
    lr_end=1e-4, decay_factor=0.1, patience=10, grad_clip_val=200, K=4, PA_backbone="gru", PA_hidden_size=23,
    PA_num_layers=1, DPD_backbone="gru", DPD_hidden_size=15, DPD_num_layers=1, quant=False, n_bits_w=8, n_bits_a=8,
    pretrained_model="", quant_dir_label="", q_pretrain=False, thx=0.0, thh=0.0, num_dvr_units=3, window_size=4)
# arguments the reference does not have (kept apart: DEFAULTS mirrors arguments.py exactly, tests/test_api_cpu.py).
#   frame_storage: storage of the resident TRAINING streams on the device — "fp32" (the reference's data) or "bf16" (BASELINE
#   configs[1]: 4 bytes per I/Q sample, widened exactly, fp32 arithmetic; read in place by the GRU-family train kernels)
EXTENSIONS = dict(frame_storage="fp32")


def count_net_params(net):
    return sum(p.numel() for p in net.parameters())


_stream_share = None      # sweeps: {(dataset, step, device, storage): (x, y) device streams} while the runs of a sweep are set up


class DeviceFrameLoader:
    """Iterates (features, targets) batches of frames gathered ON DEVICE from the resident I/Q streams (no
    materialised frame tensor: a frame is a start offset into the stream, data_collector.py:239-247).
    The epoch's frame order is what ONE pass of a torch DataLoader over range(n) with shuffle=True yields, with exactly
    the global-RNG consumption (base seed + sampler seed) and permutation of the reference's
    DataLoader(train_set, shuffle=True) (project.py:236) — tests/test_api_cpu.py compares it with a real DataLoader; each step is then two index_select launches on an
    overlapping-window view of the stream and no host<->device traffic."""

    def __init__(self, x, y, frame_length, stride, batch_size, device, shuffle=True, storage="fp32", share_key=None):
        if storage not in ("fp32", "bf16"):
            raise ValueError(f"frame_storage={storage!r}: expected 'fp32' or 'bf16'")
        # storage="bf16": the resident streams hold bf16 (I, Q) pairs (round-to-nearest-even of the data, 4 bytes per sample); the
        # fused kernels read them in place (odpd_frames_t.sample_format) and batches gathered for the generic path are widened to fp32
        dt = torch.bfloat16 if storage == "bf16" else torch.float32
        # the K runs of a sweep train on the SAME resident streams (read-only): one upload, not K (opendpd_amd/sweep.py sets _stream_share)
        key = None if (_stream_share is None or share_key is None) else (share_key, str(device), storage)
        if key is not None and key in _stream_share:
            self.x, self.y = _stream_share[key]
        else:
            self.x = torch.as_tensor(np.asarray(x), dtype=torch.float32).to(device).to(dt).contiguous()
            self.y = torch.as_tensor(np.asarray(y), dtype=torch.float32).to(device).to(dt).contiguous()
            if key is not None:
                _stream_share[key] = (self.x, self.y)
        self.n = (len(x) - frame_length) // stride + 1
        self.batch_size, self.device, self.frame_length, self.stride = batch_size, device, frame_length, stride
        win = lambda s: torch.as_strided(s, (self.n, frame_length, 2), (2 * stride, 2, 1))
        self.fx, self.fy = win(self.x), win(self.y)
        self.shuffle, self.storage = shuffle, storage

    def __len__(self):
        return (self.n + self.batch_size - 1) // self.batch_size

    def epoch_order(self):
        """Frame indices of one epoch in visiting order (device int64); one call == one pass of the reference DataLoader."""
        # What one pass of DataLoader(range(n), shuffle=...) does to the global RNG and yields, without iterating n Python ints:
        # the iterator draws its base seed (dataloader.py _BaseDataLoaderIter.__init__), RandomSampler then draws the seed of a
        # private generator and yields torch.randperm(n, generator) (sampler.py RandomSampler.__iter__, num_samples == n)
        torch.empty((), dtype=torch.int64).random_()
        if not self.shuffle:
            return torch.arange(self.n, device=self.device)
        g = torch.Generator()
        g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
        return torch.randperm(self.n, generator=g).to(self.device)

    def __iter__(self):
        order = self.epoch_order()
        for i in range(0, self.n, self.batch_size):
            idx = order[i:i + self.batch_size]
            yield self.fx.index_select(0, idx).float(), self.fy.index_select(0, idx).float()


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', threshold=1e-4, threshold_mode='rel', cooldown=0, eps=1e-8) as
    configured in project.py:289-296, for optimisers that only expose `param_groups` (FusedAdamW).  Torch's rule is kept to the
    letter: "better" means metric < best * (1 - threshold) whatever the sign of best — the monitored values are NMSE / ACLR in dB,
    i.e. negative, where this lets a metric up to 0.01 % WORSE than the best still count as an improvement."""

    def __init__(self, optimizer, factor, patience, min_lr, threshold=1e-4, eps=1e-8):
        self.opt, self.factor, self.patience, self.min_lr, self.threshold, self.eps = optimizer, factor, patience, min_lr, threshold, eps
        self.best, self.bad = float("inf"), 0

    def step(self, metric):
        metric = float(metric)
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            for g in self.opt.param_groups:
                new_lr = max(float(g["lr"]) * self.factor, self.min_lr)
                if float(g["lr"]) - new_lr > self.eps:
                    g["lr"] = new_lr
            self.bad = 0


class CsvLogger:
    """History / best CSV files with the reference's formatting (loggers.py:119-163).

    The reference REWRITES the whole history file through pandas after every epoch (loggers.py:131-140).  The bytes of that file are header +
    one line per epoch and earlier lines never change, so the same file is produced by appending the new line — `csv.writer` with pandas'
    own dialect (minimal quoting, "\n") on values that are already strings or integers (tests/test_api_cpu.py compares the two writers byte
    for byte); anything else in a row (a float32 scalar, None, a changed column set) falls back to the pandas rewrite.
    `defer_checkpoints` (sweeps, opendpd_amd/sweep.py): an improved model is remembered as a device-side copy of its state dict and written
    once, by `flush()`, instead of one `torch.save` per improving epoch — the file is the one the last improving epoch would have written."""

    def __init__(self, path_save_best, path_log_best, path_log_hist, precision=8, writes=True):
        self.path_save_best, self.path_log_best, self.path_log_hist, self.precision = path_save_best, path_log_best, path_log_hist, precision
        self.headers, self.rows, self.best_val_metric = [], [], None
        self.writes = writes          # data-parallel runs: every rank keeps the bookkeeping, rank 0 alone touches the files
        self.defer_checkpoints = False
        self._pending = None          # (net, {key: device copy}) of the best model not yet on disk
        self._hist_ok = False         # the history file on disk = header + self.rows[:-1], written by the fast writer

    @staticmethod
    def _plain(row):
        return all(type(v) in (str, int) or (isinstance(v, np.integer) and not isinstance(v, np.bool_)) for v in row)

    @staticmethod
    def _write_rows(path, mode, lines):
        import csv
        with open(path, mode, newline="") as f:
            csv.writer(f, lineterminator="\n", quoting=csv.QUOTE_MINIMAL).writerows(lines)

    def write_log(self, stat):
        headers = list(stat.keys())
        fmt = "{:." + str(self.precision) + "f}"
        row = [fmt.format(v) if isinstance(v, float) else v for v in stat.values()]
        same_cols = headers == self.headers
        self.headers = headers
        self.rows.append(row)
        if not self.writes:
            return
        if self._plain(row) and self._plain(headers) and all(isinstance(h, str) for h in headers):
            if self._hist_ok and same_cols and len(self.rows) > 1:
                self._write_rows(self.path_log_hist, "a", [row])
                return
            if len(self.rows) == 1 or all(self._plain(r) and len(r) == len(headers) for r in self.rows):
                self._write_rows(self.path_log_hist, "w", [headers] + self.rows)
                self._hist_ok = True
                return
        self._hist_ok = False
        pd.DataFrame(self.rows, columns=self.headers).to_csv(self.path_log_hist, index=False)

    def _write_best(self, idx):
        if self.writes:
            if self._plain(self.rows[idx]) and all(isinstance(h, str) for h in self.headers):
                self._write_rows(self.path_log_best, "w", [self.headers, self.rows[idx]])
            else:
                pd.DataFrame([self.rows[idx]], columns=self.headers).to_csv(self.path_log_best, index=False)

    def save_best_model(self, net, epoch, val_stat, metric_name):
        crit = val_stat[metric_name]
        if epoch == 0 or crit < self.best_val_metric:
            self.best_val_metric = crit
            if self.writes:
                if self.defer_checkpoints:
                    self._pending = (net, {k: v.detach().clone() for k, v in net.state_dict().items()})
                else:
                    torch.save(net.state_dict(), self.path_save_best)
            self._write_best(epoch)

    def flush(self):
        """write the remembered best model: its values are copied into the live module for the duration of the `torch.save` call, so that
        the file has the layout a save at that epoch had (every parameter a view into the one flat storage), then the live values return"""
        if self._pending is None:
            return
        net, best = self._pending
        self._pending = None
        with torch.no_grad():
            sd = net.state_dict()
            live = {k: v.detach().clone() for k, v in sd.items()}
            for k, v in sd.items():
                v.copy_(best[k])
            torch.save(net.state_dict(), self.path_save_best)
            for k, v in net.state_dict().items():
                v.copy_(live[k])


class Project:
    def __init__(self, **overrides):
        hp = dict(DEFAULTS, **EXTENSIONS)
        unknown = set(overrides) - set(hp)
        if unknown:
            raise TypeError(f"unknown arguments: {sorted(unknown)}")
        hp.update(overrides)
        self.args = argparse.Namespace(**hp)
        self.hparams = vars(self.args)
        for k, v in hp.items():
            setattr(self, k, v)
        for k, v in D.load_spec(self.dataset_name, self.dataset_path).items():     # project.py:163-166
            setattr(self, k, v)
            setattr(self.args, k, v)
        self.log_train, self.log_val, self.log_test = {}, {}, {}
        random.seed(self.seed); np.random.seed(self.seed); torch.manual_seed(self.seed)   # project.py:108-112
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(self.seed)
        ds = self.dataset_name or os.path.splitext(os.path.basename(str(self.dataset_path)))[0]
        self.dataset_label = ds
        if self.step == "train_pa":
            base = (ds, self.step, self.quant_dir_label)
        else:
            base = (ds, self.step, self.pa_dir_id(), self.quant_dir_label)
        self.path_dir_save = os.path.join("./save", *base)
        self.path_dir_log_hist = os.path.join("./log", *base, "history")
        self.path_dir_log_best = os.path.join("./log", *base, "best")
        for d in (self.path_dir_save, self.path_dir_log_hist, self.path_dir_log_best):
            os.makedirs(d, exist_ok=True)

    # ---- ids (project.py:57-92, paths.py:105-117) ----------------------------------------------------------------
    def pa_dir_id(self):
        return f"PA_S_{self.seed}_M_{self.PA_backbone.upper()}_H_{self.PA_hidden_size:d}_F_{self.frame_length:d}"

    def gen_pa_model_id(self, n):
        return f"{self.pa_dir_id()}_P_{n:d}"

    def gen_dpd_model_id(self, n):
        s = f"DPD_S_{self.seed}_M_{self.DPD_backbone.upper()}_H_{self.DPD_hidden_size:d}_F_{self.frame_length:d}_P_{n:d}"
        if "delta" in self.DPD_backbone:
            s += f"_THX_{self.thx:.3f}_THH_{self.thh:.3f}"
        return s

    def set_device(self):
        """One process per GPU: under torchrun (WORLD_SIZE > 1) the process takes device LOCAL_RANK and joins the default
        process group (backend nccl = RCCL; OPENDPD_DIST_BACKEND overrides, OPENDPD_DIST_SINGLE_DEVICE=1 keeps `devices` for
        functional checks of the N > 1 path on a one-GPU box); a single process keeps the reference's `--devices` index."""
        from . import dist as DP
        self.rank, local, self.world = DP.env_world()
        if self.accelerator == "cpu" and torch.cuda.is_available():
            # the reference's default (arguments.py:20, opendpd/api.py:35) names the CPU; there is no CPU path here, and a call that
            # relies on the mirrored defaults (od.train_pa(dataset_name=...), OpenDPDTrainer's implicit train_pa) must still run
            warnings.warn("opendpd_amd has no CPU path: accelerator='cpu' (the reference's default) runs on the HIP device "
                          f"cuda:{self.devices}; pass accelerator='cuda' to silence this", stacklevel=2)
            self.accelerator = "cuda"
        if self.accelerator == "cuda" and torch.cuda.is_available():
            index = local if (self.world > 1 and os.environ.get("OPENDPD_DIST_SINGLE_DEVICE") != "1") else self.devices
            dev = torch.device("cuda:" + str(index))
            torch.cuda.set_device(dev)
            if self.world > 1:
                DP.init(os.environ.get("OPENDPD_DIST_BACKEND") or None, dev)
        elif self.accelerator in ("cpu", "cuda"):
            raise ValueError("opendpd_amd runs on a HIP device only and none is visible (torch.cuda.is_available() is False): "
                             "there is no CPU fallback")
        else:
            raise ValueError(f"The select device {self.accelerator} is not supported.")
        self.device = dev
        return dev

    def build_dataloaders(self):
        Xtr, ytr, Xv, yv, Xte, yte = D.load_dataset(self.dataset_name, self.dataset_path)
        self.target_gain = D.set_target_gain(Xtr, ytr)
        if self.step == "train_dpd":
            ytr, yv, yte = self.target_gain * Xtr, self.target_gain * Xv, self.target_gain * Xte
        train = DeviceFrameLoader(Xtr, ytr, self.frame_length, self.frame_stride, self.batch_size, self.device, shuffle=True,
                                  storage=self.frame_storage, share_key=(str(D.resolve_dataset(self.dataset_name, self.dataset_path)), self.step))
        val = DataLoader(D.IQSegmentDataset(Xv, yv, nperseg=self.args.nperseg), batch_size=self.batch_size_eval, shuffle=False)
        test = DataLoader(D.IQSegmentDataset(Xte, yte, nperseg=self.args.nperseg), batch_size=self.batch_size_eval, shuffle=False)
        return (train, val, test), Xtr.shape[-1]

    def build_logger(self, model_id):
        self.path_save_file_best = os.path.join(self.path_dir_save, model_id + ".pt")
        self.path_log_file_hist = os.path.join(self.path_dir_log_hist, model_id + ".csv")
        self.path_log_file_best = os.path.join(self.path_dir_log_best, model_id + ".csv")
        self.logger = CsvLogger(self.path_save_file_best, self.path_log_file_best, self.path_log_file_hist, self.log_precision,
                                writes=getattr(self, "rank", 0) == 0)

    def build_criterion(self):
        return {"l2": nn.MSELoss(), "l1": nn.L1Loss()}[self.loss_type]

    def build_optimizer(self, net):
        # project.py:274-297.  A kernel-backed model steps through the fused HIP optimiser of its kind; a registry configuration beyond the
        # kernels' envelope (backbones/extras.py, wide.py: ATen path) gets the torch optimiser the reference builds
        fused = {"adamw": FusedAdamW, "adam": FusedAdam, "sgd": FusedSGD, "rmsprop": FusedRMSprop}
        if self.opt_type == "adabound":
            import adabound  # noqa: F401  — as the reference (project.py:284-286): a package it does not ship (ModuleNotFoundError there too)
            raise NotImplementedError("--opt_type adabound: the reference builds adabound.AdaBound (project.py:284-286); there is no fused HIP step of that kind")
        if self.opt_type not in fused:
            raise RuntimeError("Please use a valid optimizer.")
        try:
            opt = fused[self.opt_type](net, lr=self.lr)
        except TypeError:
            params = [p for p in net.parameters() if p.requires_grad] if self.opt_type == "adamw" else net.parameters()
            opt = {"adamw": lambda: torch.optim.AdamW(params, lr=self.lr), "adam": lambda: torch.optim.Adam(params, lr=self.lr),
                   "sgd": lambda: torch.optim.SGD(params, lr=self.lr, momentum=0.9),
                   "rmsprop": lambda: torch.optim.RMSprop(params, lr=self.lr)}[self.opt_type]()
        if getattr(self, "world", 1) > 1 and not isinstance(opt, FusedAdamW):
            raise RuntimeError("data-parallel training (WORLD_SIZE > 1) runs through the fused HIP optimiser: --opt_type adamw on "
                               "a kernel-backed model")
        return opt, ReduceLROnPlateau(opt, self.decay_factor, self.patience, self.lr_end)

    def gen_log_stat(self, elapsed, net, optimizer, epoch):
        backbone, hidden = (self.PA_backbone, self.PA_hidden_size) if self.step == "train_pa" else (self.DPD_backbone, self.DPD_hidden_size)
        stat = {"EPOCH": epoch, "N_EPOCH": self.n_epochs, "TIME:": elapsed, "LR": optimizer.param_groups[-1]["lr"],
                "BATCH_SIZE": self.batch_size, "N_PARAM": count_net_params(net), "FRAME_LENGTH": self.frame_length,
                "BACKBONE": backbone, "HIDDEN_SIZE": hidden}
        if self.step == "train_dpd" and "delta" in net.dpd_model.backbone_type:
            bb = net.dpd_model.backbone
            stat["THX"], stat["THH"] = bb.thx, bb.thh
            stat.update(bb.get_temporal_sparsity())
            bb.set_debug(1)
        for prefix, d in (("TRAIN", self.log_train), ("VAL", self.log_val), ("TEST", self.log_test)):
            stat.update({f"{prefix}_{k.upper()}": (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in d.items()})
        return stat

    def train(self, net, criterion, optimizer, lr_scheduler, loaders, best_model_metric):
        train_loader, val_loader, test_loader = loaders
        start = time.time()
        for epoch in range(self.n_epochs):
            net = net_train(self.log_train, net, train_loader, optimizer, criterion, self.grad_clip_val, self.device)
            if self.eval_val and self.eval_test:      # both splits in one forward launch (latency-bound tiny batches)
                _, (pv, tv), (pt, tt) = net_eval_pair(self.log_val, self.log_test, net, val_loader, test_loader, criterion, self.device)
                self.log_val = calculate_metrics(self.args, self.log_val, pv, tv)
                self.log_test = calculate_metrics(self.args, self.log_test, pt, tt)
            elif self.eval_val:
                _, pred, truth = net_eval(self.log_val, net, val_loader, criterion, self.device)
                self.log_val = calculate_metrics(self.args, self.log_val, pred, truth)
            elif self.eval_test:
                _, pred, truth = net_eval(self.log_test, net, test_loader, criterion, self.device)
                self.log_test = calculate_metrics(self.args, self.log_test, pred, truth)
            self.finish_epoch(net, optimizer, lr_scheduler, epoch, start, best_model_metric)

    def finish_epoch(self, net, optimizer, lr_scheduler, epoch, start, best_model_metric):
        """the tail of one epoch of project.py's train loop: log row, best-model checkpoint, plateau scheduler (also opendpd_amd/sweep.py)"""
        self.log_all = self.gen_log_stat((time.time() - start) / 60.0, net, optimizer, epoch)
        self.logger.write_log(self.log_all)
        best_net = net.dpd_model if self.step == "train_dpd" else net
        self.logger.save_best_model(best_net, epoch, self.log_val, best_model_metric)
        if self.lr_schedule:
            lr_scheduler.step(self.log_val[best_model_metric])


def run_train_pa(proj):
    """steps/train_pa.py:10-59"""
    proj.set_device()
    loaders, input_size = proj.build_dataloaders()
    net = CoreModel(input_size, proj.PA_hidden_size, proj.PA_num_layers, proj.PA_backbone, window_size=proj.window_size,
                    num_dvr_units=proj.num_dvr_units).to(proj.device)
    proj.build_logger(proj.gen_pa_model_id(count_net_params(net)))
    opt, sched = proj.build_optimizer(net)
    proj.train(net, proj.build_criterion(), opt, sched, loaders, best_model_metric="NMSE")
    return net


def run_train_dpd(proj):
    """steps/train_dpd.py:14-90"""
    proj.set_device()
    loaders, input_size = proj.build_dataloaders()
    pa = CoreModel(input_size, proj.PA_hidden_size, proj.PA_num_layers, proj.PA_backbone, window_size=proj.window_size,
                   num_dvr_units=proj.num_dvr_units)
    pa_id = proj.gen_pa_model_id(count_net_params(pa))
    pa.load_state_dict(torch.load(os.path.join("save", proj.dataset_label, "train_pa", pa_id + ".pt"), map_location="cpu"))
    dpd = CoreModel(input_size, proj.DPD_hidden_size, proj.DPD_num_layers, proj.DPD_backbone, window_size=proj.window_size,
                    num_dvr_units=proj.num_dvr_units, thx=proj.thx, thh=proj.thh)
    dpd = get_quant_model(proj, dpd)
    proj.build_logger(proj.gen_dpd_model_id(count_net_params(dpd)))
    net = CascadedModel(dpd_model=dpd, pa_model=pa)
    net.freeze_pa_model()
    net = net.to(proj.device)
    opt, sched = proj.build_optimizer(net)
    proj.train(net, proj.build_criterion(), opt, sched, loaders, best_model_metric="ACLR_AVG")
    return net


def run_run_dpd(proj):
    """steps/run_dpd.py:19-94: the whole test split as ONE sequence through the trained DPD -> dpd_out/<id>.csv.
    (Like the reference, the DPD is rebuilt WITHOUT thx/thh here: delta models run dense at export time.)"""
    proj.set_device()
    X_test = D.load_dataset(proj.dataset_name, proj.dataset_path)[4]
    os.makedirs("dpd_out", exist_ok=True)
    pa = CoreModel(2, proj.PA_hidden_size, proj.PA_num_layers, proj.PA_backbone, num_dvr_units=proj.num_dvr_units)
    pa_id = proj.gen_pa_model_id(count_net_params(pa))
    dpd = get_quant_model(proj, CoreModel(2, proj.DPD_hidden_size, proj.DPD_num_layers, proj.DPD_backbone))
    dpd_id = proj.gen_dpd_model_id(count_net_params(dpd))
    sub = [proj.quant_dir_label] if proj.quant else []
    path = os.path.join("save", proj.dataset_label, "train_dpd", pa_id.split("_P_")[0], *sub, dpd_id + ".pt")
    dpd.load_state_dict(torch.load(path, map_location="cpu"))
    dpd = dpd.to(proj.device).eval()
    with torch.no_grad():
        out = dpd(torch.Tensor(X_test).unsqueeze(0).to(proj.device)).squeeze(0).cpu().numpy()
    out_dir = os.path.join("dpd_out", *sub)
    os.makedirs(out_dir, exist_ok=True)
    out_path = os.path.join(out_dir, dpd_id + ".csv")
    pd.DataFrame({"I": X_test[:, 0], "Q": X_test[:, 1], "I_dpd": out[:, 0], "Q_dpd": out[:, 1]}).to_csv(out_path, index=False)
    return out_path
