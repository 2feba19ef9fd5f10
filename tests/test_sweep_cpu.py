"""Host logic of the lockstep sweeps (opendpd_amd/sweep.py) that needs no GPU: the per-run RNG copies that make a run of a sweep draw exactly
what its solo run draws (reference: every run is a process of its own, bash_scripts/train_all_pa.sh:26-57; project.py:108-112 seeds the
process-global RNGs), and the argument contract of `train_pa_sweep`."""
import random

import numpy as np
import pytest
import torch

from opendpd_amd import data as D
from opendpd_amd.sweep import _Rng, train_pa_sweep


def _seed(s):
    torch.manual_seed(s); np.random.seed(s); random.seed(s)


def _draw():
    return float(torch.rand(())), float(np.random.rand()), random.random()


def test_interleaved_runs_draw_what_they_would_draw_alone():
    # two runs alone: three draws each from their own seeds
    alone = {}
    for s in (11, 12):
        _seed(s)
        alone[s] = [_draw() for _ in range(3)]
    # the same two runs interleaved inside one process, an outer stream running around them
    _seed(99)
    outer_ref = [_draw() for _ in range(4)]
    _seed(99)
    runs = {11: _Rng(), 12: _Rng()}
    got = {11: [], 12: []}
    outer = [_draw()]
    for s, r in runs.items():
        with r:            # (Project.__init__ seeds the global RNGs inside the run's first block)
            _seed(s)
            got[s].append(_draw())
    outer.append(_draw())
    for _ in range(2):
        for s, r in runs.items():
            with r:
                got[s].append(_draw())
        outer.append(_draw())
    assert got == alone
    assert outer == outer_ref       # the process-global stream is untouched by what the runs drew


def test_a_block_that_raises_still_restores_the_outer_stream():
    _seed(5)
    ref = [_draw(), _draw()]
    _seed(5)
    r = _Rng()
    first = _draw()
    with pytest.raises(RuntimeError):
        with r:
            _seed(1)
            _draw()
            raise RuntimeError("inside a run")
    assert [first, _draw()] == ref


def test_argument_contract():
    with pytest.raises(ValueError, match="dataset_name"):
        train_pa_sweep(dataset_name=None, seeds=(0, 1))
    # a failing setup must not leave the CSV cache of the sweep behind for later solo runs
    with pytest.raises(Exception):
        train_pa_sweep(dataset_name="no_such_dataset_anywhere", seeds=(0, 1), accelerator="cpu", n_epochs=1)
    assert D._share is None
