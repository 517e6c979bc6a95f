"""Seeded random-geometry sweep of the autograd operators under the SIMT emulator.  The fixed cases in op_checks.run_all pin
the shapes the networks use; this sweep walks ragged / odd geometries through the same checks so that the tile plans, the
stride-2 parity-class data gradient (with and without the zero-fill skip), the split-K paths and the grouped reductions are all
hit off the beaten track."""
import random

import pytest
import torch

import op_checks as oc

DEV = torch.device("cpu")


def _conv_cases(nr, seed):
    rng = random.Random(seed)
    cases = []
    for i in range(nr):
        k, s = rng.choice([(1, 1), (3, 1), (3, 2), (4, 2), (4, 1), (2, 2), (2, 1)])       # <= 16 taps: the generic conv limit (FlowNet2's 5x5 / 7x7 go through flownet2.FConv)
        p = rng.choice([0, k // 2]) if k > 1 else 0
        h, w = rng.randint(max(k, 4), 13), rng.randint(max(k, 4), 13)
        cin = rng.choice([3, 4, 6, 8, 20, 33, 48])
        cout = rng.choice([1, 2, 5, 16, 40, 70, 136])
        n = rng.choice([1, 2, 3])
        act = rng.choice(['none', 'lrelu', 'tanh', 'sigmoid'])
        cases.append((n, cin, h, w, cout, k, s, p, act, rng.random() < 0.7, 100 + i))
    return cases


@pytest.mark.parametrize("case", _conv_cases(80, 7))
def test_conv_sweep(emu_lib, case):
    n, cin, h, w, cout, k, s, p, act, bias, seed = case
    oc.check_conv(DEV, n, cin, h, w, cout, k, s, p, act=act, bias=bias, seed=seed, tol=2e-4)


@pytest.mark.parametrize("case", _conv_cases(40, 11))
def test_conv_sweep_cached_deferred(emu_lib, case):
    """same sweep with the persistent layouts and the deferred weight / bias gradient finalisation switched on"""
    ops, conv = oc.pkg()
    import importlib
    lc = importlib.import_module(ops.__name__.rsplit('.', 1)[0] + '.layout_cache')
    gf = importlib.import_module(ops.__name__.rsplit('.', 1)[0] + '.grad_finalize')
    n, cin, h, w, cout, k, s, p, act, bias, seed = case
    oc.check_conv(DEV, n, cin, h, w, cout, k, s, p, act=act, bias=bias, seed=seed, tol=2e-4,
                  cache=lc.LayoutCache(), fin=gf.GradFinalizer())


@pytest.mark.parametrize("seed", range(16))
def test_batch_conv_sweep(emu_lib, seed):
    rng = random.Random(300 + seed)
    oc.check_batch_conv(DEV, b=rng.randint(1, 3), cin=rng.choice([4, 8, 12, 36]), cout=rng.choice([3, 12, 70]),
                        h=rng.randint(3, 9), w=rng.randint(3, 9), seed=300 + seed)


@pytest.mark.parametrize("seed", range(16))
def test_norm_sweep(emu_lib, seed):
    rng = random.Random(400 + seed)
    oc.check_norm(DEV, instance=rng.random() < 0.5, n=rng.randint(1, 4), c=rng.choice([1, 3, 10, 65, 130]),
                  h=rng.randint(2, 11), w=rng.randint(2, 11), affine=rng.random() < 0.6,
                  act=rng.choice(['none', 'lrelu']), seed=400 + seed)


@pytest.mark.parametrize("seed", range(16))
def test_spade_sweep(emu_lib, seed):
    rng = random.Random(500 + seed)
    oc.check_spade(DEV, nmaps=rng.randint(1, 3), generated=rng.random() < 0.6, n=rng.randint(1, 3),
                   c=rng.choice([4, 12, 40, 68]), ch=rng.choice([4, 8, 12]), h=rng.randint(3, 8), w=rng.randint(3, 8),
                   act=rng.choice(['none', 'lrelu']), seed=500 + seed)


@pytest.mark.parametrize("seed", range(4))
def test_warp_softmax_upsample_sweep(emu_lib, seed):
    rng = random.Random(600 + seed)
    oc.check_warp(DEV, b=rng.randint(1, 3), c=rng.choice([1, 3, 5]), h=rng.randint(5, 20), w=rng.randint(5, 20),
                  mag=rng.choice([0.5, 3.0, 30.0]), seed=600 + seed)
    oc.check_softmax(DEV, n=rng.randint(1, 3), c=rng.choice([2, 17, 70, 300]), h=rng.randint(1, 5), w=rng.randint(1, 5),
                     seed=600 + seed)
    oc.check_upsample(DEV, n=rng.randint(1, 3), c=rng.choice([1, 6, 33]), h=rng.randint(1, 6), w=rng.randint(1, 6),
                      seed=600 + seed)
