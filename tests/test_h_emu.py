"""Half-precision convolution kernels (csrc/conv_h.hip) under the SIMT emulator: every tile / buffer count / split, data and
weight gradients, grouped launches, epilogue statistics, element conversion."""
import pytest
import torch

import h_checks as hc

DEV = torch.device("cpu")


def test_cast(emu_lib):
    hc.check_cast(DEV)


@pytest.mark.parametrize('gi', range(len(hc.GEOMS)))
@pytest.mark.parametrize('out_half', [True, False])
def test_forward_plan(emu_lib, gi, out_half):
    hc.check_forward(DEV, hc.GEOMS[gi], -1, 0, out_half)


@pytest.mark.parametrize('tile,split', hc.FWD_TILES)
def test_forward_tiles(emu_lib, tile, split):
    hc.check_forward(DEV, hc.GEOMS[1], tile, split, True)
    hc.check_forward(DEV, hc.GEOMS[4], tile, split, False, res_half=False)


def test_forward_half_residual_no_activation(emu_lib):
    hc.check_forward(DEV, hc.GEOMS[0], -1, 0, True, res_half=True, act=False)


@pytest.mark.parametrize('gi', range(len(hc.GEOMS)))
def test_dgrad(emu_lib, gi):
    hc.check_dgrad(DEV, hc.GEOMS[gi], True)
    hc.check_dgrad(DEV, hc.GEOMS[gi], False)


@pytest.mark.parametrize('tile,split', hc.WGRAD_TILES)
def test_wgrad_tiles(emu_lib, tile, split):
    assert hc.check_wgrad(DEV, hc.WG_GEOMS[0], tile, split)
    assert hc.check_wgrad(DEV, hc.WG_GEOMS[1], tile, split)


def test_wgrad_geometries(emu_lib):
    ran = [hc.check_wgrad(DEV, g, 0, 0) for g in hc.GEOMS + hc.WG_GEOMS]
    assert sum(ran) >= 5, ran


def test_group(emu_lib):
    hc.check_group(DEV)


def test_stats(emu_lib):
    hc.check_stats(DEV)


@pytest.mark.parametrize('nmaps,up', [(1, False), (3, True)])
def test_spade_half_output_and_half_gradient(emu_lib, nmaps, up):
    import op_checks as oc
    oc.check_spade(DEV, nmaps=nmaps, generated=True, c=32, ch=16, h=8, w=10, up=up, half_out=True)
