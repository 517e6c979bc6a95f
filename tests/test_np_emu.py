"""Narrow-operand (`--amp`) GEMM kernels of csrc/conv_np.hip against their CPU definition (oracle/np_oracle.py) under the SIMT
emulator: forward, data gradient (stride 1 and the stride-2 parity classes) and weight gradient for both operand modes, every
tile of the narrow launchers, split-K, ragged geometries, per-sample weights (checks in np_checks.py)."""
import pytest
import torch

import np_checks as nc

DEV = torch.device("cpu")


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("geom", nc.GEOMS)
@pytest.mark.parametrize("tile,split", nc.FWD_TILES)
def test_forward(emu_lib, mode, geom, tile, split):
    nc.check_forward(DEV, mode, geom, tile, split)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("geom", nc.GEOMS)
@pytest.mark.parametrize("tile,split", nc.WGRAD_TILES)
def test_wgrad(emu_lib, mode, geom, tile, split):
    nc.check_wgrad(DEV, mode, geom, tile, split)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("geom", nc.AUTOGRAD_GEOMS)
def test_autograd(emu_lib, mode, geom):
    nc.check_autograd(DEV, mode, geom)


@pytest.mark.parametrize("mode", [1, 2])
def test_batch_conv(emu_lib, mode):
    nc.check_batch_conv(DEV, mode)


def test_modes_differ_from_fp32(emu_lib):
    nc.check_modes_differ(DEV)


def test_scalar_gather_layers_stay_fp32(emu_lib):
    nc.check_scalar_gather_stays_fp32(DEV)


def test_mode_is_restored(emu_lib):
    _, conv = nc.oc.pkg()
    assert conv.mfma_mode() == 0
