"""Every instantiated tile of the fp32 GEMM kernels on the SIMT emulator (tests/tile_checks.py)."""
import pytest
import torch

import tile_checks as tc

CPU = torch.device('cpu')


@pytest.mark.parametrize('tile', tc.FWD_TILES)
def test_forward_tile(tile):
    tc.check_forward_tiles(CPU, tiles=(tile,), geoms=tc.GEOMS)


def test_per_sample_tiles():
    tc.check_per_sample(CPU)


@pytest.mark.parametrize('tile', tc.WGRAD_TILES)
def test_wgrad_tile(tile):
    tc.check_wgrad_tiles(CPU, tiles=(tile,))
