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


@pytest.mark.parametrize('variant,base', tc.EXPERIMENTAL_FWD_TILES)
def test_prefetch_two_variant(variant, base):
    """force_tile-only prefetch-distance-2 kernels: against F.conv2d and bit-equal to the plan's tile of the same shape, over
    chunk counts 1 ... 11 per K split (odd and even: the loop runs two chunks per trip)"""
    tc.check_forward_tiles(CPU, tiles=(variant,), geoms=tc.GEOMS)
    tc.check_variant_equals_plan_tile(CPU, variant, base)


@pytest.mark.parametrize('tile', tc.REORDERED_FWD_TILES)
def test_lds_direct_variant(tile):
    """loads straight into LDS, three buffers, whole trips of three chunks (chunk counts 1 ... 11 per split: every remainder)"""
    tc.check_forward_tiles(CPU, tiles=(tile,), geoms=tc.GEOMS)
