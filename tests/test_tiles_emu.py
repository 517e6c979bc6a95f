"""Forced-tile sweep of the gather-GEMM and weight-gradient kernels under the SIMT emulator: every tile template and split
factor the launch plans can pick is checked on ragged geometries (pixel / channel counts that are not multiples of the tile,
stride 2, 4x4 and 1x1 kernels, scalar-gather input channels) against torch."""
import pytest
import torch
import torch.nn.functional as F

import op_checks as oc

DEV = torch.device("cpu")

GEOMS = [  # n, cin, h, w, cout, k, stride, pad
    (1, 8, 9, 7, 70, 3, 1, 1),
    (2, 12, 11, 13, 130, 3, 2, 1),
    (1, 20, 10, 9, 40, 4, 2, 2),
    (1, 36, 6, 5, 200, 1, 1, 0),
    (1, 6, 8, 8, 33, 3, 1, 1),          # Cin % 4 != 0: scalar gather path
]


@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("tile,split", [(0, 1), (0, 3), (1, 1), (1, 2), (2, 1), (4, 1), (4, 2), (9, 1), (9, 2),
                                        (10, 1), (10, 2), (11, 1), (11, 3), (12, 1), (12, 2), (13, 1), (13, 2), (14, 1), (14, 3), (15, 1), (15, 2),
                                        (16, 1), (16, 2), (16, 5), (17, 1), (17, 3), (18, 1), (18, 2), (19, 1), (19, 2), (20, 1), (20, 3), (21, 1), (21, 2)])
def test_forward_tiles(emu_lib, geom, tile, split):
    ops, conv = oc.pkg()
    n, cin, h, w, cout, k, s, p = geom
    g = torch.Generator().manual_seed(1000 + tile * 10 + split)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    ref = F.leaky_relu(F.conv2d(x, wt, b, stride=s, padding=p), 0.2)
    geo = conv.Geom(k, k, s, p)
    wf, _, ldw = conv.prep_weight(wt, 0, geo)
    nchunks = (geo.ntaps * cin + 31) // 32
    y = conv.conv_forward(conv.to_nhwc(x), wf, ldw, cout, geo, bias=b, act=conv.ACT_LRELU, force_tile=tile,
                          force_split=min(split, nchunks))
    oc.assert_close('tile %d split %d' % (tile, split), y, ref, 1e-4)


@pytest.mark.parametrize("geom", GEOMS[:4])
@pytest.mark.parametrize("tile,split", [(0, 0), (1, 1), (1, 3), (2, 2), (3, 1), (3, 2), (5, 1), (5, 3), (6, 1), (6, 2), (7, 1), (7, 3), (8, 1), (8, 2)])
def test_wgrad_tiles(emu_lib, geom, tile, split):
    ops, conv = oc.pkg()
    n, cin, h, w, cout, k, s, p = geom
    g = torch.Generator().manual_seed(2000 + tile * 10 + split)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = (torch.randn(cout, cin, k, k, generator=g) * 0.2).requires_grad_(True)
    y = F.conv2d(x, wt, None, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    geo = conv.Geom(k, k, s, p)
    dw = conv.conv_wgrad(conv.to_nhwc(x), conv.to_nhwc(dy), geo, (cout, cin, k, k), force_tile=tile, force_split=split)
    oc.assert_close('wgrad tile %d split %d' % (tile, split), dw, wt.grad, 1e-4)



def test_all_tiles_are_bitwise_identical_without_split(emu_lib):
    """every tile template walks K in the same order (chunk by chunk, two k per MFMA, one fma chain per output element), so
    without split-K the choice of tile - including the few-wave and double-buffered variants - cannot change a single bit"""
    import tile_checks as tc
    tc.check_bitwise_tiles(DEV)


def test_experimental_tiles_with_split(emu_lib):
    import tile_checks as tc
    tc.check_split_tiles(DEV)
    tc.check_wgrad_few_wave(DEV)
