"""Forced-tile parity of the fp32 gather-GEMM and weight-gradient kernels: every tile the launcher instantiates (the plan only
ever picks a few per shape) against F.conv2d on ragged geometries - partial row / column tiles, K not a multiple of 32, taps
that leave the image, split-K with atomics, stride 2, per-sample weights.  Shared by test_tiles_emu.py (SIMT emulator) and the
GPU suite; `python tests/tile_checks.py` runs it on cuda:0."""
import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import op_checks as oc  # noqa: E402

# (n, cin, h, w, cout, k, stride, pad)
GEOMS = [(2, 8, 9, 11, 40, 3, 1, 1),          # K = 72 (tail chunk), ragged M and N
         (1, 36, 7, 5, 130, 3, 1, 1),         # Cin not a multiple of 32: a chunk spans two taps; N > 128
         (2, 64, 6, 6, 64, 1, 1, 0),          # 1x1
         (1, 16, 10, 9, 33, 4, 2, 2),         # discriminator geometry (k4 s2 p2)
         (1, 4, 13, 12, 96, 3, 1, 1),         # Cin = 4: eight taps per chunk
         (3, 32, 1, 40, 64, 1, 1, 0),         # one image row (nn.Linear as a 1x1 convolution): 32-pixel steps cross images
         (2, 8, 3, 5, 64, 3, 1, 1)]           # 15 pixels per image: the weight-gradient kernel's scalar-twin geometry
FWD_TILES = (0, 1, 2, 4, 9)
# force_tile-only variants that the launch plan never picks (prefetch distance 2, csrc/conv_igemm.hip): checked on the emulator
# only until they have been measured on hardware; (variant, the plan's tile with the same dimensions)
# kernel variants of the plan's tile shapes (csrc/conv_igemm.hip fsv_conv_variant): 10 - 12 prefetch distance 2, 13 - 15 the same with
# in-place A fragments, 16 - 18 in-place A fragments on the prefetch-distance-1 tiles; (variant, the base tile of the same shape)
EXPERIMENTAL_FWD_TILES = ((10, 9), (11, 0), (12, 1), (13, 9), (14, 0), (15, 1), (16, 0), (17, 4), (18, 2), (20, 4))
# 21 / 22: global loads straight into LDS; they pair the k of an MFMA step as (k, k + 2) - same products, another order of the fp32
# chain - so they are held to F.conv2d at the kernels' tolerance, not to the base tile's bits
REORDERED_FWD_TILES = (21, 22, 27)
WGRAD_TILES = (0, 1, 2, 3, 4)


def _fwd_case(device, geom, tile, split, seed):
    ops, conv = oc.pkg()
    n, cin, h, w, cout, k, s, p = geom
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
    b = torch.randn(cout, generator=g)
    ref = F.leaky_relu(F.conv2d(x, wt, b, stride=s, padding=p), 0.2)
    ge = conv.Geom(k, k, s, p)
    wf, kpad, ldw = conv.prep_weight(wt.to(device), 0, ge)
    nchunks = (k * k * cin + 31) // 32
    y = conv.conv_forward(conv.to_nhwc(x.to(device)), wf, ldw, cout, ge, bias=b.to(device), act=conv.ACT_LRELU,
                          force_tile=tile, force_split=min(split, nchunks))
    oc.assert_close('tile %d split %d %s' % (tile, split, geom), y, ref, 2e-5)


def check_forward_tiles(device, tiles=FWD_TILES, geoms=GEOMS):
    for gi, geom in enumerate(geoms):
        for tile in tiles:
            bn = {0: 128, 1: 64, 2: 32, 4: 64, 9: 128, 10: 128, 11: 128, 12: 64, 13: 128, 14: 128, 15: 64, 16: 128, 17: 64, 18: 32, 20: 64, 21: 128, 22: 64, 27: 64}[tile]
            if geom[4] < bn // 2 and bn > 32:
                continue                       # a tile twice as wide as the layer: not a configuration the plan can produce
            for split in (1, 3):
                _fwd_case(device, geom, tile, split, 100 + gi)


def check_variant_equals_plan_tile(device, variant, base, geoms=GEOMS, splits=(1, 2, 3, 5)):
    """a variant that only moves the global loads in time must reproduce the plan's tile bit for bit (same fma chains)"""
    ops, conv = oc.pkg()
    for gi, geom in enumerate(geoms):
        n, cin, h, w, cout, k, s, p = geom
        g = torch.Generator().manual_seed(300 + gi)
        x = conv.to_nhwc(torch.randn(n, cin, h, w, generator=g).to(device))
        wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
        ge = conv.Geom(k, k, s, p)
        wf, kpad, ldw = conv.prep_weight(wt.to(device), 0, ge)
        nchunks = (k * k * cin + 31) // 32
        for split in splits:
            sp = min(split, nchunks)
            a = conv.conv_forward(x, wf, ldw, cout, ge, force_tile=variant, force_split=sp)
            b = conv.conv_forward(x, wf, ldw, cout, ge, force_tile=base, force_split=sp)
            assert torch.equal(a.cpu(), b.cpu()), (variant, base, geom, split)


def check_per_sample(device, tiles=(1, 4, 9)):
    """batch_conv form: one weight matrix per sample (grid.z = sample)"""
    ops, conv = oc.pkg()
    g = torch.Generator().manual_seed(7)
    n, cin, h, w, cout = 3, 16, 9, 10, 72
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(n, cout, cin, 1, 1, generator=g) * 0.2
    ref = torch.cat([F.conv2d(x[i:i + 1], wt[i]) for i in range(n)])
    ge = conv.Geom(1, 1, 1, 0)
    wf, kpad, ldw = conv.prep_weight(wt.to(device), 0, ge)
    for tile in tiles:
        y = conv.conv_forward(conv.to_nhwc(x.to(device)), wf, ldw, cout, ge, per_sample=True, force_tile=tile)
        oc.assert_close('per-sample tile %d' % tile, y, ref, 2e-5)


def check_wgrad_tiles(device, tiles=WGRAD_TILES, geoms=GEOMS):
    ops, conv = oc.pkg()
    for gi, geom in enumerate(geoms):
        n, cin, h, w, cout, k, s, p = geom
        g = torch.Generator().manual_seed(200 + gi)
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.zeros(cout, cin, k, k, requires_grad=True)
        y = F.conv2d(x, wt, stride=s, padding=p)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        ge = conv.Geom(k, k, s, p)
        for tile in tiles:
            for split in (1, 2):
                dw = conv.conv_wgrad(conv.to_nhwc(x.to(device)), conv.to_nhwc(dy.to(device)), ge, (cout, cin, k, k),
                                     force_tile=tile, force_split=split)
                oc.assert_close('wgrad tile %d split %d %s' % (tile, split, geom), dw, wt.grad, 2e-5)


def run_all(device):
    check_forward_tiles(device)
    check_per_sample(device)
    check_wgrad_tiles(device)
    # the kernel variants the plan's shapes run as: against F.conv2d and (unsplit: split launches add through atomics, whose
    # order is not reproducible on hardware) bit-equal to the base tile of the same shape
    for variant, base in EXPERIMENTAL_FWD_TILES:
        check_forward_tiles(device, tiles=(variant,))
        check_variant_equals_plan_tile(device, variant, base, splits=(1,))
    check_forward_tiles(device, tiles=REORDERED_FWD_TILES)


if __name__ == '__main__':
    run_all(torch.device('cuda:0'))
    print('TILES_GPU_OK')
