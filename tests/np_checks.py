"""Checks of the narrow-operand (`--amp`) GEMM kernels (csrc/conv_np.hip) against their CPU definition
(oracle/np_oracle.py), parameterised by device: the emulator tests (test_np_emu.py) and the GPU test (test_zz_np_gpu.py, which
runs this file as a subprocess) share them.  Products of 16-bit operands are exact in fp32, so kernel and oracle differ only by
the fp32 summation order: the tolerance is an fp32 one, far below the gap between the narrow and the exact result."""
import os
import sys

import torch
import torch.nn.functional as F

import op_checks as oc

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import np_oracle as NO   # noqa: E402

TOL = 2e-5

GEOMS = [  # n, cin, h, w, cout, k, stride, pad
    (1, 8, 9, 7, 70, 3, 1, 1),
    (2, 12, 11, 13, 130, 3, 2, 1),
    (1, 20, 10, 9, 40, 4, 2, 2),
    (1, 36, 6, 5, 200, 1, 1, 0),
    (2, 64, 8, 8, 64, 3, 1, 1),
]
# the data gradient is a gather-GEMM over the OUTPUT channels: it only has a narrow kernel when Cout % 4 == 0 (otherwise it
# runs exact fp32, like every scalar-gather layer), so the end-to-end cases use such channel counts
AUTOGRAD_GEOMS = [(1, 8, 9, 7, 72, 3, 1, 1), (2, 12, 11, 13, 132, 3, 2, 1), (1, 20, 10, 9, 40, 4, 2, 2), (1, 36, 6, 5, 200, 1, 1, 0),
                  (2, 64, 8, 8, 64, 3, 1, 1), (1, 16, 7, 8, 24, 4, 2, 1)]
# sizes at which whole tiles, several workgroups per dimension and real split-K factors occur (GPU run)
BIG_GEOMS = [(2, 64, 64, 64, 128, 3, 1, 1), (2, 128, 32, 32, 256, 3, 2, 1), (1, 256, 16, 16, 512, 4, 2, 1), (2, 512, 16, 16, 64, 1, 1, 0)]
FWD_TILES = [(-1, 0), (0, 1), (0, 3), (1, 1), (2, 1), (4, 2), (9, 1), (9, 2)]
WGRAD_TILES = [(0, 0), (1, 1), (1, 3), (2, 2), (3, 1), (3, 2), (4, 1)]


class mode_scope:
    def __init__(self, mode):
        _, self.conv = oc.pkg()
        self.mode = mode

    def __enter__(self):
        self.prev = self.conv.set_mfma_mode(self.mode)
        # these checks pin the staging-time narrowing kernels (csrc/conv_np.hip); since round 4 the product's f16 mode runs the
        # half-precision kernels (csrc/conv_h.hip, tests/h_checks.py) unless they are switched off
        self.prev_h = self.conv.set_h_kernels(False)
        return self.conv

    def __exit__(self, *a):
        self.conv.set_mfma_mode(self.prev)
        self.conv.set_h_kernels(self.prev_h)


def check_forward(device, mode, geom, tile, split, seed=3000):
    n, cin, h, w, cout, k, s, p = geom
    g = torch.Generator().manual_seed(seed + mode * 100 + tile * 10 + split)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    ref = F.leaky_relu(NO.conv2d(x, wt, b, s, p, mode), 0.2)
    with mode_scope(mode) as conv:
        geo = conv.Geom(k, k, s, p)
        wf, _, ldw = conv.prep_weight(wt.to(device), 0, geo)
        nchunks = (geo.ntaps * cin + 31) // 32
        y = conv.conv_forward(conv.to_nhwc(x.to(device)), wf, ldw, cout, geo, bias=b.to(device), act=conv.ACT_LRELU,
                              force_tile=tile, force_split=min(split, nchunks))
    oc.assert_close('np fwd mode %d tile %d split %d' % (mode, tile, split), y, ref, TOL)


def check_wgrad(device, mode, geom, tile, split, seed=4000):
    n, cin, h, w, cout, k, s, p = geom
    g = torch.Generator().manual_seed(seed + mode * 100 + tile * 10 + split)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = (torch.randn(cout, cin, k, k, generator=g) * 0.2).requires_grad_(True)
    y = NO.conv2d(x, wt, None, s, p, mode)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    with mode_scope(mode) as conv:
        geo = conv.Geom(k, k, s, p)
        dw = conv.conv_wgrad(conv.to_nhwc(x.to(device)), conv.to_nhwc(dy.to(device)), geo, (cout, cin, k, k), force_tile=tile,
                             force_split=split)
    oc.assert_close('np wgrad mode %d tile %d split %d' % (mode, tile, split), dw, wt.grad, TOL)


def check_autograd(device, mode, geom, seed=5000):
    """ops.conv2d end to end (forward, data gradient incl. the stride-2 classes, weight and bias gradient)"""
    ops, _ = oc.pkg()
    n, cin, h, w, cout, k, s, p = geom
    g = torch.Generator().manual_seed(seed + mode)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = NO.conv2d(xr, wr, br, s, p, mode)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xd, wd, bd = [t.to(device).requires_grad_(True) for t in (x, wt, b)]
    with mode_scope(mode):
        y = ops.conv2d(xd, wd, bd, stride=s, padding=p)
        y.backward(dy.to(device))
    oc.assert_close('np y', y, ref, TOL)
    oc.assert_close('np dx', xd.grad, xr.grad, TOL)
    oc.assert_close('np dw', wd.grad, wr.grad, TOL)
    oc.assert_close('np db', bd.grad, br.grad, TOL)


def check_batch_conv(device, mode, seed=6000):
    """per-sample generated weights (base_network.py:56-71) on the narrow kernel"""
    ops, _ = oc.pkg()
    g = torch.Generator().manual_seed(seed + mode)
    bsz, cin, cout, h, w = 2, 16, 24, 6, 5
    x = torch.randn(bsz, cin, h, w, generator=g)
    wt = torch.randn(bsz, cout, cin, 1, 1, generator=g) * 0.3
    b = torch.randn(bsz, cout, generator=g)
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    ref = torch.cat([NO.conv2d(xr[i:i + 1], wr[i], b[i], 1, 0, mode) for i in range(bsz)])
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xd, wd = x.to(device).requires_grad_(True), wt.to(device).requires_grad_(True)
    with mode_scope(mode):
        y = ops.batch_conv(xd, wd, b.to(device))
        y.backward(dy.to(device))
    oc.assert_close('np bc y', y, ref, TOL)
    oc.assert_close('np bc dx', xd.grad, xr.grad, TOL)
    oc.assert_close('np bc dw', wd.grad, wr.grad, TOL)


def check_modes_differ(device, seed=7000):
    """the narrow results sit where their definition puts them: ~5e-4 (f16) and ~1e-5 (bf16x3) away from the exact convolution"""
    _, conv = oc.pkg()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, 64, 12, 12, generator=g)
    wt = torch.randn(64, 64, 3, 3, generator=g) * 0.1
    exact = F.conv2d(x.double(), wt.double(), padding=1)
    geo = conv.Geom(3, 3, 1, 1)
    wf, _, ldw = conv.prep_weight(wt.to(device), 0, geo)
    errs = {}
    for mode in (0, 1, 2):
        with mode_scope(mode):
            y = conv.conv_forward(conv.to_nhwc(x.to(device)), wf, ldw, 64, geo)
        errs[mode] = float((y.double().cpu() - exact).abs().max() / exact.abs().max())
    assert errs[0] < 2e-6, errs
    assert 5e-5 < errs[1] < 3e-3, errs
    assert errs[0] < errs[2] < 5e-5, errs
    return errs


def check_scalar_gather_stays_fp32(device, seed=7001):
    """Cin % 4 != 0 (3-channel images, label maps) has no narrow kernel: those layers run the exact fp32 path"""
    _, conv = oc.pkg()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, 3, 9, 9, generator=g)
    wt = torch.randn(16, 3, 3, 3, generator=g)
    geo = conv.Geom(3, 3, 1, 1)
    wf, _, ldw = conv.prep_weight(wt.to(device), 0, geo)
    with mode_scope(1):
        y = conv.conv_forward(conv.to_nhwc(x.to(device)), wf, ldw, 16, geo)
    oc.assert_close('fp32 fallthrough', y, F.conv2d(x, wt, padding=1), 1e-5)


def run_all(device, big=False):
    """everything above in one go (the GPU subprocess); `big` adds tile-filling sizes"""
    for mode in (1, 2):
        for geom in GEOMS + (BIG_GEOMS if big else []):
            for tile, split in FWD_TILES:
                check_forward(device, mode, geom, tile, split)
            for tile, split in WGRAD_TILES:
                check_wgrad(device, mode, geom, tile, split)
        for geom in AUTOGRAD_GEOMS + (BIG_GEOMS if big else []):
            check_autograd(device, mode, geom)
        check_batch_conv(device, mode)
    errs = check_modes_differ(device)
    check_scalar_gather_stays_fp32(device)
    return errs


if __name__ == '__main__':
    # GPU entry (tests/test_zz_np_gpu.py): operator checks at small and tile-filling sizes, then one --amp training
    # iteration of the tiny model in both modes and the overflow / skip rule, all on cuda:0
    import model_checks as mc
    dev = torch.device('cuda', 0)
    errs = run_all(dev, big=True)
    print('operator checks ok; |narrow - exact| / max:', errs, flush=True)
    mc.check_amp_overflow_skip(dev)
    kw = dict(warp_ref=True, spade_combine=True, remove_face_labels=True)
    mc.check_amp_step(dev, kw, 'O1')
    mc.check_amp_step(dev, kw, 'bf16x3', loss_tol=1e-3, image_tol=1e-3, grad_l2_tol=2e-2)
    print('NP_GPU_OK', flush=True)
