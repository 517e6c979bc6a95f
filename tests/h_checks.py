"""Checks of the half-precision convolution kernels (csrc/conv_h.hip, few-shot-vid2vid_amd/hconv.py) against their CPU definition
(oracle/np_oracle.py: operands in IEEE half, exact products, fp32 accumulation, one rounding at a half output), parameterised by
device: the emulator tests (test_h_emu.py) and the GPU tests (test_h_gpu.py) share them.  Kernel and definition differ by the fp32
summation order only; a half OUTPUT may in addition land on the neighbouring half value when the fp32 sums straddle a rounding
boundary, hence the half-ulp term of the half-output tolerance."""
import os
import sys

import torch
import torch.nn.functional as F

import op_checks as oc

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import np_oracle as NO   # noqa: E402

TOL = 3e-5
HALF_ULP = 2.0 ** -10          # relative spacing of halves: one ulp of a rounded result

GEOMS = [  # n, cin, h, w, cout, k, stride, pad
    (1, 8, 9, 7, 72, 3, 1, 1),
    (2, 16, 11, 13, 136, 3, 2, 1),
    (1, 24, 10, 9, 40, 4, 2, 2),
    (1, 40, 6, 5, 200, 1, 1, 0),
    (2, 64, 8, 8, 64, 3, 1, 1),
    (1, 32, 12, 12, 3, 3, 1, 1),
]
# weight-gradient geometries (output maps large enough for the 64-pixel walk): stride 2 with odd sizes, 4x4 taps, a ragged Cout
WG_GEOMS = [(2, 16, 21, 27, 136, 3, 2, 1), (1, 64, 8, 8, 64, 3, 1, 1), (1, 24, 30, 17, 40, 4, 2, 2), (3, 8, 9, 16, 200, 1, 1, 0)]
BIG_GEOMS = [(2, 64, 64, 64, 128, 3, 1, 1), (2, 128, 32, 32, 256, 3, 2, 1), (1, 256, 16, 16, 512, 4, 2, 1), (2, 512, 16, 16, 64, 1, 1, 0),
             (1, 32, 96, 128, 32, 3, 1, 1)]
FWD_TILES = [(-1, 0), (0, 1), (0, 3), (1, 1), (2, 1), (3, 1), (3, 2), (4, 2), (5, 1), (5, 3), (9, 1), (16, 1), (16, 2), (17, 1), (18, 1), (19, 1),
             (20, 3), (21, 2), (25, 1)]
WGRAD_TILES = [(0, 0), (1, 1), (1, 3), (2, 2), (3, 1), (4, 1), (4, 2), (5, 1), (6, 1), (6, 2)]


def _mods():
    from importlib import import_module
    import fsv2v_amd  # noqa: F401
    return import_module('few-shot-vid2vid_amd.conv'), import_module('few-shot-vid2vid_amd.hconv')


def _h(x):
    return x.to(torch.float16).to(torch.float32)


def close_half(name, got, want, tol=TOL):
    """got: a half tensor the kernel stored; want: the fp32 value before rounding"""
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    err = (got - want).abs()
    lim = tol * want.abs().max() + HALF_ULP * want.abs() * 1.01 + 6e-8
    bad = err > lim
    assert not bool(bad.any()), (name, float(err.max()), float(want.abs().max()), int(bad.sum()))


def check_forward(device, geom, tile, split, out_half, seed=7000, res_half=None, act=True):
    conv, hc = _mods()
    n, cin, h, w, cout, k, s, p = geom
    g = torch.Generator().manual_seed(seed + tile * 10 + split)
    x = _h(torch.randn(n, cin, h, w, generator=g))
    wt = torch.randn(cout, cin, k, k, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, _h(wt), b, stride=s, padding=p)
    if act:
        ref = F.leaky_relu(ref, 0.2)
    res = None
    if res_half is not None:
        res = torch.randn(ref.shape, generator=g)
        if res_half:
            res = _h(res)
        ref = ref + res
    geo = conv.Geom(k, k, s, p)
    wf, _, ldw = conv.prep_weight(wt.to(device), 0, geo)
    wh, kpad, nrows = hc.prep_weight_h(wf)
    nchunks = (geo.ntaps * cin + 63) // 64
    rd = None if res is None else conv.to_nhwc(res.to(device).to(torch.float16 if res_half else torch.float32))
    y = hc.conv_forward_h(hc.to_half_nhwc(x.to(device)), wh, kpad, nrows, cout, geo, bias=b.to(device),
                          act=conv.ACT_LRELU if act else conv.ACT_NONE, res=rd, out_half=out_half, force_tile=tile,
                          force_split=min(split, nchunks))
    assert y.dtype == (torch.float16 if out_half else torch.float32)
    name = 'h fwd tile %d split %d half %d %s' % (tile, split, out_half, geom)
    if out_half:
        close_half(name, y, ref)
    else:
        oc.assert_close(name, y, ref, TOL)


def check_wgrad(device, geom, tile, split, seed=8000):
    conv, hc = _mods()
    n, cin, h, w, cout, k, s, p = geom
    g = torch.Generator().manual_seed(seed + tile * 10 + split)
    x = _h(torch.randn(n, cin, h, w, generator=g))
    wt = (torch.randn(cout, cin, k, k, generator=g) * 0.2).requires_grad_(True)
    y = F.conv2d(x, wt, None, stride=s, padding=p)
    dy = _h(torch.randn(y.shape, generator=g))
    y.backward(dy)
    geo = conv.Geom(k, k, s, p)
    oh, ow = geo.out_hw(h, w)
    if not hc.wgrad_eligible(cin, cout, oh, ow):
        return False
    dwt = hc.conv_wgrad_h(hc.to_half_nhwc(x.to(device)), hc.to_half_nhwc(dy.to(device)), geo, force_tile=tile, force_split=split)
    dw = conv.unprep_weight_grad(dwt, (cout, cin, k, k), geo)
    oc.assert_close('h wgrad tile %d split %d %s' % (tile, split, geom), dw, wt.grad, TOL)
    return True


def check_dgrad(device, geom, out_half, seed=9000):
    conv, hc = _mods()
    n, cin, h, w, cout, k, s, p = geom
    if cout % 8:
        return False
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g).requires_grad_(True)
    wt = torch.randn(cout, cin, k, k, generator=g) * 0.2
    y = F.conv2d(x, _h(wt), None, stride=s, padding=p)
    dy = _h(torch.randn(y.shape, generator=g))
    y.backward(dy)
    geo = conv.Geom(k, k, s, p)
    layouts = []
    for c in geo.dgrad_classes:
        if not c['khs']:
            layouts.append(None)
            continue
        wd, _, _ = conv.prep_weight(wt.to(device), 1, geo, c['khs'], c['kws'])
        layouts.append(hc.prep_weight_h(wd))
    dx = hc.conv_dgrad_h(hc.to_half_nhwc(dy.to(device)), layouts, geo, (h, w), cin, out_half=out_half)
    name = 'h dgrad half %d %s' % (out_half, geom)
    if out_half:
        close_half(name, dx, x.grad)
    else:
        oc.assert_close(name, dx, x.grad, TOL)
    return True


def check_group(device, seed=9500):
    """independent problems of different sizes in ONE grid == the same problems one by one"""
    conv, hc = _mods()
    g = torch.Generator().manual_seed(seed)
    probs = [(8, 16, 24, 40), (3, 40, 8, 136), (16, 64, 64, 64), (5, 8, 200, 72)]     # rows, cin, cout
    singles, outs, refs = [], [], []
    geo = conv.Geom(1, 1, 1, 0)
    with conv.launch_group(True):
        for rows, cin, cout, _ in probs:
            x = _h(torch.randn(1, cin, 1, rows, generator=g))
            wt = torch.randn(cout, cin, 1, 1, generator=g) * 0.3
            b = torch.randn(cout, generator=g)
            refs.append(F.leaky_relu(F.conv2d(x, _h(wt), b), 0.2))
            wf, _, _ = conv.prep_weight(wt.to(device), 0, geo)
            wh, kpad, nrows = hc.prep_weight_h(wf)
            outs.append(hc.conv_forward_h(hc.to_half_nhwc(x.to(device)), wh, kpad, nrows, cout, geo, bias=b.to(device),
                                          act=conv.ACT_LRELU, out_half=False))
    for o, r in zip(outs, refs):
        oc.assert_close('h group', o, r, TOL)


def check_stats(device, seed=9700):
    """the epilogue's per-channel sums == sums over the stored (rounded) output"""
    conv, hc = _mods()
    g = torch.Generator().manual_seed(seed)
    n, cin, h, w, cout = 2, 16, 16, 16, 48
    x = _h(torch.randn(n, cin, h, w, generator=g))
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.2
    geo = conv.Geom(3, 3, 1, 1)
    wf, _, _ = conv.prep_weight(wt.to(device), 0, geo)
    wh, kpad, nrows = hc.prep_weight_h(wf)
    for groups in (1, n):
        st = dict(groups=groups)
        with conv.stats_pass(x.device if False else torch.device(device)):
            y = hc.conv_forward_h(hc.to_half_nhwc(x.to(device)), wh, kpad, nrows, cout, geo, out_half=True, stats=st)
        if 'part' not in st:
            continue
        part = st['part'].view(groups, st['slots'], cout, 2).sum(dim=1).cpu()
        yf = y.float().cpu().view(groups, n // groups, cout, h * w)
        s1 = yf.double().sum(dim=(1, 3))
        s2 = (yf.double() ** 2).sum(dim=(1, 3))
        assert float((part[..., 0] - s1).abs().max()) <= 1e-3 * float(s1.abs().max() + 1), (groups, 'sum')
        assert float((part[..., 1] - s2).abs().max()) <= 1e-3 * float(s2.abs().max() + 1), (groups, 'sumsq')


def check_cast(device, seed=9800):
    conv, hc = _mods()
    g = torch.Generator().manual_seed(seed)
    for nelem in (1, 7, 1024, 4099):
        x = torch.randn(nelem, generator=g) * 100
        y = hc.cast(x.to(device), torch.float16)
        assert y.dtype == torch.float16 and bool((y.cpu() == x.to(torch.float16)).all())
        z = hc.cast(y, torch.float32)
        assert bool((z.cpu() == x.to(torch.float16).float()).all())


if __name__ == '__main__':
    dev = torch.device('cuda', 0)
    check_cast(dev)
    for geom in GEOMS + BIG_GEOMS:
        for half in (True, False):
            check_forward(dev, geom, -1, 0, half)
            check_dgrad(dev, geom, half)
        check_wgrad(dev, geom, 0, 0)
    for tile, split in FWD_TILES:
        check_forward(dev, BIG_GEOMS[0], tile, split, True)
        check_forward(dev, BIG_GEOMS[2], tile, split, False, res_half=True)
    for tile, split in WGRAD_TILES:
        check_wgrad(dev, BIG_GEOMS[1], tile, split)
    check_group(dev)
    check_stats(dev)
    print('H_GPU_OK', flush=True)
