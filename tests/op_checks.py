"""Operator parity checks shared by the emulated (`not gpu`) and the real-hardware (`gpu`) test modules.

Every check runs one HIP operator (through the C ABI) and the CPU oracle on the same seeded inputs and compares
values and gradients.  Tolerance: 1e-3 relative to the largest reference magnitude (BASELINE.json north_star),
most checks are far tighter; integer tap indices of the warp must match bit for bit.
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import fsv_oracle as O  # noqa: E402

REL_TOL = 1e-3


def pkg():
    import fsv2v_amd  # noqa: F401
    from importlib import import_module
    return import_module('few-shot-vid2vid_amd.ops'), import_module('few-shot-vid2vid_amd.conv')


def assert_close(name, got, ref, tol=REL_TOL):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, '%s: shape %s vs %s' % (name, tuple(got.shape), tuple(ref.shape))
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, '%s: max|diff| %.3e > %.1e * %.3e' % (name, err, tol, scale)
    return err / scale


def _dev(t, device):
    return t.to(device) if t is not None else None


def _sinkify(t, cache, fin):
    """what FlatAdam does to a parameter: persistent layouts, gradient sink, deferred weight-gradient jobs"""
    if cache is not None:
        t._fsv_cache = cache
    if fin is not None:
        t.grad = torch.zeros_like(t)
        t._fsv_sink = True
        t._fsv_finalizer = fin


def check_conv(device, n, cin, h, w, cout, k, s, p, act='lrelu', bias=True, seed=0, tol=REL_TOL, cache=None, fin=None):
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
    b = torch.randn(cout, generator=g) if bias else None
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    ref = F.conv2d(xr, wr, br, stride=s, padding=p)
    actc = {'none': conv.ACT_NONE, 'lrelu': conv.ACT_LRELU, 'tanh': conv.ACT_TANH, 'sigmoid': conv.ACT_SIGMOID}[act]
    ref = {'none': lambda t: t, 'lrelu': O.actvn, 'tanh': torch.tanh, 'sigmoid': torch.sigmoid}[act](ref)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xd, wd = _dev(x, device).requires_grad_(True), _dev(wt, device).requires_grad_(True)
    bd = _dev(b, device).requires_grad_(True) if bias else None
    _sinkify(wd, cache, fin)
    if bias:
        _sinkify(bd, None, fin)             # bias gradient: deferred grouped column sums
    y = ops.conv2d(xd, wd, bd, stride=s, padding=p, act=actc)
    y.backward(_dev(dy, device))
    if fin is not None:
        assert fin.pending() and len(fin.bias_jobs) == (1 if bias else 0)
        fin.run()
    assert_close('conv y', y, ref, tol)
    assert_close('conv dx', xd.grad, xr.grad, tol)
    assert_close('conv dw', wd.grad, wr.grad, tol)
    if bias:
        assert_close('conv db', bd.grad, br.grad, tol)


def check_conv_up(device, n=2, cin=16, h=6, w=10, cout=24, k=3, act='lrelu', stats=0, tile=-1, split=0, seed=71, expect_fold=True,
                  amp=False, cache=None):
    """conv(nearest_x2(x)) (nn.Upsample in front of a convolution: generator.py:124,489-493,559-563) with the up-sampling folded
    into the gather (csrc/conv_igemm.hip ConvP::up, forward and weight gradient; round 5) against F.conv2d(F.interpolate(x)):
    output, dx (the 2 x 2 pooled data gradient), dw, db; bit-equal to the same convolution on the materialised tensor
    (FSV_UP_FOLD=0) - the fold changes addresses, not arithmetic.  h, w: size of x (the convolution sees 2h x 2w).  stats: the
    BatchNorm-statistics epilogue rides along.  expect_fold=False: layers the fold does not cover materialise silently."""
    from importlib import import_module
    ops, conv = pkg()
    lib = import_module('few-shot-vid2vid_amd.lib')
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
    b = torch.randn(cout, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(F.interpolate(xr, scale_factor=2, mode='nearest'), wr, br, padding=k // 2)
    actc = {'none': conv.ACT_NONE, 'lrelu': conv.ACT_LRELU}[act]
    ref = O.actvn(ref) if act == 'lrelu' else ref
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)

    def run(fold, direct=True, subpixel=False):
        xd = _dev(x, device).detach().clone().requires_grad_(True)
        wd = _dev(wt, device).detach().clone().requires_grad_(True)
        bd = _dev(b, device).detach().clone().requires_grad_(True)
        if cache is not None:          # an optimiser-owned weight: persistent layouts, the summed-tap ones included (round 6)
            wd._fsv_cache = cache
        seen, real_call = [], lib.call

        def recording_call(name, *a):
            seen.append((name if name != 'fsv_prep_weight' else 'fsv_prep_weight:mode%d' % a[3], a))
            return real_call(name, *a)
        os.environ['FSV_UP_FOLD'] = '1' if fold else '0'
        os.environ['FSV_UP_DGRAD'] = '1' if direct else '0'
        os.environ['FSV_UP_SUBPIXEL'] = '1' if subpixel else '0'
        lib.call = recording_call
        try:
            with conv.stats_pass(xd.device):
                y = ops.conv2d(xd, wd, bd, stride=1, padding=k // 2, act=actc, stats_groups=stats, up=True)
            got_stats = getattr(y, '_fsv_stats', None) is not None
            y.backward(_dev(dy, device))
        finally:
            lib.call = real_call
            os.environ.pop('FSV_UP_FOLD', None)
            os.environ.pop('FSV_UP_DGRAD', None)
            os.environ.pop('FSV_UP_SUBPIXEL', None)
        return y.detach(), xd.grad, wd.grad, bd.grad, [s_[0] for s_ in seen], got_stats
    prev = conv.set_mfma_mode(1 if amp else conv.mfma_mode())
    try:
        a = run(True)
        bm = run(False)
        pooled = run(True, direct=False)
        sub = run(True, subpixel=True)
    finally:
        conv.set_mfma_mode(prev)
    # the forward of the large layers: four 2x2-tap launches over the source pixels, one per output parity class, with summed
    # weights (ops._up_subpixel_forward: 2.25x fewer MACs) - one fp32 rounding per summed weight away from the single gather
    want_sub = bool(not amp and k == 3 and n * h * w >= 8192 and cin % 4 == 0 and 'fsv_upsample2x_fwd' not in a[4])
    gathers = lambda names: sum(1 for s_ in names if s_.startswith('fsv_conv_gather_fwd'))      # (..._stats: the statistics epilogue)
    assert (gathers(sub[4]) == gathers(a[4]) + 3) == want_sub, (want_sub, sub[4])
    for name, u, v in zip(('y', 'dx', 'dw', 'db'), sub[:4], a[:4]):
        if want_sub:
            assert_close('conv(up2x) %s: sub-pixel forward vs one gather' % name, u, v, 2e-6 if act == 'none' else REL_TOL)
        elif name == 'dw':
            assert_close('conv(up2x) dw, same launches twice', u, v, 1e-6)            # (pixel-split atomics: summation order)
        else:
            assert bool((u == v).all()), name
    folded = 'fsv_upsample2x_fwd' not in a[4]
    assert folded == expect_fold, a[4]
    # the data gradient w.r.t. x: ONE gather-GEMM (4x4 stride-2 convolution over dy with the 2 x 2 pooling folded into summed weights,
    # ops._up_dgrad_weight) where the layer allows - 3x3, exact fp32, float4 channels -, else data gradient at the up-sampled size +
    # fsv_upsample2x_bwd; the two agree to the rounding of the summed weights
    direct = k == 3 and not amp and cout % 4 == 0 and cin % 4 == 0 and cin > 4
    assert 'fsv_upsample2x_fwd' in bm[4] and ('fsv_upsample2x_bwd' in a[4]) == (not direct) and 'fsv_upsample2x_bwd' in pooled[4], (a[4], pooled[4])
    assert_close('conv(up2x) dx: one-launch data gradient vs pooled', a[1], pooled[1], 1e-5 if not amp else 3e-3)
    if stats and not amp:
        assert a[5] == bm[5]
    tol = 3e-3 if amp else REL_TOL
    if folded and not amp:
        # the folded forward gathers through the up-sampling index: the materialised tensor's bits; the data gradient is the
        # one-launch form on both sides
        for name, u, v in zip(('y', 'dx', 'db'), (a[0], a[1], a[3]), (bm[0], bm[1], bm[3])):
            assert bool((u == v).all()), 'folded up-sampling changed the bits of %s' % name
        assert_close('conv(up2x) dw folded vs materialised', a[2], bm[2], 1e-6)      # (pixel-split atomics: summation order)
    # (with a LeakyReLU epilogue and ~10^6 outputs a few pre-activations lie within rounding of the kink: their gradient takes slope 1
    # on one side and 0.2 on the other - the large cases are run with act='none'; hardware record: 7.5e-3 of max|dx| on one element)
    for name, got, want in (('y', a[0], ref), ('dx', a[1], xr.grad), ('dw', a[2], wr.grad), ('db', a[3], br.grad)):
        assert_close('conv(up2x) %s' % name, got, want, tol)
    if cache is not None and k == 3 and not amp and cin % 4 == 0 and folded:
        # the summed-tap layouts live in the layout cache: no per-call re-arrangement in any of the folded runs, and the cached
        # operands equal the per-call construction (0 / 1 coefficient product + re-arrangement) to the last rounding of a sum
        for r in (a, pooled, sub):
            assert 'fsv_prep_weight:mode0' not in r[4] and 'fsv_prep_weight:mode1' not in r[4], r[4]
        ups = [e for e in cache.entries if e.up_fwd is not None and tuple(e.weight.shape) == (cout, cin, 3, 3)]
        assert ups, 'no summed-tap layouts were registered'
        e = ups[-1]
        w4 = e.weight.detach()
        v, khs, kws, _, _ = ops._up_dgrad_weight(w4)
        refd, _, _ = conv.prep_weight(v, 1, conv.Geom(3, 3, 1, 1), khs, kws, None)
        assert_close('cached summed-tap data-gradient layout', e.up_dgrad[0], refd, 1e-6)
        cls = [(ry, rx) for ry in (0, 1) for rx in (0, 1)]
        khs = [2 * ry + iy for ry, rx in cls for iy in (0, 1) for _ in (0, 1)]
        kws = [2 * rx + ix for ry, rx in cls for _ in (0, 1) for ix in (0, 1)]
        reff, _, _ = conv.prep_weight(ops._tap_sums(w4, ops._SUBPIXEL_ROWS), 0, conv.Geom(3, 3, 1, 1), khs, kws, None)
        assert_close('cached summed-tap forward layout', e.up_fwd[0], reff, 1e-6)
    if want_sub:
        # the sub-pixel forward's own arithmetic (summed weights: one extra fp32 rounding) against F.conv2d(F.interpolate(x))
        # itself, not only against the product's other path (round-5 review, weak #2)
        for name, got, want in (('y', sub[0], ref), ('dx', sub[1], xr.grad), ('dw', sub[2], wr.grad), ('db', sub[3], br.grad)):
            assert_close('conv(up2x) %s, sub-pixel forward vs torch' % name, got, want, tol)
    if expect_fold and not amp and cin % 4 == 0 and cout % 32 == 0:
        # every tile shape of the plan has its own UP instantiation (and the LD form its own address swizzle): forced, forward,
        # against the same tile on the materialised tensor - bit for bit, K splits included
        geo = conv.Geom(k, k, 1, k // 2)
        xd = conv.to_nhwc(_dev(x, device))
        xu = conv.to_nhwc(_dev(F.interpolate(x, scale_factor=2, mode='nearest'), device))
        wf, _, ldw = conv.prep_weight(_dev(wt, device), 0, geo)
        for tile in (16, 15, 18, 20, 21):          # the variants the plan's shapes run as (fsv_conv_variant): same template either way
            for sp in (1, 2):
                if sp == 2 and (k * k * cin + 31) // 32 < 2:
                    continue
                u = conv.conv_forward(xd, wf, ldw, cout, geo, bias=_dev(b, device), act=actc, force_tile=tile, force_split=sp, up=True)
                v = conv.conv_forward(xu, wf, ldw, cout, geo, bias=_dev(b, device), act=actc, force_tile=tile, force_split=sp)
                assert bool((u == v).all()), 'folded up-sampling, tile %d split %d' % (tile, sp)
                assert_close('conv(up2x) tile %d' % tile, u, ref, REL_TOL)


def check_conv_up_spectral(device, n=2, cin=16, h=6, w=5, cout=24, seed=72, with_res=True):
    """conv3x3(nearest_x2(x)) of a SPECTRAL-NORMALISED layer (every up-sampling decoder convolution of the flow network,
    generator.py:479-496) with a residual: 1 / sigma reaches the sub-pixel forward and the one-launch data gradient as the epilogue
    scalar (their summed weights are built from the un-normalised W), the weight gradient carries the spectral-norm correction"""
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.2
    res = torch.randn(n, cout, 2 * h, 2 * w, generator=g)
    u = F.normalize(torch.randn(cout, generator=g), dim=0)
    v = F.normalize(torch.randn(cin * 9, generator=g), dim=0)
    sd = {'weight_orig': wt.clone().requires_grad_(True), 'weight_u': u.clone(), 'weight_v': v.clone()}
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    ref = F.conv2d(F.interpolate(xr, scale_factor=2, mode='nearest'), O.spectral_weight(sd, '', training=True), None, padding=1)
    ref = ref + rr if with_res else ref
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    ud, vd = _dev(u.clone(), device), _dev(v.clone(), device)
    wd = _dev(wt, device).detach().clone().requires_grad_(True)
    xd, rd = (_dev(t, device).detach().clone().requires_grad_(True) for t in (x, res))
    sig = ops.SpectralState.update(wd, ud, vd, training=True)
    y = ops.conv2d(xd, wd, None, stride=1, padding=1, res=rd if with_res else None, sn=(sig, ud, vd), up=True)
    y.backward(_dev(dy, device))
    assert_close('sn conv(up2x) y', y, ref)
    assert_close('sn conv(up2x) dx', xd.grad, xr.grad)
    assert_close('sn conv(up2x) dw', wd.grad, sd['weight_orig'].grad)
    if with_res:
        assert_close('sn conv(up2x) dres', rd.grad, rr.grad)


def check_conv_sn_res(device, seed=1, cache=None, fin=None):
    """spectral-norm conv with fused residual add (SPADEResnetBlock conv_1 + shortcut)."""
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    n, cin, h, w, cout = 2, 8, 6, 5, 12
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    res = torch.randn(n, cout, h, w, generator=g)
    u = F.normalize(torch.randn(cout, generator=g), dim=0)
    v = F.normalize(torch.randn(cin * 9, generator=g), dim=0)
    sd = {'weight_orig': wt.clone().requires_grad_(True), 'weight_u': u.clone(), 'weight_v': v.clone()}
    xr, br, rr = x.clone().requires_grad_(True), b.clone().requires_grad_(True), res.clone().requires_grad_(True)
    wsn = O.spectral_weight(sd, '', training=True)
    ref = F.conv2d(xr, wsn, br, padding=1) + rr
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    ud, vd = _dev(u.clone(), device), _dev(v.clone(), device)
    wd = _dev(wt, device).requires_grad_(True)
    xd, bd, rd = (_dev(t, device).requires_grad_(True) for t in (x, b, res))
    _sinkify(wd, cache, fin)
    _sinkify(bd, None, fin)
    sig = ops.SpectralState.update(wd, ud, vd, training=True)
    y = ops.conv2d(xd, wd, bd, stride=1, padding=1, res=rd, sn=(sig, ud, vd))
    y.backward(_dev(dy, device))
    if fin is not None:
        assert fin.pending()
        fin.run()
    assert_close('sn u', ud, sd['weight_u'])
    assert_close('sn v', vd, sd['weight_v'])
    assert_close('sn conv y', y, ref)
    assert_close('sn conv dx', xd.grad, xr.grad)
    assert_close('sn conv dw', wd.grad, sd['weight_orig'].grad)
    assert_close('sn conv db', bd.grad, br.grad)
    assert_close('sn conv dres', rd.grad, rr.grad)


def check_layout_cache(device, seed=14):
    """Persistent K-major layouts (layout_cache.py): cached forward / data-gradient operands (stride 1 and 2, padded
    input channels, spectral norm through the epilogue scale) give the reference results, and the single grouped launch
    re-derives every layout after the parameters changed behind autograd's back (what the fused Adam kernel does)."""
    ops, conv = pkg()
    from fsv2v_amd.layout_cache import LayoutCache
    cache = LayoutCache()
    check_conv(device, 2, 6, 9, 7, 5, 3, 2, 1, act='none', cache=cache)        # cpad 2, stride-2 parity classes
    check_conv(device, 1, 20, 9, 9, 32, 4, 2, 2, cache=cache)
    check_conv(device, 1, 8, 7, 7, 16, 4, 1, 2, act='sigmoid', bias=False, cache=cache)
    check_conv(device, 1, 3, 8, 8, 8, 3, 1, 1, act='tanh', cache=cache)          # RGB input: cpad 1
    check_conv(device, 1, 8, 6, 6, 36, 2, 1, 0, act='none', cache=cache)         # 2x2: the tap count without a compiled-in constant
    check_conv_sn_res(device, cache=cache)
    assert len(cache.entries) == 6
    # raw parameter update (no version bump), then ONE grouped refresh
    g = torch.Generator().manual_seed(seed)
    olds = [torch.randn(e.weight.shape, generator=g) * 0.1 for e in cache.entries]
    for e, new in zip(cache.entries, olds):
        e.weight.data.copy_(_dev(new, device))
    for e in cache.entries:               # .data writes do not bump _version: only refresh() can repair the layouts
        assert e.version == e.weight._version
    cache.refresh()
    for e in cache.entries:
        w4 = e.weight.detach()
        for (wt, d, lo, hi) in e.jobs:
            cout, cinp, cin, kh, kw, nt, kpad, ldw, mode = d
            wp = F.pad(w4, (0, 0, 0, 0, 0, cinp - cin))
            khs = [((lo if j < 8 else hi) >> ((j & 7) * 8)) & 15 for j in range(nt)]
            kws = [((lo if j < 8 else hi) >> ((j & 7) * 8 + 4)) & 15 for j in range(nt)]
            if mode & 4:
                import model_checks as _mc
                wp, khs, kws = _mc.masked_tap_sums(wp, khs, kws), list(range(nt)), [0] * nt
            ref, _, _ = conv.prep_weight(wp, mode & 1, None, khs, kws)
            assert torch.equal(ref, wt), d


def check_deferred_wgrad(device, seed=16):
    """grad_finalize.GradFinalizer + csrc/wgrad_finalize.hip: K-major weight gradients queued by backward and folded
    into the (flat-buffer) gradient by one grouped call - plain, channel-padded, 16-tap, spectral-normalised layers, a
    queue holding several layers at once, and one weight used twice in a pass (shared sink -> atomic adds)."""
    ops, conv = pkg()
    from fsv2v_amd.layout_cache import LayoutCache
    from fsv2v_amd.grad_finalize import GradFinalizer
    cache, fin = LayoutCache(), GradFinalizer()
    check_conv(device, 2, 6, 9, 7, 5, 3, 2, 1, act='none', cache=cache, fin=fin)
    check_conv(device, 1, 20, 9, 9, 40, 4, 2, 2, cache=cache, fin=fin)
    check_conv(device, 1, 64, 6, 6, 130, 1, 1, 0, cache=cache, fin=fin)
    check_conv(device, 1, 3, 8, 8, 8, 3, 1, 1, act='tanh', cache=cache, fin=fin)
    check_conv(device, 1, 8, 6, 6, 36, 2, 1, 0, act='none', cache=cache, fin=fin)
    check_conv_sn_res(device, cache=cache, fin=fin)
    # several jobs in one queue, one weight used twice, spectral + plain mixed; K-major results in the per-pass arena
    fin.begin_pass()
    assert fin.arena is not None and fin.arena.numel() > 0
    g = torch.Generator().manual_seed(seed)
    x1, x2 = torch.randn(2, 8, 7, 6, generator=g), torch.randn(2, 8, 7, 6, generator=g)
    w1 = torch.randn(12, 8, 3, 3, generator=g) * 0.2
    w2 = torch.randn(5, 12, 3, 3, generator=g) * 0.2
    u = F.normalize(torch.randn(5, generator=g), dim=0)
    v = F.normalize(torch.randn(12 * 9, generator=g), dim=0)
    sd = {'weight_orig': w2.clone().requires_grad_(True), 'weight_u': u.clone(), 'weight_v': v.clone()}
    w1r = w1.clone().requires_grad_(True)
    b1 = torch.randn(12, generator=g)
    b1r = b1.clone().requires_grad_(True)
    ref = F.conv2d(F.conv2d(x1, w1r, b1r, padding=1) + F.conv2d(x2, w1r, b1r, padding=1),
                   O.spectral_weight(sd, '', training=True), padding=1)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    w1d, w2d = _dev(w1, device).requires_grad_(True), _dev(w2, device).requires_grad_(True)
    b1d = _dev(b1, device).requires_grad_(True)
    _sinkify(w1d, cache, fin); _sinkify(w2d, cache, fin); _sinkify(b1d, None, fin)
    ud, vd = _dev(u.clone(), device), _dev(v.clone(), device)
    sig = ops.SpectralState.update(w2d, ud, vd, training=True)
    y = ops.conv2d(ops.conv2d(_dev(x1, device), w1d, b1d, padding=1) + ops.conv2d(_dev(x2, device), w1d, b1d, padding=1),
                   w2d, None, padding=1, sn=(sig, ud, vd))
    y.backward(_dev(dy, device))
    assert len(fin.jobs) == 3 and fin._arena_off > 0 and len(fin.bias_jobs) == 2     # one bias used twice: shared sink
    assert all(j[1].data_ptr() >= fin.arena.data_ptr() for j in fin.jobs)
    fin.run()
    assert_close('deferred y', y, ref)
    assert_close('deferred shared dw', w1d.grad, w1r.grad)
    assert_close('deferred shared db', b1d.grad, b1r.grad)
    assert_close('deferred sn dw', w2d.grad, sd['weight_orig'].grad)


def check_spectral_power_iteration(device, shapes=((40, 300), (130, 70), (512, 4608)), repeats=1, seed=21):
    """One power iteration of torch.nn.utils.spectral_norm (v = normalize(W^T u), u = normalize(W v), sigma = u.W v): the single-layer
    entry and the grouped one (ops.SpectralGroup) against the same arithmetic in torch, and - `repeats` > 1, on hardware - the SAME
    bits on every run from the same state: W^T u is summed over 64-row slabs in a fixed order (the slabs used to arrive through
    fp32 atomics, which made sigma - and with it every normalised weight of the network - differ in the last bits between two
    runs of the same step)."""
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)

    class _L:
        pass
    layers, refs = [], []
    for (r, c) in shapes:
        w = torch.randn(r, c, generator=g) * 0.1
        u = F.normalize(torch.randn(r, generator=g), dim=0)
        v = F.normalize(torch.randn(c, generator=g), dim=0)
        v_ref = F.normalize(torch.mv(w.double().t(), u.double()), dim=0, eps=1e-12)
        u_ref = F.normalize(torch.mv(w.double(), v_ref), dim=0, eps=1e-12)
        refs.append((float(torch.dot(u_ref, torch.mv(w.double(), v_ref))), u_ref.float(), v_ref.float()))
        l = _L()
        l.weight_orig, l.weight_u, l.weight_v = _dev(w, device), _dev(u.clone(), device), _dev(v.clone(), device)
        l.u0, l.v0 = u, v
        layers.append(l)
    first = None
    for it in range(repeats):
        outs = []
        for l in layers:                                  # single-layer entry, from the initial state
            ud, vd = _dev(l.u0.clone(), device), _dev(l.v0.clone(), device)
            sig = ops.SpectralState.update(l.weight_orig, ud, vd, training=True)
            outs.append((sig.cpu().clone(), ud.cpu().clone(), vd.cpu().clone()))
        for l in layers:
            l.weight_u.copy_(_dev(l.u0, device)); l.weight_v.copy_(_dev(l.v0, device))
        grp = ops.SpectralGroup(layers)
        grp.update(training=True)
        for l, (sg, uu, vv), (s_ref, u_ref, v_ref) in zip(layers, outs, refs):
            sb, ub, vb = l._sig_cached
            assert abs(float(sg[0]) - s_ref) <= 2e-5 * abs(s_ref), (float(sg[0]), s_ref)
            assert_close('sn u', uu, u_ref, tol=2e-5)
            assert_close('sn v', vv, v_ref, tol=2e-5)
            assert torch.equal(sb.cpu(), sg) and torch.equal(ub.cpu(), uu) and torch.equal(vb.cpu(), vv), 'grouped != single-layer'
            assert torch.equal(l.weight_u.cpu(), uu) and torch.equal(l.weight_v.cpu(), vv)
        if first is None:
            first = outs
        else:
            for a, b in zip(first, outs):
                assert all(torch.equal(x, y) for x, y in zip(a, b)), 'power iteration differs between two runs from the same state'


def check_cat_and_pad(device, seed=33):
    """ops.cat_channels (forward + gradient slices) and ops.pad_channels_nhwc against torch, bit for bit, over the layouts the step
    produces: channels-last sources with channel counts that are / are not multiples of four (float4 and element forms of
    csrc/elementwise.hip), NCHW sources, channel slices of a wider tensor, an odd pixel count."""
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    cases = [((2, 8, 5, 7), (2, 12, 5, 7), True), ((2, 8, 5, 7), (2, 3, 5, 7), True), ((1, 16, 4, 6), (1, 4, 4, 6), False),
             ((3, 5, 3, 3), (3, 6, 3, 3), False)]
    for sa, sb, nhwc in cases:
        a, b = torch.randn(sa, generator=g), torch.randn(sb, generator=g)
        wide = torch.randn(sa[0], sa[1] + 8, sa[2], sa[3], generator=g)
        ad, bd, wd = _dev(a, device), _dev(b, device), _dev(wide, device)
        if nhwc:
            ad, bd, wd = conv.to_nhwc(ad), conv.to_nhwc(bd), conv.to_nhwc(wd)
        ad.requires_grad_(True); bd.requires_grad_(True)
        sl = wd[:, 4:4 + sa[1]]                       # a channel slice: strided source (offset multiple of four)
        y = ops.cat_channels([ad, bd, sl])
        ref = torch.cat([a, b, wide[:, 4:4 + sa[1]]], dim=1)
        assert torch.equal(y.detach().cpu(), ref), ('cat', sa, sb, nhwc)
        dy = torch.randn(ref.shape, generator=g)
        y.backward(_dev(dy, device))
        assert torch.equal(ad.grad.cpu(), dy[:, :sa[1]]) and torch.equal(bd.grad.cpu(), dy[:, sa[1]:sa[1] + sb[1]]), ('cat grad', sa, sb)
    for shape, cpad, nhwc in (((2, 3, 6, 5), 1, False), ((2, 6, 6, 5), 2, False), ((1, 10, 4, 4), 2, False), ((2, 15, 3, 5), 1, False),
                              ((2, 6, 6, 5), 2, True), ((1, 17, 4, 4), 3, False)):
        x = torch.randn(shape, generator=g)
        xd = conv.to_nhwc(_dev(x, device)) if nhwc else _dev(x, device)
        y = ops.pad_channels_nhwc(xd, cpad)
        ref = F.pad(x, (0, 0, 0, 0, 0, cpad))
        assert y.shape == ref.shape and torch.equal(y.cpu(), ref), ('pad', shape, cpad, nhwc)
        assert y.permute(0, 2, 3, 1).is_contiguous()
    # wide paddings (one-hot labels 35 -> 40, a packed discriminator input 76 -> 80) and the half-output form of the `--amp` path
    for shape, cpad in (((2, 35, 6, 5), 5), ((1, 76, 4, 7), 4), ((2, 3, 6, 5), 5), ((1, 11, 3, 9), 5), ((1, 20, 5, 5), 4)):
        x = torch.randn(shape, generator=g)
        ref = F.pad(x, (0, 0, 0, 0, 0, cpad))
        y = ops.pad_channels_nhwc(_dev(x, device), cpad)
        assert torch.equal(y.cpu(), ref) and y.permute(0, 2, 3, 1).is_contiguous(), ('pad wide', shape, cpad)
        yh = ops.pad_channels_nhwc(_dev(x, device), cpad, half=True)
        assert yh.dtype == torch.float16 and yh.permute(0, 2, 3, 1).is_contiguous(), ('pad half', shape, cpad)
        assert torch.equal(yh.cpu(), ref.to(torch.float16)), ('pad half values', shape, cpad)


def check_linear(device, r=40, cin=16, cout=50, seed=2):
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(r, cin, generator=g)
    w = torch.randn(cout, cin, generator=g) * 0.3
    b = torch.randn(cout, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = O.actvn(F.linear(xr, wr, br))
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xd, wd, bd = (_dev(t, device).requires_grad_(True) for t in (x, w, b))
    y = ops.linear(xd, wd, bd, act=conv.ACT_LRELU)
    y.backward(_dev(dy, device))
    assert_close('linear y', y, ref)
    assert_close('linear dx', xd.grad, xr.grad)
    assert_close('linear dw', wd.grad, wr.grad)
    assert_close('linear db', bd.grad, br.grad)


def check_batch_conv(device, b=2, cin=8, cout=12, h=5, w=6, seed=3):
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(b, cout, cin, 1, 1, generator=g) * 0.3
    bias = torch.randn(b, cout, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, wt, bias))
    ref = O.actvn(O.batch_conv(xr, wr, br))
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xd, wd, bd = (_dev(t, device).requires_grad_(True) for t in (x, wt, bias))
    y = ops.batch_conv(xd, wd, bd, act=conv.ACT_LRELU)
    y.backward(_dev(dy, device))
    assert_close('batch_conv y', y, ref)
    assert_close('batch_conv dx', xd.grad, xr.grad)
    assert_close('batch_conv dw', wd.grad, wr.grad)
    assert_close('batch_conv db', bd.grad, br.grad)
    # weights / bias as strided views of one FC output row per sample (read in place, no copies)
    f = torch.randn(b, cout * cin + cout + 5, generator=g) * 0.3
    fr, fd = f.clone().requires_grad_(True), _dev(f, device).requires_grad_(True)

    def parts(t):
        wp, bp, _ = torch.split(t, [cout * cin, cout, 5], dim=1)
        return wp.reshape(b, cout, cin, 1, 1), bp
    ref = O.actvn(O.batch_conv(xr.detach(), *parts(fr)))
    ref.backward(dy)
    wv, bv = parts(fd)
    assert b == 1 or not wv.is_contiguous()
    y = ops.batch_conv(_dev(x, device), wv, bv, act=conv.ACT_LRELU)
    y.backward(_dev(dy, device))
    assert_close('batch_conv strided y', y, ref)
    assert_close('batch_conv strided df', fd.grad, fr.grad)


def check_norm(device, instance, n=3, c=10, h=7, w=5, affine=True, act='lrelu', seed=4):
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, c, h, w, generator=g) * 2 + 0.5
    wt = torch.randn(c, generator=g) if affine else None
    b = torch.randn(c, generator=g) if affine else None
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True) if affine else None
    br = b.clone().requires_grad_(True) if affine else None
    rm, rv = torch.zeros(c), torch.ones(c)
    if instance:
        ref = O.instance_norm(xr, wr, br, eps=0.1)
    else:
        ref = F.batch_norm(xr, rm, rv, wr, br, True, 0.1, 1e-5)
    if act == 'lrelu':
        ref = O.actvn(ref)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xd = _dev(x, device).requires_grad_(True)
    wd = _dev(wt, device).requires_grad_(True) if affine else None
    bd = _dev(b, device).requires_grad_(True) if affine else None
    rmd, rvd = _dev(torch.zeros(c), device), _dev(torch.ones(c), device)
    y = ops.norm_act(xd, wd, bd, None if instance else rmd, None if instance else rvd, instance=instance,
                     eps=0.1 if instance else 1e-5, act=conv.ACT_LRELU if act == 'lrelu' else conv.ACT_NONE)
    y.backward(_dev(dy, device))
    assert_close('norm y', y, ref)
    assert_close('norm dx', xd.grad, xr.grad)
    if affine:
        assert_close('norm dw', wd.grad, wr.grad)
        assert_close('norm db', bd.grad, br.grad)
    if not instance:
        assert_close('running_mean', rmd, rm)
        assert_close('running_var', rvd, rv)


def check_spade(device, nmaps=1, generated=True, n=2, c=12, ch=8, h=6, w=5, act='lrelu', seed=5, strided=False, up=False,
                half_out=False, f16=False):
    """SPADE with per-sample generated weights for map 0 and fixed weights for the extra maps.  strided: the generated
    weights / biases are views into one [n, L] tensor, the way the weight-generating FC hands them over
    (generator.py reshape_weight); c % 16 == 0 takes the single-preparation-launch path of ops._SpadeFn.  up: x is handed
    over at half resolution and the kernels read it through the nearest x2 up-sampling index (generator.py:124 folded in);
    the reference up-samples explicitly (h, w must be even).  half_out: the `--amp` form on the half-precision kernels - h is
    stored as IEEE half (one rounding) and its gradient arrives as half; the arithmetic in between stays fp32, so with the
    gradient rounded beforehand on both sides the gradients agree at the fp32 tolerance (c % 16 == 0 required).  f16 (with
    half_out; ch % 8 == 0): the gamma / beta GEMMs on the f16 matrix instructions too - half maps, half weights, half
    d(gamma|beta), bias gradients from the backward twin's epilogue - against the oracle with the `--amp` arithmetic installed."""
    import contextlib
    from oracle import np_oracle
    ops, conv = pkg()
    assert half_out or not f16
    os.environ['FSV_SPADE_F16'] = '1' if f16 else '0'
    arith = (lambda: O.arithmetic(np_oracle.amp_conv2d)) if f16 else contextlib.nullcontext
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, c, h // 2, w // 2, generator=g) + 0.3 if up else torch.randn(n, c, h, w, generator=g) + 0.3
    maps = [torch.randn(n, ch, h, w, generator=g) for _ in range(nmaps)]
    leaves_ref, leaves_dev = [], []

    def leaf(t):
        a = t.clone().requires_grad_(True)
        d = _dev(t, device).requires_grad_(True)
        leaves_ref.append(a); leaves_dev.append(d)
        return a, d
    xr, xd = leaf(x)
    maps_r, maps_d, fixed_r, weights_d = [], [], [], []
    gen_r = None
    for k in range(nmaps):
        mr, md = leaf(maps[k])
        maps_r.append(mr); maps_d.append(md)
        if k == 0 and generated and strided:
            half = c * ch + c
            f_r, f_d = leaf(torch.randn(n, 2 * half + 3, generator=g) * 0.3)

            def views(f):
                a, b = f[:, :half], f[:, half:2 * half]
                return (a[:, :-c].view(n, c, ch, 1, 1), b[:, :-c].view(n, c, ch, 1, 1), a[:, -c:], b[:, -c:])
            wg_r, wb_r, bg_r, bb_r = views(f_r)
            wg_d, wb_d, bg_d, bb_d = views(f_d)
            gen_r = ((wg_r, bg_r), (wb_r, bb_r))
            fixed_r.append(None)
        elif k == 0 and generated:
            wg_r, wg_d = leaf(torch.randn(n, c, ch, 1, 1, generator=g) * 0.3)
            wb_r, wb_d = leaf(torch.randn(n, c, ch, 1, 1, generator=g) * 0.3)
            bg_r, bg_d = leaf(torch.randn(n, c, generator=g) * 0.3)
            bb_r, bb_d = leaf(torch.randn(n, c, generator=g) * 0.3)
            gen_r = ((wg_r, bg_r), (wb_r, bb_r))
            fixed_r.append(None)
        else:
            wg_r, wg_d = leaf(torch.randn(c, ch, 1, 1, generator=g) * 0.3)
            wb_r, wb_d = leaf(torch.randn(c, ch, 1, 1, generator=g) * 0.3)
            bg_r, bg_d = leaf(torch.randn(c, generator=g) * 0.3)
            bb_r, bb_d = leaf(torch.randn(c, generator=g) * 0.3)
            fixed_r.append((wg_r, bg_r, wb_r, bb_r))
        weights_d.append((wg_d, wb_d, bg_d, bb_d))
    run_mean_r, run_var_r = torch.zeros(c), torch.ones(c)
    x_in = F.interpolate(xr, scale_factor=2, mode='nearest') if up else xr
    with arith():
        ref = O.spade(x_in, maps_r, fixed_r, gen_r)
        F.batch_norm(x_in.detach(), run_mean_r, run_var_r, training=True, momentum=0.1, eps=1e-5)      # running statistics
        if act == 'lrelu':
            ref = O.actvn(ref)
        dy = torch.randn(ref.shape, generator=g)
        if half_out:
            dy = dy.to(torch.float16).to(torch.float32)
        ref.backward(dy)
    run_mean_d, run_var_d = _dev(torch.zeros(c), device), _dev(torch.ones(c), device)
    prev = conv.set_mfma_mode(1 if half_out else conv.mfma_mode())
    from importlib import import_module
    lib = import_module('few-shot-vid2vid_amd.lib')
    seen, real_call = [], lib.call

    def recording_call(name, *a):
        seen.append((name, a))
        return real_call(name, *a)
    lib.call = recording_call
    try:
        y = ops.spade_mod(xd, maps_d, weights_d, run_mean_d, run_var_d, act=conv.ACT_LRELU if act == 'lrelu' else conv.ACT_NONE,
                          up=up)
        if half_out:
            assert y.dtype == torch.float16, 'the modulated tensor should have been stored as half'
            y.backward(_dev(dy, device).to(torch.float16))
            err = (y.detach().float().cpu() - ref.detach()).abs()
            assert bool((err <= 2.0 ** -10 * ref.detach().abs() * 1.01 + 1e-5 * float(ref.detach().abs().max())).all()), float(err.max())
        else:
            y.backward(_dev(dy, device))
            assert_close('spade h', y, ref)
    finally:
        lib.call = real_call
        conv.set_mfma_mode(prev)
        os.environ.pop('FSV_SPADE_F16', None)
    if f16:                    # the launches really were the f16 forms (flags bit 2), with half d(gamma|beta) and fused bias sums
        fw = [a for nm_, a in seen if nm_ == 'fsv_spade_mod_fwd_h']
        bw = [a for nm_, a in seen if nm_ == 'fsv_spade_mod_bwd_h']
        assert fw and bw and all(a[-2] & 4 for a in fw) and all((a[-6] & 6) == 6 and a[-5] is not None for a in bw), \
            [nm_ for nm_, _ in seen]
        assert not any(nm_ == 'fsv_colsum_fused' for nm_, _ in seen)
    assert_close('spade running mean', run_mean_d, run_mean_r, 1e-5)
    assert_close('spade running var', run_var_d, run_var_r, 1e-5)
    for i, (a, d) in enumerate(zip(leaves_ref, leaves_dev)):
        assert_close('spade grad %d' % i, d.grad, a.grad)


def check_upsample(device, n=2, c=6, h=3, w=5, seed=6):
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, c, h, w, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = F.interpolate(xr, scale_factor=2)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xd = _dev(x, device).requires_grad_(True)
    y = ops.upsample2x(xd)
    y.backward(_dev(dy, device))
    assert_close('up y', y, ref, 1e-7)
    assert_close('up dx', xd.grad, xr.grad, 1e-6)


def check_warp(device, b=2, c=3, h=16, w=24, mag=6.0, seed=7, zero_flow=False):
    """values/gradients within tolerance; integer tap indices bit-exact against ATen's selection."""
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(b, c, h, w, generator=g)
    if zero_flow:
        flow = torch.zeros(b, 2, h, w)
        flow[1:] = torch.randint(-3, 4, (b - 1, 2, h, w), generator=g).float()
    else:
        flow = (torch.rand(b, 2, h, w, generator=g) - 0.5) * 2 * mag
    ir, fr = img.clone().requires_grad_(True), flow.clone().requires_grad_(True)
    ref = O.resample(ir, fr)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    idd, fd = _dev(img, device).requires_grad_(True), _dev(flow, device).requires_grad_(True)
    y = ops.resample(idd, fd)
    y.backward(_dev(dy, device))
    assert_close('warp y', y, ref, 1e-5)
    assert_close('warp dimg', idd.grad, ir.grad, 1e-5)
    assert_close('warp dflow', fd.grad, fr.grad, 1e-4)
    _, taps = ops.warp_taps(_dev(img, device), _dev(flow, device))
    ref_taps = O.resample_taps(flow)
    assert torch.equal(taps.cpu(), ref_taps), 'warp tap indices differ from the oracle'
    # independent confirmation through ATen itself: warp an image whose value is its own x (resp. y) coordinate;
    # with integer-valued taps the east/south neighbours differ by exactly 1, so floor(out) reveals x_w / y_n.
    return taps


def check_warp_index_image(device, h=32, w=512, seed=8):
    """tap indices revealed by ATen's own grid_sample (one-hot column image), compared bit for bit."""
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    flow = (torch.rand(1, 2, h, w, generator=g) - 0.5) * 40.0
    flow[:, :, : h // 4] = 0.0                      # zero-flow rows: the fp32 round trip must be reproduced
    flow[:, :, h // 4: h // 2] = torch.randint(-5, 6, (1, 2, h // 2 - h // 4, w), generator=g).float()
    _, taps = ops.warp_taps(_dev(torch.zeros(1, 1, h, w), device), _dev(flow, device))
    taps = taps.cpu()
    # ATen evidence: gradient of grid_sample w.r.t. the image scatters onto exactly the taps it selected
    img = torch.zeros(1, 1, h, w, requires_grad=True)
    out = O.resample(img, flow)
    for (yy, xx) in [(0, 0), (h // 8, w // 3), (h // 3, w - 1), (h - 1, w // 2), (h // 2 + 3, 7), (h // 8, 100)]:
        gsel = torch.zeros_like(out)
        gsel[0, 0, yy, xx] = 1.0
        (gi,) = torch.autograd.grad(out, img, gsel, retain_graph=True)
        nz = gi[0, 0].nonzero()
        x0, y0 = int(taps[0, yy, xx, 0]), int(taps[0, yy, xx, 1])
        for (ty, tx) in nz.tolist():
            assert ty in (y0, y0 + 1) and tx in (x0, x0 + 1), 'ATen touched tap (%d,%d), kernel chose (%d,%d)' % (ty, tx, y0, x0)
        assert gi[0, 0, y0, x0] > 0 or nz.numel() > 0
    ref_taps = O.resample_taps(flow)
    assert torch.equal(taps, ref_taps)


def check_softmax(device, n=2, c=70, h=3, w=5, seed=12):
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, c, h, w, generator=g) * 3
    xr = x.clone().requires_grad_(True)
    ref = torch.softmax(xr, dim=1)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xd = _dev(x, device).requires_grad_(True)
    y = ops.softmax_channels(xd)
    y.backward(_dev(dy, device))
    assert_close('softmax y', y, ref, 1e-5)
    assert_close('softmax dx', xd.grad, xr.grad, 1e-4)


def check_losses(device, seed=13):
    """L1 / masked L1 (all three inputs differentiable, NCHW and NHWC operands), hinge, D-input packing, 15x15 pooling."""
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    n, c, h, w = 2, 5, 9, 7
    a, b = torch.randn(n, c, h, w, generator=g), torch.randn(n, c, h, w, generator=g)
    m = torch.rand(n, 1, h, w, generator=g)
    for (use_b, use_m, nhwc) in [(True, False, False), (True, True, False), (False, True, True), (True, True, True)]:
        ar, br, mr = (t.detach().clone().requires_grad_(True) for t in (a, b, m))
        tgt = br if use_b else torch.full_like(ar, 0.7)
        mm = mr.expand_as(ar) if use_m else 1.0
        ref = F.l1_loss(ar * mm, tgt * mm)
        ref.backward()
        ad = _dev(a, device)
        if nhwc:
            ad = conv.to_nhwc(ad)
        ad = ad.detach().requires_grad_(True)
        bd, md = (_dev(t, device).detach().clone().requires_grad_(True) for t in (b, m))
        out = ops.l1_loss(ad, bd if use_b else 0.7, md if use_m else None)
        out.sum().backward()
        assert_close('l1', out.view(()), ref, 1e-5)
        assert_close('l1 da', ad.grad, ar.grad, 1e-5)
        if use_b:
            assert_close('l1 db', bd.grad, br.grad, 1e-5)
        if use_m:
            assert_close('l1 dm', md.grad, mr.grad, 1e-4)
    x = torch.randn(3, 1, 6, 5, generator=g) * 2
    for real in (True, False):
        xr = x.detach().clone().requires_grad_(True)
        ref = -torch.mean(torch.min((xr if real else -xr) - 1, xr * 0))
        ref.backward()
        xd = _dev(x, device).detach().clone().requires_grad_(True)
        out = ops.hinge_loss(xd, real)
        out.sum().backward()
        assert_close('hinge', out.view(()), ref, 1e-5)
        assert_close('hinge dx', xd.grad, xr.grad, 1e-5)
    # D input packing
    bsz, hh, ww = 2, 6, 5
    ref_c, lab, fake, real = (torch.randn(bsz, k, hh, ww, generator=g) for k in (4, 3, 3, 3))
    fr = fake.detach().clone().requires_grad_(True)
    xref = torch.cat([ref_c.repeat(2, 1, 1, 1), torch.cat([lab.repeat(2, 1, 1, 1), torch.cat([fr, real], 0)], 1)], 1)
    wgt = torch.randn(xref.shape, generator=g)
    (xref * wgt).sum().backward()
    fd = _dev(fake, device).detach().clone().requires_grad_(True)
    xo = ops.pack_d_input(_dev(ref_c, device), conv.to_nhwc(_dev(lab, device)), fd, _dev(real, device))
    (xo * _dev(wgt, device)).sum().backward()
    assert_close('pack', xo, xref, 1e-7)
    assert_close('pack dfake', fd.grad, fr.grad, 1e-6)
    # pooling
    lab1 = torch.randn(2, 1, 20, 17, generator=g)
    assert_close('pool max', ops.pool15(_dev(lab1, device), 'max_gt', 0.5), (F.max_pool2d(lab1, 15, 1, 7) > 0.5).float(), 1e-7)
    assert_close('pool avg', ops.pool15(_dev(lab1, device), 'avg'), F.avg_pool2d(lab1, 15, 1, 7), 1e-5)
    lab2 = torch.randn(1, 3, 70, 45, generator=g)[:, 1:2]          # several 32x32 tiles, ragged edges, strided input
    assert_close('pool max tiles', ops.pool15(_dev(lab2, device), 'max_gt', -0.2), (F.max_pool2d(lab2, 15, 1, 7) > -0.2).float(), 1e-7)
    assert_close('pool avg tiles', ops.pool15(_dev(lab2, device), 'avg'), F.avg_pool2d(lab2, 15, 1, 7), 1e-5)


def check_part_masks(device, seed=15):
    """DensePose part-group masks in one launch == the reference's 25 x (gt, lt, and, or) chain, bit for bit
    (models/input_process.py:64-94), on part ids stored as (p / 24) * 2 - 1 plus off-grid values."""
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    b, t, c, h, w = 2, 2, 6, 9, 13
    ids = torch.randint(0, 25, (b, t, h, w), generator=g).float()
    pose = ids / 24 * 2 - 1
    noise = (torch.rand((b, t, h, w), generator=g) < 0.2).float() * (torch.rand((b, t, h, w), generator=g) - 0.5) * 0.02
    label = torch.randn(b, t, c, h, w, generator=g)
    label[:, :, 2] = pose + noise
    ch = label[:, :, 2]
    part = (ch / 2 + 0.5) * 24
    groups = [[0], [1, 2], [3, 4], [5, 6], [7, 9, 8, 10], [11, 13, 12, 14], [15, 17, 16, 18], [19, 21, 20, 22], [23, 24]]
    ref = []
    for grp in groups:
        m = torch.zeros_like(part, dtype=torch.bool)
        for j in grp:
            m |= (part > j - 0.1) & (part < j + 0.1)
        ref.append(m)
    ref = torch.stack(ref, dim=2).float()
    got = ops.part_masks(_dev(label, device)[:, :, 2], 0, 9).cpu()
    assert torch.equal(got, ref)
    face = ops.part_masks(_dev(label, device)[:, 0:1, 2], 8, 1).cpu()
    assert torch.equal(face[:, :, 0], ref[:, 0:1, 8])
    assert 0 < float(ref.mean()) < 1


def check_face_ops(device, seed=17):
    """csrc/face.hip: device-side face boxes == the reference's get_face_region integers for DensePose / OpenPose / empty
    label maps, and the one-launch crop + nearest resize (and its gradient) == the per-sample slicing + F.interpolate."""
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    n, c, h, w, size = 5, 6, 64, 48, 16
    label = torch.rand(n, c, h, w, generator=g) * 2 - 1
    label[:, 2] = -1.0
    label[:, 3:] = -1.0
    label[0, 2, 10:31, 20:29] = 0.95                 # tall DensePose face
    label[1, 2, 60:64, 0:3] = 1.0                    # at the border -> clamped centre
    label[1, 3:, 5:12, 30:44] = 0.5                  # OpenPose key-points present
    label[2, 2, 3, 3] = 0.92                         # single pixel -> minimum size 32
    label[3, 2, 0:64, 0:48] = 0.91                   # everything -> capped at the width
    # sample 4: no face at all -> default box
    image = torch.randn(n, 5, h, w, generator=g)
    for remove_face_labels in (True, False):
        cfg = O.Cfg(fineSize=size * 4, aspect_ratio=1.0, remove_face_labels=remove_face_labels)
        for crop_smaller in (0, 4):
            want = [O.get_face_region(cfg, label[i:i + 1], crop_smaller) for i in range(n)]
            got = ops.face_boxes(_dev(label, device), use_openpose=not remove_face_labels, crop_smaller=crop_smaller).cpu()
            assert got.tolist() == [list(b) for b in want], (remove_face_labels, crop_smaller, got.tolist(), want)
        img_r = image.clone().requires_grad_(True)
        ref = O.crop_face_region(cfg, img_r, label)
        dy = torch.randn(ref.shape, generator=g)
        ref.backward(dy)
        for nhwc in (False, True):
            img_d = _dev(image, device)
            if nhwc:
                img_d = conv.to_nhwc(img_d)
            img_d = img_d.detach().requires_grad_(True)
            boxes = ops.face_boxes(_dev(label, device), use_openpose=not remove_face_labels)
            out = ops.crop_face(img_d, boxes, size)
            out.backward(_dev(dy, device))
            assert torch.equal(out.detach().cpu(), ref.detach())
            assert_close('crop face grad', img_d.grad, img_r.grad, 1e-6)
        # --refine_face: paste the (coarse + refined) face back (replace_face_region, crop_smaller 4)
        fake = torch.randn(n, 3, h, w, generator=g).clamp(-1.2, 1.2)
        face = torch.randn(n, 3, size, size, generator=g) * 0.8
        fr, cr = fake.clone().requires_grad_(True), face.clone().requires_grad_(True)
        ref = O.replace_face_region(cfg, fr, cr, label, None, crop_smaller=4)
        dy = torch.randn(ref.shape, generator=g)
        ref.backward(dy)
        fd, cd = _dev(fake, device).requires_grad_(True), _dev(face, device).requires_grad_(True)
        boxes = ops.face_boxes(_dev(label, device), use_openpose=not remove_face_labels, crop_smaller=4)
        out = ops.paste_face(fd, cd, boxes)
        out.backward(_dev(dy, device))
        assert_close('paste face', out, ref, 1e-5)       # (ATen evaluates the same bilinear weights in another order)
        assert_close('paste face dimg', fd.grad, fr.grad, 1e-5)
        assert_close('paste face dface', cd.grad, cr.grad, 1e-5)


def check_flownet_ops(device, seed=18, checker=None):
    """FlowNet2's native operators (csrc/flownet_ops.hip) vs `checker`: the reference's own CUDA kernels compiled for the host
    (oracle/flownet_ref.py, built by oracle/build_ref.py) when given, else their Python restatement (oracle/flownet_oracle.py,
    itself held to that build by tests/test_flownet_ref.py): cost volume 1e-5 (summation order), warp and channel norm
    bit-exact."""
    from importlib import import_module
    import fsv2v_amd  # noqa: F401
    fo = import_module('few-shot-vid2vid_amd.flownet_ops')
    from oracle import flownet_oracle as FO
    if checker is not None:
        FO = checker
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        # FlowNetC's configuration (FlowNetC.py:28-31) on a small map, plus odd sizes / strides
        for (n, c, h, w, pad, md, s1, s2) in [(2, 70, 9, 37, 20, 20, 1, 2), (1, 64, 12, 40, 4, 4, 1, 1),
                                               (1, 5, 11, 13, 6, 6, 2, 3)]:
            f1, f2 = torch.randn(n, c, h, w, generator=g), torch.randn(n, c, h, w, generator=g)
            ref = FO.correlation(f1, f2, pad, 1, md, s1, s2)
            got = fo.correlation(_dev(f1, device), _dev(f2, device), pad, 1, md, s1, s2)
            assert tuple(got.shape) == tuple(ref.shape), (got.shape, ref.shape)
            assert_close('correlation', got, ref, 1e-5)
        for (n, c, h, w, mag) in [(2, 3, 17, 23, 6.0), (1, 2, 8, 64, 80.0)]:
            img = torch.randn(n, c, h, w, generator=g)
            flow = (torch.rand(n, 2, h, w, generator=g) - 0.5) * mag
            flow[:, :, 0] = torch.round(flow[:, :, 0])               # integer displacements: alpha = beta = 0
            ref = FO.resample2d(img, flow)
            got = fo.resample2d(_dev(img, device), _dev(flow, device)).cpu()
            assert torch.equal(got, ref), float((got - ref).abs().max())
            nhwc = fo.resample2d(conv_nhwc(_dev(img, device)), _dev(flow, device)).cpu()
            assert torch.equal(nhwc, ref)
        # bilinear resize (align_corners = False): x4 up-sampling of a flow field, resize to a multiple of 64 and back
        for (n, c, h, w, size, sf) in [(2, 2, 8, 12, None, 4), (1, 3, 70, 90, (64, 64), None), (1, 2, 64, 64, (70, 90), None),
                                        (1, 1, 5, 7, (11, 3), None)]:
            t = torch.randn(n, c, h, w, generator=g)
            ref = F.interpolate(t, size=size, scale_factor=sf, mode='bilinear')
            got = fo.bilinear_resize(_dev(t, device), size=size, scale_factor=sf).cpu()
            assert got.shape == ref.shape and float((got - ref).abs().max()) <= 1e-5 * max(float(ref.abs().max()), 1.0), \
                ('bilinear resize', (n, c, h, w), size, sf, float((got - ref).abs().max()))
        x = torch.randn(2, 7, 5, 9, generator=g) * 3
        assert torch.equal(fo.channelnorm(_dev(x, device)).cpu(), FO.channelnorm(x))
        assert torch.equal(fo.channelnorm(conv_nhwc(_dev(x, device))).cpu(), FO.channelnorm(x))


def conv_nhwc(t):
    return pkg()[1].to_nhwc(t)


def check_adam(device, n=1000, seed=9):
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    p0 = torch.randn(n, generator=g)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=2e-4, betas=(0.0, 0.999))
    pd = _dev(p0.clone(), device)
    m, v = torch.zeros_like(pd), torch.zeros_like(pd)
    state = _dev(torch.tensor([0.0, 0.0, 0.0, 2e-4]), device)
    for it in range(3):
        grad = torch.randn(n, generator=g)
        pr.grad = grad.clone()
        opt.step()
        ops.adam_step(pd, _dev(grad, device), m, v, state, 0.0, 0.999, 1e-8)
    assert_close('adam', pd, pr, 1e-5)


def run_all(device, big=False):
    check_conv(device, 1, 8, 8, 8, 8, 3, 1, 1)
    check_conv(device, 2, 4, 9, 7, 5, 3, 2, 1, act='none')
    check_conv(device, 1, 6, 10, 10, 12, 3, 1, 1, act='tanh')
    check_conv(device, 1, 20, 9, 9, 32, 4, 2, 2)
    check_conv(device, 1, 8, 7, 7, 16, 4, 1, 2, act='sigmoid', bias=False)
    check_conv(device, 1, 64, 6, 6, 130, 1, 1, 0)
    check_conv_sn_res(device)
    check_layout_cache(device)
    check_deferred_wgrad(device)
    check_linear(device)
    check_batch_conv(device)
    check_norm(device, instance=False)
    check_norm(device, instance=True)
    check_norm(device, instance=False, affine=False, act='none')
    check_spade(device, nmaps=1, generated=True)
    check_spade(device, nmaps=3, generated=True, act='none')
    check_spade(device, nmaps=2, generated=False, c=40, ch=12)
    check_upsample(device)
    check_warp(device)
    check_warp(device, zero_flow=True)
    check_warp_index_image(device)
    check_adam(device)
    check_part_masks(device)
    check_face_ops(device)
    check_flownet_ops(device)


def check_avgpool3s2(device, seed=90):
    """MultiscaleDiscriminator's pyramid step: AvgPool2d(3, stride 2, padding 1, count_include_pad=False), odd and even sizes"""
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    for shp in [(2, 5, 9, 7), (1, 20, 64, 64), (2, 3, 8, 10), (1, 4, 1, 5)]:
        x = torch.randn(*shp, generator=g)
        xr = x.clone().requires_grad_(True)
        ref = F.avg_pool2d(xr, 3, stride=2, padding=1, count_include_pad=False)
        dy = torch.randn(ref.shape, generator=g)
        ref.backward(dy)
        xd = _dev(x, device).requires_grad_(True)
        y = ops.avgpool3s2(xd)
        y.backward(_dev(dy, device))
        assert_close('avgpool y %s' % (shp,), y, ref, 1e-6)
        assert_close('avgpool dx %s' % (shp,), xd.grad, xr.grad, 1e-6)


def check_ordered_split(device, seed=93):
    """split-K launches of the fp32 gather-GEMM sum their splits in ascending order (one output copy per split + a finishing pass,
    the explicit split_ws argument of include/fsv2v.h fsv_conv_gather_fwd): the same bits on every run, equal to the atomic form up to summation order,
    epilogue (bias, LeakyReLU, residual) applied once by the finishing pass"""
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    n, cin, h, w, cout = 2, 256, 6, 5, 96
    x = conv.to_nhwc(_dev(torch.randn(n, cin, h, w, generator=g), device))
    wt = _dev(torch.randn(cout, cin, 3, 3, generator=g) * 0.05, device)
    b = _dev(torch.randn(cout, generator=g), device)
    res = conv.to_nhwc(_dev(torch.randn(n, cout, h, w, generator=g), device))
    geo = conv.Geom(3, 3, 1, 1)
    wf, _, ldw = conv.prep_weight(wt, 0, geo)

    def run(split):
        return conv.conv_forward(x, wf, ldw, cout, geo, bias=b, res=res, act=conv.ACT_NONE, force_split=split).clone()
    ref = F.conv2d(x.cpu(), wt.cpu(), b.cpu(), padding=1) + res.cpu()
    outs = [run(4) for _ in range(4)]
    for o in outs[1:]:
        assert bool((o == outs[0]).all()), 'ordered split-K is not bit-reproducible'
    assert_close('ordered split', outs[0], ref, 1e-5)
    os.environ['FSV_ORDERED_SPLIT'] = '0'
    try:
        atomic = run(4)
    finally:
        os.environ.pop('FSV_ORDERED_SPLIT', None)
    assert_close('atomic split', atomic, ref, 1e-5)
    # round 6: the finishing pass handles four channels per work-item where Cout % 4 == 0 - the same sums in the same order as the
    # scalar pass (bit for bit); Cout % 4 != 0 stays on the scalar pass
    os.environ['FSV_SPLIT_FIN4'] = '0'
    try:
        scalar = run(4)
    finally:
        os.environ.pop('FSV_SPLIT_FIN4', None)
    assert bool((scalar == outs[0]).all()), 'vectorised finishing pass changed the bits'
    cout2 = 94
    wt2 = _dev(torch.randn(cout2, cin, 3, 3, generator=g) * 0.05, device)
    b2 = _dev(torch.randn(cout2, generator=g), device)
    wf2, _, ldw2 = conv.prep_weight(wt2, 0, geo)
    y2 = conv.conv_forward(x, wf2, ldw2, cout2, geo, bias=b2, act=conv.ACT_LRELU, force_split=3)
    assert_close('ordered split, Cout % 4 != 0, LeakyReLU', y2, O.actvn(F.conv2d(x.cpu(), wt2.cpu(), b2.cpu(), padding=1)), 1e-5)


def check_adaptive_avgpool(device, seed=92):
    """nn.AdaptiveAvgPool2d (AdaptiveDiscriminator.gen_conv_weights): windows that do not divide (33 -> 8), that do (32 -> 8),
    identity, a rectangular case, and maps that GROW (discriminator.py:146,153 pool to fineSize / 8 whatever the encoded map's
    size: 33 -> 64 at the second scale with adaptive_D_layers = 3, a non-integer ratio, and growth on one axis only)"""
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    for shp, out in [((2, 8, 33, 33), (8, 8)), ((1, 16, 32, 32), (8, 8)), ((2, 4, 8, 8), (8, 8)), ((1, 5, 17, 33), (4, 8)),
                     ((1, 3, 9, 5), (1, 1)), ((1, 4, 33, 33), (64, 64)), ((2, 3, 5, 7), (8, 16)), ((1, 2, 6, 20), (9, 4))]:
        x = torch.randn(*shp, generator=g)
        xr = x.clone().requires_grad_(True)
        ref = F.adaptive_avg_pool2d(xr, out)
        dy = torch.randn(ref.shape, generator=g)
        ref.backward(dy)
        xd = _dev(x, device).requires_grad_(True)
        y = ops.adaptive_avgpool(xd, *out)
        y.backward(_dev(dy, device))
        assert_close('adaptive avgpool y %s' % (shp,), y, ref, 1e-6)
        assert_close('adaptive avgpool dx %s' % (shp,), xd.grad, xr.grad, 1e-6)


def check_fused_reductions(device, shapes=((1, 4096, 32), (2, 1000, 7), (1, 300, 260), (4, 64, 512), (1, 70000, 64)),
                           repeats=3, seed=91):
    """One-launch statistics / column sums (last-workgroup second stage, csrc/norm.hip) against fp64 torch, repeatedly on the
    same buffers (a stale ticket or a partial read too early would show as a changing result), and the ticket pool is left
    zeroed."""
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    for (G, P, C) in shapes:
        x = torch.randn(G, P, C, generator=g) * 1.5 + 0.25
        x64 = x.double()
        mean_ref = x64.mean(dim=1)
        var_ref = x64.var(dim=1, unbiased=False)
        rstd_ref = 1.0 / torch.sqrt(var_ref + 1e-5)
        col_ref = x64.sum(dim=1)
        xd = _dev(x, device)
        for it in range(repeats):
            rm, rv = _dev(torch.zeros(C), device), _dev(torch.ones(C), device)
            mean, rstd = ops.norm_stats(xd, G, P, C, 1e-5, rm if G == 1 else None, rv if G == 1 else None, 0.1)
            assert_close('fused mean %s #%d' % ((G, P, C), it), mean.view(G, C), mean_ref.float(), tol=1e-5)
            assert_close('fused rstd %s #%d' % ((G, P, C), it), rstd.view(G, C), rstd_ref.float(), tol=1e-5)
            if G == 1:
                unb = var_ref[0] * (P / (P - 1.0))
                assert_close('fused running_mean', rm, (0.1 * mean_ref[0]).float(), tol=1e-5)
                assert_close('fused running_var', rv, (0.9 + 0.1 * unb).float(), tol=1e-5)
            cs = ops.colsum(xd, G, P, C)
            assert_close('fused colsum %s #%d' % ((G, P, C), it), cs, col_ref.float(), tol=1e-5)
            acc = _dev(torch.ones(G, C), device)
            ops.colsum(xd, G, P, C, out=acc)
            assert_close('fused colsum accumulate', acc, (col_ref + 1.0).float(), tol=1e-5)
    pool = conv._tickets[xd.device][0]
    assert int(pool.abs().sum()) == 0, "a fused reduction left its ticket range dirty"


def check_warp_compose(device, b=2, h=16, w=24, mag=5.0, seed=93):
    """Fused warp + compositing (ops.warp_concat / ops.warp_blend) against the oracle's op chain
    (generator.py:214-227, :441-443): resample -> cat([warp, mask]) resp. raw * mask + warp * (1 - mask), with gradients
    arriving through the warped image AND the composite (the losses read the former, the decoder the latter); forward values
    bit-equal to the separate kernels of this library."""
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    img5 = torch.randn(b, 5, h, w, generator=g)           # a channel slice is what the generator passes in
    flow = (torch.rand(b, 2, h, w, generator=g) - 0.5) * 2 * mag
    mask = torch.rand(b, 1, h, w, generator=g)
    raw = torch.randn(b, 3, h, w, generator=g)
    for mode in ('concat', 'blend'):
        ir, fr, mr, rr = [t.clone().requires_grad_(True) for t in (img5, flow, mask, raw)]
        wref = O.resample(ir[:, :3], fr)
        cref = torch.cat([wref, mr], dim=1) if mode == 'concat' else rr * mr + wref * (1 - mr)
        gw = torch.randn(wref.shape, generator=g)
        gc = torch.randn(cref.shape, generator=g)
        ((wref * gw).sum() + (cref * gc).sum()).backward()
        idv, fd, md, rd = [_dev(t.clone(), device).requires_grad_(True) for t in (img5, flow, mask, raw)]
        if mode == 'concat':
            wv, cv = ops.warp_concat(idv[:, :3], fd, md)
            sep_w = ops.resample(idv[:, :3].detach(), fd.detach())
            sep_c = ops.cat_channels([sep_w, md.detach()])
        else:
            wv, cv = ops.warp_blend(rd, idv[:, :3], fd, md)
            sep_w = ops.resample(idv[:, :3].detach(), fd.detach())
            sep_c = ops.blend(rd.detach(), sep_w, md.detach())
        assert torch.equal(wv.detach().cpu(), sep_w.cpu()) and torch.equal(cv.detach().cpu(), sep_c.cpu()), mode
        ((wv * _dev(gw, device)).sum() + (cv * _dev(gc, device)).sum()).backward()
        assert_close(mode + ' warp', wv, wref, 1e-5)
        assert_close(mode + ' comp', cv, cref, 1e-5)
        assert_close(mode + ' dimg', idv.grad, ir.grad, 1e-5)
        assert_close(mode + ' dflow', fd.grad, fr.grad, 1e-4)
        assert_close(mode + ' dmask', md.grad, mr.grad, 1e-5)
        if mode == 'blend':
            assert_close('blend draw', rd.grad, rr.grad, 1e-5)
        # gradient through one output only
        fd2, md2 = _dev(flow.clone(), device).requires_grad_(True), _dev(mask.clone(), device).requires_grad_(True)
        if mode == 'concat':
            wv2, cv2 = ops.warp_concat(_dev(img5, device)[:, :3], fd2, md2)
        else:
            wv2, cv2 = ops.warp_blend(_dev(raw, device), _dev(img5, device)[:, :3], fd2, md2)
        (cv2 * _dev(gc, device)).sum().backward()
        fr2, mr2 = flow.clone().requires_grad_(True), mask.clone().requires_grad_(True)
        w2 = O.resample(img5[:, :3], fr2)
        c2 = torch.cat([w2, mr2], dim=1) if mode == 'concat' else raw * mr2 + w2 * (1 - mr2)
        (c2 * gc).sum().backward()
        assert_close(mode + ' dflow (comp only)', fd2.grad, fr2.grad, 1e-4)
        assert_close(mode + ' dmask (comp only)', md2.grad, mr2.grad, 1e-5)


def check_conv_groups(device, seed=77, big=False):
    """Grouped launches (conv.launch_group -> fsv_conv_gather_group / fsv_conv_wgrad_group): heterogeneous independent problems in
    one grid give, bit for bit, what the same launches give one by one (same kernels, same k order; split problems to rounding),
    and both equal F.conv2d.  Covers: float4 and scalar-gather groups, strided placement (the parity classes of a stride-2 data
    gradient), K-split accumulating problems, per-sample weights, the LeakyReLU'-multiplying epilogue, > 16 problems (two
    grids), grouped weight gradients with and without Cout % 4 == 0."""
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    sc = 2 if big else 1

    def problem(n, cin, h, w, cout, k, act, with_res, with_bias):
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
        b = torch.randn(cout, generator=g) if with_bias else None
        r = torch.randn(n, cout, h, w, generator=g) if with_res else None
        return x, wt, b, r, k, act

    specs = [problem(2, 16 * sc, 9, 7, 40, 3, conv.ACT_LRELU, False, True),
             problem(1, 32, 5, 33 * sc, 24, 1, conv.ACT_NONE, True, True),
             problem(3, 8, 4, 4, 130, 3, conv.ACT_TANH, False, False),
             problem(1, 64 * sc, 1, 50, 66, 1, conv.ACT_LRELU, False, True)]
    specs += [problem(1, 8, 3, 5 + i, 12 + 4 * i, 1, conv.ACT_NONE, False, True) for i in range(16)]     # 20 problems: two grids

    def run(grouped, specs, tile=-1):
        outs = []
        with conv.launch_group(grouped, force_tile=tile):
            for x, wt, b, r, k, act in specs:
                geom = conv.Geom(k, k, 1, k // 2)
                wk, _, ldw = conv.prep_weight(_dev(wt, device), 0, geom)
                outs.append(conv.conv_forward(_dev(x, device), wk, ldw, wt.shape[0], geom, bias=_dev(b, device) if b is not None else None,
                                              res=_dev(r, device) if r is not None else None, act=act))
        return outs
    specs_s0 = [problem(2, 6, 6, 5, 20, 3, conv.ACT_LRELU, False, True), problem(1, 16, 4, 9, 33, 1, conv.ACT_NONE, True, True)]
    one, grp = run(False, specs), run(True, specs)
    for i, ((x, wt, b, r, k, act), a, bb) in enumerate(zip(specs, one, grp)):
        ref = F.conv2d(x, wt, b, padding=k // 2)
        ref = {conv.ACT_NONE: lambda t: t, conv.ACT_LRELU: O.actvn, conv.ACT_TANH: torch.tanh}[act](ref)
        if r is not None:
            ref = ref + r
        assert_close('group conv %d vs torch' % i, bb, ref)
        assert float((a.cpu() - bb.cpu()).abs().max()) == 0.0, 'grouped launch differs from the single launch (problem %d)' % i
    for tile in (0, 1, 2, 9):               # every tile shape of the grouped kernel (the plan picked 64x64 above)
        for i, (a, bb) in enumerate(zip(one, run(True, specs[:6], tile))):
            if tile == 9:
                # the 64x128 shape runs as the loads-straight-into-LDS variant, which sums every output's products in the order
                # k0 k2 k1 k3 per quad instead of ascending (csrc/conv_igemm.hip, LD): same products, not the same bits
                assert_close('grouped launch, tile 9, problem %d' % i, bb, a, tol=2e-5)
                continue
            assert float((a.cpu() - bb.cpu()).abs().max()) == 0.0, 'grouped launch, tile %d, problem %d' % (tile, i)
        for i, (a, bb) in enumerate(zip(run(False, specs_s0), run(True, specs_s0, tile))):
            assert float((a.cpu() - bb.cpu()).abs().max()) == 0.0, 'scalar-gather group, tile %d, problem %d' % (tile, i)
    # scalar-gather group (Cin % 4 != 0 in one problem sends the whole group to the v1 kernel)
    specs_s = [problem(2, 6, 6, 5, 20, 3, conv.ACT_LRELU, False, True), problem(1, 16, 4, 9, 33, 1, conv.ACT_NONE, True, True)]
    one, grp = run(False, specs_s), run(True, specs_s)
    for i, (a, bb) in enumerate(zip(one, grp)):
        assert float((a.cpu() - bb.cpu()).abs().max()) == 0.0, 'scalar-gather group differs (problem %d)' % i
    # stride-2 data gradient: four parity classes in one launch, plain stores and (small maps, long K) K-split atomics
    for (n, cin, h, w, cout, k, p) in ((2, 16, 12, 10, 32, 3, 1), (1, 8, 9, 9, 24, 4, 2), (1, 128 * sc, 6, 6, 64, 3, 1)):
        x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
        wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
        y = F.conv2d(x, wt, None, stride=2, padding=p)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        geom = conv.Geom(k, k, 2, p)
        dx = conv.conv_dgrad(_dev(dy, device), _dev(wt, device), geom, (h, w))
        assert_close('grouped stride-2 dgrad %s' % ((n, cin, h, w, cout, k),), dx, x.grad)
    # the epilogue that multiplies by LeakyReLU'(aux): dgrad + act-backward of the layer below in one launch
    for cin in (24, 18):                      # float4 and scalar gather
        d = torch.randn(1, cin, 1, 70, generator=g)
        wt = torch.randn(cin, 40, 1, 1, generator=g) * 0.2        # [Cout_fwd = cin of the dgrad GEMM][Cin_fwd]
        ybelow = torch.randn(1, 40, 1, 70, generator=g)
        geom = conv.Geom(1, 1, 1, 0)
        wk, _, ldw = conv.prep_weight(_dev(wt, device), 1, geom)
        got = conv.gather_gemm(_dev(d, device), wk, ldw, 40, 1, 70, [0], [0], 1, 1, act=ops.ACT_DLRELU,
                               res=_dev(ybelow, device).contiguous(memory_format=torch.channels_last))
        ref = torch.einsum('nchw,co->nohw', d, wt[:, :, 0, 0]) * torch.where(ybelow > 0, 1.0, 0.2)
        assert_close('dgrad x leaky_relu\'(y) cin %d' % cin, got, ref)
    # grouped weight gradients against the single launches and torch
    class Arena:
        def take(self, nfloats):
            return torch.zeros(nfloats, dtype=torch.float32, device=device)
    wspecs = [(1, 32, 1, 96, 64, 1), (1, 64, 1, 40 * sc, 130, 1), (2, 16, 7, 9, 72, 3), (1, 8, 1, 64, 66, 1)]

    def wrun(grouped):
        outs = []
        with conv.launch_group(grouped):
            for (n, cin, h, w, cout, k) in wspecs:
                gg = torch.Generator().manual_seed(seed + cin + cout)
                x = torch.randn(n, cin, h, w, generator=gg)
                dy = torch.randn(n, cout, h, w, generator=gg)
                geom = conv.Geom(k, k, 1, k // 2)
                dwt = conv.conv_wgrad(_dev(x, device), _dev(dy, device), geom, (cout, cin, k, k), raw=True, arena=Arena())
                outs.append((x, dy, geom, dwt, (cout, cin, k, k)))
        return outs
    for (x, dy, geom, dwt_s, shp), (_, _, _, dwt_g, _) in zip(wrun(False), wrun(True)):
        xr = x.clone()
        wr = torch.zeros(shp, requires_grad=True)
        F.conv2d(xr, wr, None, padding=shp[-1] // 2).backward(dy)
        kpad, ldw = -(-geom.ntaps * shp[1] // 32) * 32, -(-shp[0] // 32) * 32
        got = conv.unprep_weight_grad(dwt_g.view(1, kpad, ldw), shp, geom)
        assert_close('grouped wgrad %s' % (shp,), got, wr.grad, tol=2e-5)
        one = conv.unprep_weight_grad(dwt_s.view(1, kpad, ldw), shp, geom)
        assert_close('grouped wgrad vs single %s' % (shp,), got, one, tol=2e-6)


def check_spade_pair(device, n=2, c=64, chs=(16, 8, 8), h=10, w=12, up=True, seed=55):
    """Two SPADE sites on the same x and maps (bn_s / bn_0 of a SPADEResnetBlock) through ops.spade_pair - ONE two-site launch -
    against the same two sites launched one by one: outputs bit-equal (same arithmetic per element), gradients equal (the
    backward passes are the single-site twins either way).  The single-site form is held to the oracle by check_spade."""
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    xs_h, xs_w = (h // 2, w // 2) if up else (h, w)
    x = torch.randn(n, c, xs_h, xs_w, generator=g)
    maps = [torch.randn(n, ch, h, w, generator=g) for ch in chs]

    def site_weights(gen0):
        ws = []
        for k, ch in enumerate(chs):
            if k == 0 and gen0:
                ws.append((torch.randn(n, c, ch, 1, 1, generator=g) * 0.3, torch.randn(n, c, ch, 1, 1, generator=g) * 0.3,
                           torch.randn(n, c, generator=g) * 0.3, torch.randn(n, c, generator=g) * 0.3))
            else:
                ws.append((torch.randn(c, ch, 1, 1, generator=g) * 0.3, torch.randn(c, ch, 1, 1, generator=g) * 0.3,
                           torch.randn(c, generator=g) * 0.3, torch.randn(c, generator=g) * 0.3))
        return ws
    w_s, w_0 = site_weights(True), site_weights(True)
    dy_s, dy_0 = torch.randn(n, c, h, w, generator=g), torch.randn(n, c, h, w, generator=g)

    def run(paired):
        # channels-last tensors, as between the layers of the network: both sites then see the SAME x / map buffers
        cl = lambda t: _dev(t, device).contiguous(memory_format=torch.channels_last)
        xd = cl(x).requires_grad_(True)
        md = [cl(m).requires_grad_(True) for m in maps]
        wd_s = [tuple(_dev(t, device).requires_grad_(True) for t in ws) for ws in w_s]
        wd_0 = [tuple(_dev(t, device).requires_grad_(True) for t in ws) for ws in w_0]
        rm = [_dev(torch.zeros(c), device) for _ in range(2)]
        rv = [_dev(torch.ones(c), device) for _ in range(2)]
        import contextlib, os
        os.environ['FSV_SPADE_PAIR'] = '1'            # opt-in switch of the two-site launch
        with (ops.spade_pair() if paired else contextlib.nullcontext()):
            hs = ops.spade_mod(xd, md, wd_s, rm[0], rv[0], act=conv.ACT_NONE, up=up)
            h0 = ops.spade_mod(xd, md, wd_0, rm[1], rv[1], act=conv.ACT_LRELU, up=up)
        os.environ.pop('FSV_SPADE_PAIR', None)
        ((hs * _dev(dy_s, device)).sum() + (h0 * _dev(dy_0, device)).sum()).backward()
        grads = [xd.grad] + [m.grad for m in md] + [t.grad for ws in wd_s + wd_0 for t in ws]
        return hs.detach(), h0.detach(), grads, rm, rv
    hs1, h01, g1, rm1, rv1 = run(False)
    hs2, h02, g2, rm2, rv2 = run(True)
    assert float((hs1.cpu() - hs2.cpu()).abs().max()) == 0.0 and float((h01.cpu() - h02.cpu()).abs().max()) == 0.0, \
        'two-site launch differs from the single-site launches'
    for i, (a, b) in enumerate(zip(g1, g2)):
        assert_close('paired spade grad %d' % i, b, a, tol=1e-6)
    for a, b in zip(rm1 + rv1, rm2 + rv2):
        assert float((a.cpu() - b.cpu()).abs().max()) == 0.0


def check_spade_conv_s(device, n=2, c=64, cout=32, chs=(16, 8), h=10, w=12, up=True, grad=True, spectral=True, seed=57,
                       max_gx=None, amp=False):
    """x_s = conv_s(bn_s(x, maps)) (architecture.py:103-108) through ops.spade_into_conv - ONE launch of csrc/spade_conv.hip -
    against the same two operators launched one after the other (each held to the oracle by check_spade / check_conv): x_s within
    the fp32 summation-order band (the fused kernel sums the 64 channels of a tile as two interleaved halves), gradients equal
    (the backward passes are the same two nodes; the training forward writes the modulated tensor as a side output for conv_s'
    weight gradient).  grad=False: the forward that keeps no graph - the call must not hand the kernel an hs pointer at all.
    max_gx: FSV_SPADE_MAX_GX, a workgroup then walks several pixel tiles.  amp: the `--amp` arithmetic on the half-precision
    kernels - both forms round the maps, the weights and the modulated value to half at the same places and accumulate in fp32, so
    they still agree to summation order (map channels a multiple of 8)."""
    import contextlib
    from importlib import import_module
    ops, conv = pkg()
    prev_mode = conv.set_mfma_mode(1 if amp else conv.mfma_mode())
    os.environ['FSV_SPADE_F16'] = '1'
    try:
        _check_spade_conv_s(device, ops, conv, n, c, cout, chs, h, w, up, grad, spectral, seed, max_gx, amp)
    finally:
        conv.set_mfma_mode(prev_mode)


def _check_spade_conv_s(device, ops, conv, n, c, cout, chs, h, w, up, grad, spectral, seed, max_gx, amp):
    import contextlib
    from importlib import import_module
    lib = import_module('few-shot-vid2vid_amd.lib')
    g = torch.Generator().manual_seed(seed)
    xs_h, xs_w = (h // 2, w // 2) if up else (h, w)
    x = torch.randn(n, c, xs_h, xs_w, generator=g) + 0.2
    maps = [torch.randn(n, ch, h, w, generator=g) for ch in chs]
    wts = []
    for k, ch in enumerate(chs):
        if k == 0:
            wts.append((torch.randn(n, c, ch, 1, 1, generator=g) * 0.3, torch.randn(n, c, ch, 1, 1, generator=g) * 0.3,
                        torch.randn(n, c, generator=g) * 0.3, torch.randn(n, c, generator=g) * 0.3))
        else:
            wts.append((torch.randn(c, ch, 1, 1, generator=g) * 0.3, torch.randn(c, ch, 1, 1, generator=g) * 0.3,
                        torch.randn(c, generator=g) * 0.3, torch.randn(c, generator=g) * 0.3))
    wconv = torch.randn(cout, c, 1, 1, generator=g) * (1.0 / c ** 0.5)
    u0 = F.normalize(torch.randn(cout, generator=g), dim=0)
    v0 = F.normalize(torch.randn(c, generator=g), dim=0)
    dy = torch.randn(n, cout, h, w, generator=g)

    def run(fused):
        # (fresh leaves per run: on the emulator `.to(device)` hands back the host tensor itself, and two runs that share leaves
        # would add their gradients up in the same .grad)
        cl = lambda t: _dev(t, device).detach().clone().contiguous(memory_format=torch.channels_last)
        xd = cl(x).requires_grad_(grad)
        md = [cl(m).requires_grad_(grad) for m in maps]
        wd = [tuple(_dev(t, device).detach().clone().requires_grad_(grad) for t in ws) for ws in wts]
        # (the convolution's weight is a module parameter: it requires a gradient also in the forward that keeps no graph - the
        # kernel must not be asked for its side output there, round 6)
        wc = _dev(wconv, device).detach().clone().requires_grad_(True)
        rm, rv = _dev(torch.zeros(c), device), _dev(torch.ones(c), device)
        u, v = _dev(u0.clone(), device), _dev(v0.clone(), device)
        seen, real_call = [], lib.call

        def recording_call(name, *a):
            seen.append((name, a))
            return real_call(name, *a)
        lib.call = recording_call
        os.environ['FSV_SPADE_CONV_S'] = '1' if fused else '0'
        if max_gx:
            os.environ['FSV_SPADE_MAX_GX'] = str(max_gx)
        try:
            with (contextlib.nullcontext() if grad else torch.no_grad()):
                with ops.spade_into_conv():
                    hs = ops.spade_mod(xd, md, wd, rm, rv, act=conv.ACT_NONE, up=up)
                    sn = ops.SpectralState.update(wc, u, v, True) if spectral else None
                    y = ops.conv2d(hs, wc, None, 1, 0, sn=(sn, u, v) if spectral else None)
            grads = None
            if grad:
                (y * _dev(dy, device)).sum().backward()
                grads = [xd.grad] + [m.grad for m in md] + [t.grad for ws in wd for t in ws] + [wc.grad]
        finally:
            lib.call = real_call
            os.environ.pop('FSV_SPADE_CONV_S', None)
            os.environ.pop('FSV_SPADE_MAX_GX', None)
        return y.detach(), grads, seen, (rm, rv)
    y1, g1, seen1, st1 = run(False)
    y2, g2, seen2, st2 = run(True)
    names1, names2 = [s_[0] for s_ in seen1], [s_[0] for s_ in seen2]
    sfx = '_h' if amp else ''
    assert 'fsv_spade_conv_s_fwd' + sfx not in names1 and 'fsv_spade_mod_fwd' + sfx in names1, names1
    assert 'fsv_spade_conv_s_fwd' + sfx in names2 and 'fsv_spade_mod_fwd' + sfx not in names2, names2
    fused_args = [a for (nm, a) in seen2 if nm == 'fsv_spade_conv_s_fwd' + sfx][0]
    assert (fused_args[3] is not None) == grad, 'the modulated tensor is a side output of the training forward only'
    assert_close('fused bn_s -> conv_s output', y2, y1, tol=2e-5)
    for a, b in zip(st1, st2):
        assert float((a.cpu() - b.cpu()).abs().max()) == 0.0
    if grad:
        for i, (a, b) in enumerate(zip(g1, g2)):
            assert_close('fused bn_s -> conv_s grad %d' % i, b, a, tol=2e-5)
    # ... and the fused launch against the ORACLE directly (round-4 review: the comparison above holds the product to the product):
    # normalization.py:37-52 + architecture.py:103-108 restated by oracle/fsv_oracle.py (spade, _sn_conv), in the `--amp` arithmetic
    # (oracle/np_oracle.amp_conv2d: operands and the half-stored modulated tensor rounded once) when amp.  fp32: the kernels' fp32
    # band; amp: two fp32 evaluations of the modulated value that straddle a half rounding boundary differ by a whole half ulp of it
    # (5e-4 of ONE of the 64 - 128 terms of an output), and the gradient of a half-stored tensor is rounded to half on both sides.
    from oracle import np_oracle
    arith = (lambda: O.arithmetic(np_oracle.amp_conv2d)) if amp else contextlib.nullcontext
    leaf = lambda t: t.detach().clone().requires_grad_(grad)
    xr, mr = leaf(x), [leaf(m) for m in maps]
    wr = [tuple(leaf(t) for t in ws) for ws in wts]
    wcr = leaf(wconv)
    fixed_r = [None if k == 0 else (wr[k][0], wr[k][2], wr[k][1], wr[k][3]) for k in range(len(chs))]
    gen_r = ((wr[0][0], wr[0][2]), (wr[0][1], wr[0][3]))
    sd = {'weight_orig': wcr, 'weight_u': u0.clone(), 'weight_v': v0.clone()}
    with (contextlib.nullcontext() if grad else torch.no_grad()):
        with arith():
            x_in = F.interpolate(xr, scale_factor=2, mode='nearest') if up else xr
            h_ref = O.spade(x_in, mr, fixed_r, gen_r)
            y_ref = O._sn_conv(sd, '', h_ref, 1, 0, dx_half=True) if spectral else O._conv2d(h_ref, wcr, None, 1, 0, dx_half=True)
            if grad:
                (y_ref * dy).sum().backward()
    otol = 3e-3 if amp else 2e-5
    assert_close('fused bn_s -> conv_s output vs the oracle', y2, y_ref, tol=otol)
    if grad:
        refs = [xr.grad] + [m.grad for m in mr] + [t.grad for ws in wr for t in ws] + [wcr.grad]
        for i, (a, b) in enumerate(zip(refs, g2)):
            assert_close('fused bn_s -> conv_s grad %d vs the oracle' % i, b, a, tol=otol * 2)


def check_spade_conv3(device, n=2, c=64, cout=32, chs=(16, 8), h=20, w=24, up=True, grad=True, spectral=True, res=False, act='lrelu',
                      seed=59):
    """dx = conv_0(actvn(bn_0(x, maps))) / conv_1(actvn(bn_1(dx, maps))) + x_s (architecture.py:96-99) through
    ops.spade_into_conv(conv3=True) - ONE launch of csrc/spade_conv3.hip (the modulated haloed tile in LDS, round 6) - against the same
    two operators launched one after the other AND against the oracle directly (normalization.py:37-52 + leaky_relu + the 3x3
    spectral-norm convolution restated by oracle/fsv_oracle.py): output within the fp32 summation-order band, gradients equal (the
    backward passes are the same two nodes; the training forward writes the modulated tensor as a side output).  Sizes that are not a
    multiple of the 8 x 16 tile exercise the partial tiles; every tile has halo pixels outside the image (the convolution's padding)."""
    import contextlib
    from importlib import import_module
    ops, conv = pkg()
    lib = import_module('few-shot-vid2vid_amd.lib')
    g = torch.Generator().manual_seed(seed)
    xs_h, xs_w = (h // 2, w // 2) if up else (h, w)
    x = torch.randn(n, c, xs_h, xs_w, generator=g) + 0.2
    maps = [torch.randn(n, ch, h, w, generator=g) for ch in chs]
    wts = []
    for k, ch in enumerate(chs):
        if k == 0:
            wts.append((torch.randn(n, c, ch, 1, 1, generator=g) * 0.3, torch.randn(n, c, ch, 1, 1, generator=g) * 0.3,
                        torch.randn(n, c, generator=g) * 0.3, torch.randn(n, c, generator=g) * 0.3))
        else:
            wts.append((torch.randn(c, ch, 1, 1, generator=g) * 0.3, torch.randn(c, ch, 1, 1, generator=g) * 0.3,
                        torch.randn(c, generator=g) * 0.3, torch.randn(c, generator=g) * 0.3))
    wconv = torch.randn(cout, c, 3, 3, generator=g) * (1.0 / (9 * c) ** 0.5)
    bconv = torch.randn(cout, generator=g) * 0.1
    u0 = F.normalize(torch.randn(cout, generator=g), dim=0)
    v0 = F.normalize(torch.randn(c * 9, generator=g), dim=0)
    rs = torch.randn(n, cout, h, w, generator=g) if res else None
    dy = torch.randn(n, cout, h, w, generator=g)
    act_code = conv.ACT_LRELU if act == 'lrelu' else conv.ACT_NONE

    def run(fused):
        cl = lambda t: _dev(t, device).detach().clone().contiguous(memory_format=torch.channels_last)
        xd = cl(x).requires_grad_(grad)
        md = [cl(m).requires_grad_(grad) for m in maps]
        wd = [tuple(_dev(t, device).detach().clone().requires_grad_(grad) for t in ws) for ws in wts]
        # (module parameters: they require a gradient also in the forward that keeps no graph - no side output there)
        wc = _dev(wconv, device).detach().clone().requires_grad_(True)
        bc = _dev(bconv, device).detach().clone().requires_grad_(True)
        rd = cl(rs).requires_grad_(grad) if res else None
        rm, rv = _dev(torch.zeros(c), device), _dev(torch.ones(c), device)
        u, v = _dev(u0.clone(), device), _dev(v0.clone(), device)
        seen, real_call = [], lib.call

        def recording_call(name, *a):
            seen.append((name, a))
            return real_call(name, *a)
        lib.call = recording_call
        os.environ['FSV_SPADE_CONV3'] = '1' if fused else '0'
        try:
            with (contextlib.nullcontext() if grad else torch.no_grad()):
                with ops.spade_into_conv(conv3=True):
                    hm = ops.spade_mod(xd, md, wd, rm, rv, act=act_code, up=up)
                    sn = ops.SpectralState.update(wc, u, v, True) if spectral else None
                    y = ops.conv2d(hm, wc, bc, 1, 1, sn=(sn, u, v) if spectral else None, res=rd, stats_groups=1)
            grads = None
            if grad:
                (y * _dev(dy, device)).sum().backward()
                grads = ([xd.grad] + [m.grad for m in md] + [t.grad for ws in wd for t in ws] + [wc.grad, bc.grad] +
                         ([rd.grad] if res else []))
        finally:
            lib.call = real_call
            os.environ.pop('FSV_SPADE_CONV3', None)
        # the BatchNorm statistics of the output from the fused launch's own epilogue (as the gather-GEMM's: fp64 partials)
        ys = getattr(y, '_fsv_stats', None)
        if fused and conv.stats_enabled():
            assert ys is not None, 'the fused launch leaves the statistics of its output'
            part, groups, slots, px, ch_ = ys
            got = part.view(slots, ch_, 2).sum(0).cpu()
            yd = y.detach().double().cpu()
            want = torch.stack([yd.sum(dim=(0, 2, 3)), (yd * yd).sum(dim=(0, 2, 3))], dim=1)
            assert groups == 1 and px == n * h * w and float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()), (got, want)
        return y.detach(), grads, seen, (rm, rv)
    y1, g1, seen1, st1 = run(False)
    y2, g2, seen2, st2 = run(True)
    names1, names2 = [s_[0] for s_ in seen1], [s_[0] for s_ in seen2]
    assert 'fsv_spade_conv3_fwd' not in names1 and 'fsv_spade_mod_fwd' in names1, names1
    assert 'fsv_spade_conv3_fwd' in names2 and 'fsv_spade_mod_fwd' not in names2, names2
    # the convolution itself is in the fused launch (the gather launches that remain are the data gradients of the backward pass)
    fwd = lambda names: sum(nm in ('fsv_conv_gather_fwd', 'fsv_conv_gather_fwd_stats') for nm in names)
    assert fwd(names2) == fwd(names1) - 1, (names1, names2)
    fused_args = [a for (nm, a) in seen2 if nm == 'fsv_spade_conv3_fwd'][0]
    assert (fused_args[3] is not None) == grad, 'the modulated tensor is a side output of the training forward only'
    assert_close('fused bn -> actvn -> conv3x3 output', y2, y1, tol=2e-5)
    for a, b in zip(st1, st2):
        assert float((a.cpu() - b.cpu()).abs().max()) == 0.0
    if grad:
        for i, (a, b) in enumerate(zip(g1, g2)):
            assert_close('fused bn -> actvn -> conv3x3 grad %d' % i, b, a, tol=2e-5)
    leaf = lambda t: t.detach().clone().requires_grad_(grad)
    xr, mr = leaf(x), [leaf(m) for m in maps]
    wr = [tuple(leaf(t) for t in ws) for ws in wts]
    wcr, bcr = leaf(wconv), leaf(bconv)
    rr = leaf(rs) if res else None
    fixed_r = [None if k == 0 else (wr[k][0], wr[k][2], wr[k][1], wr[k][3]) for k in range(len(chs))]
    gen_r = ((wr[0][0], wr[0][2]), (wr[0][1], wr[0][3]))
    sd = {'weight_orig': wcr, 'weight_u': u0.clone(), 'weight_v': v0.clone(), 'bias': bcr}
    with (contextlib.nullcontext() if grad else torch.no_grad()):
        x_in = F.interpolate(xr, scale_factor=2, mode='nearest') if up else xr
        h_ref = O.spade(x_in, mr, fixed_r, gen_r)
        if act == 'lrelu':
            h_ref = F.leaky_relu(h_ref, 0.2)
        y_ref = O._sn_conv(sd, '', h_ref, 1, 1) if spectral else O._conv2d(h_ref, wcr, bcr, 1, 1)
        if res:
            y_ref = y_ref + rr
        if grad:
            (y_ref * dy).sum().backward()
    assert_close('fused bn -> actvn -> conv3x3 output vs the oracle', y2, y_ref, tol=2e-5)
    if grad:
        refs = [xr.grad] + [m.grad for m in mr] + [t.grad for ws in wr for t in ws] + [wcr.grad, bcr.grad] + ([rr.grad] if res else [])
        for i, (a, b) in enumerate(zip(refs, g2)):
            assert_close('fused bn -> actvn -> conv3x3 grad %d vs the oracle' % i, b, a, tol=4e-5)


def check_pooled_product(device, b=2, c=64, h=16, w=16, seed=97):
    """ops.pooled_product: prod[b, i, j] = sum_p a[b, i, p] * softmax(l)[b, j, p] (generator.py:378-389: torch.bmm of the image
    features with the transposed channel softmax of the label features) issued as a per-sample 1x1 weight-gradient GEMM (round 6:
    both operands read in place) - values and both gradients against the bmm, and against the gather-GEMM form it replaces
    (the default, FSV_POOL_WGRAD=0) at summation-order distance; the launch list: one weight-gradient launch, no re-arrangement, no copy."""
    from importlib import import_module
    ops, conv = pkg()
    lib = import_module('few-shot-vid2vid_amd.lib')
    networks = import_module('few-shot-vid2vid_amd.networks')
    g = torch.Generator().manual_seed(seed)
    a0 = torch.randn(b, c, h, w, generator=g)
    l0 = torch.randn(b, c, h, w, generator=g)
    dy = torch.randn(b, c, c, 1, generator=g)

    def run(new):
        cl = lambda t: _dev(t, device).detach().clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        a, l = cl(a0), cl(l0)
        seen, real_call = [], lib.call

        def recording_call(name, *args):
            seen.append(name)
            return real_call(name, *args)
        os.environ['FSV_POOL_WGRAD'] = '1' if new else '0'
        lib.call = recording_call
        try:
            (enc,) = networks.FewShotGenerator._pooled([a], [l])
            n_fwd = len(seen)
            (enc * _dev(dy, device)).sum().backward()
        finally:
            lib.call = real_call
            os.environ.pop('FSV_POOL_WGRAD', None)
        return enc.detach(), a.grad, l.grad, seen[:n_fwd], seen[n_fwd:]
    y1, ga1, gl1, f1, b1 = run(True)
    y0, ga0, gl0, f0, b0 = run(False)
    assert f1 == ['fsv_softmax_rows_fwd', 'fsv_conv_wgrad'], f1
    assert 'fsv_conv_wgrad' not in f0 and len(f0) > len(f1), f0
    # backward: two position-major 1x1 convolutions + one c x c re-arrangement; no weight-gradient launch, no un-arrangement
    assert [n for n in b1 if n != 'fsv_conv_plan'] == ['fsv_prep_weight', 'fsv_conv_gather_fwd', 'fsv_conv_gather_fwd', 'fsv_softmax_rows_bwd'], b1
    ar, lr = a0.clone().requires_grad_(True), l0.clone().requires_grad_(True)
    sm = torch.softmax(lr, dim=1)
    ref = torch.bmm(ar.reshape(b, c, h * w), sm.reshape(b, c, h * w).transpose(1, 2)).unsqueeze(-1)
    (ref * dy).sum().backward()
    for nm, got, want in (('product', y1, ref), ('d a', ga1, ar.grad), ('d label', gl1, lr.grad),
                          ('product vs the gather-GEMM form', y1, y0), ('d a vs the gather-GEMM form', ga1, ga0),
                          ('d label vs the gather-GEMM form', gl1, gl0)):
        assert_close('softmax pooling ' + nm, got, want, tol=2e-5)


def check_weighted_sum(device, seed=96):
    """ops.weighted_sum (the loss collector's `sum(lambda_i * term_i)`, loss_collector.py:60-67,161-162,204,218-219) as one launch
    each way (round 6) against the torch expression: value to the last fp32 rounding of a sequential sum, every term's gradient
    = lambda_i * g exactly; terms that need no gradient get none; the torch form (FSV_WSUM=0) agrees."""
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    for n, with_w in ((2, False), (5, True), (17, True), (32, True)):
        vals = torch.randn(n, generator=g) * 3
        ws = [float(w) for w in (torch.rand(n, generator=g) * 4 + 0.1)] if with_w else None
        terms = [_dev(vals[i:i + 1].clone(), device).requires_grad_(i % 3 != 1) for i in range(n)]
        y = ops.weighted_sum([t.view(1, 1) if i % 2 else t for i, t in enumerate(terms)], ws)
        os.environ['FSV_WSUM'] = '0'
        try:
            terms0 = [_dev(vals[i:i + 1].clone(), device).requires_grad_(i % 3 != 1) for i in range(n)]
            y0 = ops.weighted_sum(terms0, ws)
        finally:
            os.environ.pop('FSV_WSUM', None)
        ref = sum((ws[i] if ws else 1.0) * float(vals[i]) for i in range(n))
        assert tuple(y.shape) == (1,) and abs(float(y.detach()) - ref) <= 1e-5 * max(1.0, abs(ref)), (float(y.detach()), ref)
        assert abs(float(y.detach()) - float(y0.detach())) <= 1e-5 * max(1.0, abs(ref))
        (y * 1.5).backward()
        (y0 * 1.5).backward()
        for i, (t, t0) in enumerate(zip(terms, terms0)):
            if i % 3 == 1:
                assert t.grad is None and t0.grad is None
            else:
                want = torch.tensor([(ws[i] if ws else 1.0)], dtype=torch.float32) * 1.5
                assert float(t.grad.cpu()) == float(want) == float(t0.grad.cpu()), (i, t.grad, want, t0.grad)


def check_loss_ticket(device, seed=95):
    """The loss reductions that finish in their own launch (csrc/losses.hip fsv_loss_finish, round 6: the last workgroup sums the
    partials in index order) give the BITS of the two-launch form (FSV_LOSS_TICKET=0), for several grid sizes, repeatedly (the
    ticket must come back to zero), and issue one launch instead of two."""
    from importlib import import_module
    ops, conv = pkg()
    lib = import_module('few-shot-vid2vid_amd.lib')
    g = torch.Generator().manual_seed(seed)
    for shape in ((1, 3, 5, 7), (2, 8, 33, 40), (2, 3, 128, 160)):
        a, b = _dev(torch.randn(*shape, generator=g), device), _dev(torch.randn(*shape, generator=g), device)
        m = _dev((torch.rand(shape[0], 1, *shape[2:], generator=g) > 0.4).float(), device)
        x = _dev(torch.randn(*shape, generator=g), device)
        got = {}
        for mode in ('1', '0'):
            os.environ['FSV_LOSS_TICKET'] = mode
            seen, real_call = [], lib.call
            lib.call = lambda name, *args: (seen.append(name), real_call(name, *args))[1]
            try:
                vals = []
                for _ in range(3):
                    vals += [ops.l1_loss(a, b), ops.l1_loss(a, b, m), ops.l1_loss(a, 1.0, m), ops.hinge_loss(x, True),
                             ops.hinge_loss(x, False)]
                got[mode] = torch.cat([v.detach().reshape(1).cpu() for v in vals])
            finally:
                lib.call = real_call
                os.environ.pop('FSV_LOSS_TICKET', None)
        assert torch.equal(got['1'], got['0']), (got['1'], got['0'])
        ref = torch.stack([(a - b).abs().mean().cpu(), ((a - b) * m).abs().mean().cpu()])
        assert float((got['1'][:2] - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


def check_conv_stats(device, seed=61):
    """BatchNorm / InstanceNorm statistics from the producing convolution's epilogue (ops.conv2d stats_groups -> `_fsv_stats` ->
    norm_act / spade_mod) against the separate reduction pass: same normalised output, running statistics and gradients; the
    InstanceNorm geometry makes pixel tiles straddle two samples; K-split plans and scalar-gather layers fall back silently."""
    ops, conv = pkg()
    g = torch.Generator().manual_seed(seed)
    cases = [  # n, cin, h, w, cout, k, stride, pad, instance, residual
        (2, 16, 12, 10, 24, 3, 1, 1, False, False),
        (3, 8, 23, 23, 40, 4, 2, 2, True, False),         # 3 samples of 12 * 12 = 144 pixels: pixel tiles straddle two samples
        (3, 8, 9, 7, 40, 4, 2, 2, True, False),           # 20 pixels per sample: a tile would span three groups -> fallback
        (2, 32, 8, 8, 32, 3, 1, 1, False, True),
        (2, 6, 8, 8, 16, 3, 1, 1, False, False),          # padded input channels still take the float4 path
        (1, 512, 4, 4, 64, 3, 1, 1, False, False),        # deep K on a tiny map (16 pixels): the plan splits K, and the group is
                                                          # smaller than the finishing pass's 32-pixel tile -> fallback
        # round 6: a K-split launch leaves the statistics from its FINISHING pass (fsv_split_finish4_stats_kernel)
        (2, 512, 8, 8, 64, 3, 1, 1, False, True),         # 128 pixels, 144 K chunks: split; BatchNorm, residual
        (2, 256, 8, 8, 96, 3, 1, 1, True, False),         # InstanceNorm: two groups of 64 pixels
        (2, 512, 8, 8, 40, 3, 1, 1, False, False),        # Cout % 32 != 0 -> fallback
    ]
    for (n, cin, h, w, cout, k, s, p, inst, with_res) in cases:
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
        b = torch.randn(cout, generator=g)
        gam, bet = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
        oh, ow = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        res = torch.randn(n, cout, oh, ow, generator=g) if with_res else None
        dy = torch.randn(n, cout, oh, ow, generator=g)

        def run(stats):
            xd, wd, bd = [_dev(t, device).requires_grad_(True) for t in (x, wt, b)]
            gd, be = _dev(gam, device).requires_grad_(True), _dev(bet, device).requires_grad_(True)
            rm, rv = _dev(torch.zeros(cout), device), _dev(torch.ones(cout), device)
            y = ops.conv2d(xd, wd, bd, stride=s, padding=p, res=_dev(res, device) if with_res else None,
                           stats_groups=(-1 if inst else 1) if stats else 0)
            had = hasattr(y, '_fsv_stats')
            z = ops.norm_act(y, gd, be, None if inst else rm, None if inst else rv, instance=inst, eps=0.1 if inst else 1e-5,
                             act=conv.ACT_LRELU)
            z.backward(_dev(dy, device))
            return z.detach(), [t.grad for t in (xd, wd, bd, gd, be)], rm, rv, had
        z0, g0, rm0, rv0, _ = run(False)
        conv.start_plan_log()
        try:
            z1, g1, rm1, rv1, had = run(True)
        finally:
            splits = [e[1] for e in conv.stop_plan_log() if e[0] != 'group']
        if cin >= 256:
            assert splits and splits[0] > 1, ('the forward launch of this case is meant to split K', splits)
        expect = not (inst and h == 9) and not (cin >= 256 and (h == 4 or cout % 32))
        assert had == expect, ('statistics attribute', (n, cin, h, w, cout), had)
        if had and cin >= 256:
            os.environ['FSV_SPLIT_FIN_STATS'] = '0'
            try:
                z2, _, _, _, had2 = run(True)
            finally:
                os.environ.pop('FSV_SPLIT_FIN_STATS', None)
            assert not had2, 'FSV_SPLIT_FIN_STATS=0 must leave the reduction to the consumer'
            assert_close('statistics from the split finishing pass vs the consumer\'s own pass', z1, z2, tol=2e-6)
        assert_close('norm(conv) with epilogue statistics %s' % ((n, cin, h, w, cout),), z1, z0, tol=2e-6)
        for i, (a, bb) in enumerate(zip(g0, g1)):
            if i == 2:
                continue        # the bias in front of a normalisation has no gradient: both sides hold rounding noise around 0
            assert_close('grad %d with epilogue statistics' % i, bb, a, tol=2e-5)
        if not inst:
            assert_close('running mean', rm1, rm0, tol=1e-6)
            assert_close('running var', rv1, rv0, tol=1e-6)


def check_thin_conv(device, seed=88):
    """Thin-output convolutions (Cout <= 4: image / flow / mask heads) on the vector-ALU kernels (csrc/conv_igemm.hip
    fsv_conv_thin_*): forward + gradients against F.conv2d, and the forward against the gather-GEMM kernel to summation order
    (a forced tile keeps a launch on the MFMA path)."""
    ops, conv = pkg()
    import os
    g = torch.Generator().manual_seed(seed)
    os.environ['FSV_CONV_THIN'] = '2'        # every eligible layer (the library's size rule keeps small maps on the MFMA path)
    try:
        _check_thin_conv(device, ops, conv, g)
    finally:
        os.environ.pop('FSV_CONV_THIN', None)


def _check_thin_conv(device, ops, conv, g):
    import os
    cases = [(2, 32, 17, 19, 3, 3, 1, 1, conv.ACT_TANH, 1.0), (1, 32, 9, 33, 2, 3, 1, 1, conv.ACT_NONE, 20.0),
             (2, 16, 12, 10, 1, 3, 1, 1, conv.ACT_SIGMOID, 1.0), (1, 8, 7, 9, 4, 1, 1, 0, conv.ACT_LRELU, 1.0),
             (1, 64, 10, 12, 3, 3, 2, 1, conv.ACT_NONE, 1.0), (2, 12, 6, 5, 2, 4, 2, 2, conv.ACT_NONE, 1.0)]
    for (n, cin, h, w, cout, k, s, p, act, scale) in cases:
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
        b = torch.randn(cout, generator=g)
        xr, wr, br = (t.clone().requires_grad_(True) for t in (x, wt, b))
        ref = F.conv2d(xr, wr, br, stride=s, padding=p) * scale
        ref = {conv.ACT_NONE: lambda t: t, conv.ACT_TANH: torch.tanh, conv.ACT_SIGMOID: torch.sigmoid, conv.ACT_LRELU: O.actvn}[act](ref)
        dy = torch.randn(ref.shape, generator=g)
        ref.backward(dy)
        xd, wd, bd = (_dev(t, device).requires_grad_(True) for t in (x, wt, b))
        y = ops.conv2d(xd, wd, bd, stride=s, padding=p, act=act, scale=scale)
        y.backward(_dev(dy, device))
        name = 'thin conv %s' % ((n, cin, h, w, cout, k, s),)
        assert_close(name + ' y', y, ref)
        assert_close(name + ' dx', xd.grad, xr.grad)
        assert_close(name + ' dw', wd.grad, wr.grad, tol=2e-5)
        assert_close(name + ' db', bd.grad, br.grad)
        ge = conv.Geom(k, k, s, p)
        wf, kpad, ldw = conv.prep_weight(_dev(wt, device), 0, ge)
        xn = conv.to_nhwc(_dev(x, device))
        thin = conv.conv_forward(xn, wf, ldw, cout, ge, bias=_dev(b, device))
        mfma = conv.conv_forward(xn, wf, ldw, cout, ge, bias=_dev(b, device), force_tile=4, force_split=1)
        assert_close(name + ': vector-ALU kernel vs the gather-GEMM kernel (summation order)', thin, mfma, tol=2e-6)
        if k == 3:
            # round 6 (opt-in, measured neutral): the 3x3 form keeps the lane's weights in registers and issues its nine tap loads
            # together - the same fma chain per output as the generic tap loop, bit for bit
            os.environ['FSV_THIN_T9'] = '1'
            try:
                t9 = conv.conv_forward(xn, wf, ldw, cout, ge, bias=_dev(b, device))
            finally:
                os.environ.pop('FSV_THIN_T9', None)
            assert bool((thin == t9).all()), name + ': register-resident 3x3 form changed the bits'

