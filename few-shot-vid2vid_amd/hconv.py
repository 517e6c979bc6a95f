"""Half-precision convolution operators on csrc/conv_h.hip: the `--amp O1` arithmetic of the reference (models/models.py:22-26,
options/base_options.py:127; BASELINE.json configs[4]) with activations and weights 16-bit in HBM.

Mirrors conv.py (gather_gemm / conv_forward / conv_dgrad / conv_wgrad) for half tensors: logical NCHW shape, NHWC memory,
`torch.float16`.  Weights reach the kernels N-major (`wt[co][Kpad]`, K contiguous) - `prep_weight_h` converts the K-major fp32
layouts of conv.prep_weight / layout_cache (one table-driven launch per call, or per optimiser step for a whole cache).
"""
import ctypes
import os

import torch

from . import conv, lib, profile
from .conv import (ACT_NONE, ACT_DLRELU, Geom, _ceil, empty_nhwc, to_nhwc)

c_p, c_i, c_ll, c_f = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float


class HConvDesc(ctypes.Structure):
    """include/fsv2v.h fsv_hconv_desc"""
    _fields_ = ([(k, c_p) for k in ('inp', 'wt', 'bias', 'res', 'out', 'wscale', 'ws', 'stats')] +
                [(k, c_i) for k in ('N', 'H', 'W', 'Cin', 'OH', 'OW', 'Cout', 'ntaps')] +
                [('ty', c_i * 16), ('tx', c_i * 16)] +
                [(k, c_i) for k in ('sy', 'sx', 'outH', 'outW', 'osy', 'osx', 'ooy', 'oox', 'Kpad', 'nrows', 'per_sample', 'act',
                                    'accumulate', 'out_h', 'res_h', 'force_tile', 'force_split', 'stats_groups', 'stats_slots',
                                    'stats_prezeroed')] +
                [('scale', c_f), ('w_bstride', c_ll), ('b_bstride', c_ll)])


lib.register_sigs({
    "fsv_hconv_gather": [c_p, c_i, ctypes.POINTER(c_i), c_p],
    "fsv_hconv_plan": [c_i] * 7 + [ctypes.POINTER(c_i)] * 2,
    "fsv_hconv_wgrad": [c_p, c_p, c_p] + [c_i] * 7 + [c_i, ctypes.POINTER(c_i), ctypes.POINTER(c_i), c_i, c_i] +
                       [c_i, c_i, c_ll, c_i, c_i, c_i, c_i, c_p],
    "fsv_hconv_prep_weight": [c_p, c_p, c_i, c_p],
    "fsv_hconv_prep_weight_one": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "fsv_cast_half": [c_p, c_p, c_ll, c_i, c_p],
})

# tests: called as _launch_hook(kind, info) behind every launch ('conv': x, wh, kpad, nrows, cout, oh, ow, ty, tx, sy, sx, bias, res,
# act, scale, per_sample, place, wscale, out; 'wgrad': x, dout, geom, per_sample, dwt) - tests/model_checks.verify_half_launches
# recomputes each launch of a real iteration from the same operands with plain torch.  Round 5: every other launch that produces a
# half tensor or belongs to the `--amp` arithmetic reports too - 'side' (t, h: the half side output of norm-apply / norm-backward /
# activation-backward), 'pack' (the packed half discriminator input), 'adam' (csrc/amp.hip: operands before, buffers after),
# 'spade_fwd' / 'spade_bwd' / 'spade_conv_s' (the SPADE kernels on the f16 matrix instructions / with half tensors; ops.py)
_launch_hook = None


def launch_hook():
    return _launch_hook

H_TILE_NAMES = {0: '128x128', 1: '128x64', 2: '128x32', 3: '128x128w8', 4: '64x64', 5: '256x128w8', 9: '64x128'}


def empty_nhwc_h(n, c, h, w, like):
    return torch.empty((n, h, w, c), dtype=torch.float16, device=like.device).permute(0, 3, 1, 2)


def cast(x, dtype):
    """dense element conversion fp32 <-> half on the library's own kernel (same strides, same logical shape)"""
    if x.dtype == dtype:
        return x
    if not (x.is_contiguous() or (x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous())):
        x = x.contiguous()
    y = torch.empty_like(x, dtype=dtype)
    lib.check_device(x, y)
    lib.call("fsv_cast_half", lib.ptr(x), lib.ptr(y), x.numel(), 0 if dtype == torch.float16 else 1, lib.stream_ptr())
    return y


def to_half_nhwc(x):
    """NHWC half tensor with the logical shape of x (no-op for one that already is; the copy its producer left beside an fp32
    tensor - half_side_output - when there is one)"""
    side = getattr(x, '_fsv_h16', None)
    if side is not None and side[0] == x._version and side[1].shape == x.shape:
        return side[1]
    x = to_nhwc(x)
    return x if x.dtype == torch.float16 else cast(x, torch.float16)


class half_side_output:
    """`with half_side_output(t) as side:` around ONE fsv_norm_apply / fsv_norm_bwd* / fsv_act_bwd call that writes the fp32 NHWC
    tensor t: under `--amp` on the half-precision kernels `side.ptr()` is a half buffer the call is handed as its explicit
    `y_half` / `dx_half` argument (include/fsv2v.h: the library keeps no armed pointer between calls) - it then also stores t as
    IEEE half, and the copy rides on the tensor (`_fsv_h16`): the convolution that reads t next finds its operand already
    converted.  Same values as the conversion pass it replaces (one rounding of the fp32 result).  The conditions below are the
    entry points' contract (C % 4 == 0, fewer than 2^31 elements), so a pointer that is handed over is always honoured."""

    def __init__(self, t):
        self.t = t
        self.h = None
        if (conv.h_kernels() and t.dim() == 4 and t.shape[1] % 8 == 0 and t.numel() < 2 ** 31 and t.dtype == torch.float32
                and os.environ.get('FSV_HALF_SIDE', '1') == '1' and t.permute(0, 2, 3, 1).is_contiguous()):
            n, c, h, w = t.shape
            self.h = empty_nhwc_h(n, c, h, w, t)

    def ptr(self):
        return lib.ptr(self.h)

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if self.h is not None and et is None:
            self.t._fsv_h16 = (self.t._version, self.h)
            if _launch_hook is not None:
                _launch_hook('side', dict(t=self.t, h=self.h))
        return False


def planned(mz, cout, nchunks, nsamp, force_tile=-1, force_split=0, can_split=True):
    tile, nsplit = c_i(0), c_i(1)
    lib.call("fsv_hconv_plan", mz, cout, nchunks, nsamp, force_tile, force_split, 1 if can_split else 0, ctypes.byref(tile),
             ctypes.byref(nsplit))
    return tile.value, nsplit.value


# ------------------------------------------------------------------------------------------------ weights
def _prep_tables(jobs, dev):
    """jobs: list of (src fp32 [nb, k32, ldw], dst half [nb, nrows, k64])"""
    words, tmap = [], []
    for j, (src, dst) in enumerate(jobs):
        nb, k32, ldw = src.shape
        _, nrows, k64 = dst.shape
        words += [src.data_ptr(), dst.data_ptr(), k32, ldw, nrows, k64, nb, 0]
        for z in range(nb):
            for a in range((k64 + 63) // 64):
                for b in range((nrows + 63) // 64):
                    tmap += [j, a, b, z]
    return (torch.tensor(words, dtype=torch.int64).to(dev), torch.tensor(tmap, dtype=torch.int32).to(dev), len(tmap) // 4)


def launch_prep(tables):
    words, tmap, nblocks = tables
    lib.call("fsv_hconv_prep_weight", lib.ptr(words), lib.ptr(tmap), nblocks, lib.stream_ptr())


def half_layout_like(wt):
    """zeroed N-major half buffer for the K-major fp32 layout wt [nb, k32, ldw] -> (wh [nb, ldw, k64], k64, nrows)"""
    nb, k32, ldw = wt.shape
    k64 = _ceil(k32, 64)
    return torch.zeros((nb, ldw, k64), dtype=torch.float16, device=wt.device), k64, ldw


def prep_weight_h(wt):
    """K-major fp32 operand of conv.prep_weight -> N-major half operand (one launch, geometry in the kernel arguments: legal
    inside a graph capture).  Returns (wh, Kpad64, nrows)."""
    nb, k32, ldw = wt.shape
    k64 = _ceil(k32, 64)
    wh = torch.empty((nb, ldw, k64), dtype=torch.float16, device=wt.device)       # every element is written (zero padding included)
    lib.check_device(wt, wh)
    lib.call("fsv_hconv_prep_weight_one", lib.ptr(wt), lib.ptr(wh), k32, ldw, ldw, k64, nb, lib.stream_ptr())
    return wh, k64, ldw


# ------------------------------------------------------------------------------------------------ gather-GEMM
def eligible(cin, per_sample=False):
    """layers the half kernels take: input channels a multiple of 8 (16-byte fragments)"""
    return cin % 8 == 0




def gather_gemm_h(x, wh, kpad, nrows, cout, oh, ow, ty, tx, sy, sx, bias=None, res=None, act=ACT_NONE, scale=1.0,
                  per_sample=False, out=None, place=None, out_half=True, force_tile=-1, force_split=0, wscale=None, stats=None):
    """conv.gather_gemm on half operands: x NHWC half (Cin % 8 == 0), wh N-major half [nb, nrows, kpad]; `out` half
    (out_half) or fp32; res half or fp32 (its dtype says which).  stats: as conv.gather_gemm."""
    x = to_nhwc(x)
    if x.dtype != torch.float16:
        raise ValueError("gather_gemm_h wants a half tensor")
    n, cin, h, w = x.shape
    if place is None:
        out_h, out_w, osy, osx, ooy, oox = oh, ow, 1, 1, 0, 0
    else:
        out_h, out_w, osy, osx, ooy, oox = place
    if out is None:
        out = (empty_nhwc_h if out_half else empty_nhwc)(n, cout, out_h, out_w, x)
    else:
        out_half = out.dtype == torch.float16
        conv.drop_half_side(out)          # (written through its raw pointer: a half copy beside it would go stale)
    if res is not None:
        res = to_nhwc(res)
    lib.check_device(x, wh, bias, res, out, wscale)
    w_bs = wh.shape[-2] * wh.shape[-1] if per_sample else 0
    b_bs = 0
    if per_sample and bias is not None:
        if bias.dim() == 2 and bias.stride(1) == 1:
            b_bs = bias.stride(0)
        else:
            bias = bias.contiguous()
            b_bs = cout
    d = HConvDesc()
    d.inp, d.wt, d.bias, d.res, d.out, d.wscale = (t.data_ptr() if t is not None else None for t in (x, wh, bias, res, out, wscale))
    d.N, d.H, d.W, d.Cin, d.OH, d.OW, d.Cout, d.ntaps = n, h, w, cin, oh, ow, cout, len(ty)
    for i, (a, b) in enumerate(zip(ty, tx)):
        d.ty[i], d.tx[i] = a, b
    d.sy, d.sx, d.outH, d.outW, d.osy, d.osx, d.ooy, d.oox = sy, sx, out_h, out_w, osy, osx, ooy, oox
    d.Kpad, d.nrows = kpad, nrows
    d.per_sample, d.act, d.accumulate, d.scale = (1 if per_sample else 0), act, 0, float(scale)
    d.out_h, d.res_h = (1 if out_half else 0), (1 if (res is not None and res.dtype == torch.float16) else 0)
    d.force_tile, d.force_split = force_tile, force_split
    d.w_bstride, d.b_bstride = w_bs, b_bs
    mz = oh * ow if per_sample else n * oh * ow
    nsamp = n if per_sample else 1
    nchunks = (len(ty) * cin + 63) // 64
    dense = place is None
    flops = 2.0 * n * oh * ow * cout * cin * len(ty)
    keep = [x, wh, bias, res, out, wscale]
    info = None
    if _launch_hook is not None:
        info = dict(x=x, wh=wh, kpad=kpad, nrows=nrows, cout=cout, oh=oh, ow=ow, ty=list(ty), tx=list(tx), sy=sy, sx=sx, bias=bias,
                    res=res, act=act, scale=scale, per_sample=per_sample, place=place, wscale=wscale, out=out)
    grp = conv._active_group()
    if grp is not None and force_tile < 0 and force_split == 0:
        label = 'fsv_hconv_kernel'
        grp.hconvs.append((d, label, flops, keep, info))
        return out
    # split-K plans of a half output accumulate in an fp32 workspace
    ws = None
    can_split = dense and act != ACT_DLRELU
    tile, nsplit = planned(mz, cout, nchunks, nsamp, force_tile, force_split, can_split)
    if nsplit > 1 and can_split and out_half:
        ws = torch.empty(n * out_h * out_w * cout, dtype=torch.float32, device=x.device)
        d.ws = ws.data_ptr()
        keep.append(ws)
    part = None
    if (stats is not None and dense and not per_sample and force_tile < 0 and force_split == 0 and conv.stats_enabled()):
        groups = int(stats['groups'])
        part = conv.stats_arena(x.device).take(groups * conv.STATS_SLOTS * cout * 2)
        d.stats_prezeroed = 1 if part is not None else 0
        if part is None:
            part = torch.empty(groups * conv.STATS_SLOTS * cout * 2, dtype=torch.float64, device=x.device)
        d.stats, d.stats_groups, d.stats_slots = part.data_ptr(), groups, conv.STATS_SLOTS
        keep.append(part)
    produced = c_i(0)
    label = 'fsv_hconv_kernel<%s>' % H_TILE_NAMES[tile & 15]
    if profile.detail():
        label += ' M%d N%d K%d z%d split%d' % (mz, cout, nchunks * 64, nsamp, nsplit)

    def go(d=d, keep=keep):
        lib.call("fsv_hconv_gather", ctypes.byref(d), 1, ctypes.byref(produced), lib.stream_ptr())
    with profile.scope(label, flops, replay=go):
        go()
    if part is not None and produced.value:
        stats['part'], stats['slots'] = part, conv.STATS_SLOTS
    if info is not None:
        _launch_hook('conv', info)
    return out


def issue_group(items):
    """one grouped launch over the (HConvDesc, label, flops, keep) entries a conv.launch_group collected"""
    if not items:
        return
    for i in range(0, len(items), 64):
        chunk = items[i:i + 64]
        arr = (HConvDesc * len(chunk))(*[it[0] for it in chunk])
        keep = [it[3] for it in chunk]
        flops = sum(it[2] for it in chunk)

        def go(arr=arr, n=len(chunk), keep=keep):
            lib.call("fsv_hconv_gather", ctypes.cast(arr, c_p), n, None, lib.stream_ptr())
        with profile.scope('fsv_hconv_group_kernel' if len(chunk) > 1 else chunk[0][1], flops, replay=go):
            go()
        if _launch_hook is not None:
            for it in chunk:
                if it[4] is not None:
                    _launch_hook('conv', it[4])


def conv_forward_h(x, wh, kpad, nrows, cout, geom, **kw):
    n, cin, h, w = x.shape
    oh, ow = geom.out_hw(h, w)
    return gather_gemm_h(x, wh, kpad, nrows, cout, oh, ow, geom.ty, geom.tx, geom.stride, geom.stride, **kw)


def conv_dgrad_h(dout, layouts, geom, in_hw, cin, scale=None, per_sample=False, out_half=True, act=ACT_NONE, res=None):
    """Data gradient from half dout.  layouts: per geom.dgrad_classes element (wh, kpad, nrows) or None for empty classes."""
    dout = to_nhwc(dout)
    n, cout, oh, ow = dout.shape
    h, wd = in_hw
    s = geom.stride
    if s == 1:
        c = geom.dgrad_classes[0]
        wh, kpad, nrows = layouts[0]
        return gather_gemm_h(dout, wh, kpad, nrows, cin, h, wd, c['ty'], c['tx'], 1, 1, per_sample=per_sample, wscale=scale,
                             out_half=out_half, act=act, res=res)
    subs = [((h - c['py'] + s - 1) // s, (wd - c['px'] + s - 1) // s) for c in geom.dgrad_classes]
    plain = all(c['khs'] and sh > 0 and sw > 0 for c, (sh, sw) in zip(geom.dgrad_classes, subs))
    mk = empty_nhwc_h if out_half else empty_nhwc
    dx = mk(n, cin, h, wd, dout)
    if not plain:
        dx.zero_()
    # every output pixel belongs to exactly one parity class and the placed launches never split K: plain stores, one grid
    with conv.launch_group(True):
        for k, (c, (sub_h, sub_w)) in enumerate(zip(geom.dgrad_classes, subs)):
            if sub_h <= 0 or sub_w <= 0 or not c['khs']:
                continue
            wh, kpad, nrows = layouts[k]
            gather_gemm_h(dout, wh, kpad, nrows, cin, sub_h, sub_w, c['ty'], c['tx'], 1, 1, per_sample=per_sample, out=dx,
                          place=(h, wd, s, s, c['py'], c['px']), wscale=scale)
    return dx


def wgrad_eligible(cin, cout, oh, ow):
    return cin % 8 == 0 and cout % 8 == 0 and 64 // ow + (1 if 64 % ow else 0) <= oh


def conv_wgrad_h(x, dout, geom, per_sample=False, force_split=0, arena=None, force_tile=0):
    """Weight gradient dwt[(tap, ci)][co] (fp32, the K-major layout grad_finalize consumes) from half x / half dout."""
    x, dout = to_nhwc(x), to_nhwc(dout)
    n, cin, h, w = x.shape
    _, cout, oh, ow = dout.shape
    kpad = _ceil(geom.ntaps * cin, 32)
    ldw = _ceil(cout, 32)
    nb = n if per_sample else 1
    dwt = arena.take(kpad * ldw) if (arena is not None and not per_sample) else None
    prezeroed = dwt is not None
    if dwt is None:
        dwt = torch.empty((nb, kpad, ldw), dtype=torch.float32, device=x.device)
    lib.check_device(x, dout, dwt)
    label = 'fsv_hconv_wgrad_kernel'
    if profile.detail():
        label += ' Kdim%d N%d pix%d z%d' % (geom.ntaps * cin, cout, (oh * ow) if per_sample else n * oh * ow, nb)
    args = (lib.ptr(x), lib.ptr(dout), lib.ptr(dwt), n, h, w, cin, oh, ow, cout, geom.ntaps, lib.int_array(geom.ty),
            lib.int_array(geom.tx), geom.stride, geom.stride, ldw, kpad, kpad * ldw, 1 if per_sample else 0, force_split,
            1 if prezeroed else 0, force_tile, lib.stream_ptr())
    keep = (x, dout, dwt)
    with profile.scope(label, 2.0 * n * oh * ow * cout * cin * geom.ntaps,
                       replay=lambda args=args, keep=keep: lib.call("fsv_hconv_wgrad", *args)):
        lib.call("fsv_hconv_wgrad", *args)
    if _launch_hook is not None:
        _launch_hook('wgrad', dict(x=x, dout=dout, geom=geom, per_sample=per_sample, dwt=dwt, kpad=kpad, ldw=ldw))
    return dwt
