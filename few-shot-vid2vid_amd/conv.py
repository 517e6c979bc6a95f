"""Convolution / linear / per-sample "batch_conv" operators on the HIP gather-GEMM kernels.

Mirrors the operator surface the reference reaches through ``F.conv2d`` (models/networks/architecture.py:22-27,
generator.py:541-572, discriminator.py:67-88), ``nn.Linear`` (generator.py:103-110) and ``batch_conv``
(models/networks/base_network.py:56-71).  Tensors keep the reference's logical NCHW shape but live in
channels-last memory (NHWC), which is the layout the kernels in csrc/conv_igemm.hip read and write.
"""
import ctypes
import os
import threading

import torch

from . import lib, profile

ACT_NONE, ACT_LRELU, ACT_TANH, ACT_SIGMOID, ACT_RELU, ACT_LRELU01, ACT_DLRELU = 0, 1, 2, 3, 4, 5, 6

# Arithmetic of the GEMM operands (csrc/conv_np.hip).  0: exact fp32 MFMA (default, the measured path);
# 1: "f16" - operands rounded to half while staged through LDS, fp32 accumulate (the reference's --amp contract, needs the
# loss scale of flat.FlatAdam); 2: "bf16x3" - operands split into two bf16 terms, three MFMAs, fp32 accumulate.
MFMA_F32, MFMA_F16, MFMA_BF16X3 = 0, 1, 2
_MFMA_NAMES = {'': 0, '0': 0, 'f32': 0, 'fp32': 0, '1': 1, 'f16': 1, 'fp16': 1, '2': 2, 'bf16x3': 2}
_mfma_mode = _MFMA_NAMES[os.environ.get('FSV_MFMA_MODE', '').lower()]


def set_mfma_mode(mode):
    """Select the operand arithmetic of every convolution / linear / batch_conv GEMM launched from this process (like
    apex's amp.initialize, models/models.py:22-26, this is process-wide).  Returns the previous mode."""
    global _mfma_mode
    prev = _mfma_mode
    _mfma_mode = _MFMA_NAMES[str(mode).lower()] if not isinstance(mode, int) else int(mode)
    if _mfma_mode not in (0, 1, 2):
        _mfma_mode = prev
        raise ValueError("unknown MFMA operand mode %r" % (mode,))
    return prev


def mfma_mode():
    return _mfma_mode


# Round 4: in the f16 mode the operators (ops._ConvFn and friends) hand HALF activations to these functions, which then run the
# half-precision kernels of csrc/conv_h.hip (activations and weights 16-bit in HBM, hconv.py); an fp32 activation tensor stays on
# the exact fp32 kernels.  FSV_HCONV=0 / set_h_kernels(False) restores round 2's behaviour - fp32 tensors narrowed while staged
# (csrc/conv_np.hip) - for A/B runs and for the tests that pin those kernels to the definition.
_h_kernels = os.environ.get('FSV_HCONV', '1') == '1'


def set_h_kernels(on):
    global _h_kernels
    prev, _h_kernels = _h_kernels, bool(on)
    return prev


def h_kernels():
    """True when the f16 mode runs on the half-precision kernels (activations cast to / kept in half by the operators)"""
    return _h_kernels and _mfma_mode == MFMA_F16


def narrow_staging_mode():
    """operand mode of the staging-time narrowing kernels (csrc/conv_np.hip) for an fp32 activation tensor: bf16x3 always,
    f16 only with the half kernels switched off"""
    if _mfma_mode == MFMA_BF16X3 or (_mfma_mode == MFMA_F16 and not _h_kernels):
        return _mfma_mode
    return 0


def half_twin(wt):
    """(wh, Kpad64, nrows): the N-major half operand (hconv.py) of a K-major fp32 layout.  Layouts owned by a LayoutCache get a
    persistent twin that the cache rewrites with the layout itself once per optimiser step; any other (a per-call
    re-arrangement of generated weights) is converted now and the twin kept on the tensor."""
    t = getattr(wt, '_fsv_half', None)
    if t is None:
        owner = getattr(wt, '_fsv_owner', None)
        if owner is not None:
            t = owner.add_half(wt)
        else:
            from . import hconv
            t = hconv.prep_weight_h(wt)
        wt._fsv_half = t
    return t


def drop_half_side(t):
    """A kernel is about to write into the EXISTING tensor t through its raw pointer (`out=` / `place=` / accumulate paths): that
    does not bump t._version, so a half copy a producer left beside it (hconv.half_side_output: `_fsv_h16`, validated by version and
    shape only) would silently go stale - drop it (round-4 advisor)."""
    if getattr(t, '_fsv_h16', None) is not None:
        try:
            del t._fsv_h16
        except AttributeError:
            t._fsv_h16 = None


def to_nhwc(x):
    """Return a tensor with the same logical NCHW shape whose memory is dense NHWC."""
    if x.dim() != 4:
        raise ValueError("expected a 4-D NCHW tensor")
    if not x.permute(0, 2, 3, 1).is_contiguous():
        x = x.contiguous(memory_format=torch.channels_last)
        if not x.permute(0, 2, 3, 1).is_contiguous():   # degenerate shapes (C==1 or H==W==1)
            x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    return x


def empty_nhwc(n, c, h, w, like):
    return torch.empty((n, h, w, c), dtype=torch.float32, device=like.device).permute(0, 3, 1, 2)


def zeros_nhwc(n, c, h, w, like):
    return torch.zeros((n, h, w, c), dtype=torch.float32, device=like.device).permute(0, 3, 1, 2)


def _ceil(a, b):
    return (a + b - 1) // b * b


class Geom:
    """Static geometry of one convolution (kernel, stride, padding) and its tap tables."""

    _cache = {}

    def __new__(cls, kh, kw, stride, pad):
        key = (kh, kw, stride, pad)
        g = cls._cache.get(key)
        if g is None:
            g = super().__new__(cls)
            g._init(kh, kw, stride, pad)
            cls._cache[key] = g
        return g

    def _init(self, kh, kw, stride, pad):
        self.kh, self.kw, self.stride, self.pad = kh, kw, stride, pad
        self.ntaps = kh * kw
        self.khs = [i for i in range(kh) for _ in range(kw)]
        self.kws = [j for _ in range(kh) for j in range(kw)]
        self.ty = [i - pad for i in self.khs]
        self.tx = [j - pad for j in self.kws]
        # data-gradient tap classes: for stride s the input pixel (y, x) with ((y + pad) % s, (x + pad) % s) == (a, b)
        # only receives taps kh = a (mod s), kw = b (mod s); the source row in dout is (y + pad - kh) / s.
        self.dgrad_classes = []
        s = stride
        for py in range(s):
            for px in range(s):
                a, b = (py + pad) % s, (px + pad) % s
                khs = [i for i in range(kh) if i % s == a]
                kws = [j for j in range(kw) if j % s == b]
                taps = [(i, j) for i in khs for j in kws]
                # output pixel (s*y' + py, s*x' + px) reads dout[y' + (py + pad - kh) / s]
                ty = [(py + pad - i) // s for (i, j) in taps]
                tx = [(px + pad - j) // s for (i, j) in taps]
                self.dgrad_classes.append(dict(py=py, px=px, khs=[t[0] for t in taps], kws=[t[1] for t in taps],
                                               ty=ty, tx=tx))

    def out_hw(self, h, w):
        return ((h + 2 * self.pad - self.kh) // self.stride + 1, (w + 2 * self.pad - self.kw) // self.stride + 1)


def prep_weight(w, mode, geom, khs=None, kws=None, scale=None, out=None):
    """OIHW weights (optionally a leading per-sample batch dim) -> K-major GEMM operand.

    mode 0: forward  wt[(tap, ci)][co];  mode 1: dgrad  wt[(tap, co)][ci].  Returns (wt, Kpad, ldw).
    """
    batched = w.dim() == 5
    nb = w.shape[0] if batched else 1
    cout, cin, kh, kw = w.shape[-4:]
    khs = geom.khs if khs is None else khs
    kws = geom.kws if kws is None else kws
    ntaps = len(khs)
    rowlen, ncols = (cout, cin) if mode == 1 else (cin, cout)
    kpad = _ceil(max(ntaps * rowlen, 1), 32)
    ldw = _ceil(ncols, 32)
    # per-sample (generated) weights are views into the weight-generating FC's output: each sample's [Cout][Cin][k][k]
    # block is contiguous, only the sample stride is larger - read them in place instead of copying
    w_bstride = cout * cin * kh * kw
    if batched and w[0].is_contiguous() and w.stride(0) >= w_bstride:
        w_bstride = w.stride(0)
    else:
        w = w.contiguous()
    lib.check_device(w, scale)
    if out is None:
        out = torch.empty((nb, kpad, ldw), dtype=torch.float32, device=w.device)
    if ntaps == 0:
        out.zero_()
        return out, kpad, ldw
    lib.call("fsv_prep_weight", lib.ptr(w), lib.ptr(out), lib.ptr(scale), mode, nb, cout, cin, kh, kw, ntaps,
             lib.int_array(khs), lib.int_array(kws), kpad, ldw, w_bstride, kpad * ldw, lib.stream_ptr())
    return out, kpad, ldw


def unprep_weight_grad(dwt, w_shape, geom, scale=None, out=None):
    """Inverse of mode-0 prep for gradients: dwt[(tap, ci)][co] -> OIHW (optionally batched).  With `out` (a
    contiguous buffer of w_shape elements, e.g. a slice of the flat gradient buffer) the result is ADDED into it."""
    batched = len(w_shape) == 5
    nb = w_shape[0] if batched else 1
    cout, cin, kh, kw = w_shape[-4:]
    dw = out if out is not None else torch.empty(w_shape, dtype=torch.float32, device=dwt.device)
    kpad, ldw = dwt.shape[-2], dwt.shape[-1]
    lib.call("fsv_prep_weight", lib.ptr(dw), lib.ptr(dwt), lib.ptr(scale), 3 if out is not None else 2, nb, cout, cin, kh, kw, geom.ntaps,
             lib.int_array(geom.khs), lib.int_array(geom.kws), kpad, ldw, cout * cin * kh * kw, kpad * ldw,
             lib.stream_ptr())
    return dw


# ------------------------------------------------------------------------------------------------ grouped launches
class ConvDesc(ctypes.Structure):
    """include/fsv2v.h fsv_conv_desc"""
    _fields_ = ([(k, ctypes.c_void_p) for k in ('inp', 'wt', 'bias', 'res', 'out', 'wscale')] +
                [(k, ctypes.c_int) for k in ('N', 'H', 'W', 'Cin', 'OH', 'OW', 'Cout', 'ntaps')] +
                [('ty', ctypes.c_int * 16), ('tx', ctypes.c_int * 16)] +
                [(k, ctypes.c_int) for k in ('sy', 'sx', 'outH', 'outW', 'osy', 'osx', 'ooy', 'oox', 'ldw',
                                             'per_sample', 'act', 'accumulate')] +
                [('scale', ctypes.c_float), ('w_bstride', ctypes.c_longlong), ('b_bstride', ctypes.c_longlong)])


class WgradDesc(ctypes.Structure):
    """include/fsv2v.h fsv_wgrad_desc"""
    _fields_ = ([(k, ctypes.c_void_p) for k in ('inp', 'dout', 'dwt')] +
                [(k, ctypes.c_int) for k in ('N', 'H', 'W', 'Cin', 'OH', 'OW', 'Cout', 'ntaps')] +
                [('ty', ctypes.c_int * 16), ('tx', ctypes.c_int * 16)] +
                [(k, ctypes.c_int) for k in ('sy', 'sx', 'ldw', 'Kpad', 'per_sample', 'reserved')] +
                [('w_bstride', ctypes.c_longlong)])


lib.register_sigs({
    "fsv_conv_gather_group": [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p],
    "fsv_conv_wgrad_group": [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p],
    "fsv_conv_group_plan": [ctypes.POINTER(ctypes.c_int)] * 4 + [ctypes.c_int, ctypes.POINTER(ctypes.c_int)],
    "fsv_conv_gather_fwd_stats": [ctypes.c_void_p] * 5 + [ctypes.c_int] * 8 + [ctypes.POINTER(ctypes.c_int)] * 2 +
                                 [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                       ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p,
                                                       ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p],
})

GROUP_LIMIT = 64          # FSV_GROUP_LIMIT (csrc/conv_igemm.hip)
_TILE_DIMS = {0: (128, 128), 1: (128, 64), 2: (128, 32), 4: (64, 64), 9: (64, 128)}
_tls = threading.local()  # autograd runs backward on its own thread: the active group is per thread


def group_enabled():
    return os.environ.get('FSV_CONV_GROUPS', '1') == '1'


def stats_enabled():
    """FSV_CONV_STATS=0: in-box A/B switch - normalisation statistics from their own read pass instead of the producing
    convolution's epilogue"""
    # the fixed-order mode (FSV_DETERMINISTIC=1) takes the read pass: the epilogue adds its partial sums with fp64 atomics
    return os.environ.get('FSV_CONV_STATS', '1') == '1' and os.environ.get('FSV_DETERMINISTIC', '0') != '1'


class launch_group:
    """`with launch_group():` - the gather-GEMM / weight-gradient launches issued inside are collected and go to the device as
    ONE grid per kind when the block ends (csrc/conv_igemm.hip fsv_conv_igemm_group_kernel).  The caller guarantees that the
    launches are independent (none reads what another writes) and that nobody reads their outputs before the block ends;
    fills / copies issued inside run BEFORE the grouped launch (stream order).  Launches the grouped kernels cannot take
    (narrow-operand modes, forced tiles / splits) are issued at once.  Nested blocks join the outer one."""

    def __init__(self, enabled=True, force_tile=-1):
        self.on = enabled and group_enabled()
        self.convs, self.wgrads, self.hconvs = [], [], []
        self.outer = None
        self.force_tile = force_tile          # tests: tile id for the grouped gather-GEMM grid (-1: the library's plan)

    def __enter__(self):
        if self.on:
            self.outer = getattr(_tls, 'group', None)
            if self.outer is None:
                _tls.group = self
        return self

    def __exit__(self, et, ev, tb):
        if self.on and self.outer is None:
            _tls.group = None
            if et is None:
                self.flush()
        return False

    def flush(self):
        for kind, items in (('conv', self.convs), ('wgrad', self.wgrads)):
            for i in range(0, len(items), GROUP_LIMIT):
                self._issue(kind, items[i:i + GROUP_LIMIT], self.force_tile)
        self.convs, self.wgrads = [], []
        if self.hconvs:             # half-precision problems (hconv.gather_gemm_h): their own grid
            from . import hconv
            items, self.hconvs = self.hconvs, []
            hconv.issue_group(items)

    @staticmethod
    def _issue(kind, items, force_tile=-1):
        if not items:
            return
        if len(items) == 1:
            entry, args = items[0][1], items[0][2]
            with profile.scope(items[0][3], items[0][4], replay=lambda: lib.call(entry, *args)):
                lib.call(entry, *args)
            return
        Desc = ConvDesc if kind == 'conv' else WgradDesc
        arr = (Desc * len(items))(*[it[0] for it in items])
        if kind == 'conv':
            gargs = (ctypes.cast(arr, ctypes.c_void_p), len(items), force_tile, lib.stream_ptr())
            name = "fsv_conv_gather_group"
            label = 'fsv_conv_igemm_group_kernel'
            if profile.enabled() or _plan_log is not None:
                tile = force_tile if force_tile >= 0 else group_planned([(it[0].OH * it[0].OW * (1 if it[0].per_sample else it[0].N), it[0].Cout,
                                       (it[0].ntaps * it[0].Cin + 31) // 32, it[0].N if it[0].per_sample else 1) for it in items])
                vec4 = all(it[0].Cin % 4 == 0 for it in items)
                label = 'fsv_conv_igemm_group_kernel<%s,V%d>' % (profile.TILE_NAMES[tile], 4 if vec4 else 1)
                if _plan_log is not None:
                    _plan_log.append(('group', tile, vec4, len(items)))
        else:
            gargs = (ctypes.cast(arr, ctypes.c_void_p), len(items), lib.stream_ptr())
            name = "fsv_conv_wgrad_group"
            label = 'fsv_conv_wgrad_group_kernel<64x64,V4>'
        keep = [it[5] for it in items]
        flops = sum(it[4] for it in items)

        def go(arr=arr, keep=keep):
            rc = lib.call_status(name, *gargs)
            if rc == -2:                     # FSV_ERR_UNSUPPORTED, nothing was launched: one by one
                for it in items:
                    lib.call(it[1], *it[2])
            elif rc != 0:
                raise lib.FsvError("%s failed with fsv_status %d" % (name, rc))
        with profile.scope(label, flops, replay=go):
            go()


def group_planned(shapes):
    """tile id of a grouped gather-GEMM launch over problems (Mz, Cout, nchunks, nsamp)"""
    n = len(shapes)
    cols = [lib.int_array([s[k] for s in shapes]) for k in range(4)]
    tile = ctypes.c_int(0)
    lib.call("fsv_conv_group_plan", cols[0], cols[1], cols[2], cols[3], n, ctypes.byref(tile))
    return tile.value


def _active_group():
    return getattr(_tls, 'group', None)


STATS_SLOTS = 32      # partial-sum slots per (group, channel) of the statistics a convolution leaves for its normalisation


class _StatsArena:
    """Zeroed fp64 storage for the statistics partials of ONE forward pass (one memset per chunk and pass instead of one in front
    of every convolution that feeds a normalisation: 80 fills of ~5 us per step).  `begin()` - called by the model-level entry
    points before anything of the pass is launched - zeroes the storage; `take()` hands out slices until it runs out (first
    pass of a new configuration: the caller falls back to a per-call buffer and the arena grows at the next begin).  Partials
    are consumed right behind their producer, on the producer's stream.

    The storage only ever GROWS BY APPENDING CHUNKS and no chunk is ever released: a hipGraph captured earlier has the raw
    addresses of the chunks it saw baked into its memsets, the fp64 atomics of the convolution epilogues and
    fsv_norm_stats_finish, and keeps replaying into them (round-3 advisor: a buffer that was re-allocated when a later eager
    pass needed more slots left those graphs writing into freed caching-allocator memory).  Slices are handed out in the same
    order every pass, so a replayed graph and an eager pass of the same configuration use the same slots."""

    def __init__(self, device):
        self.device, self.chunks, self.cur, self.off, self.need, self.high, self.depth = device, [], 0, 0, 0, 0, 0

    def capacity(self):
        return sum(c.numel() for c in self.chunks)

    def begin(self):
        self.depth += 1
        if self.depth > 1:            # a nested entry point (the generator inside the model): the pass is already open
            return
        capturing = self.device.type == 'cuda' and torch.cuda.is_current_stream_capturing()
        if self.need > self.high:
            self.high = self.need
        short = self.high - self.capacity()
        if short > 0 and not capturing:
            # (allocated outside any capture: eager allocations are never recycled while the chunk list holds them)
            self.chunks.append(torch.zeros(max(short, 4096), dtype=torch.float64, device=self.device))
        for c in self.chunks:
            c.zero_()
        self.cur, self.off, self.need = 0, 0, 0

    def end(self):
        self.depth = max(self.depth - 1, 0)

    def take(self, n):
        n = (n + 15) // 16 * 16
        if self.depth == 0:           # a bare op call outside a pass: per-call buffer, and nothing to plan for
            return None
        self.need += n
        while self.cur < len(self.chunks) and self.off + n > self.chunks[self.cur].numel():
            # (the tail of a chunk that cannot hold the request stays unused; `need` counts requests, so the next growth
            # may over-allocate by those tails - a few KB)
            self.need += self.chunks[self.cur].numel() - self.off
            self.cur, self.off = self.cur + 1, 0
        if self.cur >= len(self.chunks):
            return None
        out = self.chunks[self.cur][self.off:self.off + n]
        self.off += n
        return out


_stats_arenas = {}


def stats_arena(device):
    a = _stats_arenas.get(device)
    if a is None:
        a = _stats_arenas[device] = _StatsArena(device)
    return a


class stats_pass:
    """`with stats_pass(device):` around one forward pass (Vid2VidModel.forward, a bare generator / discriminator call)"""

    def __init__(self, device):
        self.arena = stats_arena(device) if stats_enabled() else None

    def __enter__(self):
        if self.arena is not None:
            self.arena.begin()
        return self

    def __exit__(self, *exc):
        if self.arena is not None:
            self.arena.end()
        return False


def up_foldable(cin, cout, per_sample=False, out_hw=None):
    """can a nearest x2 up-sampling in front of this convolution be folded into the gather (csrc/conv_igemm.hip ConvP::up)?
    The exact-fp32 float4-gather MFMA launches only - forward AND weight gradient (whose float4 kernel walks the output pixels 32
    at a time and wants 32 / OW + 1 <= OH: tiny maps take its scalar twin, which has no fold); FSV_UP_FOLD=0: materialise (in-box
    A/B)."""
    if out_hw is not None and 32 // out_hw[1] + 1 > out_hw[0]:
        return False
    return (os.environ.get('FSV_UP_FOLD', '1') == '1' and cin % 4 == 0 and cout > 4 and cout % 4 == 0 and not per_sample and
            _mfma_mode == MFMA_F32 and _active_group() is None)


def gather_gemm(x, wt, ldw, cout, oh, ow, ty, tx, sy, sx, bias=None, res=None, act=ACT_NONE, scale=1.0,
                per_sample=False, out=None, place=None, accumulate=False, force_tile=-1, force_split=0, wscale=None,
                stats=None, up=False):
    """out[n, oy, ox, :] = act((sum_taps x[n, oy*sy+ty, ox*sx+tx, :] @ wt[tap]) + bias) * scale) + res.

    stats: None, or a dict with 'groups' (1: BatchNorm, n: InstanceNorm) - the launch then also leaves the per-channel sums of
    its output (csrc/conv_igemm.hip ConvP::stats) and fills stats['part'] (fp64 partials), stats['slots']; when the launch
    cannot (K-split plan, scalar gather, narrow-operand modes, inside a launch group) 'part' stays absent."""
    x = to_nhwc(x)
    if x.dtype == torch.float16:
        # half activations: the half-precision kernels (fp32 output unless `out` is a half tensor)
        if up:
            raise ValueError("a folded up-sampling needs the exact-fp32 kernels (conv.up_foldable)")
        from . import hconv
        wh, k64, nrows = half_twin(wt)
        if accumulate:
            raise ValueError("half launches store plainly (placed outputs never split K)")
        return hconv.gather_gemm_h(x, wh, k64, nrows, cout, oh, ow, ty, tx, sy, sx, bias=bias, res=res, act=act, scale=scale,
                                   per_sample=per_sample, out=out, place=place, out_half=False, wscale=wscale, stats=stats)
    n, cin, h, w = x.shape
    if up:            # x is stored at half the size the convolution sees (the nearest x2 up-sampling is folded into the gather)
        if x.dtype != torch.float32 or place is not None or accumulate or per_sample or _active_group() is not None:
            raise ValueError("a folded up-sampling needs a plain fp32 launch (conv.up_foldable)")
        h, w = 2 * h, 2 * w
    if place is None:
        out_h, out_w, osy, osx, ooy, oox = oh, ow, 1, 1, 0, 0
    else:
        out_h, out_w, osy, osx, ooy, oox = place
    if out is None:
        out = empty_nhwc(n, cout, out_h, out_w, x)
    else:
        drop_half_side(out)
    if res is not None:
        res = to_nhwc(res)
    lib.check_device(x, wt, bias, res, out, wscale)
    _np_mode = narrow_staging_mode()
    w_bs = wt.shape[-2] * wt.shape[-1] if per_sample else 0
    b_bs = 0
    if per_sample and bias is not None:
        if bias.dim() == 2 and bias.stride(1) == 1:
            b_bs = bias.stride(0)               # rows of the FC output: read in place
        else:
            bias = bias.contiguous()
            b_bs = cout
    head = (lib.ptr(x), lib.ptr(wt), lib.ptr(bias), lib.ptr(res), lib.ptr(out),
            n, h, w, cin, oh, ow, cout, len(ty), lib.int_array(ty), lib.int_array(tx), sy, sx,
            out_h, out_w, osy, osx, ooy, oox, ldw, w_bs, b_bs, 1 if per_sample else 0,
            act, float(scale), force_tile, force_split, 1 if accumulate else 0, lib.ptr(wscale))
    if _plan_log is not None:
        _plan_log.append(planned(oh * ow if per_sample else n * oh * ow, cout, (len(ty) * cin + 31) // 32, n if per_sample else 1,
                                 force_tile, force_split) + (cin % 4 == 0,))
    entry = "fsv_conv_gather_fwd"
    if _np_mode and cin % 4 == 0:                 # scalar-gather layers (3-channel images, labels) stay on the fp32 kernel
        entry = "fsv_conv_gather_fwd_np"
    grp = _active_group()
    split_ws = None
    if entry == "fsv_conv_gather_fwd" and grp is None and place is None and not accumulate and ordered_split():
        # ordered split-K: every split stores its own copy of the output, the finishing pass sums them in ascending order
        ns = planned(oh * ow if per_sample else n * oh * ow, cout, (len(ty) * cin + 31) // 32, n if per_sample else 1, force_tile,
                     force_split)[1]
        if ns > 1 and act != ACT_DLRELU:
            split_ws = torch.empty(ns * out.numel(), dtype=torch.float32, device=x.device)
    # the ordered-split workspace is an explicit (nullable) argument of the call that may use it (include/fsv2v.h)
    ws_args = (lib.ptr(split_ws), split_ws.numel() if split_ws is not None else 0)
    if entry == "fsv_conv_gather_fwd_np":
        if up:
            raise ValueError("a folded up-sampling needs the exact-fp32 kernels (conv.up_foldable)")
        args = head + (_np_mode, lib.stream_ptr())
    else:
        args = head + ws_args + (1 if up else 0, lib.stream_ptr())
    if (stats is not None and grp is None and entry == "fsv_conv_gather_fwd" and place is None and not per_sample
            and not accumulate and force_tile < 0 and force_split == 0 and cin % 4 == 0 and stats_enabled()):
        groups = int(stats['groups'])
        part = stats_arena(x.device).take(groups * STATS_SLOTS * cout * 2)
        prezeroed = part is not None
        if part is None:
            part = torch.empty(groups * STATS_SLOTS * cout * 2, dtype=torch.float64, device=x.device)
        produced = ctypes.c_int(0)
        sargs = (lib.ptr(x), lib.ptr(wt), lib.ptr(bias), lib.ptr(res), lib.ptr(out), n, h, w, cin, oh, ow, cout, len(ty),
                 lib.int_array(ty), lib.int_array(tx), sy, sx, ldw, act, float(scale), lib.ptr(wscale), lib.ptr(part), groups,
                 STATS_SLOTS, 1 if prezeroed else 0, ctypes.byref(produced)) + ws_args + (1 if up else 0, lib.stream_ptr())
        label = 'fsv_conv_igemm_kernel'
        if profile.enabled():
            label = profile.conv_label(n * oh * ow, cout, (len(ty) * cin + 31) // 32, 1, True, force_tile, force_split)
        keep = (x, wt, bias, res, out, wscale, part, split_ws)

        def go_stats(sargs=sargs, keep=keep):
            lib.call("fsv_conv_gather_fwd_stats", *sargs)
        with profile.scope(label, 2.0 * n * oh * ow * cout * cin * len(ty), replay=go_stats):
            go_stats()
        if produced.value:
            stats['part'], stats['slots'] = part, STATS_SLOTS
        return out
    if grp is not None and entry == "fsv_conv_gather_fwd" and force_tile < 0 and force_split == 0:
        d = ConvDesc()
        d.inp, d.wt, d.bias, d.res, d.out, d.wscale = (t.data_ptr() if t is not None else None
                                                       for t in (x, wt, bias, res, out, wscale))
        d.N, d.H, d.W, d.Cin, d.OH, d.OW, d.Cout, d.ntaps = n, h, w, cin, oh, ow, cout, len(ty)
        for i, (a, b) in enumerate(zip(ty, tx)):
            d.ty[i], d.tx[i] = a, b
        d.sy, d.sx, d.outH, d.outW, d.osy, d.osx, d.ooy, d.oox, d.ldw = sy, sx, out_h, out_w, osy, osx, ooy, oox, ldw
        d.per_sample, d.act, d.accumulate, d.scale = (1 if per_sample else 0), act, (1 if accumulate else 0), float(scale)
        d.w_bstride, d.b_bstride = w_bs, b_bs
        label = 'fsv_conv_igemm_kernel'
        if profile.enabled():
            label = profile.conv_label(oh * ow if per_sample else n * oh * ow, cout, (len(ty) * cin + 31) // 32,
                                       n if per_sample else 1, cin % 4 == 0, force_tile, force_split)
        grp.convs.append((d, entry, args, label, 2.0 * n * oh * ow * cout * cin * len(ty), (x, wt, bias, res, out, wscale)))
        return out
    if profile.enabled():
        mz = oh * ow if per_sample else n * oh * ow
        label = profile.conv_label(mz, cout, (len(ty) * cin + 31) // 32, n if per_sample else 1, cin % 4 == 0,
                                   force_tile, force_split,
                                   thin=0 if entry.endswith('_np') else _thin_k(cout, cin, len(ty), per_sample, place, accumulate,
                                                                                force_tile, force_split, act))
        if entry.endswith('_np'):
            label = label.replace('fsv_conv_igemm_kernel', 'fsv_np_conv_kernel[%s]' % ('f16' if _np_mode == 1 else 'bf16x3'))
        keep = (x, wt, bias, res, out, wscale, split_ws)          # the replay re-issues the launch on the same buffers

        def go(entry=entry, args=args, keep=keep):
            lib.call(entry, *args)
        with profile.scope(label, 2.0 * n * oh * ow * cout * cin * len(ty), replay=go):
            go()
    else:
        lib.call(entry, *args)
    return out


_tickets = {}
_TICKET_POOL = 1 << 16


def ticket_range(like, n):
    """address of n (rounded up to 64) ZEROED ints for one launch that finishes in its last workgroup (the fused reductions of
    csrc/norm.hip): a ring over a per-device pool, so that launches which may overlap
    (branch streams, neighbouring graph nodes) never share a range; every launch leaves its range zeroed.  None when n does not
    fit the pool (the caller takes its two-launch form)."""
    n = (max(int(n), 1) + 63) // 64 * 64
    if n > _TICKET_POOL // 4:
        return None
    ent = _tickets.get(like.device)
    if ent is None:
        from . import streams
        ent = _tickets[like.device] = [streams.shared(lambda: torch.zeros(_TICKET_POOL, dtype=torch.int32, device=like.device)), 0]
    pool, cur = ent
    if cur + n > _TICKET_POOL:
        cur = 0
    ent[1] = (cur + n) % _TICKET_POOL
    return ctypes.c_void_p(pool.data_ptr() + 4 * cur)


def ordered_split():
    """split-K launches of the fp32 gather-GEMM sum their splits in a fixed order (FSV_ORDERED_SPLIT=0: the atomic form, A/B)"""
    return os.environ.get('FSV_ORDERED_SPLIT', '1') == '1'


def _thin_k(cout, cin, ntaps, per_sample, place, accumulate, force_tile, force_split, act):
    """K of a launch that fsv_conv_gather_fwd hands to the thin-output (vector-ALU) kernels if the size rule agrees
    (csrc/conv_igemm.hip fsv_conv_gather_impl), else 0 - for profiler labels only"""
    q = cin >> 2
    ok = (cout <= 4 and cin % 4 == 0 and cin <= 256 and q & (q - 1) == 0 and ntaps * cin <= 1152 and not per_sample and
          place is None and not accumulate and force_tile < 0 and force_split <= 0 and act != ACT_DLRELU)
    return ntaps * cin if ok else 0


def conv_forward(x, wt_f, ldw, cout, geom, bias=None, res=None, act=ACT_NONE, scale=1.0, per_sample=False,
                 force_tile=-1, force_split=0, wscale=None, stats=None, up=False):
    """up: x stands for its nearest x2 up-sampling (gather_gemm)"""
    n, cin, h, w = x.shape
    oh, ow = geom.out_hw(2 * h, 2 * w) if up else geom.out_hw(h, w)
    return gather_gemm(x, wt_f, ldw, cout, oh, ow, geom.ty, geom.tx, geom.stride, geom.stride, bias=bias, res=res,
                       act=act, scale=scale, per_sample=per_sample, force_tile=force_tile, force_split=force_split,
                       wscale=wscale, stats=stats, up=up)


def planned(mz, cout, nchunks, nsamp, force_tile=-1, force_split=0):
    """(tile id, split-K factor) fsv_conv_gather_fwd will use for this launch (csrc/conv_igemm.hip fsv_conv_plan)"""
    lib.register_sigs({"fsv_conv_plan": [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_int)] * 2})
    tile, nsplit = ctypes.c_int(0), ctypes.c_int(1)
    lib.call("fsv_conv_plan", mz, cout, nchunks, nsamp, force_tile, force_split, ctypes.byref(tile), ctypes.byref(nsplit))
    return tile.value, nsplit.value


def _planned_split(mz, cout, nchunks, nsamp):
    return planned(mz, cout, nchunks, nsamp)[1]


# tests: which (tile, split-K, vectorised gather) combinations a piece of work actually launched
_plan_log = None


def start_plan_log():
    global _plan_log
    _plan_log = []


def stop_plan_log():
    global _plan_log
    log, _plan_log = _plan_log, None
    return log or []


def conv_dgrad(dout, w, geom, in_hw, scale=None, per_sample=False, cached=None, cin=None, out_half=False):
    """Data gradient of a convolution: dx[n, y, x, ci] from dout (NHWC) and OIHW weights.

    cached: optional list (one entry per geom.dgrad_classes element, None for empty classes) of (wt, ldw) un-scaled
    layouts kept by the optimiser's LayoutCache; `scale` (device 1/sigma) is then applied in the GEMM epilogue."""
    dout = to_nhwc(dout)
    n, cout, oh, ow = dout.shape
    cin = w.shape[-3] if cin is None else cin
    h, wd = in_hw
    s = geom.stride
    if dout.dtype == torch.float16:
        from . import hconv
        layouts = []
        for k, c in enumerate(geom.dgrad_classes):
            if not c['khs']:
                layouts.append(None)
            elif cached is not None:
                layouts.append(half_twin(cached[k][0]))
            else:
                wt_k, _, _ = prep_weight(w, 1, geom, c['khs'], c['kws'], None)
                layouts.append(half_twin(wt_k))
        return hconv.conv_dgrad_h(dout, layouts, geom, in_hw, cin, scale=scale, per_sample=per_sample, out_half=out_half)

    def layout(k, c):
        if cached is not None:
            return cached[k][0], cached[k][1], scale
        wt, _, ldw = prep_weight(w, 1, geom, c['khs'], c['kws'], scale)
        return wt, ldw, None
    if s == 1:
        c = geom.dgrad_classes[0]
        wt, ldw, ws = layout(0, c)
        return gather_gemm(dout, wt, ldw, cin, h, wd, c['ty'], c['tx'], 1, 1, per_sample=per_sample, wscale=ws)
    # every output pixel belongs to exactly one parity class: when all classes have taps and none of their launches
    # needs split-K (atomics), each class stores its own pixels and the zero fill of dx is unnecessary
    subs = [((h - c['py'] + s - 1) // s, (wd - c['px'] + s - 1) // s) for c in geom.dgrad_classes]
    plain = all(c['khs'] and sh > 0 and sw > 0 for c, (sh, sw) in zip(geom.dgrad_classes, subs))
    # the parity classes are independent problems (disjoint output pixels): ONE grouped launch instead of s * s small ones.
    # The group splits K (atomics into a zeroed dx) only when all classes together would leave most CUs idle.
    grouped = group_enabled() and not narrow_staging_mode() and cout % 4 == 0
    if grouped:
        live = [(c, sub) for c, sub in zip(geom.dgrad_classes, subs) if sub[0] > 0 and sub[1] > 0 and c['khs']]
        shapes = [((sh * sw) if per_sample else n * sh * sw, cin, (len(c['khs']) * cout + 31) // 32, n if per_sample else 1)
                  for c, (sh, sw) in live]
        bm, bn = _TILE_DIMS[group_planned(shapes)]
        wgs = sum(-(-m // bm) * -(-co // bn) * z for m, co, _, z in shapes)
        plain = plain and not (wgs <= 384 and max(k for _, _, k, _ in shapes) >= 16 and os.environ.get('FSV_DETERMINISTIC') != '1')
    elif plain:
        for c, (sh, sw) in zip(geom.dgrad_classes, subs):
            mz = sh * sw if per_sample else n * sh * sw
            if _planned_split(mz, cin, (len(c['khs']) * cout + 31) // 32, n if per_sample else 1) > 1:
                plain = False
                break
    dx = empty_nhwc(n, cin, h, wd, dout) if plain else zeros_nhwc(n, cin, h, wd, dout)
    with launch_group(grouped):
        for k, (c, (sub_h, sub_w)) in enumerate(zip(geom.dgrad_classes, subs)):
            if sub_h <= 0 or sub_w <= 0 or not c['khs']:
                continue
            wt, ldw, ws = layout(k, c)
            gather_gemm(dout, wt, ldw, cin, sub_h, sub_w, c['ty'], c['tx'], 1, 1, per_sample=per_sample, out=dx,
                        place=(h, wd, s, s, c['py'], c['px']), accumulate=not plain, wscale=ws)
    return dx


def conv_wgrad(x, dout, geom, w_shape, per_sample=False, scale=None, force_split=0, out=None, raw=False, arena=None,
               force_tile=0, up=False):
    """Weight gradient in OIHW layout (batched when per_sample); raw=True returns the GEMM's K-major result
    dwt[(tap, ci)][co] instead (consumed by grad_finalize.GradFinalizer).  up: x stands for its nearest x2 up-sampling (the
    forward pass read it through the folded index, gather_gemm)."""
    x = to_nhwc(x)
    dout = to_nhwc(dout)
    n, cin, h, w = x.shape
    if up:
        if x.dtype != torch.float32 or dout.dtype != torch.float32 or per_sample or narrow_staging_mode():
            raise ValueError("a folded up-sampling needs the exact-fp32 kernels (conv.up_foldable)")
        h, w = 2 * h, 2 * w
    _, cout, oh, ow = dout.shape
    if x.dtype == torch.float16 or dout.dtype == torch.float16:
        from . import hconv
        if hconv.wgrad_eligible(cin, cout, oh, ow) and force_tile == 0:
            dwt = hconv.conv_wgrad_h(hconv.to_half_nhwc(x), hconv.to_half_nhwc(dout), geom, per_sample=per_sample,
                                     force_split=force_split, arena=arena if raw else None)
            if raw:
                return dwt
            return unprep_weight_grad(dwt, tuple(w_shape), geom, scale, out)
        x, dout = hconv.cast(x, torch.float32), hconv.cast(dout, torch.float32)       # geometry outside the half kernels' contract
    kpad = _ceil(geom.ntaps * cin, 32)
    ldw = _ceil(cout, 32)
    nb = n if per_sample else 1
    dwt = arena.take(kpad * ldw) if (arena is not None and raw and not per_sample) else None
    prezeroed = dwt is not None
    if dwt is None:
        dwt = torch.empty((nb, kpad, ldw), dtype=torch.float32, device=x.device)
    lib.check_device(x, dout)
    # tile the launcher picks (csrc/conv_igemm.hip fsv_conv_wgrad): rows = taps * Cin, columns = Cout
    kdim, vec4 = geom.ntaps * cin, cin % 4 == 0
    bn = 32 if cout <= 32 else (64 if cout <= 64 else 128)
    bm = 128
    if vec4 and bn >= 64:
        bm = 32 if kdim <= 32 else (64 if kdim <= 64 else 128)
    if force_tile == 0 and vec4 and cout >= 64 and kdim > 64 and os.environ.get('FSV_WGRAD_PLAN', '1') == '1':
        bm, bn = 64, 64
    label = 'fsv_conv_wgrad_kernel<%dx%d,V%d>' % (bm, bn, 4 if vec4 else 1)
    if (profile.enabled() and cout <= 4 and vec4 and not per_sample and force_tile == 0 and force_split <= 0 and
            geom.ntaps * (cin >> 2) <= 256 and not narrow_staging_mode() and profile.thin_rule(n * oh * ow, kdim)):
        label = 'fsv_conv_thin_wgrad_kernel<Cout%d>' % cout          # csrc/conv_igemm.hip fsv_conv_wgrad: vector-ALU reduction
    if profile.detail():
        label += ' Kdim%d N%d pix%d z%d' % (geom.ntaps * cin, cout, (oh * ow) if per_sample else n * oh * ow, nb)
    _np_mode = narrow_staging_mode()
    narrow = _np_mode and vec4
    if narrow:
        label = 'fsv_np_wgrad_kernel[%s]' % ('f16' if _np_mode == 1 else 'bf16x3')
    wargs = (lib.ptr(x), lib.ptr(dout), lib.ptr(dwt), n, h, w, cin, oh, ow, cout,
             geom.ntaps, lib.int_array(geom.ty), lib.int_array(geom.tx), geom.stride, geom.stride,
             ldw, kpad, kpad * ldw, 1 if per_sample else 0, force_split, 1 if prezeroed else 0, force_tile)
    grp = _active_group()
    if (grp is not None and not narrow and prezeroed and raw and vec4 and force_split == 0 and force_tile == 0
            and 32 // ow + 1 <= oh and not up):
        d = WgradDesc()
        d.inp, d.dout, d.dwt = x.data_ptr(), dout.data_ptr(), dwt.data_ptr()
        d.N, d.H, d.W, d.Cin, d.OH, d.OW, d.Cout, d.ntaps = n, h, w, cin, oh, ow, cout, geom.ntaps
        for i, (a, b) in enumerate(zip(geom.ty, geom.tx)):
            d.ty[i], d.tx[i] = a, b
        d.sy, d.sx, d.ldw, d.Kpad, d.per_sample, d.reserved, d.w_bstride = geom.stride, geom.stride, ldw, kpad, 0, 0, kpad * ldw
        grp.wgrads.append((d, "fsv_conv_wgrad", wargs + (0, lib.stream_ptr()), label,
                           2.0 * n * oh * ow * cout * cin * geom.ntaps, (x, dout, dwt)))
        return dwt
    with profile.scope(label, 2.0 * n * oh * ow * cout * cin * geom.ntaps):
        if narrow:
            lib.call("fsv_conv_wgrad_np", *wargs, _np_mode, lib.stream_ptr())
        else:
            lib.call("fsv_conv_wgrad", *wargs, 1 if up else 0, lib.stream_ptr())
    if raw:
        return dwt
    return unprep_weight_grad(dwt, tuple(w_shape), geom, scale, out)
