"""Differentiable operators of the G/D hot path, each a torch.autograd.Function over the HIP C-ABI kernels.

Operator surface mirrored from the reference (file:line under /root/reference):
  conv2d / linear / batch_conv ........ F.conv2d, nn.Linear, models/networks/base_network.py:56-71
  spectral norm ........................ torch.nn.utils.spectral_norm at architecture.py:60,81-84; generator.py:106-109
  norm_act ............................. BatchNorm2d / InstanceNorm2d(+LeakyReLU) normalization.py:33-35,78-82
  spade_mod ............................ SPADE.forward normalization.py:37-52 (+ actvn architecture.py:15-17)
  upsample2x ........................... F.interpolate(scale_factor=2) generator.py:124, nn.Upsample
  resample ............................. models/networks/base_network.py:28-37
All tensors are fp32, logical NCHW, channels-last memory.  No CPU fallback: lib.check_device refuses host
tensors unless the emulated test library was requested explicitly.
"""
import ctypes
import os
import threading as _threading

import torch

from . import lib, profile
from . import conv as _conv
from . import hconv as _hconv
from .conv import (ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, Geom, conv_dgrad, conv_forward, conv_wgrad, empty_nhwc,
                   gather_gemm, launch_group,
                   prep_weight, to_nhwc, zeros_nhwc)

c_p, c_i, c_ll, c_f = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float
c_llp = ctypes.POINTER(ctypes.c_longlong)
c_pp = ctypes.POINTER(ctypes.c_void_p)
c_ip = ctypes.POINTER(ctypes.c_int)

lib.register_sigs({
    "fsv_warp_fwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_llp, c_llp, c_llp, c_p],
    "fsv_warp_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_llp, c_llp, c_llp, c_llp, c_llp, c_p],
    "fsv_norm_stats": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p, c_p, c_f, c_p],
    "fsv_norm_stats_rep": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p, c_p, c_f, c_i, c_p],
    "fsv_norm_stats_fused": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p, c_p, c_f, c_i, c_p, c_p],
    "fsv_norm_bwd_fused": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p],
    "fsv_colsum_fused": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p],
    "fsv_norm_apply": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p],
    "fsv_norm_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p],
    "fsv_norm_sums": [c_p, c_p, c_p, c_i, c_i, c_p],
    "fsv_norm_stats_from_sums": [c_p, ctypes.c_double, c_p, c_p, c_i, c_f, c_p, c_p, c_f, c_p],
    "fsv_norm_bwd_sums": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "fsv_norm_bwd_apply": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p],
    "fsv_colsum": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "fsv_spade_prep": [c_p, c_p, c_p, c_p, c_ll, c_ll, c_ll, c_ll, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "fsv_spade_mod_fwd": [c_p, c_p, c_p, c_p, c_i, c_pp, c_pp, c_pp, c_pp, c_pp, c_ip, c_llp, c_llp,
                          c_i, c_i, c_i, c_i, c_ll, c_i, c_i, c_i, c_p],
    "fsv_spade_mod_fwd_h": [c_p, c_p, c_p, c_p, c_i, c_pp, c_pp, c_pp, c_pp, c_pp, c_ip, c_llp, c_llp,
                            c_i, c_i, c_i, c_i, c_ll, c_i, c_i, c_i, c_p],
    "fsv_spade_mod_bwd_h": [c_p, c_p, c_p, c_p, c_i, c_pp, c_pp, c_pp, c_pp, c_pp, c_ip, c_llp, c_llp, c_pp, c_p,
                            c_i, c_i, c_i, c_i, c_ll, c_i, c_i, c_i, c_i, c_p],
    "fsv_spade_mod_fwd2": [c_p, c_p, c_p, c_p, c_p, c_i, c_pp, c_pp, c_pp, c_pp, c_pp, c_ip, c_llp, c_llp,
                           c_i, c_i, c_i, c_i, c_ll, c_i, c_i, c_i, c_i, c_p],
    "fsv_spade_bwd_elem": [c_p, c_p, c_p, c_p, c_p, c_i, c_pp, c_pp, c_p, c_i, c_i, c_i, c_ll, c_i, c_i, c_i, c_p],
    "fsv_spade_mod_bwd": [c_p, c_p, c_p, c_p, c_i, c_pp, c_pp, c_pp, c_pp, c_pp, c_ip, c_llp, c_llp, c_pp, c_p,
                          c_i, c_i, c_i, c_i, c_ll, c_i, c_i, c_i, c_p],
    "fsv_spade_conv_s_supported": [c_i, c_i, c_i],
    "fsv_spade_conv_s_fwd": [c_p, c_p, c_p, c_p, c_p, c_i, c_pp, c_pp, c_pp, c_pp, c_pp, c_ip, c_llp, c_llp,
                             c_i, c_i, c_i, c_i, c_ll, c_i, c_i, c_p, c_i, c_i, c_p, c_p],
    "fsv_spade_conv3_supported": [c_i, c_i, c_i],
    # x mean rstd hs out | nmaps maps wg wb bg bb ch w_bstride b_bstride | N H W C ldw stat_bstride up act | wc ldwc Cout bias res wscale |
    # stats stats_slots stats_prezeroed stream
    "fsv_spade_conv3_fwd": [c_p, c_p, c_p, c_p, c_p, c_i, c_pp, c_pp, c_pp, c_pp, c_pp, c_ip, c_llp, c_llp,
                            c_i, c_i, c_i, c_i, c_i, c_ll, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_p],
    "fsv_spade_conv_s_fwd_h": [c_p, c_p, c_p, c_p, c_p, c_i, c_pp, c_pp, c_pp, c_pp, c_pp, c_ip, c_llp, c_llp,
                               c_i, c_i, c_i, c_ll, c_i, c_i, c_p, c_i, c_i, c_p, c_p],
    "fsv_upsample2x_fwd": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "fsv_upsample2x_bwd": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "fsv_act_fwd": [c_p, c_p, c_ll, c_i, c_p],
    "fsv_act_bwd": [c_p, c_p, c_p, c_ll, c_i, c_f, c_p, c_p],
    "fsv_softmax_rows_fwd": [c_p, c_p, c_ll, c_i, c_p],
    "fsv_softmax_rows_bwd": [c_p, c_p, c_p, c_ll, c_i, c_p],
    "fsv_adam_step": [c_p, c_p, c_p, c_p, c_p, c_ll, c_f, c_f, c_f, c_f, c_p],
    "fsv_adam_step_range": [c_p, c_p, c_p, c_p, c_p, c_ll, c_f, c_f, c_f, c_f, c_i, c_p],
    "fsv_sn_power_iter": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_i, c_p],
    "fsv_sn_backward": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "fsv_sn_power_iter_batched": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_i, c_f, c_p],
})

_ws_fn = None


def _ws(g, p, c, like):
    """fp64 scratch for the two-stage column reductions; the size comes from the library's own launch plan."""
    global _ws_fn
    if _ws_fn is None:
        _ws_fn = getattr(lib.get_lib(), "fsv_norm_workspace_doubles")
        _ws_fn.argtypes = [c_i, c_i, c_i]
        _ws_fn.restype = c_i
    return torch.empty(max(int(_ws_fn(g, p, c)), 2), dtype=torch.float64, device=like.device)


def _ticket(like):
    """address of 64 zeroed ints for one fused reduction launch (include/fsv2v.h, fsv_norm_stats_fused): a range of the per-device
    ticket ring (conv.ticket_range)"""
    return _conv.ticket_range(like, 64)


class _SplitColsFn(torch.autograd.Function):
    """torch.split(f, sizes, dim=1) whose backward is ONE concatenation: pieces nobody used (the unread tail of an FC output,
    weights of a site that is switched off) come back as slices of a cached zero row block instead of one zero-fill launch
    each (autograd's SplitWithSizesBackward materialises every undefined gradient first)"""
    _zeros = {}

    @staticmethod
    def forward(ctx, f, sizes):
        ctx.set_materialize_grads(False)
        ctx.sizes, ctx.rows, ctx.meta = sizes, f.shape[0], (f.dtype, f.device)
        return torch.split(f, sizes, dim=1)

    @staticmethod
    def backward(ctx, *grads):
        dtype, device = ctx.meta
        need = max([n for n, g in zip(ctx.sizes, grads) if g is None] + [0])
        key = (dtype, device, ctx.rows)
        z = _SplitColsFn._zeros.get(key)
        if need and (z is None or z.shape[1] < need):
            z = _SplitColsFn._zeros[key] = torch.zeros(ctx.rows, need, dtype=dtype, device=device)
        return torch.cat([g if g is not None else z[:, :n] for n, g in zip(ctx.sizes, grads)], dim=1), None


def split_cols(f, sizes):
    return _SplitColsFn.apply(f, list(sizes))


def _ll(vals):
    return (ctypes.c_longlong * len(vals))(*[int(v) for v in vals])


def _pp(tensors):
    return (ctypes.c_void_p * max(len(tensors), 1))(*[t.data_ptr() for t in tensors])


# ------------------------------------------------------------------------------------------------ activations
def act_backward(dy, y, act, scale=1.0):
    if act == ACT_NONE and scale == 1.0:
        return dy
    dy = to_nhwc(dy) if dy.dim() == 4 else dy.contiguous()
    dx = torch.empty_like(y)
    lib.check_device(dy, y)
    with _hconv.half_side_output(dx) as side:
        lib.call("fsv_act_bwd", lib.ptr(dy), lib.ptr(y), lib.ptr(dx), y.numel(), act, float(scale), side.ptr(), lib.stream_ptr())
    return dx


class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        x = to_nhwc(x) if x.dim() == 4 else x.contiguous()
        y = torch.empty_like(x)
        lib.check_device(x)
        lib.call("fsv_act_fwd", lib.ptr(x), lib.ptr(y), x.numel(), act, lib.stream_ptr())
        ctx.act = act
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return act_backward(dy, y, ctx.act), None


def activation(x, act=ACT_LRELU):
    return _ActFn.apply(x, act)


class _SoftmaxChFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        y = torch.empty_like(x)
        lib.check_device(x)
        lib.call("fsv_softmax_rows_fwd", lib.ptr(x), lib.ptr(y), n * h * w, c, lib.stream_ptr())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = to_nhwc(dy)
        n, c, h, w = y.shape
        dx = torch.empty_like(y)
        lib.call("fsv_softmax_rows_bwd", lib.ptr(dy), lib.ptr(y), lib.ptr(dx), n * h * w, c, lib.stream_ptr())
        return dx


def softmax_channels(x):
    """nn.Softmax(dim=1) on an NCHW tensor (generator.py:384): a row softmax in channels-last memory."""
    return _SoftmaxChFn.apply(x)


# ------------------------------------------------------------------------------------------------ column sums
def colsum(x2d_nhwc, groups, pixels, channels, out=None):
    """x viewed as [groups][pixels][channels] -> [groups, channels]; with `out` the sums are ADDED into it."""
    acc = out is not None
    if out is None:
        out = torch.empty((groups, channels), dtype=torch.float32, device=x2d_nhwc.device)
    lib.check_device(x2d_nhwc)
    ws = _ws(groups, pixels, channels, x2d_nhwc)     # must outlive the call (host allocator frees eagerly)
    lib.call("fsv_colsum_fused", lib.ptr(x2d_nhwc), lib.ptr(ws), lib.ptr(out), groups, pixels, channels, 1 if acc else 0,
             _ticket(out), lib.stream_ptr())
    return out


# ------------------------------------------------------------------------------------------------ spectral norm
class SpectralState:
    """sigma bookkeeping for one weight: persistent u / v buffers live in the owning module."""

    @staticmethod
    def update(weight, u, v, training, eps=1e-12):
        """One power iteration (training) or sigma = u.(Wv) (eval).  Returns sig = [sigma, 1/sigma] (device)."""
        rows = weight.shape[0]
        cols = weight.numel() // rows
        w = weight.detach()
        if not w.is_contiguous():
            w = w.contiguous()
        scratch = torch.empty(rows + cols * (1 + (rows + 63) // 64), dtype=torch.float32, device=w.device)   # fsv_sn_scratch_floats
        sig = torch.empty(2, dtype=torch.float32, device=w.device)
        lib.check_device(w, u, v)
        lib.call("fsv_sn_power_iter", lib.ptr(w), lib.ptr(u), lib.ptr(v), lib.ptr(scratch), lib.ptr(sig), rows, cols,
                 float(eps), 1 if training else 0, lib.stream_ptr())
        return sig


class SpectralGroup:
    """All spectral-normalised layers of one network, power-iterated together in four launches
    (csrc/specnorm.hip, fsv_sn_power_iter_batched) at the start of the network's forward pass.

    `layers` are modules exposing weight_orig / weight_u / weight_v.  After `update()` every layer holds a fresh
    (sigma, 1/sigma) view in `_sig_cached`; a layer consumes it on its first call of the forward pass and falls back
    to its own iteration if it is called again (modules shared between two branches iterate twice, like the
    reference's per-call forward pre-hook)."""

    def __init__(self, layers):
        self.layers = list(layers)
        self._key = None

    def _build(self, device):
        import numpy as np
        rows = [l.weight_orig.shape[0] for l in self.layers]
        cols = [l.weight_orig.numel() // r for l, r in zip(self.layers, rows)]
        t_off, s_off, off = [], [], 0
        for r, c in zip(rows, cols):
            # the layer's t region: t[c], then one partial row per 64-row slab of W (csrc/specnorm.hip: W^T u is summed over the
            # slabs in a fixed order, no atomics)
            t_off.append(off); off += c * (1 + (r + 63) // 64)
            s_off.append(off); off += r
        self.scratch_floats = off
        u_off, v_off, off2 = [], [], 0
        for r, c in zip(rows, cols):
            u_off.append(off2); off2 += r
            v_off.append(off2); off2 += c
        self.snap_floats, self.u_off, self.v_off, self.rows, self.cols = off2, u_off, v_off, rows, cols
        tmap_t, tmap_s = [], []
        for li, (r, c) in enumerate(zip(rows, cols)):
            ntile = ((c + 255) // 256) * ((r + 63) // 64)
            tmap_t += [(li, t) for t in range(ntile)]
            tmap_s += [(li, g) for g in range((r + 3) // 4)]
        mk = lambda a, dt: torch.tensor(a, dtype=dt, device=device)
        self.d_W = mk([l.weight_orig.data_ptr() for l in self.layers], torch.int64)
        self.d_u = mk([l.weight_u.data_ptr() for l in self.layers], torch.int64)
        self.d_v = mk([l.weight_v.data_ptr() for l in self.layers], torch.int64)
        self.d_rows, self.d_cols = mk(rows, torch.int32), mk(cols, torch.int32)
        self.d_toff, self.d_soff = mk(t_off, torch.int32), mk(s_off, torch.int32)
        self.d_uoff, self.d_voff = mk(u_off, torch.int32), mk(v_off, torch.int32)
        self.d_tmap_t = mk(tmap_t, torch.int32).reshape(-1)
        self.d_tmap_s = mk(tmap_s, torch.int32).reshape(-1)
        self.nblk_t, self.nblk_s = len(tmap_t), len(tmap_s)
        self.scratch = torch.empty(off, dtype=torch.float32, device=device)

    def update(self, training, eps=1e-12):
        if not self.layers or not training:
            return
        w0 = self.layers[0].weight_orig
        key = (w0.data_ptr(), self.layers[-1].weight_orig.data_ptr(), self.layers[0].weight_u.data_ptr(), str(w0.device))
        if key != self._key:           # parameters were moved (e.g. into the flat optimiser buffer): rebuild the table
            for l in self.layers:
                if not l.weight_orig.is_contiguous():
                    raise lib.FsvError("spectral weights must be contiguous")
            self._build(w0.device)
            self._key = key
        n = len(self.layers)
        sig = torch.empty((n, 2), dtype=torch.float32, device=w0.device)
        # this pass's u / v: autograd keeps views of this buffer, the persistent buffers are overwritten by the next pass
        snap = torch.empty(self.snap_floats, dtype=torch.float32, device=w0.device)
        lib.check_device(w0)
        lib.call("fsv_sn_power_iter_batched", lib.ptr(self.d_W), lib.ptr(self.d_u), lib.ptr(self.d_v), lib.ptr(self.d_rows),
                 lib.ptr(self.d_cols), lib.ptr(self.d_toff), lib.ptr(self.d_soff), lib.ptr(self.scratch),
                 self.scratch_floats, lib.ptr(sig), lib.ptr(snap), lib.ptr(self.d_uoff), lib.ptr(self.d_voff), n,
                 lib.ptr(self.d_tmap_t), self.nblk_t, lib.ptr(self.d_tmap_s), self.nblk_s, float(eps), lib.stream_ptr())
        for i, l in enumerate(self.layers):
            l._sig_cached = (sig[i], snap[self.u_off[i]:self.u_off[i] + self.rows[i]],
                             snap[self.v_off[i]:self.v_off[i] + self.cols[i]])


def sn_backward(dwsn, weight, u, v, sig, out=None):
    """dW = (dW_sn - <dW_sn, W_sn> u v^T) / sigma; with `out` (a flat-buffer gradient view) the result is ADDED into it."""
    rows = weight.shape[0]
    cols = weight.numel() // rows
    dwsn = dwsn.contiguous()
    w = weight.detach().contiguous()
    part = torch.empty(256, dtype=torch.float64, device=w.device)
    acc = out is not None
    dw = out if acc else torch.empty_like(w)
    lib.call("fsv_sn_backward", lib.ptr(dwsn), lib.ptr(w), lib.ptr(u), lib.ptr(v), lib.ptr(sig), lib.ptr(part),
             lib.ptr(dw), rows, cols, 1 if acc else 0, lib.stream_ptr())
    return dw


# ------------------------------------------------------------------------------------------------ backward concurrency
# The data gradient and the weight gradient of a layer are independent, and the weight-gradient chain only feeds the
# optimiser's gradient sinks, so backward COULD run it on a side HIP stream.  Measured twice on MI355X and removed:
# round 1, joined per layer: 84.9 ms/step against 83.4 ms without; round 2 (profiles/r02_notes.md), joined once before the
# gradient finalisation: 60.0 ms against 56.5 ms, and the per-layer variant crashed inside the ROCm 7.2 graph executor.
# Every cross-stream edge of a captured graph costs tens of microseconds there; only coarse forks pay (streams.py).
import os as _os


_conv_stats_tls = _threading.local()


def _stats_from_producer(x, groups, pixels, channels, eps, run_mean, run_var, momentum, rep=1):
    """(mean, rstd) from the partial sums the producing convolution's epilogue attached to x (`_fsv_stats`, conv2d), or None:
    the second stage of the statistics alone, no read pass over x"""
    st = getattr(x, '_fsv_stats', None)
    if st is None or bn_sync_world(groups) > 1:
        return None
    part, g, slots, p, c = st
    if (g, p, c) != (groups, pixels, channels):
        return None
    lib.register_sigs({"fsv_norm_stats_finish": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_f, c_i, c_p]})
    mean = torch.empty(groups * channels, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    lib.check_device(x, run_mean, run_var)
    lib.call("fsv_norm_stats_finish", lib.ptr(part), lib.ptr(mean), lib.ptr(rstd), groups, pixels, channels, slots, float(eps),
             lib.ptr(run_mean), lib.ptr(run_var), float(momentum), int(rep), lib.stream_ptr())
    return mean, rstd


# `ctx.needs_input_grad` of a custom Function says which inputs REQUIRE a gradient - it is (.., True, ..) for a module's weight also
# inside `torch.no_grad()` (round 6: the no-graph generator pass of the discriminator step had been asking the fused SPADE kernels for
# their side outputs, a 134 MB write per level-0 site that nobody read), and grad mode is always off inside forward().  The wrappers
# below note the caller's grad mode right before .apply (same thread, synchronous).
_outer_grad = _threading.local()


def _note_grad_mode():
    _outer_grad.on = torch.is_grad_enabled()


def _keeps_graph(ctx):
    """will this forward have a backward? (some input requires a gradient AND the caller records a graph)"""
    return bool(any(ctx.needs_input_grad) and getattr(_outer_grad, 'on', True))


class _ConvFn(torch.autograd.Function):
    """y = act((conv(x, W * inv_sigma) + bias) * scale) + res.  W: OIHW, or [B]OIHW for per-sample weights."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, sig, u, v, geom, act, scale, uv_owned=False, stats_groups=0, allow_half=True, up=False):
        weight._fsv_conv_param = True          # FlatAdam: these parameters take their gradients through the sink
        if bias is not None:
            bias._fsv_conv_param = True
        w4 = weight.detach()
        if w4.dim() == 2:                      # nn.Linear weight: a 1x1 convolution over a [1, in, 1, rows] "image"
            w4 = w4.view(w4.shape[0], w4.shape[1], 1, 1)
        per_sample = w4.dim() == 5
        cout = w4.shape[-4]
        # input channel counts that are not a multiple of 4 (labels 6, RGB 3, flow-net input 15) would fall onto the
        # scalar gather path of the kernels; zero-pad channels (input and weights) once so the float4 path is used
        cin = x.shape[1]
        # `--amp O1` on the half-precision kernels (csrc/conv_h.hip): this layer's activations go to HBM as IEEE half - x here
        # (cast once, kept for the weight gradient), the pre-activation gradient in backward (cast once, shared by the data and
        # the weight gradient) - when its GEMMs fit the kernels' contract: input channels padded to a multiple of 8, output
        # channels a multiple of 8 (the data gradient gathers over them; the image / flow / mask heads and the
        # discriminator's one-channel output layer stay on the exact fp32 kernels).  Outputs are fp32: every consumer of a
        # convolution here (normalisation, SPADE, residual sums, losses) computes in fp32, like apex O1's fp32 list.
        # a producer may hand over an input that already carries the zero channels (ops.pack_d_* under `--amp`: padding and
        # conversion done while packing): the weight says how many channels are real
        cin_w = w4.shape[-3]
        prepadded = (not per_sample) and cin > cin_w
        half = (allow_half and _conv.h_kernels() and cout % 8 == 0 and (per_sample and cin % 8 == 0 or not per_sample) and
                x.dtype in (torch.float32, torch.float16))
        if prepadded:
            cpad, cin = cin - cin_w, cin_w
            if (cin + cpad) % (8 if half else 4) != 0:
                raise ValueError("pre-padded convolution input with %d channels for a weight of %d" % (cin + cpad, cin))
        else:
            cpad = ((-cin) % 8 if half else (-cin) % 4) if not per_sample else 0
        # x may be the (still unwritten) output of a held-back bn_s modulation (spade_into_conv): settle that BEFORE anything below
        # reads or converts x - either this convolution issues both as one kernel (it then never reads x), or the modulation is
        # launched now (round-4 advisor: with `--amp` and the unfused backward A/B switch the cast below used to read the unwritten
        # fp32 tensor, the pointer comparison then failed and the modulation ran after its consumer)
        st_wanted = bool(stats_groups and not per_sample)
        site = _spade_pending_for(x)
        site3 = None
        if site is not None and site.get('conv3'):
            if _spade_conv3_fits(site, geom, per_sample, res, act, scale, half, cpad, cout, up):
                site3, site = site, None
            else:
                _spade_launch(site)
                site = None
        elif site is not None and not _spade_conv_s_fits(site, geom, per_sample, bias, res, act, scale, half, cpad, st_wanted, cout):
            _spade_launch(site)
            site = None
        # up: x stands for its nearest x2 up-sampling (generator.py:124,497-504,541-572: nn.Upsample in front of a 3x3
        # convolution).  Folded into the gather where the launch allows (round 5, csrc/conv_igemm.hip ConvP::up: the kernels read x
        # at (y >> 1, x >> 1), forward and weight gradient; the up-sampled tensor is never written), materialised here otherwise
        # (the half-precision path, scalar-gather layers).  Either way the data gradient is pooled 2 x 2 in backward.
        fold = bool(up and not half and cpad == 0 and not prepadded and site is None and x.dtype == torch.float32 and
                    _conv.up_foldable(cin, cout, per_sample, geom.out_hw(2 * x.shape[2], 2 * x.shape[3])))
        if up and not fold:
            xs = to_nhwc(_hconv.cast(x, torch.float32) if x.dtype == torch.float16 else x)
            x = empty_nhwc(xs.shape[0], xs.shape[1], 2 * xs.shape[2], 2 * xs.shape[3], xs)
            lib.check_device(xs)
            lib.call("fsv_upsample2x_fwd", lib.ptr(xs), lib.ptr(x), xs.shape[0], xs.shape[2], xs.shape[3], xs.shape[1], lib.stream_ptr())
        ctx.up, ctx.fold = bool(up), fold
        ctx.x_half = x.dtype == torch.float16          # a half input (a producer that rounded at its store) gets a half gradient
        if x.dtype == torch.float16 and not half:
            x = _hconv.cast(x, torch.float32)
        x = pad_channels_nhwc(x, cpad, half=half) if (cpad and not prepadded) else to_nhwc(x)
        if half:
            x = _hconv.to_half_nhwc(x)
        ctx.cpad, ctx.cin, ctx.half, ctx.prepadded = cpad, cin, half, prepadded
        inv = sig[1:2] if sig is not None else None
        # parameters owned by a FlatAdam keep persistent K-major layouts (layout_cache.py); 1/sigma then rides in the
        # GEMM epilogue instead of the re-arrangement
        cache = getattr(weight, '_fsv_cache', None) if not per_sample else None
        entry = cache.lookup(weight, tuple(w4.shape), geom, cpad, up=bool(up and fold)) if cache is not None else None
        ctx.entry = entry
        if entry is not None:
            wt, ldw = entry.fwd
            wscale = inv
        else:
            if cpad:
                w4 = torch.nn.functional.pad(w4, (0, 0, 0, 0, 0, cpad))
            # (half path: W itself is rounded and 1 / sigma applied to the fp32 accumulator, like the cached layouts - the
            # arithmetic must not depend on whether an optimiser's layout cache owns the weight)
            wt, _, ldw = prep_weight(w4, 0, geom, scale=None if half else inv)
            wscale = inv if half else None
        b = bias.detach() if bias is not None else None
        if b is not None and not (per_sample and b.dim() == 2 and b.stride(1) == 1):
            b = b.contiguous()       # (per-sample bias rows are read in place by gather_gemm)
        if res is not None and act != ACT_NONE:
            raise ValueError("residual add is only fused after a linear epilogue")
        if scale != 1.0 and act != ACT_NONE:
            raise ValueError("output scale is only fused with a linear epilogue")
        st = dict(groups=stats_groups) if st_wanted else None
        y = None
        if site is not None:
            # x is the output of the held-back bn_s modulation and the fused kernel covers the pair (checked above)
            if half:          # the N-major half twin of the layout the half-precision convolution would read
                wh, kpad_h, _ = _conv.half_twin(wt)
                y = _spade_conv_s_launch(site, wh, kpad_h, cout, wscale, want_hs=_keeps_graph(ctx))
            else:
                y = _spade_conv_s_launch(site, wt, ldw, cout, wscale, want_hs=_keeps_graph(ctx))
        elif site3 is not None:
            y = _spade_conv3_launch(site3, wt, ldw, cout, wscale, b, res.detach() if res is not None else None,
                                    want_hs=_keeps_graph(ctx), st=st)
        # (a layer whose output feeds a BatchNorm - the flow decoder's three - loses the statistics epilogue on the placed class
        # launches and reduces in a pass of its own: FSV_UP_SUBPIXEL_STATS=0 keeps such layers on the single gather + fused
        # statistics instead; in-box A/B of round 6: profiles/r06_notes.md)
        if (y is None and fold and _up_subpixel_wanted(x.shape[0], x.shape[2], x.shape[3], cin, cout, geom, per_sample, res) and
                (not st_wanted or _os.environ.get('FSV_UP_SUBPIXEL_STATS', '1') == '1')):
            y = _up_subpixel_forward(x, w4, cout, b, act, scale, inv, cached=entry.up_fwd if entry is not None else None)
        if y is None:
            y = conv_forward(x, wt, ldw, cout, geom, bias=b, res=res.detach() if res is not None else None, act=act,
                             scale=scale, per_sample=per_sample, wscale=wscale, stats=st, up=fold)
        # statistics of y left by the epilogue (conv2d() hands them to the normalisation that follows through the output tensor)
        _conv_stats_tls.last = st if (st is not None and 'part' in st) else None
        ctx.geom, ctx.act, ctx.scale, ctx.per_sample = geom, act, scale, per_sample
        ctx.has_bias, ctx.has_res, ctx.has_sn = bias is not None, res is not None, sig is not None
        ctx.bias_ref = bias                      # the leaf itself (its .grad slice is the sink target), not saved data
        ctx.x_shape = tuple(x.shape)
        if not _keeps_graph(ctx):
            return y                     # forward under no_grad (the D step's generator pass): nothing to keep
        if ctx.has_sn:
            # u / v of the persistent buffers are overwritten by the next power iteration: keep this call's values
            # (the batched pass already hands out per-pass snapshots)
            ctx.save_for_backward(x, weight, y, sig, u if uv_owned else u.clone(), v if uv_owned else v.clone())
        else:
            ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.has_sn:
            x, weight, y, sig, u, v = ctx.saved_tensors
        else:
            x, weight, y = ctx.saved_tensors
            sig = u = v = None
        dy = to_nhwc(dy)
        geom = ctx.geom
        n, _, h, w = ctx.x_shape
        if ctx.fold:                # x was kept at half the size the convolution saw
            h, w = 2 * h, 2 * w
        cpad, cin = ctx.cpad, ctx.cin
        dpre = act_backward(dy, y, ctx.act, ctx.scale) if (ctx.act != ACT_NONE or ctx.scale != 1.0) else dy
        # half path: ONE rounding of the pre-activation gradient, shared by the data gradient and the weight gradient (the bias
        # gradient below is the fp32 column sum of the unrounded tensor: the bias add is fp32 epilogue work)
        dpre_g = _hconv.to_half_nhwc(dpre) if ctx.half else dpre
        inv = sig[1:2] if sig is not None else None
        dx = dw = db = dres = None
        w4 = weight.detach()
        if w4.dim() == 2:
            w4 = w4.view(w4.shape[0], w4.shape[1], 1, 1)
        entry = ctx.entry
        w_shape = tuple(w4.shape[:-3]) + (w4.shape[-3] + cpad,) + tuple(w4.shape[-2:])
        if cpad and entry is None:
            w4 = torch.nn.functional.pad(w4, (0, 0, 0, 0, 0, cpad))
        # gradient sink: parameters owned by a FlatAdam expose their slice of the flat gradient buffer as .grad; the
        # last kernel of the weight-gradient chain adds into it directly and autograd gets None (no AccumulateGrad add)
        w_sink = weight.grad if (getattr(weight, '_fsv_sink', False) and weight.grad is not None) else None
        b_sink = None
        if ctx.has_bias:
            bias_t = ctx.bias_ref
            b_sink = bias_t.grad if (getattr(bias_t, '_fsv_sink', False) and bias_t.grad is not None) else None
        want_w = ctx.needs_input_grad[1]
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        want_x = ctx.needs_input_grad[0]
        if want_w:
            fin = getattr(weight, '_fsv_finalizer', None) if (w_sink is not None and entry is not None) else None
            # conv3x3(nearest_x2(x)): four 2x2-tap weight gradients over the source pixels instead of one 3x3 over the up-sampled
            # ones (2.25x fewer MACs; _up_wgrad_classes)
            up_dwt = _up_wgrad_classes(x, dpre_g) if (ctx.up and _up_wgrad_direct(ctx, geom, x, dpre_g, cpad)) else None
            if fin is not None:
                # deferred: leave the K-major result to the optimiser's grouped finalisation (grad_finalize.py)
                dwt = up_dwt if up_dwt is not None else conv_wgrad(x, dpre_g, geom, w_shape, raw=True, arena=fin, up=ctx.fold)
                if ctx.has_sn:
                    fin.add(entry, dwt, w_sink, sig, u, v)
                else:
                    fin.add(entry, dwt, w_sink)
                dw = None
            elif ctx.has_sn or cpad:
                dwsn = (_conv.unprep_weight_grad(up_dwt, tuple(w_shape), geom) if up_dwt is not None else
                        conv_wgrad(x, dpre_g, geom, w_shape, per_sample=ctx.per_sample, up=ctx.fold))
                if cpad:
                    dwsn = dwsn[:, :cin].contiguous()
                if ctx.has_sn:
                    dw = sn_backward(dwsn, weight, u, v, sig, out=w_sink)
                elif w_sink is not None:
                    dw = w_sink.add_(dwsn.view_as(w_sink))
                else:
                    dw = dwsn
                dw = None if w_sink is not None else dw.view_as(weight)
            else:
                dw = (_conv.unprep_weight_grad(up_dwt, tuple(w_shape), geom, None, w_sink) if up_dwt is not None else
                      conv_wgrad(x, dpre_g, geom, w_shape, per_sample=ctx.per_sample, out=w_sink, up=ctx.fold))
                dw = None if w_sink is not None else dw.view_as(weight)
        if want_b:
            cout = dpre.shape[1]
            hw = dpre.shape[2] * dpre.shape[3]
            if ctx.per_sample:
                db = colsum(dpre, n, hw, cout)
            elif b_sink is not None:
                fin_b = getattr(bias_t, '_fsv_finalizer', None) if _os.environ.get('FSV_DEFER_BIAS', '1') == '1' else None
                if fin_b is not None and dpre.is_contiguous(memory_format=torch.channels_last):
                    fin_b.add_bias(dpre, b_sink)          # one grouped column-sum pass for the whole backward
                else:
                    colsum(dpre, 1, n * hw, cout, out=b_sink)
            else:
                db = colsum(dpre, 1, n * hw, cout).view(cout)
        if want_x and ctx.up and _up_dgrad_direct(ctx, geom, dpre_g, w4, cpad):
            # conv(nearest_x2(x)): the data gradient w.r.t. x itself in ONE gather-GEMM (round 5) - pooling the up-sampled gradient
            # 2 x 2 and the flipped 3x3 taps combine into a 4x4 stride-2 convolution over dy with summed weights (_up_dgrad_weight):
            # 16 taps on a quarter of the pixels = 2.25x fewer MACs than the data gradient at the up-sampled size, whose tensor
            # (4x the size of x) is neither written nor pooled
            if entry is not None and entry.up_dgrad is not None:
                # the summed-tap layout is kept by the optimiser's layout cache (refreshed once per step with the others)
                wt4, ldw4 = entry.up_dgrad
                ty = [2 - a for a in range(4) for _ in range(4)]
                tx = [2 - b for _ in range(4) for b in range(4)]
            else:
                v, khs, kws, ty, tx = _up_dgrad_weight(w4)
                wt4, _, ldw4 = prep_weight(v, 1, geom, khs, kws, None)
            dx = _conv.gather_gemm(dpre_g, wt4, ldw4, w4.shape[-3], h // 2, w // 2, ty, tx, 2, 2, wscale=inv)
        elif want_x:
            dx = conv_dgrad(dpre_g, w4, geom, (h, w), scale=inv, per_sample=ctx.per_sample,
                            cached=entry.dgrad if entry is not None else None, cin=w_shape[-3],
                            out_half=ctx.x_half and ctx.half)
            if cpad and not ctx.prepadded:
                dx = dx[:, :cin]
            if ctx.up:              # the gradient of the nearest x2 up-sampling: 2 x 2 sums (exact fp32, fixed order)
                dxu = to_nhwc(_hconv.cast(dx, torch.float32) if dx.dtype == torch.float16 else dx)
                dx = empty_nhwc(n, dxu.shape[1], h // 2, w // 2, dxu)
                lib.call("fsv_upsample2x_bwd", lib.ptr(dxu), lib.ptr(dx), n, h // 2, w // 2, dxu.shape[1], lib.stream_ptr())
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres = dy
        return dx, dw, db, dres, None, None, None, None, None, None, None, None, None, None


# (The FORWARD pass has the same algebra - conv3x3(nearest_x2(x)) = four 2x2-tap convolutions over x with summed weights, one per
# output parity class: 16 instead of 36 taps per source pixel.  As ONE grouped launch of four placed problems it measured slower than
# the single gather through the up-sampling index (46.42 against 45.84 ms per step: the grouped kernel's rate on short-K placed
# problems) and was removed; as four PLAIN launches on the layers large enough not to K-split it wins 0.65 ms: _up_subpixel_forward
# below; profiles/r05_notes.md sections 10 - 11.)
class _TapGeom:
    """the part of conv.Geom the raw weight-gradient launch reads, for an explicit tap list"""

    def __init__(self, ty, tx):
        self.ntaps, self.ty, self.tx, self.stride = len(ty), list(ty), list(tx), 1


_up_wgrad_m = {}


def _up_wgrad_direct(ctx, geom, x, dpre, cpad):
    """the class-wise weight gradient of conv3x3(nearest_x2(x)) covers: x kept at source resolution (the folded forward), 3x3 /
    stride 1 / padding 1, shared weights, exact fp32, float4 channels (FSV_UP_WGRAD=0: the gather through the up-sampling index, A/B)"""
    return (_os.environ.get('FSV_UP_WGRAD', '1') == '1' and ctx.fold and geom.kh == 3 and geom.kw == 3 and geom.stride == 1 and
            geom.pad == 1 and not ctx.per_sample and not ctx.half and cpad == 0 and x.dtype == torch.float32 and
            dpre.dtype == torch.float32 and x.shape[1] % 4 == 0 and dpre.shape[1] % 4 == 0 and dpre.shape[1] > 4 and
            32 // x.shape[3] + 1 <= x.shape[2] and _conv.narrow_staging_mode() == 0 and _os.environ.get('FSV_DETERMINISTIC', '0') != '1')


def _up_wgrad_classes(x, dy):
    """Weight gradient of y = conv3x3(nearest_x2(x)) in the K-major layout of the plain launch ([(tap, ci)][co], 9 taps), from
    FOUR 2x2-tap weight gradients over the source pixels - one per output parity class r: the pixels 2s + r of dy against x[s + d],
    d in {-1, 0} (r = 0) resp. {0, 1} (r = 1) per axis - 16 instead of 36 products per source pixel, channel pair.  The 3x3 taps are
    sums of those blocks: tap t of the kernel is seen by class r through offset d = floor((r + t) / 2) (the transpose of the forward
    relation W0 | W1 + W2 and W0 + W1 | W2).  x: NHWC fp32 at source resolution, dy: NHWC fp32 at twice that."""
    n, cin, h, w = x.shape
    cout = dy.shape[1]
    # the four parity classes of dy as dense NHWC tensors: ONE strided copy
    d6 = dy.permute(0, 2, 3, 1).reshape(n, h, 2, w, 2, cout).permute(2, 4, 0, 1, 3, 5).contiguous()        # [2, 2, n, h, w, cout]
    kpad_c, ldw = (4 * cin + 31) // 32 * 32, (cout + 31) // 32 * 32
    slab = _Slab(torch.zeros(4 * kpad_c * ldw, dtype=torch.float32, device=x.device))        # ONE fill for the four results
    for ry in (0, 1):
        for rx in (0, 1):
            ty = [ry - 1 + iy for iy in (0, 1) for _ in (0, 1)]
            tx = [rx - 1 + ix for _ in (0, 1) for ix in (0, 1)]
            conv_wgrad(x, d6[ry, rx].permute(0, 3, 1, 2), _TapGeom(ty, tx), (cout, cin, 2, 2), raw=True, arena=slab)
    g = slab.buf.view(4, kpad_c, ldw)[:, :4 * cin]                                     # [class][(tap of the class, cin)][ldw]
    m = _up_wgrad_m.get(x.device)
    if m is None:
        from . import streams as _streams
        sm = torch.tensor([[1., 0., 0.], [0., 1., 1.], [1., 1., 0.], [0., 0., 1.]])          # rows (r, d), columns kernel index
        mm = torch.einsum('pk,ql->pqkl', sm, sm).reshape(2, 2, 2, 2, 9).permute(0, 2, 1, 3, 4).reshape(16, 9)   # [(ry, rx), (iy, ix)][t]
        m = _up_wgrad_m[x.device] = _streams.shared(lambda: mm.t().contiguous().to(x.device))
    # dW[t] = sum_{class, tap of the class} M[t][(class, tap)] G[(class, tap)]: one [9 x 16] x [16 x cin ldw] product
    dwt9 = torch.mm(m, g.reshape(16, cin * ldw)).view(1, 9 * cin, ldw)
    kpad = (9 * cin + 31) // 32 * 32
    if kpad != 9 * cin:
        dwt9 = torch.nn.functional.pad(dwt9, (0, 0, 0, kpad - 9 * cin))
    return dwt9


class _Slab:
    """conv.conv_wgrad's `arena` protocol over one zeroed buffer: consecutive slices"""

    def __init__(self, buf):
        self.buf, self.off = buf, 0

    def take(self, nfloats):
        out = self.buf[self.off:self.off + nfloats]
        self.off += nfloats
        return out


def _up_subpixel_wanted(n, h, w, cin, cout, geom, per_sample, res):
    """conv3x3(nearest_x2(x)) forward as four plain 2x2-tap launches over the source pixels (one per output parity class) where
    every class is a chip-filling GEMM of its own (>= 8192 source pixels: no K split, which a placed launch cannot take).
    FSV_UP_SUBPIXEL=0 / FSV_UP_SUBPIXEL_MIN: in-box A/B switches (profiles/r05_notes.md section 11: -0.65 ms per step at 8192,
    nothing at 2048; section 10: ONE grouped launch of the four classes measured slower than the single gather)"""
    return (_os.environ.get('FSV_UP_SUBPIXEL', '1') == '1' and not per_sample and res is None and geom.kh == 3 and geom.kw == 3 and
            geom.stride == 1 and geom.pad == 1 and n * h * w >= int(_os.environ.get('FSV_UP_SUBPIXEL_MIN', '8192')) and cin % 4 == 0)


_SUBPIXEL_ROWS = ((1., 0., 0.), (0., 1., 1.), (1., 1., 0.), (0., 0., 1.))      # per axis (r, tap): W0 | W1 + W2 || W0 + W1 | W2
_DGRAD_ROWS = ((1., 0., 0.), (1., 1., 0.), (0., 1., 1.), (0., 0., 1.))         # per axis: W0, W0 + W1, W1 + W2, W2
_tap_sum_m = {}


def _tap_sums(w4, rows):
    """V[o, i, p, q] = sum_{k, l} rows[p][k] rows[q][l] W[o, i, k, l] for a 3x3 OIHW kernel - the summed weights of the sub-pixel
    forms of conv3x3(nearest_x2(x)) - as ONE product [Cout Cin x 9] x [9 x 16] (0 / 1 coefficients: exact products, the sums are
    the additions themselves)"""
    key = (w4.device, rows)
    k = _tap_sum_m.get(key)
    if k is None:
        from . import streams as _streams
        r = torch.tensor(rows)
        kk = torch.einsum('pk,ql->klpq', r, r).reshape(9, 16).contiguous()
        k = _tap_sum_m[key] = _streams.shared(lambda: kk.to(w4.device))
    cout, cin = w4.shape[:2]
    return torch.mm(w4.reshape(cout * cin, 9), k).view(cout, cin, 4, 4)




def _up_subpixel_forward(x, w4, cout, bias, act, scale, wscale, cached=None):
    """y = act((conv3x3(nearest_x2(x), W) * wscale + bias) * scale) without the up-sampled tensor AND without its redundant
    products: output pixel 2s + r (r in {0,1} per axis) sees x[s - 1], x[s] (r = 0: weights W0, W1 + W2) resp. x[s], x[s + 1]
    (r = 1: W0 + W1, W2) - per parity class (ry, rx) a 2x2-tap convolution over x whose outputs are placed at stride 2
    (generator.py:489-493, 559-563: nn.Upsample(2) -> conv3x3).  The summed weights carry one fp32 rounding each against the
    reference's sum of the separate products.  x: NHWC fp32 at source resolution; w4: OIHW (un-normalised under spectral norm:
    1 / sigma rides in wscale)."""
    n, cin, h, w = x.shape
    y = empty_nhwc(n, cout, 2 * h, 2 * w, x)
    # ONE re-arrangement for the four classes: 16 taps in class-major order, class c = K rows [c * 4 cin, (c + 1) * 4 cin)
    cls = [(ry, rx) for ry in (0, 1) for rx in (0, 1)]
    one = (4 * cin) % 32 == 0
    if one and cached is not None:
        # kept by the optimiser's layout cache (layout_cache.LayoutCache._add_up_jobs): no per-call product / re-arrangement
        wall, ldw = cached
    else:
        v = _tap_sums(w4, _SUBPIXEL_ROWS)                                      # [cout, cin, 4, 4]: rows / columns (r, tap)
        khs = [2 * ry + iy for ry, rx in cls for iy in (0, 1) for _ in (0, 1)]
        kws = [2 * rx + ix for ry, rx in cls for _ in (0, 1) for ix in (0, 1)]
        if one:
            wall, _, ldw = prep_weight(v, 0, Geom(3, 3, 1, 1), khs, kws, None)
    for c, (ry, rx) in enumerate(cls):
        ty = [ry - 1 + iy for iy in (0, 1) for _ in (0, 1)]                 # r = 0: x[s - 1], x[s]; r = 1: x[s], x[s + 1]
        tx = [rx - 1 + ix for _ in (0, 1) for ix in (0, 1)]
        if one:
            wt = wall[0, c * 4 * cin:(c + 1) * 4 * cin]
        else:
            wt, _, ldw = prep_weight(v, 0, Geom(3, 3, 1, 1), khs[4 * c:4 * c + 4], kws[4 * c:4 * c + 4], None)
        _conv.gather_gemm(x, wt, ldw, cout, h, w, ty, tx, 1, 1, bias=bias, act=act, scale=scale, out=y,
                          place=(2 * h, 2 * w, 2, 2, ry, rx), force_split=1, wscale=wscale)
    return y


def _up_dgrad_direct(ctx, geom, dpre, w4, cpad):
    """the one-launch data gradient of conv(nearest_x2(x)) covers: 3x3 / stride 1 / padding 1, shared weights, the exact-fp32
    float4 gather (FSV_UP_DGRAD=0: data gradient at the up-sampled size + 2 x 2 pooling, in-box A/B)"""
    return (_os.environ.get('FSV_UP_DGRAD', '1') == '1' and geom.kh == 3 and geom.kw == 3 and geom.stride == 1 and geom.pad == 1 and
            not ctx.per_sample and not ctx.half and cpad == 0 and w4.dim() == 4 and dpre.dtype == torch.float32 and
            dpre.shape[1] % 4 == 0 and w4.shape[1] > 4 and _conv.narrow_staging_mode() == 0)


def _up_dgrad_weight(w4):
    """Weights and taps of the data gradient of y = conv3x3(nearest_x2(x)) w.r.t. x as ONE convolution over dy:
        dx[s] = sum_{r in {0,1}^2} dxu[2s + r],  dxu[q] = sum_t W[t]^T dy[q - t]   =>   dx[s] = sum_{o in {-1..2}^2} V[o]^T dy[2s + o]
    with V[o] = sum_{r - t = o} W[t] - per axis (W0, W1, W2) -> o = -1: W2, 0: W1 + W2, 1: W0 + W1, 2: W0: the 2-wide running sums of
    the zero-padded kernel, read backwards.  Returns (V as a 4x4 OIHW tensor, kernel rows / columns of the 16 taps, their offsets
    into dy).  Sums of at most four weights: one fp32 rounding each, against the reference's sum of four products."""
    v = _tap_sums(w4, _DGRAD_ROWS)                                                                  # v[a] = (W0, W0+W1, W1+W2, W2)
    khs = [a for a in range(4) for _ in range(4)]
    kws = [b for _ in range(4) for b in range(4)]
    return v.contiguous(), khs, kws, [2 - a for a in khs], [2 - b for b in kws]


def _unpack_sn(sn):
    """sn: None | (sig, u, v) from SpectralState.update | (sig, u_snapshot, v_snapshot, True) from SpectralGroup"""
    if sn is None:
        return None, None, None, False
    if len(sn) == 4:
        return sn
    return sn[0], sn[1], sn[2], False


def conv2d(x, weight, bias=None, stride=1, padding=0, act=ACT_NONE, scale=1.0, res=None, sn=None, stats_groups=0, up=False):
    """sn: None or (sig, u, v) from SpectralState.update for this call.

    stats_groups: 1 (a BatchNorm follows) / -1 (an InstanceNorm follows: one group per sample) - the convolution's epilogue
    then also leaves the per-channel sums of its output (csrc/conv_igemm.hip ConvP::stats) and the returned tensor carries them
    as `_fsv_stats` for norm_act / spade_mod, which skip their own reduction pass over it.  Only a hint: launches that cannot
    produce them (K-split plans, scalar gather) leave the attribute off."""
    kh, kw = weight.shape[-2:]
    geom = Geom(kh, kw, stride, padding)
    sig, u, v, owned = _unpack_sn(sn)
    groups = (x.shape[0] if stats_groups < 0 else stats_groups) if stats_groups else 0
    # up: the convolution of nearest_x2(x) (ops._ConvFn.forward: folded into the gather where the launch allows)
    _note_grad_mode()
    y = _ConvFn.apply(x, weight, bias, res, sig, u, v, geom, act, scale, owned, groups, True, bool(up))
    st = getattr(_conv_stats_tls, 'last', None)
    if groups and st is not None:
        _conv_stats_tls.last = None
        y._fsv_stats = (st['part'], groups, st['slots'], y.shape[0] * y.shape[2] * y.shape[3] // groups, y.shape[1])
    return y


def linear(x2d, weight, bias=None, act=ACT_NONE, sn=None):
    """y[R, out] = act(x2d[R, in] @ weight[out, in]^T + bias) on the same gather-GEMM kernel (1x1, H=1, W=R)."""
    r, cin = x2d.shape
    x4 = x2d.contiguous().view(1, 1, r, cin).permute(0, 3, 1, 2)
    sig, u, v, owned = _unpack_sn(sn)
    # (nn.Linear = the weight generators: fp32 also under `--amp`, like the grouped bank below - see mlp_bank)
    _note_grad_mode()
    y4 = _ConvFn.apply(x4, weight, bias, None, sig, u, v, Geom(1, 1, 1, 0), act, 1.0, owned, 0, False)
    return y4.permute(0, 2, 3, 1).reshape(r, weight.shape[0])


# ------------------------------------------------------------------------------------------------ banks of small MLPs
ACT_DLRELU = 6       # FSV_ACT_DLRELU (include/fsv2v.h): gather-GEMM epilogue v * leaky_relu'(res)


class _MlpBankFn(torch.autograd.Function):
    """Several independent Linear(+LeakyReLU) chains - the weight generators of generator.py:103-110, 245-273: per adaptive
    level four MLPs of `n_fc_layers` + 1 spectral-normalised Linears on the same rows - advanced LAYER BY LAYER with one
    grouped launch per layer (conv.launch_group) instead of one small launch per Linear: 3 launches forward instead of 48
    at C3, and in backward per layer one grouped weight-gradient launch and one grouped data-gradient launch whose epilogue
    already multiplies by LeakyReLU'(output of the layer below) (FSV_ACT_DLRELU; no separate activation-backward pass).
    Arithmetic per output element is that of ops.linear (same kernels, same k order).

    apply(meta, rows_0 .. rows_{L-1}, (weight, bias) per chain per layer ...).  meta: dict(chains=[(level, nlayers)],
    sn=[[(sig, u, v) per layer] per chain], entries=[[layout-cache entry per layer] per chain]).  Requirements (checked by
    the caller, networks.FewShotGenerator._mlp_bank): every weight is owned by a FlatAdam with layout cache; for a pass that
    records gradients also the gradient sinks + deferred finalisation."""

    @staticmethod
    def forward(ctx, meta, *tensors):
        chains, sns, entries = meta['chains'], meta['sn'], meta['entries']
        nlev = meta['nlevels']
        rows = tensors[:nlev]
        params = tensors[nlev:]
        geom = Geom(1, 1, 1, 0)
        # per chain: list of (weight, bias)
        wb, k = [], 0
        for (_, nl) in chains:
            wb.append([(params[k + 2 * j], params[k + 2 * j + 1]) for j in range(nl)])
            k += 2 * nl
        for ch in wb:
            for w, b in ch:
                w._fsv_conv_param = True
                b._fsv_conv_param = True
        nl_max = max(nl for _, nl in chains)
        x4 = [None] * nlev
        for l, r in enumerate(rows):
            r = r.detach().contiguous()
            x4[l] = r.view(1, 1, r.shape[0], r.shape[1]).permute(0, 3, 1, 2)
        cur = [x4[lev] for (lev, _) in chains]
        xs = [[] for _ in chains]          # input of every layer
        ys = [[] for _ in chains]          # output of every layer
        for j in range(nl_max):
            with launch_group():
                for c, (lev, nl) in enumerate(chains):
                    if j >= nl:
                        continue
                    w, b = wb[c][j]
                    e = entries[c][j]
                    wt, ldw = e.fwd
                    sig = sns[c][j][0]
                    act = ACT_LRELU if j < nl - 1 else ACT_NONE
                    y = gather_gemm(cur[c], wt, ldw, w.shape[0], 1, cur[c].shape[3], geom.ty, geom.tx, 1, 1,
                                    bias=b.detach(), act=act, wscale=sig[1:2])
                    xs[c].append(cur[c]); ys[c].append(y)
                    cur[c] = y
        ctx.meta, ctx.geom = meta, geom
        ctx.wb = wb
        ctx.xs, ctx.ys = (xs, ys) if any(ctx.needs_input_grad) else (None, None)
        return tuple(y[-1].permute(0, 2, 3, 1).reshape(y[-1].shape[3], y[-1].shape[1]) for y in ys)

    @staticmethod
    def backward(ctx, *douts):
        meta, geom, wb, xs, ys = ctx.meta, ctx.geom, ctx.wb, ctx.xs, ctx.ys
        chains, sns, entries, nlev = meta['chains'], meta['sn'], meta['entries'], meta['nlevels']
        d = []
        for c, g in enumerate(douts):
            y = ys[c][-1]
            if g is None:
                g = torch.zeros((y.shape[3], y.shape[1]), dtype=torch.float32, device=y.device)
            g = g.contiguous()
            d.append(g.view(1, 1, g.shape[0], g.shape[1]).permute(0, 3, 1, 2))
        nl_max = max(nl for _, nl in chains)
        cls0 = geom.dgrad_classes[0]
        drows = [None] * nlev
        for j in reversed(range(nl_max)):
            live = [c for c, (_, nl) in enumerate(chains) if j < nl]
            # weight + bias gradients of layer j (deferred finalisation: K-major result + job, grouped column sums)
            with launch_group():
                for c in live:
                    w, b = wb[c][j]
                    e = entries[c][j]
                    fin = w._fsv_finalizer
                    w_shape = (w.shape[0], w.shape[1], 1, 1)
                    dwt = conv_wgrad(xs[c][j], d[c], geom, w_shape, raw=True, arena=fin)
                    sig, u, v = sns[c][j]
                    fin.add(e, dwt, w.grad, sig, u, v)
                    b._fsv_finalizer.add_bias(d[c], b.grad)
            # data gradient of layer j.  j > 0: handed to layer j - 1 as its pre-activation gradient (x LeakyReLU'(y_{j-1}) in the
            # epilogue).  j == 0: the chains of one level share their input rows; their gradients are summed in a fixed
            # order by chaining the residual operand through one grouped launch per chain position.
            if j > 0:
                with launch_group():
                    nxt = {}
                    for c in live:
                        w, _ = wb[c][j]
                        wt, ldw = entries[c][j].dgrad[0]
                        sig = sns[c][j][0]
                        nxt[c] = gather_gemm(d[c], wt, ldw, w.shape[1], 1, d[c].shape[3], cls0['ty'], cls0['tx'], 1, 1,
                                             act=ACT_DLRELU, res=ys[c][j - 1], wscale=sig[1:2])
                for c in live:
                    d[c] = nxt[c]
            else:
                terms = {}
                with launch_group():
                    for c in live:
                        w, _ = wb[c][0]
                        wt, ldw = entries[c][0].dgrad[0]
                        sig = sns[c][0][0]
                        terms.setdefault(chains[c][0], []).append(
                            gather_gemm(d[c], wt, ldw, w.shape[1], 1, d[c].shape[3], cls0['ty'], cls0['tx'], 1, 1,
                                        wscale=sig[1:2]))
                jobs = [(lev, ts) for lev, ts in sorted(terms.items()) if ctx.needs_input_grad[1 + lev]]
                for lev, ts in jobs:
                    drows[lev] = ts[0] if len(ts) == 1 else empty_nhwc(1, ts[0].shape[1], 1, ts[0].shape[3], ts[0])
                multi = [(lev, ts) for lev, ts in jobs if len(ts) > 1]
                for i in range(0, len(multi), 8):
                    _sum_terms([(drows[lev], ts) for lev, ts in multi[i:i + 8]])
        outs = [None]
        for lev in range(nlev):
            g = drows[lev]
            outs.append(None if g is None or not ctx.needs_input_grad[1 + lev]
                        else g.permute(0, 2, 3, 1).reshape(g.shape[3], g.shape[1]))
        outs += [None] * (len(ctx.needs_input_grad) - len(outs))
        return tuple(outs)


def _sum_terms(jobs):
    """jobs: [(dst, [src ...])] dense fp32 tensors of equal numel per job (at most 8 jobs, any number of terms): dst = sum of
    its terms, left to right, in one launch per four terms (csrc/wgrad_finalize.hip fsv_sum_terms)"""
    lib.register_sigs({"fsv_sum_terms": [c_p, c_p, c_p, c_p, c_i, c_p]})
    while jobs:
        dsts = (ctypes.c_void_p * len(jobs))(*[d.data_ptr() for d, _ in jobs])
        srcs = (ctypes.c_void_p * (4 * len(jobs)))()
        ns = []
        for j, (d, ts) in enumerate(jobs):
            take = ts[:4]
            ns.append(len(take))
            for t, src in enumerate(take):
                srcs[4 * j + t] = src.data_ptr()
        lib.check_device(*[d for d, _ in jobs])
        lib.call("fsv_sum_terms", dsts, srcs, lib.int_array(ns), _ll([d.numel() for d, _ in jobs]), len(jobs), lib.stream_ptr())
        jobs = [(d, [d] + ts[4:]) for d, ts in jobs if len(ts) > 4]


def mlp_bank(rows, chains):
    """rows: list of [R_l, c_l] tensors (one per level); chains: list of (level, [modules with weight_orig / bias / _sn()]).
    Returns one [R_l, out] tensor per chain, or None when the grouped path does not apply (the caller then runs the chains
    Linear by Linear through ops.linear)."""
    from . import conv as _conv
    # the staging-time narrowing kernels have no grouped form; under `--amp` on the half-precision kernels the bank runs as in the
    # exact mode - grouped fp32 launches: the weight generators (6 GFLOP of the step's 1.8 TFLOP per frame) produce PARAMETERS of
    # the SPADE layers, 100 small GEMMs that one by one cost more in launches than in arithmetic
    if not _conv.group_enabled() or _conv.narrow_staging_mode() != 0:
        return None
    grad = torch.is_grad_enabled() and (any(r.requires_grad for r in rows) or
                                        any(m.weight_orig.requires_grad for _, ms in chains for m in ms))
    geom = Geom(1, 1, 1, 0)
    entries = []
    for lev, ms in chains:
        es = []
        for m in ms:
            w, b = m.weight_orig, m.bias
            if not getattr(m, 'spectral', False) or b is None or w.dim() != 2 or rows[lev].shape[1] % 4 != 0 and m is ms[0]:
                return None
            cache = getattr(w, '_fsv_cache', None)
            e = cache.lookup(w, (w.shape[0], w.shape[1], 1, 1), geom, 0) if cache is not None else None
            if e is None or w.shape[1] % 4 != 0:
                return None
            if grad:
                ok = (getattr(w, '_fsv_sink', False) and w.grad is not None and getattr(w, '_fsv_finalizer', None) is not None
                      and getattr(b, '_fsv_sink', False) and b.grad is not None and getattr(b, '_fsv_finalizer', None) is not None
                      and w.requires_grad and b.requires_grad)
                if not ok:
                    return None
            es.append(e)
        entries.append(es)
    sns = []
    for lev, ms in chains:
        per = []
        for m in ms:
            sig, u, v, owned = _unpack_sn(m._sn())
            per.append((sig, u if owned else u.clone(), v if owned else v.clone()))
        sns.append(per)
    meta = dict(chains=[(lev, len(ms)) for lev, ms in chains], sn=sns, entries=entries, nlevels=len(rows))
    flat = list(rows)
    for lev, ms in chains:
        for m in ms:
            flat += [m.weight_orig, m.bias]
    return list(_MlpBankFn.apply(meta, *flat))


class _PooledFn(torch.autograd.Function):
    """prod[b, i, j] = sum_p a[b, i, p] * sm[b, j, p]: the softmax pooling of the reference encoder (generator.py:378-389,
    `torch.bmm(x, softmax(label).transpose)`), a and sm NCHW tensors in channels-last memory.

    Both operands lie in memory as [position p][channel]: the product over positions is exactly the shape of a per-sample 1x1
    WEIGHT-GRADIENT GEMM, `dwt[ci][co] = sum_pixels in[pixel][ci] * dout[pixel][co]`, whose kernel reads both of them in place
    (round 6; before: a transposed copy of each operand + a K-major re-arrangement + a gather-GEMM whose "pixels" were the 32 ...
    1024 channels - four launches per level and pass, 17 - 26 us for the GEMM alone on the 16x16 maps).  One launch, one split
    (direct stores, fixed summation order).  Backward in the operands' own layouts as well: d a[p][i] = sum_j sm[p][j] dprod[i][j] and
    d sm[p][j] = sum_i a[p][i] dprod[i][j] are per-sample 1x1 convolutions over the positions whose K-major weight operands are
    dprod re-arranged (a c x c matrix) and dprod itself."""

    @staticmethod
    def forward(ctx, a, sm):
        a_, sm_ = to_nhwc(a), to_nhwc(sm)
        b, c, h, w = a_.shape
        g1 = Geom(1, 1, 1, 0)
        dwt = conv_wgrad(a_, sm_, g1, (c, c, 1, 1), per_sample=True, raw=True, force_split=1)       # [b, c, c] (c % 32 == 0)
        ctx.save_for_backward(a_, sm_)
        return dwt.view(b, c, c, 1)

    @staticmethod
    def backward(ctx, dprod):
        a_, sm_ = ctx.saved_tensors
        b, c, h, w = a_.shape
        g1 = Geom(1, 1, 1, 0)
        d = dprod.reshape(b, c, c).contiguous()
        da = dsm = None
        if ctx.needs_input_grad[0]:
            wt, _, ldw = prep_weight(d.view(b, c, c, 1, 1), 0, g1)            # K-major [j][i] per sample
            da = conv_forward(sm_, wt, ldw, c, g1, per_sample=True)
        if ctx.needs_input_grad[1]:
            dsm = conv_forward(a_, d, c, c, g1, per_sample=True)              # K-major [i][j] = dprod as it lies in memory
        return da, dsm


def pooled_product_ready(a, sm):
    """can _PooledFn take this pair?  (channel counts a multiple of 32: no padding rows / columns in the c x c results; the float4
    weight-gradient kernel's geometry; the exact-fp32 kernels.)  Opt-in, FSV_POOL_WGRAD=1: measured in-box against the gather-GEMM
    form over three alternating pairs - 42.49 / 42.42 / 42.46 ms per step with the gather-GEMM, 42.48 / 42.59 / 42.55 with this one
    (profiles/r06_notes.md section 9): the ~40 launches it removes sit on the reference-encoder chain, which runs next to the flow
    branch and is not what the step waits for."""
    if a.shape != sm.shape or a.dim() != 4 or a.dtype != torch.float32 or sm.dtype != torch.float32:
        return False
    b, c, h, w = a.shape
    return (c % 32 == 0 and 32 // w + 1 <= h and _conv.narrow_staging_mode() == 0 and
            _os.environ.get('FSV_POOL_WGRAD', '0') == '1')


def pooled_product(a, sm):
    """[b, c(i), c(j), 1] = sum over positions of a[b, i, p] * sm[b, j, p] (see _PooledFn)"""
    return _PooledFn.apply(a, sm)


def batch_conv(x, weight, bias=None, act=ACT_NONE, stride=1, allow_half=True):
    """Per-sample 1x1 (or kxk) convolution with generated weights [B, Cout, Cin, k, k] (base_network.py:56-71);
    stride 1 or 2 (padding k // 2, as the reference).  allow_half=False: a call site that only borrows the kernel for a
    batched matrix product (torch.bmm in the reference: softmax pooling, attention) and stays fp32 under `--amp`."""
    if weight is None:
        return x
    k = weight.shape[-1]
    geom = Geom(k, k, int(stride), k // 2)
    # weights / biases are usually strided views into the weight-generating FC's output: conv.prep_weight / gather_gemm
    # read them in place (sample stride), no copies here
    _note_grad_mode()
    return _ConvFn.apply(x, weight, bias, None, None, None, None, geom, act, 1.0, False, 0, allow_half)


# ------------------------------------------------------------------------------------------------ normalisation
# Cross-replica BatchNorm statistics (opt-in).  The reference's multi-process path makes every BatchNorm of the generator an
# apex.parallel.SyncBatchNorm (models/networks/normalization.py:15,33,80): statistics over the GLOBAL batch.  Default here is
# per-replica statistics (DESIGN.md "Multi-GPU"); set_bn_sync(world, group) switches every train-mode BatchNorm site (affine,
# and the parameter-free one inside SPADE) to: local fp64 sums -> one all-reduce of 2C doubles -> mean / rstd, and the same for
# the two sums of the backward pass.  Every rank must hold the same number of pixels per site (equal per-rank batches).
_bn_sync = None        # (world_size, process group) or None


def set_bn_sync(world_size=1, group=None):
    global _bn_sync
    _bn_sync = (int(world_size), group) if world_size and int(world_size) > 1 else None


def bn_sync_world(groups=1, instance=False):
    """number of replicas whose statistics are pooled for a BatchNorm site (InstanceNorm is never pooled - also not at one
    sample per replica, where it has a single group like BatchNorm)"""
    return _bn_sync[0] if (_bn_sync is not None and groups == 1 and not instance) else 1


def norm_stats(x, groups, pixels, channels, eps, run_mean=None, run_var=None, momentum=0.1, rep=1, instance=False):
    """rep > 1: the normalised tensor repeats every value of x `rep` times (nearest x2 up-sampling folded into the consumer)"""
    mean = torch.empty(groups * channels, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    lib.check_device(x, run_mean, run_var)
    ws = _ws(groups, pixels, channels, x)
    if bn_sync_world(groups, instance) > 1:
        import torch.distributed as dist
        sums = torch.empty(2 * channels, dtype=torch.float64, device=x.device)
        lib.call("fsv_norm_sums", lib.ptr(x), lib.ptr(ws), lib.ptr(sums), pixels, channels, lib.stream_ptr())
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=_bn_sync[1])
        lib.call("fsv_norm_stats_from_sums", lib.ptr(sums), float(pixels) * _bn_sync[0], lib.ptr(mean), lib.ptr(rstd),
                 channels, float(eps), lib.ptr(run_mean), lib.ptr(run_var), float(momentum), lib.stream_ptr())
        return mean, rstd
    lib.call("fsv_norm_stats_fused", lib.ptr(x), lib.ptr(ws), lib.ptr(mean), lib.ptr(rstd),
             groups, pixels, channels, float(eps), lib.ptr(run_mean), lib.ptr(run_var), float(momentum), int(rep),
             _ticket(x), lib.stream_ptr())
    return mean, rstd


def bn_backward(dy, y, x, mean, rstd, w, g, p, c, act, fixed_stats, affine, world=1):
    """dx (and dw, db when affine) of a normalisation whose forward pooled its statistics over `world` replicas"""
    dx = torch.empty_like(x)
    if world > 1 and not fixed_stats:
        import torch.distributed as dist
        ws = _ws(1, p, c, x)
        sums = torch.empty(2 * c, dtype=torch.float64, device=x.device)
        lib.call("fsv_norm_bwd_sums", lib.ptr(dy), lib.ptr(y), lib.ptr(x), lib.ptr(mean), lib.ptr(rstd), lib.ptr(ws),
                 lib.ptr(sums), p, c, act, lib.stream_ptr())
        # parameter gradients are the local sums: they are averaged over the replicas with every other gradient
        db = sums[:c].float() if affine else None
        dw = sums[c:].float() if affine else None
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=_bn_sync[1])
        s = sums.float()
        lib.call("fsv_norm_bwd_apply", lib.ptr(dy), lib.ptr(y), lib.ptr(x), lib.ptr(mean), lib.ptr(rstd), lib.ptr(w),
                 lib.ptr(s[:c]), lib.ptr(s[c:]), lib.ptr(dx), p, c, p * world, act, None, lib.stream_ptr())
        return dx, dw, db
    s1 = torch.empty(g * c, dtype=torch.float32, device=x.device)
    s2 = torch.empty_like(s1)
    dw = torch.empty(c, dtype=torch.float32, device=x.device) if affine else None
    db = torch.empty_like(dw) if affine else None
    ws = _ws(g, p, c, x)
    with _hconv.half_side_output(dx) as side:
        lib.call("fsv_norm_bwd_fused", lib.ptr(dy), lib.ptr(y), lib.ptr(x), lib.ptr(mean), lib.ptr(rstd), lib.ptr(w), lib.ptr(ws),
                 lib.ptr(s1), lib.ptr(s2), lib.ptr(dx), lib.ptr(dw), lib.ptr(db), g, p, c, act, 1 if fixed_stats else 0,
                 _ticket(x), side.ptr(), lib.stream_ptr())
    return dx, dw, db


class _NormActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, run_mean, run_var, instance, eps, momentum, act, training):
        x_arg = x
        x = to_nhwc(x)
        n, c, h, w = x.shape
        g, p = (n, h * w) if instance else (1, n * h * w)
        if training or instance or run_mean is None:
            got = None if (instance and bn_sync_world(g, True) > 1) else _stats_from_producer(
                x_arg, g, p, c, eps, None if instance else run_mean, None if instance else run_var, momentum)
            mean, rstd = got if got is not None else norm_stats(
                x, g, p, c, eps, None if instance else run_mean, None if instance else run_var, momentum, instance=instance)
        else:
            mean = run_mean.detach().clone()
            rstd = torch.rsqrt(run_var.detach() + eps)
        y = torch.empty_like(x)
        wd = weight.detach().contiguous() if weight is not None else None
        bd = bias.detach().contiguous() if bias is not None else None
        with _hconv.half_side_output(y) as side:
            lib.call("fsv_norm_apply", lib.ptr(x), lib.ptr(mean), lib.ptr(rstd), lib.ptr(wd), lib.ptr(bd), lib.ptr(y), g, p,
                     c, act, side.ptr(), lib.stream_ptr())
        ctx.dims = (g, p, c)
        ctx.act, ctx.affine = act, weight is not None
        ctx.batch_stats = bool(training or instance or run_mean is None)
        ctx.world = bn_sync_world(g, instance) if ctx.batch_stats else 1
        ctx.save_for_backward(x, y, mean, rstd, wd if wd is not None else mean)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, rstd, wd = ctx.saved_tensors
        g, p, c = ctx.dims
        dy = to_nhwc(dy)
        dx, dw, db = bn_backward(dy, y, x, mean, rstd, wd if ctx.affine else None, g, p, c, ctx.act, not ctx.batch_stats,
                                 ctx.affine, ctx.world)
        return dx, dw, db, None, None, None, None, None, None, None


def norm_act(x, weight=None, bias=None, run_mean=None, run_var=None, instance=False, eps=1e-5, momentum=0.1,
             act=ACT_NONE, training=True):
    return _NormActFn.apply(x, weight, bias, run_mean, run_var, instance, eps, momentum, act, training)


# ------------------------------------------------------------------------------------------------ SPADE
_spade_tls = _threading.local()


class spade_pair:
    """`with spade_pair():` around the two SPADE sites of one SPADEResnetBlock that normalise the SAME tensor with the same
    maps (bn_s and bn_0, architecture.py:95-96,103): the first site's modulation launch is held back and issued together with
    the second one's as ONE two-site launch (csrc/spade.hip NS = 2: x, the statistics and the label-map tiles are read once).
    Nobody may read the first site's output before the block ends.  Autograd is untouched (two nodes, two backward twins).
    Opt-in (FSV_SPADE_PAIR=1): measured in-box in round 3 (profiles/r03_notes.md) the two-site launch is no faster than the two
    single-site launches (51.64 vs 51.45 ms per step) - its 80 KB of LDS and 206 registers leave one or two workgroups per CU, and
    the reads it saves (x at a quarter of the resolution, the 32-channel maps) are L2 hits for the second single-site launch
    anyway."""

    def __enter__(self):
        self.pending = None
        self.outer = getattr(_spade_tls, 'pair', None)
        _spade_tls.pair = self if _os.environ.get('FSV_SPADE_PAIR', '0') == '1' else None
        return self

    def __exit__(self, et, ev, tb):
        _spade_tls.pair = self.outer
        if self.pending is not None and et is None:
            site, self.pending = self.pending, None
            _spade_launch(site)
        return False


class spade_into_conv:
    """`with spade_into_conv():` around `x_s = conv_s(bn_s(x, maps))` of a SPADEResnetBlock (architecture.py:103-108): the
    modulation launch of bn_s is held back, and the 1x1 convolution that consumes its output issues BOTH as one kernel
    (csrc/spade_conv.hip: the modulated tensor stays in registers between the two GEMMs).  In a forward that keeps no graph the
    modulated tensor is never written; a training forward gets it as a side output (conv_s' weight gradient reads it) - autograd
    is untouched (the same two nodes, the same two backward passes).  A consumer the fused kernel does not cover (other widths,
    `--amp`, a convolution with bias / residual) launches the held-back modulation first and proceeds as usual.
    FSV_SPADE_CONV_S=0 switches the fusion off (in-box A/B).

    conv3=True (round 6): around `conv_0(actvn(bn_0(x)))` / `conv_1(actvn(bn_1(dx)))` (architecture.py:96-99) - the 3x3 consumer
    issues modulation + activation + convolution as one kernel (csrc/spade_conv3.hip: the modulated haloed tile lives in LDS) where
    that kernel covers the widths.  Built and measured against the two launches (profiles/r06_notes.md section 8); an opt-in:
    FSV_SPADE_CONV3=1."""

    def __init__(self, conv3=False):
        self.conv3 = bool(conv3)

    def __enter__(self):
        self.pending = None
        self.outer = getattr(_spade_tls, 'defer', None)
        if self.conv3:
            _spade_tls.defer = self if spade_conv3_enabled() else None
        else:
            _spade_tls.defer = self if _os.environ.get('FSV_SPADE_CONV_S', '1') == '1' else None
        return self

    def __exit__(self, et, ev, tb):
        _spade_tls.defer = self.outer
        if self.pending is not None and et is None:
            site, self.pending = self.pending, None
            _spade_launch(site)
        return False


def spade_pair_enabled():
    return _os.environ.get('FSV_SPADE_PAIR', '0') == '1'


def spade_conv3_enabled():
    return _os.environ.get('FSV_SPADE_CONV3', '0') == '1'


def _spade_pending_for(x):
    """the held-back modulation whose output tensor is `x` (consumed: the caller launches it, fused or not), or None"""
    defer = getattr(_spade_tls, 'defer', None)
    site = defer.pending if defer is not None else None
    if site is None:
        return None
    defer.pending = None
    if site['h'].data_ptr() != x.data_ptr():
        _spade_launch(site)                 # somebody else's input: nothing to fuse with
        return None
    return site


def _spade_conv_s_fits(site, geom, per_sample, bias, res, act, scale, half, cpad, st_wanted, cout):
    """every condition of the fused bn_s -> conv_s launch (csrc/spade_conv.hip fsv_spade_conv_s_impl), evaluated from shapes alone
    before anything is launched: the widths the kernel is instantiated for AND the per-launch limits the entry point checks (map
    channels a multiple of 4 / 8, tensors below the 2 GiB descriptor range, even geometry behind a folded up-sampling) - a site
    that passes here cannot come back FSV_ERR_UNSUPPORTED after the held-back modulation has been given up (round-4 advisor)"""
    if not (geom.kh == 1 and geom.kw == 1 and geom.stride == 1 and geom.pad == 0 and not per_sample and bias is None and
            res is None and act == ACT_NONE and scale == 1.0 and cpad == 0 and not st_wanted):
        return False
    f16 = bool(site.get('f16'))
    if half != f16 or bool(site.get('half')) != f16:
        return False
    n, hw, c, _, w, up = site['dims']
    chs = site['chs']
    if lib.call_status("fsv_spade_conv_s_supported", c, cout, len(chs)) != 1:
        return False
    if any(ch < 1 or ch % (8 if f16 else 4) for ch in chs):
        return False
    lim = 2 ** 31
    if hw * c * 4 > lim or any(hw * ch * 4 > lim for ch in chs):
        return False
    if up and (w < 2 or w % 2 or hw % w or (hw // w) % 2):
        return False
    return True


def _spade_conv_s_launch(site, wt, ldws, cout, wscale, want_hs):
    """x_s = conv_s(bn_s(x)) in one launch; returns x_s (NHWC storage, logical NCHW).  wt / ldws: the K-major fp32 forward operand of
    the 1x1 weight, or - for a site on the f16 GEMMs (`--amp`) - its N-major half twin and row length"""
    arr = lambda v: (ctypes.c_void_p * max(len(v), 1))(*v)
    n, hw, c, ldw, w, up = site['dims']
    chs = site['chs']
    hgt = hw // w
    xs = empty_nhwc(n, cout, hgt, w, site['x'])
    lib.check_device(site['x'], wt, wscale)
    f16 = bool(site.get('f16'))
    with profile.scope('fsv_spade_conv_s_kernel' + ('[f16]' if f16 else '') +
                       (' P%d C%d N%d K%s' % (n * hw, c, cout, '+'.join(map(str, chs))) if profile.detail() else ''),
                       site['flops'] + 2.0 * n * hw * c * cout):
        if f16:
            lib.call("fsv_spade_conv_s_fwd_h", lib.ptr(site['x']), lib.ptr(site['mean']), lib.ptr(site['rstd']),
                     lib.ptr(site['h']) if want_hs else None, lib.ptr(xs), len(chs), _pp(site['maps']), arr(site['wg']),
                     arr(site['wb']), arr(site['bg']), arr(site['bb']), lib.int_array(chs + [0]), _ll(site['wstr'] + [0]),
                     _ll(site['bstr'] + [0]), n, hw, c, 0, w, up, lib.ptr(wt), ldws, cout, lib.ptr(wscale), lib.stream_ptr())
            if _hconv.launch_hook() is not None:
                _hconv.launch_hook()('spade_conv_s', dict(site=site, wt=wt, cout=cout, wscale=wscale, xs=xs, want_hs=want_hs))
        else:
            lib.call("fsv_spade_conv_s_fwd", lib.ptr(site['x']), lib.ptr(site['mean']), lib.ptr(site['rstd']),
                     lib.ptr(site['h']) if want_hs else None, lib.ptr(xs), len(chs), _pp(site['maps']), arr(site['wg']),
                     arr(site['wb']), arr(site['bg']), arr(site['bb']), lib.int_array(chs + [0]), _ll(site['wstr'] + [0]),
                     _ll(site['bstr'] + [0]), n, hw, c, ldw, 0, w, up, lib.ptr(wt), ldws, cout, lib.ptr(wscale), lib.stream_ptr())
    return xs


def _spade_conv3_fits(site, geom, per_sample, res, act, scale, half, cpad, cout, up):
    """every condition of the fused bn -> actvn -> 3x3 convolution launch (csrc/spade_conv3.hip fsv_spade_conv3_fwd), from shapes alone
    (as _spade_conv_s_fits: a site that passes here cannot come back FSV_ERR_UNSUPPORTED).  A bias, a residual and a statistics hint
    are fine (the statistics of the output are then reduced by the normalisation that follows, in a pass of its own)."""
    if not (site.get('conv3') and geom.kh == 3 and geom.kw == 3 and geom.stride == 1 and geom.pad == 1 and not per_sample and
            act == ACT_NONE and scale == 1.0 and cpad == 0 and not half and not up):
        return False
    if site.get('f16') or site.get('half') or site['act'] not in (ACT_NONE, ACT_LRELU):
        return False
    n, hw, c, _, w, s_up = site['dims']
    chs = site['chs']
    if lib.call_status("fsv_spade_conv3_supported", c, cout, len(chs)) != 1:
        return False
    if any(ch < 1 or ch % 4 for ch in chs):
        return False
    lim = 2 ** 31
    if hw * c * 4 > lim or hw * cout * 4 > lim or any(hw * ch * 4 > lim for ch in chs):
        return False
    if s_up and (w % 2 or (hw // w) % 2):
        return False
    if res is not None and tuple(res.shape) != (n, cout, hw // w, w):
        return False
    return True


def _spade_conv3_launch(site, wt, ldwc, cout, wscale, bias, res, want_hs, st=None):
    """out = conv3x3(actvn(bn(x))) (+ bias, + res) in one launch; returns out (NHWC storage, logical NCHW).  st: the statistics hint
    of conv_forward ({'groups': 1}: a BatchNorm follows) - filled with the epilogue's partials like the gather-GEMM's"""
    arr = lambda v: (ctypes.c_void_p * max(len(v), 1))(*v)
    n, hw, c, ldw, w, up = site['dims']
    chs = site['chs']
    hgt = hw // w
    out = empty_nhwc(n, cout, hgt, w, site['x'])
    if res is not None:
        res = to_nhwc(res)
    lib.check_device(site['x'], wt, wscale, bias, res)
    part, prezeroed = None, False
    if st is not None and int(st['groups']) == 1 and _conv.stats_enabled():
        part = _conv.stats_arena(out.device).take(_conv.STATS_SLOTS * cout * 2)
        prezeroed = part is not None
        if part is None:
            part = torch.empty(_conv.STATS_SLOTS * cout * 2, dtype=torch.float64, device=out.device)
    with profile.scope('fsv_spade_conv3_kernel' + (' P%d C%d N%d K%s' % (n * hw, c, cout, '+'.join(map(str, chs))) if profile.detail() else ''),
                       site['flops'] + 2.0 * n * hw * 9 * c * cout):
        lib.call("fsv_spade_conv3_fwd", lib.ptr(site['x']), lib.ptr(site['mean']), lib.ptr(site['rstd']),
                 lib.ptr(site['h']) if want_hs else None, lib.ptr(out), len(chs), _pp(site['maps']), arr(site['wg']),
                 arr(site['wb']), arr(site['bg']), arr(site['bb']), lib.int_array(chs + [0]), _ll(site['wstr'] + [0]),
                 _ll(site['bstr'] + [0]), n, hgt, w, c, ldw, 0, up, site['act'], lib.ptr(wt), ldwc, cout, lib.ptr(bias),
                 lib.ptr(res), lib.ptr(wscale), lib.ptr(part), _conv.STATS_SLOTS, 1 if prezeroed else 0, lib.stream_ptr())
    if part is not None:
        st['part'], st['slots'] = part, _conv.STATS_SLOTS
    return out


def _spade_same_input(a, b):
    return (a['x'].data_ptr() == b['x'].data_ptr() and a['dims'] == b['dims'] and a['chs'] == b['chs'] and
            [m.data_ptr() for m in a['maps']] == [m.data_ptr() for m in b['maps']])


def _half_map(key, m):
    """IEEE-half copy of a label map (NHWC), made once per map tensor: every SPADE site of a block and their backward twins and
    weight-gradient GEMMs read the same copy"""
    got = getattr(key, '_fsv_h16', None)
    if got is not None and got[0] == key._version and got[1].shape == m.shape:
        return got[1]
    mh = _hconv.to_half_nhwc(m)
    try:
        key._fsv_h16 = (key._version, mh)
    except Exception:
        pass
    return mh


def _spade_launch(a, b=None):
    arr = lambda v: (ctypes.c_void_p * max(len(v), 1))(*v)
    n, hw, c, ldw, w, up = a['dims']
    chs = a['chs']
    if b is None:
        with profile.scope('fsv_spade_mod_kernel' + (' P%d C%d K%s' % (n * hw, c, '+'.join(map(str, chs))) if profile.detail() else ''),
                           a['flops']):
            if a.get('half'):
                lib.call("fsv_spade_mod_fwd_h", lib.ptr(a['x']), lib.ptr(a['mean']), lib.ptr(a['rstd']), lib.ptr(a['h']), len(chs),
                         _pp(a['maps']), arr(a['wg']), arr(a['wb']), arr(a['bg']), arr(a['bb']), lib.int_array(chs + [0]),
                         _ll(a['wstr'] + [0]), _ll(a['bstr'] + [0]), n, hw, c, ldw, 0, a['act'], w, up,
                         1 | (4 if a.get('f16') else 0), lib.stream_ptr())
                if _hconv.launch_hook() is not None:
                    _hconv.launch_hook()('spade_fwd', dict(site=a))
            else:
                lib.call("fsv_spade_mod_fwd", lib.ptr(a['x']), lib.ptr(a['mean']), lib.ptr(a['rstd']), lib.ptr(a['h']), len(chs),
                         _pp(a['maps']), arr(a['wg']), arr(a['wb']), arr(a['bg']), arr(a['bb']), lib.int_array(chs + [0]),
                         _ll(a['wstr'] + [0]), _ll(a['bstr'] + [0]), n, hw, c, ldw, 0, a['act'], w, up, lib.stream_ptr())
        return
    with profile.scope('fsv_spade_mod_kernel', a['flops'] + b['flops']):
        lib.call("fsv_spade_mod_fwd2", lib.ptr(a['x']), lib.ptr(a['mean']), lib.ptr(a['rstd']), lib.ptr(a['h']), lib.ptr(b['h']),
                 len(chs), _pp(a['maps']), arr(a['wg'] + b['wg']), arr(a['wb'] + b['wb']), arr(a['bg'] + b['bg']),
                 arr(a['bb'] + b['bb']), lib.int_array(chs + [0]), _ll(a['wstr'] + b['wstr'] + [0]),
                 _ll(a['bstr'] + b['bstr'] + [0]), n, hw, c, ldw, 0, a['act'], b['act'], w, up, lib.stream_ptr())


def _streams_mod():
    from . import streams
    return streams


class _SpadeFn(torch.autograd.Function):
    """h = act(spade(x; maps, weights)).  Argument list: x, run_mean, run_var, then per map (map, wg, wb, bg, bb).

    wg/wb are OIHW 1x1 weights: [C, Ch, 1, 1] (fixed) or [B, C, Ch, 1, 1] (generated per sample).
    """

    @staticmethod
    def forward(ctx, act, training, eps, momentum, x, run_mean, run_var, *rest):
        # act may be (act, up): up = 1 -> x is the HALF-resolution tensor; the kernels read it through the nearest x2
        # up-sampling index (generator.py:124 folded into SPADE: the up-sampled tensor is never written) and the statistics
        # are those of x with every value counted four times
        up = 0
        if isinstance(act, tuple):
            act, up = act
        x_arg = x
        x = to_nhwc(x)
        n, c, h, w = x.shape
        xs_h, xs_w = h, w                   # geometry of x as stored
        if up:
            h, w = 2 * h, 2 * w             # geometry of the normalised / modulated tensor
        nmaps = len(rest) // 5
        maps = [to_nhwc(rest[5 * k]) for k in range(nmaps)]
        wgs = [rest[5 * k + 1] for k in range(nmaps)]
        wbs = [rest[5 * k + 2] for k in range(nmaps)]
        bgs = [rest[5 * k + 3] for k in range(nmaps)]
        bbs = [rest[5 * k + 4] for k in range(nmaps)]
        if training or run_mean is None:
            got = _stats_from_producer(x_arg, 1, n * xs_h * xs_w, c, eps, run_mean, run_var, momentum, rep=4 if up else 1)
            mean, rstd = got if got is not None else norm_stats(x, 1, n * xs_h * xs_w, c, eps, run_mean, run_var, momentum,
                                                                rep=4 if up else 1)
        else:
            mean = run_mean.detach().clone()
            rstd = torch.rsqrt(run_var.detach() + eps)
        g1 = Geom(1, 1, 1, 0)
        for k in range(nmaps):
            if tuple(maps[k].shape[2:]) != (h, w):
                raise ValueError("SPADE maps must already be at the resolution of x")
        chs = [m.shape[1] for m in maps]
        ctx.up = up
        lib.check_device(x, *maps)
        # fast path (every production width): ONE preparation launch per map builds the combined [gamma | beta] operands
        # that the modulation kernel, the backward recompute and the data gradient all use as they are
        ctx.fast = (c % 16 == 0) and _os.environ.get('FSV_SPADE_FAST', '1') == '1'
        # `--amp` on the half-precision kernels: the modulated tensor is only ever read by convolutions (conv_0 / conv_1 / conv_s of
        # a SPADEResnetBlock, architecture.py:92-99) that would round it to half anyway - the kernel rounds at the store and the
        # fp32 tensor + conversion pass disappear; its gradient then arrives as half (the convolutions' data gradient) and the
        # backward twin reads it as such
        ctx.half_out = bool(_conv.h_kernels() and ctx.fast and _os.environ.get('FSV_SPADE_FUSED_BWD', '1') == '1'
                            and getattr(_spade_tls, 'pair', None) is None and nmaps > 0)
        hout = _hconv.empty_nhwc_h(n, c, h, w, x) if ctx.half_out else empty_nhwc(n, c, h, w, x)
        # ... and the gamma / beta GEMMs themselves on the f16 matrix instructions: half label maps (one conversion per map tensor,
        # shared by every SPADE site that reads it and by the backward), half weights, half d(gamma|beta) for the half data /
        # weight-gradient kernels, the bias gradients from the backward twin's own epilogue
        ctx.f16 = bool(ctx.half_out and all(ch % 8 == 0 for ch in chs) and _os.environ.get('FSV_SPADE_F16', '1') == '1')
        if ctx.f16:
            maps = [_half_map(rest[5 * k], maps[k]) for k in range(nmaps)]
        if ctx.fast:
            ldw = 2 * c
            prepped, wg_p, wb_p, bg_p, bb_p, wstr, bstr = [], [], [], [], [], [], []
            for k in range(nmaps):
                wg, wb, bg, bb = wgs[k].detach(), wbs[k].detach(), bgs[k].detach(), bbs[k].detach()
                per_sample = wg.dim() == 5
                ch = chs[k]

                def inner_ok(t):
                    return t.stride(-4) == ch and t.stride(-3) == 1
                if not inner_ok(wg):
                    wg = wg.contiguous()
                if not inner_ok(wb):
                    wb = wb.contiguous()
                if bg.stride(-1) != 1:
                    bg = bg.contiguous()
                if bb.stride(-1) != 1:
                    bb = bb.contiguous()
                nb = n if per_sample else 1
                kt, kd, ld = (ch + 31) // 32 * 32, (2 * c + 31) // 32 * 32, (ch + 31) // 32 * 32
                # fixed (optimiser-owned) weights change in the optimiser step only: their combined operands are built once per
                # step - by the first SPADE call that meets them, i.e. the discriminator step's generator pass - and reused by the
                # generator-mode pass and its backward (LayoutCache.epoch: bumped by the refresh that ends every optimiser step;
                # `_version` catches load_state_dict / copy_).  Round 6: 22 of the 46 preparation launches per step.
                owner = getattr(wgs[k], '_fsv_cache', None) if (not per_sample and _os.environ.get('FSV_SPADE_PREP_CACHE', '1') == '1') else None
                if owner is not None:
                    key = (owner.epoch, wgs[k]._version, wbs[k]._version, bgs[k]._version, bbs[k]._version, wg.data_ptr(),
                           wb.data_ptr(), bg.data_ptr(), bb.data_ptr(), c, ch, bool(ctx.f16))
                    hit = getattr(wgs[k], '_fsv_spade_prep', None)
                    if hit is not None and hit[0] == key:
                        wcat_x, wcat_d, bcat = hit[1]
                        if len(hit) > 2 and hit[2] is not None and x.is_cuda:
                            cur_s = torch.cuda.current_stream(x.device)
                            if cur_s != hit[3]:          # built by the other of two passes issued next to each other (streams.CROSS)
                                cur_s.wait_event(hit[2])
                                for t_ in hit[1]:
                                    t_.record_stream(cur_s)
                        prepped += [wcat_x, wcat_d, bcat]
                        if ctx.f16:
                            wg_p.append(wcat_x.data_ptr()); wb_p.append(wcat_x.data_ptr() + 2 * c * kt)
                        else:
                            wg_p.append(wcat_x.data_ptr()); wb_p.append(wcat_x.data_ptr() + 4 * c)
                        wstr.append(0)
                        bg_p.append(bcat.data_ptr()); bb_p.append(bcat.data_ptr() + 4 * c)
                        bstr.append(0)
                        continue
                # + 64 floats of slack: the beta half is addressed as (wcat_t + C) with the same row stride, so a channel
                # tile that overhangs C (C not a multiple of the 32 / 64-wide tile) reads past the last row's end; those
                # columns only feed output channels >= C, which are never stored
                flat_t = torch.empty(nb * kt * 2 * c + 64, dtype=torch.float32, device=x.device)
                wcat_t = flat_t[:nb * kt * 2 * c].view(nb, kt, 2 * c)
                need_d = (ctx.needs_input_grad[7 + 5 * k] and getattr(_outer_grad, 'on', True)) or owner is not None      # (cached: whichever pass comes next may need it)
                wcat_d = torch.empty((nb, kd, ld), dtype=torch.float32, device=x.device) if need_d else None
                bcat = torch.empty((nb, 2 * c), dtype=torch.float32, device=x.device)
                lib.check_device(wg, wb, bg, bb)
                lib.call("fsv_spade_prep", lib.ptr(wg), lib.ptr(wb), lib.ptr(bg), lib.ptr(bb),
                         wg.stride(0) if per_sample else 0, wb.stride(0) if per_sample else 0,
                         bg.stride(0) if per_sample else 0, bb.stride(0) if per_sample else 0,
                         lib.ptr(wcat_t), lib.ptr(wcat_d), lib.ptr(bcat), nb, c, ch, lib.stream_ptr())
                if ctx.f16:
                    # N-major half operand [nb][gamma rows (C) | beta rows (C)][Kh]
                    wcat_h = torch.empty((nb, 2 * c, kt), dtype=torch.float16, device=x.device)
                    lib.call("fsv_spade_prep_h", lib.ptr(wg), lib.ptr(wb), wg.stride(0) if per_sample else 0,
                             wb.stride(0) if per_sample else 0, lib.ptr(wcat_h), nb, c, ch, lib.stream_ptr())
                    prepped += [wcat_h, wcat_d if need_d else flat_t[:0], bcat]
                    wg_p.append(wcat_h.data_ptr()); wb_p.append(wcat_h.data_ptr() + 2 * c * kt)
                    wstr.append(2 * c * kt if per_sample else 0)
                else:
                    prepped += [wcat_t, wcat_d if need_d else wcat_t[:0], bcat]
                    wg_p.append(wcat_t.data_ptr()); wb_p.append(wcat_t.data_ptr() + 4 * c)
                    wstr.append(kt * 2 * c if per_sample else 0)
                if owner is not None:
                    ev_ = st_ = None
                    if _streams_mod().CROSS and x.is_cuda:
                        st_ = torch.cuda.current_stream(x.device)
                        ev_ = torch.cuda.Event()
                        ev_.record(st_)
                    wgs[k]._fsv_spade_prep = (key, tuple(prepped[-3:]), ev_, st_)
                bg_p.append(bcat.data_ptr()); bb_p.append(bcat.data_ptr() + 4 * c)
                bstr.append(2 * c if per_sample else 0)
            site = dict(x=x, mean=mean, rstd=rstd, h=hout, maps=maps, wg=wg_p, wb=wb_p, bg=bg_p, bb=bb_p, chs=chs, wstr=wstr,
                        bstr=bstr, dims=(n, h * w, c, ldw, w, up), act=act, keep=prepped, half=ctx.half_out, f16=ctx.f16,
                        flops=2.0 * n * h * w * c * 2 * sum(chs))
            pair = getattr(_spade_tls, 'pair', None)
            defer = getattr(_spade_tls, 'defer', None)
            conv3 = bool(defer is not None and defer.conv3)
            if (defer is not None and defer.pending is None and pair is None and nmaps > 0 and
                    ((not conv3 and (ctx.f16 or not ctx.half_out) and act == ACT_NONE and c in (64, 128)) or
                     (conv3 and not ctx.half_out and act in (ACT_NONE, ACT_LRELU) and c == 64))):
                site['conv3'] = conv3
                defer.pending = site            # the convolution that reads hout issues both (spade_into_conv)
            elif pair is None or nmaps == 0:
                _spade_launch(site)
            elif pair.pending is None:
                pair.pending = site             # the partner site of this block issues both (spade_pair)
            else:
                first, pair.pending = pair.pending, None
                if _spade_same_input(first, site):
                    _spade_launch(first, site)
                else:
                    _spade_launch(first)
                    _spade_launch(site)
            ctx.nmaps, ctx.act = nmaps, act
            ctx.batch_stats = bool(training or run_mean is None)
            ctx.world = bn_sync_world(1) if ctx.batch_stats else 1
            ctx.per_sample = [wgs[k].dim() == 5 for k in range(nmaps)]
            ctx.w_shapes = [tuple(wgs[k].shape) for k in range(nmaps)]
            ctx.save_for_backward(x, hout, mean, rstd, *maps, *prepped)
            return hout
        wg_t, wb_t, bg_c, bb_c, wbs_stride, bbs_stride = [], [], [], [], [], []
        ldw = (c + 31) // 32 * 32
        for k in range(nmaps):
            per_sample = wgs[k].dim() == 5
            tg, kpad, _ = prep_weight(wgs[k].detach(), 0, g1)
            tb, _, _ = prep_weight(wbs[k].detach(), 0, g1)
            wg_t.append(tg); wb_t.append(tb)
            bg_c.append(bgs[k].detach().contiguous()); bb_c.append(bbs[k].detach().contiguous())
            wbs_stride.append(kpad * ldw if per_sample else 0)
            bbs_stride.append(c if per_sample else 0)
        with profile.scope('fsv_spade_mod_kernel', 2.0 * n * h * w * c * 2 * sum(chs)):
            lib.call("fsv_spade_mod_fwd", lib.ptr(x), lib.ptr(mean), lib.ptr(rstd), lib.ptr(hout), nmaps, _pp(maps),
                     _pp(wg_t), _pp(wb_t), _pp(bg_c), _pp(bb_c), lib.int_array(chs + [0]), _ll(wbs_stride + [0]),
                     _ll(bbs_stride + [0]), n, h * w, c, ldw, 0, act, w, up, lib.stream_ptr())
        ctx.nmaps, ctx.act = nmaps, act
        ctx.batch_stats = bool(training or run_mean is None)
        ctx.world = bn_sync_world(1) if ctx.batch_stats else 1
        ctx.save_for_backward(x, hout, mean, rstd, *maps, *wgs, *wbs, *bgs, *bbs)
        return hout

    @staticmethod
    def backward(ctx, dh):
        nm = ctx.nmaps
        saved = ctx.saved_tensors
        x, hout, mean, rstd = saved[:4]
        maps = saved[4:4 + nm]
        dh = to_nhwc(dh)
        n, c, xs_h, xs_w = x.shape
        up = ctx.up
        h, w = (2 * xs_h, 2 * xs_w) if up else (xs_h, xs_w)
        g1 = Geom(1, 1, 1, 0)
        fast = ctx.fast
        gbs, wcats = [], []
        dxhat = empty_nhwc(n, c, h, w, x)
        fused = fast and _os.environ.get('FSV_SPADE_FUSED_BWD', '1') == '1'          # in-box A/B switch
        if fused:
            # fused backward twin of the modulation kernel: gamma / beta are recomputed in registers, never materialised
            prepped = saved[4 + nm:]
            f16 = getattr(ctx, 'f16', False)
            dgbs = [(_hconv.empty_nhwc_h if f16 else empty_nhwc)(n, 2 * c, h, w, x) for _ in range(nm)]
            wg_p, wb_p, bg_p, bb_p, wstr, bstr, chs = [], [], [], [], [], [], []
            for k in range(nm):
                wcat_t, bcat = prepped[3 * k], prepped[3 * k + 2]            # f16: the half operand [nb][2C][Kh]
                wg_p.append(wcat_t.data_ptr())
                wb_p.append(wcat_t.data_ptr() + (2 * c * wcat_t.shape[-1] if f16 else 4 * c))
                bg_p.append(bcat.data_ptr()); bb_p.append(bcat.data_ptr() + 4 * c)
                wstr.append(wcat_t.shape[-2] * wcat_t.shape[-1] if ctx.per_sample[k] else 0)
                bstr.append(2 * c if ctx.per_sample[k] else 0)
                chs.append(maps[k].shape[1])
            arr = lambda v: (ctypes.c_void_p * max(len(v), 1))(*v)
            lib.check_device(x, dh, *maps)
            dbsum = None
            # bias gradients (per-channel sums of d(gamma|beta)) from the twin's own epilogue instead of a column-sum pass over the
            # [P][2C] tensors it writes: on the f16 GEMMs always; on the exact-fp32 kernels an opt-in (FSV_SPADE_DBSUM=1) - measured
            # neutral there (profiles/r04_notes.md section 10: the grouped column-sum pass costs what the fp64 atomics cost), and
            # the fixed-order mode wants the read pass anyway
            twin_sums = f16 or (_os.environ.get('FSV_SPADE_DBSUM', '0') == '1' and _os.environ.get('FSV_DETERMINISTIC', '0') != '1'
                                and any(ctx.needs_input_grad[7 + 5 * k + 3] or ctx.needs_input_grad[7 + 5 * k + 4]
                                        for k in range(nm)))
            if twin_sums and nm:
                # (h w / 64 pixel tiles) x 2 waves add into every address: spread over copies beyond 512 adds per address
                slots = int(_os.environ.get('FSV_SPADE_DB_SLOTS', '0'))          # (tests force the multi-copy form on small maps)
                while slots < 64 and (slots < 1 or (h * w // 32) // slots > 512):
                    slots = max(1, slots * 2)
                dbsum = torch.empty((slots, n, nm, 2 * c), dtype=torch.float64, device=x.device)
                zstr = [nm * 2 * c if ctx.per_sample[k] else 0 for k in range(nm)]
            # (labelled as the backward twin; FLOPs = the gamma / beta GEMMs it recomputes)
            with profile.scope('fsv_spade_mod_kernel<bwd>' + (' P%d C%d K%s' % (n * h * w, c, '+'.join(map(str, chs)))
                                                              if profile.detail() else ''), 2.0 * n * h * w * c * 2 * sum(chs)):
                if dh.dtype == torch.float16 or f16 or dbsum is not None:
                    flags = (1 if dh.dtype == torch.float16 else 0) | (6 if f16 else 0)
                    lib.call("fsv_spade_mod_bwd_h", lib.ptr(x), lib.ptr(mean), lib.ptr(rstd), lib.ptr(dh), nm, _pp(maps), arr(wg_p),
                             arr(wb_p), arr(bg_p), arr(bb_p), lib.int_array(chs + [0]), _ll(wstr + [0]), _ll(bstr + [0]),
                             _pp(dgbs), lib.ptr(dxhat), n, h * w, c, 2 * c, 0, ctx.act, w, up, flags,
                             lib.ptr(dbsum) if dbsum is not None else None, _ll(zstr + [0]) if dbsum is not None else None,
                             slots if dbsum is not None else 1, ctypes.c_longlong(n * nm * 2 * c), lib.stream_ptr())
                    if dbsum is not None:
                        dbsum = dbsum.sum(0, dtype=torch.float32) if slots > 1 else dbsum[0].float()
                    if _hconv.launch_hook() is not None:
                        _hconv.launch_hook()('spade_bwd', dict(x=x, mean=mean, rstd=rstd, dh=dh, maps=list(maps), prepped=list(prepped),
                                                              per_sample=list(ctx.per_sample), dgbs=dgbs, dxhat=dxhat, dbsum=dbsum,
                                                              act=ctx.act, up=up, f16=f16))
                else:
                    lib.call("fsv_spade_mod_bwd", lib.ptr(x), lib.ptr(mean), lib.ptr(rstd), lib.ptr(dh), nm, _pp(maps), arr(wg_p),
                             arr(wb_p), arr(bg_p), arr(bb_p), lib.int_array(chs + [0]), _ll(wstr + [0]), _ll(bstr + [0]),
                             _pp(dgbs), lib.ptr(dxhat), n, h * w, c, 2 * c, 0, ctx.act, w, up, lib.stream_ptr())
        else:
            if fast:
                prepped = saved[4 + nm:]
                for k in range(nm):
                    wcat_t, bcat = prepped[3 * k], prepped[3 * k + 2]
                    gbs.append(conv_forward(maps[k], wcat_t, 2 * c, 2 * c, g1, bias=bcat, per_sample=ctx.per_sample[k]))
            else:
                wgs = saved[4 + nm:4 + 2 * nm]
                wbs = saved[4 + 2 * nm:4 + 3 * nm]
                bgs = saved[4 + 3 * nm:4 + 4 * nm]
                bbs = saved[4 + 4 * nm:4 + 5 * nm]
                # 1) recompute gamma|beta of every map with the gather-GEMM kernel ([P][2C] each)
                for k in range(nm):
                    per_sample = wgs[k].dim() == 5
                    wcat = torch.cat([wgs[k].detach(), wbs[k].detach()], dim=-4)
                    bcat = torch.cat([bgs[k].detach(), bbs[k].detach()], dim=-1).contiguous()
                    wt, _, ldw = prep_weight(wcat, 0, g1)
                    gbs.append(conv_forward(maps[k], wt, ldw, 2 * c, g1, bias=bcat, per_sample=per_sample))
                    wcats.append(wcat)
            # 2) elementwise chain backward
            dgbs = [torch.empty_like(gb) for gb in gbs]
            lib.call("fsv_spade_bwd_elem", lib.ptr(x), lib.ptr(mean), lib.ptr(rstd), lib.ptr(dh), lib.ptr(hout), nm,
                     _pp(gbs), _pp(dgbs), lib.ptr(dxhat), n, h * w, c, 0, ctx.act, w, up, lib.stream_ptr())
        # 3) param-free BatchNorm backward.  With the up-sampling folded in, xhat of the four children of a source pixel
        # is the same value, so with dxhat summed over the children the backward is exactly the BatchNorm backward of
        # the half-resolution tensor (s1 = sum dxhat, s2 = sum dxhat * xhat, count = source pixels).
        dx = None
        if ctx.needs_input_grad[4]:
            if up:
                pooled = empty_nhwc(n, c, xs_h, xs_w, dxhat)
                lib.call("fsv_upsample2x_bwd", lib.ptr(dxhat), lib.ptr(pooled), n, xs_h, xs_w, c, lib.stream_ptr())
                dxhat = pooled
            if ctx.batch_stats:
                dx, _, _ = bn_backward(dxhat, None, x, mean, rstd, None, 1, n * xs_h * xs_w, c, ACT_NONE, False, False,
                                       ctx.world)
            else:
                dx = dxhat * rstd.view(1, c, 1, 1)
        grads = []
        for k in range(nm):
            per_sample = ctx.per_sample[k] if fast else wgs[k].dim() == 5
            base = 7 + 5 * k
            dm = dwg = dwb = dbg = dbb = None
            if fast:
                ch = maps[k].shape[1]
                if ctx.needs_input_grad[base]:
                    wcat_d = prepped[3 * k + 1]
                    dm = conv_dgrad(dgbs[k], None, g1, (h, w), per_sample=per_sample,
                                    cached=[(wcat_d, wcat_d.shape[-1])], cin=ch)
                if ctx.needs_input_grad[base + 1] or ctx.needs_input_grad[base + 2]:
                    # operands swapped: rows = the 2C gamma|beta channels, columns = map channels, which IS the OIHW
                    # layout of a 1x1 weight - no re-arrangement pass afterwards
                    dwt = conv_wgrad(dgbs[k], maps[k], g1, (ch, 2 * c, 1, 1), per_sample=per_sample, raw=True)
                    dwcat = dwt[:, :2 * c, :ch]
                    dwcat = dwcat.unsqueeze(-1).unsqueeze(-1) if per_sample else dwcat[0].unsqueeze(-1).unsqueeze(-1)
                    dwg, dwb = dwcat.narrow(-4, 0, c), dwcat.narrow(-4, c, c)
                    if dwg.shape != ctx.w_shapes[k]:
                        dwg, dwb = dwg.reshape(ctx.w_shapes[k]), dwb.reshape(ctx.w_shapes[k])
            else:
                if ctx.needs_input_grad[base]:
                    dm = conv_dgrad(dgbs[k], wcats[k], g1, (h, w), per_sample=per_sample)
                if ctx.needs_input_grad[base + 1] or ctx.needs_input_grad[base + 2]:
                    dwcat = conv_wgrad(maps[k], dgbs[k], g1, tuple(wcats[k].shape), per_sample=per_sample)
                    dwg, dwb = torch.split(dwcat, c, dim=-4)
            if ctx.needs_input_grad[base + 3] or ctx.needs_input_grad[base + 4]:
                if fused and dbsum is not None:
                    dbcat = dbsum[:, k] if per_sample else dbsum[0, k]
                elif per_sample:
                    dbcat = colsum(dgbs[k], n, h * w, 2 * c)
                else:
                    dbcat = colsum(dgbs[k], 1, n * h * w, 2 * c).view(2 * c)
                dbg, dbb = torch.split(dbcat, c, dim=-1)
            grads += [dm, dwg, dwb, dbg, dbb]
        return (None, None, None, None, dx, None, None, *grads)


def spade_mod(x, maps, weights, run_mean=None, run_var=None, act=ACT_LRELU, training=True, eps=1e-5, momentum=0.1, up=False):
    """maps: list of tensors; weights: list of (wg, wb, bg, bb) per map (see _SpadeFn).  up=True: x is at half the
    resolution of the maps and stands for its nearest x2 up-sampling (never materialised)."""
    flat = []
    for m, (wg, wb, bg, bb) in zip(maps, weights):
        flat += [m, wg, wb, bg, bb]
    _note_grad_mode()
    return _SpadeFn.apply((act, 1) if up else act, training, eps, momentum, x, run_mean, run_var, *flat)


def spade_can_fold_upsample():
    """the folded form needs per-replica statistics (the cross-replica path exchanges sums of the materialised tensor)"""
    return bn_sync_world(1) == 1 and _os.environ.get('FSV_SPADE_FOLD', '1') == '1'            # env: in-box A/B switch


# ------------------------------------------------------------------------------------------------ upsample
class _Up2xFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, 2 * h, 2 * w, x)
        lib.check_device(x)
        lib.call("fsv_upsample2x_fwd", lib.ptr(x), lib.ptr(y), n, h, w, c, lib.stream_ptr())
        ctx.dims = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w = ctx.dims
        dy = to_nhwc(dy)
        dx = empty_nhwc(n, c, h, w, dy)
        lib.call("fsv_upsample2x_bwd", lib.ptr(dy), lib.ptr(dx), n, h, w, c, lib.stream_ptr())
        return dx


def upsample2x(x):
    return _Up2xFn.apply(x)


# ------------------------------------------------------------------------------------------------ flow warp
_lin_cache = {}


def _linspace(n, device):
    key = (n, str(device))
    t = _lin_cache.get(key)
    if t is None:
        # built on the host exactly like the reference's get_grid (base_network.py:13-26), then uploaded
        from . import streams
        t = streams.shared(lambda: torch.linspace(-1.0, 1.0, n).to(device))
        _lin_cache[key] = t
    return t


def warp_taps(image, flow):
    """Forward warp that also returns the integer tap indices (x_w, y_n) - used by the index-parity tests."""
    b, c, h, w = image.shape
    out = torch.empty((b, c, h, w), dtype=torch.float32, device=image.device)
    taps = torch.empty((b, h, w, 2), dtype=torch.int32, device=image.device)
    lib.check_device(image, flow)
    lib.call("fsv_warp_fwd", lib.ptr(image), lib.ptr(flow), lib.ptr(_linspace(w, image.device)),
             lib.ptr(_linspace(h, image.device)), lib.ptr(out), lib.ptr(taps), b, c, h, w, _ll(image.stride()),
             _ll(flow.stride()), _ll(out.stride()), lib.stream_ptr())
    return out, taps


class _WarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, flow):
        b, c, h, w = image.shape
        if flow.shape != (b, 2, h, w):
            raise ValueError("flow must be [B, 2, H, W]")
        image = image if image.stride(-1) in (1, c) else image.contiguous()
        out = torch.empty((b, c, h, w), dtype=torch.float32, device=image.device)
        lib.check_device(image, flow)
        lib.call("fsv_warp_fwd", lib.ptr(image), lib.ptr(flow), lib.ptr(_linspace(w, image.device)),
                 lib.ptr(_linspace(h, image.device)), lib.ptr(out), None, b, c, h, w, _ll(image.stride()),
                 _ll(flow.stride()), _ll(out.stride()), lib.stream_ptr())
        ctx.save_for_backward(image, flow)
        return out

    @staticmethod
    def backward(ctx, gout):
        image, flow = ctx.saved_tensors
        b, c, h, w = image.shape
        gout = gout.contiguous()
        gimg = torch.zeros((b, c, h, w), dtype=torch.float32, device=image.device) if ctx.needs_input_grad[0] else None
        gflow = torch.empty((b, 2, h, w), dtype=torch.float32, device=image.device) if ctx.needs_input_grad[1] else None
        z4 = _ll([0, 0, 0, 0])
        lib.call("fsv_warp_bwd", lib.ptr(image), lib.ptr(flow), lib.ptr(_linspace(w, image.device)),
                 lib.ptr(_linspace(h, image.device)), lib.ptr(gout), lib.ptr(gimg), lib.ptr(gflow), b, c, h, w,
                 _ll(image.stride()), _ll(flow.stride()), _ll(gout.stride()),
                 _ll(gimg.stride()) if gimg is not None else z4, _ll(gflow.stride()) if gflow is not None else z4,
                 lib.stream_ptr())
        return gimg, gflow


def resample(image, flow):
    """Drop-in for models.networks.base_network.resample (file:28-37)."""
    return _WarpFn.apply(image, flow)


lib.register_sigs({
    "fsv_warp_compose_fwd": [c_p] * 8 + [c_i] * 5 + [c_llp] * 6 + [c_p],
    "fsv_warp_compose_bwd": [c_p] * 12 + [c_i] * 5 + [c_llp] * 7 + [c_p],
})
COMPOSE_CONCAT, COMPOSE_BLEND = 0, 1


class _WarpComposeFn(torch.autograd.Function):
    """(warp, comp) = the flow warp of `image` and its composite with the occlusion mask in one launch (csrc/warp.hip):
    comp = cat([warp, mask], 1) (mode 0, --spade_combine: generator.py:441-443) or raw * mask + warp * (1 - mask) (mode 1,
    generator.py:217,224).  Same taps as resample()."""

    @staticmethod
    def forward(ctx, image, flow, mask, raw, mode):
        b, c, h, w = image.shape
        if flow.shape != (b, 2, h, w) or mask.shape != (b, 1, h, w):
            raise ValueError("flow must be [B, 2, H, W] and mask [B, 1, H, W]")
        warp = torch.empty((b, c, h, w), dtype=torch.float32, device=image.device)
        if mode == COMPOSE_CONCAT:
            comp = empty_nhwc(b, c + 1, h, w, image)
            raw_t, cstr = None, None
        else:
            raw_t = raw
            comp = torch.empty((b, c, h, w), dtype=torch.float32, device=image.device)
            cstr = _ll(comp.stride())
        lib.check_device(image, flow, mask, raw_t)
        mstr = _ll([mask.stride(0), mask.stride(2), mask.stride(3)])
        lib.call("fsv_warp_compose_fwd", lib.ptr(image), lib.ptr(flow), lib.ptr(_linspace(w, image.device)),
                 lib.ptr(_linspace(h, image.device)), lib.ptr(mask), lib.ptr(raw_t), lib.ptr(warp), lib.ptr(comp), mode,
                 b, c, h, w, _ll(image.stride()), _ll(flow.stride()), mstr,
                 _ll(raw_t.stride()) if raw_t is not None else None, _ll(warp.stride()), cstr, lib.stream_ptr())
        ctx.mode = mode
        ctx.save_for_backward(image, flow, mask, raw_t if raw_t is not None else image)
        return warp, comp

    @staticmethod
    def backward(ctx, g_warp, g_comp):
        image, flow, mask, raw_t = ctx.saved_tensors
        mode = ctx.mode
        b, c, h, w = image.shape
        if mode == COMPOSE_CONCAT:
            raw_t = None
            if g_comp is not None:
                g_comp = to_nhwc(g_comp)
        dev = image.device
        gimg = torch.zeros_like(image) if ctx.needs_input_grad[0] else None
        gflow = torch.empty((b, 2, h, w), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        gmask = torch.empty((b, 1, h, w), dtype=torch.float32, device=dev) if ctx.needs_input_grad[2] else None
        graw = torch.empty((b, c, h, w), dtype=torch.float32, device=dev) if (mode == COMPOSE_BLEND and
                                                                                ctx.needs_input_grad[3]) else None
        mstr = _ll([mask.stride(0), mask.stride(2), mask.stride(3)])
        lib.call("fsv_warp_compose_bwd", lib.ptr(image), lib.ptr(flow), lib.ptr(_linspace(w, dev)), lib.ptr(_linspace(h, dev)),
                 lib.ptr(mask), lib.ptr(raw_t), lib.ptr(g_warp), lib.ptr(g_comp), lib.ptr(gimg), lib.ptr(gflow), lib.ptr(gmask),
                 lib.ptr(graw), mode, b, c, h, w, _ll(image.stride()), _ll(flow.stride()), mstr,
                 _ll(raw_t.stride()) if raw_t is not None else None,
                 _ll(g_warp.stride()) if g_warp is not None else None,
                 _ll(g_comp.stride()) if (g_comp is not None and mode == COMPOSE_BLEND) else None,
                 _ll(gimg.stride()) if gimg is not None else None, lib.stream_ptr())
        return gimg, gflow, gmask, graw, None


def warp_concat(image, flow, mask):
    """(resample(image, flow), cat([that, mask], dim=1)) - the warped image and the image-embedding input - in one launch"""
    return _WarpComposeFn.apply(image, flow, mask, None, COMPOSE_CONCAT)


def warp_blend(raw, image, flow, mask):
    """(resample(image, flow), raw * mask + that * (1 - mask)) in one launch"""
    return _WarpComposeFn.apply(image, flow, mask, raw, COMPOSE_BLEND)


# ------------------------------------------------------------------------------------------------ Adam
def adam_step(param, grad, m, v, state, beta1, beta2, eps, gscale=1.0, tick=None):
    """Fused Adam on flat buffers; state = [t, 1-b1^t, 1-b2^t, lr] (device, fp32).  tick (True / False): one piece of a step
    issued range by range - exactly one piece advances the step count, the others are ordered behind it (FlatAdam)."""
    lib.check_device(param, grad, m, v, state)
    if tick is None:
        lib.call("fsv_adam_step", lib.ptr(param), lib.ptr(grad), lib.ptr(m), lib.ptr(v), lib.ptr(state), param.numel(),
                 float(beta1), float(beta2), float(eps), float(gscale), lib.stream_ptr())
    else:
        lib.call("fsv_adam_step_range", lib.ptr(param), lib.ptr(grad), lib.ptr(m), lib.ptr(v), lib.ptr(state), param.numel(),
                 float(beta1), float(beta2), float(eps), float(gscale), 1 if tick else 0, lib.stream_ptr())


lib.register_sigs({
    "fsv_amp_check": [c_p, c_ll, c_p, c_p],
    "fsv_amp_adam": [c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_f, c_f, c_f, c_f, c_p],
    "fsv_amp_update": [c_p, c_p],
})


def amp_adam_step(param, grad, m, v, state, scaler, beta1, beta2, eps, gscale=1.0):
    """Adam step of the fp16-operand mode (csrc/amp.hip): overflow test of the scaled gradients, the step with
    grad * gscale / scale (skipped when a gradient is not finite), and apex's scale update - all on the device."""
    lib.check_device(param, grad, m, v, state, scaler)
    st = lib.stream_ptr()
    hook = _hconv.launch_hook()
    before = dict(param=param.clone(), m=m.clone(), v=v.clone(), state=state.clone(), scaler=scaler.clone()) if hook is not None else None
    lib.call("fsv_amp_check", lib.ptr(grad), grad.numel(), lib.ptr(scaler), st)
    lib.call("fsv_amp_adam", lib.ptr(param), lib.ptr(grad), lib.ptr(m), lib.ptr(v), lib.ptr(state), lib.ptr(scaler),
             param.numel(), float(beta1), float(beta2), float(eps), float(gscale), st)
    lib.call("fsv_amp_update", lib.ptr(scaler), st)
    if hook is not None:
        hook('adam', dict(before=before, param=param, grad=grad, m=m, v=v, state=state, scaler=scaler, beta1=float(beta1),
                          beta2=float(beta2), eps=float(eps), gscale=float(gscale)))


# ------------------------------------------------------------------------------------------------ losses / packing / masks
lib.register_sigs({
    "fsv_l1_fwd": [c_p, c_p, c_f, c_p, c_i, c_i, c_ll, c_llp, c_llp, c_p, c_p, c_p, c_p],
    "fsv_l1_bwd": [c_p, c_p, c_f, c_p, c_i, c_i, c_ll, c_llp, c_llp, c_p, c_p, c_p, c_p, c_p],
    "fsv_wsum_fwd": [c_pp, ctypes.POINTER(ctypes.c_float), c_i, c_p, c_p],
    "fsv_wsum_bwd": [ctypes.POINTER(ctypes.c_float), c_i, c_p, c_p, c_p],
    "fsv_hinge_fwd": [c_p, c_ll, c_f, c_p, c_p, c_p, c_p],
    "fsv_hinge_bwd": [c_p, c_ll, c_f, c_p, c_p, c_p],
    "fsv_pack_d_input": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_ll, c_llp, c_llp, c_llp, c_llp, c_p],
    "fsv_unpack_d_grad": [c_p, c_p, c_i, c_i, c_i, c_i, c_ll, c_p],
    "fsv_pool15": [c_p, c_p, c_i, c_i, c_i, c_ll, c_ll, c_ll, c_i, c_f, c_p],
    "fsv_part_masks": [c_p, c_p, c_ll, c_ll, c_i, c_ll, c_ll, c_i, c_i, c_p],
})


def _dense4(t):
    """a 4-D tensor whose memory is dense NCHW or dense NHWC (no copy when it already is)"""
    if t.is_contiguous() or t.permute(0, 2, 3, 1).is_contiguous():
        return t
    return t.contiguous()


def _ncp_strides(t):
    """(batch, channel, pixel) element strides of a dense NCHW / NHWC 4-D tensor"""
    n, c, h, w = t.shape
    if t.is_contiguous():
        return [c * h * w, h * w, 1]
    return [h * w * c, 1, c]


class _L1Fn(torch.autograd.Function):
    """mean |a*m - b*m| (nn.L1Loss / MaskedL1Loss, models/networks/loss.py:130-138).  b: tensor or python float;
    m: None or a [N, 1, H, W] mask broadcast over channels.  All three tensor inputs may require grad."""

    @staticmethod
    def forward(ctx, a, b, m):
        a = _dense4(a)
        bconst = 0.0
        bt = None
        if isinstance(b, torch.Tensor):
            bt = _dense4(b)
            if bt.shape != a.shape:
                raise ValueError("l1: shape mismatch")
        else:
            bconst = float(b)
        n, c, h, w = a.shape
        mt = None
        if m is not None:
            if m.shape != (n, 1, h, w):
                raise ValueError("l1: mask must be [N, 1, H, W]")
            mt = m.contiguous()
        sa = _ncp_strides(a)
        sb = _ncp_strides(bt) if bt is not None else [0, 0, 0]
        flat = mt is None and (bt is None or sb == sa)
        if flat:        # same layout, no mask: one coalesced stream
            dims, sa_, sb_ = (1, 1, a.numel()), [0, 0, 1], ([0, 0, 1] if bt is not None else [0, 0, 0])
        else:
            dims, sa_, sb_ = (n, c, h * w), sa, sb
        part = torch.empty(512, dtype=torch.float64, device=a.device)
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        lib.check_device(a, bt, mt)
        lib.call("fsv_l1_fwd", lib.ptr(a), lib.ptr(bt), bconst, lib.ptr(mt), dims[0], dims[1], dims[2], _ll(sa_), _ll(sb_),
                 lib.ptr(part), lib.ptr(loss), _loss_ticket(a), lib.stream_ptr())
        ctx.meta = (dims, sa_, sb_, bconst, bt is not None, mt is not None)
        ctx.save_for_backward(a, bt if bt is not None else a, mt if mt is not None else a)
        return loss

    @staticmethod
    def backward(ctx, g):
        a, bt, mt = ctx.saved_tensors
        dims, sa_, sb_, bconst, has_b, has_m = ctx.meta
        g = g.contiguous()
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(bt) if (has_b and ctx.needs_input_grad[1]) else None
        dm = torch.empty_like(mt) if (has_m and ctx.needs_input_grad[2]) else None
        lib.call("fsv_l1_bwd", lib.ptr(a), lib.ptr(bt) if has_b else None, bconst, lib.ptr(mt) if has_m else None,
                 dims[0], dims[1], dims[2], _ll(sa_), _ll(sb_), lib.ptr(g), lib.ptr(da), lib.ptr(db), lib.ptr(dm),
                 lib.stream_ptr())
        return da, db, dm


def l1_loss(a, b, mask=None):
    return _L1Fn.apply(a, b, mask)


_wvec_cache = {}


def _loss_ticket(like):
    """the ticket of a loss reduction that finishes in its own launch (csrc/losses.hip fsv_loss_finish; round 6: a dozen 5-us
    finishing launches per step at the serial point between the forward and the backward pass).  None: the two-launch form - the
    default.  Opt-in (FSV_LOSS_TICKET=1): bit-identical, and measured SLOWER in-box, 42.41 / 42.52 / 42.53 ms per step without,
    42.92 / 42.87 / 42.85 with (profiles/r06_step_ab_serial_point.txt) - every workgroup of the fourteen launches pays a
    device-scope release (an L2 write-back on this part) in front of its ticket, which costs more than the 5-us launches it saves
    (the round-3 and round-5 findings about producer-side tickets, once more)."""
    if _os.environ.get('FSV_LOSS_TICKET', '0') != '1':
        return None
    return _conv.ticket_range(like, 64)


class _WsumFn(torch.autograd.Function):
    """sum_i w_i * t_i of one-element fp32 tensors: ONE launch (csrc/losses.hip fsv_wsum_fwd), ONE launch for all gradients"""

    @staticmethod
    def forward(ctx, weights, *terms):
        ctx.weights = weights
        ts = [t.detach() if t.is_contiguous() else t.detach().contiguous() for t in terms]
        out = torch.empty(1, dtype=torch.float32, device=ts[0].device)
        lib.check_device(*ts)
        wv = (ctypes.c_float * len(ts))(*weights)
        lib.call("fsv_wsum_fwd", (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts]), wv, len(ts), lib.ptr(out), lib.stream_ptr())
        return out

    @staticmethod
    def backward(ctx, g):
        n = len(ctx.weights)
        g = g.contiguous()
        d = torch.empty(n, dtype=torch.float32, device=g.device)
        lib.call("fsv_wsum_bwd", (ctypes.c_float * n)(*ctx.weights), n, lib.ptr(g), lib.ptr(d), lib.stream_ptr())
        return (None,) + tuple(d[i:i + 1] if ctx.needs_input_grad[1 + i] else None for i in range(n))


def weighted_sum(terms, weights=None):
    """sum_i weights[i] * terms[i] over one-element loss tensors -> shape [1].  The loss collector combines two dozen such
    scalars per iteration; one add / mul / div launch per term (and as many again in backward) became cat + mul + sum in round 2
    and is one launch each way since round 6 (FSV_WSUM=0: the torch form, in-box A/B)."""
    terms = [t.reshape(1) for t in terms]
    if weights is None:
        weights = [1.0] * len(terms)
    weights = [float(w) for w in weights]
    if len(terms) == 1:
        return terms[0] if weights[0] == 1.0 else terms[0] * weights[0]
    if (len(terms) <= 32 and all(t.dtype == torch.float32 for t in terms) and _os.environ.get('FSV_WSUM', '1') == '1'):
        return _WsumFn.apply(tuple(weights), *terms)
    cat = torch.cat(terms)
    if any(w != 1.0 for w in weights):
        key = (tuple(weights), cat.device, cat.dtype)
        wv = _wvec_cache.get(key)
        if wv is None:
            if len(_wvec_cache) > 256:
                _wvec_cache.clear()
            from . import streams
            wv = _wvec_cache[key] = streams.shared(lambda: torch.tensor(weights, dtype=cat.dtype).to(cat.device))
        cat = cat * wv
    return cat.sum(dim=0, keepdim=True)


class _HingeFn(torch.autograd.Function):
    """-mean(min(sign * x - 1, 0))  (GANLoss hinge, models/networks/loss.py:69-79): sign = +1 real, -1 fake."""

    @staticmethod
    def forward(ctx, x, sign):
        x = x if (x.is_contiguous() or (x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous())) else x.contiguous()
        part = torch.empty(512, dtype=torch.float64, device=x.device)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        lib.check_device(x)
        lib.call("fsv_hinge_fwd", lib.ptr(x), x.numel(), float(sign), lib.ptr(part), lib.ptr(loss), _loss_ticket(x), lib.stream_ptr())
        ctx.sign = float(sign)
        ctx.save_for_backward(x)
        return loss

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        lib.call("fsv_hinge_bwd", lib.ptr(x), x.numel(), ctx.sign, lib.ptr(g.contiguous()), lib.ptr(dx), lib.stream_ptr())
        return dx, None


def hinge_loss(x, real):
    return _HingeFn.apply(x, 1.0 if real else -1.0)


class _PackDFn(torch.autograd.Function):
    """Discriminator input [ref | label | image] written once in NHWC (loss_collector.py:47-58 builds it with three torch.cat +
    two repeat): fake and real stacked on the batch axis (real given), or one image set (real None: the G step's two
    discriminator passes - real images without autograd, generated images with - each pack their own input).  Only `fake`
    receives a gradient.  for_conv: the `--amp` form on the half-precision kernels - the tensor goes straight into the
    discriminator's first convolution, so it is written with the zero channels that convolution's float / half gather needs
    (76 -> 80) and as IEEE half: no padding pass, no conversion pass (ops._ConvFn recognises the pre-padded input)."""

    @staticmethod
    def forward(ctx, ref, lab, fake, real, for_conv):
        fake = _dense4(fake)
        real = _dense4(real) if real is not None else None
        b, ci, h, w = fake.shape
        cr = ref.shape[1] if ref is not None else 0
        cl = lab.shape[1] if lab is not None else 0
        ref = _dense4(ref) if ref is not None else None
        lab = _dense4(lab) if lab is not None else None
        halves = 2 if real is not None else 1
        ct = cr + cl + ci
        half = bool(for_conv and _conv.h_kernels())
        cto = (ct + 7) // 8 * 8 if half else ct
        out = (_hconv.empty_nhwc_h if half else empty_nhwc)(halves * b, cto, h, w, fake)
        z3 = _ll([0, 0, 0])
        lib.check_device(ref, lab, fake, real)
        lib.call("fsv_pack_d_x", lib.ptr(ref), lib.ptr(lab), lib.ptr(fake), lib.ptr(real), lib.ptr(out), b, cr, cl, ci,
                 h * w, _ll(_ncp_strides(ref)) if ref is not None else z3, _ll(_ncp_strides(lab)) if lab is not None else z3,
                 _ll(_ncp_strides(fake)), _ll(_ncp_strides(real)) if real is not None else z3, halves, cto, 1 if half else 0,
                 lib.stream_ptr())
        if half and _hconv.launch_hook() is not None:
            _hconv.launch_hook()('pack', dict(ref=ref, lab=lab, fake=fake, real=real, out=out, cto=cto))
        ctx.dims = (b, cr, cl, ci, h, w, cto)
        return out

    @staticmethod
    def backward(ctx, dout):
        b, cr, cl, ci, h, w, cto = ctx.dims
        dfake = None
        if ctx.needs_input_grad[2]:
            dout = to_nhwc(dout)
            dfake = torch.empty((b, ci, h, w), dtype=torch.float32, device=dout.device)
            lib.check_device(dout)
            lib.call("fsv_unpack_d_grad_h" if dout.dtype == torch.float16 else "fsv_unpack_d_grad", lib.ptr(dout), lib.ptr(dfake),
                     b, ci, cr + cl, cto, h * w, lib.stream_ptr())
        return None, None, dfake, None, None


lib.register_sigs({"fsv_pack_d_x": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_ll, c_llp, c_llp, c_llp, c_llp, c_i, c_i, c_i, c_p],
                   "fsv_unpack_d_grad_h": [c_p, c_p, c_i, c_i, c_i, c_i, c_ll, c_p]})


def pack_d_input(ref, lab, fake, real, for_conv=False):
    return _PackDFn.apply(ref, lab, fake, real, for_conv)


def pack_d_single(ref, lab, img, for_conv=False):
    return _PackDFn.apply(ref, lab, img, None, for_conv)


def part_masks(pose_ch, g0=0, ngroups=9):
    """pose_ch [B, T, H, W] (any batch / frame strides, contiguous rows) -> [B, T, ngroups, H, W] float masks of the
    DensePose part groups g0 .. g0+ngroups-1 (input_process.py:64-94; group 8 = face) in one launch."""
    b, t, h, w = pose_ch.shape
    if pose_ch.stride(3) != 1 or pose_ch.stride(2) != w:
        pose_ch = pose_ch.contiguous()
    y = torch.empty((b, t, ngroups, h, w), dtype=torch.float32, device=pose_ch.device)
    lib.check_device(pose_ch)
    lib.call("fsv_part_masks", lib.ptr(pose_ch), lib.ptr(y), b * t, h * w, t, pose_ch.stride(0), pose_ch.stride(1), g0, ngroups,
             lib.stream_ptr())
    return y


lib.register_sigs({
    "fsv_face_boxes": [c_p, c_ll, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p],
    "fsv_crop_resize_fwd": [c_p, c_ll, c_ll, c_ll, c_ll, c_i, c_p, c_p, c_i, c_i, c_p],
    "fsv_crop_resize_bwd": [c_p, c_p, c_p, c_ll, c_ll, c_ll, c_ll, c_i, c_i, c_i, c_p],
    "fsv_paste_face_fwd": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_ll, c_ll, c_ll, c_ll, c_p],
    "fsv_paste_face_bwd": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
})


def face_boxes(label, use_openpose, crop_smaller=0):
    """face_refiner.py:56-87 get_face_region for every sample of label [N, C, H, W]: int32 [N, 4] = (ys, ye, xs, xe),
    computed and kept on the device (the reference syncs the host through .nonzero() / .item() per sample)."""
    n, c, h, w = label.shape
    if label.stride(3) != 1 or label.stride(2) != w:
        label = label.contiguous()
    boxes = torch.empty((n, 4), dtype=torch.int32, device=label.device)
    lib.check_device(label)
    lib.call("fsv_face_boxes", lib.ptr(label), label.stride(0), label.stride(1), n, c, h, w, 1 if use_openpose else 0,
             int(crop_smaller), lib.ptr(boxes), lib.stream_ptr())
    return boxes


class _CropFaceFn(torch.autograd.Function):
    """face_refiner.py:32-39 crop_face_region: F.interpolate(image[i, -3:, ys:ye, xs:xe], size=(S, S)) for all i."""

    @staticmethod
    def forward(ctx, image, boxes, size):
        n, c, h, w = image.shape
        out = torch.empty((n, 3, size, size), dtype=torch.float32, device=image.device)
        lib.check_device(image, boxes)
        lib.call("fsv_crop_resize_fwd", lib.ptr(image), image.stride(0), image.stride(1), image.stride(2), image.stride(3),
                 c, lib.ptr(boxes), lib.ptr(out), n, size, lib.stream_ptr())
        ctx.save_for_backward(boxes)
        ctx.meta = (tuple(image.shape), size)
        return out

    @staticmethod
    def backward(ctx, dout):
        (boxes,) = ctx.saved_tensors
        (n, c, h, w), size = ctx.meta
        dimg = zeros_nhwc(n, c, h, w, dout)
        dout = dout.contiguous()
        lib.call("fsv_crop_resize_bwd", lib.ptr(dout), lib.ptr(boxes), lib.ptr(dimg), dimg.stride(0), dimg.stride(1),
                 dimg.stride(2), dimg.stride(3), c, n, size, lib.stream_ptr())
        return dimg, None, None


def crop_face(image, boxes, size):
    return _CropFaceFn.apply(image, boxes, size)


class _PasteFaceFn(torch.autograd.Function):
    """face_refiner.py:42-54 replace_face_region for the whole batch: the refined face (face [N, 3, S, S], already
    `fake_face + coarse`) is resized bilinearly to each sample's box, clamped to [-1, 1] and written over the image."""

    @staticmethod
    def forward(ctx, image, face, boxes):
        n, c, h, w = image.shape
        face = face.contiguous()
        out = torch.empty((n, 3, h, w), dtype=torch.float32, device=image.device)
        lib.check_device(image, face, boxes)
        lib.call("fsv_paste_face_fwd", lib.ptr(image), lib.ptr(face), lib.ptr(boxes), lib.ptr(out), n, h, w, face.shape[-1],
                 image.stride(0), image.stride(1), image.stride(2), image.stride(3), lib.stream_ptr())
        ctx.save_for_backward(out, boxes)
        ctx.size = face.shape[-1]
        return out

    @staticmethod
    def backward(ctx, dout):
        out, boxes = ctx.saved_tensors
        n, _, h, w = out.shape
        dout = dout.contiguous()
        dimg = torch.empty_like(out)
        dface = torch.zeros((n, 3, ctx.size, ctx.size), dtype=torch.float32, device=out.device)
        lib.call("fsv_paste_face_bwd", lib.ptr(dout), lib.ptr(out), lib.ptr(boxes), lib.ptr(dimg), lib.ptr(dface), n, h, w,
                 ctx.size, lib.stream_ptr())
        return dimg, dface, None


def paste_face(image, face, boxes):
    return _PasteFaceFn.apply(image, face, boxes)


def pool15(x, mode, thresh=0.0):
    """15x15 stride-1 pooling of a [N, 1, H, W] mask: mode 'max_gt' -> (maxpool(x) > thresh).float()
    (input_process.py:59-60), mode 'avg' -> AvgPool2d(15, padding=7, stride=1) (loss_collector.py:180)."""
    n, c, h, w = x.shape
    if c != 1:
        raise ValueError("pool15 expects a single-channel mask")
    y = torch.empty((n, 1, h, w), dtype=torch.float32, device=x.device)
    lib.check_device(x)
    lib.call("fsv_pool15", lib.ptr(x), lib.ptr(y), n, h, w, x.stride(0), x.stride(2), x.stride(3),
             0 if mode == 'max_gt' else 1, float(thresh), lib.stream_ptr())
    return y


lib.register_sigs({
    "fsv_cat_put": [c_p, c_p, c_ll, c_i, c_ll, c_llp, c_i, c_i, c_p],
    "fsv_pad_channels": [c_p, c_p, c_ll, c_i, c_ll, c_llp, c_i, c_p],
    "fsv_pad_channels_h": [c_p, c_p, c_ll, c_i, c_ll, c_llp, c_i, c_p],
    "fsv_cat_get": [c_p, c_p, c_ll, c_i, c_ll, c_i, c_i, c_p],
    "fsv_blend_fwd": [c_p, c_p, c_p, c_p, c_i, c_i, c_ll, c_llp, c_llp, c_llp, c_p],
    "fsv_blend_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_ll, c_llp, c_llp, c_llp, c_p],
})


class _CatFn(torch.autograd.Function):
    """torch.cat(tensors, dim=1) written once into channels-last memory (U-Net skips, flow-network input, ds_ref)."""

    @staticmethod
    def forward(ctx, *tensors):
        ts = [_dense4(t) for t in tensors]
        n, _, h, w = ts[0].shape
        cs = [t.shape[1] for t in ts]
        ct = sum(cs)
        out = empty_nhwc(n, ct, h, w, ts[0])
        lib.check_device(*ts)
        off = 0
        for t, c in zip(ts, cs):
            lib.call("fsv_cat_put", lib.ptr(t), lib.ptr(out), n, c, h * w, _ll(_ncp_strides(t)), ct, off, lib.stream_ptr())
            off += c
        ctx.meta = (n, h, w, cs, ct)
        return out

    @staticmethod
    def backward(ctx, dout):
        n, h, w, cs, ct = ctx.meta
        dout = to_nhwc(dout)
        grads, off = [], 0
        for i, c in enumerate(cs):
            if ctx.needs_input_grad[i]:
                g = empty_nhwc(n, c, h, w, dout)
                lib.call("fsv_cat_get", lib.ptr(dout), lib.ptr(g), n, c, h * w, ct, off, lib.stream_ptr())
                grads.append(g)
            else:
                grads.append(None)
            off += c
        return tuple(grads)


def cat_channels(tensors):
    return _CatFn.apply(*tensors)


def pad_channels_nhwc(x, cpad, half=False):
    """x [N, C, H, W] in any layout whose H x W plane is one strided run -> dense NHWC [N, C + cpad, H, W] with zero channels
    appended, in one launch (no autograd: callers slice the gradient themselves).  half: the result as IEEE half (the `--amp`
    path: padding + conversion of a convolution input in one pass) when the source planes are pixel-contiguous; otherwise the
    fp32 result (the caller converts)."""
    x = x.detach()
    n, c, h, w = x.shape
    if not (x.dtype == torch.float32 and (h == 1 or x.stride(2) == w * x.stride(3))):
        return to_nhwc(torch.nn.functional.pad(to_nhwc(x), (0, 0, 0, 0, 0, cpad)))
    lib.check_device(x)
    if half and x.stride(3) == 1 and (c + cpad) % 8 == 0 and n * h * w < (1 << 31):
        out = _hconv.empty_nhwc_h(n, c + cpad, h, w, x)
        lib.call("fsv_pad_channels_h", lib.ptr(x), lib.ptr(out), n, c, h * w, _ll([x.stride(0), x.stride(1), 1]), c + cpad,
                 lib.stream_ptr())
        return out
    out = empty_nhwc(n, c + cpad, h, w, x)
    lib.call("fsv_pad_channels", lib.ptr(x), lib.ptr(out), n, c, h * w, _ll([x.stride(0), x.stride(1), x.stride(3)]), c + cpad,
             lib.stream_ptr())
    return out


class _BlendFn(torch.autograd.Function):
    """a * m + b * (1 - m) with a [N, 1, H, W] soft occlusion mask (generator.py:217,224)."""

    @staticmethod
    def forward(ctx, a, b, m):
        a, b = _dense4(a), _dense4(b)
        m = m.contiguous()
        n, c, h, w = a.shape
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=a.device)
        lib.check_device(a, b, m)
        lib.call("fsv_blend_fwd", lib.ptr(a), lib.ptr(b), lib.ptr(m), lib.ptr(out), n, c, h * w, _ll(_ncp_strides(a)),
                 _ll(_ncp_strides(b)), _ll(_ncp_strides(out)), lib.stream_ptr())
        ctx.save_for_backward(a, b, m)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, m = ctx.saved_tensors
        g = _dense4(g)
        n, c, h, w = a.shape
        da = torch.empty((n, c, h, w), dtype=torch.float32, device=a.device) if ctx.needs_input_grad[0] else None
        db = torch.empty((n, c, h, w), dtype=torch.float32, device=a.device) if ctx.needs_input_grad[1] else None
        dm = torch.empty_like(m) if ctx.needs_input_grad[2] else None
        lib.call("fsv_blend_bwd", lib.ptr(a), lib.ptr(b), lib.ptr(m), lib.ptr(g), lib.ptr(da), lib.ptr(db), lib.ptr(dm), n, c,
                 h * w, _ll(_ncp_strides(a)), _ll(_ncp_strides(b)), _ll(_ncp_strides(g)), lib.stream_ptr())
        return da, db, dm


def blend(a, b, mask):
    return _BlendFn.apply(a, b, mask)


lib.register_sigs({
    "fsv_maxpool2_fwd": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "fsv_maxpool2_bwd": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
})


class _MaxPool2Fn(torch.autograd.Function):
    """nn.MaxPool2d(2, 2) of the VGG19 feature stack."""

    @staticmethod
    def forward(ctx, x):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, h // 2, w // 2, x)
        lib.check_device(x)
        lib.call("fsv_maxpool2_fwd", lib.ptr(x), lib.ptr(y), n, h, w, c, lib.stream_ptr())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        n, c, h, w = x.shape
        dy = to_nhwc(dy)
        dx = torch.empty_like(x)
        lib.call("fsv_maxpool2_bwd", lib.ptr(x), lib.ptr(dy), lib.ptr(dx), n, h, w, c, lib.stream_ptr())
        return dx


def maxpool2(x):
    return _MaxPool2Fn.apply(x)


lib.register_sigs({
    "fsv_avgpool3s2_fwd": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "fsv_avgpool3s2_bwd": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
})


class _AvgPool3s2Fn(torch.autograd.Function):
    """nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False): the pyramid step between the discriminators of
    the reference's MultiscaleDiscriminator (discriminator.py:28,56)."""

    @staticmethod
    def forward(ctx, x):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1, x)
        lib.check_device(x)
        lib.call("fsv_avgpool3s2_fwd", lib.ptr(x), lib.ptr(y), n, h, w, c, lib.stream_ptr())
        ctx.dims = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w = ctx.dims
        dy = to_nhwc(dy)
        dx = empty_nhwc(n, c, h, w, dy)
        lib.call("fsv_avgpool3s2_bwd", lib.ptr(dy), lib.ptr(dx), n, h, w, c, lib.stream_ptr())
        return dx


def avgpool3s2(x):
    return _AvgPool3s2Fn.apply(x)


class _AdaptiveAvgPoolFn(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d((oh, ow)) (discriminator.py:146,153)"""

    @staticmethod
    def forward(ctx, x, oh, ow):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, oh, ow, x)
        lib.check_device(x)
        lib.call("fsv_adaptive_avgpool_fwd", lib.ptr(x), lib.ptr(y), n, h, w, c, oh, ow, lib.stream_ptr())
        ctx.dims = (n, c, h, w, oh, ow)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w, oh, ow = ctx.dims
        dy = to_nhwc(dy)
        dx = empty_nhwc(n, c, h, w, dy)
        lib.call("fsv_adaptive_avgpool_bwd", lib.ptr(dy), lib.ptr(dx), n, h, w, c, oh, ow, lib.stream_ptr())
        return dx, None, None


def adaptive_avgpool(x, oh, ow):
    return _AdaptiveAvgPoolFn.apply(x, int(oh), int(ow))
