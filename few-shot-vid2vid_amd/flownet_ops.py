"""Host wrappers of the FlowNet2 teacher's native operators (csrc/flownet_ops.hip), forward only.

Mirrors the module surface of the reference's extension packages
(models/networks/flownet2_pytorch/networks/{correlation,resample2d,channelnorm}_package): `Correlation(pad_size,
kernel_size, max_displacement, stride1, stride2, corr_multiply)(f1, f2)`, `Resample2d()(img, flow)`, `ChannelNorm()(x)`.
The reference only ever runs them under torch.no_grad() (models/flownet.py:41); calling them on tensors that require
gradients is refused instead of silently returning a constant.
"""
import ctypes

import torch

from . import lib
from .conv import empty_nhwc, to_nhwc

c_p, c_i, c_ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
c_llp = ctypes.POINTER(ctypes.c_longlong)
lib.register_sigs({
    "fsv_correlation_fwd": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "fsv_resample2d_fwd": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_llp, c_llp, c_llp, c_p],
    "fsv_channelnorm_fwd": [c_p, c_p, c_i, c_i, c_ll, c_ll, c_ll, c_ll, c_p],
    "fsv_bilinear_resize_fwd": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_llp, c_llp, c_p],
})


def _no_grad(*ts):
    if torch.is_grad_enabled() and any(t.requires_grad for t in ts):
        raise lib.FsvError("the FlowNet2 teacher operators are forward-only (the reference runs them under no_grad)")


def _ll4(t):
    return (ctypes.c_longlong * 4)(*[int(s) for s in t.stride()])


def correlation(f1, f2, pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2, corr_multiply=1):
    """correlation_cuda.cc:10-87 + kernel :74-147.  f1, f2 [N, C, H, W] -> [N, D*D, OH, OW] (channels-last memory)."""
    _no_grad(f1, f2)
    if corr_multiply != 1:
        raise NotImplementedError("corr_multiply != 1 (the reference kernel ignores it as well)")
    f1, f2 = to_nhwc(f1), to_nhwc(f2)
    n, c, h, w = f1.shape
    drad = max_displacement // stride2
    d = 2 * drad + 1
    oh = -((h + 2 * pad_size - 2 * max_displacement) // -stride1)
    ow = -((w + 2 * pad_size - 2 * max_displacement) // -stride1)
    out = empty_nhwc(n, d * d, oh, ow, f1)
    lib.check_device(f1, f2)
    lib.call("fsv_correlation_fwd", lib.ptr(f1), lib.ptr(f2), lib.ptr(out), n, h, w, c, pad_size, kernel_size,
             max_displacement, stride1, stride2, lib.stream_ptr())
    return out


def resample2d(img, flow):
    """resample2d_kernel.cu:16-64 (kernel_size 1): out[b, c, y, x] = bilinear(img[b, c], x + flow_x, y + flow_y) with the
    tap indices clamped to the image."""
    _no_grad(img, flow)
    n, c, h, w = img.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=img.device)
    lib.check_device(img, flow)
    lib.call("fsv_resample2d_fwd", lib.ptr(img), lib.ptr(flow), lib.ptr(out), n, c, h, w, _ll4(img), _ll4(flow), _ll4(out),
             lib.stream_ptr())
    return out


def bilinear_resize(x, size=None, scale_factor=None):
    """F.interpolate(x, size / scale_factor, mode='bilinear') with align_corners=False (the default the reference relies on:
    flownet2_pytorch/models.py:119 `nn.Upsample(scale_factor=4, mode='bilinear')`, models/flownet.py:66-77)"""
    _no_grad(x)
    n, c, h, w = x.shape
    oh, ow = (int(h * scale_factor), int(w * scale_factor)) if size is None else (int(size[0]), int(size[1]))
    out = torch.empty((n, c, oh, ow), dtype=torch.float32, device=x.device)
    lib.check_device(x)
    lib.call("fsv_bilinear_resize_fwd", lib.ptr(x), lib.ptr(out), n, c, h, w, oh, ow, _ll4(x), _ll4(out), lib.stream_ptr())
    return out


def channelnorm(x):
    """channelnorm_kernel.cu:18-60 (norm_deg 2): [N, C, H, W] -> [N, 1, H, W]"""
    _no_grad(x)
    n, c, h, w = x.shape
    if x.stride(3) * w != x.stride(2):
        x = x.contiguous()
    out = torch.empty((n, 1, h, w), dtype=torch.float32, device=x.device)
    lib.check_device(x)
    lib.call("fsv_channelnorm_fwd", lib.ptr(x), lib.ptr(out), n, c, h * w, x.stride(0), x.stride(1), x.stride(3),
             lib.stream_ptr())
    return out


class Correlation(torch.nn.Module):
    def __init__(self, pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2, corr_multiply=1):
        super().__init__()
        self.args = (pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)

    def forward(self, f1, f2):
        return correlation(f1, f2, *self.args)


class Resample2d(torch.nn.Module):
    def forward(self, img, flow):
        return resample2d(img, flow)


class ChannelNorm(torch.nn.Module):
    def forward(self, x):
        return channelnorm(x)
