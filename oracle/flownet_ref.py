"""ctypes face of oracle/_ref/libflownet2_ref.so - the reference's OWN correlation / resample2d / channelnorm CUDA kernels,
compiled for the host from the reference tree by oracle/build_ref.py.  TEST INFRASTRUCTURE ONLY (checker for
oracle/flownet_oracle.py and for the HIP kernels of csrc/flownet_ops.hip; never imported by the product path).

`load()` returns None when neither the reference tree (to build from) nor a prebuilt library is present."""
import ctypes
import os

import torch

from . import build_ref

_lib = None


def load():
    global _lib
    if _lib is None:
        path = build_ref.build()
        if path is None or not os.path.exists(path):
            return None
        _lib = ctypes.CDLL(path)
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def correlation(f1, f2, pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2):
    """correlation_cuda.cc:8-83 forward + the two kernels it launches, on CPU tensors (NCHW fp32)"""
    lib = load()
    f1, f2 = f1.contiguous().float(), f2.contiguous().float()
    n, c, h, w = f1.shape
    oh, ow = ctypes.c_int(0), ctypes.c_int(0)
    lib.fsv_ref_correlation_out_hw(h, w, pad_size, kernel_size, max_displacement, stride1, ctypes.byref(oh), ctypes.byref(ow))
    d = 2 * (max_displacement // stride2) + 1
    out = torch.zeros(n, d * d, oh.value, ow.value)
    ph, pw = h + 2 * pad_size, w + 2 * pad_size
    r1, r2 = torch.empty(n * ph * pw * c), torch.empty(n * ph * pw * c)
    lib.fsv_ref_correlation_forward(_p(f1), _p(f2), _p(r1), _p(r2), _p(out), n, c, h, w, pad_size, kernel_size, max_displacement,
                                    stride1, stride2)
    return out


def resample2d(img, flow, kernel_size=1):
    lib = load()
    img, flow = img.contiguous().float(), flow.contiguous().float()
    n, c, h, w = img.shape
    out = torch.empty_like(img)
    lib.fsv_ref_resample2d_forward(_p(img), _p(flow), _p(out), n, c, h, w, kernel_size)
    return out


def channelnorm(x, norm_deg=2):
    lib = load()
    x = x.contiguous().float()
    n, c, h, w = x.shape
    out = torch.empty(n, 1, h, w)
    lib.fsv_ref_channelnorm_forward(_p(x), _p(out), n, c, h, w, norm_deg)
    return out
