"""CPU restatement of the FlowNet2 teacher's three native operators (forward).  TEST INFRASTRUCTURE ONLY.

The reference implements them as CUDA kernels (models/networks/flownet2_pytorch/networks/*_package/*.cu) that cannot be
built or run here (no nvcc, no CUDA device), and its repository holds no test vectors for them: **parity unpinned**
for these three operators - the restatements below follow the kernel sources line by line instead.
"""
import numpy as np
import torch
import torch.nn.functional as F


def correlation(f1, f2, pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2):
    """correlation_cuda.cc:28-42 (shapes) + correlation_cuda_kernel.cu:74-147: zero-padded inputs; output channel
    tc = (tj + r) * D + (ti + r) holds mean over the k*k*C products of the patches at (y1, x1) and (y1 + tj*s2, x1 + ti*s2)."""
    n, c, h, w = f1.shape
    p1 = F.pad(f1, (pad_size,) * 4)
    p2 = F.pad(f2, (pad_size,) * 4)
    krad = (kernel_size - 1) // 2
    border = max_displacement + krad
    ph, pw = h + 2 * pad_size, w + 2 * pad_size
    oh = int(np.ceil((ph - 2 * border) / stride1))
    ow = int(np.ceil((pw - 2 * border) / stride1))
    drad = max_displacement // stride2
    d = 2 * drad + 1
    out = torch.zeros(n, d * d, oh, ow, dtype=f1.dtype)
    ys = torch.arange(oh) * stride1 + max_displacement
    xs = torch.arange(ow) * stride1 + max_displacement
    nelems = kernel_size * kernel_size * c
    for tj in range(-drad, drad + 1):
        for ti in range(-drad, drad + 1):
            acc = torch.zeros(n, oh, ow, dtype=f1.dtype)
            for j in range(-krad, krad + 1):
                for i in range(-krad, krad + 1):
                    a = p1[:, :, (ys + j)][:, :, :, (xs + i)]
                    b = p2[:, :, (ys + tj * stride2 + j)][:, :, :, (xs + ti * stride2 + i)]
                    acc = acc + (a * b).sum(1)
            out[:, (tj + drad) * d + (ti + drad)] = acc / nelems
    return out


def resample2d(img, flow):
    """resample2d_kernel.cu:16-64, kernel_size 1, bit for bit: alpha / beta in float, each weighted tap evaluated in
    double (the `1. - alpha` literals) and rounded to float, accumulated in float in the order LT, RT, LB, RB."""
    n, c, h, w = img.shape
    im = img.numpy().astype(np.float32)
    fl = flow.numpy().astype(np.float32)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing='ij')
    xf = (xs[None] + fl[:, 0]).astype(np.float32)
    yf = (ys[None] + fl[:, 1]).astype(np.float32)
    fx, fy = np.floor(xf), np.floor(yf)
    alpha = (xf - fx).astype(np.float32)
    beta = (yf - fy).astype(np.float32)
    xL = np.clip(fx.astype(np.int64), 0, w - 1)
    xR = np.clip((fx + np.float32(1)).astype(np.int64), 0, w - 1)
    yT = np.clip(fy.astype(np.int64), 0, h - 1)
    yB = np.clip((fy + np.float32(1)).astype(np.int64), 0, h - 1)
    a, b = alpha.astype(np.float64), beta.astype(np.float64)
    out = np.zeros((n, c, h, w), dtype=np.float32)
    bi = np.arange(n)[:, None, None]
    for ch in range(c):
        v = im[:, ch]
        val = np.zeros((n, h, w), dtype=np.float32)
        val = (val + ((1.0 - a) * (1.0 - b) * v[bi, yT, xL].astype(np.float64)).astype(np.float32)).astype(np.float32)
        val = (val + (a * (1.0 - b) * v[bi, yT, xR].astype(np.float64)).astype(np.float32)).astype(np.float32)
        val = (val + ((1.0 - a) * b * v[bi, yB, xL].astype(np.float64)).astype(np.float32)).astype(np.float32)
        val = (val + (a * b * v[bi, yB, xR].astype(np.float64)).astype(np.float32)).astype(np.float32)
        out[:, ch] = val
    return torch.from_numpy(out)


def channelnorm(x):
    """channelnorm_kernel.cu:18-60: float accumulation of x*x in channel order, then sqrt"""
    r = torch.zeros_like(x[:, 0])
    for ch in range(x.shape[1]):
        r = r + x[:, ch] * x[:, ch]
    return torch.sqrt(r).unsqueeze(1)
