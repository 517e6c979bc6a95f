"""CPU restatement of the FlowNet2 teacher's three native operators (forward).  TEST INFRASTRUCTURE ONLY.

The reference implements them as CUDA kernels (models/networks/flownet2_pytorch/networks/*_package/*.cu) that cannot be
built with nvcc or run on a CUDA device here, and its repository holds no test vectors for them.  They are pinned instead to
the reference's own kernel templates compiled FOR THE HOST from the reference tree (oracle/build_ref.py ->
oracle/_ref/libflownet2_ref.so, tests/test_flownet_ref.py): the restatements below follow the kernel sources line by line and
agree with that build bit for bit (resample2d, channelnorm) resp. to summation order (correlation).
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
    """resample2d_kernel.cu:16-64, kernel_size 1, bit for bit: alpha / beta in float; the three weighted taps that carry a
    `1. - alpha` / `1. - beta` literal are evaluated in double and rounded to float, the fourth (`alpha * beta * x`) in float;
    accumulated in float in the order LT, RT, LB, RB.  Pinned to the reference's own kernel compiled for the host
    (oracle/build_ref.py, tests/test_flownet_ref.py)."""
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
        # `(alpha)*(beta) * x` carries no double literal: float products (resample2d_kernel.cu:60)
        val = (val + ((alpha * beta).astype(np.float32) * v[bi, yB, xR]).astype(np.float32)).astype(np.float32)
        out[:, ch] = val
    return torch.from_numpy(out)


def channelnorm(x):
    """channelnorm_kernel.cu:18-60: float accumulation of x*x in channel order, then sqrt"""
    r = torch.zeros_like(x[:, 0])
    for ch in range(x.shape[1]):
        r = r + x[:, ch] * x[:, ch]
    # correctly rounded square root, as CUDA's sqrtf: some CPU builds of torch.sqrt(float32) are 1 ulp off (seen on the
    # GPU box's host), the float rounding of the double-precision root is not
    return torch.sqrt(r.double()).float().unsqueeze(1)


# ------------------------------------------------------------------------------------------------ FlowNet2 (forward)
def _conv(sd, p, x, stride=1, act=True):
    """submodules.py conv / i_conv / predict_flow: Conv2d(k, stride, padding=(k-1)//2) [+ LeakyReLU(0.1)]"""
    w = sd[p + 'weight']
    y = F.conv2d(x, w, sd.get(p + 'bias'), stride=stride, padding=(w.shape[-1] - 1) // 2)
    return F.leaky_relu(y, 0.1) if act else y


def _deconv(sd, p, x, act=True):
    """ConvTranspose2d(4, 2, 1) [+ LeakyReLU(0.1)] (submodules.py:36-40; upsampled_flow*: no activation)"""
    y = F.conv_transpose2d(x, sd[p + 'weight'], sd.get(p + 'bias'), stride=2, padding=1)
    return F.leaky_relu(y, 0.1) if act else y


def _refine(sd, p, c6, c5, c4, c3, c2):
    """FlowNetS.py:64-89 / FlowNetC.py:101-122"""
    flow6 = _conv(sd, p + 'predict_flow6.', c6, act=False)
    concat5 = torch.cat((c5, _deconv(sd, p + 'deconv5.0.', c6), _deconv(sd, p + 'upsampled_flow6_to_5.', flow6, False)), 1)
    flow5 = _conv(sd, p + 'predict_flow5.', concat5, act=False)
    concat4 = torch.cat((c4, _deconv(sd, p + 'deconv4.0.', concat5), _deconv(sd, p + 'upsampled_flow5_to_4.', flow5, False)), 1)
    flow4 = _conv(sd, p + 'predict_flow4.', concat4, act=False)
    concat3 = torch.cat((c3, _deconv(sd, p + 'deconv3.0.', concat4), _deconv(sd, p + 'upsampled_flow4_to_3.', flow4, False)), 1)
    flow3 = _conv(sd, p + 'predict_flow3.', concat3, act=False)
    concat2 = torch.cat((c2, _deconv(sd, p + 'deconv2.0.', concat3), _deconv(sd, p + 'upsampled_flow3_to_2.', flow3, False)), 1)
    return _conv(sd, p + 'predict_flow2.', concat2, act=False)


def flownet_c(sd, p, x):
    """networks/FlowNetC.py:71-128 (eval: returns flow2)"""
    def tower(im):
        a1 = _conv(sd, p + 'conv1.0.', im, 2)
        a2 = _conv(sd, p + 'conv2.0.', a1, 2)
        return a2, _conv(sd, p + 'conv3.0.', a2, 2)
    a2, a3 = tower(x[:, 0:3])
    _, b3 = tower(x[:, 3:])
    corr = F.leaky_relu(correlation(a3, b3, 20, 1, 20, 1, 2), 0.1)
    c31 = _conv(sd, p + 'conv3_1.0.', torch.cat((_conv(sd, p + 'conv_redir.0.', a3), corr), 1))
    c4 = _conv(sd, p + 'conv4_1.0.', _conv(sd, p + 'conv4.0.', c31, 2))
    c5 = _conv(sd, p + 'conv5_1.0.', _conv(sd, p + 'conv5.0.', c4, 2))
    c6 = _conv(sd, p + 'conv6_1.0.', _conv(sd, p + 'conv6.0.', c5, 2))
    return _refine(sd, p, c6, c5, c4, c31, a2)


def flownet_s(sd, p, x):
    """networks/FlowNetS.py:58-95"""
    c2 = _conv(sd, p + 'conv2.0.', _conv(sd, p + 'conv1.0.', x, 2), 2)
    c3 = _conv(sd, p + 'conv3_1.0.', _conv(sd, p + 'conv3.0.', c2, 2))
    c4 = _conv(sd, p + 'conv4_1.0.', _conv(sd, p + 'conv4.0.', c3, 2))
    c5 = _conv(sd, p + 'conv5_1.0.', _conv(sd, p + 'conv5.0.', c4, 2))
    c6 = _conv(sd, p + 'conv6_1.0.', _conv(sd, p + 'conv6.0.', c5, 2))
    return _refine(sd, p, c6, c5, c4, c3, c2)


def flownet_sd(sd, p, x):
    """networks/FlowNetSD.py:62-106"""
    c0 = _conv(sd, p + 'conv0.0.', x)
    c1 = _conv(sd, p + 'conv1_1.0.', _conv(sd, p + 'conv1.0.', c0, 2))
    c2 = _conv(sd, p + 'conv2_1.0.', _conv(sd, p + 'conv2.0.', c1, 2))
    c3 = _conv(sd, p + 'conv3_1.0.', _conv(sd, p + 'conv3.0.', c2, 2))
    c4 = _conv(sd, p + 'conv4_1.0.', _conv(sd, p + 'conv4.0.', c3, 2))
    c5 = _conv(sd, p + 'conv5_1.0.', _conv(sd, p + 'conv5.0.', c4, 2))
    c6 = _conv(sd, p + 'conv6_1.0.', _conv(sd, p + 'conv6.0.', c5, 2))
    flow6 = _conv(sd, p + 'predict_flow6.', c6, act=False)
    concat5 = torch.cat((c5, _deconv(sd, p + 'deconv5.0.', c6), _deconv(sd, p + 'upsampled_flow6_to_5.', flow6, False)), 1)
    flow5 = _conv(sd, p + 'predict_flow5.', _conv(sd, p + 'inter_conv5.0.', concat5, act=False), act=False)
    concat4 = torch.cat((c4, _deconv(sd, p + 'deconv4.0.', concat5), _deconv(sd, p + 'upsampled_flow5_to_4.', flow5, False)), 1)
    flow4 = _conv(sd, p + 'predict_flow4.', _conv(sd, p + 'inter_conv4.0.', concat4, act=False), act=False)
    concat3 = torch.cat((c3, _deconv(sd, p + 'deconv3.0.', concat4), _deconv(sd, p + 'upsampled_flow4_to_3.', flow4, False)), 1)
    flow3 = _conv(sd, p + 'predict_flow3.', _conv(sd, p + 'inter_conv3.0.', concat3, act=False), act=False)
    concat2 = torch.cat((c2, _deconv(sd, p + 'deconv2.0.', concat3), _deconv(sd, p + 'upsampled_flow3_to_2.', flow3, False)), 1)
    return _conv(sd, p + 'predict_flow2.', _conv(sd, p + 'inter_conv2.0.', concat2, act=False), act=False)


def flownet_fusion(sd, p, x):
    """networks/FlowNetFusion.py:46-67"""
    c0 = _conv(sd, p + 'conv0.0.', x)
    c1 = _conv(sd, p + 'conv1_1.0.', _conv(sd, p + 'conv1.0.', c0, 2))
    c2 = _conv(sd, p + 'conv2_1.0.', _conv(sd, p + 'conv2.0.', c1, 2))
    flow2 = _conv(sd, p + 'predict_flow2.', c2, act=False)
    concat1 = torch.cat((c1, _deconv(sd, p + 'deconv1.0.', c2), _deconv(sd, p + 'upsampled_flow2_to_1.', flow2, False)), 1)
    flow1 = _conv(sd, p + 'predict_flow1.', _conv(sd, p + 'inter_conv1.0.', concat1, act=False), act=False)
    concat0 = torch.cat((c0, _deconv(sd, p + 'deconv0.0.', concat1), _deconv(sd, p + 'upsampled_flow1_to_0.', flow1, False)), 1)
    return _conv(sd, p + 'predict_flow0.', _conv(sd, p + 'inter_conv0.0.', concat0, act=False), act=False)


def flownet2(sd, inputs, div_flow=20.0, rgb_max=1.0):
    """FlowNet2.forward (flownet2_pytorch/models.py:117-180); inputs [B, 3, 2, H, W]"""
    b = inputs.shape[0]
    rgb_mean = inputs.contiguous().view(b, 3, -1).mean(dim=-1).view(b, 3, 1, 1, 1)
    x = (inputs - rgb_mean) / rgb_max
    x = torch.cat((x[:, :, 0], x[:, :, 1]), dim=1)
    img0, img1 = x[:, :3], x[:, 3:]
    up = lambda t, mode: F.interpolate(t, scale_factor=4, mode=mode)

    def stage(flow):
        warped = resample2d(img1, flow)
        return torch.cat((x, warped, flow / div_flow, channelnorm(img0 - warped)), dim=1)
    flow_c = up(flownet_c(sd, 'flownetc.', x) * div_flow, 'bilinear')
    flow_s1 = up(flownet_s(sd, 'flownets_1.', stage(flow_c)) * div_flow, 'bilinear')
    flow_s2 = up(flownet_s(sd, 'flownets_2.', stage(flow_s1)) * div_flow, 'nearest')
    norm_s2 = channelnorm(flow_s2)
    diff_s2 = channelnorm(img0 - resample2d(img1, flow_s2))
    flow_sd = up(flownet_sd(sd, 'flownets_d.', x) / div_flow, 'nearest')
    norm_sd = channelnorm(flow_sd)
    diff_sd = channelnorm(img0 - resample2d(img1, flow_sd))
    concat3 = torch.cat((img0, flow_sd, flow_s2, norm_sd, norm_s2, diff_sd, diff_s2), dim=1)
    return flownet_fusion(sd, 'flownetfusion.', concat3)
