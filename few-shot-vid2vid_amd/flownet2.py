"""FlowNet2 teacher (forward / inference only) on the HIP kernels.

Reference: models/networks/flownet2_pytorch/models.py:22-180 (FlowNet2), networks/{FlowNetC,FlowNetS,FlowNetSD,
FlowNetFusion}.py, networks/submodules.py, and its caller models/flownet.py:15-83 (ground-truth flow + confidence for the
flow losses when training without --no_flow_gt).  Module / parameter names are the reference's, so its checkpoint
(`FlowNet2_checkpoint.pth.tar`, 162.5 M parameters; not available in this environment) loads with `load_state_dict`.

Everything runs under torch.no_grad() exactly as in the reference (models/flownet.py:41); the weights are frozen, so the
K-major operand layouts are prepared once and kept.  Operators:
  * Conv2d(+LeakyReLU 0.1) k = 1 / 3 (one gather-GEMM launch, fused bias + activation), k = 5 / 7 (25 / 49 taps: groups of
    <= 16 taps chained through the kernel's residual input, then one bias + activation pass);
  * ConvTranspose2d(4, 2, 1): the stride-2 data-gradient path of the convolution kernels (4 parity-class launches) + bias /
    activation pass;
  * correlation, resample2d, channelnorm: csrc/flownet_ops.hip;
  * bilinear / nearest x4 up-sampling of the 2-channel flows, channel concatenation, mean subtraction: torch glue.
`width_div` shrinks every hidden width (tests only; 1 = the reference network).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import lib, ops
from .conv import (ACT_LRELU01, ACT_NONE, Geom, conv_dgrad, gather_gemm, prep_weight, to_nhwc)
from .flownet_ops import bilinear_resize, channelnorm, correlation, resample2d


def _pad4(t, dim):
    pad = (-t.shape[dim]) % 4
    if pad == 0:
        return t
    spec = [0, 0] * (t.dim() - 1 - dim) + [0, pad]
    return F.pad(t, spec)


def bias_act_(x, bias, act):
    """in place: x = act(x + bias[c]) on an NHWC tensor"""
    lib.check_device(x, bias)
    lib.call("fsv_bias_act", lib.ptr(x), lib.ptr(bias), x.numel(), x.shape[1], act, lib.stream_ptr())
    return x


class FConv(nn.Module):
    """nn.Conv2d(cin, cout, k, stride, padding=(k-1)//2) with an optional fused LeakyReLU(0.1)"""

    def __init__(self, cin, cout, k=3, stride=1, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        nn.init.xavier_uniform_(self.weight)
        if bias:
            nn.init.uniform_(self.bias)
        self.k, self.stride = k, stride
        self._layout = None

    def _prepare(self):
        key = (self.weight._version, self.weight.data_ptr())
        if self._layout is not None and self._layout[0] == key:
            return self._layout[1]
        k, pad = self.k, (self.k - 1) // 2
        w = _pad4(self.weight.detach(), 1)
        taps = [(i, j) for i in range(k) for j in range(k)]
        groups = []
        for g0 in range(0, len(taps), 16):
            grp = taps[g0:g0 + 16]
            khs, kws = [t[0] for t in grp], [t[1] for t in grp]
            wt, _, ldw = prep_weight(w, 0, None, khs, kws)
            groups.append((wt, ldw, [i - pad for i in khs], [j - pad for j in kws]))
        self._layout = (key, groups)
        return groups

    def forward(self, x, act=ACT_NONE):
        x = to_nhwc(_pad4(x, 1))
        n, _, h, w = x.shape
        k, s, pad = self.k, self.stride, (self.k - 1) // 2
        oh, ow = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        cout = self.weight.shape[0]
        groups = self._prepare()
        b = self.bias.detach() if self.bias is not None else None
        if len(groups) == 1:
            wt, ldw, ty, tx = groups[0]
            return gather_gemm(x, wt, ldw, cout, oh, ow, ty, tx, s, s, bias=b, act=act)
        out = None
        for wt, ldw, ty, tx in groups:
            out = gather_gemm(x, wt, ldw, cout, oh, ow, ty, tx, s, s, res=out)
        return bias_act_(out, b, act) if (b is not None or act != ACT_NONE) else out


class FDeconv(nn.Module):
    """nn.ConvTranspose2d(cin, cout, 4, 2, 1) with an optional LeakyReLU(0.1): the transposed convolution IS the data
    gradient of a k4 s2 p1 convolution whose OIHW weight is this [cin, cout, 4, 4] tensor."""

    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cin, cout, 4, 4))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        nn.init.xavier_uniform_(self.weight)
        if bias:
            nn.init.uniform_(self.bias)
        self.geom = Geom(4, 4, 2, 1)
        self._layout = None

    def _prepare(self):
        key = (self.weight._version, self.weight.data_ptr())
        if self._layout is not None and self._layout[0] == key:
            return self._layout[1]
        w = _pad4(_pad4(self.weight.detach(), 0), 1)
        cached = []
        for c in self.geom.dgrad_classes:
            wt, _, ldw = prep_weight(w, 1, self.geom, c['khs'], c['kws'])
            cached.append((wt, ldw))
        self._layout = (key, cached)
        return cached

    def forward(self, x, act=ACT_NONE):
        x = to_nhwc(_pad4(x, 1))
        n, _, h, w = x.shape
        cout = self.weight.shape[1]
        coutp = cout + (-cout) % 4
        y = conv_dgrad(x, None, self.geom, (2 * h, 2 * w), cached=self._prepare(), cin=coutp)
        if coutp != cout:
            y = to_nhwc(y[:, :cout])
        if self.bias is not None or act != ACT_NONE:
            bias_act_(y, self.bias.detach() if self.bias is not None else None, act)
        return y


class _Seq(nn.Module):
    """nn.Sequential(layer, LeakyReLU(0.1)) / nn.Sequential(layer): the parameters live under '.0.' as in the reference"""

    def __init__(self, layer, act):
        super().__init__()
        self.add_module('0', layer)
        self.act = act

    def forward(self, x):
        return getattr(self, '0')(x, self.act)


def conv(cin, cout, kernel_size=3, stride=1):            # submodules.py:7-19, batchNorm False
    return _Seq(FConv(cin, cout, kernel_size, stride), ACT_LRELU01)


def i_conv(cin, cout):                                   # submodules.py:21-31
    return _Seq(FConv(cin, cout, 3, 1), ACT_NONE)


def deconv(cin, cout):                                   # submodules.py:36-40
    return _Seq(FDeconv(cin, cout), ACT_LRELU01)


def predict_flow(cin):                                   # submodules.py:33-34
    return FConv(cin, 2, 3, 1)


def _cat(*ts):
    return to_nhwc(torch.cat([to_nhwc(t) for t in ts], dim=1))


class _Decoder:
    """the refinement ladder shared by FlowNetC and FlowNetS (FlowNetS.py:64-89, FlowNetC.py:101-122)"""

    def _refine(self, c6, c5, c4, c3, c2):
        flow6 = self.predict_flow6(c6)
        concat5 = _cat(c5, self.deconv5(c6), self.upsampled_flow6_to_5(flow6))
        flow5 = self.predict_flow5(concat5)
        concat4 = _cat(c4, self.deconv4(concat5), self.upsampled_flow5_to_4(flow5))
        flow4 = self.predict_flow4(concat4)
        concat3 = _cat(c3, self.deconv3(concat4), self.upsampled_flow4_to_3(flow4))
        flow3 = self.predict_flow3(concat3)
        concat2 = _cat(c2, self.deconv2(concat3), self.upsampled_flow3_to_2(flow3))
        return self.predict_flow2(concat2)

    def _make_decoder(self, d, up_bias):
        c = lambda v: max(v // d, 2)
        self.deconv5 = deconv(c(1024), c(512))
        self.deconv4 = deconv(c(512) + c(512) + 2, c(256))
        self.deconv3 = deconv(c(512) + c(256) + 2, c(128))
        self.deconv2 = deconv(c(256) + c(128) + 2, c(64))
        self.predict_flow6 = predict_flow(c(1024))
        self.predict_flow5 = predict_flow(c(512) + c(512) + 2)
        self.predict_flow4 = predict_flow(c(512) + c(256) + 2)
        self.predict_flow3 = predict_flow(c(256) + c(128) + 2)
        self.predict_flow2 = predict_flow(c(128) + c(64) + 2)
        for name in ('6_to_5', '5_to_4', '4_to_3', '3_to_2'):
            setattr(self, 'upsampled_flow' + name, FDeconv(2, 2, bias=up_bias))


class FlowNetC(nn.Module, _Decoder):
    """networks/FlowNetC.py"""

    def __init__(self, d=1):
        super().__init__()
        c = lambda v: max(v // d, 2)
        self.conv1 = conv(3, c(64), 7, 2)
        self.conv2 = conv(c(64), c(128), 5, 2)
        self.conv3 = conv(c(128), c(256), 5, 2)
        self.conv_redir = conv(c(256), c(32), 1, 1)
        self.conv3_1 = conv(c(32) + 441, c(256))
        self.conv4 = conv(c(256), c(512), stride=2)
        self.conv4_1 = conv(c(512), c(512))
        self.conv5 = conv(c(512), c(512), stride=2)
        self.conv5_1 = conv(c(512), c(512))
        self.conv6 = conv(c(512), c(1024), stride=2)
        self.conv6_1 = conv(c(1024), c(1024))
        self._make_decoder(d, up_bias=True)

    def forward(self, x):
        a1 = self.conv1(x[:, 0:3])
        a2 = self.conv2(a1)
        a3 = self.conv3(a2)
        b3 = self.conv3(self.conv2(self.conv1(x[:, 3:])))
        corr = correlation(a3, b3, 20, 1, 20, 1, 2, 1)
        from .ops import activation
        corr = activation(corr, ACT_LRELU01)
        c31 = self.conv3_1(_cat(self.conv_redir(a3), corr))
        c4 = self.conv4_1(self.conv4(c31))
        c5 = self.conv5_1(self.conv5(c4))
        c6 = self.conv6_1(self.conv6(c5))
        return self._refine(c6, c5, c4, c31, a2)


class FlowNetS(nn.Module, _Decoder):
    """networks/FlowNetS.py"""

    def __init__(self, d=1, input_channels=12):
        super().__init__()
        c = lambda v: max(v // d, 2)
        self.conv1 = conv(input_channels, c(64), 7, 2)
        self.conv2 = conv(c(64), c(128), 5, 2)
        self.conv3 = conv(c(128), c(256), 5, 2)
        self.conv3_1 = conv(c(256), c(256))
        self.conv4 = conv(c(256), c(512), stride=2)
        self.conv4_1 = conv(c(512), c(512))
        self.conv5 = conv(c(512), c(512), stride=2)
        self.conv5_1 = conv(c(512), c(512))
        self.conv6 = conv(c(512), c(1024), stride=2)
        self.conv6_1 = conv(c(1024), c(1024))
        self._make_decoder(d, up_bias=False)

    def forward(self, x):
        c2 = self.conv2(self.conv1(x))
        c3 = self.conv3_1(self.conv3(c2))
        c4 = self.conv4_1(self.conv4(c3))
        c5 = self.conv5_1(self.conv5(c4))
        c6 = self.conv6_1(self.conv6(c5))
        return self._refine(c6, c5, c4, c3, c2)


class FlowNetSD(nn.Module):
    """networks/FlowNetSD.py"""

    def __init__(self, d=1):
        super().__init__()
        c = lambda v: max(v // d, 2)
        self.conv0 = conv(6, c(64))
        self.conv1 = conv(c(64), c(64), stride=2)
        self.conv1_1 = conv(c(64), c(128))
        self.conv2 = conv(c(128), c(128), stride=2)
        self.conv2_1 = conv(c(128), c(128))
        self.conv3 = conv(c(128), c(256), stride=2)
        self.conv3_1 = conv(c(256), c(256))
        self.conv4 = conv(c(256), c(512), stride=2)
        self.conv4_1 = conv(c(512), c(512))
        self.conv5 = conv(c(512), c(512), stride=2)
        self.conv5_1 = conv(c(512), c(512))
        self.conv6 = conv(c(512), c(1024), stride=2)
        self.conv6_1 = conv(c(1024), c(1024))
        self.deconv5 = deconv(c(1024), c(512))
        self.deconv4 = deconv(c(512) + c(512) + 2, c(256))
        self.deconv3 = deconv(c(512) + c(256) + 2, c(128))
        self.deconv2 = deconv(c(256) + c(128) + 2, c(64))
        self.inter_conv5 = i_conv(c(512) + c(512) + 2, c(512))
        self.inter_conv4 = i_conv(c(512) + c(256) + 2, c(256))
        self.inter_conv3 = i_conv(c(256) + c(128) + 2, c(128))
        self.inter_conv2 = i_conv(c(128) + c(64) + 2, c(64))
        self.predict_flow6 = predict_flow(c(1024))
        self.predict_flow5 = predict_flow(c(512))
        self.predict_flow4 = predict_flow(c(256))
        self.predict_flow3 = predict_flow(c(128))
        self.predict_flow2 = predict_flow(c(64))
        for name in ('6_to_5', '5_to_4', '4_to_3', '3_to_2'):
            setattr(self, 'upsampled_flow' + name, FDeconv(2, 2, bias=True))

    def forward(self, x):
        c0 = self.conv0(x)
        c1 = self.conv1_1(self.conv1(c0))
        c2 = self.conv2_1(self.conv2(c1))
        c3 = self.conv3_1(self.conv3(c2))
        c4 = self.conv4_1(self.conv4(c3))
        c5 = self.conv5_1(self.conv5(c4))
        c6 = self.conv6_1(self.conv6(c5))
        flow6 = self.predict_flow6(c6)
        concat5 = _cat(c5, self.deconv5(c6), self.upsampled_flow6_to_5(flow6))
        flow5 = self.predict_flow5(self.inter_conv5(concat5))
        concat4 = _cat(c4, self.deconv4(concat5), self.upsampled_flow5_to_4(flow5))
        flow4 = self.predict_flow4(self.inter_conv4(concat4))
        concat3 = _cat(c3, self.deconv3(concat4), self.upsampled_flow4_to_3(flow4))
        flow3 = self.predict_flow3(self.inter_conv3(concat3))
        concat2 = _cat(c2, self.deconv2(concat3), self.upsampled_flow3_to_2(flow3))
        return self.predict_flow2(self.inter_conv2(concat2))


class FlowNetFusion(nn.Module):
    """networks/FlowNetFusion.py"""

    def __init__(self, d=1):
        super().__init__()
        c = lambda v: max(v // d, 2)
        self.conv0 = conv(11, c(64))
        self.conv1 = conv(c(64), c(64), stride=2)
        self.conv1_1 = conv(c(64), c(128))
        self.conv2 = conv(c(128), c(128), stride=2)
        self.conv2_1 = conv(c(128), c(128))
        self.deconv1 = deconv(c(128), c(32))
        self.deconv0 = deconv(c(128) + c(32) + 2, c(16))
        self.inter_conv1 = i_conv(c(128) + c(32) + 2, c(32))
        self.inter_conv0 = i_conv(c(64) + c(16) + 2, c(16))
        self.predict_flow2 = predict_flow(c(128))
        self.predict_flow1 = predict_flow(c(32))
        self.predict_flow0 = predict_flow(c(16))
        self.upsampled_flow2_to_1 = FDeconv(2, 2, bias=True)
        self.upsampled_flow1_to_0 = FDeconv(2, 2, bias=True)

    def forward(self, x):
        c0 = self.conv0(x)
        c1 = self.conv1_1(self.conv1(c0))
        c2 = self.conv2_1(self.conv2(c1))
        flow2 = self.predict_flow2(c2)
        concat1 = _cat(c1, self.deconv1(c2), self.upsampled_flow2_to_1(flow2))
        flow1 = self.predict_flow1(self.inter_conv1(concat1))
        concat0 = _cat(c0, self.deconv0(concat1), self.upsampled_flow1_to_0(flow1))
        return self.predict_flow0(self.inter_conv0(concat0))


class FlowNet2(nn.Module):
    """models.py:22-180 (batchNorm False, fp32, rgb_max 1, div_flow 20)"""

    def __init__(self, width_div=1, div_flow=20.0, rgb_max=1.0):
        super().__init__()
        self.div_flow, self.rgb_max = div_flow, rgb_max
        self.flownetc = FlowNetC(width_div)
        self.flownets_1 = FlowNetS(width_div)
        self.flownets_2 = FlowNetS(width_div)
        self.flownets_d = FlowNetSD(width_div)
        self.flownetfusion = FlowNetFusion(width_div)

    @torch.no_grad()
    def forward(self, inputs):
        """inputs [B, 3, 2, H, W] (image pair stacked on dim 2), H and W multiples of 64 -> flow [B, 2, H, W]"""
        b = inputs.shape[0]
        rgb_mean = inputs.contiguous().view(b, 3, -1).mean(dim=-1).view(b, 3, 1, 1, 1)
        x = (inputs - rgb_mean) / self.rgb_max
        x = torch.cat((x[:, :, 0], x[:, :, 1]), dim=1)
        img0, img1 = x[:, :3], x[:, 3:]
        def up(t, mode):            # nn.Upsample(scale_factor=4, mode=...) of models.py:119-120, on the HIP kernels
            if mode == 'bilinear':
                return bilinear_resize(t, scale_factor=4)
            return ops.upsample2x(ops.upsample2x(t))          # nearest x4 = nearest x2 twice

        def stage(flow):                                 # models.py:127-135
            warped = resample2d(img1, flow)
            return _cat(x, warped, flow / self.div_flow, channelnorm(img0 - warped))
        flow_c = up(self.flownetc(x) * self.div_flow, 'bilinear')
        flow_s1 = up(self.flownets_1(stage(flow_c)) * self.div_flow, 'bilinear')
        flow_s2 = up(self.flownets_2(stage(flow_s1)) * self.div_flow, 'nearest')
        norm_s2 = channelnorm(flow_s2)
        diff_s2 = channelnorm(img0 - resample2d(img1, flow_s2))
        flow_sd = up(self.flownets_d(x) / self.div_flow, 'nearest')
        norm_sd = channelnorm(flow_sd)
        diff_sd = channelnorm(img0 - resample2d(img1, flow_sd))
        concat3 = _cat(img0, flow_sd, flow_s2, norm_sd, norm_s2, diff_sd, diff_s2)
        return self.flownetfusion(concat3).contiguous()


class FlowNet(nn.Module):
    """models/flownet.py:15-83: teacher flow + confidence for the flow losses.  forward([image_now, image_ref], epoch) ->
    (flow_gt, conf_gt), each [ref, prev] with [B, T, 2|1, H, W] tensors or None."""

    def __init__(self, opt, width_div=1):
        super().__init__()
        self.opt = opt
        self.flowNet = FlowNet2(width_div)
        self.flowNet.eval()

    @torch.no_grad()
    def forward(self, data_list, epoch=0, dummy_bs=0):
        image_now, image_ref = data_list
        image_now, image_ref = image_now[:, :, :3], image_ref[:, 0:1, :3]
        flow_prev = conf_prev = flow_ref = conf_ref = None
        if not self.opt.isTrain or epoch > self.opt.niter_single:
            image_prev = torch.cat([image_now[:, 0:1], image_now[:, :-1]], dim=1)
            flow_prev, conf_prev = self.flowNet_forward(image_now, image_prev)
        if self.opt.warp_ref:
            flow_ref, conf_ref = self.flowNet_forward(image_now, image_ref.expand_as(image_now))
        return [flow_ref, flow_prev], [conf_ref, conf_prev]

    def flowNet_forward(self, a, b):
        if a.dim() == 5:
            bs, n, c, h, w = a.shape
            flow, conf = self.compute_flow_and_conf(a.contiguous().view(-1, c, h, w), b.contiguous().view(-1, c, h, w))
            return flow.view(bs, n, 2, h, w), conf.view(bs, n, 1, h, w)
        return self.compute_flow_and_conf(a, b)

    def compute_flow_and_conf(self, im1, im2):
        old_h, old_w = im1.shape[2:]
        new_h, new_w = old_h // 64 * 64, old_w // 64 * 64
        if old_h != new_h:                               # (the reference tests the height only, flownet.py:66)
            im1 = bilinear_resize(im1, size=(new_h, new_w))
            im2 = bilinear_resize(im2, size=(new_h, new_w))
        flow1 = self.flowNet(torch.cat([im1.unsqueeze(2), im2.unsqueeze(2)], dim=2))
        diff = im1 - resample2d(im2, flow1)
        conf = (torch.sum(diff * diff, dim=1, keepdim=True) < 0.02).float()
        if old_h != new_h:
            flow1 = bilinear_resize(flow1, size=(old_h, old_w)) * old_h / new_h
            conf = bilinear_resize(conf, size=(old_h, old_w))
        return flow1, conf
