"""Generator / discriminator networks of the few-shot-vid2vid hot path on the HIP operators in ops.py.

Same operator surface and - deliberately - the same ``state_dict`` key layout as the reference's
``models.networks`` (generator.py, architecture.py, normalization.py, discriminator.py), so that reference
checkpoints load unchanged and these classes can be patched into the reference's train.py loop
(see INTEGRATION.md).  The bodies are new: every convolution, normalisation, SPADE modulation, up-sampling and
warp goes through the gfx950 kernels, activations stay channels-last between layers, and element-wise chains
of the reference are folded into kernel epilogues (bias+LeakyReLU, tanh, sigmoid, flow scale, residual add,
BN+LeakyReLU, SPADE denorm+modulate+LeakyReLU).
"""
import math
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, streams
from .conv import ACT_LRELU, ACT_NONE, ACT_SIGMOID, ACT_TANH


# ------------------------------------------------------------------------------------------------ parameter holders
class _SpectralMixin:
    def _init_spectral(self, weight):
        rows = weight.shape[0]
        cols = weight.numel() // rows
        self.weight_orig = nn.Parameter(weight)
        self.register_buffer('weight_u', F.normalize(torch.randn(rows), dim=0, eps=1e-12))
        self.register_buffer('weight_v', F.normalize(torch.randn(cols), dim=0, eps=1e-12))

    _sig_cached = None

    def _sn(self):
        cached = self._sig_cached        # (sigma pair, u snapshot, v snapshot) from the network-level batched pass
        if cached is not None:
            self._sig_cached = None
            return (cached[0], cached[1], cached[2], True)
        sig = ops.SpectralState.update(self.weight_orig, self.weight_u, self.weight_v, self.training)
        return (sig, self.weight_u, self.weight_v, False)


class Conv2d(nn.Module, _SpectralMixin):
    """nn.Conv2d (+ optional torch.nn.utils.spectral_norm) with a fused epilogue."""

    def __init__(self, cin, cout, k, stride=1, padding=0, bias=True, spectral=False):
        super().__init__()
        self.stride, self.padding, self.spectral = stride, padding, spectral
        w = torch.empty(cout, cin, k, k)
        if spectral:
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))       # what survives the reference's init
            self._init_spectral(w)
        else:
            nn.init.xavier_normal_(w, gain=0.02)              # init_type 'xavier', init_variance 0.02
            self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None

    def forward(self, x, act=ACT_NONE, res=None, scale=1.0, stats=0, up=False):
        """stats: 1 / -1 when a BatchNorm / InstanceNorm consumes the output next (ops.conv2d stats_groups: the statistics then
        come out of this launch's epilogue instead of a read pass over the output).  up: the convolution of the nearest x2
        up-sampling of x (nn.Upsample in front of this layer in the reference: ops.conv2d folds it into the gather)"""
        if self.spectral:
            return ops.conv2d(x, self.weight_orig, self.bias, self.stride, self.padding, act, scale, res, self._sn(), stats, up)
        return ops.conv2d(x, self.weight, self.bias, self.stride, self.padding, act, scale, res, None, stats, up)


class Linear(nn.Module, _SpectralMixin):
    def __init__(self, cin, cout, spectral=True):
        super().__init__()
        w = torch.empty(cout, cin)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.spectral = spectral
        if spectral:
            self._init_spectral(w)
        else:
            self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(cout))

    def forward(self, x, act=ACT_NONE):
        if self.spectral:
            return ops.linear(x, self.weight_orig, self.bias, act, self._sn())
        return ops.linear(x, self.weight, self.bias, act, None)


class BatchNorm(nn.Module):
    """Train-mode BatchNorm2d statistics holder (apex SyncBatchNorm in one process)."""

    def __init__(self, c, affine=True):
        super().__init__()
        self.affine = affine
        if affine:
            self.weight = nn.Parameter(torch.ones(c))
            self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.ones(c))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        # the counter does not enter the maths (momentum is fixed); it is kept on the host and folded into the buffer
        # when a checkpoint is taken, so that no extra kernel is launched per normalisation call
        self._pending = 0
        self.register_state_dict_pre_hook(BatchNorm._flush)

    @staticmethod
    def _flush(module, prefix, keep_vars):
        if module._pending:
            module.num_batches_tracked += module._pending
            module._pending = 0

    def note_forward(self):
        if self.training:
            self._pending += 1

    # Two forward passes of ONE network issued next to each other on different streams (model.Vid2VidModel's twin generator passes)
    # must not both read-modify-write the running statistics.  While `_redirect` is set, a module found in it hands its kernels a
    # ZEROED stand-in pair instead: the kernel's update `r = (1 - m) r + m s` leaves exactly `m s` there, and the caller folds it
    # into the real buffers in pass order afterwards (`running = (1 - m) running + stand-in`: the same two products and one sum as
    # the in-place update, bit for bit).  A module may be met once per redirected pass (its second update would need (1 - m)^2).
    _redirect = None
    MOMENTUM = 0.1

    def buffers(self):
        red = BatchNorm._redirect
        if red is not None and self.training:
            hit = red.get(id(self))
            if hit is not None:
                if hit[2]:
                    raise RuntimeError("a BatchNorm site was met twice in a redirected pass")
                hit[2] = True
                return hit[0], hit[1]
        return self.running_mean, self.running_var

    def forward(self, x, act=ACT_NONE):
        self.note_forward()
        rm, rv = self.buffers()
        return ops.norm_act(x, self.weight if self.affine else None, self.bias if self.affine else None,
                            rm, rv, instance=False, eps=1e-5, momentum=BatchNorm.MOMENTUM, act=act,
                            training=self.training)


class InstanceNorm(nn.Module):
    """nn.InstanceNorm2d(affine=True, eps=0.1) of the discriminator."""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))

    def forward(self, x, act=ACT_NONE):
        return ops.norm_act(x, self.weight, self.bias, None, None, instance=True, eps=0.1, act=act, training=True)


class _Slot(nn.Module):
    """Parameter-free placeholder that keeps nn.Sequential-style numeric keys aligned with the reference."""

    def forward(self, x):
        return x


def _seq(*mods):
    return nn.ModuleList(list(mods))


# ------------------------------------------------------------------------------------------------ building blocks
class SPADEConv2d(nn.Module):
    """conv3x3(SN) -> BatchNorm(affine) -> LeakyReLU  (reference architecture.py:57-69 with norm='spectralsyncbatch')."""

    def __init__(self, fin, fout, stride=1):
        super().__init__()
        self.conv = Conv2d(fin, fout, 3, stride=stride, padding=1, spectral=True)
        self.bn = BatchNorm(fout, affine=True)

    def forward(self, x):
        # (the hint only pays when the consumer reduces batch statistics: eval-mode BatchNorm takes its running buffers)
        return self.bn(self.conv(x, stats=1 if self.training else 0), act=ACT_LRELU)


class SPADE(nn.Module):
    """Reference normalization.py:18-52 with ks = 1: param-free BatchNorm + sequential (1+gamma)*x+beta per map."""

    def __init__(self, norm_nc, hidden_nc, params_free=False):
        super().__init__()
        if not isinstance(hidden_nc, list):
            hidden_nc = [hidden_nc]
        self.n_hidden = len(hidden_nc)
        self.params_free = params_free
        for i, nh in enumerate(hidden_nc):
            if not params_free or i != 0:
                s = str(i + 1) if i > 0 else ''
                setattr(self, 'mlp_gamma%s' % s, Conv2d(nh, norm_nc, 1))
                setattr(self, 'mlp_beta%s' % s, Conv2d(nh, norm_nc, 1))
        self.norm = BatchNorm(norm_nc, affine=False)
        self.norm_nc = norm_nc

    def forward(self, x, maps, weights=None, act=ACT_NONE, up=False):
        if not isinstance(maps, list):
            maps = [maps]
        use_maps, use_w = [], []
        for i, m in enumerate(maps):
            if m is None:
                continue
            if weights is None or i != 0:
                s = str(i + 1) if i > 0 else ''
                g, b = getattr(self, 'mlp_gamma%s' % s), getattr(self, 'mlp_beta%s' % s)
                use_w.append((g.weight, b.weight, g.bias, b.bias))
            else:
                # generated weights of map 0: the reference indexes weights[0][j] / weights[1][j], i.e. the weight
                # tensors only - the generated biases never reach batch_conv (normalization.py:48-50)
                wg, wb = weights[0][0], weights[1][0]
                zb = getattr(self, '_zero_bias', None)
                if zb is None or zb.shape[0] != wg.shape[0] or zb.device != wg.device:
                    zb = self._zero_bias = streams.shared(
                        lambda: torch.zeros(wg.shape[0], self.norm_nc, dtype=wg.dtype, device=wg.device))
                use_w.append((wg, wb, zb, zb))
            use_maps.append(m)
        self.norm.note_forward()
        rm, rv = self.norm.buffers()
        return ops.spade_mod(x, use_maps, use_w, rm, rv, act=act, training=self.training, up=up)


class SPADEResnetBlock(nn.Module):
    """Reference architecture.py:71-108 (conv_params_free=False; SPADE or plain-BatchNorm flavour)."""

    def __init__(self, fin, fout, hidden_nc=0, spade=True, norm_params_free=False):
        super().__init__()
        fhidden = min(fin, fout)
        self.learned_shortcut = fin != fout
        self.spade = spade
        self.conv_0 = Conv2d(fin, fhidden, 3, padding=1, spectral=True)
        self.conv_1 = Conv2d(fhidden, fout, 3, padding=1, spectral=True)
        if self.learned_shortcut:
            self.conv_s = Conv2d(fin, fout, 1, bias=False, spectral=True)
        if spade:
            self.bn_0 = SPADE(fin, hidden_nc, norm_params_free)
            self.bn_1 = SPADE(fhidden, hidden_nc, norm_params_free)
            if self.learned_shortcut:
                self.bn_s = SPADE(fin, hidden_nc, norm_params_free)
        else:
            self.bn_0 = BatchNorm(fin)
            self.bn_1 = BatchNorm(fhidden)
            if self.learned_shortcut:
                self.bn_s = BatchNorm(fin)

    def forward(self, x, label=None, norm_weights=None, up=False, feeds_norm=True):
        """up=True: x is the block input BEFORE the nearest x2 up-sampling of generator.py:124.  With a learned shortcut
        the up-sampled tensor is consumed only by bn_0 and bn_s, which read x through the up-sampling index (ops.spade_mod
        up=True) - it is never written; otherwise it is materialised here."""
        nw = norm_weights if norm_weights else [None] * 3
        fold = up and self.spade and self.learned_shortcut and x.shape[1] % 16 == 0 and ops.spade_can_fold_upsample()
        if up and not fold:
            x = ops.upsample2x(x)
        if self.spade:
            conv3 = ops.spade_conv3_enabled() and not ops.spade_pair_enabled()      # (two opt-ins that both want bn_0: the pair wins)
            if self.learned_shortcut:
                # bn_s and bn_0 normalise the same x with the same maps: one two-site launch (ops.spade_pair)
                if ops.spade_pair_enabled():
                    with ops.spade_pair():
                        hs = self.bn_s(x, label, nw[2], act=ACT_NONE, up=fold)
                        h0 = self.bn_0(x, label, nw[0], act=ACT_LRELU, up=fold)
                    x_s = self.conv_s(hs)
                else:
                    # bn_s -> conv_s as ONE kernel where csrc/spade_conv.hip covers the widths (ops.spade_into_conv)
                    with ops.spade_into_conv():
                        x_s = self.conv_s(self.bn_s(x, label, nw[2], act=ACT_NONE, up=fold))
                    h0 = None if conv3 else self.bn_0(x, label, nw[0], act=ACT_LRELU, up=fold)
            else:
                x_s = x
                h0 = None if conv3 else self.bn_0(x, label, nw[0], act=ACT_LRELU, up=fold)
            # conv_0 feeds bn_1, conv_1 (+ shortcut) the next block's bn_0 / bn_s: BatchNorm statistics from their epilogues
            # (in training mode only: eval-mode BatchNorm takes its running buffers and would leave the partials unused; a next
            # block that materialises the up-sampling - up and not fold - reduces over the up-sampled tensor itself)
            hint = 1 if self.training else 0
            if conv3:
                # round 6, opt-in (FSV_SPADE_CONV3=1): actvn(bn_*) -> 3x3 convolution as ONE kernel where csrc/spade_conv3.hip covers
                # the widths (the modulated tensor stays in LDS); anything else falls through to the two launches
                with ops.spade_into_conv(conv3=True):
                    dx = self.conv_0(self.bn_0(x, label, nw[0], act=ACT_LRELU, up=fold), stats=hint)
                with ops.spade_into_conv(conv3=True):
                    return self.conv_1(self.bn_1(dx, label, nw[1], act=ACT_LRELU), res=x_s, stats=hint if feeds_norm else 0)
            dx = self.conv_0(h0, stats=hint)
            return self.conv_1(self.bn_1(dx, label, nw[1], act=ACT_LRELU), res=x_s, stats=hint if feeds_norm else 0)
        hint = 1 if self.training else 0
        x_s = self.conv_s(self.bn_s(x)) if self.learned_shortcut else x
        dx = self.conv_0(self.bn_0(x, act=ACT_LRELU), stats=hint)
        return self.conv_1(self.bn_1(dx, act=ACT_LRELU), res=x_s, stats=hint if feeds_norm else 0)


def _decode_early():
    import os
    return streams.ENABLED and os.environ.get('FSV_DECODE_EARLY', '1') == '1'


def spectral_layers(module):
    """every spectral-normalised Conv2d / Linear below `module`, each once, in registration order"""
    seen, out = set(), []
    for m in module.modules():
        if getattr(m, 'spectral', False) and id(m) not in seen:
            seen.add(id(m))
            out.append(m)
    return out


def _channels(nf, n, cap=1024):
    return [min(cap, nf * (2 ** i)) for i in range(n)]


class LabelEmbedder(nn.Module):
    """Reference generator.py:506-572: encoder(-decoder / U-Net) producing one SPADE map per generator level."""

    def __init__(self, opt, input_nc, netS, params_free_layers=0):
        super().__init__()
        nf = opt.ngf
        self.unet = 'unet' in netS
        self.decode = 'decoder' in netS or self.unet
        self.n = n = opt.n_downsample_G
        self.params_free_layers = params_free_layers if params_free_layers != -1 else n
        ch = _channels(nf, n + 1)
        self.conv_first = _seq(Conv2d(input_nc, nf, 3, padding=1), _Slot())
        for i in range(n):
            if i >= params_free_layers or 'decoder' in netS:
                setattr(self, 'down_%d' % i, _seq(Conv2d(ch[i], ch[i + 1], 3, stride=2, padding=1), _Slot()))
        if self.decode:
            for i in reversed(range(n)):
                ch_i = ch[i + 1] * (2 if self.unet and i != n - 1 else 1)
                if i >= params_free_layers:
                    setattr(self, 'up_%d' % i, _seq(_Slot(), Conv2d(ch_i, ch[i], 3, padding=1), _Slot()))

    def forward(self, x, weights=None):
        return self.decode_maps(self.encode_maps(x), weights)

    def encode_maps(self, x):
        """the part that needs no generated weights: the encoder and the decoder levels with parameters of their own (it
        runs next to the reference encoders that produce those weights, see FewShotGenerator.flow_branch)"""
        if x is None:
            return None
        n = self.n
        out = [self.conv_first[0](x, act=ACT_LRELU)]
        for i in range(n):
            if i >= self.params_free_layers or self.decode:
                out.append(getattr(self, 'down_%d' % i)[0](out[-1], act=ACT_LRELU))
            else:
                raise NotImplementedError("adaptive strided embedding convs are not used by any shipped config")
        if not self.decode:
            return out
        skips = out
        if not self.unet:
            out = [out[-1]]
        out = list(out)
        for i in reversed(range(n)):
            if i < self.params_free_layers:
                break
            out.append(self._up(i, out[-1], skips, None))
        return out, skips

    def _up(self, i, cur, skips, weights):
        if self.unet and i != self.n - 1:
            cur = ops.cat_channels([cur, skips[i + 1]])
        if i >= self.params_free_layers:
            return getattr(self, 'up_%d' % i)[1](cur, act=ACT_LRELU, up=True)      # generator.py:559-563: Upsample -> conv3x3
        # a 1x1 convolution commutes with nearest up-sampling: run the generated-weight conv on the quarter-size
        # tensor, then up-sample (bit-identical, 4x fewer MACs and bytes)
        w, b = weights[i]
        return ops.upsample2x(ops.batch_conv(cur, w, b, act=ACT_LRELU))

    def decode_maps(self, state, weights=None):
        if state is None or not self.decode:
            return state
        out, skips = state
        out = list(out)
        for i in reversed(range(min(self.n, self.params_free_layers))):
            out.append(self._up(i, out[-1], skips, weights))
        if self.unet:
            out = out[self.n:]
        return out[::-1]


class FlowGenerator(nn.Module):
    """Reference generator.py:456-504."""

    def __init__(self, opt, n_frames_G):
        super().__init__()
        input_nc = (opt.label_nc if opt.label_nc != 0 else opt.input_nc) * n_frames_G + opt.output_nc * (n_frames_G - 1)
        nf, nd = opt.nff, opt.n_downsample_F
        self.nd = nd
        self.flow_multiplier = opt.flow_multiplier
        ch = _channels(nf, nd + 1)

        def normed(cin, cout, stride=1):
            return _seq(Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False, spectral=True), BatchNorm(cout))
        down = [normed(input_nc, nf), _Slot()]
        for i in range(nd):
            down += [normed(ch[i], ch[i + 1], 2), _Slot()]
        self.down_flow = _seq(*down)
        self.res_flow = _seq(*[SPADEResnetBlock(ch[nd], ch[nd], spade=False) for _ in range(opt.n_blocks_F)])
        up = []
        for i in reversed(range(nd)):
            up += [_Slot(), normed(ch[i + 1], ch[i]), _Slot()]
        self.up_flow = _seq(*up)
        self.conv_flow = _seq(Conv2d(nf, 2, 3, padding=1))
        self.conv_mask = _seq(Conv2d(nf, 1, 3, padding=1), _Slot())

    def forward(self, label, label_prev, img_prev, for_ref=False):
        x = ops.cat_channels([label, label_prev, img_prev])
        for k in range(0, 2 * (self.nd + 1), 2):
            conv, bn = self.down_flow[k]
            x = bn(conv(x, stats=1 if self.training else 0), act=ACT_LRELU)
        for k, blk in enumerate(self.res_flow):
            x = blk(x, feeds_norm=k + 1 < len(self.res_flow))
        for k in range(1, 3 * self.nd, 3):
            conv, bn = self.up_flow[k]
            x = bn(conv(x, stats=1 if self.training else 0, up=True), act=ACT_LRELU)           # generator.py:489-493: Upsample -> conv
        flow = self.conv_flow[0](x, scale=float(self.flow_multiplier))
        mask = self.conv_mask[0](x, act=ACT_SIGMOID)
        return flow, mask


def pick_ref(refs, ref_idx):
    """base_network.py:40-47: the reference with the largest attention mass per sample (the first one for n_shot == 1)"""
    if ref_idx is None:
        return refs[:, 0]
    return refs[torch.arange(refs.shape[0], device=refs.device), ref_idx.long()]


class FewShotGenerator(nn.Module):
    """Reference generator.py:20-454 for use_label_ref == 'mul', no KLD, no adaptive_conv; n_shot >= 1 (with more than one
    reference image the attention module of generator.py:291-316 merges the reference features)."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        if getattr(opt, 'adaptive_conv', False) or getattr(opt, 'lambda_kld', 0) > 0:
            raise NotImplementedError("adaptive_conv and the KLD branch are outside the hot-path scope")
        self.n_shot = getattr(opt, 'n_shot', 1)
        self.n_downsample_A = getattr(opt, 'n_downsample_A', 2)
        if getattr(opt, 'use_label_ref', 'mul') != 'mul' or getattr(opt, 'res_for_ref', False):
            raise NotImplementedError("only use_label_ref='mul' with SPADEConv2d encoders is on the hot path")
        if opt.spade_ks != 1 or opt.embed_ks != 1 or opt.conv_ks != 3:
            raise NotImplementedError("spade_ks = embed_ks = 1, conv_ks = 3 (the defaults of every shipped script)")
        self.n_downsample_G = n = opt.n_downsample_G
        nf = opt.ngf
        nf_max = min(1024, nf * (2 ** n))
        self.ch = ch = [min(nf_max, nf * (2 ** i)) for i in range(n + 2)]
        self.spade_combine = opt.spade_combine
        self.n_sc_layers = opt.n_sc_layers
        self.add_raw_output_loss = getattr(opt, 'add_raw_output_loss', False) and opt.spade_combine
        ch_hidden = []
        for i in range(n + 1):
            ch_hidden += [[ch[i]]] if not self.spade_combine or i >= self.n_sc_layers else [[ch[i]] * 3]
        self.ch_hidden = ch_hidden
        self.adap_spade = opt.adaptive_spade
        self.adap_embed = opt.adaptive_spade and not getattr(opt, 'no_adaptive_embed', False)
        self.n_adaptive_layers = opt.n_adaptive_layers if opt.n_adaptive_layers != -1 else n
        self.n_fc_layers = opt.n_fc_layers
        input_nc = opt.label_nc if opt.label_nc != 0 else opt.input_nc
        self.ref_img_first = SPADEConv2d(opt.output_nc, nf)
        self.ref_label_first = SPADEConv2d(input_nc, nf)
        for i in range(n):
            setattr(self, 'ref_img_down_%d' % i, SPADEConv2d(ch[i], ch[i + 1], stride=2))
            setattr(self, 'ref_img_up_%d' % i, SPADEConv2d(ch[i + 1], ch[i]))
            setattr(self, 'ref_label_down_%d' % i, SPADEConv2d(ch[i], ch[i + 1], stride=2))
            setattr(self, 'ref_label_up_%d' % i, SPADEConv2d(ch[i + 1], ch[i]))
        if self.adap_spade:
            for i in range(self.n_adaptive_layers):
                ch_in, ch_out = ch[i], ch[i + 1]
                ch_h = ch_hidden[i][0]
                names = ['fc_spade_0', 'fc_spade_1', 'fc_spade_s']
                outs = [(ch_h + 1) * 2, (ch_h + 1) * (1 if ch_in != ch_out else 2), (ch_h + 1) * 2]
                if self.adap_embed:
                    names.append('fc_spade_e')
                    outs.append(ch_in + 1)
                for name, fo in zip(names, outs):
                    layers = [Linear(ch_out, ch_out), _Slot()]
                    for _ in range(1, self.n_fc_layers):
                        layers += [Linear(ch_out, ch_out), _Slot()]
                    layers += [Linear(ch_out, fo)]
                    setattr(self, '%s_%d' % (name, i), _seq(*layers))
        self.label_embedding = LabelEmbedder(opt, input_nc, opt.netS,
                                             params_free_layers=(self.n_adaptive_layers if self.adap_embed else 0))
        for i in reversed(range(n + 1)):
            setattr(self, 'up_%d' % i, SPADEResnetBlock(ch[i + 1], ch[i], hidden_nc=ch_hidden[i], spade=True,
                                                       norm_params_free=(self.adap_spade and i < self.n_adaptive_layers)))
        self.conv_img = Conv2d(nf, 3, 3, padding=1)
        if self.n_shot > 1:                # generator.py:128-134: key / query encoders of the attention module
            self.atn_query_first = SPADEConv2d(input_nc, nf)
            self.atn_key_first = SPADEConv2d(input_nc, nf)
            for i in range(self.n_downsample_A):
                setattr(self, 'atn_key_%d' % i, SPADEConv2d(ch[i], ch[i + 1], stride=2))
                setattr(self, 'atn_query_%d' % i, SPADEConv2d(ch[i], ch[i + 1], stride=2))
        self._sn_group, self._sn_count = None, -1
        self.warp_prev = False
        self.warp_ref = opt.warp_ref and not getattr(opt, 'for_face', False)
        if self.warp_ref:
            self.flow_network_ref = FlowGenerator(opt, 2)
            if self.spade_combine:
                self.img_ref_embedding = LabelEmbedder(opt, opt.output_nc + 1, opt.sc_arch)

    # -- temporal extension (reference generator.py:153-179) -------------------------------------------------------
    def init_temporal_network(self):
        opt = self.opt
        self.warp_prev = True
        self.sep_prev_flownet = opt.sep_flow_prev or (opt.n_frames_G != 2) or not opt.warp_ref
        self.sep_prev_embedding = self.spade_combine and (not opt.no_sep_warp_embed or not opt.warp_ref)
        dev = self.conv_img.weight.device
        if self.sep_prev_flownet:
            self.flow_network_temp = FlowGenerator(opt, opt.n_frames_G).to(dev)
        else:
            self.flow_network_temp = self.flow_network_ref
        if self.spade_combine:
            if self.sep_prev_embedding:
                self.img_prev_embedding = LabelEmbedder(opt, opt.output_nc + 1, opt.sc_arch).to(dev)
            else:
                self.img_prev_embedding = self.img_ref_embedding
        if self.warp_ref:
            if self.sep_prev_flownet:
                self.load_pretrained_net(self.flow_network_ref, self.flow_network_temp)
            if self.sep_prev_embedding:
                self.load_pretrained_net(self.img_ref_embedding, self.img_prev_embedding)
            self.flow_temp_is_initalized = True

    @staticmethod
    def load_pretrained_net(net_src, net_dst):
        src, dst = net_src.state_dict(), net_dst.state_dict()
        for k, v in src.items():
            if k in dst and dst[k].size() == v.size():
                dst[k] = v
        net_dst.load_state_dict(dst)

    # -- weight generation ---------------------------------------------------------------------------------------------
    def _mlp(self, name, i, rows):
        layers = getattr(self, '%s_%d' % (name, i))
        x = rows
        last = len(layers) - 1
        for k in range(0, last, 2):
            x = layers[k](x, act=ACT_LRELU)
        return layers[last](x)

    @staticmethod
    def _pairs(f, npairs, cout, cin):
        """f [b, L] -> npairs x [weight [b, cout, cin, 1, 1], bias [b, cout]] read off the front of each row
        (generator.py reshape_weight slices the flattened FC output the same way).  One split instead of nested slicing:
        its backward is ONE concatenation (ops.split_cols) instead of a zero-fill + copy + add per slice."""
        sizes = [cout * cin, cout] * npairs
        rest = f.shape[1] - sum(sizes)
        parts = ops.split_cols(f, sizes + ([rest] if rest > 0 else []))
        b = f.shape[0]
        return [[parts[2 * k].reshape(b, cout, cin, 1, 1), parts[2 * k + 1]] for k in range(npairs)]

    _MLP_NAMES = ('fc_spade_e', 'fc_spade_0', 'fc_spade_1', 'fc_spade_s')

    def _mlp_names(self):
        return self._MLP_NAMES if self.adap_embed else self._MLP_NAMES[1:]

    def _mlp_bank(self, feats):
        """All weight-generator MLPs of all adaptive levels, advanced layer by layer with one grouped launch per layer
        (ops.mlp_bank) instead of one small launch per Linear; {(name, level): FC output} or None when the grouped path does
        not apply (no optimiser-owned layouts yet, narrow-operand modes, FSV_CONV_GROUPS=0)."""
        rows = [f.reshape(f.shape[0] * f.shape[1], -1) for f in feats]
        chains, keys = [], []
        for i in range(len(feats)):
            for name in self._mlp_names():
                layers = getattr(self, '%s_%d' % (name, i))
                chains.append((i, [layers[k] for k in range(0, len(layers), 2)]))
                keys.append((name, i))
        outs = ops.mlp_bank(rows, chains)
        return None if outs is None else dict(zip(keys, outs))

    def get_SPADE_weights(self, feat, i, fc=None):
        """fc: {(name, level): FC output rows} from _mlp_bank, or None: run this level's MLPs here"""
        ch_in, ch_out = self.ch[i], self.ch[i + 1]
        ch_h = self.ch_hidden[i][0]
        b = feat.shape[0]
        rows = feat.reshape(b * feat.shape[1], -1) if fc is None else None

        def mlp(name):
            return fc[(name, i)] if fc is not None else self._mlp(name, i, rows)
        embedding_weights = None
        if self.adap_embed:
            fe = mlp('fc_spade_e').view(b, -1)
            # the reference drops the trailing ch_in entries, then reads weight | bias off what is left: the same split
            embedding_weights = self._pairs(fe, 1, ch_in, ch_out)[0]

        def two(name, co):
            f = mlp(name).view(b, -1)
            return self._pairs(f, 2, co, ch_h)
        return embedding_weights, [two('fc_spade_0', ch_out), two('fc_spade_1', ch_in), two('fc_spade_s', ch_out)]

    def attention_encode(self, img, name):
        x = getattr(self, name + '_first')(img)
        for i in range(self.n_downsample_A):
            x = getattr(self, '%s_%d' % (name, i))(x)
        return x

    def attention_module(self, x, label, label_ref, attention=None):
        """generator.py:298-316.  energy = key^T query over all N*HW reference positions, softmax over them, then the
        attention-weighted sum of the N reference feature maps.  Both batched matrix products run on the gather-GEMM
        kernel as per-sample 1x1 convolutions over the h x w query positions, kept in the transposed arrangement
        attention_t[b, (n, p_key), y, x] so that the softmax is the channel softmax kernel."""
        bn, c, h, w = x.shape
        n = self.n_shot
        b = bn // n
        hw = h * w
        if attention is None:
            key = self.attention_encode(label_ref, 'atn_key')            # [b*n, c, h, w]
            query = self.attention_encode(label, 'atn_query')            # [b, c, h, w]
            kmat = key.reshape(b, n, c, hw).permute(0, 1, 3, 2).reshape(b, n * hw, c, 1, 1)
            energy_t = ops.batch_conv(query, kmat, allow_half=False)                        # [b, n*hw, h, w]
            attention = ops.softmax_channels(energy_t)
        xmat = x.reshape(b, n, c, hw).permute(0, 2, 1, 3).reshape(b, c, n * hw, 1, 1)
        out = ops.batch_conv(attention, xmat, allow_half=False)                             # [b, c, h, w]
        atn_vis = attention.reshape(b, n, hw, h, w).sum(2)[-1:, 0:1]
        return out, attention, atn_vis

    def reference_encoding(self, img_ref, label_ref, encode=True, label=None):
        n = self.n_downsample_G
        # (the two encoders as parallel branches of the captured graph: +2 ... 3 ms in round 2, re-measured in round 6 on the final
        # kernels: 42.32 / 42.65 -> 43.50 / 43.51 ms per step - profiles/r06_step_ab_schedule.txt; not kept)
        x = self.ref_img_first(img_ref)
        xl = self.ref_label_first(label_ref)
        atn = atn_vis = ref_idx = None
        for i in range(n):
            x = getattr(self, 'ref_img_down_%d' % i)(x)
            xl = getattr(self, 'ref_label_down_%d' % i)(xl)
            if self.n_shot > 1 and i == self.n_downsample_A - 1:          # generator.py:359-366
                x, atn, atn_vis = self.attention_module(x, label, label_ref)
                xl, _, _ = self.attention_module(xl, None, None, atn)
                ref_idx = torch.argmax(atn.reshape(label.shape[0], self.n_shot, -1).sum(2), dim=1)
        self._atn = (atn_vis, ref_idx)
        if not encode:           # generator.py:370: test-time frames after the first re-use the cached weights
            return x, None
        fi, fl = [x], [xl]
        for i in reversed(range(n)):
            fi.append(getattr(self, 'ref_img_up_%d' % i)(fi[-1]))
            fl.append(getattr(self, 'ref_label_up_%d' % i)(fl[-1]))
        return x, self._pooled(fi, fl)

    @staticmethod
    def _pooled(fi, fl):
        enc = []
        for a, l in zip(fi, fl):
            b, c, h, w = a.shape
            sm = ops.softmax_channels(l)
            if ops.pooled_product_ready(a, sm):
                # round 6, opt-in (FSV_POOL_WGRAD=1, measured neutral): the product over positions as a per-sample 1x1 weight-gradient
                # GEMM - both operands read in place
                enc.append(ops.pooled_product(a, sm))                               # [b, c(i), c(j), 1]
                continue
            # prod[b, i, j] = sum_p a[b, i, p] * sm[b, j, p]  as a per-sample 1x1 "convolution" on the gather-GEMM
            # kernel: pixels = image channels i, input channels = positions p, generated weights = softmax rows j
            a_rows = a.reshape(b, c, 1, h * w).permute(0, 3, 1, 2)              # logical [b, hw, c, 1]
            wts = sm.reshape(b, c, h * w, 1, 1)
            prod = ops.batch_conv(a_rows, wts, allow_half=False)                                    # logical [b, c(j), c(i), 1]
            enc.append(prod.permute(0, 2, 1, 3))                                  # [b, c(i), c(j), 1]
        return enc[::-1]

    def weight_generation(self, img_ref, label_ref, label, t=0, label_maps_elsewhere=False):
        """returns (x, label maps, SPADE weights); with `label_maps_elsewhere` the middle entry is the generated embedding
        weights instead (label_embedding.encode_maps runs next to the flow network, forward() finishes with decode_maps)"""
        b, n, c, h, w = img_ref.shape
        img_ref, label_ref = img_ref.reshape(b * n, -1, h, w), label_ref.reshape(b * n, -1, h, w)
        # generator.py:370,403-416: at test time (isTrain False, one reference) the generated weights of frame 0 are kept
        # and every later frame only runs the down path of the reference encoder
        fresh = bool(self.opt.isTrain) or n > 1 or t == 0
        x, enc = self.reference_encoding(img_ref, label_ref, encode=fresh, label=label)
        cut2 = getattr(self, 'bwd_cut2', None)
        if cut2 is not None and fresh and torch.is_grad_enabled():
            # second stage boundary (three-piece backward): what the reference encoders hand on - the deepest feature map and
            # the pooled products the weight generators read - becomes detached leaves; the encoders' backward is the third piece
            cut2.begin_forward()
            x, enc = cut2.split((x, enc))
        if fresh:
            embed_w, norm_w = [], []
            if self.adap_spade:
                feats = [enc[min(len(enc) - 1, i + 1)] for i in range(self.n_adaptive_layers)]
                fc = self._mlp_bank(feats)
                for i in range(self.n_adaptive_layers):
                    e, nw = self.get_SPADE_weights(feats[i], i, fc)
                    embed_w.append(e)
                    norm_w.append(nw)
            if not self.opt.isTrain:
                self._cached_weights = (embed_w, norm_w)
        else:
            embed_w, norm_w = self._cached_weights
        embed_w = embed_w if self.adap_embed else None
        if label_maps_elsewhere:
            return x, embed_w, norm_w
        return x, self.label_embedding(label, weights=embed_w), norm_w

    def forward_face(self, label, label_refs, img_refs, img_coarse):
        """generator.py:232-242 (the --refine_face generator): the decoder starts from the encoding of the COARSE face
        (compute_kld with img_coarse, generator.py:321-325: reference-image encoder applied to it) instead of the
        reference image's; SPADE weights still come from the reference crops."""
        _, enc_label, norm_w = self.weight_generation(img_refs, label_refs, label)
        x = self.ref_img_first(img_coarse)
        for i in range(self.n_downsample_G):
            x = getattr(self, 'ref_img_down_%d' % i)(x)
        for i in range(self.n_downsample_G, -1, -1):
            nw = norm_w[i] if (self.adap_spade and i < self.n_adaptive_layers) else None
            x = getattr(self, 'up_%d' % i)(x, enc_label[i], nw, up=(i != self.n_downsample_G))
        return self.conv_img(ops.activation(x, ACT_LRELU), act=ACT_TANH)

    def flow_generation(self, label, label_ref, img_ref, prev):
        """generator.py:430-449.  The warp is fused with what consumes it (ops.warp_concat / ops.warp_blend, csrc/warp.hip):
        with --spade_combine the warped image and ds = cat([warp, mask]) come out of one launch here; otherwise the warp is
        left to the blend in forward() (`sources` carries the images to warp)."""
        label_prev, img_prev = prev
        flow, mask, warp, ds = [None, None], [None, None], [None, None], [None, None]
        sources = [None, None]
        if self.warp_ref:
            flow[0], mask[0] = self.flow_network_ref(label, label_ref, img_ref, for_ref=True)
            sources[0] = img_ref[:, :3]
        if self.warp_prev and label_prev is not None:
            flow[1], mask[1] = self.flow_network_temp(label, label_prev, img_prev)
            sources[1] = img_prev[:, -3:]
        if self.spade_combine:
            for k in range(2):
                if sources[k] is not None:
                    warp[k], ds[k] = ops.warp_concat(sources[k], flow[k], mask[k])
        self._warp_sources = sources
        return flow, mask, warp, ds

    def combine_embeddings(self, ds):
        """generator.py:218-225 (--spade_combine): SPADE maps from the warped reference / previous image"""
        if not self.spade_combine:
            return None
        return [self.img_ref_embedding(ds[0]), self.img_prev_embedding(ds[1]) if ds[1] is not None else None]

    def flow_branch(self, label, label_ref, img_ref, prev, with_label_maps=False):
        """everything of the forward pass that needs neither the reference encoders nor the generated weights: flow network,
        warp, the SPADE maps of the warped image and (with_label_maps) the weight-free part of the label embedding"""
        maps = None
        early = with_label_maps and _decode_early() and label.is_cuda
        if early:
            # round 6 (FSV_DECODE_EARLY=0: the old order): the label maps FIRST, with an event behind them - the other branch then
            # decodes them right behind its weight generators instead of behind the join of both branches.  In the replayed graph the
            # twelve small launches of decode_maps sat ~90 us apart behind that join in the generator-mode pass (a 0.7 ms hole in
            # profiles/r06_step_sequence.txt; back to back in the no-grad pass and, here, on the branch's own queue); unprofiled the
            # step gains 0.1 ms (profiles/r06_step_ab_schedule.txt) - most of that hole is the profiler's cross-queue cost
            maps = self.label_embedding.encode_maps(label)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(label.device))
            self._maps_ready = (ev, maps)
        flow, mask, warp, ds = self.flow_generation(label, label_ref, img_ref, prev)
        emb = self.combine_embeddings(ds)
        if with_label_maps and not early:
            maps = self.label_embedding.encode_maps(label)
        return flow, mask, warp, emb, maps

    def stage2_parameters(self):
        """parameters below the BackwardCut boundary of forward(): decoder blocks, output conv"""
        names = ('up_', 'conv_img')
        return [p for n, p in self.named_parameters() if n.startswith(names)]

    def stage3_parameters(self):
        """parameters above the SECOND boundary (three-piece backward, `bwd_cut2` on the outputs of reference_encoding): the
        reference encoders and the attention encoders - the last part of the network the backward pass reaches"""
        names = ('ref_img_', 'ref_label_', 'atn_')
        return [p for n, p in self.named_parameters() if n.startswith(names)]

    def _sn_update(self):
        if self._sn_group is None or self._sn_count != sum(1 for _ in self.modules()):
            self._sn_group = ops.SpectralGroup(spectral_layers(self))
            self._sn_count = sum(1 for _ in self.modules())
        self._sn_group.update(self.training)

    def begin_pass(self):
        """One power iteration of every spectral layer NOW, for a forward pass that is issued later (possibly on another stream,
        possibly behind further begin_pass() calls): the (layer, sigma / u / v snapshot) list is queued and the next forward pass
        takes it instead of iterating itself - the iterations keep their order whatever the order the passes run in."""
        self._sn_update()
        snap = [(l, l._sig_cached) for l in self._sn_group.layers]
        for l, _ in snap:
            l._sig_cached = None
        if getattr(self, '_sn_presets', None) is None:
            self._sn_presets = []
        self._sn_presets.append(snap)

    def forward(self, label, label_refs, img_refs, prev=(None, None), t=0, img_coarse=None):
        from .conv import stats_pass
        with stats_pass(label.device):          # a no-op inside Vid2VidModel.forward's pass; opens one for bare generator calls
            return self._forward(label, label_refs, img_refs, prev, t, img_coarse)

    def _forward(self, label, label_refs, img_refs, prev=(None, None), t=0, img_coarse=None):
        presets = getattr(self, '_sn_presets', None)
        if presets:
            for l, c in presets.pop(0):          # this pass's power iteration was issued ahead of it (begin_pass)
                l._sig_cached = c
        else:
            self._sn_update()
        if img_coarse is not None:
            return self.forward_face(label, label_refs, img_refs, img_coarse)
        if self.n_shot == 1 and label_refs.shape[1] == 1:
            # The flow branch (flow network, warp, the SPADE maps of the warped image) depends on nothing the reference
            # encoders / weight generators produce: two parallel branches (streams.fork).  With attention (n_shot > 1) it
            # needs ref_idx first.
            self._maps_ready = None

            def generate_and_decode():
                x, embed_w, norm_w = self.weight_generation(img_refs, label_refs, label, t=t, label_maps_elsewhere=True)
                ready = self._maps_ready            # (set by the flow branch, which streams.fork issues first)
                if ready is None:
                    return x, embed_w, norm_w, None
                ev, maps = ready
                cur = torch.cuda.current_stream(label.device)
                cur.wait_event(ev)
                streams._record(maps, cur)
                return x, embed_w, norm_w, self.label_embedding.decode_maps(maps, embed_w)
            (x, embed_w, norm_w, enc_label), (flow, mask, warp, emb, maps) = streams.fork(label, [
                generate_and_decode,
                lambda: self.flow_branch(label, label_refs[:, 0], img_refs[:, 0], prev, with_label_maps=True)])
            self._maps_ready = None
            if enc_label is None:
                enc_label = self.label_embedding.decode_maps(maps, embed_w)
            atn_vis, ref_idx = self._atn
        else:
            x, enc_label, norm_w = self.weight_generation(img_refs, label_refs, label, t=t)
            atn_vis, ref_idx = self._atn
            label_ref, img_ref = pick_ref(label_refs, ref_idx), pick_ref(img_refs, ref_idx)
            flow, mask, warp, emb, _ = self.flow_branch(label, label_ref, img_ref, prev)
        cut = getattr(self, 'bwd_cut', None)
        if cut is not None and torch.is_grad_enabled():
            cut.begin_forward()
            # stage boundary (see BackwardCut): everything above is "stage 1", the decoder below is "stage 2"
            # (stage2_parameters)
            x, enc_label, norm_w, flow, mask, warp, emb = cut.split((x, enc_label, norm_w, flow, mask, warp, emb))
            enc_label = list(enc_label)
        # --add_raw_output_loss (generator.py:195, 202-205, 227): the last n_sc_layers blocks run a second time on the label
        # embedding alone (no warped-image maps) - same modules, so their spectral norms and BatchNorm running statistics take a
        # second update, as the reference's do
        enc_raw = [enc_label[i] for i in range(self.n_sc_layers)] if self.add_raw_output_loss else None
        x_raw = None
        if self.spade_combine:
            for i in range(self.n_sc_layers):
                enc_label[i] = [enc_label[i]] + [e[i] if e is not None else None for e in emb]
        for i in range(self.n_downsample_G, -1, -1):
            nw = norm_w[i] if (self.adap_spade and i < self.n_adaptive_layers) else None
            # generator.py:121-124: the nearest x2 up-sampling after block i + 1 is handed to block i (up=True), whose SPADE
            # kernels read through the up-sampling index
            if enc_raw is not None and i < self.n_sc_layers:
                if i == self.n_sc_layers - 1:
                    x_raw = x
                x_raw = getattr(self, 'up_%d' % i)(x_raw, enc_raw[i], nw, up=(i != self.n_downsample_G), feeds_norm=i > 0)
            x = getattr(self, 'up_%d' % i)(x, enc_label[i], nw, up=(i != self.n_downsample_G), feeds_norm=i > 0)
        img_raw = self.conv_img(ops.activation(x, ACT_LRELU), act=ACT_TANH)
        if not self.spade_combine:
            img_final = img_raw
            sources = self._warp_sources
            warp = list(warp)
            if self.warp_ref:
                warp[0], img_final = ops.warp_blend(img_raw, sources[0], flow[0], mask[0])
            elif not self.warp_prev:
                img_raw = None
            if sources[1] is not None:
                warp[1], img_final = ops.warp_blend(img_final, sources[1], flow[1], mask[1])
        else:
            img_final = img_raw
            img_raw = self.conv_img(ops.activation(x_raw, ACT_LRELU), act=ACT_TANH) if x_raw is not None else None
        return img_final, flow, mask, img_raw, warp, None, None, atn_vis, ref_idx


class BackwardCut:
    """A stage boundary inside one forward pass, for backward passes that run in two pieces (graph_step / bench.py at
    N > 1: the gradients of the decoder stage are complete after the first piece and are all-reduced on a side stream
    while the second piece - the encoders, weight generators and the flow network - is still running).

    split() replaces every tensor that crosses the boundary by a detached leaf; `loss.backward()` then stops at the leaves
    (first piece), and backward_rest() continues from the recorded originals with the gradients the leaves collected."""

    # cuts that hold an unfinished second piece.  Every backward driver (model.loss_backward, graph_step, a caller's own
    # `.backward()` followed by finish_all()) completes them from here, so the forward pass that detached at the boundary -
    # not an attribute of whichever optimiser happens to be stepped - decides whether a second piece has to run.
    _live = weakref.WeakSet()

    def __init__(self):
        self.pairs = []

    def begin_forward(self):
        """a new forward pass supersedes the boundary tensors of one whose backward never ran"""
        self.pairs = []
        BackwardCut._live.discard(self)

    def split(self, obj):
        if torch.is_tensor(obj):
            if not obj.requires_grad:
                return obj
            leaf = obj.detach().requires_grad_(True)
            self.pairs.append((obj, leaf))
            BackwardCut._live.add(self)
            return leaf
        if isinstance(obj, (list, tuple)):
            return type(obj)(self.split(o) for o in obj)
        return obj

    def has_grads(self):
        """the first piece of THIS cut's backward pass has run and the second is still due"""
        return any(l.grad is not None for _, l in self.pairs)

    def backward_rest(self):
        """second piece: continue from the recorded originals with the gradients the leaves collected.  A cut whose leaves have
        no gradient yet stays registered: its own backward pass has not run (a G forward with grad followed by a D
        loss_backward - whose finish_all() reaches every live cut - and only then the G backward: round-3 advisor)."""
        outs = [o for o, l in self.pairs if l.grad is not None]
        grads = [l.grad for o, l in self.pairs if l.grad is not None]
        if not outs:
            return
        self.pairs = []
        BackwardCut._live.discard(self)
        torch.autograd.backward(outs, grads)

    def abandon(self):
        """drop the boundary tensors of a forward pass whose backward will never run (an interrupted graph capture)"""
        self.pairs = []
        BackwardCut._live.discard(self)

    @classmethod
    def finish_all(cls):
        """run the remaining pieces of every forward pass that detached at a stage boundary and whose first piece has run (a cut
        inside the region behind another cut gets its gradients from that one's piece: repeat until nothing moves)"""
        moved = True
        while moved:
            moved = False
            for cut in list(cls._live):
                if cut.has_grads():
                    cut.backward_rest()
                    moved = True

    @classmethod
    def abandon_all(cls):
        for cut in list(cls._live):
            cut.abandon()


# ------------------------------------------------------------------------------------------------ discriminator
class NLayerDiscriminator(nn.Module):
    """Reference discriminator.py:61-102 with norm 'spectralinstance': k4 p2 PatchGAN returning every feature."""

    def __init__(self, input_nc, ndf=64, n_layers=3, getIntermFeat=False, stride=2):
        super().__init__()
        self.getIntermFeat, self.n_layers = getIntermFeat, n_layers
        self.model0 = _seq(Conv2d(input_nc, ndf, 4, stride=stride, padding=2), _Slot())
        nf = ndf
        for n in range(1, n_layers):
            nf_prev, nf = nf, min(nf * 2, 512)
            setattr(self, 'model%d' % n, _seq(_seq(Conv2d(nf_prev, nf, 4, stride=stride, padding=2, bias=False, spectral=True),
                                                   InstanceNorm(nf)), _Slot()))
        nf_prev, nf = nf, min(nf * 2, 512)
        setattr(self, 'model%d' % n_layers, _seq(_seq(Conv2d(nf_prev, nf, 4, stride=1, padding=2, bias=False, spectral=True),
                                                      InstanceNorm(nf)), _Slot()))
        setattr(self, 'model%d' % (n_layers + 1), _seq(Conv2d(nf, 1, 4, stride=1, padding=2)))
        self._sn_group = None

    def forward(self, x, sn=None):
        """sn: None - one power iteration, then the pass (the reference's forward pre-hook); or the sigmas of an earlier
        begin_pass() - Vid2VidModel's G step runs the real and the generated images in separate passes that count as ONE
        forward of the reference's batched call."""
        snap = self.begin_pass() if sn is None else sn
        for l, c in snap:
            l._sig_cached = c
        try:
            return self._run(x)
        finally:
            for l, _ in snap:
                l._sig_cached = None

    def begin_pass(self):
        """one power iteration of every spectral layer; returns the (layer, sigma / u / v snapshot) list for forward(sn=...)"""
        if self._sn_group is None:
            self._sn_group = ops.SpectralGroup(spectral_layers(self))
        self._sn_group.update(self.training)
        snap = [(l, l._sig_cached) for l in self._sn_group.layers]
        for l, _ in snap:
            l._sig_cached = None
        return snap

    def _run(self, x):
        res = []
        x = self.model0[0](x, act=ACT_LRELU)
        res.append(x)
        for n in range(1, self.n_layers + 1):
            conv, norm = getattr(self, 'model%d' % n)[0]
            x = norm(conv(x, stats=-1), act=ACT_LRELU)
            res.append(x)
        x = getattr(self, 'model%d' % (self.n_layers + 1))[0](x)
        res.append(x)
        return res if self.getIntermFeat else res[-1]


class AdaptiveDiscriminator(NLayerDiscriminator):
    """Reference discriminator.py:104-209 (`--netD_subarch adaptive`): the first `adaptive_layers` convolutions take weights
    GENERATED from the reference image - encoder_n (k4 s2 p2 + LeakyReLU) on [ref label | ref image], adaptive average pooling of
    every encoded channel to (fineSize / 8 / aspect, fineSize / 8), one Linear per layer from the pooled map to a k4 x k4 filter
    row - applied per sample with stride 2, InstanceNorm (no affine) and LeakyReLU; the remaining layers are the spectral
    PatchGAN's.  Same attribute names / state_dict keys as the reference."""

    def __init__(self, opt, input_nc, ndf=64, n_layers=3, getIntermFeat=False, adaptive_layers=1):
        nn.Module.__init__(self)
        self.getIntermFeat, self.n_layers, self.adaptive_layers = getIntermFeat, n_layers, adaptive_layers
        self.input_nc, self.ndf = input_nc, ndf
        self.sw = opt.fineSize // 8
        self.sh = int(self.sw / opt.aspect_ratio)
        ch = self.sh * self.sw
        nf = ndf
        self.fc_0 = Linear(ch, input_nc * 16, spectral=False)
        self.encoder_0 = _seq(Conv2d(input_nc, ndf, 4, stride=2, padding=2), _Slot())
        for n in range(1, adaptive_layers):
            nf_prev, nf = nf, min(nf * 2, 512)
            setattr(self, 'fc_%d' % n, Linear(ch, nf_prev * 16, spectral=False))
            setattr(self, 'encoder_%d' % n, _seq(Conv2d(nf_prev, nf, 4, stride=2, padding=2), _Slot()))
        nf = ndf * (2 ** (adaptive_layers - 1))
        for n in range(adaptive_layers, n_layers + 1):
            nf_prev, nf = nf, min(nf * 2, 512)
            setattr(self, 'model%d' % n, _seq(_seq(Conv2d(nf_prev, nf, 4, stride=2 if n != n_layers else 1, padding=2, bias=False,
                                                          spectral=True), InstanceNorm(nf)), _Slot()))
        setattr(self, 'model%d' % (n_layers + 1), _seq(Conv2d(nf, 1, 4, stride=1, padding=2)))
        self._sn_group = None

    def forward(self, x, ref=None, sn=None):
        if ref is None:
            raise ValueError("the adaptive discriminator needs the reference [label | image] tensor")
        snap = self.begin_pass() if sn is None else sn
        for l, c in snap:
            l._sig_cached = c
        try:
            return self._run_adaptive(x, ref)
        finally:
            for l, _ in snap:
                l._sig_cached = None

    def _run_adaptive(self, x, ref):
        enc, r = [], ref
        for n in range(self.adaptive_layers):                       # encode (discriminator.py:186-190)
            r = getattr(self, 'encoder_%d' % n)[0](r, act=ACT_LRELU)
            enc.append(r)
        res = []
        nf, nf_prev = self.ndf, self.input_nc
        for n in range(self.adaptive_layers):                       # gen_conv_weights + batch_conv (discriminator.py:142-170,192-197)
            e = enc[n]
            b, ch = e.shape[0], e.shape[1]
            pooled = ops.adaptive_avgpool(e, self.sh, self.sw).reshape(b * ch, self.sh * self.sw)
            wgt = getattr(self, 'fc_%d' % n)(pooled).view(b, nf, nf_prev, 4, 4)
            x = ops.batch_conv(x, wgt, None, stride=2, allow_half=False)
            x = ops.norm_act(x, None, None, None, None, instance=True, eps=1e-5, act=ACT_LRELU)
            res.append(x)
            nf_prev, nf = nf, min(nf * 2, 512)
        for n in range(self.adaptive_layers, self.n_layers + 1):
            conv, norm = getattr(self, 'model%d' % n)[0]
            x = norm(conv(x, stats=-1), act=ACT_LRELU)
            res.append(x)
        x = getattr(self, 'model%d' % (self.n_layers + 1))[0](x)
        res.append(x)
        return res if self.getIntermFeat else res[-1]


class MultiscaleDiscriminator(nn.Module):
    """Reference discriminator.py:16-58 (subarch 'n_layers' or 'adaptive')."""

    def __init__(self, opt, input_nc, ndf=64, n_layers=3, num_D=1, getIntermFeat=False, stride=2, subarch='n_layers'):
        super().__init__()
        self.num_D, self.getIntermFeat, self.subarch = num_D, getIntermFeat, subarch
        for i in range(num_D):
            if subarch == 'adaptive':
                d = AdaptiveDiscriminator(opt, input_nc, ndf, n_layers, getIntermFeat, getattr(opt, 'adaptive_D_layers', 1))
            else:
                d = NLayerDiscriminator(input_nc, ndf, n_layers, getIntermFeat, stride)
            setattr(self, 'discriminator_%d' % i, d)

    def begin_pass(self):
        return [getattr(self, 'discriminator_%d' % i).begin_pass() for i in range(self.num_D)]

    def forward(self, x, ref=None, sn=None):
        result = []
        adaptive = self.subarch == 'adaptive'
        for i in range(self.num_D):
            d = getattr(self, 'discriminator_%d' % i)
            out = d(x, ref, sn=sn[i] if sn is not None else None) if adaptive else d(x, sn=sn[i] if sn is not None else None)
            result.append(out if self.getIntermFeat else [out])
            if i + 1 < self.num_D:
                x = ops.avgpool3s2(x)            # discriminator.py:28,56 (scripts/face/train_g8_512.sh: --num_D 2)
                if adaptive:
                    ref = ops.avgpool3s2(ref)
        return result


def define_G(opt):
    """Reference models/networks/__init__.py:29-39."""
    return FewShotGenerator(opt)


def define_D(opt, input_nc, ndf, n_layers_D, norm='spectralinstance', subarch='n_layers', num_D=1, getIntermFeat=False,
             stride=2, gpu_ids=()):
    """Reference models/networks/__init__.py:41-55."""
    if norm != 'spectralinstance' or subarch not in ('n_layers', 'adaptive'):
        raise NotImplementedError("only the 'spectralinstance' PatchGANs (n_layers, adaptive) are on the hot path")
    if subarch == 'adaptive' and str(getattr(opt, 'amp', 'O0')) != 'O0':
        raise NotImplementedError("--netD_subarch adaptive under --amp")
    return MultiscaleDiscriminator(opt, input_nc, ndf, n_layers_D, num_D, getIntermFeat, stride, subarch)
