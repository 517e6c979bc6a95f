"""CPU oracle for the few-shot-vid2vid G/D hot path.  TEST INFRASTRUCTURE - never imported by the product path.

A plain fp32 PyTorch-on-CPU restatement of the reference's algorithm, written functionally (state_dict in,
tensors out) so that it shares no code with the product modules in few-shot-vid2vid_amd/.  Every function cites
the reference lines it follows (paths relative to /root/reference).  The restatement is pinned against the real
reference by the golden fixtures in tests/golden/ (minted by oracle/make_golden.py from the unmodified reference modules,
imported in the build container through oracle/ref_import.py) and the tests of tests/test_golden.py that replay them
(`test_oracle_reproduces_reference_*`).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------- operators
def actvn(x):
    """models/networks/architecture.py:15-17"""
    return F.leaky_relu(x, 0.2)


def get_grid(b, rows, cols, dtype=torch.float32):
    """models/networks/base_network.py:13-26 (dtype is fp32 in the reference; fp64 only for the noise-floor runs)"""
    hor = torch.linspace(-1.0, 1.0, cols, dtype=dtype).view(1, 1, 1, cols).expand(b, 1, rows, cols)
    ver = torch.linspace(-1.0, 1.0, rows, dtype=dtype).view(1, 1, rows, 1).expand(b, 1, rows, cols)
    return torch.cat([hor, ver], 1)


def resample(image, flow):
    """models/networks/base_network.py:28-37"""
    b, c, h, w = image.size()
    grid = get_grid(b, h, w, flow.dtype)
    flow = torch.cat([flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)], dim=1)
    final_grid = (grid + flow).permute(0, 2, 3, 1)
    image = image.to(flow.dtype)
    return F.grid_sample(image, final_grid, mode='bilinear', padding_mode='border', align_corners=True)


def resample_taps(flow):
    """Integer tap indices ATen's grid_sample selects for `resample` (ATen/native/GridSampler.h:27-60,164-172)."""
    b, _, h, w = flow.size()
    grid = get_grid(b, h, w)
    f = torch.cat([flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)], dim=1)
    g = grid + f
    ix = ((g[:, 0] + 1) / 2) * (w - 1)
    iy = ((g[:, 1] + 1) / 2) * (h - 1)
    ix = torch.clamp(ix, 0, w - 1)
    iy = torch.clamp(iy, 0, h - 1)
    return torch.stack([torch.floor(ix), torch.floor(iy)], dim=-1).to(torch.int32)


# Arithmetic of the convolutions.  None: the reference's own (F.conv2d in the tensors' dtype).  The `--amp` parity tests install
# oracle/np_oracle.amp_conv2d here (`with arithmetic(...)`): the call sites below that the product runs on its half-precision
# kernels go through _conv2d - including the gamma / beta convolutions inside a SPADE layer whose widths fit the half form of the
# product's fused modulation kernel (_spade_exact); nn.Linear (the weight generators: grouped fp32 launches) keeps calling
# F.linear directly.
_ARITH = {'conv2d': None}


class arithmetic:
    """`with arithmetic(fn):` fn(x, w, b, stride, padding, dx_half, per_sample) replaces F.conv2d at the routed call sites"""

    def __init__(self, conv2d):
        self.fn = conv2d

    def __enter__(self):
        self.prev = _ARITH['conv2d']
        _ARITH['conv2d'] = self.fn

    def __exit__(self, *exc):
        _ARITH['conv2d'] = self.prev
        return False


def _conv2d(x, w, b=None, stride=1, padding=0, dx_half=False, per_sample=False):
    f = _ARITH['conv2d']
    if f is None:
        return F.conv2d(x, w, b, stride=stride, padding=padding)
    return f(x, w, b, stride, padding, dx_half, per_sample)


def batch_conv(x, weight, bias=None, stride=1, exact=False):
    """models/networks/base_network.py:56-71 (stride >= 1 branch; every call site uses it).  exact: a call from inside a SPADE
    layer (see _ARITH)"""
    if weight is None:
        return x
    padding = weight.size(-1) // 2
    ys = []
    for i in range(x.size(0)):
        if exact:
            ys.append(F.conv2d(x[i:i + 1], weight[i], None if bias is None else bias[i], padding=padding, stride=stride))
        else:
            ys.append(_conv2d(x[i:i + 1], weight[i], None if bias is None else bias[i], stride, padding, per_sample=True))
    return torch.cat(ys, 0)


_MODE = {'training': True}
_RAW_SN = {'on': False}      # _sn_conv under an `--amp` arithmetic asks spectral_weight for (W, sigma) instead of W / sigma


class eval_mode:
    """`with eval_mode():` evaluates the state-dict networks the way model.eval() does (test.py:27): BatchNorm with
    its running statistics, spectral norm from the stored u / v without a power iteration."""

    def __enter__(self):
        self.prev = _MODE['training']
        _MODE['training'] = False

    def __exit__(self, *exc):
        _MODE['training'] = self.prev
        return False


def spectral_weight(sd, prefix, training=None, eps=1e-12):
    """torch.nn.utils.spectral_norm semantics (one power iteration in training mode); updates u, v in `sd`."""
    if training is None:
        training = _MODE['training']
    w = sd[prefix + 'weight_orig']
    u, v = sd[prefix + 'weight_u'], sd[prefix + 'weight_v']
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v_new = F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps)
            u_new = F.normalize(torch.mv(wm, v_new), dim=0, eps=eps)
            sd[prefix + 'weight_u'] = u_new
            sd[prefix + 'weight_v'] = v_new
            u, v = u_new, v_new
    sigma = torch.dot(u, torch.mv(wm, v))
    if _RAW_SN['on']:
        return w, sigma
    return w / sigma


def batch_norm_train(x, weight=None, bias=None, eps=1e-5):
    """nn.BatchNorm2d / apex SyncBatchNorm in one process, training mode (normalization.py:33,80)"""
    return F.batch_norm(x, None, None, weight, bias, True, 0.1, eps)


def instance_norm(x, weight=None, bias=None, eps=0.1):
    """nn.InstanceNorm2d(affine=True, eps=0.1) (normalization.py:35,82)"""
    return F.instance_norm(x, None, None, weight, bias, True, 0.1, eps)


def spade(x, maps, fixed, generated=None):
    """SPADE.forward, models/networks/normalization.py:37-52.

    maps: list of tensors / None; fixed[k] = (wg, bg, wb, bb) conv weights of mlp_gamma{k}/mlp_beta{k} (k>0, or k==0
    when not params_free); generated = ((wg, bg), (wb, bb)) per-sample weights for map 0 or None.
    """
    out = batch_norm_train(x)
    exact = _spade_exact(x, maps)
    for i, m in enumerate(maps):
        if m is None:
            continue
        m = F.interpolate(m, size=x.shape[2:])
        if generated is None or i != 0:
            wg, bg, wb, bb = fixed[i]
            pad = wg.shape[-1] // 2
            gamma = F.conv2d(m, wg, bg, padding=pad) if exact else _conv2d(m, wg, bg, 1, pad)
            beta = F.conv2d(m, wb, bb, padding=pad) if exact else _conv2d(m, wb, bb, 1, pad)
        else:
            (wg, bg), (wb, bb) = generated
            gamma = batch_conv(m, wg, bg, exact=exact)
            beta = batch_conv(m, wb, bb, exact=exact)
        out = out * (1 + gamma) + beta
    return out


def _spade_exact(x, maps):
    """Under an installed arithmetic (_ARITH) the gamma / beta convolutions of a SPADE layer follow it when the whole layer fits the
    half-precision form - normalised channels a multiple of 16, every map's channels a multiple of 8 (all layers of the BASELINE
    configurations) - and are exact otherwise."""
    if _ARITH['conv2d'] is None:
        return True
    return not (x.shape[1] % 16 == 0 and all(m is None or m.shape[1] % 8 == 0 for m in maps))


# ----------------------------------------------------------------------------------------------- configuration
class Cfg:
    """The subset of the reference's `opt` the hot path reads (options/base_options.py:52-132 defaults)."""

    def __init__(self, **kw):
        self.ngf = 32; self.ndf = 32; self.nff = 32
        self.n_downsample_G = 5; self.n_adaptive_layers = 4; self.n_fc_layers = 2
        self.n_downsample_F = 3; self.n_blocks_F = 6; self.flow_multiplier = 20
        self.adaptive_spade = True; self.spade_combine = False; self.n_sc_layers = 2; self.warp_ref = False
        self.add_raw_output_loss = False
        self.netD_subarch = 'n_layers'; self.adaptive_D_layers = 1
        self.label_nc = 0; self.input_nc = 6; self.output_nc = 3
        self.basic_point_only = False; self.add_face_D = False; self.lambda_face = 10.0
        self.lambda_temp = 0.0; self.n_frames_D = 2; self.n_frames_G = 2
        self.n_shot = 1; self.n_downsample_A = 2
        self.isTrain = True; self.lr = 0.0004; self.beta1 = 0.5; self.beta2 = 0.999; self.no_TTUR = False
        self.n_layers_D = 4; self.num_D = 1
        self.dataset_mode = 'fewshot_pose'; self.remove_face_labels = False; self.pose_type = 'both'
        self.lambda_feat = 10.0; self.lambda_flow = 10.0; self.lambda_mask = 10.0; self.lambda_vgg = 10.0
        self.fineSize = 512; self.aspect_ratio = 1.0
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)

    @property
    def label_channels(self):
        return self.label_nc if self.label_nc != 0 else self.input_nc

    @property
    def ch(self):
        """models/networks/generator.py:25-29"""
        nf_max = min(1024, self.ngf * (2 ** self.n_downsample_G))
        return [min(nf_max, self.ngf * (2 ** i)) for i in range(self.n_downsample_G + 2)]

    @property
    def pose(self):
        return 'pose' in self.dataset_mode


def cfg_from_opt(opt):
    keys = Cfg().__dict__.keys()
    return Cfg(**{k: getattr(opt, k) for k in keys if hasattr(opt, k)})


# ----------------------------------------------------------------------------------------------- layers over a state dict
def _bn(sd, p, x, affine):
    return F.batch_norm(x, sd[p + 'running_mean'], sd[p + 'running_var'], sd[p + 'weight'] if affine else None,
                        sd[p + 'bias'] if affine else None, _MODE['training'], 0.1, 1e-5)


def _sn_conv(sd, p, x, stride=1, padding=1, dx_half=False):
    """dx_half (only read by an `--amp` arithmetic): the input is a tensor the product stores as half - the output of a SPADE
    layer with at least one map, the packed discriminator input -, so its gradient is rounded to half as well"""
    if _ARITH['conv2d'] is None:
        return _conv2d(x, spectral_weight(sd, p), sd.get(p + 'bias'), stride, padding, dx_half=dx_half)
    # the product keeps W itself in its (half) operand layout and applies 1 / sigma to the fp32 accumulator (it changes every
    # forward; csrc/conv_h.hip `wscale`): the rounding to half sees W, not W / sigma
    _RAW_SN['on'] = True
    try:
        w, sigma = spectral_weight(sd, p)
    finally:
        _RAW_SN['on'] = False
    y = _conv2d(x, w, None, stride, padding, dx_half=dx_half) / sigma
    b = sd.get(p + 'bias')
    return y if b is None else y + b.view(1, -1, 1, 1)


def _plain_conv(sd, p, x, stride=1, padding=1, dx_half=False):
    return _conv2d(x, sd[p + 'weight'], sd.get(p + 'bias'), stride, padding, dx_half=dx_half)


def spade_conv2d(sd, p, x, stride=1):
    """SPADEConv2d with a plain (affine) BatchNorm: architecture.py:57-69, norm='spectralsyncbatch'."""
    return actvn(_bn(sd, p + 'bn.', _sn_conv(sd, p + 'conv.', x, stride, 1), True))


def _spade_layer(sd, p, x, maps, gen):
    """One SPADE instance `p` (normalization.py:18-52): fixed weights come from sd, generated ones from `gen`."""
    out = _bn(sd, p + 'norm.', x, False)
    if not isinstance(maps, list):
        maps = [maps]
    exact = _spade_exact(x, maps)
    for i, m in enumerate(maps):
        if m is None:
            continue
        m = F.interpolate(m, size=x.shape[2:])
        if gen is None or i != 0:
            s = str(i + 1) if i > 0 else ''
            cv = F.conv2d if exact else _conv2d
            gamma = cv(m, sd[p + 'mlp_gamma%s.weight' % s], sd[p + 'mlp_gamma%s.bias' % s])
            beta = cv(m, sd[p + 'mlp_beta%s.weight' % s], sd[p + 'mlp_beta%s.bias' % s])
        else:
            # normalization.py:48-50: weights[0][j] / weights[1][j] with j = min(i, len(weights[0]) - 1) picks the
            # weight TENSOR of the (weight, bias) pair, so the generated biases are not applied.
            j = min(i, len(gen[0]) - 1)
            gamma = batch_conv(m, gen[0][j], exact=exact)
            beta = batch_conv(m, gen[1][j], exact=exact)
        out = out * (1 + gamma) + beta
    return out


def spade_resblock(sd, p, x, maps, norm_weights, fin, fout):
    """SPADEResnetBlock with SPADE norms (architecture.py:71-108), conv_params_free=False."""
    nw = norm_weights if norm_weights else [None] * 3
    hm = any(m is not None for m in (maps if isinstance(maps, list) else [maps]))       # a modulated (half-stored) conv input
    if fin != fout:
        x_s = _sn_conv(sd, p + 'conv_s.', _spade_layer(sd, p + 'bn_s.', x, maps, nw[2]), 1, 0, dx_half=hm)
    else:
        x_s = x
    dx = _sn_conv(sd, p + 'conv_0.', actvn(_spade_layer(sd, p + 'bn_0.', x, maps, nw[0])), dx_half=hm)
    dx = _sn_conv(sd, p + 'conv_1.', actvn(_spade_layer(sd, p + 'bn_1.', dx, maps, nw[1])), dx_half=hm)
    return x_s + 1.0 * dx


def bn_resblock(sd, p, x):
    """SPADEResnetBlock(ch, ch, norm='spectralsyncbatch') of the FlowGenerator (generator.py:479)."""
    dx = _sn_conv(sd, p + 'conv_0.', actvn(_bn(sd, p + 'bn_0.', x, True)))
    dx = _sn_conv(sd, p + 'conv_1.', actvn(_bn(sd, p + 'bn_1.', dx, True)))
    return x + 1.0 * dx


def _mlp(sd, p, x, n_fc):
    """fc_spade_*: Linear-LReLU x n_fc then Linear, all spectral-normalised (generator.py:103-110)."""
    for k in range(n_fc):
        x = actvn(F.linear(x, spectral_weight(sd, p + '%d.' % (2 * k)), sd[p + '%d.bias' % (2 * k)]))
    k = n_fc
    return F.linear(x, spectral_weight(sd, p + '%d.' % (2 * k)), sd[p + '%d.bias' % (2 * k)])


def _pair(x, cout, cin):
    """reshape_weight leaf (base_network.py:154-167): per-sample (weight[B,cout,cin,1,1], bias[B,cout])."""
    return [x[:, :-cout].reshape(x.shape[0], cout, cin, 1, 1), x[:, -cout:]]


# ----------------------------------------------------------------------------------------------- generator
def attention_module(sd, cfg, x, label, label_ref, attention=None):
    """generator.py:291-316: key / query encoders on the label maps, energy over all n_shot * HW reference positions,
    softmax over them, attention-weighted sum of the reference features."""
    bn, c, h, w = x.shape
    n = cfg.n_shot
    b = bn // n

    def encode(img, name):
        y = spade_conv2d(sd, name + '_first.', img)
        for i in range(cfg.n_downsample_A):
            y = spade_conv2d(sd, '%s_%d.' % (name, i), y, 2)
        return y
    if attention is None:
        key = encode(label_ref, 'atn_key')
        query = encode(label, 'atn_query')
        key = key.view(b, n, c, -1).permute(0, 1, 3, 2).contiguous().view(b, -1, c)
        query = query.view(b, c, -1)
        attention = torch.softmax(torch.bmm(key, query), dim=1)
    x = x.view(b, n, c, h * w).permute(0, 2, 1, 3).contiguous().view(b, c, -1)
    out = torch.bmm(x, attention).view(b, c, h, w)
    atn_vis = attention.view(b, n, h * w, h * w).sum(2).view(b, n, h, w)
    return out, attention, atn_vis[-1:, 0:1]


def pick_ref(refs, ref_idx):
    """base_network.py:40-47"""
    if ref_idx is None:
        return refs[:, 0]
    idx = ref_idx.long().view(-1, 1, 1, 1, 1)
    return refs.gather(1, idx.expand_as(refs)[:, 0:1])[:, 0]


def reference_encoding(sd, cfg, img_ref, label_ref, label=None):
    """generator.py:341-393, use_label_ref == 'mul'; returns (x, encoded_ref, atn_vis, ref_idx)."""
    n = cfg.n_downsample_G
    x = spade_conv2d(sd, 'ref_img_first.', img_ref)
    xl = spade_conv2d(sd, 'ref_label_first.', label_ref)
    atn_vis = ref_idx = None
    for i in range(n):
        x = spade_conv2d(sd, 'ref_img_down_%d.' % i, x, 2)
        xl = spade_conv2d(sd, 'ref_label_down_%d.' % i, xl, 2)
        if cfg.n_shot > 1 and i == cfg.n_downsample_A - 1:
            x, atn, atn_vis = attention_module(sd, cfg, x, label, label_ref)
            xl, _, _ = attention_module(sd, cfg, xl, None, None, atn)
            ref_idx = torch.argmax(atn.view(label.shape[0], cfg.n_shot, -1).sum(2), dim=1)
    feats_i, feats_l = [x], [xl]
    for i in reversed(range(n)):
        feats_i.append(spade_conv2d(sd, 'ref_img_up_%d.' % i, feats_i[-1]))
        feats_l.append(spade_conv2d(sd, 'ref_label_up_%d.' % i, feats_l[-1]))
    enc = []
    for fi, fl in zip(feats_i, feats_l):
        b, c, h, w = fi.shape
        sm = torch.softmax(fl, dim=1)
        enc.append((fi.view(b, c, 1, h * w) * sm.view(b, 1, c, h * w)).sum(3, keepdim=True))
    return x, enc[::-1], atn_vis, ref_idx


def spade_weights(sd, cfg, feat, i):
    """get_SPADE_weights (generator.py:245-273) for spade_ks = embed_ks = 1."""
    ch = cfg.ch
    cin, cout = ch[i], ch[i + 1]
    b = feat.shape[0]
    rows = feat.reshape(b * feat.shape[1], -1)
    nfc = cfg.n_fc_layers
    fe = _mlp(sd, 'fc_spade_e_%d.' % i, rows, nfc).view(b, -1)
    embed = _pair(fe[:, :-cin], cin, cout)

    def two(name, co):
        f = _mlp(sd, 'fc_spade_%s_%d.' % (name, i), rows, nfc).view(b, -1)
        half = co * cin + co
        return [_pair(f[:, :half], co, cin), _pair(f[:, half:2 * half], co, cin)]
    return embed, [two('0', cout), two('1', cin), two('s', cout)]


def label_embedder(sd, p, cfg, x, weights=None, unet=False, params_free_layers=0):
    """LabelEmbedder.forward (generator.py:541-572); netS='encoderdecoder' (label) or 'unet' (warp embeddings)."""
    n = cfg.n_downsample_G
    out = [actvn(_plain_conv(sd, p + 'conv_first.0.', x))]
    for i in range(n):
        out.append(actvn(_plain_conv(sd, p + 'down_%d.0.' % i, out[-1], 2)))
    if not unet:
        out = [out[-1]]
    for i in reversed(range(n)):
        cur = out[-1]
        if unet and i != n - 1:
            cur = torch.cat([cur, out[i + 1]], dim=1)
        cur = F.interpolate(cur, scale_factor=2)
        if i >= params_free_layers:
            out.append(actvn(_plain_conv(sd, p + 'up_%d.1.' % i, cur)))
        else:
            out.append(actvn(batch_conv(cur, weights[i][0], weights[i][1])))
    if unet:
        out = out[n:]
    return out[::-1]


def flow_generator(sd, p, cfg, label, label_prev, img_prev):
    """FlowGenerator.forward (generator.py:456-504)."""
    x = torch.cat([label, label_prev, img_prev], dim=1)
    nd = cfg.n_downsample_F
    x = actvn(_bn(sd, p + 'down_flow.0.1.', _sn_conv(sd, p + 'down_flow.0.0.', x), True))
    for i in range(nd):
        k = 2 * (i + 1)
        x = actvn(_bn(sd, p + 'down_flow.%d.1.' % k, _sn_conv(sd, p + 'down_flow.%d.0.' % k, x, 2), True))
    for i in range(cfg.n_blocks_F):
        x = bn_resblock(sd, p + 'res_flow.%d.' % i, x)
    for i in range(nd):
        k = 3 * i + 1
        x = F.interpolate(x, scale_factor=2)
        x = actvn(_bn(sd, p + 'up_flow.%d.1.' % k, _sn_conv(sd, p + 'up_flow.%d.0.' % k, x), True))
    flow = _plain_conv(sd, p + 'conv_flow.0.', x) * cfg.flow_multiplier
    mask = torch.sigmoid(_plain_conv(sd, p + 'conv_mask.0.', x))
    return flow, mask


def generator_forward(sd, cfg, label, label_refs, img_refs, prev=(None, None), warp_prev=False,
                      flow_temp_prefix='flow_network_ref.', return_ref=False):
    """FewShotGenerator.forward (generator.py:181-229).  return_ref: also return (atn_vis, ref_idx) of the attention
    module (n_shot > 1; None otherwise).

    Returns (img_final, [flow_ref, flow_prev], [mask_ref, mask_prev], img_raw, [warp_ref, warp_prev]).
    flow_temp_prefix: the reference shares ONE FlowGenerator between the reference and the previous-frame branch when
    n_frames_G == 2, warp_ref and not sep_flow_prev (generator.py:159-166) - both state_dict names then alias the
    same tensors, so the oracle reads (and power-iterates) them under a single prefix.
    """
    ch, n = cfg.ch, cfg.n_downsample_G
    b = img_refs.shape[0]
    img_ref = img_refs.reshape(b * img_refs.shape[1], -1, *img_refs.shape[-2:])
    label_ref = label_refs.reshape(b * label_refs.shape[1], -1, *label_refs.shape[-2:])
    x, enc, atn_vis, ref_idx = reference_encoding(sd, cfg, img_ref, label_ref, label)
    if ref_idx is not None:            # flow_generation works on the attended reference (generator.py:425)
        img_ref, label_ref = pick_ref(img_refs, ref_idx), pick_ref(label_refs, ref_idx)
    embed_w, norm_w = [], []
    for i in range(cfg.n_adaptive_layers):
        e, nw = spade_weights(sd, cfg, enc[min(len(enc) - 1, i + 1)], i)
        embed_w.append(e); norm_w.append(nw)
    enc_label = label_embedder(sd, 'label_embedding.', cfg, label, embed_w, False, cfg.n_adaptive_layers)
    # flow (generator.py:424-445)
    flow, mask, warp = [None, None], [None, None], [None, None]
    ds = [None, None]
    label_prev, img_prev = prev
    if cfg.warp_ref:
        flow[0], mask[0] = flow_generator(sd, 'flow_network_ref.', cfg, label, label_ref, img_ref)
        warp[0] = resample(img_ref, flow[0])[:, :3]
    if warp_prev and label_prev is not None:
        flow[1], mask[1] = flow_generator(sd, flow_temp_prefix, cfg, label, label_prev, img_prev)
        warp[1] = resample(img_prev[:, -3:], flow[1])
    if cfg.spade_combine:
        if cfg.warp_ref:
            ds[0] = torch.cat([warp[0], mask[0]], dim=1)
        if warp[1] is not None:
            ds[1] = torch.cat([warp[1], mask[1]], dim=1)
        emb_ref = label_embedder(sd, 'img_ref_embedding.', cfg, ds[0], None, True, 0)
        emb_prev = label_embedder(sd, 'img_prev_embedding.', cfg, ds[1], None, True, 0) if ds[1] is not None else None
        enc_raw = [enc_label[i] for i in range(cfg.n_sc_layers)] if cfg.add_raw_output_loss else None      # generator.py:195
        for i in range(cfg.n_sc_layers):
            enc_label[i] = [enc_label[i], emb_ref[i], emb_prev[i] if emb_prev is not None else None]
    else:
        enc_raw = None
    x_raw = None
    for i in range(n, -1, -1):
        nw = norm_w[i] if (cfg.adaptive_spade and i < cfg.n_adaptive_layers) else None
        if enc_raw is not None and i < cfg.n_sc_layers:           # generator.py:202-205 (the raw branch runs first)
            if i == cfg.n_sc_layers - 1:
                x_raw = x
            x_raw = spade_resblock(sd, 'up_%d.' % i, x_raw, enc_raw[i], nw, ch[i + 1], ch[i])
            if i != 0:
                x_raw = F.interpolate(x_raw, scale_factor=2)
        x = spade_resblock(sd, 'up_%d.' % i, x, enc_label[i], nw, ch[i + 1], ch[i])
        if i != 0:
            x = F.interpolate(x, scale_factor=2)
    img_raw = torch.tanh(_plain_conv(sd, 'conv_img.', actvn(x)))
    if not cfg.spade_combine:
        img_final = img_raw
        if cfg.warp_ref:
            img_final = img_raw * mask[0] + warp[0] * (1 - mask[0])
        elif not warp_prev:
            img_raw = None
        if warp[1] is not None:
            img_final = img_final * mask[1] + warp[1] * (1 - mask[1])
    else:
        img_final = img_raw
        img_raw = torch.tanh(_plain_conv(sd, 'conv_img.', actvn(x_raw))) if x_raw is not None else None       # generator.py:227
    if return_ref:
        return img_final, flow, mask, img_raw, warp, atn_vis, ref_idx
    return img_final, flow, mask, img_raw, warp


# ----------------------------------------------------------------------------------------------- discriminator
def nlayer_discriminator(sd, p, x, n_layers=4, packed_input=False):
    """NLayerDiscriminator.forward (discriminator.py:61-102), norm 'spectralinstance', getIntermFeat=True.  packed_input: x is the
    packed [ref | label | image] tensor of a single-scale discriminator (see _sn_conv: dx_half)"""
    feats = []
    x = actvn(_plain_conv(sd, p + 'model0.0.', x, 2, 2, dx_half=packed_input))
    feats.append(x)
    for n in range(1, n_layers + 1):
        stride = 2 if n < n_layers else 1
        x = _sn_conv(sd, p + 'model%d.0.0.' % n, x, stride, 2)
        x = actvn(instance_norm(x, sd[p + 'model%d.0.1.weight' % n], sd[p + 'model%d.0.1.bias' % n], 0.1))
        feats.append(x)
    x = _plain_conv(sd, p + 'model%d.0.' % (n_layers + 1), x, 1, 2)
    feats.append(x)
    return feats


def adaptive_discriminator(sd, p, cfg, x, ref, n_layers=4, adaptive_layers=1):
    """AdaptiveDiscriminator.forward (discriminator.py:104-209), norm 'spectralinstance', getIntermFeat=True: encode the reference
    (encoder_n: k4 s2 p2 + LeakyReLU), pool every channel to (sh, sw), one Linear per layer from the pooled map to a filter row, the
    generated convolution per sample (stride 2, InstanceNorm2d without affine, LeakyReLU), then the spectral PatchGAN layers."""
    sw = cfg.fineSize // 8
    sh = int(sw / cfg.aspect_ratio)
    enc, r = [], ref
    for n in range(adaptive_layers):
        r = actvn(F.conv2d(r, sd[p + 'encoder_%d.0.weight' % n], sd[p + 'encoder_%d.0.bias' % n], stride=2, padding=2))
        enc.append(r)
    feats = []
    nf_prev, nf = x.shape[1], cfg.ndf
    for n in range(adaptive_layers):
        e = enc[n]
        b, ch = e.shape[0], e.shape[1]
        pooled = F.adaptive_avg_pool2d(e, (sh, sw)).reshape(b * ch, -1)
        w = F.linear(pooled, sd[p + 'fc_%d.weight' % n], sd[p + 'fc_%d.bias' % n]).view(b, nf, nf_prev, 4, 4)
        x = torch.cat([F.conv2d(x[i:i + 1], w[i], stride=2, padding=2) for i in range(x.shape[0])], 0)
        x = actvn(F.instance_norm(x, eps=1e-5))
        feats.append(x)
        nf_prev, nf = nf, min(nf * 2, 512)
    for n in range(adaptive_layers, n_layers + 1):
        stride = 2 if n != n_layers else 1
        x = _sn_conv(sd, p + 'model%d.0.0.' % n, x, stride, 2)
        x = actvn(instance_norm(x, sd[p + 'model%d.0.1.weight' % n], sd[p + 'model%d.0.1.bias' % n], 0.1))
        feats.append(x)
    x = _plain_conv(sd, p + 'model%d.0.' % (n_layers + 1), x, 1, 2)
    feats.append(x)
    return feats


def multiscale_discriminator(sd, cfg, x, num_D=None, ref=None):
    """MultiscaleDiscriminator.forward (discriminator.py:49-58).  ref: the second input of the adaptive sub-architecture."""
    res = []
    nd = cfg.num_D if num_D is None else num_D
    for i in range(nd):
        if ref is not None:
            res.append(adaptive_discriminator(sd, 'discriminator_%d.' % i, cfg, x, ref, cfg.n_layers_D, cfg.adaptive_D_layers))
            ref = F.avg_pool2d(ref, 3, stride=2, padding=1, count_include_pad=False)
        else:
            res.append(nlayer_discriminator(sd, 'discriminator_%d.' % i, x, cfg.n_layers_D, packed_input=(nd == 1)))
        x = F.avg_pool2d(x, 3, stride=2, padding=1, count_include_pad=False)
    return res


# ----------------------------------------------------------------------------------------------- per-step losses
def face_mask(pose_ch):
    """models/input_process.py:80-94"""
    if pose_ch.dim() == 3:
        pose_ch = pose_ch.unsqueeze(1)
    part = (pose_ch / 2 + 0.5) * 24
    mask = torch.zeros_like(part, dtype=torch.bool)
    for j in (23, 24):
        mask = mask | ((part > j - 0.1) & (part < j + 0.1))
    return mask.float()


def part_mask(pose):
    """models/input_process.py:64-78; pose [B, T, H, W] -> [B, T, 9, H, W]"""
    groups = [[0], [1, 2], [3, 4], [5, 6], [7, 9, 8, 10], [11, 13, 12, 14], [15, 17, 16, 18], [19, 21, 20, 22], [23, 24]]
    bo, t, h, w = pose.shape
    p = pose.reshape(-1, h, w)
    part = (p / 2 + 0.5) * 24
    mask = torch.zeros(p.shape[0], len(groups), h, w, dtype=torch.bool)
    for i, grp in enumerate(groups):
        for j in grp:
            mask[:, i] = mask[:, i] | ((part > j - 0.1) & (part < j + 0.1))
    return mask.view(bo, t, -1, h, w).float()


def use_valid_labels(cfg, pose):
    """models/input_process.py:97-113"""
    if not cfg.pose or pose is None:
        return pose
    if cfg.pose_type == 'open':
        return pose[:, 3:] if pose.dim() == 4 else pose[:, :, 3:]
    if cfg.remove_face_labels:
        if pose.dim() == 4:
            fm = face_mask(pose[:, 2])
            return torch.cat([pose[:, :3] * (1 - fm) - fm, pose[:, 3:]], dim=1)
        fm = face_mask(pose[:, :, 2]).unsqueeze(2)
        return torch.cat([pose[:, :, :3] * (1 - fm) - fm, pose[:, :, 3:]], dim=2)
    return pose


def get_fg_mask(cfg, label):
    """models/input_process.py:52-61 (pose only)"""
    if label.dim() == 5:
        label = label[:, 0]
    mask = label[:, 2:3] if cfg.label_nc == 0 else -label[:, 0:1]
    mask = F.max_pool2d(mask, 15, padding=7, stride=1)
    return (mask > -1).float()


def get_face_region(cfg, pose, crop_smaller=0):
    """models/face_refiner.py:56-87: face bounding box (ys, ye, xs, xe) of one label map [1, C, H, W]."""
    _, _, h, w = pose.shape
    use_openpose = not cfg.basic_point_only and not cfg.remove_face_labels
    if use_openpose:
        face = ((pose[:, -3] > 0) & (pose[:, -2] > 0) & (pose[:, -1] > 0)).nonzero()
    else:
        face = (pose[:, 2] > 0.9).nonzero()
    if face.size(0):
        y, x = face[:, 1], face[:, 2]
        ys, ye, xs, xe = y.min().item(), y.max().item(), x.min().item(), x.max().item()
        if use_openpose:
            xc, yc = (xs + xe) // 2, (ys * 3 + ye * 2) // 5
            ylen = int((xe - xs) * 2.5)
        else:
            xc, yc = (xs + xe) // 2, (ys + ye) // 2
            ylen = int((ye - ys) * 1.25)
        ylen = xlen = min(w, max(32, ylen))
        yc = max(ylen // 2, min(h - 1 - ylen // 2, yc))
        xc = max(xlen // 2, min(w - 1 - xlen // 2, xc))
    else:
        yc = h // 4
        xc = w // 2
        ylen = xlen = h // 32 * 8
    ys, ye, xs, xe = yc - ylen // 2, yc + ylen // 2, xc - xlen // 2, xc + xlen // 2
    if crop_smaller != 0:
        ys += crop_smaller; xs += crop_smaller
        ye -= crop_smaller; xe -= crop_smaller
    return ys, ye, xs, xe


def crop_face_region(cfg, image, input_label, crop_smaller=0):
    """models/face_refiner.py:32-39; face_size = (fineSize / aspect_ratio) // 4 (:21)."""
    size = int(cfg.fineSize / cfg.aspect_ratio) // 4
    outs = []
    for i in range(input_label.size(0)):
        ys, ye, xs, xe = get_face_region(cfg, input_label[i:i + 1], crop_smaller)
        outs.append(F.interpolate(image[i:i + 1, -3:, ys:ye, xs:xe], size=(size, size)))
    return torch.cat(outs)


def replace_face_region(cfg, fake_image, fake_face, input_label, fake_face_coarse=None, crop_smaller=0):
    """models/face_refiner.py:42-54"""
    fake_image = fake_image.clone()
    for i in range(input_label.shape[0]):
        ys, ye, xs, xe = get_face_region(cfg, input_label[i:i + 1], crop_smaller)
        face_i = fake_face[i:i + 1] + (fake_face_coarse[i:i + 1] if fake_face_coarse is not None else 0)
        face_i = F.interpolate(face_i, size=(ye - ys, xe - xs), mode='bilinear')
        fake_image[i:i + 1, :, ys:ye, xs:xe] = torch.clamp(face_i, -1, 1)
    return fake_image


def _hinge_d(pred, real):
    """models/networks/loss.py:69-79"""
    return -torch.mean(torch.min((pred if real else -pred) - 1, pred * 0))


def _masked_l1(inp, target, mask):
    """models/networks/loss.py:130-138"""
    mask = mask.expand_as(inp)
    return F.l1_loss(inp * mask, target * mask)


def face_cfg(cfg):
    """base_model.py:175-181: the options of the --refine_face generator"""
    import copy
    cf = copy.copy(cfg)
    cf.n_downsample_G = cfg.n_downsample_G - 1
    if cf.n_adaptive_layers > 0:
        cf.n_adaptive_layers = cfg.n_adaptive_layers - 1
    cf.label_nc, cf.input_nc = 0, cfg.output_nc
    cf.fineSize, cf.aspect_ratio = int(cfg.fineSize / cfg.aspect_ratio) // 4, 1
    cf.warp_ref = False                                  # generator.py:148 (for_face)
    cf.n_shot = 1
    return cf


def generator_forward_face(sd, cfg, label, label_refs, img_refs, img_coarse):
    """FewShotGenerator.forward_face (generator.py:232-242); cfg = face_cfg(...)"""
    ch, n = cfg.ch, cfg.n_downsample_G
    b = img_refs.shape[0]
    img_ref = img_refs.reshape(b * img_refs.shape[1], -1, *img_refs.shape[-2:])
    label_ref = label_refs.reshape(b * label_refs.shape[1], -1, *label_refs.shape[-2:])
    _, enc, _, _ = reference_encoding(sd, cfg, img_ref, label_ref, label)
    embed_w, norm_w = [], []
    for i in range(cfg.n_adaptive_layers):
        e, nw = spade_weights(sd, cfg, enc[min(len(enc) - 1, i + 1)], i)
        embed_w.append(e); norm_w.append(nw)
    enc_label = label_embedder(sd, 'label_embedding.', cfg, label, embed_w, False, cfg.n_adaptive_layers)
    x = spade_conv2d(sd, 'ref_img_first.', img_coarse)                     # compute_kld with img_coarse, generator.py:321-325
    for i in range(n):
        x = spade_conv2d(sd, 'ref_img_down_%d.' % i, x, 2)
    for i in range(n, -1, -1):
        nw = norm_w[i] if (cfg.adaptive_spade and i < cfg.n_adaptive_layers) else None
        x = spade_resblock(sd, 'up_%d.' % i, x, enc_label[i], nw, ch[i + 1], ch[i])
        if i != 0:
            x = F.interpolate(x, scale_factor=2)
    return torch.tanh(_plain_conv(sd, 'conv_img.', actvn(x)))


def refine_face_region(sdGf, cfg, label_valid, fake_image, label, ref_label_valid, ref_image, ref_label):
    """FaceRefineModel.refine_face_region (face_refiner.py:24-30)"""
    label_face = crop_face_region(cfg, label_valid, label, crop_smaller=4)
    coarse = crop_face_region(cfg, fake_image, label, crop_smaller=4)
    ref_label_face = crop_face_region(cfg, ref_label_valid, ref_label, crop_smaller=4)
    ref_image_face = crop_face_region(cfg, ref_image, ref_label, crop_smaller=4)
    fake_face = generator_forward_face(sdGf, face_cfg(cfg), label_face, ref_label_face.unsqueeze(1),
                                       ref_image_face.unsqueeze(1), coarse.detach())
    return replace_face_region(cfg, fake_image, fake_face, label, coarse.detach(), crop_smaller=4)


def step_generate(sdG, cfg, tgt_label, tgt_image, ref_labels, ref_images, prevs=None, warp_prev=False, sdGf=None):
    """Vid2VidModel.generate_images (vid2vid_model.py:130-158), n_frames_per_gpu = 1.  prevs = [prev_label,
    prev_real, prev_fake] ([B, n_frames_G-1, C, H, W]) or None for the first frame; warp_prev mirrors
    netG.warp_prev (set by init_temporal_network)."""
    ref_labels_valid = use_valid_labels(cfg, ref_labels)
    tgt_label_t = tgt_label[:, 0]
    tgt_label_valid = use_valid_labels(cfg, tgt_label_t)
    prev_t = (None, None)
    if prevs is not None and prevs[0] is not None:
        b, _, _, h, w = tgt_label.shape
        prev_t = (prevs[0].reshape(b, -1, h, w), prevs[2].reshape(b, -1, h, w))       # vid2vid_model.py:163-166
    fake, flow, mask, raw, warp, atn_vis, ref_idx = generator_forward(sdG, cfg, tgt_label_valid, ref_labels_valid, ref_images,
                                                                      prev_t, warp_prev, return_ref=True)
    ref_label_valid, ref_label_t, ref_image_t = pick_ref(ref_labels_valid, ref_idx), pick_ref(ref_labels, ref_idx), \
        pick_ref(ref_images, ref_idx)                                                    # vid2vid_model.py:144
    if sdGf is not None:                                                                 # vid2vid_model.py:146-148
        fake = refine_face_region(sdGf, cfg, tgt_label_valid, fake, tgt_label_t, ref_label_valid, ref_image_t, ref_label_t)
    if cfg.pose:
        fg, ref_fg = get_fg_mask(cfg, tgt_label_t), get_fg_mask(cfg, ref_label_t)
        union = ((fg > 0) | (ref_fg > 0)).float()
    else:
        fg = ref_fg = None
        union = 1
    if raw is not None:
        raw = raw * union
    new_prevs = []                                                                   # concat_prev, vid2vid_model.py:169-176
    for old, now in zip(prevs if prevs is not None else [None] * 3, (tgt_label_valid, tgt_image[:, 0], fake)):
        if old is None:
            new_prevs.append(now.detach().unsqueeze(1))
        else:
            new_prevs.append(torch.cat([old[:, 1:], now.detach().unsqueeze(1)], dim=1))
    return dict(fake=fake, raw=raw, warp=warp, flow=flow, mask=mask, fg=fg, ref_fg=ref_fg, union=union,
                ref_label=ref_label_valid, ref_image=ref_image_t, prevs=new_prevs)


def _gan_terms(out, half, cfg, for_discriminator):
    """loss_collector.py:56-67 on the multi-scale predictions of a [fake ; real] batch."""
    pf = [[t[:half] for t in s] for s in out]
    pr = [[t[half:] for t in s] for s in out]
    if for_discriminator:
        return [sum(_hinge_d(s[-1], True) for s in pr) / len(pr), sum(_hinge_d(s[-1], False) for s in pf) / len(pf)]
    feat = 0
    for sf, sr in zip(pf, pr):
        for a, b in zip(sf[:-1], sr[:-1]):
            feat = feat + F.l1_loss(a, b.detach()) / len(pf)
    return [sum(_hinge_d(s[-1], True) for s in pf) / len(pf), feat * cfg.lambda_feat]


def _discriminate_face(sdDf, cfg, fake, tgt_label4, real, ref_label_cat, ref_image, for_discriminator, vgg_weights):
    """LossCollector.discriminate_face (loss_collector.py:69-85).  ref_label_cat is compute_GAN_losses' local ref_label
    at that point: valid labels with the foreground mask appended (:104-107)."""
    real_region = crop_face_region(cfg, real, tgt_label4)
    fake_region = crop_face_region(cfg, fake, tgt_label4)
    ref_region = crop_face_region(cfg, ref_image, ref_label_cat)
    x = torch.cat([fake_region, real_region], dim=0)
    x = torch.cat([ref_region.repeat(2, 1, 1, 1), x], dim=1)
    out = multiscale_discriminator(sdDf, cfg, x, num_D=1)
    losses = [l * cfg.lambda_face for l in _gan_terms(out, x.shape[0] // 2, cfg, for_discriminator)]
    if for_discriminator:
        return losses
    gf_gan, gf_feat = losses
    gf_feat = gf_feat + F.l1_loss(fake_region, real_region) * cfg.lambda_feat
    gf_feat = gf_feat + vgg_loss(vgg_weights, fake_region, real_region) * cfg.lambda_vgg
    return [gf_gan, gf_feat]


def _temporal_losses(sdDT, cfg, real_all, fake_all, for_discriminator):
    """compute_GAN_losses(for_temporal=True) (loss_collector.py:87-90,109-113): tD = min(n_frames_D, n_frames_G) frames
    stacked on the channel axis (base_model.py:120-139, isTrain), temporal discriminator without label / reference."""
    tD = min(cfg.n_frames_D, cfg.n_frames_G)

    def stack(x):
        bs, t, ch, h, w = x.shape
        if t > tD:
            if t % tD != 0:
                x = x[:, -(t // tD) * tD:]
            return x.contiguous().view(-1, ch * tD, h, w)
        return x.contiguous().view(bs, ch * t, h, w)
    x = torch.cat([stack(fake_all), stack(real_all)], dim=0)
    out = multiscale_discriminator(sdDT, cfg, x, num_D=1)
    losses = _gan_terms(out, x.shape[0] // 2, cfg, for_discriminator)
    if not for_discriminator:
        losses = [l * cfg.lambda_temp for l in losses]
    return losses


def _discriminate(sdD, cfg, tgt_label4, fake, real, ref_label, ref_image, for_discriminator, sdDf=None, vgg_weights=None):
    """LossCollector.compute_GAN_losses + discriminate (loss_collector.py:47-68,87-120): per-frame D, plus the face
    discriminator's two terms when cfg.add_face_D (4 entries then)."""
    inp = use_valid_labels(cfg, tgt_label4)
    rl = ref_label
    if cfg.pose:
        inp = torch.cat([inp, get_fg_mask(cfg, tgt_label4)], dim=1)
        rl = torch.cat([ref_label, get_fg_mask(cfg, ref_label)], dim=1)
    ref_concat = torch.cat([rl, ref_image], dim=1)
    x = torch.cat([fake, real], dim=0)
    x = torch.cat([inp.repeat(2, 1, 1, 1), x], dim=1)
    if cfg.netD_subarch == 'adaptive':          # loss_collector.py:52-58: the reference tensor is netD's second input
        out = multiscale_discriminator(sdD, cfg, x, ref=ref_concat.repeat(2, 1, 1, 1))
    else:
        x = torch.cat([ref_concat.repeat(2, 1, 1, 1), x], dim=1)
        out = multiscale_discriminator(sdD, cfg, x)
    losses = _gan_terms(out, x.shape[0] // 2, cfg, for_discriminator)
    if cfg.add_face_D:
        losses = losses + _discriminate_face(sdDf, cfg, fake, tgt_label4, real, rl, ref_image, for_discriminator, vgg_weights)
    return losses


VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512]


def vgg_activations(weights, x):
    """VGG_Activations.forward (models/networks/vgg.py:45-59) on torchvision's vgg19.features up to index 29;
    weights: [(w, b)] of the convolutions in order; returns the activations at indices 1, 6, 11, 20, 29."""
    res, wi, idx = [], 0, 0
    for v in VGG_CFG:
        if v == 'M':
            x = F.max_pool2d(x, 2, 2); idx += 1
        else:
            w, b = weights[wi]; wi += 1
            x = F.relu(_conv2d(x, w.to(x.dtype), b.to(x.dtype), 1, 1)); idx += 2
            if idx - 1 in (1, 6, 11, 20, 29):
                res.append(x)
    return res


def vgg_loss(weights, x, y):
    """VGGLoss.forward / compute_loss (models/networks/loss.py:107-128)"""
    xf, yf = vgg_activations(weights, x), vgg_activations(weights, y)
    loss = 0
    for wgt, a, b in zip([1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0], xf, yf):
        loss = loss + wgt * F.l1_loss(a, b.detach())
    return loss


def finetune(sdG, sdD, cfg, ref_labels, ref_images, iterations=100, vgg_weights=None, sdDf=None):
    """Vid2VidModel.finetune (vid2vid_model.py:207-237) on leaf state dicts (updated in place): layers of the generator
    whose names contain 'fc' / 'conv_img' / 'up' (base_model.py:149-165) and the discriminator(s) are trained with Adam
    (base_model.py:39-48) on randomly rolled / flipped reference images (util/util.py:157-168; Python `random` stream).
    The model is in eval mode (test.py:27) and opt.isTrain is False, so only the GAN (+ VGG / face) terms are non-zero."""
    import random
    names = ('fc', 'conv_img', 'up')

    def trainable(k, v):
        return v.is_floating_point() and v.requires_grad

    if cfg.no_TTUR:
        b1, b2, g_lr, d_lr = cfg.beta1, 0.999, cfg.lr, cfg.lr
    else:
        b1, b2, g_lr, d_lr = 0.0, cfg.beta2, cfg.lr / 2, cfg.lr * 2
    opt_g = torch.optim.Adam([v for k, v in sdG.items() if trainable(k, v) and any(t in k for t in names)], lr=g_lr,
                             betas=(float(b1), float(b2)))
    d_leaves = [v for k, v in sdD.items() if trainable(k, v)] + [v for k, v in (sdDf or {}).items() if trainable(k, v)]
    opt_d = torch.optim.Adam(d_leaves, lr=d_lr, betas=(float(b1), float(b2)))
    all_leaves = [v for v in list(sdG.values()) + list(sdD.values()) + list((sdDf or {}).values())
                  if v.is_floating_point() and v.requires_grad]

    def roll(t, ny, nx, flip):
        t = torch.cat([t[:, :, -ny:], t[:, :, :-ny]], dim=2)
        t = torch.cat([t[:, :, :, -nx:], t[:, :, :, :-nx]], dim=3)
        return torch.flip(t, dims=[3]) if flip else t
    history = []
    with eval_mode():
        for it in range(iterations):
            idx = random.randrange(ref_labels.size(1))
            h, w = ref_labels.shape[-2:]
            ny = random.choice([random.randrange(h // 16), h - random.randrange(h // 16)])
            nx = random.choice([random.randrange(w // 16), w - random.randrange(w // 16)])
            flip = random.random() > 0.5
            tl = roll(ref_labels[:, idx], ny, nx, flip).unsqueeze(1)
            ti = roll(ref_images[:, idx], ny, nx, flip).unsqueeze(1)
            g_losses, _ = g_step_losses(sdG, sdD, cfg, tl, ti, ref_labels, ref_images, vgg_weights=vgg_weights, sdDf=sdDf)
            for v in all_leaves:
                v.grad = None
            sum(l.mean() for l in g_losses.values()).backward()
            opt_g.step()
            d_losses = d_step_losses(sdG, sdD, cfg, tl, ti, ref_labels, ref_images, sdDf=sdDf)
            for v in all_leaves:
                v.grad = None
            sum(l.mean() for l in d_losses).backward()
            opt_d.step()
            history.append([float(l) for l in g_losses.values()] + [float(l) for l in d_losses])
    for v in all_leaves:
        v.grad = None
    return history


def inference_frames(sdG, cfg, frames, ref_labels, ref_images, warp_prev=False, n_frames_G=2):
    """Vid2VidModel.inference (vid2vid_model.py:179-205) over a sequence: frames = list of tgt_label [B, 1, C, H, W];
    returns the list of synthesized images.  (With one reference the reference keeps the frame-0 weights for later
    frames, generator.py:403-416 - the same values as recomputing them in eval mode.)"""
    outs, prevs = [], None
    ref_labels = encode_label(cfg, ref_labels)
    ref_labels_valid = use_valid_labels(cfg, ref_labels)
    with eval_mode(), torch.no_grad():
        for tgt_label in frames:
            tgt_label = encode_label(cfg, tgt_label)
            lab = use_valid_labels(cfg, tgt_label[:, -1])
            b, _, h, w = lab.shape
            prev_t = (None, None) if prevs is None else tuple(p.reshape(b, -1, h, w) for p in prevs)
            fake, flow, mask, raw, warp = generator_forward(sdG, cfg, lab, ref_labels_valid, ref_images, prev_t, warp_prev)
            new = []
            for old, now in zip(prevs if prevs is not None else [None, None], (lab, fake)):
                if old is None:
                    new.append(now.unsqueeze(1).repeat(1, n_frames_G - 1, 1, 1, 1))
                else:
                    new.append(torch.cat([old[:, 1:], now.unsqueeze(1)], dim=1))
            prevs = new
            outs.append(fake)
    return outs


def encode_label(cfg, label_map):
    """models/input_process.py:25-45: label_nc != 0 -> one-hot scatter of the integer class map (street); the pose / face
    label maps (label_nc == 0) pass through."""
    if cfg.label_nc == 0:
        return label_map
    size = label_map.shape
    flat = label_map.reshape(-1, *size[-3:])
    one_hot = torch.zeros(flat.shape[0], cfg.label_nc, size[-2], size[-1], dtype=label_map.dtype)
    one_hot.scatter_(1, flat.long(), 1.0)
    return one_hot.view(*size[:-3], cfg.label_nc, size[-2], size[-1])


def d_step_losses(sdG, sdD, cfg, tgt_label, tgt_image, ref_labels, ref_images, prevs=None, warp_prev=False, sdDf=None,
                  sdDT=None, sdGf=None):
    """Vid2VidModel.forward_discriminator (vid2vid_model.py:106-128) -> [D_real, D_fake(, Df_real, Df_fake)]
    (lambda_temp = 0)."""
    tgt_label, ref_labels = encode_label(cfg, tgt_label), encode_label(cfg, ref_labels)
    with torch.no_grad():
        g = step_generate(sdG, cfg, tgt_label, tgt_image, ref_labels, ref_images, prevs, warp_prev, sdGf)
    lab4 = tgt_label.reshape(-1, *tgt_label.shape[-3:])
    real = tgt_image[:, 0]
    total = None
    for fake, r in ((g['fake'], real), (g['raw'], real * g['union'])):
        if fake is None:
            continue
        l = _discriminate(sdD, cfg, lab4, fake, r, g['ref_label'], g['ref_image'], True, sdDf)
        total = l if total is None else [a + b for a, b in zip(total, l)]
    if cfg.lambda_temp > 0 and prevs is not None and prevs[0] is not None and sdDT is not None:   # vid2vid_model.py:115-119
        if len(total) == 2:
            total = total + [torch.zeros(1), torch.zeros(1)]                                      # Df slots
        total = total + _temporal_losses(sdDT, cfg, torch.cat([prevs[1], tgt_image], dim=1),
                                         torch.cat([prevs[2], g['fake'].unsqueeze(1)], dim=1), True)
    return total


def g_step_losses(sdG, sdD, cfg, tgt_label, tgt_image, ref_labels, ref_images, prevs=None, warp_prev=False,
                  vgg_weights=None, sdDf=None, sdDT=None, flow_gt=(None, None), conf_gt=(None, None), sdGf=None):
    """Vid2VidModel.forward_generator (vid2vid_model.py:62-104) with --no_vgg_loss --no_flow_gt, first frame.

    Returns dict(G_GAN, G_GAN_Feat, F_Warp, F_Mask) plus the generated tensors.
    """
    tgt_label, ref_labels = encode_label(cfg, tgt_label), encode_label(cfg, ref_labels)
    g = step_generate(sdG, cfg, tgt_label, tgt_image, ref_labels, ref_images, prevs, warp_prev, sdGf)
    lab4 = tgt_label.reshape(-1, *tgt_label.shape[-3:])
    real = tgt_image[:, 0]
    gan = None
    for fake, r in ((g['fake'], real), (g['raw'], real * g['union'])):
        if fake is None:
            continue
        l = _discriminate(sdD, cfg, lab4, fake, r, g['ref_label'], g['ref_image'], False, sdDf, vgg_weights)
        gan = l if gan is None else [a + b for a, b in zip(gan, l)]
    # flow losses (loss_collector.py:132-162)
    f_warp = torch.zeros(1)
    f_flow = torch.zeros(1)
    for k, (f, wimg) in enumerate(zip(g['flow'], g['warp'])):
        if f is not None and cfg.isTrain:                                          # loss_collector.py:158
            f_warp = f_warp + F.l1_loss(wimg, real)
            if flow_gt[k] is not None and cfg.n_shot == 1:                     # compute_flow_loss, loss_collector.py:154-161
                gt = flow_gt[k].reshape(-1, *flow_gt[k].shape[-3:])
                conf = conf_gt[k].reshape(-1, *conf_gt[k].shape[-3:])
                f_flow = f_flow + _masked_l1(f, gt, conf * g['fg'] if g['fg'] is not None else conf)
    body_diff = None
    if cfg.isTrain and cfg.pose and g['flow'][0] is not None:                       # loss_collector.py:141
        body = part_mask(tgt_label[:, :, 2])
        ref_body = part_mask(g['ref_label'][:, 2].unsqueeze(1)).expand_as(body)
        body, ref_body = body.reshape(-1, *body.shape[-3:]), ref_body.reshape(-1, *ref_body.shape[-3:])
        ref_body_warp = resample(ref_body, g['flow'][0])
        f_warp = f_warp + F.l1_loss(ref_body_warp, body)
        fg2, ref_fg2 = get_fg_mask(cfg, tgt_label), get_fg_mask(cfg, g['ref_label'])
        f_warp = f_warp + F.l1_loss(resample(ref_fg2, g['flow'][0]), fg2)
        body_diff = torch.sum(abs(ref_body_warp - body), dim=1, keepdim=True)
    # mask losses (loss_collector.py:164-204)
    f_mask = torch.zeros(1)
    for m, wimg in zip(g['mask'], g['warp']):
        if m is None or not cfg.isTrain:                                            # loss_collector.py:192
            continue
        conf = torch.clamp(1 - torch.sum(abs(wimg - real), dim=1, keepdim=True), 0, 1)
        f_mask = f_mask + _masked_l1(m, torch.zeros_like(m), conf) + _masked_l1(m, torch.ones_like(m), 1 - conf)
    if cfg.isTrain and cfg.pose and cfg.warp_ref:                                   # loss_collector.py:172
        m_ref = g['mask'][0]
        h, w = tgt_label.shape[-2:]
        fm = face_mask(tgt_label[:, :, 2]).view(-1, 1, h, w)
        fm = F.avg_pool2d(fm, 15, padding=7, stride=1)
        f_mask = f_mask + _masked_l1(m_ref, torch.zeros_like(m_ref), fm)
        if cfg.spade_combine:
            f_mask = f_mask + _masked_l1(g['fake'], g['warp'][0].detach(), fm)
        fg_diff = ((g['ref_fg'] - g['fg']) > 0).float()
        f_mask = f_mask + _masked_l1(m_ref, torch.ones_like(m_ref), fg_diff)
        f_mask = f_mask + _masked_l1(m_ref, torch.ones_like(m_ref), body_diff)
    out = dict(G_GAN=gan[0], G_GAN_Feat=gan[1], F_Warp=f_warp * cfg.lambda_flow, F_Mask=f_mask * cfg.lambda_mask)
    if flow_gt[0] is not None or flow_gt[1] is not None:
        out['F_Flow'] = f_flow * cfg.lambda_flow
    if cfg.add_face_D:
        out['Gf_GAN'], out['Gf_GAN_feat'] = gan[2], gan[3]
    if cfg.lambda_temp > 0 and prevs is not None and prevs[0] is not None and sdDT is not None:   # vid2vid_model.py:70-75
        out['GT_GAN'], out['GT_GAN_Feat'] = _temporal_losses(
            sdDT, cfg, torch.cat([prevs[1], tgt_image], dim=1), torch.cat([prevs[2], g['fake'].unsqueeze(1)], dim=1), False)
    if vgg_weights is not None:            # compute_VGG_losses, loss_collector.py:122-130
        v = vgg_loss(vgg_weights, g['fake'], real)
        if g['raw'] is not None:
            v = v + vgg_loss(vgg_weights, g['raw'], real * g['union'])
        out['G_VGG'] = v * cfg.lambda_vgg
    return out, g


def iteration(sdG0, sdD0, cfg, data, dtype, vgg_weights=None, sdDf0=None, flow_gt=None, conf_gt=None, sdGf0=None, loss_scale=1.0):
    """One reference iteration (train.py:58-62) on the oracle: D step then G step; returns losses and gradients
    (the face discriminator's gradients, when present, as a 6th entry).  loss_scale: `amp.scale_loss` (loss_collector.py:221-224):
    the backward passes run on loss * scale and the gradients are divided by it - it only matters under a half arithmetic."""
    tl, ti, rl, ri = [t.to(dtype) for t in data]

    def leafify(sd0):
        sd = {}
        for k, v in sd0.items():
            t = v.clone().to(dtype).detach() if v.is_floating_point() else v.clone()
            if v.is_floating_point() and ('running' not in k) and not k.endswith(('_u', '_v')):
                t.requires_grad_(True)
            sd[k] = t
        return sd
    sdG, sdD = leafify(sdG0), leafify(sdD0)
    sdDf = leafify(sdDf0) if sdDf0 is not None else None
    sdGf = leafify(sdGf0) if sdGf0 is not None else None
    d_losses = d_step_losses(sdG, sdD, cfg, tl, ti, rl, ri, sdDf=sdDf, sdGf=sdGf)
    (sum(l.mean() for l in d_losses) * loss_scale).backward()
    gD = {k: v.grad.clone() / loss_scale for k, v in sdD.items() if v.is_floating_point() and v.grad is not None}
    gDf = {k: v.grad.clone() / loss_scale for k, v in (sdDf or {}).items() if v.is_floating_point() and v.grad is not None}
    # (the optimiser step is checked separately in check_adam; gradients are what the comparison needs)
    for v in list(sdG.values()) + list(sdD.values()) + list((sdDf or {}).values()):
        if v.is_floating_point() and v.grad is not None:
            v.grad = None
    fg = [None if f is None else f.to(dtype) for f in (flow_gt or [None, None])]
    cg = [None if f is None else f.to(dtype) for f in (conf_gt or [None, None])]
    g_losses, gen = g_step_losses(sdG, sdD, cfg, tl, ti, rl, ri, vgg_weights=vgg_weights, sdDf=sdDf, flow_gt=fg,
                                    conf_gt=cg, sdGf=sdGf)
    (sum(l.mean() for l in g_losses.values()) * loss_scale).backward()
    gG = {k: v.grad.clone() / loss_scale for k, v in sdG.items() if v.is_floating_point() and v.grad is not None}
    if sdGf is not None:                         # the face generator's gradients ride along under a 'netGf.' prefix
        gG.update({'netGf.' + k: v.grad.clone() / loss_scale for k, v in sdGf.items() if v.is_floating_point() and v.grad is not None})
    return d_losses, gD, g_losses, gG, gen, gDf
