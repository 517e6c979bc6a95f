"""Per-step orchestration of the G/D training hot path: the drop-in for the reference's
``models.vid2vid_model.Vid2VidModel`` (forward modes 'generator' / 'discriminator'), its loss collector
(models/loss_collector.py) and ``loss_backward``.

The call protocol is the reference's (train.py:55-62):

    d_losses = model(data_list_t, mode='discriminator');   loss_backward(opt, d_losses, optimizer_D, 1)
    g_losses, generated, prev = model(data_list_t, mode='generator');   loss_backward(opt, g_losses, optimizer_G, 0)

with ``data_list_t = [tgt_label, tgt_image, flow_gt[2], conf_gt[2], ref_label, ref_image, prev_label, prev_real,
prev_fake]`` (5-D ``[B, T, C, H, W]`` tensors).  Loss order and names follow loss_collector.py:42-45.
Networks run on the HIP kernels (networks.py); parameters and gradients live in flat buffers (flat.py) so the
optimiser is one fused Adam launch and the data-parallel exchange is a handful of large RCCL all-reduces.
"""
import contextlib
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import conv, networks, ops, streams
from .flat import FlatAdam

LOSS_NAMES_G = ['G_GAN', 'G_GAN_Feat', 'G_VGG', 'Gf_GAN', 'Gf_GAN_feat', 'GT_GAN', 'GT_GAN_Feat', 'F_Flow', 'F_Warp',
                'F_Mask']
LOSS_NAMES_D = ['D_real', 'D_fake', 'Df_real', 'Df_fake', 'DT_real', 'DT_fake']


# ------------------------------------------------------------------------------------------------ label helpers
# Pure functions of the iteration's LABEL tensors (face-label removal, foreground / face masks, their union): one iteration of
# train.py:58-62 evaluates each of them three to five times on the same data - the no-grad and the generator-mode pass of
# generate_images, the discriminator conditioning of both steps and of the real-image pass, the flow / mask losses.  Inside a
# Vid2VidModel.forward call they are memoised per (function, storage, shape, strides, version) of their inputs; the table is
# emptied at the top of every `mode='discriminator'` call, i.e. it never outlives the iteration's data (a memo entry holds its
# input tensors, so an address cannot be reused under it).  Same kernels on the same data: bit-identical results; ~40 small
# launches per iteration less (round 6).  FSV_LABEL_MEMO=0 switches it off (A/B).
_MEMO = None


def _memo_key(t):
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype, t._version)


def _memoised(tag, tensors, make):
    if _MEMO is None or any(t.requires_grad for t in tensors):
        return make()
    key = (tag,) + tuple(_memo_key(t) for t in tensors)
    hit = _MEMO.get(key)
    cross = streams.CROSS and tensors[0].is_cuda          # two passes of the iteration are being issued on different streams
    if hit is None:
        val = make()
        ev = stream = None
        if cross:
            stream = torch.cuda.current_stream(tensors[0].device)
            ev = torch.cuda.Event()
            ev.record(stream)
        hit = _MEMO[key] = (tensors, val, ev, stream)
    elif hit[2] is not None and cross:
        cur = torch.cuda.current_stream(tensors[0].device)
        if cur != hit[3]:                                   # made on the other pass's stream: order this stream behind it
            cur.wait_event(hit[2])
            streams._record(hit[1], cur)
    return hit[1]


class _memo_scope:
    def __init__(self, table):
        self.table = table if os.environ.get('FSV_LABEL_MEMO', '1') == '1' else None

    def __enter__(self):
        global _MEMO
        self.prev, _MEMO = _MEMO, self.table

    def __exit__(self, *exc):
        global _MEMO
        _MEMO = self.prev
        return False


def face_mask_of(pose_ch):
    """models/input_process.py:80-94: parts 23 / 24 are the face.  pose_ch: [B, H, W] or [B, T, H, W]."""
    if pose_ch.dim() == 3:
        pose_ch = pose_ch.unsqueeze(1)
    return _memoised('face', (pose_ch,), lambda: ops.part_masks(pose_ch, 8, 1).squeeze(2))


PART_GROUPS = [[0], [1, 2], [3, 4], [5, 6], [7, 9, 8, 10], [11, 13, 12, 14], [15, 17, 16, 18], [19, 21, 20, 22],
               [23, 24]]            # the table csrc/losses.hip:fsv_part_masks_kernel carries


def part_masks(pose_ch):
    """models/input_process.py:64-78: 9 body-part group masks.  pose_ch [B, T, H, W] -> [B, T, 9, H, W]."""
    return _memoised('parts', (pose_ch,), lambda: ops.part_masks(pose_ch, 0, 9))


def valid_labels(opt, pose):
    """models/input_process.py:97-113."""
    if 'pose' not in opt.dataset_mode or pose is None:
        return pose
    cdim = pose.dim() - 3
    if opt.pose_type == 'open':
        return pose.narrow(cdim, 3, pose.shape[cdim] - 3)
    if opt.remove_face_labels:
        def make():
            fm = face_mask_of(pose.select(cdim, 2))
            if pose.dim() == 5:
                fm = fm.unsqueeze(2)
            head = pose.narrow(cdim, 0, 3) * (1 - fm) - fm
            return torch.cat([head, pose.narrow(cdim, 3, pose.shape[cdim] - 3)], dim=cdim)
        return _memoised('valid', (pose,), make)
    return pose


def fg_mask_of(opt, label, has_fg):
    """models/input_process.py:52-61 (15x15 dilation of the body channel)."""
    if not has_fg:
        return None
    if label.dim() == 5:
        label = label[:, 0]
    return _memoised('fg%d' % int(opt.label_nc == 0), (label,),
                     lambda: ops.pool15(label[:, 2:3] if opt.label_nc == 0 else -label[:, 0:1], 'max_gt', -1.0))


def union_fg(fg, ref_fg, has_fg):
    return _memoised('union', (fg, ref_fg), lambda: ((fg > 0) | (ref_fg > 0)).float()) if has_fg else 1


def encode_label(opt, label_map):
    """models/input_process.py:25-45."""
    if opt.label_nc == 0:
        return label_map
    size = label_map.shape
    flat = label_map.reshape(-1, *size[-3:])
    one_hot = torch.zeros(flat.shape[0], opt.label_nc, size[-2], size[-1], dtype=torch.float32, device=label_map.device)
    one_hot.scatter_(1, flat.long(), 1.0)
    return one_hot.view(*size[:-3], opt.label_nc, size[-2], size[-1])


# ------------------------------------------------------------------------------------------------ losses
def l1(a, b):
    """nn.L1Loss (mean absolute difference) as one fused reduction (csrc/losses.hip)."""
    return ops.l1_loss(a, b)


def masked_l1(inp, target, mask):
    """models/networks/loss.py:130-138: L1(input * mask, target * mask), mask [N, 1, H, W] broadcast over channels."""
    return ops.l1_loss(inp, target, mask)


def gan_loss(preds, real):
    """GANLoss.__call__ with the hinge objective on a list-of-lists prediction (loss.py:69-79, 92-104): last
    feature of every scale; the reference calls it with for_discriminator=True in both steps."""
    return ops.weighted_sum([ops.hinge_loss(p[-1], real) for p in preds], [1.0 / len(preds)] * len(preds))


_ZEROS = {}
_SPLIT_D_PASS = os.environ.get('FSV_SPLIT_D_PASS', '1') == '1'       # in-box A/B switch (profiles/r02_notes.md section 10)


class LossCollector:
    """models/loss_collector.py restated on top of the HIP networks."""

    def __init__(self, opt):
        self.opt = opt
        self.pose = 'pose' in opt.dataset_mode
        self.has_fg = self.pose
        self.warp_ref = opt.warp_ref
        self.add_face_D = opt.add_face_D
        self.concat_ref_for_D = (opt.isTrain or getattr(opt, 'finetune', False)) and opt.netD_subarch == 'n_layers'
        self.concat_fg_mask_for_D = self.has_fg
        self.loss_names_G, self.loss_names_D = LOSS_NAMES_G, LOSS_NAMES_D
        self.loss_names = LOSS_NAMES_G + LOSS_NAMES_D
        self.tD = 1
        self.face_size = int(opt.fineSize / opt.aspect_ratio) // 4                  # face_refiner.py:21
        self.use_openpose = not opt.basic_point_only and not opt.remove_face_labels     # face_refiner.py:62
        if opt.add_face_D and opt.no_vgg_loss:
            # the reference adds criterionVGG(fake_region, real_region) unconditionally (loss_collector.py:83) and
            # fails with an AttributeError in this combination
            raise ValueError("--add_face_D needs the VGG loss (drop --no_vgg_loss), as in the reference")
        self.vgg = None
        if not opt.no_vgg_loss:
            from .vgg import VGGLoss
            self.vgg = VGGLoss()

    def to(self, device):
        if self.vgg is not None:
            self.vgg = self.vgg.to(device)
        return self

    def vgg_losses(self, fake, raw, real, fg_union):
        """loss_collector.py:122-130"""
        if self.vgg is None:
            return self.zero(fake)
        terms = [self.vgg(fake, real)]
        if raw is not None:
            terms.append(self.vgg(raw, real * fg_union))
        return ops.weighted_sum(terms, [self.opt.lambda_vgg] * len(terms))

    @staticmethod
    def zero(ref):
        """the constant 0 of absent loss terms (one shared read-only tensor per device: no fill launch per use)"""
        z = _ZEROS.get(ref.device)
        if z is None:
            z = _ZEROS[ref.device] = streams.shared(lambda: torch.zeros(1, dtype=torch.float32, device=ref.device))
        return z

    @classmethod
    def outward(cls, losses):
        """the loss list as it leaves the module: [1, 1] views, except that entries which ARE the shared zero constant are
        handed out as copies - a caller that reduces or scales its losses in place (the reference's dist_all_reduce_tensor
        does all_reduce + div_ on the tensor it is given, util/distributed.py:66-72) must not corrupt the constant that later
        iterations and captured graphs read"""
        shared = set(z.data_ptr() for z in _ZEROS.values())
        n_shared = sum(1 for l in losses if l.data_ptr() in shared)
        if not n_shared:
            return [l.view(1, 1) for l in losses]
        # (one fill for all of them: six one-element copies cost 9 us each inside a replayed graph - round 6)
        fresh = torch.zeros(n_shared, dtype=losses[0].dtype, device=losses[0].device)
        out, k = [], 0
        for l in losses:
            if l.data_ptr() in shared:
                out.append(fresh[k:k + 1].view(1, 1))
                k += 1
            else:
                out.append(l.view(1, 1))
        return out

    def discriminate(self, netD, label, fake, real, ref, for_discriminator, real_out=None):
        """loss_collector.py:47-68: D sees [ref | label | image] with fake and real stacked on the batch axis.  real_out:
        (outputs, sigmas) of the discriminator for the real images from real_pass() - then only the generated images go through
        it here, on the same sigmas."""
        # --netD_subarch adaptive: the reference tensor is the discriminator's second input instead of extra channels
        side = ref if (ref is not None and not self.concat_ref_for_D) else None
        if real_out is not None:
            x = ops.pack_d_single(ref if self.concat_ref_for_D else None, label, fake, for_conv=_single_scale(netD))
            pred_fake, pred_real = netD(x, side, sn=real_out[1]), real_out[0]
        else:
            x = ops.pack_d_input(ref if self.concat_ref_for_D else None, label, fake, real, for_conv=_single_scale(netD))
            out = netD(x, side.repeat(2, 1, 1, 1) if side is not None else None)
            half = x.shape[0] // 2
            pred_fake = [[t[:half] for t in scale] for scale in out]
            pred_real = [[t[half:] for t in scale] for scale in out]
        if for_discriminator:
            return [gan_loss(pred_real, True), gan_loss(pred_fake, False)]
        terms = []
        if not self.opt.no_ganFeat_loss:
            for sf, sr in zip(pred_fake, pred_real):
                for a, b in zip(sf[:-1], sr[:-1]):
                    terms.append(l1(a, b.detach()))
        feat = ops.weighted_sum(terms, [self.opt.lambda_feat / len(pred_fake)] * len(terms)) if terms else self.zero(fake)
        return [gan_loss(pred_fake, True), feat]

    def crop_face_region(self, image, label):
        """face_refiner.py:32-39: device-side boxes + one crop/resize launch (csrc/face.hip)."""
        boxes = ops.face_boxes(label, self.use_openpose)
        if isinstance(image, (list, tuple)):
            return [ops.crop_face(im, boxes, self.face_size) for im in image]
        return ops.crop_face(image, boxes, self.face_size)

    def discriminate_face(self, netDf, fake, tgt_label, real, ref_label, ref_image, for_discriminator):
        """loss_collector.py:69-85.  ref_label is what compute_GAN_losses holds at that point: the valid-label version
        with the foreground mask appended (so the OpenPose branch of get_face_region reads shifted channels and the
        DensePose branch sees the face already blanked by --remove_face_labels) - reproduced as is."""
        if not self.add_face_D:
            return [self.zero(fake), self.zero(fake)]
        real_region, fake_region = self.crop_face_region([real, fake], tgt_label)
        ref_region = self.crop_face_region(ref_image, ref_label)
        losses = self.discriminate(netDf, ref_region, fake_region, real_region, None, for_discriminator)
        losses = [l * self.opt.lambda_face for l in losses]
        if for_discriminator:
            return losses
        gf_gan, gf_feat = losses
        gf_feat = gf_feat + l1(fake_region, real_region) * self.opt.lambda_feat
        gf_feat = gf_feat + self.vgg(fake_region, real_region) * self.opt.lambda_vgg
        return [gf_gan, gf_feat]

    def temporal_losses(self, netDT, real_all, fake_all, for_discriminator):
        """loss_collector.py:87-90,109-113 with for_temporal=True: the temporal discriminator sees tD consecutive frames
        stacked on the channel axis ([B, t, 3, H, W] -> reshape of base_model.py:120-139), no label / reference input."""
        if self.tD < 2:
            return [self.zero(real_all), self.zero(real_all)]

        def stack(x):
            bs, t, ch, h, w = x.shape
            nd = self.tD
            if t > nd:
                if t % nd != 0:
                    x = x[:, -(t // nd) * nd:]
                return x.contiguous().view(-1, ch * nd, h, w)
            return x.contiguous().view(bs, ch * t, h, w)
        losses = self.discriminate(netDT, None, stack(fake_all), stack(real_all), None, for_discriminator)
        if not for_discriminator:
            losses = [l * self.opt.lambda_temp for l in losses]
        return losses

    def d_conditioning(self, tgt_label, ref_label, ref_image):
        """what the per-frame discriminator sees next to the image (loss_collector.py:92-104): the valid target labels
        (+ foreground mask) and [reference labels (+ mask) | reference image]"""
        opt = self.opt
        lab = tgt_label.reshape(-1, *tgt_label.shape[-3:])

        def make():
            inp = valid_labels(opt, lab)
            rl = ref_label
            if self.concat_fg_mask_for_D:
                inp = torch.cat([inp, fg_mask_of(opt, lab, True)], dim=1)
                rl = torch.cat([ref_label, fg_mask_of(opt, ref_label, True)], dim=1)
            return inp, rl, torch.cat([rl, ref_image], dim=1)
        # (a pure function of the iteration's labels and reference image: the discriminator step, the generator step and the
        # real-image pass all ask for it - built once per iteration, see _memoised)
        inp, rl, ref_concat = _memoised('dcond%d' % int(self.concat_fg_mask_for_D), (lab, ref_label, ref_image), make)
        return lab, inp, rl, ref_concat

    def real_pass(self, netD, tgt_label, reals, ref_label, ref_image, sigmas):
        """The G step's discriminator pass over the REAL images, without autograd (only their features are needed, as
        feature-matching targets): it depends on the data alone, so Vid2VidModel.forward_generator issues it next to the
        generator's forward pass, and the backward pass of the step no longer drags a zero gradient through the real half
        of a stacked batch.  `sigmas`: one netD.begin_pass() per entry of `reals` (every stacked call of the reference is one
        power iteration); returns (outputs, sigmas) per entry (None for absent ones) for discriminate(real_out=...)."""
        with torch.no_grad():
            _, inp, _, ref_concat = self.d_conditioning(tgt_label, ref_label, ref_image)
            outs = []
            for real, sn in zip(reals, sigmas):
                if real is None:
                    outs.append(None)
                    continue
                real4 = real.reshape(-1, *real.shape[-3:])
                x = ops.pack_d_single(ref_concat if self.concat_ref_for_D else None, inp, real4, for_conv=_single_scale(netD))
                outs.append((netD(x, None if self.concat_ref_for_D else ref_concat, sn=sn), sn))
        return outs

    def gan_losses(self, netD, tgt_label, reals, fakes, ref_label, ref_image, for_discriminator, netDf=None, real_outs=None):
        """loss_collector.py:87-120 for the per-frame discriminator and the face discriminator (no temporal branch)."""
        total = None
        lab, inp, rl, ref_concat = self.d_conditioning(tgt_label, ref_label, ref_image)
        for k, (fake, real) in enumerate(zip(fakes, reals)):
            if fake is None:
                continue
            real4 = real.reshape(-1, *real.shape[-3:])
            fake4 = fake.reshape(-1, *fake.shape[-3:])
            losses = self.discriminate(netD, inp, fake4, real4, ref_concat, for_discriminator,
                                       real_out=real_outs[k] if real_outs is not None else None)
            losses = losses + self.discriminate_face(netDf, fake4, lab, real4, rl, ref_image, for_discriminator)
            total = losses if total is None else [a + b for a, b in zip(total, losses)]
        return total

    def flow_losses(self, flow, warped, tgt_image, fg_mask, tgt_label, ref_label, flow_gt=(None, None),
                    conf_gt=(None, None)):
        """loss_collector.py:132-162.  flow_gt / conf_gt: the FlowNet2 teacher's flow and confidence for the reference
        and the previous-frame branch ([B, T, 2|1, H, W] or None each; None with --no_flow_gt)."""
        opt = self.opt
        warp_terms, flow_terms = [], []
        flow_gt = list(flow_gt) if flow_gt is not None else [None, None]
        conf_gt = list(conf_gt) if conf_gt is not None else [None, None]

        def total(terms):                 # lambda_flow * sum of the terms
            return ops.weighted_sum(terms, [opt.lambda_flow] * len(terms)) if terms else self.zero(tgt_image)
        if not opt.isTrain:               # loss_collector.py:141,158: test-time finetune has no flow / warp terms
            return total([]), total([]), None
        for k, (f, wimg) in enumerate(zip(flow, warped)):
            if f is not None:
                warp_terms.append(l1(wimg, tgt_image))
                if flow_gt[k] is not None and getattr(opt, 'n_shot', 1) == 1:      # loss_collector.py:158-159
                    gt = flow_gt[k].reshape(-1, *flow_gt[k].shape[-3:])
                    conf = conf_gt[k].reshape(-1, *conf_gt[k].shape[-3:])
                    flow_terms.append(masked_l1(f, gt, conf * fg_mask if fg_mask is not None else conf))
        body_diff = None
        if self.pose and flow[0] is not None:
            body = part_masks(tgt_label[:, :, 2])
            ref_body = part_masks(ref_label[:, 2].unsqueeze(1)).expand_as(body)
            body = body.reshape(-1, *body.shape[-3:])
            ref_body = ref_body.reshape(-1, *ref_body.shape[-3:])
            ref_body_warp = ops.resample(ref_body.contiguous(), flow[0])
            warp_terms.append(l1(ref_body_warp, body))
            if self.has_fg:
                fg, ref_fg = fg_mask_of(opt, tgt_label, True), fg_mask_of(opt, ref_label, True)
                warp_terms.append(l1(ops.resample(ref_fg, flow[0]), fg))
            body_diff = (ref_body_warp - body).abs().sum(dim=1, keepdim=True)
        return total(flow_terms), total(warp_terms), body_diff

    def mask_losses(self, flow_mask, fake_image, warped, tgt_label, tgt_image, fg_mask, ref_fg_mask, body_diff):
        """loss_collector.py:164-204."""
        opt = self.opt
        terms = []
        if not opt.isTrain:               # loss_collector.py:172,192
            return self.zero(tgt_image)
        for m, wimg in zip(flow_mask, warped):
            if m is None:
                continue
            conf = torch.clamp(1 - (wimg - tgt_image).abs().sum(dim=1, keepdim=True), 0, 1)
            terms += [masked_l1(m, 0.0, conf), masked_l1(m, 1.0, 1 - conf)]
        if self.pose and self.warp_ref:
            m_ref = flow_mask[0]
            h, w = tgt_label.shape[-2:]
            face = face_mask_of(tgt_label[:, :, 2]).view(-1, 1, h, w)
            face = ops.pool15(face, 'avg')
            terms.append(masked_l1(m_ref, 0.0, face))
            if opt.spade_combine:
                terms.append(masked_l1(fake_image, warped[0].detach(), face))
            fg_diff = ((ref_fg_mask - fg_mask) > 0).float()
            terms += [masked_l1(m_ref, 1.0, fg_diff), masked_l1(m_ref, 1.0, body_diff)]
        return ops.weighted_sum(terms, [opt.lambda_mask] * len(terms)) if terms else self.zero(tgt_image)


def _single_scale(netD):
    """the packed discriminator input is read by ONE convolution and nothing else (a multi-scale pyramid, --num_D > 1, also
    average-pools it): ops.pack_d_* may then hand it over padded and - under `--amp` - as half"""
    return getattr(netD, 'num_D', 2) == 1 and getattr(netD, 'subarch', 'n_layers') == 'n_layers'


def amp_mode(opt):
    """`--amp` (options/base_options.py:127: an apex opt-level string, '' = off) -> operand arithmetic of the GEMM kernels.
    Every apex level that computes in half ('O1' ... 'O3') selects the fp16-operand kernels; 'bf16x3' (not an apex level)
    selects the split-bf16 kernels, which need no loss scale."""
    level = str(getattr(opt, 'amp', '') or '').lower()
    if level in ('', 'o0', 'fp32', 'f32'):
        return conv.MFMA_F32
    if level in ('o1', 'o2', 'o3', 'fp16', 'f16'):
        return conv.MFMA_F16
    if level == 'bf16x3':
        return conv.MFMA_BF16X3
    raise ValueError("unknown --amp level %r" % (getattr(opt, 'amp', ''),))


def tgt_label_device(data_list):
    return data_list[0].device


def mean_and_total(losses):
    """loss_collector.py:218-219: `losses = [torch.mean(x) ...]; loss = sum(losses)`.  Without DataParallel every loss is one
    element: its mean is itself (a view), and the total is one cat + sum instead of a chain of scalar adds."""
    means = [x if isinstance(x, int) else (x.reshape(()) if x.numel() == 1 else torch.mean(x)) for x in losses]
    tensors = [m for m in means if not isinstance(m, int)]
    total = ops.weighted_sum(tensors).reshape(()) if tensors else 0
    extra = sum(m for m in means if isinstance(m, int))
    return means, (total + extra if extra else total)


def loss_backward(opt, losses, optimizer, loss_id):
    """models/loss_collector.py:217-228: sum of means -> zero_grad -> backward -> optimiser step.  With `--amp` the
    reference scales the loss per `loss_id` (:221-224); here every optimiser owns its scaler (flat.FlatAdam.scale_loss:
    identity unless the fp16-operand mode is on) and un-scales inside its fused step."""
    # the side stream (if any) is resolved ONCE, from the losses the forward pass tagged: `mean_and_total` below rebinds `losses` to
    # fresh views that do not carry the tag (round-5 advisor: the final ordering then relied on the optimiser's tag alone)
    branch = _branch(losses, optimizer)
    with (branch.on() if branch is not None else contextlib.nullcontext()):
        losses, loss = mean_and_total(losses)
        optimizer.zero_grad()
        scale_loss = getattr(optimizer, 'scale_loss', None)
        (scale_loss(loss) if scale_loss is not None else loss).backward()
        # two-piece backward (build_optimizers(split_backward=True)): the forward pass detached at the generator's stage boundary,
        # run the second piece too - whichever optimiser is being stepped (finetune() builds its own).  Between the pieces the
        # decoder stage's gradients are final: on one GPU its Adam launch and layout refresh start now, on a side stream
        cut = getattr(optimizer, 'bwd_cut', None)
        if cut is not None and cut.has_grads() and hasattr(optimizer, 'step_stage2_early'):
            optimizer.step_stage2_early()
        networks.BackwardCut.finish_all()
        optimizer.step()
    _order_behind_branch(losses, optimizer, branch)
    return losses


def _branch(losses, optimizer=None):
    """the open side-stream branch a list of losses was computed on (tag on the first loss, else on the optimiser), or None"""
    branch = getattr(losses[0], '_fsv_branch', None) if len(losses) and torch.is_tensor(losses[0]) else None
    if branch is None or branch.stream is None:
        branch = getattr(optimizer, '_fsv_branch', None)
    return branch if (branch is not None and branch.stream is not None) else None


def _order_behind_branch(losses, optimizer, branch=None):
    """The discriminator step of an iteration with `early_generator` ran on a side stream: whoever reads the returned losses on
    the caller's stream (train.py logs float(loss) right after loss_backward and synchronises only ITS stream) must come behind it.
    One stream wait, no host synchronisation; the generator-mode pass was issued before this point and still runs next to the
    step, and the branch stays open for the real-image pass (join_early).  Round-4 advisor finding."""
    if branch is None:
        branch = _branch(losses, optimizer)
    if branch is not None and branch.stream is not None:
        cur = torch.cuda.current_stream(branch.stream.device)
        cur.wait_stream(branch.stream)
        for t in losses:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(cur)


def branch_of(losses, optimizer=None):
    """context of the stream a list of losses was computed on: the discriminator step of an iteration with
    `Vid2VidModel.early_generator` lives on a side stream (its forward pass tagged the first loss and the discriminator's
    optimiser), everything else on the caller's"""
    branch = _branch(losses, optimizer)
    return branch.on() if branch is not None else contextlib.nullcontext()


def set_random_seed(seed):
    """util/distributed.py `set_random_seed`: Python, numpy and torch streams together."""
    import random
    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def get_rank():
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def optimizer_rates(opt):
    """models/base_model.py:39-48 `get_optimizer`: (beta1, beta2, lr_G, lr_D), always derived from opt.lr."""
    if opt.no_TTUR:
        return opt.beta1, 0.999, opt.lr, opt.lr
    return 0.0, opt.beta2, opt.lr / 2, opt.lr * 2


# ------------------------------------------------------------------------------------------------ the model
class Vid2VidModel(nn.Module):
    """Drop-in for models.vid2vid_model.Vid2VidModel on the training path (modes 'generator', 'discriminator')."""

    def name(self):
        return 'Vid2VidModel'

    def initialize(self, opt, epoch=0):
        self.opt = opt
        self.isTrain = opt.isTrain
        self.pose = 'pose' in opt.dataset_mode
        self.has_fg = self.pose
        self.add_face_D = opt.add_face_D
        self.temporal = False
        self.old_lr = opt.lr
        self.save_dir = os.path.join(getattr(opt, 'checkpoints_dir', './checkpoints'), getattr(opt, 'name', 'test'))
        self.refine_face = bool(getattr(opt, 'refine_face', False))
        self.lossCollector = LossCollector(opt)
        set_random_seed(0)              # vid2vid_model.py:27: every replica builds identical weights
        opt.for_face = False
        self.netG = networks.define_G(opt)
        self.netGf = None
        if self.refine_face:               # base_model.py:174-182: a smaller generator on face_size x face_size crops
            import copy
            opt_face = copy.deepcopy(opt)
            opt_face.n_downsample_G -= 1
            if opt_face.n_adaptive_layers > 0:
                opt_face.n_adaptive_layers -= 1
            opt_face.input_nc = opt.output_nc
            opt_face.fineSize = self.lossCollector.face_size
            opt_face.aspect_ratio = 1
            opt_face.for_face = True
            self.netGf = networks.define_G(opt_face)
        input_nc = opt.label_nc if (opt.label_nc != 0 and not self.pose) else opt.input_nc
        netD_input_nc = input_nc + opt.output_nc + (1 if self.lossCollector.concat_fg_mask_for_D else 0)
        if self.lossCollector.concat_ref_for_D:
            netD_input_nc *= 2
        self.netD = networks.define_D(opt, netD_input_nc, opt.ndf, opt.n_layers_D, opt.norm_D, opt.netD_subarch,
                                      opt.num_D, not opt.no_ganFeat_loss)
        self.netDf = None
        if self.add_face_D:                # base_model.py:191-193: 6 channels = reference face region | image face region
            self.netDf = networks.define_D(opt, opt.output_nc * 2, opt.ndf, opt.n_layers_D, opt.norm_D, 'n_layers', 1,
                                           not opt.no_ganFeat_loss)
        self.netDT = None
        self.optimizer_G = self.optimizer_D = None
        # base_model.py:213-215 (inside define_networks): test mode, or a run resumed past the single-frame epochs,
        # starts temporal.  The optimisers do not exist yet here: build_optimizers() covers whatever networks exist then.
        self.start_epoch = epoch
        if (not opt.isTrain or epoch > getattr(opt, 'niter_single', 0)) and opt.n_frames_G > 1:
            self.init_temporal_model()
        set_random_seed(get_rank())     # vid2vid_model.py:45: per-rank streams from here on (load_networks() runs once the
        return self                     # module sits on its device: integration.create_model / Vid2VidModel.load_networks)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if getattr(self, 'lossCollector', None) is not None and self.lossCollector.vgg is not None:
            self.lossCollector.vgg._apply(fn)
        return out

    # optimisers are created once the module sits on its device (flat buffers are device allocations)
    def build_optimizers(self, world_size=1, process_group=None, force_exchange=False, overlap=True, sync_bn=None,
                         split_backward=False):
        """sync_bn: pool the BatchNorm statistics over the replicas (the reference's apex SyncBatchNorm under its
        multi-process path, normalization.py:15,33,80) instead of the default per-replica statistics; None reads
        FSV_SYNC_BN.  Needs equal per-rank batches and the eager step (collectives inside forward / backward)."""
        opt = self.opt
        if sync_bn is None:
            sync_bn = os.environ.get('FSV_SYNC_BN', '0') == '1'
        ops.set_bn_sync(world_size if sync_bn else 1, process_group)
        beta1, beta2, g_lr, d_lr = optimizer_rates(opt)
        # models/models.py:22-26 `amp.initialize(model, [optimizer_G, optimizer_D], opt_level=opt.amp, num_losses=2)`:
        # process-wide operand arithmetic + one dynamic loss scale per optimiser (fp16 only)
        mode = amp_mode(opt)
        conv.set_mfma_mode(mode)
        loss_scale = True if mode == conv.MFMA_F16 else None
        g_params = self._g_params(split_backward)
        self.optimizer_G = FlatAdam(g_params, g_lr, (beta1, beta2), world_size, process_group,
                                    force_exchange=force_exchange, overlap=overlap, loss_scale=loss_scale)
        self._set_split(split_backward)
        d_params = list(self.netD.parameters())
        if self.netDT is not None:         # built temporal from the start (resume past niter_single): base_model.py:274
            d_params += list(self.netDT.parameters())
        if self.netDf is not None:         # base_model.py:209-211
            d_params += list(self.netDf.parameters())
        self.optimizer_D = FlatAdam(d_params, d_lr, (beta1, beta2), world_size, process_group,
                                    force_exchange=force_exchange, overlap=overlap, loss_scale=loss_scale)
        return self.optimizer_G, self.optimizer_D

    def _g_params(self, split_backward):
        """generator parameters in optimiser order.  split_backward: the generator's stage-2 parameters (below its
        BackwardCut boundary) go LAST - FlatAdam lays its buffers out in reverse order, so their gradients occupy
        flat_g[:split_at] and can be exchanged while the rest of backward is still running (graph_step.py)."""
        g_params = list(self.netG.parameters())
        if split_backward:
            late = set(id(p) for p in self.netG.stage2_parameters())
            # three pieces (split_backward=3): the reference / attention encoders - reached last by the backward pass - first in
            # the list, i.e. LAST in the flat buffers: flat_g = [decoder stage | weight generators, embeddings, flow | encoders]
            last = set(id(p) for p in self.netG.stage3_parameters()) if int(split_backward) >= 3 else set()
            g_params = ([p for p in g_params if id(p) in last] + [p for p in g_params if id(p) not in late and id(p) not in last] +
                        [p for p in g_params if id(p) in late])
        if self.netGf is not None:         # base_model.py:204-205
            g_params = list(self.netGf.parameters()) + g_params if split_backward else g_params + list(self.netGf.parameters())
        return g_params

    def _set_split(self, split_backward):
        """split_backward: False / True (= 2) / 3 - the number of pieces of the generator's backward pass"""
        pieces = 0 if not split_backward else max(2, int(split_backward))
        self.split_backward = pieces if pieces >= 3 else bool(pieces)
        self.netG.bwd_cut = networks.BackwardCut() if pieces else None
        self.netG.bwd_cut2 = networks.BackwardCut() if pieces >= 3 else None
        self.optimizer_G.bwd_cut = self.netG.bwd_cut
        self.optimizer_G.split_at = (sum(p.numel() for p in self.netG.stage2_parameters() if p.requires_grad)
                                     if pieces else 0)
        # gradients of flat_g[:split_at2] are complete after the SECOND piece (three pieces only; else = split_at)
        self.optimizer_G.split_at2 = (self.optimizer_G.total - sum(p.numel() for p in self.netG.stage3_parameters() if p.requires_grad)
                                      - (sum(p.numel() for p in self.netGf.parameters() if p.requires_grad) if self.netGf is not None else 0)
                                      if pieces >= 3 else self.optimizer_G.split_at)

    def init_temporal_model(self):
        """models/base_model.py:259-279: the generator grows its previous-frame flow / embedding branches, the temporal
        discriminator netDT (input = tD stacked frames) is created, and both optimisers are re-laid over the new
        parameter sets with fresh Adam state and the opt.lr-derived rates of `get_optimizer`.  The optimiser OBJECTS stay
        the ones create_model handed to train.py (train.py:36 keeps them for the whole run; models.update_models calls
        this method at epoch niter_single + 1), so the loop keeps stepping live parameters."""
        opt = self.opt
        self.temporal = True
        set_random_seed(0)                 # generator.py:157
        self.netG.init_temporal_network()
        dev = next(self.netG.parameters()).device
        self.netG.to(dev)
        if opt.isTrain:
            self.lossCollector.tD = min(opt.n_frames_D, opt.n_frames_G)
            self.netDT = networks.define_D(opt, opt.output_nc * self.lossCollector.tD, opt.ndf, opt.n_layers_D, opt.norm_D,
                                           'n_layers', 1, not opt.no_ganFeat_loss).to(dev)
        set_random_seed(get_rank())        # generator.py:179
        if self.optimizer_G is not None:
            _, _, g_lr, d_lr = optimizer_rates(opt)
            split = getattr(self, 'split_backward', False)
            self.optimizer_G.rebuild(self._g_params(split), lr=g_lr)          # the loss scale found so far carries over
            self._set_split(split)
            d_params = list(self.netD.parameters()) + (list(self.netDT.parameters()) if self.netDT is not None else [])
            if self.netDf is not None:
                d_params += list(self.netDf.parameters())
            self.optimizer_D.rebuild(d_params, lr=d_lr)
        return self.optimizer_G

    def update_learning_rate(self, epoch):
        """models/base_model.py:245-257."""
        opt = self.opt
        new_lr = opt.lr * (1 - (epoch - opt.niter) / (opt.niter_decay + 1))
        g_lr, d_lr = (new_lr, new_lr) if opt.no_TTUR else (new_lr / 2, new_lr * 2)
        self.optimizer_G.set_lr(g_lr)
        self.optimizer_D.set_lr(d_lr)
        self.old_lr = new_lr

    def save_network(self, network, network_label, epoch_label):
        """models/base_model.py:51-56: `<epoch>_net_<label>.pth` holding the CPU state_dict (same keys as the reference)"""
        os.makedirs(self.save_dir, exist_ok=True)
        sd = {k: v.detach().cpu() for k, v in network.state_dict().items()}
        torch.save(sd, os.path.join(self.save_dir, '%s_net_%s.pth' % (epoch_label, network_label)))

    def save_networks(self, which_epoch):
        """models/base_model.py:219-227"""
        self.save_network(self.netG, 'G', which_epoch)
        if self.netGf is not None:
            self.save_network(self.netGf, 'Gf', which_epoch)
        self.save_network(self.netD, 'D', which_epoch)
        if self.temporal and self.netDT is not None:
            self.save_network(self.netDT, 'DT', which_epoch)
        if self.add_face_D:
            self.save_network(self.netDf, 'Df', which_epoch)

    def load_network(self, network, network_label, epoch_label, save_dir=''):
        """models/base_model.py:59-93: exact load, else only the keys the network has, else every tensor whose shape
        matches (a checkpoint from before init_temporal_model leaves flow_network_temp uninitialised).  Returns the set
        of layers that were not initialised (empty on an exact load), None when the file does not exist."""
        path = os.path.join(save_dir or self.save_dir, '%s_net_%s.pth' % (epoch_label, network_label))
        if not os.path.isfile(path):
            return None
        loaded = torch.load(path, map_location='cpu')
        try:
            network.load_state_dict(loaded)
            return set()
        except Exception:
            pass
        model_dict = network.state_dict()
        try:
            network.load_state_dict({k: v for k, v in loaded.items() if k in model_dict})
            return set()
        except Exception:
            pass
        not_initialized = set()
        for k, v in loaded.items():
            if k in model_dict and v.shape == model_dict[k].shape:
                model_dict[k] = v
        for k, v in model_dict.items():
            if k not in loaded or v.shape != loaded[k].shape:
                not_initialized.add('.'.join(k.split('.')[:2]))
                if 'flow_network_temp' in k:
                    network.flow_temp_is_initalized = False
        network.load_state_dict(model_dict)
        return not_initialized

    def load_networks(self):
        """models/base_model.py:229-243"""
        opt = self.opt
        if not self.isTrain or getattr(opt, 'continue_train', False) or getattr(opt, 'load_pretrain', ''):
            path = '' if (not self.isTrain or getattr(opt, 'continue_train', False)) else opt.load_pretrain
            epoch = getattr(opt, 'which_epoch', 'latest')
            self.load_network(self.netG, 'G', epoch, path)
            if self.temporal and opt.warp_ref and not self.netG.flow_temp_is_initalized:     # base_model.py:234-235
                self.netG.load_pretrained_net(self.netG.flow_network_ref, self.netG.flow_network_temp)
            if self.netGf is not None:
                self.load_network(self.netGf, 'Gf', epoch, path)
            if (self.isTrain and not getattr(opt, 'load_pretrain', '')) or getattr(opt, 'finetune', False):
                self.load_network(self.netD, 'D', epoch, path)
                if self.isTrain and self.temporal and self.netDT is not None:
                    self.load_network(self.netDT, 'DT', epoch, path)
                if self.add_face_D:
                    self.load_network(self.netDf, 'Df', epoch, path)
            for o in (self.optimizer_G, self.optimizer_D):      # cached GEMM layouts follow the loaded values
                if o is not None:
                    o.refresh_layouts()

    # ---------------------------------------------------------------------------------------------- forward
    # Discriminator step next to the generator pass (opt-in: bench.py / GraphedIteration set `early_generator` for iterations
    # without a gradient exchange).  The generator-mode forward pass of train.py:61 depends on nothing the discriminator step of
    # train.py:58-59 produces: the generator's weights, spectral-norm vectors and BatchNorm buffers are final once the step's own
    # (no-grad) generator pass has run.  With the switch on, `mode='discriminator'` puts the discriminator's forward pass on a
    # side stream and issues the generator-mode pass of the SAME data_list right behind the no-grad one on the caller's stream;
    # loss_backward runs the rest of the discriminator step (backward, finalisation, Adam, layout refresh - hundreds of launches
    # that each fill a fraction of the chip) on that side stream too, and the following `mode='generator'` call joins it and picks
    # the generator pass up.  Same kernels on the same data in the same per-network order: bit-identical results.  The returned
    # discriminator losses live on the side stream: read them after the generator-mode call (or after `join_early()`).
    early_generator = False
    _pre_g = None

    @staticmethod
    def _data_key(data_list):
        flat = []
        for item in data_list:
            flat.extend(item if isinstance(item, (list, tuple)) else [item])
        return tuple(None if t is None else (id(t), t.data_ptr(), t._version) for t in flat)

    def early_generate(self, data_list):
        """The generator-mode forward pass of train.py:61 issued AHEAD of the `mode='generator'` call that will use it, on the
        caller's stream (no side stream): graph_step.GraphedIteration puts it, as a segment of its own, between the launch of the
        discriminator's gradient all-reduce and the wait for it - at N > 1 the exchange then runs next to the generator's whole
        forward pass instead of in front of it.  The pass depends on nothing the discriminator step produces (see
        `early_generator`); same kernels, same per-network order: bit-identical to the serial schedule."""
        tgt_label, tgt_image, _, _, ref_label, ref_image, p_label, p_real, p_fake = data_list
        with conv.stats_pass(tgt_label_device(data_list)):
            gen = self.generate_images(encode_label(self.opt, tgt_label), tgt_image, encode_label(self.opt, ref_label), ref_image,
                                       [p_label, p_real, p_fake])
        self._pre_g = (self._data_key(data_list), streams.Branch(None), gen)

    def join_early(self, data_list=None):
        """join the discriminator step's side stream into the current one; returns (generator pass, real-image pass or None) if
        `data_list` is the data the early passes were computed from (else None: they are dropped)"""
        if self._pre_g is None:
            return None
        key, branch, gen = self._pre_g
        ours = data_list is not None and key == self._data_key(data_list)
        real_outs = None
        if ours and branch.stream is not None and _SPLIT_D_PASS and os.environ.get('FSV_EARLY_REAL', '1') == '1':     # (in-box A/B)
            # the G step's discriminator pass over the REAL images (no autograd; needs the discriminator as the step that is
            # finishing on the side stream leaves it): behind that step on the same stream, i.e. still next to the generator pass
            tgt_label, tgt_image, _, _, ref_label, ref_image = data_list[:6]
            with branch.on():
                rb = self._real_branch(encode_label(self.opt, tgt_label), tgt_image, encode_label(self.opt, ref_label), ref_image)
                real_outs = rb() if rb is not None else None
        self._pre_g = None
        branch.finish(real_outs)
        self._fold_bn_standins()             # (twin generator passes: the second pass's running-statistics update, in pass order)
        if self.optimizer_D is not None:
            self.optimizer_D._fsv_branch = None
        return (gen, real_outs) if ours else None

    def _real_branch(self, tgt_label, tgt_image, ref_labels, ref_images):
        """closure for the G step's discriminator pass over the real images (LossCollector.real_pass), or None when the pass
        cannot be split off (attention picks the conditioning reference after the generator ran)"""
        opt = self.opt
        if not (_SPLIT_D_PASS and getattr(opt, 'n_shot', 1) == 1 and ref_labels.shape[1] == 1):
            return None
        real = tgt_image[:, 0]
        ref_labels_valid = valid_labels(opt, ref_labels)
        fg0 = fg_mask_of(opt, tgt_label[:, 0], self.has_fg)
        ref_fg0 = fg_mask_of(opt, ref_labels[:, 0], self.has_fg)
        G = self.netG
        with_raw = ((not G.spade_combine) and (G.warp_ref or G.warp_prev)) or G.add_raw_output_loss
        reals0 = [real, real * union_fg(fg0, ref_fg0, self.has_fg) if with_raw else None]
        # ONE power iteration per stacked call of the reference, shared by its real and its generated half
        sigmas = [self.netD.begin_pass() if r is not None else None for r in reals0]
        lc = self.lossCollector
        return lambda: lc.real_pass(self.netD, tgt_label, reals0, ref_labels_valid[:, 0], ref_images[:, 0], sigmas)

    def forward(self, data_list, save_images=False, mode='inference', dummy_bs=0):
        # (joined BEFORE this pass zeroes the statistics arena the side stream may still be working in)
        pre = self.join_early(data_list if mode == 'generator' else None)
        if mode != 'generator' or getattr(self, '_label_memo', None) is None:
            self._label_memo = {}          # a new iteration (train.py:58) / an inference call: nothing of the last one survives
        # one zeroed arena per pass for the normalisation statistics the convolutions leave behind (conv.stats_pass)
        with conv.stats_pass(tgt_label_device(data_list)), _memo_scope(self._label_memo):
            return self._forward(data_list, save_images, mode, pre)

    def _forward(self, data_list, save_images, mode, pre=None):
        opt = self.opt
        tgt_label, tgt_image, flow_gt, conf_gt, ref_label, ref_image, p_label, p_real, p_fake = data_list
        tgt_label, ref_label = encode_label(opt, tgt_label), encode_label(opt, ref_label)
        prevs = [p_label, p_real, p_fake]
        if mode == 'generator':
            losses, generated, prev = self.forward_generator(tgt_label, tgt_image, ref_label, ref_image, prevs,
                                                             flow_gt, conf_gt, pre=pre)
            return losses, generated if save_images else [], prev
        if mode == 'discriminator':
            early = self._data_key(data_list) if (self.early_generator and self.isTrain and torch.is_grad_enabled()) else None
            return self.forward_discriminator(tgt_label, tgt_image, ref_label, ref_image, prevs, early=early)
        return self.inference(tgt_label, ref_label, ref_image)

    def inference(self, tgt_label, ref_labels, ref_images):
        """vid2vid_model.py:179-205 (test.py:39-41): one frame per call, previous labels / outputs carried in self.prevs;
        call reset_inference() between sequences.  --finetune and --refine_face are not part of this build."""
        opt = self.opt
        if getattr(self, 'prevs', None) is None:
            self.prevs = [None, None]
            prevs = [None, None]
            self.t = 0
        else:
            b, _, _, h, w = tgt_label.shape
            prevs = [p.contiguous().view(b, -1, h, w) for p in self.prevs]
            self.t += 1
        tgt_label_valid = valid_labels(opt, tgt_label[:, -1])
        ref_labels_valid = valid_labels(opt, ref_labels)
        if getattr(opt, 'finetune', False) and self.t == 0:
            self.finetune(ref_labels, ref_images)
        with torch.no_grad():
            fake, flow, mask, raw, warped, _, _, atn_score, ref_idx = self.netG(tgt_label_valid, ref_labels_valid,
                                                                                ref_images, prevs, t=self.t)
            if self.refine_face:                     # vid2vid_model.py:198-201
                pick = networks.pick_ref
                fake = self.refine_face_region(tgt_label_valid, fake, tgt_label[:, -1], pick(ref_labels_valid, ref_idx),
                                               pick(ref_images, ref_idx), pick(ref_labels, ref_idx))
            n_prev = opt.n_frames_G - 1
            new = []
            for old, now in zip(self.prevs, (tgt_label_valid, fake)):          # concat_prev, vid2vid_model.py:169-176
                if old is None:
                    new.append(now.unsqueeze(1).repeat(1, n_prev, 1, 1, 1).detach())
                else:
                    new.append(torch.cat([old[:, 1:], now.unsqueeze(1)], dim=1).detach())
            self.prevs = new
        return fake, raw, warped, flow, mask, atn_score

    def reset_inference(self):
        self.prevs = None

    def refine_face_region(self, label_valid, fake_image, label, ref_label_valid, ref_image, ref_label):
        """face_refiner.py:24-30: crop (4 px inside the face box) label / coarse output and the reference's, run the face
        generator on the crops, paste `coarse + refinement` back (bilinear resize to the box, clamp)."""
        lc = self.lossCollector
        boxes = ops.face_boxes(label, lc.use_openpose, crop_smaller=4)
        ref_boxes = ops.face_boxes(ref_label, lc.use_openpose, crop_smaller=4)
        size = lc.face_size
        label_face = ops.crop_face(label_valid, boxes, size)
        coarse = ops.crop_face(fake_image, boxes, size).detach()
        ref_label_face = ops.crop_face(ref_label_valid, ref_boxes, size)
        ref_image_face = ops.crop_face(ref_image, ref_boxes, size)
        fake_face = self.netGf(label_face, ref_label_face.unsqueeze(1), ref_image_face.unsqueeze(1), img_coarse=coarse)
        return ops.paste_face(fake_image, fake_face + coarse, boxes)

    def finetune(self, ref_labels, ref_images):
        """vid2vid_model.py:207-237: test-time adaptation on the reference images - the generator layers whose names
        contain 'fc' / 'conv_img' / 'up' (base_model.py:149-165) and the discriminator(s), 100 iterations (opt.
        finetune_iterations to override), target = a randomly rolled / flipped reference (util/util.py:157-168, Python's
        `random` stream, same draw order as the reference), G step before D step."""
        import random
        opt = self.opt
        names = ('fc', 'conv_img', 'up')
        g_params = [p for n, p in self.netG.named_parameters() if any(t in n for t in names)]
        frozen = [p for n, p in self.netG.named_parameters() if not any(t in n for t in names) and p.requires_grad]
        for p in frozen:                 # the reference leaves them trainable but never steps them: same result, less work
            p.requires_grad_(False)
        beta1, beta2, g_lr, d_lr = optimizer_rates(opt)
        mode = amp_mode(opt)               # models/models.py:22-26: amp.initialize covers the finetune optimisers too
        conv.set_mfma_mode(mode)
        loss_scale = True if mode == conv.MFMA_F16 else None
        self.optimizer_G = FlatAdam(g_params, g_lr, (beta1, beta2), loss_scale=loss_scale)
        d_params = list(self.netD.parameters()) + (list(self.netDf.parameters()) if self.netDf is not None else [])
        self.optimizer_D = FlatAdam(d_params, d_lr, (beta1, beta2), loss_scale=loss_scale)

        def roll(t, ny, nx, flip):
            t = torch.cat([t[:, :, -ny:], t[:, :, :-ny]], dim=2)
            t = torch.cat([t[:, :, :, -nx:], t[:, :, :, :-nx]], dim=3)
            return torch.flip(t, dims=[3]) if flip else t
        history = []
        try:
            for it in range(1, int(getattr(opt, 'finetune_iterations', 100)) + 1):
                idx = random.randrange(ref_labels.size(1))
                h, w = ref_labels.shape[-2:]
                ny = random.choice([random.randrange(h // 16), h - random.randrange(h // 16)])
                nx = random.choice([random.randrange(w // 16), w - random.randrange(w // 16)])
                flip = random.random() > 0.5
                tgt_label = roll(ref_labels[:, idx], ny, nx, flip).unsqueeze(1)
                tgt_image = roll(ref_images[:, idx], ny, nx, flip).unsqueeze(1)
                g_losses, _, _ = self.forward_generator(tgt_label, tgt_image, ref_labels, ref_images, [None] * 3)
                g_losses = loss_backward(opt, g_losses, self.optimizer_G, 0)
                d_losses = self.forward_discriminator(tgt_label, tgt_image, ref_labels, ref_images, [None] * 3)
                d_losses = loss_backward(opt, d_losses, self.optimizer_D, 1)
                history.append([float(x) for x in list(g_losses) + list(d_losses)])
        finally:
            for p in frozen:
                p.requires_grad_(True)
        return history

    def generate_images(self, tgt_labels, tgt_images, ref_labels, ref_images, prevs, want_prevs=True, ref_labels_valid=None):
        """vid2vid_model.py:130-158 with n_frames_per_gpu == 1.  want_prevs=False: the caller drops the updated
        previous-frame buffers (the D step), so they are not built."""
        opt = self.opt
        if ref_labels_valid is None:
            ref_labels_valid = valid_labels(opt, ref_labels)
        b, _, _, h, w = tgt_labels.shape
        tgt_label_t = tgt_labels[:, 0]
        tgt_label_valid = valid_labels(opt, tgt_label_t)
        tgt_image = tgt_images[:, 0]
        prev_t = [p.contiguous().view(b, -1, h, w) if p is not None else None for p in (prevs[0], prevs[2])]
        # round 6: the generator optimiser's two big fills (flat gradient buffer + weight-gradient arena: 392 + 312 MB at the bench
        # widths, 93 us) used to sit in zero_grad() - at the one serial point of the step, between the losses and the backward pass.
        # They are issued here on a side stream next to the generator's forward pass (nothing touches those buffers before the
        # backward pass) and joined before this call returns; zero_grad() then skips them (flat.FlatAdam.zero_early).  Measured
        # slower (+0.2 ms: one more fork in the captured pass) - opt-in, FSV_ZERO_EARLY=1.
        zero_early = None
        opt_g = getattr(self, 'optimizer_G', None)
        if self.isTrain and torch.is_grad_enabled() and hasattr(opt_g, 'zero_early'):
            zero_early = opt_g.zero_early(tgt_label_valid)
        try:
            fake, flow, mask, raw, warped, _, _, atn_score, ref_idx = self.netG(tgt_label_valid, ref_labels_valid, ref_images,
                                                                                prev_t)
        finally:
            if zero_early is not None:
                zero_early.finish()
        self.atn_score = atn_score
        pick = networks.pick_ref                     # vid2vid_model.py:144 (the attended reference when n_shot > 1)
        ref_label_valid, ref_label_t, ref_image_t = pick(ref_labels_valid, ref_idx), pick(ref_labels, ref_idx), \
            pick(ref_images, ref_idx)
        if self.refine_face:                         # vid2vid_model.py:146-148
            fake = self.refine_face_region(tgt_label_valid, fake, tgt_label_t, ref_label_valid, ref_image_t, ref_label_t)
        fg, ref_fg = fg_mask_of(opt, tgt_label_t, self.has_fg), fg_mask_of(opt, ref_label_t, self.has_fg)
        if raw is not None:
            raw = raw * union_fg(fg, ref_fg, self.has_fg)
        n_prev = opt.n_frames_G - 1
        new_prevs = []
        for old, now in (zip(prevs, (tgt_label_valid, tgt_image, fake)) if want_prevs else ()):
            if old is None:
                new_prevs.append(now.detach().unsqueeze(1).repeat(1, n_prev, 1, 1, 1))
            else:
                new_prevs.append(torch.cat([old[:, 1:], now.detach().unsqueeze(1)], dim=1).detach())
        return (fake, raw, warped, flow, mask), (fg, ref_fg), (ref_label_valid, ref_image_t), new_prevs

    def forward_discriminator(self, tgt_label, tgt_image, ref_labels, ref_images, prevs, early=None):
        """vid2vid_model.py:106-128.  early: key of the data_list when the discriminator's part goes to a side stream and the
        generator-mode pass is issued here (see `early_generator`)."""
        if early is not None and self._twin_ready(tgt_label, ref_labels, prevs):
            return self._forward_discriminator_twin(tgt_label, tgt_image, ref_labels, ref_images, prevs, early)
        with torch.no_grad():
            (fake, raw, _, _, _), (fg, ref_fg), (ref_label, ref_image), _ = \
                self.generate_images(tgt_label, tgt_image, ref_labels, ref_images, prevs, want_prevs=False)
        branch = streams.Branch(tgt_label) if early is not None else None
        if branch is not None:
            branch.uses([fake, raw, fg, ref_fg, ref_label, ref_image, tgt_label])
        with (branch.guarded() if branch is not None else contextlib.nullcontext()):
            fg_union = union_fg(fg, ref_fg, self.has_fg)
            real = tgt_image[:, 0]
            losses = self.lossCollector.gan_losses(self.netD, tgt_label, [real, real * fg_union], [fake, raw], ref_label,
                                                   ref_image, for_discriminator=True, netDf=self.netDf)
            if self.isTrain and self.opt.lambda_temp > 0 and prevs[0] is not None and self.netDT is not None:   # vid2vid_model.py:115
                real_all = torch.cat([prevs[1], tgt_image], dim=1)
                fake_all = torch.cat([prevs[2], fake.unsqueeze(1)], dim=1)
                losses = list(losses) + self.lossCollector.temporal_losses(self.netDT, real_all, fake_all, True)
            losses = LossCollector.outward(losses)
        if branch is not None:
            try:
                with torch.enable_grad():
                    gen = self.generate_images(tgt_label, tgt_image, ref_labels, ref_images, prevs)
            except BaseException:
                branch.finish()              # (never leave the side stream reserved behind an exception)
                raise
            self._pre_g = (early, branch, gen)
            # loss_backward continues the discriminator step on the branch's stream: the tag rides on the losses AND on the
            # optimiser that will step them (a caller may hand loss_backward clones of the losses)
            losses[0]._fsv_branch = branch
            if self.optimizer_D is not None:
                self.optimizer_D._fsv_branch = branch
        return losses

    # ---- the iteration's TWO generator passes next to each other (round 6) ------------------------------------------------------
    # train.py:58 runs the generator without autograd (the discriminator step's fake image), train.py:61 runs it again with: same
    # weights, same data - the passes differ in the spectral-norm iteration they see and in what they keep.  Nothing the second
    # reads is produced by the first once (1) both power iterations are issued up front, in pass order (netG.begin_pass x 2), and
    # (2) the second pass writes its BatchNorm running-statistics updates into zeroed stand-ins that are folded into the buffers
    # behind the first pass's in-place updates (networks.BatchNorm._redirect: the same two products and one sum per element).
    # With `early_generator` on, `mode='discriminator'` then issues the no-grad pass AND the whole discriminator step on a side
    # stream and the generator-mode pass on the caller's stream right away (the pass with autograd has to be the one on the
    # caller's stream: its backward runs where its forward ran, see `early_generator`).  Same kernels, same data, same
    # per-network order of everything stateful: bit-identical to the sequential schedule (tests/test_graph_step_emu.py on the
    # emulator's single queue; tests/graph_step_checks.py on hardware).  Only for passes that meet every BatchNorm / spectral layer
    # once (no previous frames, no --add_raw_output_loss, one reference image, no face generator) and in exact fp32 (the half
    # twins of the `--amp` path are made lazily on whichever stream asks first).
    # MEASURED AND NOT ADOPTED (profiles/r06_step_ab_schedule.txt, one box): sequential 42.91 / 42.96 ms per step; twin passes with the
    # no-grad pass's own fork rooted at the caller's stream (four streams) 43.48 / 44.14; twin passes with that pass on ONE stream
    # 43.01 / 43.01 - two heavy passes next to each other take their time from each other, and what the overlap of their launch-bound
    # stretches buys is what the no-grad pass loses with its own fork.  Opt-in: FSV_TWIN_G=1 (2: also on host tensors, test-suite).
    def _twin_ready(self, tgt_label, ref_labels, prevs):
        # (FSV_TWIN_G=2: also on host tensors - the emulator's single queue runs the two passes in issue order, which exercises the
        # spectral-norm queue and the stand-in fold of the schedule in the CPU test-suite)
        sw = os.environ.get('FSV_TWIN_G', '0')
        if sw not in ('1', '2') or not ((streams.ENABLED and tgt_label.is_cuda) or sw == '2'):
            return False
        G = self.netG
        if (not self.isTrain or not G.training or any(p is not None for p in prevs) or G.add_raw_output_loss or
                getattr(self.opt, 'n_shot', 1) != 1 or ref_labels.shape[1] != 1 or self.netGf is not None or self.refine_face or
                conv.mfma_mode() != conv.MFMA_F32 or ops.bn_sync_world() > 1):
            return False
        return self._bn_standins(tgt_label.device) is not None

    def _bn_standins(self, device):
        mods = [m for m in self.netG.modules() if isinstance(m, networks.BatchNorm)]
        st = getattr(self, '_bn_twin', None)
        if st is None or st['n'] != len(mods) or st['flat'].device != device:
            if device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
                return None                  # (built by the eager iterations in front of a capture)
            total = sum(2 * m.running_mean.numel() for m in mods)
            flat = torch.zeros(total, dtype=torch.float32, device=device)
            table, off = {}, 0
            for m in mods:
                c = m.running_mean.numel()
                table[id(m)] = [flat[off:off + c], flat[off + c:off + 2 * c], False]
                off += 2 * c
            st = self._bn_twin = dict(n=len(mods), flat=flat, table=table, mods=mods, pending=False)
        return st

    def _fold_bn_standins(self):
        """running = (1 - m) running + stand-in for every BatchNorm site of the generator: the second pass's update, applied behind
        the first's (on the caller's stream, after the side stream has been joined)"""
        st = getattr(self, '_bn_twin', None)
        if st is None or not st['pending']:
            return
        st['pending'] = False
        import numpy as np
        keep = float(np.float32(1.0) - np.float32(networks.BatchNorm.MOMENTUM))        # the kernel's (1.f - momentum)
        real, stand = [], []
        for m in st['mods']:
            a, b, _ = st['table'][id(m)]
            real += [m.running_mean, m.running_var]
            stand += [a, b]
        with torch.no_grad():
            torch._foreach_mul_(real, keep)
            torch._foreach_add_(real, stand)

    def _forward_discriminator_twin(self, tgt_label, tgt_image, ref_labels, ref_images, prevs, early):
        G = self.netG
        st = self._bn_standins(tgt_label.device)
        self._fold_bn_standins()             # (an iteration whose generator-mode call never came)
        st['flat'].zero_()
        for ent in st['table'].values():
            ent[2] = False
        G.begin_pass()                       # power iteration of the no-grad pass ...
        G.begin_pass()                       # ... and of the generator-mode pass, both on the caller's stream, in pass order
        # everything the no-grad pass launches in front of its own fork (the label helpers of generate_images) runs HERE, on the
        # caller's stream, so that the fork's branches can start from this stream (streams.FORK_ROOT)
        rooted = _MEMO is not None and os.environ.get('FSV_TWIN_INNER', '1') == '1'
        if rooted:
            valid_labels(self.opt, ref_labels)
            valid_labels(self.opt, tgt_label[:, 0])
        main = torch.cuda.current_stream(tgt_label.device) if tgt_label.is_cuda else None
        branch = streams.Branch(tgt_label)
        branch.uses([tgt_label, tgt_image, ref_labels, ref_images])
        was = streams.CROSS
        streams.CROSS = True
        try:
            with streams.hold():
                with branch.guarded():
                    inner = streams.ENABLED
                    if not rooted:
                        streams.ENABLED = False          # the no-grad pass's own branches in issue order on its stream (FORK_ROOT)
                    else:
                        streams.FORK_ROOT = main
                    try:
                        with torch.no_grad():
                            (fake, raw, _, _, _), (fg, ref_fg), (ref_label, ref_image), _ = \
                                self.generate_images(tgt_label, tgt_image, ref_labels, ref_images, prevs, want_prevs=False)
                    finally:
                        streams.ENABLED = inner
                        streams.FORK_ROOT = None
                    fg_union = union_fg(fg, ref_fg, self.has_fg)
                    real = tgt_image[:, 0]
                    losses = self.lossCollector.gan_losses(self.netD, tgt_label, [real, real * fg_union], [fake, raw], ref_label,
                                                           ref_image, for_discriminator=True, netDf=self.netDf)
                    losses = LossCollector.outward(losses)
                networks.BatchNorm._redirect = st['table']
                try:
                    with torch.enable_grad():
                        gen = self.generate_images(tgt_label, tgt_image, ref_labels, ref_images, prevs)
                except BaseException:
                    branch.finish()
                    G._sn_presets = []
                    raise
                finally:
                    networks.BatchNorm._redirect = None
        finally:
            streams.CROSS = was
        st['pending'] = True
        self._pre_g = (early, branch, gen)
        losses[0]._fsv_branch = branch
        if self.optimizer_D is not None:
            self.optimizer_D._fsv_branch = branch
        return losses

    def forward_generator(self, tgt_label, tgt_image, ref_labels, ref_images, prevs, flow_gt=(None, None),
                          conf_gt=(None, None), pre=None):
        """vid2vid_model.py:62-104.  pre: the result of generate_images when the discriminator-mode call already issued it."""
        lc = self.lossCollector
        opt = self.opt
        real = tgt_image[:, 0]
        # The G step only needs d(loss)/d(fake) from the discriminator.  The reference lets autograd also fill the
        # discriminator's (unused, later zeroed) weight gradients; skipping them changes no result of either step.
        d_params = [p for p in self.netD.parameters() if p.requires_grad]
        if self.netDf is not None:
            d_params += [p for p in self.netDf.parameters() if p.requires_grad]
        if self.netDT is not None:
            d_params += [p for p in self.netDT.parameters() if p.requires_grad]
        real_outs = None
        pre_gen, pre_real = pre if pre is not None else (None, None)
        # The discriminator's pass over the real images needs the data only (with one reference image; with attention the
        # conditioning is the attended reference, known after the generator ran): it is issued as a branch next to the
        # generator's forward pass (streams.fork; with an early generator pass: behind the discriminator step on ITS side stream,
        # join_early) and carries no autograd graph - loss_collector.py:47-68 stacks real and generated images into one batch,
        # whose backward pass then moves a zero gradient through the real half.
        real_branch = self._real_branch(tgt_label, tgt_image, ref_labels, ref_images) if pre_real is None else None
        if pre_gen is not None:
            gen, real_outs = pre_gen, (pre_real if pre_real is not None else (real_branch() if real_branch is not None else None))
        elif real_branch is not None:
            gen, real_outs = streams.fork(tgt_label, [
                lambda: self.generate_images(tgt_label, tgt_image, ref_labels, ref_images, prevs), real_branch])
        else:
            gen = self.generate_images(tgt_label, tgt_image, ref_labels, ref_images, prevs)
        (fake, raw, warped, flow, mask), (fg, ref_fg), (ref_label, ref_image), prevs_new = gen
        fg_union = union_fg(fg, ref_fg, self.has_fg)
        for p in d_params:
            p.requires_grad_(False)
        gt_gan = gt_feat = None
        try:
            g_gan, g_feat, gf_gan, gf_feat = lc.gan_losses(self.netD, tgt_label, [real, real * fg_union], [fake, raw],
                                                           ref_label, ref_image, for_discriminator=False,
                                                           netDf=self.netDf, real_outs=real_outs)
            if self.isTrain and self.opt.lambda_temp > 0 and prevs[0] is not None and self.netDT is not None:  # :70-75
                real_all = torch.cat([prevs[1], tgt_image], dim=1)
                fake_all = torch.cat([prevs[2], fake.unsqueeze(1)], dim=1)
                gt_gan, gt_feat = lc.temporal_losses(self.netDT, real_all, fake_all, False)
        finally:
            for p in d_params:
                p.requires_grad_(True)
        z = lc.zero(fake)
        g_vgg = lc.vgg_losses(fake, raw, real, fg_union)
        f_flow, f_warp, body_diff = lc.flow_losses(flow, warped, real, fg, tgt_label, ref_label, flow_gt, conf_gt)
        f_mask = lc.mask_losses(mask, fake, warped, tgt_label, real, fg, ref_fg, body_diff)
        losses = [g_gan, g_feat, g_vgg, gf_gan, gf_feat, gt_gan if gt_gan is not None else z,
                  gt_feat if gt_feat is not None else z, f_flow, f_warp, f_mask]
        # the reference returns fake / raw as [B, T, ...] and - because forward_generator rebinds them through
        # self.reshape (vid2vid_model.py:88-89) - warped / flow / mask as 4-D tensors
        up = lambda t: t.unsqueeze(1) if t is not None else None
        generated = [up(fake), up(raw), list(warped), list(flow), list(mask), self.atn_score]
        return LossCollector.outward(losses), generated, prevs_new


def create_model(opt, epoch=0, device=None):
    """models/models.py:16-38 without the DataParallel wrapper: returns (model, optimizers)."""
    model = Vid2VidModel().initialize(opt, epoch)
    if device is not None:
        model = model.to(device)
    model.train()
    return model
