"""CPU statement of the narrow-operand (`--amp`) convolution arithmetic of csrc/conv_np.hip.  TEST INFRASTRUCTURE ONLY.

The reference reaches fp16 through NVIDIA apex (`amp.initialize(..., opt_level=opt.amp)`, models/models.py:22-26; scaled
backward in models/loss_collector.py:221-224).  apex is neither vendored in the reference nor installed here, and there is
no CUDA device to run its fp16 kernels on, so no reference output exists to pin this path to: **parity unpinned**.  What
this file pins is the *definition* the HIP kernels implement, in plain fp32 torch on the CPU:

  mode 1 "f16"     every GEMM operand (activation, weight, and in the backward GEMMs the incoming gradient) is rounded to
                   IEEE half (round to nearest even); products are exact in fp32 and are summed in fp32.  This is the
                   contraction apex O1 runs (fp16 operands, fp32 accumulation inside the tensor-core GEMM) - apex
                   additionally rounds the *result* to half, which this path does not (outputs stay fp32).
  mode 2 "bf16x3"  x = hi + lo with hi = bf16(x), lo = bf16(x - hi); a*b is replaced by a_lo*b_hi + a_hi*b_lo + a_hi*b_hi.

plus apex's dynamic loss-scale rule (apex/amp/scaler.py semantics: start at 2**16, halve and skip the step when a gradient
is not finite, double after 2000 consecutive good steps, cap 2**24), restated in `LossScaler`.
"""
import torch
import torch.nn.functional as F


def planes(x, mode):
    """the fp32 values of the operand planes the kernel stages for tensor x"""
    # (back in the tensor's own dtype: the fp64 runs of the whole-iteration oracle keep exact products and fp64 sums)
    if mode == 1:
        return [x.to(torch.float16).to(x.dtype)]
    if mode == 2:
        hi = x.to(torch.bfloat16).to(x.dtype)
        lo = (x - hi).to(torch.bfloat16).to(x.dtype)
        return [hi, lo]
    return [x]


def _pairs(a, b):
    """(a-plane, b-plane) products kept by the kernel, smallest first"""
    if len(a) == 1:
        return [(a[0], b[0])]
    return [(a[1], b[0]), (a[0], b[1]), (a[0], b[0])]


class _Conv2dNp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, padding, mode, dx_half=False):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, mode)
        ctx.dx_half = dx_half
        y = 0
        for xa, wb in _pairs(planes(x, mode), planes(w, mode)):
            y = y + F.conv2d(xa, wb, None, stride=stride, padding=padding)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, mode = ctx.cfg
        dx = dw = 0
        for ga, wb in _pairs(planes(dy, mode), planes(w, mode)):
            dx = dx + torch.nn.grad.conv2d_input(x.shape, wb, ga, stride=stride, padding=padding)
        for xa, gb in _pairs(planes(x, mode), planes(dy, mode)):
            dw = dw + torch.nn.grad.conv2d_weight(xa, w.shape, gb, stride=stride, padding=padding)
        if ctx.dx_half:            # the gradient of an input that lives in HBM as half is stored as half
            dx = dx.to(torch.float16).to(dx.dtype)
        return dx, dw, None, None, None, None


def conv2d(x, w, bias=None, stride=1, padding=0, mode=1, dx_half=False):
    """convolution whose three GEMMs (forward, data gradient, weight gradient) use narrowed operands; the bias and its
    gradient are fp32 (they are applied in the kernel epilogue / reduced by a separate fp32 kernel)"""
    y = _Conv2dNp.apply(x, w, stride, padding, mode, dx_half)
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    return y


def amp_conv2d(x, w, b, stride, padding, dx_half=False, per_sample=False):
    """The convolution arithmetic of the product's `--amp O1` path on the half-precision kernels (csrc/conv_h.hip; installed into
    oracle/fsv_oracle.py with `fsv_oracle.arithmetic(amp_conv2d)`): layers whose GEMMs fit those kernels - output channels a
    multiple of 8 (per-sample generated weights: input channels too) - run with operands rounded to IEEE half, exact products
    and fp32 sums (mode 1 above; the output stays fp32, the bias is added in fp32), every other convolution exactly.
    *Parity unpinned* against apex (not vendored, no CUDA): this is the definition the product is held to."""
    cout, cin = w.shape[0], w.shape[1]
    if cout % 8 != 0 or (per_sample and cin % 8 != 0):
        return F.conv2d(x, w, b, stride=stride, padding=padding)
    return conv2d(x, w, b, stride, padding, mode=1, dx_half=dx_half)


class LossScaler:
    """apex dynamic loss scaling, one instance per loss (`num_losses=2`, loss_id 0 = G, 1 = D; models/models.py:24-26)."""

    def __init__(self, init_scale=2.0 ** 16, window=2000, max_scale=2.0 ** 24, min_scale=1.0):
        self.scale, self.window, self.max_scale, self.min_scale = float(init_scale), int(window), float(max_scale), float(min_scale)
        self.good = 0

    def unscale_and_check(self, grads):
        """returns (found_inf, unscaled grads)"""
        found = any(not bool(torch.isfinite(g).all()) for g in grads)
        return found, [g / self.scale for g in grads]

    def update(self, found_inf):
        """returns True when the optimiser step must be skipped"""
        if found_inf:
            self.scale = max(self.scale * 0.5, self.min_scale)
            self.good = 0
            return True
        self.good += 1
        if self.good >= self.window:
            self.scale = min(self.scale * 2.0, self.max_scale)
            self.good = 0
        return False
