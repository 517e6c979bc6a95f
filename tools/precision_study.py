"""CPU prediction of what the narrow-operand modes do to the generator's output: the oracle's generator forward (full channel
widths, random-init weights, synthetic pose inputs) with every convolution / linear operand narrowed as csrc/conv_np.hip does
(oracle/np_oracle.py definition), against the same forward in fp32 and fp64.  SPADE's fixed 1x1 gamma / beta convolutions stay
fp32 (the fused modulation kernel is not narrowed).  Prints relative errors of the generated image.

    python tools/precision_study.py [size=128] [ngf=32]
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import fsv_oracle as O        # noqa: E402
from oracle import np_oracle as NO        # noqa: E402
import model_checks as mc                 # noqa: E402
from importlib import import_module       # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ngf = int(sys.argv[2]) if len(sys.argv) > 2 else 32
os.environ['FSV2V_EMU'] = '1'             # only to import the package for its random-init weight shapes (no kernels run)
import fsv2v_amd  # noqa: F401,E402
M = import_module('few-shot-vid2vid_amd.model')
opt = mc.make_opt(fineSize=size, loadSize=size, batchSize=1, warp_ref=True, spade_combine=True, remove_face_labels=True, ngf=ngf,
                  no_vgg_loss=True, no_flow_gt=True, n_downsample_G=5 if size >= 128 else 3)
torch.manual_seed(0)
model = M.create_model(opt)
sd = {k: v.detach().clone() for k, v in model.netG.state_dict().items()}
cfg = O.cfg_from_opt(opt)
tl, ti, rl, ri = mc.synth_pose_inputs(1, size, size, 7)
lab = tl[:, 0]


def run(dtype, mode):
    conv0, lin0 = F.conv2d, F.linear

    def conv(x, w, b=None, stride=1, padding=0, *a, **k):
        if mode and x.shape[1] % 4 == 0:            # scalar-gather layers stay fp32 in the kernels as well
            y = 0
            for xa, wb in NO._pairs(NO.planes(x, mode), NO.planes(w, mode)):
                y = y + conv0(xa, wb, None, stride, padding)
            return y if b is None else y + b.view(1, -1, 1, 1)
        return conv0(x, w, b, stride, padding, *a, **k)

    def lin(x, w, b=None):
        if mode and x.shape[-1] % 4 == 0:
            y = 0
            for xa, wb in NO._pairs(NO.planes(x, mode), NO.planes(w, mode)):
                y = y + lin0(xa, wb)
            return y if b is None else y + b
        return lin0(x, w, b)
    F.conv2d, F.linear = conv, lin
    try:
        s = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        with torch.no_grad():
            out = O.generator_forward(s, cfg, lab.to(dtype), rl.to(dtype), ri.to(dtype))
    finally:
        F.conv2d, F.linear = conv0, lin0
    return out[0] if isinstance(out, (tuple, list)) else out


ref64 = run(torch.float64, 0)
for name, dtype, mode in (('fp32', torch.float32, 0), ('bf16x3', torch.float32, 2), ('f16', torch.float32, 1)):
    y = run(dtype, mode).double()
    d = (y - ref64)
    print('%-7s image vs fp64: max|diff| %.3e  rel L2 %.3e   (|image| max %.3f)' % (
        name, float(d.abs().max()), float(d.norm() / ref64.norm()), float(ref64.abs().max())), flush=True)
