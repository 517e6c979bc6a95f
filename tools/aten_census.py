"""Census of the torch-native (aten) operators one training iteration issues next to the library's own launches.

Every aten op on a device tensor is a launch (add / copy_ / fill_ / zeros ...) that does no work of the hot path: gradient
accumulation of multi-consumer tensors, layout conversions, zero fills, scalar loss arithmetic.  This tool runs one D+G
iteration of the product model on the SIMT emulator (CPU) under a TorchDispatchMode and prints the ops grouped by the innermost
frame of this package that issued them, so that they can be folded into kernels one source at a time.

    FSV2V_EMU=1 python tools/aten_census.py [--size 64] [--top 60]
"""
import argparse
import collections
import os
import sys
import traceback

os.environ.setdefault('FSV2V_EMU', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch
from torch.utils._python_dispatch import TorchDispatchMode

SKIP = ('aten.view', 'aten._unsafe_view', 'aten.reshape', 'aten.permute', 'aten.transpose', 'aten.t.', 'aten.expand',
        'aten.slice', 'aten.select', 'aten.narrow', 'aten.as_strided', 'aten.detach', 'aten.alias', 'aten.unsqueeze',
        'aten.squeeze', 'aten.empty', 'aten.split', 'aten.unbind', 'aten.is_', 'aten.sym_', 'aten.stride', 'aten.size',
        'aten._local_scalar_dense', 'aten.lift_fresh', 'aten.set_', 'aten.resize_', 'aten.unfold', 'aten.item', 'aten.chunk',
        'aten.new_empty', 'aten.result_type', 'aten.can_cast', 'aten._to_copy')      # (_to_copy: host->device uploads of scalars show separately)


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()
        self.phase = '?'
        self.shapes = False

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not name.startswith(SKIP):
            numel = 0
            for a in list(args) + [out]:
                if torch.is_tensor(a):
                    numel = max(numel, a.numel())
            where = 'autograd engine'
            for fr in reversed(traceback.extract_stack()[:-1]):
                if 'few-shot-vid2vid_amd' in fr.filename and 'aten_census' not in fr.filename:
                    where = '%s:%d %s' % (os.path.basename(fr.filename), fr.lineno, fr.name)
                    break
            if where == 'autograd engine' or 'loss_backward' in where:
                node = torch._C._current_autograd_node()
                where = 'engine, node %s' % (type(node).__name__ if node is not None else 'none (gradient accumulation)')
            kind = 'scalar' if numel <= 8 else 'tensor'
            if self.shapes and kind == 'tensor':
                kind = 'x'.join(str(d) for d in (out.shape if torch.is_tensor(out) else ()))
            self.rows[(self.phase, name, where, kind)] += 1
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=64)
    ap.add_argument('--top', type=int, default=80)
    ap.add_argument('--shapes', action='store_true', help='key tensor ops by output shape')
    a = ap.parse_args()
    import model_checks as mc
    M = mc._model()
    opt = mc.tiny_opt(ngf=8, ndf=8, nff=8, warp_ref=True, spade_combine=True, remove_face_labels=True, fineSize=a.size,
                      loadSize=a.size)
    model = M.create_model(opt)
    model.train()
    opt_G, opt_D = model.build_optimizers()
    tl, ti, rl, ri = mc.synth_pose_inputs(1, a.size, a.size, 3, opt.input_nc)
    data = [tl, ti, [None, None], [None, None], rl, ri, None, None, None]

    def iteration(c=None):
        def ph(p):
            if c is not None:
                c.phase = p
        ph('D fwd'); d = model(data, mode='discriminator')
        ph('D bwd'); M.loss_backward(opt, d, opt_D, 1)
        ph('G fwd'); g, gen, prev = model(data, save_images=False, mode='generator')
        ph('G bwd'); M.loss_backward(opt, g, opt_G, 0)
    iteration(); iteration()           # settle caches
    c = Census()
    c.shapes = a.shapes
    with c:
        iteration(c)
    total = sum(c.rows.values())
    print('aten ops in one iteration: %d' % total)
    by_op = collections.Counter()
    for (ph, name, where, kind), n in c.rows.items():
        by_op[name] += n
    for name, n in by_op.most_common(25):
        print('  %5d  %s' % (n, name))
    print()
    for (ph, name, where, kind), n in c.rows.most_common(a.top):
        print('%4d  %-6s %-28s %-14s %s' % (n, ph, name, kind, where))


if __name__ == '__main__':
    main()
