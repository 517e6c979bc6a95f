"""Census of the torch-native (aten) operations one training step issues besides the library's own kernels, by call site.

Runs anywhere: on the GPU box against the real library, in the build container against the emulated one (FSV2V_EMU=1; a
reduced network - the op census per layer is the same, the layer count is not).  Every aten op that touches tensor data is
one or more device launches (copy_ = copyBuffer or an elementwise kernel, zero_ / fill_ = fillBuffer, add / mul / cat ...).
Usage: python tools/aten_census.py [--workload street] [--amp O1] [--tiny]
"""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

VIEW_OPS = ('view', 'reshape', 'permute', 'transpose', 'expand', 'slice', 'select', 'narrow', 'unsqueeze', 'squeeze', 'as_strided',
            'detach', 'alias', 't.default', 'unbind', 'split', 'chunk', '_unsafe_view', 'empty', 'sym_', 'stride', 'size',
            'is_', 'unfold', 'movedim', 'lift_fresh', 'set_', 'resize_', '_local_scalar_dense', 'item', 'new_empty', 'empty_like',
            'empty_strided', 'contiguous', 'clone', '_to_copy', 'to.')


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.ops = collections.Counter()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace('aten.', '')
        base = name.split('.')[0]
        is_view = any(base.startswith(v.split('.')[0]) and not base.endswith('_copy') for v in VIEW_OPS
                      if v not in ('contiguous', 'clone', '_to_copy', 'to.'))
        if not is_view:
            self.ops[name] += 1
            st = traceback.extract_stack()
            fr = [f for f in st if 'few-shot-vid2vid_amd' in f.filename]
            key = name + '  @ ' + ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in fr[-3:][::-1])
            self.sites[key] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='pose')
    ap.add_argument('--amp', default='O0')
    ap.add_argument('--tiny', action='store_true', help='reduced widths / resolution (the emulator)')
    ap.add_argument('--top', type=int, default=60)
    args = ap.parse_args()
    import bench
    bench.WORKLOAD, bench.AMP = args.workload, args.amp
    wl = bench.WORKLOADS[args.workload]
    from importlib import import_module
    import fsv2v_amd  # noqa: F401
    M = import_module('few-shot-vid2vid_amd.model')
    lib = import_module('few-shot-vid2vid_amd.lib')
    dev = torch.device('cpu') if lib.is_emu() else torch.device('cuda:0')
    size, batch = (64, 1) if args.tiny else (wl['size'], wl['batch'])
    opt = bench.build_opt(size, batch)
    if args.tiny:
        opt.ngf = opt.ndf = opt.nff = 16
        opt.n_downsample_G, opt.n_adaptive_layers = 3, 2
    model = M.create_model(opt).to(dev).train()
    opt_G, opt_D = model.build_optimizers()
    data = bench.make_data(batch, size, 1234, dev, opt)

    def step():
        M.loss_backward(opt, model(data, mode='discriminator'), opt_D, 1)
        g_losses, _, _ = model(data, mode='generator')
        M.loss_backward(opt, g_losses, opt_G, 0)
    for _ in range(2):
        step()
    c = Census()
    with c:
        step()
    print('aten ops that launch work: %d per step' % sum(c.ops.values()))
    for k, v in c.ops.most_common(30):
        print('%5d  %s' % (v, k))
    print('--- by site')
    for k, v in c.sites.most_common(args.top):
        print('%5d  %s' % (v, k))


if __name__ == '__main__':
    main()
