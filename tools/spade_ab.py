"""In-box timing of the SPADE modulation kernels (forward and backward twin) at the shapes of a workload, isolated, warm:
python tools/spade_ab.py [--f16 1] ; FSV_SPADE_DBG bits switch parts of the kernels off (csrc/spade.hip)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

SHAPES = [  # (n, c, ch list, h, w, up)      street --amp: one map everywhere
    (1, 64, [32], 512, 1024, 1), (1, 32, [32], 512, 1024, 0), (1, 128, [64], 256, 512, 1), (1, 256, [128], 128, 256, 1),
    (1, 512, [256], 64, 128, 1), (1, 1024, [512], 32, 64, 1), (1, 1024, [1024], 16, 32, 0),
    (2, 64, [32, 32, 32], 512, 512, 1), (2, 128, [64, 64, 64], 256, 256, 1),          # pose: three maps
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--f16', type=int, default=1)
    ap.add_argument('--reps', type=int, default=20)
    args = ap.parse_args()
    from importlib import import_module
    import fsv2v_amd  # noqa: F401
    ops = import_module('few-shot-vid2vid_amd.ops')
    conv = import_module('few-shot-vid2vid_amd.conv')
    dev = torch.device('cuda:0')
    conv.set_mfma_mode(1 if args.f16 else 0)
    g = torch.Generator().manual_seed(1)
    for (n, c, chs, h, w, up) in SHAPES:
        xs = (h // 2, w // 2) if up else (h, w)
        x = torch.randn(n, c, *xs, generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        maps = [torch.randn(n, ch, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
                for ch in chs]
        ws = [tuple((torch.randn(*s, generator=g) * 0.1).to(dev).requires_grad_(True)
                    for s in ((n, c, ch, 1, 1), (n, c, ch, 1, 1), (n, c), (n, c))) for ch in chs]
        lib = import_module('few-shot-vid2vid_amd.lib')
        times = {}
        real = lib.call

        def timed(name, *a):
            if name.startswith('fsv_spade_mod'):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    rc = real(name, *a)
                e1.record()
                torch.cuda.synchronize()
                times[name] = e0.elapsed_time(e1) * 1e3 / args.reps
                return rc
            return real(name, *a)
        y = ops.spade_mod(x, maps, ws, None, None, act=conv.ACT_LRELU, up=bool(up))           # warm (and layout caches)
        y.backward(torch.ones_like(y))
        lib.call = timed
        try:
            y = ops.spade_mod(x, maps, ws, None, None, act=conv.ACT_LRELU, up=bool(up))
            y.backward(torch.ones_like(y))
        finally:
            lib.call = real
        px = n * h * w
        fb = px * c * ((1 if up else 4) + 2) + sum(px * ch * 2 for ch in chs) if args.f16 else 0
        print('P%-7d C%-5d K%-12s up%d  ' % (px, c, '+'.join(map(str, chs)), up) +
              '  '.join('%s %7.1f us' % (k.replace('fsv_spade_mod_', ''), v) for k, v in sorted(times.items())) +
              ('   fwd floor %.0f us' % (fb / 8e6) if fb else ''), flush=True)


if __name__ == '__main__':
    main()
