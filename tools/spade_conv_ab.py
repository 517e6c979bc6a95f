"""In-box A/B of the shortcut branch of a SPADEResnetBlock, `x_s = conv_s(bn_s(x, maps))` (architecture.py:103-108), isolated and warm,
at the widths of the two bench workloads:
    two launches    fsv_spade_mod_fwd (writes the modulated tensor) + fsv_conv_gather_fwd (1x1, reads it back)
    one launch      fsv_spade_conv_s_fwd (csrc/spade_conv.hip), without / with the modulated tensor as a side output
python tools/spade_conv_ab.py [--reps 20]     -> one JSON line per shape (microseconds per call, median of 5 rounds)"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

SHAPES = [  # (tag, n, c, cout, ch list, h, w, up)
    ('pose level 0', 2, 64, 32, [32, 32, 32], 512, 512, 1),
    ('pose level 1', 2, 128, 64, [64, 64, 64], 256, 256, 1),
    ('street level 0', 1, 64, 32, [32], 512, 1024, 1),
    ('street level 1', 1, 128, 64, [64], 256, 512, 1),
    ('pose level 0, one map', 2, 64, 32, [32], 512, 512, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--amp', type=int, default=0, help='1: the --amp arithmetic (half maps / weights, f16 GEMMs)')
    args = ap.parse_args()
    from importlib import import_module
    import fsv2v_amd  # noqa: F401
    ops = import_module('few-shot-vid2vid_amd.ops')
    conv = import_module('few-shot-vid2vid_amd.conv')
    lib = import_module('few-shot-vid2vid_amd.lib')
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(1)
    watch = ('fsv_spade_mod_fwd', 'fsv_conv_gather_fwd', 'fsv_spade_conv_s_fwd', 'fsv_spade_mod_fwd_h', 'fsv_hconv_gather',
             'fsv_spade_conv_s_fwd_h')
    conv.set_mfma_mode(1 if args.amp else 0)
    for (tag, n, c, cout, chs, h, w, up) in SHAPES:
        xs = (h // 2, w // 2) if up else (h, w)
        cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
        x = cl(torch.randn(n, c, *xs, generator=g))
        maps = [cl(torch.randn(n, ch, h, w, generator=g)) for ch in chs]
        ws = [tuple((torch.randn(*s, generator=g) * 0.1).to(dev)
                    for s in (((n, c, ch, 1, 1), (n, c, ch, 1, 1), (n, c), (n, c)) if k == 0 else
                              ((c, ch, 1, 1), (c, ch, 1, 1), (c,), (c,)))) for k, ch in enumerate(chs)]
        wc = (torch.randn(cout, c, 1, 1, generator=g) * (1.0 / c ** 0.5)).to(dev)
        real = lib.call
        out = dict(shape=tag, pixels=n * h * w, C=c, Cout=cout, maps=chs)

        def run(fused, grad):
            times = {}

            def timed(name, *a):
                if name in watch:
                    samples = []
                    for _ in range(args.rounds):
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(args.reps):
                            rc = real(name, *a)
                        e1.record()
                        torch.cuda.synchronize()
                        samples.append(e0.elapsed_time(e1) * 1e3 / args.reps)
                    times[name] = sorted(samples)[len(samples) // 2]
                    return rc
                return real(name, *a)
            os.environ['FSV_SPADE_CONV_S'] = '1' if fused else '0'
            xx = x.clone().requires_grad_(grad)
            wcc = wc.clone().requires_grad_(grad)
            for timing in (False, True):
                lib.call = timed if timing else real
                try:
                    with (torch.enable_grad() if grad else torch.no_grad()):
                        with ops.spade_into_conv():
                            hs = ops.spade_mod(xx, maps, ws, None, None, act=conv.ACT_NONE, up=bool(up))
                            y = ops.conv2d(hs, wcc, None, 1, 0)
                finally:
                    lib.call = real
            os.environ.pop('FSV_SPADE_CONV_S', None)
            return times, y
        t2, y2 = run(False, False)
        t1, y1 = run(True, False)
        t1g, _ = run(True, True)
        out['two_launches_us'] = {k.replace('fsv_', ''): round(v, 1) for k, v in t2.items()}
        out['two_launches_total_us'] = round(sum(t2.values()), 1)
        fk = 'fsv_spade_conv_s_fwd_h' if args.amp else 'fsv_spade_conv_s_fwd'
        out['arithmetic'] = 'amp O1 (f16 GEMMs)' if args.amp else 'fp32'
        out['fused_us'] = round(t1.get(fk, float('nan')), 1)
        out['fused_with_side_output_us'] = round(t1g.get(fk, float('nan')), 1)
        out['max_rel_diff'] = float((y1 - y2).abs().max() / y2.abs().max())
        # algorithmic HBM bytes: x (a quarter of the pixels when the up-sampling is folded in) + maps + x_s (+ the modulated tensor)
        px = n * h * w
        base = px * c * 4 // (4 if up else 1) + sum(px * ch * 4 for ch in chs) + px * cout * 4
        out['fused_GBps'] = round(base / (out['fused_us'] * 1e-6) / 1e9, 1)
        out['two_launches_bytes_MB'] = round((base + 2 * px * c * 4) / 1e6, 1)
        out['fused_bytes_MB'] = round(base / 1e6, 1)
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
