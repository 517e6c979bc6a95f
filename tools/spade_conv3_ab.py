"""In-box A/B of the 3x3 half of a SPADEResnetBlock, `conv_0(actvn(bn_0(x, maps)))` / `conv_1(actvn(bn_1(dx, maps))) + x_s`
(architecture.py:96-99), isolated and warm, at the widths the fused kernel covers in the two bench workloads:
    two launches    fsv_spade_mod_fwd (writes the modulated tensor) + fsv_conv_gather_fwd[_stats] (3x3, reads it back nine times
                    through L2; the statistics of the output for the next normalisation come from its epilogue)
    one launch      fsv_spade_conv3_fwd (csrc/spade_conv3.hip), without / with the modulated tensor as a side output (statistics of
                    the output from its epilogue as well)
python tools/spade_conv3_ab.py [--reps 20]     -> one JSON line per shape (microseconds per call, median of 5 rounds)"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

SHAPES = [  # (tag, n, c, cout, ch list, h, w, up, res)
    ('pose level 0 conv_0 (64 -> 32)', 2, 64, 32, [32, 32], 512, 512, 1, 0),
    ('pose level 1 conv_1 (64 -> 64) + x_s', 2, 64, 64, [64, 64], 256, 256, 0, 1),
    ('street level 0 conv_0 (64 -> 32)', 1, 64, 32, [32], 512, 1024, 1, 0),
    ('street level 1 conv_1 (64 -> 64) + x_s', 1, 64, 64, [64], 256, 512, 0, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--only', type=int, default=-1, help='one shape of the list (counter passes)')
    args = ap.parse_args()
    from importlib import import_module
    import fsv2v_amd  # noqa: F401
    ops = import_module('few-shot-vid2vid_amd.ops')
    conv = import_module('few-shot-vid2vid_amd.conv')
    lib = import_module('few-shot-vid2vid_amd.lib')
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(1)
    watch = ('fsv_spade_mod_fwd', 'fsv_conv_gather_fwd', 'fsv_conv_gather_fwd_stats', 'fsv_spade_conv3_fwd')
    for (tag, n, c, cout, chs, h, w, up, res) in (SHAPES if args.only < 0 else SHAPES[args.only:args.only + 1]):
        xs = (h // 2, w // 2) if up else (h, w)
        cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
        x = cl(torch.randn(n, c, *xs, generator=g))
        maps = [cl(torch.randn(n, ch, h, w, generator=g)) for ch in chs]
        ws = [tuple((torch.randn(*s, generator=g) * 0.1).to(dev)
                    for s in (((n, c, ch, 1, 1), (n, c, ch, 1, 1), (n, c), (n, c)) if k == 0 else
                              ((c, ch, 1, 1), (c, ch, 1, 1), (c,), (c,)))) for k, ch in enumerate(chs)]
        wc = (torch.randn(cout, c, 3, 3, generator=g) * (1.0 / (9 * c) ** 0.5)).to(dev)
        bc = (torch.randn(cout, generator=g) * 0.1).to(dev)
        rs = cl(torch.randn(n, cout, h, w, generator=g)) if res else None
        real = lib.call
        out = dict(shape=tag, pixels=n * h * w, C=c, Cout=cout, maps=chs)

        def run(fused, grad, stats):
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
            os.environ['FSV_SPADE_CONV3'] = '1' if fused else '0'
            xx = x.clone().requires_grad_(grad)
            wcc = wc.clone().requires_grad_(grad)
            for timing in (False, True):
                lib.call = timed if timing else real
                try:
                    with (torch.enable_grad() if grad else torch.no_grad()):
                        with ops.spade_into_conv(conv3=True):
                            hm = ops.spade_mod(xx, maps, ws, None, None, act=conv.ACT_LRELU, up=bool(up))
                            y = ops.conv2d(hm, wcc, bc, 1, 1, res=rs, stats_groups=1 if stats else 0)
                finally:
                    lib.call = real
            os.environ.pop('FSV_SPADE_CONV3', None)
            return times, y
        t2, y2 = run(False, False, True)
        t1, y1 = run(True, False, True)
        t1g, _ = run(True, True, True)
        os.environ['FSV_S3_RW'] = '32'                   # 32-row weight chunks in phase 2 (Cout 32: 64 by default)
        t1w, _ = run(True, False, True)
        os.environ.pop('FSV_S3_RW', None)
        out['two_launches_us'] = {k.replace('fsv_', ''): round(v, 1) for k, v in t2.items()}
        out['two_launches_total_us'] = round(sum(t2.values()), 1)
        out['fused_us'] = round(t1.get('fsv_spade_conv3_fwd', float('nan')), 1)
        out['fused_rw32_us'] = round(t1w.get('fsv_spade_conv3_fwd', float('nan')), 1)
        out['fused_with_side_output_us'] = round(t1g.get('fsv_spade_conv3_fwd', float('nan')), 1)
        out['max_rel_diff'] = float((y1 - y2).abs().max() / y2.abs().max())
        px = n * h * w
        flop = 2.0 * px * c * 2 * sum(chs) + 2.0 * px * 9 * c * cout
        out['gflop_reference'] = round(flop / 1e9, 2)
        out['fused_tflops_of_reference_work'] = round(flop / (out['fused_us'] * 1e-6) / 1e12, 1)
        out['two_launches_tflops'] = round(flop / (out['two_launches_total_us'] * 1e-6) / 1e12, 1)
        base = px * c * 4 // (4 if up else 1) + sum(px * ch * 4 for ch in chs) + px * cout * 4 * (2 if res else 1)
        out['two_launches_bytes_MB'] = round((base + 2 * px * c * 4) / 1e6, 1)
        out['fused_bytes_MB'] = round(base / 1e6, 1)
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
