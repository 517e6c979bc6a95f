"""Per-GEMM-shape timing of one eager training step of the bench workload (MI355X only).

Every MFMA launch is bracketed with HIP events (few-shot-vid2vid_amd/profile.py, detail mode) and the launches are
grouped by kernel + GEMM shape, sorted by total time.  Shows which layer shapes the step's MFMA time goes to and how
far each is from the 157.3 TFLOP/s fp32-matrix peak.  Usage: python tools/shape_profile.py [--size 512] [--batch 2]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='pose', choices=sorted(bench.WORKLOADS))
    ap.add_argument('--amp', default='O0')
    ap.add_argument('--size', type=int, default=None)
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--top', type=int, default=60)
    ap.add_argument('--out', default='')
    ap.add_argument('--streams', type=int, default=0,
                    help='1: keep the branch streams on - a launch that shares the chip with another stream then reads long (the '
                         'K64 / K96 rows of round 3); default 0: one stream, every launch timed alone')
    args = ap.parse_args()
    bench.WORKLOAD, bench.AMP = args.workload, args.amp
    wl = bench.WORKLOADS[args.workload]
    args.size = wl['size'] if args.size is None else args.size
    args.batch = wl['batch'] if args.batch is None else args.batch
    from importlib import import_module
    import fsv2v_amd  # noqa: F401
    M = import_module('few-shot-vid2vid_amd.model')
    prof = import_module('few-shot-vid2vid_amd.profile')
    import_module('few-shot-vid2vid_amd.streams').ENABLED = bool(args.streams)
    dev = torch.device('cuda:0')
    opt = bench.build_opt(args.size, args.batch)
    model = M.create_model(opt).to(dev).train()
    opt_G, opt_D = model.build_optimizers()
    data = bench.make_data(args.batch, args.size, 1234, dev, opt)

    def step():
        M.loss_backward(opt, model(data, mode='discriminator'), opt_D, 1)
        g_losses, _, _ = model(data, mode='generator')
        M.loss_backward(opt, g_losses, opt_G, 0)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    prof.enable(detail=True)
    for _ in range(3):
        step()
    s = prof.summary()['by_kernel']
    prof.disable()
    rows = sorted(s.items(), key=lambda kv: -kv[1]['total_ms'])
    tot = sum(v['total_ms'] for _, v in rows) / 3
    print('MFMA kernels: %.2f ms / step over %d distinct shapes' % (tot, len(rows)))
    acc = 0.0
    for k, v in rows[:args.top]:
        acc += v['total_ms'] / 3
        print('%-82s n=%3d %8.1f us %6.1f TF/s %6.2f ms (cum %5.1f)' % (k, v['launches'] // 3, v['avg_us'], v['tflops'],
                                                                     v['total_ms'] / 3, acc))
    if args.out:
        with open(args.out, 'w') as f:
            for k, v in rows:
                f.write(json.dumps(dict(kernel=k, **v)) + '\n')


if __name__ == '__main__':
    main()
