"""Per-kernel micro-benchmarks at the layer shapes of config C3 (pose 512x512, B=2).  GPU only.

Prints one JSON line per case: achieved TFLOP/s (2*MAC, dense-equivalent) or GB/s, against the gfx950 peaks
(157.3 TFLOP/s fp32 matrix, 8 TB/s HBM; /opt/skills/guides/MI355X_MICROARCH.md).
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fsv2v_amd  # noqa
from importlib import import_module

conv = import_module('few-shot-vid2vid_amd.conv')
ops = import_module('few-shot-vid2vid_amd.ops')

dev = torch.device('cuda:0')


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def conv_case(name, n, cin, h, w, cout, k, s, p, tile=-1, split=0, which=('fwd', 'dgrad', 'wgrad')):
    x = conv.to_nhwc(torch.randn(n, cin, h, w, device=dev))
    wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    g = conv.Geom(k, k, s, p)
    oh, ow = g.out_hw(h, w)
    flops = 2.0 * n * oh * ow * cout * cin * k * k
    wf, kpad, ldw = conv.prep_weight(wt, 0, g)
    dy = conv.to_nhwc(torch.randn(n, cout, oh, ow, device=dev))
    res = {}
    if 'fwd' in which:
        t = timeit(lambda: conv.conv_forward(x, wf, ldw, cout, g, force_tile=tile, force_split=split))
        res['fwd_tflops'] = round(flops / t / 1e12, 2)
        res['fwd_us'] = round(t * 1e6, 1)
    if 'dgrad' in which:
        t = timeit(lambda: conv.conv_dgrad(dy, wt, g, (h, w)))
        res['dgrad_tflops'] = round(flops / t / 1e12, 2)
        res['dgrad_us'] = round(t * 1e6, 1)
    if 'wgrad' in which:
        t = timeit(lambda: conv.conv_wgrad(x, dy, g, wt.shape))
        res['wgrad_tflops'] = round(flops / t / 1e12, 2)
        res['wgrad_us'] = round(t * 1e6, 1)
    res.update(case=name, gflop=round(flops / 1e9, 2), tile=tile, split=split)
    print(json.dumps(res), flush=True)


def spade_case(name, n, c, ch, h, w, nmaps):
    x = conv.to_nhwc(torch.randn(n, c, h, w, device=dev))
    maps = [conv.to_nhwc(torch.randn(n, ch, h, w, device=dev)) for _ in range(nmaps)]
    weights = []
    for k in range(nmaps):
        if k == 0:
            weights.append((torch.randn(n, c, ch, 1, 1, device=dev) * 0.1, torch.randn(n, c, ch, 1, 1, device=dev) * 0.1,
                            torch.randn(n, c, device=dev) * 0.1, torch.randn(n, c, device=dev) * 0.1))
        else:
            weights.append((torch.randn(c, ch, 1, 1, device=dev) * 0.1, torch.randn(c, ch, 1, 1, device=dev) * 0.1,
                            torch.randn(c, device=dev) * 0.1, torch.randn(c, device=dev) * 0.1))
    flops = 2.0 * n * h * w * c * ch * 2 * nmaps
    with torch.no_grad():
        t = timeit(lambda: ops.spade_mod(x, maps, weights))
    print(json.dumps(dict(case=name, gflop=round(flops / 1e9, 2), us=round(t * 1e6, 1),
                          tflops=round(flops / t / 1e12, 2))), flush=True)


def ew_case(name, fn, nbytes):
    t = timeit(fn)
    print(json.dumps(dict(case=name, us=round(t * 1e6, 1), gbps=round(nbytes / t / 1e9, 1))), flush=True)


def main():
    B = 2
    print(json.dumps(dict(device=torch.cuda.get_device_name(0))), flush=True)
    conv_case('up0.conv0 64->32 @512', B, 64, 512, 512, 32, 3, 1, 1)
    conv_case('up0.conv0 64->32 @512 tile2', B, 64, 512, 512, 32, 3, 1, 1, tile=2, which=('fwd',))
    conv_case('up0.conv1 32->32 @512', B, 32, 512, 512, 32, 3, 1, 1)
    conv_case('up0.convs 64->32 1x1 @512', B, 64, 512, 512, 32, 1, 1, 0)
    conv_case('up1.conv0 128->64 @256', B, 128, 256, 256, 64, 3, 1, 1)
    conv_case('up2.conv0 256->128 @128', B, 256, 128, 128, 128, 3, 1, 1)
    conv_case('up3.conv0 512->256 @64', B, 512, 64, 64, 256, 3, 1, 1)
    conv_case('up4.conv0 1024->512 @32', B, 1024, 32, 32, 512, 3, 1, 1)
    conv_case('up5.conv0 1024->1024 @16', B, 1024, 16, 16, 1024, 3, 1, 1)
    conv_case('up5.conv0 1024->1024 @16 nosplit', B, 1024, 16, 16, 1024, 3, 1, 1, split=1, which=('fwd',))
    conv_case('flow.res 256->256 @64', B, 256, 64, 64, 256, 3, 1, 1)
    conv_case('enc.down 64->128 s2 @256', B, 64, 256, 256, 128, 3, 2, 1)
    conv_case('D.first 20->32 k4s2 @512 (2B)', 2 * B, 20, 512, 512, 32, 4, 2, 2)
    conv_case('D.l2 64->128 k4s2 @129', 2 * B, 64, 129, 129, 128, 4, 2, 2)
    conv_case('embed.first 6->32 @512', B, 6, 512, 512, 32, 3, 1, 1)
    conv_case('conv_img 32->3 @512', B, 32, 512, 512, 3, 3, 1, 1)
    conv_case('fc 1024->1024 rows2048', 1, 1024, 1, 2048, 1024, 1, 1, 0)
    spade_case('spade L0 bn0 C64 ch32 x3maps @512', B, 64, 32, 512, 512, 3)
    spade_case('spade L0 bn1 C32 ch32 x3maps @512', B, 32, 32, 512, 512, 3)
    spade_case('spade L2 bn0 C256 ch128 @128', B, 256, 128, 128, 128, 1)
    spade_case('spade L3 bn0 C512 ch256 @64', B, 512, 256, 64, 64, 1)
    # HBM-bound helpers
    x = conv.to_nhwc(torch.randn(B, 32, 512, 512, device=dev))
    nb = x.numel() * 4
    ew_case('norm_stats C32 @512', lambda: ops.norm_stats(x, 1, B * 512 * 512, 32, 1e-5), nb)
    with torch.no_grad():
        ew_case('norm_act C32 @512 (stats+apply)', lambda: ops.norm_act(x, act=1), nb * 3)
        xs = conv.to_nhwc(torch.randn(B, 64, 256, 256, device=dev))
        ew_case('upsample2x C64 256->512', lambda: ops.upsample2x(xs), xs.numel() * 4 * 5)
        img = torch.randn(B, 3, 512, 512, device=dev)
        flow = torch.randn(B, 2, 512, 512, device=dev) * 8
        ew_case('warp 3ch @512 (32 B/px algorithmic)', lambda: ops.resample(img, flow), B * 512 * 512 * 32)
    p = torch.randn(100_000_000, device=dev)
    g = torch.randn_like(p); m = torch.zeros_like(p); v = torch.zeros_like(p)
    st = torch.tensor([0., 0., 0., 2e-4], device=dev)
    ew_case('adam 100M params', lambda: ops.adam_step(p, g, m, v, st, 0.0, 0.999, 1e-8), p.numel() * 4 * 7)


if __name__ == '__main__':
    main()
