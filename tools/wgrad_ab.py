"""A/B of weight-gradient GEMM tiles / pixel-split factors on layer shapes of the bench step (B=2).

Each configuration: hipGraph of 20 launches (zero-fill + GEMM as the product path issues them), interleaved replays,
median of 5.  Columns: tile (auto / 64x64 / 128x64 / 64x128) x target workgroup count (split = target / tiles)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fsv2v_amd  # noqa
from importlib import import_module
conv = import_module('few-shot-vid2vid_amd.conv')
dev = torch.device('cuda:0')
shapes = [('Kdim2304 N256 pix8192', 2, 256, 64, 64, 256, 3), ('Kdim1152 N256 pix8192', 2, 128, 64, 64, 256, 3),
          ('Kdim576 N128 pix32768', 2, 64, 128, 128, 128, 3), ('Kdim2304 N128 pix32768', 2, 256, 128, 128, 128, 3),
          ('Kdim9216 N512 pix2048', 2, 1024, 32, 32, 512, 3), ('Kdim4608 N1024 pix512', 2, 512, 16, 16, 1024, 3),
          ('Kdim2304 N512 pix2048', 2, 256, 32, 32, 512, 3), ('Kdim288 N64 pix131072', 2, 32, 256, 256, 64, 3),
          ('Kdim1152 N64 pix131072', 2, 128, 256, 256, 64, 3), ('Kdim512 N512 pix1024', 1, 512, 1, 1024, 512, 1)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if any(a in s[0] for a in sys.argv[1:])]
TILES = {0: None, 1: (64, 64), 2: (128, 64), 3: (64, 128), 4: (128, 128)}
NREP = 20
for name, n, cin, h, w, cout, k in shapes:
    x = conv.to_nhwc(torch.randn(n, cin, h, w, device=dev))
    g = conv.Geom(k, k, 1, k // 2)
    dy = conv.to_nhwc(torch.randn(n, cout, h, w, device=dev))
    flops = 2.0 * n * h * w * cout * cin * k * k
    kdim, pch = k * k * cin, (n * h * w + 31) // 32
    cfgs = [(0, 0)]
    for t in (1, 2, 3, 4):
        bm, bn = TILES[t]
        if cout < bn:
            continue
        tiles = ((kdim + bm - 1) // bm) * ((cout + bn - 1) // bn)
        for target in (1024, 2048, 4096):
            sp = max(1, min((target + tiles - 1) // tiles, pch // 4))
            cfgs.append((t, sp))
    graphs = {}
    for c in cfgs:
        f = lambda: conv.conv_wgrad(x, dy, g, (cout, cin, k, k), raw=True, force_tile=c[0], force_split=c[1])
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            f(); f()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(NREP):
                f()
        graphs[c] = gr
    res = {c: [] for c in graphs}
    for rnd in range(5):
        for c, gr in graphs.items():
            gr.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            res[c].append(flops / (e0.elapsed_time(e1) / NREP * 1e-3) / 1e12)
    print(json.dumps({'case': name, **{('auto' if c[0] == 0 else '%dx%d/s%d' % (TILES[c[0]] + (c[1],))): round(sorted(v)[len(v) // 2], 1)
                                       for c, v in res.items()}}), flush=True)
