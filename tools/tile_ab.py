"""A/B of conv tile / split-K configurations on C3 layer shapes at B=2 (interleaved rounds in one process)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fsv2v_amd  # noqa
from importlib import import_module
conv = import_module('few-shot-vid2vid_amd.conv')
dev = torch.device('cuda:0')
cfgs = [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (1, 2), (4, 0), (4, 1)]
shapes = [('up2 256->128@128', 2, 256, 128, 128, 128, 3), ('up3 512->256@64', 2, 512, 64, 64, 256, 3),
          ('up4 1024->512@32', 2, 1024, 32, 32, 512, 3), ('up5 1024->1024@16', 2, 1024, 16, 16, 1024, 3),
          ('flow 256->256@64', 2, 256, 64, 64, 256, 3), ('fc 1024->1024 r2048', 1, 1024, 1, 2048, 1024, 1),
          ('emb 512->256@64 (cat)', 2, 512, 64, 64, 128, 3)]
for name, n, cin, h, w, cout, k in shapes:
    x = conv.to_nhwc(torch.randn(n, cin, h, w, device=dev)); wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    g = conv.Geom(k, k, 1, k // 2)
    wf, kpad, ldw = conv.prep_weight(wt, 0, g)
    flops = 2.0 * n * h * w * cout * cin * k * k
    res = {c: [] for c in cfgs}
    for rnd in range(3):
        for c in cfgs:
            f = lambda: conv.conv_forward(x, wf, ldw, cout, g, force_tile=c[0], force_split=c[1])
            f(); f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            res[c].append(flops / (e0.elapsed_time(e1) / 5 * 1e-3) / 1e12)
    print(json.dumps({'case': name, **{'t%d/s%d' % c: round(sorted(v)[len(v) // 2], 1) for c, v in res.items()}}), flush=True)
