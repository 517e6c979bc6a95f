"""A few convolution launches at C3 layer shapes, for rocprofv3 --pmc runs (one launch per shape and direction)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fsv2v_amd  # noqa
from importlib import import_module
conv = import_module('few-shot-vid2vid_amd.conv')
dev = torch.device('cuda:0')
cases = [(2, 128, 256, 256, 64, 3, 1, 1), (2, 512, 64, 64, 256, 3, 1, 1), (2, 64, 512, 512, 32, 3, 1, 1), (2, 256, 128, 128, 128, 3, 1, 1)]
for (n, cin, h, w, cout, k, s, p) in cases:
    x = conv.to_nhwc(torch.randn(n, cin, h, w, device=dev)); wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    g = conv.Geom(k, k, s, p); oh, ow = g.out_hw(h, w)
    wf, kpad, ldw = conv.prep_weight(wt, 0, g)
    dy = conv.to_nhwc(torch.randn(n, cout, oh, ow, device=dev))
    for _ in range(2):
        conv.conv_forward(x, wf, ldw, cout, g)
        conv.conv_wgrad(x, dy, g, wt.shape)
    torch.cuda.synchronize()
