"""Fused second stage of the BatchNorm-shaped reductions against the two-launch form, shared by the emulator test and - run as a
script - the first hardware run (where the last-ticket logic meets real concurrency: repeated launches, many workgroups)."""
import torch

import op_checks as oc


def check_bit_identical(device, shapes, reps=1, seed=91):
    ops, conv = oc.pkg()
    g = torch.Generator().manual_seed(seed)
    for (n, c, h, w) in shapes:
        x = torch.randn(n, c, h, w, generator=g)
        wt, b = torch.randn(c, generator=g), torch.randn(c, generator=g)
        dy = torch.randn(n, c, h, w, generator=g)
        ref = None
        for fused in (False,) + (True,) * reps:
            prev = ops.set_fused_final(fused)
            try:
                rm, rv = torch.zeros(c, device=device), torch.ones(c, device=device)
                xd, wd, bd = [t.clone().to(device).requires_grad_(True) for t in (x, wt, b)]
                y = ops.norm_act(xd, wd, bd, rm, rv, instance=False, act=conv.ACT_LRELU, training=True)
                y.backward(dy.to(device))
                cs = ops.colsum(conv.to_nhwc(x.to(device)), 1, n * h * w, c)
                got = [t.detach().cpu() for t in (y, xd.grad, wd.grad, bd.grad, rm, rv, cs)]
            finally:
                ops.set_fused_final(prev)
            if ref is None:
                ref = got
            else:
                for k, (a, bb) in enumerate(zip(ref, got)):
                    assert torch.equal(a, bb), ('shape', (n, c, h, w), 'output', k)


if __name__ == '__main__':
    dev = torch.device('cuda', 0)
    check_bit_identical(dev, [(2, 12, 9, 7), (3, 130, 5, 6), (1, 7, 33, 31), (4, 64, 16, 16), (2, 64, 256, 256), (2, 1024, 16, 16),
                              (2, 32, 512, 512), (2, 256, 64, 64)], reps=10)
    print('FUSED_FINAL_GPU_OK', flush=True)
