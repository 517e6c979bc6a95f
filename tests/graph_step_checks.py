"""GraphedIteration (few-shot-vid2vid_amd/graph_step.py) against the plain eager loop of train.py:58-62: same losses, images and
weights over several iterations with changing data and changing learning rate.  Under the emulator the "graph" is an eager
re-run on the static buffers (buffer plumbing only); run as a script on a GPU it captures and replays real hipGraphs."""
import torch

import model_checks as mc


def _run(device, graphed, iters, seed, opt_kw, b=1, split=False):
    from importlib import import_module
    M = mc._model()
    gs = import_module('few-shot-vid2vid_amd.graph_step')
    opt = mc.tiny_opt(**opt_kw)
    model = M.create_model(opt)
    mc.fill_state(model.netG); mc.fill_state(model.netD)
    model = model.to(device).train()
    opt_G, opt_D = model.build_optimizers(split_backward=split)
    step = gs.GraphedIteration(model, opt, warmup=2) if graphed else None
    h, w = int(opt.fineSize / opt.aspect_ratio), opt.fineSize
    log = []
    for it in range(iters):
        tl, ti, rl, ri = mc.synth_pose_inputs(b, h, w, seed + it, opt.input_nc)
        data = [tl, ti, [None, None], [None, None], rl, ri, None, None, None]
        if it == iters - 2:
            opt_G.set_lr(1e-5); opt_D.set_lr(3e-5)               # the device-side lr must reach a replayed graph
        if graphed:
            d, g, gen, prev = step(data, save_images=True)
        else:
            data = [None if t is None else ([x if x is None else x.to(device) for x in t] if isinstance(t, list) else t.to(device))
                    for t in data]
            d = M.loss_backward(opt, model(data, mode='discriminator'), opt_D, 1)
            g, gen, prev = model(data, save_images=True, mode='generator')
            g = M.loss_backward(opt, g, opt_G, 0)
        log.append(dict(d=[float(x.detach()) for x in d], g=[float(x.detach()) for x in g if not isinstance(x, int)],
                        img=gen[0].detach().clone().cpu()))
    return log, opt_G.flat_p.detach().clone().cpu(), opt_D.flat_p.detach().clone().cpu(), step


def check_graphed_iteration(device, iters=5, seed=500, tol=0.0):
    kw = dict(warp_ref=True, spade_combine=True, remove_face_labels=True, fineSize=32, loadSize=32, n_downsample_G=3, n_adaptive_layers=2)
    ref, pG, pD, _ = _run(device, False, iters, seed, kw)
    got, qG, qD, step = _run(device, True, iters, seed, kw)
    assert len(step.entries) == 1
    for it, (a, b) in enumerate(zip(ref, got)):
        for k in ('d', 'g'):
            for x, y in zip(a[k], b[k]):
                assert abs(x - y) <= tol * max(abs(x), 1.0) + 0.0, (it, k, a[k], b[k])
        assert float((a['img'] - b['img']).abs().max()) <= tol * 2.0 + 0.0, it
    assert float((pG - qG).abs().max()) <= tol and float((pD - qD).abs().max()) <= tol
    return step


def check_split_backward_single_rank(device, iters=3, seed=520):
    """build_optimizers(split_backward=True) WITHOUT a gradient exchange: the generator's forward pass still detaches at its
    stage boundary, so both drivers (the eager loss_backward and GraphedIteration) have to run the second backward piece -
    weights equal to the unsplit loop bit for bit (round-2 advisor finding: the graphed driver dropped every stage-1 gradient)."""
    kw = dict(warp_ref=True, spade_combine=True, remove_face_labels=True, fineSize=32, loadSize=32, n_downsample_G=3, n_adaptive_layers=2)
    _, pG, pD, _ = _run(device, False, iters, seed, kw)

    def by_name(step_or_none, split, graphed):
        log, qG, qD, step = _run(device, graphed, iters, seed, kw, split=split)
        return qG, qD
    for graphed in (False, True):
        qG, qD = by_name(None, True, graphed)
        # the split lays the generator's parameters out in another order: compare as sorted multisets of values
        assert qG.numel() == pG.numel()
        assert float((torch.sort(qG)[0] - torch.sort(pG)[0]).abs().max()) == 0.0, ("generator weights differ", graphed)
        assert float((qD - pD).abs().max()) == 0.0, ("discriminator weights differ", graphed)


if __name__ == '__main__':
    dev = torch.device('cuda', 0)
    # atomics (split-K, scatter-add) make GPU runs differ in the last bits from run to run, and Adam turns a rounding-level
    # gradient difference into a +-lr step: compare losses / images loosely and not the weights
    kw = dict(warp_ref=True, spade_combine=True, remove_face_labels=True)
    ref, _, _, _ = _run(dev, False, 5, 500, kw)
    got, _, _, step = _run(dev, True, 5, 500, kw)
    assert any(e.graphs is not None for e in step.entries.values()), "nothing was captured"
    for it, (a, b) in enumerate(zip(ref, got)):
        for k in ('d', 'g'):
            for x, y in zip(a[k], b[k]):
                assert abs(x - y) <= 5e-2 * max(abs(x), 1.0), (it, k, a[k], b[k])
    # the same loop on one stream (streams.ENABLED off): the forked branches only reorder launches in time
    from importlib import import_module
    streams = import_module('few-shot-vid2vid_amd.streams')
    assert streams.ENABLED, "branch streams are expected on by default"
    streams.ENABLED = False
    one, _, _, _ = _run(dev, False, 5, 500, kw)
    streams.ENABLED = True
    for it, (a, b, c) in enumerate(zip(one, ref, got)):
        for other in (b, c):
            for k in ('d', 'g'):
                for x, y in zip(a[k], other[k]):
                    assert abs(x - y) <= 5e-2 * max(abs(x), 1.0), (it, k, a[k], other[k])
    for other in (ref, got):      # first iteration, before any weight update: rounding-level agreement
        assert float((one[0]['img'] - other[0]['img']).abs().max()) <= 1e-4
        for x, y in zip(one[0]['d'] + one[0]['g'], other[0]['d'] + other[0]['g']):
            assert abs(x - y) <= 1e-4 * max(abs(x), 1.0), (one[0], other[0])
    print('GRAPH_STEP_GPU_OK', flush=True)
