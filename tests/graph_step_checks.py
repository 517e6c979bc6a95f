"""GraphedIteration (few-shot-vid2vid_amd/graph_step.py) against the plain eager loop of train.py:58-62: same losses, images and
weights over several iterations with changing data and changing learning rate.  Under the emulator the "graph" is an eager
re-run on the static buffers (buffer plumbing only); run as a script on a GPU it captures and replays real hipGraphs."""
import torch

import model_checks as mc


def _run(device, graphed, iters, seed, opt_kw, b=1, split=False, early=False):
    from importlib import import_module
    M = mc._model()
    gs = import_module('few-shot-vid2vid_amd.graph_step')
    opt = mc.tiny_opt(**opt_kw)
    model = M.create_model(opt)
    mc.fill_state(model.netG); mc.fill_state(model.netD)
    model = model.to(device).train()
    opt_G, opt_D = model.build_optimizers(split_backward=split)
    model.early_generator = early        # the plain loop with the discriminator step on a side stream (model.py)
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
    if graphed is False and early:
        model.join_early()                  # (an open side-stream step; also folds the twin passes' running-statistics stand-ins)
    # every buffer of both networks (BatchNorm running statistics and counters, spectral-norm u / v) rides along on the last log entry
    log[-1]['buffers'] = {('G.' if net is model.netG else 'D.') + k: v.detach().clone().cpu()
                          for net in (model.netG, model.netD) for k, v in net.state_dict().items()
                          if k.endswith(('running_mean', 'running_var', 'num_batches_tracked', 'weight_u', 'weight_v'))}
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


def check_early_generator(device, iters=3, seed=560, tol=0.0):
    """Vid2VidModel.early_generator in the plain train.py loop: the discriminator step on a side stream next to the
    generator-mode forward pass issued behind the step's own no-grad pass - the same kernels on the same data in the same
    per-network order, so losses, images and both networks' weights equal the sequential loop (bit for bit under the emulator,
    which runs the branches in issue order)."""
    kw = dict(warp_ref=True, spade_combine=True, remove_face_labels=True, fineSize=32, loadSize=32, n_downsample_G=3, n_adaptive_layers=2)
    ref, pG, pD, _ = _run(device, False, iters, seed, kw)
    got, qG, qD, _ = _run(device, False, iters, seed, kw, early=True)
    for it, (a, b) in enumerate(zip(ref, got)):
        for k in ('d', 'g'):
            for x, y in zip(a[k], b[k]):
                assert abs(x - y) <= tol * max(abs(x), 1.0), (it, k, a[k], b[k])
        assert float((a['img'] - b['img']).abs().max()) <= tol * 2.0, it
    assert float((pG - qG).abs().max()) <= tol and float((pD - qD).abs().max()) <= tol


def check_twin_generator_passes(device, iters=3, seed=570, graphed=False):
    """Round 6: the iteration's two generator passes issued next to each other (model.Vid2VidModel._forward_discriminator_twin: the
    no-grad pass and the discriminator step on a side stream, the generator-mode pass on the caller's stream; both spectral-norm
    iterations up front, the second pass's BatchNorm running-statistics updates through zeroed stand-ins folded in afterwards)
    against the same loop with FSV_TWIN_G=0: losses, images, all weights AND every buffer (running statistics, counters,
    spectral-norm vectors) bit for bit under the emulator, which runs the two passes in issue order."""
    import os
    from importlib import import_module
    M = mc._model()
    kw = dict(warp_ref=True, spade_combine=True, remove_face_labels=True, fineSize=32, loadSize=32, n_downsample_G=3, n_adaptive_layers=2)
    calls = []
    real = M.Vid2VidModel._forward_discriminator_twin

    def counted(self, *a, **k):
        calls.append(1)
        return real(self, *a, **k)
    M.Vid2VidModel._forward_discriminator_twin = counted
    old = os.environ.get('FSV_TWIN_G')
    try:
        os.environ['FSV_TWIN_G'] = '0'
        ref, pG, pD, _ = _run(device, graphed, iters, seed, kw, split=True, early=True)
        assert not calls
        os.environ['FSV_TWIN_G'] = '2' if device.type == 'cpu' else '1'
        got, qG, qD, _ = _run(device, graphed, iters, seed, kw, split=True, early=True)
        assert len(calls) >= iters - (2 if graphed else 0), calls
    finally:
        M.Vid2VidModel._forward_discriminator_twin = real
        if old is None:
            os.environ.pop('FSV_TWIN_G', None)
        else:
            os.environ['FSV_TWIN_G'] = old
    exact = device.type == 'cpu'
    for it, (a, b) in enumerate(zip(ref, got)):
        for k in ('d', 'g'):
            for x, y in zip(a[k], b[k]):
                assert (x == y) if exact else abs(x - y) <= 5e-2 * max(abs(x), 1.0), (it, k, a[k], b[k])
        if exact:
            assert torch.equal(a['img'], b['img']), it
    if exact:
        assert torch.equal(pG, qG) and torch.equal(pD, qD)
        ba, bb = ref[-1]['buffers'], got[-1]['buffers']
        assert set(ba) == set(bb)
        for k in ba:
            assert torch.equal(ba[k], bb[k]), 'buffer %s differs between the twin and the sequential schedule' % k
    else:
        # hardware: atomics reorder sums from run to run (see __main__ below) - first iteration to rounding, buffers loosely
        assert float((ref[0]['img'] - got[0]['img']).abs().max()) <= 1e-4
        ba, bb = ref[-1]['buffers'], got[-1]['buffers']
        for k in ba:
            if ba[k].dtype.is_floating_point:
                assert float((ba[k] - bb[k]).abs().max()) <= 2e-2 * max(float(ba[k].abs().max()), 1e-3), k
            else:
                assert torch.equal(ba[k], bb[k]), k


def check_serial_point_opt_ins(device, iters=4, seed=580):
    """Round 6, second session - the two opt-in forms of the step's serial point that were measured slower and kept as switches:
    `FSV_ZERO_EARLY=1` (the generator optimiser's gradient-buffer and weight-gradient-arena fills on a side stream next to the
    generator's forward pass, flat.FlatAdam.zero_early) and `FSV_LOSS_TICKET=1` (loss reductions that finish in their own launch).
    Neither changes a value: in the fixed-order mode the loop gives the same losses, images and weights bit for bit with and
    without them - eager and captured -, and the early fills must actually have been issued (on a device)."""
    import os
    from importlib import import_module
    flat = import_module('few-shot-vid2vid_amd.flat')
    kw = dict(warp_ref=True, spade_combine=True, remove_face_labels=True, fineSize=32, loadSize=32, n_downsample_G=3, n_adaptive_layers=2)
    issued = []
    real = flat.FlatAdam.zero_early

    def counted(self, ref):
        br = real(self, ref)
        issued.append(br is not None)
        return br
    keys = ('FSV_DETERMINISTIC', 'FSV_ZERO_EARLY', 'FSV_LOSS_TICKET')
    old = {k: os.environ.get(k) for k in keys}
    flat.FlatAdam.zero_early = counted
    try:
        os.environ['FSV_DETERMINISTIC'] = '1'
        for graphed in (False, True):
            os.environ['FSV_ZERO_EARLY'], os.environ['FSV_LOSS_TICKET'] = '0', '0'
            ref, pG, pD, _ = _run(device, graphed, iters, seed, kw, split=True, early=True)
            assert not any(issued), issued
            os.environ['FSV_ZERO_EARLY'], os.environ['FSV_LOSS_TICKET'] = '1', '1'
            got, qG, qD, _ = _run(device, graphed, iters, seed, kw, split=True, early=True)
            if device.type != 'cpu':
                assert sum(issued) >= iters - 2, issued          # (not before the first optimiser step; one GPU)
            del issued[:]
            for it, (a, b) in enumerate(zip(ref, got)):
                assert a['d'] == b['d'] and a['g'] == b['g'], (graphed, it, a['d'], b['d'], a['g'], b['g'])
                assert torch.equal(a['img'], b['img']), (graphed, it)
            assert torch.equal(pG, qG) and torch.equal(pD, qD), graphed
    finally:
        flat.FlatAdam.zero_early = real
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def check_capture_failure_falls_back(device, iters=4, seed=540):
    """A capture that raises half-way through the iteration body (what a foreign hipEventQuery inside a capture does) must not
    cost the caller its iteration: GraphedIteration drops to the eager step for that signature, says so (launch_mode,
    capture_failures) and the loop's losses and weights stay those of the plain eager loop - with the two-piece backward on."""
    from importlib import import_module
    gs = import_module('few-shot-vid2vid_amd.graph_step')
    kw = dict(warp_ref=True, spade_combine=True, remove_face_labels=True, fineSize=32, loadSize=32, n_downsample_G=3, n_adaptive_layers=2)
    ref, pG, pD, _ = _run(device, False, iters, seed, kw, split=True)
    orig_init = gs.GraphedIteration.__init__

    def patched(self, *a, **k):
        orig_init(self, *a, **k)
        self._can_capture = True                     # the emulator has no graphs: force the capture branch ...

        def broken_capture(e, save_images):          # ... and interrupt the body after the generator's forward + first piece
            # a capture RECORDS the body, it does not execute it: the emulator runs it for real, so the device-side state the
            # body touched is put back before the "capture" fails - what stays behind is exactly what a failed capture leaves
            # on the host (autograd tape, detached .grad slots, queued finaliser jobs, live BackwardCut pairs)
            import torch
            sd = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
            os_ = [(o, o.m.clone(), o.v.clone(), o.state.clone(), o.flat_g.clone()) for o in (self.opt_G, self.opt_D)]
            try:
                self._seg_d(e)
                self._seg_g(e, save_images)
            finally:
                with torch.no_grad():
                    self.model.load_state_dict(sd)
                    for o, m, v, st, g in os_:
                        o.m.copy_(m); o.v.copy_(v); o.state.copy_(st); o.flat_g.copy_(g)
                        o.refresh_layouts()
            raise RuntimeError("injected: operation not permitted on an event last recorded in a capturing stream")
        self._capture = broken_capture
    gs.GraphedIteration.__init__ = patched
    try:
        got, qG, qD, step = _run(device, True, iters, seed, kw, split=True)
    finally:
        gs.GraphedIteration.__init__ = orig_init
    assert len(step.capture_failures) == 1 and 'eager fallback' in step.launch_mode(), step.launch_mode()
    assert all(e.eager_only and e.graphs is None for e in step.entries.values())
    for it, (a, b) in enumerate(zip(ref, got)):
        for k in ('d', 'g'):
            assert a[k] == b[k], (it, k, a[k], b[k])
    assert float((pG - qG).abs().max()) == 0.0 and float((pD - qD).abs().max()) == 0.0


def check_split_backward_single_rank(device, iters=3, seed=520, pieces=True):
    """build_optimizers(split_backward=True) WITHOUT a gradient exchange: the generator's forward pass still detaches at its
    stage boundary, so both drivers (the eager loss_backward and GraphedIteration) have to run the second backward piece -
    weights equal to the unsplit loop bit for bit (round-2 advisor finding: the graphed driver dropped every stage-1 gradient)."""
    kw = dict(warp_ref=True, spade_combine=True, remove_face_labels=True, fineSize=32, loadSize=32, n_downsample_G=3, n_adaptive_layers=2)
    _, pG, pD, _ = _run(device, False, iters, seed, kw)

    def by_name(step_or_none, split, graphed):
        log, qG, qD, step = _run(device, graphed, iters, seed, kw, split=split)
        return qG, qD
    for graphed in (False, True):
        qG, qD = by_name(None, pieces, graphed)          # pieces: True (two) or 3 (a second boundary behind the reference encoders)
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
    # the plain loop with the discriminator step on a side stream next to the generator-mode forward pass (early_generator)
    early, _, _, _ = _run(dev, False, 5, 500, kw, early=True)
    for it, (a, b, c, d) in enumerate(zip(one, ref, got, early)):
        for other in (b, c, d):
            for k in ('d', 'g'):
                for x, y in zip(a[k], other[k]):
                    assert abs(x - y) <= 5e-2 * max(abs(x), 1.0), (it, k, a[k], other[k])
    for other in (ref, got, early):      # first iteration, before any weight update: rounding-level agreement
        assert float((one[0]['img'] - other[0]['img']).abs().max()) <= 1e-4
        for x, y in zip(one[0]['d'] + one[0]['g'], other[0]['d'] + other[0]['g']):
            assert abs(x - y) <= 1e-4 * max(abs(x), 1.0), (one[0], other[0])
    # round 6: the two generator passes of an iteration next to each other (twin schedule) against FSV_TWIN_G=0 - plain loop and
    # captured graph
    check_twin_generator_passes(dev, iters=4)
    check_twin_generator_passes(dev, iters=5, graphed=True)
    # second session: the opt-in forms of the serial point (early fills on a side stream, ticketed loss reductions) change no bit
    check_serial_point_opt_ins(dev)
    print('GRAPH_STEP_GPU_OK', flush=True)
