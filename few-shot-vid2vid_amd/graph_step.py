"""The body of the reference's training loop (train.py:58-62) as replayable hipGraphs.

    d_losses = model(data_list_t, mode='discriminator');  loss_backward(opt, d_losses, optimizer_D, 1)
    g_losses, generated, prevs = model(data_list_t, save_images=..., mode='generator');  loss_backward(opt, g_losses, optimizer_G, 0)

Launched eagerly this body is host-bound on the ROCm stack (about 2500 launches per iteration: 116 ms against 67 ms as a graph
on the C3 bench workload), so a drop-in training loop should not pay for it on every iteration.  `GraphedIteration` keeps one
captured graph per *signature* of `data_list_t` (which entries are None, the shapes of the others, the save_images flag - the
first frame of a sequence has no previous frames and therefore its own graph), copies each call's tensors into the static
input buffers and replays.  The first `warmup` calls of a new signature run eagerly (real iterations on real data; they also
settle lazily-built caches such as the optimisers' layout tables), the next call captures and replays.

With a process group the step is cut into graphs around the gradient all-reduces (the optimisers must have been built with
`overlap=False`), exactly like bench.py does at N > 1: no collective is captured.  With `build_optimizers(split_backward=True)`
the generator's backward pass itself is two graphs (networks.BackwardCut): D | Adam(D) + G forward + first piece of backward |
rest of backward | Adam(G) - the all-reduce of the decoder-stage gradients (half of the generator's parameters) runs on a side
stream next to the third graph, the remaining ranges follow it, and the fused Adam waits for both.

Everything that changes between iterations lives on the device and is read by the captured kernels: learning rate and Adam
step count (`FlatAdam.state`), loss scale (`FlatAdam.scaler`), BatchNorm / spectral-norm buffers.  Rebuilding the optimisers
(`init_temporal_model`) or changing the networks invalidates the graphs: call `reset()` (done automatically when the model's
optimiser objects are no longer the captured ones).

The returned losses / images are the graphs' static output tensors: consume (or clone) them before the next call.
"""
import os

import torch

from . import lib
from . import model as _model
from . import networks as _networks


_EARLY_G = os.environ.get('FSV_EARLY_G', '1') == '1'          # in-box A/B switch


def _seg_early_g():
    """N > 1 (segmented): the generator-mode forward pass as a segment of its own next to the discriminator's gradient exchange
    (FSV_SEG_EARLY_G=0: the serial order of round 4, for the bit-identity tests and A/B runs; read at bind time)"""
    return os.environ.get('FSV_SEG_EARLY_G', '1') == '1'


def _flat(data_list):
    out = []
    for item in data_list:
        if isinstance(item, (list, tuple)):
            out.extend(item)
        else:
            out.append(item)
    return out


def _like(data_list, flat):
    it = iter(flat)
    out = []
    for item in data_list:
        if isinstance(item, (list, tuple)):
            out.append([next(it) for _ in item])
        else:
            out.append(next(it))
    return out


class _Entry:
    def __init__(self):
        self.static = None          # data_list with static device tensors
        self.calls = 0
        self.graphs = None
        self.outputs = None
        self.eager_only = False     # a capture failed for this signature: the step keeps running eagerly


class GraphedIteration:
    def __init__(self, model, opt, warmup=2):
        self.model = getattr(model, 'module', model)
        self.opt = opt
        self.warmup = max(int(warmup), 1)
        self.entries = {}
        self.capture_failures = []      # (signature, first line of the error) of every capture that fell back to the eager step
        self._bind()
        # the SIMT-emulated library (CPU test infrastructure) has no graphs: the body is re-run eagerly on the static buffers,
        # which exercises everything here except the capture itself
        self._emulated = lib.emu_requested()
        if not self._emulated and not torch.cuda.is_available():
            raise RuntimeError("GraphedIteration needs a GPU (hipGraph capture)")
        self._can_capture = not self._emulated

    # ------------------------------------------------------------------------------------------------ bookkeeping
    def _bind(self):
        self.opt_G, self.opt_D = self.model.optimizer_G, self.model.optimizer_D
        if self.opt_G is None or self.opt_D is None:
            raise RuntimeError("build the optimisers (model.build_optimizers) before graphing the iteration")
        self._generations = (self.opt_G.generation, self.opt_D.generation)      # bumped by FlatAdam.rebuild (init_temporal_model)
        from . import ops
        if ops.bn_sync_world() > 1 and not lib.emu_requested():
            raise RuntimeError("cross-replica BatchNorm issues collectives inside forward / backward, which cannot be captured: "
                               "use the eager loop with sync_bn, or per-replica statistics with GraphedIteration")
        self.segmented = bool(self.opt_G.exchange)
        # two-piece generator backward (model.build_optimizers(split_backward=True)): the generator's forward pass detaches at
        # its stage boundary whether or not there is an exchange to overlap, so the second piece always has to run
        self.split = bool(getattr(self.model, 'split_backward', False))
        # three pieces (build_optimizers(split_backward=3)): a second boundary behind the reference encoders - the exchange of the
        # middle range (weight generators, embeddings, flow network) runs next to the encoders' backward as well
        self.pieces = 3 if getattr(self.model, 'split_backward', False) is not True and getattr(self.model, 'split_backward', 0) == 3 \
            else (2 if self.split else 1)
        self.seg_early = self.segmented and _seg_early_g()
        if self.segmented and self.opt_G.overlap:
            raise RuntimeError("with a process group build the optimisers with overlap=False: bucket hooks issue collectives "
                               "inside backward, which cannot be captured")

    def reset(self):
        """drop every captured graph (after init_temporal_model, load_networks, architecture changes)"""
        self.entries = {}
        self._bind()

    @staticmethod
    def _signature(flat, save_images):
        return (bool(save_images),) + tuple(None if t is None else (tuple(t.shape), t.dtype) for t in flat)

    # ------------------------------------------------------------------------------------------------ the body
    def _backward(self, losses, optimizer):
        with _model.branch_of(losses, optimizer):   # (the discriminator step of a single-graph iteration lives on a side stream)
            losses, loss = _model.mean_and_total(losses)
            optimizer.zero_grad()
            optimizer.scale_loss(loss).backward()
            _networks.BackwardCut.finish_all()      # a forward pass that detached at a stage boundary (no-op otherwise)
            optimizer.finalize_grads()
            if optimizer is self.opt_D and self.model._pre_g is not None:
                optimizer.adam()                    # ... together with its Adam step (else issued at the top of _seg_g)
                self._d_stepped = True
        return losses

    def _seg_d(self, e):
        # iterations without a gradient exchange: the discriminator step runs on a side stream next to the generator-mode forward
        # pass (model.early_generator); with segments the branch would have to cross the all-reduce between two graphs - there the
        # generator-mode pass becomes a segment of its own next to that all-reduce (_seg_gf)
        was = self.model.early_generator
        self.model.early_generator = (not self.segmented) and _EARLY_G
        self._d_stepped = False
        try:
            e.out_d = self._backward(self.model(e.static, mode='discriminator'), self.opt_D)
        finally:
            self.model.early_generator = was

    def _seg_gf(self, e):
        # N > 1: the generator-mode forward pass of train.py:61 while the discriminator's gradients are being exchanged
        self.model.early_generate(e.static)

    def _seg_g(self, e, save_images):
        if not getattr(self, '_d_stepped', False):
            self.opt_D.adam()
        self._d_stepped = False
        g_losses, generated, prevs = self.model(e.static, save_images=save_images, mode='generator')
        if not self.split:
            e.out_g = self._backward(g_losses, self.opt_G)
        else:                                  # first piece: down to the generator's stage boundary
            losses, loss = _model.mean_and_total(g_losses)
            self.opt_G.zero_grad()
            self.opt_G.scale_loss(loss).backward()
            if not self.opt_G.step_stage2_early():      # (one GPU: the decoder stage's Adam + layouts next to the second piece)
                self.opt_G.finalize_grads(partial=True)
            e.out_g = losses
        e.out_gen, e.out_prev = generated, prevs

    def _seg_g2(self):
        self.model.netG.bwd_cut.backward_rest()
        self.opt_G.finalize_grads(partial=self.pieces == 3)

    def _seg_g3(self):
        self.model.netG.bwd_cut2.backward_rest()
        self.opt_G.finalize_grads()

    def _seg_a(self):
        self.opt_G.adam()

    # round 6: the generator's optimiser step in two segments - the decoder-stage range (its exchange completed next to the second
    # backward piece) steps while the last range of the exchange is still in flight (FlatAdam.adam_part)
    def _seg_a1(self):
        self.opt_G.adam_part(0)

    def _seg_a2(self):
        self.opt_G.adam_part(1)

    # ---- the collectives between the segments (host side; never captured) -------------------------------------------------
    def _exchange_d(self):
        # the discriminator's 11 MB on the optimiser's side stream: it runs next to the segment that follows (_seg_gf)
        self.opt_D.exchange_range(0, self.opt_D.total, side=True, label='D')

    def _wait_d(self):
        self.opt_D.wait_exchange()

    def _exchange_d_serial(self):
        self.opt_D.exchange_all('D')

    def _exchange_g_first(self):
        self.opt_G.exchange_range(0, self.opt_G.split_at, side=True, label='G decoder stage')

    def _exchange_g_middle(self):
        # (the side stream runs its collectives in order: this one queues behind the first range)
        self.opt_G.exchange_range(self.opt_G.split_at, self.opt_G.split_at2, side=True, label='G middle')

    def _exchange_g_rest(self):
        self.opt_G.exchange_range(self.opt_G.split_at2 if self.pieces == 3 else self.opt_G.split_at, self.opt_G.total, label='G rest')
        self.opt_G.wait_exchange()

    def _exchange_g_rest_side(self):
        # the last range on the side stream (behind the decoder-stage range); the caller's stream only waits for the decoder range
        self.opt_G.exchange_range(self.opt_G.split_at, self.opt_G.total, side=True, label='G rest')
        self.opt_G.wait_exchange_of('G decoder stage')

    def _wait_g(self):
        self.opt_G.wait_exchange()

    def _exchange_g_all(self):
        self.opt_G.exchange_all('G')

    def _steps(self, e, save_images):
        """The iteration as (body, after) pairs: `body` is what one hipGraph segment captures, `after` runs on the host behind it
        (the RCCL calls and the stream waits for them - nothing of that is captured).  Per iteration at N > 1 (two-piece backward):

            D fwd + bwd             | all-reduce(D) -> side stream
            G forward (train.py:61) |                   ... runs next to it | wait
            Adam(D), D pass, losses, backward piece 1 | all-reduce(G decoder range) -> side stream
            backward piece 2        |                   ... runs next to it | all-reduce(G rest) -> side stream; wait for the decoder range
            Adam + layouts, decoder range |                ... runs next to it | wait                    <- what is left of it is exposed
            Adam + layouts, the rest                                            (round 6; FSV_SEG_SPLIT_ADAM=0: one Adam segment behind the wait)
        """
        seg = self.segmented
        steps = []
        if seg and self.seg_early:
            steps.append((lambda: self._seg_d(e), self._exchange_d))
            steps.append((lambda: self._seg_gf(e), self._wait_d))
        else:
            steps.append((lambda: self._seg_d(e), self._exchange_d_serial if seg else None))
        if self.split:
            steps.append((lambda: self._seg_g(e, save_images), self._exchange_g_first))
            if self.pieces == 3:
                steps.append((self._seg_g2, self._exchange_g_middle))
                steps.append((self._seg_g3, self._exchange_g_rest))
            elif self._split_adam():
                # backward piece 2 | all-reduce(G rest) -> side stream, wait for the decoder range only
                # Adam + layouts of the decoder range ... next to it | wait
                # Adam + layouts of the rest
                steps.append((self._seg_g2, self._exchange_g_rest_side))
                steps.append((self._seg_a1, self._wait_g))
                steps.append((self._seg_a2, None))
                return steps
            else:
                steps.append((self._seg_g2, self._exchange_g_rest))
        else:
            steps.append((lambda: self._seg_g(e, save_images), self._exchange_g_all if seg else None))
        steps.append((self._seg_a, None))
        return steps

    def _split_adam(self):
        return self.segmented and self.split and self.pieces == 2 and self.opt_G.split_adam_ready()

    def n_segments(self):
        return (3 + (self.pieces - 1 if self.split else 0) + (1 if (self.segmented and self.seg_early) else 0) +
                (1 if self._split_adam() else 0))

    def _eager(self, e, save_images):
        for body, after in self._steps(e, save_images):
            body()
            if after is not None:
                after()

    # one poll period of ProcessGroupNCCL's watchdog thread (kWatchdogThreadSleepMillis = 100 ms in torch 2.x) plus margin
    WATCHDOG_PERIOD_S = 0.15

    def _quiesce(self):
        """Before the captures: nothing of ours may be in flight on the process group while a stream captures.  A watchdog
        `hipEventQuery` that lands inside a capture was seen to fail with hipErrorCapturedEvent on ROCm 7 (one run in eight,
        once the captured body itself records and destroys events - the branch streams).  Round 3 slept 0.35 s in front of each
        of the four captures and hoped; now (1) every exchange collective carries a work handle (FlatAdam._all_reduce) and this
        function blocks until each handle reports completion, (2) the segments are captured back to back with NO collective
        between them (a capture only records: it does not need the previous segment's results), so the process group sees no
        new work from the first capture to the last, (3) if handles were outstanding the watchdog gets one poll period to reap
        them - it drops a work object at the first poll that finds it complete -, and (4) a capture that fails all the same is
        caught by __call__, which falls back to the eager segmented step for that signature (`launch_mode()` says so)."""
        torch.cuda.synchronize(self.opt_G.device)
        if self.segmented:
            pending = self.opt_D.drain_works() + self.opt_G.drain_works()
            if pending:
                import time
                time.sleep(self.WATCHDOG_PERIOD_S)

    def _capture(self, e, save_images):
        self._quiesce()
        if not self.segmented:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._eager(e, save_images)
            e.graphs = [g]
            e.afters = [None]
            return
        # the captures only police their own thread (the RCCL watchdog polls its events from another one); the caller replays
        # the whole sequence - with the all-reduces between the graphs - once all of them exist
        steps = self._steps(e, save_images)
        gs = [torch.cuda.CUDAGraph() for _ in steps]
        for k, (body, _) in enumerate(steps):
            kw = dict(capture_error_mode='thread_local')
            if k:
                kw['pool'] = gs[0].pool()
            with torch.cuda.graph(gs[k], **kw):
                body()
        e.graphs = gs
        e.afters = [after for _, after in steps]

    def _replay(self, e):
        for g, after in zip(e.graphs, e.afters):
            g.replay()
            if after is not None:
                after()

    def _capture_failed(self, e, key, ex):
        """a capture raised (e.g. hipErrorCapturedEvent from a foreign event query, an allocation the capture could not make):
        drop the half-built graphs, clear what the interrupted body left behind and keep this signature on the eager step"""
        e.graphs, e.eager_only = None, True
        self.capture_failures.append((key, str(ex).split('\n')[0][:200]))
        if torch.cuda.is_available() and not self._emulated:
            try:
                torch.cuda.synchronize(self.opt_G.device)
            except Exception:                # noqa: BLE001
                pass
        _networks.BackwardCut.abandon_all()      # boundary tensors of the interrupted forward pass
        try:
            self.model.join_early()              # a side-stream discriminator step the interrupted body left open
        except Exception:                        # noqa: BLE001
            self.model._pre_g = None
        for o in (self.opt_D, self.opt_G):
            o._reattach()
            try:
                o.abandon_early()                # a decoder-stage step the interrupted body issued
            except Exception:                    # noqa: BLE001
                o._early = None

    def launch_mode(self):
        """how the iterations of this object reach the device - for bench.py's `config.launch`"""
        n = self.n_segments()
        if self.capture_failures:
            return ('eager fallback (hipGraph capture failed: %s)%s' % (self.capture_failures[-1][1],
                    ', all-reduce between %d eager segments' % n if self.segmented else ''))
        if self._emulated:
            return 'emulated kernels on host tensors, %d eager segments' % n if self.segmented else 'emulated kernels, eager'
        if not self.segmented:
            return 'hipgraph'
        return 'hipgraph x%d + RCCL all-reduce between segments%s%s' % (
            n, ' (decoder-stage range on a side stream)' if self.split else '',
            ' (discriminator range on a side stream next to the generator-mode forward pass)' if self.seg_early else '')

    # ------------------------------------------------------------------------------------------------ call
    def __call__(self, data_list, save_images=False):
        """one iteration; returns (d_losses, g_losses, generated, prevs_new) like the two model calls of train.py:58-62"""
        if (self.model.optimizer_G is not self.opt_G or self.model.optimizer_D is not self.opt_D or
                (self.opt_G.generation, self.opt_D.generation) != self._generations):
            self.reset()
        flat = _flat(data_list)
        key = self._signature(flat, save_images)
        e = self.entries.get(key)
        if e is None:
            e = self.entries[key] = _Entry()
            dev = self.opt_G.device
            e.static = _like(data_list, [None if t is None else t.detach().to(dev).clone() for t in flat])
        else:
            for dst, src in zip(_flat(e.static), flat):
                if dst is not None:
                    dst.copy_(src, non_blocking=True)
        e.calls += 1
        if not self._can_capture or e.eager_only or e.calls <= self.warmup:
            if not self._emulated and e.calls == 1 and torch.cuda.is_available():
                # warm-up runs on a side stream so that allocations made now do not end up in the capture's private pool
                s = torch.cuda.Stream(self.opt_G.device)
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self._eager(e, save_images)
                torch.cuda.current_stream().wait_stream(s)
            else:
                self._eager(e, save_images)
        elif e.graphs is None:
            try:
                self._capture(e, save_images)
            except Exception as ex:          # noqa: BLE001 - a capture fault must never cost the caller its iteration
                self._capture_failed(e, key, ex)
                self._eager(e, save_images)
            else:
                self._replay(e)
        else:
            self._replay(e)
        return e.out_d, e.out_g, e.out_gen, e.out_prev


def graphed_iteration(model, opt, warmup=2):
    """convenience: `step = graphed_iteration(model, opt); d, g, generated, prevs = step(data_list_t, save_images)`"""
    return GraphedIteration(model, opt, warmup)


loss_backward = _model.loss_backward      # the eager counterpart, for loops that mix both
