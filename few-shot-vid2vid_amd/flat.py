"""Flat parameter / gradient storage, fused Adam and the RCCL gradient exchange.

Replaces, for the hot path, ``torch.optim.Adam`` (models/base_model.py:39-48) and the reference's data-parallel
wrapper (apex DistributedDataParallel with delay_allreduce=True, models/models.py:40-43; util/distributed.py).

* every parameter of one optimiser is a view into ONE contiguous fp32 buffer, and so is its ``.grad``:
  the optimiser step is a single fused HIP kernel (csrc/elementwise.hip, fsv_adam_step) and ``zero_grad`` is one
  memset;
* with world_size > 1 the gradient buffer is cut into large buckets (default 64 MiB - sized for the point-to-point
  xGMI links, not for many small NCCL-style messages) and each bucket is all-reduced on a side stream as soon as
  autograd has produced its last gradient, i.e. overlapped with the rest of backward.  Parameters are laid out in
  reverse registration order so that buckets complete front to back during backward.  The 1/world_size average is
  folded into the Adam kernel.  BatchNorm statistics stay per replica (DESIGN.md "Multi-GPU").
* `overlap=False` selects the second exchange mode used by bench.py at N > 1: no autograd hooks at all - the step is
  captured as hipGraph segments (backward | Adam + next forward/backward | Adam) and ONE all-reduce over the whole
  flat gradient buffer runs between the segments (`exchange()`), i.e. the fewest, largest collectives possible and
  no per-launch host overhead; the un-overlapped transfer (0.39 GB over xGMI) costs a few ms of an 80 ms step.
"""
import os

import torch
import torch.distributed as dist

from . import ops
from .layout_cache import LayoutCache
from .grad_finalize import GradFinalizer


class FlatAdam:
    def __init__(self, params, lr, betas=(0.0, 0.999), world_size=1, process_group=None, eps=1e-8, bucket_mb=64,
                 force_exchange=False, overlap=True, loss_scale=None):
        self.world_size = world_size
        # force_exchange: run the bucket / hook / side-stream machinery even in a one-rank group (smoke test of the
        # exact multi-GPU code path on a single-GPU box: the all-reduce is then an identity)
        self.exchange = world_size > 1 or force_exchange
        self.overlap = overlap and self.exchange
        self.group = process_group
        self.betas, self.eps = betas, eps
        self.bucket_mb = bucket_mb
        # bumped by rebuild(): captured graphs / cached pointers of an older layout are stale (graph_step.py checks it)
        self.generation = 0
        self._hook_handles = []
        # split_at: element offset in the flat buffers below which the gradients are complete after the FIRST piece of a
        # two-piece backward (set by Vid2VidModel.build_optimizers(split_backward=True): stage-2 parameters are laid out first)
        self.split_at = 0
        self.split_at2 = 0           # three-piece backward: flat_g[split_at:split_at2] is complete after the second piece
        self._side_events = {}       # label -> event behind that side-stream collective (wait_exchange_of)
        self._lay_out(params, lr, loss_scale)

    def rebuild(self, params, lr=None, loss_scale='keep'):
        """Re-lay this optimiser over a new parameter set IN PLACE: fresh flat buffers, Adam moments and step count
        (what the reference's `get_optimizer` call inside init_temporal_model does, base_model.py:259-279), but the
        Python object stays the one train.py got from create_model (train.py:36 keeps those handles for the whole
        run - a second FlatAdam over the same parameters would leave the first one stepping orphaned buffers)."""
        for h in self._hook_handles:
            h.remove()
        self._hook_handles = []
        self._reattach()
        for p in self.params:
            for a in ('_fsv_sink', '_fsv_cache', '_fsv_finalizer'):
                if hasattr(p, a):
                    delattr(p, a)
        if lr is None:
            lr = float(self.state[3])
        if loss_scale == 'keep':
            loss_scale = None if self.scaler is None else (float(self.scaler[0]), int(self.scaler[3]))
        self.generation += 1
        self._lay_out(params, lr, loss_scale)
        return self

    def _lay_out(self, params, lr, loss_scale):
        world_size, bucket_mb = self.world_size, self.bucket_mb
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("no trainable parameters")
        # reverse order: the last layers' gradients (first to be produced by backward) sit at the front
        self.params = list(reversed(params))
        self.device = params[0].device
        total = sum(p.numel() for p in self.params)
        self.total = total
        self.flat_p = torch.empty(total, dtype=torch.float32, device=self.device)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.m = torch.zeros_like(self.flat_g)
        self.v = torch.zeros_like(self.flat_g)
        self.state = torch.tensor([0.0, 0.0, 0.0, float(lr)], dtype=torch.float32, device=self.device)
        # fp16-operand mode (conv.MFMA_F16): apex-style dynamic loss scale kept on the device (csrc/amp.hip);
        # scaler = [scale, good_steps, found_inf, window, max_scale, min_scale].  None: plain fp32 step.
        self.scaler = None
        if loss_scale is not None:
            init, window = (2.0 ** 16, 2000) if loss_scale is True else loss_scale
            self.scaler = torch.tensor([float(init), 0.0, 0.0, float(window), 2.0 ** 24, 1.0], dtype=torch.float32,
                                       device=self.device)
        # persistent K-major layouts of every weight this optimiser owns, rewritten once per step (layout_cache.py)
        self.layouts = LayoutCache() if os.environ.get('FSV_LAYOUT_CACHE', '1') == '1' else None
        # weight gradients stay in the GEMM's layout until one grouped launch folds them into flat_g (grad_finalize.py)
        self.finalizer = (GradFinalizer() if (self.layouts is not None and not self.overlap and
                                              os.environ.get('FSV_GRAD_SINK', '1') == '1' and
                                              os.environ.get('FSV_DEFER_WGRAD', '1') == '1') else None)
        self.offsets = []
        self._grad_views = []          # the flat_g slice of every parameter (its .grad outside a backward pass)
        self._loose = []               # parameters whose .grad is detached from flat_g for the current pass
        self._steps_done = 0
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.flat_p[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + n].view(p.shape)
                p.grad = self.flat_g[off:off + n].view(p.shape)
                self._grad_views.append(p.grad)
                # single process: kernels may add gradients straight into the slice (ops._ConvFn "gradient sink");
                # with a process group the autograd hooks below have to see every gradient, so the sink stays off
                p._fsv_sink = (not self.overlap) and os.environ.get('FSV_GRAD_SINK', '1') == '1'
                if self.layouts is not None and p.dim() in (2, 4):
                    p._fsv_cache = self.layouts
                    if self.finalizer is not None:
                        p._fsv_finalizer = self.finalizer
                if self.finalizer is not None and p.dim() == 1:
                    p._fsv_finalizer = self.finalizer      # conv biases: grouped column sums (grad_finalize.add_bias)
                self.offsets.append((off, n))
                off += n
        # ---- data-parallel buckets ------------------------------------------------------------------------
        self._armed = False
        self.buckets = []            # (start, end) element ranges of flat_g
        self._pending, self._handles = [], []
        self._works = []             # hook-free mode: work handles of the collectives issued since the last drain_works()
        self._param_bucket = {}
        self.side_stream = None
        if self.overlap:
            cap = bucket_mb * (1 << 20) // 4
            start, count = 0, 0
            cur = []
            for idx, (o, n) in enumerate(self.offsets):
                cur.append(idx)
                count += n
                if count >= cap or idx == len(self.offsets) - 1:
                    self.buckets.append((start, start + count, list(cur)))
                    start += count
                    count, cur = 0, []
            for b, (_, _, idxs) in enumerate(self.buckets):
                for i in idxs:
                    self._param_bucket[i] = b
            self._remaining = [len(b[2]) for b in self.buckets]
            self._launched = [False] * len(self.buckets)
            if self.device.type == 'cuda':
                self.side_stream = torch.cuda.Stream(device=self.device)
            for i, p in enumerate(self.params):
                self._hook_handles.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    # ------------------------------------------------------------------------------------------------ DDP
    def _make_hook(self, i):
        def hook(param):
            if not self._armed:        # gradients produced for another optimiser's step (e.g. D's during the G step)
                return
            b = self._param_bucket[i]
            self._remaining[b] -= 1
            if self._remaining[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        if self._launched[b]:
            return
        self._launched[b] = True
        s, e, _ = self.buckets[b]
        view = self.flat_g[s:e]
        if self.side_stream is not None:
            self.side_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.side_stream):
                h = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            h = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._handles.append(h)

    def _finish_exchange(self):
        if not self.overlap:
            return
        for b in range(len(self.buckets)):       # buckets whose parameters were not all touched this step
            self._launch(b)
        for h in self._handles:
            h.wait()
        if self.side_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.side_stream)
        self._handles = []
        self._armed = False

    # ------------------------------------------------------------------------------------------------ optimiser API
    _prezeroed = False           # zero_early() has cleared the gradient buffers of the pass that comes next
    _dirty = False               # a backward pass may have written gradients that no step has consumed yet

    def zero_early(self, ref):
        """The fills of zero_grad() ahead of time, on a side stream: returns the open streams.Branch (the caller joins it with
        .finish() before anything can write a gradient - model.Vid2VidModel.generate_images joins at the end of the generator's
        forward pass) or None.  Only between a step and the next zero_grad() (no un-stepped gradients), one GPU.
        Opt-in (FSV_ZERO_EARLY=1): measured in-box 42.59 / 42.63 ms per step without, 42.80 / 42.88 with
        (profiles/r06_step_ab_serial_point.txt) - the 93 us of fills leave the serial point, but one more fork in the captured
        forward pass costs the graph executor more than that (the finding of rounds 2 and 6 about finer forks, once more)."""
        if (self._dirty or self._prezeroed or self.exchange or self._early is not None or not torch.is_tensor(ref) or
                not ref.is_cuda or os.environ.get('FSV_ZERO_EARLY', '0') != '1' or self._steps_done < 1):
            return None
        from . import streams
        br = streams.Branch(ref)
        if br.stream is None:
            return None
        with br.on():
            self.flat_g.zero_()
            if self.finalizer is not None:
                self.finalizer.zero_arena()
        self._prezeroed = True
        return br

    def zero_grad(self, set_to_none=False):
        """Called by loss_backward right before backward: clears the flat gradient and arms the bucket hooks."""
        self.abandon_early()
        pre, self._prezeroed = self._prezeroed, False
        self._dirty = True
        if not pre:
            self.flat_g.zero_()
        if self.finalizer is not None:
            self.finalizer.begin_pass(prezeroed=pre)       # drops jobs of a pass that was never stepped, zeroes the wgrad arena
            # Small parameters that do not take their gradient through a kernel-side sink (norm weights / biases, fixed
            # SPADE weights) would each cost an AccumulateGrad add launch into their flat_g slice.  Detach .grad for the
            # pass instead - autograd then just keeps the incoming tensor - and fold all of them into flat_g with one
            # grouped launch in finalize_grads().  Which parameters are sink-fed is known after the first step.
            self._reattach()
            if self._steps_done >= 1 and os.environ.get('FSV_LOOSE_GRADS', '1') == '1':
                for i, p in enumerate(self.params):
                    if not getattr(p, '_fsv_conv_param', False):
                        p.grad = None
                        self._loose.append(i)
        if self.overlap:
            self._remaining = [len(b[2]) for b in self.buckets]
            self._launched = [False] * len(self.buckets)
            self._handles = []
            self._armed = True

    def set_lr(self, lr):
        self.state[3:4].fill_(float(lr))

    @property
    def param_groups(self):
        """torch.optim-style view used by the reference's update_learning_rate (base_model.py:253-256)."""
        outer = self

        class _Group(dict):
            def __setitem__(self, k, v):
                if k == 'lr':
                    outer.set_lr(v)
                super().__setitem__(k, v)
        return [_Group(lr=float(self.state[3]), params=self.params)]

    def finalize_grads(self, partial=False):
        """Fold the queued K-major weight gradients of the last backward pass into the flat gradient buffer.  Runs by
        itself before the exchange / Adam step; call it explicitly to read `.grad` of a weight before stepping.
        partial=True: the backward pass is not over (two-piece backward, networks.BackwardCut) - fold what has been
        produced so far and leave the parameters that have no gradient yet detached."""
        if self.finalizer is not None:
            self.finalizer.run()
            if self._loose:
                pairs, done = [], []
                for i in self._loose:
                    g = self.params[i].grad
                    if g is not None:
                        pairs.append((g.contiguous().view(-1), self._grad_views[i].view(-1)))
                        done.append(i)
                self.finalizer.gather_dense(pairs)
                if partial:
                    for i in done:
                        self.params[i].grad = self._grad_views[i]
                    keep = set(done)
                    self._loose = [i for i in self._loose if i not in keep]
                else:
                    self._reattach()

    def _reattach(self):
        for i in self._loose:
            self.params[i].grad = self._grad_views[i]
        self._loose = []

    exchange_log = None          # set to a list to record every collective (exchange_summary)

    def _all_reduce(self, view, label='', side=False):
        """hook-free mode: the collective is issued with a work handle (async_op=True), the issuing stream waits for it at once
        (`wait()` on an RCCL work is a stream-side dependency, the host does not block) and the handle is kept until
        drain_works(): graph_step waits for every handle to report completion before it starts a hipGraph capture, so the
        process group's watchdog thread has nothing of ours left in flight while streams are capturing."""
        log = self.exchange_log
        if log is not None:                    # bench.py at N > 1: bytes and issue -> complete time of every collective
            import time
            cuda = view.is_cuda
            if cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            t0 = time.perf_counter()
        h = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if h is not None:
            h.wait()
            if len(self._works) >= 16:         # a long replay loop never drains: keep only what is still in flight
                self._works = [w for w in self._works if not w.is_completed()]
            self._works.append(h)
        if log is not None:
            if cuda:
                e1.record()                    # behind the stream-side wait: reached when the collective has completed
            log.append(dict(name=label, bytes=view.numel() * view.element_size(), side=bool(side),
                            events=(e0, e1) if cuda else None, host_s=time.perf_counter() - t0))

    @staticmethod
    def exchange_summary(log):
        """per collective name: launches, bytes and the average issue -> complete time (device events on the issuing stream; the
        host-side wall time for a CPU group).  Call after a device synchronise."""
        agg = {}
        for r in log:
            a = agg.setdefault(r['name'], dict(name=r['name'], n=0, bytes=r['bytes'], ms=0.0, side_stream=r['side']))
            a['n'] += 1
            a['ms'] += (r['events'][0].elapsed_time(r['events'][1]) if r['events'] is not None else r['host_s'] * 1e3)
        out = []
        for a in agg.values():
            a['ms'] = round(a['ms'] / max(a['n'], 1), 3)
            a['GB_per_s'] = round(a['bytes'] / max(a['ms'], 1e-6) / 1e6, 1)
            out.append(a)
        return out

    def drain_works(self, timeout_s=60.0):
        """Block the host until every collective issued through _all_reduce has completed (work.is_completed(): the end event of
        the collective has been reached on the device); returns the number of handles that were outstanding."""
        import time
        works, self._works = self._works, []
        t0 = time.monotonic()
        for h in works:
            while not h.is_completed():
                if time.monotonic() - t0 > timeout_s:
                    raise RuntimeError("gradient exchange did not complete within %.0f s" % timeout_s)
                time.sleep(0.001)
        return len(works)

    def exchange_all(self, label=''):
        """Non-overlapped mode: one all-reduce over the whole flat gradient buffer on the current stream."""
        self.finalize_grads()
        if self.exchange and not self.overlap:
            self._all_reduce(self.flat_g, label or 'all')

    def exchange_range(self, lo, hi, side=False, label=''):
        """All-reduce flat_g[lo:hi] (hook-free mode; the caller has finalised the gradients in that range).  side=True:
        on this optimiser's side stream, which first waits for the work queued on the current stream - the collective then
        runs next to whatever the caller enqueues on the current stream afterwards; join with wait_exchange()."""
        if not (self.exchange and not self.overlap) or hi <= lo:
            return
        view = self.flat_g[lo:hi]
        if side and self.device.type == 'cuda':
            if self.side_stream is None:
                self.side_stream = torch.cuda.Stream(device=self.device)
            self.side_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.side_stream):
                self._all_reduce(view, label, side=True)
                # (the side stream runs its collectives in order: an event behind each lets a consumer wait for ONE range)
                ev = torch.cuda.Event()
                ev.record(self.side_stream)
            self._side_events[label] = ev
            self._side_pending = True
        else:
            self._all_reduce(view, label)

    def wait_exchange_of(self, label):
        """the current stream waits for the side-stream collective issued under `label` (not for the ones queued behind it)"""
        ev = self._side_events.pop(label, None)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)

    def split_adam_ready(self):
        """the optimiser step of a segmented iteration in two launches (adam_part): needs the two-piece layout, plain fp32 (the
        `--amp` overflow test spans all gradients) and the layout cache (refresh_split)"""
        return (0 < self.split_at < self.total and self.scaler is None and self.layouts is not None and self._early is None and
                os.environ.get('FSV_SEG_SPLIT_ADAM', '1') == '1')

    def adam_part(self, part):
        """Round 6 (N > 1): Adam + layout refresh of flat[:split_at] (part 0: the decoder stage, whose gradients were exchanged next
        to the second backward piece) issued while the LAST range of the exchange is still in flight on the side stream, then part 1
        for the rest.  One optimiser step: the state ticks once (with part 0), every parameter sees the arithmetic of adam() -
        bit-identical weights (tests/test_ddp_gloo.py)."""
        s = self.split_at
        sl = slice(0, s) if part == 0 else slice(s, self.total)
        ops.adam_step(self.flat_p[sl], self.flat_g[sl], self.m[sl], self.v[sl], self.state, self.betas[0], self.betas[1], self.eps,
                      1.0 / self.world_size, tick=(part == 0))
        base = self.flat_p.data_ptr()
        self.layouts.refresh_split(part, base, base + 4 * s)
        if part == 1:
            self._steps_done += 1
            self._dirty = False

    def wait_exchange(self):
        if getattr(self, '_side_pending', False):
            torch.cuda.current_stream(self.device).wait_stream(self.side_stream)
            self._side_pending = False

    def scale_loss(self, loss):
        """models/loss_collector.py:221-224 `amp.scale_loss`: the loss times this optimiser's current scale (a device
        scalar - nothing is read back); identity without a scaler."""
        return loss if self.scaler is None else loss * self.scaler[0]

    # ---- the step of the decoder stage between the two pieces of a split backward (one GPU) -------------------------------
    # With build_optimizers(split_backward=True) the gradients of flat_g[:split_at] (the generator's decoder stage) are final
    # once the first piece of the backward pass has run.  Without a gradient exchange nothing else has to happen to them: their
    # Adam launch and the refresh of their GEMM layouts go to a side stream (streams.Branch) and run next to the second piece
    # (encoders, weight generators, flow network: several ms of MFMA kernels); adam() then joins the branch and steps the rest.
    # Same arithmetic per parameter: bit-identical weights.  Never under `--amp` (the overflow test spans all gradients) or with an
    # exchange (the ranges are all-reduced first).  Measured neutral on one GPU (profiles/r04_notes.md section 11: 47.04 / 47.10 ms
    # without, 47.08 / 47.14 ms with - the optimiser's streaming kernels take from the second piece what they hide), hence an
    # opt-in: FSV_EARLY_ADAM=1.
    _early = None

    def abandon_early(self):
        """a pass whose second piece / Adam never ran (an interrupted capture, an exception): join the branch, forget it"""
        if self._early is not None:
            branch, self._early = self._early, None
            branch.finish()

    def early_step_ready(self):
        return (self.split_at > 0 and self.scaler is None and not self.exchange and self.layouts is not None and
                os.environ.get('FSV_EARLY_ADAM', '0') == '1')

    def step_stage2_early(self):
        if not self.early_step_ready() or self._early is not None:
            return False
        from . import streams
        self.finalize_grads(partial=True)          # (on the caller's stream: its temporaries belong to that stream's allocator)
        s = self.split_at
        branch = streams.Branch(self.flat_g)
        with branch.on():
            ops.adam_step(self.flat_p[:s], self.flat_g[:s], self.m[:s], self.v[:s], self.state, self.betas[0], self.betas[1],
                          self.eps, 1.0 / self.world_size, tick=True)
            base = self.flat_p.data_ptr()
            self.layouts.refresh_split(0, base, base + 4 * s)
        self._early = branch
        return True

    def adam(self):
        self._dirty = False
        self.finalize_grads()
        if self._early is not None:
            branch, self._early = self._early, None
            branch.finish()                         # (the state was ticked on the branch; the second piece took ms longer than it)
            s = self.split_at
            ops.adam_step(self.flat_p[s:], self.flat_g[s:], self.m[s:], self.v[s:], self.state, self.betas[0], self.betas[1],
                          self.eps, 1.0 / self.world_size, tick=False)
            self._steps_done += 1
            base = self.flat_p.data_ptr()
            self.layouts.refresh_split(1, base, base + 4 * s)
            return
        if self.scaler is not None:
            # gradients carry the loss scale: test them, step with grad / scale unless one is inf / nan, adapt the scale
            ops.amp_adam_step(self.flat_p, self.flat_g, self.m, self.v, self.state, self.scaler, self.betas[0],
                              self.betas[1], self.eps, 1.0 / self.world_size)
        else:
            ops.adam_step(self.flat_p, self.flat_g, self.m, self.v, self.state, self.betas[0], self.betas[1], self.eps,
                          1.0 / self.world_size)
        self._steps_done += 1
        self.refresh_layouts()

    def refresh_layouts(self):
        """Re-derive the cached GEMM layouts from the current parameter values (also call this after modifying
        parameters outside the optimiser when replaying a captured graph)."""
        if self.layouts is not None:
            self.layouts.refresh()

    def step(self):
        if self.overlap:
            self._finish_exchange()
        else:
            self.exchange_all()
        self.adam()

    def state_dict(self):
        d = dict(m=self.m, v=self.v, state=self.state)
        if self.scaler is not None:
            d['scaler'] = self.scaler
        return d
