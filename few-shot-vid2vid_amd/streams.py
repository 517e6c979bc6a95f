"""Independent branches of one forward pass on side HIP streams.

The few-shot generator is a wide graph of small problems: most launches of the reference encoders and weight generators
(16x16 ... 64x64 maps, weight-generator MLPs, per-channel reductions) fill a fraction of the 256 CUs, and the flow network /
warp / image-embedding branch does not depend on them.  `fork` issues such branches on different streams, so that they become
parallel branches of the captured hipGraph (and overlap in the eager loop as far as the host keeps up).  Autograd runs every
backward node on the stream of its forward op and orders the streams itself, so the backward pass overlaps the same way.

Measured on the C3 bench step (profiles/r02_notes.md): one coarse fork with few tensors crossing it pays (-3.6 ms of 60);
finer forks (the two reference encoders against each other, the loss terms against the discriminator pass) each cost 2-3 ms
with the ROCm 7.2 graph executor - every cross-stream edge autograd adds splits the graph's packet batches - and were dropped.

Results are bit-identical to the sequential order: the same kernels run on the same data, only concurrently; no kernel of the
library shares scratch memory with another launch (workspaces come from the stream-aware caching allocator).

FSV_BRANCH_STREAMS=0 restores the single-stream order (A/B measurements, debugging).
"""
import os

import torch

ENABLED = os.environ.get('FSV_BRANCH_STREAMS', '1') == '1'
_pool = {}
_busy = []          # side streams with an open (not yet joined) branch: nested forks take others


def _side_streams(device, n, avoid):
    got = _pool.setdefault(device, [])
    out, k = [], 0
    while len(out) < n:
        if k == len(got):
            got.append(torch.cuda.Stream(device=device))
        if got[k] != avoid and not any(got[k] is b for b in _busy):
            out.append(got[k])
        k += 1
    return out


def _record(obj, stream):
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _record(o, stream)
    elif isinstance(obj, dict):
        for o in obj.values():
            _record(o, stream)


def shared(make):
    """A device tensor that is created once and then read by kernels of ANY stream (zero constants, the ticket pool of the
    fused reductions): `make()` enqueues its fill on the current stream only, so the fill is waited for here - otherwise the
    first use on another stream races with it (seen on hardware: a first iteration whose branch zeroed the ticket pool
    while the other branch was already counting in it)."""
    t = make()
    if torch.is_tensor(t) and t.is_cuda:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("a shared device constant was first needed inside a graph capture: run one eager iteration first")
        torch.cuda.current_stream(t.device).synchronize()
    return t


CROSS = False       # set while two passes of one iteration are issued on different streams: values one pass caches for the
                    # other (model._memoised, ops._SpadeFn's fixed-weight operands) then carry an event

FORK_ROOT = None    # set by a caller that issues a pass on a SIDE stream (model.Vid2VidModel's twin generator passes): the branches
                    # of a fork inside that pass then start from this stream - the capture's origin - instead of the side stream.
                    # hipStreamEndCapture of ROCm 7.2 crashes on a branch of a branch (round 6: segmentation fault in capture_end,
                    # gone with the inner fork off); a sibling of the side stream is fine.  Only valid while the side stream has
                    # launched nothing since it was forked from FORK_ROOT that the branches read (the caller's responsibility).

_held = None        # while a `hold()` block is open: side streams that forks inside it took (they stay reserved until it closes)


class hold:
    """`with hold():` - the side streams that forks inside the block take stay reserved until the block closes.  A forward pass that
    is issued on one stream while ANOTHER pass of the same shape is issued next to it (model.Vid2VidModel's twin generator passes)
    must not hand its branch streams to that pass: the second pass's branches would queue behind the first's on the device."""

    def __enter__(self):
        global _held
        self.prev, _held = _held, []
        return self

    def __exit__(self, *exc):
        global _held
        for s in _held:
            if any(s is b for b in _busy):
                _busy.remove(s)
        _held = self.prev
        return False


def fork(ref, fns):
    """[f() for f in fns]; on a GPU fns[1:] run on side streams next to fns[0] on the current one and are joined before the
    return.  `ref` is any tensor of the pass (it names the device; CPU / emulated tensors run the branches in order)."""
    if not (ENABLED and torch.is_tensor(ref) and ref.is_cuda) or len(fns) < 2:
        return [f() for f in fns]
    cur = torch.cuda.current_stream(ref.device)
    sides = _side_streams(ref.device, len(fns) - 1, cur)
    outs = [None] * len(fns)
    _busy.extend(sides)
    try:
        for s, i in zip(sides, range(1, len(fns))):
            s.wait_stream(cur if FORK_ROOT is None else FORK_ROOT)
            with torch.cuda.stream(s):
                outs[i] = fns[i]()
        outs[0] = fns[0]()
    finally:
        for s in sides:
            if _held is not None:
                _held.append(s)          # stays in _busy until the enclosing hold() closes
            else:
                _busy.remove(s)
    for s, i in zip(sides, range(1, len(fns))):
        cur.wait_stream(s)
        _record(outs[i], cur)
    return outs


class Branch:
    """An open-ended branch: a side stream that stays reserved from `start` to `finish`, possibly across several entry points of
    the model (the discriminator step of an iteration next to the generator-mode forward pass, model.Vid2VidModel).  Work is put
    on it with `with branch.on():`; nested forks - on the branch and on the caller's stream - take other streams meanwhile."""

    def __init__(self, ref):
        self.stream = None
        if ENABLED and torch.is_tensor(ref) and ref.is_cuda:
            cur = torch.cuda.current_stream(ref.device)
            self.stream = _side_streams(ref.device, 1, cur)[0]
            _busy.append(self.stream)
            self.stream.wait_stream(cur)

    def on(self):
        import contextlib
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def guarded(self):
        """`on()` that gives the stream back (finish) if the block raises"""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            try:
                with self.on():
                    yield self
            except BaseException:
                self.finish()
                raise
        return ctx()

    def uses(self, obj):
        """tensors made on the caller's stream that the branch reads: their memory must not be recycled under it"""
        if self.stream is not None:
            _record(obj, self.stream)

    def finish(self, made=None):
        """the current stream waits for everything the branch has been given; `made`: tensors produced on the branch that the
        caller's stream reads from now on"""
        if self.stream is not None:
            cur = torch.cuda.current_stream(self.stream.device)
            cur.wait_stream(self.stream)
            _record(made, cur)
            if any(self.stream is b for b in _busy):
                _busy.remove(self.stream)
            self.stream = None
