"""Persistent K-major weight layouts, refreshed by ONE grouped launch per optimiser step.

The gather-GEMM kernels read weights as wt[(tap, ci)][co] (forward) and wt[(tap, co)][ci] (data gradient, one layout
per stride parity class).  Parameters only change in the optimiser step, so instead of re-arranging every layer's
weights on every use (~400 small launches per training step) the optimiser owns a cache of the layouts and rewrites
all of them with a single `fsv_prep_weight_grouped` launch right after the fused Adam kernel.  The spectral-norm
1/sigma (which changes every forward) is not baked into the layouts: the GEMM epilogue applies it (`wscale`).

Entries are registered lazily the first time a parameter goes through ops._ConvFn (the geometry is only known there).
Parameters modified behind the optimiser's back (load_state_dict, copy_) are detected through `Tensor._version` on the
eager path; a captured hipGraph cannot see that, so call `FlatAdam.refresh_layouts()` after such a modification.
"""
import torch

from . import lib


def _pack_taps(khs, kws):
    lo = hi = 0
    for j, (a, b) in enumerate(zip(khs, kws)):
        code = (a | (b << 4)) & 0xff
        if j < 8:
            lo |= code << (8 * j)
        else:
            hi |= code << (8 * (j - 8))
    return lo, hi


def _i64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _ceil(a, b):
    return (a + b - 1) // b * b


def _pack_masks(mhs, mws):
    """summed-tap layouts: per output tap a 4-bit mask over kh and one over kw (csrc/conv_igemm.hip fsv_prep_pick)"""
    return _pack_taps(mhs, mws)


# per axis (r, tap) -> mask over the 3x3 kernel's index: ops._SUBPIXEL_ROWS (W0 | W1 + W2 || W0 + W1 | W2) and
# ops._DGRAD_ROWS (W0, W0 + W1, W1 + W2, W2) as bit masks
SUBPIXEL_MASKS = (0b001, 0b110, 0b011, 0b100)
DGRAD_MASKS = (0b001, 0b011, 0b110, 0b100)


class _Entry:
    __slots__ = ("weight", "key", "src_ptr", "version", "fwd", "dgrad", "jobs", "up_fwd", "up_dgrad")


class LayoutCache:
    def __init__(self):
        self.entries = []
        # bumped by every refresh: derived operands that are NOT in the tables (the combined [gamma | beta] operands of fixed SPADE
        # weights, ops._SpadeFn) are valid for the epoch they were built in - the parameters only change in the optimiser step,
        # which ends with a refresh
        self.epoch = 0
        self._tables = None
        self._nblocks = 0
        self._dirty = True
        # half-precision twins (hconv.py: N-major half operands of the `--amp` kernels) of the layouts above, created the first
        # time a half activation meets a layout (conv.half_twin) and rewritten right behind the layouts themselves
        self.half_jobs = []
        self._half_tables = None
        self._half_dirty = True

    # ------------------------------------------------------------------------------------------ registration
    def lookup(self, weight, shape4, geom, cpad, up=False):
        """shape4: (Cout, Cin_real, KH, KW) of the parameter seen as a convolution weight.  up: the layer convolves
        nearest_x2(x) (generator.py:489-493, 559-563) - a 3x3 / stride 1 / padding 1 layer then also keeps the two summed-tap
        layouts of DESIGN.md 4d (`up_fwd`: the four sub-pixel classes of the forward pass, 16 taps class-major; `up_dgrad`: the 4x4
        stride-2 data gradient), refreshed with the others instead of being rebuilt on every call."""
        key = (tuple(shape4), geom.kh, geom.kw, geom.stride, geom.pad, cpad)
        e = getattr(weight, "_fsv_layout", None)
        if e is None or e.src_ptr != weight.data_ptr() or e.key != key:
            if e is not None and e.key != key:
                return None                   # one parameter used with two geometries: leave it on the per-call path
            if not lib.is_emu() and torch.cuda.is_current_stream_capturing():
                return None                   # cannot grow the tables inside a capture; this call re-arranges itself
            e = self._register(weight, key, shape4, geom, cpad)
        if (up and e.up_fwd is None and cpad == 0 and (geom.kh, geom.kw, geom.stride, geom.pad) == (3, 3, 1, 1) and shape4[1] % 4 == 0
                and (lib.is_emu() or not torch.cuda.is_current_stream_capturing())):
            self._add_up_jobs(e, shape4)
        if e.version != weight._version:
            self._refresh_entry(e)
        return e

    def _add_up_jobs(self, e, shape4):
        cout, cin, kh, kw = shape4
        dev = e.weight.device

        def job(mode, mhs, mws):
            ntaps = len(mhs)
            rowlen, ncols = (cout, cin) if mode == 1 else (cin, cout)
            kpad = _ceil(ntaps * rowlen, 32)
            ldw = _ceil(ncols, 32)
            wt = torch.zeros((1, kpad, ldw), dtype=torch.float32, device=dev)
            wt._fsv_owner = self
            lo, hi = _pack_masks(mhs, mws)
            e.jobs.append((wt, [cout, cin, cin, kh, kw, ntaps, kpad, ldw, mode | 4], lo, hi))
            return wt, ldw
        # forward: 16 taps in class-major order - class (ry, rx) = K rows [c * 4 cin, (c + 1) * 4 cin), tap (iy, ix) inside it
        cls = [(ry, rx) for ry in (0, 1) for rx in (0, 1)]
        ph = [2 * ry + iy for ry, rx in cls for iy in (0, 1) for _ in (0, 1)]
        pw = [2 * rx + ix for ry, rx in cls for _ in (0, 1) for ix in (0, 1)]
        e.up_fwd = job(0, [SUBPIXEL_MASKS[p] for p in ph], [SUBPIXEL_MASKS[q] for q in pw])
        # data gradient: V[a][b], a, b = 0 .. 3 (ops._up_dgrad_weight: khs = a, kws = b)
        e.up_dgrad = job(1, [DGRAD_MASKS[a] for a in range(4) for _ in range(4)], [DGRAD_MASKS[b] for _ in range(4) for b in range(4)])
        self._dirty = True
        e.version = None                      # the new layouts are filled by the refresh that follows

    def _register(self, weight, key, shape4, geom, cpad):
        old = getattr(weight, "_fsv_layout", None)
        if old is not None and old in self.entries:
            self.entries.remove(old)
            stale = [j[0] for j in old.jobs]
            if any(any(wt is t for t in stale) for wt, _ in self.half_jobs):
                self.half_jobs = [(wt, wh) for wt, wh in self.half_jobs if not any(wt is t for t in stale)]
                self._half_dirty = True
        cout, cin, kh, kw = shape4
        cinp = cin + cpad
        dev = weight.device
        e = _Entry()
        e.weight, e.key, e.src_ptr, e.version = weight, key, weight.data_ptr(), None
        e.jobs = []
        e.up_fwd = e.up_dgrad = None

        def job(mode, khs, kws):
            ntaps = len(khs)
            rowlen, ncols = (cout, cinp) if mode == 1 else (cinp, cout)
            kpad = _ceil(max(ntaps * rowlen, 1), 32)
            ldw = _ceil(ncols, 32)
            # zero once: the refresh kernel only rewrites the valid region, padding rows / columns stay zero
            wt = torch.zeros((1, kpad, ldw), dtype=torch.float32, device=dev)
            wt._fsv_owner = self
            lo, hi = _pack_taps(khs, kws)
            e.jobs.append((wt, [cout, cinp, cin, kh, kw, ntaps, kpad, ldw, mode], lo, hi))
            return wt, ldw
        e.fwd = job(0, geom.khs, geom.kws)
        e.dgrad = []
        for c in geom.dgrad_classes:
            e.dgrad.append(job(1, c['khs'], c['kws']) if c['khs'] else None)
        weight._fsv_layout = e
        self.entries.append(e)
        self._dirty = True
        return e

    # ------------------------------------------------------------------------------------------ refresh
    def _launch(self, tables, nblocks):
        src, dst, dims, taps, tmap = tables
        lib.call("fsv_prep_weight_grouped", lib.ptr(src), lib.ptr(dst), lib.ptr(dims), lib.ptr(taps), lib.ptr(tmap),
                 nblocks, lib.stream_ptr())

    @staticmethod
    def _build(entries, dev):
        src, dst, dims, taps, tmap = [], [], [], [], []
        for e in entries:
            first = len(src)
            for (wt, d, lo, hi) in e.jobs:
                src.append(e.src_ptr)
                dst.append(wt.data_ptr())
                dims += d
                taps += [_i64(lo), _i64(hi)]
            # one workgroup per source tile writes ALL layouts of the weight (the jobs share Cout / Cin_pad / Cin_real / KH / KW)
            d = e.jobs[0][1]
            assert all(j[1][:5] == d[:5] for j in e.jobs)
            ci_t = 32 if d[3] * d[4] <= 8 else 16            # source taps KH * KW
            for a in range((d[0] + 31) // 32):
                for b in range((d[1] + ci_t - 1) // ci_t):
                    tmap += [first, len(e.jobs), a, b]
        mk = lambda v, dt: torch.tensor(v, dtype=dt).to(dev)
        return ((mk(src, torch.int64), mk(dst, torch.int64), mk(dims, torch.int32), mk(taps, torch.int64),
                 mk(tmap, torch.int32)), len(tmap) // 4)

    def add_half(self, wt):
        """persistent half twin of the cached layout wt (one conversion now; from here on refresh() rewrites it)"""
        from . import hconv
        if not lib.is_emu() and torch.cuda.is_current_stream_capturing():
            raise lib.FsvError("a half twin of a cached weight layout is needed inside a graph capture; run one eager step first")
        wh, k64, nrows = hconv.prep_weight_h(wt)
        self.half_jobs.append((wt, wh))
        self._half_dirty = True
        return wh, k64, nrows

    def _refresh_half(self, only=None):
        if not self.half_jobs:
            return
        from . import hconv
        if only is not None:
            jobs = [(wt, wh) for wt, wh in self.half_jobs if any(wt is j[0] for j in only.jobs)]
            if jobs:
                tables = hconv._prep_tables(jobs, jobs[0][0].device)
                hconv.launch_prep(tables)
            return
        if self._half_dirty:
            if not lib.is_emu() and torch.cuda.is_current_stream_capturing():
                raise lib.FsvError("weight-layout cache changed inside a graph capture; run one eager step first")
            self._half_tables = hconv._prep_tables(self.half_jobs, self.half_jobs[0][0].device)
            self._half_dirty = False
        hconv.launch_prep(self._half_tables)

    def _refresh_entry(self, e):
        tables, nblocks = self._build([e], e.weight.device)
        self._launch(tables, nblocks)
        self._refresh_half(only=e)
        e.version = e.weight._version
        if not lib.is_emu():
            # the tables are temporaries: keep them alive until the launch has consumed them
            torch.cuda.current_stream().synchronize()

    def refresh_split(self, part, lo, hi):
        """Rewrite the layouts of the weights that lie in bytes [lo, hi) of the optimiser's flat parameter buffer (part 0) or
        outside of it (part 1): an optimiser step issued in two pieces refreshes each piece's layouts behind its own Adam
        launch (FlatAdam.step_stage2_early).  fp32 layouts only (the caller keeps `--amp` on the one-piece step)."""
        if not self.entries:
            return
        key = (lo, hi)
        if self._dirty or getattr(self, '_split_key', None) != key:
            if not lib.is_emu() and torch.cuda.is_current_stream_capturing():
                raise lib.FsvError("weight-layout cache changed inside a graph capture; run one eager step first")
            inside = [e for e in self.entries if lo <= e.src_ptr < hi]
            outside = [e for e in self.entries if not (lo <= e.src_ptr < hi)]
            dev = self.entries[0].weight.device
            self._split_tables = [self._build(es, dev) if es else None for es in (inside, outside)]
            self._split_entries = (inside, outside)
            self._split_key = key
            if self._dirty:           # keep the whole-cache tables in step (refresh() may still be called)
                self._tables, self._nblocks = self._build(self.entries, dev)
                self._dirty = False
        self.epoch += 1
        t = self._split_tables[part]
        if t is not None:
            self._launch(t[0], t[1])
        for e in self._split_entries[part]:
            e.version = e.weight._version

    def refresh(self):
        """Rewrite every registered layout from the current parameter values (one launch)."""
        self.epoch += 1
        if not self.entries:
            return
        if self._dirty:
            if not lib.is_emu() and torch.cuda.is_current_stream_capturing():
                raise lib.FsvError("weight-layout cache changed inside a graph capture; run one eager step first")
            self._tables, self._nblocks = self._build(self.entries, self.entries[0].weight.device)
            self._dirty = False
        self._launch(self._tables, self._nblocks)
        self._refresh_half()
        for e in self.entries:
            e.version = e.weight._version
