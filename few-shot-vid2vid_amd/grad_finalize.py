"""Queue of deferred weight-gradient jobs, drained by ONE fsv_wgrad_finalize call per backward pass.

ops._ConvFn.backward leaves the weight gradient of a cached, sink-enabled parameter in the K-major layout the wgrad
GEMM produces and appends a job here instead of launching the re-layout (+ spectral-norm correction) itself;
FlatAdam drains the queue before the gradient exchange / Adam step (csrc/wgrad_finalize.hip).  The job table holds raw
device pointers of this pass's temporaries, so it is rebuilt per pass: geometry / block maps are cached per job
sequence, only the pointer array is re-uploaded - through kernel arguments (fsv_upload_i64), which is legal inside a
hipGraph capture and leaves no host buffer for a captured graph to depend on.
"""
import ctypes

import torch

from . import lib

c_p, c_i = ctypes.c_void_p, ctypes.c_int
lib.register_sigs({"fsv_wgrad_finalize": [c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_i, c_p],
                   "fsv_upload_i64": [c_p, c_p, c_i, c_p],
                   "fsv_gather_add": [c_p, c_i, c_p, c_i, c_p],
                   "fsv_colsum_plan": [c_i, c_i, ctypes.POINTER(c_i)],
                   "fsv_colsum_grouped": [c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p]})

DOT_CHUNK = 4096


def _i64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


class GradFinalizer:
    def __init__(self):
        self.jobs = []            # (entry, dwt, sink, sig, u, v)
        self.bias_jobs = []       # (dpre [P][C] NHWC rows, sink float[C]): bias gradients = column sums, grouped like the rest
        self._static = {}         # job-sequence signature -> (dims, taps, tmap_dot, nblk_dot, tmap_apply, nblk_apply)
        # per-pass arena for the K-major gradients: zeroed by ONE fill in begin_pass() instead of a zero-fill in front
        # of every split-K wgrad launch; sized from the previous pass (the first pass allocates per layer)
        self.arena = None
        self._arena_off = 0
        self._arena_need = 0

    def add(self, entry, dwt, sink, sig=None, u=None, v=None):
        self._arena_dev = dwt.device
        self.jobs.append((entry, dwt, sink, sig, u, v))

    def add_bias(self, dpre, sink):
        """bias gradient of a convolution: sink[c] += sum over the pixels of dpre[:, c] (dpre is kept alive until run())"""
        self.bias_jobs.append((dpre, sink))

    def _run_bias(self):
        jobs, self.bias_jobs = self.bias_jobs, []
        if not jobs:
            return
        dev = jobs[0][0].device
        capturing = (not lib.is_emu()) and torch.cuda.is_current_stream_capturing()
        shapes = tuple((t.numel() // t.shape[1], t.shape[1]) for t, _ in jobs)
        key = ('bias',) + shapes
        st = self._static.get(key)
        if st is None:
            if capturing:
                raise lib.FsvError("new bias-gradient job sequence inside a graph capture; run one eager step first")
            plans, tmap1, tmap2, off = [], [], [], 0
            out = (c_i * 5)()
            for j, (p_rows, c) in enumerate(shapes):
                lib.call("fsv_colsum_plan", p_rows, c, out)
                v, tx, nslabs, rpb, nch = [int(x) for x in out]
                plans.append((off, p_rows, c, rpb, nch, v))
                for ch in range(nch):
                    for sl in range(nslabs):
                        tmap1 += [j, ch, sl]
                for cb in range((c + 3) // 4):
                    tmap2 += [j, cb]
                off += nch * c
            mk = lambda vals: torch.tensor(vals, dtype=torch.int32).to(dev)
            st = self._static[key] = (plans, mk(tmap1), len(tmap1) // 3, mk(tmap2), len(tmap2) // 2, off)
        plans, tmap1, nblk1, tmap2, nblk2, ndoubles = st
        words = []
        uses = {}
        for _, sink in jobs:
            uses[sink.data_ptr()] = uses.get(sink.data_ptr(), 0) + 1
        for (t, sink), (off, p_rows, c, rpb, nch, v) in zip(jobs, plans):
            lib.check_device(t, sink)
            shared = 1 if uses[sink.data_ptr()] > 1 else 0            # a module used twice in the pass: atomic adds
            words += [t.data_ptr(), sink.data_ptr(), off, p_rows, c, rpb, nch, v | (shared << 8)]
        host = (ctypes.c_longlong * len(words))(*words)
        table = torch.empty(len(words), dtype=torch.int64, device=dev)
        part = torch.empty(max(ndoubles, 1), dtype=torch.float64, device=dev)
        lib.call("fsv_upload_i64", lib.ptr(table), host, len(words), lib.stream_ptr())
        lib.call("fsv_colsum_grouped", lib.ptr(table), len(jobs), lib.ptr(tmap1), nblk1, lib.ptr(tmap2), nblk2,
                 lib.ptr(part), lib.stream_ptr())

    def zero_arena(self):
        """the arena's fill ahead of time (FlatAdam.zero_early, on a side stream); begin_pass(prezeroed=True) then skips it unless
        it has to allocate a larger arena"""
        if self.arena is not None:
            self.arena.zero_()

    def begin_pass(self, prezeroed=False):
        """Called by FlatAdam.zero_grad right before a backward pass."""
        self.jobs = []
        self.bias_jobs = []
        capturing = (not lib.is_emu()) and torch.cuda.is_current_stream_capturing()
        if self._arena_need and not capturing and (self.arena is None or self.arena.numel() < self._arena_need):
            self.arena = None
            self.arena = torch.empty(self._arena_need, dtype=torch.float32, device=self._arena_dev)
            prezeroed = False
        if self.arena is not None and not prezeroed:
            self.arena.zero_()
        self._arena_off = 0
        self._arena_need = 0

    def take(self, nfloats):
        """A zeroed [1, nfloats] slice of the arena, or None when it does not fit (caller allocates and zero-fills)."""
        nfloats = (nfloats + 63) // 64 * 64
        self._arena_need += nfloats
        if self.arena is None or self._arena_off + nfloats > self.arena.numel():
            return None
        out = self.arena[self._arena_off:self._arena_off + nfloats]
        self._arena_off += nfloats
        return out

    def pending(self):
        return bool(self.jobs) or bool(self.bias_jobs)

    def _build_static(self, dev):
        dims, taps, tmap_dot, tmap_apply = [], [], [], []
        sinks = {}
        for (e, dwt, sink, sig, u, v) in self.jobs:
            sinks[sink.data_ptr()] = sinks.get(sink.data_ptr(), 0) + 1
        for j, (e, dwt, sink, sig, u, v) in enumerate(self.jobs):
            _, d, lo, hi = e.jobs[0]                      # the forward layout: same geometry as dwt
            cout, cinp, cin, kh, kw, ntaps, kpad, ldw, _mode = d
            shared = 1 if sinks[sink.data_ptr()] > 1 else 0
            dims += [cout, cinp, cin, kh, kw, ntaps, ldw, shared]
            taps += [_i64(lo), _i64(hi)]
            if sig is not None:
                total = ntaps * cinp * ldw
                for ch in range((total + DOT_CHUNK - 1) // DOT_CHUNK):
                    tmap_dot += [j, ch]
            ci_t = 32 if ntaps <= 8 else 16
            for a in range((cout + 31) // 32):
                for b in range((cinp + ci_t - 1) // ci_t):
                    tmap_apply += [j, a, b]
        mk = lambda vals, dt: torch.tensor(vals if vals else [0], dtype=dt).to(dev)
        return (mk(dims, torch.int32), mk(taps, torch.int64), mk(tmap_dot, torch.int32), len(tmap_dot) // 2,
                mk(tmap_apply, torch.int32), len(tmap_apply) // 3)

    def gather_dense(self, pairs):
        """pairs: [(src, dst)] contiguous fp32 tensors of equal numel; dst += src for all of them in one launch."""
        pairs = [(a, b) for a, b in pairs if a.numel() > 0]
        if not pairs:
            return
        dev = pairs[0][1].device
        capturing = (not lib.is_emu()) and torch.cuda.is_current_stream_capturing()
        key = ('dense',) + tuple(a.numel() for a, _ in pairs)
        st = self._static.get(key)
        if st is None:
            if capturing:
                raise lib.FsvError("new dense-gradient set inside a graph capture; run one eager step first")
            tmap = []
            for j, (a, _) in enumerate(pairs):
                for ch in range((a.numel() + 4095) // 4096):
                    tmap += [j, ch]
            st = self._static[key] = (torch.tensor(tmap, dtype=torch.int32).to(dev), len(tmap) // 2)
        tmap, nblk = st
        words = []
        for a, b in pairs:
            lib.check_device(a, b)
            words += [a.data_ptr(), b.data_ptr(), a.numel()]
        host = (ctypes.c_longlong * len(words))(*words)
        table = torch.empty(len(words), dtype=torch.int64, device=dev)
        lib.call("fsv_upload_i64", lib.ptr(table), host, len(words), lib.stream_ptr())
        lib.call("fsv_gather_add", lib.ptr(table), len(pairs), lib.ptr(tmap), nblk, lib.stream_ptr())

    def run(self):
        self._run_bias()
        if not self.jobs:
            return
        dev = self.jobs[0][1].device
        capturing = (not lib.is_emu()) and torch.cuda.is_current_stream_capturing()
        sig_key = tuple((id(e), s is not None) for (e, _, _, s, _, _) in self.jobs)
        st = self._static.get(sig_key)
        if st is None:
            if capturing:
                raise lib.FsvError("new weight-gradient job sequence inside a graph capture; run one eager step first")
            if len(self._static) > 8:
                self._static.clear()
            st = self._static[sig_key] = self._build_static(dev)
        dims, taps, tmap_dot, nblk_dot, tmap_apply, nblk_apply = st
        ptrs = []
        for (e, dwt, sink, sig, u, v) in self.jobs:
            sn = sig is not None
            ptrs += [dwt.data_ptr(), e.fwd[0].data_ptr() if sn else 0, sink.data_ptr(),
                     u.data_ptr() if sn else 0, v.data_ptr() if sn else 0, sig.data_ptr() if sn else 0]
        host = (ctypes.c_longlong * len(ptrs))(*ptrs)
        d_ptrs = torch.empty(len(ptrs), dtype=torch.int64, device=dev)
        lib.call("fsv_upload_i64", lib.ptr(d_ptrs), host, len(ptrs), lib.stream_ptr())
        dots = torch.empty(len(self.jobs), dtype=torch.float64, device=dev)
        lib.call("fsv_wgrad_finalize", lib.ptr(d_ptrs), lib.ptr(dims), lib.ptr(taps), lib.ptr(dots), len(self.jobs),
                 lib.ptr(tmap_dot), nblk_dot, lib.ptr(tmap_apply), nblk_apply, lib.stream_ptr())
        self.jobs = []
