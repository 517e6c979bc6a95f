"""In-process kernel timing for bench.py's `roofline` object.

When enabled, the host wrappers of the MFMA kernels bracket every launch with a pair of HIP events recorded on the
stream the kernel is launched on (torch.cuda.Event records on torch's current stream, which is the stream handed to
the C ABI).  `summary()` turns the pairs into per-kernel averages: algorithmic FLOPs per launch (2 * MAC of the
convolution / GEMM the launch computes) divided by the average launch duration.

The instrumented pass is eager and therefore host-bound: the GPU idles between launches and the per-launch brackets come
out a few per cent longer than the same kernels inside the replayed graph (rocprofv3 --kernel-trace of the bench command,
profiles/).  The bracketed average is nevertheless what `roofline.achieved` is priced on (`avg_launch_us`): it times every
launch where it sits in the step.  For the dominant kernel `summary()` also re-issues the step's launches of that kernel back to
back - same arguments, same stream, one event pair around the whole run, three repetitions after a warm-up; that figure has warm
caches and nothing between the launches, so it is reported beside the headline as an upper bound
(`replay_us_warm_cache_upper_bound`), never as the headline.
"""
import ctypes

import torch

from . import lib

_enabled = False
_only = None          # second pass of bench.py: bracket the launches of ONE kernel label only
_detail = False       # tools/shape_profile.py: append the GEMM shape to every label
_records = []          # (label, flops, ev0, ev1)
FP32_MFMA_PEAK = 157.3e12
F16_MFMA_PEAK = 2.5e15        # dense f16 / bf16 matrix peak (MI355X_MICROARCH.md; the 2:1-sparse figure is not used)

TILE_NAMES = {0: '128x128', 1: '128x64', 2: '128x32', 4: '64x64', 9: '64x128', 10: '64x128pf2', 11: '128x128pf2', 12: '128x64pf2',
              13: '64x128pf2af', 14: '128x128pf2af', 15: '128x64pf2af', 16: '128x128af', 17: '64x64af', 18: '128x32af', 20: '64x64pf2af', 21: '64x128lds', 22: '128x64lds', 27: '64x64lds'}


_events = True        # False: labels are produced (scopes are entered) but no HIP events are recorded - a stamped graph capture


def enable(detail=False, only=None, events=True):
    """only: bracket the launches with this label alone (no replay closures kept) - with ~130 event pairs instead of ~450 in the
    pass the eager step idles less between launches and the brackets read closer to the kernel's time inside the replayed graph"""
    global _enabled, _records, _detail, _only, _events
    _enabled, _records, _detail, _only, _events = True, [], detail, only, events


def disable():
    global _enabled
    _enabled = False


def enabled():
    return _enabled


def detail():
    return _detail


# ---- brackets inside a captured graph: device-side time stamps (csrc/stamp.hip) -----------------------------------------------------
# HIP events cannot be recorded into a replayable graph; a one-work-item kernel that writes the GPU's wall clock can.  stamp_begin(label,
# slots) makes every scope of `label` launch such a kernel in front of and behind its launch (slot 2i, 2i + 1 of a device buffer), plus
# `calibration` empty pairs per real pair (two stamps with nothing between them: the pair's own cost).  bench.py captures the step
# once more that way, replays it and prices the dominant kernel on the in-graph durations.
_stamp = None          # dict(label, buf, n, cal) while a stamped capture is being recorded

lib.register_sigs({"fsv_stamp": [ctypes.c_void_p, ctypes.c_void_p], "fsv_stamp_rate_khz": []})


def stamp_begin(label, max_launches, device):
    global _stamp
    buf = torch.zeros(4 * max_launches + 8, dtype=torch.int64, device=device)
    _stamp = dict(label=label, buf=buf, n=0, cap=max_launches)
    return buf


def stamp_end():
    global _stamp
    st, _stamp = _stamp, None
    return st


def _stamp_slot(st, k):
    return ctypes.c_void_p(st['buf'].data_ptr() + 8 * k)


def stamp_result(st):
    """(launches, average seconds per launch, average seconds of an empty pair) from the slots of the last replay"""
    torch.cuda.synchronize()
    n = st['n']
    if n == 0:
        return 0, 0.0, 0.0
    khz = int(lib.get_lib().fsv_stamp_rate_khz())
    if khz <= 0:
        return 0, 0.0, 0.0
    t = st['buf'][:4 * n].cpu().view(n, 4).double()
    full = (t[:, 1] - t[:, 0]).mean().item() / (khz * 1e3)
    empty = (t[:, 3] - t[:, 2]).mean().item() / (khz * 1e3)
    return n, full, empty


class scope:
    def __init__(self, label, flops, replay=None):
        """replay: callable that re-issues this launch (it must keep its operand tensors alive)"""
        self.label, self.flops, self.replay = label, flops, replay

    def __enter__(self):
        self.on = _enabled and _events and not lib.is_emu() and (_only is None or self.label == _only)
        self.st = _stamp if (_stamp is not None and self.label == _stamp['label'] and _stamp['n'] < _stamp['cap']) else None
        if self.st is not None:
            k = 4 * self.st['n']
            # an empty pair first (its second stamp also separates the launch from whatever ran before), then the real one
            lib.call("fsv_stamp", _stamp_slot(self.st, k + 2), lib.stream_ptr())
            lib.call("fsv_stamp", _stamp_slot(self.st, k + 3), lib.stream_ptr())
            lib.call("fsv_stamp", _stamp_slot(self.st, k), lib.stream_ptr())
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            _records.append((self.label, self.flops, self.e0, self.e1, None if _only is not None else self.replay))
        if self.st is not None:
            lib.call("fsv_stamp", _stamp_slot(self.st, 4 * self.st['n'] + 1), lib.stream_ptr())
            self.st['n'] += 1
        return False


def bracket_average(label):
    """(launches, average seconds) of the recorded launches of `label`; clears the records"""
    torch.cuda.synchronize()
    ts = [e0.elapsed_time(e1) * 1e-3 for lab, _, e0, e1, _ in _records if lab == label]
    _records.clear()
    return len(ts), (sum(ts) / len(ts) if ts else 0.0)


def thin_rule(mz, k):
    """csrc/conv_igemm.hip fsv_conv_thin: the size rule of the vector-ALU kernels for Cout <= 4 layers"""
    lib.register_sigs({"fsv_conv_thin_rule": [ctypes.c_int] * 2})
    return lib.call_status("fsv_conv_thin_rule", int(mz), int(k)) == 1


def conv_label(mz, cout, nchunks, nsamp, vec4, force_tile=-1, force_split=0, thin=0):
    """thin: K = taps * Cin when the caller's launch meets every condition of the thin-output dispatch inside
    fsv_conv_gather_fwd but the size rule (0: it does not)"""
    if thin and thin_rule(mz, thin):
        base = 'fsv_conv_thin_fwd_kernel<Cout%d>' % cout
        if _detail:
            base += ' M%d N%d K%d' % (mz, cout, nchunks * 32)
        return base
    lib.register_sigs({"fsv_conv_plan": [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_int)] * 2})
    tile, nsplit = ctypes.c_int(0), ctypes.c_int(1)
    lib.call("fsv_conv_plan", mz, cout, nchunks, nsamp, force_tile, force_split, ctypes.byref(tile), ctypes.byref(nsplit))
    base = 'fsv_conv_igemm_kernel<%s,V%d>' % (TILE_NAMES[tile.value], 4 if vec4 else 1)
    if _detail:
        base += ' M%d N%d K%d z%d split%d' % (mz, cout, nchunks * 32, nsamp, nsplit.value)
    return base


def summary():
    torch.cuda.synchronize()
    agg = {}
    for label, flops, e0, e1, _ in _records:
        a = agg.setdefault(label, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += flops
        a[2] += e0.elapsed_time(e1) * 1e-3
    by_kernel = {}
    for label, (n, fl, t) in agg.items():
        by_kernel[label] = dict(launches=n, avg_us=round(t / n * 1e6, 2), total_ms=round(t * 1e3, 3),
                                tflops=round(fl / t / 1e12, 2) if t > 0 else 0.0,
                                gflop_per_launch=round(fl / n / 1e9, 3))
    dominant = None
    if agg:
        lab = max(agg, key=lambda k: agg[k][2])
        n, fl, t = agg[lab]
        ach = fl / t / 1e12
        peak = FP32_MFMA_PEAK / 1e12
        if '[f16]' in lab or lab.startswith('fsv_hconv'):
            peak = F16_MFMA_PEAK / 1e12                 # half-precision / narrow-operand kernels against the dense 16-bit matrix peak
        elif '[bf16x3]' in lab:
            peak = F16_MFMA_PEAK / 3e12                 # three MFMAs per algorithmic product
        bracketed = t / n
        replays = [r[4] for r in _records if r[0] == lab and r[4] is not None]
        replay_us = None
        if len(replays) == n:
            reps = 3
            for r in replays:                            # warm-up: clocks, instruction cache
                r()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                for r in replays:
                    r()
            e1.record()
            torch.cuda.synchronize()
            replay_us = e0.elapsed_time(e1) * 1e3 / reps / n
        # Headline = the per-launch brackets of the instrumented pass: every launch timed where it sits in the step, with the
        # other kernels of the step between its launches (cold-ish caches).  It reads a few per cent LONG against the same
        # kernel inside the replayed graph (rocprofv3 kernel trace of the bench command, profiles/) because the eager pass idles
        # between launches; the back-to-back replay of the same launches reads a few per cent SHORT (warm L2 / MALL, nothing
        # between the launches) and is reported beside it as an upper bound, not as the headline.
        dominant = dict(kernel=lab, bound='mfma', achieved=round(ach, 2), peak=round(peak, 1), unit='TFLOP/s',
                        frac=round(ach / peak, 4), traffic=None, launches=n,
                        avg_launch_us=round(bracketed * 1e6, 2), bracketed_us=round(bracketed * 1e6, 2),
                        replay_us_warm_cache_upper_bound=None if replay_us is None else round(replay_us, 2),
                        gflop_per_launch=round(fl / n / 1e9, 3),
                        timing='HIP events around every launch of this kernel in an instrumented eager pass of the step '
                               '(bracketed_us: every MFMA kernel of the step bracketed; avg_launch_us, when bench.py ran its second '
                               'pass: this kernel alone bracketed); replay_us_warm_cache_upper_bound = the same launches re-issued '
                               'back to back')
    _records.clear()          # the replay closures pin every operand of the step
    return dict(dominant=dominant, by_kernel=by_kernel)
