"""ctypes binding of the C-ABI kernel library declared in include/fsv2v.h.

The product path loads ``libfsv2v_hip.so`` (gfx950 code objects) and refuses to work without it: there is no
CPU fallback.  The only other library this module will ever load is the SIMT-emulated build of the *same* kernel
sources, and only when the test-suite asks for it explicitly with ``FSV2V_EMU=1`` (see tests/emu/hip_emu.h).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_is_emu = False

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_ll = ctypes.c_longlong
c_f = ctypes.c_float
c_ip = ctypes.POINTER(ctypes.c_int)

# name -> argtypes (all functions return int status; 0 == FSV_OK).  Order mirrors include/fsv2v.h.
_SIGS = {
    "fsv_conv_gather_fwd": [c_p, c_p, c_p, c_p, c_p,
                            c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                            c_i, c_ip, c_ip, c_i, c_i,
                            c_i, c_i, c_i, c_i, c_i, c_i,
                            c_i, c_ll, c_ll, c_i,
                            c_i, c_f, c_i, c_i, c_i, c_p, c_p, c_ll, c_i, c_p],
    "fsv_conv_wgrad": [c_p, c_p, c_p,
                       c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                       c_i, c_ip, c_ip, c_i, c_i,
                       c_i, c_i, c_ll, c_i, c_i, c_i, c_i, c_i, c_p],
    "fsv_bias_act": [c_p, c_p, c_ll, c_i, c_i, c_p],
    # narrow-operand (--amp) variants, csrc/conv_np.hip: the same arguments plus `mode` before the stream
    "fsv_conv_gather_fwd_np": [c_p, c_p, c_p, c_p, c_p,
                               c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                               c_i, c_ip, c_ip, c_i, c_i,
                               c_i, c_i, c_i, c_i, c_i, c_i,
                               c_i, c_ll, c_ll, c_i,
                               c_i, c_f, c_i, c_i, c_i, c_p, c_i, c_p],
    "fsv_conv_wgrad_np": [c_p, c_p, c_p,
                          c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                          c_i, c_ip, c_ip, c_i, c_i,
                          c_i, c_i, c_ll, c_i, c_i, c_i, c_i, c_i, c_p],
    "fsv_prep_weight_grouped": [c_p, c_p, c_p, c_p, c_p, c_i, c_p],
    "fsv_prep_weight": [c_p, c_p, c_p, c_i, c_i,
                        c_i, c_i, c_i, c_i, c_i, c_ip, c_ip,
                        c_i, c_i, c_ll, c_ll, c_p],
}


class FsvError(RuntimeError):
    pass


def emu_requested():
    return os.environ.get("FSV2V_EMU", "0") == "1"


def get_lib():
    global _lib, _is_emu
    if _lib is not None:
        return _lib
    if emu_requested():
        path = os.path.join(_HERE, "libfsv2v_emu.so")
        _is_emu = True
    else:
        # FSV2V_LIB: a diagnostic build of the SAME sources (build.build_hip_diag, tools/knockout.py); never set by the product
        path = os.environ.get("FSV2V_LIB") or os.path.join(_HERE, "libfsv2v_hip.so")
        _is_emu = False
    if not os.path.exists(path):
        raise FsvError("fsv2v kernel library %s is missing; run `python __graft_entry__.py` (build()) first. "
                       "There is no CPU fallback for the product path." % path)
    lib = ctypes.CDLL(path)
    for name, argtypes in _SIGS.items():
        fn = getattr(lib, name)      # AttributeError here == header/library drift; let it propagate loudly
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    _lib = lib
    return lib


def register_sigs(sigs):
    """Used by sibling modules to add entry points (keeps one table per kernel file small)."""
    _SIGS.update(sigs)
    global _lib
    if _lib is not None:
        for name, argtypes in sigs.items():
            fn = getattr(_lib, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int


def is_emu():
    get_lib()
    return _is_emu


def stream_ptr(t=None):
    if is_emu():
        return None
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check_device(*tensors):
    """The HIP library dereferences raw device pointers: refuse host tensors unless emulating."""
    emu = is_emu()
    for t in tensors:
        if t is None:
            continue
        if emu:
            if t.is_cuda:
                raise FsvError("emulated library got a device tensor")
        elif not t.is_cuda:
            raise FsvError("fsv2v HIP kernels need device tensors (got a CPU tensor); no CPU fallback exists")
        if t.dtype not in (torch.float32, torch.int32, torch.float64, torch.float16):
            raise FsvError("unsupported dtype %s" % t.dtype)


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def int_array(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def call_status(name, *args):
    """like call(), but returns the fsv_status instead of raising (entry points that may decline with FSV_ERR_UNSUPPORTED)"""
    return int(getattr(get_lib(), name)(*args))


def call(name, *args):
    rc = getattr(get_lib(), name)(*args)
    if rc != 0:
        raise FsvError("%s failed with fsv_status %d" % (name, rc))
