"""Static resource check of every gfx950 kernel in libfsv2v_hip.so (metadata of the embedded code object, tools/kernel_meta.py):
no kernel may touch scratch memory or spill registers - both silently turn a compute-bound kernel into a memory-bound one, and
neither shows up in the emulator tests - and the MFMA kernels must leave room for at least one workgroup per CU."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_no_kernel_uses_scratch_or_spills():
    import fsv2v_amd  # noqa: F401
    import kernel_meta
    build = importlib.import_module('few-shot-vid2vid_amd.build')
    ks = kernel_meta.kernels(build.build_hip())
    assert len(ks) >= 100, len(ks)
    # (scalar registers may spill: they go to lanes of a vector register, not to memory)
    bad = [(k['name'], k['private_segment_fixed_size'], k['vgpr_spill_count']) for k in ks
           if k['private_segment_fixed_size'] or k['vgpr_spill_count']]
    assert not bad, bad
    for k in ks:
        assert k['group_segment_fixed_size'] <= 160 * 1024 and k['vgpr_count'] <= 512, k      # gfx950: 160 KB of LDS per CU
