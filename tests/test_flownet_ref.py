"""The FlowNet2 teacher's three native operators pinned to the reference's OWN kernel code: oracle/build_ref.py compiles the
`__global__` templates of correlation_cuda_kernel.cu / resample2d_kernel.cu / channelnorm_kernel.cu for the host from the
reference tree (SIMT fibre emulator underneath, launch geometry of the reference's host wrappers) into
oracle/_ref/libflownet2_ref.so.  Held to it here: the Python restatement (oracle/flownet_oracle.py), the HIP kernels of
csrc/flownet_ops.hip on the emulator, and - on the GPU box, where the prebuilt library travels with the snapshot - the HIP kernels
on the MI355X.  Skipped where neither the reference tree nor a prebuilt library exists."""
import pytest
import torch

import op_checks as oc


def _ref():
    from oracle import flownet_ref
    if flownet_ref.load() is None:
        pytest.skip("no reference tree to build from and no prebuilt oracle/_ref/libflownet2_ref.so")
    return flownet_ref


def test_restatement_equals_the_reference_kernels():
    R = _ref()
    from oracle import flownet_oracle as FO
    g = torch.Generator().manual_seed(404)
    # FlowNetC's configuration (FlowNetC.py:28-31: pad 20, kernel 1, displacement 20, strides 1 / 2) and odd geometries
    for (n, c, h, w, pad, md, s1, s2) in [(1, 256, 6, 10, 20, 20, 1, 2), (2, 70, 9, 37, 20, 20, 1, 2), (1, 64, 12, 40, 4, 4, 1, 1),
                                           (1, 5, 11, 13, 6, 6, 2, 3)]:
        f1, f2 = torch.randn(n, c, h, w, generator=g), torch.randn(n, c, h, w, generator=g)
        a, b = R.correlation(f1, f2, pad, 1, md, s1, s2), FO.correlation(f1, f2, pad, 1, md, s1, s2)
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 1e-6 * max(float(a.abs().max()), 1.0)        # summation order over the channels
    for (n, c, h, w, mag) in [(2, 3, 17, 23, 6.0), (1, 2, 8, 64, 80.0), (1, 3, 33, 31, 2.5)]:
        img = torch.randn(n, c, h, w, generator=g)
        flow = (torch.rand(n, 2, h, w, generator=g) - 0.5) * mag
        flow[:, :, 0] = torch.round(flow[:, :, 0])
        assert torch.equal(R.resample2d(img, flow), FO.resample2d(img, flow))              # bit for bit
    for shp in [(2, 7, 5, 9), (1, 3, 16, 16), (1, 2, 4, 4)]:
        x = torch.randn(*shp, generator=g) * 3
        assert torch.equal(R.channelnorm(x), FO.channelnorm(x))


def test_emulated_hip_kernels_equal_the_reference_kernels(emu_lib):
    oc.check_flownet_ops(torch.device('cpu'), checker=_ref())


@pytest.mark.gpu
def test_hip_kernels_equal_the_reference_kernels(hip_lib):
    oc.check_flownet_ops(torch.device('cuda:0'), seed=19, checker=_ref())
