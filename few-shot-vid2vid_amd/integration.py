"""Drop-in wiring for the reference's train.py loop (walk-through in INTEGRATION.md).

The reference has no plugin / FFI registry - its operator seam is Python name binding - so the integration is a
set of name re-bindings applied before `train.py` builds its model:

    import fsv2v_amd; fsv2v_amd.integration.patch_reference()     # top of train.py, nothing else changes

After that `models.models.create_model(opt, epoch)` returns this package's Vid2VidModel (behind the `.module`
attribute train.py expects) with flat-buffer Adam optimisers, `models.loss_collector.loss_backward` is ours, and
the four namespaces that imported `batch_conv` / `resample` by name see the HIP versions.
"""
import sys

import torch

from . import lib as _lib
from . import model as _model
from . import networks as _networks
from . import ops as _ops


class _ModuleHandle(torch.nn.Module):
    """train.py / trainer.py talk to `model.module.*` (DataParallel convention, models/models.py:79-117)."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def create_model(opt, epoch=0):
    """models/models.py:16-38: returns (model, flowNet, [optimizer_G, optimizer_D])."""
    if len(opt.gpu_ids) and (torch.cuda.is_available() or not _lib.is_emu()):
        device = torch.device('cuda', opt.gpu_ids[0])
    else:
        # host tensors exist for the emulated kernel library only (FSV2V_EMU=1, the test-suite's explicit opt-in); the product
        # library refuses them (lib.check_device), so there is no silent CPU path behind this branch
        device = torch.device('cpu')
    # Vid2VidModel.initialize(opt, epoch) already started temporal when the run resumes past the single-frame epochs
    # (base_model.py:213-215); the checkpoints are read once the networks sit on their device and before the optimisers
    # copy the parameters into their flat buffers (vid2vid_model.py:44 `self.load_networks()`)
    m = _model.create_model(opt, epoch, device)
    m.load_networks()
    world = torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1
    opt_G, opt_D = m.build_optimizers(world_size=world)
    flow_net = None
    if not opt.no_flow_gt:
        # models/models.py:31-35: the FlowNet2 teacher.  Its checkpoint is looked up where the reference keeps it
        # (models/flownet.py:28); without it the teacher would run on random weights, which is refused.
        from .flownet2 import FlowNet
        import os
        ckpt = getattr(opt, 'flownet2_checkpoint', 'models/networks/flownet2_pytorch/FlowNet2_checkpoint.pth.tar')
        if not os.path.exists(ckpt):
            raise FileNotFoundError("FlowNet2 checkpoint %s not found: pass --no_flow_gt or provide it" % ckpt)
        flow_net = FlowNet(opt)
        flow_net.flowNet.load_state_dict(torch.load(ckpt, map_location='cpu')['state_dict'])
        flow_net = _ModuleHandle(flow_net.to(device))
    return _ModuleHandle(m), flow_net, [opt_G, opt_D]


def init_dist(launcher='pytorch', backend='nccl', **kwargs):
    """util/distributed.py:15-26, made to work (the reference's raises on its first line): one process per GPU, launched
    by torchrun - bind this process to its GPU, join the process group (backend 'nccl' is RCCL over xGMI on ROCm), and
    de-correlate the per-rank random streams like the reference intends (`set_random_seed(get_rank())` :20; the model
    weights are still initialised under seed 0 on every rank, vid2vid_model.py:27).  Returns the GPU index."""
    import os
    import random
    import numpy as np
    import torch.distributed as dist
    # dmabuf IPC is the only mode the host driver supports for RCCL's cross-process buffers
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    gpu_id = 0
    if torch.cuda.is_available():
        local = os.environ.get('LOCAL_RANK')
        gpu_id = int(local) if local is not None else int(os.environ.get('RANK', '0')) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(gpu_id)
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(backend=backend, **kwargs)
    seed = dist.get_rank()
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return gpu_id


def batch_conv(x, weight, bias=None, stride=1, group_size=-1):
    """models/networks/base_network.py:56-71 on the per-sample gather-GEMM: stride 1 (every call site under the flags of
    the shipped scripts) and stride 2 (generator.py:552, architecture.py:35).  The transposed form (stride < 1) and grouped
    weights (group_size != -1) are not built and raise instead of silently computing something else."""
    if isinstance(weight, (list, tuple)):
        weight, bias = weight[0], weight[1]
    if weight is None:
        return x
    if stride not in (1, 2):
        raise NotImplementedError("batch_conv: stride %r (1 and 2 are built)" % (stride,))
    if group_size != -1:
        raise NotImplementedError("batch_conv: group_size %r" % (group_size,))
    return _ops.batch_conv(x, weight, bias, stride=stride)


def patch_reference():
    """Re-bind the reference's names to this package.  The reference tree must already be importable."""
    import importlib
    mm = importlib.import_module('models.models')
    lc = importlib.import_module('models.loss_collector')
    nets = importlib.import_module('models.networks')
    patched = []
    mm.create_model = create_model; patched.append('models.models.create_model')
    mm.Vid2VidModel = _model.Vid2VidModel; patched.append('models.models.Vid2VidModel')
    lc.loss_backward = _model.loss_backward; patched.append('models.loss_collector.loss_backward')
    nets.define_G = _networks.define_G; nets.define_D = _networks.define_D
    patched += ['models.networks.define_G', 'models.networks.define_D']
    for modname in ('models.networks.base_network', 'models.networks.generator', 'models.networks.normalization',
                    'models.networks.architecture', 'models.loss_collector'):
        mod = sys.modules.get(modname) or importlib.import_module(modname)
        if hasattr(mod, 'batch_conv'):
            mod.batch_conv = batch_conv
            patched.append(modname + '.batch_conv')
        if hasattr(mod, 'resample'):
            mod.resample = _ops.resample
            patched.append(modname + '.resample')
    try:                                   # `--distributed`: train.py:22-23 calls util.distributed.init_dist()
        ud = importlib.import_module('util.distributed')
        ud.init_dist = init_dist; patched.append('util.distributed.init_dist')
    except ImportError:
        pass
    # train.py binds these names at import time (train.py:13-16)
    tr = sys.modules.get('__main__')
    for name, obj in (('create_model', create_model), ('loss_backward', _model.loss_backward), ('init_dist', init_dist)):
        if tr is not None and hasattr(tr, name):
            setattr(tr, name, obj)
    return patched
