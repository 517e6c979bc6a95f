"""Which aten operations ran, and in which dtype - TEST INFRASTRUCTURE ONLY (oracle/make_golden.py mints
tests/golden/autocast_ops.json with it from the unmodified reference under torch.autocast; tests/test_golden.py runs the oracle under
it and compares the cast lists).  Nothing here is imported by the product."""
import torch


def census_mode():
    """TorchDispatchMode that records (operation, dtype of its first floating tensor argument) and the weight shapes of the
    contraction operations - what actually ran BELOW a cast policy (shared with the test, which runs the oracle under it)"""
    from torch.utils._python_dispatch import TorchDispatchMode

    class Census(TorchDispatchMode):
        def __init__(self):
            super().__init__()
            self.ops, self.convs, self.mms = {}, [], []

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = str(func.overloadpacket).replace('aten.', '')
            flt = [a for a in args if torch.is_tensor(a) and a.is_floating_point()]
            if flt and name not in ('_to_copy', 'detach', 'view', '_unsafe_view', 'expand', 'permute', 'slice', 'select', 't', 'transpose',
                                    'reshape', 'unsqueeze', 'squeeze', 'clone', 'alias', 'as_strided', 'empty_like', 'zeros_like',
                                    'ones_like', 'copy_', 'contiguous', 'unbind', 'split', 'split_with_sizes', 'narrow', 'cat', 'stack',
                                    'index', 'repeat', 'fill_', 'zero_', 'new_zeros', 'new_empty', 'lift_fresh', 'flatten'):
                d = str(flt[0].dtype).replace('torch.', '')
                self.ops.setdefault(name, {}).setdefault(d, 0)
                self.ops[name][d] += 1
                if name == 'convolution':
                    stride = args[3][0] if len(args) > 3 else 1
                    self.convs.append([list(args[1].shape), int(stride), d])
                elif name in ('mm', 'addmm', 'bmm', 'baddbmm', 'linear', 'matmul'):
                    self.mms.append([name, [list(a.shape) for a in flt[:3]], d])
            return out
    return Census()


