"""Run a block (or a hand-built operator chain) of the HIP path on cuda:0 and hand back NCHW fp32 results."""
from collections import OrderedDict

import torch

import salt_amd
from salt_amd.engine import Graph
from salt_amd.runtime import Engine

DEV = 'cuda:0'


class BlockRun:
    """Compile ``emit_fn(g, *acts) -> act`` over NCHW inputs; forward + backward with upstream grad ``gy``."""

    def __init__(self, module, inputs, emit_fn, train=True, dtype='f32', backward=True):
        self.module = module.to(DEV)
        self.eng = Engine(self.module, torch.device(DEV), dtype)
        g = Graph(self.eng, train)
        self.g = g
        self.xs = [g.alloc(tuple(x.shape), torch.float32) for x in inputs]
        for t, x in zip(self.xs, inputs):
            t.copy_(x)
        acts = [g.from_nchw(t) for t in self.xs]
        y = emit_fn(g, *acts)
        self.out = g.alloc((y.B, y.C, y.H, y.W), torch.float32)
        g.to_nchw(y, self.out)
        if train and backward:
            g.build_backward()
        elif train:
            g._resolve_lazies()                 # forward-only graph (SALT_FWD_BN_FOLD): what build_backward would have done to the forward program
        g.finalize()

    def forward(self):
        self.eng.refresh(self.g.train)
        self.g.fwd.run(side=self.eng.side_stream)
        torch.cuda.synchronize()
        return self.out.cpu()

    def backward(self, gy):
        self.g.dlogits.copy_(gy)
        self.g.bwd.run(side=self.eng.side_stream)
        torch.cuda.synchronize()
        gx = [t.cpu() for t in getattr(self.g, 'input_grads', [])]
        grads = OrderedDict()
        for name, p in self.module.named_parameters():
            if id(p) in self.eng._off:
                off, n = self.eng.grad_range(p)
                grads[name] = self.eng.grads[off:off + n].view(p.shape).cpu()
        return gx, grads


def load_into(module, sd):
    """Load a {key: tensor} dict (reference key names) into a module built from salt_amd classes."""
    own = module.state_dict()
    missing = [k for k in own if k not in sd]
    assert not missing, missing[:5]
    module.load_state_dict({k: sd[k] for k in own})
    return module
