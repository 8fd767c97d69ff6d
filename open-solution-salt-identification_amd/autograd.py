"""torch.autograd bridge: lets reference-style code (``out = model(X); loss = fn(out, t); loss.backward()``,
models.py:121-134) drive the compiled HIP programs.  The fused path in models.SegmentationModel._fit_loop
does not go through autograd at all; this exists so that the nn.Module surface stays a drop-in."""
import torch

from ._abi import SaltError


class HipNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        eng = module.engine(x.device)
        net = eng.forward(x, True)
        ctx.net, ctx.eng = net, eng
        return net.logits.clone()

    @staticmethod
    def backward(ctx, dlogits):
        net, eng = ctx.net, ctx.eng
        if net.dlogits is None:
            raise SaltError('network was compiled without a backward program')
        net.dlogits.copy_(dlogits)
        net.bwd.run(side=eng.side_stream)
        grads = []
        for p in eng.live_params:
            off, n = eng.grad_range(p)
            view = eng.grads[off:off + n].view(p.shape)
            # gradients are already in place when p.grad still aliases the flat gradient buffer
            grads.append(None if (p.grad is not None and p.grad.data_ptr() == view.data_ptr()) else view.clone())
        return (None, None) + tuple(grads)


class _NativeLoss(torch.autograd.Function):
    """Hand-written HIP loss kernels as differentiable torch functions of fp32 NCHW logits."""

    @staticmethod
    def forward(ctx, logits, target, kind):
        from . import losses
        loss, dl = losses.native_loss(logits.detach(), target, kind, want_grad=True)
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None
