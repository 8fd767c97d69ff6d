"""Oracle: losses and the optimizer step (TEST INFRASTRUCTURE ONLY).

Reference anchors (relative to /root/reference/common_blocks):
  lovasz_losses.py:21-33    lovasz_grad            -> lovasz_grad
  lovasz_losses.py:81-115   lovasz_hinge(_flat)    -> lovasz_hinge   (F.elu variant, both channels flattened)
  lovasz_losses.py:133-145  flatten_binary_scores  -> the .reshape(-1) below
  models.py:326-328         lovasz_loss            -> lovasz_loss    (target cast .long())
  models.py:315-323         DiceLoss               -> dice_loss
  models.py:361-388         multiclass_dice_loss   -> multiclass_dice_loss
  models.py:331-340         mixed_dice_bce_loss    -> mixed_dice_bce_loss
  models.py:74-75,289-297   Adam + L2 weight decay -> adam_l2_step (torch.optim.Adam semantics, un-vendored)

The implementations are torch-differentiable (autograd gives the reference gradients);
``lovasz_hinge_grad_closed_form`` restates the gradient analytically so the HIP
backward can be checked without autograd and tie handling can be reasoned about.
"""
import torch
import torch.nn.functional as F


def lovasz_grad(gt_sorted):
    """Gradient of the Lovasz extension of the Jaccard loss w.r.t. sorted errors (Alg. 1)."""
    gt = gt_sorted.float()
    total = gt.sum()
    inter = total - gt.cumsum(0)
    union = total + (1.0 - gt).cumsum(0)
    jac = 1.0 - inter / union
    if gt.numel() > 1:
        jac = torch.cat([jac[:1], jac[1:] - jac[:-1]])
    return jac


def _lovasz_hinge_flat(logits, labels):
    if labels.numel() == 0:
        return logits.sum() * 0.0
    signs = 2.0 * labels.float() - 1.0
    errors = 1.0 - logits * signs
    errors_sorted, perm = torch.sort(errors, dim=0, descending=True)
    grad = lovasz_grad(labels[perm.detach()])
    return torch.dot(F.elu(errors_sorted), grad)


def lovasz_hinge(logits, labels, per_image=True):
    """Binary Lovasz hinge; logits/labels [B, ...]; per image the trailing dims are flattened together."""
    if per_image:
        vals = [_lovasz_hinge_flat(lg.reshape(-1), lb.reshape(-1)) for lg, lb in zip(logits, labels)]
        return sum(vals) / len(vals)
    return _lovasz_hinge_flat(logits.reshape(-1), labels.reshape(-1))


def lovasz_loss(output, target):
    return lovasz_hinge(output, target.long())


def lovasz_hinge_grad_closed_form(logits, labels, dtype=torch.float64):
    """(loss, dloss/dlogits) for per_image=True, computed without autograd.

    d/dz_{pi(k)} = -s_{pi(k)} * elu'(e_{pi(k)}) * g_k / B ; ties are ordered by torch.sort(stable=True) (flat-index
    order, the rule the HIP kernel's stable radix sort follows).  ``dtype=torch.float32`` reproduces the reference's
    fp32 operation sequence for g_k (its first difference cancels ~3 digits), float64 is the well-conditioned truth."""
    B = logits.shape[0]
    z = logits.detach().to(dtype).reshape(B, -1)
    y = labels.detach().to(dtype).reshape(B, -1)
    s = 2 * y - 1
    e = 1 - z * s
    grad = torch.zeros_like(z)
    total = 0.0
    for b in range(B):
        es, perm = torch.sort(e[b], descending=True, stable=True)
        g = lovasz_grad(y[b][perm]).to(dtype) if dtype == torch.float32 else _lovasz_grad64(y[b][perm])
        total += float((F.elu(es) * g).double().sum())
        d = torch.where(es > 0, torch.ones_like(es), torch.exp(es))
        grad[b, perm] = -s[b][perm] * d * g / B
    return total / B, grad.reshape(logits.shape)


def _lovasz_grad64(gt_sorted):
    gt = gt_sorted.double()
    total = gt.sum()
    jac = 1.0 - (total - gt.cumsum(0)) / (total + (1.0 - gt).cumsum(0))
    if gt.numel() > 1:
        jac = torch.cat([jac[:1], jac[1:] - jac[:-1]])
    return jac


def dice_loss(output, target, smooth=0.0, eps=1e-7):
    return 1 - (2 * torch.sum(output * target) + smooth) / (torch.sum(output) + torch.sum(target) + smooth + eps)


def multiclass_dice_loss(output, target, smooth=0.0, activation='sigmoid'):
    if activation == 'sigmoid':
        prob = torch.sigmoid(output)
    elif activation == 'softmax':
        prob = torch.softmax(output, dim=1)
    else:
        raise NotImplementedError(activation)
    target = target.float()
    n = prob.shape[1]
    return sum(dice_loss(prob[:, c], target[:, c], smooth) for c in range(n)) / n


def mixed_dice_bce_loss(output, target, dice_weight=0.2, bce_weight=0.9, smooth=0.0, dice_activation='sigmoid'):
    n = output.shape[1]
    t = target[:, :n].float()
    return (dice_weight * multiclass_dice_loss(output, t, smooth, dice_activation)
            + bce_weight * F.binary_cross_entropy_with_logits(output, t))


LOSSES = {'lovasz': lovasz_loss, 'bce_dice': mixed_dice_bce_loss}


def adam_l2_step(params, grads, m, v, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-4):
    """One torch.optim.Adam step with L2-in-gradient weight decay (not AdamW), in place.
    ``step`` is the 1-based step count.  Tensors with grad None are skipped."""
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    for p, g, mm, vv in zip(params, grads, m, v):
        if g is None:
            continue
        g = g + weight_decay * p
        mm.mul_(beta1).add_(g, alpha=1 - beta1)
        vv.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (vv.sqrt() / (bc2 ** 0.5)).add_(eps)
        p.addcdiv_(mm, denom, value=-lr / bc1)
