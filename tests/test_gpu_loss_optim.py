"""GPU parity: Lovasz hinge / BCE+Dice kernels and the fused Adam step vs reference goldens and the oracle."""
import numpy as np
import pytest
import torch

from helpers import golden, T, assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('case', ['random', 'all0', 'all1', 'p1', 'ties', 'big'])
def test_lovasz_vs_reference_golden(case):
    from salt_amd import losses
    fx = golden('F6_lovasz')
    z = T(fx[case + '_z']).to(DEV).requires_grad_(True)
    t = T(fx[case + '_t']).to(DEV)
    loss = losses.lovasz_loss(z, t)
    loss.backward()
    ref = float(fx[case + '_loss'])
    assert abs(float(loss) - ref) <= 1e-5 * max(1.0, abs(ref)), (float(loss), ref)
    g = z.grad.cpu().numpy()
    if case != 'ties':
        assert_close(g, fx[case + '_gz'], 1e-4, 'dlogits')
    else:
        # Inside a group of tied errors the per-element gradient depends on the order the sort happens to produce
        # (only the loss value is order-invariant).  The HIP sort is stable (flat-index order), so compare with
        # the oracle's closed form evaluated with a stable sort.
        from oracle import losses as OL
        lv, gref = OL.lovasz_hinge_grad_closed_form(T(fx[case + '_z']), T(fx[case + '_t']))
        assert abs(lv - float(loss)) < 1e-5
        assert_close(g, gref.float().numpy(), 2e-4, 'dlogits (stable tie order)')


def test_lovasz_full_size_vs_oracle_and_properties():
    """BASELINE C3 size (B=64, P=2*128*128): loss vs the oracle's closed form, sum(g_k)=1 => sum|grad| bound, determinism."""
    from salt_amd import losses
    from oracle import losses as OL
    g = torch.Generator().manual_seed(3)
    B, H, W = 64, 128, 128
    z = torch.randn(B, 2, H, W, generator=g) * 2
    m = (torch.rand(B, 1, H, W, generator=g) < 0.3).float()
    m[:8] = 0                                        # empty masks (dataset trait)
    t = torch.cat([1 - m, m], 1)
    l1, g1 = losses.native_loss(z.to(DEV), t.to(DEV), 'lovasz')
    l2, g2 = losses.native_loss(z.to(DEV), t.to(DEV), 'lovasz')
    assert float(l1) == float(l2) and torch.equal(g1, g2)          # deterministic
    l4, g4 = losses.native_loss(z[:4].to(DEV), t[:4].to(DEV), 'lovasz')
    zr = z[:4].clone().requires_grad_(True)
    lr_ = OL.lovasz_loss(zr, t[:4])                      # the reference's arithmetic through autograd
    lr_.backward()
    assert abs(float(l4) - float(lr_)) < 2e-5 * max(1, abs(float(lr_)))
    # With 32768 fp32 keys per image a few exact ties occur; per-element gradients inside a tie group depend on the
    # sort's tie order (torch.sort is unstable, the HIP radix sort is stable), so the element-wise check uses the
    # oracle's closed form with a STABLE sort and the reference's fp32 operation sequence ...
    ref32, gref32 = OL.lovasz_hinge_grad_closed_form(z[:4], t[:4], dtype=torch.float32)
    assert abs(float(l4) - ref32) < 2e-5 * max(1, abs(ref32))
    assert_close(g4.cpu(), gref32, 2e-5, 'grad vs fp32 closed form (stable ties)')
    # ... while vs torch's own (unstable) order only tie-group members may differ, and vs float64 the bar is the
    # fp32 cancellation of g_k = J_k - J_(k-1) (~1e-3, present in the reference itself)
    assert float((g4.cpu() - zr.grad).abs().max()) <= 4e-3 * float(zr.grad.abs().max())
    ref, gref = OL.lovasz_hinge_grad_closed_form(z[:4], t[:4])
    assert abs(float(l4) - ref) < 2e-5 * max(1, abs(ref))
    assert_close(g4.cpu(), gref.float(), 5e-3, 'grad vs float64 closed form')
    # |d loss/d z_i| = elu'(e_i) g_k / B <= g_k / B and sum_k g_k = 1 per image
    per_image = g1.abs().reshape(B, -1).sum(1).cpu()
    assert float(per_image.max()) <= 1.0 / B + 1e-6


@pytest.mark.parametrize('shape', [(3, 2, 64, 64), (2, 2, 70, 70), (5, 2, 128, 128), (2, 1, 101, 101), (1, 2, 256, 256)])
@pytest.mark.parametrize('kind', ['random', 'ties', 'all0', 'all1'])
def test_lovasz_split_sort_equals_single_workgroup_sort(shape, kind):
    """The split form (several workgroups per image, segments of 2048 positions, 1 + 4 + 1 launches) against the one-workgroup
    kernel through the C-ABI: same positions and tie order => bit-identical gradients; the loss differs by its summation tree only.
    Ragged last segments (9800, 10201 elements), heavy ties, empty and full masks."""
    import ctypes
    from salt_amd import _abi
    B, C, H, W = shape
    P = C * H * W
    g = torch.Generator().manual_seed(B * 1000 + H)
    z = torch.randn(B, C, H, W, generator=g) * 2
    if kind == 'ties':
        z = torch.round(z * 2) / 2                               # a handful of distinct values: long tie groups
    m = (torch.rand(B, 1, H, W, generator=g) < 0.3).float()
    if kind == 'all0':
        m.zero_()
    if kind == 'all1':
        m.fill_(1)
    t = torch.cat([1 - m, m], 1)[:, :C].contiguous()
    z, t = z.to(DEV).contiguous(), t.to(DEV)
    sw = int(_abi.lib.salt_lovasz_split_words(P))
    assert sw > 0

    def run(split):
        wk = torch.empty(2 * B * P, dtype=torch.int32, device=DEV)
        wv = torch.empty(2 * B * P, dtype=torch.int32, device=DEV)
        ws = torch.full((B * sw,), 0x55555555, dtype=torch.int32, device=DEV) if split else None      # garbage: the kernels initialise it
        lpi = torch.empty(B, device=DEV); loss = torch.zeros(1, device=DEV); dl = torch.full_like(z, float('nan'))
        a = _abi.STRUCTS['salt_lovasz_args']()
        _abi.fill(a, logits=z.data_ptr(), target=t.data_ptr(), B=B, P=P, ws_keys=wk.data_ptr(), ws_vals=wv.data_ptr(), loss_per_image=lpi.data_ptr(),
                  loss=loss.data_ptr(), dlogits=dl.data_ptr(), loss_scale=1.0, ws_split=ws.data_ptr() if split else None)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(2):                                       # twice: the workspace is left reusable
            _abi.check(_abi.lib.salt_lovasz_hinge(ctypes.byref(a), st), 'lovasz')
        torch.cuda.synchronize()
        return float(loss), lpi.cpu(), dl.cpu()

    l1, p1, g1 = run(False)
    l2, p2, g2 = run(True)
    assert torch.equal(g1, g2), float((g1 - g2).abs().max())
    assert abs(l1 - l2) <= 2e-6 * max(1.0, abs(l1)), (l1, l2)
    assert float((p1 - p2).abs().max()) <= 2e-6 * max(1.0, float(p1.abs().max()))


def test_bce_dice_vs_reference_golden():
    from salt_amd import losses
    fx = golden('F7_bce_dice')
    z = T(fx['z']).to(DEV).requires_grad_(True)
    loss = losses.mixed_dice_bce_loss(z, T(fx['t']).to(DEV))
    loss.backward()
    assert abs(float(loss) - float(fx['loss'])) < 1e-5
    assert_close(z.grad.cpu(), fx['gz'], 1e-4, 'dlogits')


def test_fused_adam_matches_torch_semantics():
    from salt_amd import architectures as A
    from salt_amd.optim import FusedAdam, weight_regularization
    from oracle import losses as OL
    torch.manual_seed(0)
    net = A.VanillaUNet(2, 1, 16, 2).to(DEV)
    eng = net.engine(torch.device(DEV))
    opt = FusedAdam(weight_regularization(net, True, 1e-4), lr=1e-3, model=net)
    ps = [p.detach().cpu().clone() for p in eng.live_params]
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    gen = torch.Generator().manual_seed(1)
    for step in range(1, 4):
        gs = [torch.randn(p.shape, generator=gen) * 0.1 for p in ps]
        for p, g in zip(eng.live_params, gs):
            off, n = eng.grad_range(p)
            eng.grads[off:off + n].copy_(g.reshape(-1))
        opt.step()
        OL.adam_l2_step(ps, gs, ms, vs, step, lr=1e-3, weight_decay=1e-4)
    for p, r in zip(eng.live_params, ps):
        assert_close(p.detach().cpu(), r, 1e-5, 'adam param')
    assert opt.state_dict()['param_groups'][0]['lr'] == 1e-3
    opt.param_groups[0]['lr'] = 5e-4                                   # scheduler writes lr (callbacks.py:273-275)
    opt.step()
    assert abs(float(opt.hyper[0]) - 5e-4) < 1e-9
