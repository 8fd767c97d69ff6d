"""salt_pack_batched (all weight packs of a network in one launch) against salt_pack_conv_weight job by job, bit for bit: the vector
path of the forward packs (3x3 / 1x1 layers with whole 32-channel chunks), the scalar path (ragged channel counts, other kernel
sizes) and the transposed data-gradient packs in ONE table."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_pack_batched_equals_single_packs_bitwise():
    import salt_amd  # noqa: F401
    from salt_amd._abi import STRUCTS, lib, fill, check
    g = torch.Generator().manual_seed(1)
    st = torch.cuda.current_stream().cuda_stream
    # D0 (out), D1 (in), KH, KW, transpose
    shapes = [(96, 64, 3, 3, 0), (64, 128, 1, 1, 0), (64, 16, 3, 3, 0), (40, 96, 3, 3, 0), (128, 320, 3, 3, 0), (64, 32, 3, 3, 1), (24, 64, 2, 2, 0),
              (512, 768, 3, 3, 0), (256, 64, 1, 1, 1), (64, 96, 3, 3, 1), (320, 64, 3, 3, 1), (48, 64, 3, 3, 1), (512, 512, 3, 3, 1)]
    jobs, singles, outs = [], [], []
    keep = []
    for (D0, D1, KH, KW, tr) in shapes:
        w = torch.randn(D0, D1, KH, KW, generator=g).cuda()
        nt = KH * KW
        N, C = (D1, D0) if tr else (D0, D1)
        n = lib.salt_packed_weight_elems(1, nt, N, C)
        a, b = torch.full((n,), 7.0, dtype=torch.bfloat16, device='cuda:0'), torch.full((n,), 9.0, dtype=torch.bfloat16, device='cuda:0')
        taps = [(t // KW, t % KW) for t in range(nt)]
        mk = lambda dst: fill(STRUCTS['salt_pack_conv_weight_args'](), dtype=1, w=w.data_ptr(), D0=D0, D1=D1, KH=KH, KW=KW, ntaps=nt,
                              tap_kh=[t[0] for t in taps], tap_kw=[t[1] for t in taps], transpose=tr, wp=dst.data_ptr())
        jobs.append(mk(a)); singles.append(mk(b)); outs.append((a, b)); keep.append(w)
    blocks = [lib.salt_pack_job_blocks(ctypes.byref(s)) for s in jobs]
    assert blocks[0] == (96 * 2 * 4 + 255) // 256 and blocks[2] == (64 * 32 + 255) // 256      # vector path / scalar path (16 -> 32 padded channels)
    pref = np.concatenate([[0], np.cumsum(blocks)]).astype(np.int32)
    table = torch.frombuffer(bytearray(b''.join(bytes(s) for s in jobs)), dtype=torch.uint8).cuda()
    pref_t = torch.from_numpy(pref).cuda()
    check(lib.salt_pack_batched(ctypes.byref(fill(STRUCTS['salt_pack_batched_args'](), jobs=table.data_ptr(), job_block0=pref_t.data_ptr(), njobs=len(jobs),
                                                  total_blocks=int(pref[-1]), dtype=1)), st))
    for s in singles:
        check(lib.salt_pack_conv_weight(ctypes.byref(s), st))
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(outs):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), shapes[i]


def test_adam_pack_equals_adam_then_pack_bitwise():
    """salt_adam_pack (round 5: Adam + L2 and the bf16 forward packs in one pass) against salt_adam followed by salt_pack_conv_weight on
    the same buffers: parameters, both moments and every packed copy bit for bit; the ranges no job covers take the plain update."""
    import salt_amd  # noqa: F401
    from salt_amd._abi import STRUCTS, lib, fill, check
    g = torch.Generator().manual_seed(5)
    st = torch.cuda.current_stream().cuda_stream
    # flat layout: [bias 64 | W1 96x64x3x3 | bn 128 | W2 40x128x3x3 (a partial last block of segments) | tail 36]
    shapes = [None, (96, 64, 3, 3), None, (40, 128, 3, 3), None]
    sizes = [64, 96 * 64 * 9, 128, 40 * 128 * 9, 36]
    n = sum(sizes)
    P = torch.randn(n, generator=g).cuda(); G = torch.randn(n, generator=g).cuda()
    M = (0.1 * torch.randn(n, generator=g)).cuda(); V = (0.01 * torch.rand(n, generator=g)).cuda()
    hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 1e-4, 1 - 0.9 ** 3, 1 - 0.999 ** 3, 0.5], dtype=torch.float32).cuda()
    ref = [t.clone() for t in (P, M, V)]
    a = fill(STRUCTS['salt_adam_args'](), param=ref[0].data_ptr(), grad=G.data_ptr(), exp_avg=ref[1].data_ptr(), exp_avg_sq=ref[2].data_ptr(), n=n, hyper=hyper.data_ptr())
    check(lib.salt_adam(ctypes.byref(a), st))
    jobs, packs, rest, off = [], [], [], 0
    for shp, sz in zip(shapes, sizes):
        if shp is None:
            rest.append((off, sz))
        else:
            D0, D1, KH, KW = shp
            nt = KH * KW
            wp = torch.full((lib.salt_packed_weight_elems(1, nt, D0, D1),), 3.0, dtype=torch.bfloat16, device='cuda:0')
            jobs.append(fill(STRUCTS['salt_pack_conv_weight_args'](), dtype=1, w=P.data_ptr() + 4 * off, D0=D0, D1=D1, KH=KH, KW=KW, ntaps=nt,
                             tap_kh=[t // KW for t in range(nt)], tap_kw=[t % KW for t in range(nt)], transpose=0, wp=wp.data_ptr()))
            assert lib.salt_pack_job_is_vec(ctypes.byref(jobs[-1])) == 1
            packs.append((wp, off, shp))
        off += sz
    blocks = [lib.salt_pack_job_blocks(ctypes.byref(s)) for s in jobs]
    pref = torch.from_numpy(np.concatenate([[0], np.cumsum(blocks)]).astype(np.int32)).cuda()
    table = torch.frombuffer(bytearray(b''.join(bytes(s) for s in jobs)), dtype=torch.uint8).cuda()
    rb = [(c + 1023) // 1024 for _, c in rest]
    rpref = torch.from_numpy(np.concatenate([[0], np.cumsum(rb)]).astype(np.int32)).cuda()
    rt = torch.tensor(rest, dtype=torch.int64).cuda()
    ap = fill(STRUCTS['salt_adam_pack_args'](), param=P.data_ptr(), grad=G.data_ptr(), exp_avg=M.data_ptr(), exp_avg_sq=V.data_ptr(), n=n, hyper=hyper.data_ptr(),
              jobs=table.data_ptr(), job_block0=pref.data_ptr(), njobs=len(jobs), pack_blocks=int(sum(blocks)), rest=rt.data_ptr(), rest_block0=rpref.data_ptr(),
              nrest=len(rest), rest_blocks=int(sum(rb)))
    check(lib.salt_adam_pack(ctypes.byref(ap), st))
    torch.cuda.synchronize()
    for got, want, what in zip((P, M, V), ref, ('param', 'exp_avg', 'exp_avg_sq')):
        assert torch.equal(got, want), what
    for wp, off, (D0, D1, KH, KW) in packs:
        nt = KH * KW
        want = torch.zeros_like(wp)
        s = fill(STRUCTS['salt_pack_conv_weight_args'](), dtype=1, w=ref[0].data_ptr() + 4 * off, D0=D0, D1=D1, KH=KH, KW=KW, ntaps=nt,
                 tap_kh=[t // KW for t in range(nt)], tap_kw=[t % KW for t in range(nt)], transpose=0, wp=want.data_ptr())
        check(lib.salt_pack_conv_weight(ctypes.byref(s), st))
        torch.cuda.synchronize()
        assert torch.equal(wp.view(torch.int16), want.view(torch.int16)), (D0, D1, KH, KW)


def test_fused_step_leaves_every_forward_pack_current():
    """The shipped step (SegmentationModel._fit_loop: FusedAdam -> salt_adam_pack, then Engine.refresh runs only the packs Adam did not
    write): after two steps every packed copy equals a full re-pack of the current masters, bit for bit."""
    from salt_amd.models import SegmentationModel
    arch = {'model_params': {'architecture': 'UNetResNet', 'out_channels': 2, 'activation': 'sigmoid', 'compute_dtype': 'bf16'},
            'optimizer_params': {'lr': 1e-3}, 'regularizer_params': {'regularize': True, 'weight_decay_conv2d': 1e-4}}
    torch.manual_seed(0)
    m = SegmentationModel(arch, {'epochs': 1}, {}); m._to_device(); m.model.train()
    X = torch.randn(2, 3, 64, 64, device='cuda'); T = (torch.rand(2, 1, 64, 64, device='cuda') < 0.3).float(); T = torch.cat([1 - T, T], 1)
    for _ in range(2):
        m._fit_loop([X, T])
    eng = m.model.engine()
    assert m.optimizer._adam_pack_program() is not None and len(eng._adam_jobs) > 30
    eng.refresh(True)                                    # what the next step would do: the rest of the forward packs + the backward packs
    torch.cuda.synchronize()
    have = {k: t.clone() for k, t in eng._packed.items()}
    eng._pack_batched.run(); eng._pack_batched_bwd.run()   # every job from the fp32 masters
    torch.cuda.synchronize()
    for k, t in eng._packed.items():
        assert torch.equal(have[k].view(torch.int16), t.view(torch.int16)), k[1:]
