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
