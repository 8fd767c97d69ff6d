#!/usr/bin/env python
"""Index-math model of conv_wgrad_ls_kernel (csrc/conv_wgrad_ls.hip): runs on the CPU, no GPU needed.

Checks, for both geometries (NB = 1: 16-pixel strips, NB = 2: two 8-pixel images side by side) and both unit sizes:
  * the loader's (piece, lane) -> (LDS row, 16-byte slot) -> (source pixel, channel slot) map fills every byte the MFMA waves read;
  * every ds_read_b64_tr_b16 address a lane forms (P operand, Q operand, own and previous ring slot, every tap) lands on the bytes
    that hold exactly the (pixel, channels) the MFMA operand layout expects;
  * the 32-lane groups of every read touch each of the 64 LDS banks at most once (conflict-free with 128-byte rows + the XOR swizzle).
usage: python tools/wgrad_ls_model.py"""
import itertools


def swz_bit(r):
    return (r >> 1) & 1


def check(NB, KU):
    HP = NB * (16 // NB + 2)
    KHS = 8 if NB == 1 else 10
    P_BYTES = KU * 16 * 128
    QROWS = KU * HP
    assert QROWS % 8 == 0
    # ---- loader: LDS content of one slot as {(region, row, slot16): (kind, row-coordinates, channel slot)}
    lds = {}
    for pi in range(P_BYTES // 1024):
        for l in range(64):
            r = pi * 8 + (l >> 3)
            cs = (l & 7) ^ (swz_bit(r) << 2)
            j, px = r >> 4, r & 15
            lds[('P', r, l & 7)] = (j, px, cs)
    for qi in range(QROWS // 8):
        for l in range(64):
            r = qi * 8 + (l >> 3)
            cs = (l & 7) ^ (swz_bit(r) << 2)
            hrl, hx = divmod(r, HP)
            lds[('Q', r, l & 7)] = (hrl, hx, cs)
    # ---- MFMA waves
    for wa, wb in itertools.product(range(2), range(2)):
        for j in range(KU):
            for h in range(2):
                banks = {0: [], 1: []}
                for lane in range(64):
                    khalf, g16, i16 = lane >> 5, (lane >> 4) & 1, lane & 15
                    prow, pcol = i16 >> 2, (i16 & 3) * 4
                    cs = wa * 4 + g16 * 2 + (pcol >> 3)
                    s = (prow >> 1) & 1
                    paL = (khalf * 8 + prow) * 128 + ((cs ^ (s << 2)) << 4) + (pcol & 7) * 2
                    addr = paL + (j * 16 + h * 4) * 128
                    row, slot, byte = addr // 128, (addr % 128) // 16, addr % 16
                    jj, px, csl = lds[('P', row, slot)]
                    assert (jj, px) == (j, khalf * 8 + h * 4 + prow), (NB, KU, 'P pixel')
                    assert csl * 8 + byte // 2 == wa * 32 + g16 * 16 + pcol, (NB, KU, 'P channel')
                    banks[khalf] += [(addr // 4) % 64, (addr // 4 + 1) % 64]
                for k in banks:
                    assert len(set(banks[k])) == 64, (NB, KU, 'P bank conflict', sorted(banks[k]))
            for dy in range(3):
                hr = j + dy - 2
                hrl = hr + KU if hr < 0 else hr          # previous slot: rows KU - 2, KU - 1
                par = hrl & 1
                for u in range(3):
                    banks = {0: [], 1: []}
                    for lane in range(64):
                        khalf, g16, i16 = lane >> 5, (lane >> 4) & 1, lane & 15
                        prow, pcol = i16 >> 2, (i16 & 3) * 4
                        cs = wb * 4 + g16 * 2 + (pcol >> 3)
                        if NB == 1:
                            s = ((prow >> 1) & 1) ^ par
                        else:
                            s = (khalf ^ (prow >> 1)) & 1
                        qL = (khalf * KHS + prow) * 128 + ((cs ^ (s << 2)) << 4) + (pcol & 7) * 2
                        addr = qL + (hrl * HP + u * 4) * 128
                        row, slot, byte = addr // 128, (addr % 128) // 16, addr % 16
                        if u < 2 or prow < 2:          # u = 2 feeds only pixels +8, +9 of the run (prow 0, 1); prow 2, 3 read past the halo row: unused
                            hh, hx, csl = lds[('Q', row, slot)]
                            assert (hh, hx) == (hrl, khalf * KHS + prow + u * 4), (NB, KU, 'Q pixel', hh, hx)
                            assert csl * 8 + byte // 2 == wb * 32 + g16 * 16 + pcol, (NB, KU, 'Q channel')
                        banks[khalf] += [(addr // 4) % 64, (addr // 4 + 1) % 64]
                    for k in banks:
                        assert len(set(banks[k])) == 64, (NB, KU, 'Q bank conflict')
    # the pre-unit fills rows (KU - 2) HP .. KU HP - 1: whole pieces from this one on (earlier rows of that piece are don't-care)
    first = ((KU - 2) * HP) // 8
    return {'HP': HP, 'pieces': P_BYTES // 1024 + QROWS // 8, 'pre_first_q_piece': first, 'pre_pieces': QROWS // 8 - first}


if __name__ == '__main__':
    for NB in (1, 2):
        for KU in (4, 8):
            print('NB', NB, 'KU', KU, check(NB, KU))
    print('ok')


def check_stride2(NB, KU):
    """Stride-2 geometry (3x3, pad 1): a Q image row is stored as TWO 18-pixel halo rows, its even-column plane E (hx = 2 k) and its odd
    one O (hx = 2 k + 1), so the fragment reads stay unit-stride: tap dx = 0 -> E[k], dx = 1 -> O[k], dx = 2 -> E[k + 1].  A unit's
    2 KU image rows are halo rows 0 .. 4 KU - 1; NB = 2: two 8-pixel images side by side, 9 plane pixels each (KHS = 9)."""
    HP = 18
    KHS = 8 if NB == 1 else 9
    P_BYTES = KU * 16 * 128
    QROWS = 4 * KU * HP
    assert QROWS % 8 == 0
    lds = {}
    for qi in range(QROWS // 8):
        for l in range(64):
            r = qi * 8 + (l >> 3)
            cs = (l & 7) ^ (swz_bit(r) << 2)
            hrl, hx = divmod(r, HP)
            lds[('Q', r, l & 7)] = (hrl >> 1, hrl & 1, hx, cs)          # image row in the unit, plane, plane pixel, channel slot
    for wb in range(2):
        for hr2 in range(2 * KU):                                       # image row of the unit
            for plane in range(2):
                hrl = 2 * hr2 + plane
                par = hrl & 1
                for kind, off in (('a0', 0), ('a1', 4), ('c1', 6)):
                    if plane == 1 and kind == 'c1':
                        continue
                    banks = {0: [], 1: []}
                    for lane in range(64):
                        khalf, g16, i16 = lane >> 5, (lane >> 4) & 1, lane & 15
                        prow, pcol = i16 >> 2, (i16 & 3) * 4
                        cs = wb * 4 + g16 * 2 + (pcol >> 3)
                        base_px = khalf * KHS + prow + (2 if kind == 'c1' else 0)
                        sL = ((khalf * KHS + prow) >> 1) & 1
                        s = sL ^ par ^ (1 if kind == 'c1' else 0)
                        roff = 4 if kind == 'c1' else off               # c1 = the read at pixel 2 + 4
                        qL = base_px * 128 + ((cs ^ (s << 2)) << 4) + (pcol & 7) * 2
                        addr = qL + (hrl * HP + roff) * 128
                        row, slot, byte = addr // 128, (addr % 128) // 16, addr % 16
                        px = khalf * KHS + prow + off
                        # pixels the MFMA operands use: a0 / a1 -> k = 0 .. 7 (all four rows); c1 -> k = 8 only (prow 2 of the read at +6)
                        used = kind != 'c1' or prow == 2
                        if used:
                            ir, pl, hx, csl = lds[('Q', row, slot)]
                            assert (ir, pl, hx) == (hr2, plane, px), (NB, KU, kind, ir, pl, hx, hr2, plane, px)
                            assert csl * 8 + byte // 2 == wb * 32 + g16 * 16 + pcol
                        banks[khalf] += [(addr // 4) % 64, (addr // 4 + 1) % 64]
                    for k in banks:
                        assert len(set(banks[k])) == 64, (NB, KU, 'stride-2 bank conflict', kind)
    first = ((4 * KU - 2) * HP) // 8
    return {'pieces': P_BYTES // 1024 + QROWS // 8, 'pre_first_q_piece': first, 'pre_pieces': QROWS // 8 - first}


if __name__ == '__main__':
    for NB in (1, 2):
        for KU in (2, 4):
            print('stride 2, NB', NB, 'KU', KU, check_stride2(NB, KU))
    print('ok')
