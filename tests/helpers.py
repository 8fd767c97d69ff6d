"""Shared helpers for the test-suite (golden loading, closed-form state, comparisons)."""
import os
from collections import OrderedDict

import numpy as np
import torch

import closed_form as CF

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden(name):
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def state_from_fixture(fx, canonical=None):
    """Closed-form state BEFORE the pass for every 's:<key>' entry of a block fixture."""
    sd = OrderedDict()
    for k, v in fx.items():
        if k.startswith('s:'):
            key = k[2:]
            ck = canonical(key) if canonical else key
            sd[key] = CF.tensor_for(ck, v.shape)
    return sd


def record_parity(name, **values):
    """SALT_PARITY_COUNTS=<file.json>: the parity tests of the BASELINE configs append their decision counts (mask pixels that differ from
    the oracle, totals, agreement) to that file - tools/r03_evidence.sh commits it as profiles/rNN_parity_counts.json."""
    path = os.environ.get('SALT_PARITY_COUNTS')
    if not path:
        return
    import json
    doc = {}
    if os.path.exists(path):
        with open(path) as f:
            doc = json.load(f)
    doc[name] = {k: (float(v) if isinstance(v, float) else int(v) if isinstance(v, (int, np.integer)) else v) for k, v in values.items()}
    with open(path, 'w') as f:
        json.dump(doc, f, indent=1, sort_keys=True)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def assert_close(a, b, tol, what=''):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().float().numpy()
    if isinstance(b, torch.Tensor):
        b = b.detach().cpu().float().numpy()
    assert a.shape == b.shape, '%s: shape %s vs %s' % (what, a.shape, b.shape)
    e = rel_err(a, b)
    assert e <= tol, '%s: max-rel err %.3e > %.1e' % (what, e, tol)
    return e
