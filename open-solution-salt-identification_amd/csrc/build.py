"""Build libsaltnet_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python open-solution-salt-identification_amd/csrc/build.py [--force]

Objects go to csrc/_obj/, the shared library next to the package's __init__.py so that it travels
to the GPU box with the source tree (built artefacts are git-ignored, not gpurun-ignored)."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
INC = os.path.join(ROOT, 'include')
OBJ = os.path.join(HERE, '_obj')
LIB = os.path.join(PKG, 'libsaltnet_hip.so')
SOURCES = ['runtime.hip', 'conv_mfma.hip', 'conv_ws.hip', 'conv_thin.hip', 'conv_wgrad_ls.hip', 'conv_small.hip', 'head_fused.hip', 'elementwise.hip', 'hyper.hip', 'se.hip', 'loss.hip', 'input.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + INC, '-I' + HERE, '-Wno-unused-value']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _digest(path):
    h = hashlib.sha1()
    for p in [path, os.path.join(HERE, 'common.h'), os.path.join(INC, 'saltnet.h')]:
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def _compile(src):
    path = os.path.join(HERE, src)
    obj = os.path.join(OBJ, src.replace('.hip', '.o'))
    stamp = obj + '.sha1'
    dg = _digest(path)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dg:
        return obj, False
    cmd = [_hipcc()] + FLAGS + ['-c', path, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr[-4000:]))
    with open(stamp, 'w') as f:
        f.write(dg)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        results = list(ex.map(_compile, SOURCES))
    objs = [o for o, _ in results]
    if any(ch for _, ch in results) or not os.path.exists(LIB):
        cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr[-4000:])
        if verbose:
            print('built', LIB)
    elif verbose:
        print('up to date', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
