"""ctypes binding of libsaltnet_hip.so, generated from include/saltnet.h (single source of truth).

The header is parsed at import: every ``typedef struct {...} salt_X;`` becomes a ``ctypes.Structure``
and every ``int salt_X(const salt_X_args*, void* stream)`` prototype gets argtypes/restype.  There is
NO fallback: if the shared library is missing or fails to load the import raises, so a GPU test
can never silently run on something else.
"""
import ctypes
import os
import re

import torch  # noqa: F401  FIRST: libsaltnet_hip.so must bind to the HIP runtime torch ships (its libamdhip64), not load a second
#                      copy from /opt/rocm - two runtimes in one process leave the later one without a device (hipErrorNoDevice)

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
HEADER = os.path.join(_ROOT, 'include', 'saltnet.h')
LIB_PATH = os.environ.get('SALT_LIB') or os.path.join(_PKG, 'libsaltnet_hip.so')      # SALT_LIB: A/B a build variant (tools/build_variant.sh)

_SCALARS = {
    'int': ctypes.c_int, 'float': ctypes.c_float, 'int64_t': ctypes.c_int64, 'uint32_t': ctypes.c_uint32,
    'double': ctypes.c_double,
}


class SaltError(RuntimeError):
    pass


def _strip_comments(src):
    return re.sub(r'/\*.*?\*/', '', src, flags=re.S)


def _parse(src):
    src = _strip_comments(src)
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r'#define\s+(SALT_\w+)\s+(-?\d+)', src)}
    structs = {}
    order = []
    for m in re.finditer(r'typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;', src, flags=re.S):
        body, name = m.group(1), m.group(2)
        fields = []
        for line in body.split(';'):
            line = ' '.join(line.split())
            if not line:
                continue
            fm = re.match(r'^(const\s+)?(\w+)\s*(\*+)?\s*(\w+)\s*(\[(\w+)\])?$', line)
            if not fm:
                raise SaltError('saltnet.h: cannot parse field %r of %s' % (line, name))
            _, typ, ptr, fname, _, arr = fm.groups()
            fields.append((fname, typ, bool(ptr), arr))
        structs[name] = fields
        order.append(name)
    protos = []
    for m in re.finditer(r'^\s*(int|int64_t)\s+(salt_\w+)\s*\(([^)]*)\)\s*;', src, flags=re.M):
        protos.append((m.group(2), m.group(1), m.group(3)))
    return consts, structs, order, protos


with open(HEADER) as _f:
    CONSTS, _STRUCT_FIELDS, _ORDER, _PROTOS = _parse(_f.read())

STRUCTS = {}
for _name in _ORDER:
    _fields = []
    for fname, typ, is_ptr, arr in _STRUCT_FIELDS[_name]:
        if is_ptr:
            ct = ctypes.c_void_p
        elif typ in _SCALARS:
            ct = _SCALARS[typ]
        elif typ in STRUCTS:
            ct = STRUCTS[typ]
        elif typ == 'salt_op_fn':
            ct = ctypes.c_void_p
        else:
            raise SaltError('saltnet.h: unknown type %s in %s' % (typ, _name))
        if arr:
            n = CONSTS[arr] if arr in CONSTS else int(arr)
            ct = ct * n
        _fields.append((fname, ct))
    STRUCTS[_name] = type(_name, (ctypes.Structure,), {'_fields_': _fields})

if not os.path.exists(LIB_PATH):
    raise SaltError('libsaltnet_hip.so is missing (%s): run __graft_entry__.build() or csrc/build.py; '
                    'there is no CPU or PyTorch fallback for the HIP path' % LIB_PATH)
try:
    lib = ctypes.CDLL(LIB_PATH)
except OSError as e:  # pragma: no cover
    raise SaltError('cannot load %s: %s' % (LIB_PATH, e))

lib.salt_last_error.restype = ctypes.c_char_p
lib.salt_last_error.argtypes = []
OP_FUNCS = {}
for _fname, _ret, _args in _PROTOS:
    try:
        fn = getattr(lib, _fname)
    except AttributeError:
        raise SaltError('libsaltnet_hip.so does not export %s declared in saltnet.h' % _fname)
    fn.restype = ctypes.c_int64 if _ret == 'int64_t' else ctypes.c_int
    am = re.match(r'^\s*const\s+(salt_\w+)\s*\*\s*(,\s*void\s*\*\s*stream)?\s*$', _args)
    if am and am.group(1) in STRUCTS:
        fn.argtypes = [ctypes.POINTER(STRUCTS[am.group(1)])] + ([ctypes.c_void_p] if am.group(2) else [])
        if am.group(2):
            OP_FUNCS[_fname] = (fn, STRUCTS[am.group(1)])
lib.salt_packed_weight_elems.argtypes = [ctypes.c_int] * 4
lib.salt_bn_stats_floats.argtypes = [ctypes.c_int] * 2
lib.salt_device_info.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, ctypes.c_int]
lib.salt_program_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
lib.salt_program_run_range.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lib.salt_program_run_timed.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
lib.salt_program_run_streams.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
lib.salt_graph_capture.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
lib.salt_graph_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.salt_graph_begin.argtypes = [ctypes.c_void_p]
lib.salt_graph_end.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
lib.salt_graph_destroy.argtypes = [ctypes.c_void_p]
lib.salt_program_run_streams_ex.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
lib.salt_program_run_streams_marks.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                               ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]
lib.salt_event_create.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
lib.salt_event_destroy.argtypes = [ctypes.c_void_p]
lib.salt_set_aux_stream.argtypes = [ctypes.c_void_p]
lib.salt_stream_wait_event.argtypes = [ctypes.c_void_p, ctypes.c_void_p]

DECLARED_SYMBOLS = [p[0] for p in _PROTOS] + ['salt_last_error']

# struct layout self-check against the compiled library
lib.salt_abi_struct_sizes.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int]
_sz = (ctypes.c_int * len(_ORDER))()
_n = lib.salt_abi_struct_sizes(_sz, len(_ORDER))
if _n != len(_ORDER):
    raise SaltError('saltnet.h declares %d structs, the library was built with %d: rebuild' % (len(_ORDER), _n))
for _i, _name in enumerate(_ORDER):
    if ctypes.sizeof(STRUCTS[_name]) != _sz[_i]:
        raise SaltError('ABI mismatch for %s: ctypes %d bytes, library %d bytes (stale build?)'
                        % (_name, ctypes.sizeof(STRUCTS[_name]), _sz[_i]))


def check(rc, what=''):
    if rc != 0:
        raise SaltError('%s failed (%d): %s' % (what or 'libsaltnet_hip call', rc, lib.salt_last_error().decode(errors='replace')))


def fill(struct, **kw):
    """Set fields of a ctypes struct; lists go to array fields (rest zero), None pointers -> NULL."""
    for k, v in kw.items():
        cur = getattr(struct, k)
        if isinstance(cur, ctypes.Array):
            if len(v) > len(cur):
                raise SaltError('%s: %d values for array of %d' % (k, len(v), len(cur)))
            for i, x in enumerate(v):
                cur[i] = x
        else:
            setattr(struct, k, v)
    return struct
