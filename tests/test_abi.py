"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/bfhip.h declares (no compute calls without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from bayesfast_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def library():
    _lib.build()
    return _lib.lib()


def test_header_symbols_exported(library):
    hdr = open(os.path.join(ROOT, 'include', 'bfhip.h')).read()
    declared = set(re.findall(r'\b(bfhip_[a-z_0-9]+)\s*\(', hdr))
    assert declared, 'no declarations found'
    raw = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), 'libbfhip.so does not export ' + name
    assert declared == set(_lib.SYMBOLS), 'python binding table and header disagree'


def test_debug_header_is_the_whole_tuning_surface(library):
    """include/bfhip_debug.h: every bfhip_debug_* export is declared there and nowhere else, the integer switches go through
    ONE setter by key (unknown keys are refused), and nothing of it is in the drop-in header."""
    import subprocess
    dbg = open(os.path.join(ROOT, 'include', 'bfhip_debug.h')).read()
    declared = set(re.findall(r'\b(bfhip_debug_[a-z_0-9]+)\s*\(', dbg))
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r'\b(bfhip_debug_[a-z_0-9]+)\b', out))
    assert exported == declared == {'bfhip_debug_set', 'bfhip_debug_get', 'bfhip_debug_buffer', 'bfhip_debug_last_kernel', 'bfhip_debug_tail_count'}
    assert 'bfhip_debug_set' not in re.findall(r'\b(bfhip_[a-z_0-9]+)\s*\(', open(os.path.join(ROOT, 'include', 'bfhip.h')).read())
    for key in re.findall(r'"([a-z_0-9]+)"', dbg.split('Integer switches by name')[1].split('*/')[0]):
        before = _lib.debug_get(key)
        _lib.debug_set(key, before + 1)
        assert _lib.debug_get(key) == before + 1
        _lib.debug_set(key, before)
    with pytest.raises(ValueError):
        _lib.debug_set('no_such_switch', 1)


def test_enums_match_header():
    hdr = open(os.path.join(ROOT, 'include', 'bfhip.h')).read()
    sc = re.search(r'BFHIP_SC_LOG_STEP = 0,(.*?)BFHIP_SC_N\n', hdr, re.S).group(1)
    assert len(re.findall(r'BFHIP_SC_[A-Z_]+', sc)) + 1 == _lib.SC_N
    vec = re.search(r'BFHIP_VEC_Q = 0,(.*?)BFHIP_VEC_N\n', hdr, re.S).group(1)
    assert len(re.findall(r'BFHIP_VEC_[A-Z_]+', vec)) + 1 == _lib.VEC_N
    assert int(re.search(r'#define BFHIP_STAT_STRIDE (\d+)', hdr).group(1)) == _lib.STAT_STRIDE == len(_lib.NSTATS)
    assert int(re.search(r'#define BFHIP_MAX_DIM (\d+)', hdr).group(1)) == _lib.MAX_DIM


def test_version_and_argument_errors(library):
    assert library.bfhip_version() >= 102   # (101: BFHIP_TREE_MODE_WORK 4162; 102: bfhip_polar_ns work size)
    # NULL context is rejected with ValueError semantics before anything touches a GPU
    rc = library.bfhip_logp_grad(None, 1, None, 0, None, None)
    assert rc == -1
    with pytest.raises(ValueError):
        _lib.check(rc)
    assert b'invalid argument' in library.bfhip_last_error()


def test_desc_flattening_masks():
    """PolyConfig masks are scattered to the full input (modules/poly.py:474-477)."""
    from bayesfast_amd.device import density_desc_from_spec
    d = 5
    q = np.arange(9.).reshape(1, 3, 3)
    q[0][np.tril_indices(3, -1)] = np.nan  # the reference leaves the lower triangle uninitialised
    spec = dict(d=d, poly=dict(input_size=d, output_size=1, use_bound=False, configs=[
        dict(order='linear', input_mask=[0, 1, 2, 3, 4], output_mask=[0], coef=np.arange(6.)[None]),
        dict(order='quadratic', input_mask=[0, 2, 4], output_mask=[0], coef=q)]))
    ds, keep = density_desc_from_spec(spec)
    assert ds.c0 == 0. and ds.use_bound == 0 and not ds.cubic2
    lin = np.ctypeslib.as_array(ds.lin, (d,))
    quad = np.ctypeslib.as_array(ds.quad, (d, d))
    assert np.array_equal(lin, [1, 2, 3, 4, 5])
    assert np.isfinite(quad).all()
    assert quad[0, 2] == 1. and quad[2, 4] == 5. and quad[4, 4] == 8. and quad[2, 0] == 0.
    with pytest.raises(ValueError):
        density_desc_from_spec(dict(d=4, poly=dict(input_size=4, output_size=2, configs=[])))
