"""ctypes binding of libbfhip.so (include/bfhip.h).  The HIP extension is mandatory: there is no CPU
fallback, and every compute entry point fails loudly when the library is missing."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# BFHIP_LIBRARY overrides the in-tree library (tuning builds of tools/gvariant.sh)
LIB_PATH = os.environ.get('BFHIP_LIBRARY') or os.path.join(_HERE, 'libbfhip.so')
CSRC = os.path.join(_HERE, 'csrc')

SC_FIELDS = ('log_step', 'log_bar', 'hbar', 'mu', 'count', 'fg_n', 'bg_n', 'n_samples', 'prev_update',
             'adapt_window', 'i_iter', 'error')
SC_N = len(SC_FIELDS)
VEC_FIELDS = ('q', 'var', 'fg_mean', 'fg_raw', 'bg_mean', 'bg_raw')
VEC_N = len(VEC_FIELDS)
MAT_N = 6  # BFHIP_MAT_N: per-chain matrices of the full-rank metric (slot 0 = covariance)
STAT_STRIDE = 11
NSTATS = ('logp', 'energy', 'tree_depth', 'tree_size', 'mean_tree_accept', 'step_size', 'step_size_bar',
          'warmup', 'energy_change', 'max_energy_change', 'diverging')
HSTATS = ('logp', 'energy', 'n_int_step', 'accept_stat', 'accepted', 'step_size', 'step_size_bar', 'warmup',
          'energy_change', 'diverging')
MAX_DIM = 128
MAX_TREEDEPTH = 12
TREE_MODE_WORK = 4162   # BFHIP_TREE_MODE_WORK
# lower edges of the size classes of bfhip_tree_size_mode_share (bf_lag_edge): 1, 2, 3, 4, 6, 8, 12, 16, ...
LAG_EDGES = tuple((j + 1) if j < 2 else ((1 << ((j + 1) // 2)) if (j & 1) else (3 << (j // 2 - 1))) for j in range(64))

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)


class DensityDesc(C.Structure):
    _fields_ = [('d', C.c_int), ('ranges', _dp), ('hard_bounds', _u8p), ('su_lo', _dp), ('su_diff', _dp),
                ('c0', C.c_double), ('lin', _dp), ('quad', _dp), ('cubic2', _dp), ('cubic3', _dp),
                ('use_bound', C.c_int), ('mu', _dp), ('hess', _dp), ('alpha', C.c_double), ('f_mu', C.c_double),
                ('use_decay', C.c_int), ('decay_mu', _dp), ('decay_hess', _dp), ('decay_alpha2', C.c_double),
                ('decay_gamma', C.c_double),
                ('link_kind', C.c_int), ('link_y', C.c_double), ('link_prec', C.c_double), ('link_logp0', C.c_double)]


class SamplerConfig(C.Structure):
    _fields_ = [('sampler', C.c_int), ('n_warmup', C.c_int), ('max_treedepth', C.c_int), ('n_int_step', C.c_int),
                ('max_change', C.c_double), ('target_accept', C.c_double), ('gamma', C.c_double), ('k', C.c_double),
                ('t_0', C.c_double), ('adapt_step_size', C.c_int), ('adapt_metric', C.c_int),
                ('update_window', C.c_int), ('doubling', C.c_int), ('full_metric', C.c_int), ('metric_mat', C.c_void_p),
                ('chain_layout', C.c_int)]


class Tempering(C.Structure):  # bfhip_tempering
    _fields_ = [('base_S', C.c_void_p), ('base_lin', C.c_void_p), ('base_c0', C.c_double), ('logxi', C.c_double)]


class PolymodelDesc(C.Structure):  # bfhip_polymodel_desc
    _fields_ = [('d', C.c_int), ('m', C.c_int), ('c0', C.POINTER(C.c_double)), ('lin', C.POINTER(C.c_double)),
                ('quad', C.POINTER(C.c_double)), ('use_bound', C.c_int), ('mu', C.POINTER(C.c_double)),
                ('hess', C.POINTER(C.c_double)), ('alpha', C.c_double), ('f_mu', C.POINTER(C.c_double)),
                ('n2', C.c_int), ('mask2', C.POINTER(C.c_int)), ('cubic2', C.POINTER(C.c_double)),
                ('n3', C.c_int), ('mask3', C.POINTER(C.c_int)), ('cubic3', C.POINTER(C.c_double))]


class PipelineDesc(C.Structure):  # bfhip_pipeline_desc
    _fields_ = [('d', C.c_int), ('m', C.c_int), ('ranges', _dp), ('hard_bounds', _u8p), ('su_lo', _dp), ('su_diff', _dp),
                ('model', PolymodelDesc), ('y', _dp), ('prec', _dp), ('prec_diag', _dp), ('logp0', C.c_double),
                ('prior_mu', _dp), ('prior_prec', _dp), ('prior_c0', C.c_double),
                ('use_decay', C.c_int), ('decay_mu', _dp), ('decay_hess', _dp), ('decay_alpha2', C.c_double),
                ('decay_gamma', C.c_double)]


# every symbol include/bfhip.h declares: (restype, argtypes)
_vp = C.c_void_p
SYMBOLS = {
    'bfhip_version': (C.c_int, []),
    'bfhip_last_error': (C.c_char_p, []),
    'bfhip_ctx_create': (C.c_int, [C.POINTER(_vp), C.c_int, _vp]),
    'bfhip_ctx_destroy': (None, [_vp]),
    'bfhip_ctx_set_stream': (C.c_int, [_vp, _vp]),
    'bfhip_ctx_synchronize': (C.c_int, [_vp]),
    'bfhip_density_upload': (C.c_int, [_vp, C.POINTER(DensityDesc)]),
    'bfhip_logp_grad': (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, _vp]),
    'bfhip_constraint': (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    'bfhip_leapfrog': (C.c_int, [_vp, C.c_int] + [_vp] * 8),
    'bfhip_sampler_run': (C.c_int, [_vp, C.POINTER(SamplerConfig), C.c_int, C.c_int, _vp, _vp, _vp, C.c_int,
                                    C.c_int, _vp, _vp, _vp]),
    'bfhip_tnuts_run': (C.c_int, [_vp, C.POINTER(SamplerConfig), C.POINTER(Tempering), C.c_int, C.c_int, _vp, _vp, _vp, _vp,
                                  C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    'bfhip_rng_seed': (C.c_int, [_vp, C.c_int, C.c_uint64, C.c_uint64, _vp]),
    'bfhip_chain_init': (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_double, _vp, _vp, C.c_double, C.c_int, _vp, _vp]),
    'bfhip_metric_init_full': (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_double, _vp, _vp]),
    'bfhip_polymodel_upload': (C.c_int, [_vp, C.POINTER(PolymodelDesc)]),
    'bfhip_polymodel_eval': (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    'bfhip_chi2_stage': (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_double, _vp, _vp]),
    'bfhip_pipeline_upload': (C.c_int, [_vp, C.POINTER(PipelineDesc)]),
    'bfhip_design_block': (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int]),
    'bfhip_gram': (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp, _vp]),
    'bfhip_tree_size_mode_share': (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_double, _vp]),
    'bfhip_solve_spd': (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    'bfhip_lstsq': (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp, _vp, C.c_int, _vp, _vp]),
    'bfhip_sort_keys': (C.c_int, [_vp, C.c_long, _vp, _vp, _vp]),
    'bfhip_order_keys': (C.c_int, [_vp, C.c_long, _vp, _vp]),
    'bfhip_count_keys': (C.c_int, [_vp, C.c_long, _vp, C.c_long, _vp, C.c_int, _vp]),
    'bfhip_importance_weights': (C.c_int, [_vp, C.c_long, _vp, _vp, C.c_double, _vp, _vp]),
    'bfhip_polar_ns': (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp]),
    'bfhip_ndtri': (C.c_int, [_vp, C.c_long, _vp, _vp]),
    'bfhip_ica_tanh': (C.c_int, [_vp, C.c_long, C.c_long, C.c_int, _vp, _vp]),
    'bfhip_ica_assemble': (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_long, C.c_long, _vp, _vp, _vp, _vp]),
    'bfhip_ica_post': (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    'bfhip_spline_build': (C.c_int, [_vp, C.c_int, C.c_long, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_double,
                                     C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    'bfhip_kde_cdf': (C.c_int, [_vp, C.c_int, C.c_long, _vp, _vp, _vp, C.c_int, _vp, _vp]),
    'bfhip_spline_apply': (C.c_int, [_vp, C.c_int, C.c_long, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    'bfhip_bridge_sums': (C.c_int, [_vp, C.c_long, _vp, C.c_long, _vp, C.c_double, _vp]),
    'bfhip_bridge_terms': (C.c_int, [_vp, C.c_long, _vp, _vp, C.c_long, _vp, _vp, C.c_double, _vp, _vp]),
}

_lib = None


def build(verbose=False):
    """Compile libbfhip.so for gfx950 in tree (hipcc cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(['make', '-C', CSRC, '-j4'], stdout=out)
    return LIB_PATH


def lib():
    """The loaded library.  Raises RuntimeError when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'bayesfast_amd: the HIP extension %s is missing; build it with '
                '`python -c "import __graft_entry__ as g; g.build()"` or `make -C bayesfast_amd/csrc`. '
                'There is no CPU fallback.' % LIB_PATH)
        # PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so): it must be in the process BEFORE this library is, so that
        # the library's libamdhip64 dependency binds to the one PyTorch uses -- two HIP runtimes in one process and the second finds
        # "no ROCm-capable device" (seen when a test fixture loaded the library ahead of the first `import torch`)
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


class BfhipError(RuntimeError):
    pass


# ---- test / tuning switches (include/bfhip_debug.h): used by tests/, tools/ and bench.py's measurement legs only ----
def debug_set(key, value):
    """One of the library's integer switches by name (process-wide)."""
    f = lib().bfhip_debug_set
    f.restype, f.argtypes = C.c_int, [C.c_char_p, C.c_longlong]
    check(f(key.encode(), int(value)))


def debug_get(key):
    f = lib().bfhip_debug_get
    f.restype, f.argtypes = C.c_longlong, [C.c_char_p]
    return int(f(key.encode()))


def debug_buffer(key, tensor_or_none):
    """Attach (a device tensor) or detach (None) one of the measurement buffers."""
    f = lib().bfhip_debug_buffer
    f.restype, f.argtypes = C.c_int, [C.c_char_p, C.c_void_p]
    check(f(key.encode(), None if tensor_or_none is None else C.c_void_p(tensor_or_none.data_ptr())))


def last_kernel():
    """The kernel the last bfhip_sampler_run dispatched to."""
    f = lib().bfhip_debug_last_kernel
    f.restype = C.c_char_p
    return f().decode()


def check(rc):
    if rc != 0:
        msg = lib().bfhip_last_error().decode()
        exc = {-1: ValueError, -2: RuntimeError, -4: NotImplementedError}.get(rc, BfhipError)
        raise exc('bfhip error %d: %s' % (rc, msg))
