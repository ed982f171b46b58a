"""CPU sanitizer runs (race / memory checking; GPU AddressSanitizer is not available on the pool).

Two builds with -fsanitize=address,undefined (-fno-sanitize-recover: any finding aborts the run):
  * `make -C oracle asan`    : the C restatement, driven by the whole replay suite of tests/test_oracle_golden.py
                               (every fixture of the reference: kernels, transforms, leapfrog states, NUTS / HMC trajectories);
  * `make -C tests/emu asan` : the group sampler kernel's OWN source (bayesfast_amd/csrc/bfhip_group.h) on the fibre emulator,
                               NUTS trajectories against the oracle with and without decay + transform (the instantiations of
                               d <= 16; the full emulator suite passes under the same sanitizers in 11 minutes).
Each runs in a child python with libasan preloaded (the sanitizer runtime must be the first DSO of the process)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asan_env(**extra):
    asan = subprocess.check_output(['gcc', '-print-file-name=libasan.so'], text=True).strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip('libasan is not installed')
    return dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1', UBSAN_OPTIONS='print_stacktrace=1',
                **extra)


def test_oracle_under_address_and_ub_sanitizers():
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), 'asan', '-s'])
    env = _asan_env(BF_ORACLE_LIB=os.path.join(ROOT, 'oracle', '_build', 'libbf_oracle_asan.so'))
    r = subprocess.run([sys.executable, '-m', 'pytest', 'tests/test_oracle_golden.py', '-q', '-x', '-p', 'no:cacheprovider'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and ' passed' in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    assert 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr, r.stderr[-3000:]


_EMU_CASES = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join('tests'))
sys.path.insert(0, os.path.join('tests', 'golden'))
import test_group_emu as t
samp = np.load(os.path.join('tests', 'golden', 'sampler.npz'))
# plain surrogate, d = 16: the replay fixture's shape (40 iterations, warm-up 25, five chains out of step)
spec = t._spec(samp, 'plain16.')
x0 = samp['plain16.x0'][:5]
t._compare_nuts(t._emu(spec, x0, 40, 25), t._oracle(spec, x0, 40, 25), 40)
# decay + constraint transform with all four kinds of bounds, d = 11 (one wave, odd size: padded lanes)
from bayesfast_amd.workloads import correlated_gaussian_spec
d = 11
spec = dict(correlated_gaussian_spec(d)[0])
spec.update(use_decay=True, decay_mu=spec['poly']['mu'], decay_hess=spec['poly']['hess'],
            decay_alpha2=float(spec['poly']['alpha'])**2 * 0.6, decay_gamma=0.1)
lo = np.full(d, -9.) + np.arange(d) * 0.01
spec.update(ranges=np.stack([lo, lo + 18.], 1), hard_bounds=np.array(([[1, 1], [1, 0], [0, 1], [0, 0]] * d)[:d], dtype=np.uint8))
x0 = np.random.default_rng(8).normal(size=(19, d)) * 0.3       # two groups, the second one ragged
t._compare_nuts(t._emu(spec, x0, 12, 8), t._oracle(spec, x0, 12, 8), 12, n_head=6, tol_head=1e-8)
print('emu-asan-ok')
'''


def test_group_kernel_source_under_address_and_ub_sanitizers():
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'tests', 'emu'), 'asan', '-s'])
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), 'asan', '-s'])
    env = _asan_env(BF_EMU_LIB=os.path.join(ROOT, 'tests', 'emu', '_build', 'libbf_emu_asan.so'),
                    BF_ORACLE_LIB=os.path.join(ROOT, 'oracle', '_build', 'libbf_oracle_asan.so'))
    r = subprocess.run([sys.executable, '-c', _EMU_CASES], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and 'emu-asan-ok' in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    assert 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr, r.stderr[-3000:]
