#!/usr/bin/env python3
"""Build-time check of the FS = 17 kernel's hand-scheduled matrix instructions.

bf_sampler_kernel<8, true, false, 17, 0> issues v_mfma_f64_4x4x4_4b through inline assembly (bfhip_sampler.hip: BF_MFMA_Q_ACC /
BF_MFMA_Q_DONE), so the compiler's hazard recogniser does not see the instructions: the wait states between dependent ones are
the s_nop counts written by hand.  This script compiles bfhip_sampler.hip to gfx950 assembly (hipcc -S, ~4 minutes), finds that
kernel and asserts, for every inline-assembly v_mfma_f64_4x4x4_4b_f64:

  * the instruction right before it is the `s_nop 1` of its own macro (nothing scheduled in between);
  * between two such MFMAs that write the SAME accumulator register there is no instruction that reads or writes that register
    (no v_mov / v_accvgpr copy, no scratch spill or reload of an accumulator in the middle of a chain);
  * every chain's last MFMA is followed, before the first read of its accumulator, by the `s_nop 7; s_nop 7` pair.

Run it after every compiler or flag change (the bit-identity test cubic_form 4 vs 8 in tests/test_gpu_sampler.py is the run-time
gate).  usage: python3 tools/check_fs17_hazards.py [existing.s]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = os.path.join(tempfile.mkdtemp(), 'bfhip_sampler.s')
        subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-S', '--cuda-device-only', '-o', path,
                               'bfhip_sampler.hip'], cwd=os.path.join(ROOT, 'bayesfast_amd', 'csrc'), stderr=subprocess.DEVNULL)
    lines = open(path).read().splitlines()
    # the kernel: bf_sampler_kernel<8, true, false, 17, 0> = _Z17bf_sampler_kernelILi8ELb1ELb0ELi17ELi0EE...
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z17bf_sampler_kernelILi8ELb1ELb0ELi17ELi0EE\w*:', l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
    # instructions of the kernel; inline assembly is bracketed by ;;#ASMSTART / ;;#ASMEND in the compiler's output (the MFMAs the
    # compiler itself emits for __builtin_amdgcn_mfma_* -- the wave reductions -- carry its own hazard handling and are not checked)
    body, in_asm, is_asm = [], False, []
    for l in lines[start:end]:
        t = l.strip()
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        t = t.split(';')[0].strip()
        if not t or t.startswith('.') or t.endswith(':'):
            continue
        body.append(t)
        is_asm.append(in_asm)
    mf = [i for i, l in enumerate(body) if l.startswith('v_mfma_f64_4x4x4_4b_f64') and is_asm[i]]
    assert mf, 'no inline-assembly v_mfma_f64_4x4x4_4b_f64 in the FS = 17 kernel'
    bad = []
    for i in mf:
        if body[i - 1] != 's_nop 1':
            bad.append((i, 'not preceded by its s_nop 1: ' + body[i - 1]))
    # chains: consecutive MFMAs with the same destination register
    def dst(l):
        return l.split()[1].rstrip(',')
    last_of = {}
    for k, i in enumerate(mf):
        d = dst(body[i])
        if d in last_of:
            for j in range(last_of[d] + 1, i):
                if re.search(r'\b' + re.escape(d) + r'\b', body[j]) and not (body[j].startswith('v_mfma_f64_4x4x4_4b_f64') and is_asm[j]):
                    bad.append((j, 'touches accumulator %s inside its chain: %s' % (d, body[j])))
        last_of[d] = i
    # after the last MFMA of every chain segment: the first later use of the accumulator must come after `s_nop 7; s_nop 7`
    n_done = 0
    for d, i in last_of.items():
        for j in range(i + 1, len(body)):
            if re.search(r'\b' + re.escape(d) + r'\b', body[j]) and not body[j].startswith('v_mfma_f64_4x4x4_4b_f64'):
                window = body[i + 1:j]
                if window.count('s_nop 7') < 2:
                    bad.append((j, 'first read of %s without the s_nop 7 pair behind the chain: %s' % (d, body[j])))
                n_done += 1
                break
    print('FS = 17 kernel: %d inline v_mfma_f64_4x4x4_4b_f64, %d accumulators, %d findings' % (len(mf), len(last_of), len(bad)))
    for b in bad[:20]:
        print('  line %d: %s' % b)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
