"""A/B of the NUTS kernels on the heterogeneous-tree workload of bench.py (hetero_rate) and on the config-4 style
workload (target_accept 0.95): LAYOUT=auto|group|wave BFHIP_NUTS_KERNEL=pipe|sliced python tools/hetero_ab.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from benchlib import blocks as bench   # (hetero_rate moved there in round 6)
from bayesfast_amd.device import get_context
ctx = get_context(0)
lay = os.environ.get('LAYOUT', 'auto')
C = int(os.environ.get('CHAINS', 4096))
r = bench.hetero_rate(ctx, 64, C, 2024, 250, layout=lay)
print('chains', C, 'BFHIP_WAVE_CPG', os.environ.get('BFHIP_WAVE_CPG', 'auto'), 'layout', lay, 'BFHIP_NUTS_KERNEL', os.environ.get('BFHIP_NUTS_KERNEL', '-'), 'hetero %.4g' % r['value'], 'mean tree', r['mean_tree_size'])
