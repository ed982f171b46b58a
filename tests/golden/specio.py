"""Flatten / rebuild density "spec" dicts (see oracle/oracle.py) into npz-storable key/value pairs."""
import numpy as np


def flatten_spec(spec, prefix=''):
    out = {}
    for k, v in spec.items():
        if k == 'poly':
            out.update(flatten_poly(v, prefix + 'poly.'))
        elif v is None:
            continue
        else:
            out[prefix + k] = np.asarray(v)
    return out


def flatten_poly(poly, prefix='poly.'):
    out = {}
    for k, v in poly.items():
        if k == 'configs':
            out[prefix + 'n_config'] = np.asarray(len(v))
            for i, cf in enumerate(v):
                out[prefix + 'cfg%d.order' % i] = np.asarray(cf['order'])
                out[prefix + 'cfg%d.input_mask' % i] = np.asarray(cf['input_mask'], dtype=np.int64)
                out[prefix + 'cfg%d.output_mask' % i] = np.asarray(cf['output_mask'], dtype=np.int64)
                out[prefix + 'cfg%d.coef' % i] = np.asarray(cf['coef'], dtype=np.float64)
        elif v is None:
            continue
        else:
            out[prefix + k] = np.asarray(v)
    return out


def _scalar(a):
    a = np.asarray(a)
    return a.item() if a.ndim == 0 else a


def rebuild_poly(z, prefix='poly.'):
    poly = {}
    n = int(z[prefix + 'n_config'])
    poly['configs'] = [dict(order=str(z[prefix + 'cfg%d.order' % i]),
                            input_mask=np.asarray(z[prefix + 'cfg%d.input_mask' % i]),
                            output_mask=np.asarray(z[prefix + 'cfg%d.output_mask' % i]),
                            coef=np.asarray(z[prefix + 'cfg%d.coef' % i])) for i in range(n)]
    for k in z.keys() if hasattr(z, 'keys') else z.files:
        if k.startswith(prefix) and '.cfg' not in k[len(prefix) - 1:] and k != prefix + 'n_config':
            poly[k[len(prefix):]] = _scalar(z[k])
    poly['input_size'] = int(poly['input_size'])
    poly['output_size'] = int(poly['output_size'])
    poly['use_bound'] = bool(poly.get('use_bound', False))
    return poly


def rebuild_spec(z, prefix=''):
    keys = list(z.keys()) if hasattr(z, 'keys') else list(z.files)
    spec = {}
    for k in keys:
        if not k.startswith(prefix):
            continue
        kk = k[len(prefix):]
        if kk.startswith('poly.') or '.' in kk:
            continue
        spec[kk] = _scalar(z[k])
    spec['poly'] = rebuild_poly(z, prefix + 'poly.')
    spec['d'] = int(spec['d'])
    spec['use_decay'] = bool(spec.get('use_decay', False))
    for k in ('ranges', 'hard_bounds', 'su_lo', 'su_diff'):
        spec.setdefault(k, None)
    return spec


def rebuild_pipeline(z):
    """pipeline.npz (make_golden.gen_pipeline): [multi-output surrogate with cubic configs, chi-square with a full
    precision matrix], no transforms -> density spec with a 'chi2' stage."""
    poly = rebuild_poly(z, 'poly.')
    return dict(d=poly['input_size'], ranges=None, hard_bounds=None, su_lo=None, su_diff=None, poly=poly, use_decay=False,
                chi2=dict(y=np.asarray(z['ydat']), prec=np.asarray(z['prec']), logp0=0.))


def rebuild_pipeline_des(z, tag):
    """pipeline_des.npz (make_golden.gen_pipeline_des), case 'a' (no decay) or 'b' (decay): [surrogate, whitened chi-square,
    like + Gaussian prior on some inputs] behind input scales and hard bounds -> density spec with 'chi2' and 'prior'."""
    poly = rebuild_poly(z, tag + '.poly.')
    d, m = poly['input_size'], poly['output_size']
    pm, pp = np.zeros(d), np.zeros(d)
    pm[z['prior.idx']] = z['prior.mu']
    pp[z['prior.idx']] = 1. / np.asarray(z['prior.sig'])**2
    spec = dict(d=d, ranges=np.asarray(z[tag + '.ranges']), hard_bounds=np.asarray(z[tag + '.hard_bounds']),
                su_lo=np.asarray(z[tag + '.su_lo']), su_diff=np.asarray(z[tag + '.su_diff']), poly=poly,
                use_decay=bool(z[tag + '.use_decay']),
                chi2=dict(y=np.asarray(z['chi2.y']), prec_diag=np.ones(m), logp0=float(z['chi2.logp0'])),
                prior=dict(mu=pm, prec_diag=pp, c0=float(z['prior.c0'])))
    if spec['use_decay']:
        spec.update(decay_mu=np.asarray(z[tag + '.decay_mu']), decay_hess=np.asarray(z[tag + '.decay_hess']),
                    decay_alpha2=float(z[tag + '.decay_alpha2']), decay_gamma=float(z[tag + '.decay_gamma']))
    return spec
