"""Shared test helpers (golden loading, error metrics)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_march_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    kw = json.loads(str(z['model_kwargs_json']))
    for k in ('xyz_min', 'xyz_max'):
        kw[k] = np.asarray(kw[k], dtype=np.float32)
    sd, inp, out = {}, {}, {}
    for k in z.files:
        if k.startswith('sd/'):
            sd[k[3:]] = torch.from_numpy(z[k])
        elif k.startswith('in/'):
            inp[k[3:]] = torch.from_numpy(z[k])
        elif k.startswith('out/'):
            out[k[4:]] = torch.from_numpy(z[k])
    return {'model_class': str(z['model_class']), 'model_kwargs': kw, 'model_state_dict': sd,
            'render_kwargs': json.loads(str(z['render_kwargs_json'])), 'rays': inp, 'out': out}


def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 200.0 if mse == 0 else -10.0 * np.log10(mse)
