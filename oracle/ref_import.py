"""Import the REAL reference Python modules in the build container.  TEST INFRASTRUCTURE ONLY.

``/root/reference/lib/{dvgo,dmpigo,grid}.py`` cannot be imported as shipped: they need
``torchvision`` (lib/dvgo.py:8) and ``torch_scatter`` (lib/dvgo.py:10), both absent here, and
JIT-compile ``lib/cuda/*.cu`` with nvcc at import time (lib/dvgo.py:14-19, lib/grid.py:12-24).
This helper installs three stubs for exactly those dependencies and then imports the
reference modules UNMODIFIED, so that their own Python control flow (mask order, in-place
aliasing, dict keys, PE layout, grid_sample call) runs on CPU and can pin ``oracle/marcher.py``:

  * ``torchvision.utils.save_image``      -> unused placeholder
  * ``torch_scatter.segment_coo``         -> ``out.index_add_`` (sum over a sorted index)
  * ``torch.utils.cpp_extension.load``    -> returns ``oracle.native_cpu`` for
    ``render_utils_cuda`` (the restated kernels) and an empty namespace for
    ``total_variation_cuda`` (training only)

``lib/sr_esrnet.py`` needs no stub at all (pure torch).

/root/reference does not exist on the GPU box: this module is only used by
``oracle/gen_golden.py`` and by tests that skip when the reference is absent.
Nothing is written to the read-only reference tree (PYTHONDONTWRITEBYTECODE).
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = '/root/reference'


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'lib'))


def _segment_coo(src, index, out=None, dim_size=None, reduce='sum'):
    assert reduce == 'sum'
    import torch
    if out is None:
        out = torch.zeros([dim_size] + list(src.shape[1:]), dtype=src.dtype)
    if src.numel():
        out.index_add_(0, index, src)
    return out


def _install_stubs():
    import torch.utils.cpp_extension as cpp_ext
    from . import native_cpu

    if 'torchvision' not in sys.modules:
        tv = types.ModuleType('torchvision')
        tvu = types.ModuleType('torchvision.utils')
        tvu.save_image = lambda *a, **k: None
        tv.utils = tvu
        sys.modules['torchvision'] = tv
        sys.modules['torchvision.utils'] = tvu
    if 'torch_scatter' not in sys.modules:
        ts = types.ModuleType('torch_scatter')
        ts.segment_coo = _segment_coo
        ts.scatter_add = lambda src, index, dim=0, out=None, dim_size=None: _segment_coo(
            src, index, out=out, dim_size=dim_size)
        sys.modules['torch_scatter'] = ts

    real_load = cpp_ext.load

    def fake_load(name, sources, **kwargs):
        if name == 'render_utils_cuda':
            return native_cpu
        return types.SimpleNamespace()
    cpp_ext.load = fake_load
    return cpp_ext, real_load


_cache = {}


def load_reference():
    """-> namespace(dvgo, dmpigo, grid, sr_esrnet) of the reference's own modules."""
    if 'ns' in _cache:
        return _cache['ns']
    if not available():
        raise RuntimeError('reference tree not present (expected on the build container only)')
    sys.dont_write_bytecode = True
    cpp_ext, real_load = _install_stubs()
    saved_path = list(sys.path)
    saved_lib = {k: v for k, v in sys.modules.items() if k == 'lib' or k.startswith('lib.')}
    for k in saved_lib:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        grid = importlib.import_module('lib.grid')
        dvgo = importlib.import_module('lib.dvgo')
        dmpigo = importlib.import_module('lib.dmpigo')
        sr_esrnet = importlib.import_module('lib.sr_esrnet')
    finally:
        cpp_ext.load = real_load
        sys.path[:] = saved_path
        # keep the reference modules reachable only through the returned namespace
        ref_mods = {k: v for k, v in sys.modules.items() if k == 'lib' or k.startswith('lib.')}
        for k in ref_mods:
            del sys.modules[k]
        sys.modules.update(saved_lib)
    ns = types.SimpleNamespace(grid=grid, dvgo=dvgo, dmpigo=dmpigo, sr_esrnet=sr_esrnet)
    _cache['ns'] = ns
    return ns
