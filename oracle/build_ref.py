"""Build the REFERENCE's own native extensions for gfx950 into oracle/_ref/.  TEST INFRASTRUCTURE ONLY.

    python oracle/build_ref.py            (also run by __graft_entry__.build() when /root/reference exists)

What it is: the three pybind11/libtorch extensions the reference JIT-builds at import time
(``render_utils_cuda``  lib/dvgo.py:14-19  <- lib/cuda/render_utils.cpp + render_utils_kernel.cu,
 ``adam_upd_cuda``      lib/masked_adam.py:7-10 <- lib/cuda/adam_upd.cpp + adam_upd_kernel.cu,
 ``total_variation_cuda`` lib/grid.py:19-24 <- lib/cuda/total_variation.cpp + total_variation_kernel.cu),
compiled from the sources WHERE THEY LIE under /root/reference by the same ``torch.utils.cpp_extension.load`` call the
reference itself makes -- on PyTorch-ROCm that call hipifies the ``.cu`` files and drives hipcc, which cross-compiles
gfx950 without a GPU.  The outputs (``oracle/_ref/<name>_ref.so``) are git-ignored and travel to the GPU box with the
repo snapshot; they are the reference-made checker that pins ``oracle/native_cpu.py`` / ``oracle/optim.py`` and the
staged gfx950 kernels (``oracle/gen_native_golden.py`` -> ``tests/golden/native_*.npz``).  Nothing under
``4k-nerf_amd/`` ever loads them.

Two things the recipe has to do around the reference's own build call, both on a SCRATCH copy under $TMPDIR (the
reference tree is read-only and hipify writes its ``.hip`` output next to each source; no reference source enters
the repository):
  1. copy the 2 files of an extension to a temporary directory;
  2. in the ``.cu`` file only, rewrite ``AT_DISPATCH_FLOATING_TYPES(x.type(), ...`` to ``x.scalar_type()`` -- torch >= 2.x no longer
     converts the deprecated ``DeprecatedTypeProperties`` to a ``ScalarType`` (a torch API drift, 13 call sites, the kernels'
     bodies and launch geometry are untouched).
Compiler flags are torch's defaults for an extension (``-O3``, hipcc's default fp contraction = nvcc's ``-fmad=true``).
"""
import glob
import os
import re
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')
REFERENCE_CUDA = '/root/reference/lib/cuda'
EXTENSIONS = {
    'render_utils_cuda_ref': ('render_utils.cpp', 'render_utils_kernel.cu'),
    'adam_upd_cuda_ref': ('adam_upd.cpp', 'adam_upd_kernel.cu'),
    'total_variation_cuda_ref': ('total_variation.cpp', 'total_variation_kernel.cu'),
}


def available():
    return os.path.isdir(REFERENCE_CUDA)


def so_path(name):
    return os.path.join(OUT, name + '.so')


def _stale(name, srcs):
    so = so_path(name)
    if not os.path.exists(so):
        return True
    t = os.path.getmtime(so)
    return any(os.path.getmtime(os.path.join(REFERENCE_CUDA, s)) > t for s in srcs) or os.path.getmtime(__file__) > t


def build(verbose=False, force=False):
    """-> {name: path} of the built libraries; {} when the reference tree is absent (GPU box: prebuilt files are used)."""
    if not available():
        return {n: so_path(n) for n in EXTENSIONS if os.path.exists(so_path(n))}
    os.makedirs(OUT, exist_ok=True)
    os.environ['PYTORCH_ROCM_ARCH'] = 'gfx950'
    os.environ.setdefault('MAX_JOBS', str(min(8, os.cpu_count() or 1)))
    from torch.utils import cpp_extension
    built = {}
    for name, srcs in EXTENSIONS.items():
        if force or _stale(name, srcs):
            tmp = tempfile.mkdtemp(prefix='k4_ref_build_')
            try:
                local = []
                for s in srcs:
                    dst = os.path.join(tmp, s)
                    with open(os.path.join(REFERENCE_CUDA, s)) as f:
                        text = f.read()
                    if s.endswith('.cu'):
                        text = re.sub(r'(AT_DISPATCH_FLOATING_TYPES\(\s*[A-Za-z_0-9]+)\.type\(\)', r'\1.scalar_type()', text)
                    with open(dst, 'w') as f:
                        f.write(text)
                    local.append(dst)
                bdir = os.path.join(tmp, 'build')
                os.makedirs(bdir)
                cpp_extension.load(name=name, sources=local, build_directory=bdir, verbose=verbose, is_python_module=False)
                sos = glob.glob(os.path.join(bdir, name + '*.so'))
                assert sos, 'no library produced for ' + name
                shutil.copyfile(sos[0], so_path(name))
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
        built[name] = so_path(name)
    return built


def load(name):
    """Import a built reference extension as a Python module (needs ``import torch`` first for libtorch's symbols)."""
    import importlib.util
    import torch  # noqa: F401
    path = so_path(name)
    if not os.path.exists(path):
        raise RuntimeError(f'{path} missing: run `python oracle/build_ref.py` in the build container (needs /root/reference)')
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == '__main__':
    res = build(verbose='-v' in sys.argv, force='--force' in sys.argv)
    for k, v in res.items():
        print(k, '->', v)
    if not res:
        print('reference tree absent and nothing prebuilt')
