"""Build lib4k_hip.so (all HIP kernels + the C ABI of include/k4nerf.h) for gfx950, in-tree.

hipcc cross-compiles without a GPU.  Objects are cached by source mtime; the .so sits next to this
file so it travels to the GPU box with the repo snapshot (git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
OUT = os.path.join(PKG, 'lib4k_hip.so')
ARCH = 'gfx950'
SOURCES = ['k4_march.hip', 'k4_staged.hip', 'k4_sr.hip', 'k4_sr_p16.hip', 'k4_sr_bwd.hip', 'k4_opt.hip', 'k4_train.hip', 'k4_tape.hip']  # missing files are skipped
# -ffp-contract=on: a*b+c fuses only where the SOURCE expression says so (and in explicit fmaf).  hipcc's default
# (fast-honor-pragmas) lets the backend fuse any fmul/fadd pair it finds after inlining, so two inlined copies of one
# routine (expf/powf included) could round differently -- a ray's result then depended on which copy served its sample.
FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-ffp-contract=on', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]


FLAGS += os.environ.get('K4_EXTRA_HIPCC_FLAGS', '').split()      # experiments only (e.g. -DK4_SHADE_WG_PER_CU=2)


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(verbose=False, force=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objdir = os.path.join(PKG, 'build')
    os.makedirs(objdir, exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + \
           [os.path.join(ROOT, 'include', 'k4nerf.h')]
    objs, relink = [], force
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(objdir, src + '.o')
        if force or _newer(s, o) or any(_newer(d, o) for d in deps):
            cmd = [hipcc] + FLAGS + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
            relink = True
        objs.append(o)
    if relink or not os.path.exists(OUT) or any(_newer(o, OUT) for o in objs):
        cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', OUT] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    print(build(verbose=True, force='--force' in sys.argv))
