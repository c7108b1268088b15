"""CPU oracle for the 4K-NeRF hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in the product package (``4k-nerf_amd/``) may import this package.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
use it, and only as the checker / the reported CPU baseline -- never as the thing
measured or shipped.

Parity pin status (see DESIGN.md "Oracle"):
  * SR decoder (``oracle/sr.py``): pinned against the UNMODIFIED reference module
    ``/root/reference/lib/sr_esrnet.py`` imported in the build container; golden
    vectors in ``tests/golden/sr_*.npz`` (generator: ``oracle/gen_golden.py``).
  * Marcher Python control flow (``oracle/marcher.py``): pinned against the
    reference's own ``DirectMPIGO.forward`` / ``DirectVoxGO.forward`` /
    ``DenseGrid`` / ``MaskGrid`` / ``get_rays_of_a_view`` executed in the build
    container with their three un-importable dependencies stubbed
    (``oracle/ref_import.py``); golden vectors in ``tests/golden/march_*.npz``.
  * The 13 native CUDA entry points (``oracle/native_cpu.py``) and the optimizer / TV
    kernels (``oracle/optim.py``): pinned against the reference's own ``lib/cuda``
    sources compiled for gfx950 (``oracle/build_ref.py`` -> ``oracle/_ref/*.so``, test-only,
    git-ignored) and run on an MI355X (``oracle/gen_native_golden.py``); vectors in
    ``tests/golden/native_*.npz`` / ``optim_ref.npz``.  ``native_march_*.npz`` re-evaluate the
    march fixtures with those compiled kernels serving every native step.
  * Training graph and joint step (``oracle/train_ops.py``): the colour MLP and the loss lines of
    ``run_sr.py:877-995`` are pinned through ``tests/golden/grad_joint.npz`` = one joint iteration
    on the reference's own ``DirectMPIGO`` + ``SFTNet`` modules; the 'patch_mimg' sampler through
    draws of the reference's own generator (``patch_sampler.npz``).  PARITY UNPINNED for one term:
    the distortion loss comes from the third-party package ``torch_efficient_distloss`` (PyPI 0.1.3,
    not vendored in /root/reference, not installed here) -- it is restated from its published
    definition (literal O(n^2) form) and the HIP kernel's prefix-sum form is checked against that.
"""
