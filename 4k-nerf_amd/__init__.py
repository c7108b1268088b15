"""4K-NeRF hot path, MI355X-native (gfx950 / CDNA4).

Product package: the fused HIP voxel-grid ray marcher and the MFMA RRDB/SFT super-resolution
decoder behind the reference's Python API (``lib.dvgo.DirectVoxGO``, ``lib.dmpigo.DirectMPIGO``,
``lib.grid``, ``lib.sr_esrnet.SFTNet``, ``render_utils_cuda``).  Kernels live in
``csrc/*.hip`` and are reached through the C ABI declared in ``include/k4nerf.h`` (loaded with
ctypes by ``_native.py``).  There is NO CPU fallback: every op raises if the HIP library is
missing or a tensor is not on the GPU.  Import as ``nerf4k_amd`` (see ``/nerf4k_amd.py``).
"""
__version__ = '0.1.0'
