"""Host-side helpers of the drop-in boundary that need neither the GPU nor the native library's compute entry points."""
import numpy as np
import torch

import nerf4k_amd  # noqa: F401


def test_batch_indices_generator_covers_every_index_between_reshuffles():
    """dvgo.batch_indices_generator (lib/dvgo.py:761-768): batches of BS out of a permutation of N, reshuffled when fewer than BS remain."""
    from nerf4k_amd.lib import dvgo as D
    np.random.seed(3)
    gen = D.batch_indices_generator(103, 10)
    seen = torch.cat([next(gen) for _ in range(10)])
    assert seen.dtype == torch.int64 and seen.numel() == 100 and seen.unique().numel() == 100 and int(seen.max()) < 103
    nxt = next(gen)                                  # 3 left < 10: new permutation
    assert nxt.numel() == 10 and nxt.unique().numel() == 10
