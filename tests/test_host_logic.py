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


def test_dense_block_backward_layout_is_disjoint_and_ordered():
    """sr_train._rdb_bwd_layout: the float offsets of a dense block's backward buffers inside its two allocations -- the five [dW | dbias] pairs first and
    contiguous (the span ONE launch zeroes), then the sixteen SFT gradients; scratch pieces disjoint, 16-byte aligned, gc0 / gc1 only without an accumulator."""
    from nerf4k_amd.lib import sr_train
    nf, g, n = 64, 32, 37 * 45
    shapes = ([(32, 32, 1, 1), (32,), (nf, 32, 1, 1), (nf,)] * 2 +
              [s for k in range(4) for s in ((g, nf + k * g, 3, 3), (g,))] + [(nf, nf + 4 * g, 3, 3), (nf,)] +
              [(32, 32, 1, 1), (32,), (g, 32, 1, 1), (g,)] * 2)
    P = [torch.zeros(s) for s in shapes]
    assert len(P) == 26
    sr_train._RDB_LAYOUTS.clear()
    for with_gc in (True, False):
        order, sizes, offs, span, soff, nb0, nb1, shp = sr_train._rdb_bwd_layout(P, n, nf, g, with_gc)
        assert order[:10] == list(range(8, 18)) and sorted(order) == list(range(26))
        assert sizes == [P[i].numel() for i in order] and offs[0] == 0 and all(b - a == s for a, b, s in zip(offs, offs[1:], sizes))
        assert span == sum(P[i].numel() for i in range(8, 18)) == offs[10]
        assert [s is None for s in shp] == [P[i].dim() == 1 for i in order]
        assert len(soff) == (9 if with_gc else 7) and all(a < b and a % 4 == 0 for a, b in zip(soff, soff[1:]))
        assert soff[1] - soff[0] == n * nf and soff[2] - soff[1] == n * (nf + 4 * g) and soff[3] - soff[2] == n * g and soff[6] - soff[5] == n * nf
        assert (soff[4] - soff[3]) * 4 == nb0 and (soff[5] - soff[4]) * 4 == nb1 and nb0 > 0 and nb1 > 0
    assert sr_train._rdb_bwd_layout(P, n, nf, g, True) is sr_train._rdb_bwd_layout(P, n, nf, g, True)        # cached per shape


def test_condition_fan_and_direct_gradient_hand_over_on_cpu_tensors():
    """sr_train._CondFan returns the accumulator (plus whatever reached it through autograd) and marks it spent; _hand_over_grads does what a leaf's
    AccumulateGrad does: assign, then add; frozen parameters and missing gradients are skipped."""
    from nerf4k_amd.lib import sr_train
    from nerf4k_amd import _native as N
    c = torch.randn(3, 4, 32, requires_grad=True)
    acc = torch.zeros_like(c)
    out = sr_train._CondFan.apply(c, acc)
    acc += 2.0                                                     # what the fused consumers do inside their backward kernels
    (out * 3.0).sum().backward()                                   # a consumer outside them
    assert torch.equal(c.grad, torch.full_like(c, 5.0)) and getattr(acc, '_k4_spent', False)
    try:
        sr_train._check_acc(acc)
        raise AssertionError('a spent accumulator must be refused')
    except N.K4Error:
        pass
    sr_train._check_acc(None)
    a, b, f = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2)), torch.nn.Parameter(torch.zeros(2), requires_grad=False)
    g1 = [torch.ones(3), None, torch.ones(2)]
    sr_train._hand_over_grads([a, b, f], g1)
    assert a.grad is g1[0] and b.grad is None and f.grad is None
    sr_train._hand_over_grads([a, b, f], [torch.full((3,), 2.0), torch.ones(2), None])
    assert torch.equal(a.grad, torch.full((3,), 3.0)) and torch.equal(b.grad, torch.ones(2))


def test_dense_grid_pending_update_hooks_are_inert_without_one():
    """lib/grid.DenseGrid: params_ready / state_dict / .to() / deepcopy with no optimizer step pending (CPU grids never have one) behave like the plain module."""
    import copy
    from nerf4k_amd.lib import grid
    g = grid.DenseGrid(3, [4, 5, 6], [0, 0, 0], [1, 1, 1])
    with torch.no_grad():
        g.grid.copy_(torch.randn_like(g.grid))
    g.params_ready()
    assert g._k4_pending is None and g._k4_seed is None
    g2 = copy.deepcopy(g)
    assert g2.grid is not g.grid and torch.equal(g2.grid, g.grid) and sorted(g2.state_dict()) == ['grid', 'xyz_max', 'xyz_min']
    g3 = copy.deepcopy(g).double()
    assert g3.grid.dtype == torch.float64 and torch.equal(g3.grid.float(), g.grid)
    assert torch.equal(g.get_dense_grid(), g.grid)


def test_fast_parameter_list_tracks_replaced_parameters():
    """SFTNet.k4_parameters: the cached (module, name, parameter) triples behind the per-call cache keys equal nn.Module.parameters() -- also after a Parameter
    object was replaced, a module moved / cast (same objects, new storage) or updated in place (version bump seen through the same objects)."""
    import torch.nn as nn
    from nerf4k_amd.lib import sr_esrnet
    net = sr_esrnet.SFTNet(3, scale=4, num_block=1)
    a = net.k4_parameters()
    assert len(a) == len(list(net.parameters())) and all(x is y for x, y in zip(a, net.parameters()))
    assert net.k4_parameters() is a                                     # validated, not rebuilt
    v0 = tuple(p._version for p in a)
    with torch.no_grad():
        net.conv_last.bias.add_(1.0)
    assert tuple(p._version for p in net.k4_parameters()) != v0
    net.conv_first.weight = nn.Parameter(torch.zeros_like(net.conv_first.weight))
    b = net.k4_parameters()
    assert b is not a and all(x is y for x, y in zip(b, net.parameters()))
    net.double()
    c = net.k4_parameters()
    assert all(x is y for x, y in zip(c, net.parameters())) and c[0].dtype == torch.float64
