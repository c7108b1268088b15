"""Data-parallel gradient exchange of the decoder (4k-nerf_amd/lib/sr_train.allreduce_gradients): world_size-2 gloo test on CPU.
The 458 gradient tensors of SFTNet(3, 4, 64, 5, 32, 1) (3,955,811 parameters = 15.8 MB) go through ONE all-reduce; every rank must
end with the mean of the ranks' gradients, identical on all ranks, parameters without a local gradient contributing zeros."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import nerf4k_amd  # noqa: F401


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _grads(rank, net):
    g = torch.Generator().manual_seed(100 + rank)
    out = []
    for i, p in enumerate(net.parameters()):
        out.append(None if (i % 97 == 5 and rank == 1) else torch.randn(p.shape, generator=g))
    return out


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from nerf4k_amd.lib import sr_esrnet, sr_train
        torch.manual_seed(0)
        net = sr_esrnet.SFTNet(3, scale=4, num_feat=64, num_block=5, num_grow_ch=32, num_cond=1)
        for p, g in zip(net.parameters(), _grads(rank, net)):
            p.grad = g
        nbytes = sr_train.allreduce_gradients(list(net.parameters()))
        flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        q.put((rank, nbytes, flat.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_of_the_decoder_gradients():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, nbytes, flat = q.get(timeout=240)
        got[r] = (nbytes, flat)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from nerf4k_amd.lib import sr_esrnet
    torch.manual_seed(0)
    net = sr_esrnet.SFTNet(3, scale=4, num_feat=64, num_block=5, num_grow_ch=32, num_cond=1)
    n_params = sum(p.numel() for p in net.parameters())
    assert len(list(net.parameters())) == 458 and n_params == 3955811 and got[0][0] == 4 * n_params      # 15.8 MB, one bucket
    parts = []
    for i, p in enumerate(net.parameters()):
        g0, g1 = _grads(0, net)[i], _grads(1, net)[i]
        parts.append(((g0 if g0 is not None else torch.zeros(p.shape)) + (g1 if g1 is not None else torch.zeros(p.shape))).reshape(-1) / 2)
    want = torch.cat(parts).numpy()
    assert np.array_equal(got[0][1], got[1][1])                      # replicas stay identical
    np.testing.assert_allclose(got[0][1], want, rtol=0, atol=1e-7)


def test_single_process_is_a_no_op_on_values():
    from nerf4k_amd.lib import sr_train
    a, b = torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(2, 2))
    a.grad = torch.tensor([1., 2., 3.])
    n = sr_train.allreduce_gradients([a, b])
    assert n == 28 and torch.equal(a.grad, torch.tensor([1., 2., 3.])) and torch.equal(b.grad, torch.zeros(2, 2))


def test_gradient_bucket_round_trip_without_a_process_group():
    """allreduce_gradients outside a job (world 1): the bucket is one concatenation in and one multi-tensor copy out -- gradients keep their values and their
    storage (the decoder's hand-over gives every parameter a view of ONE flat buffer), a parameter without a gradient receives zeros, frozen ones are skipped."""
    import torch
    from nerf4k_amd.lib import sr_train
    g = torch.Generator().manual_seed(3)
    ps = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in ([4, 3, 3, 3], [4], [5, 7], [2], [6])]
    ps[3].requires_grad_(False)
    flat = torch.randn([4 * 27 + 4 + 35], generator=g)
    want = flat.clone()
    ps[0].grad, ps[1].grad, ps[2].grad = flat[:108].view(4, 3, 3, 3), flat[108:112], flat[112:147].view(5, 7)
    ptrs = [p.grad.data_ptr() for p in ps[:3]]
    n = sr_train.allreduce_gradients(ps)
    assert n == 4 * (108 + 4 + 35 + 6)
    assert torch.equal(flat, want) and [p.grad.data_ptr() for p in ps[:3]] == ptrs
    assert ps[3].grad is None and torch.equal(ps[4].grad, torch.zeros([6]))
