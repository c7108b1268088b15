"""RCCL on ONE GPU: a world-size-1 `nccl` process group (nccl IS RCCL on ROCm) that really runs the PRODUCTION collective calls of the N > 1
paths on device buffers -- what a 1-GPU box can prove about them: librccl loads, a communicator is created on the leased device, and
  * tile_parallel.render_frame_tiles issues its ONE all_gather_into_tensor of final pixels (fp32 and the uint8 form),
  * joint_train.exchange_gradients issues the sparse voxel-grid all_gather(s) and the dense-bucket all_reduce of the decoder,
with results equal to the collective-free single-process path (_native.FORCE_COLLECTIVES makes the calls run at world size 1; production
code only takes them at world size > 1).  Prints one JSON line.  `python tools/rccl_world1_smoke.py` on the GPU box;
tests/test_rccl_gpu.py runs it in a subprocess.  Multi-GPU behaviour stays UNMEASURED ON HARDWARE (no multi-GPU node in any round)."""
import json, os, socket, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist


def main():
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    t0 = time.time()
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', world_size=1, rank=0, device_id=dev)
    import nerf4k_amd  # noqa: F401
    from nerf4k_amd import _native as N, scene, tile_parallel as tp, joint_train
    from nerf4k_amd.lib import dvgo, sr_esrnet, utils
    out = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(), 'device': torch.cuda.get_device_name(0)}
    # ---- a plain collective first: communicator creation shows up here
    x = torch.arange(1024, dtype=torch.float32, device=dev)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    assert torch.equal(x, torch.arange(1024, dtype=torch.float32, device=dev))
    out['communicator_s'] = round(time.time() - t0, 2)
    # ---- tile-parallel frame: 4 tiles, HIP marcher + HIP decoder
    ck = scene.make_llff_checkpoint(seed=11, num_voxels=48 * 48 * 32, mpi_depth=32)
    H, W = 44, 60
    K = scene.LLFF_K.copy()
    K[:2] *= W / scene.LLFF_HW[1]
    pose = scene.llff_spiral_poses()[3]
    model = utils.model_from_checkpoint_dict(ck).to(dev).eval()
    torch.manual_seed(31)
    net = sr_esrnet.SFTNet(3, scale=4, num_block=2).to(dev).eval()
    with torch.no_grad():
        rays = dvgo.get_rays_of_a_view(H, W, K, torch.from_numpy(pose).to(dev), True, False, False, False)
        march_fn, sr_fn = tp.hip_march_fn(model, ck['render_kwargs']), tp.hip_sr_fn(net)
        plain = tp.render_frame_tiles(rays, H, W, march_fn, sr_fn, 30).clone()
        plain8 = tp.render_frame_tiles(rays, H, W, march_fn, sr_fn, 30, out_dtype=torch.uint8).clone()
        N.FORCE_COLLECTIVES = True
        coll = tp.render_frame_tiles(rays, H, W, march_fn, sr_fn, 30)
        coll8 = tp.render_frame_tiles(rays, H, W, march_fn, sr_fn, 30, out_dtype=torch.uint8)
        torch.cuda.synchronize()
    assert torch.equal(plain, coll) and torch.equal(plain8, coll8)
    out['tile_all_gather'] = {'frame': list(coll.shape), 'fp32_equal': True, 'uint8_equal': True}
    # ---- gradient exchange of the joint step: sparse voxel-grid lists + the dense bucket
    model.train(); net.train()
    g = torch.Generator(device=dev).manual_seed(3)
    for p in list(model.parameters()) + list(net.parameters()):
        if not p.requires_grad:
            continue
        if p.dim() == 5 and p.numel() >= 4096:                  # a grid: ~3 % of the voxels touched, as a ray patch leaves them
            V = p[0, 0].numel()
            sel = torch.rand([V], device=dev, generator=g) < 0.03
            p.grad = (torch.randn(p.shape, device=dev, generator=g) * sel.view(1, 1, *p.shape[2:])).contiguous()
        else:
            p.grad = torch.randn(p.shape, device=dev, generator=g)
    before = {id(p): p.grad.clone() for p in list(model.parameters()) + list(net.parameters()) if p.grad is not None}
    old_min = joint_train.SPARSE_MIN_NUMEL
    joint_train.SPARSE_MIN_NUMEL = 4096                          # the small scene's grids take the sparse path like the 1.36 GB one
    stats = joint_train.exchange_gradients(model, net)
    joint_train.SPARSE_MIN_NUMEL = old_min
    torch.cuda.synchronize()
    for p in list(model.parameters()) + list(net.parameters()):
        if p.grad is not None:
            assert torch.equal(p.grad, before[id(p)]), 'a one-rank exchange must return the gradients it was given'
    out['gradient_exchange'] = {'sparse_bytes_gathered': int(stats.get('bytes_gathered', 0)), 'dense_bucket_bytes': int(stats.get('bytes_dense', 0)),
                                'touched': [[c, v] for c, v in stats.get('touched', [])], 'gradients_equal': True}
    assert out['gradient_exchange']['sparse_bytes_gathered'] > 0 and out['gradient_exchange']['dense_bucket_bytes'] > 0
    N.FORCE_COLLECTIVES = False
    dist.barrier()
    dist.destroy_process_group()
    out['ok'] = True
    print('RCCL_WORLD1 ' + json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
