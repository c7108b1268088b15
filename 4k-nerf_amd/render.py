"""Render loop with the reference's ``render_viewpoints`` interface (run_sr.py:75-182, run.py:67-171).

``render_viewpoints`` keeps the reference signature and return tuple
``(rgbs, depths, bgmaps, psnrs, viewdirs_all, rgb_features)`` (numpy, host) so ``run_sr.py``'s SR loop
(run_sr.py:1353-1395) consumes it unchanged.  The reference reads ``cfg.data.flip_x/flip_y`` from a
module global (run_sr.py:106); here they come from ``render_kwargs`` (the same dict carries them,
run_sr.py:1318-1319).  What changed inside: no 8192-ray chunk loop (the fused marcher has no
per-sample intermediates), rays are generated on the device, ONE launch per frame.

``render_frame`` is the device-resident fast path used by bench.py and the multi-GPU tile renderer.
"""
import numpy as np
import torch

from .lib import dvgo


@torch.no_grad()
def render_frame(model, H, W, K, c2w, ndc, render_kwargs, rays=None):
    """One full frame -> dict of DEVICE tensors: rgb_marched/rgb_feature [H,W,3], depth [H,W], alphainv_last [H,W]."""
    dev = next(model.parameters()).device
    if rays is None:
        c2w = torch.as_tensor(np.asarray(c2w), dtype=torch.float32).to(dev)
        rays = dvgo.get_rays_of_a_view(H, W, K, c2w, ndc, inverse_y=render_kwargs.get('inverse_y', False),
                                       flip_x=render_kwargs.get('flip_x', False),
                                       flip_y=render_kwargs.get('flip_y', False))
    ro, rd, vd = [r.reshape(-1, 3) for r in rays]
    kw = dict(render_kwargs)
    kw.setdefault('render_depth', True)
    out = model(ro, rd, vd, k4_img_w=W, **kw)
    res = {k: out[k].reshape(H, W, -1) for k in ('rgb_marched', 'rgb_feature')}
    res['depth'] = out['depth'].reshape(H, W)
    res['alphainv_last'] = out['alphainv_last'].reshape(H, W)
    return res


@torch.no_grad()
def render_viewpoints(model, render_poses, HW, Ks, ndc, render_kwargs,
                      gt_imgs=None, savedir=None, dump_images=False,
                      render_factor=0, render_video_flipy=False, render_video_rot90=0,
                      eval_ssim=False, eval_lpips_alex=False, eval_lpips_vgg=False, global_step=0,
                      arr_index=None, img_enc=None):
    '''Render images for the given viewpoints; run evaluation if gt given.'''
    assert len(render_poses) == len(HW) and len(HW) == len(Ks)
    if eval_ssim or eval_lpips_alex or eval_lpips_vgg:
        raise NotImplementedError('SSIM/LPIPS evaluation is outside the hot-path scope (SURVEY.md 2.1 #17)')
    if render_factor != 0:
        HW = np.copy(HW)
        Ks = np.copy(Ks)
        HW = (HW / render_factor).astype(int)
        Ks[:, :2, :3] /= render_factor
    rgbs, rgb_features, depths, bgmaps, psnrs, viewdirs_all = [], [], [], [], [], []
    for i, c2w in enumerate(render_poses):
        H, W = int(HW[i][0]), int(HW[i][1])
        K = Ks[i]
        dev = next(model.parameters()).device
        c2w_t = torch.as_tensor(np.asarray(c2w), dtype=torch.float32).to(dev)
        rays = dvgo.get_rays_of_a_view(H, W, K, c2w_t, ndc, inverse_y=render_kwargs.get('inverse_y', False),
                                       flip_x=render_kwargs.get('flip_x', False),
                                       flip_y=render_kwargs.get('flip_y', False))
        res = render_frame(model, H, W, K, c2w_t, ndc, render_kwargs, rays=rays)
        rgb = res['rgb_marched'].clamp(0, 1).cpu().numpy()
        rgbs.append(rgb)
        rgb_features.append(res['rgb_feature'].cpu().numpy())          # UNclamped, as run_sr.py:131
        depths.append(res['depth'].unsqueeze(-1).cpu().numpy())
        bgmaps.append(res['alphainv_last'].unsqueeze(-1).cpu().numpy())
        viewdirs_all.append(rays[2].flatten(0, -2))
        if gt_imgs is not None and render_factor == 0:
            psnrs.append(-10. * np.log10(np.mean(np.square(rgb - gt_imgs[i]))))
    if render_video_flipy:
        for i in range(len(rgbs)):
            rgbs[i], depths[i], bgmaps[i] = np.flip(rgbs[i], 0), np.flip(depths[i], 0), np.flip(bgmaps[i], 0)
    if render_video_rot90 != 0:
        for i in range(len(rgbs)):
            rgbs[i] = np.rot90(rgbs[i], k=render_video_rot90, axes=(0, 1))
            depths[i] = np.rot90(depths[i], k=render_video_rot90, axes=(0, 1))
            bgmaps[i] = np.rot90(bgmaps[i], k=render_video_rot90, axes=(0, 1))
    if savedir is not None and dump_images:
        raise NotImplementedError('PNG dumping needs imageio (not installed); save the returned arrays instead')
    return (np.array(rgbs), np.array(depths), np.array(bgmaps), psnrs, viewdirs_all, np.array(rgb_features))
