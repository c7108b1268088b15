"""Checkpoint helpers with the reference's formats (/root/reference/lib/utils.py:50-66).

DVGO/MPI checkpoints: ``{'global_step', 'model_kwargs', 'model_state_dict', 'optimizer_state_dict'}``
(run_sr.py:1173-1178).  Only what the render / training-step path needs is restated; metrics (SSIM/LPIPS) are out
of scope.
"""
import numpy as np
import torch
import torch.nn as nn

from .masked_adam import MaskedAdam

to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)     # lib/utils.py:19
mse2psnr = lambda x: -10. * torch.log10(x)


def to8b_device(x):
    """to8b for a device tensor: uint8 tensor of the same shape, packed on the GPU (k4_to8b) -- a 4K frame leaves the device
    as 36.6 MB instead of 146 MB of fp32."""
    from .. import _native as N
    xc = x.detach().float().contiguous()
    out = torch.empty(xc.shape, dtype=torch.uint8, device=xc.device)
    N.check(N.lib().k4_to8b(N.f32(xc), xc.numel(), N.ptr(out), N.stream()), 'k4_to8b')
    return out


def window_to_planes(hr, oy, ox, th, tw, dst):
    """dst[c, y, x] = hr[0, c, oy + y, ox + x] for a decoded window `hr` ([1, C, H, W] VIEW of the decoder's NHWC result) and a destination `dst` [C, th, tw]
    whose rows are contiguous (a tile of the frame, a slice of a gather buffer reshaped): one pass of k4_nhwc_window_to_planar instead of a channel-stride
    gather by slice assignment.  Falls back to the slice assignment for any other layout (CPU tensors of the gloo tests, other dtypes)."""
    C, H, W = int(hr.shape[1]), int(hr.shape[2]), int(hr.shape[3])
    ok = (hr.is_cuda and dst.is_cuda and hr.dtype == torch.float32 and dst.dtype == torch.float32 and C <= 4 and hr.shape[0] == 1
          and hr.stride(1) == 1 and hr.stride(3) == C and hr.stride(2) == W * C and dst.dim() == 3 and tuple(dst.shape) == (C, th, tw)
          and dst.stride(2) == 1 and th > 0 and tw > 0)
    if not ok:
        dst.copy_(hr[0, :, oy:oy + th, ox:ox + tw])
        return
    from .. import _native as N
    N.check(N.lib().k4_nhwc_window_to_planar(N.C.c_void_p(hr.data_ptr()), W, C, int(oy), int(ox), int(th), int(tw), N.C.c_void_p(dst.data_ptr()),
                                             int(dst.stride(0)), int(dst.stride(1)), N.stream()), 'k4_nhwc_window_to_planar')


def create_optimizer_or_freeze_model(model, cfg_train, global_step):
    """lib/utils.py:21-48: one param group per `lrate_<name>` entry of cfg_train whose attribute exists on the model;
    lr decays by 0.1 every `lrate_decay` k-steps; lr == 0 freezes the parameter.  cfg_train: attribute-style mapping
    (mmcv ConfigDict upstream; anything with .keys() and getattr works)."""
    decay = 0.1 ** (global_step / (cfg_train.lrate_decay * 1000))
    names = [key[len('lrate_'):] for key in cfg_train.keys() if key.startswith('lrate_')]
    param_group = []
    for name in names:
        target = getattr(model, name, None)
        if target is None:
            if hasattr(model, name):
                print(f'create_optimizer_or_freeze_model: param {name} not exist')
            continue
        lr = getattr(cfg_train, 'lrate_' + name) * decay
        if lr <= 0:
            print(f'create_optimizer_or_freeze_model: param {name} freeze')
            target.requires_grad = False        # (a plain attribute on an nn.Module, exactly as upstream)
            continue
        print(f'create_optimizer_or_freeze_model: param {name} lr {lr}')
        param_group.append({'params': target.parameters() if isinstance(target, nn.Module) else target,
                            'lr': lr, 'kname': name,
                            'skip_zero_grad': name in cfg_train.skip_zero_grad_fields})
    return MaskedAdam(param_group)


def load_checkpoint(model, optimizer, ckpt_path, no_reload_optimizer):
    ckpt = torch.load(ckpt_path, map_location='cpu', weights_only=False)
    start = ckpt['global_step']
    model.load_state_dict(ckpt['model_state_dict'])
    if not no_reload_optimizer:
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
    return model, optimizer, start


def load_model(model_class, ckpt_path):
    ckpt = torch.load(ckpt_path, map_location='cpu', weights_only=False)
    model = model_class(**ckpt['model_kwargs'])
    model.load_state_dict(ckpt['model_state_dict'])
    return model


def model_from_checkpoint_dict(ckpt):
    """Same as load_model for an in-memory checkpoint dict (synthetic scenes)."""
    from . import dvgo, dmpigo
    cls = {'DirectMPIGO': dmpigo.DirectMPIGO, 'DirectVoxGO': dvgo.DirectVoxGO}[ckpt['model_class']]
    model = cls(**ckpt['model_kwargs'])
    model.load_state_dict(ckpt['model_state_dict'])
    return model
