"""Checkpoint helpers with the reference's formats (/root/reference/lib/utils.py:50-66).

DVGO/MPI checkpoints: ``{'global_step', 'model_kwargs', 'model_state_dict', 'optimizer_state_dict'}``
(run_sr.py:1173-1178).  Only what the render path needs is restated; metrics (SSIM/LPIPS) and the
optimiser factory are out of scope.
"""
import numpy as np
import torch

to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)     # lib/utils.py:19
mse2psnr = lambda x: -10. * torch.log10(x)


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
