"""CPU restatement of the marcher's training-graph operators and of the joint step's loss terms.  TEST INFRASTRUCTURE ONLY.

``rgbnet_sigmoid``      torch.sigmoid(nn.Sequential(Linear, ReLU, [Linear, ReLU], Linear)(x) [+ add]) in fp64 with autograd --
                        what /root/reference/lib/dmpigo.py:375-379 and lib/dvgo.py:407-412 evaluate (pinned through the
                        reference-module gradient goldens grad_mpi / grad_dvgo / grad_joint, which run the reference's own
                        nn.Sequential).
``distortion_loss``     the distortion loss of run_sr.py:976-988.  The reference calls the third-party package
                        ``torch_efficient_distloss`` (sunset1995, PyPI 0.1.3; neither vendored in /root/reference nor installed
                        here): PARITY UNPINNED against that package's code.  Restated from its published definition
                        (Mip-NeRF 360 eq. 15, flattened form of DVGOv2):
                            L = 1/R * sum_rays [ sum_i sum_j w_i w_j |s_i - s_j| + 1/3 sum_i w_i^2 * interval ],  R = ray_id.max() + 1
                        evaluated literally (O(n^2) per ray) in fp64 with autograd -- independent of the prefix-sum form the HIP
                        kernel uses.
``joint_losses``        the loss terms of one joint iteration (run_sr.py:877-995) for fern_lg_joint_l1 (dim_rend=3, num_cond=1).
"""
import torch
import torch.nn.functional as F


def rgbnet_sigmoid(x, weights, add=None):
    """weights = [(W1, b1), (W2, b2)?, (W3, b3)] as nn.Linear stores them."""
    h = x
    for i, (w, b) in enumerate(weights):
        h = F.linear(h, w, b)
        if i + 1 < len(weights):
            h = torch.relu(h)
    return torch.sigmoid(h if add is None else h + add)


def distortion_loss(w, s, interval, ray_id):
    n_rays = int(ray_id.max()) + 1
    total = w.new_zeros([])
    for r in torch.unique(ray_id).tolist():
        m = ray_id == r
        wr, sr = w[m], s[m]
        total = total + (wr[:, None] * wr[None, :] * (sr[:, None] - sr[None, :]).abs()).sum() + (wr * wr).sum() * interval / 3
    return total / n_rays


def joint_losses(rr, rgb_sr, target, target_4x, pr, pc, cfg, sr_ratio=4):
    """rr: the marcher's training dict; cfg: mapping with the weight_* entries.  Returns (total, dict of terms)."""
    n_rays = target.shape[0]
    terms = {'photo': cfg['weight_main'] * F.l1_loss(rr['rgb_feature'], target)}                                  # run_sr.py:877-881
    rgb_hr = target_4x.detach().reshape(sr_ratio * pr, sr_ratio * pc, 3).movedim(-1, 0).unsqueeze(0)
    terms['l1'] = F.l1_loss(rgb_sr, rgb_hr)                                                                        # :925
    if cfg['weight_entropy_last'] > 0:                                                                             # :962-964
        p = rr['alphainv_last'].clamp(1e-6, 1 - 1e-6)
        terms['entropy_last'] = -(p * torch.log(p) + (1 - p) * torch.log(1 - p)).mean() * cfg['weight_entropy_last']
    if cfg['weight_distortion'] > 0:                                                                               # :976-988
        terms['distortion'] = cfg['weight_distortion'] * distortion_loss(rr['weights'], rr['s'], 1 / rr['n_max'], rr['ray_id'])
    if cfg['weight_rgbper'] > 0:                                                                                   # :993-995
        per = (rr['raw_rgb'] - target[rr['ray_id']]).pow(2).sum(-1)
        terms['rgbper'] = cfg['weight_rgbper'] * (per * rr['weights'].detach()).sum() / n_rays
    return sum(terms.values()), terms
