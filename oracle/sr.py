"""CPU oracle of the VC-Decoder (SFTNet: RRDB + SFT x4 super-resolution).  TEST INFRASTRUCTURE ONLY.

Functional fp32 restatement (``F.conv2d`` on a plain ``state_dict``) of
  * ``SFTLayer.forward``                 -- /root/reference/lib/sr_esrnet.py:112-123
  * ``ResidualDenseBlock_SFT.forward``   -- lib/sr_esrnet.py:149-158
  * ``RRDB_SFT.forward``                 -- lib/sr_esrnet.py:176-182
  * ``SFTNet.forward``                   -- lib/sr_esrnet.py:446-465
  * ``SFTNet.tile_process``              -- lib/sr_esrnet.py:467-527
with the reference's state_dict key names (458 tensors at the default configuration).

Pin: ``tests/golden/sr_*.npz`` hold outputs of the UNMODIFIED reference module (importable in
the build container, pure torch) on weights from ``make_state_dict`` -- generator
``oracle/gen_golden.py``; ``tests/test_oracle_golden.py`` checks this file against them.
"""
import math

import torch
import torch.nn.functional as F


def _conv(sd, name, x, pad):
    return F.conv2d(x, sd[name + '.weight'], sd[name + '.bias'], stride=1, padding=pad)


def _lrelu(x):
    return F.leaky_relu(x, 0.2)


def sft_layer(sd, p, x, cond):
    """x*(scale+1)+shift with scale/shift = 1x1 conv -> lrelu(0.2) -> 1x1 conv of cond (:120-123)"""
    scale = _conv(sd, p + '.SFT_scale_conv1', _lrelu(_conv(sd, p + '.SFT_scale_conv0', cond, 0)), 0)
    shift = _conv(sd, p + '.SFT_shift_conv1', _lrelu(_conv(sd, p + '.SFT_shift_conv0', cond, 0)), 0)
    return x * (scale + 1) + shift


def rdb_sft(sd, p, x, cond):
    """Dense block with SFT at the entry and on x4; x5*0.2 + x (:149-158)"""
    xc0 = sft_layer(sd, p + '.sft0', x, cond)
    x1 = _lrelu(_conv(sd, p + '.conv1', xc0, 1))
    x2 = _lrelu(_conv(sd, p + '.conv2', torch.cat((xc0, x1), 1), 1))
    x3 = _lrelu(_conv(sd, p + '.conv3', torch.cat((xc0, x1, x2), 1), 1))
    x4 = _lrelu(_conv(sd, p + '.conv4', torch.cat((xc0, x1, x2, x3), 1), 1))
    xc1 = sft_layer(sd, p + '.sft1', x4, cond)
    x5 = _conv(sd, p + '.conv5', torch.cat((xc0, x1, x2, x3, xc1), 1), 1)
    return x5 * 0.2 + x


def rrdb_sft(sd, p, x, cond):
    """3 RDBs, SFT, *0.2 + x (:176-182)"""
    out = rdb_sft(sd, p + '.rdb1', x, cond)
    out = rdb_sft(sd, p + '.rdb2', out, cond)
    out = rdb_sft(sd, p + '.rdb3', out, cond)
    out = sft_layer(sd, p + '.sft0', out, cond)
    return out * 0.2 + x


def num_blocks(sd):
    return 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('body.'))


def sftnet_forward(sd, x, cond, scale=4):
    """SFTNet.forward with fea=None (:446-465)."""
    feat = _conv(sd, 'conv_first', x, 1)
    c = _lrelu(_conv(sd, 'CondNet.0', cond, 1))
    c = _lrelu(_conv(sd, 'CondNet.2', c, 0))
    c = _lrelu(_conv(sd, 'CondNet.4', c, 0))
    c = _conv(sd, 'CondNet.6', c, 0)
    body = feat
    for b in range(num_blocks(sd)):
        body = rrdb_sft(sd, f'body.{b}', body, c)
    body = sft_layer(sd, 'sftbody', body, c)
    body = _conv(sd, 'conv_body', body, 1)
    body = body + feat
    if scale > 1:
        body = _lrelu(_conv(sd, 'conv_up1', F.interpolate(body, scale_factor=2, mode='nearest'), 1))
        if scale == 4:
            body = _lrelu(_conv(sd, 'conv_up2', F.interpolate(body, scale_factor=2, mode='nearest'), 1))
    return _conv(sd, 'conv_last', _lrelu(_conv(sd, 'conv_hr', body, 1)), 1)


def tile_geometry(height, width, tile_size, tile_pad=10):
    """The reference's tile loop as data (:478-523): list of
    (y0, y1, x0, x1, yp0, yp1, xp0, xp1) = unpadded tile and its clipped padded window."""
    tiles = []
    for y in range(math.ceil(height / tile_size)):
        for x in range(math.ceil(width / tile_size)):
            x0, y0 = x * tile_size, y * tile_size
            x1, y1 = min(x0 + tile_size, width), min(y0 + tile_size, height)
            tiles.append((y0, y1, x0, x1, max(y0 - tile_pad, 0), min(y1 + tile_pad, height),
                          max(x0 - tile_pad, 0), min(x1 + tile_pad, width)))
    return tiles


def tile_process(sd, img, cond, tile_size, tile_pad=10, scale=4):
    """SFTNet.tile_process (:467-527): img [1,3,H,W], cond [1,H,W] (unsqueezed to [1,1,H,W] :474)."""
    _, ch, height, width = img.shape
    cond = cond.unsqueeze(0)
    out = img.new_zeros((1, ch, height * scale, width * scale))
    for (y0, y1, x0, x1, yp0, yp1, xp0, xp1) in tile_geometry(height, width, tile_size, tile_pad):
        o = sftnet_forward(sd, img[:, :, yp0:yp1, xp0:xp1], cond[:, :, yp0:yp1, xp0:xp1], scale)
        oy, ox = (y0 - yp0) * scale, (x0 - xp0) * scale
        out[:, :, y0 * scale:y1 * scale, x0 * scale:x1 * scale] = \
            o[:, :, oy:oy + (y1 - y0) * scale, ox:ox + (x1 - x0) * scale]
    return out


# ---------------------------------------------------------------------------
# deterministic weights (independent of nn.Module construction order)
# ---------------------------------------------------------------------------
def state_dict_spec(n_in_colors=3, scale=4, num_feat=64, num_block=5, num_grow_ch=32, num_cond=1):
    """(key, shape) of every SFTNet tensor, in the reference module's registration order
    (lib/sr_esrnet.py:411-444)."""
    spec = []

    def conv(name, cout, cin, k):
        spec.append((name + '.weight', (cout, cin, k, k)))
        spec.append((name + '.bias', (cout,)))

    def sft(name, nf):
        conv(name + '.SFT_scale_conv0', num_grow_ch, num_grow_ch, 1)
        conv(name + '.SFT_scale_conv1', nf, num_grow_ch, 1)
        conv(name + '.SFT_shift_conv0', num_grow_ch, num_grow_ch, 1)
        conv(name + '.SFT_shift_conv1', nf, num_grow_ch, 1)

    conv('conv_first', num_feat, n_in_colors, 3)
    for b in range(num_block):
        for r in (1, 2, 3):
            p = f'body.{b}.rdb{r}'
            for i in range(4):
                conv(f'{p}.conv{i + 1}', num_grow_ch, num_feat + i * num_grow_ch, 3)
            conv(f'{p}.conv5', num_feat, num_feat + 4 * num_grow_ch, 3)
            sft(p + '.sft0', num_feat)
            sft(p + '.sft1', num_grow_ch)
        sft(f'body.{b}.sft0', num_feat)
    conv('conv_body', num_feat, num_feat, 3)
    if scale > 1:
        conv('conv_up1', num_feat, num_feat, 3)
        if scale == 4:
            conv('conv_up2', num_feat, num_feat, 3)
    conv('conv_hr', num_feat, num_feat, 3)
    conv('conv_last', 3, num_feat, 3)
    sft('sftbody', num_feat)
    conv('CondNet.0', 64, num_cond, 3)
    conv('CondNet.2', 64, 64, 1)
    conv('CondNet.4', 64, 64, 1)
    conv('CondNet.6', 32, 64, 1)
    return spec


def make_state_dict(seed=777, **cfg):
    """Seeded weights: N(0, (gain/sqrt(fan_in))^2), small biases.  Dense-block convs get the
    reference's 0.1 init scale (lib/sr_esrnet.py:147) so residual branches stay small."""
    g = torch.Generator(device='cpu')
    g.manual_seed(int(seed))
    sd = {}
    for key, shape in state_dict_spec(**cfg):
        if key.endswith('.weight'):
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 0.1 * math.sqrt(2.0) if '.rdb' in key and '.conv' in key else 1.0
            sd[key] = torch.randn(shape, generator=g) * (gain / math.sqrt(fan_in))
        else:
            sd[key] = torch.randn(shape, generator=g) * 0.02
    return sd
