"""CPU tests of the host-side weight packers against the layouts documented in include/k4nerf.h / csrc/*.hip: every packed
buffer must reproduce the original fp32 weights EXACTLY when read back through the documented index maps (the 3-term bf16
split is exact: w == t0 + t1 + t2).  Needs lib4k_hip.so only for its size functions (no GPU)."""
import numpy as np
import pytest
import torch

import nerf4k_amd  # noqa: F401
from nerf4k_amd import _native as N
from nerf4k_amd.lib import dvgo
from nerf4k_amd.lib.sr_esrnet import _Packed


def _row(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def _bf16_terms(raw_f32_view, shape):
    """fp32-typed raw bytes -> bf16 tensor of `shape` -> float"""
    return raw_f32_view.contiguous().view(torch.int16).view(torch.bfloat16).reshape(shape).float()


def test_three_term_split_is_exact_and_ordered():
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(4096, generator=g) * 10.0 ** torch.randint(-6, 6, [4096], generator=g).float(),
                   torch.tensor([0.0, 1.0, -1.0, 1.0e-30, 65504.0, 1e30])])      # (exact down to ~2^-110: bf16 shares fp32's exponent range)
    t0, t1, t2 = dvgo._split3_bf16(x)
    assert torch.equal(t0.float() + t1.float() + t2.float(), x)                       # exact (fp32 adds of nested terms)
    assert bool((t1.float().abs() <= t0.float().abs() * 2 ** -7 + 1e-45).all())
    assert bool((t2.float().abs() <= t0.float().abs() * 2 ** -15 + 1e-45).all())


@pytest.mark.parametrize('dim0,W,depth', [(15, 64, 3), (27, 32, 3), (39, 128, 2), (6, 32, 2)])
def test_pack_mlp_sections_reproduce_the_weights(dim0, W, depth):
    torch.manual_seed(dim0 + W)
    lins = [torch.nn.Linear(dim0, W)] + [torch.nn.Linear(W, W) for _ in range(depth - 2)] + [torch.nn.Linear(W, 3)]
    buf = dvgo.pack_mlp_mfma(lins)
    nh = depth - 2
    assert buf.numel() == N.lib().k4_mlp_packed_floats(dim0, W, nh)
    NB, k1p = W // 32, (dim0 + 2) & ~1
    w1ext = torch.zeros([W, k1p])
    w1ext[:, :dim0] = lins[0].weight.detach()
    w1ext[:, dim0] = lins[0].bias.detach()
    # ---- fp32 section (v_mfma_f32_32x32x2_f32 order)
    off = 0
    w1a = buf[off:off + NB * (k1p // 2) * 64].reshape(NB, k1p // 2, 64); off += w1a.numel()
    for mb in range(NB):
        for kk in range(k1p // 2):
            for l in (0, 17, 31, 32, 63):
                assert w1a[mb, kk, l] == w1ext[mb * 32 + (l & 31), 2 * kk + (l >> 5)]
    if nh:
        w2 = lins[1].weight.detach()
        w2a = buf[off:off + NB * NB * 16 * 64].reshape(NB, NB, 16, 64); off += w2a.numel()
        for mb2 in range(NB):
            for mb in range(NB):
                for r in (0, 5, 15):
                    for l in (0, 31, 32, 63):
                        assert w2a[mb2, mb, r, l] == w2[mb2 * 32 + (l & 31), mb * 32 + _row(r, l >> 5)]
        off += NB * 64
    off += NB * 16 * 2 * 4 + 4
    # ---- split-bf16 sections (v_mfma_f32_32x32x16_bf16 order): first the exact one (K4_MLP_ARITH_B3: 3 terms everywhere, the terms sum back to
    # the weights exactly), then the default one (layer 1 exact, the layer-2 weights' two LEADING terms: within 2^-16 relative)
    KB1, KB2 = (k1p + 15) // 16, W // 16
    wo = lins[-1].weight.detach()
    for nt2 in (3, 2):
        nt1 = 3 if nt2 == 3 else int(N.lib().k4_mlp_b2_layer1_terms())
        n1 = NB * KB1 * nt1 * 64 * 4
        w1s = _bf16_terms(buf[off:off + n1], [NB, KB1, nt1, 64, 8]).sum(2); off += n1
        for mb in range(NB):
            for kb in range(KB1):
                for l in (0, 9, 31, 32, 50, 63):
                    for e in range(8):
                        k = kb * 16 + 8 * (l >> 5) + e
                        want = w1ext[mb * 32 + (l & 31), k] if k < k1p else 0.0
                        assert w1s[mb, kb, l, e] == want if nt1 == 3 else abs(float(w1s[mb, kb, l, e] - want)) <= 2.0 ** -16 * abs(float(want))
        if nh:
            n2 = NB * KB2 * nt2 * 64 * 4
            terms = _bf16_terms(buf[off:off + n2], [NB, KB2, nt2, 64, 8]); off += n2
            w2s = terms.sum(2)
            t0, t1, _ = dvgo._split3_bf16(w2)
            for mb2 in range(NB):
                for kb in range(KB2):
                    for l in (0, 31, 32, 63):
                        for e in range(8):
                            h = l >> 5
                            n = (kb >> 1) * 32 + (e & 3) + 8 * (2 * (kb & 1) + (e >> 2)) + 4 * h
                            j = mb2 * 32 + (l & 31)
                            if nt2 == 3:
                                assert w2s[mb2, kb, l, e] == w2[j, n]
                            else:
                                assert terms[mb2, kb, 0, l, e] == t0[j, n].float() and terms[mb2, kb, 1, l, e] == t1[j, n].float()
                                assert abs(float(w2s[mb2, kb, l, e] - w2[j, n])) <= 2.0 ** -16 * abs(float(w2[j, n]))
            b2s = buf[off:off + NB * 2 * 16].reshape(NB, 2, 16); off += b2s.numel()
            b2 = lins[1].bias.detach()
            for mb2 in range(NB):
                for h in range(2):
                    for r in range(16):
                        assert b2s[mb2, h, r] == b2[mb2 * 32 + _row(r, h)]
        wot = buf[off:off + NB * 16 * 2 * 4].reshape(NB, 16, 2, 4); off += wot.numel()
        for mb in range(NB):
            for r in (0, 7, 15):
                for h in range(2):
                    assert torch.equal(wot[mb, r, h, :3], wo[:, mb * 32 + _row(r, h)]) and wot[mb, r, h, 3] == 0
        assert torch.equal(buf[off:off + 3], lins[-1].bias.detach())
        off += 4
    assert off == buf.numel()


@pytest.mark.parametrize('cout,cin,k', [(32, 160, 3), (64, 192, 3), (64, 3, 3), (3, 64, 3), (128, 64, 1), (64, 1, 3)])
@pytest.mark.parametrize('mode', ['fp32', 'f16x3', 'bf16x6', 'bf16x3'])
def test_conv_packers_reproduce_the_weights(cout, cin, k, mode):
    g = torch.Generator().manual_seed(cout * 7 + cin)
    w = torch.randn([cout, cin, k, k], generator=g)
    b = torch.randn([cout], generator=g)
    pk = _Packed(w, b, mode)
    nt = (cout + 31) // 32
    assert torch.equal(pk.b[:cout], b) and float(pk.b[cout:].abs().sum()) == 0
    if mode == 'bf16x3':
        # the opt-in 2-term arithmetic = the default packing + K4_ARITH_2TERM on plain 3x3 layers (the kernel skips the third term)
        from nerf4k_amd.lib.sr_esrnet import ARITH_2TERM
        ref = _Packed(w, b, 'bf16x6')
        assert pk.mode == 'bf16x6' and torch.equal(pk.w, ref.w)
        assert pk.flags_extra == (ref.flags_extra | (ARITH_2TERM if (k == 3 and cout > 3) else 0))
        return
    if mode == 'f16x3':
        # 2-term fp16 split of w * 2^kexp on plain 3x3 layers with cout > 3 (K4_ARITH_F16X3); every other layer keeps the bf16x6 packing
        from nerf4k_amd.lib.sr_esrnet import ARITH_F16X3
        ref = _Packed(w, b, 'bf16x6')
        assert pk.mode == 'bf16x6'
        if not (k == 3 and cout > 3):
            assert torch.equal(pk.w, ref.w) and pk.flags_extra == ref.flags_extra
            return
        assert pk.flags_extra == ARITH_F16X3
        nch = (cin + 15) // 16
        nb = (nch + 3) // 4 * 4
        tail = pk.w[-(nt * 32 * 2 + nb * 2):]
        unscale = tail[:nt * 32 * 2].view(torch.float32).double()                           # 2^-a[co]
        bq = tail[nt * 32 * 2:].view(torch.int32)[:nch]
        assert bool((unscale > 0).all()) and bool((torch.log2(unscale) == torch.log2(unscale).round()).all()) and int(bq.min()) >= 0
        terms = pk.w[:-(nt * 32 * 2 + nb * 2)].view(torch.float16).reshape(nch, 2, k * k, 2, nt * 32, 8)
        assert torch.isfinite(terms.float()).all()
        assert 2.0 ** 13 <= float(terms[:, 0].float().abs().max()) <= 2.0 ** 14            # operands fill fp16's range, never overflow it
        got = terms.double().sum(1).permute(1, 0, 2, 4, 3).reshape(k * k, nch * 16, nt * 32)
        got = got * unscale.view(1, 1, -1) * torch.ldexp(torch.ones(nch, dtype=torch.float64), -bq).repeat_interleave(16).view(1, -1, 1)
        want = torch.zeros_like(got)
        want[:, :cin, :cout] = w.double().permute(2, 3, 1, 0).reshape(k * k, cin, cout)
        # 22 significant bits of the largest weight of each (output channel, chunk) block
        blk = want.reshape(k * k, nch, 16, nt * 32).abs().amax((0, 2), keepdim=True).expand(k * k, nch, 16, nt * 32).reshape(want.shape)
        assert bool(((got - want).abs() <= 2.0 ** -21 * blk).all())
        return
    if mode == 'bf16x6' and k == 3 and cout <= 3:
        # few output channels: taps become the N dimension of a 1x1 layer, n = tap*cout + co (K4_W_TAPS_AS_COUT)
        from nerf4k_amd.lib.sr_esrnet import W_TAPS_AS_COUT
        assert pk.flags_extra == W_TAPS_AS_COUT
        nch = (cin + 15) // 16
        terms = pk.w.view(torch.bfloat16).reshape(nch, 3, 1, 2, 32, 8).float()
        got1 = terms.sum(1).permute(1, 0, 2, 4, 3).reshape(nch * 16, 32)                    # [channel][n]
        want1 = torch.zeros_like(got1)
        want1[:cin, :9 * cout] = w.permute(1, 2, 3, 0).reshape(cin, 9 * cout)               # n = (dy*3+dx)*cout + co
        assert torch.equal(got1, want1)
        return
    assert pk.flags_extra == 0
    if mode == 'fp32':
        nch = (cin + 7) // 8
        got = pk.w.reshape(nch, k * k, 8, nt * 32).permute(1, 0, 2, 3).reshape(k * k, nch * 8, nt * 32)
    else:
        assert pk.mode == 'bf16x6'
        nch = (cin + 15) // 16
        terms = pk.w.view(torch.bfloat16).reshape(nch, 3, k * k, 2, nt * 32, 8).float()
        got = terms.sum(1).permute(1, 0, 2, 4, 3).reshape(k * k, nch * 16, nt * 32)         # [tap][channel][cout]
    want = torch.zeros_like(got)
    want[:, :cin, :cout] = w.permute(2, 3, 1, 0).reshape(k * k, cin, cout)
    assert torch.equal(got, want)


def test_p16_format_roundtrip_and_weight_packing():
    """The pre-split activation format (tests/helpers.py restates include/k4nerf.h) keeps 22 bits inside its window, and the p16 weight
    operand decodes to w 2^a[co] 2^-E[chunk] with every output channel's largest magnitude in [2^13, 2^14)."""
    import helpers
    from nerf4k_amd.lib.sr_esrnet import _PackedP16
    g = torch.Generator().manual_seed(3)
    x = torch.randn([5, 7, 48], generator=g) * torch.logspace(-3, 1, 48)
    for E in (-3, 0, 9):
        p = helpers.to_p16(x, E)
        assert p.dtype == torch.int32 and p.shape == x.shape
        back = helpers.from_p16(p, E)
        big = (x.abs() * 2.0 ** E) >= 2.0 ** -3
        rel = ((back - x.double()).abs() / x.abs().double().clamp_min(1e-30))[big]
        assert float(rel.max()) <= 2.0 ** -21.5, (E, float(rel.max()))
        assert float((back - x.double()).abs()[~big].max()) <= 2.0 ** -25 * 2.0 ** -E * 1.0001 if (~big).any() else True
    cout, cin = 64, 96
    w = torch.randn([cout, cin, 3, 3], generator=g) * 0.05
    b = torch.randn([cout], generator=g)
    e_chunks = [3, 3, 3, 3, -2, 7]
    pk = _PackedP16(w, b, e_chunks)
    nbytes = (cin // 16) * (cout // 32) * 18432
    raw = pk.w[: nbytes // 2].view(torch.float16).reshape(cin // 16, cout // 32, 2, 9, 2, 32, 8).double()        # [chunk][nb][term][tap][kg][co][j]
    unscale = pk.w[nbytes // 2:].view(torch.float32).double()
    dec = (raw[:, :, 0] + raw[:, :, 1]).permute(1, 4, 0, 3, 5, 2).reshape(cout, cin, 9)                            # [nb][co][chunk][kg][j][tap]
    e_in = torch.tensor(e_chunks).repeat_interleave(16).double()
    want = w.double().reshape(cout, cin, 9) * (2.0 ** -e_in).view(1, -1, 1) / unscale.view(-1, 1, 1)
    assert float(((dec - want).abs() / want.abs().amax((1, 2), keepdim=True)).max()) <= 2.0 ** -22
    top = (raw[:, :, 0].abs().permute(1, 4, 0, 3, 5, 2).reshape(cout, -1)).amax(1)
    assert bool(((top >= 2.0 ** 13) & (top < 2.0 ** 14)).all())
    assert torch.equal(pk.b, b)


def test_p16_up2x_weight_packing_is_the_phase_sum_of_the_taps():
    """K4_PRE_UPSAMPLE2X operand: decoding it and applying the four 2 x 2 phase filters to an image equals the 3 x 3 layer on the nearest-x2
    upsampled image (CPU, fp64), i.e. the tap sums and the (phase, tap) -> input offset map are right."""
    import torch.nn.functional as F
    from nerf4k_amd.lib.sr_esrnet import _PackedP16Up
    g = torch.Generator().manual_seed(8)
    cout, cin = 32, 32
    w = torch.randn([cout, cin, 3, 3], generator=g) * 0.1
    b = torch.zeros([cout])
    pk = _PackedP16Up(w, b, [0, 2])
    nbytes = (cin // 16) * (cout // 32) * 4 * 8192
    raw = pk.w[: nbytes // 2].view(torch.float16).reshape(cin // 16, cout // 32, 4, 2, 4, 2, 32, 8).double()       # [chunk][nb][phase][term][tap][kg][co][j]
    unscale = pk.w[nbytes // 2:].view(torch.float32).double()
    dec = (raw[:, :, :, 0] + raw[:, :, :, 1]).permute(1, 5, 0, 4, 6, 2, 3).reshape(cout, cin, 4, 4)                  # [nb][co][chunk][kg][j][phase][tap]
    e_in = torch.tensor([0, 2]).repeat_interleave(16).double()
    wp = dec * unscale.view(-1, 1, 1, 1) * (2.0 ** e_in).view(1, -1, 1, 1)                                            # phase filters on the raw input
    x = torch.randn([1, cin, 7, 9], generator=g).double()
    want = F.conv2d(F.interpolate(x, scale_factor=2, mode='nearest'), w.double(), padding=1)
    xp = F.pad(x, (1, 1, 1, 1))
    got = torch.zeros_like(want)
    for py in (0, 1):
        for px in (0, 1):
            k2 = wp[:, :, py * 2 + px].reshape(cout, cin, 2, 2)
            got[:, :, py::2, px::2] = F.conv2d(xp[:, :, py:py + 8, px:px + 10], k2)                                    # input pixel (Y - 1 + py + a, X - 1 + px + b)
    assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())


@pytest.mark.parametrize('C', [32, 64])
def test_sft_epilogue_operand_evaluates_the_sft_layer(C):
    """k4_conv3x3_p16_sft_multi's w_sfe (include/k4nerf.h): walking the operand the way the kernel's epilogue does -- GEMM 1 on the scaled
    condition, tables, LeakyReLU, GEMM 2 with k in accumulator-register order, tables -- gives scale / shift of the SFTLayer (fp64, CPU);
    every fp16 operand is finite and the hidden activations stay below 2^P16_TARGET_EXP over the calibrated condition range."""
    from nerf4k_amd.lib.sr_esrnet import _PackedSfe, SFTLayer, P16_TARGET_EXP
    torch.manual_seed(40 + C)
    layer = SFTLayer(C, 32)
    with torch.no_grad():
        for m_ in layer.modules():
            if isinstance(m_, torch.nn.Conv2d):
                m_.weight.mul_(3.0)
                m_.bias.normal_(0, 0.5)
    e_cond = 7
    sfe = _PackedSfe(layer, e_cond)
    assert sfe.w.numel() * 2 == (C // 32) * 17408
    cmax = 2.0 ** (P16_TARGET_EXP + 1 - e_cond)
    cond = (torch.rand([50, 32], dtype=torch.float64) * 2 - 1) * cmax
    with torch.no_grad():
        c4 = cond.float().t().reshape(1, 32, 1, 50)
        want_scale = layer.SFT_scale_conv1(torch.nn.functional.leaky_relu(layer.SFT_scale_conv0(c4), 0.2))[0, :, 0].double()       # [C][50]
        want_shift = layer.SFT_shift_conv1(torch.nn.functional.leaky_relu(layer.SFT_shift_conv0(c4), 0.2))[0, :, 0].double()
    lane = torch.arange(64)
    l31, half = lane & 31, lane >> 5
    for nb in range(C // 32):
        blk = sfe.w[nb * 8704:(nb + 1) * 8704]
        assert bool(torch.isfinite(blk[:8192].view(torch.float16).float()).all())
        A1 = blk[:4096].view(torch.float16).double().reshape(2, 2, 2, 64, 8).sum(2)              # [path][kb][lane][e]  (hi + lo)
        A2 = blk[4096:8192].view(torch.float16).double().reshape(2, 2, 2, 64, 8).sum(2)
        T = blk[8192:].view(torch.float32).double().reshape(4, 2, 2, 16)                          # [us1|b1|us2|b2][path][half][16]
        B = cond * 2.0 ** e_cond                                                                  # [pix][k]
        got = []
        for pth in (0, 1):
            acc = torch.zeros([50, 32], dtype=torch.float64)                                      # [pix][row m]
            for kb in range(2):
                for e in range(8):
                    for h in (0, 1):                                                              # lane (h, m): A1 element e multiplies condition channel 16 kb + 8 h + e
                        sel = half == h
                        acc[:, l31[sel]] += A1[pth, kb, sel, e][None, :] * B[:, (16 * kb + 8 * h + e)][:, None]
            # registers of half h: entry i <-> row (i & 3) + 8 (i >> 2) + 4 h
            hid = torch.zeros([50, 2, 16], dtype=torch.float64)
            for h in (0, 1):
                for i in range(16):
                    rowi = (i & 3) + 8 * (i >> 2) + 4 * h
                    t = acc[:, rowi] * T[0, pth, h, i] + T[1, pth, h, i]
                    hid[:, h, i] = torch.where(t > 0, t, t * 0.2)
            assert float(hid.abs().max()) <= 2.0 ** P16_TARGET_EXP
            out = torch.zeros([50, 32], dtype=torch.float64)                                      # [pix][c local]
            for kb in range(2):
                for e in range(8):
                    for h in (0, 1):
                        sel = half == h
                        out[:, l31[sel]] += A2[pth, kb, sel, e][None, :] * hid[:, h, 8 * kb + e][:, None]
            res = torch.zeros([50, 32], dtype=torch.float64)
            for h in (0, 1):
                for i in range(16):
                    rowi = (i & 3) + 8 * (i >> 2) + 4 * h
                    res[:, rowi] = out[:, rowi] * T[2, pth, h, i] + T[3, pth, h, i]
            got.append(res.t())                                                                   # [32][pix]
        for g_, w_ in zip(got, (want_scale, want_shift)):
            w_nb = w_[nb * 32:(nb + 1) * 32]
            assert float((g_ - w_nb).abs().max()) <= 3e-6 * max(1.0, float(w_nb.abs().max())), (C, nb, float((g_ - w_nb).abs().max()))
