"""SFTNet's training pass -- forward AND backward -- as two launch tapes (include/k4nerf.h k4_tape_*).

The reference back-propagates its losses through ``SFTNet`` as one autograd graph over cuDNN calls (/root/reference/run_sr.py:869-1014,
modules lib/sr_esrnet.py:112-182, forward :446-465).  ``lib/sr_train.forward_train`` evaluates the same graph as ~20 autograd nodes per
RRDB whose kernels are already native -- but a 64x64 patch makes every launch 4-30 us, so the iteration was paced by the host: one
Python-to-C transition per call, ``torch.empty`` per buffer, an autograd node per fused block (10.5 ms of host time against 9.3 ms of
kernels on the main stream).  Here the whole decoder is ONE autograd node (``K4DecoderTape``):

  * a ``DecoderProgram`` owns, per patch shape, every activation / gradient buffer of the pass and ONE flat buffer of parameter gradients;
  * its forward and backward are written out call by call below (the same entry points, in the order the autograd engine ran them);
  * the first pass through each RECORDS the calls on a tape, every later pass replays the tape with one native call.

Values: the forward output and the input gradients are bit-identical to ``forward_train``'s (same kernels, same order); the weight gradients
of the 3x3 / 1x1 layers are split-K sums of fp32 atomics in both forms (equal up to the order of those additions).  The weight gradients of
the layers outside the dense blocks move to the side stream the dense blocks' own weight gradients already use.

A program is leased from forward until its backward has run (or the graph is dropped): a second forward in between takes another program
of the pool; past the pool ``forward_train`` keeps the per-block path.  There is no CPU path.
"""
import torch

from .. import _native as N
from .sr_esrnet import EPI_LRELU, EPI_RES, CONV_SMALL
from . import sr_train as T

POOL = 3            # programs per (network, shape): forwards in flight before their backward


class _Tape:
    """RAII handle of a k4_tape."""

    def __init__(self):
        self.h = None

    def record(self, fn):
        L = N.lib()
        h = L.k4_tape_begin(N.stream())
        if not h:
            raise N.K4Error('k4_tape_begin: this thread is already recording a tape')
        ok = False
        try:
            fn()
            ok = True
        finally:
            L.k4_tape_end(N.C.c_void_p(h))
            if not ok:
                L.k4_tape_free(N.C.c_void_p(h))
        self.h = N.C.c_void_p(h)
        return self

    def __len__(self):
        return 0 if self.h is None else int(N.lib().k4_tape_length(self.h))

    def replay(self):
        N.check(N.lib().k4_tape_replay(self.h, N.stream()), 'k4_tape_replay')

    def __del__(self):
        try:                                                # (at interpreter shutdown the module globals may be gone already)
            if self.h is not None and N._lib is not None:
                N._lib.k4_tape_free(self.h)
                self.h = None
        except Exception:
            pass


class _Lease:
    """Marks a program busy from a forward until its backward (or until the autograd graph that holds the context is dropped)."""

    def __init__(self, prog):
        self.prog = prog
        prog.busy = True

    def release(self):
        if self.prog is not None:
            self.prog.busy = False
            self.prog = None

    __del__ = release


def _sft_params(layer):
    return [layer.SFT_scale_conv0.weight, layer.SFT_scale_conv0.bias, layer.SFT_scale_conv1.weight, layer.SFT_scale_conv1.bias,
            layer.SFT_shift_conv0.weight, layer.SFT_shift_conv0.bias, layer.SFT_shift_conv1.weight, layer.SFT_shift_conv1.bias]


def eligible(net, x, cond):
    """The shapes the fused training kernels cover (everything the reference instantiates: SFTNet(3, scale=4), num_cond=1)."""
    if net.num_grow_ch != 32 or net.num_feat not in (32, 64) or net.CondNet[6].weight.shape[0] != 32 or net.scale not in (1, 2, 4):
        return False
    if x.dim() != 4 or cond.dim() != 4 or x.shape[0] != 1 or cond.shape[0] != 1 or x.shape[2:] != cond.shape[2:]:
        return False
    if x.shape[1] != net.conv_first.in_channels or cond.shape[1] != net.CondNet[0].in_channels or getattr(net, 'dswise', False):
        return False
    dev = net.conv_first.weight.device
    return x.device == dev and cond.device == dev and x.dtype == torch.float32 and cond.dtype == torch.float32


def _params_ok(net):
    """fp32, contiguous, on one device: checked when a program is built (a program is rebuilt whenever a parameter's storage moves)."""
    dev = net.conv_first.weight.device
    return all(p.dtype == torch.float32 and p.is_contiguous() and p.device == dev for p in net.parameters())


class DecoderProgram:
    def __init__(self, net, cache, h, w, x_grad, cond_grad):
        self.net, self.cache, self.h, self.w, self.x_grad, self.cond_grad = net, cache, h, w, x_grad, cond_grad
        self.busy, self.gen = False, 0
        self.fwd_tape = self.bwd_tape = None
        dev = self.dev = net.conv_first.weight.device
        nf, g, s, nb = net.num_feat, net.num_grow_ch, net.scale, len(net.body)
        self.nf, self.g, self.s, self.nb, self.bw = nf, g, s, nb, nf + 4 * g
        cin, ncond = net.conv_first.in_channels, net.CondNet[0].in_channels
        self.main = N.stream().value or 0
        self.side = T._side_stream(dev)
        # a third stream for what the chain does not read of the SFT layers' backward (k4_sft_train_bwd_rest + reductions: condition gradient, parameter gradients)
        self.aux = T._aux_stream(dev) if self.side is not None else None
        f32 = dict(dtype=torch.float32, device=dev)

        def E(*shape):
            return torch.empty(shape, **f32)

        # ---- packed operands: the persistent buffers of the weight cache's pack plan (forward + dgrad form of every convolution)
        self.convs = net._k4.get(('train_convs', True))
        if self.convs is None:
            self.convs = net._k4[('train_convs', True)] = [m for name, m in net.named_modules()
                                                           if isinstance(m, torch.nn.Conv2d) and '.SFT_' not in '.' + name]
        cache.prepack(self.convs)
        if cache._plan is None:
            raise N.K4Error('DecoderProgram: the weight cache has no live pack plan (non-fp32 / non-contiguous parameters)')
        self.packplan = cache._plan[1]
        lo, hi = self.packplan.wbuf.data_ptr(), self.packplan.wbuf.data_ptr() + 2 * self.packplan.wbuf.numel()

        # (looked up ONCE, right after prepack: the cache hands out a temporary packing for an operand whose weight an optimizer step has
        # changed since -- the tapes must name the plan's buffers, which their first call re-packs)
        pkf, pkb = {}, {}
        for m_ in self.convs:
            pkf[id(m_)], pkb[id(m_)] = cache.fwd(m_.weight, m_.bias), cache.bwd(m_.weight)
            for pk in (pkf[id(m_)], pkb[id(m_)]):
                if not (lo <= pk.w.data_ptr() < hi and pk.mode == 'bf16x6'):
                    raise N.K4Error('DecoderProgram: a packed operand lies outside the pack plan')

        def fw(m):
            return pkf[id(m)]

        def bw_(m):
            return pkb[id(m)]

        self.fw, self.bw_ = fw, bw_
        # ---- activations (kept for the backward pass)
        A = self.A = {'xi': E(h, w, cin), 'ci': E(h, w, ncond), 'feat': E(h, w, nf), 'c1': E(h, w, 64), 'c2': E(h, w, 64), 'c3': E(h, w, 64),
                      'c': E(h, w, 32), 'sb': E(h, w, nf), 'bf': E(h, w, nf)}
        for q in range(3 * nb):
            A[f'buf{q}'], A[f'x4{q}'], A[f'o{q}'] = E(h, w, self.bw), E(h, w, g), E(h, w, nf)
        for b in range(nb):
            A[f'body{b}'] = E(h, w, nf)
        m = 1
        if s > 1:
            A['ubf'], A['u1'] = E(2 * h, 2 * w, nf), E(2 * h, 2 * w, nf)
            m = 2
            if s == 4:
                A['uu1'], A['u2'] = E(4 * h, 4 * w, nf), E(4 * h, 4 * w, nf)
                m = 4
        A['hr'], A['out'] = E(m * h, m * w, nf), E(m * h, m * w, 3)
        # ---- gradients
        G = self.G = {'out': E(m * h, m * w, 3), 'hr': E(m * h, m * w, nf), 'top': E(m * h, m * w, nf), 'bf': E(h, w, nf), 'sb': E(h, w, nf),
                      'o3': E(h, w, nf), 'feat': E(h, w, nf), 'acc': E(h, w, 32),
                      'c3': E(h, w, 64), 'c2': E(h, w, 64), 'c1': E(h, w, 64)}
        for b in range(nb + 1):                # gradient of RRDB b's input (b = nb: of the trunk's output): one buffer each -- a layer's deferred backward on the
            G[f'gb{b}'] = E(h, w, nf)          # third stream still reads its grad_y while the chain is an RRDB further
        if s > 1:
            G['ubf'], G['u1'] = E(2 * h, 2 * w, nf), E(2 * h, 2 * w, nf)
            if s == 4:
                G['uu1'] = E(4 * h, 4 * w, nf)
        if x_grad:
            G['xi'] = E(h, w, cin)
        if cond_grad:
            G['ci'] = E(h, w, ncond)
        L = N.lib()
        n = h * w
        self.sft_ws_bytes = {C: int(L.k4_sft_train_bwd_workspace_bytes(n, C)) for C in (nf, g)}
        # the stand-alone SFT layers' backward workspaces: one per layer (their reductions run on the side stream while the chain moves on)
        self.sft_ws = {name: torch.empty([self.sft_ws_bytes[nf] // 4], **f32) for name in ['sftbody'] + [f'sft{b}' for b in range(nb)]}
        # ---- parameter gradients: ONE flat buffer; a convolution's [dW | dbias] adjacent (one launch writes both)
        self.hand = []                      # (parameter, offset, shape)
        off = 0

        def take(p):
            nonlocal off
            self.hand.append((p, off, tuple(p.shape)))
            o = off
            off += (p.numel() + 3) // 4 * 4                                           # 16-byte aligned pieces
            return o

        def take_conv(mod):
            nonlocal off
            o = off
            self.hand.append((mod.weight, off, tuple(mod.weight.shape)))
            off += mod.weight.numel()
            self.hand.append((mod.bias, off, tuple(mod.bias.shape)))
            off += mod.bias.numel()
            off = (off + 3) // 4 * 4
            return o

        self.pg_off = {}
        for name in ['conv_first', 'conv_body', 'conv_up1', 'conv_up2', 'conv_hr', 'conv_last']:
            if hasattr(net, name):
                self.pg_off[name] = take_conv(getattr(net, name))
        for i in (0, 2, 4, 6):
            self.pg_off[f'cn{i}'] = take_conv(net.CondNet[i])
        self.blocks = []
        for b, rr in enumerate(net.body):
            for r in (1, 2, 3):
                blk = getattr(rr, f'rdb{r}')
                span0 = off
                cv = [take_conv(getattr(blk, f'conv{k}')) for k in range(1, 6)]
                span1 = off
                s0 = [take(p) for p in _sft_params(blk.sft0)]
                s1 = [take(p) for p in _sft_params(blk.sft1)]
                self.blocks.append((blk, cv, (span0, span1 - span0), s0, s1))
            self.pg_off[f'sft{b}'] = [take(p) for p in _sft_params(rr.sft0)]
        self.pg_off['sftbody'] = [take(p) for p in _sft_params(net.sftbody)]
        self.pg = torch.empty([off], **f32)
        self.views = [self.pg[o:o + p.numel()].view(shape) for p, o, shape in self.hand]
        # ---- dense blocks: descriptors (forward + backward fields) and scratch
        nb0, nb1 = self.sft_ws_bytes[nf], self.sft_ws_bytes[g]
        ssz = [n * nf, n * self.bw, n * g, nb0 // 4, nb1 // 4, n * nf]
        soff = [0]
        for q in ssz:
            soff.append(soff[-1] + q)
        self.scr, self.desc, self.keep = [], [], []
        pb = self.pg.data_ptr()
        for q, (blk, cv, (span0, spanf), s0, s1) in enumerate(self.blocks):
            P = _sft_params(blk.sft0) + [t for k in range(1, 6) for t in (getattr(blk, f'conv{k}').weight, getattr(blk, f'conv{k}').bias)] + _sft_params(blk.sft1)
            t_in = A['feat'] if q == 0 else (A[f'body{q // 3 - 1}'] if q % 3 == 0 else A[f'o{q - 1}'])
            d = T._rdb_desc(t_in, A['c'], A[f'buf{q}'], A[f'x4{q}'], P, h, w, nf, g)
            d.out = A[f'o{q}'].data_ptr()
            for k in range(5):
                m_ = getattr(blk, f'conv{k + 1}')
                pf, pbk = fw(m_), bw_(m_)
                assert pf.k == 3 and pf.flags_extra == 0 and pbk.k == 3 and pbk.flags_extra == 0
                d.w_fwd[k], d.b_fwd[k] = pf.w.data_ptr(), pf.b.data_ptr()
                d.w_bwd[k], d.b_bwd[k] = pbk.w.data_ptr(), pbk.b.data_ptr()
                d.dwdb[k] = pb + 4 * cv[k]
                self.keep += [pf, pbk]
            for i in range(8):
                d.gsft0[i], d.gsft1[i] = pb + 4 * s0[i], pb + 4 * s1[i]
            d.dwdb_span, d.dwdb_span_floats = pb + 4 * span0, spanf
            scr = torch.empty([soff[-1]], **f32)
            sb = scr.data_ptr()
            d.gx0, d.G, d.gx4, d.ws0, d.ws1, d.g5 = (sb + 4 * o for o in soff[:6])
            d.ws0_bytes, d.ws1_bytes = nb0, nb1
            d.g5_from_gx0_add, d.fused_lrelu = 1, 1
            d.gc_acc = G['acc'].data_ptr()
            d.side_stream = self.side
            d.no_join = int(self.side is not None)                # ONE join, behind the whole backward pass (_backward_calls): the buffers are this program's own
            d.defer_side = int(self.side is not None)             # ... and ONE fork per block: every event record on the chain's stream costs ~7 us of the chain's time
            d.aux_stream = self.aux                               # ... and only the grad_x part of the two SFT layers' backward on the chain
            d.aux_wgrad = int(T._AUX_WGRAD and q > 0)             # ... one of the five weight gradients on the third stream (not in the last block of the pass: the chain joins that stream right after it)
            self.scr.append(scr)
            self.desc.append(d)
        self.signature = signature(net)

    # ------------------------------------------------------------------------------------------------ call helpers
    @staticmethod
    def _conv(pk, x, cin_stride, y, cout, cout_stride, H, W, flags=0, res=None, res_scale=0.0):
        N.check(N.lib().k4_conv2d_nhwc_bf16x6(N.f32(x), pk.cin, cin_stride, N.ptr(pk.w), N.f32(pk.b), pk.k, N.f32(y), cout, cout_stride, H, W,
                                              flags | pk.flags_extra | (CONV_SMALL if pk.k == 3 and not pk.flags_extra else 0), 0.2, None if res is None else N.f32(res), 0 if res is None else cout_stride,
                                              res_scale, None, 0, N.stream()), 'k4_conv2d_nhwc_bf16x6')

    def _side_call(self, fn, aux=False):
        """A launch that only produces parameter gradients (weight gradients, SFT reductions): queued for the side stream and issued by _flush_side behind a fork that
        covers a whole section of the chain (a fork per launch put an event record -- ~7 us of the chain's time -- in front of every dgrad launch); without a side
        stream it runs now, on the chain's stream.  aux: queued for the third stream instead (the tail of the pass: both streams share the last weight gradients)."""
        if self.side is None:
            fn(N.stream())
        elif aux and self.aux is not None:
            self._aux_q.append(fn)
        else:
            self._side_q.append(fn)

    def _flush_side(self, fork):
        """Issue the queued side-stream launches.  fork=False: the side stream already waits for everything queued so far (a dense block's call has just forked)."""
        if self.side is None:
            return
        for which, stream in (('_side_q', self.side), ('_aux_q', self.aux)):
            q = getattr(self, which)
            if not q:
                continue
            st = N.C.c_void_p(stream)
            if fork:
                N.check(N.lib().k4_side_wait_main(st, N.stream()), 'k4_side_wait_main')
            for fn in q:
                fn(st)
            setattr(self, which, [])

    def _wgrad(self, mod, x, gy, H, W, name, aux=False):
        """[dW | dbias] of `mod` into its piece of the flat buffer, on the side stream (_side_call)."""
        L = N.lib()
        cout, cin, k, _ = mod.weight.shape
        args = (N.f32(x), cin, cin, N.f32(gy), cout, cout, k, H, W, N.C.c_void_p(self.pg.data_ptr() + 4 * self.pg_off[name]))
        self._side_call(lambda st: N.check(L.k4_conv2d_wgrad_dbias_bf16x6(*args, st), 'k4_conv2d_wgrad_dbias_bf16x6'), aux=aux)

    def _lrelu_bwd(self, g, y):
        C = g.shape[2]
        N.check(N.lib().k4_lrelu_bwd(N.f32(g), C, N.f32(y), C, g.shape[0] * g.shape[1], C, 0.2, N.f32(g), C, N.stream()), 'k4_lrelu_bwd')

    def _sft_fwd(self, layer, x, y, res=None, res_scale=1.0):
        C = x.shape[2]
        N.check(N.lib().k4_sft_train_fwd_ex(N.f32(x), C, N.f32(self.A['c']), 32, self.h * self.w, C, *[N.f32(p) for p in _sft_params(layer)], 0.2,
                                            N.f32(y), C, None if res is None else N.f32(res), C, float(res_scale), N.stream()), 'k4_sft_train_fwd_ex')

    def _sft_bwd(self, layer, x, gy, gx, name, gy_scale, scaled_out=None):
        C = x.shape[2]
        ps = _sft_params(layer)
        pb = self.pg.data_ptr()
        L = N.lib()
        n = self.h * self.w
        if self.aux is not None:              # the chain's launch writes grad_x only; the rest of the layer on the third stream (k4_rdb_train.aux_stream does the same inside a block)
            # scaled_out: the g5 (= 0.2 grad_out) of the dense block that receives gx as its grad_out, written here instead of by a launch of its own
            N.check(L.k4_sft_train_bwd_gx(None, C, N.f32(self.A['c']), 32, N.f32(gy), C, n, C, *[N.f32(p) for p in ps[:4]], 0.2, N.f32(gx), None, 0, 0, float(gy_scale),
                                          None if scaled_out is None else N.C.c_void_p(scaled_out), 0.2, None, None, N.stream()), 'k4_sft_train_bwd_gx')
            # issued NOW, behind a fork of its own (six of these layers per pass): the third stream adds the layers' condition gradients in the chain's order
            st = N.C.c_void_p(self.aux)
            N.check(L.k4_side_wait_main(st, N.stream()), 'k4_side_wait_main')
            N.check(L.k4_sft_train_bwd_rest(N.f32(x), C, N.f32(self.A['c']), 32, N.f32(gy), C, n, C, *[N.f32(p) for p in ps[:7]], 0.2, N.f32(self.G['acc']),
                                            N.f32(self.sft_ws[name]), self.sft_ws_bytes[C], 1, float(gy_scale), st), 'k4_sft_train_bwd_rest')
            N.check(L.k4_sft_train_reduce(N.f32(self.sft_ws[name]), n, C, *[N.C.c_void_p(pb + 4 * o) for o in self.pg_off[name]], st), 'k4_sft_train_reduce')
            return
        N.check(L.k4_sft_train_bwd_main(N.f32(x), C, N.f32(self.A['c']), 32, N.f32(gy), C, n, C, *[N.f32(p) for p in ps[:7]], 0.2,
                                        N.f32(gx), N.f32(self.G['acc']), N.f32(self.sft_ws[name]), self.sft_ws_bytes[C], None, 0, 1, 0, float(gy_scale), N.stream()),
                'k4_sft_train_bwd_main')
        rargs = (N.f32(self.sft_ws[name]), n, C, *[N.C.c_void_p(pb + 4 * o) for o in self.pg_off[name]])
        self._side_call(lambda st: N.check(L.k4_sft_train_reduce(*rargs, st), 'k4_sft_train_reduce'))

    # ------------------------------------------------------------------------------------------------ the pass, call by call
    def _forward_calls(self):
        """SFTNet.forward (lib/sr_esrnet.py:446-465) on the NHWC buffers, xi / ci already filled."""
        net, A, h, w, nf = self.net, self.A, self.h, self.w, self.nf
        L = N.lib()
        self.packplan.run()                                                       # every operand from the current weights (3 launches)
        cin, ncond = A['xi'].shape[2], A['ci'].shape[2]
        self._conv(self.fw(net.conv_first), A['xi'], cin, A['feat'], nf, nf, h, w)
        cn = net.CondNet
        self._conv(self.fw(cn[0]), A['ci'], ncond, A['c1'], 64, 64, h, w, EPI_LRELU)
        self._conv(self.fw(cn[2]), A['c1'], 64, A['c2'], 64, 64, h, w, EPI_LRELU)
        self._conv(self.fw(cn[4]), A['c2'], 64, A['c3'], 64, 64, h, w, EPI_LRELU)
        self._conv(self.fw(cn[6]), A['c3'], 64, A['c'], 32, 32, h, w)
        body = A['feat']
        for b, rr in enumerate(net.body):                                         # lib/sr_esrnet.py:176-182
            for r in range(3):
                N.check(L.k4_rdb_train_fwd(N.C.byref(self.desc[3 * b + r]), N.stream()), 'k4_rdb_train_fwd')
            self._sft_fwd(rr.sft0, A[f'o{3 * b + 2}'], A[f'body{b}'], res=body, res_scale=0.2)      # out * 0.2 + x
            body = A[f'body{b}']
        self._sft_fwd(net.sftbody, body, A['sb'])
        self._conv(self.fw(net.conv_body), A['sb'], nf, A['bf'], nf, nf, h, w, EPI_RES, res=A['feat'], res_scale=1.0)      # conv_body(...) + feat
        cur, m = A['bf'], 1
        if self.s > 1:
            N.check(L.k4_upsample2x_nhwc(N.f32(cur), h, w, nf, N.f32(A['ubf']), N.stream()), 'k4_upsample2x_nhwc')
            self._conv(self.fw(net.conv_up1), A['ubf'], nf, A['u1'], nf, nf, 2 * h, 2 * w, EPI_LRELU)
            cur, m = A['u1'], 2
            if self.s == 4:
                N.check(L.k4_upsample2x_nhwc(N.f32(cur), 2 * h, 2 * w, nf, N.f32(A['uu1']), N.stream()), 'k4_upsample2x_nhwc')
                self._conv(self.fw(net.conv_up2), A['uu1'], nf, A['u2'], nf, nf, 4 * h, 4 * w, EPI_LRELU)
                cur, m = A['u2'], 4
        self._conv(self.fw(net.conv_hr), cur, nf, A['hr'], nf, nf, m * h, m * w, EPI_LRELU)
        self._conv(self.fw(net.conv_last), A['hr'], nf, A['out'], 3, 3, m * h, m * w)

    def _backward_calls(self):
        """The backward pass in the order the autograd engine ran lib/sr_train.forward_train's graph; G['out'] already filled."""
        net, A, G, h, w, nf, s = self.net, self.A, self.G, self.h, self.w, self.nf, self.s
        L = N.lib()
        m = s if s in (2, 4) else 1
        n = h * w
        self._side_q, self._aux_q = [], []
        N.check(L.k4_zero_f32(N.f32(G['acc']), G['acc'].numel(), N.stream()), 'k4_zero_f32')        # every SFT layer ADDS its condition gradient
        # conv_last, conv_hr
        self._wgrad(net.conv_last, A['hr'], G['out'], m * h, m * w, 'conv_last')
        self._conv(self.bw_(net.conv_last), G['out'], 3, G['hr'], nf, nf, m * h, m * w)
        self._lrelu_bwd(G['hr'], A['hr'])
        top_in = A['u2'] if s == 4 else (A['u1'] if s == 2 else A['bf'])
        self._wgrad(net.conv_hr, top_in, G['hr'], m * h, m * w, 'conv_hr')
        g_top = G['top'] if s > 1 else G['bf']
        self._conv(self.bw_(net.conv_hr), G['hr'], nf, g_top, nf, nf, m * h, m * w)
        if s == 4:
            self._lrelu_bwd(g_top, A['u2'])
            self._wgrad(net.conv_up2, A['uu1'], g_top, 4 * h, 4 * w, 'conv_up2')
            self._conv(self.bw_(net.conv_up2), g_top, nf, G['uu1'], nf, nf, 4 * h, 4 * w)
            N.check(L.k4_upsample2x_bwd_nhwc(N.f32(G['uu1']), 2 * h, 2 * w, nf, N.f32(G['u1']), N.stream()), 'k4_upsample2x_bwd_nhwc')
            g_top = G['u1']
        if s > 1:
            self._lrelu_bwd(g_top, A['u1'])
            self._wgrad(net.conv_up1, A['ubf'], g_top, 2 * h, 2 * w, 'conv_up1')
            self._conv(self.bw_(net.conv_up1), g_top, nf, G['ubf'], nf, nf, 2 * h, 2 * w)
            N.check(L.k4_upsample2x_bwd_nhwc(N.f32(G['ubf']), h, w, nf, N.f32(G['bf']), N.stream()), 'k4_upsample2x_bwd_nhwc')
        # bf = conv_body(sb) + feat
        self._wgrad(net.conv_body, A['sb'], G['bf'], h, w, 'conv_body')
        self._conv(self.bw_(net.conv_body), G['bf'], nf, G['sb'], nf, nf, h, w)
        nb = self.nb
        g_body = G[f'gb{nb}']
        self._sft_bwd(net.sftbody, A[f'body{nb - 1}'] if nb else A['feat'], G['sb'], g_body, 'sftbody', 1.0)
        self._flush_side(fork=True)                                                # the high-resolution layers' weight gradients run beside the trunk's chain
        for b in range(nb - 1, -1, -1):
            rr = net.body[b]
            fold = self.aux is not None          # the chain's grad_x launches also write what k_scale_f32 (a block's g5) and k4_add_f32 (the RRDB's input gradient) did
            self._sft_bwd(rr.sft0, A[f'o{3 * b + 2}'], g_body, G['o3'], f'sft{b}', 0.2,      # body_b = sft(o3) * 0.2 + body_{b-1}
                          scaled_out=self.desc[3 * b + 2].g5 if fold else None)
            go = G['o3']
            # the RRDB's input reaches its first dense block and the skip connection: the sum of both gradients.  The first RRDB's input is `feat`,
            # which the long skip connection reads too: three addends, summed in the order the autograd engine received them (same roundings)
            g_other = G[f'gb{b}']
            for r in (2, 1, 0):
                q = 3 * b + r
                d = self.desc[q]
                d.gx0_add = go.data_ptr()
                if fold:
                    d.g5_given = 1
                    d.g5_next = self.desc[q - 1].g5 if r > 0 else None
                    if r == 0 and b > 0:
                        d.gx0_add2, d.gx0_sum2 = g_body.data_ptr(), g_other.data_ptr()
                    elif r == 0:
                        N.check(L.k4_add_f32(N.f32(G['bf']), N.f32(g_body), N.f32(g_other), n * nf, N.stream()), 'k4_add_f32')
                        d.gx0_add2, d.gx0_sum2 = g_other.data_ptr(), G['feat'].data_ptr()
                N.check(L.k4_rdb_train_bwd(N.C.byref(d), N.stream()), 'k4_rdb_train_bwd')
                self._flush_side(fork=False)                                       # (the block's call ended with a fork: what was queued before it is covered)
                go = self.scr[q][:n * nf].view(h, w, nf)                           # = go + the gradient through the block's sft0
            if fold:
                g_body = g_other
            elif b > 0:
                N.check(L.k4_add_f32(N.f32(go), N.f32(g_body), N.f32(g_other), n * nf, N.stream()), 'k4_add_f32')
                g_body = g_other
            else:
                N.check(L.k4_add_f32(N.f32(G['bf']), N.f32(g_body), N.f32(g_other), n * nf, N.stream()), 'k4_add_f32')
                N.check(L.k4_add_f32(N.f32(g_other), N.f32(go), N.f32(G['feat']), n * nf, N.stream()), 'k4_add_f32')
        if nb == 0:
            N.check(L.k4_add_f32(N.f32(G['bf']), N.f32(g_body), N.f32(G['feat']), n * nf, N.stream()), 'k4_add_f32')
        self._wgrad(net.conv_first, A['xi'], G['feat'], h, w, 'conv_first')
        # The tail of the pass.  Behind the last dense block the weight gradients' stream still holds that block's five launches; conv_first's joins them at once when the
        # block's closing fork covers G['feat'] (it does when the block's grad_x launch wrote it), and the CondNet's four are dealt to both side streams behind the closing fork:
        # the pass ends ~75 us earlier than with all nine in one queue behind the chain's last launch.
        tail_split = self.aux is not None and nb > 0 and T._TAIL_SPLIT
        if tail_split:
            self._flush_side(fork=False)
        if self.x_grad:
            cin = A['xi'].shape[2]
            self._conv(self.bw_(net.conv_first), G['feat'], nf, G['xi'], cin, cin, h, w)
        # CondNet: G['acc'] holds the sum over every SFT layer -- once the third stream has added the last of them
        if self.aux is not None:
            N.check(L.k4_main_wait_side(N.C.c_void_p(self.aux), N.stream()), 'k4_main_wait_side')
        cn = net.CondNet
        self._wgrad(cn[6], A['c3'], G['acc'], h, w, 'cn6')
        self._conv(self.bw_(cn[6]), G['acc'], 32, G['c3'], 64, 64, h, w)
        self._lrelu_bwd(G['c3'], A['c3'])
        self._wgrad(cn[4], A['c2'], G['c3'], h, w, 'cn4', aux=tail_split)
        self._conv(self.bw_(cn[4]), G['c3'], 64, G['c2'], 64, 64, h, w)
        self._lrelu_bwd(G['c2'], A['c2'])
        self._wgrad(cn[2], A['c1'], G['c2'], h, w, 'cn2')
        self._conv(self.bw_(cn[2]), G['c2'], 64, G['c1'], 64, 64, h, w)
        self._lrelu_bwd(G['c1'], A['c1'])
        self._wgrad(cn[0], A['ci'], G['c1'], h, w, 'cn0', aux=tail_split)
        if self.cond_grad:
            ncond = A['ci'].shape[2]
            self._conv(self.bw_(cn[0]), G['c1'], 64, G['ci'], ncond, ncond, h, w)
        self._flush_side(fork=True)
        if self.side is not None:
            N.check(L.k4_main_wait_side(N.C.c_void_p(self.side), N.stream()), 'k4_main_wait_side')      # the weight gradients are done before the optimizer reads them
            if tail_split:
                N.check(L.k4_main_wait_side(N.C.c_void_p(self.aux), N.stream()), 'k4_main_wait_side')

    # ------------------------------------------------------------------------------------------------ entry points of the autograd node
    def run_forward(self, x, cond):
        self.gen += 1
        self.A['xi'].copy_(x[0].permute(1, 2, 0))
        self.A['ci'].copy_(cond[0].permute(1, 2, 0))
        if self.fwd_tape is None:
            self.fwd_tape = _Tape().record(self._forward_calls)
        else:
            self.fwd_tape.replay()
        return self.A['out'].clone().permute(2, 0, 1).unsqueeze(0)                 # (a copy: the buffer is rewritten by the next forward)

    def run_backward(self, g):
        self.G['out'].copy_(g[0].permute(1, 2, 0))
        # gradients a caller kept from an earlier pass through this program live in the flat buffer this pass rewrites: detach them first
        lo, hi = self.pg.data_ptr(), self.pg.data_ptr() + 4 * self.pg.numel()
        pending = []
        for (p, _, _), v in zip(self.hand, self.views):
            if p.grad is not None and p.requires_grad:
                if lo <= p.grad.data_ptr() < hi:
                    p.grad = p.grad.clone()
                pending.append((p, v))
        if self.bwd_tape is None:
            self.bwd_tape = _Tape().record(self._backward_calls)
        else:
            self.bwd_tape.replay()
        if pending:                                                                 # what AccumulateGrad does for a leaf that already has a gradient
            have = {id(p) for p, _ in pending}
            for p, v in pending:
                p.grad.add_(v)
            for (p, _, _), v in zip(self.hand, self.views):
                if id(p) not in have and p.requires_grad:
                    p.grad = v
        else:
            for (p, _, _), v in zip(self.hand, self.views):
                if p.requires_grad:
                    p.grad = v
        gx = self.G['xi'].clone().permute(2, 0, 1).unsqueeze(0) if self.x_grad else None
        gc = self.G['ci'].clone().permute(2, 0, 1).unsqueeze(0) if self.cond_grad else None
        return gx, gc


def signature(net):
    """What a recorded tape depends on besides the shapes: the parameters' storage and the stream pair."""
    return tuple(p.data_ptr() for p in net.k4_parameters())


class K4DecoderTape(torch.autograd.Function):
    """SFTNet(x, cond) as ONE autograd node.  `anchor`: a parameter that requires grad (the node must be part of the graph when neither input
    does: decoder-only training); every parameter gradient is written to ``.grad`` by the node itself."""

    @staticmethod
    def forward(ctx, x, cond, prog, anchor):
        out = prog.run_forward(x, cond)
        ctx.prog, ctx.gen, ctx.lease = prog, prog.gen, _Lease(prog)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        prog = ctx.prog
        if ctx.gen != prog.gen:
            raise N.K4Error('second backward pass through a decoder graph whose buffers a later forward has rewritten (retain_graph use: K4_TRAIN_TAPE=0)')
        try:
            gx, gc = prog.run_backward(g)
        finally:
            ctx.lease.release()
        return gx, gc, None, None


MAX_SHAPES = 4      # program pools (patch shape x gradient need x stream) kept per network: a pool owns ~150 MB of buffers per program


def program_for(net, cache, x, cond):
    """A free program for this network / shape / gradient need, or None when the pool is exhausted or the network is not eligible (forward_train then keeps
    the per-block path)."""
    h, w = int(x.shape[2]), int(x.shape[3])
    main = N.stream().value or 0
    key = ('tape_programs', h, w, bool(x.requires_grad), bool(cond.requires_grad), main, T._side_stream(net.conv_first.weight.device))
    sig = signature(net)
    pool = net._k4.get(key)
    if pool is None or pool[0] != sig or (pool[1] and (cache._plan is None or pool[1][0].packplan is not cache._plan[1])):
        pool = net._k4[key] = (sig, [] if _params_ok(net) else None)               # parameters moved / the pack plan was rebuilt: the old tapes name dead buffers
    lru = net._k4.setdefault('tape_lru', [])
    if key in lru:
        lru.remove(key)
    lru.append(key)
    for old in [k for k in lru[:-MAX_SHAPES]]:                                      # edge patches of many sizes: the oldest idle pools go
        op = net._k4.get(old)
        if op is None or op[1] is None or not any(p.busy for p in op[1]):
            net._k4.pop(old, None)
            lru.remove(old)
    if pool[1] is None:
        return None
    for prog in pool[1]:
        if not prog.busy:
            return prog
    if len(pool[1]) >= POOL:
        return None
    try:
        prog = DecoderProgram(net, cache, h, w, bool(x.requires_grad), bool(cond.requires_grad))
    except N.K4Error:
        if not pool[1]:                                                             # this network cannot be taped (no live pack plan): remember, per-block path
            net._k4[key] = (sig, None)
        return None
    pool[1].append(prog)
    return prog
