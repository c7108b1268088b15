"""Tile-parallel 4K rendering across the GPUs of one node (BASELINE configs[3]).

The reference is single-process / single-GPU (run_sr.py:3).  Rays are independent and SR tiles are independent by
construction of ``SFTNet.tile_process`` (lib/sr_esrnet.py:482-526), so the natural unit is ONE reference SR tile:
rank r owns a subset of the tiles of ``tile_geometry(H, W, tile_size, tile_pad)``, marches only the rays of each owned
tile's PADDED window (halo = tile_pad recomputed redundantly, no halo exchange), super-resolves the window, crops it,
and ONE ``all_gather_into_tensor`` per frame moves the final HR pixels -- nothing else ever crosses xGMI.  Every rank
ends with the full frame, bit-identical to the single-GPU ``tile_process`` with the same ``tile_size``.

One process per GPU, ``torch.distributed`` backend "nccl" (= RCCL on ROCm); the payload is 3*4H*4W floats per frame
(146 MB at 4032x3024, 18 MB per rank at 8 ranks): over the fully connected xGMI mesh this is one direct push per peer
(~0.12 ms of wire time at 153 GB/s/link), so it is issued asynchronously and never bucketed or ring-scheduled by hand.

``march_fn`` / ``sr_fn`` are injectable so that the sharding / gather / assembly logic is exercised by world_size-2
``gloo`` tests on CPU (tests/test_tile_parallel.py) with the CPU oracle standing in for the HIP kernels.
"""
import contextlib
import math

import torch
import torch.distributed as dist

from . import _native as _N
from .lib.utils import window_to_planes


def tile_geometry(height, width, tile_size, tile_pad=10):
    """The reference's tile loop as data (lib/sr_esrnet.py:478-497):
    (y0, y1, x0, x1, yp0, yp1, xp0, xp1) = unpadded tile and its clipped padded window, row-major tile order."""
    tiles = []
    for y in range(math.ceil(height / tile_size)):
        for x in range(math.ceil(width / tile_size)):
            x0, y0 = x * tile_size, y * tile_size
            x1, y1 = min(x0 + tile_size, width), min(y0 + tile_size, height)
            tiles.append((y0, y1, x0, x1, max(y0 - tile_pad, 0), min(y1 + tile_pad, height),
                          max(x0 - tile_pad, 0), min(x1 + tile_pad, width)))
    return tiles


def assign_tiles(tiles, world_size):
    """Longest-processing-time-first balance by padded area (cost ~ marched rays + SR pixels).
    -> list (per rank) of tile indices, each sorted; deterministic, identical on every rank."""
    order = sorted(range(len(tiles)), key=lambda i: (-(tiles[i][5] - tiles[i][4]) * (tiles[i][7] - tiles[i][6]), i))
    load = [0] * world_size
    owned = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda q: (load[q], q))
        owned[r].append(i)
        load[r] += (tiles[i][5] - tiles[i][4]) * (tiles[i][7] - tiles[i][6])
    return [sorted(o) for o in owned]


def _slots(tiles, owned, scale):
    """Per rank: number of HR pixels it contributes; the all-gather slot is the maximum (padded, equal sized)."""
    counts = [sum((tiles[i][1] - tiles[i][0]) * (tiles[i][3] - tiles[i][2]) * scale * scale for i in o) for o in owned]
    return counts, max(counts) if counts else 0


@torch.no_grad()
def render_frame_tiles(rays, H, W, march_fn, sr_fn, tile_size, tile_pad=10, scale=4, group=None, out=None, out_dtype=None):
    """One 4K frame, tiles sharded over the process group.

    rays      : (rays_o, rays_d, viewdirs) of the full LR frame, [H,W,3] each, on this rank's device
    march_fn  : (ro [n,3], rd [n,3], vd [n,3], window_w) -> (rgb_feature [n,3], depth [n])       (unclamped, run_sr.py:131)
    sr_fn     : (img [1,3,h,w], cond [1,1,h,w]) -> [1,3,scale*h,scale*w]
    -> [1,3,scale*H,scale*W] on every rank.
    out_dtype : None / torch.float32 -- the decoder's fp32 pixels (bit-identical to the single-GPU frame); torch.uint8 -- every rank
                quantises its own pixels with the reference's ``utils.to8b`` rule (clip to [0,1], x255, truncate: what run_sr.py writes to
                PNG / video) BEFORE the exchange: the all-gather moves 36.6 MB per 4K frame instead of 146 MB, and the assembled frame
                equals to8b of the fp32 frame byte for byte.

    = ``decode_frame_tiles(march_frame_tiles(...))``.  A renderer of MANY frames may issue frame i+1's march before frame i's decode
    (``march_frame_tiles`` runs on side streams that only wait for what is queued on the current stream at the time of the call): the
    march of the next frame then runs under the decoder of the current one -- same pixels, the frames are independent (bench.py four_k,
    ``frames_per_s_pipelined``).
    """
    st = march_frame_tiles(rays, H, W, march_fn, sr_fn, tile_size, tile_pad, scale, group)
    return decode_frame_tiles(st, out=out, out_dtype=out_dtype)


@torch.no_grad()
def march_frame_tiles(rays, H, W, march_fn, sr_fn, tile_size, tile_pad=10, scale=4, group=None):
    """First half of ``render_frame_tiles``: this rank's tile windows marched (each on its own HIP stream on a GPU); with a decoder that
    has no grouped form the windows are decoded here too.  -> state for ``decode_frame_tiles``."""
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    rk = dist.get_rank(group) if dist.is_initialized() else 0
    tiles = tile_geometry(H, W, tile_size, tile_pad)
    owned = assign_tiles(tiles, ws)
    counts, slot = _slots(tiles, owned, scale)
    dev = rays[0].device
    multi = getattr(sr_fn, 'k4_multi', None)            # HIP decoder: all of this rank's windows per layer in ONE grouped launch
    # the gather buffer: filled here by a decoder without a grouped form; decode_frame_tiles makes its own when it needs one (a single process with fp32
    # output writes the frame directly: no 146 MB zero-fill per 4K frame)
    send = torch.zeros([3, slot], dtype=torch.float32, device=dev) if multi is None else None
    off = 0
    # this rank's tiles are independent until the gather: on a GPU each runs on its own HIP stream (own marcher workspace
    # and decoder buffers, `slot`), so that the short last round of one tile's kernels is filled by another tile's
    n_str = _n_streams(dev, len(owned[rk]), march_fn, sr_fn)
    pool = _stream_pool(dev, n_str) if n_str > 1 else None
    cur = torch.cuda.current_stream(dev) if n_str > 1 else None
    multi_ = getattr(sr_fn, 'k4_multi', None)
    # (a single process marches the whole frame on the current stream, below: the decoder's cache check -- a sweep over its 458 parameter versions -- then
    # runs on the host while the march runs on the GPU instead of in front of it)
    whole_ = WHOLE_FRAME_MARCH and ws == 1 and multi_ is not None and dev.type == 'cuda' and getattr(march_fn, 'k4_slots', False) and len(owned[rk]) > 1
    for fn in ((march_fn,) if whole_ else (march_fn, sr_fn)):         # cold caches (k0 repack, packed rgbnet / conv weights) are built on the CURRENT stream,
        warm = getattr(fn, 'k4_warm', None)      # before the side streams fork from it -- never inside one of them
        if warm is not None:
            warm()
    if pool and not whole_:
        for st in pool:
            st.wait_stream(cur)
    pending = []
    # A single process owns every tile: ONE march of the whole frame (what render_viewpoints -> tile_process does, run_sr.py:1361-1390) instead of one per padded
    # window -- the windows overlap by their halos (1.05x the rays at tile 510) and four small launches fill the chip worse than one; a window of the frame's
    # result holds the bits of the window's own march (tests: tile == frame), so the pixels do not change.
    whole = WHOLE_FRAME_MARCH and ws == 1 and multi is not None and dev.type == 'cuda' and getattr(march_fn, 'k4_slots', False) and len(owned[rk]) > 1
    if whole:
        rgb_f, depth_f = march_fn(*[r.reshape(-1, 3) if r.is_contiguous() else r.reshape(-1, 3).contiguous() for r in rays], W)
        rgb_f, depth_f = rgb_f.reshape(H, W, 3), depth_f.reshape(H, W)
        warm = getattr(sr_fn, 'k4_warm', None)
        if warm is not None:
            warm()
        for i in owned[rk]:
            y0, y1, x0, x1, yp0, yp1, xp0, xp1 = tiles[i]
            th, tw = (y1 - y0) * scale, (x1 - x0) * scale
            img = rgb_f[yp0:yp1, xp0:xp1].permute(2, 0, 1).unsqueeze(0)          # (views: the decoder copies its windows into its own NHWC buffers)
            cond = depth_f[yp0:yp1, xp0:xp1].unsqueeze(0).unsqueeze(0)
            pending.append((img, cond, off, th, tw, (y0 - yp0) * scale, (x0 - xp0) * scale, i))
            off += th * tw
        return {'tiles': tiles, 'owned': owned, 'slot': slot, 'send': send, 'pending': pending, 'multi': multi, 'events': None, 'cur': cur,
                'ws': ws, 'group': group, 'H': H, 'W': W, 'scale': scale, 'dev': dev}
    for j, i in enumerate(owned[rk]):
        y0, y1, x0, x1, yp0, yp1, xp0, xp1 = tiles[i]
        th, tw = (y1 - y0) * scale, (x1 - x0) * scale
        with (torch.cuda.stream(pool[j % n_str]) if pool else contextlib.nullcontext()):
            kw = {'slot': j % n_str} if pool else {}
            ro, rd, vd = [r[yp0:yp1, xp0:xp1].reshape(-1, 3).contiguous() for r in rays]
            rgb, depth = march_fn(ro, rd, vd, xp1 - xp0, **kw)
            hh, ww = yp1 - yp0, xp1 - xp0
            img = rgb.reshape(hh, ww, 3).permute(2, 0, 1).unsqueeze(0)
            cond = depth.reshape(1, 1, hh, ww)
            if multi is not None:
                if pool:                                   # produced on a side stream, consumed on the current one after the join
                    rgb.record_stream(cur)
                    depth.record_stream(cur)
                pending.append((img, cond, off, th, tw, (y0 - yp0) * scale, (x0 - xp0) * scale, i))
            else:
                hr = sr_fn(img, cond, **kw)
                oy, ox = (y0 - yp0) * scale, (x0 - xp0) * scale
                window_to_planes(hr, oy, ox, th, tw, send[:, off:off + th * tw].view(3, th, tw))
        off += th * tw
    events = None
    if pool:                                               # joined by decode_frame_tiles (events: the pool's streams may carry the NEXT frame's march by then)
        events = []
        for st in pool:
            ev = torch.cuda.Event()
            ev.record(st)
            events.append(ev)
    return {'tiles': tiles, 'owned': owned, 'slot': slot, 'send': send, 'pending': pending, 'multi': multi, 'events': events, 'cur': cur,
            'ws': ws, 'group': group, 'H': H, 'W': W, 'scale': scale, 'dev': dev}


@torch.no_grad()
def decode_frame_tiles(state, out=None, out_dtype=None):
    """Second half of ``render_frame_tiles``: join the marches, decode this rank's windows (one grouped launch per layer), gather, assemble."""
    tiles, owned, slot, send, pending, multi = (state[k] for k in ('tiles', 'owned', 'slot', 'send', 'pending', 'multi'))
    ws, group, H, W, scale, dev = (state[k] for k in ('ws', 'group', 'H', 'W', 'scale', 'dev'))
    if state['events']:
        cur = torch.cuda.current_stream(dev)
        for ev in state['events']:
            cur.wait_event(ev)
    odt = torch.float32 if out_dtype is None else out_dtype
    # a single process with fp32 output needs no gather buffer: the windows' interiors go straight into the frame (one pass each, utils.window_to_planes)
    direct = multi is not None and ws == 1 and odt == torch.float32 and not (_N.FORCE_COLLECTIVES and dist.is_initialized())
    if direct:
        if out is None:
            out = torch.empty([1, 3, H * scale, W * scale], dtype=odt, device=dev)
        assert out.dtype == odt
    elif send is None:
        send = torch.zeros([3, slot], dtype=torch.float32, device=dev)
    if multi is not None:
        for p0 in range(0, len(pending), multi.max_jobs):
            part = pending[p0:p0 + multi.max_jobs]
            for hr, (_, _, o, th, tw, oy, ox, ti) in zip(multi([p[0] for p in part], [p[1] for p in part]), part):
                if direct:
                    y0, y1, x0, x1 = tiles[ti][:4]
                    window_to_planes(hr, oy, ox, th, tw, out[0, :, y0 * scale:y1 * scale, x0 * scale:x1 * scale])
                else:
                    window_to_planes(hr, oy, ox, th, tw, send[:, o:o + th * tw].view(3, th, tw))
    if direct:
        return out
    if odt == torch.uint8:
        send = _to8b(send)
    elif odt != torch.float32:
        raise ValueError('render_frame_tiles: out_dtype must be torch.float32 or torch.uint8')
    if ws > 1 or (_N.FORCE_COLLECTIVES and dist.is_initialized()):
        recv = torch.empty([ws, 3, slot], dtype=odt, device=dev)
        dist.all_gather_into_tensor(recv.view(ws * 3, slot), send, group=group)      # final pixels only
    else:
        recv = send.unsqueeze(0)
    if out is None:
        out = torch.empty([1, 3, H * scale, W * scale], dtype=odt, device=dev)
    assert out.dtype == odt
    for r in range(ws):
        off = 0
        for i in owned[r]:
            y0, y1, x0, x1 = tiles[i][:4]
            th, tw = (y1 - y0) * scale, (x1 - x0) * scale
            out[0, :, y0 * scale:y1 * scale, x0 * scale:x1 * scale] = recv[r, :, off:off + th * tw].reshape(3, th, tw)
            off += th * tw
    return out


def _to8b(x):
    """utils.to8b (lib/utils.py: (255 * clip(x, 0, 1)).astype(uint8)) of a tensor where it lives: k4_to8b on the GPU, the same rule in torch
    ops for the CPU tensors of the gloo tests."""
    if x.is_cuda:
        from .lib.utils import to8b_device
        return to8b_device(x)
    return (255.0 * x.clamp(0.0, 1.0)).to(torch.uint8)


TILE_STREAMS = 4        # HIP streams a rank deals its tiles to (1: sequential; same pixels either way, tests)
WHOLE_FRAME_MARCH = True        # a single process marches the frame ONCE and cuts its windows out of the result (False: one march per padded window, as every rank of a multi-GPU job does; same pixels, tests)


def _stream_pool(dev, n):
    """n worker streams verified to run beside EACH OTHER (_native.overlapping_stream: streams share a few hardware queues; two tiles dealt to streams on
    one queue run one after the other)."""
    from . import _native as N
    return [N.overlapping_stream(dev, f'tile worker {i}', group='tile workers', beside_main=False) for i in range(n)]


def _n_streams(dev, n_tiles, march_fn, sr_fn):
    """Streams for this rank's tiles: only for the HIP-backed functions (they take a `slot`); TILE_STREAMS (module attribute, 4)."""
    if dev.type != 'cuda' or not (getattr(march_fn, 'k4_slots', False) and getattr(sr_fn, 'k4_slots', False)):
        return 1
    return max(1, min(n_tiles, int(TILE_STREAMS)))


def hip_march_fn(model, render_kwargs):
    """march_fn backed by the fused HIP marcher (DirectMPIGO / DirectVoxGO modules of this package)."""
    kw = dict(render_kwargs)
    kw['render_depth'] = True

    def fn(ro, rd, vd, window_w, slot=0):
        o = model(ro, rd, vd, k4_img_w=window_w, k4_ws_slot=8 + slot, **kw)
        return o['rgb_feature'], o['depth']
    fn.k4_slots = True
    fn.k4_warm = lambda: model.k4_warm(stepsize=kw.get('stepsize'))
    return fn


def hip_sr_fn(net_sr):
    from . import _native as N

    def fn(img, cond, slot=0):
        return net_sr._forward_hip(img, cond, slot=slot)

    def multi(imgs, conds):
        return net_sr._forward_hip_multi(imgs, conds)
    multi.max_jobs = N.K4_MAX_JOBS
    fn.k4_slots = True
    fn.k4_warm = net_sr.k4_warm
    fn.k4_multi = multi
    return fn


def shard_rows(H, world_size, rank, align=8):
    """Marcher-only sharding (BASELINE configs[1] at N>1): contiguous bands of pixel rows, multiples of the 8-row
    wave tile.  -> (row0, row1, rows_per_rank)"""
    rows_per = ((H + world_size - 1) // world_size + align - 1) // align * align
    return min(rank * rows_per, H), min((rank + 1) * rows_per, H), rows_per
