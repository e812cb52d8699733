"""Multi-GPU shim beside the render boundary: one process per GPU, torch.distributed over RCCL/xGMI.

The reference has no distributed code at all; its training loop renders ONE view per
iteration on cuda:0 (/root/reference/train.py:54-69, utils/general_utils.py:133).  Two shardings are
provided, both with exactly one data-path collective per step and nothing else crossing GPUs:

  * view-parallel (BASELINE config 4): every rank holds a replica of the surfel parameters, renders a
    DIFFERENT training view, and the per-surfel gradients are summed with ONE all-reduce over a flat fp32
    bucket of 58 floats / surfel (xyz 3, f_dc 3, f_rest 45, opacity 1, scaling 2, rotation 4 — the Adam
    groups of /root/reference/scene/gaussian_model.py:153-160) = 232 B/surfel.  The densification statistics
    the reference keeps per view (train.py:126-128, gaussian_model.py:405-407) are reduced alongside:
    sum of per-view ||means2D.grad||, sum of visibility, max of radii.
  * tile-band sharding (BASELINE config 5): rank r renders image rows [y0, y1) (multiples of 16, so tile
    boundaries coincide) by giving the unchanged rasterizer a shifted projection (`band_settings`); bands are
    independent (no halo), the loss is a sum over pixels, so the per-surfel gradients of the bands add up —
    the same single all-reduce.

The backward kernels accumulate without atomics, so every rank's gradients are bit-reproducible and an
identical all-reduced bucket + identical Adam step keeps the replicas bit-identical.
xGMI note: 8 GPUs are fully connected (7 links x ~153 GB/s per GPU); one large bucket lets RCCL pick a
direct reduce-scatter + all-gather over all links — do not split it into per-tensor collectives.
"""
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

# One flat fp32 buffer, planar by gradient tensor (the rasterizer's own outputs, so they can be written in place):
# xyz 3 | opacity 1 | scaling 2 | rotation 4 | sh 48 (= f_dc 3 + f_rest 45 per surfel, the reference's two SH parameter groups)
# — the same layout as the surfel parameter store (include/surfel_train.h).
BUCKET_LAYOUT = (("xyz", 3), ("opacity", 1), ("scaling", 2), ("rotation", 4), ("sh", 48))
BUCKET_FLOATS = sum(n for _, n in BUCKET_LAYOUT)      # 58 floats = 232 B per surfel


def view_indices(n_views: int, world: int, rank: int, iteration: int, seed: int = 0) -> int:
    """The training view of `rank` at global step `iteration`: all ranks walk the SAME seeded permutation of the
    views (re-drawn every epoch like the reference's pop-from-shuffled-stack, train.py:64-67), rank-strided, so a
    step consumes `world` distinct views."""
    per_epoch = max(1, n_views // world)
    epoch, k = divmod(iteration, per_epoch)
    g = torch.Generator().manual_seed(seed * 1_000_003 + epoch)
    perm = torch.randperm(n_views, generator=g)
    return int(perm[(k * world + rank) % n_views])


def epoch_schedule(n_views: int, world: int, epoch: int, seed: int = 0) -> List[List[int]]:
    """The whole epoch of view_indices at once: schedule[k][rank] = view of `rank` at step k of `epoch` (identical values;
    callers cache it so that a training step costs no RNG work on the host)."""
    per_epoch = max(1, n_views // world)
    g = torch.Generator().manual_seed(seed * 1_000_003 + epoch)
    perm = torch.randperm(n_views, generator=g).tolist()
    return [[perm[(k * world + r) % n_views] for r in range(world)] for k in range(per_epoch)]


class GradBucket:
    """Flat fp32 gradient bucket of 58 floats / surfel: fill -> ONE all-reduce(SUM) -> read.

    Zero-copy use: `diff_surfel_rasterization.set_grad_arena(bucket.arena())` makes the rasterizer's backward write its
    per-surfel gradients straight into the bucket's sections, so a step is backward -> all_reduce with no packing pass.
    `pack()` (copies) serves gradients that live elsewhere, e.g. parameter .grad tensors behind activation functions."""

    def __init__(self, P: int, device, group=None):
        self.P, self.group = P, group
        self.buf = torch.empty((P * BUCKET_FLOATS,), dtype=torch.float32, device=device)
        self.views, off = {}, 0
        for name, n in BUCKET_LAYOUT:
            self.views[name] = self.buf[off:off + P * n].view(P, n)
            off += P * n

    def arena(self) -> Dict[str, torch.Tensor]:
        """Output tensors for the rasterizer's backward, aliasing the bucket (see set_grad_arena)."""
        v = self.views
        return dict(means3D=v["xyz"], sh=v["sh"].view(self.P, 16, 3), opacities=v["opacity"], scales=v["scaling"],
                    rotations=v["rotation"])

    def pack(self, grads: Dict[str, torch.Tensor]):
        """grads: xyz, opacity, scaling, rotation and either sh [P,16,3] or f_dc [P,1,3] + f_rest [P,15,3]."""
        P = self.P
        for name, n in BUCKET_LAYOUT:
            if name == "sh" and "sh" not in grads:
                dst = self.views["sh"].view(P, 16, 3)
                dst[:, :1].copy_(grads["f_dc"].reshape(P, 1, 3)); dst[:, 1:].copy_(grads["f_rest"].reshape(P, 15, 3))
            else:
                self.views[name].copy_(grads[name].reshape(P, n))
        return self.buf

    def all_reduce(self, average: bool = True, async_op: bool = False):
        work = dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if average and not async_op:
            self.buf.div_(dist.get_world_size(self.group))
        return work

    def unpack(self, like: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        out = {}
        sh = self.views["sh"].view(self.P, 16, 3)
        for name in like:
            src = sh[:, :1] if name == "f_dc" else sh[:, 1:] if name == "f_rest" else sh if name == "sh" else self.views[name]
            out[name] = src.reshape(like[name].shape)
        return out


def reduce_densification_stats(grad_norm: torch.Tensor, visible: torch.Tensor, radii: torch.Tensor, group=None):
    """Per-view statistics -> job-wide: grad_norm [P,1] = ||means2D.grad|| of THIS rank's view (norm taken locally,
    before reducing: gaussian_model.py:406), visible [P] bool, radii [P].  Returns (sum_norm, denom, max_radii)."""
    vis = visible.to(torch.float32).reshape(-1, 1)
    packed = torch.cat([grad_norm.reshape(-1, 1) * vis, vis], dim=1).contiguous()
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    mr = torch.where(visible.reshape(-1), radii.reshape(-1).to(torch.float32), torch.zeros((), device=radii.device)).contiguous()
    dist.all_reduce(mr, op=dist.ReduceOp.MAX, group=group)
    return packed[:, :1], packed[:, 1:2], mr


def broadcast_parameters(params: Sequence[torch.Tensor], src: int = 0, group=None):
    """Start / resume: make every replica bit-identical to rank `src`."""
    for p in params:
        dist.broadcast(p.data if hasattr(p, "data") else p, src=src, group=group)


# ------------------------------------------------------------------------------------------- tile bands
def band_bounds(H: int, world: int, weights: Optional[Sequence[float]] = None, multiple: int = 16) -> List[tuple]:
    """Split image rows into `world` contiguous bands whose edges are multiples of `multiple` rows (16 = tile rows; the
    halo-sharded loss uses 32 so that band edges also coincide with the loss kernels' 32-row blocks).  `weights` (one per
    16-row tile row, e.g. the previous frames' instance counts) balances the bands; default = equal rows."""
    assert multiple % 16 == 0 and world >= 1
    k = multiple // 16
    rows16 = (H + 15) // 16
    w16 = [1.0] * rows16 if weights is None else [float(x) + 1e-6 for x in weights]
    assert len(w16) == rows16
    # band granules of `multiple` rows; a trailing partial granule (H not a multiple of `multiple`) is folded into the one before it,
    # so that no band is ever shorter than `multiple` rows (the halo exchange needs bands of at least HALO = 32 rows: H = 1080 with
    # bottom-heavy weights used to end in a 24-row band)
    rows = max(1, H // multiple)
    edges = [i * multiple for i in range(rows)] + [H]
    w = [sum(w16[i * k:((i + 1) * k if i + 1 < rows else rows16)]) for i in range(rows)]
    total = sum(w)
    cuts, acc, r = [0], 0.0, 0
    for b in range(1, world):
        while r < rows and acc + w[r] <= total * b / world + 1e-9:
            acc += w[r]; r += 1
        r_min = cuts[-1] + 1 if rows >= world else cuts[-1]      # every band gets at least one granule when there are enough
        cuts.append(min(max(r, r_min), rows - (world - b) if rows >= world else rows))
        if cuts[-1] > r:
            acc += sum(w[r:cuts[-1]]); r = cuts[-1]
    cuts.append(rows)
    return [(edges[cuts[i]], edges[cuts[i + 1]]) for i in range(world)]


def band_settings(raster_settings, y0: int, y1: int):
    """Raster settings that make the UNCHANGED rasterizer render rows [y0, y1) of the full image.
    Pixel rows shift by y0 and the image height shrinks, which in the kernel's convention
    y_pix = ((y_ndc + 1) * H - 1) / 2 is the clip-space map  y_clip' = y_clip * H/Hb + w_clip * (H - Hb - 2*y0)/Hb ;
    with row-vector matrices (scene/cameras.py:56-58) that is a change of column 1 of the projection matrix."""
    H = int(raster_settings.image_height)
    Hb = int(y1 - y0)
    assert y0 % 16 == 0 and 0 < Hb and y1 <= H
    pm = raster_settings.projmatrix.clone()
    pm[:, 1] = raster_settings.projmatrix[:, 1] * (H / Hb) + raster_settings.projmatrix[:, 3] * ((H - Hb - 2.0 * y0) / Hb)
    return raster_settings._replace(image_height=Hb, projmatrix=pm, tanfovy=raster_settings.tanfovy * Hb / H)


class _GatherBands(torch.autograd.Function):
    """[C, Hb, W] band of this rank -> full [C, H, W] image on every rank (one all-gather of equally padded bands); the backward
    hands the rank the rows of the gradient that belong to its band."""

    @staticmethod
    def forward(ctx, band, bounds, group):
        world = dist.get_world_size(group); rank = dist.get_rank(group)
        C, Hb, W = band.shape
        hmax = max(y1 - y0 for y0, y1 in bounds)
        send = band.detach()
        if Hb < hmax:
            send = torch.cat([send, send.new_zeros((C, hmax - Hb, W))], dim=1)
        send = send.contiguous()
        recv = send.new_empty((world, C, hmax, W))
        try:
            dist.all_gather_into_tensor(recv, send, group=group)
        except (RuntimeError, NotImplementedError):
            dist.all_gather([recv[r] for r in range(world)], send, group=group)
        full = torch.cat([recv[r, :, :bounds[r][1] - bounds[r][0]] for r in range(world)], dim=1)
        ctx.rows = bounds[rank]
        return full

    @staticmethod
    def backward(ctx, g_full):
        y0, y1 = ctx.rows
        return g_full[:, y0:y1].contiguous(), None, None


def gather_bands(band: torch.Tensor, bounds: Sequence[tuple], group=None) -> torch.Tensor:
    """Assemble the full image from every rank's rows (bounds = band_bounds(H, world)); differentiable w.r.t. the own band."""
    return _GatherBands.apply(band, list(bounds), group)


# ------------------------------------------------------------------------------------------- halo-sharded loss
HALO = 32      # rows: >= 10 for the 11x11 SSIM window applied twice (S_q, then dS_q/dimg_p), >= 2 for the depth-to-normal stencil,
               # and a multiple of the loss kernels' 32-row blocks so that band-only partial sums can be picked out


def halo_rows(bounds: Sequence[tuple], rank: int, H: int, halo: int = HALO):
    y0, y1 = bounds[rank]
    return min(halo, y0), min(halo, H - y1)


class _ExchangeHalo(torch.autograd.Function):
    """[C, Hb, W] band of this rank -> [C, top + Hb + bottom, W]: the band plus up to `halo` rows of the neighbouring bands
    (point-to-point with the two neighbours; backends without device send/recv gather the edge strips instead).  The halo rows are
    inputs of THIS rank's loss terms only as context: every pixel's gradient is computed once, by the rank that owns the pixel
    (its own extended region holds every loss term the pixel takes part in), so the backward just hands the own rows on."""

    @staticmethod
    def forward(ctx, band, bounds, H, halo, group):
        world = dist.get_world_size(group); rank = dist.get_rank(group)
        C, Hb, W = band.shape
        top, bot = halo_rows(bounds, rank, H, halo)
        for r in range(world):
            if bounds[r][1] - bounds[r][0] < halo and world > 1:
                raise ValueError("band %d has %d rows, fewer than the %d-row halo" % (r, bounds[r][1] - bounds[r][0], halo))
        src = band.detach()
        ext = src.new_empty((C, top + Hb + bot, W))
        ext[:, top:top + Hb].copy_(src)
        send_up = src[:, :halo].contiguous() if rank > 0 else None               # my first rows -> bottom halo of rank - 1
        send_dn = src[:, Hb - halo:].contiguous() if rank < world - 1 else None  # my last rows  -> top halo of rank + 1
        recv_up = src.new_empty((C, top, W)) if top else None
        recv_dn = src.new_empty((C, bot, W)) if bot else None
        if dist.get_backend(group) == "nccl":
            ops = []
            if rank > 0:
                ops += [dist.P2POp(dist.isend, send_up, rank - 1, group), dist.P2POp(dist.irecv, recv_up, rank - 1, group)]
            if rank < world - 1:
                ops += [dist.P2POp(dist.isend, send_dn, rank + 1, group), dist.P2POp(dist.irecv, recv_dn, rank + 1, group)]
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
        else:           # rehearsal backends: all-gather of the (first, last) strips of every band
            z = src.new_zeros((C, halo, W))
            strips = torch.stack([send_up if send_up is not None else z, send_dn if send_dn is not None else z]).contiguous()
            allst = [torch.empty_like(strips) for _ in range(world)]
            dist.all_gather(allst, strips, group=group)
            if top:
                recv_up.copy_(allst[rank - 1][1])
            if bot:
                recv_dn.copy_(allst[rank + 1][0])
        if top:
            ext[:, :top].copy_(recv_up)
        if bot:
            ext[:, top + Hb:].copy_(recv_dn)
        ctx.rows = (top, top + Hb)
        return ext

    @staticmethod
    def backward(ctx, g_ext):
        a, b = ctx.rows
        return g_ext[:, a:b].contiguous(), None, None, None, None


def exchange_halo(band: torch.Tensor, bounds: Sequence[tuple], H: int, halo: int = HALO, group=None) -> torch.Tensor:
    """Band + neighbour rows for the halo-sharded loss (see _ExchangeHalo); differentiable w.r.t. the own band."""
    return _ExchangeHalo.apply(band, list(bounds), int(H), int(halo), group)


def halo_bytes(bounds: Sequence[tuple], rank: int, H: int, W: int, channels: int, halo: int = HALO) -> int:
    """Bytes this rank receives per halo exchange (= sends, by symmetry of interior bands)."""
    top, bot = halo_rows(bounds, rank, H, halo)
    return 4 * channels * W * (top + bot)


def wire_bytes_per_step(P: int, world: int, sharding: str, stats_live: bool, halo_b: int = 0) -> dict:
    """Bytes on the wire per GPU per training step (ring algorithms: all-reduce 2 (N-1)/N x payload, all-gather (N-1) x own
    payload), by collective.  views: all-gather of 12 B/surfel colour gradients + all-reduce of the 40 B/surfel geometry prefix;
    bands: ONE all-reduce of 52 B/surfel (+ 12 B/surfel while the densification statistic is live) + the halo strips."""
    f = 2.0 * (world - 1) / world
    if world <= 1:
        return {"total": 0}
    if sharding == "bands":
        d = {"all_reduce_per_surfel_B": 52 + (12 if stats_live else 0), "all_reduce": int(f * P * (52 + (12 if stats_live else 0))),
             "radii_max_all_reduce": int(f * P * 4) if stats_live else 0, "halo_p2p": int(halo_b), "loss_scalars": int(f * 16)}
    else:
        d = {"all_gather_colour": int((world - 1) * P * 12), "all_reduce_geometry": int(f * P * 40)}
    d["total"] = int(sum(v for k, v in d.items() if not k.endswith("_B")))
    d["per_surfel_B"] = round(d["total"] / P, 1)
    return d
