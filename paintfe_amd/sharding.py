"""Row-band sharding of one document across ranks (one process per GPU, torch.distributed over RCCL/xGMI).

The reference has no distributed layer at all; its natural unit of independence is the 64x64
TiledImage chunk (ref: src/canvas/canvas_state.rs:565 parallelises the compositor over chunks), so a document is cut
into bands of whole chunk rows:

  * flatten and every pointwise op are per-pixel: a band needs nothing from its neighbours;
  * a separable stencil of radius r needs r rows of the *input of the stencil* from each neighbour before its
    vertical pass (the horizontal pass is row-local).  For "flatten -> Gaussian" that input is the flattened u8 band:
    r x w x 4 bytes per neighbour and direction (sigma=16 at 8K: 48 x 7680 x 4 = 1.47 MB), one point-to-point
    message each way per neighbour pair — the only collective-free exchange the path has;
  * the final image is the concatenation of the bands (all_gather only if one rank needs the whole picture).

All functions take plain ``torch`` tensors, so the same code runs on CPU tensors over gloo (tests, world_size 2) and
on device tensors over RCCL (bench.py --shard band).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

CHUNK = 64  # ref: src/canvas/defs.rs:7


def band_rows(height: int, world: int, rank: int) -> Tuple[int, int]:
    """[y0, y1) of rank's band: whole chunk rows, remainder spread over the first ranks.  Empty bands are possible
    when there are fewer chunk rows than ranks."""
    chunk_rows = (height + CHUNK - 1) // CHUNK
    base, rem = divmod(chunk_rows, world)
    c0 = rank * base + min(rank, rem)
    c1 = c0 + base + (1 if rank < rem else 0)
    return min(c0 * CHUNK, height), min(c1 * CHUNK, height)


def all_bands(height: int, world: int) -> List[Tuple[int, int]]:
    return [band_rows(height, world, r) for r in range(world)]


def halo_plan(height: int, world: int, rank: int, radius: int):
    """What rank must receive: list of (src_rank, src_y0, src_y1) row ranges above and below its band, at most
    `radius` rows each, possibly spanning several (thin) neighbouring bands."""
    y0, y1 = band_rows(height, world, rank)
    need = []
    if y1 > y0:
        lo0, lo1 = max(0, y0 - radius), y0
        hi0, hi1 = y1, min(height, y1 + radius)
        for r, (b0, b1) in enumerate(all_bands(height, world)):
            if r == rank or b1 <= b0:
                continue
            for (a, b) in ((lo0, lo1), (hi0, hi1)):
                s0, s1 = max(a, b0), min(b, b1)
                if s1 > s0:
                    need.append((r, s0, s1))
    return sorted(need, key=lambda t: t[1])


def exchange_halo(band, height: int, radius: int, group=None):
    """band: (rows, w, C) tensor holding rows [y0, y1) of this rank.  Returns (padded, top, bottom) where padded holds
    rows [y0 - top, y1 + bottom) of the full image.  Uses batched point-to-point ops (RCCL send/recv on GPUs)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    y0, y1 = band_rows(height, world, rank)
    my_need = halo_plan(height, world, rank, radius)
    out_device = band.device
    if band.device.type == "cuda" and dist.get_backend(group) == "gloo":
        band = band.cpu()  # plumbing self-test on a single-GPU box: gloo moves host memory; RCCL moves device memory
    ops, recv_bufs = [], []
    for (src, s0, s1) in my_need:
        buf = torch.empty((s1 - s0,) + tuple(band.shape[1:]), dtype=band.dtype, device=band.device)
        recv_bufs.append((s0, s1, buf))
        ops.append(dist.P2POp(dist.irecv, buf, src, group))
    keep = []
    for other in range(world):
        if other == rank:
            continue
        for (src, s0, s1) in halo_plan(height, world, other, radius):
            if src == rank:
                piece = band[s0 - y0:s1 - y0].contiguous()
                keep.append(piece)
                ops.append(dist.P2POp(dist.isend, piece, other, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    above = [b for (s0, s1, b) in recv_bufs if s1 <= y0]
    below = [b for (s0, s1, b) in recv_bufs if s0 >= y1]
    top = sum(int(b.shape[0]) for b in above)
    bottom = sum(int(b.shape[0]) for b in below)
    padded = torch.cat(above + [band] + below, dim=0) if (above or below) else band
    return padded.to(out_device), top, bottom


def gather_bands(band, height: int, group=None):
    """all_gather of the (ragged) bands into the full image on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    bands = all_bands(height, world)
    max_rows = max(b1 - b0 for b0, b1 in bands)
    pad = torch.zeros((max_rows,) + tuple(band.shape[1:]), dtype=band.dtype, device=band.device)
    pad[:band.shape[0]] = band
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    return torch.cat([o[:b1 - b0] for o, (b0, b1) in zip(outs, bands)], dim=0)


def warp_sharded(renderer, flat_band, height: int, kind: str, disp_band=None, mesh=None, group=None):
    """Warp of a document sharded by rows, one process per GPU ("replicate the source, warp bands of the output"; ref:
    src/ops/transform.rs:1288-1345, 1687-1761): `flat_band` = this rank's flattened band (rows, w, 4) on the device.  The bands are all-gathered so
    that every rank holds the whole source, then the rank warps ITS band of the output with the band entry points of the C ABI — no halo, because a
    displacement may reach anywhere.  kind "displacement": disp_band = (rows, w, 2) float32 device tensor; kind "mesh": mesh = (orig or None, deformed,
    cols, rows).  Returns the rank's band of the result; gather_bands() assembles the whole image where one consumer wants it."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    y0, y1 = band_rows(height, world, rank)
    w = flat_band.shape[1]
    source = gather_bands(flat_band, height, group=group).contiguous()
    out = torch.empty_like(flat_band)
    if y1 > y0:
        if kind == "displacement":
            renderer.warp_displacement_band_dev(source.data_ptr(), w, height, disp_band.data_ptr(), w, y1 - y0, out.data_ptr(), y0)
        elif kind == "mesh":
            orig, deformed, cols, rows = mesh
            renderer.warp_mesh_catmull_rom_band_dev(source.data_ptr(), orig, deformed, cols, rows, w, height, out.data_ptr(), y0, y1 - y0)
        else:
            raise ValueError(kind)
    return out


class BandPipeline:
    """flatten -> halo exchange -> Gaussian (or box blur / median) -> all-gather of ONE document on this rank's band, buffers allocated once.

    Everything is enqueued on torch's current stream (the constructor points the renderer at it): the flatten writes straight into the centre of [top halo | band | bottom halo]
    (one launch; `split_edges` flattens the edge rows first so that the halo exchange overlaps the interior, `pipelined` overlaps it with the NEXT step's flatten instead), the halo rows arrive in place through one
    batched RCCL send/recv group (no concatenation, no host synchronisation), the blur
    runs on band + halo with its tiles on the whole image's grid (pfx_gaussian_blur_band_dev: results equal the single-GPU ones bit
    for bit), and the result bands are all-gathered into every rank.  Over RCCL the all-gather is asynchronous and double-buffered:
    step k's gather runs on RCCL's stream while step k + 1's flatten runs on the compute stream; a buffer set is only waited for
    when it comes round again (two steps later) or in finish().  The gather moves (N - 1) / N of the image into every GPU — at 8
    GPUs more time than a rank's kernels — so back-to-back documents are paced by max(kernels, gather), not by their sum."""

    SETS = 2

    def __init__(self, renderer, w: int, h: int, radius: int, sigma: float, device, gather: bool = True, group=None, filter: str = "gaussian",
                 split_edges: bool = False, pipelined: bool = False):
        """filter: "gaussian" (sigma; radius = ceil(3 sigma)), "box" (sigma carries the box radius; radius = ceil of it) or "median"
        (sigma carries the radius) — the three stencils whose vertical pass needs halo rows"""
        import torch
        import torch.distributed as dist

        assert filter in ("gaussian", "box", "median")
        self.filter = filter
        # split_edges: flatten the band's edge chunk rows first so that the halo exchange overlaps the interior's flatten.  OFF by default since round 4:
        # a 64-row flatten launch is a chain of 32 dependent layer steps on a nearly empty chip (~0.06 ms each at 8K) — the three launches cost 0.12-0.17 ms
        # more than one (tools/lab/band_step_cost.py: 8 GPUs' band 0.360 against 0.192 ms), the exchange they hide ~0.05.
        # pipelined (result left sharded only): step k's halo exchange travels while step k + 1 is flattened into a second padded buffer; step() then
        # returns the PREVIOUS step's blurred band (None on the first call) and finish() flushes the last one — back-to-back documents at
        # flatten + blur per step with no exposed exchange.
        self.split_edges, self.pipelined = split_edges, pipelined
        self.inflight, self.last_result = None, None
        self.r, self.w, self.h, self.radius, self.sigma, self.group, self.gather = renderer, w, h, radius, sigma, group, gather
        # every kernel of a step and RCCL's ordering (req.wait(), the asynchronous all-gather) are relative to torch's CURRENT stream:
        # the renderer must launch there too, or the halo exchange races the flatten and the blur races the halo's arrival
        if getattr(device, "type", str(device)) == "cuda" or str(device).startswith("cuda"):
            renderer.set_stream(torch.cuda.current_stream(device).cuda_stream)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.y0, self.y1 = band_rows(h, self.world, self.rank)
        self.rows = self.y1 - self.y0
        self.top = min(radius, self.y0) if self.rows else 0
        self.bottom = min(radius, h - self.y1) if self.rows else 0
        prow = max(self.top + self.rows + self.bottom, 1)
        self.bands = all_bands(h, self.world)
        self.max_rows = max(max(b1 - b0 for b0, b1 in self.bands), 1)
        self.padded = torch.empty((prow, w, 4), dtype=torch.uint8, device=device)
        self.padded_sets = [self.padded, torch.empty_like(self.padded)] if pipelined else [self.padded]
        n_sets = self.SETS if (gather or pipelined) else 1
        # the blur output doubles as the all-gather's send buffer: rows [top, top + max_rows) — bands are ragged by at most one chunk
        # row, so the buffer carries that much slack and the gather needs no staging copy
        self.blurred = [torch.zeros((self.top + self.max_rows + radius + 1, w, 4), dtype=torch.uint8, device=device) for _ in range(n_sets)]
        # all-gather target: one equal-size slot per rank (slot k holds band k in its first rows); `assemble()` makes the contiguous image
        self.slots = [torch.empty((self.world, self.max_rows, w, 4), dtype=torch.uint8, device=device) for _ in range(n_sets)] if gather else None
        self.pending = [None] * n_sets  # the asynchronous all-gather that is reading blurred[s] / writing slots[s]
        self.turn, self.last = 0, 0
        self.full = None
        # static exchange plan: (what I receive, what I send)
        lo0 = self.y0 - self.top
        self.recvs = [(src, s0 - lo0, s1 - lo0) for (src, s0, s1) in halo_plan(h, self.world, self.rank, radius)]
        self.sends = []
        for other in range(self.world):
            if other == self.rank:
                continue
            for (src, s0, s1) in halo_plan(h, self.world, other, radius):
                if src == self.rank:
                    self.sends.append((other, self.top + s0 - self.y0, self.top + s1 - self.y0))

    def flat_band(self):
        return self.padded[self.top:self.top + self.rows]

    def _exchange_via_host(self):
        import torch
        import torch.distributed as dist

        bufs = [(a, b, torch.empty((b - a, self.w, 4), dtype=torch.uint8)) for (_, a, b) in self.recvs]
        ops = [dist.P2POp(dist.irecv, buf, src, self.group) for (src, _, _), (_, _, buf) in zip(self.recvs, bufs)]
        keep = [self.padded[a:b].cpu() for (_, a, b) in self.sends]
        ops += [dist.P2POp(dist.isend, piece, dst, self.group) for (dst, _, _), piece in zip(self.sends, keep)]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for (a, b, buf) in bufs:
            self.padded[a:b].copy_(buf)

    def _start_exchange(self, pad):
        """halo rows of `pad`: one batched send/recv group, ordered behind what the current stream holds so far; returns the requests to wait for"""
        import torch.distributed as dist

        if dist.get_backend(self.group) == "gloo" and pad.device.type == "cuda":
            keep, self.padded = self.padded, pad
            self._exchange_via_host()  # plumbing self-test on a 1-GPU box (gloo moves host memory); RCCL moves device memory
            self.padded = keep
            return []
        ops = [dist.P2POp(dist.irecv, pad[a:b], src, self.group) for (src, a, b) in self.recvs]
        ops += [dist.P2POp(dist.isend, pad[a:b], dst, self.group) for (dst, a, b) in self.sends]
        return dist.batch_isend_irecv(ops) if ops else []

    def _filter(self, pad, blurred):
        prow = self.top + self.rows + self.bottom
        first = self.y0 - self.top
        if self.filter == "gaussian":
            self.r.gaussian_blur_dev(pad.data_ptr(), blurred.data_ptr(), self.w, prow, self.sigma, first_row=first)
        elif self.filter == "box":
            self.r.box_blur_band_dev(pad.data_ptr(), blurred.data_ptr(), self.w, prow, self.sigma, first_row=first)
        else:
            self.r.median_band_dev(pad.data_ptr(), blurred.data_ptr(), self.w, prow, int(self.sigma), first_row=first)

    def _complete_inflight(self):
        if self.inflight is None:
            return None
        s, reqs = self.inflight
        self.inflight = None
        for req in reqs:
            req.wait()  # the compute stream waits for the halo rows; the host does not
        if self.pending[s] is not None:
            self.pending[s].wait()  # an all-gather of an earlier (gathered) step that still reads this output buffer
            self.pending[s] = None
        if self.rows:
            self._filter(self.padded_sets[s], self.blurred[s])
        self.last_result = self.blurred[s][self.top:self.top + self.rows]
        return self.last_result

    def _step_pipelined(self, layer_ptrs, info):
        s = self.turn % 2
        self.turn += 1
        pad = self.padded_sets[s]
        self.padded = pad  # flat_band() = the band flattened last
        if self.rows:
            self.r.flatten_dev(list(layer_ptrs), info, self.w, self.rows, pad[self.top:].data_ptr())
        reqs = self._start_exchange(pad)     # travels while the previous step is blurred and the next one flattened
        done = self._complete_inflight()     # the previous step: its halo rows have had a whole flatten to arrive
        self.inflight = (s, reqs)
        return done

    def step(self, layer_ptrs, info):
        import torch.distributed as dist

        if self.pipelined and not self.gather:
            return self._step_pipelined(layer_ptrs, info)
        self._complete_inflight()  # a pipelined step left over from before the mode changed
        self.padded = self.padded_sets[0]
        s = self.turn % len(self.blurred)
        self.turn += 1
        row_bytes = self.w * 4
        base = self.padded[self.top:].data_ptr()

        def flatten_rows(r0, r1):  # rows [r0, r1) of the band: every layer's band buffer is contiguous, so a row range is a pointer offset
            if r1 > r0:
                self.r.flatten_dev([p + r0 * row_bytes for p in layer_ptrs], info, self.w, r1 - r0, base + r0 * row_bytes)

        # split_edges: the rows the neighbours need (whole chunk rows covering `radius` at either edge of the band) are flattened first, the halo
        # exchange starts on RCCL's stream, and the interior of the band is flattened while the halo rows travel
        edge = min(self.rows, 64 * ((self.radius + 63) // 64))
        split = self.split_edges and self.rows > 2 * edge and bool(self.sends or self.recvs)
        if split:
            flatten_rows(0, edge)
            flatten_rows(self.rows - edge, self.rows)
        else:
            flatten_rows(0, self.rows)
        reqs = self._start_exchange(self.padded)  # ordered behind the flatten so far (the current stream at this point)
        if split:
            flatten_rows(edge, self.rows - edge)
        for req in reqs:
            req.wait()  # RCCL: orders the current stream behind the transfer, does not block the host
        if self.pending[s] is not None:
            self.pending[s].wait()  # this set's previous all-gather (two steps ago): the compute stream waits, the host does not
            self.pending[s] = None
        blurred = self.blurred[s]
        if self.rows:
            self._filter(self.padded, blurred)
        if not self.gather:
            self.last_result = blurred[self.top:self.top + self.rows]
            return self.last_result
        send = blurred[self.top:self.top + self.max_rows]
        self.last = s
        if dist.get_backend(self.group) == "gloo" and send.device.type == "cuda":
            outs = [send.cpu() for _ in range(self.world)]
            dist.all_gather(outs, send.cpu(), group=self.group)
            for k in range(self.world):
                self.slots[s][k].copy_(outs[k])
        elif dist.get_backend(self.group) == "gloo":
            dist.all_gather(list(self.slots[s].unbind(0)), send, group=self.group)  # gloo's all_gather_into_tensor refuses the stacked (world, rows, w, 4) layout RCCL takes
        else:
            self.pending[s] = dist.all_gather_into_tensor(self.slots[s], send, group=self.group, async_op=True)
        return self.slots[s]

    def phase_probe(self, layer_ptrs, info):
        """ONE unpipelined step with events around its phases (a collective: every rank calls it): ms of flatten, of the halo exchange as this rank's stream sees it
        (send / receive + waiting for the neighbours' flattens), of the band filter and — over RCCL, when the pipeline gathers — of the all-gather.  What a first run
        on a real multi-GPU node should print per rank beside the step time (bench.py: ranks.phase_ms_per_rank)."""
        import torch
        import torch.distributed as dist

        self.finish()
        pad, blurred = self.padded_sets[0], self.blurred[0]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        torch.cuda.synchronize()
        dist.barrier(group=self.group)
        ev[0].record()
        if self.rows:
            self.r.flatten_dev(list(layer_ptrs), info, self.w, self.rows, pad[self.top:].data_ptr())
        ev[1].record()
        for req in self._start_exchange(pad):
            req.wait()
        ev[2].record()
        if self.rows:
            self._filter(pad, blurred)
        ev[3].record()
        if self.gather and dist.get_backend(self.group) != "gloo":
            dist.all_gather_into_tensor(self.slots[0], blurred[self.top:self.top + self.max_rows], group=self.group, async_op=True).wait()
        ev[4].record()
        torch.cuda.synchronize()
        names = ("flatten", "halo_exchange", "filter", "gather")
        return {k: round(ev[i].elapsed_time(ev[i + 1]), 4) for i, k in enumerate(names)}

    def finish(self):
        """order the current stream behind every all-gather still in flight and flush a pipelined step (the host does not block)"""
        self._complete_inflight()
        for s, work in enumerate(self.pending):
            if work is not None:
                work.wait()
                self.pending[s] = None

    def assemble(self):
        """the last step's gathered bands as one contiguous h x w x 4 image (not part of the timed step: every rank already holds every band)"""
        import torch

        self.finish()
        slots = self.slots[self.last]
        if self.full is None:
            self.full = torch.empty((self.h, self.w, 4), dtype=torch.uint8, device=slots.device)
        for k, (b0, b1) in enumerate(self.bands):
            if b1 > b0:
                self.full[b0:b1] = slots[k, :b1 - b0]
        return self.full


def max_over_ranks(value: float, device=None, group=None) -> float:
    """wall time of a step = slowest rank (bench.py contract)"""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return value
    if dist.get_backend(group) == "gloo":
        device = None
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
