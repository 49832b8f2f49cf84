"""Multi-process path on CPU (gloo, world_size 2 and 3): row-band sharding + halo exchange + gather.

The compute stand-in on CPU is the oracle (allowed: tests only); on GPUs bench.py --shard band runs the same
sharding code with the HIP kernels.  The sharded pipeline must reproduce the single-process result bit for bit:
flatten is per-pixel, and a band blurred with `radius` halo rows equals the same rows of the full-image blur."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from paintfe_amd import sharding as S

from . import inputs as I
from . import oracle_lib as O

W, H, NL, SIGMA = 96, 300, 5, 4.0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


FILTERS = {  # name -> (halo rows, filter on a numpy image): the three stencils whose vertical pass needs neighbours' rows
    "gaussian": (lambda: len(O.gaussian_kernel(SIGMA)) // 2, lambda img: O.gaussian_blur(img, SIGMA, threads=1)),
    "box": (lambda: 3, lambda img: O.box_blur(img, 2.5)),          # ceil(2.5) rows (blur.rs:241)
    "median": (lambda: 2, lambda img: O.median(img, 2)),
}


def _worker(rank, world, port, out_dir, filt="gaussian"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stack, modes, opac = I.layer_stack(W, H, NL, seed=11)
        radius = FILTERS[filt][0]()
        y0, y1 = S.band_rows(H, world, rank)
        if y1 > y0:
            flat_band = O.flatten_stack(np.ascontiguousarray(stack[:, y0:y1]), modes, opac, threads=1)
        else:
            flat_band = np.zeros((0, W, 4), np.uint8)
        padded, top, bottom = S.exchange_halo(torch.from_numpy(flat_band), H, radius)
        if y1 > y0:
            blurred = FILTERS[filt][1](padded.numpy())[top:top + (y1 - y0)]
        else:
            blurred = np.zeros((0, W, 4), np.uint8)
        full = S.gather_bands(torch.from_numpy(np.ascontiguousarray(blurred)), H).numpy()
        t = S.max_over_ranks(float(rank + 1))
        assert t == float(world)
        np.save(os.path.join(out_dir, f"r{rank}.npy"), full)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,filt", [(2, "gaussian"), (3, "gaussian"), (2, "box"), (3, "box"), (2, "median"), (3, "median")])
def test_band_sharded_pipeline_matches_single_process(tmp_path, world, filt):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), filt), nprocs=world, join=True)
    stack, modes, opac = I.layer_stack(W, H, NL, seed=11)
    ref = FILTERS[filt][1](O.flatten_stack(stack, modes, opac, threads=2))
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got.shape == ref.shape
        assert np.array_equal(got, ref), f"rank {r}: {(got != ref).any(-1).sum()} px differ"


class _CpuWarpRenderer:
    """CPU stand-in for the two band entry points S.warp_sharded calls (tests only): the oracle's whole-image warp, cut to the band"""

    @staticmethod
    def _np(ptr_owner):
        return ptr_owner

    def __init__(self, tensors):
        self.t = tensors  # data_ptr -> tensor (the stand-in cannot dereference raw pointers)

    def warp_displacement_band_dev(self, src_ptr, sw, sh, disp_ptr, w, band_rows, dst_ptr, first_row):
        src, disp, dst = self.t[src_ptr].numpy(), self.t[disp_ptr].numpy(), self.t[dst_ptr]
        full = np.zeros((sh, w, 2), np.float32)
        full[first_row:first_row + band_rows] = disp
        dst.copy_(torch.from_numpy(O.warp_displacement(src, full, threads=1)[first_row:first_row + band_rows].copy()))

    def warp_mesh_catmull_rom_band_dev(self, src_ptr, orig, deformed, cols, rows, w, h, dst_ptr, first_row, band_rows):
        src, dst = self.t[src_ptr].numpy(), self.t[dst_ptr]
        dst.copy_(torch.from_numpy(O.warp_mesh_catmull_rom(src, orig, deformed, cols, rows, threads=1)[first_row:first_row + band_rows].copy()))


class _Registry(dict):
    """tensors by data_ptr, filled lazily: warp_sharded allocates `source` and `out` itself, so the stand-in looks them up through torch's allocator"""

    def __missing__(self, ptr):
        raise KeyError(ptr)


def _warp_worker(rank, world, port, out_dir, kind):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stack, modes, opac = I.layer_stack(W, H, NL, seed=12)
        y0, y1 = S.band_rows(H, world, rank)
        flat_band = torch.from_numpy(O.flatten_stack(np.ascontiguousarray(stack[:, y0:y1]), modes, opac, threads=1) if y1 > y0 else np.zeros((0, W, 4), np.uint8))
        rng = np.random.default_rng(5)
        disp = (rng.standard_normal((H, W, 2)) * 6.0).astype(np.float32)
        orig, deformed = I.jittered_mesh(4, 5, W, H, seed=9)
        reg = _Registry()
        real_empty_like, real_gather = torch.empty_like, S.gather_bands

        def tracking_empty_like(t, *a, **k):
            o = real_empty_like(t, *a, **k); reg[o.data_ptr()] = o; return o

        def tracking_gather(*a, **k):
            o = real_gather(*a, **k).contiguous(); reg[o.data_ptr()] = o; return o

        torch.empty_like, S.gather_bands = tracking_empty_like, tracking_gather
        try:
            disp_band = torch.from_numpy(np.ascontiguousarray(disp[y0:y1]))
            reg[disp_band.data_ptr()] = disp_band
            r = _CpuWarpRenderer(reg)
            if kind == "displacement":
                band = S.warp_sharded(r, flat_band, H, "displacement", disp_band=disp_band)
            else:
                band = S.warp_sharded(r, flat_band, H, "mesh", mesh=(orig, deformed, 4, 5))
        finally:
            torch.empty_like, S.gather_bands = real_empty_like, real_gather
        full = S.gather_bands(band, H).numpy()
        np.save(os.path.join(out_dir, f"w{rank}.npy"), full)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "displacement"), (3, "displacement"), (2, "mesh"), (3, "mesh")])
def test_warp_of_a_sharded_document_matches_single_process(tmp_path, world, kind):
    """replicate the flattened source (all-gather of the bands), every rank warps its band of the output, the bands concatenate to
    the single-process warp — displacement field and Catmull-Rom mesh, ragged bands (300 rows over 2 and 3 ranks)"""
    port = _free_port()
    mp.spawn(_warp_worker, args=(world, port, str(tmp_path), kind), nprocs=world, join=True)
    stack, modes, opac = I.layer_stack(W, H, NL, seed=12)
    flat = O.flatten_stack(stack, modes, opac, threads=2)
    if kind == "displacement":
        rng = np.random.default_rng(5)
        ref = O.warp_displacement(flat, (rng.standard_normal((H, W, 2)) * 6.0).astype(np.float32))
    else:
        orig, deformed = I.jittered_mesh(4, 5, W, H, seed=9)
        ref = O.warp_mesh_catmull_rom(flat, orig, deformed, 4, 5)
    for r in range(world):
        got = np.load(tmp_path / f"w{r}.npy")
        assert np.array_equal(got, ref), f"rank {r}: {(got != ref).any(-1).sum()} px differ"


def test_band_partition_properties():
    for h in (1, 63, 64, 65, 300, 4320, 8640):
        for world in (1, 2, 3, 4, 8, 100):
            bands = S.all_bands(h, world)
            assert bands[0][0] == 0 and bands[-1][1] == h
            for (a0, a1), (b0, b1) in zip(bands, bands[1:]):
                assert a1 == b0 and a0 <= a1
            for (b0, b1) in bands:
                assert (b0 % 64 == 0 or b0 == h) and (b1 % 64 == 0 or b1 == h)
    # 8K at 8 GPUs: 68 chunk rows -> 9,9,9,9,8,8,8,8
    assert [(b - a) // 64 for a, b in S.all_bands(4320, 8)][:4] == [9, 9, 9, 9]


def test_halo_plan_covers_radius():
    h, world, r = 4320, 8, 48
    for rank in range(world):
        y0, y1 = S.band_rows(h, world, rank)
        rows = set()
        for (src, s0, s1) in S.halo_plan(h, world, rank, r):
            b0, b1 = S.band_rows(h, world, src)
            assert b0 <= s0 < s1 <= b1
            rows.update(range(s0, s1))
        want = set(range(max(0, y0 - r), y0)) | set(range(y1, min(h, y1 + r)))
        assert rows == want


class _CpuBandRenderer:
    """CPU stand-in for the two entry points BandPipeline.step calls (tests only): the oracle on the host buffers behind the raw pointers"""

    @staticmethod
    def _view(ptr, rows, w):
        import ctypes
        return np.ctypeslib.as_array((ctypes.c_uint8 * (rows * w * 4)).from_address(ptr)).reshape(rows, w, 4)

    def flatten_dev(self, ptrs, info, w, rows, dst_ptr):
        stack = np.stack([self._view(p, rows, w) for p in ptrs])
        modes = np.array([i[3] for i in info], np.uint8)
        opac = np.array([i[1] for i in info], np.float32)
        self._view(dst_ptr, rows, w)[...] = O.flatten_stack(stack, modes, opac, threads=1)

    def gaussian_blur_dev(self, src_ptr, dst_ptr, w, prow, sigma, first_row=0):
        # a band with its halo rows: rows at least `radius` from the buffer's ends equal the whole image's (clamp-to-edge only matters at true image edges)
        self._view(dst_ptr, prow, w)[...] = O.gaussian_blur(self._view(src_ptr, prow, w).copy(), sigma, threads=1)


def _pipeline_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        radius = len(O.gaussian_kernel(SIGMA)) // 2
        y0, y1 = S.band_rows(H, world, rank)
        docs = [I.layer_stack(W, H, NL, seed=21 + d) for d in range(3)]
        bands = [torch.from_numpy(np.ascontiguousarray(st[:, y0:y1])) for (st, _, _) in docs]
        infos = [[(k, float(op[k]), True, int(md[k])) for k in range(NL)] for (_, md, op) in docs]
        # (a) pipelined, result left sharded: step k + 1 hands back document k's band, finish() the last one
        pipe = S.BandPipeline(_CpuBandRenderer(), W, H, radius, SIGMA, "cpu", gather=False, pipelined=True)
        got = []
        for b, info in zip(bands, infos):
            res = pipe.step([b[k].data_ptr() for k in range(NL)], info)
            got.append(None if res is None else res.numpy().copy())
        pipe.finish()
        got.append(pipe.last_result.numpy().copy())
        assert got[0] is None
        np.save(os.path.join(out_dir, f"p{rank}.npy"), np.stack(got[1:]))
        # (b) unpipelined with the all-gather, and the edge-first split
        pipe2 = S.BandPipeline(_CpuBandRenderer(), W, H, radius, SIGMA, "cpu", gather=True, split_edges=(rank % 2 == 0))
        fulls = []
        for b, info in zip(bands, infos):
            pipe2.step([b[k].data_ptr() for k in range(NL)], info)
            fulls.append(pipe2.assemble().numpy().copy())
        np.save(os.path.join(out_dir, f"g{rank}.npy"), np.stack(fulls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_band_pipeline_class_pipelined_and_gathered(tmp_path, world):
    """paintfe_amd.sharding.BandPipeline itself over gloo on CPU (the oracle behind the renderer's two entry points): the pipelined form bench.py --gpus N times
    (a step's halo rows travel under the next step's flatten; results arrive one step late) and the gathered form, three different documents back to back"""
    port = _free_port()
    mp.spawn(_pipeline_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    refs = []
    for d in range(3):
        stack, modes, opac = I.layer_stack(W, H, NL, seed=21 + d)
        refs.append(O.gaussian_blur(O.flatten_stack(stack, modes, opac, threads=2), SIGMA, threads=2))
    for r in range(world):
        y0, y1 = S.band_rows(H, world, r)
        p, g = np.load(tmp_path / f"p{r}.npy"), np.load(tmp_path / f"g{r}.npy")
        for d in range(3):
            assert np.array_equal(p[d], refs[d][y0:y1]), f"rank {r} document {d}: pipelined band differs"
            assert np.array_equal(g[d], refs[d]), f"rank {r} document {d}: gathered frame differs"
