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
