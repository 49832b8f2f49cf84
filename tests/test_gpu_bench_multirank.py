"""bench.py's multi-rank paths, end to end on the (single) GPU of the test box: two ranks launched exactly as the driver
launches them (`python -m torch.distributed.run --nproc-per-node 2 ...`), sharing device 0 over gloo
(PFX_BENCH_BACKEND=gloo; RCCL refuses two ranks per device).  Checks the JSON contract and — in band mode — that a
rank's band of the sharded flatten -> halo exchange -> Gaussian equals the single-process oracle result."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(nproc, extra, port, height=400):
    env = dict(os.environ, PFX_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3", "--warmup", "1",
           "--width", "640", "--height", str(height), "--layers", "6", "--sigma", "3.0", "--no-cpu-baseline"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, p.stdout[-2000:] + p.stderr[-2000:]
    return json.loads(lines[0])


def test_doc_mode_two_ranks_contract():
    d = _run(2, ["--shard", "doc"], 29621)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3 and d["warmup"] == 1
    assert d["unit"] == "Mpixels/s" and d["higher_is_better"] is True and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - 2 * 640 * 400 / d["ms_per_step"] / 1e3) / d["value"] < 0.01  # whole-job aggregate over both ranks
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"]
    assert d["check"]["flatten_crop_bitexact"] is True


@pytest.mark.parametrize("nproc", [2, 3])
def test_band_mode_is_the_default_and_matches_single_process(nproc):
    d = _run(nproc, ["--exact"], 29630 + nproc)  # no --shard: the driver's command line
    assert d["scaling"] == "strong" and d["n_gpus"] == nproc and "stays sharded" in d["config"]["sharding"]
    assert d["ranks"]["world_size"] == nproc and len(d["ranks"]["per_rank_ms_per_step"]) == nproc and all(t > 0 for t in d["ranks"]["per_rank_ms_per_step"])
    assert d["check"]["band_blur_max_diff_vs_oracle"] == 0  # exact Gaussian: bit-identical to the unsharded pipeline
    ph = d["ranks"]["phase_ms_per_rank"]                     # one unpipelined step's phases per rank (BandPipeline.phase_probe)
    assert len(ph) == nproc and all(p["flatten"] > 0 and p["filter"] > 0 and p["halo_exchange"] >= 0 for p in ph)
    assert abs(d["value"] - 640 * 400 / d["ms_per_step"] / 1e3) / d["value"] < 0.01  # ONE document per step for the whole job
    assert d["doc_mode"]["scaling"] == "weak" and d["doc_mode"]["value"] > 0
    # the headline leaves the blurred result sharded (halo exchange only); the same pipeline + an all-gather into every rank is timed beside it
    assert d["band_gathered_result"]["scaling"] == "strong" and d["band_gathered_result"]["value"] > 0 and "band_gathered_error" not in d
    # the timed headline's own band is always checked, and the gathered variant's window beside it
    assert d["check"]["band_blur_checked_rows"].startswith("own band") and d["check"]["gathered_window_max_diff_vs_oracle"] == 0
    # the C-ABI single-process leg over the same ranks' devices (here: the one GPU twice / three times), bit-identical under every variant
    ga = d["c_abi_group"]
    assert ga["members"] == nproc and ga["check"]["window_max_diff_vs_oracle"] == 0
    assert ga["peer"]["sharded"]["value"] > 0 and ga["peer"]["gathered"]["identical_to_first_variant"] is True
    assert len(ga["peer"]["sharded"]["phase_ms_per_member"]) == nproc and all(p["flatten"] > 0 and p["filter"] > 0 for p in ga["peer"]["sharded"]["phase_ms_per_member"])
    d = _run(nproc, [], 29640 + nproc)
    assert d["check"]["band_blur_max_diff_vs_oracle"] <= 1  # matrix-core Gaussian: the stated +-1 LSB


def test_band_mode_without_the_gathered_variant_checks_the_rank_own_band():
    """--no-gather: the headline pipeline alone; rank 0's band (its edge rows depend on the received halo rows) against the oracle"""
    d = _run(2, ["--exact", "--no-gather"], 29648)
    assert d["scaling"] == "strong" and "band_gathered_result" not in d
    assert d["check"]["band_blur_checked_rows"].startswith("own band") and d["check"]["band_blur_max_diff_vs_oracle"] == 0
    assert abs(d["value"] - 640 * 400 / d["ms_per_step"] / 1e3) / d["value"] < 0.01


def test_failed_band_pipeline_still_reports_the_collective_free_mode():
    """VERDICT r03 #5b: when the band pipeline (the only timed region with collectives) fails, the line still carries `doc_mode` — a partial scaling
    curve survives — with a null headline value, the error on the line and a non-zero exit code"""
    env = dict(os.environ, PFX_BENCH_BACKEND="gloo", PFX_BENCH_TEST_FAIL_BAND="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29661",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--width", "640", "--height", "400", "--layers", "6", "--sigma", "3.0",
           "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode != 0 and len(lines) == 1, p.stdout[-2000:] + p.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["value"] is None and "band_pipeline" in d["failed_checks"] and "injected" in d["band_pipeline_error"]
    assert d["doc_mode"]["value"] > 0 and d["doc_mode"]["scaling"] == "weak"


def test_gpus_flag_without_that_many_ranks_fails_loudly():
    """`--gpus 2` started as a single process must refuse to print a number (the driver launches N ranks; a job that silently ran on
    fewer would report a wrong N-GPU value)"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--width", "256", "--height", "128",
                        "--layers", "3", "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_band_mode_with_split_flatten_matches_single_process():
    """bands taller than two edge regions: the edge rows are flattened first, the halo exchange starts, the interior follows (the overlap
    itself needs RCCL; here the ordering and the row-range pointer arithmetic are checked bit for bit)"""
    d = _run(2, ["--exact", "--band-split-edges"], 29655, height=900)
    assert d["check"]["band_blur_max_diff_vs_oracle"] == 0  # flatten + halo rows + exact Gaussian of rank 0's window, bit for bit
    assert abs(d["value"] - 640 * 900 / d["ms_per_step"] / 1e3) / d["value"] < 0.01


def test_band_mode_over_rccl_when_two_gpus_are_visible():
    """the real thing: one rank per GPU over RCCL (backend "nccl"), halo rows by send/recv over xGMI (headline), then the all-gathered variant.
    Needs two visible devices; the driver's 8-GPU run uses exactly this command line."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 visible GPUs (RCCL refuses two ranks on one device)")
    env = {k: v for k, v in os.environ.items() if k != "PFX_BENCH_BACKEND"}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29671",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--width", "1024", "--height", "640", "--layers", "6",
           "--sigma", "8.0", "--no-cpu-baseline", "--exact"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, p.stdout[-2000:] + p.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["scaling"] == "strong" and d["n_gpus"] == 2 and d["check"]["band_blur_max_diff_vs_oracle"] == 0


def test_group_api_on_two_devices_when_visible():
    """pfx_group_* with one member per physical device (peer copies over xGMI instead of device-to-device copies)"""
    import numpy as np
    from paintfe_amd import GpuRenderer, _lib as L
    from paintfe_amd.group import GpuGroup
    from tests import inputs as I
    if L.load().pfx_device_count() < 2:
        pytest.skip("needs >= 2 visible GPUs")
    w, h, n = 512, 450, 5
    stack, modes, opac = I.layer_stack(w, h, n, seed=77)
    infos = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
    r = GpuRenderer(0)
    for k in range(n):
        r.ensure_layer_texture(k, stack[k], generation=1)
    ref = r.blur_rgba(r.composite(w, h, infos), 6.0)
    g = GpuGroup([0, 1])
    g.set_document(w, h, n)
    for k in range(n):
        g.upload_layer(k, stack[k])
    g.flatten_blur(infos, 6.0, all_gather=True)
    assert np.array_equal(g.download(), ref) and np.array_equal(g.download_gathered(1), ref)


_RCCL_ONE_RANK = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["PFX_ROOT"])
from paintfe_amd import GpuRenderer, sharding as S
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:PORT", rank=0, world_size=1)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h, n, sigma = 320, 200, 4, 3.0
g = torch.Generator(device=dev); g.manual_seed(5)
docs = [torch.randint(0, 256, (n, h, w, 4), dtype=torch.uint8, device=dev, generator=g) for _ in range(3)]
info = [(k, 1.0 if k % 2 == 0 else 0.6, True, [0, 1, 8, 13][k]) for k in range(n)]
pipe = S.BandPipeline(r, w, h, 9, sigma, dev)
outs = []
for d in docs:   # three different documents back to back: the asynchronous gathers of steps k and k + 1 use different buffer sets
    res = pipe.step([d[k].data_ptr() for k in range(n)], info)
    outs.append(res)
got = [None] * 3
got[2] = pipe.assemble().cpu().numpy().copy()
flat = torch.empty((h, w, 4), dtype=torch.uint8, device=dev); blur = torch.empty_like(flat)
ok = True
for k, d in enumerate(docs):
    r.flatten_dev([d[q].data_ptr() for q in range(n)], info, w, h, flat.data_ptr())
    r.gaussian_blur_dev(flat.data_ptr(), blur.data_ptr(), w, h, sigma)
    torch.cuda.synchronize()
    if k == 2:
        ok = ok and np.array_equal(got[2], blur.cpu().numpy())
# step 1's gather went to buffer set 1, step 2's to set 0 (overwriting step 0's): set 1 still holds document 1
r.flatten_dev([docs[1][q].data_ptr() for q in range(n)], info, w, h, flat.data_ptr())
r.gaussian_blur_dev(flat.data_ptr(), blur.data_ptr(), w, h, sigma)
torch.cuda.synchronize()
ok = ok and np.array_equal(pipe.slots[1][0, :h].cpu().numpy(), blur.cpu().numpy())
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK" if ok else "RCCL_ONE_RANK_MISMATCH")
'''


def test_band_pipeline_over_rccl_single_rank():
    """the RCCL code path of BandPipeline (asynchronous, double-buffered all_gather_into_tensor; batched send/recv group) with the one
    rank a 1-GPU box allows: API usage, stream ordering and the buffer-set rotation; results bit-identical to the plain pipeline"""
    env = {k: v for k, v in os.environ.items() if k != "PFX_BENCH_BACKEND"}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["PFX_ROOT"] = ROOT
    p = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK.replace("PORT", "29683")], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert "RCCL_ONE_RANK_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


_PIPELINED_TWO_RANKS = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["PFX_ROOT"])
from paintfe_amd import GpuRenderer, sharding as S
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h, n, sigma, radius = 320, 450, 4, 3.0, 9
g = torch.Generator(device=dev); g.manual_seed(5)     # every rank draws the same documents
docs = [torch.randint(0, 256, (n, h, w, 4), dtype=torch.uint8, device=dev, generator=g) for _ in range(4)]
info = [(k, 1.0 if k % 2 == 0 else 0.6, True, [0, 1, 8, 13][k]) for k in range(n)]
y0, y1 = S.band_rows(h, world, rank)
pipe = S.BandPipeline(r, w, h, radius, sigma, dev, gather=False, pipelined=True)
got = []
for d in docs:   # different documents back to back: a stale padded buffer, halo row or output set would show
    band = d[:, y0:y1].contiguous()
    res = pipe.step([band[k].data_ptr() for k in range(n)], info)
    got.append(None if res is None else res.cpu().numpy().copy())
    torch.cuda.synchronize()
pipe.finish()
got.append(pipe.last_result.cpu().numpy().copy())
ok = got[0] is None and len(got) == 5
flat = torch.empty((h, w, 4), dtype=torch.uint8, device=dev); blur = torch.empty_like(flat)
for k, d in enumerate(docs):
    r.flatten_dev([d[q].data_ptr() for q in range(n)], info, w, h, flat.data_ptr())
    r.gaussian_blur_dev(flat.data_ptr(), blur.data_ptr(), w, h, sigma)
    torch.cuda.synchronize()
    ok = ok and np.array_equal(got[k + 1], blur[y0:y1].cpu().numpy())   # step k + 1 returns document k's band
verdict = torch.tensor([1 if ok else 0])
dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
if rank == 0:
    print("PIPELINED_OK" if int(verdict) == 1 else "PIPELINED_MISMATCH")
dist.destroy_process_group()
'''


@pytest.mark.parametrize("nproc", [2, 3])
def test_pipelined_band_steps_return_the_previous_document(nproc, tmp_path):
    """BandPipeline(pipelined=True): step k's halo exchange travels under step k + 1's flatten (two padded buffers); step() hands back the
    PREVIOUS document's blurred band and finish() the last — every band equal to the same rows of the single-process result, bit for bit"""
    script = tmp_path / "pipelined.py"
    script.write_text(_PIPELINED_TWO_RANKS)
    env = dict(os.environ, PFX_ROOT=ROOT)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
                        "--master-port", str(29690 + nproc), str(script)], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert "PIPELINED_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_single_rank_line_carries_live_hbm_traffic():
    """N = 1: roofline.traffic comes from a FETCH_SIZE and a WRITE_SIZE pass collected by the run itself (two rocprofv3 child runs after the timed legs), not
    from a committed profile; the compositor reads every layer once and writes the frame once, so the figure sits just above the algorithmic bytes"""
    import shutil
    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not installed")
    w, h, n = 2048, 1024, 20
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--width", str(w), "--height", str(h), "--layers", str(n),
           "--sigma", "3.0", "--no-cpu-baseline", "--no-group", "--headline-only"]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, p.stdout[-2000:] + p.stderr[-2000:]
    rf = json.loads(lines[0])["roofline"]
    live = rf["traffic_live"]
    assert "error" not in live, live
    alg = (4 * n + 4) * w * h
    assert rf["traffic"] == live["hbm_bytes"] and live["launches_counted"] >= 4
    assert 0.95 * alg <= live["hbm_bytes"] <= 1.35 * alg, (live, alg)
    assert abs(live["write_bytes"] - 4 * w * h) <= 0.2 * 4 * w * h
