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


def _run(nproc, extra, port):
    env = dict(os.environ, PFX_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3", "--warmup", "1",
           "--width", "640", "--height", "400", "--layers", "6", "--sigma", "3.0", "--no-cpu-baseline"] + extra
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
    assert d["scaling"] == "strong" and d["n_gpus"] == nproc and "all-gather" in d["config"]["sharding"]
    assert d["check"]["band_blur_max_diff_vs_oracle"] == 0  # exact Gaussian: bit-identical to the unsharded pipeline
    assert abs(d["value"] - 640 * 400 / d["ms_per_step"] / 1e3) / d["value"] < 0.01  # ONE document per step for the whole job
    assert d["doc_mode"]["scaling"] == "weak" and d["doc_mode"]["value"] > 0
    d = _run(nproc, [], 29640 + nproc)
    assert d["check"]["band_blur_max_diff_vs_oracle"] <= 1  # matrix-core Gaussian: the stated +-1 LSB
