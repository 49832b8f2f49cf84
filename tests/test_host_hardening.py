"""Host hardening gate (SURVEY 5 "sanitizers"): the three parsers of untrusted bytes — the CLI's PNG reader, the PFE project reader and the
script front end — compiled with AddressSanitizer + UndefinedBehaviorSanitizer (paintfe_amd/csrc: `make asan`) and driven by a mutation fuzzer
(tests/cpp/fuzz_host.cpp) over seeds this file writes: PNGs of every colour type / bit depth / interlace method, PFE files of versions 0-3 from the
independent bincode restatement (tests/pfe_format.py), and every script source the language tests use.  A sanitizer report, a crash, a hang or a
violated post-condition fails the gate.  A fourth stage drives the host math (table builders, the brush-line walk, the displacement brush) with hostile floats —
NaN, infinities, 1e30, random bit patterns — with float-to-integer conversions sanitized as well.  No GPU.  References: /root/reference/src/io.rs:477-503, 693-723; src/ops/scripting.rs:288-293, 1489-1508."""
import ast
import json
import os
import subprocess

import numpy as np
import pytest

from tests import pfe_format as F
from tests.test_gpu_script_cli import _png_bytes
from tests.test_pfe_format import full_v3_document, sparse_image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
ITERATIONS = int(os.environ.get("PFX_FUZZ_ITERATIONS", "40000"))   # per parser: 1.2e5 mutated inputs in the gate


def script_sources():
    out = []
    for name in ("test_script_lang_host.py", "test_gpu_script_lang.py", "test_gpu_script_cli.py"):
        tree = ast.parse(open(os.path.join(ROOT, "tests", name)).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.Constant) and isinstance(node.value, str) and len(node.value) > 8 and (";" in node.value or "{" in node.value) \
                    and "\x00" not in node.value and len(node.value) < 4000:
                out.append(node.value)
    return sorted(set(out))


def write_seeds(d):
    n = 0
    for color_type, depths in ((0, (1, 2, 4, 8, 16)), (2, (8, 16)), (4, (8, 16)), (6, (8, 16))):
        for bit_depth in depths:
            for interlace in (0, 1):
                rng = np.random.default_rng(color_type * 100 + bit_depth + interlace)
                ch = {0: 1, 2: 3, 4: 2, 6: 4}[color_type]
                arr = rng.integers(0, 1 << bit_depth, size=(11, 13, ch), dtype=np.uint32)
                open(os.path.join(d, f"s{n:03d}.png"), "wb").write(_png_bytes(arr, color_type, bit_depth, interlace))
                n += 1
    from PIL import Image   # palette images (with and without tRNS), written by an independent encoder
    rng = np.random.default_rng(5)
    pal = Image.fromarray(rng.integers(0, 256, size=(9, 17, 3), dtype=np.uint8)).quantize(16)
    pal.save(os.path.join(d, f"s{n:03d}.png")); n += 1
    pal.save(os.path.join(d, f"s{n:03d}.png"), transparency=3); n += 1
    Image.fromarray(rng.integers(0, 256, size=(20, 20, 3), dtype=np.uint8)).save(os.path.join(d, f"s{n:03d}.png"), transparency=(1, 2, 3)); n += 1
    doc, _ = full_v3_document()
    img = sparse_image(100, 70, 77)
    pfes = [F.encode(doc),
            F.encode({"version": 1, "width": 100, "height": 70, "active_layer_index": 0, "layers": [F.raster_layer("a", img), F.raster_layer("b", img, opacity=0.5)]}),
            F.encode({"version": 2, "width": 100, "height": 70, "active_layer_index": 0, "layers": [dict(F.raster_layer("t", img), layer_type=1, text_data=b"payload")]}),
            F.encode({"version": 0, "width": 10, "height": 6, "active_layer_index": 0,
                      "layers": [{"name": "l", "visible": True, "opacity": 1.0, "blend_mode": 0, "pixels": bytes(240)}]}),
            F.encode({"version": 3, "width": 64, "height": 64, "active_layer_index": 1, "folders": [], "next_layer_folder_id": 1,
                      "layers": [F.raster_layer("p", sparse_image(64, 64, 3)),
                                 {"name": "adj", "visible": True, "opacity": 0.5, "blend_mode": 0, "layer_type": 2, "chunks": [], "content_data": F.adjustment_bytes(1, [5.0, 5.0])}]})]
    for k, raw in enumerate(pfes):
        open(os.path.join(d, f"s{k:03d}.pfe"), "wb").write(raw)
    srcs = script_sources()
    assert len(srcs) > 100, "the language tests' sources are the script seeds"
    for k, s in enumerate(srcs):
        open(os.path.join(d, f"s{k:04d}.rhai"), "w").write(s)


@pytest.fixture(scope="module")
def harness():
    r = subprocess.run(["make", "-C", CPP, "fuzz_host"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return os.path.join(CPP, "fuzz_host")


def run(harness, seeds, iterations, seed, timeout):
    def once(leaks):
        env = dict(os.environ, ASAN_OPTIONS=f"detect_leaks={leaks}:max_allocation_size_mb=3072:allocator_may_return_null=1",
                   UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
        return subprocess.run([harness, seeds, str(iterations), str(seed)], capture_output=True, text=True, timeout=timeout, env=env)
    r = once(1)
    if r.returncode != 0 and "LeakSanitizer has encountered a fatal error" in r.stderr:   # a sandbox without ptrace: the leak pass cannot run, everything else can
        r = once(0)
    assert r.returncode == 0, f"fuzz_host seed {seed}: exit {r.returncode}\n" + r.stdout[-1000:] + r.stderr[-6000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_mutated_png_pfe_and_script_inputs_under_asan_ubsan(harness, tmp_path):
    write_seeds(str(tmp_path))
    res = run(harness, str(tmp_path), ITERATIONS, 20260930, timeout=900)
    assert res["iterations_per_kind"] == ITERATIONS
    for kind in ("png", "pfe", "script"):   # both outcomes are exercised: the mutations neither bounce off the first check nor leave the inputs intact
        assert res[kind]["ok"] > ITERATIONS // 50 and res[kind]["error"] > ITERATIONS // 50, res
    assert res["png"]["structured"] > ITERATIONS // 2 and res["png"]["max_decoded_px"] > 1000, res
    # host math on hostile floats: the brush-line walk equals the reference's full walk wherever that is affordable, incl. lines long enough to be clipped
    assert res["host_math"]["lines_equal_to_the_full_walk"] > ITERATIONS // 4 and res["host_math"]["of_them_clipped_walks"] > ITERATIONS // 400, res


def test_unmutated_seeds_all_parse(harness, tmp_path):
    """iteration count 0 is not a no-op check of the harness: the seeds themselves must load through the sanitized parsers (PNG / PFE) and the library's own loader"""
    import paintfe_amd as P
    write_seeds(str(tmp_path))
    for name in sorted(os.listdir(tmp_path)):
        raw = open(os.path.join(tmp_path, name), "rb").read()
        if name.endswith(".png"):
            px = P.png_decode(raw)
            assert px.ndim == 3 and px.shape[2] == 4, name
        elif name.endswith(".pfe"):
            assert len(P.Project.load_bytes(raw)) >= 1, name


def test_png_header_cannot_demand_more_than_the_data_can_inflate_to():
    """a 70-byte file that declares 16000 x 16000 RGBA (1 GB of scanlines) is refused before anything of that size is allocated (deflate expands at most 1032 : 1)"""
    import struct
    import zlib
    import paintfe_amd as P

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    raw = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 16000, 16000, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b"")
    import resource
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    with pytest.raises(P.PfxError) as e:
        P.png_decode(raw)
    assert "too short" in str(e.value)
    assert resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before < 64 * 1024, "peak RSS (KiB) must not jump by the declared size"
