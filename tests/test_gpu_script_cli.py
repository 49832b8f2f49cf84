"""B5/B6 on the GPU: the Rhai Effect-API front-end (call-statement subset) and the batch CLI, through the C ABI.
Goldens: tests/scripting.rs:119-152 (reference) -> tests/golden/golden.npz scripting/*."""
import os
import subprocess

import numpy as np
import pytest

from . import inputs as I
from . import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def r():
    from paintfe_amd import GpuRenderer
    return GpuRenderer(0)


GOLDEN_SCRIPTS = [
    ("scripting/apply_blur", "apply_blur(2.0);"),
    ("scripting/apply_invert", "apply_invert();"),
    ("scripting/apply_sepia", "apply_sepia();"),
    ("scripting/apply_desaturate", "apply_desaturate();"),
    ("scripting/apply_brightness_contrast", "apply_brightness_contrast(20.0, 10.0);"),
    ("scripting/apply_pixelate", "apply_pixelate(4);"),
]


@pytest.mark.parametrize("key,src", GOLDEN_SCRIPTS, ids=[k for k, _ in GOLDEN_SCRIPTS])
def test_script_goldens(r, golden, key, src):
    out, console = r.execute_script_sync(src, I.create_test_gradient(64, 64))
    assert np.array_equal(out, golden[key])


def test_script_chain_comments_and_console(r):
    img = I.random_rgba(150, 90, 3)
    src = """
    // effect chain, one upload / one download
    apply_invert();
    /* nested /* block */ comment */
    apply_blur(1.5); apply_hsl(10.0, 5.0, 0.0);
    print_line("done");
    apply_box_blur(2);
    apply_median(1);
    apply_levels(10.0, 240.0, 1.1);
    apply_exposure(-0.5); apply_sepia(0.25);
    apply_sharpen(1.5); apply_glow(2.0, 0.4); apply_motion_blur(30.0, 4.0)
    """
    r.set_exact(True)
    out, console = r.execute_script_sync(src, img)
    r.set_exact(False)
    ref = O.rhai_adjust(img, "invert")
    ref = O.gaussian_blur(ref, 1.5)
    ref = O.rhai_adjust(ref, "hsl", [10.0, 5.0, 0.0])
    ref = O.box_blur(ref, 2.0)
    ref = O.median(ref, 1)
    ref = O.rhai_adjust(ref, "levels", [10.0, 240.0, 1.1])
    ref = O.rhai_adjust(ref, "exposure", [-0.5])
    ref = O.rhai_adjust(ref, "sepia_strength", [0.25])
    ref = O.sharpen(ref, 1.5, 1.0)
    ref = O.glow(ref, 2.0, 0.4)
    ref = O.motion_blur(ref, 30.0, 4.0)
    assert np.array_equal(out, ref)
    assert console == ["done"]


def test_script_selection_mask(r):
    img = I.random_rgba(120, 80, 5)
    mask = np.zeros((80, 120), np.uint8)
    mask[20:60, 30:100] = 255
    r.set_exact(True)
    out, _ = r.execute_script_sync("apply_blur(2.0); apply_box_blur(3); apply_invert();", img, mask)
    r.set_exact(False)
    ref = O.gaussian_blur(img, 2.0, mask)
    ref = O.box_blur(ref, 3.0, mask)
    ref = O.rhai_adjust(ref, "invert")  # inline ops ignore the selection (scripting.rs:869)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("src,status,needle", [
    ("apply_blur(2);", -6, "Function not found: apply_blur (i64)"),          # Rhai does not coerce i64 -> f64
    ("apply_frobnicate(1.0);", -6, "Function not found: apply_frobnicate (f64)"),
    ("let x = 4; apply_blur(x);", -6, "Function not found: apply_blur (i64)"),  # typing is checked on values, not literals
    ("map_channels(|r, g, b, a| { print(r); [r, g, b, a] });", -5, "cannot be compiled for the GPU"),
    ("import \"effects\";", -5, "outside the supported subset"),
    ("apply_invert(); throw \"stop here\";", -6, "Runtime error: stop here"),
    ("apply_blur(2.0", -6, "Expecting ')'"),
    ('print_line("unterminated);', -6, "not terminated"),
])
def test_script_errors_leave_pixels_untouched(r, src, status, needle):
    from paintfe_amd import PfxError
    img = I.random_rgba(40, 30, 9)
    with pytest.raises(PfxError) as e:
        r.execute_script_sync(src, img)
    assert e.value.status == status and needle in str(e.value)
    assert e.value.line >= 1


def _write_png(path, arr):
    from PIL import Image
    Image.fromarray(arr, "RGBA").save(path)


def _read_png(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGBA"))


def test_cli_batch(tmp_path):
    exe = os.path.join(ROOT, "paintfe_amd", "pfx")
    a = I.create_test_gradient(200, 130)
    b = I.random_rgba(97, 150, 4)
    b[:, :64, 3] = 0  # a fully transparent chunk column: TiledImage drops it on load (colour information is lost)
    _write_png(tmp_path / "a.png", a)
    _write_png(tmp_path / "b.png", b)
    (tmp_path / "bad.png").write_bytes(b"not a png")
    (tmp_path / "s.rhai").write_text("// BASELINE config 1 style script\napply_blur(4.0);\n")
    out_dir = tmp_path / "out"
    p = subprocess.run([exe, "-i", str(tmp_path / "a.png"), str(tmp_path / "b.png"), str(tmp_path / "bad.png"), "-s", str(tmp_path / "s.rhai"),
                        "--output-dir", str(out_dir), "-v"], capture_output=True, text=True)
    assert p.returncode == 1, p.stderr                      # one file failed, the loop kept going (ref: cli.rs:204-215)
    assert "[1/3]" in p.stdout and "load failed" in p.stderr
    for name, src in (("a", a), ("b", b)):
        got = _read_png(out_dir / f"{name}.png")
        ref = O.tiled_roundtrip(O.gaussian_blur(O.tiled_roundtrip(src), 4.0))
        d = np.abs(got.astype(np.int16) - ref.astype(np.int16))
        assert d.max() <= 1 and (d > 0).mean() < 4e-3       # default-mode Gaussian: 1 LSB; on white noise at sigma 4 up to 2e-3 of the channels (one f16 per tap)
    # format conversion only (no script), explicit output, palette/greyscale decode paths are exercised in the unit below
    p = subprocess.run([exe, "-i", str(tmp_path / "a.png"), "-o", str(tmp_path / "copy.png")], capture_output=True, text=True)
    assert p.returncode == 0 and np.array_equal(_read_png(tmp_path / "copy.png"), a)
    p = subprocess.run([exe, "-i", str(tmp_path / "a.png"), "-o", str(tmp_path / "x.jpg")], capture_output=True, text=True)
    assert p.returncode == 1 and "not built into this back-end" in p.stderr


def test_cli_config1_literal_1024_and_gpus_flag(tmp_path):
    """BASELINE config 1 as written: a 1024x1024 PNG of create_test_gradient + the script `apply_gaussian_blur(4.0);` (the alias of
    the reference's apply_blur) through the CLI; and --gpus N shards a file list over worker contexts with the same results."""
    exe = os.path.join(ROOT, "paintfe_amd", "pfx")
    img = I.create_test_gradient(1024, 1024)
    _write_png(tmp_path / "in.png", img)
    (tmp_path / "blur.rhai").write_text("apply_gaussian_blur(4.0);\n")
    p = subprocess.run([exe, "-i", str(tmp_path / "in.png"), "-s", str(tmp_path / "blur.rhai"), "-o", str(tmp_path / "out.png")],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    got = _read_png(tmp_path / "out.png")
    ref = O.tiled_roundtrip(O.gaussian_blur(O.tiled_roundtrip(img), 4.0))
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16))
    assert got.shape == (1024, 1024, 4) and d.max() <= 1 and (d > 0).mean() < 1e-3
    # the same script on five files over three workers (they share device 0 on a 1-GPU box): identical outputs, reports in input order
    names = []
    for k in range(5):
        _write_png(tmp_path / f"f{k}.png", I.random_rgba(120 + 8 * k, 90 + k, k))
        names.append(str(tmp_path / f"f{k}.png"))
    p1 = subprocess.run([exe, "-i", *names, "-s", str(tmp_path / "blur.rhai"), "--output-dir", str(tmp_path / "o1")], capture_output=True, text=True)
    p3 = subprocess.run([exe, "-i", *names, "-s", str(tmp_path / "blur.rhai"), "--output-dir", str(tmp_path / "o3"), "--gpus", "3"],
                        capture_output=True, text=True)
    assert p1.returncode == 0 and p3.returncode == 0, p1.stderr + p3.stderr
    order = [l for l in p3.stdout.splitlines() if l.startswith("[")]
    assert [l.split("]")[0] for l in order] == [f"[{k + 1}/5" for k in range(5)]
    for k in range(5):
        assert np.array_equal(_read_png(tmp_path / "o1" / f"f{k}.png"), _read_png(tmp_path / "o3" / f"f{k}.png"))


def _png_bytes(arr, color_type, bit_depth, interlace):
    """a minimal PNG writer for the decoder test: any colour type / bit depth, filter 0, optional Adam7 (PNG spec 8.2)"""
    import struct, zlib
    h, w = arr.shape[:2]
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[color_type]
    a = arr.reshape(h, w, ch)

    def rows(sub):
        out = bytearray()
        for row in sub:
            out.append(0)
            if bit_depth == 16:
                out += row.astype(">u2").tobytes()
            elif bit_depth == 8:
                out += row.astype(np.uint8).tobytes()
            else:
                bits = "".join(format(int(v), f"0{bit_depth}b") for v in row.reshape(-1))
                bits += "0" * (-len(bits) % 8)
                out += int(bits, 2).to_bytes(len(bits) // 8, "big") if bits else b""
        return bytes(out)

    if interlace:
        raw = b""
        for (x0, y0, dx, dy) in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]:
            sub = a[y0::dy, x0::dx]
            if sub.size:
                raw += rows(sub)
    else:
        raw = rows(a)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, color_type, 0, 0, 1 if interlace else 0)) +
            chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


@pytest.mark.parametrize("color_type,bit_depth,interlace", [(6, 16, 0), (6, 8, 1), (2, 16, 1), (0, 16, 0), (0, 4, 1), (0, 1, 0), (4, 16, 1), (4, 8, 0)])
def test_cli_png_16bit_and_interlaced(tmp_path, color_type, bit_depth, interlace):
    """16-bit samples become 8-bit like the `image` crate's to_rgba8 ((v + 128) / 257), low-bit grey is stretched, Adam7 is undone"""
    exe = os.path.join(ROOT, "paintfe_amd", "pfx")
    rng = np.random.default_rng(color_type * 100 + bit_depth + interlace)
    w, h = 37, 23
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[color_type]
    arr = rng.integers(0, 1 << bit_depth, size=(h, w, ch), dtype=np.uint32)
    (tmp_path / "in.png").write_bytes(_png_bytes(arr, color_type, bit_depth, interlace))
    p = subprocess.run([exe, "-i", str(tmp_path / "in.png"), "-o", str(tmp_path / "out.png")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    to8 = (lambda v: (v + 128) // 257) if bit_depth == 16 else ((lambda v: v) if bit_depth == 8 else (lambda v: v * 255 // ((1 << bit_depth) - 1)))
    v = to8(arr).astype(np.uint8)
    want = np.empty((h, w, 4), np.uint8)
    if color_type == 0:
        want[..., :3] = v[..., :1]; want[..., 3] = 255
    elif color_type == 2:
        want[..., :3] = v; want[..., 3] = 255
    elif color_type == 4:
        want[..., :3] = v[..., :1]; want[..., 3] = v[..., 1]
    else:
        want[...] = v
    got = _read_png(tmp_path / "out.png")
    # the CLI stores the image as a TiledImage: chunks with no alpha at all lose their colour (tiled_image.rs:50-104)
    assert np.array_equal(got, O.tiled_roundtrip(want))


def test_cli_png_decoder_variants(tmp_path):
    from PIL import Image
    exe = os.path.join(ROOT, "paintfe_amd", "pfx")
    rgb = I.random_rgba(33, 21, 1)[..., :3]
    Image.fromarray(rgb, "RGB").save(tmp_path / "rgb.png")
    Image.fromarray(rgb[..., 0], "L").save(tmp_path / "gray.png")
    Image.fromarray(rgb, "RGB").convert("P", palette=Image.ADAPTIVE, colors=64).save(tmp_path / "pal.png")
    for name in ("rgb", "gray", "pal"):
        p = subprocess.run([exe, "-i", str(tmp_path / f"{name}.png"), "-o", str(tmp_path / f"{name}_o.png")], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        want = np.asarray(Image.open(tmp_path / f"{name}.png").convert("RGBA"))
        assert np.array_equal(_read_png(tmp_path / f"{name}_o.png"), want), name
