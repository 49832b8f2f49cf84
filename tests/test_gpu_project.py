"""PFE documents on the GPU: CanvasState::composite() of a loaded project, run_one's script step with canvas-op replay on the
other layers (ref: src/cli.rs:222-308, src/ops/scripting.rs:1640-1723), the device TiledImage import / export kernels, and the
CLI with .pfe input and output.  Oracle: tests/oracle_lib.py for the pixels, tests/pfe_format.py for the file layout."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from . import inputs as I
from . import oracle_lib as O
from . import pfe_format as F
from .test_pfe_format import dropped, sparse_image

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "paintfe_amd", "pfx")


@pytest.fixture(scope="module")
def r():
    from paintfe_amd import GpuRenderer
    return GpuRenderer(0)


def project_from(w, h, layers, folders=(), active=0):
    from paintfe_amd.project import Project
    p = Project.new(w, h)
    for fid, name, vis in folders:
        p.add_folder(fid, name, vis)
    for i, L in enumerate(layers):
        p.add_layer(L.get("name", f"L{i}"), L.get("pixels"), opacity=L.get("opacity", 1.0), blend_mode=L.get("mode", 0), visible=L.get("visible", True),
                    kind=L.get("kind", 0), adj=L.get("adj", ()))
        if L.get("folder") is not None:
            p.set_layer_folder(i, L["folder"])
    p.set_active_layer(active)
    return p


def oracle_layers(layers, hidden_folders=()):
    out = []
    for L in layers:
        d = dict(L)
        if d.get("pixels") is not None:
            d["pixels"] = dropped(d["pixels"])
        if d.get("folder") in hidden_folders:
            d["visible"] = False
        out.append(d)
    return out


@pytest.mark.parametrize("size", [(200, 150), (64, 64), (130, 67), (1, 1), (513, 260)])
def test_project_composite_matches_the_oracle(r, size):
    from paintfe_amd.project import Project
    w, h = size
    layers = [
        dict(pixels=sparse_image(w, h, 1), name="bg"),
        dict(kind=O.ADJ_INVERT, opacity=0.6),
        dict(pixels=sparse_image(w, h, 2), mode=8, opacity=0.8, folder=3),
        dict(pixels=I.random_rgba(w, h, 3), visible=False, mode=3),
        dict(pixels=sparse_image(w, h, 4), mode=21, opacity=0.5, folder=9),     # folder 9 is hidden
        dict(kind=O.ADJ_EXPOSURE, adj=[0.7]),
        dict(pixels=sparse_image(w, h, 5), mode=13, opacity=0.9),
        dict(kind=O.ADJ_BC, adj=[12.0, 30.0], opacity=0.5),
        dict(kind=O.ADJ_MIXER, adj=[0.5, 0.3, 0.2, 0.0, 0.1, 0.8, 0.1, 0.0, 0.0, 0.2, 0.8, 0.0, 0.0, 0.0, 0.0, 1.0], opacity=0.9),
    ]
    p = project_from(w, h, layers, folders=[(3, "shown", True), (9, "hidden", False)])
    ref = O.composite(oracle_layers(layers, hidden_folders=(9,)), w, h)
    assert np.array_equal(p.composite(r), ref)
    # the same through a save -> load round trip (V3: folders + adjustment layers)
    q = Project.load_bytes(p.save_bytes())
    assert q.version == 3 and np.array_equal(q.composite(r), ref)


def test_present_but_transparent_chunk_counts_for_adjustment_layers(r):
    """canvas_state.rs:528-550: an adjustment layer runs on every chunk some visible layer HOLDS (chunk_keys()), pixels or not.
    A .pfe file can store a fully transparent chunk; Invert over it yields (255, 255, 255, 0), and a layer above that blends
    against that colour.  Chunks no layer holds stay (0, 0, 0, 0).  (The flat-image oracle cannot express this: expected values
    are derived by hand from layers.rs:276-325.)"""
    from paintfe_amd.project import Project
    from tests import pfe_format as F
    w = h = 128
    solid = np.zeros((64, 64, 4), np.uint8); solid[...] = (10, 200, 90, 255)
    clear = np.zeros((64, 64, 4), np.uint8)
    base = {"name": "base", "visible": True, "folder_id": None, "opacity": 1.0, "blend_mode": 0, "layer_type": 0,
            "chunks": [(0, 0, solid.tobytes()), (1, 0, clear.tobytes())],          # chunk (1,0) is stored although transparent
            "content_data": None, "pixel_format": 0, "hdr_metadata": dict(F.DEFAULT_HDR), "source_metadata": dict(F.DEFAULT_META),
            "webp_frame_compression": 1, "deep_pixels": None}
    inv = dict(base, name="invert", layer_type=2, chunks=[], content_data=F.adjustment_bytes(2))
    raw = F.encode({"version": 3, "width": w, "height": h, "active_layer_index": 0, "folders": [], "next_layer_folder_id": 1, "layers": [base, inv]})
    got = Project.load_bytes(raw).composite(r)
    assert (got[:64, :64] == np.array([245, 55, 165, 255], np.uint8)).all()       # inverted solid chunk
    assert (got[:64, 64:] == np.array([255, 255, 255, 0], np.uint8)).all()        # held-but-transparent chunk: colour inverted, alpha 0
    assert (got[64:] == 0).all()                                                   # chunks nobody holds


def test_project_composite_nothing_visible_and_legacy_versions(r):
    from paintfe_amd.project import Project
    w, h = 100, 70
    img = sparse_image(w, h, 7)
    p = project_from(w, h, [dict(pixels=img, visible=False)])
    assert not p.composite(r).any()
    top = sparse_image(w, h, 8)
    raw = F.encode({"version": 0, "width": w, "height": h, "active_layer_index": 0,
                    "layers": [{"name": "a", "visible": True, "opacity": 1.0, "blend_mode": 0, "pixels": img.tobytes()},
                               {"name": "b", "visible": True, "opacity": 0.7, "blend_mode": 2, "pixels": top.tobytes()}]})
    ref = O.composite([dict(pixels=dropped(img)), dict(pixels=dropped(top), opacity=0.7, mode=2)], w, h)
    assert np.array_equal(Project.load_bytes(raw).composite(r), ref)
    raw2 = F.encode({"version": 2, "width": w, "height": h, "active_layer_index": 1,
                     "layers": [F.raster_layer("a", img, text_data=None),
                                dict(F.raster_layer("t", top, opacity=0.7, blend_mode=2), layer_type=1, text_data=b"text payload")]})
    assert np.array_equal(Project.load_bytes(raw2).composite(r), ref), "a text layer composites through its rasterised chunks"


def test_tiled_import_export_kernels(r):
    lib = r._lib
    for (w, h) in ((200, 150), (64, 64), (65, 129), (3, 2)):
        img = sparse_image(w, h, w + h)
        tiles = F.tiles_from_image(img)
        cxn, cyn = (w + 63) // 64, (h + 63) // 64
        slot = np.full(cxn * cyn, 0xFFFFFFFF, np.uint32)
        for k, (cx, cy, _) in enumerate(tiles):
            slot[cy * cxn + cx] = k
        packed = np.frombuffer(b"".join(t[2] for t in tiles) or b"\0" * 4, np.uint8)
        d_packed, d_flat = r.dev_alloc(max(packed.size, 4)), r.dev_alloc(w * h * 4)
        r.dev_upload(d_packed, packed)
        r._check(lib.pfx_tiled_import_dev(r.handle, C.c_void_p(d_packed), slot.ctypes.data_as(C.c_void_p), C.c_uint32(w), C.c_uint32(h), C.c_void_p(d_flat)))
        assert np.array_equal(r.dev_download(d_flat, (h, w, 4)), dropped(img)), "import = to_rgba_image"
        # export the original (undropped) image with the populated slot table: chunk bytes incl. zero padding past the edges
        r.dev_upload(d_flat, img)
        d_out = r.dev_alloc(max(packed.size, 4))
        r._check(lib.pfx_tiled_export_dev(r.handle, C.c_void_p(d_flat), C.c_uint32(w), C.c_uint32(h), slot.ctypes.data_as(C.c_void_p), C.c_void_p(d_out)))
        if tiles:
            assert np.array_equal(r.dev_download(d_out, (packed.size,)), packed), "export = from_rgba_image chunk bytes"
        for d in (d_packed, d_flat, d_out):
            r.dev_free(d)


def replay(img, ops):
    """apply_canvas_ops on one layer with the oracle: every op is followed by TiledImage::from_rgba_image (scripting.rs:1650-1705)"""
    names = ["flip_horizontal", "flip_vertical", "rotate_90cw", "rotate_90ccw", "rotate_180"]
    filters = ["nearest", "bilinear", "bicubic", "lanczos3"]
    for (kind, ow, oh, ax, ay) in ops:
        if kind <= 4:
            img = O.flip_rotate(img, names[kind])
        elif kind == 5:
            img = O.resize(img, ow, oh, filters[ax])
        else:
            img = O.resize_canvas(img, ow, oh, (ax, ay))
        img = O.tiled_roundtrip(img)
    return img


SCRIPTS = [
    ("apply_invert(); apply_blur(1.5);", False),
    ("flip_horizontal(); rotate_180();", False),            # layer-only transforms: nothing to replay (scripting.rs:640-674)
    ("flip_canvas_horizontal();", True),
    ("rotate_canvas_90cw(); apply_invert();", True),
    ("rotate_canvas_90ccw(); flip_vertical(); rotate_canvas_180();", True),
    ("resize_image(90, 50, \"bilinear\");", True),
    ("resize_image(301, 77, \"lanczos3\"); rotate_canvas_90cw();", True),
    ("resize_canvas(260, 180, \"center\");", True),
    ("resize_canvas(100, 60, \"bottom-right\"); resize_image(64, 64, \"nearest\");", True),
]


@pytest.mark.parametrize("src,has_ops", SCRIPTS, ids=[s for s, _ in SCRIPTS])
def test_project_run_script_replays_canvas_ops(r, src, has_ops):
    from paintfe_amd.project import Project
    w, h = 200, 130
    imgs = [sparse_image(w, h, 60 + k) for k in range(3)]
    layers = [dict(pixels=imgs[0], name="bg"), dict(pixels=imgs[1], mode=1, opacity=0.8), dict(pixels=imgs[2], mode=2, opacity=0.6),
              dict(kind=O.ADJ_INVERT, opacity=0.3)]
    p = project_from(w, h, layers, active=1)
    # the active layer's expected pixels: the script front-end itself (pinned against the reference's goldens elsewhere)
    r.set_exact(True)
    try:
        want_active, _, ops = r.execute_script_sync(src, dropped(imgs[1]), with_ops=True)
        p.run_script(r, src)
    finally:
        r.set_exact(False)
    assert bool(ops) == has_ops
    nh, nw = want_active.shape[:2]
    assert (p.width, p.height) == (nw, nh)
    assert np.array_equal(p.layer_pixels(1), O.tiled_roundtrip(want_active)), "active layer"
    for k in (0, 2):
        want = replay(dropped(imgs[k]), ops) if ops else dropped(imgs[k])
        assert np.array_equal(p.layer_pixels(k), want), f"layer {k}"
    assert p.layer(3)["kind"] == O.ADJ_INVERT and p.layer(3)["n_chunks"] == 0
    assert p.layer(1)["blend_mode"] == 1 and p.layer(2)["opacity"] == np.float32(0.6)
    # and the document still composites like the oracle's
    ref_layers = [dict(pixels=p.layer_pixels(0)), dict(pixels=p.layer_pixels(1), mode=1, opacity=0.8), dict(pixels=p.layer_pixels(2), mode=2, opacity=0.6),
                  dict(kind=O.ADJ_INVERT, opacity=0.3)]
    assert np.array_equal(p.composite(r), O.composite(ref_layers, nw, nh))
    q = Project.load_bytes(p.save_bytes())
    assert (q.width, q.height) == (nw, nh) and np.array_equal(q.layer_pixels(2), p.layer_pixels(2))


def test_project_script_error_leaves_the_document_alone(r):
    from paintfe_amd import PfxError
    img = sparse_image(100, 80, 5)
    p = project_from(100, 80, [dict(pixels=img), dict(pixels=sparse_image(100, 80, 6))])
    before = p.save_bytes()
    with pytest.raises(PfxError):
        p.run_script(r, "rotate_canvas_90cw(); this_function_does_not_exist();")
    assert p.save_bytes() == before


def _read_png(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGBA"))


def test_cli_pfe_in_png_and_pfe_out(tmp_path):
    """paintfe -i project.pfe -o flat.png flattens the visible layers; -f pfe keeps them (cli.rs:9,272-306)"""
    from paintfe_amd.project import Project
    w, h = 200, 130
    imgs = [sparse_image(w, h, 70 + k) for k in range(3)]
    layers = [dict(pixels=imgs[0]), dict(pixels=imgs[1], mode=8, opacity=0.7), dict(pixels=imgs[2], mode=3, visible=False), dict(kind=O.ADJ_EXPOSURE, adj=[0.5])]
    p = project_from(w, h, layers, active=1)
    p.save(str(tmp_path / "doc.pfe"))
    run = subprocess.run([EXE, "-i", str(tmp_path / "doc.pfe"), "-o", str(tmp_path / "flat.png")], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    assert np.array_equal(_read_png(tmp_path / "flat.png"), O.composite(oracle_layers(layers), w, h))
    # script on the active layer + canvas op on all of them, PFE out
    (tmp_path / "s.rhai").write_text("apply_invert();\nrotate_canvas_90cw();\nprint_line(\"turned\");\n")
    run = subprocess.run([EXE, "-i", str(tmp_path / "doc.pfe"), "-s", str(tmp_path / "s.rhai"), "-o", str(tmp_path / "out.pfe"), "-v"],
                         capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    assert "[script] turned" in run.stdout
    q = Project.load(str(tmp_path / "out.pfe"))
    assert (q.width, q.height) == (h, w) and len(q) == 4 and q.active_layer == 1
    assert np.array_equal(q.layer_pixels(0), O.tiled_roundtrip(O.flip_rotate(dropped(imgs[0]), "rotate_90cw")))
    assert np.array_equal(q.layer_pixels(1), O.tiled_roundtrip(O.flip_rotate(O.rhai_adjust(dropped(imgs[1]), "invert"), "rotate_90cw")))
    assert not q.layer(2)["visible"] and q.layer(3)["kind"] == O.ADJ_EXPOSURE
    # and flattened to PNG in the same run shape
    run = subprocess.run([EXE, "-i", str(tmp_path / "out.pfe"), "-f", "png", "--output-dir", str(tmp_path / "o")], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    ref_layers = [dict(pixels=q.layer_pixels(0)), dict(pixels=q.layer_pixels(1), mode=8, opacity=0.7), dict(pixels=q.layer_pixels(2), mode=3, visible=False),
                  dict(kind=O.ADJ_EXPOSURE, adj=[0.5])]
    assert np.array_equal(_read_png(tmp_path / "o" / "out.png"), O.composite(ref_layers, h, w))
    # a PNG can be turned into a one-layer project named after the file
    from PIL import Image
    Image.fromarray(imgs[0], "RGBA").save(tmp_path / "photo.png")
    run = subprocess.run([EXE, "-i", str(tmp_path / "photo.png"), "-o", str(tmp_path / "photo.pfe")], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    d = F.decode((tmp_path / "photo.pfe").read_bytes())
    assert d["version"] == 1 and d["layers"][0]["name"] == "photo" and d["layers"][0]["chunks"] == F.tiles_from_image(imgs[0])
    # malformed project: reported, exit code 1
    (tmp_path / "bad.pfe").write_bytes(b"\x04\0\0\0\0\0\0\0PFE1" + b"\0" * 3)
    run = subprocess.run([EXE, "-i", str(tmp_path / "bad.pfe"), "-o", str(tmp_path / "x.png")], capture_output=True, text=True)
    assert run.returncode == 1 and "load failed" in run.stderr


def test_project_composite_8k_sparse_document(r):
    """a document-sized case: 8K canvas, 6 layers with 10-60 % of their chunks present; only stored chunks cross PCIe"""
    w, h = 7680, 4320
    rng = np.random.default_rng(11)
    from paintfe_amd.project import Project
    p = Project.new(w, h)
    ref_layers = []
    for k in range(6):
        img = np.zeros((h, w, 4), np.uint8)
        keep = rng.random(((h + 63) // 64, (w + 63) // 64)) < (1.0 if k == 0 else 0.1 + 0.1 * k)
        tile = I.random_rgba(64, 64, 100 + k)
        tile[..., 3] |= 1
        for cy, cx in zip(*np.nonzero(keep)):
            part = img[cy * 64:(cy + 1) * 64, cx * 64:(cx + 1) * 64]
            part[...] = np.roll(tile, (int(cy), int(cx)), (0, 1))[:part.shape[0], :part.shape[1]]
        p.add_layer(f"L{k}", img, opacity=1.0 if k == 0 else 0.9 - 0.1 * k, blend_mode=(k * 5) % 25)
        ref_layers.append(dict(pixels=img, opacity=1.0 if k == 0 else 0.9 - 0.1 * k, mode=(k * 5) % 25))
    got = p.composite(r)
    for (x, y) in ((0, 0), (w - 512, h - 320), (3000, 2000)):
        win = [dict(L, pixels=np.ascontiguousarray(L["pixels"][y:y + 320, x:x + 512])) for L in ref_layers]
        assert np.array_equal(got[y:y + 320, x:x + 512], O.composite(win, 512, 320)), (x, y)
