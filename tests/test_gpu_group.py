"""pfx_group_* (multi-GPU behind the C ABI): band flatten + xGMI halo exchange + blur + all-gather must equal the single-GPU
result bit for bit.  On a 1-GPU box the members share device 0 (the whole code path runs, peer copies degenerate to
device-to-device copies); with >= 2 visible devices the same test runs one member per device.

The comparison here is group vs the single-GPU C-ABI calls (pfx_flatten_dev / pfx_gaussian_dev / ...), NOT group vs the oracle: those single-GPU calls are the
ones tests/test_gpu_parity.py pins to the oracle bit for bit on the same generators (tests/inputs.py), so equality with them carries the oracle parity over."""
import numpy as np
import pytest

from tests import inputs as I

pytestmark = pytest.mark.gpu


def _devices(n):
    import ctypes as C
    from paintfe_amd import _lib as L
    cnt = L.load().pfx_device_count()
    return [k % max(cnt, 1) for k in range(n)]


@pytest.mark.parametrize("world,size,sigma", [(2, (300, 200), 4.0), (3, (257, 330), 2.0), (2, (64, 640), 16.0), (4, (96, 130), 5.0),
                                             (3, (128, 700), 0.0)])
def test_group_matches_single_gpu(world, size, sigma):
    from paintfe_amd import GpuRenderer
    from paintfe_amd.group import GpuGroup, band_rows
    w, h = size
    n = 7
    stack, modes, opac = I.layer_stack(w, h, n, seed=31 + world)
    infos = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
    r = GpuRenderer(0)
    for k in range(n):
        r.ensure_layer_texture(k, stack[k], generation=1)
    ref = r.composite(w, h, infos)
    if sigma > 0:
        ref = r.blur_rgba(ref, sigma)

    g = GpuGroup(_devices(world))
    g.set_document(w, h, n)
    bands = [g.band(k) for k in range(world)]
    assert bands == [band_rows(h, world, k) for k in range(world)]
    assert bands[0][0] == 0 and bands[-1][1] == h and all(bands[k][1] == bands[k + 1][0] for k in range(world - 1))
    for k in range(n):
        g.upload_layer(k, stack[k])
    for _ in range(2):  # twice: the second call must wait for the first call's readers before overwriting its buffers
        g.flatten_blur(infos, sigma, all_gather=True)
    got = g.download()
    assert np.array_equal(got, ref)
    for k in range(world):
        assert np.array_equal(g.download_gathered(k), ref), f"gathered image on member {k}"
    g.close()


@pytest.mark.parametrize("transport", ["peer", "staged", "peer_denied"])
@pytest.mark.parametrize("filt,param", [("gaussian", 3.0), ("box", 2.0), ("box", 7.3), ("median", 1), ("median", 3), ("none", 0.0)])
def test_group_filters_and_transports(monkeypatch, transport, filt, param):
    """every band filter (Gaussian, box and median need halo rows) under every transport the group can be forced into on one
    box: peer copies, copies staged through pinned host memory, and peer access refused at creation (PFX_GROUP_DENY_PEER=1: what
    hipDeviceEnablePeerAccess failing looks like) — thin bands make halos span several members; results equal the single-GPU calls"""
    from paintfe_amd import GpuRenderer
    from paintfe_amd.group import GpuGroup
    w, h, n, world = 200, 330, 5, 4
    stack, modes, opac = I.layer_stack(w, h, n, seed=91)
    infos = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
    r = GpuRenderer(0)
    for k in range(n):
        r.ensure_layer_texture(k, stack[k], generation=1)
    flat = r.composite(w, h, infos)
    ref = {"gaussian": lambda: r.blur_rgba(flat, param), "box": lambda: r.box_blur_core(flat, param), "median": lambda: r.median_core(flat, int(param)),
           "none": lambda: flat}[filt]()
    if transport == "peer_denied":
        monkeypatch.setenv("PFX_GROUP_DENY_PEER", "1")
    g = GpuGroup(_devices(world))
    if transport == "staged":
        g.set_transport(GpuGroup.STAGED)
        assert g.transport() == GpuGroup.STAGED
    g.set_document(w, h, n)
    for k in range(n):
        g.upload_layer(k, stack[k])
    code = {"none": GpuGroup.BAND_NONE, "gaussian": GpuGroup.BAND_GAUSSIAN, "box": GpuGroup.BAND_BOX, "median": GpuGroup.BAND_MEDIAN}[filt]
    for _ in range(3):  # back to back: bounce buffers and result buffers are re-used while the previous call's copies may still run
        g.flatten_filter(infos, code, float(param), all_gather=True)
    assert np.array_equal(g.download(), ref), "concatenated bands"
    for k in range(world):
        assert np.array_equal(g.download_gathered(k), ref), f"gathered image on member {k}"
    # a larger halo re-allocates the padded buffers while nothing may be in flight, then a smaller one re-uses them
    g.flatten_filter(infos, GpuGroup.BAND_GAUSSIAN, 9.0, all_gather=True)
    g.flatten_filter(infos, code, float(param), all_gather=True)
    for k in range(world):
        assert np.array_equal(g.download_gathered(k), ref), f"after a halo change: member {k}"
    g.close()


def test_group_rccl_transport():
    """RCCL inside the group (ncclSend / ncclRecv halos, ncclBroadcast gather).  One member per device is RCCL's rule: on a 1-GPU box
    this is a one-member group (the load of librccl, communicator set-up, the broadcast group and the stream ordering all run); with
    >= 2 devices it is the real exchange and must equal the peer-copy result bit for bit."""
    from paintfe_amd import GpuRenderer, _lib as L
    from paintfe_amd.group import GpuGroup
    cnt = max(L.load().pfx_device_count(), 1)
    world = min(cnt, 4)
    w, h, n = 256, 300, 4
    stack, modes, opac = I.layer_stack(w, h, n, seed=5)
    infos = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
    r = GpuRenderer(0)
    for k in range(n):
        r.ensure_layer_texture(k, stack[k], generation=1)
    ref = r.blur_rgba(r.composite(w, h, infos), 4.0)
    g = GpuGroup(list(range(world)))
    g.set_transport(GpuGroup.RCCL)
    g.set_document(w, h, n)
    for k in range(n):
        g.upload_layer(k, stack[k])
    for _ in range(2):
        g.flatten_filter(infos, GpuGroup.BAND_GAUSSIAN, 4.0, all_gather=True)
    assert np.array_equal(g.download(), ref)
    for k in range(world):
        assert np.array_equal(g.download_gathered(k), ref)
    g.close()
    if cnt == 1:  # two members on one device: RCCL must be refused with a message, not hang
        g2 = GpuGroup([0, 0])
        with pytest.raises(L.PfxError) as e:
            g2.set_transport(GpuGroup.RCCL)
        assert e.value.status == L.ERR_UNSUPPORTED
        g2.close()


def test_group_band_rows_cover_and_chunk_aligned():
    from paintfe_amd.group import band_rows
    for h in (1, 63, 64, 65, 4320, 8640, 130):
        for world in (1, 2, 3, 8, 70):
            rows = [band_rows(h, world, k) for k in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == h
            for (a, b), (c, d) in zip(rows, rows[1:]):
                assert b == c and a <= b
            assert all(a % 64 == 0 or a == h for a, _ in rows)


def test_group_errors():
    from paintfe_amd import _lib as L
    from paintfe_amd.group import GpuGroup
    g = GpuGroup([0, 0])
    with pytest.raises(L.PfxError):
        g.flatten_blur([(0, 1.0, True, 0)], 2.0)  # no document yet
    g.set_document(64, 64, 1)
    with pytest.raises(L.PfxError):
        g.flatten_blur([(3, 1.0, True, 0)], 2.0)  # layer outside the document
    g.close()


def test_group_back_to_back_calls_with_overlapping_gathers():
    """the all-gather's pushes run on per-member copy streams: calls enqueued back to back — different layer selections, with and
    without the blur (the un-blurred gather reads the buffer the next flatten overwrites) — must each leave the right image behind"""
    from paintfe_amd import GpuRenderer
    from paintfe_amd.group import GpuGroup
    w, h, n, world = 384, 520, 6, 3
    stack, modes, opac = I.layer_stack(w, h, n, seed=77)
    r = GpuRenderer(0)
    for k in range(n):
        r.ensure_layer_texture(k, stack[k], generation=1)
    g = GpuGroup(_devices(world))
    g.set_document(w, h, n)
    for k in range(n):
        g.upload_layer(k, stack[k])
    plans = [(list(range(n)), 3.0), ([0, 2, 4], 0.0), ([5, 1, 3, 0], 0.0), (list(range(n)), 6.0), ([1, 2], 2.0), ([4, 3], 0.0)]
    for upto in (1, 3, 4, 6):  # enqueue the first `upto` calls without a host synchronisation in between; check the last one
        for sel, sigma in plans[:upto]:
            g.flatten_blur([(k, float(opac[k]), True, int(modes[k])) for k in sel], sigma, all_gather=True)
        sel, sigma = plans[upto - 1]
        ref = r.composite(w, h, [(k, float(opac[k]), True, int(modes[k])) for k in sel])
        if sigma > 0:
            ref = r.blur_rgba(ref, sigma)
        for k in range(world):
            assert np.array_equal(g.download_gathered(k), ref), f"after {upto} calls: gathered image on member {k}"
        assert np.array_equal(g.download(), ref)
    g.close()


@pytest.mark.parametrize("transport", ["peer", "staged"])
@pytest.mark.parametrize("world", [2, 3, 5])
def test_group_warps_of_a_sharded_document(world, transport):
    """the flattened bands are all-gathered (every member holds the whole source) and every member warps its band of the output —
    displacement field and fused Catmull-Rom mesh warp, ragged bands, back to back; equal to flatten + warp on one GPU (and to the oracle)"""
    from paintfe_amd import GpuRenderer
    from paintfe_amd.group import GpuGroup
    from tests import oracle_lib as O
    w, h, n = 208, 333, 4
    stack, modes, opac = I.layer_stack(w, h, n, seed=200 + world)
    infos = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
    rng = np.random.default_rng(world)
    disp = (rng.standard_normal((h, w, 2)) * 9.0).astype(np.float32)
    orig, deformed = I.jittered_mesh(6, 6, w, h, seed=world)
    r = GpuRenderer(0)
    for k in range(n):
        r.ensure_layer_texture(k, stack[k], generation=1)
    flat = r.composite(w, h, infos)
    ref_d = r.warp_displacement(flat, disp)
    ref_m = r.warp_mesh_catmull_rom(flat, orig, deformed, 6, 6)
    assert np.array_equal(ref_d, O.warp_displacement(flat, disp))
    g = GpuGroup(_devices(world))
    g.set_transport({"peer": g.PEER, "staged": g.STAGED}[transport])
    g.set_document(w, h, n)
    for k in range(n):
        g.upload_layer(k, stack[k])
    for _ in range(2):
        g.flatten_warp_displacement(infos, disp)
        assert np.array_equal(g.download(), ref_d), "displacement warp"
        g.flatten_warp_mesh(infos, orig, deformed, 6, 6)
        assert np.array_equal(g.download(), ref_m), "mesh warp"
    g.flatten_warp_mesh(infos, None, deformed, 6, 6)          # uniform original grid (the `_fast` form, transform.rs:1735-1736)
    fast = r.warp_displacement(flat, r.generate_displacement(deformed, 6, 6, w, h))
    assert np.array_equal(g.download(), fast), "mesh warp, uniform original grid"
    g.flatten_blur(infos, 2.0, all_gather=True)               # the filter pipeline still works behind a warp (buffers, events)
    assert np.array_equal(g.download(), r.blur_rgba(flat, 2.0))
    g.close()


def test_group_watchdog_names_the_late_member(monkeypatch):
    """a peer that never sends must become an error, not a hang.  The last member's stream sleeps in a host function in front of the
    event its neighbours wait for (PFX_GROUP_TEST_STALL_MS); with a 60 ms watchdog the call returns PFX_ERR_HIP naming the busy members and the halo
    transfers they take part in; once the sleep is over the group works again"""
    from paintfe_amd import GpuRenderer
    from paintfe_amd._lib import PfxError, ERR_HIP
    from paintfe_amd.group import GpuGroup
    w, h, n, world = 128, 400, 3, 3
    stack, modes, opac = I.layer_stack(w, h, n, seed=77)
    infos = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
    monkeypatch.setenv("PFX_GROUP_TEST_STALL_MS", "700")      # read once, when the group is created: every pipeline call of THIS group stalls its last member
    g = GpuGroup(_devices(world))
    monkeypatch.delenv("PFX_GROUP_TEST_STALL_MS")
    g.set_document(w, h, n)
    for k in range(n):
        g.upload_layer(k, stack[k])
    g.flatten_blur(infos, 3.0)                                 # warm: allocations, first launches (no watchdog yet: it simply takes 700 ms)
    g.synchronize()
    g.set_watchdog(60, 1)
    with pytest.raises(PfxError) as e:
        g.flatten_blur(infos, 3.0)
    assert e.value.status == ERR_HIP
    msg = str(e.value)
    assert "did not finish within 60 ms" in msg and "member(s)" in msg and "<-" in msg and "2 (device" in msg, msg
    with pytest.raises(PfxError):
        g.synchronize_timeout(5)                              # still sleeping
    g.synchronize()                                           # ... and over
    r = GpuRenderer(0)
    for k in range(n):
        r.ensure_layer_texture(k, stack[k], generation=1)
    g.flatten_blur(infos, 3.0)                                 # the watchdog's budget of calls is spent; the group is intact
    assert np.array_equal(g.download(), r.blur_rgba(r.composite(w, h, infos), 3.0))
    g.synchronize_timeout(2000)
    g.close()


def test_warp_band_entry_points_equal_the_rows_of_the_whole_image_call():
    """pfx_warp_displacement_band_dev / pfx_warp_mesh_catmull_rom_band_dev: ragged bands of the output, the source whole — bit-identical to the same
    rows of the whole-image calls (what paintfe_amd.sharding.warp_sharded and pfx_group_flatten_warp_* rest on)"""
    import torch
    from paintfe_amd import GpuRenderer
    from paintfe_amd.group import band_rows
    w, h = 300, 333
    r = GpuRenderer(0)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    dev = torch.device("cuda", 0)
    img = I.random_rgba(w, h, 3)
    rng = np.random.default_rng(4)
    disp = (rng.standard_normal((h, w, 2)) * 11.0).astype(np.float32)
    orig, deformed = I.jittered_mesh(6, 6, w, h, seed=8)
    ref_d = r.warp_displacement(img, disp)
    ref_m = r.warp_mesh_catmull_rom(img, orig, deformed, 6, 6)
    src = torch.from_numpy(img).to(dev)
    for world in (1, 3, 6):
        for k in range(world):
            y0, y1 = band_rows(h, world, k)
            if y1 == y0:
                continue
            dband = torch.from_numpy(np.ascontiguousarray(disp[y0:y1])).to(dev)
            out = torch.zeros((y1 - y0, w, 4), dtype=torch.uint8, device=dev)
            r.warp_displacement_band_dev(src.data_ptr(), w, h, dband.data_ptr(), w, y1 - y0, out.data_ptr(), y0)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), ref_d[y0:y1]), f"displacement band {k} of {world}"
            r.warp_mesh_catmull_rom_band_dev(src.data_ptr(), orig, deformed, 6, 6, w, h, out.data_ptr(), y0, y1 - y0)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), ref_m[y0:y1]), f"mesh band {k} of {world}"
    r.close()
