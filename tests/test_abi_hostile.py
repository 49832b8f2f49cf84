"""The C ABI under hostile arguments (include/pfx.h: "No C++ exception leaves the library"; every entry point reports bad arguments as a status).
The prototypes are parsed from the header, so a new entry point is covered the day it is declared.

* no GPU needed: every entry point that takes a context is called with a NULL context and zero / NULL everything else — it must return an error status
  (never PFX_OK, never a fault); the `_free` / `_destroy` functions must accept NULL.
* GPU: the same sweep with a LIVE context (NULL buffers, zero sizes): an error status or a documented no-op, never a fault, and the context still computes the
  right pixels afterwards.

Each sweep runs in a child process: a fault would otherwise take the test runner with it, and the child prints the entry point it is about to call."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCALARS = {"int": C.c_int, "int32_t": C.c_int32, "uint32_t": C.c_uint32, "uint64_t": C.c_uint64, "int64_t": C.c_int64, "size_t": C.c_size_t, "float": C.c_float,
           "double": C.c_double, "uint8_t": C.c_uint8, "uint16_t": C.c_uint16, "unsigned": C.c_uint, "char": C.c_char, "pfx_chain_kind": C.c_int}


def prototypes():
    text = open(os.path.join(ROOT, "include", "pfx.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    out = []
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(pfx_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret or "(" in params:     # function-pointer typedefs / callbacks in parameter lists are out of this sweep
            continue
        plist = [] if params in ("", "void") else [p.strip() for p in params.split(",")]
        out.append((ret, name, plist))
    return out


def zero_for(param):
    """a ctypes zero of the parameter's type: NULL for pointers and arrays, 0 for scalars; None = a by-value struct (the entry point is skipped)"""
    if "*" in param or "[" in param:
        return C.c_void_p(None)
    words = [w for w in re.sub(r"\bconst\b|\bunsigned\b(?=\s+\w+\s+\w)", " ", param).split()]
    tname = words[0] if len(words) >= 1 else ""
    if param.split()[0] == "unsigned" and len(param.split()) == 2:
        return C.c_uint(0)
    if tname == "enum" and len(words) >= 2:
        return C.c_int(0)
    return SCALARS[tname](0) if tname in SCALARS else None


def sweep(live_ctx, start=0):
    """runs in the child: prints one line per call (flushed before the call), then SWEEP-OK; `start` skips the first entry points (the parent restarts a child
    that faulted behind the entry point that killed it, so one run lists every offender)"""
    lib = C.CDLL(os.environ.get("PFX_LIB_PATH") or os.path.join(ROOT, "paintfe_amd", "libpfx.so"))
    lib.pfx_last_error.restype = C.c_char_p
    ctx = C.c_void_p(None)
    if live_ctx:
        assert lib.pfx_ctx_create(C.c_int(0), C.byref(ctx)) == 0
    called = ok_returns = 0
    for index, (ret, name, plist) in enumerate(prototypes()):
        if index < start or name in ("pfx_ctx_create", "pfx_ctx_destroy", "pfx_abi_version") or not plist:
            continue
        takes_ctx = plist[0].replace(" ", "").startswith("pfx_ctx*")
        if live_ctx and not takes_ctx:
            continue
        args = [zero_for(p) for p in plist]
        if any(a is None for a in args):
            print("SKIP by-value struct:", name, flush=True)
            continue
        if live_ctx:
            args[0] = ctx
        fn = getattr(lib, name)
        is_status = ret.split()[-1] == "int" and "*" not in ret
        fn.restype = C.c_int if is_status else (None if ret.strip() == "void" else C.c_void_p)
        print("CALL", name, index, flush=True)
        st = fn(*args)
        called += 1
        if is_status and st == 0:
            ok_returns += 1
            print("RETURNED-OK", name, flush=True)
        if live_ctx and takes_ctx and hasattr(lib, "pfx_synchronize"):
            lib.pfx_synchronize(ctx)    # a launch that should not have happened faults here, under this entry point's name
    if live_ctx:
        # the context survived: 64 x 64 invert through the host-buffer entry point
        import numpy as np
        img = (np.arange(64 * 64 * 4, dtype=np.uint32) * 7 % 251).astype(np.uint8).reshape(64, 64, 4)
        out = np.zeros_like(img)
        assert lib.pfx_invert_rgba(ctx, img.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_uint32(64), C.c_uint32(64)) == 0
        want = img.copy()
        want[..., :3] = 255 - want[..., :3]
        assert (out == want).all()
        lib.pfx_ctx_destroy(ctx)
    lib.pfx_ctx_destroy(C.c_void_p(None))
    print("SWEEP-OK", called, ok_returns, flush=True)


DIM_NAMES = {"w", "h", "sw", "sh", "new_w", "new_h", "canvas_w", "canvas_h", "src_w", "src_h", "width", "height"}


def oversize_sweep(start=0):
    """child, GPU: live context, VALID (small) buffers behind every pointer, every image dimension 20 000 (4 * 10^8 pixels: over the 256 Mpx document limit of
    tiled_image.rs:15-26).  An entry point with a width / height pair must refuse before it touches a buffer — one that did not would walk off a 1 MB allocation."""
    import numpy as np
    lib = C.CDLL(os.environ.get("PFX_LIB_PATH") or os.path.join(ROOT, "paintfe_amd", "libpfx.so"))
    ctx = C.c_void_p(None)
    assert lib.pfx_ctx_create(C.c_int(0), C.byref(ctx)) == 0
    host = np.zeros(1 << 20, np.uint8)
    dev = C.c_void_p(None)
    assert lib.pfx_dev_alloc(ctx, C.c_size_t(1 << 20), C.byref(dev)) == 0 and lib.pfx_dev_memset(ctx, dev, C.c_int(0), C.c_size_t(1 << 20)) == 0
    for index, (ret, name, plist) in enumerate(prototypes()):
        if index < start or not plist or not plist[0].replace(" ", "").startswith("pfx_ctx*") or ret.split()[-1] != "int" or "*" in ret:
            continue
        pnames = [re.sub(r"\[.*\]", "", p).replace("*", " ").split()[-1] for p in plist]
        if len(DIM_NAMES & set(pnames)) < 2:
            continue
        args = [ctx]
        for p, pn in list(zip(plist, pnames))[1:]:
            if "*" in p or "[" in p:
                args.append(dev if pn.endswith("_dev") else host.ctypes.data_as(C.c_void_p))
            elif pn in DIM_NAMES:
                args.append(C.c_uint32(20000))
            else:
                z = zero_for(p)
                args.append(type(z)(1))
        fn = getattr(lib, name)
        fn.restype = C.c_int
        print("CALL", name, index, flush=True)
        st = fn(*args)
        lib.pfx_ctx_synchronize(ctx)
        if st == 0:
            print("RETURNED-OK", name, flush=True)
    print("SWEEP-OK", flush=True)


# entry points for which PFX_OK on all-zero arguments is the documented behaviour (nothing to do is not an error)
NOOP_OK_NULL_CTX = set()
NOOP_OK_LIVE_CTX = {"pfx_ctx_set_exact", "pfx_ctx_set_stream", "pfx_ctx_synchronize", "pfx_timing_enable", "pfx_timing_reset",   # settings / no arguments to get wrong
                    "pfx_layer_clear", "pfx_layer_remove", "pfx_warp_invalidate_source", "pfx_dev_free", "pfx_host_free",                      # removing what is not there
                    "pfx_dev_upload", "pfx_dev_download", "pfx_dev_memset"}                                                     # zero bytes


def run_child(mode):
    lines, faulted, start = [], [], 0
    for _ in range(40):   # every restart is one offender: far fewer than this many are tolerable
        call = f"oversize_sweep({start})" if mode == "oversize" else f"sweep({mode}, {start})"
        r = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); from tests.test_abi_hostile import sweep, oversize_sweep; {call}"],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONFAULTHANDLER="1"))
        out = r.stdout.splitlines()
        lines += out
        if r.returncode == 0 and out and out[-1].startswith("SWEEP-OK"):
            break
        last_call = next((l for l in reversed(out) if l.startswith("CALL ")), None)
        assert last_call is not None, f"child exit {r.returncode} before any call\n{r.stderr[-3000:]}"
        faulted.append(f"{last_call.split()[1]} (child exit {r.returncode})")
        start = int(last_call.split()[2]) + 1
    assert not faulted, f"entry points that took the process down: {faulted}"
    return lines


def test_header_parses_to_the_exported_prototypes():
    protos = prototypes()
    names = {n for _, n, _ in protos}
    assert len(protos) >= 150 and {"pfx_flatten_dev", "pfx_chain_dev", "pfx_script_execute", "pfx_group_create", "pfx_png_decode_mem"} <= names


def test_null_context_is_an_error_status_everywhere_and_never_a_fault():
    lines = run_child(False)
    n_called = sum(1 for l in lines if l.startswith("CALL "))
    assert n_called >= 120, n_called
    returned_ok = {l.split()[1] for l in lines if l.startswith("RETURNED-OK")}
    ctx_takers = {n for _, n, p in prototypes() if p and p[0].replace(" ", "").startswith("pfx_ctx*")}
    wrong = sorted((returned_ok & ctx_takers) - NOOP_OK_NULL_CTX)
    assert not wrong, f"PFX_OK with a NULL context: {wrong}"


@pytest.mark.gpu
def test_live_context_with_null_buffers_and_zero_sizes_never_faults():
    lines = run_child(True)
    assert sum(1 for l in lines if l.startswith("CALL ")) >= 100
    returned_ok = {l.split()[1] for l in lines if l.startswith("RETURNED-OK")}
    unexpected = sorted(returned_ok - NOOP_OK_LIVE_CTX)
    assert not unexpected, f"PFX_OK for all-zero arguments, not on the documented no-op list: {unexpected}"


@pytest.mark.gpu
def test_dimensions_over_the_document_limit_are_refused_before_any_buffer_is_touched():
    lines = run_child("oversize")
    assert sum(1 for l in lines if l.startswith("CALL ")) >= 60
    accepted = sorted(l.split()[1] for l in lines if l.startswith("RETURNED-OK"))
    assert not accepted, f"accepted a 20000 x 20000 image: {accepted}"
