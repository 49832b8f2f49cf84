#!/bin/bash
# tools/r06_vmap_ab.sh — bit-exact fused Gaussian: the vertical pass's lane -> column map (PFX_GF_VMAP; libpfx_vmap0.so = the old map, built by
# tools/build_variant.sh vmap0 k_gauss_exact.hip -DPFX_GF_VMAP=0), alternating processes on ONE box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r06_vmap; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_effects.py tests/test_gpu_chain.py -q -m gpu -x -k "gauss or blur or sharpen or glow or shadow or chain" 2>&1 | tail -2 | tee $OUT/parity.txt
cat > /tmp/vmap_time.py <<'PY'
import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream); r.set_exact(True)
w, h = 7680, 4320
src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
col = (0, 0, 0, 255)
def ops():
    return [("exact gaussian sigma=1", lambda: r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, 1.0)),
            ("exact gaussian sigma=3", lambda: r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, 3.0)),
            ("exact gaussian sigma=4", lambda: r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, 4.0)),
            ("exact gaussian sigma=5.3", lambda: r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, 5.3)),
            ("sharpen 1, 1", lambda: r.sharpen_dev(src.data_ptr(), dst.data_ptr(), w, h, 1.0, 1.0)),
            ("glow 3, 0.5", lambda: r.glow_dev(src.data_ptr(), dst.data_ptr(), w, h, 3.0, 0.5))]
for name, f in ops():
    for _ in range(30): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): f()
    b.record(); torch.cuda.synchronize()
    print(f"{sys.argv[1]:8s} {name:26s} {a.elapsed_time(b) / 50:.4f} ms")
PY
for rnd in 1 2 3; do
  PFX_LIB_PATH=$ROOT/paintfe_amd/libpfx_vmap0.so python /tmp/vmap_time.py old-map 2>&1 | grep -v amdgpu.ids
  python /tmp/vmap_time.py new-map 2>&1 | grep -v amdgpu.ids
done | tee $OUT/ab.txt
