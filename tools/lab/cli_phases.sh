#!/bin/bash
# tools/lab/cli_phases.sh — wall clock of the batch tool on a 1024 x 1024 PNG against a bare HIP process on the same box
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<'PY'
import numpy as np
from PIL import Image
import sys; sys.path.insert(0, ".")
from tests import inputs as I
Image.fromarray(I.create_test_gradient(1024, 1024), "RGBA").save("/tmp/in.png")
open("/tmp/blur.rhai", "w").write("apply_blur(4.0);\n")
open("/tmp/none.rhai", "w").write("\n")
PY
for i in 1 2 3; do TIMEFORMAT="bare HIP process %R s wall"; time tools/lab/hip_start; done
for s in /tmp/none.rhai /tmp/blur.rhai; do for i in 1 2 3; do TIMEFORMAT="$s %R s wall %U user %S sys"; time paintfe_amd/pfx -i /tmp/in.png -s $s -o /tmp/out.png; done; done
