#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for rep in 1 2 3; do for x in 0 1; do echo -n "mesh_xcd=$x "; python tools/time_mesh.py mesh_xcd=$x 2>/dev/null | grep "fused"; done; done
timeout 600 python -m pytest tests -m gpu -x -q -k "mesh or warp" 2>&1 | grep -E "passed|failed"
