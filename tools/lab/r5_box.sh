#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "box" 2>&1 | tail -5
python tools/time_box.py 0 2>/dev/null | grep "r=[5689]\|r=16\|r=24\|r=48"
echo "-- two-pass"; python tools/time_box.py 0 box_strip=0 2>/dev/null | grep "r=[59]\.\|r=24\|r=48"
for f in 50 200; do echo "-- fill $f"; python tools/time_box.py 0 box_strip_fill=$f 2>/dev/null | grep "r=[59]\.\|r=24\|r=48"; done
