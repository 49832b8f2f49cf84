#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for f in 100 150 200 300; do echo "-- fill $f"; python tools/time_box.py 0 box_strip_fill=$f 2>/dev/null | grep "r=[59]\.\|r=16\|r=24\|r=48"; done
