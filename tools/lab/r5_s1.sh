#!/bin/bash
# round-5 session 1: baseline + instruction-cache A/B + counters + thread-trace attempt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r5s1; mkdir -p $O
cd $ROOT
python tools/dle_stats.py > $O/base.txt 2>&1
python tools/lab/icache_ab.py > $O/icache_ab.txt 2>&1
bash tools/pmc_quick.sh mix - > $O/pmc_mix.txt 2>&1
bash tools/pmc_quick.sh m1 - mode=1 > $O/pmc_m1.txt 2>&1
bash tools/pmc_quick.sh m21 - mode=21 > $O/pmc_m21.txt 2>&1
bash tools/pmc_quick.sh m16 - mode=16 > $O/pmc_m16.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --att --att-target-cu 1 --kernel-trace -d $O/att -o att -- python $ROOT/tools/dle_stats.py > $O/att.log 2>&1; echo "att rc=$?" >> $O/att.log; ls -R $O/att 2>/dev/null | head -30 >> $O/att.log)
tail -3 $O/base.txt; cat $O/icache_ab.txt | tail -2; cat $O/pmc_mix.txt $O/pmc_m1.txt $O/pmc_m21.txt $O/pmc_m16.txt | grep -v "^$" | tail -80; tail -15 $O/att.log
