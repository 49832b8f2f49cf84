#!/bin/bash
# round-4 lab run 1: wide-load variants and in-mix VALU costs
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4lab1; mkdir -p $OUT
timeout 300 tools/lab/wide_load 8 > $OUT/wide_load.jsonl 2>&1; echo "wide_load rc=$?"
cat $OUT/wide_load.jsonl
timeout 120 tools/lab/valu_mix 6 > $OUT/valu_mix_6.jsonl 2>&1; echo "valu_mix rc=$?"
cat $OUT/valu_mix_6.jsonl
timeout 120 tools/lab/valu_mix 8 > $OUT/valu_mix_8.jsonl 2>&1
cat $OUT/valu_mix_8.jsonl
