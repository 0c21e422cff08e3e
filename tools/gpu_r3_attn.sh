#!/bin/bash
# round 3, GPU call 1: correctness of every attention dispatch arm + in-process A/B at the bench shape (40 heads) and at
# the 8-way Ulysses shard (5 heads), then the new SP tests.  Writes gpurun_out/r3_*.log.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 300 tools/kernel_check attnarms > gpurun_out/r3_attnarms.log 2>&1; echo "attnarms rc=$?"
tail -3 gpurun_out/r3_attnarms.log
ARMS=("attn_fast=1" "attn_fast=0,attn_ref=1" "attn_fast=0,attn_ref=2" "attn_w4=0,attn_fast=0")
timeout 200 tools/kernel_check attnx 40 "${ARMS[@]}" > gpurun_out/r3_attnx40.log 2>&1; echo "attnx40 rc=$?"; cat gpurun_out/r3_attnx40.log
timeout 200 tools/kernel_check attnx 5 "${ARMS[@]}" > gpurun_out/r3_attnx5.log 2>&1; echo "attnx5 rc=$?"; cat gpurun_out/r3_attnx5.log
timeout 900 python -m pytest tests/test_gpu_sp.py -x -q -s > gpurun_out/r3_pytest_sp.log 2>&1; echo "pytest sp rc=$?"; tail -15 gpurun_out/r3_pytest_sp.log
