#!/bin/bash
# fuzz campaign with every ILP on the HBM storage policy (MHT_BLP_FORCE_HBM=1: the giant clusters' code path -- batched sweeps, a thread per
# row bit, column ranges from LDS) against the live oracle, then the ordinary kinds on new seeds
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=${1:-1100000}
mkdir -p gpurun_out/fuzz_hbm
MHT_BLP_FORCE_HBM=1 timeout 900 python tools/fuzz_parity.py 400 $((91000 + O)) 2>&1 | grep -v amdgpu | grep "BAD\|cases," > gpurun_out/fuzz_hbm/plain_hbm.txt
MHT_BLP_FORCE_HBM=1 timeout 900 python tools/fuzz_streamed.py 300 $((94000 + O)) 2>&1 | grep -v amdgpu | grep "BAD\|cases" > gpurun_out/fuzz_hbm/streamed_hbm.txt
timeout 900 python tools/fuzz_parity.py 300 $((95000 + O)) 2>&1 | grep -v amdgpu | grep "BAD\|cases," > gpurun_out/fuzz_hbm/plain.txt
for f in gpurun_out/fuzz_hbm/*.txt; do echo "== $f"; tail -n 2 $f; done
