#!/bin/bash
# fuzz campaigns of the build against the live oracle: plain / similar-state pruning / AIS / streamed hosts
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=${1:-0}      # seed offset: a campaign on other scenarios than the recorded ones
mkdir -p gpurun_out/fuzz
timeout 900 python tools/fuzz_parity.py 500 $((91000 + O)) 2>&1 | grep -v amdgpu | grep "BAD\|cases," > gpurun_out/fuzz/plain.txt
timeout 900 python tools/fuzz_parity.py 400 $((92000 + O)) similar 2>&1 | grep -v amdgpu | grep "BAD\|cases," > gpurun_out/fuzz/similar.txt
timeout 900 python tools/fuzz_ais.py $((93000 + O)) 300 2>&1 | grep -v amdgpu | grep "BAD\|cases\|bad" | tail -5 > gpurun_out/fuzz/ais.txt
timeout 900 python tools/fuzz_streamed.py 600 $((94000 + O)) 2>&1 | grep -v amdgpu | grep "BAD\|cases" > gpurun_out/fuzz/streamed.txt
for f in gpurun_out/fuzz/*.txt; do echo "== $f"; tail -n 2 $f; done
