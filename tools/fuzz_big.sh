#!/bin/bash
# a larger campaign of the four fuzz kinds against the live oracle (seeds away from every recorded campaign)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=${1:-700000}
mkdir -p gpurun_out/fuzz_big
timeout 1500 python tools/fuzz_parity.py 1500 $((91000 + O)) 2>&1 | grep -v amdgpu | grep "BAD\|cases," > gpurun_out/fuzz_big/plain.txt
timeout 1500 python tools/fuzz_parity.py 1200 $((92000 + O)) similar 2>&1 | grep -v amdgpu | grep "BAD\|cases," > gpurun_out/fuzz_big/similar.txt
timeout 1500 python tools/fuzz_ais.py $((93000 + O)) 900 2>&1 | grep -v amdgpu | grep "BAD\|cases\|bad" | tail -5 > gpurun_out/fuzz_big/ais.txt
timeout 1500 python tools/fuzz_streamed.py 2500 $((94000 + O)) 2>&1 | grep -v amdgpu | grep "BAD\|cases" > gpurun_out/fuzz_big/streamed.txt
for f in gpurun_out/fuzz_big/*.txt; do echo "== $f"; tail -n 2 $f; done
