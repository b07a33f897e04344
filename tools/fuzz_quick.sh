#!/bin/bash
# a shorter set of the fuzz campaigns of tools/fuzz_all.sh (about 25 minutes on one MI355X): other seeds, same checks
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/fuzzq
timeout 540 python tools/fuzz_parity.py 300 191000 2>&1 | grep -v amdgpu | grep "BAD\|cases," > gpurun_out/fuzzq/plain.txt
timeout 420 python tools/fuzz_parity.py 200 192000 similar 2>&1 | grep -v amdgpu | grep "BAD\|cases," > gpurun_out/fuzzq/similar.txt
timeout 540 python tools/fuzz_streamed.py 350 194000 2>&1 | grep -v amdgpu | grep "BAD\|cases" > gpurun_out/fuzzq/streamed.txt
timeout 240 python tools/fuzz_ais.py 193000 300 2>&1 | grep -v amdgpu | grep "BAD\|cases\|bad" | tail -5 > gpurun_out/fuzzq/ais.txt
tail -3 gpurun_out/fuzzq/*.txt
