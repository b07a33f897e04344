#!/bin/bash
# AIS fuzz campaign against the live oracle (tests/fuzz_util.py::run_case_ais: states and covariances of all leaves bit for bit):
# usage: fuzz_ais_campaign.sh SEED0 COUNT PROCS -- COUNT scenarios from SEED0 on, split over PROCS processes on the one GPU
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/fuzz
S0=${1:-1000}; N=${2:-1800}; P=${3:-6}
per=$(( (N + P - 1) / P ))
for q in $(seq 0 $((P - 1))); do
  OPENBLAS_NUM_THREADS=1 timeout 2400 python tools/fuzz_ais.py $((S0 + q * per)) $per 2>&1 | grep -v amdgpu > gpurun_out/fuzz/ais_$q.txt &
done
wait
grep -h "^BAD" gpurun_out/fuzz/ais_[0-9]*.txt > gpurun_out/fuzz/ais_bad.txt
grep -h "^cases" gpurun_out/fuzz/ais_*.txt
echo "bad lines: $(wc -l < gpurun_out/fuzz/ais_bad.txt)"; head -5 gpurun_out/fuzz/ais_bad.txt
