#!/bin/bash
# one AIS fuzz seed against the live oracle on several builds / switches (development)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
echo "== tree build"; python tools/fuzz_ais.py $1 1 2>&1 | grep -v amdgpu | tail -2
echo "== tree build, MHT_NO_UF=1 (clustering kernel)"; MHT_NO_UF=1 python tools/fuzz_ais.py $1 1 2>&1 | grep -v amdgpu | tail -2
echo "== tree build, MHT_NO_OVERLAP=1"; MHT_NO_OVERLAP=1 python tools/fuzz_ais.py $1 1 2>&1 | grep -v amdgpu | tail -2
if [ -f pymht_amd/libmht_amd.so.old ]; then echo "== build of the round's start (dee88f3)"; MHT_LIB_VARIANT=.old python tools/fuzz_ais.py $1 1 2>&1 | grep -v amdgpu | tail -2; fi
echo "== why"; python tools/fuzz_ais_why.py $1 2>&1 | grep -v amdgpu | tail -15
} > gpurun_out/ais_seed.txt 2>&1
cat gpurun_out/ais_seed.txt
