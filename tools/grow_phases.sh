#!/bin/bash
# per-phase stamps of the target workgroups (tools/grow_profile.py): a scratch copy of the library built with -DMHT_GROW_STAMPS
cd ${GRAFT_REPO_ROOT:-/root/repo}
MHT_EXTRA_HIPCC_FLAGS=-DMHT_GROW_STAMPS python - <<'PY'
import os, subprocess, sys
sys.path.insert(0, ".")
from pymht_amd import build as b
srcs = [os.path.join(b.CSRC, s) for s in b.SOURCES]
out = b.LIB + ".stamps"
subprocess.check_call(["hipcc"] + b.FLAGS + ["-DMHT_GROW_STAMPS"] + srcs + ["-o", out])
PY
MHT_LIB_VARIANT=.stamps python tools/grow_profile.py raw 2>&1 | grep -v amdgpu | tail -12
