#!/bin/bash
# build both libraries; non-zero exit on any compile error (so that `tools/b.sh && gpurun ...` never ships a stale library)
cd "$(dirname "$0")/.."
out=$(python -m pymht_amd.build 2>&1); rc=$?
echo "$out" | grep -E "error|Error" -A6 | head -40
[ $rc -eq 0 ] && ! echo "$out" | grep -q "error:" && echo "build ok"
