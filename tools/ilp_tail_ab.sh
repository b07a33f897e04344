#!/bin/bash
# Development: the ILP tail instances (G20, G9) through the seam with the build in the tree and with variant builds next to it
# (pymht_amd/libmht_amd.so.<name>, made by hand; never rebuilt).  usage: tools/ilp_tail_ab.sh <out> "<variants>" [pytest]
out=gpurun_out/${1:-ilp_tail}; mkdir -p $out
case " ${2:-.base tree} " in *" tree "*) python -c "import pymht_amd._lib as l; l.load()" > /dev/null 2>&1;; esac      # (the tree's build, if the sources are newer)
for v in ${2:-.base tree}; do
  [ "$v" = "tree" ] && v=""
  echo "== variant '$v'" >> $out/ab.txt
  MHT_LIB_VARIANT=$v timeout 300 python tools/g20_time.py 2>&1 | grep -v amdgpu.ids | sort | uniq -c | sort -k1,1nr | head -${TRACE_LINES:-60} >> $out/ab.txt
  MHT_LIB_VARIANT=$v timeout 300 python tools/g9_time.py 0 2>&1 | grep no_reduce >> $out/ab.txt
done
cat $out/ab.txt
[ -n "$3" ] && MHT_AMD_NO_BUILD=1 timeout 900 python -m pytest tests/test_cluster_blp_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee $out/pytest.txt
