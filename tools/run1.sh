source tools/ab_merge.sh
timeout 300 python -m pytest tests/test_overlap_golden_gpu.py -x -q -m gpu > gpurun_out/s2/ovl_tests.log 2>&1; tail -n 2 gpurun_out/s2/ovl_tests.log
B="python bench.py --cpu-scans 0 --pmc off --sectors 0 --steps 400 --warmup 40"
for rep in 1 2; do
run g258_$rep X=1
run g224_$rep MHT_BLP_GRID=224
run g192_$rep MHT_BLP_GRID=192
run g160_$rep MHT_BLP_GRID=160
run g128_$rep MHT_BLP_GRID=128
done
