source tools/ab_merge.sh
timeout 600 python -m pytest tests/test_overlap_golden_gpu.py tests/test_tracker_gpu.py tests/test_forest_edge_gpu.py tests/test_cluster_blp_gpu.py -x -q -m gpu > gpurun_out/s2/hint_tests.log 2>&1; tail -n 2 gpurun_out/s2/hint_tests.log
B="python bench.py --cpu-scans 0 --pmc off --sectors 0 --steps 400 --warmup 40"
for rep in 1 2 3; do
run hint_$rep X=1
run nohint_$rep MHT_BLP_GRID_HINT=0
done
B="python bench.py --cpu-scans 0 --pmc off --sectors 0 --steps 20 --warmup 5"
for rep in 1 2 3; do
run hint20_$rep X=1
run nohint20_$rep MHT_BLP_GRID_HINT=0
done
