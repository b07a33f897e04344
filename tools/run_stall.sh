for i in 1 2 3 4 5; do MHT_STALL_DEBUG=1 timeout 350 python bench.py --steps 20 --warmup 5 --cpu-scans 0 --pmc off > gpurun_out/b_ev.json 2> gpurun_out/b.err; echo run $i rc $?; grep -v amdgpu.ids gpurun_out/b.err | cut -c1-300; done
timeout 250 python tools/api_timeline.py 2>&1 | tail -2; timeout 250 python tools/api_profile.py 416 2>&1 | grep "per scan\|scans:"
timeout 600 python -m pytest tests/test_tracker_gpu.py tests/test_initiator_gpu.py tests/test_forest_edge_gpu.py -x -q 2>&1 | tail -3
