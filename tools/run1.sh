mkdir -p gpurun_out/s2
ext2() { python -c "
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value']), [(m.get('sectors_per_gpu'), round(m.get('scans_per_sec',0)), m.get('ok'), (m.get('error') or '')[:80]) for m in d['multi_sector_all']])
except Exception as e: print(sys.argv[1], 'FAILED', e)
" $1; }
run2() { n=$1; s=$2; shift; shift; env "$@" timeout 300 python bench.py --cpu-scans 0 --pmc off --steps 20 --warmup 5 --sectors $s > gpurun_out/s2/$n.json 2>gpurun_out/s2/$n.err; ext2 gpurun_out/s2/$n.json; }
run2 c_def 4,16 X=1
run2 c_1024 4,16 MHT_BLP_CAPS=1024,512,128
run2 c_1536 4,16 MHT_BLP_CAPS=1536,768,192
run2 c_1280 4,16 MHT_BLP_CAPS=1280,640,160
