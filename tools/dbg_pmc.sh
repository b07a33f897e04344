cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc1 -o p -- python /root/repo/bench.py --config cfg3 --cpu-scans 0 --sectors 0 --steps 40 --warmup 8 --pmc off > /tmp/pmc1.out 2> /tmp/pmc1.err; echo rc $?
grep -v "amdgpu.ids\|simple_timer" /tmp/pmc1.err | tail -30 | cut -c1-500
tail -c 300 /tmp/pmc1.out
