cd rvdd-release_amd
for v in r03 bfp11; do cp librvdd_hip_$v.so librvdd_hip.so; (cd .. && timeout -k 10 300 python bench.py --steps 3 --warmup 2 --cpu-frames 0 --no-exact-ab --no-other-configs --all-kernel-events 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['value'])
        for n,k in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['total_ms']): print('   %-45s n=%5d avg=%8.1f us total=%8.1f ms'%(n,k['launches'],k['avg_us'],k['total_ms']))
"); done
cp librvdd_hip_bfp11.so librvdd_hip.so
