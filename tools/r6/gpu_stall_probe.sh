mkdir -p gpurun_out; export TMPDIR=/tmp
for P in ${PS:-none create steps30 steps120 steps120sync}; do
echo "== pre=$P"
KVFE_HOST_PROF=1 timeout 200 python tools/r6/stall_probe.py $P 2>&1 | python -c "
import sys
for l in sys.stdin:
    if 'per call' in l:
        v=[float(x) for x in l.split(':')[1].split()]
        print('   stalls', [(i,int(x)) for i,x in enumerate(v) if x>1000], 'of', len(v))
    elif 'steps,' in l: print('  ', l.strip())
"
done
