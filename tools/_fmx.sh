for n in x1 x2 x3; do echo "== $n"; NCAHIP_LIB=$PWD/video-stylization-with-nca_amd/libncahip_$n.so timeout -k 10 120 python tools/bench_paths.py cond_train 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['storage'], 'bwd us/step %.1f' % d['bwd_us_per_step'])
"; done
