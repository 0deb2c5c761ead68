# diagnostic: where the persistent DyNCA kernel's step time goes (NCAHIP_PERSIST_DBG knobs, see nca_kernels.h; results of knob runs are not valid states)
for d in ${@:-0 1 2 4 16 17}; do echo "dbg=$d"; NCAHIP_PERSIST_DBG=$d python tools/bench_paths.py video 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l)
    if not d['two_scale']: print('  C',d['C'],'us/step',round(d['us_per_step'],2),'min',round(d['min_us_per_step'],2))"; done
