for b in 32 64 128 256 512; do python bench.py --config vitb --batch $b --steps 30 --warmup 5 --no-cpu --no-extra 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print($b, d['value'], d['ms_per_step'])"; done
