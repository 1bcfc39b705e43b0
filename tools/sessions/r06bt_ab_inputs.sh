for i in 1 2 3; do
for v in 0 1; do
XG_SCATTER_INPUTS=$v python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['configs']
print('inputs_scattered=$v', d['value'], d['roofline']['frac'], [o['frac'] for o in c['config3']['ops']], c['box_probe']['frac'], [o['frac'] for o in c['config4']['ops']], [o['frac'] for o in c['config5']['ops']][:2], d['result_buffers'].get('rejected'))"
done; done
