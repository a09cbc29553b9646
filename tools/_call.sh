python bench.py --no-extras --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); r=b['roofline']
        print('DEVICE value %.0f frac %.4f kernel %.1f first %.1f frac_first %.4f attempts %s' % (b['value'], r['frac'], r['kernel_avg_us'], r['kernel_us_first_allocation'], r['frac_first_allocation'], r['output_placement']['store_GBs_per_attempt']))"
