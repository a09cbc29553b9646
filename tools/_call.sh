python tools/probe_array_rates.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
for k,v in d['fill_GBs'].items(): print(k, v)
print('sets', d['pattern_GBs_sets_i']); print(d['pattern_GBs_mixes'])"
