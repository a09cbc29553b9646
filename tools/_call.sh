python tools/probe_cull.py 2>&1 | tail -3
