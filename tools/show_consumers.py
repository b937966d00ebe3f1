#!/usr/bin/env python3
"""One line per entry of bench.py's `consumers` object: python tools/show_consumers.py <json file with the line>"""
import json
import sys
line = [l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
d = d.get("consumers", d)
for k, v in d.items():
    if isinstance(v, dict) and "value" in v:
        r = v.get("roofline") or {}
        print(f"{k:34s} {v['value'] / 1e9:8.1f} G k-mers/s {v['ms']:8.2f} ms ok={v.get('ok')} kernel={r.get('kernel')} "
              f"{r.get('kernel_ms') or 0:.2f} ms frac={r.get('frac', 0):.3f}")
