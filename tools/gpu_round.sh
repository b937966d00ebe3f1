#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e9,1), round(d['roofline']['frac'],4), {k:(round(v['value']/1e9,1), round(v['frac'],3), v['verify_ok']) for k,v in d['secondary'].items()})"
