#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
