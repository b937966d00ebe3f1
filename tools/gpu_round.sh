#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python tools/ab_ragged.py rdplain 10000000 12
RAGGED_MOSTLY=150 python tools/ab_ragged.py rdplain 10000000 12
