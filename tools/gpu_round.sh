#!/bin/bash
cd "$GRAFT_REPO_ROOT"
AB_PROBED=1 ABLATE_SHAPE=150,31,1 python tools/ab_multi.py "nohash,nohashnoload" 100000000 8 | cut -c1-125
AB_PROBED=1 ABLATE_SHAPE=150,31,1 python tools/ab_multi.py "nohash,nohashnoload" 100000000 8 | cut -c1-125
