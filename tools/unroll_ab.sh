#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for shape in 100,64,3:30000000 151,31,1:20000000 100,64,1:30000000 150,100,1:20000000 101,31,3:12000000; do
  s=${shape%%:*}; n=${shape##*:}
  echo "== shape $s reads $n"
  AB_PROBED=1 ABLATE_SHAPE=$s timeout 600 python tools/ab_multi.py "u2,u4" $n 8 2>&1 | grep -v probed
done
