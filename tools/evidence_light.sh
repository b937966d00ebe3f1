#!/bin/bash
# The short form of tools/evidence_round.sh (no counter passes, no sweeps): bench lines + kernel stats per config, the
# consumers, the GPU suite and smoke -- about half an hour of box time.   bash tools/evidence_light.sh r04
set -u
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
SKIP_PMC=1 timeout 1800 bash tools/profile_round.sh "$TAG" > "$OUT/profile_round.log" 2>&1
timeout 2400 python -m pytest tests -q -m gpu > "$OUT/pytest_gpu.txt" 2>&1
tail -3 "$OUT/pytest_gpu.txt"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" >> "$OUT/pytest_gpu.txt" 2>&1
tail -1 "$OUT/pytest_gpu.txt"
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*_agent_info.csv" -delete
du -sh "$OUT"
