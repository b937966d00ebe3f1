#!/bin/bash
# rocprofv3 kernel stats of the binned Bloom insert (tools/bloom_binned_prof.py); usage on the GPU box: bash tools/bbp.sh <tag> [args]
TAG=${1:-bbp}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o kt -- python tools/bloom_binned_prof.py "$@" > $OUT/run.log 2>&1
grep "^insert" $OUT/run.log
f=$(find $OUT/t -name "*kernel_stats.csv"); cp $f $OUT/kernel_stats.csv
python - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e6:8.3f} ms max {float(r['MaxNs'])/1e6:8.3f} total {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
