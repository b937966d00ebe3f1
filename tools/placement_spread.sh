#!/bin/bash
# Fresh-process default-shaped lines of bench.py (c2 at full size, 10 steps, no extras), one allocation policy per column:
#   tools/placement_spread.sh <runs> <out file> [policies: plain 8 32 64 probed]
# plain: one hipMalloc per buffer; N: NTHIP_TUNE_MALLOC_PIECES=N (buffers mapped from N MiB physical pieces); probed: three candidates
# measured per buffer (bench.py --placement 3, the default until round 4); default: bench.py as the driver runs it (20 steps)
runs=$1; out=$2; shift; shift
pol=${@:-plain 32 probed}
: > "$out"
for i in $(seq 1 $runs); do
  for p in $pol; do
    case $p in
      plain) env NTHIP_TUNE_MALLOC_PIECES=1 python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline --no-peak --no-placement --no-plain-pass > /tmp/ps.json 2>/dev/null ;;
      probed) python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline --no-peak --no-plain-pass --placement 3 > /tmp/ps.json 2>/dev/null ;;
      default) python bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline --no-peak > /tmp/ps.json 2>/dev/null ;;
      *) env NTHIP_TUNE_MALLOC_PIECES=$p python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline --no-peak --no-placement --no-plain-pass > /tmp/ps.json 2>/dev/null ;;
    esac
    python - "$i" "$p" >> "$out" <<'PY'
import json, sys
try:
    d = json.load(open('/tmp/ps.json'))
    pl = d['roofline'].get('frac_plain_alloc')
    print(f"run {sys.argv[1]:>2s} {sys.argv[2]:>7s}: {d['value']/1e9:7.1f} G k-mers/s  frac {d['roofline']['frac']:.3f}  kernel {d['roofline']['kernel_avg_ms']:.3f} ms" + (f"  (plain hipMalloc buffers, same process: {pl:.3f})" if pl else ""))
except Exception as e:
    print(f"run {sys.argv[1]} {sys.argv[2]}: failed {e}")
PY
  done
done
cat "$out"
