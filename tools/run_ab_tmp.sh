run() { python bench.py --config var --no-cpu-baseline --no-peak --no-secondary --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e9,1), round(d['ms_per_step'],3), round(d['roofline']['frac'],3), d['roofline']['kernel_avg_ms'])"; }
run default
for c in 7 11 13 15; do NTHIP_TUNE_READS_RUN_LEN=$c run "C=$c"; done
for r in 16 24 48 64; do NTHIP_TUNE_READS_PER_TILE=$r run "R=$r"; done
for w in 8 12; do NTHIP_TUNE_READS_WAVES=$w run "W=$w"; done
