#!/bin/bash
# Is the headline kernel power-limited?  Samples rocm-smi while the kernel runs back to back.
# usage (GPU box): bash tools/power_probe.sh [config]   (c2 | c3 | c4 | mh)
CFG=${1:-c2}
python - "$CFG" <<'PY' &
import sys, time
sys.path.insert(0, '.')
import nthash_amd
cfg = sys.argv[1]
SEEDS = ["1010101010101010101010101010101", "1101101101101101011011011011011"]
L, k, m, seeds, n = {"c2": (150, 31, 1, None, 60_000_000), "c3": (150, 31, 4, None, 20_000_000),
                     "c4": (250, 31, 3, SEEDS, 10_000_000), "mh": (150, 31, 1, None, 60_000_000)}[cfg]
per = m if seeds is None else len(seeds) * m
nwin = L - k + 1
ctx = nthash_amd.Context(0)
d_in = ctx.malloc(n * L); d_out = ctx.malloc(n * nwin * per * 8)
ctx.synth_reads_ptr(d_in, 0, n, L, 42)
sd = nthash_amd.Seeds(ctx, seeds, k) if seeds else None
t_end = time.time() + 12
it = 0
t0 = time.time()
while time.time() < t_end:
    if cfg == "mh": ctx.minhash_ptr(d_in, n, L, 0, k, m, d_out)
    elif sd is None: ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin)
    else: ctx.seed_hash_ptr(d_in, 0, n, L, 0, sd, m, d_out, n * nwin)
    it += 1
dt = time.time() - t0
print(f"{cfg}: {it} launches, {it*n*nwin/dt/1e9:.1f} G k-mers/s sustained over {dt:.1f} s", flush=True)
PY
PID=$!
sleep 5
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|memory)" | tr -s ' ' | head -8
  echo "--"
  sleep 1
done
wait $PID
