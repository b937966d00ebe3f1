#!/bin/bash
# forward-half position tables (k = 49 ... 64, kmer_runs_gen_kernel FH) against the full tables: parity + bench --config ref
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/fh
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kmer" 2>&1 | tail -3
for v in 1 0 1 0; do
  export NTHIP_TUNE_NO_FH=$v
  python bench.py --config ref --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-peak --no-plain-pass > gpurun_out/fh/ref_$v.json 2> gpurun_out/fh/ref_$v.err
  python - gpurun_out/fh/ref_$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"NO_FH={sys.argv[2]}: {d['value']/1e9:7.1f} G k-mers/s, kernel {r.get('kernel_avg_ms'):.3f} ms, frac {r['frac']:.4f}, verify {d.get('verify',{}).get('ok')}")
PY
done
unset NTHIP_TUNE_NO_FH
SWEEP_SHAPES="${FH_SHAPES:-100,64,3;100,64,1;150,64,1;150,51,1;150,56,2;250,64,4;151,63,1;100,50,3}" python tools/shape_sweep.py 2>&1 | tail -9
NTHIP_TUNE_NO_FH=1 SWEEP_SHAPES="${FH_SHAPES:-100,64,3;100,64,1;150,64,1;150,51,1;150,56,2;250,64,4;151,63,1;100,50,3}" python tools/shape_sweep.py 2>&1 | tail -9
