#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "golden_synth or fixed_vs_oracle or properties" 2>&1 | tail -2
NTHIP_TUNE_L2PF_TILES=8 timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "golden_synth or fixed_vs_oracle or properties" 2>&1 | tail -2
AB_PROBED=1 ABLATE_SHAPE=150,31,1 python tools/ab_multi.py ":NTHIP_TUNE_L2PF_TILES=4,:NTHIP_TUNE_L2PF_TILES=8,:NTHIP_TUNE_L2PF_TILES=16,:NTHIP_TUNE_L2PF_TILES=32,nohash,nohash:NTHIP_TUNE_L2PF_TILES=8,nohash:NTHIP_TUNE_L2PF_TILES=16" 100000000 8 | cut -c1-125
