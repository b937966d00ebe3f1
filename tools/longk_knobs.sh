cd $GRAFT_REPO_ROOT
S="150,65,1;150,100,1;250,200,1;1000,500,1;300,128,1;400,255,1;100,64,3;100,64,1;150,64,1;150,80,2;10000,200,1"
for spec in "default" "NTHIP_TUNE_FW=1" "NTHIP_TUNE_FW=2" "NTHIP_TUNE_WAVES=8" "NTHIP_TUNE_WAVES=16" "NTHIP_TUNE_RUN_MAX=23" "NTHIP_TUNE_RUN_MAX=13"; do
  echo "== $spec"
  if [ "$spec" = default ]; then e=""; else e="$spec"; fi
  env $e SWEEP_PROBED=0 SWEEP_SHAPES="$S" python tools/shape_sweep.py 2>&1 | awk '{printf "%s %s %s %s %s %s | ", $2,$3,$4,$10,$11,$15} END{print ""}'
done
