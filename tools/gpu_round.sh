cd $GRAFT_REPO_ROOT
python tools/fill_probe.py fs8,fs32,fs64,fw4,fw2,fw4s16 24
