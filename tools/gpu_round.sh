#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python tools/size_probe.py 2>&1 | tail -20
