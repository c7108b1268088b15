#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export K4_SR_MODE=f16x3
for d in 0 2 4 6 8 16 24 30 31; do echo "== K4_SR_DEBUG=$d"; K4_SR_DEBUG=$d python tools/conv_layer_time.py 3 4 7 2>&1 | grep "cin"; done
