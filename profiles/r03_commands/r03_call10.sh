#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export K4_SR_MODE=f16x3
for d in 0 4 8 12 16 24 32 48; do echo "== K4_SR_STAGGER=$d"; K4_SR_STAGGER=$d python tools/conv_layer_time.py 3 4 7 2>&1 | grep "cin"; done
