#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export K4_LIB=$R/4k-nerf_amd/lib4k_hip_timing.so
for cfg in "K4_SR_DEBUG=0" "K4_SR_DEBUG=1" "K4_SR_DEBUG=0 K4_SR_2T_RPW=4"; do echo "== $cfg"; env $cfg python tools/conv_phase_timing.py 2>&1 | grep -v Warn; done
