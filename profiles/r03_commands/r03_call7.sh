#!/bin/bash
# Round 3, GPU call 7: is the fp16 3x3 convolution memory-bound?  single layers with and without memory traffic (K4_SR_DEBUG=1), tile heights, FETCH / WRITE.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3g
mkdir -p $O
cd $R
export K4_SR_MODE=f16x3
for cfg in "K4_SR_DEBUG=0" "K4_SR_DEBUG=1" "K4_SR_DEBUG=0 K4_SR_2T_RPW=4" "K4_SR_DEBUG=1 K4_SR_2T_RPW=4" "K4_SR_DEBUG=0 K4_SR_NBK=2" "K4_SR_DEBUG=1 K4_SR_NBK=2" "K4_SR_DEBUG=0 K4_SR_WLDS=1" "K4_SR_DEBUG=1 K4_SR_WLDS=1"; do
  echo "== $cfg"; env $cfg python tools/conv_layer_time.py 0 3 4 7 2>&1 | grep "cin"
done
cd /tmp && export TMPDIR=/tmp
for g in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
  out=$R/gpurun_out/pmc_cl; rm -rf $out
  timeout 200 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -o run -- python $R/tools/conv_layer_time.py 3 4 7 > $out.log 2>&1 || { echo "pmc $g failed"; tail -3 $out.log; }
  python $R/tools/pmc_by_grid.py $out | grep -A1 "b6v2" 
done
rm -rf $R/gpurun_out/pmc_cl $R/gpurun_out/pmc_cl.log
