cd /root/repo
mkdir -p gpurun_out/r4c41
for d in 0 2048 0 2048 2304; do echo "K4_SR_DEBUG=$d (2048: fragment reads of a chunk only for its first sub-stages -- WRONG results, same MFMAs and DMA; 2304 = also no MFMAs)"; K4_SR_DEBUG=$d K4_TOOL_ONLY=p16 K4_TOOL_WINDOWS=4 timeout 300 python tools/p16_layer_time.py 0 3 4 2>&1 | grep "^cin" | sed 's/f16x3 per-tile.*| p16 in, p16 out/| p16 in, p16 out/'; done | tee gpurun_out/r4c41/lds_ablation.log
