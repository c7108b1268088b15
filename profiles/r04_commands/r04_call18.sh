cd /root/repo
mkdir -p gpurun_out/r4c18
for ns in 0 900 1800 2700 0 1800; do echo "K4_P16_STAGGER_NS=$ns"; K4_P16_STAGGER_NS=$ns timeout 300 python tools/sr_frame_time.py f16x3p f16x3p 2>&1 | grep ms/frame; done | tee gpurun_out/r4c18/stagger.log
