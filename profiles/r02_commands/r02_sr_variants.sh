#!/bin/bash
# compile-time variants of the v2 convolution kernel, rebuilt and timed on the GPU box (arguments: hipcc -D flag sets)
for v in "$@"; do
  touch 4k-nerf_amd/csrc/k4_sr.hip
  K4_EXTRA_HIPCC_FLAGS="$v" python 4k-nerf_amd/build.py > /dev/null 2>&1 || echo "build failed: $v"
  echo "== $v"
  python tools/conv_layer_time.py ${CASES:-3 7} 2>&1 | grep cin
done
touch 4k-nerf_amd/csrc/k4_sr.hip
