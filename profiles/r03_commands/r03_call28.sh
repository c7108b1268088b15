#!/bin/bash
# Decoder counters at the final kernels, per kernel and grid size (one 4K frame = layers of four windows per launch): SQ busy / VALU /
# MFMA, FETCH_SIZE, WRITE_SIZE.  Separate --pmc passes, --kernel-trace only; a pass is retried once (rocprofv3 --pmc segfaulted in
# the process's first torch kernel on some boxes).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3f2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
groups=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for g in "${groups[@]}"; do
  out=$R/gpurun_out/pmc_sr_$i
  for try in 1 2; do
    rm -rf $out
    if timeout 60 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -o run -- python $R/tools/sr_frame_time.py f16x3 > $out.log 2>&1; then echo "group $i ok (try $try)"; break; else echo "group $i failed (try $try)"; rm -rf $out; fi
  done
  i=$((i+1))
done
python $R/tools/pmc_by_grid.py $R/gpurun_out/pmc_sr_* > $O/sr_pmc_by_grid.md
rm -rf $R/gpurun_out/pmc_sr_[0-9] $R/gpurun_out/pmc_sr_*.log
wc -l $O/sr_pmc_by_grid.md
