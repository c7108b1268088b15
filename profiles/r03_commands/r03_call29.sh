#!/bin/bash
# second part of r03_call28.sh: WRITE_SIZE and GRBM_GUI_ACTIVE (the clock base of the busy fractions), up to three tries
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3f2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/pmc_sr_9
for try in 1 2 3; do
  rm -rf $out
  if timeout 45 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o run -- python $R/tools/sr_frame_time.py f16x3 > $out.log 2>&1; then echo "ok (try $try)"; break; else echo "failed (try $try)"; rm -rf $out; fi
done
python $R/tools/pmc_by_grid.py $out > $O/sr_pmc_by_grid_write.md
rm -rf $out $out.log
wc -l $O/sr_pmc_by_grid_write.md
