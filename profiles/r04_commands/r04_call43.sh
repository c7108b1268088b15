# texture-addresser (TA) / L1 activity of the marcher kernels: is the shading kernel bound by the rate at which its 16-byte corner gathers are turned into cache-line lookups?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c43
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for g in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  out=$R/gpurun_out/pmc_r4t_$i; rm -rf $out
  timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -o run -- python $R/bench.py --steps 3 --warmup 1 --sr-frames 0 --no-cpu-baseline --no-extras --streams 1 > $out.log 2>&1 || echo "group $i failed"
  i=$((i+1))
done
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_r4t_* > $O/marcher_ta_pmc.md 2>&1
rm -rf $R/gpurun_out/pmc_r4t_[0-9]
grep -A 14 "k4_shade_kernel\|k4_geom3_kernel" $O/marcher_ta_pmc.md | head -60
