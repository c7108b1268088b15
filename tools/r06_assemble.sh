O=gpurun_out/r06_assemble2; mkdir -p $O
timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_sr_gpu.py tests/test_sr_train_gpu.py tests/test_train_ops_gpu.py -q -x 2>&1 | tail -3
python tools/sr_frame_time.py f16x3p 2>/dev/null | tail -1
OUT=r06_assemble2/tl bash tools/four_k_timeline.sh 2>&1 | grep -v "^W2026" | cut -c1-260 | tail -14
for i in 1 2; do timeout 600 python tools/joint_phase_events.py 2>/dev/null; done
timeout 1500 python bench.py --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err; echo bench_rc=$?
python tools/bench_summary.py $O/bench_line.json 2>/dev/null | head -3
python -c "
import json; d=json.load(open('$O/bench_line.json')); print(d.get('four_k_horns',{}).get('ms_per_frame'), d.get('four_k_horns',{}).get('rank_share_8gpu')); print(d['joint_train_step']['ms_per_iteration'], d['joint_train_step']['ms_per_iteration_blocks'])"
