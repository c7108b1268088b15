O=gpurun_out/r06_assemble5; mkdir -p $O
timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_march_gpu.py -q -x 2>&1 | tail -3
OUT=r06_assemble5/tl bash tools/four_k_timeline.sh 2>&1 | grep -v "^W2026" | cut -c1-260 | tail -13
timeout 1500 python bench.py --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err; echo bench_rc=$?
python tools/bench_summary.py $O/bench_line.json 2>/dev/null | head -3
python -c "
import json; d=json.load(open('$O/bench_line.json')); print(d.get('four_k_horns',{}).get('ms_per_frame'), d.get('four_k_horns',{}).get('rank_share_8gpu')); print(d['joint_train_step']['ms_per_iteration'], d['joint_train_step']['ms_per_iteration_blocks'])"
