PIPELINED=0 FRAME_SCRIPT=tools/rank_share_timeline.py OUT=r06_share_tl/serial bash tools/four_k_timeline.sh 2>&1 | grep -v "^W2026" | cut -c1-330 | tail -14
PIPELINED=1 FRAME_SCRIPT=tools/rank_share_timeline.py OUT=r06_share_tl/pipe bash tools/four_k_timeline.sh 2>&1 | grep -v "^W2026" | cut -c1-330 | tail -14
