ALL=1 TS=510 python tools/rank_share_streams_probe.py 2>&1 | grep -v amdgpu.ids | tail -6
ALL=1 TS=510 python tools/rank_share_streams_probe.py 2>&1 | grep -v amdgpu.ids | tail -5
