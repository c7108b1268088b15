O=gpurun_out/r06_full_tests4; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log
tail -6 $O/tests.log
grep -n "^E  " $O/tests.log | head -20
