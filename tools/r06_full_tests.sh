O=gpurun_out/r06_full_tests3; mkdir -p $O
timeout 2400 python -m pytest tests/test_train_ops_gpu.py tests/test_sr_train_gpu.py tests/test_optim_gpu.py -q -m gpu > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log
tail -5 $O/tests.log
grep -n "^E  " $O/tests.log | head -20
