set -x
O=gpurun_out/r4e; mkdir -p $O
python -m pytest tests/test_gpu_group.py tests/test_gpu_seaco.py tests/test_gpu_edges.py tests/test_gpu_int8.py tests/test_gpu_baseline_sizes.py -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
bash tools/profile_round.sh round4 > $O/profile_round.log 2>&1
tail -30 $O/profile_round.log
