export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipeline_kernels.py -x -q -k "small" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_seaco.py tests/test_gpu_sensevoice.py tests/test_gpu_online.py tests/test_gpu_timestamp.py -x -q 2>&1 | tail -30
python tools/latency.py
