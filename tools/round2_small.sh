export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipeline_kernels.py -x -q -k "small or rc" 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_seaco.py tests/test_gpu_sensevoice.py tests/test_gpu_online.py tests/test_gpu_timestamp.py tests/test_gpu_edges.py -x -q 2>&1 | tail -5
python tools/latency.py
PF_SMALL_NOFUSE=1 python tools/latency.py 2>/dev/null | head -2
PF_SMALL_M=0 python tools/latency.py 2>/dev/null | head -2
