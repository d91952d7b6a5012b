set -x
mkdir -p gpurun_out/r4a
python -m pytest tests -m gpu -x -q > gpurun_out/r4a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4a/pytest.log
tail -15 gpurun_out/r4a/pytest.log
for ns in 2 3 4; do echo "NS8=$ns"; PF_ATT_NS8=$ns python tools/bench_ops.py attn; done > gpurun_out/r4a/attn_ns8.txt 2>&1
for ns in 2 3 4; do echo "NS4=$ns"; PF_ATT_NW=4 PF_ATT_NS4=$ns python tools/bench_ops.py attn; done > gpurun_out/r4a/attn_ns4.txt 2>&1
cat gpurun_out/r4a/attn_ns8.txt gpurun_out/r4a/attn_ns4.txt
for ns in 2 4; do PF_ATT_NS8=$ns python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4a/bench_ns$ns.json 2> gpurun_out/r4a/bench_ns$ns.err; done
python -c "
import json
for f in ('bench_ns2','bench_ns4'):
    d=json.load(open('gpurun_out/r4a/'+f+'.json')); print(f, round(d['ms_per_step'],3), d['ids_vs_fp32_oracle']['ok'], {k:v['ms'] for k,v in d['class_ms_per_step'].items()})
"
bash tools/feed_gap.sh r4a
