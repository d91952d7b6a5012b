# A/B runs of bench.py under environment switches: bash tools/ab_inflight.sh  (results in gpurun_out/ab_*.json)
export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $EXTRA > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err; }
for rep in 1 2; do
EXTRA="--model seaco --in-flight 1" run c1_base_$rep A=1
EXTRA="--model seaco --in-flight 2" run c2_base_$rep A=1
EXTRA="--model seaco --in-flight 2" run c2_inline_$rep PF_TS_STREAM=0
EXTRA="--model seaco --in-flight 3" run c3_inline_$rep PF_TS_STREAM=0
done
