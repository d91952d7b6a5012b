# A/B runs of bench.py under environment switches: bash tools/ab_inflight.sh  (results in gpurun_out/ab_*.json)
export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $EXTRA > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err; }
for rep in 1 2; do
EXTRA="--in-flight 1" run e1_$rep A=1
EXTRA="" run e2_$rep A=1
EXTRA="--model sensevoice" run sv_$rep A=1
done
