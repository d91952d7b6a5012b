# A/B runs of bench.py under environment switches: bash tools/ab_inflight.sh  (results in gpurun_out/ab_*.json)
export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $EXTRA > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err; }
for rep in 1 2; do
EXTRA="--in-flight 1" run e1_base_$rep A=1
EXTRA="--in-flight 1" run e1_pfd4_$rep PF_RC_PFD=4
EXTRA="--in-flight 1" run e1_pfd8_$rep PF_RC_PFD=8
EXTRA="" run e2_base_$rep A=1
EXTRA="" run e2_pfd4_$rep PF_RC_PFD=4
EXTRA="" run e2_pfd8_$rep PF_RC_PFD=8
done
