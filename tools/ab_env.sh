# usage: tools/ab_env.sh VAR v1 v2 ... -> two bench runs per value (GPU box)
VAR=$1; shift
for i in 1 2; do for v in "$@"; do
env $VAR=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 600 > gpurun_out/abq.log 2>&1
echo "$VAR=$v $(grep -o '"ms_per_step": [0-9.]*\|"full_mcmc_step_ms": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"back_to_back_replay_ms": [0-9.]*' gpurun_out/abq.log | tr '\n' ' ')"
done; done
