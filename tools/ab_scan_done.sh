for i in 1 2; do
for m in flags count; do
GRAAL_SCAN_DONE=$m timeout -k 10 200 python bench.py --no-cpu-baseline --steps 600 > gpurun_out/ab_$m.log 2>&1
echo "$m $(grep -o '"ms_per_step": [0-9.]*\|"full_mcmc_step_ms": [0-9.]*\|"avg_launch_ms": [0-9.]*' gpurun_out/ab_$m.log | tr '\n' ' ')"
done; done
GRAAL_SCAN_DONE=count timeout -k 10 600 python -m pytest tests/test_engine_gpu.py tests/test_sampler_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -2
