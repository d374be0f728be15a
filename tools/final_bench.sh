#!/bin/bash
# GPU box: the round's closing numbers -- bench.py as the driver runs it (3 times) and with its defaults, then the BASELINE stand-ins
for i in 1 2 3; do timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_drv$i.log 2> gpurun_out/bench_drv$i.err; echo "drv$i rc=$?"; done
timeout -k 10 400 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; echo "default rc=$?"
( timeout -k 10 400 python tools/run_configs.py C2 --cycles 100; timeout -k 10 400 python tools/run_configs.py C3 --cycles 60; timeout -k 10 300 python tools/run_configs.py C4 --cycles 4; timeout -k 10 300 python tools/run_configs.py C4 --cycles 2 --arithmetic exact ) 2>&1 | grep "us/step" > gpurun_out/soak_final.log
cat gpurun_out/soak_final.log | cut -c1-200
