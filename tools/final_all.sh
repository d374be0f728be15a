PROF_SQ_ONLY=1 bash tools/profile_gpu.sh r04_late late > gpurun_out/prof_r04_late.log 2>&1
PROF_SQ_ONLY=1 bash tools/profile_gpu.sh r04_c2 c2 > gpurun_out/prof_r04_c2.log 2>&1
bash tools/final_bench.sh
