mkdir -p gpurun_out/r05c
python -m pytest tests/test_gpu_round5.py -x -q > gpurun_out/r05c/t_round5.log 2>&1; tail -4 gpurun_out/r05c/t_round5.log
python bench.py > gpurun_out/r05c/bench_default.json 2> gpurun_out/r05c/bench_default.err; tail -c 400 gpurun_out/r05c/bench_default.json; tail -3 gpurun_out/r05c/bench_default.err
