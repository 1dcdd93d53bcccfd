mkdir -p gpurun_out/r05b
python -m pytest tests -m gpu -x -q > gpurun_out/r05b/gpu_tests3.log 2>&1
tail -4 gpurun_out/r05b/gpu_tests3.log
python scripts/session_timeline.py --out gpurun_out/r05b/session_timeline2.json > gpurun_out/r05b/session_timeline2.log 2>&1; tail -22 gpurun_out/r05b/session_timeline2.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r05b/bench_driver_cmd.json 2> gpurun_out/r05b/bench_driver_cmd.err; tail -c 300 gpurun_out/r05b/bench_driver_cmd.json; tail -3 gpurun_out/r05b/bench_driver_cmd.err
