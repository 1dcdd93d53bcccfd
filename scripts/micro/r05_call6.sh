mkdir -p gpurun_out/r05b
python -m pytest tests -m gpu -x -q > gpurun_out/r05b/gpu_tests.log 2>&1
tail -5 gpurun_out/r05b/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05b/smoke.log 2>&1; tail -2 gpurun_out/r05b/smoke.log
python bench.py > gpurun_out/r05b/bench_default.json 2> gpurun_out/r05b/bench_default.err; tail -c 600 gpurun_out/r05b/bench_default.json
