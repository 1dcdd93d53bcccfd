mkdir -p gpurun_out/r05e
python -m pytest tests -m gpu -q > gpurun_out/r05e/gpu_full.log 2>&1; tail -3 gpurun_out/r05e/gpu_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05e/bench_driver_cmd.json 2> gpurun_out/r05e/bench_driver_cmd.err; tail -c 200 gpurun_out/r05e/bench_driver_cmd.json
python bench.py --gaussians 1500000 --no-extras --no-cpu-baseline > gpurun_out/r05e/r05_bench_1500k.json 2>/dev/null; tail -c 200 gpurun_out/r05e/r05_bench_1500k.json
python scripts/table_numbers.py > gpurun_out/r05e/table.log 2>&1; cp gpurun_out/table_numbers.json gpurun_out/r05e/r05_table_numbers.json 2>/dev/null; tail -2 gpurun_out/r05e/table.log
