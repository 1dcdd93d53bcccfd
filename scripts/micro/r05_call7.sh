mkdir -p gpurun_out/r05b
python -m pytest tests/test_gpu_dropin_native.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_reference_fixture.py tests/test_gpu_aux.py tests/test_gpu_kat.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r05b/gpu_tests2.log 2>&1
tail -5 gpurun_out/r05b/gpu_tests2.log
python scripts/session_timeline.py --out gpurun_out/r05b/session_timeline.json > gpurun_out/r05b/session_timeline.log 2>&1; tail -40 gpurun_out/r05b/session_timeline.log
