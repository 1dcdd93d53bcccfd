mkdir -p gpurun_out/r05
python scripts/collect_profiles.py r05 > gpurun_out/r05/collect.log 2>&1; tail -3 gpurun_out/r05/collect.log
python scripts/run_session_config1.py --oracle-views 2 --out gpurun_out/r05/r05_session_configs1.json > gpurun_out/r05/session.log 2>&1; tail -3 gpurun_out/r05/session.log
python scripts/profile_dropin.py --out gpurun_out/r05/r05_dropin_phases.json > gpurun_out/r05/dropin.log 2>&1; tail -3 gpurun_out/r05/dropin.log
python scripts/multi_path_time.py --kernels --out gpurun_out/r05/r05_multi_path_time_light.json > gpurun_out/r05/multi.log 2>&1; tail -4 gpurun_out/r05/multi.log
python scripts/session_timeline.py --out gpurun_out/r05/r05_session_timeline.json > gpurun_out/r05/timeline.log 2>&1; tail -3 gpurun_out/r05/timeline.log
