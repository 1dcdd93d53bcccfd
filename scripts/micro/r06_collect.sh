# round 6: everything profiles/r06_* is made of, in one gpurun call.   bash scripts/micro/r06_collect.sh [tag]
set -x
cd $GRAFT_REPO_ROOT
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python scripts/collect_profiles.py $TAG > $OUT/collect.log 2>&1
tail -3 $OUT/collect.log
python scripts/micro/pmc_tile.py $OUT/pmc_tile_opaque_full.json --scale-add 1.6 > $OUT/pmc_full.log 2>&1
SGR_DEBUG=4096 python scripts/micro/pmc_tile.py $OUT/pmc_tile_opaque_nobwd.json --scale-add 1.6 --passes 0 > $OUT/pmc_nobwd.log 2>&1
SGR_DEBUG=2048 python scripts/micro/pmc_tile.py $OUT/pmc_tile_opaque_nowalk.json --scale-add 1.6 --passes 0 > $OUT/pmc_nowalk.log 2>&1
python scripts/micro/pmc_tile.py $OUT/pmc_tile_light_full.json --passes 0 > $OUT/pmc_light.log 2>&1
SGR_DEBUG=4096 python scripts/micro/pmc_tile.py $OUT/pmc_tile_light_nobwd.json --passes 0 > $OUT/pmc_light_nobwd.log 2>&1
SGR_DEBUG=2048 python scripts/micro/pmc_tile.py $OUT/pmc_tile_light_nowalk.json --passes 0 > $OUT/pmc_light_nowalk.log 2>&1
python scripts/run_session_config1.py --oracle-views 2 --out $OUT/${TAG}_session_configs1.json > $OUT/session.log 2>&1
tail -c 600 $OUT/session.log
python scripts/multi_path_time.py --kernels --out $OUT/${TAG}_multi_path_time_light.json > $OUT/multi.log 2>&1
tail -4 $OUT/multi.log
python bench.py --gaussians 1500000 --no-extras --no-cpu-baseline > $OUT/${TAG}_bench_1500k.json 2> $OUT/b1500.err
python scripts/table_numbers.py > $OUT/table.log 2>&1; cp gpurun_out/table_numbers.json $OUT/${TAG}_table_numbers.json
python scripts/profile_dropin.py --out $OUT/${TAG}_dropin_phases.json > $OUT/dropin.log 2>&1
tail -3 $OUT/dropin.log
ls -la $OUT
