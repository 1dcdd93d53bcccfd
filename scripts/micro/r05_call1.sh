set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
python scripts/micro/pmc_tile.py gpurun_out/r05a/pmc_tile_opaque_full.json --scale-add 1.6 > gpurun_out/r05a/full.log 2>&1
SGR_DEBUG=4096 python scripts/micro/pmc_tile.py gpurun_out/r05a/pmc_tile_opaque_nobwd.json --scale-add 1.6 --passes 0 > gpurun_out/r05a/nobwd.log 2>&1
SGR_DEBUG=2048 python scripts/micro/pmc_tile.py gpurun_out/r05a/pmc_tile_opaque_nowalk.json --scale-add 1.6 --passes 0 > gpurun_out/r05a/nowalk.log 2>&1
tail -3 gpurun_out/r05a/*.log
