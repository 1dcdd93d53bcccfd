# Build a VARIANT of libsplat_hip.so into splat_slam_amd/lib_<name>/ (git-ignored; travels with gpurun) for same-box A/B
# (scripts/micro/r06_variant_ab.sh, SPLAT_HIP_LIB):   bash scripts/micro/build_variant.sh <name> [-DSGR_X=1 ...]
set -e
name=$1; shift
root=${SGR_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$root/splat_slam_amd/lib_$name
mkdir -p $out
pids=""
for src in $root/splat_slam_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c $src -o $out/$(basename ${src%.hip}).o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libsplat_hip.so $out/*.o
echo $out/libsplat_hip.so
