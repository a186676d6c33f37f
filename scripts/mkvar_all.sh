#!/bin/bash
# dev aid: scripts/mkvar_all.sh <name> "<extra hipcc flags>" -> build/libmpcx_<name>.so with EVERY source compiled with the flags
set -e
cd "$(dirname "$0")/.."
CS=mpc_for_av_at_intersection_amd/csrc
mkdir -p build /tmp/var_$1
for f in $CS/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Wno-unused-function $2 -c $f -o /tmp/var_$1/$(basename ${f%.hip}).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libmpcx_$1.so /tmp/var_$1/*.o -L/opt/rocm/lib -lrccl
