#!/bin/bash
# dev aid: scripts/mkvar.sh <name> <source file in csrc, e.g. mpcx_expand.hip> "<extra hipcc flags>" -> build/libmpcx_<name>.so
# (the other objects are reused from the tree: run make first); load it with MPCX_LIB=build/libmpcx_<name>.so
set -e
cd "$(dirname "$0")/.."
CS=mpc_for_av_at_intersection_amd/csrc
mkdir -p build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Wno-unused-function $3 -c $CS/$2 -o /tmp/var_$1.o
OBJ=$(ls $CS/*.o | grep -v "${2%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libmpcx_$1.so $OBJ /tmp/var_$1.o -L/opt/rocm/lib -lrccl
