#!/bin/bash
# builds libmpcx variants with ablation macros into gpurun-visible files (dev aid); run scripts/ablate_run.py on the box
set -e
cd /root/repo/mpc_for_av_at_intersection_amd/csrc
mkdir -p ../../build_ablate
i=0
while IFS= read -r cfg; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I/root/repo/include $cfg -c mpcx_qp.hip -o /tmp/qp_var.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_ablate/libmpcx_$i.bin mpcx_api.o mpcx_expand.o mpcx_interaction.o mpcx_misc.o mpcx_prepare.o /tmp/qp_var.o
  echo "$i: $cfg" 
  i=$((i+1))
done
