#!/bin/bash
# usage: isa_stats.sh "<extra hipcc flags>" [NT]  -> ISA metrics of qp_kernel<NT> (dev aid)
set -e
NT=${2:-20}
D=$(mktemp -d); cd $D
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I/root/repo/include $1 -c /root/repo/mpc_for_av_at_intersection_amd/csrc/mpcx_qp.hip -o qp.o -save-temps -Rpass-analysis=kernel-resource-usage 2> log.txt || { grep error log.txt; exit 1; }
S=mpcx_qp-hip-amdgcn-amd-amdhsa-gfx950.s
B=$(grep -n "^_ZN4mpcx9qp_kernelILi${NT}EEEvNS_6QpArgsE:" $S | cut -d: -f1)
awk -v b=$B 'NR>=b' $S > k.s; END=$(grep -n "s_endpgm" k.s | head -1 | cut -d: -f1); head -$END k.s > kb.s
grep -A6 "qp_kernelILi${NT}E" log.txt | grep -E "VGPRs:|AGPRs|Scratch|Occupancy" | sed 's/.*remark: *//' | tr '\n' ';'
echo
echo "lines $(wc -l < kb.s) waitcnt $(grep -c s_waitcnt kb.s) b128 $(grep -c ds_read_b128 kb.s) b64 $(grep -c 'ds_read_b64' kb.s) scratch_ld $(grep -c scratch_load kb.s) scratch_st $(grep -c scratch_store kb.s) accvgpr $(grep -c accvgpr kb.s) fma $(grep -c 'v_fma_f64\|v_fmac_f64' kb.s) cndmask $(grep -c v_cndmask kb.s) nop $(grep -c s_nop kb.s)"
cp kb.s /tmp/kb_last.s
cd /; rm -rf $D
