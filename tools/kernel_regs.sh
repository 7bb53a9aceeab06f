#!/bin/bash
# VGPR / SGPR / LDS / scratch of every kernel in a HIP source: tools/kernel_regs.sh eppm_amd/csrc/k_patchmatch.hip [extra hipcc flags]
# (compiles the device code to assembly for gfx950 and reads the .amdhsa_ metadata)
src=$1; shift
out=$(mktemp /tmp/kregs.XXXXXX.s)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    --cuda-device-only -S -o $out "$@" $src 2>/dev/null || exit 1
awk '$1==".amdhsa_kernel"{k=$2} $1==".amdhsa_next_free_vgpr"{v=$2} $1==".amdhsa_next_free_sgpr"{s=$2} $1==".amdhsa_group_segment_fixed_size"{l=$2} $1==".amdhsa_private_segment_fixed_size"{p=$2} $1==".amdhsa_accum_offset"{a=$2} $1==".end_amdhsa_kernel"{printf "%s vgpr %s (arch %s) sgpr %s lds %s scratch %s\n", k, v, a, s, l, p}' $out | c++filt | sed 's/eppm:://g; s/(eppm::PmBatch.*)//'
rm -f $out
