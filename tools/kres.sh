#!/bin/bash
# dev (no GPU needed): registers / spills / LDS / occupancy of the kernels of one source file whose names match a pattern.  usage: tools/kres.sh network 'gb_fx_bin|grid_backward'
cd "$(dirname "$0")/../blender-ngp_amd" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../include -c csrc/$1.hip -o /tmp/kres_$1.o -Rpass-analysis=kernel-resource-usage 2>&1 | sed 's/.*remark: *//' | sed 's/ \[-Rpass.*//' | awk '/error/{print} /Function Name/{n=substr($3,1,64)} /VGPRs:/{v=$2} /SGPRs Spill/{sp=$3} /VGPRs Spill/{vs=$3} /Occupancy/{o=$3} /LDS Size/{print n, "VGPRs", v, "occ", o, "sgpr-spill", sp, "vgpr-spill", vs, "lds", $4}' | grep -E "error|${2:-.}"
