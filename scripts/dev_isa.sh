#!/bin/bash
# ISA of ONE kernel instantiation without building the library (device-only, ~10 s instead of ~2 min):
#   scripts/dev_isa.sh out.s 'template __global__ void k_norm_colsum2<VC2_BF16, 7, 0>(const void*, int, const int*, int, int, int, int64_t, float*, double*, int*, unsigned long long*, int, uint8_t*, OrderArgs);' [extra hipcc flags]
# (-DVC2_DEV_ONLY=<explicit instantiation> drops the C ABI, i.e. every other instantiation)
out=$1; inst=$2; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S \
  "-DVC2_DEV_ONLY=$inst" "$@" "$(dirname "$0")/../vidcom2_amd/csrc/vc2_kernels.hip" -o "$out" 2>&1 | grep -v "warning: unused\|^ *[0-9]* | \|^ *| \|warnings generated" | head -40
grep -E "^\s+\.(sgpr|vgpr)_count|vgpr_spill|\.lds_size|; (NumVgprs|NumSgprs|Occupancy|ScratchSize|LDSByteSize)" "$out" | head -20
