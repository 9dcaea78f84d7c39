#!/bin/bash
# Register / spill / occupancy summary of one kernel file (device-only compile with hipcc's resource-usage remarks).
#   tools/kernel_regs.sh lstm_seq.hip [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt --cuda-device-only -c "$(dirname "$0")/../speechsplit_amd/csrc/$f" -o /dev/null \
    -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
    grep -E "Function Name|  VGPRs:|Spill|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//;s/ \[-Rpass.*//' | paste - - - - - - | sed 's/Function Name: //' | c++filt | cut -c1-330
