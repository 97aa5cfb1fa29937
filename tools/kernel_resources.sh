#!/bin/bash
# Prints VGPR / scratch / LDS use of every kernel in kernels.hip (compile only, no GPU needed).
# Scratch must stay 0 for the model kernels: a scratch reload costs s_waitcnt vmcnt(0).
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -ffp-contract=on -I"$R/include" -I"$R/gtcrn_micro_amd/csrc" \
    -Rpass-analysis=kernel-resource-usage -c "$R/gtcrn_micro_amd/csrc/kernels.hip" -o /dev/null "$@" 2>&1 |
  grep -E "Function Name|VGPRs:|Spill|ScratchSize|LDS Size|Occupancy" | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/.*remark: //' |
  awk '/Function Name/{if (line) print line; sub(/.*Function Name: /,""); line=$0; next} {gsub(/^ +/,""); line=line" | "$0} END{print line}' | c++filt | sed 's/(.*)//' 
