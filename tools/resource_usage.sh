#!/bin/bash
# Per-kernel VGPRs / scratch / occupancy / static LDS of every kernel file (no GPU needed).  usage: bash tools/resource_usage.sh > profiles/rNN_kernel_resource_usage.txt
cd "$(dirname "$0")/.."
echo "# hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage (static LDS only: dynamic LDS is set at launch)"
echo "file | kernel | VGPRs | scratch B/lane | waves/SIMD | static LDS B"
for f in bfv bmul eval fused gsplit isplit ntt; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-pass-failed -I include -c --cuda-device-only \
    -Rpass-analysis=kernel-resource-usage -o /dev/null abc_amd/csrc/abc_kernels_$f.hip 2>&1 | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
    awk -v F=$f '/Function Name:/ {n=$NF} / VGPRs:/ {v=$NF} /ScratchSize/ {s=$NF} /Occupancy/ {o=$NF} /LDS Size/ {print F" | "n" | "v" | "s" | "o" | "$NF}' |
    while IFS= read -r line; do
      sym=$(echo "$line" | cut -d'|' -f2 | tr -d ' ')
      dem=$(echo "$sym" | c++filt | sed 's/(.*//; s/^void //')
      echo "$line" | awk -F'|' -v D="$dem" '{print $1"| "D" |"$3"|"$4"|"$5"|"$6}'
    done
done
