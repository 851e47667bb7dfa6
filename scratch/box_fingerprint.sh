#!/bin/bash
# what distinguishes the boxes of the pool on which S-from-memory slows an fp32 model down?  (partition modes, firmware, driver, memory vendor) + a 2-arm A/B
echo "== kernel / driver"; uname -r; cat /sys/module/amdgpu/version 2>/dev/null
for c in /sys/class/drm/card*/device; do
  [ -f $c/current_compute_partition ] || continue
  echo "== $c"; for f in unique_id current_compute_partition current_memory_partition available_memory_partition mem_info_vram_vendor vbios_version pcie_bw; do [ -r $c/$f ] && echo "$f: $(cat $c/$f 2>/dev/null | head -1)"; done
  break
done
rocm-smi --showmemvendor --showvbios --showfwinfo 2>/dev/null | grep -v "^=\|^$" | grep "GPU\[0\]" | head -40
rocminfo 2>/dev/null | grep -i "Marketing Name\|Compute Unit\|Max Clock\|L2\|L3\|Cacheline" | head -12
