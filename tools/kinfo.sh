#!/bin/bash
# per-kernel registers / LDS / spills of one HIP source (device-only compile to assembly):
#   tools/kinfo.sh aas_enhancement_amd/csrc/rnn_gru_fwd.hip [grep pattern]
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S "$1" -o /tmp/_kinfo.s 2>/dev/null
grep -E "^\s+(- )?\.(name|vgpr_count|agpr_count|sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):" /tmp/_kinfo.s | paste - - - - - - - | sed 's/  */ /g' | c++filt | sed 's/(anonymous namespace):://g' | grep -E "${2:-.}" | cut -c1-260
