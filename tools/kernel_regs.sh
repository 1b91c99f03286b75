#!/bin/bash
# Registers, spills, scratch and static LDS of every kernel of one source file, from the metadata of the gfx950
# assembly, which is left in /tmp/kregs/<file>-hip-amdgcn-amd-amdhsa-gfx950.s.   usage: tools/kernel_regs.sh spgemm.hip [extra flags]
f=${1:-spgemm.hip}; shift
mkdir -p /tmp/kregs
cd "$(dirname "$0")/../spmm_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Xclang -target-feature -Xclang -packed-fp32-ops "$@" \
  -c $f -o /tmp/kregs/${f%.hip}.o -save-temps=obj 2>/dev/null
python3 - /tmp/kregs/${f%.hip}-hip-amdgcn-amd-amdhsa-gfx950.s <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
meta = txt[txt.index('amdhsa.kernels:'):]
for blk in re.split(r'\n  - \.agpr_count', meta)[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s*(\S+)', blk) or [None, '?'])[1]
    name = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip()
    name = name.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
    print(f"{name:62s} vgpr {g('vgpr_count'):>4} spill {g('vgpr_spill_count'):>3} sgpr_spill {g('sgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size'):>6}")
PY
