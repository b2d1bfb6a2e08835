#!/bin/bash
# register / scratch / LDS use of the kernels whose name matches $1 (cross-compiles the library's translation unit; no GPU needed)
#   scripts/kernel_regs.sh frag
set -e
d=$(mktemp -d); cd "$d"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-value --save-temps -c /root/repo/blues_amd/csrc/blues_engine.hip -o /dev/null 2>/dev/null
python3 - "$1" <<'P'
import re, sys
s = open('blues_engine-hip-amdgcn-amd-amdhsa-gfx950.s').read()
md = s[s.index('amdhsa.kernels'):]
for e in md.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', e).group(1)
    if sys.argv[1] in name:
        g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, e).group(1)
        print("%-72s vgpr %4s sgpr %4s scratch %5s lds %6s" % (name[:72], g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
P
cp blues_engine-hip-amdgcn-amd-amdhsa-gfx950.s /tmp/last_engine.s
rm -rf "$d"
