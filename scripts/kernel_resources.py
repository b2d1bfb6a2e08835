"""Prints registers / scratch / LDS of every kernel in a gfx950 code object (llvm-readelf --notes output)."""
import re, subprocess, sys
t = open(sys.argv[1]).read()
for k in re.split(r'\n\s+- \.agpr_count', t)[1:]:
    g = lambda key: re.search(r'\.%s:\s+(\S+)' % key, k).group(1)
    d = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip()
    d = re.sub(r'^void ', '', d)
    print("%-64s scratch=%-5s vgpr=%-4s sgpr=%-4s lds=%s" % (d[:64], g('private_segment_fixed_size'), g('vgpr_count'), g('sgpr_count'), g('group_segment_fixed_size')))
