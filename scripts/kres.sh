# usage: bash scripts/kres.sh [lib] [pattern] -- registers / scratch / LDS of the kernels of the built code object
L=${1:-blues_amd/csrc/libblues_hip.so}; P=${2:-.}
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$L --output=$T/co 2>/dev/null || \
  python3 - "$L" "$T/co" <<'PY'
import sys
b = open(sys.argv[1], 'rb').read()
i = b.find(b'\x7fELF', b.find(b'__CLANG_OFFLOAD_BUNDLE__'))
open(sys.argv[2], 'wb').write(b[i:])
PY
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/co > $T/notes 2>/dev/null
python3 scripts/kernel_resources.py $T/notes | grep "$P"
rm -rf $T
