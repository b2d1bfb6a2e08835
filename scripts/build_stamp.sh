# the -DBLUES_STAMP build of the library (phase stamps of the kernels; loaded through BLUES_LIB_PATH): bash scripts/build_stamp.sh
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -fno-slp-vectorize -DBLUES_STAMP -o blues_amd/csrc/libblues_hip_stamp.so blues_amd/csrc/blues_engine.hip
