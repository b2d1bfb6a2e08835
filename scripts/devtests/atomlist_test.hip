// standalone check of k_build_atom_lists<float> on synthetic data
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include "../../blues_amd/csrc/kernels_nb.h"
int main() {
    const int n = 4000, jcap = 2048, acap = 960; const double L = 4.0;
    std::vector<AtomF> img(n); std::vector<double> pos(3 * n);
    srand(1);
    for (int i = 0; i < n; i++) { for (int k = 0; k < 3; k++) pos[3 * i + k] = L * (rand() / (RAND_MAX + 1.0)); img[i].x = (uint32_t)(pos[3 * i] / L * 4294967296.0); img[i].y = (uint32_t)(pos[3 * i + 1] / L * 4294967296.0); img[i].z = (uint32_t)(pos[3 * i + 2] / L * 4294967296.0); img[i].q = 0; img[i].hs = 0.1f; img[i].se = 1; img[i].flags = (i % 7 == 0) ? FLAG_MOBILE : 0; img[i].orig = i; }
    std::vector<int> tile(64, -1); for (int s = 0; s < 40; s++) tile[s] = 100 + 3 * s;
    std::vector<int> jl(jcap, 0); int count = 0; for (int j = 0; j < n && count < 1500; j += 2) jl[count++] = j;   // ascending candidates
    std::vector<int> jcount(2, 0); jcount[0] = count;
    std::vector<int> ex_start(n + 1), ex_idx; for (int i = 0; i < n; i++) { ex_start[i] = (int)ex_idx.size(); ex_idx.push_back(i); if (i + 2 < n && i % 2 == 0) ex_idx.push_back(i + 2); } ex_start[n] = (int)ex_idx.size();
    DevFlags fl; memset(&fl, 0, sizeof fl); fl.req_gen = 1;
    AtomF* d_img; int *d_tile, *d_jl, *d_jc, *d_es, *d_ei, *d_ac; unsigned short* d_al; DevFlags* d_fl;
    hipMalloc(&d_img, n * sizeof(AtomF)); hipMalloc(&d_tile, 64 * 4); hipMalloc(&d_jl, jcap * 4); hipMalloc(&d_jc, 8); hipMalloc(&d_es, (n + 1) * 4); hipMalloc(&d_ei, ex_idx.size() * 4); hipMalloc(&d_ac, 64 * 4); hipMalloc(&d_al, 64 * acap * 2); hipMalloc(&d_fl, sizeof fl);
    hipMemcpy(d_img, img.data(), n * sizeof(AtomF), hipMemcpyHostToDevice); hipMemcpy(d_tile, tile.data(), 256, hipMemcpyHostToDevice); hipMemcpy(d_jl, jl.data(), jcap * 4, hipMemcpyHostToDevice); hipMemcpy(d_jc, jcount.data(), 8, hipMemcpyHostToDevice);
    hipMemcpy(d_es, ex_start.data(), (n + 1) * 4, hipMemcpyHostToDevice); hipMemcpy(d_ei, ex_idx.data(), ex_idx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_fl, &fl, sizeof fl, hipMemcpyHostToDevice); hipMemset(d_ac, 0xff, 256);
    ListArgs a; memset(&a, 0, sizeof a);
    a.n = n; a.n_tiles = 1; a.n_itiles = 1; a.jcap = jcap; a.tile_atoms = d_tile; a.jlist = d_jl; a.jcount = d_jc; a.ex_start = d_es; a.ex_idx = d_ei; a.flags = d_fl; a.alist = d_al; a.acount = d_ac; a.acap = acap; a.S = 1; a.n_lists = 1; a.hint_count = jcap;
    NbConst<float> c; for (int k = 0; k < 3; k++) { c.dscale[k] = L / 4294967296.0; c.scale[k] = (float)c.dscale[k]; } c.rc2 = 1.0f; c.alpha = 2.0f; c.rlist2 = 1.12 * 1.12;
    const size_t lds = (size_t)jcap * 24;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_build_atom_lists<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    printf("setattr: %s\n", hipGetErrorString(e));
    hipLaunchKernelGGL(k_build_atom_lists<float>, dim3(1), dim3(LIST_THREADS), lds, 0, a, c, d_img, 1);
    printf("launch: %s\n", hipGetErrorString(hipGetLastError()));
    printf("sync: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    std::vector<int> ac(64); hipMemcpy(ac.data(), d_ac, 256, hipMemcpyDeviceToHost);
    std::vector<unsigned short> al(64 * acap); hipMemcpy(al.data(), d_al, al.size() * 2, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int s = 0; s < 64; s++) {
        int ref = 0; std::vector<int> want;
        if (tile[s] >= 0) for (int k = 0; k < count; k++) {
            const int j = jl[k]; double d2 = 0; for (int q = 0; q < 3; q++) { double d = pos[3 * j + q] - pos[3 * tile[s] + q]; d -= L * std::round(d / L); d2 += d * d; }
            const bool ex = (j == tile[s]) || (tile[s] % 2 == 0 && j == tile[s] + 2) || (j % 2 == 0 && tile[s] == j + 2);
            if (d2 < 1.12 * 1.12 && !ex) { want.push_back(k); ref++; }
        }
        bool ok = ac[s] == ref;
        for (int q = 0; ok && q < ref; q++) ok = (al[s * acap + q] & 0x7fff) == want[q] && ((al[s * acap + q] >> 15) == ((jl[want[q]] % 7 == 0) ? 1 : 0));
        if (!ok) { bad++; if (bad < 6) printf("slot %d: got %d want %d first %d/%d\n", s, ac[s], ref, ref ? (al[s * acap] & 0x7fff) : -1, ref ? want[0] : -1); }
    }
    printf("slots wrong: %d of 64 (counts e.g. %d %d %d)\n", bad, ac[0], ac[1], ac[63]);
    return 0;
}
