// standalone check of k_build_lists<float>: the tile list must hold every atom within cutoff+skin of some tile atom
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../blues_amd/csrc/kernels_nb.h"
int main(int argc, char** argv) {
    const double L = argc > 1 ? atof(argv[1]) : 2.1786, rl = argc > 2 ? atof(argv[2]) : 1.0893;
    const int n = (int)(94.3 * L * L * L), jcap = ((n + 63) / 64) * 64;
    std::vector<AtomF> img(n); std::vector<double> pos(3 * n);
    srand(7);
    for (int i = 0; i < n; i++) { for (int k = 0; k < 3; k++) pos[3 * i + k] = L * (rand() / (RAND_MAX + 1.0)); }
    // sort along x so that consecutive indices are spatially close-ish; tile = 64 atoms nearest to a point near the box corner (straddles the boundary)
    std::vector<int> order(n); for (int i = 0; i < n; i++) order[i] = i;
    const double c0[3] = {0.05, L - 0.05, 0.5 * L};
    auto d2c = [&](int i) { double s = 0; for (int k = 0; k < 3; k++) { double d = pos[3 * i + k] - c0[k]; d -= L * std::round(d / L); s += d * d; } return s; };
    std::sort(order.begin(), order.end(), [&](int a, int b) { return d2c(a) < d2c(b); });
    for (int i = 0; i < n; i++) { img[i].x = (uint32_t)(pos[3 * i] / L * 4294967296.0); img[i].y = (uint32_t)(pos[3 * i + 1] / L * 4294967296.0); img[i].z = (uint32_t)(pos[3 * i + 2] / L * 4294967296.0); img[i].q = 0; img[i].hs = 0.1f; img[i].se = 1; img[i].flags = 0; img[i].orig = i; }
    std::vector<int> tile(64); for (int s = 0; s < 64; s++) tile[s] = order[s];
    DevFlags fl; memset(&fl, 0, sizeof fl); fl.req_gen = 1;
    const int share = (((n + LIST_WAVES - 1) / LIST_WAVES) + 63) & ~63;
    AtomF* d_img; int *d_tile, *d_jl, *d_jc, *d_js, *d_bs; DevFlags* d_fl; unsigned short* d_al;
    hipMalloc(&d_img, n * sizeof(AtomF)); hipMalloc(&d_tile, 64 * 4); hipMalloc(&d_jl, jcap * 4); hipMalloc(&d_jc, 8); hipMalloc(&d_js, LIST_WAVES * share * 4); hipMalloc(&d_bs, (jcap / 64) * 4 + 64); hipMalloc(&d_fl, sizeof fl); hipMalloc(&d_al, 64);
    hipMemcpy(d_img, img.data(), n * sizeof(AtomF), hipMemcpyHostToDevice); hipMemcpy(d_tile, tile.data(), 256, hipMemcpyHostToDevice); hipMemcpy(d_fl, &fl, sizeof fl, hipMemcpyHostToDevice);
    ListArgs a; memset(&a, 0, sizeof a);
    a.n = n; a.n_tiles = 1; a.n_itiles = 1; a.jcap = jcap; a.tile_atoms = d_tile; a.jlist = d_jl; a.jstage = d_js; a.jcount = d_jc; a.batch_slot = d_bs; a.flags = d_fl; a.alist = d_al; a.S = 1; a.n_lists = 1; a.hint_count = jcap;
    a.no_sphere = argc > 3 ? atoi(argv[3]) : 0;
    NbConst<float> c; for (int k = 0; k < 3; k++) { c.dscale[k] = L / 4294967296.0; c.scale[k] = (float)c.dscale[k]; } c.rc2 = 1.0f; c.alpha = 2.0f; c.rlist2 = rl * rl;
    hipLaunchKernelGGL(k_build_lists<float>, dim3(1), dim3(LIST_THREADS), 0, 0, a, c, d_img, 1);
    printf("sync: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    int jc[2]; hipMemcpy(jc, d_jc, 8, hipMemcpyDeviceToHost);
    std::vector<int> jl(jcap); hipMemcpy(jl.data(), d_jl, jcap * 4, hipMemcpyDeviceToHost);
    std::vector<char> in(n, 0); for (int k = 0; k < jc[0]; k++) in[jl[k]] = 1;
    int need = 0, missing = 0; double worst = 1e9;
    for (int j = 0; j < n; j++) {
        double best = 1e9;
        for (int s = 0; s < 64; s++) { double d2 = 0; for (int k = 0; k < 3; k++) { double d = pos[3 * j + k] - pos[3 * tile[s] + k]; d -= L * std::round(d / L); d2 += d * d; } best = std::min(best, d2); }
        if (best < rl * rl) { need++; if (!in[j]) { missing++; worst = std::min(worst, std::sqrt(best)); } }
    }
    printf("L %.4f rl %.4f n %d: list %d entries, %d atoms within rl of a tile atom, MISSING %d (closest missing at %.4f nm)\n", L, rl, n, jc[0], need, missing, missing ? worst : 0.0);
    return 0;
}
