// kernels_pme.h -- smooth particle-mesh Ewald reciprocal space, the part of OpenMM's
// NonbondedForce under nonbondedMethod=PME (reference blues/simulation.py:219, examples/rotmove_cuda.yml:20) that the
// direct-space kernels leave out.  Same algorithm and conventions as the oracle (oracle/blues_oracle.c, reciprocal_space):
// order-`order` cardinal B-splines on the mesh points (floor(u) + k) mod K, E = 1/2 sum_m eterm(m) |Q^(m)|^2,
// forces from the spline derivatives [OpenMM 7.4.2 ReferencePME.cpp, recalled].
//
// MI355X shape.  The meshes of this path are small (18 x 27 x 36 for the 23k-atom box at ewaldErrorTolerance 0.005), and
// the sum is LINEAR in the charges: what the frozen atoms contribute -- their transformed charge mesh and their potential
// mesh -- is computed once per geometry (STATIC launch) and only the few hundred mobile charges are spread, transformed and
// gathered per step.  One workgroup per replica runs the whole pipeline (spread -> three axis transforms -> eterm ->
// three axis transforms back -> gather) with workgroup barriers between the phases: no launch boundaries, the mesh stays
// in the L2 of the replica's XCD.  Spreading accumulates in 64-bit fixed point (integer atomics commute: the result does
// not depend on the order in which the atoms arrive, so a batched replica stays bitwise identical to a lone one).
// The axis transforms are plain DFTs (K <= 64 per axis, any factorisation -- OpenMM's Reference platform takes the mesh
// sizes as ceil(...) gives them, 27 included).
#pragma once
#include "device_common.h"

#define PME_THREADS 1024
#define PME_MAX_ORDER 8
#define PME_FIX 1099511627776.0   // 2^40: fixed-point scale of the charge mesh

template <typename T> struct PmeArgs {
    int n, K[3], order, ng;
    int n_sel; const int* sel;         // atoms this launch spreads (and gathers): caller indices
    const double* x[3];                // master positions
    const double* qn;                  // [n] NonbondedForce charges (alchemical atoms: 0)
    Box3 box;
    unsigned long long* acc;           // [ng] fixed-point charge mesh
    T* a_re; T* a_im; T* b_re; T* b_im;   // work meshes
    const T* eterm;                    // [ng], 0 at m = 0
    const T* tw_cos[3]; const T* tw_sin[3];   // cos / sin(2 pi t / K) per axis
    T* qf_re; T* qf_im; T* phi_f;      // static part: transform of the frozen charges' mesh, their potential mesh
    int have_static;                   // the static meshes hold something (there are frozen charges)
    double* frec;                      // [3][n] force on the selected atoms
    double* epart;                     // [1] mesh energy 1/2 sum eterm |Q^_f + Q^_m|^2 (dynamic launch, when want_energy)
    int want_energy;
};

template <typename T>
__device__ __forceinline__ void pme_splines(T dr, int order, T* w, T* dw) {
    w[order - 1] = (T)0; w[1] = dr; w[0] = (T)1 - dr;
    for (int l = 3; l < order; l++) {
        const T div = (T)1 / (T)(l - 1);
        w[l - 1] = div * dr * w[l - 2];
        for (int k = 1; k < l - 1; k++) w[l - k - 1] = div * ((dr + (T)k) * w[l - k - 2] + ((T)(l - k) - dr) * w[l - k - 1]);
        w[0] = div * ((T)1 - dr) * w[0];
    }
    dw[0] = -w[0];
    for (int k = 1; k < order; k++) dw[k] = w[k - 1] - w[k];
    const T div = (T)1 / (T)(order - 1);
    w[order - 1] = div * dr * w[order - 2];
    for (int k = 1; k < order - 1; k++) w[order - k - 1] = div * ((dr + (T)k) * w[order - k - 2] + ((T)(order - k) - dr) * w[order - k - 1]);
    w[0] = div * ((T)1 - dr) * w[0];
}

// mesh index and spline offset of atom i along axis d (fractional coordinate in fp64: the offset must not lose the position's digits)
template <typename T>
__device__ __forceinline__ void pme_locate(const PmeArgs<T>& P, int i, int d, int& ti, T& dr) {
    double fr = P.x[d][i] * P.box.invL[d]; fr -= floor(fr);
    const double u = fr * P.K[d];
    ti = (int)u; dr = (T)(u - ti);
    if (ti >= P.K[d]) ti -= P.K[d];
}

// one axis of the separable transform: out(m) = sum_k in(k) exp(sign 2 pi i m k / K) along axis ax, thread per output element
template <typename T>
__device__ __forceinline__ void pme_dft_axis(const PmeArgs<T>& P, const T* __restrict__ ire, const T* __restrict__ iim, T* __restrict__ ore, T* __restrict__ oim, int ax, T sign) {
    const int K = P.K[ax];
    const int st = ax == 0 ? P.K[1] * P.K[2] : (ax == 1 ? P.K[2] : 1);
    const T* tc = P.tw_cos[ax]; const T* ts = P.tw_sin[ax];
    for (int g = threadIdx.x; g < P.ng; g += PME_THREADS) {
        const int m = (g / st) % K, base = g - m * st;
        T sr = (T)0, si = (T)0;
        int t = 0;
        for (int k = 0; k < K; k++) {
            const T c = tc[t], s = sign * ts[t];
            const T a = ire[base + k * st], b = iim[base + k * st];
            sr = fma(a, c, fma(-b, s, sr)); si = fma(a, s, fma(b, c, si));
            t += m; if (t >= K) t -= K;
        }
        ore[g] = sr; oim[g] = si;
    }
}

template <typename T, bool STATIC>
__device__ __forceinline__ void pme_body(const PmeArgs<T>& P) {
    const int tid = threadIdx.x, order = P.order;
    const int K0 = P.K[0], K1 = P.K[1], K2 = P.K[2];
    // ---- charge mesh in fixed point
    for (int g = tid; g < P.ng; g += PME_THREADS) P.acc[g] = 0ull;
    __syncthreads();
    {   // work item = (atom, a, b): the order^2 (x, y) offsets of an atom, each spreading its column along z
        const int per = order * order;
        for (long w = tid; w < (long)P.n_sel * per; w += PME_THREADS) {
            const int s = (int)(w / per), ab = (int)(w - (long)s * per), a = ab / order, b = ab - a * order;
            const int i = P.sel[s];
            const double q = P.qn[i];
            if (q == 0.0) continue;
            int t0, t1, t2; T d0, d1, d2;
            pme_locate(P, i, 0, t0, d0); pme_locate(P, i, 1, t1, d1); pme_locate(P, i, 2, t2, d2);
            T w0[PME_MAX_ORDER], w1[PME_MAX_ORDER], w2[PME_MAX_ORDER], dw[PME_MAX_ORDER];
            pme_splines(d0, order, w0, dw); pme_splines(d1, order, w1, dw); pme_splines(d2, order, w2, dw);
            // (constant indices only into the spline arrays: a runtime index would put them in scratch memory)
            T wa = (T)0, wb = (T)0;
#pragma unroll
            for (int k = 0; k < PME_MAX_ORDER; k++) { if (k == a) wa = w0[k]; if (k == b) wb = w1[k]; }
            int ix = t0 + a; if (ix >= K0) ix -= K0;
            int iy = t1 + b; if (iy >= K1) iy -= K1;
            const double qab = q * (double)wa * (double)wb;
#pragma unroll
            for (int c = 0; c < PME_MAX_ORDER; c++) {
                if (c >= order) break;
                int iz = t2 + c; if (iz >= K2) iz -= K2;
                const long long v = llrint(qab * (double)w2[c] * PME_FIX);
                atomicAdd(&P.acc[((size_t)ix * K1 + iy) * K2 + iz], (unsigned long long)v);
            }
        }
    }
    __syncthreads();
    for (int g = tid; g < P.ng; g += PME_THREADS) { P.a_re[g] = (T)((double)(long long)P.acc[g] * (1.0 / PME_FIX)); P.a_im[g] = (T)0; }
    __syncthreads();
    // ---- forward transform z, y, x
    pme_dft_axis(P, P.a_re, P.a_im, P.b_re, P.b_im, 2, (T)-1); __syncthreads();
    pme_dft_axis(P, P.b_re, P.b_im, P.a_re, P.a_im, 1, (T)-1); __syncthreads();
    pme_dft_axis(P, P.a_re, P.a_im, P.b_re, P.b_im, 0, (T)-1); __syncthreads();
    // ---- energy (with the frozen charges' transform), static copy, multiplication by eterm
    if (!STATIC && P.want_energy) {
        double e = 0.0;
        for (int g = tid; g < P.ng; g += PME_THREADS) {
            const double re = (double)P.b_re[g] + (P.have_static ? (double)P.qf_re[g] : 0.0), im = (double)P.b_im[g] + (P.have_static ? (double)P.qf_im[g] : 0.0);
            e += (double)P.eterm[g] * (re * re + im * im);
        }
        __shared__ double s_e[PME_THREADS / 64];
        e = wave_sum(e);
        if ((tid & 63) == 0) s_e[tid >> 6] = e;
        __syncthreads();
        if (tid == 0) { double t = 0.0; for (int w = 0; w < PME_THREADS / 64; w++) t += s_e[w]; P.epart[0] = 0.5 * t; }
    }
    for (int g = tid; g < P.ng; g += PME_THREADS) {
        if (STATIC) { P.qf_re[g] = P.b_re[g]; P.qf_im[g] = P.b_im[g]; }
        const T et = P.eterm[g];
        P.b_re[g] *= et; P.b_im[g] *= et;
    }
    __syncthreads();
    // ---- backward transform x, y, z (the potential mesh is the real part)
    pme_dft_axis(P, P.b_re, P.b_im, P.a_re, P.a_im, 0, (T)1); __syncthreads();
    pme_dft_axis(P, P.a_re, P.a_im, P.b_re, P.b_im, 1, (T)1); __syncthreads();
    pme_dft_axis(P, P.b_re, P.b_im, P.a_re, P.a_im, 2, (T)1); __syncthreads();
    if (STATIC) {
        for (int g = tid; g < P.ng; g += PME_THREADS) P.phi_f[g] = P.a_re[g];
        return;
    }
    // ---- gather: the force on every selected atom from the potential of all charges (mobile: just computed; frozen: static)
    for (int s = tid; s < P.n_sel; s += PME_THREADS) {
        const int i = P.sel[s];
        const double q = P.qn[i];
        double fx = 0.0, fy = 0.0, fz = 0.0;
        if (q != 0.0) {
            int t0, t1, t2; T d0, d1, d2;
            pme_locate(P, i, 0, t0, d0); pme_locate(P, i, 1, t1, d1); pme_locate(P, i, 2, t2, d2);
            T w0[PME_MAX_ORDER], w1[PME_MAX_ORDER], w2[PME_MAX_ORDER], e0[PME_MAX_ORDER], e1[PME_MAX_ORDER], e2[PME_MAX_ORDER];
            pme_splines(d0, order, w0, e0); pme_splines(d1, order, w1, e1); pme_splines(d2, order, w2, e2);
#pragma unroll
            for (int a = 0; a < PME_MAX_ORDER; a++) {
                if (a >= order) break;
                int ix = t0 + a; if (ix >= K0) ix -= K0;
#pragma unroll
                for (int b = 0; b < PME_MAX_ORDER; b++) {
                    if (b >= order) break;
                    int iy = t1 + b; if (iy >= K1) iy -= K1;
                    T sx = (T)0, sz = (T)0;   // sum_c w2 phi and sum_c dw2 phi of this column
#pragma unroll
                    for (int c = 0; c < PME_MAX_ORDER; c++) {
                        if (c >= order) break;
                        int iz = t2 + c; if (iz >= K2) iz -= K2;
                        const size_t g = ((size_t)ix * K1 + iy) * K2 + iz;
                        const T phi = P.a_re[g] + (P.have_static ? P.phi_f[g] : (T)0);
                        sx = fma(w2[c], phi, sx); sz = fma(e2[c], phi, sz);
                    }
                    fx += (double)(e0[a] * w1[b] * sx); fy += (double)(w0[a] * e1[b] * sx); fz += (double)(w0[a] * w1[b] * sz);
                }
            }
            fx *= -q * K0 * P.box.invL[0]; fy *= -q * K1 * P.box.invL[1]; fz *= -q * K2 * P.box.invL[2];
        }
        P.frec[i] = fx; P.frec[(size_t)P.n + i] = fy; P.frec[2 * (size_t)P.n + i] = fz;
    }
}

template <typename T, bool STATIC>
__global__ void __launch_bounds__(PME_THREADS) k_pme(PmeArgs<T> P) { pme_body<T, STATIC>(P); }
