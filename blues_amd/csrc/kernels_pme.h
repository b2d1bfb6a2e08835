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
#define PME_FIX32 67108864.0f      // 2^26: the same in 32 bits (fast path: |mesh value| < 32 e, resolution 1.5e-8 e)

template <typename T> struct PmeArgs {
    int n, K[3], order, ng;
    int n_sel; const int* sel;         // atoms this launch spreads (and gathers): caller indices
    const double* x[3];                // master positions
    const double* qn;                  // [n] NonbondedForce charges (alchemical atoms: 0)
    const double* qn_full;             // [n] every atom's own charge: the NonbondedForce of the NON-alchemical System (want_energy bit 1: the mesh energy the MD / alch context would report)
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

// the same with the order known at compile time: every loop unrolls, the arrays stay in registers (with a runtime order they are indexed
// dynamically and live in scratch memory -- measured: the per-step kernel spent most of its 280 us there)
template <typename T, int ORDER>
__device__ __forceinline__ void pme_splines_c(T dr, T* w, T* dw) {
    if (ORDER <= 0) { return; }
#pragma unroll
    for (int k = 0; k < (ORDER > 0 ? ORDER : 1); k++) w[k] = (T)0;
    w[1] = dr; w[0] = (T)1 - dr;
#pragma unroll
    for (int l = 3; l < ORDER; l++) {
        const T div = (T)1 / (T)(l - 1);
        w[l - 1] = div * dr * w[l - 2];
#pragma unroll
        for (int k = 1; k < l - 1; k++) w[l - k - 1] = div * ((dr + (T)k) * w[l - k - 2] + ((T)(l - k) - dr) * w[l - k - 1]);
        w[0] = div * ((T)1 - dr) * w[0];
    }
    dw[0] = -w[0];
#pragma unroll
    for (int k = 1; k < ORDER; k++) dw[k] = w[k - 1] - w[k];
    const T div = (T)1 / (T)(ORDER - 1);
    w[ORDER - 1] = div * dr * w[ORDER - 2];
#pragma unroll
    for (int k = 1; k < ORDER - 1; k++) w[ORDER - k - 1] = div * ((dr + (T)k) * w[ORDER - k - 2] + ((T)(ORDER - k) - dr) * w[ORDER - k - 1]);
    w[0] = div * ((T)1 - dr) * w[0];
}
// ORDER > 0: compile-time order; ORDER == 0: the runtime one
template <typename T, int ORDER>
__device__ __forceinline__ void pme_splines_any(T dr, int order, T* w, T* dw) {
    if constexpr (ORDER > 0) pme_splines_c<T, ORDER>(dr, w, dw); else pme_splines(dr, order, w, dw);
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

template <typename T, bool STATIC, int ORDER = 0>
__device__ __forceinline__ void pme_body(const PmeArgs<T>& P) {
    const int tid = threadIdx.x, order = ORDER > 0 ? ORDER : P.order;
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
            pme_splines_any<T, ORDER>(d0, order, w0, dw); pme_splines_any<T, ORDER>(d1, order, w1, dw); pme_splines_any<T, ORDER>(d2, order, w2, dw);
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
            pme_splines_any<T, ORDER>(d0, order, w0, e0); pme_splines_any<T, ORDER>(d1, order, w1, e1); pme_splines_any<T, ORDER>(d2, order, w2, e2);
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


// out-of-line copy of the general path for the fast kernel's fallbacks (kept out of its register allocation)
__device__ __noinline__ void pme_body_general_f(const PmeArgs<float>& P) { pme_body<float, false, 5>(P); }

// ---- fast path of the per-step (mobile) launch in mixed precision: everything between the spread and the gather lives in LDS, and
// the transforms are PRUNED.  The mobile charges occupy a small block of the mesh (their stencils span Lx x Ly x Lz points, ~13^3 of
// 18 x 27 x 36 for the benchmark), the potential is only needed on that block, and the charge mesh is real (half spectrum along z):
//   forward   z: (Lx, Ly) lines,  Lz inputs -> Hz = K2/2+1 outputs      y: (Lx, Hz) lines, Ly inputs -> K1      x: (K1, Hz) lines, Lx -> K0
//   backward  x: K0 inputs -> Lx outputs     y: K1 -> Ly     z: Hz (Hermitian weights 1,2,..,2[,1]) -> Lz real outputs
// = 0.5 M complex multiply-adds instead of 2.8 M for the full complex transforms.  One thread per (line, output); the inputs of a
// line are read from LDS at the same address by all the lanes that share the line (broadcast).
#define PME_LDS_Y 9280             // complex entries of buffer Y: the half spectrum K0 x K1 x (K2/2+1) (18 x 27 x 19 = 9,234 for the benchmark mesh)
#define PME_LDS_X 7424             // complex entries of buffer X: the block's lines along one full axis, Lx x K1 x (K2/2+1)
#define PME_LDS_ATOMS 768          // mobile atoms whose mesh position is cached in LDS (32 B each); more: the general path

struct PmeRegion { int s[3], L[3]; };   // block of the mesh: points (s + a) mod K, a < L, per axis

// (the buffers are addressed as offsets into ONE __shared__ array: through plain pointers the compiler cannot tell LDS from global memory
// and emits flat loads, each iteration then waiting out a full memory latency -- measured 296 us per launch instead of ~20)
extern __shared__ __align__(16) float2 pme_lds[];

template <int DIR>   // DIR = -1 forward, +1 backward
__device__ __forceinline__ void pme_pass(int in, int out, int nlines, int line_div,
                                         int in_s0, int in_s1, int in_st, int n_in, int k0, int out_s0, int out_s1, int out_st, int n_out, int m0,
                                         int K, int tw) {
    // line l = (l / line_div, l % line_div); element j of its input sits at in_s0*(l/line_div) + in_s1*(l%line_div) + in_st*j and stands for
    // mesh index (k0 + j) mod K; output o stands for index (m0 + o) mod K.  A thread keeps ONE output index o and walks the lines
    // (its phase exp(DIR 2 pi i m k0 / K) and the per-input rotation are loop invariants; no integer division per work item).  The phase
    // is advanced by a complex rotation per input (4 FMAs, no table look-up; <= 36 steps in fp32: ~2e-6 of a few % of the force).
    const int o = threadIdx.x % n_out, lg = threadIdx.x / n_out, nlg = PME_THREADS / n_out;
    if (lg >= nlg) return;   // (the threads past the last whole group of n_out sit this pass out)
    int m = m0 + o; if (m >= K) m -= K;
    int k = k0; if (k >= K) k -= K;
    const float2 p0 = pme_lds[tw + (m * k) % K], rot = pme_lds[tw + m];
    const float c0 = p0.x, s0 = DIR < 0 ? -p0.y : p0.y, rc = rot.x, rs = DIR < 0 ? -rot.y : rot.y;
    int l0 = lg / line_div, l1 = lg - l0 * line_div;
    const int d0 = nlg / line_div, d1 = nlg - d0 * line_div;
    for (int l = lg; l < nlines; l += nlg) {
        int src = in + in_s0 * l0 + in_s1 * l1;
        float c = c0, sn = s0, sr = 0.0f, si = 0.0f;
#pragma unroll 4
        for (int j = 0; j < n_in; j++) {
            const float2 v = pme_lds[src];
            src += in_st;
            sr = fmaf(v.x, c, fmaf(-v.y, sn, sr)); si = fmaf(v.x, sn, fmaf(v.y, c, si));
            const float c2 = fmaf(c, rc, -sn * rs); sn = fmaf(c, rs, sn * rc); c = c2;
        }
        pme_lds[out + out_s0 * l0 + out_s1 * l1 + out_st * o] = make_float2(sr, si);
        l0 += d0; l1 += d1; if (l1 >= line_div) { l1 -= line_div; l0++; }
    }
}

template <int ORDER>
__device__ __forceinline__ void pme_fast_body(const PmeArgs<float>& P) {
    const int X = 0, Y = PME_LDS_X, tw0 = PME_LDS_X + PME_LDS_Y, tw1 = tw0 + P.K[0], tw2 = tw1 + P.K[1];   // offsets into pme_lds
    const int AT = (tw2 + P.K[2] + 1) & ~1;   // per-atom cache: int4 {t0, t1, t2, selected index} + float4 {dr0, dr1, dr2, q}, 4 float2 slots per atom
    int4* at_i = reinterpret_cast<int4*>(pme_lds + AT); float4* at_f = reinterpret_cast<float4*>(pme_lds + AT + 2 * PME_LDS_ATOMS);
    int* acc = reinterpret_cast<int*>(pme_lds + Y);   // the block's charge mesh in 32-bit fixed point (before any transform): integer adds commute
    __shared__ int s_lo[3], s_hi[3], s_ref[3], s_first;
    __shared__ double s_e[PME_THREADS / 64];
    const int tid = threadIdx.x, order = ORDER > 0 ? ORDER : P.order;
    const int K0 = P.K[0], K1 = P.K[1], K2 = P.K[2], Hz = K2 / 2 + 1;
    NB_STAMP(blockIdx.x == 0 && tid == 0, 9);
    // ---- twiddles and the block that holds the mobile charges' stencils (relative to the first charged atom, through the periodic wrap)
    for (int k = tid; k < K0; k += PME_THREADS) pme_lds[tw0 + k] = make_float2(P.tw_cos[0][k], P.tw_sin[0][k]);
    for (int k = tid; k < K1; k += PME_THREADS) pme_lds[tw1 + k] = make_float2(P.tw_cos[1][k], P.tw_sin[1][k]);
    for (int k = tid; k < K2; k += PME_THREADS) pme_lds[tw2 + k] = make_float2(P.tw_cos[2][k], P.tw_sin[2][k]);
    if (tid < 3) { s_lo[tid] = 1 << 30; s_hi[tid] = -(1 << 30); }
    if (tid == 0) s_first = 1 << 30;
    if (P.n_sel > PME_LDS_ATOMS) { pme_body_general_f(P); return; }   // (block-uniform) too many mobile atoms to cache: the general path
    __syncthreads();
    // every selected atom's mesh position, once: three dependent global loads per atom (index -> charge, coordinates), all atoms in parallel
    // -- the spread and the gather below then run out of LDS (reading them again per work item cost ~250 us of pure memory latency)
    for (int s = tid; s < P.n_sel; s += PME_THREADS) {
        const int i = P.sel[s];
        int4 ti; float4 fr; ti.w = i;
        pme_locate(P, i, 0, ti.x, fr.x); pme_locate(P, i, 1, ti.y, fr.y); pme_locate(P, i, 2, ti.z, fr.z);
        fr.w = (float)P.qn[i];
        at_i[s] = ti; at_f[s] = fr;
        int mine = fr.w != 0.0f ? s : (1 << 30);
        for (int o = 32; o > 0; o >>= 1) mine = min(mine, __shfl_xor(mine, o, 64));
        if ((tid & 63) == 0 && mine < (1 << 30)) atomicMin(&s_first, mine);
    }
    __syncthreads();
    if (tid < 3) s_ref[tid] = s_first < P.n_sel ? (tid == 0 ? at_i[s_first].x : (tid == 1 ? at_i[s_first].y : at_i[s_first].z)) : 0;
    __syncthreads();
    {   // extent of the charged atoms' mesh indices around the reference: reduced in the wave first (64 lanes on one LDS address serialise)
        int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-(1 << 30), -(1 << 30), -(1 << 30)};
        for (int s = tid; s < P.n_sel; s += PME_THREADS) {
            if (at_f[s].w == 0.0f) continue;
            const int4 ti = at_i[s];
            const int tt[3] = {ti.x, ti.y, ti.z};
#pragma unroll
            for (int d = 0; d < 3; d++) {
                int del = tt[d] - s_ref[d]; const int K = P.K[d];
                if (2 * del >= K) del -= K; else if (2 * del < -K) del += K;
                lo[d] = min(lo[d], del); hi[d] = max(hi[d], del);
            }
        }
#pragma unroll
        for (int d = 0; d < 3; d++) {
            for (int o = 32; o > 0; o >>= 1) { lo[d] = min(lo[d], __shfl_xor(lo[d], o, 64)); hi[d] = max(hi[d], __shfl_xor(hi[d], o, 64)); }
            if ((tid & 63) == 0) { atomicMin(&s_lo[d], lo[d]); atomicMax(&s_hi[d], hi[d]); }
        }
    }
    __syncthreads();
    NB_STAMP(blockIdx.x == 0 && tid == 0, 10);
    PmeRegion R;
    bool fits = true;
    for (int d = 0; d < 3; d++) {
        const int K = P.K[d];
        if (s_hi[d] < s_lo[d]) { R.s[d] = 0; R.L[d] = 1; continue; }   // no charged atom: an empty block
        R.L[d] = s_hi[d] - s_lo[d] + order;
        int st = (s_ref[d] + s_lo[d]) % K; if (st < 0) st += K;
        R.s[d] = st;
        if (R.L[d] >= K) { R.L[d] = K; R.s[d] = 0; }
    }
    const int Lx = R.L[0], Ly = R.L[1], Lz = R.L[2];
    fits = Lx * Ly * Lz <= PME_LDS_X && Lx * K1 * Hz <= PME_LDS_X && Lx * Ly * Hz <= PME_LDS_Y && K0 * K1 * Hz <= PME_LDS_Y
           && P.n_sel * 16 <= PME_LDS_X && (Lx * Ly * Lz + 4 + P.n_sel * 48 + 1) / 2 <= PME_LDS_Y;   // (room for the atoms' spline weights beside the meshes)
    if (!fits) { pme_body_general_f(P); return; }   // (block-uniform) a spread-out mobile set: the general path
    // ---- spread into the block (LDS, 64-bit integer adds: order-independent)
    for (int g = tid; g < Lx * Ly * Lz; g += PME_THREADS) acc[g] = 0;
    __syncthreads();
    // the atoms' spline weights once, into the (still unused) X buffer: 32 floats per atom = w_x[8] w_y[8] w_z[8] + padding; then a thread per
    // (atom, x offset, y offset) column.  (A thread per column that recomputed the splines, or a thread per atom with all order^3 points, both
    // cost several times more: redundant spline arithmetic / a 125-point unrolled body spilling registers.)
    float* wts = reinterpret_cast<float*>(pme_lds + X);
    for (int s = tid; s < P.n_sel; s += PME_THREADS) {
        const float4 fr = at_f[s];
        float w0[ORDER > 0 ? ORDER : PME_MAX_ORDER], w1[ORDER > 0 ? ORDER : PME_MAX_ORDER], w2[ORDER > 0 ? ORDER : PME_MAX_ORDER], dw[ORDER > 0 ? ORDER : PME_MAX_ORDER];
        pme_splines_any<float, ORDER>(fr.x, order, w0, dw); pme_splines_any<float, ORDER>(fr.y, order, w1, dw); pme_splines_any<float, ORDER>(fr.z, order, w2, dw);
#pragma unroll
        for (int k = 0; k < (ORDER > 0 ? ORDER : PME_MAX_ORDER); k++) { wts[s * 32 + k] = w0[k]; wts[s * 32 + 8 + k] = w1[k]; wts[s * 32 + 16 + k] = w2[k]; }
    }
    __syncthreads();
    {
        const int per = order * order;
        for (int w = tid; w < P.n_sel * per; w += PME_THREADS) {
            const int s = w / per, ab = w - s * per, a = ab / order, b = ab - a * order;
            const float4 fr = at_f[s]; const int4 ti = at_i[s];
            if (fr.w == 0.0f) continue;
            int ax = ti.x + a - R.s[0]; if (ax < 0) ax += K0; if (ax >= K0) ax -= K0;
            int ay = ti.y + b - R.s[1]; if (ay < 0) ay += K1; if (ay >= K1) ay -= K1;
            int az0 = ti.z - R.s[2]; if (az0 < 0) az0 += K2;
            const float qab = fr.w * PME_FIX32 * wts[s * 32 + a] * wts[s * 32 + 8 + b];
            const int row = (ax * Ly + ay) * Lz;
#pragma unroll
            for (int c = 0; c < (ORDER > 0 ? ORDER : PME_MAX_ORDER); c++) {
                if (c >= order) break;
                int az = az0 + c; if (az >= K2) az -= K2;
                atomicAdd(&acc[row + az], __float2int_rn(qab * wts[s * 32 + 16 + c]));
            }
        }
    }
    __syncthreads();
    NB_STAMP(blockIdx.x == 0 && tid == 0, 11);
    // block -> X as complex [Lx][Ly][Lz]   (Y is being read as acc: convert through registers, then write X)
    for (int g = tid; g < Lx * Ly * Lz; g += PME_THREADS) pme_lds[X + g] = make_float2((float)acc[g] * (1.0f / PME_FIX32), 0.0f);
    __syncthreads();
    NB_STAMP(blockIdx.x == 0 && tid == 0, 12);
    // ---- forward: z (X[Lx][Ly][Lz] -> Y[Lx][Ly][Hz]), y (-> X[Lx][K1][Hz]), x (-> Y[K0][K1][Hz])
    pme_pass<-1>(X, Y, Lx * Ly, Ly, Ly * Lz, Lz, 1, Lz, R.s[2], Ly * Hz, Hz, 1, Hz, 0, K2, tw2); __syncthreads();
    pme_pass<-1>(Y, X, Lx * Hz, Hz, Ly * Hz, 1, Hz, Ly, R.s[1], K1 * Hz, 1, Hz, K1, 0, K1, tw1); __syncthreads();
    pme_pass<-1>(X, Y, K1 * Hz, Hz, Hz, 1, K1 * Hz, Lx, R.s[0], Hz, 1, K1 * Hz, K0, 0, K0, tw0); __syncthreads();
    NB_STAMP(blockIdx.x == 0 && tid == 0, 13);
    // ---- energy with the frozen charges' transform (half spectrum: interior k_z count twice), then eterm
    if (P.want_energy) {
        double e = 0.0;
        for (int g = tid; g < K0 * K1 * Hz; g += PME_THREADS) {
            const int kz = g % Hz, xy = g / Hz; const size_t gf = (size_t)xy * K2 + kz;
            const double re = (double)pme_lds[Y + g].x + (P.have_static ? (double)P.qf_re[gf] : 0.0), im = (double)pme_lds[Y + g].y + (P.have_static ? (double)P.qf_im[gf] : 0.0);
            e += ((kz == 0 || 2 * kz == K2) ? 1.0 : 2.0) * (double)P.eterm[gf] * (re * re + im * im);
        }
        e = wave_sum(e);
        if ((tid & 63) == 0) s_e[tid >> 6] = e;
        __syncthreads();
        if (tid == 0) { double t = 0.0; for (int w = 0; w < PME_THREADS / 64; w++) t += s_e[w]; P.epart[0] = 0.5 * t; }
    }
    for (int g = tid; g < K0 * K1 * Hz; g += PME_THREADS) {
        const int kz = g % Hz, xy = g / Hz;
        // (eterm and the weight of the half spectrum's missing conjugates, 1 for k_z = 0 and K2/2, 2 otherwise: the last backward pass then is a plain sum)
        const float et = P.eterm[(size_t)xy * K2 + kz] * ((kz == 0 || 2 * kz == K2) ? 1.0f : 2.0f);
        pme_lds[Y + g].x *= et; pme_lds[Y + g].y *= et;
    }
    __syncthreads();
    NB_STAMP(blockIdx.x == 0 && tid == 0, 14);
    // ---- backward: x (Y[K0][K1][Hz] -> X[Lx][K1][Hz]), y (-> Y[Lx][Ly][Hz]), z (Hermitian, -> X[Lx][Ly][Lz], real part)
    pme_pass<1>(Y, X, K1 * Hz, Hz, Hz, 1, K1 * Hz, K0, 0, Hz, 1, K1 * Hz, Lx, R.s[0], K0, tw0); __syncthreads();
    pme_pass<1>(X, Y, Lx * Hz, Hz, K1 * Hz, 1, Hz, K1, 0, Ly * Hz, 1, Hz, Ly, R.s[1], K1, tw1); __syncthreads();
    pme_pass<1>(Y, X, Lx * Ly, Ly, Ly * Hz, Hz, 1, Hz, 0, Ly * Lz, Lz, 1, Lz, R.s[2], K2, tw2); __syncthreads();
    NB_STAMP(blockIdx.x == 0 && tid == 0, 15);
    // the frozen charges' potential on the block, once (Y is free now): the gather then reads LDS only
    float* phf = reinterpret_cast<float*>(pme_lds + Y);
    if (P.have_static) for (int g = tid; g < Lx * Ly * Lz; g += PME_THREADS) {
        const int az = g % Lz, ay = (g / Lz) % Ly, ax = g / (Lz * Ly);
        int ix = R.s[0] + ax; if (ix >= K0) ix -= K0;
        int iy = R.s[1] + ay; if (iy >= K1) iy -= K1;
        int iz = R.s[2] + az; if (iz >= K2) iz -= K2;
        phf[g] = P.phi_f[((size_t)ix * K1 + iy) * K2 + iz];
    }
    __syncthreads();
    // ---- gather: weights and their derivatives once per atom (into the free part of Y, behind the frozen potential's block), then 32 lanes per
    // atom, lane = (x offset, y offset) column, summed over the atom's lanes in a fixed order
    float* wg = phf + ((Lx * Ly * Lz + 3) & ~3);   // 48 floats per atom: w_x w_y w_z e_x e_y e_z (8 each)
    for (int s = tid; s < P.n_sel; s += PME_THREADS) {
        const float4 fr = at_f[s];
        float w0[ORDER > 0 ? ORDER : PME_MAX_ORDER], w1[ORDER > 0 ? ORDER : PME_MAX_ORDER], w2[ORDER > 0 ? ORDER : PME_MAX_ORDER];
        float e0[ORDER > 0 ? ORDER : PME_MAX_ORDER], e1[ORDER > 0 ? ORDER : PME_MAX_ORDER], e2[ORDER > 0 ? ORDER : PME_MAX_ORDER];
        pme_splines_any<float, ORDER>(fr.x, order, w0, e0); pme_splines_any<float, ORDER>(fr.y, order, w1, e1); pme_splines_any<float, ORDER>(fr.z, order, w2, e2);
#pragma unroll
        for (int k = 0; k < (ORDER > 0 ? ORDER : PME_MAX_ORDER); k++) {
            wg[s * 48 + k] = w0[k]; wg[s * 48 + 8 + k] = w1[k]; wg[s * 48 + 16 + k] = w2[k];
            wg[s * 48 + 24 + k] = e0[k]; wg[s * 48 + 32 + k] = e1[k]; wg[s * 48 + 40 + k] = e2[k];
        }
    }
    __syncthreads();
    for (int s0 = (tid >> 5); s0 < P.n_sel; s0 += PME_THREADS / 32) {
        const int ab = tid & 31;
        const float4 fr = at_f[s0]; const int4 ti = at_i[s0];
        float fx = 0.0f, fy = 0.0f, fz = 0.0f;
        if (fr.w != 0.0f && ab < order * order) {
            const int a = ab / order, b = ab - a * order;
            const float* wv = wg + s0 * 48;
            int ax = ti.x + a - R.s[0]; if (ax < 0) ax += K0; if (ax >= K0) ax -= K0;
            int ay = ti.y + b - R.s[1]; if (ay < 0) ay += K1; if (ay >= K1) ay -= K1;
            int az0 = ti.z - R.s[2]; if (az0 < 0) az0 += K2;
            const int row = (ax * Ly + ay) * Lz;
            float sx = 0.0f, sz = 0.0f;
#pragma unroll
            for (int c = 0; c < (ORDER > 0 ? ORDER : PME_MAX_ORDER); c++) {
                if (c >= order) break;
                int az = az0 + c; if (az >= K2) az -= K2;
                const float phi = pme_lds[X + row + az].x + (P.have_static ? phf[row + az] : 0.0f);
                sx = fmaf(wv[16 + c], phi, sx); sz = fmaf(wv[40 + c], phi, sz);
            }
            fx = wv[24 + a] * wv[8 + b] * sx; fy = wv[a] * wv[32 + b] * sx; fz = wv[a] * wv[8 + b] * sz;
        }
        double gx = (double)fx, gy = (double)fy, gz = (double)fz;   // fixed-order sum over the 32 lanes of the atom
        for (int o = 16; o > 0; o >>= 1) { gx += __shfl_xor(gx, o, 64); gy += __shfl_xor(gy, o, 64); gz += __shfl_xor(gz, o, 64); }
        if ((tid & 31) == 0) {
            const int i = ti.w; const double q = (double)fr.w;
            P.frec[i] = -q * K0 * P.box.invL[0] * gx; P.frec[(size_t)P.n + i] = -q * K1 * P.box.invL[1] * gy; P.frec[2 * (size_t)P.n + i] = -q * K2 * P.box.invL[2] * gz;
        }
    }
    NB_STAMP(blockIdx.x == 0 && tid == 0, 16);
}

__global__ void __launch_bounds__(PME_THREADS) k_pme_fast(PmeArgs<float> P) { pme_fast_body<5>(P); }   // (the engine accepts order 5 only: OpenMM's)
#define PME_FAST_LDS(K0, K1, K2) ((size_t)(PME_LDS_X + PME_LDS_Y) * 8 + (size_t)((K0) + (K1) + (K2) + 2) * 8 + (size_t)PME_LDS_ATOMS * 32)

template <typename T, bool STATIC>
__global__ void __launch_bounds__(PME_THREADS) k_pme(PmeArgs<T> P) { pme_body<T, STATIC, 5>(P); }
