// device_common.h -- shared device-side types and math for the gfx950 NCMC engine.
//
// Data layout in HBM:
//   master state   x,v (fp64 SoA, caller's atom order)   -- integrator, bonded and alchemical kernels
//   tile image     AtomF / AtomD (32 / 64 B AoS, Hilbert-sorted order) -- nonbonded kernel
//     positions are FIXED-POINT fractions of the box edge (u32 or u64): the difference of
//     two such integers is exact and already minimum-imaged by wrap-around, so the pair
//     kernel needs no periodic-shift arithmetic and loses no precision to large coordinates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ONE_4PI_EPS0 138.935456
#define KB_KJ 0.0083144626
#define TWO_OVER_SQRT_PI 1.1283791670955126
#define TWO_OVER_SQRT_PI_F 1.1283791671f
#define WAVE 64
#define FLAG_ALCH 1u
#define FLAG_MOBILE 2u

struct AtomF {  // 32 B
    uint32_t x, y, z;
    float q;   // charge * sqrt(ONE_4PI_EPS0)
    float hs;  // sigma / 2            (sigma_ij = hs_i + hs_j)
    float se;  // 2 * sqrt(epsilon)    (4 eps_ij = se_i * se_j)
    uint32_t flags;
    int32_t orig;
};
struct AtomD {  // 64 B
    uint64_t x, y, z;
    double q, hs, se;
    uint32_t flags;
    int32_t orig;
    uint64_t pad;
};

template <typename R> struct Img;
template <> struct Img<float> {
    using Atom = AtomF; using ufix = uint32_t; using sfix = int32_t;
};
template <> struct Img<double> {
    using Atom = AtomD; using ufix = uint64_t; using sfix = int64_t;
};

struct Box3 { double L[3]; double invL[3]; };

// Device-resident step counters for hipGraph replays of the steady-state NCMC step: a captured graph has
// its kernel arguments frozen, so everything that changes from step to step is derived from here.
//   integrate reads kint and publishes kpass = kint+1 for the force kernels that follow it;
//   finalize (last kernel of a unit) publishes kint = kpass for the next integrate.
struct DevCtrl {
    int kint, kpass;       // unit counters (never read and written inside the same kernel)
    int L0;                // lambda index of slot 0 of the force pass that precedes unit 0
    unsigned draw0;        // O-substep draw index used by unit 0
    int trace0;            // work-trace slot written by unit 0
    int n_lambda;
    const double* tab_ls; const double* tab_le;  // [n_lambda+1]
};

// ---------------------------------------------------------------- Philox4x32-10 (same stream as oracle/blues_oracle.c)
__host__ __device__ inline void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__host__ __device__ inline void gaussians3(uint64_t seed, uint32_t stream, uint32_t draw, uint32_t atom, double g[3]) {
    uint32_t r[4];
    philox4x32(atom, draw, stream, 0x424C5545u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const double S = 2.3283064365386963e-10;
    double u1 = ((double)r[0] + 0.5) * S, u2 = ((double)r[1] + 0.5) * S;
    double u3 = ((double)r[2] + 0.5) * S, u4 = ((double)r[3] + 0.5) * S;
    double ra = sqrt(-2.0 * log(u1)), rb = sqrt(-2.0 * log(u3));
    const double TWO_PI = 6.283185307179586476925286766559;
    g[0] = ra * cos(TWO_PI * u2);
    g[1] = ra * sin(TWO_PI * u2);
    g[2] = rb * cos(TWO_PI * u4);
}

// ---------------------------------------------------------------- pair math
// fp32 erfc: erfc(x) = exp(-x^2) * t * P(t), t = 1/(1+0.55x); |rel err| < 3.5e-7 on [0,4] in fp32.
__device__ inline float erfc_scaled_f(float x) {  // returns erfc(x)*exp(x^2)
    float t = __builtin_amdgcn_rcpf(1.0f + 0.55f * x);  // v_rcp_f32 (1 ulp); the IEEE-rounded division costs 10 extra VALU ops per pair
    float p = -7.091228587e-02f;
    p = fmaf(p, t, 3.810680318e-01f);
    p = fmaf(p, t, -7.630493514e-01f);
    p = fmaf(p, t, 5.768974586e-01f);
    p = fmaf(p, t, -5.936511826e-02f);
    p = fmaf(p, t, 3.227979795e-01f);
    p = fmaf(p, t, 3.017286883e-01f);
    p = fmaf(p, t, 3.108345779e-01f);
    return p * t;
}

// The Ewald direct-space pair FORCE without transcendentals (mixed precision).  With x = alpha r,
//   -dU/dr / r = qq [erfc(x)/r + 2 alpha/sqrt(pi) exp(-x^2)] / r^2 = qq [1/r^3 - T(r^2)],
//   T(u) = [erf(x) - 2/sqrt(pi) x exp(-x^2)] / r^3 = alpha^3 (2/sqrt(pi)) (2/3 - 2/5 x^2 + 1/7 x^4 - ...),
// an entire function of u = r^2: a degree-9 Chebyshev fit on [0, cutoff^2] (host: fit_ewald_poly) is good to 3e-7 of T(0),
// below the fp32 rounding of the 1/r^3 it is subtracted from.  Near the cutoff the two terms agree to two digits, so the
// screened force of such a pair carries ~1e-5 of ITS OWN size in error -- 1e-7 of the 1/r^3 scale, i.e. nothing on the
// scale the 1e-5 force tolerance is stated on (the largest force component / the norm of the force vector).  This replaces
// v_exp, v_rcp and a degree-7 polynomial (22 instructions, two of them quarter-rate) by 10 FMAs.
#define EWALD_POLY_DEG 9
// The fit is only as good as T is smooth on [0, cutoff^2]: at alpha rc = 2.15 (ewald tolerance 5e-3) the fp32 Horner form is
// within 1.2e-6 nm^-3 of T everywhere, at 2.63 (5e-4) within 1.6e-5, at 3.29 (1e-5) within 6e-4 -- more than the screened force
// of a pair near the cutoff.  The host measures the residual of what it fitted (fit_ewald_poly) and sets `exact` when it
// exceeds EWALD_POLY_MAX_RESIDUAL: the kernels then evaluate the erfc/exp form instead (22 instructions more per pair).
#define EWALD_POLY_MAX_RESIDUAL 5e-6   // nm^-3, absolute: times k_e q_i q_j r <~ 50 kJ/mol/nm^-2 this stays below 1e-5 of a typical largest force
struct EwaldPoly { float c[EWALD_POLY_DEG + 1]; float wa; int exact; float residual; };   // T(u) = sum c[k] w^k, w = wa u - 1; exact != 0: do not use the fit

template <typename R> struct NbConst {
    R scale[3];   // box edge / 2^32 (or 2^64)
    R rc2;        // cutoff^2
    R alpha;      // Ewald alpha
    double rlist2;  // (cutoff+skin)^2: list radius for frozen candidates
    double rlist2_m;  // list radius for mobile candidates (derive_margins in blues_engine.hip)
    double dscale[3];
    float rp2, rp2_m;  // (cutoff + inner margin)^2 of the pruned per-atom lists, frozen candidates (current position) / mobile candidates (position at the list build) (nonbonded_atom_body)
    EwaldPoly ew;      // mixed precision only
};

// Regular (non-alchemical) pair: 12-6 LJ + erfc-screened Coulomb.  r2 < cutoff^2 is the caller's business.
// q is pre-multiplied by sqrt(ONE_4PI_EPS0); sig = hs_i+hs_j; eps4 = se_i*se_j.
// Returns fscale = -dU/dr / r ; *e_lj, *e_c energies (dead code in the callers that do not use them).
template <typename R> __device__ inline R pair_regular(R r2, R qq, R sig, R eps4, const NbConst<R>& c, R* e_lj, R* e_c);

// EXACT = false: the polynomial form above; true: erfc(x) exp(x^2) by its rational fit and v_exp (what round 2 ran everywhere).
template <bool EXACT> __device__ __forceinline__ float pair_regular_f(float r2, float qq, float sig, float eps4, const NbConst<float>& c, float* e_lj, float* e_c) {
    const float inv_r = __builtin_amdgcn_rsqf(r2);  // v_rsq_f32; r2 is never subnormal here (fixed-point resolution is 2e-9 nm)
    const float inv_r2 = inv_r * inv_r;
    const float s2 = sig * sig * inv_r2, s6 = s2 * s2 * s2, t = eps4 * s6;
    const float flj = t * fmaf(12.0f, s6, -6.0f);
    // energies (only the energy kernels keep this part alive)
    *e_lj = t * (s6 - 1.0f);
    const float ar = c.alpha * (r2 * inv_r);
    const float ex = __expf(-ar * ar), ec = erfc_scaled_f(ar) * ex;
    *e_c = qq * ec * inv_r;
    float g;
    if (EXACT) g = fmaf(TWO_OVER_SQRT_PI_F * c.alpha, ex, ec * inv_r) * inv_r2;
    else {
        const float w = fmaf(r2, c.ew.wa, -1.0f);
        float T = c.ew.c[EWALD_POLY_DEG];
#pragma unroll
        for (int k = EWALD_POLY_DEG - 1; k >= 0; k--) T = fmaf(T, w, c.ew.c[k]);
        g = fmaf(inv_r, inv_r2, -T);
    }
    return fmaf(qq, g, flj * inv_r2);
}
// (the kernels off the benchmark path decide per pair: a wave-uniform branch on a kernel argument)
template <> __device__ __forceinline__ float pair_regular<float>(float r2, float qq, float sig, float eps4, const NbConst<float>& c, float* e_lj, float* e_c) {
    return c.ew.exact ? pair_regular_f<true>(r2, qq, sig, eps4, c, e_lj, e_c) : pair_regular_f<false>(r2, qq, sig, eps4, c, e_lj, e_c);
}

template <> __device__ inline double pair_regular<double>(double r2, double qq, double sig, double eps4, const NbConst<double>& c, double* e_lj, double* e_c) {
    const double alpha = c.alpha;
    double r = sqrt(r2), inv_r = 1.0 / r, inv_r2 = inv_r * inv_r;
    double s2 = sig * sig * inv_r2, s6 = s2 * s2 * s2;
    *e_lj = eps4 * (s6 * s6 - s6);
    double f = eps4 * (12.0 * s6 * s6 - 6.0 * s6) * inv_r2;
    double ar = alpha * r;
    double ec = erfc(ar), ex = exp(-ar * ar);
    *e_c = qq * ec * inv_r;
    f += qq * (ec * inv_r + TWO_OVER_SQRT_PI * alpha * ex) * inv_r2;
    return f;
}

// fp64 forms used by the alchemical kernel.
// softcore LJ: U = ls*4 eps x (x-1), x = 1/(alpha(1-ls) + (r/sig)^6).  returns U, *fs = -dU/dr / r
__device__ inline double softcore_lj_d(double r2, double sig, double eps, double ls, double sc_alpha, double* fs) {
    if (eps == 0.0 || sig == 0.0) { *fs = 0.0; return 0.0; }
    double s2 = sig * sig, q2 = r2 / s2, q6 = q2 * q2 * q2;
    double x = 1.0 / (sc_alpha * (1.0 - ls) + q6);
    *fs = ls * 4.0 * eps * (2.0 * x - 1.0) * x * x * 6.0 * q2 * q2 / s2;
    return ls * 4.0 * eps * x * (x - 1.0);
}
// Coulomb per unit lambda_electrostatics; screened = erfc(alpha r)/r, else 1/r.
__device__ inline double coulomb_d(double r2, double qq, double alpha, bool screened, double* fs) {
    if (qq == 0.0) { *fs = 0.0; return 0.0; }
    double r = sqrt(r2), pre = ONE_4PI_EPS0 * qq;
    if (!screened) { *fs = pre / (r * r2); return pre / r; }
    double ar = alpha * r, ec = erfc(ar);
    *fs = pre * (ec / r + TWO_OVER_SQRT_PI * alpha * exp(-ar * ar)) / r2;
    return pre * ec / r;
}
__device__ inline double plain_lj_d(double r2, double sig, double eps, double* fs) {
    if (eps == 0.0) { *fs = 0.0; return 0.0; }
    double sr2 = sig * sig / r2, sr6 = sr2 * sr2 * sr2;
    *fs = 4.0 * eps * (12.0 * sr6 * sr6 - 6.0 * sr6) / r2;
    return 4.0 * eps * (sr6 * sr6 - sr6);
}

// ---- fast fp64 forms for the mixed-precision mode of the alchemical kernel: hardware reciprocal / rsqrt seeds refined
// by two Newton steps (full fp64 accuracy to ~1 ulp, 6-9 VALU ops instead of 12-22), and erfc through
// erfc(x) = exp(-x^2) P(u), u = A/(1+0.3x) + B: degree-16 fit, |rel err| < 6e-15 on [0,6] (scripts/gen_erfcx_coeffs.py);
// the libm erfc it replaces is ~140 VALU ops and shares no work with the exp() the force needs anyway.
__device__ __forceinline__ double rcp_fast_d(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ double rsqrt_fast_d(double x) {
    double r = __builtin_amdgcn_rsq(x);
    double h = 0.5 * x * r;
    r = fma(fma(-h, r, 0.5), r, r);
    h = 0.5 * x * r;
    return fma(fma(-h, r, 0.5), r, r);
}
#define ERFCX_A 3.1111111111111116
#define ERFCX_B -2.1111111111111116
__device__ __forceinline__ double erfcx_fast_d(double x) {  // erfc(x) * exp(x^2), x >= 0
    const double t = rcp_fast_d(1.0 + 0.3 * fmin(x, 6.0));
    const double u = fma(ERFCX_A, t, ERFCX_B);
    double p = -1.12889145348163940e-10;
    p = fma(p, u, 3.23086561221761168e-10);
    p = fma(p, u, 2.28679856732028227e-09);
    p = fma(p, u, -9.29632928364947494e-09);
    p = fma(p, u, -4.38466056753230290e-08);
    p = fma(p, u, 2.06382334159081147e-07);
    p = fma(p, u, 1.11132914875102761e-06);
    p = fma(p, u, -3.69277184520888158e-06);
    p = fma(p, u, -3.64511703162744848e-05);
    p = fma(p, u, -1.11587331375040937e-05);
    p = fma(p, u, 1.07582460856189941e-03);
    p = fma(p, u, 7.61123825488967468e-03);
    p = fma(p, u, 3.13096026232882255e-02);
    p = fma(p, u, 9.18900964165779988e-02);
    p = fma(p, u, 2.04911494369103497e-01);
    p = fma(p, u, 3.54125035524154419e-01);
    return fma(p, u, 3.09126743813179228e-01);
}
// screened Coulomb per unit lambda_electrostatics, fast forms (same value as coulomb_d(.., screened=true) to rounding)
__device__ __forceinline__ double coulomb_fast_d(double r2, double qq, double alpha, double* fs) {
    if (qq == 0.0) { *fs = 0.0; return 0.0; }
    const double inv_r = rsqrt_fast_d(r2), r = r2 * inv_r, inv_r2 = inv_r * inv_r, pre = ONE_4PI_EPS0 * qq;
    const double ar = alpha * r, ex = exp(-ar * ar), ec = erfcx_fast_d(ar) * ex;
    *fs = pre * (ec * inv_r + TWO_OVER_SQRT_PI * alpha * ex) * inv_r2;
    return pre * ec * inv_r;
}
// softcore LJ at three lambda slots sharing (r/sigma)^6; one reciprocal per slot
__device__ __forceinline__ void softcore_lj3_fast_d(double r2, double sig, double eps, const double ls[3], double sc_alpha, double e[3], double fs[3]) {
    if (eps == 0.0 || sig == 0.0) { e[0] = e[1] = e[2] = 0.0; fs[0] = fs[1] = fs[2] = 0.0; return; }
    const double inv_s2 = rcp_fast_d(sig * sig), q2 = r2 * inv_s2, q4 = q2 * q2, q6 = q4 * q2;
    const double g = 24.0 * eps * q4 * inv_s2;
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const double x = rcp_fast_d(sc_alpha * (1.0 - ls[s]) + q6);
        e[s] = ls[s] * 4.0 * eps * x * (x - 1.0);
        fs[s] = ls[s] * g * (2.0 * x - 1.0) * x * x;
    }
}

// the same at one lambda (all three slots carry the same lambda_sterics)
__device__ __forceinline__ void softcore_lj1_fast_d(double r2, double sig, double eps, double ls, double sc_alpha, double* e, double* fs) {
    if (eps == 0.0 || sig == 0.0) { *e = 0.0; *fs = 0.0; return; }
    const double inv_s2 = rcp_fast_d(sig * sig), q2 = r2 * inv_s2, q4 = q2 * q2, q6 = q4 * q2;
    const double g = 24.0 * eps * q4 * inv_s2;
    const double x = rcp_fast_d(sc_alpha * (1.0 - ls) + q6);
    *e = ls * 4.0 * eps * x * (x - 1.0);
    *fs = ls * g * (2.0 * x - 1.0) * x * x;
}

__device__ inline double min_image_d(double d, double L, double invL) { return d - L * rint(d * invL); }

// wave64 reductions
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// the same sum with the four in-row levels on DPP (no LDS round trips): quad xor 1, quad xor 2, row_half_mirror, row_mirror
// leave the 16-lane row sum in every lane of the row; the two cross-row levels go through __shfl_xor.  Fixed order.
template <int CTRL> __device__ __forceinline__ double dpp_add_d(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, true);
    return v + __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v = dpp_add_d<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add_d<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add_d<0x141>(v);   // row_half_mirror
    v = dpp_add_d<0x140>(v);   // row_mirror
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
// fp32 sum over the 64 lanes entirely on DPP, fixed order: quad xor 1, quad xor 2, row_half_mirror, row_mirror leave the sum of a
// 16-lane row in each of its lanes; row_bcast15 adds row 0 into row 1 and row 2 into row 3, row_bcast31 adds row 1 into rows 2, 3:
// lane 63 holds the total (one v_add_f32 with a DPP operand per level; the fp64 version above needs two moves and a 4-cycle add).
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_add_f(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}
__device__ __forceinline__ float wave_sum_dpp_f32(float v) {   // result valid in lane 63; readlane'd into every lane
    v = dpp_add_f<0xB1, 0xF>(v);
    v = dpp_add_f<0x4E, 0xF>(v);
    v = dpp_add_f<0x141, 0xF>(v);
    v = dpp_add_f<0x140, 0xF>(v);
    v = dpp_add_f<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3
    v = dpp_add_f<0x143, 0xC>(v);   // row_bcast31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ inline double seg_sum(double v, int width) {  // sum over aligned groups of `width` lanes (power of two)
    for (int off = width >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ inline float seg_sum(float v, int width) {
    for (int off = width >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---- phase stamps of development builds (-DBLUES_STAMP): the clock of thread 0 of block 0 at phase boundaries, printed by the host
#ifdef BLUES_STAMP
__device__ long long g_nb_stamps[64];   // dev builds: s_memtime of thread 0 of block 0 at phase boundaries ([0..15] list build, [16..31] force kernel, [32..47] dense alchemical kernel)
#define NB_STAMP(cond, i) do { if (cond) g_nb_stamps[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define NB_STAMP(cond, i) do { } while (0)
#endif
