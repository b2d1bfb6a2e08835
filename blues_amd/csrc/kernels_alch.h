// kernels_alch.h -- the lambda-dependent pairs (K2), entirely in fp64.
//
// Replaces what openmmtools' AbsoluteAlchemicalFactory adds to the System
// (reference blues/simulation.py:300-316; forms in SURVEY.md Appendix B) and the two
// total-energy evaluations of every H step (reference blues/integrators.py:219-230):
// instead of differencing two O(1e5 kJ/mol) totals, the protocol-work increment
// Enew - Eold is formed directly from the few thousand pairs that depend on lambda.
//
// One pass at fixed positions serves up to three consecutive lambda indices (slots):
//   electrostatics is linear in lambda_electrostatics, so the Coulomb sum C is computed once
//   and E_elec(slot) = le[slot] * C;  softcore sterics S(slot) is recomputed per slot from the
//   shared (r/sigma)^6.  Forces are produced for the slots in `slot_mask`.
//
// Thread layout: lane = (j-slot, a) with the alchemical atom index a fastest (PA = padded
// power of two), so the force on environment atom j is a segmented shuffle reduction over
// PA lanes and the force on alchemical atom a a strided one; block partials are written to
// slabs and summed in fixed order by the integrator kernel (no atomics, reproducible).
#pragma once
#include "device_common.h"

#define K2_NE 6  // energy partials per block: C_scaled, S0, S1, S2, const_lj, const_coul
#define K2_STAGE 256  // j entries an env block stages in LDS (>= (256 / PA) * jiter)
#define K2_NP 9  // K2_NE + the block's total slot-0 force on the alchemical atoms (x,y,z), for the momentum bookkeeping
// env blocks needed for `count` list entries when a block walks `jiter` groups of 256/PA j's (see alchemical_body): 1 for a
// lone replica (shortest dependent chain per block), 4 in a large batch (reductions amortised, 4x fewer partials)
__host__ __device__ inline int k2_env_blocks(int count, int PA, int jiter) { const int jpb = (256 / PA) * jiter; return (count + jpb - 1) / jpb; }

// packed per-entry records so the pair thread needs two dependent loads (record -> position) instead of four
// `eps` of both records is sqrt(epsilon): the Lorentz-Berthelot well depth of a pair is then one multiplication instead of an
// fp64 square root per pair (about 25 of the ~350 instructions of a pair)
struct AlchJRec { int jo, jsrt; double sig, eps, q; };   // one per entry of the alchemical tile's j-list (written at list build); jsrt bit 30 = j is mobile
struct AlchARec { int ao, asrt, has_env_excl, pad; double sig, eps, q; };  // one per alchemical atom (static)

struct AlchArgs {
    const AlchJRec* jrec; const AlchARec* arec;
    int n, n_alch, PA, jcap, nblocks_env, jiter;
    const int* alch_orig;       // [n_alch]
    const int* jlist;           // j-list of the alchemical tile (sorted indices)
    const int* jcount;          // -> its count
    const int* orig_of_sorted;
    const int* sorted_of_orig;
    const double* x[3];
    const double* charge; const double* sigma; const double* eps;  // caller order
    const int* ex_start; const int* ex_idx;                         // sorted index space
    // alchemical exceptions as rows per alchemical atom: partner (caller index) and parameters
    const int* exc_start;       // [n_alch+1]
    const int* exc_partner;
    const int* exc_owner;       // caller index of the alchemical atom owning the entry
    const int* exc_is_env;      // 1: the partner is a non-alchemical atom (handled by the env thread of that pair)
    const double* exc_params;   // [3*] chargeProd, sigma, epsilon
    Box3 box;
    double rc2, alpha, sc_alpha;
    int pme, annih_elec, annih_ster, slot_mask;
    int check_env_excl;  // 0 when no alchemical atom has an excluded non-alchemical partner (free ligand)
    double ls[3], le[3];
    double* fJ;         // [3 slots][3][n] force on environment atoms by sorted index
    double* self_part;  // [nblocks][3 slots][3][PA]
    double* e_part;     // [nblocks][K2_NP]
    const DevCtrl* ctrl; // non-null in graph replays: lambda slots come from the device tables
    int* flags_overflow; // -> DevFlags::list_overflow (the dense form: more mobile list entries than it has accumulators for)
    const AtomF* img;    // the fixed-point image (mixed precision; the fp32 dense form stages its records from it), or null
    float fscale[3];     // box edge / 2^32
};

__device__ inline bool excluded_sorted(const int* ex_start, const int* ex_idx, int a_sorted, int j_sorted) {
    bool ex = false;
    for (int e = ex_start[a_sorted]; e < ex_start[a_sorted + 1]; e++) ex |= (ex_idx[e] == j_sorted);
    return ex;
}

// FAST: the mixed-precision mode's math (device_common.h, fast fp64 forms); false = libm forms, the reference-grade path
// MASK: the force slot mask known at compile time (5 and 2 are what "H V R O R V H" asks for, see force_pass), so that the
// force arithmetic, accumulators and reductions of the other slots are not even compiled in; -1 = use A.slot_mask
// Returns false when the (env) block had no list entries left, i.e. every later block is empty too.
template <bool FAST, int MASK = -1>
__device__ __forceinline__ bool alchemical_body(AlchArgs& A, const int block_id) {
    auto slot_on = [&](int s) -> bool { return MASK >= 0 ? ((MASK >> s) & 1) != 0 : ((A.slot_mask >> s) & 1) != 0; };
    if (A.ctrl) {
        const int L = A.ctrl->L0 + 2 * A.ctrl->kpass;
#pragma unroll
        for (int s = 0; s < 3; s++) { const int Ls = min(L + s, A.ctrl->n_lambda); A.ls[s] = A.ctrl->tab_ls[Ls]; A.le[s] = A.ctrl->tab_le[Ls]; }
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int PA = A.PA;
    const int a = tid & (PA - 1);
    __shared__ double s_self[4][9][64];
    __shared__ double s_e[4][K2_NE];
    __shared__ double s_exc[9][256];   // self block: exception rows; env blocks: the block's j records and positions (K2_STAGE entries)
    // Force arithmetic of the env blocks: fp32 in the mixed mode (FAST) -- the force scale of a pair is formed in fp64 with its
    // energy and rounded once; its three components, the sum over the alchemical lanes (force on j) and the running sum over
    // this thread's j's (force on a) are fp32; block partials are fp64 again.  The ENERGIES -- the protocol work -- stay fp64
    // throughout.  Half the live registers of the loop were these fp64 force values.
    using FT = typename std::conditional<FAST, float, double>::type;
    FT f[3][3];  // [slot][xyz] force on the alchemical atom from this pair
#pragma unroll
    for (int s = 0; s < 3; s++) { f[s][0] = f[s][1] = f[s][2] = (FT)0; }
    double e[K2_NE];
#pragma unroll
    for (int q = 0; q < K2_NE; q++) e[q] = 0.0;

    const bool env_block = block_id < A.nblocks_env;
    const bool no_elec = A.le[0] == 0.0 && A.le[1] == 0.0 && A.le[2] == 0.0, same_ls = A.ls[0] == A.ls[1] && A.ls[1] == A.ls[2];   // (uniform)
    bool wave_hit = true;
    if (env_block) {
        // An env block covers A.jiter groups of 256/PA consecutive j's; a thread keeps its alchemical atom a and walks one
        // j per group.  The force on j is reduced over the PA lanes per group; the force on a and the energies accumulate
        // in registers across the groups and are reduced once per block.
        // (the record's pointers are generic; they all point to global memory: FLAT accesses cost an address pair per lane)
#if defined(__HIP_DEVICE_COMPILE__)
#define G1(T, p) ((__attribute__((address_space(1))) T*)(p))
#else
#define G1(T, p) ((T*)(p))
#endif
        const auto g_fJ = G1(double, A.fJ); const auto g_jrec = G1(const AlchJRec, A.jrec); const auto g_arec = G1(const AlchARec, A.arec);
        const auto g_x0 = G1(const double, A.x[0]); const auto g_x1 = G1(const double, A.x[1]); const auto g_x2 = G1(const double, A.x[2]);
        const int count = *G1(const int, A.jcount);
        const int jpg = 256 / PA, j0 = block_id * jpg * A.jiter;
        if (j0 >= count) return false;  // nothing to do; finalize sums only the used blocks
        AlchARec Ar; Ar.ao = 0; Ar.asrt = 0; Ar.has_env_excl = 0; Ar.sig = Ar.eps = Ar.q = 0.0;
        if (a < A.n_alch) { Ar.ao = g_arec[a].ao; Ar.asrt = g_arec[a].asrt; Ar.has_env_excl = g_arec[a].has_env_excl; Ar.sig = g_arec[a].sig; Ar.eps = g_arec[a].eps; Ar.q = g_arec[a].q; }
        double xa[3] = {0.0, 0.0, 0.0};
        if (a < A.n_alch) { xa[0] = g_x0[Ar.ao]; xa[1] = g_x1[Ar.ao]; xa[2] = g_x2[Ar.ao]; }
        FT fa[3][3];
#pragma unroll
        for (int s = 0; s < 3; s++) { fa[s][0] = fa[s][1] = fa[s][2] = (FT)0; }
        wave_hit = false;
        // The block's j records and their positions are staged in LDS by one round of loads (record -> position: two dependent
        // memory round trips per BLOCK; read per pair they were two per iteration, and with three waves per SIMD the kernel
        // spent 60 % of its wave-cycles waiting on them).  jpg * jiter <= K2_STAGE (the host clamps jiter to PA).
        double* const st_x = &s_exc[0][0];                                   // [3][K2_STAGE]
        double* const st_sig = st_x + 3 * K2_STAGE; double* const st_eps = st_sig + K2_STAGE; double* const st_q = st_eps + K2_STAGE;
        int* const st_jsrt = reinterpret_cast<int*>(st_q + K2_STAGE); int* const st_jo = st_jsrt + K2_STAGE;
        {
            const int nstage = min(jpg * A.jiter, count - j0);
            if (tid < nstage) {
                const int jo = g_jrec[j0 + tid].jo;
                st_jsrt[tid] = g_jrec[j0 + tid].jsrt; st_jo[tid] = jo; st_sig[tid] = g_jrec[j0 + tid].sig; st_eps[tid] = g_jrec[j0 + tid].eps; st_q[tid] = g_jrec[j0 + tid].q;
                st_x[tid] = g_x0[jo]; st_x[K2_STAGE + tid] = g_x1[jo]; st_x[2 * K2_STAGE + tid] = g_x2[jo];
            }
            __syncthreads();
        }
        for (int it = 0; it < A.jiter; it++) {
            const int sl = it * jpg + tid / PA;
            const int js = j0 + sl;
            if (j0 + it * jpg >= count) break;   // block-uniform
            const bool valid = js < count && a < A.n_alch;
            int jsrt = -1;
            bool hit = false;  // this lane holds a pair term
            bool j_mobile = false;
#pragma unroll
            for (int s = 0; s < 3; s++) { f[s][0] = f[s][1] = f[s][2] = (FT)0; }
            if (valid) {
                AlchJRec J; J.jsrt = st_jsrt[sl]; J.jo = st_jo[sl]; J.sig = st_sig[sl]; J.eps = st_eps[sl]; J.q = st_q[sl];
                jsrt = J.jsrt & 0x3fffffff; j_mobile = (J.jsrt >> 30) & 1;
                const int jo = J.jo;
                double d[3];
                for (int k = 0; k < 3; k++) d[k] = min_image_d(xa[k] - st_x[k * K2_STAGE + sl], A.box.L[k], A.box.invL[k]);
                const double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                const bool excl = Ar.has_env_excl && excluded_sorted(A.ex_start, A.ex_idx, Ar.asrt, jsrt);
                if (!excl && r2 < A.rc2) {
                    const double sig = 0.5 * (Ar.sig + J.sig), eps = Ar.eps * J.eps;   // (both are square roots)
                    const double qq = Ar.q * J.q;
                    double fc;
                    hit = true;
                    if (FAST && A.pme) {
                        // Two things the schedule makes common (reference blues/simulation.py:654-659: sterics move only for
                        // 0.2 < lambda < 0.8, electrostatics only outside) are checked on the actual slot values, not assumed:
                        // with lambda_electrostatics = 0 in every slot the Coulomb sum multiplies zeros (energy le * C, force
                        // le * fc) and is skipped; with one lambda_sterics in all slots the softcore term is evaluated once.
                        fc = 0.0;
                        if (!no_elec) e[0] += coulomb_fast_d(r2, qq, A.alpha, &fc);
                        double es[3], fs3[3];
                        if (same_ls) { softcore_lj1_fast_d(r2, sig, eps, A.ls[0], A.sc_alpha, &es[0], &fs3[0]); es[1] = es[2] = es[0]; fs3[1] = fs3[2] = fs3[0]; }
                        else softcore_lj3_fast_d(r2, sig, eps, A.ls, A.sc_alpha, es, fs3);
#pragma unroll
                        for (int s = 0; s < 3; s++) {
                            e[1 + s] += es[s];
                            if (MASK >= 0 && !((MASK >> s) & 1)) continue;
                            const FT ft = (FT)(fs3[s] + A.le[s] * fc);
                            f[s][0] = ft * (FT)d[0]; f[s][1] = ft * (FT)d[1]; f[s][2] = ft * (FT)d[2];
                        }
                    } else {
                        e[0] += coulomb_d(r2, qq, A.alpha, A.pme != 0, &fc);
#pragma unroll
                        for (int s = 0; s < 3; s++) {
                            double fs;
                            e[1 + s] += softcore_lj_d(r2, sig, eps, A.ls[s], A.sc_alpha, &fs);
                            const double ft = fs + A.le[s] * fc;
                            f[s][0] = (FT)(ft * d[0]); f[s][1] = (FT)(ft * d[1]); f[s][2] = (FT)(ft * d[2]);
                        }
                    }
                } else if (excl) {
                    // excluded pair: if it is a 1-4 exception between this alchemical atom and an environment atom it is
                    // evaluated here with its own parameters (softcore LJ scaled by lambda_sterics, bare Coulomb scaled by
                    // lambda_electrostatics, no cutoff -- SURVEY.md Appendix B), so both atoms get their force through the
                    // same reductions as a regular pair
                    for (int q = A.exc_start[a]; q < A.exc_start[a + 1]; q++) {
                        if (A.exc_partner[q] != jo) continue;
                        hit = true;
                        const double qq = A.exc_params[3 * q], sig = A.exc_params[3 * q + 1], eps = A.exc_params[3 * q + 2];
                        double fc;
                        e[0] += coulomb_d(r2, qq, 0.0, false, &fc);
#pragma unroll
                        for (int s = 0; s < 3; s++) {
                            double fs;
                            e[1 + s] += softcore_lj_d(r2, sig, eps, A.ls[s], A.sc_alpha, &fs);
                            const double ft = fs + A.le[s] * fc;
                            f[s][0] = (FT)(ft * d[0]); f[s][1] = (FT)(ft * d[1]); f[s][2] = (FT)(ft * d[2]);
                        }
                    }
                }
            }
            // A wave covers 64/PA consecutive (Hilbert-sorted) j's per group; part of the list is out of range of every
            // alchemical atom at any one time: such a group publishes zeros and skips its reductions.
            const bool group_hit = __ballot(hit) != 0ull;
            wave_hit |= group_hit;
            // force on environment atom j: minus the sum over the PA alchemical lanes -- wanted only for MOBILE j (frozen
            // atoms are never kicked; most of the list is frozen environment, so most groups skip this reduction too)
            const bool group_fj = __ballot(hit && j_mobile) != 0ull;
#pragma unroll
            for (int s = 0; s < 3; s++) {
                if (!slot_on(s)) continue;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const FT fj = group_fj ? seg_sum(f[s][k], PA) : (FT)0;
                    if (a == 0 && j_mobile) g_fJ[(size_t)(s * 3 + k) * A.n + jsrt] = -(double)fj;
                    fa[s][k] += f[s][k];
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 3; s++) { f[s][0] = fa[s][0]; f[s][1] = fa[s][1]; f[s][2] = fa[s][2]; }
#undef G1
    } else {
        // ---- alchemical x alchemical pairs: thread (a2, b) with b fastest
        const int rows_per_iter = 256 / PA;
        for (int a0 = 0; a0 < A.n_alch; a0 += rows_per_iter) {
            const int a2 = a0 + tid / PA, b = a;
            double g[3][3];
#pragma unroll
            for (int s = 0; s < 3; s++) { g[s][0] = g[s][1] = g[s][2] = 0.0; }
            if (a2 < A.n_alch && b < A.n_alch && a2 != b) {
                const int ao = A.alch_orig[a2], bo = A.alch_orig[b];
                if (!excluded_sorted(A.ex_start, A.ex_idx, A.sorted_of_orig[ao], A.sorted_of_orig[bo])) {
                    double d[3];
                    for (int k = 0; k < 3; k++) d[k] = min_image_d(A.x[k][ao] - A.x[k][bo], A.box.L[k], A.box.invL[k]);
                    const double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                    if (!A.pme || r2 < A.rc2) {
                        const double sig = 0.5 * (A.sigma[ao] + A.sigma[bo]), eps = sqrt(A.eps[ao] * A.eps[bo]);
                        double fc, fl = 0.0;
                        const double ec = coulomb_d(r2, A.charge[ao] * A.charge[bo], A.alpha, A.pme != 0, &fc);
                        if (A.annih_elec) e[0] += 0.5 * ec; else e[5] += 0.5 * ec;
                        if (!A.annih_ster) e[4] += 0.5 * plain_lj_d(r2, sig, eps, &fl);
#pragma unroll
                        for (int s = 0; s < 3; s++) {
                            double fs = fl;
                            if (A.annih_ster) e[1 + s] += 0.5 * softcore_lj_d(r2, sig, eps, A.ls[s], A.sc_alpha, &fs);
                            const double ft = fs + (A.annih_elec ? A.le[s] : 1.0) * fc;
                            g[s][0] = ft * d[0]; g[s][1] = ft * d[1]; g[s][2] = ft * d[2];
                        }
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < 3; s++)
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const double v = seg_sum(g[s][k], PA);
                    // lane b==0 of row a2 owns the row sum; park it in the lane that maps to atom a2 in the final reduce
                    if (b == 0 && a2 < A.n_alch) s_self[0][s * 3 + k][a2] = v;
                }
        }
        __syncthreads();
        // ---- exceptions that involve alchemical atoms: one thread per (row, entry), rows summed from LDS in order
        const int n_exc_ent = A.exc_start[A.n_alch];
        for (int e0 = 0; e0 < n_exc_ent; e0 += 256) {
            const int q = e0 + tid;
            double h[3][3];
#pragma unroll
            for (int s = 0; s < 3; s++) { h[s][0] = h[s][1] = h[s][2] = 0.0; }
            if (q < n_exc_ent && !A.exc_is_env[q]) {
                const int ao = A.exc_owner[q], po = A.exc_partner[q];
                const double qq = A.exc_params[3 * q], sig = A.exc_params[3 * q + 1], eps = A.exc_params[3 * q + 2];
                double d[3];
                for (int k = 0; k < 3; k++) d[k] = min_image_d(A.x[k][ao] - A.x[k][po], A.box.L[k], A.box.invL[k]);
                const double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                double fc, fl = 0.0;
                const double ec = coulomb_d(r2, qq, 0.0, false, &fc);
                if (A.annih_elec) e[0] += 0.5 * ec; else e[5] += 0.5 * ec;
                if (!A.annih_ster) e[4] += 0.5 * plain_lj_d(r2, sig, eps, &fl);
#pragma unroll
                for (int s = 0; s < 3; s++) {
                    double fs = fl;
                    if (A.annih_ster) e[1 + s] += 0.5 * softcore_lj_d(r2, sig, eps, A.ls[s], A.sc_alpha, &fs);
                    const double ft = fs + (A.annih_elec ? A.le[s] : 1.0) * fc;
                    h[s][0] = ft * d[0]; h[s][1] = ft * d[1]; h[s][2] = ft * d[2];
                }
            }
#pragma unroll
            for (int s = 0; s < 3; s++)
#pragma unroll
                for (int k = 0; k < 3; k++) s_exc[s * 3 + k][tid] = h[s][k];
            __syncthreads();
            if (tid < A.n_alch) {
                const int lo = max(A.exc_start[tid], e0), hi = min(A.exc_start[tid + 1], e0 + 256);
                for (int u = lo; u < hi; u++)
#pragma unroll
                    for (int qq2 = 0; qq2 < 9; qq2++) s_self[0][qq2][tid] += s_exc[qq2][u - e0];
            }
            __syncthreads();
        }
        __syncthreads();
        if (tid < PA) {
#pragma unroll
            for (int q = 0; q < 9; q++)
                A.self_part[((size_t)block_id * 9 + q) * PA + tid] = (tid < A.n_alch) ? s_self[0][q][tid] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < K2_NE; q++) e[q] = wave_sum(e[q]);
        if (lane == 0) for (int q = 0; q < K2_NE; q++) s_e[wv][q] = e[q];
        __syncthreads();
        if (tid < K2_NE) A.e_part[(size_t)block_id * K2_NP + tid] = s_e[0][tid] + s_e[1][tid] + s_e[2][tid] + s_e[3][tid];
        if (wv == 0) for (int k = 0; k < 3; k++) { const double t = wave_sum(lane < A.n_alch ? s_self[0][k][lane] : 0.0); if (lane == 0) A.e_part[(size_t)block_id * K2_NP + K2_NE + k] = t; }
        return true;
    }

    // ---- env blocks: force on alchemical atom a = sum over the j's of this block
    if (wave_hit) {
#pragma unroll
        for (int s = 0; s < 3; s++)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                FT v = (FT)0;
                if (slot_on(s)) {   // a slot whose force nobody applies is not reduced (its energy still is)
                    v = f[s][k];
                    for (int off = PA; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
                }
                if (lane < PA) s_self[wv][s * 3 + k][lane] = (double)v;
            }
#pragma unroll
        for (int q = 0; q < K2_NE; q++) e[q] = wave_sum(e[q]);
        if (lane == 0) for (int q = 0; q < K2_NE; q++) s_e[wv][q] = e[q];
    } else {
#pragma unroll
        for (int q = 0; q < 9; q++) if (lane < PA) s_self[wv][q][lane] = 0.0;
        if (lane == 0) for (int q = 0; q < K2_NE; q++) s_e[wv][q] = 0.0;
    }
    __syncthreads();
    if (tid < PA) {
#pragma unroll
        for (int q = 0; q < 9; q++)
            A.self_part[((size_t)block_id * 9 + q) * PA + tid] = s_self[0][q][tid] + s_self[1][q][tid] + s_self[2][q][tid] + s_self[3][q][tid];
    }
    if (tid < K2_NE) A.e_part[(size_t)block_id * K2_NP + tid] = s_e[0][tid] + s_e[1][tid] + s_e[2][tid] + s_e[3][tid];
    if (wv == 0) for (int k = 0; k < 3; k++) {
        const double v = lane < PA ? s_self[0][k][lane] + s_self[1][k][lane] + s_self[2][k][lane] + s_self[3][k][lane] : 0.0;
        const double t = wave_sum(v);
        if (lane == 0) A.e_part[(size_t)block_id * K2_NP + K2_NE + k] = t;
    }
    return true;
}

// The grid holds `nphys` env blocks (+ the alchemical x alchemical block, last): physical block p walks the logical blocks
// p, p + nphys, ... that still have list entries.  The logical block owns the partial slabs, so the sums finalize forms do
// not depend on nphys.  (Sized for the list's CAPACITY, 4 of 5 blocks of a launch were empty and each still held an
// occupancy slot for three dependent loads.)
#define K2_PHYS 24
template <bool FAST, int MASK = -1>
__device__ __forceinline__ void alchemical_blocks(AlchArgs& A, const int p, const int nphys) {
    if (p >= nphys) { alchemical_body<FAST, MASK>(A, A.nblocks_env); return; }
    for (int lb = p; lb < A.nblocks_env; lb += nphys) {
        if (!alchemical_body<FAST, MASK>(A, lb)) return;
        __syncthreads();   // the next logical block reuses the LDS staging and partial arrays
    }
}

// ---- dense form of the env pairs (large batches, mixed precision): ONE workgroup per chain.
// The lane = (alchemical atom, list entry) layout above spends a lane on every combination and finds a pair inside the cutoff
// in one of six; a wave runs the ~350 fp64 instructions of a pair whenever one of its 64 lanes does (45 % of the time at R =
// 512), and 12,800 workgroups each pay the same prologue of dependent loads (round 3 floor experiments: 60 us of the 161 us
// were prologue + epilogue, 80 us pair arithmetic at 35 % lane use).  Here the chain's list is staged once (fp64 positions in
// LDS), a cheap fp32 sweep lists the pairs within the cutoff IN ORDER (a wave per alchemical atom, list index ascending), and
// the fp64 pair arithmetic runs over those lists with every lane busy.  Forces are accumulated as 64-bit fixed point in LDS (integer addition is associative: any order gives the same
// bits, so a chain's result does not depend on what shares the launch); energies stay fp64 per thread and are reduced in a
// fixed order.  Output: the same partial slabs as alchemical_body, as logical block 0 (the host sets jiter so that
// k2_env_blocks() = 1); the alchemical x alchemical block is still alchemical_body's.
// Preconditions (host: k2_dense): no alchemical atom has an excluded environment partner; <= 16 alchemical atoms; at most
// K2D_MOB mobile atoms; every box edge > 2 (cutoff + outer margin + 2 x extent of the alchemical group).
#define K2D_THREADS 512
#define K2D_WAVES (K2D_THREADS / 64)
#define K2D_JC 2560        // list entries staged at a time (an alchemical tile's list is ~2,400 entries; longer lists take more rounds)
#define K2D_AG 8           // alchemical atoms per marking pass (the pair list holds K2D_AG * K2D_JC entries)
#define K2D_MOB 288        // mobile entries per round that have a force accumulator (with the fp32 offsets of the marking pass the record fills the 160 KB of LDS)
#define K2D_FIX 1048576.0  // 2^20: the accumulators resolve 1e-6 kJ/mol/nm and hold +-8.8e12
struct K2DLds {
    double x[3][K2D_JC];
    float rel[3][K2D_JC];   // the entry relative to the first alchemical atom (minimum image, fp32): what the marking pass measures from
    unsigned short pairs[K2D_AG][K2D_JC];   // one segment per alchemical atom of the pass, list index ascending
    unsigned short mslot[K2D_JC];
    unsigned long long fj[9][K2D_MOB];
    unsigned long long fa[9][16];
    double xa[3][16], asig[16], aeps[16], aq[16];
    float aoff[3][16];
    int mcnt[K2D_JC / 64];
    int pcnt[K2D_AG];
    double e[K2D_WAVES][K2_NE];
};

static_assert(sizeof(K2DLds) <= 160 * 1024, "the dense alchemical kernel's record must fit the LDS of a CU");

template <int MASK>
__device__ __forceinline__ void alchemical_dense_body(AlchArgs& A) {
    auto slot_on = [&](int s) -> bool { return MASK >= 0 ? ((MASK >> s) & 1) != 0 : ((A.slot_mask >> s) & 1) != 0; };
    if (A.ctrl) {
        const int L = A.ctrl->L0 + 2 * A.ctrl->kpass;
#pragma unroll
        for (int s = 0; s < 3; s++) { const int Ls = min(L + s, A.ctrl->n_lambda); A.ls[s] = A.ctrl->tab_ls[Ls]; A.le[s] = A.ctrl->tab_le[Ls]; }
    }
    extern __shared__ __align__(16) unsigned char k2d_smem[];
    K2DLds& S = *reinterpret_cast<K2DLds*>(k2d_smem);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    NB_STAMP(blockIdx.x == 0 && threadIdx.x == 0, 32);
    const int count = *A.jcount;
    if (count <= 0) return;   // (finalize reads no env slab then)
    const bool no_elec = A.le[0] == 0.0 && A.le[1] == 0.0 && A.le[2] == 0.0, same_ls = A.ls[0] == A.ls[1] && A.ls[1] == A.ls[2];   // (uniform)
    double e[K2_NE];
#pragma unroll
    for (int q = 0; q < K2_NE; q++) e[q] = 0.0;
    for (int w = tid; w < 9 * 16; w += K2D_THREADS) S.fa[w / 16][w % 16] = 0ull;
    if (tid < 16) {
        const bool on = tid < A.n_alch;
        const int ao = on ? A.arec[tid].ao : 0;
        S.xa[0][tid] = on ? A.x[0][ao] : 0.0; S.xa[1][tid] = on ? A.x[1][ao] : 0.0; S.xa[2][tid] = on ? A.x[2][ao] : 0.0;
        S.asig[tid] = on ? A.arec[tid].sig : 0.0; S.aeps[tid] = on ? A.arec[tid].eps : 0.0; S.aq[tid] = on ? A.arec[tid].q : 0.0;
    }
    __syncthreads();
    if (tid < 16) {   // the alchemical atoms relative to the first of them (fp32: the marking pass)
#pragma unroll
        for (int k = 0; k < 3; k++) S.aoff[k][tid] = (float)min_image_d(S.xa[k][tid] - S.xa[k][0], A.box.L[k], A.box.invL[k]);
    }
    const float rc2m = (float)A.rc2 * 1.0001f + 1e-5f;   // the marking pass keeps a margin; the pair pass tests r < cutoff itself, in fp64
    constexpr int NIT = K2D_JC / K2D_THREADS;
    for (int j0 = 0; j0 < count; j0 += K2D_JC) {
        const int nst = min(K2D_JC, count - j0), nchunk = (nst + 63) >> 6;
        // ---- stage the round: positions (record -> position: the two dependent loads, all entries of a level in flight together),
        // mobile entries numbered in list order
        for (int w = tid; w < 9 * K2D_MOB; w += K2D_THREADS) S.fj[w / K2D_MOB][w % K2D_MOB] = 0ull;
        int jo[NIT], js[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int kk = min(it * K2D_THREADS + tid, nst - 1);
            jo[it] = A.jrec[j0 + kk].jo; js[it] = A.jrec[j0 + kk].jsrt;
        }
        double px[NIT][3];
#pragma unroll
        for (int it = 0; it < NIT; it++) { px[it][0] = A.x[0][jo[it]]; px[it][1] = A.x[1][jo[it]]; px[it][2] = A.x[2][jo[it]]; }
        bool mob[NIT]; int mrank[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int k = it * K2D_THREADS + tid, ch = k >> 6;
            S.x[0][k] = px[it][0]; S.x[1][k] = px[it][1]; S.x[2][k] = px[it][2];   // (entries past the end repeat the last one)
            // (formed once per entry: the marking pass below ran these three fp64 minimum images for every (alchemical atom, entry)
            // combination -- 15 x the work, and two fifths of the kernel's instructions)
#pragma unroll
            for (int c3 = 0; c3 < 3; c3++) S.rel[c3][k] = (float)min_image_d(px[it][c3] - S.xa[c3][0], A.box.L[c3], A.box.invL[c3]);
            mob[it] = k < nst && ((js[it] >> 30) & 1);
            const unsigned long long b = __ballot(mob[it]);
            mrank[it] = __popcll(b & ((1ull << lane) - 1ull));
            if (lane == 0) S.mcnt[ch] = __popcll(b);
        }
        __syncthreads();
        {
            // exclusive scan of the chunks' mobile counts: every wave forms it for itself (lane = chunk; K2D_JC / 64 <= 64)
            int c = lane < K2D_JC / 64 ? S.mcnt[lane] : 0, inc = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
            const int exc = inc - c;
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int k = it * K2D_THREADS + tid, ch = k >> 6;
                const int ms = __shfl(exc, ch, 64) + mrank[it];
                S.mslot[k] = (mob[it] && ms < K2D_MOB) ? (unsigned short)ms : (unsigned short)0xffff;
            }
        }
        NB_STAMP(blockIdx.x == 0 && threadIdx.x == 0, 33);
        for (int a0 = 0; a0 < A.n_alch; a0 += K2D_AG) {
            const int na = min(K2D_AG, A.n_alch - a0);
            __syncthreads();
            NB_STAMP(blockIdx.x == 0 && threadIdx.x == 0, 34 + 3 * (a0 / K2D_AG));
            // ---- mark and list in one sweep: wave al owns alchemical atom a0 + al and walks the round's chunks in order (fp32
            // distances relative to the first alchemical atom; a running offset keeps its segment of the pair list ascending)
            if (wv < na) {
                const float ox = S.aoff[0][a0 + wv], oy = S.aoff[1][a0 + wv], oz = S.aoff[2][a0 + wv];
                int pos = 0;
                auto test = [&](int ch) -> bool {
                    const int k = ch * 64 + lane;
                    const float dx = S.rel[0][k] - ox, dy = S.rel[1][k] - oy, dz = S.rel[2][k] - oz;
                    return k < nst && fmaf(dz, dz, fmaf(dy, dy, dx * dx)) < rc2m;
                };
                auto emit = [&](int ch, bool hit) {
                    const unsigned long long b = __ballot(hit);
                    if (hit) S.pairs[wv][pos + __popcll(b & ((1ull << lane) - 1ull))] = (unsigned short)(ch * 64 + lane);
                    pos += __popcll(b);
                };
                int ch = 0;
                for (; ch + 1 < nchunk; ch += 2) {   // two chunks per trip: their LDS reads and fp64 arithmetic are independent
                    const bool h0 = test(ch), h1 = test(ch + 1);
                    emit(ch, h0); emit(ch + 1, h1);
                }
                if (ch < nchunk) emit(ch, test(ch));
                if (lane == 0) S.pcnt[wv] = pos;
            }
            __syncthreads();
            NB_STAMP(blockIdx.x == 0 && threadIdx.x == 0, 35 + 3 * (a0 / K2D_AG));
            // ---- the pairs, every lane busy: work items = (alchemical atom of the pass, 64 consecutive pairs of its segment)
            int item_al[K2D_AG + 1];   // first item of each segment (uniform)
            item_al[0] = 0;
#pragma unroll
            for (int al = 0; al < K2D_AG; al++) item_al[al + 1] = item_al[al] + (al < na ? (S.pcnt[al] + 63) >> 6 : 0);
            const int nitems = item_al[K2D_AG];
            // the first item's record (the next one's is requested before the current pair is computed: a global gather per pair)
            auto locate = [&](int item, int& al, int& p) {
                al = 0;
#pragma unroll
                for (int q = 1; q < K2D_AG; q++) if (item >= item_al[q]) al = q;
                p = (item - item_al[al]) * 64 + lane;
            };
            double njsig = 0.0, njeps = 0.0, njq = 0.0; int nk = 0, nal = 0; bool nvalid = false;
            auto request = [&](int item) {
                nvalid = false; nk = 0; nal = 0;
                if (item < nitems) {
                    int p; locate(item, nal, p);
                    nvalid = p < S.pcnt[nal];
                    nk = nvalid ? (int)S.pairs[nal][p] : 0;
                    const AlchJRec* jr = A.jrec + j0 + nk;
                    njsig = jr->sig; njeps = jr->eps; njq = jr->q;
                }
            };
            // a wave takes a CONTIGUOUS range of items: nearly all of them belong to one or two alchemical atoms, whose force is summed
            // per lane across the items and over the wave once per atom (round 3 reduced nine values over the wave after EVERY item:
            // a quarter of the loop's instructions).  The sums land in fixed-point accumulators: any order gives the same bits.
            const int per_wave = (nitems + K2D_WAVES - 1) / K2D_WAVES, item_lo = wv * per_wave, item_hi = min(nitems, item_lo + per_wave);
            float facc[3][3];
#pragma unroll
            for (int s = 0; s < 3; s++) { facc[s][0] = facc[s][1] = facc[s][2] = 0.0f; }
            int acc_a = -1;
            auto flush = [&]() {
                if (acc_a < 0) return;
#pragma unroll
                for (int s = 0; s < 3; s++) {
                    if (!slot_on(s)) continue;
#pragma unroll
                    for (int c3 = 0; c3 < 3; c3++) {
                        const float v = wave_sum_dpp_f32(facc[s][c3]);
                        if (lane == 0) atomicAdd(&S.fa[s * 3 + c3][acc_a], (unsigned long long)__double2ll_rn((double)v * K2D_FIX));
                        facc[s][c3] = 0.0f;
                    }
                }
            };
            request(item_lo);
            for (int item = item_lo; item < item_hi; item++) {
                const double jsig = njsig, jeps = njeps, jq = njq; const int k = nk, al = nal; const bool valid = nvalid;
                request(item + 1 < item_hi ? item + 1 : nitems);
                const int a = a0 + al;   // (uniform: an item lies inside one segment)
                if (a != acc_a) { flush(); acc_a = a; }
                double d[3];
#pragma unroll
                for (int c3 = 0; c3 < 3; c3++) d[c3] = min_image_d(S.xa[c3][a] - S.x[c3][k], A.box.L[c3], A.box.invL[c3]);
                const double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                const bool hit = valid && r2 < A.rc2;
                float f[3][3];
#pragma unroll
                for (int s = 0; s < 3; s++) { f[s][0] = f[s][1] = f[s][2] = 0.0f; }
                if (hit) {
                    const double sig = 0.5 * (S.asig[a] + jsig), eps = S.aeps[a] * jeps, qq = S.aq[a] * jq;   // (eps: both are square roots)
                    double fc = 0.0;
                    if (!no_elec) e[0] += coulomb_fast_d(r2, qq, A.alpha, &fc);
                    double es[3], fs3[3];
                    if (same_ls) { softcore_lj1_fast_d(r2, sig, eps, A.ls[0], A.sc_alpha, &es[0], &fs3[0]); es[1] = es[2] = es[0]; fs3[1] = fs3[2] = fs3[0]; }
                    else softcore_lj3_fast_d(r2, sig, eps, A.ls, A.sc_alpha, es, fs3);
#pragma unroll
                    for (int s = 0; s < 3; s++) {
                        e[1 + s] += es[s];
                        if (!slot_on(s)) continue;
                        const float ft = (float)(fs3[s] + A.le[s] * fc);
                        f[s][0] = ft * (float)d[0]; f[s][1] = ft * (float)d[1]; f[s][2] = ft * (float)d[2];
                    }
                }
                // force on the item's alchemical atom: into the lane's running sums (flushed when the atom changes)
#pragma unroll
                for (int s = 0; s < 3; s++) {
                    if (!slot_on(s)) continue;
#pragma unroll
                    for (int c3 = 0; c3 < 3; c3++) facc[s][c3] += f[s][c3];
                }
                // force on a mobile environment atom: minus the pair force
                const unsigned ms = hit ? (unsigned)S.mslot[k] : 0xffffu;
                if (ms != 0xffffu) {
#pragma unroll
                    for (int s = 0; s < 3; s++) {
                        if (!slot_on(s)) continue;
#pragma unroll
                        for (int c3 = 0; c3 < 3; c3++) atomicAdd(&S.fj[s * 3 + c3][ms], (unsigned long long)__double2ll_rn(-(double)f[s][c3] * K2D_FIX));
                    }
                }
            }
            flush();
            NB_STAMP(blockIdx.x == 0 && threadIdx.x == 0, 36 + 3 * (a0 / K2D_AG));
        }
        __syncthreads();
        NB_STAMP(blockIdx.x == 0 && threadIdx.x == 0, 40);
        // ---- the round's mobile entries: force by sorted index (every one of them is written: zero if it had no pair)
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int k = it * K2D_THREADS + tid;
            if (k < nst && mob[it]) {
                const unsigned ms = S.mslot[k];
                if (ms == 0xffffu) { if (A.flags_overflow) A.flags_overflow[0] = 1; continue; }
                const int jsrt = js[it] & 0x3fffffff;
#pragma unroll
                for (int s = 0; s < 3; s++) {
                    if (!slot_on(s)) continue;
#pragma unroll
                    for (int c3 = 0; c3 < 3; c3++) A.fJ[(size_t)(s * 3 + c3) * A.n + jsrt] = (double)(long long)S.fj[s * 3 + c3][ms] * (1.0 / K2D_FIX);
                }
            }
        }
        __syncthreads();
    }
    // ---- partial slabs of logical block 0
    if (tid < A.PA) {
#pragma unroll
        for (int q = 0; q < 9; q++) A.self_part[(size_t)q * A.PA + tid] = (tid < A.n_alch && slot_on(q / 3)) ? (double)(long long)S.fa[q][tid] * (1.0 / K2D_FIX) : 0.0;
    }
#pragma unroll
    for (int q = 0; q < K2_NE; q++) e[q] = wave_sum(e[q]);
    if (lane == 0) for (int q = 0; q < K2_NE; q++) S.e[wv][q] = e[q];
    __syncthreads();
    if (tid < K2_NE) { double t = 0.0; for (int w = 0; w < K2D_WAVES; w++) t += S.e[w][tid]; A.e_part[tid] = t; }
    if (tid < 3) { double t = 0.0; for (int a = 0; a < A.n_alch; a++) t += (double)(long long)S.fa[tid][a] * (1.0 / K2D_FIX); A.e_part[K2_NE + tid] = slot_on(0) ? t : 0.0; }
    NB_STAMP(blockIdx.x == 0 && threadIdx.x == 0, 41);
}

// ---- round 6: the dense form in fp32 pair arithmetic (the form the batch decomposition runs; the fp64 body above stays as
// BluesTuning.k2_dense = 2, the reference the new one is tested against).
// What changed and why.  The fp64 body spent 17 k of its ~30 k cycles per marking pass in ~350 fp64 instructions per pair (two Newton-
// refined reciprocals, an rsqrt, a libm exp, a degree-16 erfc fit), ran two marking passes (its pair lists and fp64 positions fill the
// LDS: one workgroup per CU, two waves per SIMD at 187 registers) and staged positions through two dependent gathers of 32 + 24 bytes
// per list entry.  Here:
//  * an entry is staged with ONE 32-byte gather from the fixed-point image the step kernel refreshes (kernels_integrate.h) into the
//    24-byte LDS record the nonbonded kernel uses -- {x, y, z as fractions of the box edge in u32, q sqrt(k_e)} + {sigma / 2, 2 sqrt(eps)};
//    a fixed-point difference IS the minimum image, exact, and its fp32 conversion carries one rounding (6e-8 of the separation);
//  * pair arithmetic in fp32 (~100 instructions: one v_rsq, one v_exp, four v_rcp) with fp64 accumulation of the energies per lane;
//  * THE PROTOCOL WORK keeps its precision because the kernel never differences two rounded energies: for a pair, with
//    x_s = 1 / (A_s + (r / sigma)^6), A_s = alpha (1 - ls_s), the change of the softcore energy between two lambda slots is
//        U_b - U_a = 4 eps [ ls_b (x_b - x_a)(x_a + x_b - 1) + (ls_b - ls_a) x_a (x_a - 1) ],   x_b - x_a = (A_a - A_b) x_a x_b   (exact algebra),
//    formed from the SMALL numbers (A_a - A_b), (ls_b - ls_a) -- its rounding error is ~3e-7 of the difference itself, not of U.  The
//    kernel accumulates S_0, D_01 = sum (U_1 - U_0), D_12 and publishes S_0, S_0 + D_01, S_0 + D_01 + D_12 in fp64: what finalize
//    differences comes back as D to 1e-16 of S.  Electrostatics is linear in lambda_electrostatics: one sum C, the work is (le_b - le_a) C;
//  * a wave owns an alchemical atom: it sweeps the staged list, compacts the entries within the cutoff into a 128-entry ring in LDS and
//    runs the pair arithmetic whenever 64 are waiting -- every lane busy, no pair list for the whole list, no second pass, no barrier
//    between marking and arithmetic; the force on its atom is a per-lane fp32 sum and one DPP reduction (a fixed order: batch = solo
//    stays bitwise); the force on a mobile environment atom goes to 64-bit fixed-point accumulators in LDS as before (integer
//    addition is associative);
//  * 73 KB of LDS and < 128 registers: two workgroups per CU.
#define K2F_THREADS 512
#define K2F_WAVES (K2F_THREADS / 64)
#define K2F_JC 2432        // list entries staged at a time (38 chunks; the alchemical tile's list of the benchmark system is ~2,400 entries)
#define K2F_MOB 288        // mobile entries per round with a force accumulator
#define K2F_RING 128
#define K2F_FIX 8192.0f    // 2^13: the reaction accumulators resolve 1.2e-4 kJ/mol/nm (1e-8 of the forces a clash produces; the fp32 pair force itself carries 3e-7 of its size) and hold +-2.6e5 kJ/mol/nm:
                           // 32-bit, so that two workgroups fit a CU's LDS with room to spare (64-bit accumulators put the record at 80.5 KB: ONE workgroup per CU on the hardware, 96 us per launch)
template <int NS> struct K2FLds {
    uint4 r4[K2F_JC];                       // x, y, z (fixed point), q sqrt(k_e) as float bits
    float2 r2[K2F_JC];                      // sigma / 2, 2 sqrt(eps)
    unsigned short mslot[K2F_JC];
    int fj[3 * NS][K2F_MOB];                // reaction on the mobile entries: 32-bit fixed point (K2F_FIX), see below
    unsigned short ring[K2F_WAVES][K2F_RING];
    int mcnt[(K2F_JC + 63) / 64 + 2];
    unsigned xa[3][16]; float aq[16], ahs[16], ase[16];
    float fa[16][9];
    double e[K2F_WAVES][4];
};
static_assert(sizeof(K2FLds<2>) <= 76 * 1024, "two workgroups of the fp32 dense alchemical kernel per CU, with room for the allocation granule");

template <int MASK>
__device__ __forceinline__ void alchemical_dense32_body(AlchArgs& A) {
    constexpr int NS = MASK == 5 ? 2 : (MASK == 2 ? 1 : 3);                    // force slots this instantiation produces
    auto slot_on = [&](int s) -> bool { return MASK >= 0 ? ((MASK >> s) & 1) != 0 : ((A.slot_mask >> s) & 1) != 0; };
    auto slot_ix = [&](int s) -> int { return MASK == 5 ? (s >> 1) : (MASK == 2 ? 0 : s); };   // where slot s lives among the NS accumulators
    if (A.ctrl) {
        const int L = A.ctrl->L0 + 2 * A.ctrl->kpass;
#pragma unroll
        for (int s = 0; s < 3; s++) { const int Ls = min(L + s, A.ctrl->n_lambda); A.ls[s] = A.ctrl->tab_ls[Ls]; A.le[s] = A.ctrl->tab_le[Ls]; }
    }
    extern __shared__ __align__(16) unsigned char k2d_smem[];
    K2FLds<NS>& S = *reinterpret_cast<K2FLds<NS>*>(k2d_smem);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool stamp_wg = blockIdx.x == (gridDim.x * 5u) / 8u;   // (stamped builds: one workgroup of the launch's third quarter)
    NB_STAMP(stamp_wg && tid == 0, 32);
    const int count = *A.jcount;
    if (count <= 0) return;   // (finalize reads no env slab then)
    const AtomF* __restrict__ img = A.img;
    // uniform lambda arithmetic in fp64, handed to the pair body as floats of SMALL numbers (differences) and of the slots' own values
    const bool no_elec = A.le[0] == 0.0 && A.le[1] == 0.0 && A.le[2] == 0.0, same_ls = A.ls[0] == A.ls[1] && A.ls[1] == A.ls[2];
    const float sA[3] = {(float)(A.sc_alpha * (1.0 - A.ls[0])), (float)(A.sc_alpha * (1.0 - A.ls[1])), (float)(A.sc_alpha * (1.0 - A.ls[2]))};
    const float ls[3] = {(float)A.ls[0], (float)A.ls[1], (float)A.ls[2]}, le[3] = {(float)A.le[0], (float)A.le[1], (float)A.le[2]};
    const float dA01 = (float)(A.sc_alpha * (A.ls[1] - A.ls[0])), dA12 = (float)(A.sc_alpha * (A.ls[2] - A.ls[1]));   // A_a - A_b = alpha (ls_b - ls_a)
    const float dl01 = (float)(A.ls[1] - A.ls[0]), dl12 = (float)(A.ls[2] - A.ls[1]);
    const float rc2 = (float)A.rc2, alpha = (float)A.alpha, c2 = TWO_OVER_SQRT_PI_F * (float)A.alpha;
    const float sc0 = A.fscale[0], sc1 = A.fscale[1], sc2 = A.fscale[2];
    if (tid < 16) {
        const bool on = tid < A.n_alch;
        const AlchARec ar = A.arec[on ? tid : 0];
        const AtomF me = img[ar.asrt];
        S.xa[0][tid] = me.x; S.xa[1][tid] = me.y; S.xa[2][tid] = me.z;
        S.aq[tid] = on ? (float)(ar.q * sqrt(ONE_4PI_EPS0)) : 0.0f; S.ahs[tid] = on ? (float)(0.5 * ar.sig) : 0.0f; S.ase[tid] = on ? (float)(2.0 * ar.eps) : 0.0f;   // (ar.eps is sqrt(epsilon))
    }
    for (int w = tid; w < 16 * 9; w += K2F_THREADS) S.fa[w / 9][w % 9] = 0.0f;
    double eC = 0.0, eS0 = 0.0, eD01 = 0.0, eD12 = 0.0;
    constexpr int NIT = (K2F_JC + K2F_THREADS - 1) / K2F_THREADS;
    for (int j0 = 0; j0 < count; j0 += K2F_JC) {
        const int nst = min(K2F_JC, count - j0), nchunk = (nst + 63) >> 6;
        // ---- stage the round: list entry -> image record (two dependent loads, every entry of a level in flight together); mobile
        // entries numbered in list order
        for (int w = tid; w < 3 * NS * K2F_MOB; w += K2F_THREADS) S.fj[w / K2F_MOB][w % K2F_MOB] = 0;
        int js[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) { const int k = it * K2F_THREADS + tid; js[it] = k < nst ? A.jlist[j0 + k] : -1; }
        bool mob[NIT]; int mrank[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int k = it * K2F_THREADS + tid, ch = k >> 6;
            mob[it] = false;
            if (k < K2F_JC) {
                if (js[it] >= 0) {
                    const uint4 lo = *reinterpret_cast<const uint4*>(&img[js[it]]);            // x, y, z, q
                    const uint4 hi = *(reinterpret_cast<const uint4*>(&img[js[it]]) + 1);      // hs, se, flags, orig
                    S.r4[k] = lo; S.r2[k] = make_float2(__uint_as_float(hi.x), __uint_as_float(hi.y));
                    mob[it] = (hi.z & FLAG_MOBILE) != 0;
                }
            }
            const unsigned long long b = __ballot(mob[it]);
            mrank[it] = __popcll(b & ((1ull << lane) - 1ull));
            if (lane == 0 && k < K2F_JC) S.mcnt[ch] = __popcll(b);
        }
        __syncthreads();
        {
            int c = lane < (K2F_JC + 63) / 64 ? S.mcnt[lane] : 0, inc = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
            const int exc = inc - c;
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int k = it * K2F_THREADS + tid, ch = k >> 6;
                const int ms = __shfl(exc, min(ch, 63), 64) + mrank[it];
                if (k < K2F_JC) S.mslot[k] = (mob[it] && ms < K2F_MOB) ? (unsigned short)ms : (unsigned short)0xffff;
            }
        }
        __syncthreads();
        NB_STAMP(stamp_wg && tid == 0, 33);
        // ---- a wave per alchemical atom: sweep, compact, compute
        for (int a = wv; a < A.n_alch; a += K2F_WAVES) {
            const unsigned xa = S.xa[0][a], ya = S.xa[1][a], za = S.xa[2][a];
            const float qa = S.aq[a], hsa = S.ahs[a], sea = S.ase[a];
            float fa[NS][3];
#pragma unroll
            for (int s = 0; s < NS; s++) { fa[s][0] = fa[s][1] = fa[s][2] = 0.0f; }
            unsigned short* const ring = S.ring[wv];
            int head = 0, fill = 0;   // (uniform)
            auto pairs = [&](int nvalid) {
                const bool valid = lane < nvalid;
                const int k = valid ? (int)ring[(head + lane) & (K2F_RING - 1)] : 0;
                const uint4 p = S.r4[k]; const float2 t = S.r2[k];
                const float dx = (float)(int)(xa - p.x) * sc0, dy = (float)(int)(ya - p.y) * sc1, dz = (float)(int)(za - p.z) * sc2;
                const float r2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                float ft[NS];
#pragma unroll
                for (int s = 0; s < NS; s++) ft[s] = 0.0f;
                if (valid) {
                    const float sig = hsa + t.x, eps4 = sea * t.y, qq = qa * __uint_as_float(p.w);
                    float fc = 0.0f;
                    if (!no_elec && qq != 0.0f) {
                        const float inv_r = __builtin_amdgcn_rsqf(r2), inv_r2 = inv_r * inv_r, arr = alpha * (r2 * inv_r);
                        const float ex = __expf(-arr * arr), ec = erfc_scaled_f(arr) * ex;
                        eC += (double)(qq * ec * inv_r);
                        fc = qq * fmaf(c2, ex, ec * inv_r) * inv_r2;
                    }
                    float fs[3] = {0.0f, 0.0f, 0.0f};
                    if (eps4 != 0.0f) {
                        const float inv_s2 = __builtin_amdgcn_rcpf(fmaxf(sig * sig, 1e-12f)), q2 = r2 * inv_s2, q4 = q2 * q2, q6 = q4 * q2;
                        const float g6 = 6.0f * eps4 * q4 * inv_s2;
                        const float den0 = sA[0] + q6, x0 = __builtin_amdgcn_rcpf(den0);
                        const float u0 = x0 * x0 * (1.0f - den0);          // x (x - 1) without the cancellation at x ~ 1
                        eS0 += (double)(ls[0] * eps4 * u0);
                        fs[0] = ls[0] * g6 * fmaf(2.0f, x0, -1.0f) * x0 * x0;
                        if (same_ls) { fs[1] = fs[2] = fs[0]; }
                        else {
                            const float den1 = sA[1] + q6, x1 = __builtin_amdgcn_rcpf(den1), den2 = sA[2] + q6, x2 = __builtin_amdgcn_rcpf(den2);
                            const float u1 = x1 * x1 * (1.0f - den1);
                            // (A_0 - A_1) = alpha (ls_1 - ls_0) = dA01; x_1 - x_0 = dA01 x_0 x_1
                            const float d01 = dA01 * x0 * x1, d12 = dA12 * x1 * x2;
                            eD01 += (double)(eps4 * fmaf(ls[1] * d01, x0 + x1 - 1.0f, dl01 * u0));
                            eD12 += (double)(eps4 * fmaf(ls[2] * d12, x1 + x2 - 1.0f, dl12 * u1));
                            fs[1] = ls[1] * g6 * fmaf(2.0f, x1, -1.0f) * x1 * x1;
                            fs[2] = ls[2] * g6 * fmaf(2.0f, x2, -1.0f) * x2 * x2;
                        }
                    }
#pragma unroll
                    for (int s = 0; s < 3; s++) if (slot_on(s)) ft[slot_ix(s)] = fmaf(le[s], fc, fs[s]);
                }
                const unsigned ms = valid ? (unsigned)S.mslot[k] : 0xffffu;
#pragma unroll
                for (int s = 0; s < 3; s++) {
                    if (!slot_on(s)) continue;
                    const int q = slot_ix(s);
                    const float fx = ft[q] * dx, fy = ft[q] * dy, fz = ft[q] * dz;
                    fa[q][0] += fx; fa[q][1] += fy; fa[q][2] += fz;
                    if (ms != 0xffffu) {   // force on a mobile environment atom: minus the pair force
                        atomicAdd(&S.fj[3 * q + 0][ms], __float2int_rn(-fx * K2F_FIX));
                        atomicAdd(&S.fj[3 * q + 1][ms], __float2int_rn(-fy * K2F_FIX));
                        atomicAdd(&S.fj[3 * q + 2][ms], __float2int_rn(-fz * K2F_FIX));
                    }
                }
            };
            for (int ch = 0; ch < nchunk; ch++) {
                const int k = ch * 64 + lane;
                const uint4 p = S.r4[min(k, nst - 1)];
                const float dx = (float)(int)(xa - p.x) * sc0, dy = (float)(int)(ya - p.y) * sc1, dz = (float)(int)(za - p.z) * sc2;
                const bool hit = k < nst && fmaf(dz, dz, fmaf(dy, dy, dx * dx)) < rc2;
                const unsigned long long b = __ballot(hit);
                if (hit) ring[(head + fill + __popcll(b & ((1ull << lane) - 1ull))) & (K2F_RING - 1)] = (unsigned short)k;
                fill += __popcll(b);
                if (fill >= 64) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (the ring is written and read by this wave only)
                    pairs(64);
                    head = (head + 64) & (K2F_RING - 1); fill -= 64;
                }
            }
            if (fill > 0) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); pairs(fill); }
            // the force on this wave's atom: one reduction per component, added to what earlier rounds left (this wave is the only writer)
#pragma unroll
            for (int s = 0; s < 3; s++) {
                if (!slot_on(s)) continue;
                const int q = slot_ix(s);
#pragma unroll
                for (int c3 = 0; c3 < 3; c3++) { const float v = wave_sum_dpp_f32(fa[q][c3]); if (lane == 0) S.fa[a][s * 3 + c3] += v; }
            }
            NB_STAMP(stamp_wg && lane == 0 && wv == 0, 34 + min(a / K2F_WAVES, 1));
        }
        NB_STAMP(stamp_wg && lane == 0, 36 + min(wv, 7));
        __syncthreads();
        // ---- the round's mobile entries: force by sorted index (every one of them is written: zero if it had no pair)
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int k = it * K2F_THREADS + tid;
            if (k < nst && mob[it]) {
                const unsigned ms = S.mslot[k];
                if (ms == 0xffffu) { if (A.flags_overflow) A.flags_overflow[0] = 1; continue; }
#pragma unroll
                for (int s = 0; s < 3; s++) {
                    if (!slot_on(s)) continue;
#pragma unroll
                    for (int c3 = 0; c3 < 3; c3++) A.fJ[(size_t)(s * 3 + c3) * A.n + js[it]] = (double)S.fj[3 * slot_ix(s) + c3][ms] * (1.0 / (double)K2F_FIX);
                }
            }
        }
        __syncthreads();
    }
    // ---- partial slabs of logical block 0
    if (tid < A.PA) {
#pragma unroll
        for (int q = 0; q < 9; q++) A.self_part[(size_t)q * A.PA + tid] = (tid < A.n_alch && tid < 16 && slot_on(q / 3)) ? (double)S.fa[tid][q] : 0.0;
    }
    eC = wave_sum(eC); eS0 = wave_sum(eS0); eD01 = wave_sum(eD01); eD12 = wave_sum(eD12);
    if (lane == 0) { S.e[wv][0] = eC; S.e[wv][1] = eS0; S.e[wv][2] = eD01; S.e[wv][3] = eD12; }
    __syncthreads();
    if (tid == 0) {
        double t[4] = {0.0, 0.0, 0.0, 0.0};
        for (int w = 0; w < K2F_WAVES; w++) for (int q = 0; q < 4; q++) t[q] += S.e[w][q];
        A.e_part[0] = t[0]; A.e_part[1] = t[1]; A.e_part[2] = t[1] + t[2]; A.e_part[3] = t[1] + t[2] + t[3];
        A.e_part[4] = 0.0; A.e_part[5] = 0.0;
    }
    if (tid < 3) { double t = 0.0; for (int a = 0; a < A.n_alch && a < 16; a++) t += (double)S.fa[a][tid]; A.e_part[K2_NE + tid] = slot_on(0) ? t : 0.0; }
    NB_STAMP(stamp_wg && tid == 0, 60);
}

template <int MASK>
__global__ void __launch_bounds__(K2F_THREADS, 4) k_alchemical_dense32(AlchArgs A) { alchemical_dense32_body<MASK>(A); }

template <int MASK>
__global__ void __launch_bounds__(K2D_THREADS) k_alchemical_dense(AlchArgs A) { alchemical_dense_body<MASK>(A); }

template <bool FAST>
__global__ void __launch_bounds__(256) k_alchemical(AlchArgs A) { alchemical_blocks<FAST>(A, blockIdx.x, gridDim.x - 1); }
