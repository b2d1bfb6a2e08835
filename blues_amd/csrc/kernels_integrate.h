// kernels_integrate.h -- the Langevin splitting substeps (K5), constraints (K6), COM-motion
// removal (K8) and the protocol-work bookkeeping (K7), fp64, one thread per constraint cluster.
//
// The reference assembles ~40 CustomIntegrator sub-steps per NCMC step
// (reference blues/integrators.py:159-231 on top of openmmtools' R/V/O steps); each becomes
// at least one kernel launch plus host-evaluated conditions on OpenMM's GPU platforms.
// Here the host (which mirrors the deterministic control state: step, lambda_step, prop)
// hands the device a short PROGRAM of ops per launch and one launch covers everything
// between two force passes, e.g. for "H V R O R V H":
//       V(slot0) H(0->1) END | CM H(1->2) V(slot2) R O R
//   = finish step k          | first half of step k+1
// HBonds / rigid-water constraints couple at most 4 atoms (X-H1..3 stars, water triangles),
// so SHAKE/RATTLE iterate inside one thread's registers: no cross-thread iteration.
#pragma once
#include "device_common.h"
#include "kernels_alch.h"
#include "kernels_nb.h"
#include "kernels_bonded.h"

enum {
    OP_V0 = 0, OP_V1, OP_V2,   // v += hV f/m with the alchemical force of slot 0/1/2, then RATTLE
    OP_R,                      // x += hR v ; SHAKE ; v += (x - x1)/hR ; RATTLE
    OP_O,                      // v = a v + b sqrt(kT/m) xi ; RATTLE
    OP_H01, OP_H12,            // protocol_work += E(slot+1) - E(slot)
    OP_END,                    // end of an NCMC step: record the work trace
    OP_PREP,                   // first-step block: constrain positions and velocities
    OP_CM_BLOCK,               // CMMotionRemover, whole system inside this (single) block
    OP_CM_REDUCE,              // write this block's momentum partial (last op of a launch)
    OP_CM_APPLY,               // subtract total momentum / total mass (first op of the next launch)
    OP_RATTLE,                 // velocity constraints only (setVelocitiesToTemperature)
    OP_CM_PART,                // CMMotionRemover from the momentum partials of k_finalize (only inside the specialised step kernels)
    OP_L,                      // OpenMM LangevinIntegrator step (MD leg): v' = a v + (1-a)/g f/m + sqrt(kT(1-a^2)/m) xi; x' = x + dt v'; SHAKE; v = (x'-x)/dt
    OP_A0, OP_A1, OP_A2,       // first half of a velocity-Verlet step (reference blues/switching.py:944-948): v += hV f/m (slot 0/1/2, NO RATTLE);
                               // x += hR v ; SHAKE ; v += (x - x1)/hR (no RATTLE: the second half, OP_V*, kicks and then constrains velocities)
};

#define MAX_OPS 24
struct Program { int n; unsigned char ops[MAX_OPS]; };

struct DevAccum {
    double protocol_work, dE_last, e_slot[3], heat;
};

// one packed record per cluster slot: everything the thread needs to know about its cluster arrives with a few
// wide contiguous loads instead of a dozen dependent gathers
struct ClusterRec {
    int atoms[4];    // caller index or -1
    int alch[4];     // local alchemical index or -1
    int mobile[4];   // index into the mobile list (noise buffer)
    int sorted[4];   // index into the tile image
    int islot[4];    // i-slot of the nonbonded kernel (mobile non-alchemical atoms) or -1
    int e0[4], e1[4];  // the atom's bonded entries [e0, e1) (BondedArgs::fent): what the step kernel sums when it forms the forces itself
    int type, nc, na, pad;
    double dist[3];
    double w[4];     // 1/mass (0 for empty slots)
};

struct IntArgs {
    int n, n_clusters;
    const ClusterRec* recs;
    double* x[3]; double* v[3];
    const double* mass;
    // force sources
    const double* ftot;       // [3 slots][3][n] total force on each mobile non-alchemical atom per lambda slot; slot 0 also
                              // carries the bonded force of alchemical atoms (k_finalize)
    const double* alch_self;  // [9][64] force on the alchemical atoms per slot (k_finalize)
    const double* fJ; const int* sorted_of_orig; int n_alch;
    const int* alch_local_of_orig;
    // ... or, when the step kernel forms them itself from the force kernels' slabs (step_default_body<CM, true>: no k_finalize launch):
    const double* fpart; const double* fent; const double* frec; const double* self_part; const double* e_part; const int* jcount_alch;
    int n_islots, npart, n_entries, PA, k2_nblocks_env, k2_jiter;
    double fin_le[3]; int fin_mask;   // lambda_electrostatics of the pass's three slots, slots whose force is applied (FinArgs::le, slot_mask)
    // constants
    double hV, hR, inv_hR, aO, bO, kT, tol;
    double dtL, aL, fsL, nsL;  // OP_L coefficients
    unsigned long long seed; unsigned stream, draw_base;
    const double* noise; const int* mobile_index; int n_mobile, n_noise; unsigned noise_draw_base;  // precomputed N(0,1), see k_bonded_entries
    Box3 box; int periodic; int cl_periodic;  // cl_periodic = 0: every cluster is stored as one whole image
    // tile image refresh + list validity
    AtomF* img_f; AtomD* img_d;
    const double* xbuild[3]; double half_skin2;
    // pruned per-atom lists of the nonbonded kernel (kernels_nb.h; null: none): where each i-atom was when its list was last pruned
    // (fixed point, by i-slot), the flag that asks for the next prune, (displacement that raises it)^2, box edge / 2^32
    const unsigned* xprune[3]; int* pneed; float prune_trig2; float fscale[3];
    DevFlags* flags; int* batch_req;   // batch_req: see ListArgs (null outside a list-synchronised batch)
    // COM removal
    double total_mass; double* cm_part; int cm_nblocks;
    const double* mom_part; int n_mom;  // momentum partials written by k_finalize (see FinArgs)
    DevAccum* acc; double* work_trace; int trace_index;
    DevCtrl* ctrl;
    long long* stamps;  // debug builds (-DBLUES_STAMP): cycle stamp of thread 0 at every op boundary
    Program prog;
};

struct Cluster {
    double x[4][3], v[4][3], w[4];  // w = 1/m
    int id[4], al[4];
    int na, nc, type;
    double d2[3];
};

// ---- cluster-local constraint solves (no register-array indexing by runtime values)
//   type 1 (star):     constraints (0,1) (0,2) (0,3)
//   type 2 (triangle): constraints (0,1) (0,2) (1,2)
// Round 6: ONE code path for both kinds.  The tables differ in the third constraint only -- (0,3) against (1,2) -- so the solver
// selects that constraint's two atoms and the five entries of the coupling matrix that involve it per lane (v_cndmask), and applies
// the third multiplier to all four atoms through weights of which two are zero (fma(t, 0, v) = v exactly).  A wave that holds stars
// AND triangles -- the second wave of a chain whose clusters are packed into two, BluesTuning.pack_clusters -- used to run the star
// solver and the triangle solver one after the other; the arithmetic per cluster is what the two templates did.
struct ClSel {
    bool star;
    double wa0, wa1, wb2, wb3;   // third constraint: weight of atoms 0 / 1 as its first atom, of atoms 2 / 3 as its second (two of them zero)
    double cf[3][3];             // coef(c, c2) = w[ci(c)] s(c2, ci(c)) - w[cj(c)] s(c2, cj(c)), s(c2, a) = +1 / -1 if a is the first / second atom of c2
};
__device__ __forceinline__ ClSel cl_select(const Cluster& C) {
    ClSel S; S.star = C.type == 1;
    const double w0 = C.w[0], w1 = C.w[1], w2 = C.w[2], w3 = C.w[3];
    S.wa0 = S.star ? w0 : 0.0; S.wa1 = S.star ? 0.0 : w1; S.wb2 = S.star ? 0.0 : w2; S.wb3 = S.star ? w3 : 0.0;
    S.cf[0][0] = w0 + w1; S.cf[0][1] = w0;      S.cf[0][2] = S.star ? w0 : -w1;
    S.cf[1][0] = w0;      S.cf[1][1] = w0 + w2; S.cf[1][2] = S.star ? w0 : w2;
    S.cf[2][0] = S.star ? w0 : -w1; S.cf[2][1] = S.star ? w0 : w2; S.cf[2][2] = S.star ? w0 + w3 : w1 + w2;
    return S;
}
// r[c] = a[first atom of c] - a[second atom of c]
__device__ __forceinline__ void cl_diffs(const double a[4][3], const bool star, const int nc, double r[3][3]) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        r[0][k] = a[0][k] - a[1][k];
        r[1][k] = nc > 1 ? a[0][k] - a[2][k] : 0.0;
        const double p = star ? a[0][k] : a[1][k], q = star ? a[3][k] : a[2][k];
        r[2][k] = nc > 2 ? p - q : 0.0;
    }
}
// a[first atom of c] += mu[c] r[c] w, a[second atom of c] -= mu[c] r[c] w, constraint by constraint
__device__ __forceinline__ void cl_apply(double a[4][3], const Cluster& C, const ClSel& S, const double mu[3], const double r[3][3]) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        { const double t = mu[0] * r[0][k]; a[0][k] += t * C.w[0]; a[1][k] -= t * C.w[1]; }
        if (C.nc > 1) { const double t = mu[1] * r[1][k]; a[0][k] += t * C.w[0]; a[2][k] -= t * C.w[2]; }
        if (C.nc > 2) { const double t = mu[2] * r[2][k]; a[0][k] += t * S.wa0; a[1][k] += t * S.wa1; a[2][k] -= t * S.wb2; a[3][k] -= t * S.wb3; }
    }
}

__device__ __forceinline__ void solve_small(int n, const double M[3][3], const double b[3], double x[3]) {
    if (n == 1) { x[0] = b[0] / M[0][0]; x[1] = x[2] = 0.0; return; }
    if (n == 2) {
        const double inv = 1.0 / (M[0][0] * M[1][1] - M[0][1] * M[1][0]);
        x[0] = (b[0] * M[1][1] - M[0][1] * b[1]) * inv;
        x[1] = (M[0][0] * b[1] - b[0] * M[1][0]) * inv;
        x[2] = 0.0;
        return;
    }
    const double c00 = M[1][1] * M[2][2] - M[1][2] * M[2][1], c01 = M[1][2] * M[2][0] - M[1][0] * M[2][2], c02 = M[1][0] * M[2][1] - M[1][1] * M[2][0];
    const double inv = 1.0 / (M[0][0] * c00 + M[0][1] * c01 + M[0][2] * c02);
    x[0] = (b[0] * c00 + b[1] * (M[0][2] * M[2][1] - M[0][1] * M[2][2]) + b[2] * (M[0][1] * M[1][2] - M[0][2] * M[1][1])) * inv;
    x[1] = (b[0] * c01 + b[1] * (M[0][0] * M[2][2] - M[0][2] * M[2][0]) + b[2] * (M[0][2] * M[1][0] - M[0][0] * M[1][2])) * inv;
    x[2] = (b[0] * c02 + b[1] * (M[0][1] * M[2][0] - M[0][0] * M[2][1]) + b[2] * (M[0][0] * M[1][1] - M[0][1] * M[1][0])) * inv;
}

// RATTLE: (v_i - v_j).r_ij = 0 is linear in the multipliers -> one small solve (same algebra as the oracle)
__device__ __forceinline__ void rattle_u(Cluster& C, double tol, const IntArgs& A) {
    if (C.nc == 0) return;
    const ClSel S = cl_select(C);
    double r[3][3], dv[3][3], M[3][3], b[3], mu[3];
    cl_diffs(C.x, S.star, C.nc, r);
    cl_diffs(C.v, S.star, C.nc, dv);
#pragma unroll
    for (int c = 0; c < 3; c++) b[c] = -(dv[c][0] * r[c][0] + dv[c][1] * r[c][1] + dv[c][2] * r[c][2]);
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int c2 = 0; c2 < 3; c2++) M[c][c2] = (r[c][0] * r[c2][0] + r[c][1] * r[c2][1] + r[c][2] * r[c2][2]) * S.cf[c][c2];
    solve_small(C.nc, M, b, mu);
    cl_apply(C.v, C, S, mu, r);
}

// SHAKE: Newton on the multipliers, directions from the reference geometry xr; converged to
// |r^2 - d^2| <= 2 tol d^2 plus one polishing iteration (same sequence as the oracle)
__device__ __forceinline__ bool shake_u(Cluster& C, const double xr[4][3], double tol, const IntArgs& A) {
    if (C.nc == 0) return true;
    const ClSel S = cl_select(C);
    double r[3][3];
    cl_diffs(xr, S.star, C.nc, r);
    int it;
    for (it = 0; it < 50; it++) {
        double D[3][3], g[3], J[3][3], dl[3];
        bool conv = true, tight = true;
        cl_diffs(C.x, S.star, C.nc, D);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            g[c] = 0.0;
            if (c < C.nc) {
                g[c] = -(D[c][0] * D[c][0] + D[c][1] * D[c][1] + D[c][2] * D[c][2] - C.d2[c]);
                if (fabs(g[c]) > 2.0 * tol * C.d2[c]) conv = false;
                if (fabs(g[c]) > 1e-13 * C.d2[c]) tight = false;
            }
        }
        if (tight) break;
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int c2 = 0; c2 < 3; c2++) J[c][c2] = 2.0 * (D[c][0] * r[c2][0] + D[c][1] * r[c2][1] + D[c][2] * r[c2][2]) * S.cf[c][c2];
        solve_small(C.nc, J, g, dl);
        cl_apply(C.x, C, S, dl, r);
        if (conv) break;
    }
    return it < 50;
}

// ---- the two-table form (compile-time index tables per kind of cluster): what every kernel but the fused steady-state step kernels runs --
// a wave that holds one kind of cluster (the lone chain's layout, the MD leg, the general interpreter) pays no selects; a wave with both
// kinds runs the two instantiations one after the other.  Same algebra as rattle_u / shake_u above.
template <int TYPE> struct ClTab {
    static __device__ constexpr int ci(int c) { return TYPE == 1 ? 0 : (c == 2 ? 1 : 0); }
    static __device__ constexpr int cj(int c) { return TYPE == 1 ? c + 1 : (c == 0 ? 1 : 2); }
    // s(c2, a): +1 if a is the first atom of constraint c2, -1 if the second
    static __device__ constexpr int s(int c2, int a) { return (a == ci(c2)) - (a == cj(c2)); }
};

// RATTLE: (v_i - v_j).r_ij = 0 is linear in the multipliers -> one small solve (same algebra as the oracle)
template <int TYPE> __device__ __forceinline__ void rattle_t(Cluster& C, const IntArgs& A) {
    using T = ClTab<TYPE>;
    double r[3][3], M[3][3], b[3], mu[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double rv = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            r[c][k] = 0.0;
            if (c < C.nc) {
                r[c][k] = C.x[T::ci(c)][k] - C.x[T::cj(c)][k];
                rv += (C.v[T::ci(c)][k] - C.v[T::cj(c)][k]) * r[c][k];
            }
        }
        b[c] = -rv;
    }
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int c2 = 0; c2 < 3; c2++) {
            const double coef = C.w[T::ci(c)] * T::s(c2, T::ci(c)) - C.w[T::cj(c)] * T::s(c2, T::cj(c));
            M[c][c2] = (r[c][0] * r[c2][0] + r[c][1] * r[c2][1] + r[c][2] * r[c2][2]) * coef;
        }
    solve_small(C.nc, M, b, mu);
#pragma unroll
    for (int c = 0; c < 3; c++) if (c < C.nc) {
#pragma unroll
        for (int k = 0; k < 3; k++) { C.v[T::ci(c)][k] += mu[c] * r[c][k] * C.w[T::ci(c)]; C.v[T::cj(c)][k] -= mu[c] * r[c][k] * C.w[T::cj(c)]; }
    }
}

__device__ __forceinline__ void rattle(Cluster& C, double tol, const IntArgs& A) {
    if (C.nc == 0) return;
    if (C.type == 1) rattle_t<1>(C, A); else rattle_t<2>(C, A);
}

// SHAKE: Newton on the multipliers, directions from the reference geometry xr; converged to
// |r^2 - d^2| <= 2 tol d^2 plus one polishing iteration (same sequence as the oracle)
template <int TYPE> __device__ __forceinline__ bool shake_t(Cluster& C, const double xr[4][3], double tol, const IntArgs& A) {
    using T = ClTab<TYPE>;
    double r[3][3];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            r[c][k] = 0.0;
            if (c < C.nc) { r[c][k] = xr[T::ci(c)][k] - xr[T::cj(c)][k]; }
        }
    int it;
    for (it = 0; it < 50; it++) {
        double D[3][3], g[3], J[3][3], dl[3];
        bool conv = true, tight = true;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            g[c] = 0.0;
#pragma unroll
            for (int k = 0; k < 3; k++) D[c][k] = 0.0;
            if (c < C.nc) {
#pragma unroll
                for (int k = 0; k < 3; k++) { D[c][k] = C.x[T::ci(c)][k] - C.x[T::cj(c)][k]; }
                g[c] = -(D[c][0] * D[c][0] + D[c][1] * D[c][1] + D[c][2] * D[c][2] - C.d2[c]);
                if (fabs(g[c]) > 2.0 * tol * C.d2[c]) conv = false;
                if (fabs(g[c]) > 1e-13 * C.d2[c]) tight = false;
            }
        }
        if (tight) break;
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int c2 = 0; c2 < 3; c2++) {
                const double coef = C.w[T::ci(c)] * T::s(c2, T::ci(c)) - C.w[T::cj(c)] * T::s(c2, T::cj(c));
                J[c][c2] = 2.0 * (D[c][0] * r[c2][0] + D[c][1] * r[c2][1] + D[c][2] * r[c2][2]) * coef;
            }
        solve_small(C.nc, J, g, dl);
#pragma unroll
        for (int c = 0; c < 3; c++) if (c < C.nc) {
#pragma unroll
            for (int k = 0; k < 3; k++) { C.x[T::ci(c)][k] += dl[c] * r[c][k] * C.w[T::ci(c)]; C.x[T::cj(c)][k] -= dl[c] * r[c][k] * C.w[T::cj(c)]; }
        }
        if (conv) break;
    }
    return it < 50;
}

__device__ __forceinline__ bool shake(Cluster& C, const double xr[4][3], double tol, const IntArgs& A) {
    if (C.nc == 0) return true;
    return C.type == 1 ? shake_t<1>(C, xr, tol, A) : shake_t<2>(C, xr, tol, A);
}
// U: the one-path form (the fused steady-state step kernels: a batch's packed layout mixes the kinds in one wave)
template <bool U> __device__ __forceinline__ void rattle_sel(Cluster& C, double tol, const IntArgs& A) { if constexpr (U) rattle_u(C, tol, A); else rattle(C, tol, A); }
template <bool U> __device__ __forceinline__ bool shake_sel(Cluster& C, const double xr[4][3], double tol, const IntArgs& A) { if constexpr (U) return shake_u(C, xr, tol, A); else return shake(C, xr, tol, A); }

__device__ __forceinline__ void load_force(const IntArgs& A, const Cluster& C, int slot, double F[4][3]) {
#pragma unroll
    for (int a = 0; a < 4; a++) {
        F[a][0] = F[a][1] = F[a][2] = 0.0;
        if (a < C.na) {
            const int i = C.id[a];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (C.al[a] >= 0) F[a][k] = A.ftot[(size_t)k * A.n + i] + A.alch_self[(slot * 3 + k) * 64 + C.al[a]];
                else F[a][k] = A.ftot[(size_t)(slot * 3 + k) * A.n + i];
            }
        }
    }
}

// Sums every per-block partial the force kernels left behind, once per pass and in a fixed order:
//   ftot[k][i]  = sum over K1 partial slabs + sum over the atom's bonded entries
//   alch_self   = sum over K2 blocks of the force on each alchemical atom (3 slots)
//   acc->e_slot = le[slot]*C + S[slot]  (alchemical energy per lambda slot, for the H ops)
// per-slot record of k_finalize: one 16-byte load replaces the chain islot -> atom -> row -> row_start
struct FinRec { int atom, sorted, e0, e1; };  // caller index (-1: empty), image index, bonded entry range

// ---- the sums of a force pass, as device functions shared by k_finalize and by the step kernel that forms the forces itself
// (step_default_body<CM, true>): ONE order of summation, so that a chain gets the same bits whichever kernel does it.
// lambda-independent force on an atom: nonbonded slabs (i-slot isl, or -1), then its bonded entries in order, then reciprocal space
__device__ __forceinline__ void fin_atom_base(const double* __restrict__ fpart, int n_islots, int npart, const double* __restrict__ fent, int n_entries,
                                              const double* __restrict__ frec, int n, int isl, int i, int e0, int e1, double f[3], const double* start = nullptr) {
    f[0] = start ? start[0] : 0.0; f[1] = start ? start[1] : 0.0; f[2] = start ? start[2] : 0.0;   // (start: slab sums formed elsewhere, k_finalize with npart > 1)
    if (isl >= 0) for (int p = 0; p < npart; p += 8) {   // (a lone chain's tile kernel leaves dozens of partial slabs: 8 x 3 loads in flight, added in slab order)
        double t[8][3];
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int k = 0; k < 3; k++) t[u][k] = p + u < npart ? fpart[((size_t)(p + u) * 3 + k) * n_islots + isl] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; u++) if (p + u < npart) { f[0] += t[u][0]; f[1] += t[u][1]; f[2] += t[u][2]; }
    }
    for (int e = e0; e < e1; e += 8) {   // 8 entries x 3 components requested together, added in entry order (a loop of dependent load -> add took one round trip per entry)
        double t[8][3];
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int k = 0; k < 3; k++) t[u][k] = e + u < e1 ? fent[(size_t)k * n_entries + e + u] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; u++) if (e + u < e1) { f[0] += t[u][0]; f[1] += t[u][1]; f[2] += t[u][2]; }
    }
    if (frec) { f[0] += frec[i]; f[1] += frec[(size_t)n + i]; f[2] += frec[2 * (size_t)n + i]; }
}
// slab q = slot * 3 + component of the alchemical atoms' pair force: a block of 256 threads, thread = (alchemical atom a, block
// group g), fixed order; the value for thread tid < n_alch (0 elsewhere).  Two block barriers.
__device__ __forceinline__ double fin_alch_self(const double* __restrict__ self_part, int PA, int nb_env, int k2_nblocks_env, int q, bool on, int n_alch, double* s_red) {
    const int tid = threadIdx.x, NG = 256 / PA, a = tid & (PA - 1), g = tid / PA;
    double s = 0.0;
    if (on) {
        for (int b = g; b < nb_env; b += NG) s += self_part[((size_t)b * 9 + q) * PA + a];
        if (g == 0) s += self_part[((size_t)k2_nblocks_env * 9 + q) * PA + a];
    }
    __syncthreads();   // (s_red may still be read from the previous slab)
    s_red[tid] = s;
    __syncthreads();
    double t = 0.0;
    if (on && tid < n_alch) for (int u = 0; u < NG; u++) t += s_red[u * PA + tid];   // fixed order
    return t;
}
// the K2_NP sums over the alchemical kernel's blocks (energies per kind, slot-0 force on the alchemical atoms) -> s_e[K2_NP]; whole waves (256 threads; 128 in the packed step kernel), one barrier at the end
__device__ __forceinline__ void fin_energy_sums(const double* __restrict__ e_part, int nb_env, int k2_nblocks_env, double* s_e) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nwv = (int)(blockDim.x >> 6);   // (a wave forms a sum in lane order: the same bits whichever wave takes it)
    for (int q = wv; q < K2_NP; q += nwv) {
        double s = 0.0;
        for (int b = lane; b < nb_env; b += 64) s += e_part[(size_t)b * K2_NP + q];
        if (lane == 0) s += e_part[(size_t)k2_nblocks_env * K2_NP + q];
        s = wave_sum(s);
        if (lane == 0) s_e[q] = s;
    }
    __syncthreads();
}

struct FinArgs {
    int n, n_islots, npart, n_alch, PA, k2_nblocks_env, k2_jiter, n_entries;
    const FinRec* recs;         // [n_islots + 64]: i-slots, then the alchemical atoms
    const int* orig_of_islot;   // [n_islots] caller index or -1
    const int* row_of_orig;     // [n] bonded row or -1
    const int* row_start;
    const double* fpart; const double* fent; const double* fJ; const int* sorted_of_orig;
    const int* alch_orig;
    const double* self_part; const double* e_part; const int* jcount_alch;
    double le[3];
    int slot_mask;              // lambda slots whose force will be applied (bit s); the others are neither read nor written
    double* ftot; double* alch_self; DevAccum* acc;
    DevCtrl* ctrl;
    // momentum bookkeeping for CMMotionRemover without a grid-wide reduction inside the step kernel:
    // mom_part[block][0..2] = sum m v, [3..5] = sum of the slot-0 force, over the block's atoms
    const double* v[3]; const double* mass; double* mom_part;
    const double* frec;   // [3][n] reciprocal-space force (kernels_pme.h), lambda-independent; null without BLUES_NB_PME
};

// grid: [0, n_itiles) one block per i-tile | [n_itiles, +nb_alch_atoms) alchemical atoms' bonded rows |
//       then 9 blocks (one per slot x component) for the alchemical self-force slabs | 1 block for the energies.
// Every sum is spread over the 4 waves of a block with 4 loads in flight per lane, then combined in LDS in a
// fixed order (deterministic, and no chain of dependent global loads).
// LEAN: same sums in the same order without the deep load batching -- for replica batches, where thousands of these blocks
// are resident and occupancy (registers) counts for more than the latency of one block's load chain
template <bool LEAN>   // (LEAN: kept as a name of the batched instantiation; the sums are the same shared functions either way)
__device__ __forceinline__ void finalize_body(FinArgs& A) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (A.ctrl) {
        const int L = A.ctrl->L0 + 2 * A.ctrl->kpass;
#pragma unroll
        for (int s = 0; s < 3; s++) A.le[s] = A.ctrl->tab_le[min(L + s, A.ctrl->n_lambda)];
        if (blockIdx.x == 0 && tid == 0) A.ctrl->kint = A.ctrl->kpass;  // nothing in this kernel reads kint
    }
    const int n_itiles = A.n_islots / 64;
    const int nb_alch_atoms = A.n_alch > 0 ? 1 : 0;
    int blk = blockIdx.x;
    if (blk < n_itiles + nb_alch_atoms) {
        const int isl = blk < n_itiles ? blk * 64 + lane : -1;
        const FinRec rec = A.recs[blk < n_itiles ? isl : A.n_islots + lane];
        const int i = rec.atom;
        // A lone chain's tile kernel leaves npart > 1 partial slabs: their sum is split over the block's four waves (8 loads in flight
        // each) and combined in LDS in a fixed order, then handed to fin_atom_base as its starting value.  With ONE slab -- the only case
        // in which the step kernel forms these sums itself -- wave 0 does everything, in fin_atom_base's order: the same bits either way.
        double pre[3] = {0.0, 0.0, 0.0};
        if (A.npart > 1) {   // (uniform)
            __shared__ double red[4][3][64];
            double t3[3] = {0.0, 0.0, 0.0};
            if (i >= 0 && isl >= 0) for (int p0 = wv; p0 < A.npart; p0 += 32) {
                double t[8][3];
#pragma unroll
                for (int u = 0; u < 8; u++)
#pragma unroll
                    for (int k = 0; k < 3; k++) t[u][k] = p0 + 4 * u < A.npart ? A.fpart[((size_t)(p0 + 4 * u) * 3 + k) * A.n_islots + isl] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; u++) if (p0 + 4 * u < A.npart) { t3[0] += t[u][0]; t3[1] += t[u][1]; t3[2] += t[u][2]; }
            }
            red[wv][0][lane] = t3[0]; red[wv][1][lane] = t3[1]; red[wv][2][lane] = t3[2];
            __syncthreads();
            if (wv != 0) return;
#pragma unroll
            for (int k = 0; k < 3; k++) pre[k] = red[0][k][lane] + red[1][k][lane] + red[2][k][lane] + red[3][k][lane];
        } else if (wv != 0) return;   // one thread per atom
        double f[3] = {0.0, 0.0, 0.0};
        double fj[3][3];  // alchemical force on this (environment) atom per slot
#pragma unroll
        for (int sl = 0; sl < 3; sl++)
#pragma unroll
            for (int k = 0; k < 3; k++) fj[sl][k] = (i >= 0 && isl >= 0 && A.n_alch > 0 && ((A.slot_mask >> sl) & 1)) ? A.fJ[(size_t)(sl * 3 + k) * A.n + rec.sorted] : 0.0;
        if (i >= 0) {
            fin_atom_base(A.fpart, A.n_islots, A.npart > 1 ? 0 : A.npart, A.fent, A.n_entries, A.frec, A.n, isl, i, rec.e0, rec.e1, f, pre);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (isl >= 0 && A.n_alch > 0) {
#pragma unroll
                    for (int sl = 0; sl < 3; sl++) if ((A.slot_mask >> sl) & 1) A.ftot[(size_t)(sl * 3 + k) * A.n + i] = f[k] + fj[sl][k];
                } else if (isl >= 0) {
#pragma unroll
                    for (int sl = 0; sl < 3; sl++) if ((A.slot_mask >> sl) & 1) A.ftot[(size_t)(sl * 3 + k) * A.n + i] = f[k];
                } else A.ftot[(size_t)k * A.n + i] = f[k];  // alchemical atom: bonded part; integrator adds alch_self[slot]
            }
        }
        double pm[6];
        const double m = i >= 0 ? A.mass[i] : 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            pm[k] = i >= 0 ? m * A.v[k][i] : 0.0;
            pm[3 + k] = i >= 0 ? f[k] + ((isl >= 0 && A.n_alch > 0) ? fj[0][k] : 0.0) : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 6; q++) { pm[q] = wave_sum(pm[q]); if (lane == 0) A.mom_part[(size_t)blk * 6 + q] = pm[q]; }
        return;
    }
    if (A.n_alch == 0) return;
    blk -= n_itiles + nb_alch_atoms;
    const int cnt = *A.jcount_alch;
    const int nb_env = k2_env_blocks(cnt, A.PA, A.k2_jiter);
    if (blk < 9) {  // slab q = slot*3 + component
        __shared__ double s_red[256];
        const bool on = (A.slot_mask >> (blk / 3)) & 1;
        const double t = fin_alch_self(A.self_part, A.PA, nb_env, A.k2_nblocks_env, blk, on, A.n_alch, s_red);
        if (on && tid < A.n_alch) A.alch_self[blk * 64 + tid] = t;
        return;
    }
    __shared__ double s_e[K2_NP];
    fin_energy_sums(A.e_part, nb_env, A.k2_nblocks_env, s_e);
    // (constant indices only: a runtime index into A -- even a select between its fields -- keeps the whole argument
    // struct in scratch memory and turns every A.field access of the kernel into a scratch load)
    if (tid == 0) A.acc->e_slot[0] = A.le[0] * s_e[0] + s_e[1];
    if (tid == 1) A.acc->e_slot[1] = A.le[1] * s_e[0] + s_e[2];
    if (tid == 2) A.acc->e_slot[2] = A.le[2] * s_e[0] + s_e[3];
    if (tid < 6) A.mom_part[(size_t)(n_itiles + 1) * 6 + tid] = tid < 3 ? 0.0 : s_e[K2_NE + tid - 3];  // alchemical pair force on the alchemical atoms
}

__global__ void __launch_bounds__(256) k_finalize(FinArgs A) { finalize_body<false>(A); }

// `prog` is passed beside A so that the batched launch (kernels_batch.h) can hand over the shared program from its own
// kernel arguments while A is a per-replica copy
__device__ __forceinline__ void integrate_body(IntArgs& A, const Program& prog) {
    const int tid = threadIdx.x;
    if (A.ctrl) {
        const int u = A.ctrl->kint;
        A.draw_base = A.ctrl->draw0 + (unsigned)u; A.noise_draw_base = A.draw_base; A.trace_index = A.ctrl->trace0 + u;
        if (blockIdx.x == 0 && tid == 0) A.ctrl->kpass = u + 1;  // nothing in this kernel reads kpass
    }
    const int cl = blockIdx.x * blockDim.x + tid;
    __shared__ double s_red[4][4];
    __shared__ double s_cm[3];

    Cluster C;
    const bool active = cl < A.n_clusters;
    ClusterRec R;
    if (active) R = A.recs[cl];
    else { for (int a = 0; a < 4; a++) { R.atoms[a] = -1; R.alch[a] = -1; R.mobile[a] = 0; R.sorted[a] = 0; R.islot[a] = -1; R.w[a] = 0.0; } R.type = 0; R.nc = 0; R.na = 0; R.dist[0] = R.dist[1] = R.dist[2] = 0.0; }
    C.na = R.na; C.nc = R.nc; C.type = R.type;
    bool moved = false, ok = true;
    unsigned draw = A.draw_base;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const int i = max(R.atoms[a], 0);
        C.id[a] = R.atoms[a]; C.al[a] = R.alch[a]; C.w[a] = R.w[a];
#pragma unroll
        for (int k = 0; k < 3; k++) { C.x[a][k] = A.x[k][i]; C.v[a][k] = A.v[k][i]; }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) C.d2[c] = R.dist[c] * R.dist[c];
#pragma unroll
    for (int a = 0; a < 4; a++) if (a >= C.na) {
#pragma unroll
        for (int k = 0; k < 3; k++) { C.x[a][k] = 0.0; C.v[a][k] = 0.0; }
    }
#ifdef BLUES_STAMP
    if (cl == 0 && A.stamps) A.stamps[0] = clock64();
#endif
    for (int op_i = 0; op_i < prog.n; op_i++) {
        const int op = prog.ops[op_i];
#ifdef BLUES_STAMP
        if (cl == 0 && A.stamps) A.stamps[1 + op_i] = clock64();
#endif
        switch (op) {
        case OP_V0: case OP_V1: case OP_V2: {
            if (active) {
                double F[4][3];
                load_force(A, C, op - OP_V0, F);
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int k = 0; k < 3; k++) C.v[a][k] += A.hV * F[a][k] * C.w[a];
                rattle(C, A.tol, A);
            }
        } break;
        case OP_R: {
            if (active) {
                double xr[4][3], x1[4][3];
#pragma unroll
                for (int a = 0; a < 4; a++) for (int k = 0; k < 3; k++) {
                    xr[a][k] = C.x[a][k];
                    if (a < C.na) C.x[a][k] += A.hR * C.v[a][k];
                    x1[a][k] = C.x[a][k];
                }
                ok &= shake(C, xr, A.tol, A);
#pragma unroll
                for (int a = 0; a < 4; a++) if (a < C.na) for (int k = 0; k < 3; k++) C.v[a][k] += (C.x[a][k] - x1[a][k]) * A.inv_hR;
                rattle(C, A.tol, A);
                moved = true;
            }
        } break;
        case OP_A0: case OP_A1: case OP_A2: {
            if (active) {
                double F[4][3], xr[4][3], x1[4][3];
                load_force(A, C, op - OP_A0, F);
#pragma unroll
                for (int a = 0; a < 4; a++) for (int k = 0; k < 3; k++) {
                    C.v[a][k] += A.hV * F[a][k] * C.w[a];
                    xr[a][k] = C.x[a][k];
                    if (a < C.na) C.x[a][k] += A.hR * C.v[a][k];
                    x1[a][k] = C.x[a][k];
                }
                ok &= shake(C, xr, A.tol, A);
#pragma unroll
                for (int a = 0; a < 4; a++) if (a < C.na) for (int k = 0; k < 3; k++) C.v[a][k] += (C.x[a][k] - x1[a][k]) * A.inv_hR;
                moved = true;
            }
        } break;
        case OP_O: {
            if (active) {
#pragma unroll
                for (int a = 0; a < 4; a++) if (a < C.na) {
                    double g[3];
                    const unsigned nd = draw - A.noise_draw_base;
                    if (nd < (unsigned)A.n_noise) {
                        const int m = R.mobile[a];
                        for (int k = 0; k < 3; k++) g[k] = A.noise[(size_t)(nd * 3 + k) * A.n_mobile + m];
                    } else gaussians3(A.seed, A.stream, draw, (unsigned)C.id[a], g);
                    const double s = sqrt(A.kT * C.w[a]);
                    for (int k = 0; k < 3; k++) C.v[a][k] = A.aO * C.v[a][k] + A.bO * s * g[k];
                }
                rattle(C, A.tol, A);
            }
            draw++;
        } break;
        case OP_L: {
            if (active) {
                double F[4][3], x0[4][3];
                load_force(A, C, 0, F);
#pragma unroll
                for (int a = 0; a < 4; a++) {
#pragma unroll
                    for (int k = 0; k < 3; k++) x0[a][k] = C.x[a][k];
                    if (a < C.na) {
                        double g[3];
                        const unsigned nd = draw - A.noise_draw_base;
                        if (nd < (unsigned)A.n_noise) {
                            const int m = R.mobile[a];
                            for (int k = 0; k < 3; k++) g[k] = A.noise[(size_t)(nd * 3 + k) * A.n_mobile + m];
                        } else gaussians3(A.seed, A.stream, draw, (unsigned)C.id[a], g);
                        const double sd = sqrt(A.kT * C.w[a]);
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            C.v[a][k] = A.aL * C.v[a][k] + A.fsL * C.w[a] * F[a][k] + A.nsL * sd * g[k];
                            C.x[a][k] += A.dtL * C.v[a][k];
                        }
                    }
                }
                ok &= shake(C, x0, A.tol, A);
                const double inv_dt = 1.0 / A.dtL;
#pragma unroll
                for (int a = 0; a < 4; a++) if (a < C.na) for (int k = 0; k < 3; k++) C.v[a][k] = (C.x[a][k] - x0[a][k]) * inv_dt;
                moved = true;
            }
            draw++;
        } break;
        case OP_H01: case OP_H12: {
            if (blockIdx.x == 0 && tid == 0) {
                const int s = op - OP_H01;
                const double dE = A.acc->e_slot[s + 1] - A.acc->e_slot[s];
                A.acc->protocol_work += dE; A.acc->dE_last = dE;
            }
        } break;
        case OP_END: {
            if (blockIdx.x == 0 && tid == 0 && A.work_trace) A.work_trace[A.trace_index] = A.acc->protocol_work;
        } break;
        case OP_RATTLE: {
            if (active) rattle(C, A.tol, A);
        } break;
        case OP_PREP: {
            if (active) {
                double xr[4][3];
#pragma unroll
                for (int a = 0; a < 4; a++) for (int k = 0; k < 3; k++) xr[a][k] = C.x[a][k];
                ok &= shake(C, xr, A.tol, A);
                rattle(C, A.tol, A);
                moved = true;
            }
        } break;
        case OP_CM_BLOCK: case OP_CM_REDUCE: {
            double p[3] = {0.0, 0.0, 0.0};
            if (active) {
_Pragma("unroll") for (int a = 0; a < 4; a++) if (a < C.na) for (int k = 0; k < 3; k++) p[k] += C.v[a][k] / C.w[a];
            }
            for (int k = 0; k < 3; k++) { p[k] = wave_sum(p[k]); if ((tid & 63) == 0) s_red[tid >> 6][k] = p[k]; }
            __syncthreads();
            if (tid < 3) {
                double s = 0.0;
                for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += s_red[w][tid];
                if (op == OP_CM_BLOCK) s_cm[tid] = s / A.total_mass; else A.cm_part[blockIdx.x * 3 + tid] = s;
            }
            __syncthreads();
            if (op == OP_CM_BLOCK && active) {
_Pragma("unroll") for (int a = 0; a < 4; a++) if (a < C.na) for (int k = 0; k < 3; k++) C.v[a][k] -= s_cm[k];
            }
        } break;
        case OP_CM_APPLY: {
            if (tid < 3) {
                double s = 0.0;
                for (int b = 0; b < A.cm_nblocks; b++) s += A.cm_part[b * 3 + tid];
                s_cm[tid] = s / A.total_mass;
            }
            __syncthreads();
            if (active) {
_Pragma("unroll") for (int a = 0; a < 4; a++) if (a < C.na) for (int k = 0; k < 3; k++) C.v[a][k] -= s_cm[k];
            }
        } break;
        }
    }

#ifdef BLUES_STAMP
    if (cl == 0 && A.stamps) A.stamps[1 + prog.n] = clock64();
#endif
    if (!active) return;
    // ---- write back, refresh the fixed-point image, check list validity
    bool need_rebuild = false, bad = false;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        if (a < C.na) {
            const int i = C.id[a];
            for (int k = 0; k < 3; k++) { A.v[k][i] = C.v[a][k]; bad |= !(C.x[a][k] == C.x[a][k]) || !(C.v[a][k] == C.v[a][k]); }
            if (moved) {
                double d2 = 0.0;
                for (int k = 0; k < 3; k++) { A.x[k][i] = C.x[a][k]; const double d = C.x[a][k] - A.xbuild[k][i]; d2 += d * d; }
                need_rebuild |= d2 > A.half_skin2;
                const int s = R.sorted[a];
                if (A.img_f) {
                    unsigned u[3]; to_fixed32(C.x[a], A.box, u); A.img_f[s].x = u[0]; A.img_f[s].y = u[1]; A.img_f[s].z = u[2];
                    if (A.pneed && R.islot[a] >= 0) {
                        float p2 = 0.0f;
                        for (int k = 0; k < 3; k++) { const float e = (float)(int)(u[k] - A.xprune[k][R.islot[a]]) * A.fscale[k]; p2 += e * e; }
                        if (p2 > A.prune_trig2) A.pneed[R.islot[a]] = 1;
                    }
                }
                else { unsigned long long u[3]; to_fixed(C.x[a], A.box, u); A.img_d[s].x = u[0]; A.img_d[s].y = u[1]; A.img_d[s].z = u[2]; }
            }
        }
    }
    if (need_rebuild) { A.flags->req_gen = A.flags->list_gen + 1; if (A.batch_req) *A.batch_req = 1; }
#ifdef BLUES_STAMP
    if (cl == 0 && A.stamps) A.stamps[2 + prog.n] = clock64();
#endif
    if (bad) A.flags->nan_flag = 1;
    if (!ok) A.flags->constraint_fail = 1;
}

__global__ void __launch_bounds__(256) k_integrate(IntArgs A) { integrate_body(A, A.prog); }

// ---- state transfer (setPositions / State snapshots): new positions arrive either interleaved from the host
// ([n][3], stride 3) or as a device-resident snapshot (SoA, stride 1).  One pass writes the master positions, refreshes
// the fixed-point tile image and gathers what the host needs to decide about re-sorting and cached energies:
//   out[0] != 0  a frozen environment atom changed (the frozen-frozen energy is stale)
//   out[1]       max squared displacement of a mobile non-alchemical atom since the last spatial sort (float bits)
//   out[2]       number of atoms farther than 0.5 nm from where they were sorted
struct LoadPosArgs {
    int n, stride;
    int keep_out;   // batched call: an older verdict is still pending in `out`, do not zero it
    const double* src[3];
    double* x[3];
    const double* x_sort[3];
    const double* mass; const int* alch_local; const int* sorted_of_orig;
    AtomF* img_f; AtomD* img_d; Box3 box;
    unsigned* out;
};
__device__ __forceinline__ void load_positions_body(const LoadPosArgs& a, const int block) {
    const int i = block * 256 + threadIdx.x;
    unsigned changed = 0, far = 0; float worst = 0.0f;
    if (i < a.n) {
        double p[3], d2 = 0.0;
        const bool frozen = a.mass[i] == 0.0 && a.alch_local[i] < 0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            p[k] = a.src[k][(size_t)i * a.stride];
            if (frozen && p[k] != a.x[k][i]) changed = 1;
            a.x[k][i] = p[k];
            const double d = min_image_d(p[k] - a.x_sort[k][i], a.box.L[k], a.box.invL[k]);
            d2 += d * d;
        }
        if (a.mass[i] != 0.0 && a.alch_local[i] < 0) worst = (float)d2;
        far = d2 > 0.25;
        const int s = a.sorted_of_orig[i];
        if (a.img_f) { unsigned u[3]; to_fixed32(p, a.box, u); a.img_f[s].x = u[0]; a.img_f[s].y = u[1]; a.img_f[s].z = u[2]; }
        else { unsigned long long u[3]; to_fixed(p, a.box, u); a.img_d[s].x = u[0]; a.img_d[s].y = u[1]; a.img_d[s].z = u[2]; }
    }
    // block-level combine, then one atomic per block and quantity
    __shared__ unsigned s_c[4], s_f[4]; __shared__ float s_w[4];
    for (int o = 32; o > 0; o >>= 1) { changed |= __shfl_xor(changed, o, 64); far += __shfl_xor(far, o, 64); worst = fmaxf(worst, __shfl_xor(worst, o, 64)); }
    if ((threadIdx.x & 63) == 0) { s_c[threadIdx.x >> 6] = changed; s_f[threadIdx.x >> 6] = far; s_w[threadIdx.x >> 6] = worst; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned c = s_c[0] | s_c[1] | s_c[2] | s_c[3], f = s_f[0] + s_f[1] + s_f[2] + s_f[3];
        const float w = fmaxf(fmaxf(s_w[0], s_w[1]), fmaxf(s_w[2], s_w[3]));
        if (c) atomicOr(&a.out[0], 1u);
        if (w > 0.0f) atomicMax(&a.out[1], __float_as_uint(w));   // non-negative floats order like their bit patterns
        if (f) atomicAdd(&a.out[2], f);
    }
}
__global__ void __launch_bounds__(256) k_load_positions(LoadPosArgs a) { load_positions_body(a, blockIdx.x); }

// a few atoms of the positions just loaded are overwritten (a Move's edit of a State handed back otherwise unchanged):
// one thread per edited atom; same statistics as k_load_positions, for the new coordinates
struct EditPosArgs {
    int n_edit; const int* idx; const double* xyz;   // [n_edit], [n_edit][3]
    double* x[3]; const double* x_sort[3];
    const double* mass; const int* alch_local; const int* sorted_of_orig;
    AtomF* img_f; AtomD* img_d; Box3 box;
    unsigned* out;
};
__device__ __forceinline__ void edit_positions_body(const EditPosArgs& a, const int block) {
    const int e = block * 64 + threadIdx.x;
    if (e >= a.n_edit) return;
    const int i = a.idx[e];
    double p[3], d2 = 0.0;
    const bool frozen = a.mass[i] == 0.0 && a.alch_local[i] < 0;
    bool changed = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        p[k] = a.xyz[3 * e + k];
        changed |= p[k] != a.x[k][i];
        a.x[k][i] = p[k];
        const double d = min_image_d(p[k] - a.x_sort[k][i], a.box.L[k], a.box.invL[k]);
        d2 += d * d;
    }
    if (frozen && changed) atomicOr(&a.out[0], 1u);
    if (a.mass[i] != 0.0 && a.alch_local[i] < 0) atomicMax(&a.out[1], __float_as_uint((float)d2));
    if (d2 > 0.25) atomicAdd(&a.out[2], 1u);
    const int s = a.sorted_of_orig[i];
    if (a.img_f) { unsigned u[3]; to_fixed32(p, a.box, u); a.img_f[s].x = u[0]; a.img_f[s].y = u[1]; a.img_f[s].z = u[2]; }
    else { unsigned long long u[3]; to_fixed(p, a.box, u); a.img_d[s].x = u[0]; a.img_d[s].y = u[1]; a.img_d[s].z = u[2]; }
}
__global__ void __launch_bounds__(64) k_edit_positions(EditPosArgs a) { edit_positions_body(a, blockIdx.x); }

// gather of a few atoms' coordinates (snapshot or live) into a dense [n][3] buffer
__global__ void __launch_bounds__(64) k_gather_atoms(int n_idx, const int* idx, const double* s0, const double* s1, const double* s2, double* out) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= n_idx) return;
    const int i = idx[e];
    out[3 * e] = s0[i]; out[3 * e + 1] = s1[i]; out[3 * e + 2] = s2[i];
}

// plain copies between engine state and snapshots / the host staging buffer: up to 6 arrays of n doubles in one launch;
// dst_stride 3 interleaves (device-side transpose for getPositions), src_stride 3 de-interleaves
struct Copy6Args { int n, count, src_stride, dst_stride; const double* src[6]; double* dst[6]; };
__global__ void __launch_bounds__(256) k_copy_arrays(Copy6Args a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    for (int q = 0; q < a.count; q++) a.dst[q][(size_t)i * a.dst_stride] = a.src[q][(size_t)i * a.src_stride];
}

// protocol_work += delta (the work of an instantaneous Move, reference blues/integrators.py:184-191)
__global__ void k_add_work(DevAccum* acc, double delta) { acc->protocol_work += delta; }

// v = -v_saved (momentum flip of a rejected GHMC trial, reference blues/switching.py:1002-1005)
__global__ void __launch_bounds__(256) k_negated_copy3(int n, const double* s0, const double* s1, const double* s2, double* d0, double* d1, double* d2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    d0[i] = -s0[i]; d1[i] = -s1[i]; d2[i] = -s2[i];
}

// kinetic energy: per-block partials of sum 0.5 m v^2 over mobile atoms
__global__ void __launch_bounds__(256) k_kinetic(int n, const double* __restrict__ mass, const double* vx, const double* vy, const double* vz, double* part) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double ke = 0.0;
    if (i < n && mass[i] != 0.0) ke = 0.5 * mass[i] * (vx[i] * vx[i] + vy[i] * vy[i] + vz[i] * vz[i]);
    __shared__ double s[4];
    ke = wave_sum(ke);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = ke;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// setVelocitiesToTemperature: v = sqrt(kT/m) N(0,1) for mobile atoms, 0 for frozen ones
__global__ void k_maxwell(int n, const double* __restrict__ mass, double* vx, double* vy, double* vz, double kT, unsigned long long seed, unsigned stream) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double g[3] = {0.0, 0.0, 0.0};
    if (mass[i] != 0.0) { gaussians3(seed, stream, 0u, (unsigned)i, g); const double s = sqrt(kT / mass[i]); g[0] *= s; g[1] *= s; g[2] *= s; }
    vx[i] = g[0]; vy[i] = g[1]; vz[i] = g[2];
}

// total forces in caller order for getState(getForces=True); frozen atoms report 0
__global__ void k_gather_forces(IntArgs A, int slot, double* out /*[n][3]*/) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    const int al = A.n_alch > 0 ? A.alch_local_of_orig[i] : -1;
    for (int k = 0; k < 3; k++) {
        double f = al >= 0 ? A.ftot[(size_t)k * A.n + i] + A.alch_self[(slot * 3 + k) * 64 + al] : A.ftot[(size_t)(slot * 3 + k) * A.n + i];
        out[3 * i + k] = A.mass[i] != 0.0 ? f : 0.0;
    }
}

// The three force kernels of a pass are independent of each other; when the i-set is small they are all
// latency-bound, so one launch with block-role dispatch runs them side by side on different CUs and saves
// two kernel boundaries:  blocks [0,nb1) nonbonded | [nb1,nb1+nb2) alchemical | the rest bonded entries + noise.
template <typename R>
__global__ void __launch_bounds__(256) k_forces_fused(NbArgs<R> a, NbConst<R> c, const typename Img<R>::Atom* __restrict__ img,
                                                      AlchArgs A, BondedArgs B, int nb1, int nb2) {
    const int b = blockIdx.x;
    if (b < nb1) { nonbonded_body<R, false, 4>(a, c, img, b); return; }
    if (b < nb1 + nb2) { alchemical_body<sizeof(R) == 4>(A, b - nb1); return; }
    bonded_entries_body(B, b - nb1 - nb2, 256);
}

// ---- straight-line specialisation of the steady-state program of "H V R O R V H":
//        V(slot0) H(0->1) END [CM] H(1->2) V(slot2) R O R
// Same device functions and arithmetic as the interpreter above (bitwise identical results), but with no op
// dispatch the compiler sees one basic-block chain: every gather is issued at the top and waited for once, which is
// what this latency-bound kernel needs (the interpreter spent >65 % of its wave cycles in s_waitcnt).
// FUSED (round 4): the kernel forms the pass's forces itself from the force kernels' slabs -- what k_finalize would have left in
// ftot / alch_self / acc->e_slot / mom_part, by the same shared functions in the same order -- so a steady-state step has one
// launch fewer and no round trip of the summed forces through memory.  One block of 256 threads holds every cluster of the chain
// (host: int_blocks == 1, int_threads == 256 -- or 128 where BluesTuning.pack_clusters packed the chain's clusters into two waves: a
// batch of more than 512 chains runs its 256-register waves in rounds), so the total momentum for CMMotionRemover is a block reduction.
// LATE (with FUSED in a batch): the noise of the O substep and the reference positions of the list checks are loaded where they are
// used instead of at the top -- 60 registers less across the constraint solves, which lets two waves share a SIMD (256 registers:
// kernels_batch.h) where the kernel otherwise runs one wave per SIMD in two rounds.
#ifdef BLUES_STAMP
#define STEP_STAMP(i) do { if (threadIdx.x == 0 && A.stamps) A.stamps[40 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define STEP_STAMP(i) do { } while (0)
#endif
#define STEP_FENT_LDS 512   // bonded entries of a chain the fused step kernel keeps in LDS (a chain with more is not fused: blues_engine.hip, fin_fusable)
template <bool CM, bool FUSED = false, bool LATE = false>
__device__ __forceinline__ void step_default_body(IntArgs& A) {
    const int tid = threadIdx.x;
    STEP_STAMP(0);
    const int cl = blockIdx.x * blockDim.x + tid;
    if (A.ctrl) {
        const int u = A.ctrl->kint;
        A.draw_base = A.ctrl->draw0 + (unsigned)u; A.noise_draw_base = A.draw_base; A.trace_index = A.ctrl->trace0 + u;
        if (blockIdx.x == 0 && tid == 0) A.ctrl->kpass = u + 1;
    }
    __shared__ double s_red[4][4];
    __shared__ double s_cm[3];
    const bool active = cl < A.n_clusters;
    ClusterRec R;
    if (active) R = A.recs[cl];
    else { for (int a = 0; a < 4; a++) { R.atoms[a] = -1; R.alch[a] = -1; R.mobile[a] = 0; R.sorted[a] = 0; R.islot[a] = -1; R.w[a] = 0.0; } R.type = 0; R.nc = 0; R.na = 0; R.dist[0] = R.dist[1] = R.dist[2] = 0.0; }
#if defined(STEPX_DENSE)   // (floor experiment: every gather and scatter of the kernel at consecutive addresses -- wrong numbers, only the time matters)
    for (int a = 0; a < 4; a++) if (R.atoms[a] >= 0) { R.atoms[a] = cl * 4 + a; R.sorted[a] = cl * 4 + a; }
#endif
    Cluster C;
    C.na = R.na; C.nc = R.nc; C.type = R.type;
    const unsigned nd0 = A.draw_base - A.noise_draw_base;
    const bool pre_noise = nd0 < (unsigned)A.n_noise;
    double FA[4][3], FB[4][3], G0[4][3], XB[4][3];
    unsigned XP[4][3];
    const bool pruned = A.pneed != nullptr;   // (uniform)
    __shared__ double s_fin[FUSED ? 6 : 1][FUSED ? 256 : 1];   // FUSED: reduction scratch of the alchemical slabs (slots 0 and 2: the two kicks of this program)
    __shared__ double s_aself[FUSED ? 6 : 1][64];    // ... the alchemical atoms' pair force, slots 0 and 2
    __shared__ double s_esum[FUSED ? K2_NP : 1];     // ... the alchemical kernel's energy sums
    // FUSED (host: fin_fusable -- ONE nonbonded slab, at most STEP_FENT_LDS bonded entries): the bonded entries of the chain are
    // staged by the whole block in one round trip (the few threads that hold the solute's clusters add a few dozen entries per
    // atom: from memory that was sixteen dependent round trips on the critical path of wave 0 while every other thread had none),
    // and every load of the top of the kernel is REQUESTED before any is waited for -- in a batch a round trip to memory takes
    // ~7,000 cycles (512 workgroups gather at once).  Same values added in the same order as k_finalize (fin_atom_base).
    __shared__ double s_fent[FUSED ? 3 * STEP_FENT_LDS : 1];
    const int jc_alch = (FUSED && A.n_alch > 0) ? *A.jcount_alch : 0;   // (asked for here: on its way while the entries are staged)
    STEP_STAMP(10);
    if (FUSED) {
        for (int e = tid; e < 3 * A.n_entries; e += (int)blockDim.x) s_fent[e] = A.fent[e];
        __syncthreads();
    }
    STEP_STAMP(11);
    double FP[FUSED ? 4 : 1][3];
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const int i = max(R.atoms[a], 0), al = R.alch[a], alc = max(al, 0);
        C.id[a] = R.atoms[a]; C.al[a] = al; C.w[a] = R.w[a];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            C.x[a][k] = A.x[k][i]; C.v[a][k] = A.v[k][i];
            if (!LATE) { XB[a][k] = A.xbuild[k][i]; XP[a][k] = pruned ? A.xprune[k][max(R.islot[a], 0)] : 0u; }
            if (FUSED) {
                const bool env = R.islot[a] >= 0 && A.n_alch > 0 && R.atoms[a] >= 0;
                FA[a][k] = env ? A.fJ[(size_t)(0 + k) * A.n + R.sorted[a]] : 0.0; FB[a][k] = env ? A.fJ[(size_t)(6 + k) * A.n + R.sorted[a]] : 0.0;   // (the lambda-independent part is added below)
                FP[a][k] = (R.atoms[a] >= 0 && R.islot[a] >= 0) ? A.fpart[(size_t)k * A.n_islots + R.islot[a]] : 0.0;
            } else {
                const double fa0 = A.ftot[(size_t)((al >= 0 ? 0 : 0) + k) * A.n + i];
                const double fb0 = A.ftot[(size_t)((al >= 0 ? 0 : 6) + k) * A.n + i];
                const double sa = A.alch_self[(0 + k) * 64 + alc], sb = A.alch_self[(6 + k) * 64 + alc];
                FA[a][k] = fa0 + (al >= 0 ? sa : 0.0);
                FB[a][k] = fb0 + (al >= 0 ? sb : 0.0);
            }
            if (!LATE) G0[a][k] = pre_noise ? A.noise[(size_t)(nd0 * 3 + k) * A.n_mobile + R.mobile[a]] : 0.0;
        }
    }
    STEP_STAMP(1);
    // (placed behind the gathers above: their loads are in flight while the block sums the alchemical slabs)
    if (FUSED && A.n_alch > 0) {   // (uniform) what the last 10 blocks of k_finalize do, the same sums in the same order (fin_alch_self, fin_energy_sums),
        // with every slab's loads in flight together and ONE barrier for all of them
        const int nb_env = k2_env_blocks(jc_alch, A.PA, A.k2_jiter);
        const int PA = A.PA, NG = 256 / PA;
        // (a 128-thread block -- BluesTuning.pack_clusters -- plays both halves of the 256 threads k_finalize sums with: same slabs
        // per partial sum, same order of the partial sums)
        for (int vt = tid; vt < 256; vt += (int)blockDim.x) {
            const int aa = vt & (PA - 1), gg = vt / PA;
            double sv[6], s0[6], sx[6];
#pragma unroll
            for (int q6 = 0; q6 < 6; q6++) {   // (the thread's first slab and the alchemical x alchemical slab of all six sums requested together)
                const int q = q6 < 3 ? q6 : q6 + 3;
                s0[q6] = gg < nb_env ? A.self_part[((size_t)gg * 9 + q) * PA + aa] : 0.0;
                sx[q6] = gg == 0 ? A.self_part[((size_t)A.k2_nblocks_env * 9 + q) * PA + aa] : 0.0;
            }
#pragma unroll
            for (int q6 = 0; q6 < 6; q6++) {
                const int q = q6 < 3 ? q6 : q6 + 3;
                double sq = 0.0;
                if (gg < nb_env) sq += s0[q6];
                for (int b = gg + NG; b < nb_env; b += NG) sq += A.self_part[((size_t)b * 9 + q) * PA + aa];   // (the dense kernel leaves one slab)
                if (gg == 0) sq += sx[q6];
                sv[q6] = sq;
            }
#pragma unroll
            for (int q6 = 0; q6 < 6; q6++) s_fin[q6][vt] = sv[q6];
        }
        fin_energy_sums(A.e_part, nb_env, A.k2_nblocks_env, s_esum);   // (ends with a barrier: s_fin is published too)
        if (tid < 64) {
#pragma unroll
            for (int q6 = 0; q6 < 6; q6++) {
                double t = 0.0;
                if (tid < A.n_alch) for (int u = 0; u < NG; u++) t += s_fin[q6][u * PA + tid];   // fixed order
                s_aself[q6][tid] = t;
            }
        }
        __syncthreads();
    }
    if (FUSED) {
#pragma unroll
        for (int a = 0; a < 4; a++) {
            // fin_atom_base's sum: 0 + the slab, the atom's bonded entries in order, reciprocal space
            double fb[3] = {0.0, 0.0, 0.0};
            if (R.atoms[a] >= 0) {
                if (R.islot[a] >= 0) { fb[0] += FP[a][0]; fb[1] += FP[a][1]; fb[2] += FP[a][2]; }
                for (int e = R.e0[a]; e < R.e1[a]; e++) { fb[0] += s_fent[e]; fb[1] += s_fent[A.n_entries + e]; fb[2] += s_fent[2 * A.n_entries + e]; }
                if (A.frec) { const int i = R.atoms[a]; fb[0] += A.frec[i]; fb[1] += A.frec[(size_t)A.n + i]; fb[2] += A.frec[2 * (size_t)A.n + i]; }
            }
#pragma unroll
            for (int k = 0; k < 3; k++) { FA[a][k] = fb[k] + FA[a][k]; FB[a][k] = fb[k] + FB[a][k]; }
        }
    }
    if (FUSED && A.n_alch > 0) {
#pragma unroll
        for (int a = 0; a < 4; a++) if (R.alch[a] >= 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) { FA[a][k] += s_aself[k][R.alch[a]]; FB[a][k] += s_aself[3 + k][R.alch[a]]; }
        }
    }
    STEP_STAMP(2);
    double e_sl[3] = {0.0, 0.0, 0.0};
    if (FUSED) {
        if (A.n_alch > 0) { e_sl[0] = A.fin_le[0] * s_esum[0] + s_esum[1]; e_sl[1] = A.fin_le[1] * s_esum[0] + s_esum[2]; e_sl[2] = A.fin_le[2] * s_esum[0] + s_esum[3]; }
        else if (blockIdx.x == 0 && tid == 0) { e_sl[0] = A.acc->e_slot[0]; e_sl[1] = A.acc->e_slot[1]; e_sl[2] = A.acc->e_slot[2]; }   // (no alchemical atoms: nobody updates them)
        if (blockIdx.x == 0 && tid == 0 && A.n_alch > 0) { A.acc->e_slot[0] = e_sl[0]; A.acc->e_slot[1] = e_sl[1]; A.acc->e_slot[2] = e_sl[2]; }
    }
    double p_pre[3] = {0.0, 0.0, 0.0};   // FUSED + CM: this thread's share of sum m v + hV sum F(slot 0), before anything is kicked
    if (FUSED && CM && active) {
#pragma unroll
        for (int a = 0; a < 4; a++) if (a < R.na) {
            const double m = A.mass[max(R.atoms[a], 0)];
#pragma unroll
            for (int k = 0; k < 3; k++) p_pre[k] += m * C.v[a][k] + A.hV * FA[a][k];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) C.d2[c] = R.dist[c] * R.dist[c];
    bool ok = true;
    // V(slot 0), finishing the previous step
    if (active) {
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int k = 0; k < 3; k++) C.v[a][k] += A.hV * FA[a][k] * C.w[a];
        rattle_sel<FUSED>(C, A.tol, A);
    }
    STEP_STAMP(3);
    if (blockIdx.x == 0 && tid == 0) {
        const double dE = FUSED ? e_sl[1] - e_sl[0] : A.acc->e_slot[1] - A.acc->e_slot[0];
        A.acc->protocol_work += dE; A.acc->dE_last = dE;
        if (A.work_trace) A.work_trace[A.trace_index] = A.acc->protocol_work;
    }
    if (CM) {
        // total momentum after the finishing kick = sum m v (before this launch) + hV * sum F(slot 0): both sums were left
        // behind per block by k_finalize, so no grid-wide reduction of the kicked velocities is needed here
        // (FUSED: the one block of the chain holds every atom; the sums are formed here, in thread order)
        double p[3] = {0.0, 0.0, 0.0};
        if (FUSED) { p[0] = p_pre[0]; p[1] = p_pre[1]; p[2] = p_pre[2]; }
        else for (int q = tid; q < A.n_mom; q += blockDim.x)
            for (int k = 0; k < 3; k++) p[k] += A.mom_part[(size_t)q * 6 + k] + A.hV * A.mom_part[(size_t)q * 6 + 3 + k];
        for (int k = 0; k < 3; k++) { p[k] = wave_sum(p[k]); if ((tid & 63) == 0) s_red[tid >> 6][k] = p[k]; }
        __syncthreads();
        if (tid < 3) { double s = 0.0; for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += s_red[w][tid]; s_cm[tid] = s / A.total_mass; }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int a = 0; a < 4; a++) if (a < C.na) for (int k = 0; k < 3; k++) C.v[a][k] -= s_cm[k];
        }
    }
    STEP_STAMP(4);
    if (blockIdx.x == 0 && tid == 0) {
        const double dE = FUSED ? e_sl[2] - e_sl[1] : A.acc->e_slot[2] - A.acc->e_slot[1];
        A.acc->protocol_work += dE; A.acc->dE_last = dE;
    }
    bool need_rebuild = false, bad = false;
    if (active) {
        // V(slot 2)
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int k = 0; k < 3; k++) C.v[a][k] += A.hV * FB[a][k] * C.w[a];
        rattle_sel<FUSED>(C, A.tol, A);
        STEP_STAMP(5);
        for (int half = 0; half < 2; half++) {
            // R
            double xr[4][3], x1[4][3];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int k = 0; k < 3; k++) { xr[a][k] = C.x[a][k]; if (a < C.na) C.x[a][k] += A.hR * C.v[a][k]; x1[a][k] = C.x[a][k]; }
            ok &= shake_sel<FUSED>(C, xr, A.tol, A);
#pragma unroll
            for (int a = 0; a < 4; a++) if (a < C.na) for (int k = 0; k < 3; k++) C.v[a][k] += (C.x[a][k] - x1[a][k]) * A.inv_hR;
            rattle_sel<FUSED>(C, A.tol, A);
            STEP_STAMP(6 + 2 * half);
            if (half == 0) {
                // O
                double GL[LATE ? 4 : 1][3];
                if (LATE && pre_noise) {   // (all twelve requested before the first is used)
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int k = 0; k < 3; k++) GL[a][k] = a < C.na ? A.noise[(size_t)(nd0 * 3 + k) * A.n_mobile + R.mobile[a]] : 0.0;
                }
#pragma unroll
                for (int a = 0; a < 4; a++) if (a < C.na) {
                    double g[3];
                    if (pre_noise) { for (int k = 0; k < 3; k++) g[k] = LATE ? GL[a][k] : G0[a][k]; }
                    else gaussians3(A.seed, A.stream, A.draw_base, (unsigned)C.id[a], g);
                    const double s = sqrt(A.kT * C.w[a]);
#pragma unroll
                    for (int k = 0; k < 3; k++) C.v[a][k] = A.aO * C.v[a][k] + A.bO * s * g[k];
                }
                rattle_sel<FUSED>(C, A.tol, A);
                STEP_STAMP(7);
            }
        }
        // write back, refresh the image, list validity
        double XL[LATE ? 4 : 1][3]; unsigned PL[LATE ? 4 : 1][3];
        if (LATE) {   // (the reference positions of both list checks, requested together)
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    XL[a][k] = a < C.na ? A.xbuild[k][max(C.id[a], 0)] : 0.0;
                    PL[a][k] = (pruned && a < C.na && R.islot[a] >= 0) ? A.xprune[k][R.islot[a]] : 0u;
                }
        }
#pragma unroll
        for (int a = 0; a < 4; a++) if (a < C.na) {
            const int i = C.id[a];
            double d2 = 0.0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                A.v[k][i] = C.v[a][k]; A.x[k][i] = C.x[a][k];
                bad |= !(C.x[a][k] == C.x[a][k]) || !(C.v[a][k] == C.v[a][k]);
                const double d = C.x[a][k] - (LATE ? XL[a][k] : XB[a][k]); d2 += d * d;
            }
            need_rebuild |= d2 > A.half_skin2;
            const int s = R.sorted[a];
            if (A.img_f) {
                unsigned u[3]; to_fixed32(C.x[a], A.box, u); A.img_f[s].x = u[0]; A.img_f[s].y = u[1]; A.img_f[s].z = u[2];
                if (pruned && R.islot[a] >= 0) {
                    float p2 = 0.0f;
#pragma unroll
                    for (int k = 0; k < 3; k++) { const float e = (float)(int)(u[k] - (LATE ? PL[a][k] : XP[a][k])) * A.fscale[k]; p2 += e * e; }
                    if (p2 > A.prune_trig2) A.pneed[R.islot[a]] = 1;
                }
            }
            else { unsigned long long u[3]; to_fixed(C.x[a], A.box, u); A.img_d[s].x = u[0]; A.img_d[s].y = u[1]; A.img_d[s].z = u[2]; }
        }
        if (need_rebuild) { A.flags->req_gen = A.flags->list_gen + 1; if (A.batch_req) *A.batch_req = 1; }
        if (bad) A.flags->nan_flag = 1;
        if (!ok) A.flags->constraint_fail = 1;
    }
    STEP_STAMP(9);
}

template <bool CM, bool FUSED = false>
__global__ void __launch_bounds__(256) k_step_default(IntArgs A) { step_default_body<CM, FUSED>(A); }

// ---- straight-line specialisation of one MD-leg step: [CM] L   (OpenMM LangevinIntegrator)
template <bool CM>
__device__ __forceinline__ void step_md_body(IntArgs& A) {
    const int tid = threadIdx.x;
    const int cl = blockIdx.x * blockDim.x + tid;
    __shared__ double s_red[4][4];
    __shared__ double s_cm[3];
    const bool active = cl < A.n_clusters;
    ClusterRec R;
    if (active) R = A.recs[cl];
    else { for (int a = 0; a < 4; a++) { R.atoms[a] = -1; R.alch[a] = -1; R.mobile[a] = 0; R.sorted[a] = 0; R.islot[a] = -1; R.w[a] = 0.0; } R.type = 0; R.nc = 0; R.na = 0; R.dist[0] = R.dist[1] = R.dist[2] = 0.0; }
    Cluster C;
    C.na = R.na; C.nc = R.nc; C.type = R.type;
    const unsigned nd0 = A.draw_base - A.noise_draw_base;
    const bool pre_noise = nd0 < (unsigned)A.n_noise;
    double F[4][3], G0[4][3], XB[4][3], x0[4][3];
    unsigned XP[4][3];
    const bool pruned = A.pneed != nullptr;   // (uniform)
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const int i = max(R.atoms[a], 0), al = R.alch[a], alc = max(al, 0);
        C.id[a] = R.atoms[a]; C.al[a] = al; C.w[a] = R.w[a];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            C.x[a][k] = A.x[k][i]; C.v[a][k] = A.v[k][i]; XB[a][k] = A.xbuild[k][i]; XP[a][k] = pruned ? A.xprune[k][max(R.islot[a], 0)] : 0u;
            F[a][k] = A.ftot[(size_t)k * A.n + i] + (al >= 0 ? A.alch_self[k * 64 + alc] : 0.0);
            G0[a][k] = pre_noise ? A.noise[(size_t)(nd0 * 3 + k) * A.n_mobile + R.mobile[a]] : 0.0;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) C.d2[c] = R.dist[c] * R.dist[c];
    if (CM) {
        double p[3] = {0.0, 0.0, 0.0};
        for (int q = tid; q < A.n_mom; q += blockDim.x) for (int k = 0; k < 3; k++) p[k] += A.mom_part[(size_t)q * 6 + k];
        for (int k = 0; k < 3; k++) { p[k] = wave_sum(p[k]); if ((tid & 63) == 0) s_red[tid >> 6][k] = p[k]; }
        __syncthreads();
        if (tid < 3) { double s = 0.0; for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += s_red[w][tid]; s_cm[tid] = s / A.total_mass; }
        __syncthreads();
    }
    if (!active) return;
    bool ok = true, need_rebuild = false, bad = false;
#pragma unroll
    for (int a = 0; a < 4; a++) {
#pragma unroll
        for (int k = 0; k < 3; k++) x0[a][k] = C.x[a][k];
        if (a < C.na) {
            double g[3];
            if (pre_noise) { for (int k = 0; k < 3; k++) g[k] = G0[a][k]; }
            else gaussians3(A.seed, A.stream, A.draw_base, (unsigned)C.id[a], g);
            const double sd = sqrt(A.kT * C.w[a]);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const double v0 = C.v[a][k] - (CM ? s_cm[k] : 0.0);
                C.v[a][k] = A.aL * v0 + A.fsL * C.w[a] * F[a][k] + A.nsL * sd * g[k];
                C.x[a][k] += A.dtL * C.v[a][k];
            }
        }
    }
    ok &= shake(C, x0, A.tol, A);
    const double inv_dt = 1.0 / A.dtL;
#pragma unroll
    for (int a = 0; a < 4; a++) if (a < C.na) {
        const int i = C.id[a];
        double d2 = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            C.v[a][k] = (C.x[a][k] - x0[a][k]) * inv_dt;
            A.v[k][i] = C.v[a][k]; A.x[k][i] = C.x[a][k];
            bad |= !(C.x[a][k] == C.x[a][k]) || !(C.v[a][k] == C.v[a][k]);
            const double d = C.x[a][k] - XB[a][k]; d2 += d * d;
        }
        need_rebuild |= d2 > A.half_skin2;
        const int s = R.sorted[a];
        if (A.img_f) {
            unsigned u[3]; to_fixed32(C.x[a], A.box, u); A.img_f[s].x = u[0]; A.img_f[s].y = u[1]; A.img_f[s].z = u[2];
            if (pruned && R.islot[a] >= 0) {
                float p2 = 0.0f;
#pragma unroll
                for (int k = 0; k < 3; k++) { const float e = (float)(int)(u[k] - XP[a][k]) * A.fscale[k]; p2 += e * e; }
                if (p2 > A.prune_trig2) A.pneed[R.islot[a]] = 1;
            }
        }
        else { unsigned long long u[3]; to_fixed(C.x[a], A.box, u); A.img_d[s].x = u[0]; A.img_d[s].y = u[1]; A.img_d[s].z = u[2]; }
    }
    if (need_rebuild) { A.flags->req_gen = A.flags->list_gen + 1; if (A.batch_req) *A.batch_req = 1; }
    if (bad) A.flags->nan_flag = 1;
    if (!ok) A.flags->constraint_fail = 1;
}

template <bool CM>
__global__ void __launch_bounds__(256) k_step_md(IntArgs A) { step_md_body<CM>(A); }

// the same role dispatch around the sub-tile throughput kernel (large i-sets): the alchemical and bonded blocks are
// queued behind the nonbonded ones and fill CUs as those drain, instead of costing two more launches
template <int IW>
__global__ void __launch_bounds__(256) k_forces_fused_sub(NbArgs<float> a, NbConst<float> c, const AtomF* __restrict__ img,
                                                          AlchArgs A, BondedArgs B, int nb1, int nb2) {
    const int b = blockIdx.x;
    if (b < nb2) { alchemical_body<true>(A, b); return; }             // the longest-latency blocks first
    if (b < nb2 + nb1) { nonbonded_sub_body<false, IW>(a, c, img, b - nb2); return; }
    bonded_entries_body(B, b - nb1 - nb2, 256);
}
