// kernels_bonded.h -- harmonic bonds/angles, periodic torsions, env-env 1-4 exceptions and
// the positional restraint (K3/K4), fp64, gather form.
//
// OpenMM's HarmonicBondForce / HarmonicAngleForce / PeriodicTorsionForce / NonbondedForce
// exceptions / CustomExternalForce('k_restr*periodicdistance(x,y,z,x0,y0,z0)^2',
// reference blues/simulation.py:347) scatter each term's force to its atoms.  Here every
// MOBILE atom that takes part in a term owns a row of (term, role) entries and recomputes the
// terms it belongs to, keeping only its own component: 2-4x redundant flops on a few
// thousand cheap terms, but no atomics and a fixed summation order.
#pragma once
#include "device_common.h"

enum { T_BOND = 0, T_ANGLE = 1, T_TORSION = 2, T_EXC = 3, T_RESTR = 4, T_EWEX = 5, T_NTYPES = 6 };   // T_EWEX: Ewald correction of an excluded pair (BLUES_NB_PME)

struct BondedArgs {
    int n_rows;
    const int* row_atom;    // caller index of the row's atom
    const int* row_start;   // [n_rows+1]
    const int* ent_type;    // per entry
    const int* ent_term;
    const int* ent_role;
    // all terms (caller indices)
    int n_terms[T_NTYPES];
    const int* atoms[T_NTYPES];      // 2,3,4,2,1,2 indices per term
    const double* params[T_NTYPES];  // 2,2,3,3,3,1 doubles per term (restraint: x0,y0,z0; Ewald correction: q_i q_j)
    double restr_k;
    double ewald_alpha;
    const double* x[3];
    Box3 box;
    int periodic;
    double* fent;    // [3][n_entries] per-entry forces
    int n_entries;
    int n;
    double* epart;   // [nblocks][T_NTYPES] (energy kernel)
    // Gaussian noise for the next O substeps, generated here (extra blocks of the same launch) so that the
    // log/sqrt/sincos of the Box-Muller transform are off the integrator's serial per-cluster path
    int n_mobile, n_noise; const int* mobile_atoms; double* noise;  // noise[(d*3+k)*n_mobile + m]
    unsigned long long seed; unsigned stream, draw_base;
    int n_entry_blocks;
    const DevCtrl* ctrl;
};

__device__ inline void mi3(const BondedArgs& B, double d[3]) {
    if (B.periodic) for (int k = 0; k < 3; k++) d[k] = min_image_d(d[k], B.box.L[k], B.box.invL[k]);
}
__device__ inline void cross3(const double a[3], const double b[3], double c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ inline double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// energy of term (type, idx); if role >= 0 also the force on the atom in that role
__device__ inline double bonded_term(const BondedArgs& B, int type, int idx, int role, double F[3]) {
    F[0] = F[1] = F[2] = 0.0;
    if (type == T_BOND) {
        const int i = B.atoms[T_BOND][2 * idx], j = B.atoms[T_BOND][2 * idx + 1];
        const double r0 = B.params[T_BOND][2 * idx], k = B.params[T_BOND][2 * idx + 1];
        double d[3] = {B.x[0][i] - B.x[0][j], B.x[1][i] - B.x[1][j], B.x[2][i] - B.x[2][j]};
        mi3(B, d);
        const double r = sqrt(dot3(d, d)), dr = r - r0, fs = -k * dr / r;
        const double sgn = role == 0 ? 1.0 : -1.0;
        if (role >= 0) for (int c = 0; c < 3; c++) F[c] = sgn * fs * d[c];
        return 0.5 * k * dr * dr;
    }
    if (type == T_ANGLE) {
        const int i = B.atoms[T_ANGLE][3 * idx], j = B.atoms[T_ANGLE][3 * idx + 1], k_ = B.atoms[T_ANGLE][3 * idx + 2];
        const double t0 = B.params[T_ANGLE][2 * idx], kk = B.params[T_ANGLE][2 * idx + 1];
        double u[3] = {B.x[0][i] - B.x[0][j], B.x[1][i] - B.x[1][j], B.x[2][i] - B.x[2][j]};
        double w[3] = {B.x[0][k_] - B.x[0][j], B.x[1][k_] - B.x[1][j], B.x[2][k_] - B.x[2][j]};
        mi3(B, u); mi3(B, w);
        const double lu = sqrt(dot3(u, u)), lw = sqrt(dot3(w, w));
        double c = dot3(u, w) / (lu * lw);
        c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
        const double th = acos(c), dth = th - t0;
        if (role >= 0) {
            double s = sqrt(1.0 - c * c); if (s < 1e-12) s = 1e-12;
            const double pre = kk * dth / s;
            for (int m = 0; m < 3; m++) {
                const double fi = pre * (w[m] / (lu * lw) - c * u[m] / (lu * lu));
                const double fk = pre * (u[m] / (lu * lw) - c * w[m] / (lw * lw));
                F[m] = role == 0 ? fi : (role == 2 ? fk : -(fi + fk));
            }
        }
        return 0.5 * kk * dth * dth;
    }
    if (type == T_TORSION) {
        const int* q = B.atoms[T_TORSION] + 4 * idx;
        const double per = B.params[T_TORSION][3 * idx], ph = B.params[T_TORSION][3 * idx + 1], kk = B.params[T_TORSION][3 * idx + 2];
        double rij[3], rkj[3], rkl[3], m[3], nn[3];
        for (int c = 0; c < 3; c++) { rij[c] = B.x[c][q[0]] - B.x[c][q[1]]; rkj[c] = B.x[c][q[2]] - B.x[c][q[1]]; rkl[c] = B.x[c][q[2]] - B.x[c][q[3]]; }
        mi3(B, rij); mi3(B, rkj); mi3(B, rkl);
        cross3(rij, rkj, m); cross3(rkj, rkl, nn);
        const double m2 = dot3(m, m), n2 = dot3(nn, nn), lkj2 = dot3(rkj, rkj), lkj = sqrt(lkj2);
        double cs = dot3(m, nn) / sqrt(m2 * n2);
        cs = cs > 1.0 ? 1.0 : (cs < -1.0 ? -1.0 : cs);
        double phi = acos(cs); if (dot3(rij, nn) < 0.0) phi = -phi;
        if (role >= 0) {
            const double ddphi = -kk * per * sin(per * phi - ph);
            const double p = dot3(rij, rkj) / lkj2, qq = dot3(rkl, rkj) / lkj2;
            for (int c = 0; c < 3; c++) {
                const double fi = -ddphi * lkj / m2 * m[c], fl = ddphi * lkj / n2 * nn[c];
                const double sv = p * fi - qq * fl;
                F[c] = role == 0 ? fi : (role == 1 ? -(fi - sv) : (role == 2 ? -(fl + sv) : fl));
            }
        }
        return kk * (1.0 + cos(per * phi - ph));
    }
    if (type == T_EXC) {  // env-env 1-4: plain LJ + bare Coulomb, no cutoff
        const int i = B.atoms[T_EXC][2 * idx], j = B.atoms[T_EXC][2 * idx + 1];
        const double qq = B.params[T_EXC][3 * idx], sig = B.params[T_EXC][3 * idx + 1], eps = B.params[T_EXC][3 * idx + 2];
        double d[3] = {B.x[0][i] - B.x[0][j], B.x[1][i] - B.x[1][j], B.x[2][i] - B.x[2][j]};
        mi3(B, d);
        const double r2 = dot3(d, d);
        double fs, fc;
        const double e = plain_lj_d(r2, sig, eps, &fs) + coulomb_d(r2, qq, 0.0, false, &fc);
        const double sgn = role == 0 ? 1.0 : -1.0;
        if (role >= 0) for (int c = 0; c < 3; c++) F[c] = sgn * (fs + fc) * d[c];
        return e;
    }
    if (type == T_EWEX) {  // excluded pair under PME: its reciprocal-space interaction is taken out, -ONE_4PI_EPS0 q_i q_j erf(alpha r) / r
        const int i = B.atoms[T_EWEX][2 * idx], j = B.atoms[T_EWEX][2 * idx + 1];
        const double pre = ONE_4PI_EPS0 * B.params[T_EWEX][idx], al = B.ewald_alpha;
        double d[3] = {B.x[0][i] - B.x[0][j], B.x[1][i] - B.x[1][j], B.x[2][i] - B.x[2][j]};
        mi3(B, d);
        const double r2 = dot3(d, d), r = sqrt(r2), er = erf(al * r);
        if (role >= 0) {
            const double dEdr = -pre * (TWO_OVER_SQRT_PI * al * exp(-al * al * r2) / r - er / r2);
            const double fs = -dEdr / r, sgn = role == 0 ? 1.0 : -1.0;
            for (int c = 0; c < 3; c++) F[c] = sgn * fs * d[c];
        }
        return -pre * er / r;
    }
    // T_RESTR
    const int i = B.atoms[T_RESTR][idx];
    double d[3] = {B.x[0][i] - B.params[T_RESTR][3 * idx], B.x[1][i] - B.params[T_RESTR][3 * idx + 1], B.x[2][i] - B.params[T_RESTR][3 * idx + 2]};
    mi3(B, d);
    if (role >= 0) for (int c = 0; c < 3; c++) F[c] = -2.0 * B.restr_k * d[c];
    return B.restr_k * dot3(d, d);
}

// one thread per (row, entry): the force of one term on one of its mobile atoms -> fent[3][n_entries];
// blocks past n_entry_blocks draw the N(0,1) numbers of the coming O substeps (counter-based, so they can be
// produced before the velocities they will be applied to exist).
__device__ __forceinline__ void bonded_entries_body(const BondedArgs& B, const int block_id, const int nthreads) {
    if (block_id >= B.n_entry_blocks) {
        const int g = (block_id - B.n_entry_blocks) * nthreads + threadIdx.x;
        if (g >= B.n_mobile * B.n_noise) return;
        const int d = g / B.n_mobile, m = g - d * B.n_mobile;
        double z[3];
        const unsigned base = B.ctrl ? B.ctrl->draw0 + (unsigned)B.ctrl->kpass : B.draw_base;
        gaussians3(B.seed, B.stream, base + (unsigned)d, (unsigned)B.mobile_atoms[m], z);
        for (int k = 0; k < 3; k++) B.noise[(size_t)(d * 3 + k) * B.n_mobile + m] = z[k];
        return;
    }
    const int e = block_id * nthreads + threadIdx.x;
    if (e >= B.n_entries) return;
    double F[3];
    bonded_term(B, B.ent_type[e], B.ent_term[e], B.ent_role[e], F);
    B.fent[e] = F[0]; B.fent[B.n_entries + e] = F[1]; B.fent[2 * B.n_entries + e] = F[2];
}

__global__ void __launch_bounds__(128) k_bonded_entries(BondedArgs B) { bonded_entries_body(B, blockIdx.x, 128); }

// energy of every term (frozen ones included): per-block partial sums per type
__device__ __forceinline__ void bonded_energy_body(const BondedArgs& B) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    double e[T_NTYPES] = {0, 0, 0, 0, 0, 0};
    int base = 0;
    for (int ty = 0; ty < T_NTYPES; ty++) {
        const int idx = gid - base;
        if (idx >= 0 && idx < B.n_terms[ty]) { double F[3]; e[ty] = bonded_term(B, ty, idx, -1, F); }
        base += B.n_terms[ty];
    }
    __shared__ double s[4][T_NTYPES];
    for (int ty = 0; ty < T_NTYPES; ty++) { double v = wave_sum(e[ty]); if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6][ty] = v; }
    __syncthreads();
    if (threadIdx.x < T_NTYPES) B.epart[blockIdx.x * T_NTYPES + threadIdx.x] = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
}

__global__ void __launch_bounds__(256) k_bonded_energy(BondedArgs B) { bonded_energy_body(B); }
