/*
 * blues_oracle.c -- CPU fp64 restatement of the BLUES NCMC switching path.
 * TEST INFRASTRUCTURE ONLY (see blues_oracle.h for the parity-pinning status).
 *
 * What is restated, and from where:
 *   step program ............ reference blues/integrators.py:159-231
 *                             (AlchemicalExternalLangevinIntegrator._add_integrator_steps,
 *                              _add_alchemical_perturbation_step); SURVEY.md Appendix A
 *   globals / reset ......... blues/integrators.py:129-145, 240-249
 *   _get_prop_lambda ........ blues/integrators.py:147-157
 *   default lambda functions  blues/simulation.py:654-659
 *   calculateNCMCSteps ...... blues/utils.py:89-145
 *   R / V / O substeps ...... openmmtools 0.15.0 LangevinIntegrator (published forms,
 *                             SURVEY.md Appendix A/B) called from blues/integrators.py:192,200
 *   alchemical potential .... openmmtools 0.15.0 AbsoluteAlchemicalFactory with the
 *                             arguments at blues/simulation.py:225-236, 300-316
 *   forces .................. OpenMM NonbondedForce (PME direct space), HarmonicBondForce,
 *                             HarmonicAngleForce, PeriodicTorsionForce, CustomExternalForce
 *                             (blues/simulation.py:347) -- published functional forms
 *   constraints ............. SHAKE / RATTLE to the tolerance of blues/integrators.py:104
 *   CMMotionRemover ......... OpenMM, enabled by removeCMMotion (examples/rotmove_cuda.yml:24)
 *
 * Like OpenMM's Reference platform this evaluates energies and forces of ALL
 * atoms at every distinct (x, lambda) the step program touches (three per
 * step for "H V R O R V H"), in double precision, on one thread.
 */
#include "blues_oracle.h"

#include <math.h>
#ifdef ORC_OPENMP
#include <omp.h>
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ONE_4PI_EPS0 138.935456
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define KB_KJ 0.0083144626 /* kJ/mol/K (k_B * N_A) */
#define TWO_OVER_SQRT_PI 1.1283791670955126
#define MAX_CLUSTER_ATOMS 16
#define MAX_SPLIT 64

static char g_err[512];
const char *orc_last_error(void) { return g_err; }

typedef struct {
    double lambda, step, lambda_step, n_lambda_steps, nsteps, protocol_work, shadow_work;
    double perturbed_pe, unperturbed_pe, first_step, nprop, prop, prop_lambda_min, prop_lambda_max;
    double Eold, Enew, heat, debug;
    double lambda_sterics, lambda_electrostatics;
    /* switching.py integrators (energies in kJ/mol; the Python mirror divides by kT) */
    double Epert, total_work, initial_energy, final_energy, accept, naccept, ntrials;
} Globals;

struct Oracle {
    uint32_t ghmc_draw;   /* Metropolis uniforms drawn so far by the GHMC integrator: the Philox counter (never reset: ntrials is a statistic) */
    int n;
    double box[3];
    double *mass, *charge, *sigma, *eps;
    int n_excl;
    int *excl_start, *excl_list; /* CSR, both directions */
    int n_exc;
    int *exc_atoms;
    double *exc_params;
    int n_bonds, *bond_atoms;
    double *bond_params;
    int n_angles, *angle_atoms;
    double *angle_params;
    int n_tors, *tors_atoms;
    double *tors_params;
    int n_cons, *cons_atoms;
    double *cons_dist;
    int n_clusters, *cluster_start, *cluster_cons; /* constraints grouped by connected component */
    int n_alch, *alch_atoms;
    unsigned char *is_alch;
    int n_restr, *restr_atoms;
    double *restr_x0, restr_k;
    int nb_method;
    double cutoff, alpha_ewald, sc_alpha;
    int annihilate_elec, annihilate_sterics, remove_cm;
    /* integrator */
    double dt, temperature, gamma, kT, tol;
    int nsteps, nprop_i, n_lambda_steps;
    char split[MAX_SPLIT];
    int n_split, n_R, n_V, n_O, n_H;
    int switch_mode, psteps;     /* BLUES_SWITCH_* (reference blues/switching.py), steps_per_propagation */
    double *tab_ls, *tab_le;
    uint64_t seed;
    uint32_t replica, draw;
    Globals g;
    /* state */
    double *x, *v, *xref, *f;
    /* energy cache (OpenMM re-uses an evaluation while x and parameters are unchanged) */
    long xver, cache_xver;
    double cache_ls, cache_le, cache_E;
    int cache_valid;
    long n_evals;
    /* oracle-only extras for the reference's ethylene known-answer system (tests/data/ethylene_system.xml) */
    int custom_pair_mode;       /* 1: CustomNonbondedForce 'q/(r^2) + 4*epsilon*((sigma/r)^12-(sigma/r)^6)' between set1 (non-alchemical)
                                   and set2 (alchemical) atoms, sigma scaled by lambda_sterics, epsilon by lambda_electrostatics */
    int n_centroid; int cb_n1[4], cb_n2[4]; int cb_i1[4][8], cb_i2[4][8]; double cb_w1[4][8], cb_w2[4][8], cb_k[4];
    /* cell list scratch */
    int *cell_head, *cell_next;
    int ncell[3];
    /* reciprocal space (BLUES_NB_PME) */
    int pme_K[3], pme_order, disp_corr;
    double *pme_mod[3];          /* B-spline moduli per axis */
    double *pme_q, *pme_re, *pme_im, *pme_tr, *pme_ti;   /* charge mesh, its transform, scratch */
    double *pme_cos[3], *pme_sin[3];
};

static void pme_setup(Oracle *o);
static void reciprocal_space(Oracle *o, double *F, double *T);

/* number of threads of the OpenMP pair loop (no-op in the serial build).  The environment variable is read when the OpenMP runtime
 * initialises, which in a process that imported torch has long happened. */
void orc_set_threads(int n) {
#ifdef ORC_OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------ RNG */
static inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

void orc_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; r++) {
        uint32_t hi0 = mulhi32(M0, c0), lo0 = M0 * c0;
        uint32_t hi1 = mulhi32(M1, c2), lo1 = M1 * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* three N(0,1) numbers for (seed, stream, draw, atom): counter-based, identical on CPU and GPU */
void orc_gaussians(uint64_t seed, uint32_t stream, uint32_t draw, uint32_t atom, double out[3]) {
    uint32_t r[4];
    orc_philox4x32(atom, draw, stream, 0x424C5545u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const double S = 2.3283064365386963e-10; /* 2^-32 */
    double u1 = ((double)r[0] + 0.5) * S, u2 = ((double)r[1] + 0.5) * S;
    double u3 = ((double)r[2] + 0.5) * S, u4 = ((double)r[3] + 0.5) * S;
    double ra = sqrt(-2.0 * log(u1)), rb = sqrt(-2.0 * log(u3));
    out[0] = ra * cos(2.0 * M_PI * u2);
    out[1] = ra * sin(2.0 * M_PI * u2);
    out[2] = rb * cos(2.0 * M_PI * u4);
}

/* ------------------------------------------------------------------ schedule helpers */
/* Lepton step(x): 0 if x < 0, 1 otherwise */
static inline double lep_step(double x) { return x < 0.0 ? 0.0 : 1.0; }

/* 'min(1, (1/0.3)*abs(lambda-0.5))'  -- reference blues/simulation.py:655-656 */
double orc_default_lambda_sterics(double l) { return fmin(1.0, (1.0 / 0.3) * fabs(l - 0.5)); }
/* 'step(0.2-lambda) - 1/0.2*lambda*step(0.2-lambda) + 1/0.2*(lambda-0.8)*step(lambda-0.8)' -- simulation.py:657-658 */
double orc_default_lambda_electrostatics(double l) {
    return lep_step(0.2 - l) - 1.0 / 0.2 * l * lep_step(0.2 - l) + 1.0 / 0.2 * (l - 0.8) * lep_step(l - 0.8);
}

/* Python's round(x, 4): the decimal string of the EXACT binary value, correctly rounded to 4 places (half-to-even on
 * the exact value), read back.  round(x*1e4)/1e4 differs where x*1e4 rounds onto a tie (0.5-0.12345 -> 0.3765, not
 * 0.3766) -- found by running the reference's own _get_prop_lambda text (tests/golden/make_reference_vectors.py). */
static double round4(double x) { char b[64]; snprintf(b, sizeof b, "%.4f", x); return strtod(b, NULL); }

/* reference blues/integrators.py:147-157 */
void orc_get_prop_lambda(double prop_lambda, double out[2]) {
    double mx = round4(prop_lambda + 0.5), mn = round4(0.5 - prop_lambda);
    if (mx - mn <= 0.0) { mn = 2.0; mx = -1.0; }
    out[0] = mn; out[1] = mx;
}

/* reference blues/utils.py:89-145; out = {nstepsNC, propSteps, moveStep}; nstepsNC=-1 for the sys.exit(1) case */
void orc_calculate_ncmc_steps(int nstepsNC, int nprop, double pl, int out[3]) {
    if (nstepsNC % 2 != 0) {
        int r = nstepsNC & ~1;
        if (r) nstepsNC = r; else { out[0] = -1; out[1] = 0; out[2] = 0; return; }
    }
    double ls = nstepsNC / (2.0 * (nprop * pl + 0.5 - pl));
    int lambdaSteps = ((int)ls % 2 == 0) ? (int)ls : (int)ls + 1;
    double in_portion = pl * lambdaSteps, out_portion = (0.5 - pl) * lambdaSteps;
    int in_prop = (int)(nprop * (2 * floor(in_portion)));
    int out_prop = (int)(2 * ceil(out_portion));
    int propSteps = in_prop + out_prop;
    if (propSteps != nstepsNC) nstepsNC = lambdaSteps;
    out[0] = nstepsNC; out[1] = propSteps; out[2] = nstepsNC / 2;
}

/* ------------------------------------------------------------------ construction */
static void *dup_mem(const void *src, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (src && bytes) memcpy(p, src, bytes);
    return p;
}

static int uf_find(int *p, int a) { while (p[a] != a) { p[a] = p[p[a]]; a = p[a]; } return a; }

static int build_clusters(Oracle *o) {
    int n = o->n, nc = o->n_cons;
    int *parent = malloc(sizeof(int) * (n + 1));
    for (int i = 0; i < n; i++) parent[i] = i;
    /* drop constraints between two massless atoms (OpenMM ignores them); mixed is an error */
    int kept = 0;
    for (int c = 0; c < nc; c++) {
        int i = o->cons_atoms[2 * c], j = o->cons_atoms[2 * c + 1];
        int mi = o->mass[i] == 0.0, mj = o->mass[j] == 0.0;
        if (mi && mj) continue;
        if (mi != mj) { snprintf(g_err, sizeof g_err, "A constraint cannot involve a massless particle (%d-%d)", i, j); free(parent); return 1; }
        o->cons_atoms[2 * kept] = i; o->cons_atoms[2 * kept + 1] = j; o->cons_dist[kept] = o->cons_dist[c];
        kept++;
        int a = uf_find(parent, i), b = uf_find(parent, j);
        if (a != b) parent[a < b ? b : a] = a < b ? a : b;
    }
    o->n_cons = nc = kept;
    /* order clusters by their smallest atom index, constraints inside a cluster keep list order */
    int *root_of = malloc(sizeof(int) * (nc + 1));
    int *order = malloc(sizeof(int) * (nc + 1));
    for (int c = 0; c < nc; c++) { root_of[c] = uf_find(parent, o->cons_atoms[2 * c]); order[c] = c; }
    /* stable insertion sort by root (lists are short or already nearly sorted) -> use simple merge via counting */
    int *count = calloc(n + 1, sizeof(int));
    for (int c = 0; c < nc; c++) count[root_of[c]]++;
    int *start = malloc(sizeof(int) * (n + 1));
    int acc = 0, ncl = 0;
    for (int r = 0; r < n; r++) { start[r] = acc; acc += count[r]; if (count[r]) ncl++; }
    o->cluster_cons = malloc(sizeof(int) * (nc + 1));
    o->cluster_start = malloc(sizeof(int) * (ncl + 1));
    int *fill = calloc(n + 1, sizeof(int));
    for (int c = 0; c < nc; c++) { int r = root_of[c]; o->cluster_cons[start[r] + fill[r]++] = c; }
    int k = 0;
    for (int r = 0; r < n; r++) if (count[r]) o->cluster_start[k++] = start[r];
    o->cluster_start[ncl] = nc;
    o->n_clusters = ncl;
    free(parent); free(root_of); free(order); free(count); free(start); free(fill);
    return 0;
}

Oracle *orc_create(const BluesSystemDesc *s, const BluesIntegratorDesc *it) {
    g_err[0] = 0;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
        if (r != c && s->box[3 * r + c] != 0.0) { snprintf(g_err, sizeof g_err, "only orthorhombic boxes are supported"); return NULL; }
    Oracle *o = calloc(1, sizeof(Oracle));
    int n = o->n = s->n_atoms;
    o->box[0] = s->box[0]; o->box[1] = s->box[4]; o->box[2] = s->box[8];
    o->mass = dup_mem(s->mass, sizeof(double) * n);
    o->charge = dup_mem(s->charge, sizeof(double) * n);
    o->sigma = dup_mem(s->sigma, sizeof(double) * n);
    o->eps = dup_mem(s->epsilon, sizeof(double) * n);
    /* exclusions -> CSR both directions */
    o->n_excl = s->n_exclusions;
    o->excl_start = calloc(n + 1, sizeof(int));
    for (int e = 0; e < s->n_exclusions; e++) { o->excl_start[s->exclusions[2 * e] + 1]++; o->excl_start[s->exclusions[2 * e + 1] + 1]++; }
    for (int i = 0; i < n; i++) o->excl_start[i + 1] += o->excl_start[i];
    o->excl_list = malloc(sizeof(int) * (2 * s->n_exclusions + 1));
    int *fill = calloc(n, sizeof(int));
    for (int e = 0; e < s->n_exclusions; e++) {
        int a = s->exclusions[2 * e], b = s->exclusions[2 * e + 1];
        o->excl_list[o->excl_start[a] + fill[a]++] = b;
        o->excl_list[o->excl_start[b] + fill[b]++] = a;
    }
    free(fill);
    o->n_exc = s->n_exceptions;
    o->exc_atoms = dup_mem(s->exception_atoms, sizeof(int) * 2 * s->n_exceptions);
    o->exc_params = dup_mem(s->exception_params, sizeof(double) * 3 * s->n_exceptions);
    o->n_bonds = s->n_bonds; o->bond_atoms = dup_mem(s->bond_atoms, sizeof(int) * 2 * s->n_bonds); o->bond_params = dup_mem(s->bond_params, sizeof(double) * 2 * s->n_bonds);
    o->n_angles = s->n_angles; o->angle_atoms = dup_mem(s->angle_atoms, sizeof(int) * 3 * s->n_angles); o->angle_params = dup_mem(s->angle_params, sizeof(double) * 2 * s->n_angles);
    o->n_tors = s->n_torsions; o->tors_atoms = dup_mem(s->torsion_atoms, sizeof(int) * 4 * s->n_torsions); o->tors_params = dup_mem(s->torsion_params, sizeof(double) * 3 * s->n_torsions);
    o->n_cons = s->n_constraints; o->cons_atoms = dup_mem(s->constraint_atoms, sizeof(int) * 2 * s->n_constraints); o->cons_dist = dup_mem(s->constraint_dist, sizeof(double) * s->n_constraints);
    o->n_alch = s->n_alchemical; o->alch_atoms = dup_mem(s->alchemical_atoms, sizeof(int) * s->n_alchemical);
    o->is_alch = calloc(n, 1);
    for (int a = 0; a < o->n_alch; a++) o->is_alch[o->alch_atoms[a]] = 1;
    o->n_restr = s->n_restraints; o->restr_atoms = dup_mem(s->restraint_atoms, sizeof(int) * s->n_restraints); o->restr_x0 = dup_mem(s->restraint_x0, sizeof(double) * 3 * s->n_restraints);
    o->restr_k = s->restraint_k;
    o->nb_method = s->nonbonded_method; o->cutoff = s->cutoff; o->alpha_ewald = s->ewald_alpha; o->sc_alpha = s->softcore_alpha;
    o->annihilate_elec = s->annihilate_electrostatics; o->annihilate_sterics = s->annihilate_sterics; o->remove_cm = s->remove_cm_motion;
    if (o->nb_method == BLUES_NB_PME) {
        o->pme_order = s->pme_order; o->disp_corr = s->dispersion_correction;
        for (int d = 0; d < 3; d++) o->pme_K[d] = s->pme_grid[d];
        if (o->pme_order < 2 || o->pme_order > 8 || o->pme_K[0] < o->pme_order || o->pme_K[1] < o->pme_order || o->pme_K[2] < o->pme_order) {
            snprintf(g_err, sizeof g_err, "PME needs 2 <= order <= 8 and a mesh of at least `order` points per edge"); orc_destroy(o); return NULL; }
        pme_setup(o);
    }
    if (o->nb_method == BLUES_NB_PME_DIRECT || o->nb_method == BLUES_NB_PME)
        for (int d = 0; d < 3; d++) if (o->box[d] < 2.0 * o->cutoff) { snprintf(g_err, sizeof g_err, "box edge %g < 2*cutoff", o->box[d]); orc_destroy(o); return NULL; }
    if (build_clusters(o)) { orc_destroy(o); return NULL; }
    /* integrator */
    o->dt = it->timestep; o->temperature = it->temperature; o->gamma = it->collision_rate; o->kT = KB_KJ * it->temperature;
    o->tol = it->constraint_tolerance; o->nsteps = it->nsteps_neq; o->nprop_i = it->nprop; o->n_lambda_steps = it->n_lambda_steps;
    o->seed = it->seed; o->replica = (uint32_t)it->replica; o->draw = 0;
    o->n_split = 0;
    o->switch_mode = it->switching_mode; o->psteps = it->steps_per_propagation;
    if (o->switch_mode != BLUES_SWITCH_NONE) {
        if (o->switch_mode != BLUES_SWITCH_VV && o->switch_mode != BLUES_SWITCH_GHMC) { snprintf(g_err, sizeof g_err, "unknown switching_mode %d", o->switch_mode); orc_destroy(o); return NULL; }
        /* steps_per_propagation == 0: the reference's nsteps = 0 program, an instantaneous toggle without propagation (switching.py:1198-1207) */
        if (o->psteps < 0 || o->nsteps < 1) { snprintf(g_err, sizeof g_err, "switching integrators need nsteps_neq >= 1 and steps_per_propagation >= 0"); orc_destroy(o); return NULL; }
        o->n_H = 1;
    }
    for (const char *p = o->switch_mode ? "" : it->splitting; *p; p++) {
        if (*p == ' ') continue;
        if (*p != 'R' && *p != 'V' && *p != 'O' && *p != 'H' && *p != 'L') { snprintf(g_err, sizeof g_err, "unsupported splitting token '%c'", *p); orc_destroy(o); return NULL; }
        if (o->n_split >= MAX_SPLIT - 1) { snprintf(g_err, sizeof g_err, "splitting too long"); orc_destroy(o); return NULL; }
        o->split[o->n_split++] = *p;
        if (*p == 'R') o->n_R++; else if (*p == 'V') o->n_V++; else if (*p == 'O') o->n_O++; else if (*p == 'H') o->n_H++;
    }
    if (o->n_lambda_steps != o->nsteps * o->n_H) { snprintf(g_err, sizeof g_err, "n_lambda_steps must be nsteps_neq * n_H"); orc_destroy(o); return NULL; }
    o->tab_ls = dup_mem(it->lambda_sterics, sizeof(double) * (o->n_lambda_steps + 1));
    o->tab_le = dup_mem(it->lambda_electrostatics, sizeof(double) * (o->n_lambda_steps + 1));
    memset(&o->g, 0, sizeof o->g);
    o->g.n_lambda_steps = o->n_lambda_steps; o->g.nsteps = o->nsteps; o->g.nprop = it->nprop; o->g.prop = 1;
    o->g.prop_lambda_min = it->prop_lambda_min; o->g.prop_lambda_max = it->prop_lambda_max;
    /* context parameter defaults of the alchemical system are 1.0 */
    o->g.lambda_sterics = 1.0; o->g.lambda_electrostatics = 1.0;
    o->x = calloc(3 * n, sizeof(double)); o->v = calloc(3 * n, sizeof(double));
    o->xref = calloc(3 * n, sizeof(double)); o->f = calloc(3 * n, sizeof(double));
    o->cell_next = malloc(sizeof(int) * n);
    o->cell_head = NULL;
    return o;
}

void orc_destroy(Oracle *o) {
    if (!o) return;
    free(o->mass); free(o->charge); free(o->sigma); free(o->eps); free(o->excl_start); free(o->excl_list);
    free(o->exc_atoms); free(o->exc_params); free(o->bond_atoms); free(o->bond_params); free(o->angle_atoms);
    free(o->angle_params); free(o->tors_atoms); free(o->tors_params); free(o->cons_atoms); free(o->cons_dist);
    free(o->cluster_start); free(o->cluster_cons); free(o->alch_atoms); free(o->is_alch); free(o->restr_atoms);
    free(o->restr_x0); free(o->tab_ls); free(o->tab_le); free(o->x); free(o->v); free(o->xref); free(o->f);
    free(o->cell_head); free(o->cell_next);
    for (int d = 0; d < 3; d++) { free(o->pme_mod[d]); free(o->pme_cos[d]); free(o->pme_sin[d]); }
    free(o->pme_q); free(o->pme_re); free(o->pme_im); free(o->pme_tr); free(o->pme_ti);
    free(o);
}

void orc_set_positions(Oracle *o, const double *xyz) { memcpy(o->x, xyz, sizeof(double) * 3 * o->n); o->xver++; }
void orc_set_velocities(Oracle *o, const double *xyz) { memcpy(o->v, xyz, sizeof(double) * 3 * o->n); }
void orc_get_positions(Oracle *o, double *xyz) { memcpy(xyz, o->x, sizeof(double) * 3 * o->n); }
void orc_get_velocities(Oracle *o, double *xyz) { memcpy(xyz, o->v, sizeof(double) * 3 * o->n); }
void orc_set_box(Oracle *o, const double box[9]) { o->box[0] = box[0]; o->box[1] = box[4]; o->box[2] = box[8]; o->xver++; }
long orc_num_evaluations(Oracle *o) { return o->n_evals; }

/* ------------------------------------------------------------------ potential */
static inline void min_image(const Oracle *o, double d[3]) {
    if (o->nb_method == BLUES_NB_NOCUTOFF) return;
    for (int k = 0; k < 3; k++) d[k] -= o->box[k] * nearbyint(d[k] / o->box[k]);
}

static inline int is_excluded(const Oracle *o, int i, int j) {
    for (int e = o->excl_start[i]; e < o->excl_start[i + 1]; e++) if (o->excl_list[e] == j) return 1;
    return 0;
}

/* softcore LJ of the alchemical CustomNonbondedForce: U = ls*4 eps x (x-1),
 * x = 1/(alpha (1-ls) + (r/sigma)^6)  (SURVEY.md Appendix B). returns U, *fs = -dU/dr / r */
static inline double softcore_lj(double r2, double sig, double eps, double ls, double alpha, double *fs) {
    if (eps == 0.0 || sig == 0.0) { *fs = 0.0; return 0.0; }
    double s2 = sig * sig, q2 = r2 / s2, q6 = q2 * q2 * q2;
    double x = 1.0 / (alpha * (1.0 - ls) + q6);
    double U = ls * 4.0 * eps * x * (x - 1.0);
    /* dU/dr = ls 4 eps (2x-1) dx/dr, dx/dr = -x^2 * 6 r^5 / sig^6 ;  -dU/dr / r = ls 4 eps (2x-1) x^2 6 r^4/sig^6 */
    *fs = ls * 4.0 * eps * (2.0 * x - 1.0) * x * x * 6.0 * q2 * q2 / s2;
    return U;
}

static inline double plain_lj(double r2, double sig, double eps, double *fs) {
    if (eps == 0.0) { *fs = 0.0; return 0.0; }
    double sr2 = sig * sig / r2, sr6 = sr2 * sr2 * sr2;
    *fs = 4.0 * eps * (12.0 * sr6 * sr6 - 6.0 * sr6) / r2;
    return 4.0 * eps * (sr6 * sr6 - sr6);
}

/* Coulomb: erfc-screened (PME direct space) or bare. returns U, *fs = -dU/dr / r */
static inline double coulomb(double r2, double qq, double alpha, int screened, double *fs) {
    if (qq == 0.0) { *fs = 0.0; return 0.0; }
    double r = sqrt(r2), pre = ONE_4PI_EPS0 * qq;
    if (!screened) { *fs = pre / (r * r2); return pre / r; }
    double ar = alpha * r, ec = erfc(ar);
    *fs = pre * (ec / r + TWO_OVER_SQRT_PI * alpha * exp(-ar * ar)) / r2;
    return pre * ec / r;
}

/* pair energy helper exposed for analytic two-body tests */
double orc_pair_energy(double r, double qq, double sigma, double epsilon, double alpha_ewald, int alchemical,
                       double ls, double le, double sc_alpha, double *dEdr) {
    double r2 = r * r, f1, f2, U;
    if (alchemical) {
        U = softcore_lj(r2, sigma, epsilon, ls, sc_alpha, &f1);
        double Uc = coulomb(r2, qq, alpha_ewald, alpha_ewald > 0.0, &f2);
        U += le * Uc; f2 *= le;
    } else {
        U = plain_lj(r2, sigma, epsilon, &f1) + coulomb(r2, qq, alpha_ewald, alpha_ewald > 0.0, &f2);
    }
    if (dEdr) *dEdr = -(f1 + f2) * r;
    return U;
}

static inline void nb_pair(const Oracle *o, int i, int j, double ls, double le, double *F, double *T) {
    double d[3] = {o->x[3 * i] - o->x[3 * j], o->x[3 * i + 1] - o->x[3 * j + 1], o->x[3 * i + 2] - o->x[3 * j + 2]};
    min_image(o, d);
    double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    int pme = (o->nb_method == BLUES_NB_PME_DIRECT || o->nb_method == BLUES_NB_PME);
    if (pme && r2 >= o->cutoff * o->cutoff) return;
    if (is_excluded(o, i, j)) return;
    if (o->custom_pair_mode == 1) { /* ethylene_system.xml:52: interaction group set1 x set2 only */
        if (o->is_alch[i] == o->is_alch[j]) return;
        double q = o->charge[i] * o->charge[j];
        double sg = 0.5 * (o->sigma[i] + o->sigma[j]) * ls, ep = sqrt(o->eps[i] * o->eps[j]) * le;
        double sr2 = sg * sg / r2, sr6 = sr2 * sr2 * sr2;
        T[6] += q / r2; T[5] += 4.0 * ep * (sr6 * sr6 - sr6);
        double fsc = 2.0 * q / (r2 * r2) + 4.0 * ep * (12.0 * sr6 * sr6 - 6.0 * sr6) / r2;
        if (F) for (int k = 0; k < 3; k++) { F[3 * i + k] += fsc * d[k]; F[3 * j + k] -= fsc * d[k]; }
        return;
    }
    double sig = 0.5 * (o->sigma[i] + o->sigma[j]), eps = sqrt(o->eps[i] * o->eps[j]), qq = o->charge[i] * o->charge[j];
    int ai = o->is_alch[i], aj = o->is_alch[j];
    double fs, fc, fscale;
    if (!ai && !aj) {
        double U = plain_lj(r2, sig, eps, &fs) + coulomb(r2, qq, o->alpha_ewald, pme, &fc);
        T[3] += U; fscale = fs + fc;
    } else {
        int both = ai && aj;
        double ls_eff = (both && !o->annihilate_sterics) ? 1.0 : ls;
        double le_eff = (both && !o->annihilate_elec) ? 1.0 : le;
        T[5] += softcore_lj(r2, sig, eps, ls_eff, o->sc_alpha, &fs);
        T[6] += le_eff * coulomb(r2, qq, o->alpha_ewald, pme, &fc);
        fscale = fs + le_eff * fc;
    }
    if (F) for (int k = 0; k < 3; k++) { F[3 * i + k] += fscale * d[k]; F[3 * j + k] -= fscale * d[k]; }
}

static void nonbonded(Oracle *o, double ls, double le, double *F, double *T) {
    int n = o->n;
    int use_cells = (o->nb_method == BLUES_NB_PME_DIRECT || o->nb_method == BLUES_NB_PME);
    int nc[3] = {1, 1, 1};
    if (use_cells) for (int d = 0; d < 3; d++) { nc[d] = (int)floor(o->box[d] / o->cutoff); if (nc[d] < 3) use_cells = 0; }
    if (!use_cells) {
        for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) nb_pair(o, i, j, ls, le, F, T);
        return;
    }
    int ncell = nc[0] * nc[1] * nc[2];
    o->cell_head = realloc(o->cell_head, sizeof(int) * ncell);
    for (int c = 0; c < ncell; c++) o->cell_head[c] = -1;
    int *cell_of = malloc(sizeof(int) * n);
    for (int i = n - 1; i >= 0; i--) {
        int ci[3];
        for (int d = 0; d < 3; d++) {
            double fr = o->x[3 * i + d] / o->box[d]; fr -= floor(fr);
            ci[d] = (int)(fr * nc[d]); if (ci[d] >= nc[d]) ci[d] = nc[d] - 1;
        }
        int c = (ci[0] * nc[1] + ci[1]) * nc[2] + ci[2];
        cell_of[i] = c; o->cell_next[i] = o->cell_head[c]; o->cell_head[c] = i;
    }
#ifdef ORC_OPENMP
    /* CPU baseline (ii) of SURVEY.md 8(d): the same pair loop on all cores.  Cells are dealt to the threads statically and the
     * thread-private force / energy arrays are summed in thread order, so a run is reproducible for a given thread count. */
    const int nt = omp_get_max_threads() < ncell ? omp_get_max_threads() : ncell;   /* (more threads than cells would only add empty force arrays to sum) */
    /* the threads' force arrays live as long as the process and are cleared by their owners inside the parallel region (allocated
     * and zeroed afresh by one thread for every evaluation, 192 x 23,400 x 3 doubles of page faults cost more than the pair loop) */
    static double *Fp = NULL; static size_t Fp_cap = 0;
    if (F && Fp_cap < (size_t)nt * 3 * n) { free(Fp); Fp_cap = (size_t)nt * 3 * n; Fp = malloc(Fp_cap * sizeof(double)); }
    double *Tp = calloc((size_t)nt * BLUES_N_ENERGY_TERMS, sizeof(double));
#pragma omp parallel num_threads(nt)
    {
        const int th = omp_get_thread_num();
        double *Ft = F ? Fp + (size_t)th * 3 * n : NULL, *Tt = Tp + (size_t)th * BLUES_N_ENERGY_TERMS;
        if (Ft) memset(Ft, 0, sizeof(double) * 3 * n);
        /* every cell pairs with itself and with the 13 neighbours at a lexicographically positive offset: each unordered pair of
         * neighbouring cells once (nc >= 3 per edge), and the same amount of work for every cell (testing c2 >= c instead leaves
         * the low-numbered cells with twice the mean) */
#pragma omp for schedule(static)
        for (int c = 0; c < ncell; c++) {
            const int cx = c / (nc[1] * nc[2]), cy = (c / nc[2]) % nc[1], cz = c % nc[2];
            for (int dx = 0; dx <= 1; dx++) for (int dy = -1; dy <= 1; dy++) for (int dz = -1; dz <= 1; dz++) {
                if (dx == 0 && (dy < 0 || (dy == 0 && dz < 0))) continue;
                int ox = (cx + dx + nc[0]) % nc[0], oy = (cy + dy + nc[1]) % nc[1], oz = (cz + dz + nc[2]) % nc[2];
                int c2 = (ox * nc[1] + oy) * nc[2] + oz;
                for (int i = o->cell_head[c]; i >= 0; i = o->cell_next[i])
                    for (int j = o->cell_head[c2]; j >= 0; j = o->cell_next[j]) {
                        if (c2 == c && j <= i) continue;
                        nb_pair(o, i, j, ls, le, Ft, Tt);
                    }
            }
        }
    }
    if (F) {
        /* thread order: reproducible.  Blocks of coordinates, the threads' arrays walked one after the other inside a block
         * (reading element q of 192 arrays 560 KB apart for one q at a time cost more than the pair loop) */
        const int n3 = 3 * n, BLK = 2048;
#pragma omp parallel for schedule(static)
        for (int q0 = 0; q0 < n3; q0 += BLK) {
            double acc[2048];
            const int len = n3 - q0 < BLK ? n3 - q0 : BLK;
            for (int q = 0; q < len; q++) acc[q] = 0.0;
            for (int th = 0; th < nt; th++) { const double *src = Fp + (size_t)th * n3 + q0; for (int q = 0; q < len; q++) acc[q] += src[q]; }
            for (int q = 0; q < len; q++) F[q0 + q] += acc[q];
        }
    }
    for (int th = 0; th < nt; th++) for (int q = 0; q < BLUES_N_ENERGY_TERMS; q++) T[q] += Tp[(size_t)th * BLUES_N_ENERGY_TERMS + q];
    free(Tp);
#else
    for (int cx = 0; cx < nc[0]; cx++) for (int cy = 0; cy < nc[1]; cy++) for (int cz = 0; cz < nc[2]; cz++) {
        int c = (cx * nc[1] + cy) * nc[2] + cz;
        for (int dx = -1; dx <= 1; dx++) for (int dy = -1; dy <= 1; dy++) for (int dz = -1; dz <= 1; dz++) {
            int ox = (cx + dx + nc[0]) % nc[0], oy = (cy + dy + nc[1]) % nc[1], oz = (cz + dz + nc[2]) % nc[2];
            int c2 = (ox * nc[1] + oy) * nc[2] + oz;
            if (c2 < c) continue; /* each unordered cell pair once (nc>=3 so neighbours are distinct) */
            for (int i = o->cell_head[c]; i >= 0; i = o->cell_next[i])
                for (int j = o->cell_head[c2]; j >= 0; j = o->cell_next[j]) {
                    if (c2 == c && j <= i) continue;
                    nb_pair(o, i, j, ls, le, F, T);
                }
        }
    }
#endif
    free(cell_of);
}

static void exceptions(const Oracle *o, double ls, double le, double *F, double *T) {
    for (int e = 0; e < o->n_exc; e++) {
        int i = o->exc_atoms[2 * e], j = o->exc_atoms[2 * e + 1];
        double qq = o->exc_params[3 * e], sig = o->exc_params[3 * e + 1], eps = o->exc_params[3 * e + 2];
        double d[3] = {o->x[3 * i] - o->x[3 * j], o->x[3 * i + 1] - o->x[3 * j + 1], o->x[3 * i + 2] - o->x[3 * j + 2]};
        min_image(o, d);
        double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], fs, fc, fscale;
        int ai = o->is_alch[i], aj = o->is_alch[j];
        if (!ai && !aj) {
            T[4] += plain_lj(r2, sig, eps, &fs) + coulomb(r2, qq, 0.0, 0, &fc);
            fscale = fs + fc;
        } else {
            int both = ai && aj;
            double ls_eff = (both && !o->annihilate_sterics) ? 1.0 : ls;
            double le_eff = (both && !o->annihilate_elec) ? 1.0 : le;
            T[5] += softcore_lj(r2, sig, eps, ls_eff, o->sc_alpha, &fs);
            T[6] += le_eff * coulomb(r2, qq, 0.0, 0, &fc);
            fscale = fs + le_eff * fc;
        }
        if (F) for (int k = 0; k < 3; k++) { F[3 * i + k] += fscale * d[k]; F[3 * j + k] -= fscale * d[k]; }
    }
}

static inline void cross(const double a[3], const double b[3], double c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
static inline double dot(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static void bonded(const Oracle *o, double *F, double *T) {
    const double *x = o->x;
    for (int b = 0; b < o->n_bonds; b++) {
        int i = o->bond_atoms[2 * b], j = o->bond_atoms[2 * b + 1];
        double r0 = o->bond_params[2 * b], k = o->bond_params[2 * b + 1];
        double d[3] = {x[3 * i] - x[3 * j], x[3 * i + 1] - x[3 * j + 1], x[3 * i + 2] - x[3 * j + 2]};
        min_image(o, d);
        double r = sqrt(dot(d, d)), dr = r - r0;
        T[0] += 0.5 * k * dr * dr;
        if (F) { double fs = -k * dr / r; for (int c = 0; c < 3; c++) { F[3 * i + c] += fs * d[c]; F[3 * j + c] -= fs * d[c]; } }
    }
    for (int a = 0; a < o->n_angles; a++) {
        int i = o->angle_atoms[3 * a], j = o->angle_atoms[3 * a + 1], k = o->angle_atoms[3 * a + 2];
        double t0 = o->angle_params[2 * a], kk = o->angle_params[2 * a + 1];
        double u[3] = {x[3 * i] - x[3 * j], x[3 * i + 1] - x[3 * j + 1], x[3 * i + 2] - x[3 * j + 2]};
        double w[3] = {x[3 * k] - x[3 * j], x[3 * k + 1] - x[3 * j + 1], x[3 * k + 2] - x[3 * j + 2]};
        min_image(o, u); min_image(o, w);
        double lu = sqrt(dot(u, u)), lw = sqrt(dot(w, w)), c = dot(u, w) / (lu * lw);
        if (c > 1.0) c = 1.0; if (c < -1.0) c = -1.0;
        double th = acos(c), dth = th - t0;
        T[1] += 0.5 * kk * dth * dth;
        if (F) {
            double s = sqrt(1.0 - c * c); if (s < 1e-12) s = 1e-12;
            double dEdth = kk * dth; /* F = -dE/dtheta * dtheta/dx ; dtheta/dcos = -1/sin */
            for (int m = 0; m < 3; m++) {
                double fi = dEdth / s * (w[m] / (lu * lw) - c * u[m] / (lu * lu));
                double fk = dEdth / s * (u[m] / (lu * lw) - c * w[m] / (lw * lw));
                F[3 * i + m] += fi; F[3 * k + m] += fk; F[3 * j + m] -= fi + fk;
            }
        }
    }
    for (int t = 0; t < o->n_tors; t++) {
        int i = o->tors_atoms[4 * t], j = o->tors_atoms[4 * t + 1], k = o->tors_atoms[4 * t + 2], l = o->tors_atoms[4 * t + 3];
        double per = o->tors_params[3 * t], ph = o->tors_params[3 * t + 1], kk = o->tors_params[3 * t + 2];
        double rij[3], rkj[3], rkl[3], m[3], nn[3];
        for (int c = 0; c < 3; c++) { rij[c] = x[3 * i + c] - x[3 * j + c]; rkj[c] = x[3 * k + c] - x[3 * j + c]; rkl[c] = x[3 * k + c] - x[3 * l + c]; }
        min_image(o, rij); min_image(o, rkj); min_image(o, rkl);
        cross(rij, rkj, m); cross(rkj, rkl, nn);
        double m2 = dot(m, m), n2 = dot(nn, nn), lkj2 = dot(rkj, rkj), lkj = sqrt(lkj2);
        double cs = dot(m, nn) / sqrt(m2 * n2);
        if (cs > 1.0) cs = 1.0; if (cs < -1.0) cs = -1.0;
        double phi = acos(cs); if (dot(rij, nn) < 0.0) phi = -phi;
        T[2] += kk * (1.0 + cos(per * phi - ph));
        if (F) {
            double ddphi = -kk * per * sin(per * phi - ph); /* dE/dphi */
            double fi[3], fl[3], sv[3];
            double p = dot(rij, rkj) / lkj2, q = dot(rkl, rkj) / lkj2;
            for (int c = 0; c < 3; c++) { fi[c] = -ddphi * lkj / m2 * m[c]; fl[c] = ddphi * lkj / n2 * nn[c]; }
            for (int c = 0; c < 3; c++) { sv[c] = p * fi[c] - q * fl[c]; }
            for (int c = 0; c < 3; c++) {
                F[3 * i + c] += fi[c]; F[3 * j + c] -= fi[c] - sv[c]; F[3 * k + c] -= fl[c] + sv[c]; F[3 * l + c] += fl[c];
            }
        }
    }
    /* CustomCentroidBondForce '0.5*k*distance(g1,g2)^2' (ethylene_system.xml:94-112), oracle-only */
    for (int b = 0; b < o->n_centroid; b++) {
        double c1[3] = {0, 0, 0}, c2[3] = {0, 0, 0}, W1 = 0, W2 = 0;
        for (int q = 0; q < o->cb_n1[b]; q++) { W1 += o->cb_w1[b][q]; for (int k = 0; k < 3; k++) c1[k] += o->cb_w1[b][q] * x[3 * o->cb_i1[b][q] + k]; }
        for (int q = 0; q < o->cb_n2[b]; q++) { W2 += o->cb_w2[b][q]; for (int k = 0; k < 3; k++) c2[k] += o->cb_w2[b][q] * x[3 * o->cb_i2[b][q] + k]; }
        double d[3]; for (int k = 0; k < 3; k++) d[k] = c1[k] / W1 - c2[k] / W2;
        T[7] += 0.5 * o->cb_k[b] * dot(d, d);
        if (F) {
            for (int q = 0; q < o->cb_n1[b]; q++) for (int k = 0; k < 3; k++) F[3 * o->cb_i1[b][q] + k] -= o->cb_k[b] * d[k] * o->cb_w1[b][q] / W1;
            for (int q = 0; q < o->cb_n2[b]; q++) for (int k = 0; k < 3; k++) F[3 * o->cb_i2[b][q] + k] += o->cb_k[b] * d[k] * o->cb_w2[b][q] / W2;
        }
    }
    /* k_restr*periodicdistance(x,y,z,x0,y0,z0)^2  (reference blues/simulation.py:347) */
    for (int r = 0; r < o->n_restr; r++) {
        int i = o->restr_atoms[r];
        double d[3] = {x[3 * i] - o->restr_x0[3 * r], x[3 * i + 1] - o->restr_x0[3 * r + 1], x[3 * i + 2] - o->restr_x0[3 * r + 2]};
        min_image(o, d);
        T[7] += o->restr_k * dot(d, d);
        if (F) for (int c = 0; c < 3; c++) F[3 * i + c] -= 2.0 * o->restr_k * d[c];
    }
}


/* ------------------------------------------------------------------ reciprocal space (BLUES_NB_PME)
 * What OpenMM's NonbondedForce adds under nonbondedMethod=PME on the Reference platform, restated from the published
 * smooth particle-mesh Ewald algorithm (Essmann et al., J. Chem. Phys. 103, 8577 (1995)) in the form OpenMM 7.4.2's
 * ReferencePME.cpp uses it [recalled -- OpenMM is not under /root/reference]: order-5 cardinal B-splines, charges spread on
 * the mesh points (floor(u) + k) mod K, k = 0..order-1, forward transform, multiplication by
 *     eterm(m) = ONE_4PI_EPS0 exp(-pi^2 m^2 / alpha^2) / (pi V m^2 |b_x(m_x) b_y(m_y) b_z(m_z)|^2),
 * E = 1/2 sum_m eterm |Q^(m)|^2, backward transform, forces from the spline derivatives; then the Ewald self term
 * -ONE_4PI_EPS0 alpha/sqrt(pi) sum q_i^2, -ONE_4PI_EPS0 q_i q_j erf(alpha r)/r for every excluded pair, and the neutralising
 * background -pi ONE_4PI_EPS0 Q^2 / (2 V alpha^2).  The charges are the NonbondedForce's: alchemical atoms enter with 0
 * (alchemical_pme_treatment='direct-space', reference blues/simulation.py:225-236), and with epsilon 0 in the dispersion
 * correction (disable_alchemical_dispersion_correction=True).  Transforms are plain separable DFTs (meshes are ~20-40 points
 * per edge): this is a checker, not a fast code. */
static void pme_bsplines(double dr, int order, double *w, double *dw) {
    w[order - 1] = 0.0; w[1] = dr; w[0] = 1.0 - dr;
    for (int l = 3; l < order; l++) {
        const double div = 1.0 / (l - 1.0);
        w[l - 1] = div * dr * w[l - 2];
        for (int k = 1; k < l - 1; k++) w[l - k - 1] = div * ((dr + k) * w[l - k - 2] + (l - k - dr) * w[l - k - 1]);
        w[0] = div * (1.0 - dr) * w[0];
    }
    if (dw) { dw[0] = -w[0]; for (int k = 1; k < order; k++) dw[k] = w[k - 1] - w[k]; }
    const double div = 1.0 / (order - 1.0);
    w[order - 1] = div * dr * w[order - 2];
    for (int k = 1; k < order - 1; k++) w[order - k - 1] = div * ((dr + k) * w[order - k - 2] + (order - k - dr) * w[order - k - 1]);
    w[0] = div * (1.0 - dr) * w[0];
}

static void pme_setup(Oracle *o) {
    const int order = o->pme_order;
    double w[8];
    pme_bsplines(0.0, order, w, NULL);
    for (int d = 0; d < 3; d++) {
        const int K = o->pme_K[d];
        o->pme_mod[d] = malloc(sizeof(double) * K); o->pme_cos[d] = malloc(sizeof(double) * K); o->pme_sin[d] = malloc(sizeof(double) * K);
        for (int m = 0; m < K; m++) {
            o->pme_cos[d][m] = cos(2.0 * M_PI * m / K); o->pme_sin[d][m] = sin(2.0 * M_PI * m / K);
            double sc = 0.0, ss = 0.0;
            for (int k = 0; k < order; k++) { const double a = 2.0 * M_PI * m * (k + 1) / K; sc += w[k] * cos(a); ss += w[k] * sin(a); }
            o->pme_mod[d][m] = sc * sc + ss * ss;
        }
        for (int m = 0; m < K; m++) if (o->pme_mod[d][m] < 1e-7) o->pme_mod[d][m] = 0.5 * (o->pme_mod[d][(m + K - 1) % K] + o->pme_mod[d][(m + 1) % K]);
    }
    const size_t ng = (size_t)o->pme_K[0] * o->pme_K[1] * o->pme_K[2];
    o->pme_q = malloc(sizeof(double) * ng); o->pme_re = malloc(sizeof(double) * ng); o->pme_im = malloc(sizeof(double) * ng);
    o->pme_tr = malloc(sizeof(double) * ng); o->pme_ti = malloc(sizeof(double) * ng);
}

/* separable DFT of (re, im) in place; sign -1 forward, +1 backward (no normalisation) */
static void pme_dft3(const Oracle *o, double *re, double *im, double *tr, double *ti, int sign) {
    const int K0 = o->pme_K[0], K1 = o->pme_K[1], K2 = o->pme_K[2];
    const int dims[3] = {K0, K1, K2};
    const size_t stride[3] = {(size_t)K1 * K2, (size_t)K2, 1};
    for (int ax = 2; ax >= 0; ax--) {
        const int K = dims[ax]; const size_t st = stride[ax];
        const size_t ng = (size_t)K0 * K1 * K2;
        for (size_t base = 0; base < ng; base++) {
            if ((base / st) % K) continue;   /* first element of a line along ax */
            for (int m = 0; m < K; m++) {
                double sr = 0.0, si = 0.0;
                for (int k = 0; k < K; k++) {
                    const int t = (int)(((long)m * k) % K);
                    const double c = o->pme_cos[ax][t], s_ = sign * o->pme_sin[ax][t];
                    const double a = re[base + k * st], b = im[base + k * st];
                    sr += a * c - b * s_; si += a * s_ + b * c;
                }
                tr[base + m * st] = sr; ti[base + m * st] = si;
            }
        }
        memcpy(re, tr, sizeof(double) * ng); memcpy(im, ti, sizeof(double) * ng);
    }
}

static double dispersion_correction(const Oracle *o) {
    /* [recalled: OpenMM NonbondedForceImpl::calcDispersionCorrection, no switching function]: atoms grouped by (sigma, epsilon);
     * pairs of classes counted n_i (n_i + 1) / 2 and n_i n_j, normalised by N (N + 1) / 2; E = 8 pi N^2 (S12 / (9 rc^9) - S6 / (3 rc^3)) / V */
    const int n = o->n;
    double *cs = malloc(sizeof(double) * n), *ce = malloc(sizeof(double) * n); long *cn = calloc(n, sizeof(long));
    int nc = 0;
    for (int i = 0; i < n; i++) {
        const double e = o->is_alch[i] ? 0.0 : o->eps[i], sg = o->sigma[i];
        int c = -1;
        for (int q = 0; q < nc; q++) if (cs[q] == sg && ce[q] == e) { c = q; break; }
        if (c < 0) { c = nc++; cs[c] = sg; ce[c] = e; }
        cn[c]++;
    }
    double s12 = 0.0, s6 = 0.0;
    for (int a = 0; a < nc; a++) for (int b = a; b < nc; b++) {
        const double cnt = a == b ? 0.5 * cn[a] * (cn[a] + 1.0) : (double)cn[a] * cn[b];
        const double sg = 0.5 * (cs[a] + cs[b]), e = sqrt(ce[a] * ce[b]);
        const double s2 = sg * sg, s6_ = s2 * s2 * s2;
        s12 += cnt * e * s6_ * s6_; s6 += cnt * e * s6_;
    }
    const double tot = 0.5 * n * (n + 1.0);
    s12 /= tot; s6 /= tot;
    free(cs); free(ce); free(cn);
    const double rc = o->cutoff, rc3 = rc * rc * rc, rc9 = rc3 * rc3 * rc3;
    const double V = o->box[0] * o->box[1] * o->box[2];
    return 8.0 * n * (double)n * M_PI * (s12 / (9.0 * rc9) - s6 / (3.0 * rc3)) / V;
}

static void reciprocal_space(Oracle *o, double *F, double *T) {
    const int n = o->n, order = o->pme_order;
    const int K0 = o->pme_K[0], K1 = o->pme_K[1], K2 = o->pme_K[2];
    const size_t ng = (size_t)K0 * K1 * K2;
    const double V = o->box[0] * o->box[1] * o->box[2], alpha = o->alpha_ewald;
    memset(o->pme_q, 0, sizeof(double) * ng);
    int *idx = malloc(sizeof(int) * 3 * n); double *w = malloc(sizeof(double) * 3 * 8 * n), *dw = malloc(sizeof(double) * 3 * 8 * n);
    double qsum = 0.0, q2sum = 0.0;
    for (int i = 0; i < n; i++) {
        const double q = o->is_alch[i] ? 0.0 : o->charge[i];
        qsum += q; q2sum += q * q;
        for (int d = 0; d < 3; d++) {
            double fr = o->x[3 * i + d] / o->box[d]; fr -= floor(fr);
            double u = fr * o->pme_K[d]; int ti = (int)u; double dr = u - ti;
            if (ti >= o->pme_K[d]) { ti -= o->pme_K[d]; }
            idx[3 * i + d] = ti;
            pme_bsplines(dr, order, w + (3 * i + d) * 8, dw + (3 * i + d) * 8);
        }
        if (q == 0.0) continue;
        for (int a = 0; a < order; a++) for (int b = 0; b < order; b++) for (int c = 0; c < order; c++) {
            const int ix = (idx[3 * i] + a) % K0, iy = (idx[3 * i + 1] + b) % K1, iz = (idx[3 * i + 2] + c) % K2;
            o->pme_q[((size_t)ix * K1 + iy) * K2 + iz] += q * w[(3 * i) * 8 + a] * w[(3 * i + 1) * 8 + b] * w[(3 * i + 2) * 8 + c];
        }
    }
    memcpy(o->pme_re, o->pme_q, sizeof(double) * ng); memset(o->pme_im, 0, sizeof(double) * ng);
    pme_dft3(o, o->pme_re, o->pme_im, o->pme_tr, o->pme_ti, -1);
    double esum = 0.0;
    for (int kx = 0; kx < K0; kx++) for (int ky = 0; ky < K1; ky++) for (int kz = 0; kz < K2; kz++) {
        const size_t g = ((size_t)kx * K1 + ky) * K2 + kz;
        if (kx == 0 && ky == 0 && kz == 0) { o->pme_re[g] = 0.0; o->pme_im[g] = 0.0; continue; }
        const double mx = (kx < (K0 + 1) / 2 ? kx : kx - K0) / o->box[0], my = (ky < (K1 + 1) / 2 ? ky : ky - K1) / o->box[1], mz = (kz < (K2 + 1) / 2 ? kz : kz - K2) / o->box[2];
        const double m2 = mx * mx + my * my + mz * mz;
        const double eterm = ONE_4PI_EPS0 * exp(-M_PI * M_PI * m2 / (alpha * alpha)) / (M_PI * V * m2 * o->pme_mod[0][kx] * o->pme_mod[1][ky] * o->pme_mod[2][kz]);
        esum += eterm * (o->pme_re[g] * o->pme_re[g] + o->pme_im[g] * o->pme_im[g]);
        o->pme_re[g] *= eterm; o->pme_im[g] *= eterm;
    }
    double E = 0.5 * esum;
    if (F) {
        pme_dft3(o, o->pme_re, o->pme_im, o->pme_tr, o->pme_ti, +1);
        for (int i = 0; i < n; i++) {
            const double q = o->is_alch[i] ? 0.0 : o->charge[i];
            if (q == 0.0) continue;
            double fx = 0.0, fy = 0.0, fz = 0.0;
            for (int a = 0; a < order; a++) for (int b = 0; b < order; b++) for (int c = 0; c < order; c++) {
                const int ix = (idx[3 * i] + a) % K0, iy = (idx[3 * i + 1] + b) % K1, iz = (idx[3 * i + 2] + c) % K2;
                const double phi = o->pme_re[((size_t)ix * K1 + iy) * K2 + iz];
                const double wx = w[(3 * i) * 8 + a], wy = w[(3 * i + 1) * 8 + b], wz = w[(3 * i + 2) * 8 + c];
                fx += dw[(3 * i) * 8 + a] * wy * wz * phi; fy += wx * dw[(3 * i + 1) * 8 + b] * wz * phi; fz += wx * wy * dw[(3 * i + 2) * 8 + c] * phi;
            }
            F[3 * i] -= q * fx * K0 / o->box[0]; F[3 * i + 1] -= q * fy * K1 / o->box[1]; F[3 * i + 2] -= q * fz * K2 / o->box[2];
        }
    }
    free(idx); free(w); free(dw);
    /* self term, neutralising background */
    E -= ONE_4PI_EPS0 * alpha / sqrt(M_PI) * q2sum;
    E -= M_PI * ONE_4PI_EPS0 * qsum * qsum / (2.0 * V * alpha * alpha);
    /* excluded pairs: their reciprocal-space interaction is taken out again */
    for (int i = 0; i < n; i++) {
        const double qi = o->is_alch[i] ? 0.0 : o->charge[i];
        if (qi == 0.0) continue;
        for (int e = o->excl_start[i]; e < o->excl_start[i + 1]; e++) {
            const int j = o->excl_list[e];
            if (j <= i) continue;
            const double qj = o->is_alch[j] ? 0.0 : o->charge[j];
            if (qj == 0.0) continue;
            double d[3] = {o->x[3 * i] - o->x[3 * j], o->x[3 * i + 1] - o->x[3 * j + 1], o->x[3 * i + 2] - o->x[3 * j + 2]};
            min_image(o, d);
            const double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], r = sqrt(r2), pre = ONE_4PI_EPS0 * qi * qj;
            const double er = erf(alpha * r);
            E -= pre * er / r;
            if (F) {
                /* -d/dr [ -pre erf(ar)/r ] / r along d */
                const double dEdr = -pre * (2.0 * alpha / sqrt(M_PI) * exp(-alpha * alpha * r2) / r - er / r2);
                const double fs = -dEdr / r;
                for (int k = 0; k < 3; k++) { F[3 * i + k] += fs * d[k]; F[3 * j + k] -= fs * d[k]; }
            }
        }
    }
    T[8] += E;
    if (o->disp_corr) T[9] += dispersion_correction(o);
}

/* exact Ewald reciprocal sum of the same charges (test helper: what the mesh approximates):
 * (2 pi ONE_4PI_EPS0 / V) sum_{k != 0} exp(-k^2 / 4 alpha^2) / k^2 |S(k)|^2 over k = 2 pi (m_x/L_x, ...) with |m_d| <= mmax */
double orc_ewald_reciprocal_exact(Oracle *o, int mmax) {
    const int n = o->n;
    const double V = o->box[0] * o->box[1] * o->box[2], alpha = o->alpha_ewald;
    double E = 0.0;
    for (int mx = -mmax; mx <= mmax; mx++) for (int my = -mmax; my <= mmax; my++) for (int mz = -mmax; mz <= mmax; mz++) {
        if (!mx && !my && !mz) continue;
        const double k[3] = {2.0 * M_PI * mx / o->box[0], 2.0 * M_PI * my / o->box[1], 2.0 * M_PI * mz / o->box[2]};
        const double k2 = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
        const double g = exp(-k2 / (4.0 * alpha * alpha)) / k2;
        if (g < 1e-16) continue;
        double sr = 0.0, si = 0.0;
        for (int i = 0; i < n; i++) {
            const double q = o->is_alch[i] ? 0.0 : o->charge[i];
            if (q == 0.0) continue;
            const double ph = k[0] * o->x[3 * i] + k[1] * o->x[3 * i + 1] + k[2] * o->x[3 * i + 2];
            sr += q * cos(ph); si += q * sin(ph);
        }
        E += g * (sr * sr + si * si);
    }
    return 2.0 * M_PI * ONE_4PI_EPS0 / V * E;
}

double orc_energy_forces(Oracle *o, double ls, double le, double *forces, double *terms) {
    double T[BLUES_N_ENERGY_TERMS] = {0};
    if (forces) memset(forces, 0, sizeof(double) * 3 * o->n);
#ifdef ORC_OPENMP
    static int prof = -1; if (prof < 0) prof = getenv("ORC_PROFILE") != NULL;
    double t0 = omp_get_wtime();
#endif
    bonded(o, forces, T);
#ifdef ORC_OPENMP
    double t1 = omp_get_wtime();
#endif
    nonbonded(o, ls, le, forces, T);
#ifdef ORC_OPENMP
    double t2 = omp_get_wtime();
#endif
    exceptions(o, ls, le, forces, T);
#ifdef ORC_OPENMP
    if (prof) fprintf(stderr, "[orc] bonded %.1f ms  nonbonded %.1f ms  exceptions %.1f ms\n", 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (omp_get_wtime() - t2));
#endif
    if (o->nb_method == BLUES_NB_PME) reciprocal_space(o, forces, T);
    double E = 0.0;
    for (int t = 0; t < BLUES_N_ENERGY_TERMS; t++) E += T[t];
    if (terms) memcpy(terms, T, sizeof T);
    o->n_evals++;
    return E;
}

/* "energy" / "f" of the CustomIntegrator: cached while (x, parameters) are unchanged */
static double cached_energy(Oracle *o) {
    if (o->cache_valid && o->cache_xver == o->xver && o->cache_ls == o->g.lambda_sterics && o->cache_le == o->g.lambda_electrostatics)
        return o->cache_E;
    o->cache_E = orc_energy_forces(o, o->g.lambda_sterics, o->g.lambda_electrostatics, o->f, NULL);
    o->cache_xver = o->xver; o->cache_ls = o->g.lambda_sterics; o->cache_le = o->g.lambda_electrostatics; o->cache_valid = 1;
    return o->cache_E;
}

double orc_potential_energy(Oracle *o) { return cached_energy(o); }
void orc_get_forces(Oracle *o, double *forces) { cached_energy(o); memcpy(forces, o->f, sizeof(double) * 3 * o->n); }

double orc_kinetic_energy(Oracle *o) {
    double ke = 0.0;
    for (int i = 0; i < o->n; i++) if (o->mass[i] != 0.0)
        ke += 0.5 * o->mass[i] * (o->v[3 * i] * o->v[3 * i] + o->v[3 * i + 1] * o->v[3 * i + 1] + o->v[3 * i + 2] * o->v[3 * i + 2]);
    return ke;
}

/* ------------------------------------------------------------------ constraints
 * Cluster-local solves (HBonds / rigid water couple at most 4 atoms through at most 3 constraints):
 *   velocities (RATTLE): the condition (v_i - v_j).r_ij = 0 is linear in the multipliers -> one k x k solve;
 *   positions  (SHAKE):  Newton on the k multipliers with displacement directions taken from the last
 *                        constrained geometry (xref), as OpenMM's constraint kernels do; iterated until
 *                        |r^2 - d^2| <= 2 tol d^2 (tol from blues/integrators.py:104) plus one polishing
 *                        iteration, so the result is converged far below the tolerance.
 * Clusters with more than 3 constraints fall back to Gauss-Seidel sweeps. */
static int solve_small(int n, double M[3][3], const double b[3], double x[3]) {
    if (n == 1) { x[0] = b[0] / M[0][0]; return 0; }
    if (n == 2) {
        double det = M[0][0] * M[1][1] - M[0][1] * M[1][0];
        x[0] = (b[0] * M[1][1] - M[0][1] * b[1]) / det;
        x[1] = (M[0][0] * b[1] - b[0] * M[1][0]) / det;
        return 0;
    }
    double c00 = M[1][1] * M[2][2] - M[1][2] * M[2][1], c01 = M[1][2] * M[2][0] - M[1][0] * M[2][2], c02 = M[1][0] * M[2][1] - M[1][1] * M[2][0];
    double det = M[0][0] * c00 + M[0][1] * c01 + M[0][2] * c02;
    double inv = 1.0 / det;
    x[0] = (b[0] * c00 + b[1] * (M[0][2] * M[2][1] - M[0][1] * M[2][2]) + b[2] * (M[0][1] * M[1][2] - M[0][2] * M[1][1])) * inv;
    x[1] = (b[0] * c01 + b[1] * (M[0][0] * M[2][2] - M[0][2] * M[2][0]) + b[2] * (M[0][2] * M[1][0] - M[0][0] * M[1][2])) * inv;
    x[2] = (b[0] * c02 + b[1] * (M[0][1] * M[2][0] - M[0][0] * M[2][1]) + b[2] * (M[0][0] * M[1][1] - M[0][1] * M[1][0])) * inv;
    return 0;
}

/* coupling coefficient of multiplier c2 in constraint c: w_i s(c2,i) - w_j s(c2,j) */
static inline double coupling(const Oracle *o, int c, int c2) {
    int i = o->cons_atoms[2 * c], j = o->cons_atoms[2 * c + 1], i2 = o->cons_atoms[2 * c2], j2 = o->cons_atoms[2 * c2 + 1];
    double si = (i == i2) - (i == j2), sj = (j == i2) - (j == j2);
    return si / o->mass[i] - sj / o->mass[j];
}

static int shake_gauss_seidel(Oracle *o, int cl) {
    int it;
    for (it = 0; it < 500; it++) {
        int done = 1;
        for (int q = o->cluster_start[cl]; q < o->cluster_start[cl + 1]; q++) {
            int c = o->cluster_cons[q], i = o->cons_atoms[2 * c], j = o->cons_atoms[2 * c + 1];
            double d2 = o->cons_dist[c] * o->cons_dist[c];
            double rp[3] = {o->x[3 * i] - o->x[3 * j], o->x[3 * i + 1] - o->x[3 * j + 1], o->x[3 * i + 2] - o->x[3 * j + 2]};
            min_image(o, rp);
            double diff = d2 - dot(rp, rp);
            if (fabs(diff) <= 2.0 * o->tol * d2) continue;
            done = 0;
            double r[3] = {o->xref[3 * i] - o->xref[3 * j], o->xref[3 * i + 1] - o->xref[3 * j + 1], o->xref[3 * i + 2] - o->xref[3 * j + 2]};
            min_image(o, r);
            double wi = 1.0 / o->mass[i], wj = 1.0 / o->mass[j];
            double delta = diff / (2.0 * dot(rp, r) * (wi + wj));
            for (int k = 0; k < 3; k++) { o->x[3 * i + k] += wi * delta * r[k]; o->x[3 * j + k] -= wj * delta * r[k]; }
        }
        if (done) break;
    }
    return it == 500;
}

static int constrain_positions(Oracle *o) {
    int fail = 0;
    for (int cl = 0; cl < o->n_clusters; cl++) {
        const int q0 = o->cluster_start[cl], nc = o->cluster_start[cl + 1] - q0;
        if (nc > 3) { fail |= shake_gauss_seidel(o, cl); continue; }
        double r[3][3];
        for (int a = 0; a < nc; a++) {
            int c = o->cluster_cons[q0 + a], i = o->cons_atoms[2 * c], j = o->cons_atoms[2 * c + 1];
            for (int k = 0; k < 3; k++) r[a][k] = o->xref[3 * i + k] - o->xref[3 * j + k];
            min_image(o, r[a]);
        }
        int it;
        for (it = 0; it < 50; it++) {
            double D[3][3], g[3], J[3][3], dl[3];
            int conv = 1, tight = 1;
            for (int a = 0; a < nc; a++) {
                int c = o->cluster_cons[q0 + a], i = o->cons_atoms[2 * c], j = o->cons_atoms[2 * c + 1];
                double d2 = o->cons_dist[c] * o->cons_dist[c];
                for (int k = 0; k < 3; k++) D[a][k] = o->x[3 * i + k] - o->x[3 * j + k];
                min_image(o, D[a]);
                g[a] = -(dot(D[a], D[a]) - d2);
                if (fabs(g[a]) > 2.0 * o->tol * d2) conv = 0;
                if (fabs(g[a]) > 1e-13 * d2) tight = 0;
            }
            if (tight) break; /* already converged to rounding: nothing to polish */
            for (int a = 0; a < nc; a++) for (int b = 0; b < nc; b++)
                J[a][b] = 2.0 * dot(D[a], r[b]) * coupling(o, o->cluster_cons[q0 + a], o->cluster_cons[q0 + b]);
            solve_small(nc, J, g, dl);
            for (int b = 0; b < nc; b++) {
                int c = o->cluster_cons[q0 + b], i = o->cons_atoms[2 * c], j = o->cons_atoms[2 * c + 1];
                for (int k = 0; k < 3; k++) { o->x[3 * i + k] += dl[b] * r[b][k] / o->mass[i]; o->x[3 * j + k] -= dl[b] * r[b][k] / o->mass[j]; }
            }
            if (conv) break; /* the update just applied was the polishing iteration */
        }
        if (it == 50) fail = 1;
    }
    memcpy(o->xref, o->x, sizeof(double) * 3 * o->n);
    o->xver++;
    return fail;
}

static void constrain_velocities(Oracle *o) {
    for (int cl = 0; cl < o->n_clusters; cl++) {
        const int q0 = o->cluster_start[cl], nc = o->cluster_start[cl + 1] - q0;
        if (nc > 3) {
            for (int it = 0; it < 500; it++) {
                int done = 1;
                for (int q = q0; q < q0 + nc; q++) {
                    int c = o->cluster_cons[q], i = o->cons_atoms[2 * c], j = o->cons_atoms[2 * c + 1];
                    double r[3] = {o->x[3 * i] - o->x[3 * j], o->x[3 * i + 1] - o->x[3 * j + 1], o->x[3 * i + 2] - o->x[3 * j + 2]};
                    min_image(o, r);
                    double rv[3] = {o->v[3 * i] - o->v[3 * j], o->v[3 * i + 1] - o->v[3 * j + 1], o->v[3 * i + 2] - o->v[3 * j + 2]};
                    double wi = 1.0 / o->mass[i], wj = 1.0 / o->mass[j];
                    double delta = -dot(rv, r) / (dot(r, r) * (wi + wj));
                    if (fabs(delta) <= o->tol) continue;
                    done = 0;
                    for (int k = 0; k < 3; k++) { o->v[3 * i + k] += wi * delta * r[k]; o->v[3 * j + k] -= wj * delta * r[k]; }
                }
                if (done) break;
            }
            continue;
        }
        double r[3][3], M[3][3], b[3], mu[3];
        for (int a = 0; a < nc; a++) {
            int c = o->cluster_cons[q0 + a], i = o->cons_atoms[2 * c], j = o->cons_atoms[2 * c + 1];
            double rv[3];
            for (int k = 0; k < 3; k++) { r[a][k] = o->x[3 * i + k] - o->x[3 * j + k]; rv[k] = o->v[3 * i + k] - o->v[3 * j + k]; }
            min_image(o, r[a]);
            b[a] = -dot(rv, r[a]);
        }
        for (int a = 0; a < nc; a++) for (int c2 = 0; c2 < nc; c2++)
            M[a][c2] = dot(r[a], r[c2]) * coupling(o, o->cluster_cons[q0 + a], o->cluster_cons[q0 + c2]);
        solve_small(nc, M, b, mu);
        for (int a = 0; a < nc; a++) {
            int c = o->cluster_cons[q0 + a], i = o->cons_atoms[2 * c], j = o->cons_atoms[2 * c + 1];
            for (int k = 0; k < 3; k++) { o->v[3 * i + k] += mu[a] * r[a][k] / o->mass[i]; o->v[3 * j + k] -= mu[a] * r[a][k] / o->mass[j]; }
        }
    }
}

void orc_set_velocities_to_temperature(Oracle *o, double T, uint64_t seed) {
    double kT = KB_KJ * T;
    for (int i = 0; i < o->n; i++) {
        double g[3] = {0, 0, 0};
        if (o->mass[i] != 0.0) { orc_gaussians(seed, o->replica * 4u + 1u, 0, (uint32_t)i, g); double s = sqrt(kT / o->mass[i]); g[0] *= s; g[1] *= s; g[2] *= s; }
        o->v[3 * i] = g[0]; o->v[3 * i + 1] = g[1]; o->v[3 * i + 2] = g[2];
    }
    constrain_velocities(o);
}

/* ------------------------------------------------------------------ integrator substeps */
static void remove_cm_motion(Oracle *o) {
    double p[3] = {0, 0, 0}, M = 0.0;
    for (int i = 0; i < o->n; i++) { double m = o->mass[i]; M += m; for (int k = 0; k < 3; k++) p[k] += m * o->v[3 * i + k]; }
    if (M == 0.0) return;
    for (int i = 0; i < o->n; i++) if (o->mass[i] != 0.0) for (int k = 0; k < 3; k++) o->v[3 * i + k] -= p[k] / M;
}

static void step_V(Oracle *o) {
    cached_energy(o); /* forces at the current (x, lambda) */
    double h = o->dt / o->n_V;
    for (int i = 0; i < o->n; i++) if (o->mass[i] != 0.0) for (int k = 0; k < 3; k++) o->v[3 * i + k] += h * o->f[3 * i + k] / o->mass[i];
    constrain_velocities(o);
}

static void step_R(Oracle *o) {
    double h = o->dt / o->n_R;
    int n3 = 3 * o->n;
    double *x1 = malloc(sizeof(double) * n3);
    for (int i = 0; i < o->n; i++) if (o->mass[i] != 0.0) for (int k = 0; k < 3; k++) o->x[3 * i + k] += h * o->v[3 * i + k];
    memcpy(x1, o->x, sizeof(double) * n3);
    constrain_positions(o);
    for (int i = 0; i < o->n; i++) if (o->mass[i] != 0.0) for (int k = 0; k < 3; k++) o->v[3 * i + k] += (o->x[3 * i + k] - x1[3 * i + k]) / h;
    constrain_velocities(o);
    free(x1);
}

static void step_O(Oracle *o) {
    double h = o->dt / (o->n_O > 0 ? o->n_O : 1);
    double a = exp(-o->gamma * h), b = sqrt(1.0 - exp(-2.0 * o->gamma * h));
    double ke0 = orc_kinetic_energy(o);
    for (int i = 0; i < o->n; i++) if (o->mass[i] != 0.0) {
        double g[3]; orc_gaussians(o->seed, o->replica * 4u, o->draw, (uint32_t)i, g);
        double s = sqrt(o->kT / o->mass[i]);
        for (int k = 0; k < 3; k++) o->v[3 * i + k] = a * o->v[3 * i + k] + b * s * g[k];
    }
    o->draw++;
    constrain_velocities(o);
    o->g.heat += orc_kinetic_energy(o) - ke0;
}

/* OpenMM LangevinIntegrator step (the MD leg, reference blues/simulation.py:647):
 *   v' = a v + (1-a)/gamma f/m + sqrt(kT (1-a^2)/m) xi ;  x' = x + dt v' ; constrain x' ; v = (x'-x)/dt */
static void step_L(Oracle *o) {
    cached_energy(o);
    const double a = exp(-o->gamma * o->dt), fs = o->gamma == 0.0 ? o->dt : (1.0 - a) / o->gamma, ns = sqrt(1.0 - a * a);
    int n3 = 3 * o->n;
    double *x0 = malloc(sizeof(double) * n3);
    memcpy(x0, o->x, sizeof(double) * n3);
    for (int i = 0; i < o->n; i++) if (o->mass[i] != 0.0) {
        double g[3]; orc_gaussians(o->seed, o->replica * 4u, o->draw, (uint32_t)i, g);
        double w = 1.0 / o->mass[i], sd = sqrt(o->kT * w);
        for (int k = 0; k < 3; k++) {
            o->v[3 * i + k] = a * o->v[3 * i + k] + fs * w * o->f[3 * i + k] + ns * sd * g[k];
            o->x[3 * i + k] += o->dt * o->v[3 * i + k];
        }
    }
    o->draw++;
    memcpy(o->xref, x0, sizeof(double) * n3);
    constrain_positions(o);
    for (int i = 0; i < o->n; i++) if (o->mass[i] != 0.0) for (int k = 0; k < 3; k++) o->v[3 * i + k] = (o->x[3 * i + k] - x0[3 * i + k]) / o->dt;
    free(x0);
}

void orc_set_custom_pair_mode(Oracle *o, int mode) { o->custom_pair_mode = mode; o->cache_valid = 0; }
int orc_add_centroid_bond(Oracle *o, int n1, const int *i1, const double *w1, int n2, const int *i2, const double *w2, double k) {
    if (o->n_centroid >= 4 || n1 > 8 || n2 > 8) return 1;
    int b = o->n_centroid++;
    o->cb_n1[b] = n1; o->cb_n2[b] = n2; o->cb_k[b] = k;
    for (int q = 0; q < n1; q++) { o->cb_i1[b][q] = i1[q]; o->cb_w1[b][q] = w1[q]; }
    for (int q = 0; q < n2; q++) { o->cb_i2[b][q] = i2[q]; o->cb_w2[b][q] = w2[q]; }
    o->cache_valid = 0;
    return 0;
}

static void update_alchemical_parameters(Oracle *o, int lambda_step) {
    o->g.lambda_sterics = o->tab_ls[lambda_step];
    o->g.lambda_electrostatics = o->tab_le[lambda_step];
}

/* reference blues/integrators.py:211-231 */
static void step_H(Oracle *o) {
    if (o->g.prop != 1.0) return;
    o->g.debug += 1;
    o->g.Eold = cached_energy(o);
    int ls = (int)o->g.lambda_step;
    o->g.lambda = (o->g.lambda_step + 1.0) / o->g.n_lambda_steps;
    o->g.lambda_step += 1.0;
    update_alchemical_parameters(o, ls + 1 <= o->n_lambda_steps ? ls + 1 : o->n_lambda_steps);
    o->g.Enew = cached_energy(o);
    o->g.protocol_work += o->g.Enew - o->g.Eold;
}

static void splitting_pass(Oracle *o) {
    if (o->remove_cm) remove_cm_motion(o); /* addUpdateContextState at the head of each pass */
    for (int s = 0; s < o->n_split; s++) {
        switch (o->split[s]) {
        case 'V': step_V(o); break;
        case 'R': step_R(o); break;
        case 'O': step_O(o); break;
        case 'H': step_H(o); break;
        case 'L': step_L(o); break;
        }
    }
}

/* ------------------------------------------------------------------ reference blues/switching.py (dead code there)
 * Work is accumulated in kJ/mol here (the reference divides every increment by kT). */
static double uniform_for_trial(const Oracle *o, uint32_t trial) {
    uint32_t r[4];
    orc_philox4x32(trial, 0u, o->replica * 4u + 2u, 0x47484D43u, (uint32_t)o->seed, (uint32_t)(o->seed >> 32), r);
    return ((double)r[0] + 0.5) * 2.3283064365386963e-10;
}

/* addVelocityVerletStep, switching.py:934-959, without its energy bookkeeping */
static void velocity_verlet(Oracle *o) {
    const int n3 = 3 * o->n;
    const double dt = o->dt;
    cached_energy(o);   /* f at the current (x, parameters) */
    for (int i = 0; i < o->n; i++) if (o->mass[i] != 0.0) for (int k = 0; k < 3; k++) o->v[3 * i + k] += 0.5 * dt * o->f[3 * i + k] / o->mass[i];
    memcpy(o->xref, o->x, sizeof(double) * n3);
    for (int i = 0; i < o->n; i++) if (o->mass[i] != 0.0) for (int k = 0; k < 3; k++) o->x[3 * i + k] += dt * o->v[3 * i + k];
    o->xver++;
    double *x1 = malloc(sizeof(double) * n3);
    memcpy(x1, o->x, sizeof(double) * n3);
    constrain_positions(o);
    cached_energy(o);   /* f at the new positions */
    for (int i = 0; i < o->n; i++) if (o->mass[i] != 0.0) for (int k = 0; k < 3; k++)
        o->v[3 * i + k] += 0.5 * dt * o->f[3 * i + k] / o->mass[i] + (o->x[3 * i + k] - x1[3 * i + k]) / dt;
    constrain_velocities(o);
    free(x1);
}

/* addVelocityVerletStep with the shadow work, switching.py:941-959 */
static void vv_with_shadow_work(Oracle *o) {
    if (o->remove_cm) remove_cm_motion(o);   /* addUpdateContextState, switching.py:939 */
    const double Eold = cached_energy(o) + orc_kinetic_energy(o);
    velocity_verlet(o);
    const double Enew = cached_energy(o) + orc_kinetic_energy(o);
    o->g.shadow_work += Enew - Eold;
    o->g.Epert = cached_energy(o);
}

/* velocity randomisation of addGHMCStep: v = sqrt(b) v + sqrt(1-b) sigma xi, b = exp(-gamma dt); constrain (switching.py:980-981, 1008-1009) */
static void ghmc_randomise(Oracle *o) {
    const double b = exp(-o->gamma * o->dt), sb = sqrt(b), s1b = sqrt(1.0 - b);
    for (int i = 0; i < o->n; i++) if (o->mass[i] != 0.0) {
        double g[3]; orc_gaussians(o->seed, o->replica * 4u, o->draw, (uint32_t)i, g);
        const double sg = sqrt(o->kT / o->mass[i]);
        for (int k = 0; k < 3; k++) o->v[3 * i + k] = sb * o->v[3 * i + k] + s1b * sg * g[k];
    }
    o->draw++;
    constrain_velocities(o);
}

/* addGHMCStep, switching.py:961-1017.  The reference sums the new kinetic energy into a variable `ke` it never declares and then
 * forms Enew from the OLD `kinetic` (switching.py:997-998) -- OpenMM would refuse that program; the evident intent (Enew = new
 * kinetic + energy) is what is restated here. */
static void ghmc_step(Oracle *o) {
    const int n3 = 3 * o->n;
    if (o->remove_cm) remove_cm_motion(o);   /* addUpdateContextState, switching.py:976 */
    ghmc_randomise(o);
    const double Eold = orc_kinetic_energy(o) + o->g.Epert;
    double *xold = malloc(sizeof(double) * n3), *vold = malloc(sizeof(double) * n3);
    memcpy(xold, o->x, sizeof(double) * n3); memcpy(vold, o->v, sizeof(double) * n3);
    velocity_verlet(o);
    const double Enew = orc_kinetic_energy(o) + cached_energy(o);
    const double u = uniform_for_trial(o, o->ghmc_draw++);   /* (keyed on ntrials, which reset() zeroes, every switch would meet the same thresholds) */
    o->g.accept = (exp(-(Enew - Eold) / o->kT) - u >= 0.0) ? 1.0 : 0.0;   /* Lepton step(x) = 1 for x >= 0 */
    if (o->g.accept != 1.0) {
        memcpy(o->x, xold, sizeof(double) * n3); o->xver++;
        for (int q = 0; q < n3; q++) o->v[q] = -vold[q];
    }
    free(xold); free(vold);
    ghmc_randomise(o);
    o->g.naccept += o->g.accept; o->g.ntrials += 1.0;
}

/* NCMCVVAlchemicalIntegrator (switching.py:1173-1241) / NCMCGHMCAlchemicalIntegrator (:1318-1360), nsteps > 0: one integrator step */
static void switching_step(Oracle *o) {
    Globals *g = &o->g;
    if (g->step == 0.0) {
        g->Epert = cached_energy(o);
        memcpy(o->xref, o->x, sizeof(double) * 3 * o->n);
        constrain_positions(o);
        constrain_velocities(o);
        g->total_work = 0.0; g->protocol_work = 0.0; g->shadow_work = 0.0;   /* addWorkResetStep */
        g->lambda_step = 0.0; update_alchemical_parameters(o, 0);             /* addAlchemicalResetStep: table entry 0 = the initial state */
        g->initial_energy = cached_energy(o);
        g->Epert = g->initial_energy;   /* (the reference takes Epert before the reset, switching.py:1211, and not at all for nsteps = 0: the first GHMC
                                           test / the toggle's work would then be measured from another state's energy) */
        if (o->switch_mode == BLUES_SWITCH_VV) for (int p = 0; p < o->psteps; p++) vv_with_shadow_work(o);
        else if (o->psteps > 0) { ghmc_step(o); g->Epert = cached_energy(o); }   /* (the reference leaves Epert stale here, switching.py:1341-1348: the first
                                                                 perturbation would book the GHMC move's energy change as protocol work) */
    }
    if (g->step < g->nsteps) {
        /* addAlchemicalPerturbationStep, switching.py:881-907 */
        g->Eold = g->Epert;
        g->lambda_step = g->step + 1.0; g->lambda = g->lambda_step / g->nsteps;
        update_alchemical_parameters(o, (int)g->lambda_step);
        g->Enew = cached_energy(o);
        g->protocol_work += g->Enew - g->Eold;
        if (o->switch_mode == BLUES_SWITCH_VV) for (int p = 0; p < o->psteps; p++) vv_with_shadow_work(o);
        else if (o->psteps > 0) { ghmc_step(o); g->Epert = cached_energy(o); }
        g->step += 1.0;
        g->total_work = g->protocol_work + g->shadow_work;   /* addComputeTotalWorkStep */
        g->final_energy = cached_energy(o);
    }
}

/* reference blues/integrators.py:159-209, SURVEY.md Appendix A */
int orc_step(Oracle *o, int nsteps) {
    if (o->switch_mode != BLUES_SWITCH_NONE) {
        for (int s = 0; s < nsteps; s++) {
            switching_step(o);
            for (int i = 0; i < 3 * o->n; i++) if (isnan(o->x[i])) return 1;
        }
        return 0;
    }
    for (int s = 0; s < nsteps; s++) {
        Globals *g = &o->g;
        if (g->step == 0.0) {
            g->perturbed_pe = cached_energy(o);
            g->unperturbed_pe = cached_energy(o);
            memcpy(o->xref, o->x, sizeof(double) * 3 * o->n);
            constrain_positions(o);
            constrain_velocities(o);
            g->protocol_work = 0.0;
            g->lambda = 0.0; g->protocol_work = 0.0; g->step = 0.0; g->lambda_step = 0.0;
            update_alchemical_parameters(o, 0);
        }
        if (g->step < g->nsteps) {
            g->perturbed_pe = cached_energy(o);
            if (g->first_step < 1.0) { g->first_step = 1.0; g->unperturbed_pe = cached_energy(o); }
            g->protocol_work += g->perturbed_pe - g->unperturbed_pe;
            memcpy(o->xref, o->x, sizeof(double) * 3 * o->n);
            splitting_pass(o);
            if (g->lambda > g->prop_lambda_min && g->lambda <= g->prop_lambda_max)
                while (g->prop < g->nprop) { g->prop += 1.0; splitting_pass(o); }
            g->unperturbed_pe = cached_energy(o);
            g->step += 1.0;
            g->prop = 1.0;
        }
        for (int i = 0; i < 3 * o->n; i++) if (isnan(o->x[i])) return 1;
    }
    return 0;
}

/* reference blues/integrators.py:240-249 plus the base-class reset of lambda_step */
void orc_reset(Oracle *o) {
    Globals *g = &o->g;
    g->step = 0; g->lambda = 0; g->protocol_work = 0; g->shadow_work = 0; g->first_step = 0;
    g->perturbed_pe = 0; g->unperturbed_pe = 0; g->prop = 1; g->lambda_step = 0;
    g->total_work = 0; g->initial_energy = 0; g->final_energy = 0; g->naccept = 0; g->ntrials = 0;   /* switching.py:1023-1036 */
}

#define GLOBAL_LIST(X) \
    X(lambda) X(step) X(lambda_step) X(n_lambda_steps) X(nsteps) X(protocol_work) X(shadow_work) X(perturbed_pe) \
    X(unperturbed_pe) X(first_step) X(nprop) X(prop) X(prop_lambda_min) X(prop_lambda_max) X(Eold) X(Enew) X(heat) \
    X(debug) X(lambda_sterics) X(lambda_electrostatics) X(Epert) X(total_work) X(initial_energy) X(final_energy) X(accept) X(naccept) X(ntrials)

int orc_get_global(Oracle *o, const char *name, double *value) {
#define X(f) if (!strcmp(name, #f)) { *value = o->g.f; return 0; }
    GLOBAL_LIST(X)
#undef X
    if (!strcmp(name, "kT")) { *value = o->kT; return 0; }
    return 1;
}

int orc_set_global(Oracle *o, const char *name, double value) {
#define X(f) if (!strcmp(name, #f)) { o->g.f = value; return 0; }
    GLOBAL_LIST(X)
#undef X
    return 1;
}

/* ------------------------------------------------------------------ fixture preparation */
double orc_minimize(Oracle *o, int max_iter, double step0) {
    int n3 = 3 * o->n;
    double *xold = malloc(sizeof(double) * n3);
    double step = step0;
    memcpy(o->xref, o->x, sizeof(double) * n3);
    constrain_positions(o);
    double E = orc_energy_forces(o, 1.0, 1.0, o->f, NULL);
    for (int it = 0; it < max_iter; it++) {
        double fmax = 0.0;
        for (int i = 0; i < o->n; i++) if (o->mass[i] != 0.0) for (int k = 0; k < 3; k++) fmax = fmax > fabs(o->f[3 * i + k]) ? fmax : fabs(o->f[3 * i + k]);
        if (fmax == 0.0) break;
        memcpy(xold, o->x, sizeof(double) * n3);
        for (int i = 0; i < o->n; i++) if (o->mass[i] != 0.0) for (int k = 0; k < 3; k++) o->x[3 * i + k] += step * o->f[3 * i + k] / fmax;
        memcpy(o->xref, xold, sizeof(double) * n3);
        constrain_positions(o);
        double *fnew = malloc(sizeof(double) * n3);
        double En = orc_energy_forces(o, 1.0, 1.0, fnew, NULL);
        if (En < E) { E = En; memcpy(o->f, fnew, sizeof(double) * n3); step *= 1.2; }
        else { memcpy(o->x, xold, sizeof(double) * n3); memcpy(o->xref, xold, sizeof(double) * n3); o->xver++; step *= 0.5; }
        free(fnew);
        if (step < 1e-7) break;
    }
    free(xold);
    o->cache_valid = 0;
    return E;
}
