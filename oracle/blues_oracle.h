/*
 * blues_oracle.h -- CPU (fp64, single-thread C) restatement of the BLUES NCMC
 * switching path.  TEST INFRASTRUCTURE ONLY: nothing under blues_amd/ may link,
 * import or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do, and only as the checker / the timed CPU baseline.
 *
 * PARITY STATUS: "parity unpinned" for forces, energies and per-step protocol
 * work -- the reference holds no golden vectors for them and its arithmetic
 * lives in un-vendored third-party packages that are absent from this image:
 * openmmtools==0.15.0 (reference devtools/conda-recipe/meta.yaml:42) and
 * OpenMM 7.4.2 (reference README.md:49).  What IS pinned, against values the
 * reference's own files state (tests/golden/reference_known_answers.json):
 * the step-program structure (blues/integrators.py:159-231), the lambda
 * schedules (blues/simulation.py:654-659), _get_prop_lambda
 * (integrators.py:147-157), calculateNCMCSteps (blues/utils.py:89-145), the
 * integrator attribute values asserted by blues/tests/test_simulation.py:262-289,
 * and the unit constants.  The third-party functional forms follow their
 * published definitions and are checked by analytic
 * two-body values and finite differences.
 */
#ifndef BLUES_ORACLE_H
#define BLUES_ORACLE_H

#include "../include/blues_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct Oracle Oracle;

Oracle *orc_create(const BluesSystemDesc *sys, const BluesIntegratorDesc *integ);
void orc_destroy(Oracle *o);
const char *orc_last_error(void);

void orc_set_positions(Oracle *o, const double *xyz);
void orc_set_velocities(Oracle *o, const double *xyz);
void orc_get_positions(Oracle *o, double *xyz);
void orc_get_velocities(Oracle *o, double *xyz);
void orc_set_box(Oracle *o, const double box[9]);
void orc_set_velocities_to_temperature(Oracle *o, double temperature, uint64_t seed);

/* potential energy (kJ/mol) at the current positions and explicit lambdas;
 * forces [n][3] and the BLUES_N_ENERGY_TERMS breakdown are optional */
double orc_energy_forces(Oracle *o, double lambda_sterics, double lambda_electrostatics, double *forces,
                         double *terms);
/* same at the integrator's current alchemical parameters */
double orc_potential_energy(Oracle *o);
double orc_kinetic_energy(Oracle *o);
void orc_get_forces(Oracle *o, double *forces);

/* AlchemicalExternalLangevinIntegrator.step(n); returns 0, or 1 on NaN positions */
int orc_step(Oracle *o, int n);
int orc_get_global(Oracle *o, const char *name, double *value);
int orc_set_global(Oracle *o, const char *name, double value);
void orc_reset(Oracle *o);
/* number of full (x, lambda) energy/force evaluations performed so far */
long orc_num_evaluations(Oracle *o);

/* oracle-only extras needed to restate the reference's ethylene known-answer system
 * (blues/tests/data/ethylene_system.xml: CustomNonbondedForce interaction group + CustomCentroidBondForce) */
void orc_set_custom_pair_mode(Oracle *o, int mode);
int orc_add_centroid_bond(Oracle *o, int n1, const int *i1, const double *w1, int n2, const int *i2, const double *w2, double k);

/* exact Ewald reciprocal-space energy of the NonbondedForce charges (what the PME mesh approximates), |m| <= mmax per axis */
double orc_ewald_reciprocal_exact(Oracle *o, int mmax);

/* plain steepest-descent relaxation with constraints (used to prepare fixtures) */
double orc_minimize(Oracle *o, int max_iter, double step0);

/* pieces exposed for pinning against golden values and for RNG parity with the GPU */
double orc_default_lambda_sterics(double lambda);
double orc_default_lambda_electrostatics(double lambda);
void orc_get_prop_lambda(double prop_lambda, double out[2]);
void orc_calculate_ncmc_steps(int nstepsNC, int nprop, double prop_lambda, int out[3]);
void orc_set_threads(int n);   /* threads of the OpenMP pair loop (ORC_OPENMP build; no-op otherwise) */
void orc_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]);
void orc_gaussians(uint64_t seed, uint32_t stream, uint32_t draw, uint32_t atom, double out[3]);
double orc_pair_energy(double r, double qq, double sigma, double epsilon, double alpha_ewald, int alchemical,
                       double lambda_s, double lambda_e, double softcore_alpha, double *dEdr);

#ifdef __cplusplus
}
#endif
#endif
