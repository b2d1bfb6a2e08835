/*
 * blues_engine.h -- C-ABI of the MI355X-native NCMC alchemical-Langevin engine.
 *
 * This is the drop-in boundary for the one hot path of MobleyLab/blues: the
 * switching trajectory that blues.simulation.BLUESSimulation._stepNCMC
 * (reference blues/simulation.py:1039-1098) drives through
 * blues.integrators.AlchemicalExternalLangevinIntegrator
 * (reference blues/integrators.py:98-249).  BLUES itself has no FFI: the
 * surface it consumes is the OpenMM Python object triple
 * Simulation / Context / CustomIntegrator.  Every entry point below names the
 * OpenMM-side call it replaces and the reference line that issues that call.
 *
 * Conventions
 *   - plain C, caller-allocated buffers, no torch / HIP types in signatures;
 *   - units are OpenMM's: nm, ps, amu, kJ/mol, elementary charge, kelvin;
 *   - xyz arrays are [n_atoms][3] row-major doubles in the caller's atom order;
 *   - every function returns 0 on success, nonzero on error; the message is
 *     available from blues_last_error(h) (or blues_last_error(NULL) for
 *     failures of blues_engine_create);
 *   - a handle is owned by one host thread at a time (OpenMM contexts have the
 *     same rule); handles on different devices are independent.
 *
 * The same two descriptor structs are consumed by the CPU oracle
 * (oracle/blues_oracle.h), so a parity test builds one description and hands
 * it to both sides.
 */
#ifndef BLUES_ENGINE_H
#define BLUES_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLUES_ABI_VERSION 6

/* nonbonded_method */
#define BLUES_NB_NOCUTOFF 0   /* oracle only: vacuum systems (vacDivaline, two-body checks) */
#define BLUES_NB_PME_DIRECT 1 /* periodic cutoff, erfc(alpha r)/r direct-space Coulomb + 12-6 LJ */
#define BLUES_NB_PME 2        /* the same plus what OpenMM's NonbondedForce adds under nonbondedMethod=PME (reference
                               * blues/simulation.py:219, examples/rotmove_cuda.yml:20): smooth-PME reciprocal space on
                               * pme_grid (order-pme_order B-splines), the Ewald self term, the erf correction of every
                               * excluded pair and, with dispersion_correction, the analytic long-range LJ correction.
                               * Alchemical atoms take part with charge 0 and epsilon 0, as
                               * alchemical_pme_treatment='direct-space' and disable_alchemical_dispersion_correction=True
                               * leave them in the NonbondedForce (reference blues/simulation.py:225-236). */

/*
 * Flat description of the alchemical System that
 * SystemFactory.generateAlchSystem builds (reference blues/simulation.py:221-317:
 * openmmtools AbsoluteAlchemicalFactory(alchemical_pme_treatment='direct-space',
 * disable_alchemical_dispersion_correction=True) over AlchemicalRegion(
 * softcore_alpha=0.5, a=b=1, c=6, beta=0, d=e=1, f=2, annihilate_electrostatics=True,
 * annihilate_sterics=False)), after freeze_radius / freeze_atoms
 * (simulation.py:364-480) zeroed the masses of frozen atoms and after
 * restrain_positions (simulation.py:319-362) added its CustomExternalForce.
 */
typedef struct BluesSystemDesc {
    int32_t n_atoms;
    double box[9];                 /* rows a,b,c (nm); only orthorhombic boxes: off-diagonals must be 0 */
    const double *mass;            /* [n] amu; 0 => frozen (utils.zero_masses, reference blues/utils.py:202-221) */
    const double *charge;          /* [n] e   */
    const double *sigma;           /* [n] nm  */
    const double *epsilon;         /* [n] kJ/mol */
    int32_t n_exclusions;          /* pairs with no regular nonbonded term (1-2, 1-3 and every exception pair) */
    const int32_t *exclusions;     /* [2*n_exclusions] */
    int32_t n_exceptions;          /* 1-4 style exceptions with their own parameters */
    const int32_t *exception_atoms;   /* [2*n_exceptions] */
    const double *exception_params;   /* [3*n_exceptions] chargeProd (e^2), sigma (nm), epsilon (kJ/mol) */
    int32_t n_bonds;
    const int32_t *bond_atoms;     /* [2*n] */
    const double *bond_params;     /* [2*n] r0 (nm), k (kJ/mol/nm^2); E = 0.5 k (r-r0)^2 */
    int32_t n_angles;
    const int32_t *angle_atoms;    /* [3*n] */
    const double *angle_params;    /* [2*n] theta0 (rad), k (kJ/mol/rad^2); E = 0.5 k (t-t0)^2 */
    int32_t n_torsions;
    const int32_t *torsion_atoms;  /* [4*n] */
    const double *torsion_params;  /* [3*n] periodicity, phase (rad), k (kJ/mol); E = k (1+cos(n phi - phase)) */
    int32_t n_constraints;
    const int32_t *constraint_atoms; /* [2*n] */
    const double *constraint_dist;   /* [n] nm */
    int32_t n_alchemical;
    const int32_t *alchemical_atoms; /* [n_alchemical] */
    int32_t n_restraints;          /* k_restr*periodicdistance(x,y,z,x0,y0,z0)^2, simulation.py:347 */
    const int32_t *restraint_atoms;
    const double *restraint_x0;    /* [3*n_restraints] nm */
    double restraint_k;            /* kJ/mol/nm^2 */
    int32_t nonbonded_method;      /* BLUES_NB_* */
    double cutoff;                 /* nm */
    double ewald_alpha;            /* nm^-1: sqrt(-ln(2 tol))/cutoff */
    double softcore_alpha;         /* 0.5 */
    int32_t annihilate_electrostatics; /* 1 */
    int32_t annihilate_sterics;        /* 0 */
    int32_t remove_cm_motion;      /* CMMotionRemover from removeCMMotion: True (examples/rotmove_cuda.yml:24) */
    /* BLUES_NB_PME only (ignored otherwise) */
    int32_t pme_grid[3];           /* mesh points along a,b,c: ceil(2 alpha L / (3 tol^(1/5))) for ewaldErrorTolerance tol */
    int32_t pme_order;             /* B-spline order; OpenMM uses 5 */
    int32_t dispersion_correction; /* NonbondedForce.getUseDispersionCorrection() (OpenMM's default: on) */
} BluesSystemDesc;

/*
 * Parameters of AlchemicalExternalLangevinIntegrator.__init__
 * (reference blues/integrators.py:98-145) as generateNCMCIntegrator passes
 * them (reference blues/simulation.py:650-705).  The Lepton strings in
 * alchemical_functions are evaluated by the host language into two tables
 * indexed by lambda_step: entry i is f(i / n_lambda_steps),
 * n_lambda_steps = nsteps_neq * (number of 'H' in splitting).
 */
typedef struct BluesIntegratorDesc {
    double timestep;               /* ps */
    double temperature;            /* K  */
    double collision_rate;         /* 1/ps; BLUES always ends up with 1.0 (simulation.py:664,697-704) */
    int32_t nsteps_neq;
    int32_t nprop;
    double prop_lambda_min;        /* _get_prop_lambda, integrators.py:147-157 */
    double prop_lambda_max;
    const char *splitting;         /* e.g. "H V R O R V H"; letters R V O H */
    int32_t n_lambda_steps;
    const double *lambda_sterics;        /* [n_lambda_steps+1] */
    const double *lambda_electrostatics; /* [n_lambda_steps+1] */
    double constraint_tolerance;   /* 1e-8, integrators.py:104 */
    uint64_t seed;                 /* Philox key; stream = replica */
    int32_t replica;
    int32_t precision;             /* 0 = mixed (f32 pair math, f64 accumulation), 1 = double */
    /* ABI 3: the switching integrators of reference blues/switching.py (dead code there: nothing imports the module).
     * BLUES_SWITCH_NONE: the program above.  Otherwise `splitting`, nprop and the prop window are ignored and one call of
     * integrator.step(1) runs the program of
     *   BLUES_SWITCH_VV   NCMCVVAlchemicalIntegrator   (switching.py:1083-1241): first step: constrain, reset, lambda table
     *                     entry 0, steps_per_propagation velocity-Verlet steps; every step: perturbation (protocol work), the
     *                     same number of velocity-Verlet steps, each adding its change of energy + kinetic to the shadow work;
     *   BLUES_SWITCH_GHMC NCMCGHMCAlchemicalIntegrator (switching.py:1244-1360): one GHMC step instead (velocity randomisation
     *                     with b = exp(-collision_rate * timestep), Metropolised velocity Verlet with momentum flip on rejection,
     *                     velocity randomisation), steps_per_propagation is not used, as in the reference.
     * timestep is the velocity-Verlet step; n_lambda_steps = nsteps_neq; the tables hold the functions at t = i / nsteps_neq
     * for direction 'insert', 1 - i / nsteps_neq for 'delete' (evaluated by the host language).  Work is kept in kJ/mol. */
    int32_t switching_mode;
    int32_t steps_per_propagation;
} BluesIntegratorDesc;
#define BLUES_SWITCH_NONE 0
#define BLUES_SWITCH_VV 1
#define BLUES_SWITCH_GHMC 2

typedef struct BluesEngine BluesEngine;

/* openmm.app.Simulation(topology, system, integrator, platform, properties)
 * (reference blues/simulation.py:730-737).  device = HIP device ordinal. */
int blues_engine_create(const BluesSystemDesc *sys, const BluesIntegratorDesc *integ, int device, BluesEngine **out);
int blues_engine_destroy(BluesEngine *h);
const char *blues_last_error(const BluesEngine *h);
int blues_abi_version(void);

/* ---- Launch policy (ABI 4) -------------------------------------------------
 * OpenMM takes such choices as platform properties (the `properties` dict of
 * openmm.app.Simulation, reference blues/simulation.py:730-737).  The engine
 * chooses every one of them itself; this struct exists so that tests and
 * profiling runs can pin a decomposition without touching the environment:
 * the library reads NO environment variables.  The process-wide tuning is copied
 * into every engine and batch when it is created; none of it changes results
 * beyond floating-point summation order (pruned lists, list margins) or nothing
 * at all (stream forks, graph replay, capacities).
 * -1 / 0 = "engine default" unless stated otherwise. */
typedef struct BluesTuning {
    int32_t struct_size;       /* sizeof(BluesTuning) of the caller: set by blues_tuning_default */
    int32_t plain_skin;        /* 1: every list candidate gets the margin `skin`, rebuild at skin/2 (plain Verlet rule) instead of
                                * the split margins for a mostly frozen system */
    double skin;               /* nm; outer margin of the neighbour lists; <= 0: by system size */
    double prune_margin;       /* nm; inner margin of the pruned per-atom lists; < 0: default (0.04 where most atoms are frozen);
                                * 0: the nonbonded kernel walks the full lists */
    double jcap_scale;         /* > 0: scales the list capacities (tests shrink them to reach the re-sort path) */
    double acap_scale;
    int32_t k1_mode;           /* -1 auto; 1: never the per-atom-list kernel nor fragment lists (sub-tile kernel in their place);
                                * 2: never fragment lists; 3: fragment lists (kernels_frag.h) wherever the engine can use them (mixed
                                * precision), also where most atoms are frozen or the system is a handful of tiles */
    int32_t list_group;        /* i-tiles per list of the per-atom-list kernel; 0 auto */
    int32_t sub_iw;            /* i-atoms per wave of the sub-tile kernel (8, 16, 32); 0 auto */
    int32_t sub_chunks;        /* its list chunks per i-tile; 0 auto */
    int32_t seg_len;           /* tile kernel: j-atoms per wave segment; 0 auto */
    int32_t waves_per_block;   /* tile kernel; 0 auto */
    int32_t k2_jiter;          /* j groups per environment block of the alchemical kernel; 0 auto */
    int32_t fuse_forces;       /* -1 auto; 0 / 1: separate force kernels / one launch with block roles */
    int32_t fuse_big;          /* -1 auto (off) */
    int32_t fast_step;         /* -1 auto (on): straight-line step kernels */
    int32_t slot_mask;         /* -1 auto; else the lambda slots whose force a pass produces */
    int32_t fork;              /* 1 (default): a batch runs the alchemical / bonded kernels on a side stream beside the builder of the atoms'
                                * lists -- of a pass with the dense alchemical kernel only the two small ones (alchemical x alchemical block,
                                * bonded entries); 0: every kernel alone; 2: the dense kernel too, joined before the nonbonded kernel (no gain);
                                * 3: every alchemical kernel and the bonded entries on the side stream, joined before the sums;
                                * 4: the small kernels and the dense kernel of the members that do not rebuild their lists beside the work list
                                * and the group-list builder (which leave the chip idle), the dense kernel of the others behind the group lists;
                                * all joined before the nonbonded kernel */
    int32_t use_graph;         /* -1 auto (off): hipGraph replay of the steady-state step of a lone engine */
    int32_t graph_units;       /* 0 auto */
    int32_t graph_fork;        /* -1 auto (off) */
    int32_t batch_sync_lists;  /* 1: every member of a batch rebuilds when one asks (drops batch = solo bitwise identity) */
    int32_t force_lists;       /* 1: rebuild the lists at every launch (development) */
    int32_t no_sphere;         /* 1: bounding-box list test only (development) */
    int32_t pme_general;       /* 1: reciprocal space through the global-memory kernel */
    int32_t debug_lists;       /* 1: print layout shapes, re-sorts and overflow diagnostics to stderr */
    int32_t assume_batch;      /* > 0: a lone engine lays itself out as a member of a batch of this many would (the decomposition
                                * depends on how many chains share a launch): what makes "a batch member equals the same
                                * chain advanced alone, bit for bit" testable */
    int32_t k1_threads;        /* threads per workgroup of the per-atom-list kernel (multiple of 64, <= 1024); 0: 1024 */
    int32_t k2_dense;          /* -1 auto (on where it applies: large batch, mixed precision, free alchemical group of <= 16 atoms);
                                * 0: the alchemical kernel keeps its (atom, list entry) lane layout; 2: the dense form with the fp64 pair
                                * arithmetic of round 5 (the reference the fp32 form of round 6 is tested against) */
    int32_t k2_early;          /* 1: with `fork`, the alchemical kernel of the members of a batch that do not rebuild their lists
                                * in a force pass starts beside the rebuild of the others; 0 (default): after the group lists */
    int32_t fuse_finalize;     /* 1 (default): where one workgroup holds every constraint cluster of a chain, the steady-state step kernel
                                * forms the summed forces of the pass itself (no k_finalize launch); 0: always the separate kernel */
    int32_t host_threads;      /* host threads for work the members of a batch share (re-sorts and re-layouts of several members at a poll);
                                * 0: the cores this process may use, at most 16.  N ranks on one host: cores / N (blues_amd/replicas.py) */
    int32_t pack_clusters;     /* order of the constraint clusters among the step kernel's threads.  1: each kind (star, single atom, triangle)
                                * starts on a wave boundary, so no wave runs two solvers (shortest chain per wave: a lone chain, small batches);
                                * 2: kinds packed back to back, largest first -- fewer waves per chain, one of them mixed (a batch of more than
                                * 512 chains, whose step kernel otherwise runs in two rounds at two 256-register waves per SIMD);
                                * 0: auto (2 where assume_batch > 512 and it saves waves) */
} BluesTuning;
void blues_tuning_default(BluesTuning *t);
/* NULL restores the defaults.  Applies to engines and batches created afterwards. */
int blues_set_tuning(const BluesTuning *t);
int blues_get_tuning(BluesTuning *t);

/* context.setPositions / setVelocities / setPeriodicBoxVectors
 * (reference blues/simulation.py:957-962, blues/moves.py:307). */
int blues_set_positions(BluesEngine *h, const double *xyz_nm, int32_t n_atoms);
int blues_set_velocities(BluesEngine *h, const double *xyz_nm_per_ps, int32_t n_atoms);
int blues_set_box(BluesEngine *h, const double box[9]);
/* context.getState(getPositions/getVelocities/getForces...) (simulation.py:904-910) */
int blues_get_positions(BluesEngine *h, double *xyz_nm, int32_t n_atoms);
int blues_get_velocities(BluesEngine *h, double *xyz_nm_per_ps, int32_t n_atoms);
int blues_get_forces(BluesEngine *h, double *xyz_kj_per_mol_nm, int32_t n_atoms);
int blues_get_box(BluesEngine *h, double box[9]);
/* context.setVelocitiesToTemperature(T) (simulation.py:743, 1187) */
int blues_set_velocities_to_temperature(BluesEngine *h, double temperature, uint64_t seed);
/* state.getPotentialEnergy() / getKineticEnergy() (simulation.py:908-909) */
int blues_get_energy(BluesEngine *h, double *potential, double *kinetic);
/* The potential energy of the current coordinates with lambda_sterics / lambda_electrostatics set to the given values, the
 * engine's own parameters left as they are: what the reference obtains from a second Context (`alch`, wrapping the MD system)
 * that it copies the coordinates into (simulation.py:1107-1110, 791-792).  For (1, 1) the value normally lies in the energy
 * cache already (every evaluation leaves it there); otherwise one evaluation. */
int blues_get_energy_at(BluesEngine *h, double lambda_sterics, double lambda_electrostatics, double *potential);
/* The reciprocal-space MESH energy 1/2 sum_m eterm(m) |Q^(m)|^2 of the current coordinates (0 unless BLUES_NB_PME), with the charges
 * of this System's NonbondedForce -- alchemical atoms 0 (alchemical_pme_treatment='direct-space') -- or, with_alchemical_charges != 0,
 * with every atom's own charge: the mesh energy of the NON-alchemical System the reference's md / alch contexts wrap
 * (blues/simulation.py:791-792).  Their difference is the x-dependent part of (U_alch - U_ncmc)(x) that no direct-space term holds:
 * what blues_amd/simulation.py forms the alchemical correction from (simulation.py:1100-1119) instead of four total energies, two of
 * them in another context.  Overwrites the reciprocal-space forces of the last pass (the next step evaluates a fresh one). */
int blues_mesh_energy(BluesEngine *h, int32_t with_alchemical_charges, double *out);
/* per-term potential energies at the current state, for parity tests:
 * [0] bonds [1] angles [2] torsions [3] nonbonded env-env [4] exceptions
 * [5] alchemical sterics [6] alchemical electrostatics [7] restraint
 * [8] Ewald reciprocal space (mesh + self term + erf corrections of excluded pairs + neutralising background)
 * [9] long-range dispersion correction; [8], [9] are 0 unless BLUES_NB_PME */
#define BLUES_N_ENERGY_TERMS 10
int blues_get_energy_terms(BluesEngine *h, double terms[BLUES_N_ENERGY_TERMS]);

/* integrator.step(n)  (reference blues/simulation.py:1082) */
int blues_step(BluesEngine *h, int32_t n_steps);
/* the same n calls of step(1) executed device-resident without returning to
 * the host between steps; optional per-step protocol-work trace (kJ/mol). */
int blues_run_switch(BluesEngine *h, int32_t n_steps, double *work_trace /* [n_steps] or NULL */);

/* integrator.getGlobalVariableByName / setGlobalVariableByName
 * (reference blues/simulation.py:935, blues/moves.py:1082, blues/reporters.py:415-418).
 * names: lambda step lambda_step n_lambda_steps nsteps protocol_work shadow_work
 *        perturbed_pe unperturbed_pe first_step nprop prop prop_lambda_min
 *        prop_lambda_max Eold Enew heat kT lambda_sterics lambda_electrostatics
 *        and, with a switching_mode: total_work Epert initial_energy final_energy
 *        accept naccept ntrials psteps (reference blues/switching.py:1062-1080; kJ/mol) */
int blues_get_global(BluesEngine *h, const char *name, double *value);
int blues_set_global(BluesEngine *h, const char *name, double value);
/* AlchemicalExternalLangevinIntegrator.reset (reference blues/integrators.py:240-249) */
int blues_reset(BluesEngine *h);

/* engine counters for bench.py / profiling: [0] force passes [1] neighbour
 * rebuilds [2] kernel launches [3] n i-tiles [4] n clusters [5] j-list capacity
 * [6] partial slabs [7] segment length*1000 + waves per block [8] longest
 * j-list at the last rebuild [9] re-sorts of the tile layout [10] list
 * builds executed [11] potential-energy evaluations launched by this engine
 * alone (those served by blues_batch_prefetch_energies are not counted)
 * [12] nonbonded kernel in use: 0 tile kernel, 1 sub-tile kernel, 2 per-atom
 * lists over an LDS image, 3 fragment lists [13] i-tiles per j-list [14] entries of all per-atom
 * lists at the last rebuild (fragment lists: of the outer lists) [15] 64-entry wave iterations they take
 * [16] pruned per-atom lists re-derived so far (one count per atom and prune; fragment lists: prunes of the chain)
 * [17] entries of all pruned lists now (fragment lists: of the inner lists) [18] 64-entry wave iterations they take
 * [19] 1 if the nonbonded kernel walks pruned (inner) lists [20] alchemical kernel in
 * use for the env pairs: 0 lane = (atom, list entry), 1 dense pair lists in fp32 pair arithmetic, 2 in fp64
 * [21] threads per block of the step kernel (BluesTuning.pack_clusters) */
#define BLUES_N_STATS 22
int blues_get_stats(BluesEngine *h, int64_t stats[BLUES_N_STATS]);
/* time `reps` launches of the dominant nonbonded kernel alone with HIP events
 * on the engine's own stream; returns mean microseconds per launch. */
int blues_time_nonbonded(BluesEngine *h, int32_t reps, double *usec_per_launch);
/* the same for a forced neighbour-list rebuild */
int blues_time_list_build(BluesEngine *h, int32_t reps, double *usec_per_launch);
/* Diagnostic: audits the per-atom neighbour lists at the current positions.  out[0] = pairs within the cutoff the nonbonded
 * kernel is responsible for (neither alchemical nor excluded), out[1] = how many of them are entries of no list (must be
 * 0: the list margins and rebuild / prune triggers exist to guarantee it).  Per-atom-list mode only (mixed precision,
 * mostly frozen system in a large batch); an error elsewhere. */
int blues_audit_lists(BluesEngine *h, int64_t out[2]);

/* ---- Device-resident State -------------------------------------------------
 * BLUES passes whole openmm.State objects around: getStateFromContext takes
 * positions and velocities three times per iteration (simulation.py:883-911,
 * 1035, 1056, 1096) only to hand them back unchanged through
 * setContextFromState (simulation.py:938-963, 1037, 1150-1163).  A snapshot
 * is that State kept in HBM: capture = device-to-device copy, restore =
 * setPositions / setVelocities from the copy (same bookkeeping as the host
 * calls), read = download for a consumer that really wants the numbers (a
 * Move, a reporter).  `what`: bit 0 positions, bit 1 velocities.  A snapshot
 * can be restored into any engine of the same size on the same device (the
 * MD -> NCMC hand-over).  Release every snapshot before destroying the engine
 * it was captured from. */
typedef struct BluesSnapshot BluesSnapshot;
int blues_snapshot_capture(BluesEngine *h, int32_t what, BluesSnapshot **out);
int blues_snapshot_release(BluesSnapshot *s);
int blues_snapshot_read(BluesSnapshot *s, int32_t what, double *out_xyz, int32_t n_atoms);
int blues_set_positions_from_snapshot(BluesEngine *h, const BluesSnapshot *s);
int blues_set_velocities_from_snapshot(BluesEngine *h, const BluesSnapshot *s);
/* A Move reads a few atoms of a State and writes them back (positions[idx],
 * positions[i] = ..., reference blues/moves.py:292-307): the few atoms travel,
 * the rest is restored device to device.  _edited returns 3 without doing
 * anything when an edited atom is constrained to an unedited one (take the
 * plain blues_set_positions route then). */
int blues_snapshot_read_atoms(BluesSnapshot *s, int32_t what, const int32_t *idx, int32_t n_idx, double *out_xyz);
int blues_set_positions_from_snapshot_edited(BluesEngine *h, const BluesSnapshot *s, const int32_t *idx, int32_t n_idx, const double *xyz_nm);

/* ---- Replica batches ------------------------------------------------------
 * BLUES chains are independent (reference examples run one
 * BLUESSimulation per process), and one replica of the 23k-atom system keeps
 * only a few percent of an MI355X busy.  A batch makes `count` congruent
 * engines on ONE device (same topology and protocol; own coordinates, box,
 * velocities and Philox stream) share every launch of integrator.step(n):
 * gridDim.y = count.  Each engine keeps its full C-ABI above (moves and
 * state queries stay per replica, as in the reference where they go through
 * each Simulation's own context); only stepping goes through the batch.
 * Results are bitwise identical to stepping each engine alone.
 *
 * blues_batch_step = integrator.step(n_steps) on every member (reference
 * blues/simulation.py:1082, once per replica).  status[r] != 0 means member r
 * raised (what would be an OpenMMException out of its step(); message via
 * blues_last_error(engines[r])); the other members are unaffected, matching
 * the reference's per-simulation error policy (simulation.py:1088-1094).
 * work_trace: [count][n_steps] per-step protocol work in kJ/mol, or NULL.
 * The return value is non-zero only for failures of the batch itself
 * (blues_batch_last_error).  Destroying a member dissolves the batch. */
typedef struct BluesBatch BluesBatch;
int blues_batch_create(BluesEngine *const *engines, int32_t count, BluesBatch **out);
int blues_batch_destroy(BluesBatch *b);
const char *blues_batch_last_error(const BluesBatch *b);
int blues_batch_size(const BluesBatch *b);
int blues_batch_step(BluesBatch *b, int32_t n_steps, double *work_trace, int32_t *status);
/* mask[r] == 0: member r sits out the following blues_batch_step calls (its
 * state is left untouched) -- a chain whose switch was abandoned after an
 * exception must not be advanced further (simulation.py:1088-1094).
 * NULL = everyone. */
int blues_batch_set_active(BluesBatch *b, const int32_t *mask);
/* Evaluates the total potential (what bit 0) and / or kinetic (bit 1) energy of
 * every active member -- bit 2: also the potential energy with lambda_sterics =
 * lambda_electrostatics = 1 at the same coordinates (the "alch"/MD energy of
 * _computeAlchemicalCorrection, simulation.py:1100-1119) -- with shared launches and one read-back, and leaves the
 * values in the members' energy caches: the blues_get_energy /
 * blues_set_positions calls that follow for each member (state.getPotentialEnergy
 * at simulation.py:908, the perturbed / unperturbed energies of
 * integrators.py:184-205) then cost no launch.  Purely an accelerator: results
 * are those of the per-member calls.  (Every potential-energy evaluation, batched or
 * not, also leaves the energy at lambda = (1, 1) in the cache -- only the alchemical terms
 * differ and the kernel has a spare lambda slot -- so bit 2 normally finds its values there.
 * With bit 1 the members' accumulators (protocol work) come back in the same read and serve
 * the blues_get_global("protocol_work") calls that follow, until the member is stepped.) */
int blues_batch_prefetch_energies(BluesBatch *b, int32_t what);
/* ---- The plugin boundary for a whole batch ---------------------------------
 * What BLUES does to each Context around the stepping -- getStateFromContext /
 * setContextFromState (reference blues/simulation.py:883-963, called at :1035, :1056,
 * :1096, :1150-1163), positions[atom_indices] / setPositions of a Move (blues/moves.py:
 * 292-307), integrator.reset() and setVelocitiesToTemperature (simulation.py:1184-1187)
 * -- for every member of a batch at once: the per-member entry points' host part in a
 * loop, ONE launch per kernel.  Semantics are those of the member calls named with each
 * function; out / snaps arrays have one slot per member, mask[r] == 0 / snaps[r] == NULL
 * leaves member r alone (mask NULL: everyone).
 * blues_batch_restore* return 3, having done nothing, when some member cannot take the
 * device route (no layout yet; an edited atom constrained to an unedited one): use the
 * member calls then. */
int blues_batch_snapshot_capture(BluesBatch *b, int32_t what, const int32_t *mask, BluesSnapshot **out);        /* blues_snapshot_capture */
int blues_batch_restore(BluesBatch *b, BluesSnapshot *const *snaps, int32_t what);                              /* blues_set_positions_from_snapshot (bit 0), blues_set_velocities_from_snapshot (bit 1) */
int blues_batch_restore_edited(BluesBatch *b, BluesSnapshot *const *snaps, const int32_t *idx, int32_t n_idx,
                               const double *xyz_nm /* [count][n_idx][3] */);                                    /* blues_set_positions_from_snapshot_edited */
int blues_batch_read_atoms(BluesBatch *b, BluesSnapshot *const *snaps /* NULL: the live state */, int32_t what,
                           const int32_t *idx, int32_t n_idx, double *out /* [count][n_idx][3] */);              /* blues_snapshot_read_atoms */
int blues_batch_reset(BluesBatch *b, const int32_t *mask);                                                       /* blues_reset */
int blues_batch_set_velocities_to_temperature(BluesBatch *b, double temperature, const uint64_t *seeds,
                                              const int32_t *mask);                                              /* blues_set_velocities_to_temperature */
int blues_batch_mesh_energy(BluesBatch *b, int32_t with_alchemical_charges, const int32_t *mask, double *out /* [count] */);   /* blues_mesh_energy */

/* [0] steps issued in lock step (one launch for all members) [1] steps that
 * fell back to per-member launches [2] members [3] batched energy evaluations */
int blues_batch_get_stats(BluesBatch *b, int64_t stats[4]);
/* What the layout of a batch has cost so far (ABI 6; bench.py reports the per-iteration differences, so that a slow iteration names its
 * cause): [0] re-plans of the layout shape (a member's group lists outgrew it) [1] seconds spent in them [2] members laid out again from a new
 * sort inside batch calls (creation included) [3] members re-sorted at the 64-step polls (their builders asked, or by age) [4] seconds spent
 * there [5] members moved to another shape in place (no sort) [6] re-sorts of all members, whoever asked [7] tiles per group list now
 * [8] list capacity [9] nonbonded kernel (blues_get_stats [12]) -- of the members in the batch's layout [10] members that step on launches
 * of their own now because their lists have outgrown the batch's shape (stragglers) [11] times a member became one [12] times one came back
 * [13] seconds spent laying those members out (leaving and coming back) [14] phases of a step in which a few members, out of step with the
 * rest (laid out again inside the call), took their own launches while the others stayed in lock step. */
#define BLUES_N_BATCH_COUNTERS 15
int blues_batch_get_counters(BluesBatch *b, double out[BLUES_N_BATCH_COUNTERS]);
/* as blues_time_nonbonded, for one batched launch covering all members.  With pruned per-atom lists an atom is served in
 * one of two ways: from its current pruned list, or from its full list while the pruned one is re-derived.  usec[0] / usec[1]:
 * the launch with every atom served the first / the second way; prune_fraction: the share of (atom, pass) pairs served the
 * second way in the force passes this batch has run so far; blues_batch_time_nonbonded returns the mean weighted with it. */
int blues_batch_time_nonbonded_modes(BluesBatch *b, int32_t reps, double usec[2], double *prune_fraction);
int blues_batch_time_nonbonded(BluesBatch *b, int32_t reps, double *usec_per_launch);
/* The same kernel timed WHERE IT RUNS: `every` > 0 brackets every `every`-th lock-step force launch of blues_batch_step with two
 * HIP events on the batch's stream (0 switches it off; either call resets the statistics).  blues_batch_get_kernel_timing drains
 * the pending events: out[0] = mean duration in us, out[1] = launches sampled, out[2] = the longest one.  This is the figure
 * bench.py reports as roofline.usec_per_launch (the reference has no counterpart: OpenMM times nothing per kernel). */
/* Diagnostic: where the set-up time of this process went so far (seconds): [0] blues_engine_create, [1] laying out tiles and lists
 * (sort_and_tile), [2] of which image + uploads, [3] device allocations, [4] zero-fills, [5] host-to-device copies, [6] allocations made, [7] stream and event creation. */
int blues_debug_setup_seconds(double *out8);
/* Diagnostic: the engines' device buffers of up to 8 MiB are carved from zero-filled slabs, 256 bytes that nobody owns behind each.  out8[0] blocks
 * checked, out8[1] blocks whose guard carries a mark (a kernel wrote past the end of a buffer), then for the first three of them: size of the
 * block in bytes (rounded to 256, guard included), byte offset of the first marked word within the guard.  Synchronises the device. */
int blues_debug_check_guards(int64_t *out8);
int blues_batch_kernel_timing(BluesBatch *b, int32_t every);
int blues_batch_get_kernel_timing(BluesBatch *b, double out[3]);

#ifdef __cplusplus
}
#endif
#endif /* BLUES_ENGINE_H */
