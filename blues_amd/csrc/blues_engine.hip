// blues_engine.hip -- host side of the MI355X-native NCMC engine and its C-ABI
// (include/blues_engine.h).  gfx950 only; no torch types; one engine per replica per GPU.
//
// The host mirrors the *deterministic* control state of
// AlchemicalExternalLangevinIntegrator (reference blues/integrators.py:159-231): step,
// lambda_step, prop, first_step.  It walks the splitting string, batches the substeps
// between two force evaluations into one k_integrate launch (a "program"), and schedules a
// force pass only when positions have changed -- one pass per NCMC step for "H V R O R V H"
// instead of the reference's three full energy/force evaluations.
// Accumulators (protocol work, forces, flags) stay on the device; nothing is read back
// between steps.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/blues_engine.h"
#include "device_common.h"
#include "kernels_alch.h"
#include "kernels_bonded.h"
#include <chrono>
#include <mutex>
#include <thread>
#include <functional>
#include <atomic>
#include "kernels_integrate.h"
#include "kernels_nb.h"
#include "kernels_pme.h"
#include "kernels_batch.h"

static thread_local std::string g_create_error;
// set-up cost accounting (diagnostic, blues_debug_setup_seconds): [0] blues_engine_create, [1] sort_and_tile, [2] its device uploads,
// [3] device allocations (hipMalloc), [4] zero-fills, [5] host-to-device copies of DBuf::upload, [6] number of allocations, [7] streams and events
static double g_setup_sec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
static std::mutex& setup_mu() { static std::mutex m; return m; }   // (engines are built from several host threads: bench.py --groups)
struct SetupTimer { double* acc; std::chrono::steady_clock::time_point t0; explicit SetupTimer(int k) : acc(&g_setup_sec[k]), t0(std::chrono::steady_clock::now()) {} ~SetupTimer() { const double d = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); std::lock_guard<std::mutex> lk(setup_mu()); *acc += d; } };

// Launch-policy overrides (include/blues_engine.h, BluesTuning): process-wide defaults copied into every engine / batch at its
// creation.  The library reads no environment variables.
static BluesTuning g_tuning = [] { BluesTuning t; blues_tuning_default(&t); return t; }();

// workgroups of the kernels of a batch that run from the rebuild's work list (kernels_batch.h: k_gather_stale_b): two per CU
static const int REBUILD_GRID = 512;
static const int SHAPE_CAND[9] = {1, 2, 3, 4, 5, 6, 8, 12, 16};   // tiles per group list the per-atom-list layout chooses from
static inline int shape_index(int S) { for (int q = 0; q < 9; q++) if (SHAPE_CAND[q] == S) return q; return -1; }

// ---- device memory of the engines' buffers (round 6).  An engine holds ~100 device buffers; 2048 chains were 134,000 hipMalloc calls at
// set-up -- 40 % of its thread-time, behind the runtime's lock whatever the number of host threads -- and as many zero-fills.  Buffers of up
// to SLAB_DIRECT bytes now come from slabs: 128 MiB allocations zero-filled once, handed out by a bump pointer under a mutex (a fresh
// block needs no fill); released blocks wait in a free list by size for the next buffer of that size (the engines of a System ask for the
// same sizes).  Slabs live as long as the process.  Every block carries a guard of 256 bytes behind it.
#define SLAB_BYTES ((size_t)128 << 20)
#define SLAB_DIRECT ((size_t)8 << 20)
struct DevSlabs {
    struct Slab { char* base; size_t size, used; };
    std::mutex mu;
    std::map<int, std::vector<Slab>> slabs;                              // by device
    std::map<std::pair<int, size_t>, std::vector<void*>> free_blocks;    // (device, bytes) -> released blocks
    static size_t round(size_t bytes) { return ((std::max<size_t>(bytes, 1) + 255) & ~(size_t)255) + 256; }
    // returns a block of round(bytes) bytes; *zeroed: it has never been handed out (still holds the slab's zeros)
    std::vector<std::pair<char*, size_t>> blocks;   // every block ever handed out (base, rounded bytes): blues_debug_check_guards
    void* get(int dev, size_t bytes, bool* zeroed) {
        const size_t rb = round(bytes);
        std::lock_guard<std::mutex> lk(mu);
        auto f = free_blocks.find({dev, rb});
        if (f != free_blocks.end() && !f->second.empty()) { void* q = f->second.back(); f->second.pop_back(); *zeroed = false; return q; }
        std::vector<Slab>& v = slabs[dev];
        if (v.empty() || v.back().used + rb > v.back().size) {
            Slab sl; sl.size = std::max(SLAB_BYTES, rb); sl.used = 0; sl.base = nullptr;
            if (hipMalloc((void**)&sl.base, sl.size) != hipSuccess) return nullptr;
            if (hipMemset(sl.base, 0, sl.size) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) { hipFree(sl.base); return nullptr; }   // (done before any stream can touch a block)
            v.push_back(sl);
        }
        void* q = v.back().base + v.back().used; v.back().used += rb; *zeroed = true;
        blocks.push_back({(char*)q, rb});
        return q;
    }
    void put(int dev, void* q, size_t bytes) {
        std::lock_guard<std::mutex> lk(mu);
        free_blocks[{dev, round(bytes)}].push_back(q);
    }
};
static DevSlabs& dev_slabs() { static DevSlabs* s = new DevSlabs; return *s; }   // (never destroyed: buffers of static engines may outlive any destructor order)

template <typename T> struct DBuf {
    T* p = nullptr; size_t n = 0, cap = 0;
    bool direct = false; int dev = 0;   // direct: its own hipMalloc (large buffers); the device the block lives on
    // an existing buffer that is large enough is reused (a re-sort keeps every address).  Round 6: large ENOUGH, not of the same size -- a
    // member that leaves its batch's layout for fragment lists and comes back changes the size of a dozen buffers twice, and with ~10^5
    // allocations in the process every hipFree / hipMalloc pair took milliseconds (one member's return: 0.2 s with the device idle)
    // returns true if the block is NEW and still holds the slab's zeros (alloc then needs no fill)
    bool reserve(size_t count) {
        if (p && count <= cap) { n = count; return false; }
        release(); n = count; cap = count;
        SetupTimer tm(3); { std::lock_guard<std::mutex> lk(setup_mu()); g_setup_sec[6] += 1.0; }
        const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
        if (bytes > ((size_t)1 << 31)) fprintf(stderr, "[blues] a single device allocation of %.1f GiB (%zu elements of %zu bytes)\n", bytes / 1073741824.0, count, sizeof(T));
        if (hipGetDevice(&dev) != hipSuccess) { n = cap = 0; throw std::string("hipGetDevice failed"); }
        if (bytes > SLAB_DIRECT) {
            direct = true;
            if (hipMalloc((void**)&p, bytes) != hipSuccess) { p = nullptr; n = cap = 0; throw std::string("hipMalloc failed"); }
            return false;
        }
        direct = false;
        bool zeroed = false;
        p = (T*)dev_slabs().get(dev, bytes, &zeroed);
        if (!p) { n = cap = 0; throw std::string("hipMalloc failed"); }
        return zeroed;
    }
    void alloc(size_t count) {   // zero-filled
        if (reserve(count)) return;   // (a block nobody has written to)
        SetupTimer tm(4);
        hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T));
    }
    void upload(const std::vector<T>& h) {   // (every element is overwritten: no fill first)
        if (h.size() != n || !p) reserve(h.size());
        SetupTimer tm(5);
        if (!h.empty() && hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) throw std::string("hipMemcpy H2D failed");
    }
    void download(std::vector<T>& h) const {
        h.resize(n);
        if (n && hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) throw std::string("hipMemcpy D2H failed");
    }
    void release() {
        if (p) { if (direct) hipFree(p); else dev_slabs().put(dev, p, std::max<size_t>(cap, 1) * sizeof(T)); }
        p = nullptr; n = 0; cap = 0; direct = false;
    }
    ~DBuf() { release(); }
};

struct HostCluster { int atoms[4]; int type, nc; double dist[3]; };

// ---- 3-D Hilbert index (Skilling's transpose algorithm), 10 bits per axis
static uint32_t hilbert3(uint32_t x, uint32_t y, uint32_t z, int bits) {
    uint32_t X[3] = {x, y, z};
    const uint32_t M = 1u << (bits - 1);
    for (uint32_t Q = M; Q > 1; Q >>= 1) {
        const uint32_t P = Q - 1;
        for (int i = 0; i < 3; i++) {
            if (X[i] & Q) X[0] ^= P;
            else { const uint32_t t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    for (int i = 1; i < 3; i++) X[i] ^= X[i - 1];
    uint32_t t = 0;
    for (uint32_t Q = M; Q > 1; Q >>= 1) if (X[2] & Q) t ^= Q - 1;
    for (int i = 0; i < 3; i++) X[i] ^= t;
    uint32_t key = 0;
    for (int b = bits - 1; b >= 0; b--) for (int i = 0; i < 3; i++) key = (key << 1) | ((X[i] >> b) & 1u);
    return key;
}

struct BluesBatch;
struct BluesEngine;

// Total potential energy of the current positions for up to two settings of (lambda_sterics, lambda_electrostatics): BLUES
// asks for the energy at the protocol's current parameters AND at lambda = 1 for the same coordinates
// (simulation.py:1100-1119; the default lambda_electrostatics function ends at 0.9999999999999998, not 1.0).
struct ECache {
    double e[2] = {0, 0}, ls[2] = {0, 0}, le[2] = {0, 0}; bool ok[2] = {false, false}; int last = 0;
    void clear() { ok[0] = ok[1] = false; }
    bool any() const { return ok[0] || ok[1]; }
    bool find(double s_, double e_, double* out) { for (int k = 0; k < 2; k++) if (ok[k] && ls[k] == s_ && le[k] == e_) { if (out) *out = e[k]; last = k; return true; } return false; }
    void put(double s_, double e_, double val) {
        int k = -1;
        for (int q = 0; q < 2; q++) if (ok[q] && ls[q] == s_ && le[q] == e_) k = q;
        if (k < 0) k = !ok[0] ? 0 : (!ok[1] ? 1 : 1 - last);
        e[k] = val; ls[k] = s_; le[k] = e_; ok[k] = true; last = k;
    }
};

// openmm.State as BLUES uses it (simulation.py:883-911, 938-963): a copy of positions and velocities taken at one
// moment, handed back later to setPositions / setVelocities.  Kept in HBM; the host sees it only if it asks.
struct BluesSnapshot {
    BluesEngine* owner = nullptr;
    int n = 0; bool has_x = false, has_v = false;
    double* x[3] = {nullptr, nullptr, nullptr}; double* v[3] = {nullptr, nullptr, nullptr};
    double* block = nullptr;   // one allocation: x[0..2], v[0..2]
    ECache ecache;   // potential energies that were known for the captured positions ...
    uint64_t box_epoch = 0;   // ... in the box the owner had then (a barostat move or set_box in between makes them stale)
};

// What the engines made from ONE System share on the host (round 6): per-atom parameters, exclusion lists, the bond / constraint graph.
// 2048 chains of the benchmark are 2048 engines of one System: each used to keep its own copy (~3 MB of the 5.6 MB of host memory per
// chain; the exclusion lists are 23,400 small vectors).  blues_engine_create finds the System's table by a hash of its arrays.
struct HostTopology {
    std::vector<double> mass, charge, sigma, eps;
    std::vector<double> qn;                      // NonbondedForce charges: the alchemical atoms' are 0
    std::vector<std::vector<int>> excl;          // per atom, both directions
    std::vector<int> alch_local;                 // [n] -> local alchemical index or -1
    std::vector<int32_t> link_pairs;             // bonds and constraints: what holds a molecule together
};

struct BluesEngine {
    std::string err;
    std::shared_ptr<HostTopology> T;   // (shared with the other engines of the same System)
    int device = 0;
    // replica batching (see BluesBatch below): while the batch is in lock step the leader's launches cover every
    // replica (gridDim.y) and the other members only advance their mirrored control state
    BluesBatch* batch = nullptr; int batch_index = -1; hipStream_t own_stream = nullptr;
    uint64_t box_epoch = 1;   // bumped by every change of the box: cached energies of a State are only good in the box they were computed in
    int batch_R = 1;  // replicas sharing this engine's launches (shapes the launch decomposition in sort_and_tile)
    uint64_t args_epoch = 1;  // bumped whenever a device buffer referenced by the argument records is (re)allocated
    hipStream_t stream = nullptr, s1 = nullptr, s2 = nullptr, cur = nullptr; bool stream_pooled = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evFork = nullptr, evJ1 = nullptr, evJ2 = nullptr, evA = nullptr, evB = nullptr;
    hipGraphExec_t gexec = nullptr; int graph_units = 16; bool graph_valid = false, use_graph = false, graph_fork = false;
    DevCtrl* ctrl_arg = nullptr; DevCtrl host_ctrl;
    int64_t st_graph_steps = 0;
    // ---- system (host copies, caller order)
    int n = 0;
    double box[3] = {0, 0, 0};
    std::vector<int> alch;               // caller indices
    int nb_method = 1; double cutoff = 1, alpha = 0, sc_alpha = 0.5;
    int annih_elec = 1, annih_ster = 0, remove_cm = 0, check_env_excl = 0;
    BluesTuning tune;   // the process-wide tuning at the time this engine was created
    double skin = 0.12; bool skin_fixed = false;
    double skin_m = 0.12, trig = 0.06;   // margin of MOBILE list candidates, and the displacement that asks for a rebuild (derive_margins)
    // pruned per-atom lists (nonbonded_atom_body): inner margins for frozen / mobile candidates, displacement that asks for a prune
    bool prune_on = false; double prune_m = 0.0, ptrig = 0.0;
    int steps_since_sort = 0;   // whole steps taken in the current order (fragment lists re-sort by age: resort_by_age)
    double frag_grow_o = 1.0, frag_grow_i = 1.0;   // room given to the fragment lists beyond the density estimate (frag_grow_caps: a row reached 90 % of its capacity)
    EwaldPoly ewpoly;   // degree-9 fit of the smooth part of the Ewald pair force (fit_ewald_poly), mixed precision
    // ---- integrator
    double dt = 0, temperature = 0, gamma = 1, kT = 0, tol = 1e-8;
    int nsteps = 0, nprop = 1, n_lambda = 0, precision = 0;
    double prop_min = 0, prop_max = 0;
    std::string split; int n_R = 0, n_V = 0, n_O = 0, n_H = 0, n_L = 0;
    // switching integrators of reference blues/switching.py (BLUES_SWITCH_VV / BLUES_SWITCH_GHMC): `split` is synthesised
    // ("H" + "AV" x psteps, or "HOAVO"), `split_first` is the propagation of the first-step block; the energy bookkeeping
    // around each velocity-Verlet step (shadow work / Metropolis test) is done on the host with synchronous energies
    int switch_mode = 0, psteps = 1; std::string split_first;
    double sw_Epert = 0.0, sw_shadow = 0.0, sw_Einit = 0.0, sw_Efinal = 0.0, sw_bracket_E0 = 0.0; int sw_accept = 0, sw_naccept = 0, sw_ntrials = 0;
    unsigned sw_draw = 0;   // Metropolis uniforms drawn so far by the GHMC integrator: the Philox counter; reset() does NOT touch it (ntrials is a statistic)
    bool sw_bracket_open = false; struct BluesSnapshot* sw_saved = nullptr;
    std::vector<double> tab_ls, tab_le;
    uint64_t seed = 0; int replica = 0;
    // ---- mirrored control state
    int h_step = 0, h_lambda_step = 0, h_prop = 1, h_first_step = 0;
    unsigned h_draw = 0;
    double cur_ls = 1.0, cur_le = 1.0, h_lambda = 0.0;
    double h_perturbed = 0, h_unperturbed = 0; bool unpert_valid = false, x_edited = false;
    double work_pending = 0.0; bool work_pending_set = false;   // a work increment the batch is about to book for this member (add_work)
    bool pass_valid = false; int pass_L = 0, pass_fmask = 7;
    // the sums of the last force pass are still to be formed (launch_finalize_deferred): by the step kernel itself if the next
    // program is the steady-state one, by k_finalize otherwise
    bool fin_pending = false; double fin_le[3] = {0, 0, 0}; int fin_mask = 7;
    bool have_positions = false, sorted_ok = false, lists_forced = true, relayout_failed = false;
    bool solo_now = false;    // out of step with the rest of its batch for the phase that is running: it takes it on launches of its own (batch_do_steps: phase)
    bool straggler = false;   // member of a batch whose own layout no longer fits the batch's shape: it steps on launches of its own until it fits again (batch_do_steps)
    std::vector<double> hx;        // host copy of the positions the tiles were last laid out from (caller order, [n][3])
    double e_frozen[2] = {0, 0}; bool e_frozen_valid = false;
    ECache ecache;  // total potential energy at the current positions
    double ke_cache = 0; bool ke_cache_valid = false;
    // setPositions' 16-byte verdict (frozen atom changed / how far the i-atoms are from their sort positions) is read back lazily:
    // the copy into pinned memory is queued with the load and looked at by the next evaluation (resolve_xfer)
    unsigned* h_xfer = nullptr; bool xfer_pending = false; hipStream_t xfer_stream = nullptr;
    const unsigned* xfer_src = nullptr;   // where the pending verdict will be found: h_xfer, or this member's slot of a batch-wide read-back
    bool xfer_foreign = false;            // the positions just loaded are another engine's State (the MD leg's hand-over): resolve_xfer
    hipEvent_t ev_edit = nullptr;                             // recorded after the last copy OUT of the pinned staging area
    unsigned char* h_edit = nullptr; size_t h_edit_cap = 0;   // pinned staging of a Move's edited atoms and of read-backs of a few atoms
    std::vector<int> edit_idx_host;                           // what d_edit_idx holds (a Move asks for the same atoms every time)
    DevAccum acc_cache; bool acc_cache_valid = false; int64_t acc_cache_stamp = 0;   // accumulators as read back by a batch prefetch; valid until the next launch
    // ---- derived topology
    std::vector<int> mobile;       // caller indices with mass > 0
    std::vector<HostCluster> clusters;
    int n_itiles = 0, n_tiles = 0, jcap = 0, n_islots = 0, pool_cap = 0, PA = 1, k2_nblocks_env = 0, k2_jiter = 1;
    bool k2_dense = false;   // the alchemical kernel's env pairs in their dense form (kernels_alch.h: alchemical_dense_body)
    bool k2_f32 = false;     // ... in fp32 pair arithmetic with the work from per-pair differences (alchemical_dense32_body; round 6)
    int seg_len = 64, waves_tile = 4, wpb = 4, npart = 1;  // K1 decomposition
    bool fuse_forces = false, fast_step = true, fuse_big = false;  // fuse_big: measured slower (the alchemical role's 140 VGPRs and 36 KB LDS cap the occupancy of the nonbonded role)
    int k1_iw = 64;  // i-atoms per wave in the nonbonded kernel: 64 = classic tile kernel, 8/16 = sub-tile throughput kernel
    int k1_mode = 0;  // 0: tile kernel (lane = i-atom), 1: sub-tile kernel, 2: per-atom Verlet lists + LDS tile image (nonbonded_atom_body), 3: fragment lists (kernels_frag.h)
    // fragment lists (every environment atom mobile): the static cut of the environment into fragments of <= 3 atoms
    // (build_fragments), the layout of the current sort, the lists
    std::vector<std::array<int, 3>> frag_atoms; std::vector<int> frag_cnt, frag_of_atom, frag_pos_of_atom; bool frag_ok = false, frag_built = false;   // (bonds and constraints: what holds a molecule together)
    std::vector<int> sp_start_h; std::vector<int2> sp_ent_h;
    int frag_F = 0, frag_NI = 0, frag_nblk = 0, frag_ocap = 0, frag_icap = 0, frag_fpw = 1, frag_nwg = 0; double frag_m = 0.0; DBuf<int> d_ifrag; bool frag_rel = false;
    DBuf<FragRec> d_fimg; DBuf<uint4> d_fpos0; DBuf<float2> d_ljtab; int frag_ntypes = 0; DBuf<int> d_sp_start, d_ocount, d_icount; DBuf<int2> d_sp_ent; DBuf<FragBox> d_fbb; DBuf<unsigned> d_olist, d_ilist, d_xprune_s;
    int acap = 0;     // capacity of one atom's list (mode 2)
    // layout shape of the per-atom-list mode.  A lone engine derives (S, jcap) from its own geometry at every re-sort; members
    // of a batch must stay congruent, so the batch fixes the shape for all of them (shape_S > 0) and re-plans it for
    // everybody when some member's list outgrows it (shape_overflow)
    int hint_count = 0;   // list length that raises resort_hint
    int shape_S = 0, shape_jcap = 0; bool shape_overflow = false, forbid_atom = false; double shape_need = 0.0;
    // what every candidate shape would need with the atoms where the last sort found them (sort_and_tile; SHAPE_CAND): the expected
    // length of the longest group list, the radius of the widest group.  A batch plans its shape from its members' tables and moves
    // everybody to it WITHOUT a new sort (reshape_groups): the sorted order, the image and the tiles do not depend on S
    double need_by_S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, rad_by_S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; bool need_table_ok = false;
    int lists_alloc = 0;   // group lists the per-atom-list buffers have room for (mode 2: one per i-tile, the finest shape)
    int layout_R = 0;   // the batch size (batch_R) the current layout was derived for (sort_and_tile)
    // a member of a batch whose layout is sized from the density (tile / sub-tile kernels): the capacities the batch agreed on, kept while
    // they hold this member's lists -- a MonteCarloBarostat leaves every member in its own box, and a capacity re-derived from n / V
    // would differ between members by a step of 64 after a 1 % change of volume (batch_congruent)
    int pin_jcap = 0, pin_seg = 0, pin_nw = 0, pin_wpb = 0;
    int S = 1, n_lists = 0;   // S consecutive i-tiles share one j-list (mode 2; 1 otherwise); n_lists = ceil(n_itiles / S)
    DBuf<unsigned short> d_alist, d_plist; DBuf<int> d_acount, d_pcount, d_pneed; DBuf<unsigned> d_xprune[3];   // pruned lists (by i-slot)
    DBuf<uint4> d_pimg4; DBuf<float2> d_pimg2; DBuf<int> d_mlist, d_mcount; int mcap = 0;   // packed group images (ListArgs)
    DBuf<unsigned short> d_aself; DBuf<uint4> d_pimgb; DBuf<int> d_sx_row;
    int n_entries = 0;
    int int_blocks = 1, int_threads = 128;
    bool clusters_packed = false;   // BluesTuning.pack_clusters took effect (build_clusters): a 128-thread block may run the fused step kernel
    double total_mass = 0;
    // ---- device buffers
    DBuf<double> d_x[3], d_v[3], d_xbuild[3], d_mass, d_charge, d_sigma, d_eps;
    DBuf<double> d_x_sort[3], d_stage;   // positions at the last spatial sort; [n][3] staging for host transfers
    DBuf<unsigned> d_xfer_out;
    DBuf<int> d_edit_idx; DBuf<double> d_edit_xyz;   // a Move's edit of a few atoms (blues_set_positions_from_snapshot_edited)
    std::vector<struct BluesSnapshot*> snap_pool;   // released snapshots, reused (hipMalloc per getState would dominate)
    DBuf<AtomF> d_img_f; DBuf<AtomD> d_img_d;
    DBuf<int> d_sorted_of_orig, d_orig_of_sorted, d_tile_atoms, d_jlist, d_jstage, d_jcount, d_batch_slot, d_ex_start, d_ex_idx, d_islot;
    DBuf<unsigned long long> d_mask_pool;
    DBuf<DevFlags> d_flags; DBuf<DevAccum> d_acc; DBuf<DevCtrl> d_ctrl; DBuf<double> d_tab_ls, d_tab_le; DBuf<long long> d_stamps;
    DBuf<double> d_fpart, d_epart_nb, d_fJ, d_self_part, d_e_part, d_fent, d_ftot, d_alch_self, d_epart_b, d_cm_part, d_trace, d_scratch;
    DBuf<int> d_orig_of_islot, d_row_of_orig, d_mobile_atoms, d_mobile_index, d_exc_owner, d_exc_is_env;
    DBuf<double> d_noise; unsigned noise_draw_base = 0; int n_noise = 0; bool noise_valid = false;
    DBuf<int> d_alch_orig, d_alch_local, d_exc_start, d_exc_partner; DBuf<double> d_exc_params;
    DBuf<int> d_cl_atoms, d_cl_type, d_cl_nc, d_cl_alch, d_cl_mobile, d_cl_sorted; DBuf<double> d_cl_dist;
    DBuf<ClusterRec> d_recs; std::vector<ClusterRec> h_recs;
    DBuf<AlchJRec> d_jrec; DBuf<AlchARec> d_arec;
    DBuf<double> d_mom_part; bool vel_clean = false, pass_valid_for_l = true;  // velocities unchanged since the last force pass (momentum partials valid)
    DBuf<FinRec> d_finrecs; std::vector<int> h_row_of_orig, h_row_start;
    // reciprocal space (BLUES_NB_PME, kernels_pme.h)
    template <typename T> struct PmeBufs { DBuf<T> a_re, a_im, b_re, b_im, eterm, tw_cos[3], tw_sin[3], qf_re, qf_im, phi_f; };
    bool pme = false; int pme_K[3] = {0, 0, 0}, pme_order = 5, disp_corr = 0;
    std::vector<int> pme_frozen;   // NonbondedForce charges (alchemical atoms 0); frozen atoms that carry charge
    DBuf<double> d_qn, d_qn_full, d_frec, d_pme_e; DBuf<int> d_pme_frozen; DBuf<unsigned long long> d_pme_acc;
    PmeBufs<float> pme_f; PmeBufs<double> pme_d;
    bool pme_static_valid = false;
    double e_ewald_const = 0.0, e_disp = 0.0;   // self term + neutralising background; dispersion correction (functions of the box)
    // bonded
    DBuf<int> d_row_atom, d_row_start, d_ent_type, d_ent_term, d_ent_role;
    DBuf<int> d_term_atoms[T_NTYPES]; DBuf<double> d_term_params[T_NTYPES];
    int n_terms[T_NTYPES] = {0, 0, 0, 0, 0, 0}; int n_rows = 0; double restr_k = 0;
    // pending integrate program
    Program prog; unsigned prog_draw_base = 0; int prog_trace = -1; bool tracing = false;
    // stats
    int64_t st_passes = 0, st_launches = 0, st_resorts = 0, st_energy_evals = 0, st_reshapes = 0;
    std::vector<int> h_sorted_of_orig, h_orig_of_sorted;

    ~BluesEngine() {
        for (BluesSnapshot* sn : snap_pool) { if (sn->block) hipFree(sn->block); delete sn; }
        if (gexec) hipGraphExecDestroy(gexec);
        for (hipEvent_t e : {ev0, ev1, evFork, evJ1, evJ2, evA, evB}) if (e) hipEventDestroy(e);
        if (stream && !stream_pooled) hipStreamDestroy(stream);
        for (hipStream_t q : {s1, s2}) if (q) hipStreamDestroy(q);
    }
};

#define E_FAIL(h, ...) do { char _b[512]; snprintf(_b, sizeof _b, __VA_ARGS__); (h)->err = _b; return 1; } while (0)
#define HIP_OK(h, call) do { hipError_t _e = (call); if (_e != hipSuccess) { E_FAIL(h, "%s: %s", #call, hipGetErrorString(_e)); } } while (0)

// ------------------------------------------------------------------ replica batch
// R congruent engines (same topology and protocol; own coordinates, velocities, Philox stream) that share one HIP
// stream.  In lock step every launch of the stepping path is issued once, by the leader, with gridDim.y = R; the
// per-replica argument records live in HBM (kernels_batch.h).  A step in which the members' mirrored control states
// differ (e.g. one replica's move edited positions and another's did not) falls back to per-replica launches.
struct BluesBatch {
    std::string err;
    std::vector<BluesEngine*> eng;
    std::vector<char> failed;           // per call: members whose host-side stepping raised an error
    std::vector<char> active, rec_active;  // caller's mask (blues_batch_set_active) / what the device records currently say
    std::vector<uint64_t> seen_epoch;
    std::vector<unsigned> rec_delta; BluesEngine* rec_leader = nullptr;   // what the device records currently say
    BluesEngine* leader = nullptr;
    bool lockstep = false;
    DBuf<RepNb<float>> d_nb_f; DBuf<RepNb<double>> d_nb_d; DBuf<RepCore> d_core; DBuf<int> d_hints;
    // list synchronisation: in a large batch some member needs new lists at almost every step, and the launch lasts as long
    // as one rebuild whoever asks; when they all rebuild together (any request rebuilds all) most steps see no rebuild at all
    DBuf<int> d_req; bool sync_lists = false;
    DBuf<int> d_work;   // [1 + 2R] members that rebuild their lists in the current force pass (kernels_batch.h: k_gather_stale_b)
    bool replanning = false;   // inside batch_plan_shape on behalf of a member (ensure_sorted)
    DBuf<int> d_frozen_work; DBuf<double> d_frozen_sum;   // the members without a frozen-frozen energy constant, their constants (batch_prefetch)
    DBuf<int> d_work_frag;   // [1 + 2R] fragment lists: members that rebuild or prune in the current pass (k_gather_frag_b), and a flag per member
    BluesTuning tune;   // the process-wide tuning at the time the batch was created
    // argument arena of the batched boundary calls (blues_batch_capture ...): pinned host side, device side, one upload per call
    unsigned char* h_arena = nullptr; size_t arena_cap = 0; DBuf<unsigned char> d_arena;
    unsigned* h_xfer_all = nullptr; DBuf<unsigned> d_xfer_all;   // [R][4] verdicts of a batched setPositions
    // Members keep their own streams for everything that is per replica (moves, state exchange, energies) so that host
    // threads serving different chains overlap; stepping runs on the batch's stream.  enter: the batch stream waits for
    // each member's pending work and the members issue into it; leave: the batch stream is drained, members go home.
    hipStream_t stream = nullptr; bool entered = false;
    DBuf<double> d_gather; int64_t st_prefetch_pe = 0, st_prefetch_ke = 0;
    int64_t st_lockstep_steps = 0, st_fallback_steps = 0, st_replans = 0, st_relayouts = 0, st_poll_resorts = 0, st_stragglers = 0, st_rejoined = 0, st_partial_steps = 0; double st_replan_sec = 0.0, st_resort_sec = 0.0, st_straggle_sec = 0.0;   // (blues_batch_get_counters)
    // in-situ timing of the nonbonded force kernel (blues_batch_kernel_timing): every `k1t_every`-th lock-step force launch of the
    // stepping loop is bracketed by two events on the stream it runs on; finished pairs are harvested lazily (no synchronisation)
    struct EvPair { hipEvent_t a = nullptr, b = nullptr; bool busy = false; };
    std::vector<const BluesEngine*> congr_lead; std::vector<uint64_t> congr_le, congr_me;   // congruence already established (batch_do_steps)
    bool defer_work = false; std::vector<DevAccum*> h_wacc; std::vector<double> h_wdelta; DBuf<DevAccum*> d_wacc; DBuf<double> d_wdelta;   // add_work of all members in one launch
    std::vector<EvPair> k1t_pairs; int k1t_every = 0; int64_t k1t_seen = 0, k1t_n = 0; double k1t_sum_us = 0.0, k1t_max_us = 0.0;
    int R() const { return (int)eng.size(); }
};
static void k1t_harvest(BluesBatch* B, bool wait) {
    for (auto& p : B->k1t_pairs) if (p.busy) {
        if (wait) hipEventSynchronize(p.b);
        if (hipEventQuery(p.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { B->k1t_sum_us += 1000.0 * ms; B->k1t_max_us = std::max(B->k1t_max_us, 1000.0 * (double)ms); B->k1t_n++; }
        p.busy = false;
    }
    (void)hipGetLastError();   // (hipErrorNotReady of a query is not an error of ours)
}
// a free event pair for the launch that is about to be issued, or null (not sampled / all pairs still in flight)
static BluesBatch::EvPair* k1t_begin(BluesBatch* B, hipStream_t st) {
    if (!B || B->k1t_every <= 0) return nullptr;
    if ((B->k1t_seen++ % B->k1t_every) != 0) return nullptr;
    k1t_harvest(B, false);
    for (auto& p : B->k1t_pairs) if (!p.busy) { if (hipEventRecord(p.a, st) != hipSuccess) return nullptr; return &p; }
    return nullptr;
}
static void k1t_end(BluesBatch::EvPair* p, hipStream_t st) { if (p && hipEventRecord(p->b, st) == hipSuccess) p->busy = true; }
static int batch_enter(BluesBatch* B);
static void batch_leave(BluesBatch* B);
static inline bool batch_dry(const BluesEngine* h) { return h->batch && h->batch->lockstep && h->batch->leader != h; }
static inline bool batch_lead(const BluesEngine* h) { return h->batch && h->batch->lockstep && h->batch->leader == h; }
static inline int* batch_req_ptr(const BluesEngine* h) { return h->batch && h->batch->sync_lists ? h->batch->d_req.p : nullptr; }
template <typename R> static const RepNb<R>* batch_reps_nb(const BluesBatch* b) { if constexpr (sizeof(R) == 4) return b->d_nb_f.p; else return b->d_nb_d.p; }

static Box3 make_box(const BluesEngine* h) {
    Box3 b; for (int k = 0; k < 3; k++) { b.L[k] = h->box[k]; b.invL[k] = 1.0 / h->box[k]; } return b;
}
// List margins.  A pair's separation changes by at most the sum of its atoms' displacements since the build, and a rebuild
// is requested as soon as ONE mobile atom has moved `trig`: a frozen candidate therefore needs a margin of trig, a mobile one
// 2 trig.  Where nearly everything is frozen (freeze_radius, reference blues/simulation.py:394-480) the few mobile
// candidates get the double margin and trig = skin: the same lists last twice the displacement.  Otherwise every candidate
// has the margin `skin` and trig = skin / 2, the usual Verlet rule.
// Pruned lists (prune_on; kernels_nb.h: nonbonded_atom_body): see the rule inside derive_margins.
static void derive_margins(BluesEngine* h) {
    double room = 1e30;
    for (int k = 0; k < 3; k++) room = std::min(room, 0.5 * h->box[k] - h->cutoff - 1e-6);
    h->skin = std::min(h->skin, std::max(0.0, room));
    const bool mostly_frozen = 4 * h->mobile.size() <= (size_t)h->n && !h->tune.plain_skin;
    h->skin_m = mostly_frozen ? std::min(2.0 * h->skin, std::max(0.0, room)) : h->skin;
    h->trig = std::min(h->skin, 0.5 * h->skin_m);
    // the dual list pays where most candidates are frozen (one displacement per pair); elsewhere the full lists are walked.
    // Full lists are built for the whole chain at once; until the next rebuild every atom stays within `trig` of its build
    // position.  Frozen candidates within cutoff + skin, mobile candidates within cutoff + skin_m of the atom's build position.
    // Pruned list of atom i, formed at time t_i from its full list and valid while i stays within m of x_i(t_i) (its own
    // trigger, ptrig = m):
    //   frozen candidate j: kept if |x_i(t_i) - x_j| < cutoff + m.  The full list holds all of those while i has moved at most
    //     skin - m since the build: trig <= skin - m.
    //   mobile candidate j: kept if |x_i(t_i) - xb_j| < cutoff + trig + m, xb_j = where j was at the BUILD: j is within trig of
    //     xb_j until the next rebuild and i within m of x_i(t_i) until its next prune, so a pair that is left out stays beyond
    //     the cutoff.  The full list holds all of those if skin_m >= 2 trig + m.  j's CURRENT position cannot be used: j's own
    //     prunes reset j's reference, so nothing bounds how far j moves after t_i except the chain-level trigger (round 3: the
    //     first version kept mobile candidates within cutoff + 3m of their current position, and the list audit --
    //     blues_audit_lists -- found pairs inside the cutoff missing from the lists in 17 of 150 steps of a hot chain).
    const double m = h->tune.prune_margin < 0.0 ? 0.04 : h->tune.prune_margin;
    h->prune_on = mostly_frozen && m > 0.0 && m < 0.75 * h->skin;
    h->prune_m = h->ptrig = 0.0;
    if (h->prune_on) {
        h->prune_m = m; h->ptrig = m;
        h->trig = std::min(h->skin - m, 0.5 * (std::max(0.0, room) - m));
        h->skin_m = 2.0 * h->trig + m;
        if (h->trig <= 0.25 * h->skin) { h->prune_on = false; h->prune_m = h->ptrig = 0.0; h->trig = std::min(h->skin, 0.5 * h->skin_m); }
    }
}

// Degree-9 Chebyshev fit of T(u) = [erf(x) - 2/sqrt(pi) x exp(-x^2)] / r^3, x = alpha r, u = r^2 on [0, cutoff^2], in the
// variable w = 2u/cutoff^2 - 1, converted to monomials in w (device_common.h: pair_regular<float>)
static void fit_ewald_poly(double alpha, double rc, EwaldPoly* P) {
    const int N = 64, D = EWALD_POLY_DEG;
    const double two_sqrtpi = 2.0 / std::sqrt(M_PI);
    double f[N], ck[D + 1];
    for (int j = 0; j < N; j++) {
        const double xj = std::cos(M_PI * (j + 0.5) / N), u = 0.5 * (xj + 1.0) * rc * rc, r = std::sqrt(u), x = alpha * r, x2 = x * x;
        if (x < 0.1) f[j] = alpha * alpha * alpha * two_sqrtpi * (2.0 / 3.0 - 0.4 * x2 + x2 * x2 / 7.0 - x2 * x2 * x2 / 27.0 + x2 * x2 * x2 * x2 / 132.0);
        else f[j] = (std::erf(x) - two_sqrtpi * x * std::exp(-x2)) / (r * r * r);
    }
    for (int k = 0; k <= D; k++) {
        double sum = 0.0;
        for (int j = 0; j < N; j++) sum += f[j] * std::cos(M_PI * k * (j + 0.5) / N);
        ck[k] = (k == 0 ? 1.0 : 2.0) * sum / N;
    }
    // Chebyshev -> monomial: T_0 = 1, T_1 = w, T_{k+1} = 2 w T_k - T_{k-1}
    double mono[D + 1] = {0}, t0[D + 1] = {0}, t1[D + 1] = {0}, t2[D + 1];
    t0[0] = 1.0; t1[1] = 1.0;
    for (int q = 0; q <= D; q++) mono[q] += ck[0] * t0[q] + (D >= 1 ? ck[1] * t1[q] : 0.0);
    for (int k = 2; k <= D; k++) {
        for (int q = 0; q <= D; q++) t2[q] = (q > 0 ? 2.0 * t1[q - 1] : 0.0) - t0[q];
        for (int q = 0; q <= D; q++) { mono[q] += ck[k] * t2[q]; t0[q] = t1[q]; t1[q] = t2[q]; }
    }
    for (int q = 0; q <= D; q++) P->c[q] = (float)mono[q];
    P->wa = (float)(2.0 / (rc * rc));
    // what the kernels will evaluate (fp32 Horner) against T itself on a dense grid of r: the fit is only kept where it is
    // good enough in absolute terms (device_common.h: EWALD_POLY_MAX_RESIDUAL); tighter Ewald tolerances take the erfc/exp form
    double worst = 0.0;
    for (int j = 0; j < 2000; j++) {
        const double r = rc * (0.05 + 0.95 * (j + 0.5) / 2000.0), x = alpha * r;
        const float w = fmaf((float)(r * r), P->wa, -1.0f);
        float T = P->c[D];
        for (int k = D - 1; k >= 0; k--) T = fmaf(T, w, P->c[k]);
        worst = std::max(worst, std::fabs((double)T - (std::erf(x) - two_sqrtpi * x * std::exp(-x * x)) / (r * r * r)));
    }
    P->residual = worst;
    P->exact = worst > EWALD_POLY_MAX_RESIDUAL ? 1 : 0;
}

template <typename R> static NbConst<R> make_nbconst(const BluesEngine* h) {
    NbConst<R> c;
    const double two = sizeof(R) == 4 ? 4294967296.0 : 18446744073709551616.0;
    for (int k = 0; k < 3; k++) { c.dscale[k] = h->box[k] / two; c.scale[k] = (R)c.dscale[k]; }
    c.rc2 = (R)(h->cutoff * h->cutoff); c.alpha = (R)h->alpha;
    c.rlist2 = (h->cutoff + h->skin) * (h->cutoff + h->skin);
    c.rlist2_m = (h->cutoff + h->skin_m) * (h->cutoff + h->skin_m);
    c.rp2 = (float)((h->cutoff + h->prune_m) * (h->cutoff + h->prune_m)) * 1.00001f; c.rp2_m = (float)((h->cutoff + h->trig + h->prune_m) * (h->cutoff + h->trig + h->prune_m)) * 1.00001f;   // (from the candidate's BUILD position: derive_margins)
    c.ew = h->ewpoly;
    return c;
}

// ------------------------------------------------------------------ fragments (kernels_frag.h), once per engine
// The environment (every atom that is not alchemical) cut into fragments of up to three atoms: walking a molecule in breadth-first
// order over its bonds and constraints, an atom not yet in a fragment takes up to two of its own free neighbours -- a rigid water is one
// fragment, the atoms of a solute's fragment are a bond or two apart.  With them the static table of fragment pairs that hold an excluded atom pair (and of every
// fragment with itself): the 9-bit masks the list builder copies into such entries.
static void build_fragments(BluesEngine* h) {
    const int n = h->n;
    h->frag_built = true;
    std::vector<std::vector<int>> adj(n);
    auto link = [&](int a, int b) { if (a >= 0 && b >= 0 && a < n && b < n && a != b) { adj[a].push_back(b); adj[b].push_back(a); } };
    for (size_t e = 0; e + 1 < h->T->link_pairs.size(); e += 2) link(h->T->link_pairs[e], h->T->link_pairs[e + 1]);
    h->frag_atoms.clear(); h->frag_cnt.clear();
    h->frag_of_atom.assign(n, -1); h->frag_pos_of_atom.assign(n, 0);
    std::vector<char> seen(n, 0);
    std::vector<int> order;
    for (int i = 0; i < n; i++) {
        if (seen[i] || h->T->alch_local[i] >= 0) continue;
        order.clear(); order.push_back(i); seen[i] = 1;
        for (size_t q = 0; q < order.size(); q++) {
            std::vector<int>& nb = adj[order[q]];
            std::sort(nb.begin(), nb.end());
            for (int b : nb) if (!seen[b] && h->T->alch_local[b] < 0) { seen[b] = 1; order.push_back(b); }
        }
        // a fragment = an atom and up to two of its own neighbours (bonded or constrained to it): every pair of its atoms is at most two
        // bonds apart, whatever the molecule -- a fragment's reach stays ~0.25 nm and constant over time.  (Round 5 cut the breadth-first
        // order into consecutive triples: at a branch of a large molecule -- a side chain beside the backbone frontier, a ring closure --
        // consecutive entries are not bonded to each other, the reach grows to several bonds and the list margins, sized for water-like
        // fragments, no longer cover it: kernels_frag.h, FR_OVER_REACH.)  A rigid water is still one fragment: O, H, H.
        for (size_t q = 0; q < order.size(); q++) {
            const int a0 = order[q];
            if (h->frag_of_atom[a0] >= 0) continue;
            std::array<int, 3> fr = {a0, -1, -1};
            int c = 1;
            for (int b : adj[a0]) if (c < 3 && h->frag_of_atom[b] < 0 && h->T->alch_local[b] < 0 && b != fr[1]) fr[c++] = b;   // (adj is sorted: see the walk above)
            for (int k = 0; k < c; k++) { h->frag_of_atom[fr[k]] = (int)h->frag_atoms.size(); h->frag_pos_of_atom[fr[k]] = k; }
            h->frag_atoms.push_back(fr); h->frag_cnt.push_back(c);
        }
    }
    const int F = (int)h->frag_atoms.size();
    auto colmask = [](int cnt) { return cnt >= 3 ? 0x1ffu : (cnt == 2 ? 0x0DBu : 0x049u); };
    auto rowmask = [](int cnt) { return cnt >= 3 ? 0x1ffu : (cnt == 2 ? 0x03Fu : 0x007u); };
    std::vector<std::map<int, unsigned>> sp(F);
    for (int fa = 0; fa < F; fa++) {
        unsigned self = colmask(h->frag_cnt[fa]);
        for (int a = 0; a < 3; a++) self &= ~(1u << (4 * a));   // an atom with itself
        sp[fa][fa] = self;
    }
    for (int fa = 0; fa < F; fa++)
        for (int a = 0; a < h->frag_cnt[fa]; a++)
            for (int p : h->T->excl[h->frag_atoms[fa][a]]) {
                const int fb = h->frag_of_atom[p];
                if (fb < 0) continue;   // (an alchemical partner: the alchemical kernel's pair)
                auto it = sp[fa].find(fb);
                if (it == sp[fa].end()) it = sp[fa].emplace(fb, colmask(h->frag_cnt[fb])).first;
                it->second &= ~(1u << (3 * a + h->frag_pos_of_atom[p]));
            }
    h->sp_start_h.assign(F + 1, 0); h->sp_ent_h.clear();
    h->frag_ok = F > 0 && F < (1 << FR_MASK_SHIFT);
    {   // (the fragment kernel keeps the environment's Lennard-Jones types in a table of FR_TYPES_MAX entries; one more for absent atoms)
        std::map<std::pair<float, float>, int> types;
        for (int i = 0; i < n; i++) if (h->T->alch_local[i] < 0) types[{(float)(0.5 * h->T->sigma[i]), (float)(2.0 * std::sqrt(h->T->eps[i]))}] = 1;
        if ((int)types.size() + 1 > FR_TYPES_MAX) h->frag_ok = false;
    }
    for (int fa = 0; fa < F; fa++) {
        for (auto& kv : sp[fa]) {
            unsigned m = kv.second;
            if (!(m & rowmask(h->frag_cnt[fa]))) m = 0u;   // no live pair among the atoms this fragment has
            h->sp_ent_h.push_back(make_int2(kv.first, (int)m));
        }
        h->sp_start_h[fa + 1] = (int)h->sp_ent_h.size();
        if (h->sp_start_h[fa + 1] - h->sp_start_h[fa] > FR_SP_MAX) h->frag_ok = false;
    }
}

// ------------------------------------------------------------------ topology set-up (once)
static int build_clusters(BluesEngine* h, const BluesSystemDesc* s) {
    const int n = h->n;
    std::vector<std::vector<std::pair<int, double>>> adj(n);
    std::vector<std::pair<int, int>> cons; std::vector<double> cdist;
    for (int c = 0; c < s->n_constraints; c++) {
        const int i = s->constraint_atoms[2 * c], j = s->constraint_atoms[2 * c + 1];
        const bool mi = h->T->mass[i] == 0.0, mj = h->T->mass[j] == 0.0;
        if (mi && mj) continue;  // OpenMM ignores constraints between two massless particles
        if (mi != mj) E_FAIL(h, "A constraint cannot involve a massless particle (%d-%d)", i, j);
        cons.push_back({i, j}); cdist.push_back(s->constraint_dist[c]);
    }
    // connected components, ordered by smallest atom index, constraints in list order (as the oracle)
    std::vector<int> parent(n); std::iota(parent.begin(), parent.end(), 0);
    auto find = [&](int a) { while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; } return a; };
    for (auto& c : cons) { int a = find(c.first), b = find(c.second); if (a != b) parent[std::max(a, b)] = std::min(a, b); }
    std::map<int, std::vector<int>> by_root;
    for (size_t c = 0; c < cons.size(); c++) by_root[find(cons[c].first)].push_back((int)c);
    std::vector<char> in_cluster(n, 0);
    std::vector<HostCluster> alch_first, rest;
    auto push = [&](const HostCluster& hc) {
        bool is_alch = false;
        for (int a = 0; a < 4; a++) if (hc.atoms[a] >= 0 && h->T->alch_local[hc.atoms[a]] >= 0) is_alch = true;
        (is_alch ? alch_first : rest).push_back(hc);
    };
    for (auto& kv : by_root) {
        const std::vector<int>& cl = kv.second;
        HostCluster hc; for (int a = 0; a < 4; a++) hc.atoms[a] = -1; hc.dist[0] = hc.dist[1] = hc.dist[2] = 0; hc.nc = (int)cl.size();
        if (cl.size() > 3) E_FAIL(h, "constraint cluster with %zu constraints is not supported (HBonds / rigid water only)", cl.size());
        std::vector<int> atoms;
        for (int c : cl) for (int a : {cons[c].first, cons[c].second}) if (std::find(atoms.begin(), atoms.end(), a) == atoms.end()) atoms.push_back(a);
        if (atoms.size() > 4) E_FAIL(h, "constraint cluster with %zu atoms is not supported", atoms.size());
        if (cl.size() == 3 && atoms.size() == 3) {  // triangle: canonical order (0,1),(0,2),(1,2)
            int a = cons[cl[0]].first, b = cons[cl[0]].second, c2a = cons[cl[1]].first, c2b = cons[cl[1]].second;
            int shared, other1, third;
            if (c2a == a || c2b == a) { shared = a; other1 = b; third = (c2a == a) ? c2b : c2a; }
            else if (c2a == b || c2b == b) { shared = b; other1 = a; third = (c2a == b) ? c2b : c2a; }
            else E_FAIL(h, "malformed triangle constraint cluster");
            hc.type = 2; hc.atoms[0] = shared; hc.atoms[1] = other1; hc.atoms[2] = third;
            hc.dist[0] = cdist[cl[0]]; hc.dist[1] = cdist[cl[1]]; hc.dist[2] = cdist[cl[2]];
        } else {  // star: common centre, leaves in list order
            int centre = -1;
            if (cl.size() == 1) centre = cons[cl[0]].first;
            else {
                for (int cand : {cons[cl[0]].first, cons[cl[0]].second}) {
                    bool all = true;
                    for (int c : cl) if (cons[c].first != cand && cons[c].second != cand) all = false;
                    if (all) centre = cand;
                }
                if (centre < 0) E_FAIL(h, "constraint cluster is neither a star nor a triangle");
            }
            hc.type = 1; hc.atoms[0] = centre;
            for (size_t q = 0; q < cl.size(); q++) {
                hc.atoms[q + 1] = cons[cl[q]].first == centre ? cons[cl[q]].second : cons[cl[q]].first;
                hc.dist[q] = cdist[cl[q]];
            }
        }
        for (int a = 0; a < 4; a++) if (hc.atoms[a] >= 0) in_cluster[hc.atoms[a]] = 1;
        push(hc);
    }
    for (int i : h->mobile) if (!in_cluster[i]) {
        HostCluster hc; hc.atoms[0] = i; hc.atoms[1] = hc.atoms[2] = hc.atoms[3] = -1; hc.type = 0; hc.nc = 0; hc.dist[0] = hc.dist[1] = hc.dist[2] = 0;
        push(hc);
    }
    // a wave executes the star and the triangle solver one after the other if it holds both kinds, so group
    // the clusters by kind and start each kind on a wave boundary (padding slots hold empty clusters)
    {
        std::vector<HostCluster> all = alch_first; all.insert(all.end(), rest.begin(), rest.end());
        HostCluster empty; for (int a = 0; a < 4; a++) empty.atoms[a] = -1; empty.type = 0; empty.nc = 0; empty.dist[0] = empty.dist[1] = empty.dist[2] = 0;
        h->clusters.clear();
        for (int kind : {1, 0, 2}) {
            bool any = false;
            for (const HostCluster& c : all) if (c.type == kind) { h->clusters.push_back(c); any = true; }
            if (any) while (h->clusters.size() % 64) h->clusters.push_back(empty);
        }
        while (!h->clusters.empty() && h->clusters.back().atoms[0] < 0) h->clusters.pop_back();
        // BluesTuning.pack_clusters: in a batch of more than 512 chains the step kernel (two 256-register waves per SIMD: 2048 waves
        // on the device) runs in rounds, and what counts is the NUMBER of waves per chain, not the length of one wave's chain: the
        // kinds back to back, largest first -- the flagship's 87 triangles, 6 stars and 1 single atom are 2 waves instead of 4, the
        // second one runs both solvers.  Same arithmetic per cluster; the sums over the block (centre of mass) take another order.
        const int pack = h->tune.pack_clusters;
        if ((pack == 2 || (pack == 0 && h->tune.assume_batch > 512)) && all.size() <= 256 && (all.size() + 63) / 64 < (h->clusters.size() + 63) / 64) {   // (one block per chain: the fused step kernel's case)
            int cnt[3] = {0, 0, 0};
            for (const HostCluster& c : all) cnt[c.type]++;
            int order[3] = {0, 1, 2};
            std::stable_sort(order, order + 3, [&](int a, int b) { return cnt[a] > cnt[b]; });
            h->clusters.clear();
            for (int q = 0; q < 3; q++) for (const HostCluster& c : all) if (c.type == order[q]) h->clusters.push_back(c);
            h->clusters_packed = true;
        }
    }
    const int ncl = (int)h->clusters.size();
    std::vector<int> ca(ncl * 4), ct(ncl), cn(ncl); std::vector<double> cd(ncl * 3);
    for (int c = 0; c < ncl; c++) {
        for (int a = 0; a < 4; a++) ca[c * 4 + a] = h->clusters[c].atoms[a];
        ct[c] = h->clusters[c].type; cn[c] = h->clusters[c].nc;
        for (int q = 0; q < 3; q++) cd[c * 3 + q] = h->clusters[c].dist[q];
    }
    h->d_cl_atoms.upload(ca); h->d_cl_type.upload(ct); h->d_cl_nc.upload(cn); h->d_cl_dist.upload(cd);
    {
        std::vector<int> cal(ncl * 4, -1), cmo(ncl * 4, 0), mi(n, 0);
        for (size_t m = 0; m < h->mobile.size(); m++) mi[h->mobile[m]] = (int)m;
        for (int c = 0; c < ncl * 4; c++) if (ca[c] >= 0) { cal[c] = h->T->alch_local[ca[c]]; cmo[c] = mi[ca[c]]; }
        h->d_cl_alch.upload(cal); h->d_cl_mobile.upload(cmo); h->d_cl_sorted.alloc((size_t)ncl * 4);
        h->h_recs.assign(ncl, ClusterRec());
        for (int c = 0; c < ncl; c++) {
            ClusterRec& r = h->h_recs[c]; memset(&r, 0, sizeof r);
            r.type = ct[c]; r.nc = cn[c]; r.na = 0;
            for (int a = 0; a < 4; a++) {
                r.atoms[a] = ca[c * 4 + a]; r.alch[a] = cal[c * 4 + a]; r.mobile[a] = cmo[c * 4 + a]; r.sorted[a] = 0; r.islot[a] = -1;
                if (r.atoms[a] >= 0) { r.na = a + 1; r.w[a] = 1.0 / h->T->mass[r.atoms[a]]; }
            }
            for (int q = 0; q < 3; q++) r.dist[q] = cd[c * 3 + q];
        }
        h->d_recs.upload(h->h_recs);
    }
    h->int_threads = ncl <= 256 ? std::max(128, ((ncl + 63) / 64) * 64) : 256;
    h->int_blocks = std::max(1, (ncl + h->int_threads - 1) / h->int_threads);
    h->d_cm_part.alloc((size_t)h->int_blocks * 3);
    return 0;
}

static int build_bonded(BluesEngine* h, const BluesSystemDesc* s) {
    // all terms are kept for energies; rows (force gather) only for mobile atoms
    std::vector<int> ta[T_NTYPES]; std::vector<double> tp[T_NTYPES];
    auto is_alch = [&](int i) { return h->T->alch_local[i] >= 0; };
    for (int b = 0; b < s->n_bonds; b++) { ta[T_BOND].push_back(s->bond_atoms[2 * b]); ta[T_BOND].push_back(s->bond_atoms[2 * b + 1]); tp[T_BOND].push_back(s->bond_params[2 * b]); tp[T_BOND].push_back(s->bond_params[2 * b + 1]); }
    for (int a = 0; a < s->n_angles; a++) { for (int q = 0; q < 3; q++) ta[T_ANGLE].push_back(s->angle_atoms[3 * a + q]); tp[T_ANGLE].push_back(s->angle_params[2 * a]); tp[T_ANGLE].push_back(s->angle_params[2 * a + 1]); }
    for (int t = 0; t < s->n_torsions; t++) { for (int q = 0; q < 4; q++) ta[T_TORSION].push_back(s->torsion_atoms[4 * t + q]); for (int q = 0; q < 3; q++) tp[T_TORSION].push_back(s->torsion_params[3 * t + q]); }
    // exceptions: env-env go to the bonded kernel, those touching alchemical atoms to the alchemical kernel
    std::vector<std::vector<int>> arow_partner(h->alch.size()); std::vector<std::vector<double>> arow_par(h->alch.size());
    for (int e = 0; e < s->n_exceptions; e++) {
        const int i = s->exception_atoms[2 * e], j = s->exception_atoms[2 * e + 1];
        const double* p = s->exception_params + 3 * e;
        if (!is_alch(i) && !is_alch(j)) { ta[T_EXC].push_back(i); ta[T_EXC].push_back(j); for (int q = 0; q < 3; q++) tp[T_EXC].push_back(p[q]); }
        else if (is_alch(i) && is_alch(j)) {
            arow_partner[h->T->alch_local[i]].push_back(j); for (int q = 0; q < 3; q++) arow_par[h->T->alch_local[i]].push_back(p[q]);
            arow_partner[h->T->alch_local[j]].push_back(i); for (int q = 0; q < 3; q++) arow_par[h->T->alch_local[j]].push_back(p[q]);
        } else if (p[0] != 0.0 || p[2] != 0.0) {  // alchemical x environment: one row entry on the alchemical side
            const int a = is_alch(i) ? i : j, e2 = is_alch(i) ? j : i;
            arow_partner[h->T->alch_local[a]].push_back(e2); for (int q = 0; q < 3; q++) arow_par[h->T->alch_local[a]].push_back(p[q]);
        }
    }
    if (h->pme) for (int e = 0; e < s->n_exclusions; e++) {   // every excluded pair: its reciprocal-space interaction is removed again
        const int i = s->exclusions[2 * e], j = s->exclusions[2 * e + 1];
        const double qq = h->T->qn[i] * h->T->qn[j];
        if (qq != 0.0) { ta[T_EWEX].push_back(i); ta[T_EWEX].push_back(j); tp[T_EWEX].push_back(qq); }
    }
    for (int r = 0; r < s->n_restraints; r++) { ta[T_RESTR].push_back(s->restraint_atoms[r]); for (int q = 0; q < 3; q++) tp[T_RESTR].push_back(s->restraint_x0[3 * r + q]); }
    h->restr_k = s->restraint_k;
    const int width[T_NTYPES] = {2, 3, 4, 2, 1, 2};
    std::vector<std::vector<int>> rtype(h->n), rterm(h->n), rrole(h->n);
    for (int ty = 0; ty < T_NTYPES; ty++) {
        h->n_terms[ty] = (int)ta[ty].size() / width[ty];
        for (int t = 0; t < h->n_terms[ty]; t++) for (int q = 0; q < width[ty]; q++) {
            const int i = ta[ty][t * width[ty] + q];
            if (i < 0 || i >= h->n) E_FAIL(h, "bonded term references atom %d", i);
            if (h->T->mass[i] != 0.0) { rtype[i].push_back(ty); rterm[i].push_back(t); rrole[i].push_back(q); }
        }
        h->d_term_atoms[ty].upload(ta[ty]); h->d_term_params[ty].upload(tp[ty]);
    }
    std::vector<int> row_atom, row_start(1, 0), et, ei, er;
    for (int i = 0; i < h->n; i++) if (!rtype[i].empty()) {
        row_atom.push_back(i);
        et.insert(et.end(), rtype[i].begin(), rtype[i].end()); ei.insert(ei.end(), rterm[i].begin(), rterm[i].end()); er.insert(er.end(), rrole[i].begin(), rrole[i].end());
        row_start.push_back((int)et.size());
    }
    h->n_rows = (int)row_atom.size(); h->n_entries = (int)et.size();
    { std::vector<int> row_of(h->n, -1); for (int r = 0; r < h->n_rows; r++) row_of[row_atom[r]] = r; h->d_row_of_orig.upload(row_of); h->h_row_of_orig = row_of; h->h_row_start = row_start; }
    h->d_fent.alloc((size_t)3 * std::max(1, h->n_entries));
    h->d_row_atom.upload(row_atom); h->d_row_start.upload(row_start); h->d_ent_type.upload(et); h->d_ent_term.upload(ei); h->d_ent_role.upload(er);
    int total_terms = 0; for (int ty = 0; ty < T_NTYPES; ty++) total_terms += h->n_terms[ty];
    h->d_epart_b.alloc((size_t)((total_terms + 255) / 256 + 1) * T_NTYPES);
    // alchemical exception rows
    std::vector<int> es(1, 0), ep, eo; std::vector<double> epar;
    for (size_t a = 0; a < h->alch.size(); a++) { eo.insert(eo.end(), arow_partner[a].size(), h->alch[a]); ep.insert(ep.end(), arow_partner[a].begin(), arow_partner[a].end()); epar.insert(epar.end(), arow_par[a].begin(), arow_par[a].end()); es.push_back((int)ep.size()); }
    { std::vector<int> ie(ep.size()); for (size_t q = 0; q < ep.size(); q++) ie[q] = h->T->alch_local[ep[q]] < 0; h->d_exc_is_env.upload(ie); }
    h->d_exc_start.upload(es); h->d_exc_partner.upload(ep); h->d_exc_owner.upload(eo); h->d_exc_params.upload(epar);
    return 0;
}

// ------------------------------------------------------------------ spatial sort + tile image (host, at set_positions)
static int download_xyz(BluesEngine* h, double* xyz, DBuf<double>* src);
static int sort_and_tile(BluesEngine* h) {
    SetupTimer tm_all(1);
    const int n = h->n;
    if (h->hx.size() != (size_t)3 * n) {   // (the positions this layout is made from: normally filled by the caller; the copy is dropped when the layout is done)
        if (!h->have_positions) E_FAIL(h, "positions have not been set");
        h->hx.resize((size_t)3 * n);
        if (download_xyz(h, h->hx.data(), h->d_x)) return 1;
    }
    h->steps_since_sort = 0;
    h->shape_overflow = false;   // (set again below if this layout still does not fit the batch's shape; a layout that left the per-atom lists has no shape to outgrow)
    // fragment lists (kernels_frag.h) where every environment atom moves: mixed precision, more than the lone-chain handful of tiles
    // fragment lists (kernels_frag.h), mixed precision: where every environment atom moves (more than the lone-chain handful of
    // tiles); where the per-atom lists over group images cannot hold the system (forbid_atom: the mobile atoms of an NCMC System
    // have scattered over MD legs, reference blues/simulation.py:1028-1037 hands such a State over every iteration); on request
    bool want_frag = false;
    {
        size_t mobile_env = 0; for (int o : h->mobile) mobile_env += h->T->alch_local[o] < 0;
        const int nit = ((int)mobile_env + 63) / 64;
        const bool all_mobile = mobile_env + h->alch.size() == (size_t)n;
        const bool asked = h->precision == 0 && mobile_env > 0 && h->tune.k1_mode != 1 && h->tune.k1_mode != 2
                           && (h->tune.k1_mode == 3 || h->forbid_atom || (all_mobile && nit * h->batch_R > 32 && h->tune.fuse_forces != 1));
        if (asked && !h->frag_built) build_fragments(h);   // (a few ms of host work per engine: only where a layout wants them)
        want_frag = h->frag_ok && h->precision == 0 && mobile_env > 0 && h->tune.k1_mode != 1 && h->tune.k1_mode != 2
                    && (h->tune.k1_mode == 3 || h->forbid_atom || (all_mobile && nit * h->batch_R > 32 && h->tune.fuse_forces != 1));   // (fuse_forces = 1 pins the one-launch force pass)
    }
    std::vector<std::pair<uint64_t, int>> keys(n);
    auto hkey = [&](int i) {
        uint32_t c[3];
        for (int k = 0; k < 3; k++) { double fr = h->hx[3 * i + k] / h->box[k]; fr -= std::floor(fr); c[k] = std::min<uint32_t>(1023u, (uint32_t)(fr * 1024.0)); }
        return (uint64_t)hilbert3(c[0], c[1], c[2], 10);
    };
    for (int i = 0; i < n; i++) {
        // (fragment lists: a fragment's atoms stay together, in fragment order, at the place of its first atom)
        const int fr = want_frag ? h->frag_of_atom[i] : -1;
        if (fr >= 0) keys[i] = {(hkey(h->frag_atoms[fr][0]) << 32) | (uint64_t)((uint32_t)fr * 4u + (uint32_t)h->frag_pos_of_atom[i]), i};
        else keys[i] = {(hkey(i) << 32) | 0xffffffffull, i};
    }
    std::stable_sort(keys.begin(), keys.end());
    h->h_orig_of_sorted.resize(n); h->h_sorted_of_orig.resize(n);
    for (int s = 0; s < n; s++) { h->h_orig_of_sorted[s] = keys[s].second; h->h_sorted_of_orig[keys[s].second] = s; }
    // i-tiles: mobile, non-alchemical atoms in sorted order
    std::vector<int> tile_atoms, islot(n, -1);
    for (int s = 0; s < n; s++) {
        const int o = h->h_orig_of_sorted[s];
        if (h->T->mass[o] != 0.0 && h->T->alch_local[o] < 0) { islot[o] = (int)tile_atoms.size(); tile_atoms.push_back(s); }
    }
    h->n_itiles = ((int)tile_atoms.size() + 63) / 64;
    tile_atoms.resize((size_t)h->n_itiles * 64, -1);
    if (!h->alch.empty()) { for (size_t a = 0; a < 64; a++) tile_atoms.push_back(a < h->alch.size() ? h->h_sorted_of_orig[h->alch[a]] : -1); }
    if (tile_atoms.empty()) { tile_atoms.assign(64, -1); }
    h->n_islots = std::max(1, h->n_itiles) * 64;
    // Verlet skin: a small i-set is latency-bound (longer j-lists cost nothing, rebuilds do); a large one is
    // throughput-bound (every extra j costs pair evaluations)
    if (!h->skin_fixed) {
        // batch_R: a large batch is throughput-bound like a large i-set.  With pruned lists the nonbonded kernel no longer pays for
        // the outer margin, and the rebuild trigger is the margin minus the inner one: the margin grows by it (same rebuild rate)
        const bool mostly_frozen = 4 * h->mobile.size() <= (size_t)h->n && !h->tune.plain_skin;
        const double m = h->tune.prune_margin < 0.0 ? 0.04 : h->tune.prune_margin;
        // (measured at R = 512, us per step: margin 0.12: 661, 0.14: 587, 0.16: 574, 0.20: 554, 0.24: 550, 0.28: 557 -- the rebuild
        // kernels cost in proportion to the members that rebuild, the pruned lists make the nonbonded kernel indifferent)
        h->skin = h->n_itiles * h->batch_R <= 32 ? 0.3 : (mostly_frozen && m > 0.0 ? 0.16 + m : 0.12);
        // fragment lists: the outer list is cheap to keep long (4 bytes per nine pairs, walked only by the prune), a rebuild is not
        // (measured at R = 16, us per chain-step: skin 0.3: 37.6, 0.4: 36.5, 0.5: 36.4 with 77 / 53 / 37 rebuilds per 1000 steps)
        if (want_frag) h->skin = 0.40;   // (also where most atoms are frozen: the lists are a few hundred rows)
    }
    derive_margins(h);
    if (want_frag) {   // plain Verlet margins (a frozen candidate gets half of them in the builder itself)
        h->prune_on = false; h->prune_m = h->ptrig = 0.0; h->skin_m = h->skin; h->trig = 0.5 * h->skin;
    }
    if (want_frag) {   // inner margin of the dual fragment lists (kernels_frag.h); <= 0: the force kernel walks the outer lists
        // (bare stepping at R = 16, us per chain-step: 0.07: 37.9, 0.10: 36.5, 0.13: 35.8, 0.16: 36.4; configs[3] through the driver, ns/day: 0.11: 9,584,
        // 0.12: 9,712, 0.13: 9,784, 0.15: 9,897 -- a list of 220-256 fragments is four chunks either way, a wider margin prunes less often)
        const double m = h->tune.prune_margin < 0.0 ? 0.15 : h->tune.prune_margin;
        h->frag_m = m > 0.0 ? std::min(m, 0.6 * h->skin) : h->skin;
    }
    // capacities
    const double rl = h->cutoff + h->skin;
    const double vol = h->box[0] * h->box[1] * h->box[2], rho = n / vol;
    const double a = std::cbrt(64.0 / std::max(rho, 1e-9)) * 1.35;  // tile edge incl. slack for diffusion
    double est = rho * (a * a * a + 6 * a * a * rl + 3 * M_PI * a * rl * rl + 4.0 / 3.0 * M_PI * rl * rl * rl) * 1.8;
    if (h->tune.jcap_scale > 0.0) est *= h->tune.jcap_scale;   // tests: shrink the capacity to exercise the re-sort path
    int jcap = (int)std::min<double>(n, est);
    jcap = std::max(64, ((jcap + 63) / 64) * 64);
    jcap = std::min(jcap, 16384);
    if (((n + 63) / 64) * 64 <= jcap) jcap = ((n + 63) / 64) * 64;
    h->jcap = jcap;
    {   // K1 decomposition: segments of seg_len j-atoms, waves_tile waves per i-tile in blocks of wpb waves
        const int nit = std::max(1, h->n_itiles);
        const double est_count = std::min<double>(jcap, est / 1.8);
        int CH = 64;
        while (CH > 8 && nit * (est_count / CH) < 4096.0) CH >>= 1;
        if (h->tune.seg_len > 0) CH = std::max(4, std::min(64, h->tune.seg_len));
        int NW = std::max(1, std::min(jcap / CH, (8192 + nit - 1) / nit));
        int WPB = h->precision == 0 ? 16 : 8;
        while (WPB > 1 && (WPB > NW || nit * NW / WPB < 64)) WPB >>= 1;
        if (h->tune.waves_per_block > 0) WPB = std::max(1, std::min(h->precision == 0 ? 16 : 8, h->tune.waves_per_block));
        // small i-set: every force kernel is latency-bound -> one fused launch.  In a replica batch the launch holds
        // batch_R times the work: from a few replicas on it is throughput that counts, and the separate kernels (each
        // with its own register / LDS budget, sub-tile nonbonded variant) win -- measured 259 vs 197 us at R = 64.
        h->fuse_forces = nit * h->batch_R <= 32;
        if (h->tune.fuse_forces >= 0) h->fuse_forces = h->tune.fuse_forces != 0;
        if (h->fuse_forces) { WPB = 4; NW = std::max(4, NW / 3); }  // ~3 segments per wave: as long as the alchemical role
        NW = std::max(WPB, (NW / WPB) * WPB);
        if (h->pin_jcap > 0 && h->batch && std::min<double>(n, est / 1.8 * 1.2) + 64 <= h->pin_jcap) { jcap = h->pin_jcap; h->jcap = jcap; CH = h->pin_seg; NW = h->pin_nw; WPB = h->pin_wpb; }
        h->seg_len = CH; h->waves_tile = NW; h->wpb = WPB; h->npart = NW / WPB;
        h->k1_iw = 64;
        if (!h->fuse_forces && h->precision == 0) {
            h->k1_iw = 8;
            if (h->tune.sub_iw > 0) h->k1_iw = h->tune.sub_iw;
            if (h->k1_iw != 8 && h->k1_iw != 16 && h->k1_iw != 32) h->k1_iw = 64;
        }
        if (h->k1_iw != 64) {
            int NC = 8;
            if (h->tune.sub_chunks > 0) NC = h->tune.sub_chunks;
            h->waves_tile = NC; h->npart = NC; h->wpb = 4; h->seg_len = 64;
        }
        // throughput regime in mixed precision: per-atom Verlet lists over an LDS-resident image of the j-list
        // (nonbonded_atom_body).  S consecutive i-tiles share one list / one image; S and the list capacity come from the
        // geometry: the list holds what lies within cutoff+skin of the group's bounding sphere.
        h->k1_mode = h->k1_iw != 64 ? 1 : 0;
        h->S = 1; h->acap = 0; h->need_table_ok = false;
        bool want_atom = h->k1_mode == 1;
        if (h->tune.k1_mode == 1) want_atom = false;
        if (h->forbid_atom) want_atom = false;
        for (int o : h->mobile) if ((int)h->T->excl[o].size() + 1 > SX_MAX) want_atom = false;   // (the builder of the atoms' lists keeps an atom's excluded partners in one fixed row: kernels_nb.h SX_ROW)
        if (want_atom && h->n_itiles > 0) {
            double est_rad = 0.0;   // radius of the widest group of the last estimate
            auto group_est = [&](int S) {   // largest expected list length over the groups of S tiles
                double worst = 0.0; est_rad = 0.0;
                for (int g0 = 0; g0 < h->n_itiles; g0 += S) {
                    double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30};
                    std::vector<std::array<double, 3>> pts;
                    const int first = h->h_orig_of_sorted[tile_atoms[(size_t)g0 * 64]];
                    for (int q = g0 * 64; q < std::min(h->n_itiles, g0 + S) * 64; q++) {
                        if (tile_atoms[q] < 0) continue;
                        const int o = h->h_orig_of_sorted[tile_atoms[q]];
                        std::array<double, 3> d;
                        for (int k = 0; k < 3; k++) { d[k] = h->hx[3 * o + k] - h->hx[3 * first + k]; d[k] -= h->box[k] * std::nearbyint(d[k] / h->box[k]); lo[k] = std::min(lo[k], d[k]); hi[k] = std::max(hi[k], d[k]); }
                        pts.push_back(d);
                    }
                    double r2 = 0.0;
                    for (auto& d : pts) { double q2 = 0.0; for (int k = 0; k < 3; k++) { const double e = d[k] - 0.5 * (lo[k] + hi[k]); q2 += e * e; } r2 = std::max(r2, q2); }
                    const double rs = std::sqrt(r2) + rl;
                    est_rad = std::max(est_rad, std::sqrt(r2));
                    const double box_v = (hi[0] - lo[0] + 2 * rl) * (hi[1] - lo[1] + 2 * rl) * (hi[2] - lo[2] + 2 * rl);
                    worst = std::max(worst, rho * std::min(4.0 / 3.0 * M_PI * rs * rs * rs, box_v));
                }
                return worst;
            };
            double slack = 1.5;    // room for the i-atoms to spread before the next re-sort (a re-sort is asked for at 1.25x)
            if (h->tune.jcap_scale > 0.0) slack *= h->tune.jcap_scale;
            const int lds_max = 6400;     // list entries whose image (24 B each) fits the 160 KB of LDS beside the kernel's statics
            double best_cost = 1e300; int best_S = 0, best_cap = 0;
            h->shape_overflow = false;
            for (int q = 0; q < 9; q++) {
                const int S = SHAPE_CAND[q];
                if (S > 1 && S > h->n_itiles) { h->need_by_S[q] = 1e30; h->rad_by_S[q] = 1e30; continue; }
                h->need_by_S[q] = group_est(S); h->rad_by_S[q] = est_rad;
            }
            h->need_table_ok = true;
            if (h->shape_S > 0) {   // member of a batch: the batch's shape, as long as this member's lists fit it
                best_S = std::min(h->shape_S, std::max(1, h->n_itiles)); best_cap = h->shape_jcap;
                h->shape_need = group_est(best_S);
                if (h->shape_need * 1.1 + 64 > best_cap) h->shape_overflow = true;   // a re-sort (asked for at 90 % of the capacity) that does not make room
                if (h->tune.debug_lists) fprintf(stderr, "[shape] member %d: S=%d need %.0f cap %d hint %d overflow %d\n", h->batch_index, best_S, h->shape_need, best_cap, h->hint_count, (int)h->shape_overflow);
            } else {
                for (int S : {1, 2, 3, 4, 5, 6, 8, 12, 16}) {
                    if (S > 1 && S > h->n_itiles) continue;
                    const double need = group_est(S);
                    // Two capacities only: 3,328 entries (an 80 KB image: two workgroups per CU) or 6,400 (the whole LDS, one
                    // workgroup per CU).  Quantised so that a chain gets the same capacity -- hence re-sorts at the same steps --
                    // alone and inside a batch (whose capacity follows its largest member): their bitwise identity rests on it.
                    if (need * 1.3 + 64 > lds_max) continue;
                    const bool one_wg = need * 1.3 + 64 > 3328;
                    int cap = one_wg ? lds_max : 3328;
                    if (slack != 1.5) cap = std::max(64, (((int)(cap * slack / 1.5) + 63) / 64) * 64);   // tests shrink it to exercise the re-sort path
                    const double cost = need / (std::min(S, h->n_itiles) * 64.0) * (one_wg ? 1.3 : 1.0);   // staging work per i-atom
                    if (cost < best_cost) { best_cost = cost; best_S = S; best_cap = cap; }
                }
                if (h->tune.list_group > 0) { const int S = std::max(1, std::min(16, h->tune.list_group)); best_S = S; best_cap = group_est(S) * 1.3 + 64 > 3328 ? lds_max : 3328; }
                if (best_S > 0) h->shape_need = group_est(best_S);
            }
            // the lists are padded with a ghost record placed 0.5 nm outside the group's bounding sphere along the longest box edge
            // (kernels_nb.h: build_lists_body): that edge must exceed the group's diameter + 1 nm, with room for the group to spread
            if (best_S > 0) {
                const double longest = std::max(h->box[0], std::max(h->box[1], h->box[2]));
                if (2.0 * est_rad + 1.0 + 0.6 >= longest) best_S = 0;   // (est_rad: of group_est(best_S), the call just above)
            }
            if (best_S > 0 && best_cap <= NB_JCAP_MAX) {
                h->k1_mode = 2; h->S = best_S; jcap = std::min(best_cap, ((n + 63) / 64) * 64); h->jcap = jcap;
                h->waves_tile = 1; h->npart = 1; h->wpb = 4; h->seg_len = 64;
                double cap = rho * 4.0 / 3.0 * M_PI * rl * rl * rl * 1.7;   // mean neighbour count within cutoff+skin, with slack for dense regions
                if (h->tune.acap_scale > 0.0) cap *= h->tune.acap_scale;
                h->acap = std::max(64 * NB_ATOM_U, std::min(jcap, std::max(64, (((int)cap + 63) / 64) * 64)));   // (at least the prefetch window of nonbonded_atom_body: its loads stay inside the atom's row)
            }
        }
    }
    if (!want_frag && h->k1_mode != 2 && h->k1_mode == 1 && h->tune.k1_mode < 0 && !h->forbid_atom && h->precision == 0 && h->shape_S == 0) {
        // the per-atom lists were wanted and no group shape holds this system (scattered mobile atoms): fragment lists instead of the
        // sub-tile kernel -- they need the fragment-wise sort, so once more from the top
        h->forbid_atom = true;
        return sort_and_tile(h);
    }
    if (!want_frag) { h->frag_F = 0; h->frag_NI = 0; h->frag_nblk = 0; h->frag_ocap = 0; h->frag_icap = 0; h->frag_fpw = 1; h->frag_nwg = 0; h->frag_rel = false; }   // (a layout that left the fragment lists -- a straggler that came back -- is congruent with members that never had them)
    if (want_frag) {
        // fragment lists: no group lists but the alchemical tile's, whose capacity must not depend on the density (members of a
        // batch keep their own boxes under a barostat and still have to agree on the launch geometry)
        h->k1_mode = 3; h->S = 1; h->acap = 0; h->k1_iw = 64; h->seg_len = 64; h->waves_tile = 1; h->npart = 1; h->wpb = 4; h->fuse_forces = false;
        jcap = h->alch.empty() ? 64 : std::min(((n + 63) / 64) * 64, 8192); h->jcap = jcap;
    }
    h->hint_count = h->k1_mode == 2 ? h->jcap - h->jcap / 10 : h->jcap - h->jcap / 7;
    h->n_lists = (std::max(1, h->n_itiles) + h->S - 1) / h->S;
    if (h->n_itiles == 0 || h->k1_mode == 3) h->n_lists = 0;
    h->n_tiles = h->n_lists + (h->alch.empty() ? 0 : 1);
    const int nt = std::max(1, h->n_tiles);
    h->pool_cap = nt * MASK_QUOTA;
    h->PA = 1; while (h->PA < (int)h->alch.size()) h->PA <<= 1;
    h->k2_jiter = h->n_itiles * h->batch_R <= 32 && h->batch_R < 8 ? 1 : 8;   // measured at R = 256 (us per launch): 2: 160, 4: 123, 8: 110, 16: 113
    if (h->tune.k2_jiter > 0) h->k2_jiter = h->tune.k2_jiter;
    h->k2_jiter = std::min(h->k2_jiter, h->PA);   // an env block stages (256 / PA) * jiter <= K2_STAGE list entries in LDS
    {   // dense form: large batches in mixed precision whose alchemical group is free (no excluded environment partner) and small
        double half_min = 1e30; for (int k = 0; k < 3; k++) half_min = std::min(half_min, 0.5 * h->box[k]);
        int mobile_env = 0; for (int o : h->mobile) mobile_env += h->T->alch_local[o] < 0;
        // (its marking pass measures every list entry from the first alchemical atom and subtracts the other atoms' offsets: a pair
        // within the cutoff comes out as its minimum image as long as half a box edge exceeds cutoff + the group's extent; the
        // pair pass takes the minimum image itself)
        double ext = 0.0;
        for (size_t a = 1; a < h->alch.size(); a++) {
            double r2 = 0.0;
            for (int k = 0; k < 3; k++) { double d = h->hx[3 * h->alch[a] + k] - h->hx[3 * h->alch[0] + k]; d -= h->box[k] * std::rint(d / h->box[k]); r2 += d * d; }
            ext = std::max(ext, std::sqrt(r2));
        }
        h->k2_dense = h->precision == 0 && (h->k1_mode == 2 || want_frag) && !h->fuse_forces && h->batch_R >= 8 && !h->check_env_excl && !h->alch.empty() && h->alch.size() <= 16 &&
                      mobile_env <= K2D_MOB && half_min > h->cutoff + ext + 0.3 && h->tune.k2_dense != 0;
        if (h->k2_dense) h->k2_jiter = 1 << 20;   // one logical env block: k2_env_blocks() = 1 wherever the partial slabs are summed
        h->k2_f32 = h->k2_dense && h->tune.k2_dense != 2;   // (2: the round-5 fp64 body, kept as the reference the fp32 form is tested against)
    }
    h->k2_nblocks_env = k2_env_blocks(jcap, h->PA, h->k2_jiter);
    // exclusions in sorted space (self included)
    std::vector<int> ex_start(n + 1, 0), ex_idx;
    for (int s = 0; s < n; s++) {
        const int o = h->h_orig_of_sorted[s];
        std::vector<int> row; row.push_back(s);
        for (int p : h->T->excl[o]) row.push_back(h->h_sorted_of_orig[p]);
        std::sort(row.begin(), row.end());
        ex_idx.insert(ex_idx.end(), row.begin(), row.end());
        ex_start[s + 1] = (int)ex_idx.size();
    }
    // image
    SetupTimer tm_up(2);   // (from here on: the image and the uploads)
    const double sq = std::sqrt(ONE_4PI_EPS0);
    std::vector<AtomF> imf; std::vector<AtomD> imd;
    if (h->precision == 0) imf.resize(n); else imd.resize(n);
    for (int s = 0; s < n; s++) {
        const int o = h->h_orig_of_sorted[s];
        const bool al = h->T->alch_local[o] >= 0;
        const unsigned flags = (al ? FLAG_ALCH : 0u) | ((h->T->mass[o] != 0.0 && !al) ? FLAG_MOBILE : 0u);
        unsigned long long u[3];
        for (int k = 0; k < 3; k++) { double fr = h->hx[3 * o + k] / h->box[k]; fr -= std::floor(fr); if (fr >= 1.0) fr = 0.0; u[k] = (unsigned long long)(fr * 18446744073709551616.0); }
        const double q = al ? 0.0 : h->T->charge[o] * sq, hs = 0.5 * h->T->sigma[o], se = al ? 0.0 : 2.0 * std::sqrt(h->T->eps[o]);
        if (h->precision == 0) {
            AtomF& A = imf[s];
            A.x = (uint32_t)((u[0] + 0x80000000ull) >> 32); A.y = (uint32_t)((u[1] + 0x80000000ull) >> 32); A.z = (uint32_t)((u[2] + 0x80000000ull) >> 32);
            A.q = (float)q; A.hs = (float)hs; A.se = (float)se; A.flags = flags; A.orig = o;
        } else {
            AtomD& A = imd[s];
            A.x = u[0]; A.y = u[1]; A.z = u[2]; A.q = q; A.hs = hs; A.se = se; A.flags = flags; A.orig = o; A.pad = 0;
        }
    }
    try {
        if (h->precision == 0) h->d_img_f.upload(imf); else h->d_img_d.upload(imd);
        h->d_sorted_of_orig.upload(h->h_sorted_of_orig); h->d_orig_of_sorted.upload(h->h_orig_of_sorted);
        { std::vector<int> cs(h->clusters.size() * 4, 0);
          for (size_t c = 0; c < h->clusters.size(); c++) for (int a = 0; a < 4; a++) if (h->clusters[c].atoms[a] >= 0) cs[c * 4 + a] = h->h_sorted_of_orig[h->clusters[c].atoms[a]];
          h->d_cl_sorted.upload(cs);
          for (size_t c = 0; c < h->clusters.size(); c++) for (int a = 0; a < 4; a++) {
              const int at = h->clusters[c].atoms[a];
              h->h_recs[c].sorted[a] = cs[c * 4 + a]; h->h_recs[c].islot[a] = at >= 0 ? islot[at] : -1;
              const int row = at >= 0 && !h->h_row_of_orig.empty() ? h->h_row_of_orig[at] : -1;
              h->h_recs[c].e0[a] = row >= 0 ? h->h_row_start[row] : 0; h->h_recs[c].e1[a] = row >= 0 ? h->h_row_start[row + 1] : 0;
          }
          h->d_recs.upload(h->h_recs); }
        h->d_tile_atoms.upload(tile_atoms); h->d_islot.upload(islot);
        h->d_ex_start.upload(ex_start); h->d_ex_idx.upload(ex_idx);
        h->d_jrec.alloc((size_t)jcap);
        { std::vector<AlchARec> ar(h->alch.size());
          for (size_t a2 = 0; a2 < h->alch.size(); a2++) { AlchARec& r = ar[a2]; const int ao = h->alch[a2]; r.ao = ao; r.asrt = h->h_sorted_of_orig[ao]; r.pad = 0; r.sig = h->T->sigma[ao]; r.eps = std::sqrt(h->T->eps[ao]); r.q = h->T->charge[ao];   // (eps: its square root, see AlchARec)
            r.has_env_excl = 0; for (int p2 : h->T->excl[ao]) if (h->T->alch_local[p2] < 0) r.has_env_excl = 1; }
          h->d_arec.upload(ar); }
        // (per-atom lists: room for one group list per i-tile, so that a batch can move its members to a finer shape -- fewer tiles per
        // list -- without laying anybody out again: reshape_groups; 1.5 MB per member of the benchmark system)
        h->lists_alloc = h->k1_mode == 2 ? std::max(1, h->n_itiles) : 0;
        const int nt_alloc = h->k1_mode == 2 ? h->lists_alloc + 1 : nt;
        h->d_jlist.alloc((size_t)nt_alloc * jcap); h->d_jstage.alloc((size_t)nt_alloc * LIST_WAVES * ((((n + LIST_WAVES - 1) / LIST_WAVES) + 63) & ~63)); h->d_jcount.alloc(nt_alloc); h->d_batch_slot.alloc((size_t)nt_alloc * (jcap / 64));
        h->d_mask_pool.alloc((size_t)nt_alloc * MASK_QUOTA * 64);
        h->d_fpart.alloc((size_t)h->npart * 3 * h->n_islots);
        if (h->k1_mode == 2) {
            h->d_alist.alloc((size_t)h->n_islots * h->acap); h->d_acount.alloc(h->n_islots); h->d_aself.alloc(h->n_islots);
            {   // static exclusion rows per i-slot (kernels_nb.h: SX_ROW): the atom's sorted index, then its excluded partners'
                std::vector<int> rows((size_t)h->n_islots * SX_ROW, -1);
                for (int q = 0; q < h->n_islots; q++) {
                    int* r = rows.data() + (size_t)q * SX_ROW;
                    const int sidx = tile_atoms[q];
                    r[28] = 0; r[29] = 0x7fffffff; r[30] = -1;
                    if (sidx < 0) continue;
                    int nrow = 0; r[nrow++] = sidx;
                    for (int e = ex_start[sidx]; e < ex_start[sidx + 1]; e++) if (ex_idx[e] != sidx && nrow < SX_MAX) r[nrow++] = ex_idx[e];
                    r[28] = nrow;
                    for (int e = 0; e < nrow; e++) { r[29] = std::min(r[29], r[e]); r[30] = std::max(r[30], r[e]); }
                }
                h->d_sx_row.upload(rows);
            }
            if (h->prune_on) {
                h->d_plist.alloc((size_t)h->n_islots * h->acap); h->d_pcount.alloc(h->n_islots); h->d_pneed.alloc(h->n_islots);
                for (int k = 0; k < 3; k++) h->d_xprune[k].alloc(h->n_islots);
            }
            // packed group images: what the builder of the atoms' lists streams its candidates from (always), and what the nonbonded
            // kernel stages its LDS image from where most list entries are frozen atoms (the same condition as the pruned lists)
            h->mcap = std::min(jcap, (((int)h->mobile.size() + 63) / 64) * 64);
            h->d_pimg4.alloc((size_t)h->lists_alloc * jcap); h->d_pimg2.alloc((size_t)h->lists_alloc * jcap); h->d_pimgb.alloc((size_t)h->lists_alloc * jcap);
            h->d_mlist.alloc((size_t)h->lists_alloc * h->mcap * 2); h->d_mcount.alloc(h->lists_alloc);
        }
        if (h->k1_mode == 3) {
            // fragment layout of this sort: the fragments in image order, capacities from the fragment density
            const int F = (int)h->frag_atoms.size();
            // Lennard-Jones types of the environment: the distinct {sigma / 2, 2 sqrt(eps)} of the image, one byte per atom
            std::vector<float2> ljt; std::map<std::pair<uint32_t, uint32_t>, int> type_of;
            auto type_id = [&](const AtomF& A) {
                uint32_t a, b; memcpy(&a, &A.hs, 4); memcpy(&b, &A.se, 4);
                auto it = type_of.find({a, b});
                if (it != type_of.end()) return it->second;
                const int id = (int)ljt.size(); type_of[{a, b}] = id; ljt.push_back(make_float2(A.hs, A.se));
                return id;
            };
            std::vector<FragRec> recs; recs.reserve(F); std::vector<int> ifrag;
            for (int sx = 0; sx < n; sx++) {
                const int o = h->h_orig_of_sorted[sx], fr = h->frag_of_atom[o];
                if (fr < 0 || h->frag_pos_of_atom[o] != 0) continue;
                FragRec r; memset(&r, 0, sizeof r);
                const int cnt = h->frag_cnt[fr];
                r.types = (unsigned)cnt << 24;
                r.islot0 = -1;
                for (int b = 0; b < cnt; b++) {
                    const int ob = h->h_orig_of_sorted[sx + b];
                    if (h->T->mass[ob] != 0.0) { r.types |= 1u << (26 + b); if (r.islot0 < 0) r.islot0 = islot[ob]; }
                }
                if (FR_MOB(r.types)) ifrag.push_back((int)recs.size());
                for (int b = 0; b < 3; b++) {
                    const AtomF& A = imf[sx + std::min(b, cnt - 1)];
                    r.p[b] = make_uint4(A.x, A.y, A.z, b < cnt ? __builtin_bit_cast(unsigned, A.q) : 0u);
                    AtomF G = A; if (b >= cnt) { G.hs = 0.0f; G.se = 0.0f; }   // (an atom the fragment does not have: no charge, no epsilon; the masks leave its pairs out anyway)
                    r.types |= (unsigned)type_id(G) << (8 * b);
                }
                r.sid = fr; r.s0 = sx;
                recs.push_back(r);
            }
            if ((int)recs.size() != F) E_FAIL(h, "internal: %zu fragments laid out, %d expected", recs.size(), F);
            {   // the ghost the list rows are padded with (kernels_frag.h): record F
                FragRec g; memset(&g, 0, sizeof g);
                AtomF G; memset(&G, 0, sizeof G);
                const unsigned gt = (unsigned)type_id(G);
                for (int b = 0; b < 3; b++) g.p[b] = make_uint4(0x12345677u + 7919u * b, 0x9abcdef1u + 104729u * b, 0x31415927u + 15485863u * b, 0u);
                g.types = gt | (gt << 8) | (gt << 16) | (3u << 24);
                g.sid = -1; g.islot0 = -1; g.s0 = 0;
                recs.push_back(g);
            }
            if ((int)ljt.size() > FR_TYPES_MAX) E_FAIL(h, "internal: %zu Lennard-Jones types exceed the fragment kernel's table", ljt.size());
            h->frag_ntypes = (int)ljt.size();
            const double rho_f = F / vol;
            auto cap_for = [&](double margin) {
                const double r = h->cutoff + margin + 0.16;   // (0.16: two fragment radii of a water)
                double c = rho_f * 4.0 / 3.0 * M_PI * r * r * r * 1.35;
                if (h->tune.acap_scale > 0.0) c *= h->tune.acap_scale;
                return std::max(128, std::min(((F + 63) / 64) * 64 + 64, (((int)c + 63) / 64) * 64));
            };
            const int NI = (int)ifrag.size();
            h->frag_F = F; h->frag_NI = NI; h->frag_nblk = (F + 63) / 64;
            auto grown = [&](int cap, double g) { return std::min(((F + 63) / 64) * 64 + 64, (((int)(cap * g) + 63) / 64) * 64); };
            h->frag_ocap = grown(cap_for(h->skin), h->frag_grow_o); h->frag_icap = std::min(h->frag_ocap, grown(cap_for(h->frag_m), h->frag_grow_i));
            // the relative form of a pair's separation (kernels_frag.h: REL) wants cutoff + a fragment's reach (bonded atoms: well under
            // 0.6 nm) below half the shortest edge; smaller boxes (the 975-atom test box) take the fixed-point difference per pair
            h->frag_rel = 0.5 * std::min(h->box[0], std::min(h->box[1], h->box[2])) > h->cutoff + 0.6;
            // (4096 waves fill the chip at four per SIMD: a lone chain's 7,800 fragments go two per wave -- one round, the second fragment's
            // head requested while the first computes -- instead of one per wave in two rounds)
            h->frag_fpw = (int)std::max<long>(1, std::min<long>(4, ((long)NI * h->batch_R + 4095) / 4096));
            h->frag_nwg = (NI + 4 * h->frag_fpw - 1) / (4 * h->frag_fpw);
            h->d_ifrag.upload(ifrag);
            h->d_fimg.upload(recs); h->d_ljtab.upload(ljt); h->d_sp_start.upload(h->sp_start_h);
            { std::vector<int2> e = h->sp_ent_h; if (e.empty()) e.push_back(make_int2(-1, 0)); h->d_sp_ent.upload(e); }
            h->d_fbb.reserve(h->frag_nblk); h->d_fpos0.reserve(F);
            h->d_olist.reserve((size_t)NI * h->frag_ocap); h->d_ilist.reserve((size_t)NI * h->frag_icap); h->d_ocount.alloc(NI); h->d_icount.alloc(NI);
            h->d_xprune_s.alloc((size_t)3 * n);
        }
        // (behind the force kernels' partial sums: two per tile of the frozen-frozen energy kernel, then its tile boxes)
        h->d_epart_nb.alloc((h->k1_mode == 3 ? (size_t)2 * h->frag_nwg
                                             : (size_t)std::max(1, h->n_itiles) * h->npart * 2 * (h->k1_mode == 2 ? 1 : (h->k1_iw != 64 ? 64 / h->k1_iw : 1)))
                            + (size_t)(2 + FROZEN_BOX_DOUBLES) * ((n + FROZEN_TILE - 1) / FROZEN_TILE));
        { std::vector<int> ooi(h->n_islots, -1); for (int o = 0; o < n; o++) if (islot[o] >= 0) ooi[islot[o]] = o; h->d_orig_of_islot.upload(ooi);
          std::vector<FinRec> fr(h->n_islots + 64);
          auto fill = [&](FinRec& r, int atom) {
              r.atom = atom; r.sorted = 0; r.e0 = r.e1 = 0;
              if (atom >= 0) { r.sorted = h->h_sorted_of_orig[atom]; const int row = h->h_row_of_orig[atom]; if (row >= 0) { r.e0 = h->h_row_start[row]; r.e1 = h->h_row_start[row + 1]; } }
          };
          for (int q = 0; q < h->n_islots; q++) fill(fr[q], ooi[q]);
          for (int a = 0; a < 64; a++) fill(fr[h->n_islots + a], a < (int)h->alch.size() ? h->alch[a] : -1);
          h->d_finrecs.upload(fr); }
        h->d_fJ.alloc((size_t)9 * n);
        h->d_self_part.alloc((size_t)(h->k2_nblocks_env + 1) * 9 * 64); h->d_e_part.alloc((size_t)(h->k2_nblocks_env + 1) * K2_NP); h->d_mom_part.alloc((size_t)(h->n_islots / 64 + 2) * 6);
    } catch (std::string& e) { E_FAIL(h, "%s", e.c_str()); }
    for (int k = 0; k < 3; k++)   // the master positions on the device are the ones just sorted (every caller uploads / downloads first)
        if (hipMemcpy(h->d_x_sort[k].p, h->d_x[k].p, sizeof(double) * n, hipMemcpyDeviceToDevice) != hipSuccess) E_FAIL(h, "hipMemcpy D2D failed");
    h->sorted_ok = true; h->lists_forced = true; h->pass_valid = false; h->graph_valid = false;
    h->layout_R = h->batch_R;
    h->args_epoch++;
    std::vector<double>().swap(h->hx);   // (only a layout reads it, and every caller fills it first: 560 KB x 2048 chains of host memory otherwise)
    return 0;
}

// ------------------------------------------------------------------ launches
static IntArgs make_int_args(BluesEngine* h) {
    IntArgs A; memset(&A, 0, sizeof A);
    A.n = h->n; A.n_clusters = (int)h->clusters.size();
    A.recs = h->d_recs.p;
    for (int k = 0; k < 3; k++) { A.x[k] = h->d_x[k].p; A.v[k] = h->d_v[k].p; A.xbuild[k] = h->d_xbuild[k].p; }
    A.mass = h->d_mass.p;
    A.ftot = h->d_ftot.p; A.alch_self = h->d_alch_self.p;
    A.fJ = h->d_fJ.p; A.sorted_of_orig = h->d_sorted_of_orig.p; A.n_alch = (int)h->alch.size();
    A.alch_local_of_orig = h->d_alch_local.p;
    A.fpart = h->d_fpart.p; A.fent = h->d_fent.p; A.frec = h->pme ? h->d_frec.p : nullptr; A.self_part = h->d_self_part.p; A.e_part = h->d_e_part.p;
    A.jcount_alch = h->d_jcount.p + h->n_lists; A.n_islots = h->n_islots; A.npart = h->npart; A.n_entries = h->n_entries; A.PA = h->PA;
    A.k2_nblocks_env = h->k2_nblocks_env; A.k2_jiter = h->k2_jiter;
    for (int s3 = 0; s3 < 3; s3++) A.fin_le[s3] = h->fin_le[s3];
    A.fin_mask = h->fin_mask;
    A.hV = h->dt / std::max(1, h->n_V); A.hR = h->dt / std::max(1, h->n_R); A.inv_hR = 1.0 / A.hR;
    const double hO = h->dt / std::max(1, h->n_O);
    A.aO = std::exp(-h->gamma * hO); A.bO = std::sqrt(1.0 - std::exp(-2.0 * h->gamma * hO));
    A.dtL = h->dt; A.aL = std::exp(-h->gamma * h->dt); A.fsL = h->gamma == 0.0 ? h->dt : (1.0 - A.aL) / h->gamma; A.nsL = std::sqrt(1.0 - A.aL * A.aL);
    A.kT = h->kT; A.tol = h->tol; A.seed = h->seed; A.stream = (unsigned)h->replica * 4u; A.draw_base = h->prog_draw_base;
    A.noise = h->d_noise.p; A.mobile_index = h->d_mobile_index.p; A.n_mobile = (int)h->mobile.size(); A.n_noise = h->noise_valid ? h->n_noise : 0; A.noise_draw_base = h->noise_draw_base;
    A.box = make_box(h); A.periodic = h->nb_method == BLUES_NB_PME_DIRECT; A.cl_periodic = 0;
    A.img_f = h->precision == 0 ? h->d_img_f.p : nullptr; A.img_d = h->precision == 0 ? nullptr : h->d_img_d.p;
    A.half_skin2 = h->trig * h->trig; A.flags = h->d_flags.p; A.batch_req = batch_req_ptr(h);
    const bool pruned = h->prune_on && h->k1_mode == 2;
    for (int k = 0; k < 3; k++) { A.xprune[k] = pruned ? h->d_xprune[k].p : nullptr; A.fscale[k] = (float)(h->box[k] / 4294967296.0); }
    A.pneed = pruned ? h->d_pneed.p : nullptr;
    A.prune_trig2 = (float)(h->ptrig * h->ptrig);
    A.total_mass = h->total_mass; A.cm_part = h->d_cm_part.p; A.cm_nblocks = h->int_blocks;
    A.mom_part = h->d_mom_part.p; A.n_mom = h->n_islots / 64 + 2;
    A.acc = h->d_acc.p; A.work_trace = (h->tracing || h->ctrl_arg) ? h->d_trace.p : nullptr; A.trace_index = h->prog_trace;
    A.ctrl = h->ctrl_arg; A.stamps = h->d_stamps.p;
    A.prog = h->prog;
    return A;
}

static int resolve_finalize(BluesEngine* h);
static int flush_program(BluesEngine* h) {
    if (h->prog.n == 0) return 0;
    h->ecache.clear(); h->ke_cache_valid = false; h->acc_cache_valid = false;   // the launch may move atoms
    IntArgs A = make_int_args(h);
    // the steady-state program of "H V R O R V H" has a straight-line specialisation (same arithmetic)
    static const unsigned char P_CM[9] = {OP_V0, OP_H01, OP_END, OP_CM_PART, OP_H12, OP_V2, OP_R, OP_O, OP_R};
    static const unsigned char P_NC[8] = {OP_V0, OP_H01, OP_END, OP_H12, OP_V2, OP_R, OP_O, OP_R};
    static const unsigned char P_MD_CM[2] = {OP_CM_PART, OP_L};
    static const unsigned char P_MD_NC[1] = {OP_L};
    bool has_part = false;
    for (int q = 0; q < h->prog.n; q++) has_part |= h->prog.ops[q] == OP_CM_PART;
    const dim3 grid(h->int_blocks, batch_lead(h) ? h->batch->R() : 1), block(h->int_threads);
    IntDyn D; D.draw_base = A.draw_base; D.noise_draw_base = A.noise_draw_base; D.n_noise = A.n_noise; D.trace_index = A.trace_index; D.tracing = A.work_trace != nullptr; D.prog = A.prog;
    for (int s3 = 0; s3 < 3; s3++) D.fin_le[s3] = h->fin_le[s3];
    D.fin_mask = h->fin_mask;
    const RepCore* reps = h->batch ? h->batch->d_core.p : nullptr;
    const bool lead = batch_lead(h), dry = batch_dry(h);
    const bool is_cm = h->fast_step && h->prog.n == 9 && !memcmp(h->prog.ops, P_CM, 9), is_nc = h->fast_step && h->prog.n == 8 && !memcmp(h->prog.ops, P_NC, 8);
    const bool fused = h->fin_pending && (is_cm || is_nc);   // the pending sums of the last pass: formed by this launch itself ...
    if (h->fin_pending && !fused) { if (resolve_finalize(h)) return 1; }   // ... or by k_finalize first (any other program)
    h->fin_pending = false;
    if (is_cm) {
        if (fused) { if (lead) { if (h->batch->R() >= 64) hipLaunchKernelGGL((k_step_default_late_b<true>), grid, block, 0, h->cur, reps, D); else hipLaunchKernelGGL((k_step_default_b<true, true>), grid, block, 0, h->cur, reps, D); } else if (!dry) hipLaunchKernelGGL((k_step_default<true, true>), grid, block, 0, h->cur, A); }
        else if (lead) hipLaunchKernelGGL(k_step_default_b<true>, grid, block, 0, h->cur, reps, D);
        else if (!dry) hipLaunchKernelGGL(k_step_default<true>, grid, block, 0, h->cur, A);
    } else if (is_nc) {
        if (fused) { if (lead) { if (h->batch->R() >= 64) hipLaunchKernelGGL((k_step_default_late_b<false>), grid, block, 0, h->cur, reps, D); else hipLaunchKernelGGL((k_step_default_b<false, true>), grid, block, 0, h->cur, reps, D); } else if (!dry) hipLaunchKernelGGL((k_step_default<false, true>), grid, block, 0, h->cur, A); }
        else if (lead) hipLaunchKernelGGL(k_step_default_b<false>, grid, block, 0, h->cur, reps, D);
        else if (!dry) hipLaunchKernelGGL(k_step_default<false>, grid, block, 0, h->cur, A);
    } else if (h->fast_step && h->prog.n == 2 && !memcmp(h->prog.ops, P_MD_CM, 2)) {
        if (lead) hipLaunchKernelGGL(k_step_md_b<true>, grid, block, 0, h->cur, reps, D);
        else if (!dry) hipLaunchKernelGGL(k_step_md<true>, grid, block, 0, h->cur, A);
    } else if (h->fast_step && h->prog.n == 1 && !memcmp(h->prog.ops, P_MD_NC, 1) && h->pass_valid_for_l) {
        if (lead) hipLaunchKernelGGL(k_step_md_b<false>, grid, block, 0, h->cur, reps, D);
        else if (!dry) hipLaunchKernelGGL(k_step_md<false>, grid, block, 0, h->cur, A);
    } else {
        if (has_part) E_FAIL(h, "internal: OP_CM_PART outside a specialised program");
        if (lead) hipLaunchKernelGGL(k_integrate_b, grid, block, 0, h->cur, reps, D);
        else if (!dry) hipLaunchKernelGGL(k_integrate, grid, block, 0, h->cur, A);
    }
    h->vel_clean = false;
    h->st_launches++;
    h->prog.n = 0; h->prog_trace = -1; h->prog_draw_base = h->h_draw;
    HIP_OK(h, hipGetLastError());
    return 0;
}

static int emit(BluesEngine* h, int op) {
    if (h->prog.n == 0) h->prog_draw_base = h->h_draw;
    if (h->prog.n >= MAX_OPS) { if (flush_program(h)) return 1; h->prog_draw_base = h->h_draw; }
    h->prog.ops[h->prog.n++] = (unsigned char)op;
    if (op == OP_O || op == OP_L) h->h_draw++;
    return 0;
}

static ListArgs make_list_args(BluesEngine* h) {
    ListArgs a; memset(&a, 0, sizeof a);
    a.n = h->n; a.n_tiles = h->n_tiles; a.n_itiles = h->n_itiles; a.jcap = h->jcap; a.pool_cap = h->pool_cap;
    a.tile_atoms = h->d_tile_atoms.p; a.jlist = h->d_jlist.p; a.jstage = h->d_jstage.p; a.jcount = h->d_jcount.p; a.batch_slot = h->d_batch_slot.p;
    a.mask_pool = h->d_mask_pool.p; a.batch_req = batch_req_ptr(h); a.ex_start = h->d_ex_start.p; a.ex_idx = h->d_ex_idx.p; a.flags = h->d_flags.p;
    for (int k = 0; k < 3; k++) { a.x[k] = h->d_x[k].p; a.xbuild[k] = h->d_xbuild[k].p; }
    a.fJ = h->d_fJ.p; a.n_fJ = 9 * h->n;
    a.alch_jrec = h->alch.empty() ? nullptr : (void*)h->d_jrec.p; a.p_sigma = h->d_sigma.p; a.p_eps = h->d_eps.p; a.p_charge = h->d_charge.p;
    if (h->k1_mode == 2) { a.alist = h->d_alist.p; a.acount = h->d_acount.p; a.acap = h->acap; }
    a.S = h->S; a.n_lists = h->n_lists; a.hint_count = h->hint_count; a.no_sphere = h->tune.no_sphere;
    if (h->k1_mode == 2 && h->prune_on) { a.pneed = h->d_pneed.p; a.plist = h->d_plist.p; a.pcount = h->d_pcount.p; for (int k = 0; k < 3; k++) a.xprune[k] = h->d_xprune[k].p; }
    if (h->k1_mode == 2) { a.pimg4 = h->d_pimg4.p; a.pimg2 = h->d_pimg2.p; a.mlist = h->d_mlist.p; a.mcount = h->d_mcount.p; a.mcap = h->mcap; a.aself = h->d_aself.p; a.pimgb = h->d_pimgb.p; a.sx_row = h->d_sx_row.p; }
    return a;
}

static FragArgs make_frag_args(BluesEngine* h) {
    FragArgs a; memset(&a, 0, sizeof a);
    if (h->k1_mode != 3) return a;
    a.F = h->frag_F; a.NI = h->frag_NI; a.ifrag = h->d_ifrag.p; a.nblk = h->frag_nblk; a.n = h->n; a.ocap = h->frag_ocap; a.icap = h->frag_icap; a.fpw = h->frag_fpw; a.nwg = h->frag_nwg; a.n_islots = h->n_islots;
    a.fimg = h->d_fimg.p; a.ljtab = h->d_ljtab.p; a.ntypes = h->frag_ntypes; a.sp_start = h->d_sp_start.p; a.sp_ent = h->d_sp_ent.p; a.bb = h->d_fbb.p; a.fpos0 = h->d_fpos0.p;
    a.olist = h->d_olist.p; a.ocount = h->d_ocount.p; a.ilist = h->d_ilist.p; a.icount = h->d_icount.p; a.xprune = h->d_xprune_s.p;
    const double ro = h->cutoff + h->skin, ri = h->cutoff + std::min(h->frag_m, h->skin);
    a.ro2 = (float)(ro * ro) * 1.0001f + 1e-5f; a.ri2 = (float)(ri * ri) * 1.0001f + 1e-5f;
    {   // a frozen candidate: the pair's separation changes by ONE atom's displacement
        const double rof = h->cutoff + 0.5 * h->skin, rif = h->cutoff + 0.5 * std::min(h->frag_m, h->skin);
        a.ro2_f = (float)(rof * rof) * 1.0001f + 1e-5f; a.ri2_f = (float)(rif * rif) * 1.0001f + 1e-5f;
    }
    a.ptrig2 = h->frag_m < h->skin ? (float)(0.25 * h->frag_m * h->frag_m) : 1e30f;   // (inner = outer list: never pruned between rebuilds)
    for (int k = 0; k < 3; k++) a.scale[k] = (float)(h->box[k] / 4294967296.0);
    a.flags = h->d_flags.p; a.batch_req = batch_req_ptr(h); a.fpart = h->d_fpart.p; a.epart = h->d_epart_nb.p;
    a.half_min_edge = (float)(0.5 * std::min(h->box[0], std::min(h->box[1], h->box[2])));
    a.count_builds = h->n_tiles == 0;
    a.hint_blocks = h->frag_nblk < 64 ? 0x7fffffff : std::max(48, h->frag_nblk * 85 / 100);   // (a re-sort is 5-10 ms of host work per chain and a rebuild of everything: only when the blocks' boxes have stopped pruning anything -- asked for at 45 / 65 / 85 % of the blocks a full iteration of 64 chains took 2.80 / 2.25 / 2.14 s)
    return a;
}

// phase 0: the whole rebuild; 1: the group lists only (k_build_lists); 2: the atoms' own lists only (k_build_atom_lists);
// a batch's leader also: 3: the work list only (k_gather_stale_b); 4: the group lists without the work list (1 = 3 then 4)
template <typename R> static int launch_lists(BluesEngine* h, int force, int phase = 0) {
    if (h->tune.force_lists) force = 1;   // development: every launch rebuilds every list
    const ListArgs a = make_list_args(h);
    const typename Img<R>::Atom* img;
    if constexpr (sizeof(R) == 4) img = h->d_img_f.p; else img = h->d_img_d.p;
    if (phase == 2 || phase == 5) { }   // (5: the fragment-list kernels only)
    else if (batch_lead(h)) {
        if (phase != 4) hipLaunchKernelGGL(k_gather_stale_b<R>, dim3(1), dim3(LIST_THREADS), 0, h->cur, batch_reps_nb<R>(h->batch), h->batch->R(), force, h->batch->d_work.p);
        if (phase != 3) hipLaunchKernelGGL(k_build_lists_b<R>, dim3(std::min((h->n_tiles + 2) * h->batch->R(), REBUILD_GRID)), dim3(LIST_THREADS), 0, h->cur, batch_reps_nb<R>(h->batch), h->batch->d_work.p, h->n_tiles + 2, force);
    } else if (!batch_dry(h)) {
        hipLaunchKernelGGL(k_build_lists<R>, dim3(h->n_tiles + 2), dim3(LIST_THREADS), 0, h->cur, a, make_nbconst<R>(h), img, force);
    }
    if (phase != 1 && phase != 3 && phase != 4 && h->k1_mode == 2 && h->n_itiles > 0) {
        // second kernel of a rebuild (same gate): the atoms' own lists, ATOM_LIST_PARTS blocks per i-tile (kernels_nb.h: build_atom_lists_body)
        if (!batch_dry(h)) {
            const int items = h->n_itiles * ATOM_LIST_PARTS;
            if (batch_lead(h)) hipLaunchKernelGGL(k_build_atom_lists_b<R>, dim3(std::min(items * h->batch->R(), 4 * REBUILD_GRID)), dim3(ATOM_LIST_THREADS), 0, h->cur, batch_reps_nb<R>(h->batch), h->batch->d_work.p, items, force);
            else hipLaunchKernelGGL(k_build_atom_lists<R>, dim3(items), dim3(ATOM_LIST_THREADS), 0, h->cur, a, make_nbconst<R>(h), img, force);
        }
        h->st_launches++;
    }
    if constexpr (sizeof(R) == 4) {
        if (h->k1_mode == 3 && (phase == 0 || phase == 5) && !batch_dry(h)) {
            // fragment lists (kernels_frag.h): block boxes / prune trigger, then rebuild or prune where one is due
            const int wpb = FR_THREADS / 64;
            const int nb_pre = ((h->frag_NI + 63) / 64 + wpb - 1) / wpb, nb_box = (h->frag_nblk + wpb - 1) / wpb, nb_lists = (h->frag_NI + wpb - 1) / wpb;
            const int merged = nb_pre == nb_box;   // (every fragment is an i-fragment: the boxes ride in the first kernel)
            if (batch_lead(h)) {
                const int nrep = h->batch->R();
                hipLaunchKernelGGL(k_frag_pre_b, dim3(nb_pre * nrep), dim3(FR_THREADS), 0, h->cur, h->batch->d_nb_f.p, nb_pre, nrep, force, merged);
                hipLaunchKernelGGL(k_gather_frag_b, dim3(1), dim3(LIST_THREADS), 0, h->cur, h->batch->d_nb_f.p, nrep, force, h->batch->d_work_frag.p);
                if (!merged) hipLaunchKernelGGL(k_frag_boxes_b, dim3(std::min(nb_box * nrep, 4 * REBUILD_GRID)), dim3(FR_THREADS), 0, h->cur, h->batch->d_nb_f.p, h->batch->d_work_frag.p, nb_box, force);
                hipLaunchKernelGGL(k_frag_lists_b, dim3(std::min(nb_lists * nrep, 16 * REBUILD_GRID)), dim3(FR_THREADS), 0, h->cur, h->batch->d_nb_f.p, h->batch->d_work_frag.p, nb_lists, force);
            } else {
                const FragArgs fa = make_frag_args(h);
                hipLaunchKernelGGL(k_frag_pre, dim3(nb_pre), dim3(FR_THREADS), 0, h->cur, fa, h->d_img_f.p, force, merged);
                if (!merged) hipLaunchKernelGGL(k_frag_boxes, dim3(nb_box), dim3(FR_THREADS), 0, h->cur, fa, h->d_img_f.p, force);
                hipLaunchKernelGGL(k_frag_lists, dim3(nb_lists), dim3(FR_THREADS), 0, h->cur, fa, force);
            }
            h->st_launches += 2;
        }
    }
    h->st_launches++;
    HIP_OK(h, hipGetLastError());
    return 0;
}

template <typename R, bool ENERGY, int WPB> static void launch_nb_wpb(BluesEngine* h, const NbArgs<R>& a, const typename Img<R>::Atom* img) {
    const int blocks = std::max(1, h->n_itiles) * (h->waves_tile / WPB);
    if (batch_lead(h)) hipLaunchKernelGGL((k_nonbonded_b<R, ENERGY, WPB>), dim3(blocks * h->batch->R()), dim3(WPB * 64), 0, h->cur, batch_reps_nb<R>(h->batch), blocks, h->batch->R());
    else if (!batch_dry(h)) hipLaunchKernelGGL((k_nonbonded<R, ENERGY, WPB>), dim3(blocks), dim3(WPB * 64), 0, h->cur, a, make_nbconst<R>(h), img);
}

template <typename R> static NbArgs<R> make_nb_args(BluesEngine* h) {
    NbArgs<R> a; memset(&a, 0, sizeof a);
    a.n_itiles = h->n_itiles; a.jcap = h->jcap; a.n_islots = h->n_islots;
    a.seg_len = h->seg_len; a.waves_tile = h->waves_tile; a.npart = h->npart;
    a.tile_atoms = h->d_tile_atoms.p; a.jlist = h->d_jlist.p; a.jcount = h->d_jcount.p; a.batch_slot = h->d_batch_slot.p; a.mask_pool = h->d_mask_pool.p;
    a.fpart = h->d_fpart.p; a.epart = h->d_epart_nb.p; a.flags = h->d_flags.p; a.batch_req = batch_req_ptr(h);
    if (h->k1_mode == 2) {
        a.alist = h->d_alist.p; a.acount = h->d_acount.p; a.acap = h->acap;
        if (h->prune_on) {
            a.plist = h->d_plist.p; a.pcount = h->d_pcount.p; a.pneed = h->d_pneed.p;
            for (int k = 0; k < 3; k++) a.xprune[k] = h->d_xprune[k].p;
        }
        a.pimg4 = h->d_pimg4.p; a.pimg2 = h->d_pimg2.p; a.mlist = h->d_mlist.p; a.mcount = h->d_mcount.p; a.mcap = h->mcap; a.aself = h->d_aself.p;   // (the kernel always stages the packed image: it holds the ghost record the lists are padded with)
    }
    a.S = h->S; a.n_lists = h->n_lists;
    return a;
}

// per-atom-list kernel: one 1024-thread workgroup per group of i-tiles, the group's j-image in dynamic LDS: the 8-byte records at
// the bottom (room for the largest capacity), the 16-byte records behind them at a compile-time offset (kernels_nb.h: NB_LQ_BYTES)
template <bool ENERGY> static int launch_nb_atom(BluesEngine* h, const NbArgs<float>& a) {
    if (h->jcap > NB_JCAP_MAX) E_FAIL(h, "internal: list capacity %d beyond the per-atom-list kernel's LDS layout", h->jcap);
    const size_t lds = (size_t)NB_LQ_BYTES + (size_t)h->jcap * 16 + NB_LDS_TAIL + (size_t)8 * 64 * std::min(h->S, std::max(1, h->n_itiles));
    static thread_local size_t lds_set[2][2] = {{0, 0}, {0, 0}};   // [batched][ENERGY]: largest dynamic-LDS size the kernel was opened for
    const bool lead = batch_lead(h);
    if (batch_dry(h)) return 0;
    if (lds > lds_set[lead][ENERGY]) {
        hipError_t e = lead ? hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nonbonded_atom_b<ENERGY>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                            : hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nonbonded_atom<ENERGY>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) E_FAIL(h, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", lds, hipGetErrorString(e));
        lds_set[lead][ENERGY] = lds;
    }
    const int nb = std::max(1, h->n_lists);
    const int nthr = h->tune.k1_threads > 0 ? std::min(1024, std::max(64, (h->tune.k1_threads / 64) * 64)) : 1024;
    if (lead) {
        BluesBatch::EvPair* tp = ENERGY ? nullptr : k1t_begin(h->batch, h->cur);
        hipLaunchKernelGGL((k_nonbonded_atom_b<ENERGY>), dim3(nb * h->batch->R()), dim3(nthr), lds, h->cur, h->batch->d_nb_f.p, nb, h->batch->R());
        k1t_end(tp, h->cur);
    } else hipLaunchKernelGGL((k_nonbonded_atom<ENERGY>), dim3(nb), dim3(nthr), lds, h->cur, a, make_nbconst<float>(h), h->d_img_f.p);
    return 0;
}

template <bool ENERGY> static void launch_nb_sub(BluesEngine* h, const NbArgs<float>& a) {
    const int subs = 64 / h->k1_iw;
    const int waves = std::max(1, h->n_itiles) * subs * h->waves_tile;
    const dim3 grid((waves + 3) / 4), block(256);
    if (batch_dry(h)) return;
    if (batch_lead(h)) {
        const RepNb<float>* reps = h->batch->d_nb_f.p;
        const int nb = (waves + 3) / 4, nrep = h->batch->R();
        const dim3 bgrid(nb * nrep);
        if (h->k1_iw == 8) hipLaunchKernelGGL((k_nonbonded_sub_b<ENERGY, 8>), bgrid, block, 0, h->cur, reps, nb, nrep);
        else if (h->k1_iw == 16) hipLaunchKernelGGL((k_nonbonded_sub_b<ENERGY, 16>), bgrid, block, 0, h->cur, reps, nb, nrep);
        else hipLaunchKernelGGL((k_nonbonded_sub_b<ENERGY, 32>), bgrid, block, 0, h->cur, reps, nb, nrep);
        return;
    }
    if (h->k1_iw == 8) hipLaunchKernelGGL((k_nonbonded_sub<ENERGY, 8>), grid, block, 0, h->cur, a, make_nbconst<float>(h), h->d_img_f.p);
    else if (h->k1_iw == 16) hipLaunchKernelGGL((k_nonbonded_sub<ENERGY, 16>), grid, block, 0, h->cur, a, make_nbconst<float>(h), h->d_img_f.p);
    else hipLaunchKernelGGL((k_nonbonded_sub<ENERGY, 32>), grid, block, 0, h->cur, a, make_nbconst<float>(h), h->d_img_f.p);
}

template <typename R, bool ENERGY> static int launch_nonbonded(BluesEngine* h) {
    NbArgs<R> a = make_nb_args<R>(h);
    if constexpr (sizeof(R) == 4) {
        if (h->k1_mode == 3) {
            if (!batch_dry(h)) {
                if (batch_lead(h)) {
                    BluesBatch::EvPair* tp = ENERGY ? nullptr : k1t_begin(h->batch, h->cur);
                    if (h->frag_rel) hipLaunchKernelGGL((k_nonbonded_frag_b<ENERGY, true>), dim3(h->frag_nwg * h->batch->R()), dim3(FR_THREADS), 0, h->cur, h->batch->d_nb_f.p, h->frag_nwg, h->batch->R());
                    else hipLaunchKernelGGL((k_nonbonded_frag_b<ENERGY, false>), dim3(h->frag_nwg * h->batch->R()), dim3(FR_THREADS), 0, h->cur, h->batch->d_nb_f.p, h->frag_nwg, h->batch->R());
                    k1t_end(tp, h->cur);
                } else if (h->frag_rel) hipLaunchKernelGGL((k_nonbonded_frag<ENERGY, true>), dim3(h->frag_nwg), dim3(FR_THREADS), 0, h->cur, make_frag_args(h), make_nbconst<float>(h));
                else hipLaunchKernelGGL((k_nonbonded_frag<ENERGY, false>), dim3(h->frag_nwg), dim3(FR_THREADS), 0, h->cur, make_frag_args(h), make_nbconst<float>(h));
            }
            h->st_launches++;
            HIP_OK(h, hipGetLastError());
            return 0;
        }
        if (h->k1_mode == 2) {
            if (launch_nb_atom<ENERGY>(h, a)) return 1;
            h->st_launches++;
            HIP_OK(h, hipGetLastError());
            return 0;
        }
        if (h->k1_iw != 64) {
            launch_nb_sub<ENERGY>(h, a);
            h->st_launches++;
            HIP_OK(h, hipGetLastError());
            return 0;
        }
    }
    const typename Img<R>::Atom* img;
    if constexpr (sizeof(R) == 4) img = h->d_img_f.p; else img = h->d_img_d.p;
    switch (h->wpb) {
    case 1: launch_nb_wpb<R, ENERGY, 1>(h, a, img); break;
    case 2: launch_nb_wpb<R, ENERGY, 2>(h, a, img); break;
    case 4: launch_nb_wpb<R, ENERGY, 4>(h, a, img); break;
    case 8: launch_nb_wpb<R, ENERGY, 8>(h, a, img); break;
    default:
        if constexpr (sizeof(R) == 4) launch_nb_wpb<R, ENERGY, 16>(h, a, img); else launch_nb_wpb<R, ENERGY, 8>(h, a, img);
        break;
    }
    h->st_launches++;
    HIP_OK(h, hipGetLastError());
    return 0;
}

static AlchArgs make_alch_args(BluesEngine* h, const double ls[3], const double le[3], int slot_mask);
static AlchDyn make_alch_dyn(const AlchArgs& A) { AlchDyn d; for (int s = 0; s < 3; s++) { d.ls[s] = A.ls[s]; d.le[s] = A.le[s]; } d.slot_mask = A.slot_mask; return d; }

// part (a batch's leader only; kernels_batch.h: k_alchemical_b): 0 every member, 1 those that do not rebuild their lists in this
// force pass, 2 those that do (both from the work list of the rebuild)
static int launch_alchemical(BluesEngine* h, const double ls[3], const double le[3], int slot_mask, int part = 0) {
    if (h->alch.empty()) return 0;
    AlchArgs A = make_alch_args(h, ls, le, slot_mask);
    const bool fast = h->precision == 0;
    if (h->k2_dense) {
        // dense env pairs: one workgroup per chain; the alchemical x alchemical block is the other kernel's (a grid of self blocks only)
        if (batch_dry(h)) { h->st_launches++; return 0; }
        const bool lead = batch_lead(h);
        if (h->k2_f32) {
            // the fp32 form (kernels_alch.h: alchemical_dense32_body): its LDS record depends on how many force slots the pass produces
            static thread_local bool opened32[2] = {false, false};
            if (!opened32[lead]) {
                hipError_t e = hipSuccess;
#define OPEN32(M, NS) do { if (e == hipSuccess) e = lead ? hipFuncSetAttribute(reinterpret_cast<const void*>(&k_alchemical_dense32_b<M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(K2FLds<NS>)) \
                                                          : hipFuncSetAttribute(reinterpret_cast<const void*>(&k_alchemical_dense32<M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(K2FLds<NS>)); } while (0)
                OPEN32(5, 2); OPEN32(2, 1); OPEN32(-1, 3);
#undef OPEN32
                if (e != hipSuccess) E_FAIL(h, "hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(e));
                opened32[lead] = true;
                if (h->tune.debug_lists && lead) { int nb = -1; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_alchemical_dense32_b<5>, K2F_THREADS, sizeof(K2FLds<2>)); fprintf(stderr, "[dense32] %zu bytes of LDS per workgroup: %d workgroups per CU\n", sizeof(K2FLds<2>), nb); }
            }
            if (lead) {
                const int nrep = h->batch->R();
                const AlchDyn D = make_alch_dyn(A);
                // (part 5 / 6: the dense kernel only, for the members that do not / do rebuild their lists in this pass -- force_pass, fork mode 4)
                const int* stale = part >= 5 ? h->batch->d_work.p : nullptr;
#define DENSE32_B(M, NS) do { if (part != 3) hipLaunchKernelGGL((k_alchemical_dense32_b<M>), dim3(nrep), dim3(K2F_THREADS), sizeof(K2FLds<NS>), h->cur, h->batch->d_core.p, D, stale, part == 6 ? 1 : 0); \
                              if (part < 4) hipLaunchKernelGGL((k_alchemical_b<true, M>), dim3(nrep), dim3(256), 0, h->cur, h->batch->d_core.p, D, 1, nrep, (const int*)nullptr); } while (0)
                if (slot_mask == 5) DENSE32_B(5, 2); else if (slot_mask == 2) DENSE32_B(2, 1); else DENSE32_B(-1, 3);
#undef DENSE32_B
            } else {
                if (slot_mask == 5) hipLaunchKernelGGL(k_alchemical_dense32<5>, dim3(1), dim3(K2F_THREADS), sizeof(K2FLds<2>), h->cur, A);
                else if (slot_mask == 2) hipLaunchKernelGGL(k_alchemical_dense32<2>, dim3(1), dim3(K2F_THREADS), sizeof(K2FLds<1>), h->cur, A);
                else hipLaunchKernelGGL(k_alchemical_dense32<-1>, dim3(1), dim3(K2F_THREADS), sizeof(K2FLds<3>), h->cur, A);
                hipLaunchKernelGGL(k_alchemical<true>, dim3(1), dim3(256), 0, h->cur, A);
            }
            h->st_launches++;
            HIP_OK(h, hipGetLastError());
            return 0;
        }
        static thread_local bool opened[2] = {false, false};
        const size_t lds = sizeof(K2DLds);
        if (!opened[lead]) {
            hipError_t e = hipSuccess;
#define OPEN(M) do { if (e == hipSuccess) e = lead ? hipFuncSetAttribute(reinterpret_cast<const void*>(&k_alchemical_dense_b<M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) \
                                                    : hipFuncSetAttribute(reinterpret_cast<const void*>(&k_alchemical_dense<M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); } while (0)
            OPEN(5); OPEN(2); OPEN(-1);
#undef OPEN
            if (e != hipSuccess) E_FAIL(h, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", lds, hipGetErrorString(e));
            opened[lead] = true;
        }
        if (lead) {
            const int nrep = h->batch->R();
            const AlchDyn D = make_alch_dyn(A);
            // (part 3: the alchemical x alchemical block only; part 4: the dense kernel only -- force_pass, fork mode 3)
#define DENSE_B(M) do { if (part != 3) hipLaunchKernelGGL((k_alchemical_dense_b<M>), dim3(nrep), dim3(K2D_THREADS), lds, h->cur, h->batch->d_core.p, D); \
                        if (part != 4) hipLaunchKernelGGL((k_alchemical_b<true, M>), dim3(nrep), dim3(256), 0, h->cur, h->batch->d_core.p, D, 1, nrep, (const int*)nullptr); } while (0)
            if (slot_mask == 5) DENSE_B(5); else if (slot_mask == 2) DENSE_B(2); else DENSE_B(-1);
#undef DENSE_B
        } else {
            if (slot_mask == 5) hipLaunchKernelGGL(k_alchemical_dense<5>, dim3(1), dim3(K2D_THREADS), lds, h->cur, A);
            else if (slot_mask == 2) hipLaunchKernelGGL(k_alchemical_dense<2>, dim3(1), dim3(K2D_THREADS), lds, h->cur, A);
            else hipLaunchKernelGGL(k_alchemical_dense<-1>, dim3(1), dim3(K2D_THREADS), lds, h->cur, A);
            hipLaunchKernelGGL(k_alchemical<true>, dim3(1), dim3(256), 0, h->cur, A);
        }
        h->st_launches++;
        HIP_OK(h, hipGetLastError());
        return 0;
    }
    if (batch_lead(h)) {
        const int nb = std::min(h->k2_nblocks_env, K2_PHYS) + 1, nrep = h->batch->R();   // (a lone chain keeps one block per logical block: shortest chain)
        const AlchDyn D = make_alch_dyn(A);
        const dim3 g(part == 2 ? std::min(nb * nrep, 8 * REBUILD_GRID) : nb * nrep), b(256);
        const int* work = h->batch->d_work.p;
#define ALCH_B(F, M) do { if (part == 2) hipLaunchKernelGGL((k_alchemical_stale_b<F, M>), g, b, 0, h->cur, h->batch->d_core.p, D, nb, work); \
                          else hipLaunchKernelGGL((k_alchemical_b<F, M>), g, b, 0, h->cur, h->batch->d_core.p, D, nb, nrep, part == 1 ? work : (const int*)nullptr); } while (0)
        if (fast) { if (slot_mask == 5) ALCH_B(true, 5); else if (slot_mask == 2) ALCH_B(true, 2); else ALCH_B(true, -1); }
        else { if (slot_mask == 5) ALCH_B(false, 5); else if (slot_mask == 2) ALCH_B(false, 2); else ALCH_B(false, -1); }
#undef ALCH_B
    } else if (!batch_dry(h)) {
        if (fast) hipLaunchKernelGGL(k_alchemical<true>, dim3(h->k2_nblocks_env + 1), dim3(256), 0, h->cur, A);
        else hipLaunchKernelGGL(k_alchemical<false>, dim3(h->k2_nblocks_env + 1), dim3(256), 0, h->cur, A);
    }
    h->st_launches++;
    HIP_OK(h, hipGetLastError());
    return 0;
}

static AlchArgs make_alch_args(BluesEngine* h, const double ls[3], const double le[3], int slot_mask) {
    AlchArgs A; memset(&A, 0, sizeof A);
    A.jrec = h->d_jrec.p; A.arec = h->d_arec.p;
    A.n = h->n; A.n_alch = (int)h->alch.size(); A.PA = h->PA; A.jcap = h->jcap; A.nblocks_env = h->k2_nblocks_env; A.jiter = h->k2_jiter;
    A.flags_overflow = h->d_flags.p ? &h->d_flags.p->list_overflow : nullptr;
    A.alch_orig = h->d_alch_orig.p; A.jlist = h->d_jlist.p + (size_t)h->n_lists * h->jcap; A.jcount = h->d_jcount.p + h->n_lists;
    A.orig_of_sorted = h->d_orig_of_sorted.p; A.sorted_of_orig = h->d_sorted_of_orig.p;
    for (int k = 0; k < 3; k++) A.x[k] = h->d_x[k].p;
    A.charge = h->d_charge.p; A.sigma = h->d_sigma.p; A.eps = h->d_eps.p; A.ex_start = h->d_ex_start.p; A.ex_idx = h->d_ex_idx.p;
    A.exc_start = h->d_exc_start.p; A.exc_partner = h->d_exc_partner.p; A.exc_owner = h->d_exc_owner.p; A.exc_is_env = h->d_exc_is_env.p; A.exc_params = h->d_exc_params.p;
    A.box = make_box(h); A.rc2 = h->cutoff * h->cutoff; A.alpha = h->alpha; A.sc_alpha = h->sc_alpha;
    A.pme = h->nb_method == BLUES_NB_PME_DIRECT; A.annih_elec = h->annih_elec; A.annih_ster = h->annih_ster; A.slot_mask = slot_mask; A.check_env_excl = h->check_env_excl;
    for (int s = 0; s < 3; s++) { A.ls[s] = ls[s]; A.le[s] = le[s]; }
    A.fJ = h->d_fJ.p; A.self_part = h->d_self_part.p; A.e_part = h->d_e_part.p; A.ctrl = h->ctrl_arg;
    A.img = h->precision == 0 ? h->d_img_f.p : nullptr;
    for (int k = 0; k < 3; k++) A.fscale[k] = (float)(h->box[k] / 4294967296.0);
    return A;
}

static BondedArgs make_bonded_args(BluesEngine* h) {
    BondedArgs B; memset(&B, 0, sizeof B);
    B.n_rows = h->n_rows; B.row_atom = h->d_row_atom.p; B.row_start = h->d_row_start.p;
    B.ent_type = h->d_ent_type.p; B.ent_term = h->d_ent_term.p; B.ent_role = h->d_ent_role.p;
    for (int ty = 0; ty < T_NTYPES; ty++) { B.n_terms[ty] = h->n_terms[ty]; B.atoms[ty] = h->d_term_atoms[ty].p; B.params[ty] = h->d_term_params[ty].p; }
    B.restr_k = h->restr_k; B.ewald_alpha = h->alpha;
    for (int k = 0; k < 3; k++) B.x[k] = h->d_x[k].p;
    B.box = make_box(h); B.periodic = h->nb_method == BLUES_NB_PME_DIRECT; B.fent = h->d_fent.p; B.n_entries = h->n_entries; B.n = h->n; B.epart = h->d_epart_b.p;
    B.n_mobile = (int)h->mobile.size(); B.n_noise = h->n_noise; B.mobile_atoms = h->d_mobile_atoms.p; B.noise = h->d_noise.p;
    B.seed = h->seed; B.stream = (unsigned)h->replica * 4u; B.draw_base = h->h_draw; B.n_entry_blocks = (h->n_entries + 127) / 128; B.ctrl = h->ctrl_arg;
    return B;
}

static int launch_bonded(BluesEngine* h, bool with_noise) {
    BondedArgs B = make_bonded_args(h);
    const int nb_noise = with_noise ? ((int)h->mobile.size() * h->n_noise + 127) / 128 : 0;
    if (with_noise) { h->noise_draw_base = h->h_draw; h->noise_valid = true; }
    if (B.n_entry_blocks + nb_noise > 0) {
        BondedDyn D; D.draw_base = B.draw_base; D.n_entry_blocks = B.n_entry_blocks;
        if (batch_lead(h)) hipLaunchKernelGGL(k_bonded_entries_b, dim3(B.n_entry_blocks + nb_noise, h->batch->R()), dim3(128), 0, h->cur, h->batch->d_core.p, D);
        else if (!batch_dry(h)) hipLaunchKernelGGL(k_bonded_entries, dim3(B.n_entry_blocks + nb_noise), dim3(128), 0, h->cur, B);
        h->st_launches++;
    }
    HIP_OK(h, hipGetLastError());
    return 0;
}

static FinArgs make_fin_args(BluesEngine* h, const double le[3], int slot_mask = 7) {
    FinArgs F; memset(&F, 0, sizeof F);
    F.n = h->n; F.n_islots = h->n_islots; F.npart = h->npart; F.n_alch = (int)h->alch.size(); F.PA = h->PA; F.k2_nblocks_env = h->k2_nblocks_env; F.k2_jiter = h->k2_jiter; F.n_entries = h->n_entries;
    F.recs = h->d_finrecs.p; F.orig_of_islot = h->d_orig_of_islot.p; F.row_of_orig = h->d_row_of_orig.p; F.row_start = h->d_row_start.p;
    F.fpart = h->d_fpart.p; F.fent = h->d_fent.p; F.fJ = h->d_fJ.p; F.sorted_of_orig = h->d_sorted_of_orig.p; F.alch_orig = h->d_alch_orig.p;
    F.self_part = h->d_self_part.p; F.e_part = h->d_e_part.p; F.jcount_alch = h->d_jcount.p + h->n_lists;
    for (int s = 0; s < 3; s++) F.le[s] = le[s];
    F.slot_mask = slot_mask;
    F.ftot = h->d_ftot.p; F.alch_self = h->d_alch_self.p; F.acc = h->d_acc.p; F.ctrl = h->ctrl_arg;
    for (int k = 0; k < 3; k++) F.v[k] = h->d_v[k].p;
    F.mass = h->d_mass.p; F.mom_part = h->d_mom_part.p;
    F.frec = h->pme ? h->d_frec.p : nullptr;
    return F;
}

static int launch_finalize(BluesEngine* h, const double le[3], int slot_mask = 7) {
    FinArgs F = make_fin_args(h, le, slot_mask);
    const int nblk = h->n_islots / 64 + (F.n_alch > 0 ? 1 + 9 + 1 : 0);
    FinDyn FD; for (int s = 0; s < 3; s++) FD.le[s] = le[s];
    FD.slot_mask = slot_mask;
    if (batch_lead(h)) {
        if (h->batch->R() >= 8) hipLaunchKernelGGL(k_finalize_b<true>, dim3(std::max(1, nblk), h->batch->R()), dim3(256), 0, h->cur, h->batch->d_core.p, FD);
        else hipLaunchKernelGGL(k_finalize_b<false>, dim3(std::max(1, nblk), h->batch->R()), dim3(256), 0, h->cur, h->batch->d_core.p, FD);
    }
    else if (!batch_dry(h)) hipLaunchKernelGGL(k_finalize, dim3(std::max(1, nblk)), dim3(256), 0, h->cur, F);
    h->st_launches++;
    HIP_OK(h, hipGetLastError());
    return 0;
}

// Round 4: the force pass of a step does not launch k_finalize where the step kernel can form the sums itself
// (kernels_integrate.h: step_default_body<CM, true>): one block holds every cluster of the chain, both kicks of the steady-state
// program are served by this pass (slots 0 and 2).  What else needs the summed forces resolves the pending sums first.
// (npart == 1: the per-atom-list kernel's single slab; a lone chain's tile kernel leaves dozens of partial slabs, and four atoms per
// thread summing them one after the other took longer than k_finalize's thread per atom: 77 against 52 us per step)
static bool fin_fusable(const BluesEngine* h) { return h->fast_step && h->int_blocks == 1 && (h->int_threads == 256 || (h->int_threads == 128 && h->clusters_packed)) && h->npart == 1 && h->n_entries <= STEP_FENT_LDS && h->tune.fuse_finalize != 0 && !h->ctrl_arg; }   // (one slab, the bonded entries fit the step kernel's LDS: step_default_body<CM, true>)
static int launch_finalize_deferred(BluesEngine* h, const double le[3], int slot_mask) {
    if (!fin_fusable(h) || (slot_mask & 5) != 5) { h->fin_pending = false; return launch_finalize(h, le, slot_mask); }
    h->fin_pending = true; h->fin_mask = slot_mask;
    for (int s = 0; s < 3; s++) h->fin_le[s] = le[s];
    return 0;
}
static int resolve_finalize(BluesEngine* h) {
    if (!h->fin_pending) return 0;
    h->fin_pending = false;
    return launch_finalize(h, h->fin_le, h->fin_mask);
}

static int launch_bonded_and_finalize(BluesEngine* h, const double le[3], bool with_noise, int slot_mask = 7) {
    if (launch_bonded(h, with_noise)) return 1;
    return launch_finalize(h, le, slot_mask);
}

template <typename R> static int launch_forces_fused(BluesEngine* h, const double ls[3], const double le[3]) {
    NbArgs<R> a = make_nb_args<R>(h);
    const typename Img<R>::Atom* img;
    if constexpr (sizeof(R) == 4) img = h->d_img_f.p; else img = h->d_img_d.p;
    AlchArgs A = make_alch_args(h, ls, le, h->pass_fmask);
    BondedArgs B = make_bonded_args(h);
    B.n_entry_blocks = (h->n_entries + 255) / 256;
    const int nb1 = std::max(1, h->n_itiles) * (h->waves_tile / 4);
    const int nb2 = h->alch.empty() ? 0 : h->k2_nblocks_env + 1;
    const int nb3 = B.n_entry_blocks + ((int)h->mobile.size() * h->n_noise + 255) / 256;
    h->noise_draw_base = h->h_draw; h->noise_valid = true;
    BondedDyn BD; BD.draw_base = B.draw_base; BD.n_entry_blocks = B.n_entry_blocks;
    if (batch_lead(h)) hipLaunchKernelGGL(k_forces_fused_b<R>, dim3(nb1 + nb2 + nb3, h->batch->R()), dim3(256), 0, h->cur, batch_reps_nb<R>(h->batch), h->batch->d_core.p, make_alch_dyn(A), BD, nb1, nb2);
    else if (!batch_dry(h)) hipLaunchKernelGGL(k_forces_fused<R>, dim3(nb1 + nb2 + nb3), dim3(256), 0, h->cur, a, make_nbconst<R>(h), img, A, B, nb1, nb2);
    h->st_launches++;
    HIP_OK(h, hipGetLastError());
    return 0;
}

static int launch_forces_fused_sub(BluesEngine* h, const double ls[3], const double le[3]) {
    NbArgs<float> a = make_nb_args<float>(h);
    AlchArgs A = make_alch_args(h, ls, le, h->pass_fmask);
    BondedArgs B = make_bonded_args(h);
    B.n_entry_blocks = (h->n_entries + 255) / 256;
    const int subs = 64 / h->k1_iw;
    const int nb1 = (std::max(1, h->n_itiles) * subs * h->waves_tile + 3) / 4;
    const int nb2 = h->alch.empty() ? 0 : h->k2_nblocks_env + 1;
    const int nb3 = B.n_entry_blocks + ((int)h->mobile.size() * h->n_noise + 255) / 256;
    h->noise_draw_base = h->h_draw; h->noise_valid = true;
    const dim3 grid(nb1 + nb2 + nb3, batch_lead(h) ? h->batch->R() : 1), block(256);
    if (batch_lead(h)) {
        BondedDyn BD; BD.draw_base = B.draw_base; BD.n_entry_blocks = B.n_entry_blocks;
        const AlchDyn AD = make_alch_dyn(A);
        if (h->k1_iw == 8) hipLaunchKernelGGL(k_forces_fused_sub_b<8>, grid, block, 0, h->cur, h->batch->d_nb_f.p, h->batch->d_core.p, AD, BD, nb1, nb2);
        else if (h->k1_iw == 16) hipLaunchKernelGGL(k_forces_fused_sub_b<16>, grid, block, 0, h->cur, h->batch->d_nb_f.p, h->batch->d_core.p, AD, BD, nb1, nb2);
        else hipLaunchKernelGGL(k_forces_fused_sub_b<32>, grid, block, 0, h->cur, h->batch->d_nb_f.p, h->batch->d_core.p, AD, BD, nb1, nb2);
    } else if (!batch_dry(h)) {
        if (h->k1_iw == 8) hipLaunchKernelGGL(k_forces_fused_sub<8>, grid, block, 0, h->cur, a, make_nbconst<float>(h), h->d_img_f.p, A, B, nb1, nb2);
        else if (h->k1_iw == 16) hipLaunchKernelGGL(k_forces_fused_sub<16>, grid, block, 0, h->cur, a, make_nbconst<float>(h), h->d_img_f.p, A, B, nb1, nb2);
        else hipLaunchKernelGGL(k_forces_fused_sub<32>, grid, block, 0, h->cur, a, make_nbconst<float>(h), h->d_img_f.p, A, B, nb1, nb2);
    }
    h->st_launches++;
    HIP_OK(h, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ reciprocal space (kernels_pme.h)
// B-spline moduli, eterm(m) = ONE_4PI_EPS0 exp(-pi^2 m^2 / alpha^2) / (pi V m^2 |b_x b_y b_z|^2), the transform's twiddle tables and the
// box-dependent constants (self term, neutralising background, dispersion correction): same formulas as the oracle.
template <typename T> static BluesEngine::PmeBufs<T>& pme_bufs(BluesEngine* h) { if constexpr (sizeof(T) == 4) return h->pme_f; else return h->pme_d; }

static void pme_splines_host(double dr, int order, double* w) {
    w[order - 1] = 0.0; w[1] = dr; w[0] = 1.0 - dr;
    for (int l = 3; l < order; l++) {
        const double div = 1.0 / (l - 1.0);
        w[l - 1] = div * dr * w[l - 2];
        for (int k = 1; k < l - 1; k++) w[l - k - 1] = div * ((dr + k) * w[l - k - 2] + (l - k - dr) * w[l - k - 1]);
        w[0] = div * (1.0 - dr) * w[0];
    }
    const double div = 1.0 / (order - 1.0);
    w[order - 1] = div * dr * w[order - 2];
    for (int k = 1; k < order - 1; k++) w[order - k - 1] = div * ((dr + k) * w[order - k - 2] + (order - k - dr) * w[order - k - 1]);
    w[0] = div * (1.0 - dr) * w[0];
}

template <typename T> static int pme_tables(BluesEngine* h) {
    auto& P = pme_bufs<T>(h);
    const int K[3] = {h->pme_K[0], h->pme_K[1], h->pme_K[2]};
    const size_t ng = (size_t)K[0] * K[1] * K[2];
    std::vector<double> mod[3]; double w[8];
    pme_splines_host(0.0, h->pme_order, w);
    try {
        for (int d = 0; d < 3; d++) {
            mod[d].resize(K[d]);
            std::vector<T> tc(K[d]), ts(K[d]);
            for (int m = 0; m < K[d]; m++) {
                tc[m] = (T)std::cos(2.0 * M_PI * m / K[d]); ts[m] = (T)std::sin(2.0 * M_PI * m / K[d]);
                double sc = 0.0, ss = 0.0;
                for (int k = 0; k < h->pme_order; k++) { const double a = 2.0 * M_PI * m * (k + 1) / K[d]; sc += w[k] * std::cos(a); ss += w[k] * std::sin(a); }
                mod[d][m] = sc * sc + ss * ss;
            }
            for (int m = 0; m < K[d]; m++) if (mod[d][m] < 1e-7) mod[d][m] = 0.5 * (mod[d][(m + K[d] - 1) % K[d]] + mod[d][(m + 1) % K[d]]);
            P.tw_cos[d].upload(tc); P.tw_sin[d].upload(ts);
        }
        const double V = h->box[0] * h->box[1] * h->box[2], al = h->alpha;
        std::vector<T> et(ng);
        for (int kx = 0; kx < K[0]; kx++) for (int ky = 0; ky < K[1]; ky++) for (int kz = 0; kz < K[2]; kz++) {
            const size_t g = ((size_t)kx * K[1] + ky) * K[2] + kz;
            if (!kx && !ky && !kz) { et[g] = (T)0; continue; }
            const double mx = (kx < (K[0] + 1) / 2 ? kx : kx - K[0]) / h->box[0], my = (ky < (K[1] + 1) / 2 ? ky : ky - K[1]) / h->box[1], mz = (kz < (K[2] + 1) / 2 ? kz : kz - K[2]) / h->box[2];
            const double m2 = mx * mx + my * my + mz * mz;
            et[g] = (T)(ONE_4PI_EPS0 * std::exp(-M_PI * M_PI * m2 / (al * al)) / (M_PI * V * m2 * mod[0][kx] * mod[1][ky] * mod[2][kz]));
        }
        P.eterm.upload(et);
        P.a_re.alloc(ng); P.a_im.alloc(ng); P.b_re.alloc(ng); P.b_im.alloc(ng); P.qf_re.alloc(ng); P.qf_im.alloc(ng); P.phi_f.alloc(ng);
        h->d_pme_acc.alloc(ng); h->d_pme_e.alloc(4); h->d_frec.alloc((size_t)3 * h->n);
    } catch (std::string& e) { E_FAIL(h, "%s", e.c_str()); }
    // constants of the box: self term, neutralising background, dispersion correction
    double qsum = 0.0, q2 = 0.0;
    for (double q : h->T->qn) { qsum += q; q2 += q * q; }
    const double V = h->box[0] * h->box[1] * h->box[2];
    h->e_ewald_const = -ONE_4PI_EPS0 * h->alpha / std::sqrt(M_PI) * q2 - M_PI * ONE_4PI_EPS0 * qsum * qsum / (2.0 * V * h->alpha * h->alpha);
    h->e_disp = 0.0;
    if (h->disp_corr) {   // [recalled: OpenMM NonbondedForceImpl::calcDispersionCorrection without switching function]; alchemical atoms enter with epsilon 0
        std::vector<double> cs, ce; std::vector<long long> cn;
        for (int i = 0; i < h->n; i++) {
            const double e = h->T->alch_local[i] >= 0 ? 0.0 : h->T->eps[i], sg = h->T->sigma[i];
            size_t c = 0; for (; c < cs.size(); c++) if (cs[c] == sg && ce[c] == e) break;
            if (c == cs.size()) { cs.push_back(sg); ce.push_back(e); cn.push_back(0); }
            cn[c]++;
        }
        double s12 = 0.0, s6 = 0.0;
        for (size_t a = 0; a < cs.size(); a++) for (size_t b = a; b < cs.size(); b++) {
            const double cnt = a == b ? 0.5 * cn[a] * (cn[a] + 1.0) : (double)cn[a] * cn[b];
            const double sg = 0.5 * (cs[a] + cs[b]), e = std::sqrt(ce[a] * ce[b]), s2 = sg * sg, s6_ = s2 * s2 * s2;
            s12 += cnt * e * s6_ * s6_; s6 += cnt * e * s6_;
        }
        const double tot = 0.5 * h->n * (h->n + 1.0), rc = h->cutoff, rc3 = rc * rc * rc, rc9 = rc3 * rc3 * rc3;
        h->e_disp = 8.0 * h->n * (double)h->n * M_PI * (s12 / tot / (9.0 * rc9) - s6 / tot / (3.0 * rc3)) / V;
    }
    h->pme_static_valid = false;
    return 0;
}

template <typename T> static PmeArgs<T> make_pme_args(BluesEngine* h, bool is_static) {
    PmeArgs<T> P; memset(&P, 0, sizeof P);
    if (!h->pme) return P;
    auto& b = pme_bufs<T>(h);
    P.n = h->n; P.order = h->pme_order; for (int d = 0; d < 3; d++) P.K[d] = h->pme_K[d];
    P.ng = h->pme_K[0] * h->pme_K[1] * h->pme_K[2];
    P.n_sel = is_static ? (int)h->pme_frozen.size() : (int)h->mobile.size();
    P.sel = is_static ? h->d_pme_frozen.p : h->d_mobile_atoms.p;
    for (int k = 0; k < 3; k++) { P.x[k] = h->d_x[k].p; P.tw_cos[k] = b.tw_cos[k].p; P.tw_sin[k] = b.tw_sin[k].p; }
    P.qn = h->d_qn.p; P.qn_full = h->d_qn_full.p; P.box = make_box(h); P.acc = h->d_pme_acc.p;
    P.a_re = b.a_re.p; P.a_im = b.a_im.p; P.b_re = b.b_re.p; P.b_im = b.b_im.p; P.eterm = b.eterm.p;
    P.qf_re = b.qf_re.p; P.qf_im = b.qf_im.p; P.phi_f = b.phi_f.p; P.have_static = !h->pme_frozen.empty();
    P.frec = h->d_frec.p; P.epart = h->d_pme_e.p; P.want_energy = 0;
    return P;
}

// reciprocal-space forces on the mobile atoms (and the mesh energy when asked); the frozen charges' meshes first if they are stale
template <typename T> static void pme_ensure_static(BluesEngine* h) {
    if (h->pme_static_valid) return;
    if (!h->pme_frozen.empty()) { hipLaunchKernelGGL((k_pme<T, true>), dim3(1), dim3(PME_THREADS), 0, h->cur, make_pme_args<T>(h, true)); h->st_launches++; }
    h->pme_static_valid = true;
}
template <typename T> static int launch_pme_t(BluesEngine* h, int want_energy) {
    if (batch_dry(h)) return 0;
    // mixed precision with a mesh whose half spectrum fits two LDS buffers: the pruned in-LDS pipeline (k_pme_fast); otherwise
    // (double precision, large meshes) the general one on global memory
    bool fast = false; size_t lds = 0;
    if constexpr (sizeof(T) == 4) {
        fast = (size_t)h->pme_K[0] * h->pme_K[1] * (h->pme_K[2] / 2 + 1) <= PME_LDS_Y && !h->tune.pme_general;
        lds = PME_FAST_LDS(h->pme_K[0], h->pme_K[1], h->pme_K[2]);
        if (fast) {
            static thread_local size_t lds_set[2] = {0, 0};
            const bool lead = batch_lead(h);
            if (lds > lds_set[lead]) {
                hipError_t e = lead ? hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pme_fast_b), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                    : hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pme_fast), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) E_FAIL(h, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", lds, hipGetErrorString(e));
                lds_set[lead] = lds;
            }
        }
    }
    if (batch_lead(h)) {
        // the shared launch reads every member's static meshes: whoever's are stale gets them first (lone launches, same stream)
        for (BluesEngine* m : h->batch->eng) if (m->pme) pme_ensure_static<T>(m);
        if constexpr (sizeof(T) == 4) {
            if (fast) hipLaunchKernelGGL(k_pme_fast_b, dim3(h->batch->R()), dim3(PME_THREADS), lds, h->cur, h->batch->d_nb_f.p, want_energy);
            else hipLaunchKernelGGL(k_pme_b<T>, dim3(h->batch->R()), dim3(PME_THREADS), 0, h->cur, batch_reps_nb<T>(h->batch), want_energy);
        } else hipLaunchKernelGGL(k_pme_b<T>, dim3(h->batch->R()), dim3(PME_THREADS), 0, h->cur, batch_reps_nb<T>(h->batch), want_energy);
    } else {
        pme_ensure_static<T>(h);
        PmeArgs<T> P = make_pme_args<T>(h, false); P.want_energy = want_energy & 1; if (want_energy & 2) P.qn = P.qn_full;
        if constexpr (sizeof(T) == 4) {
            if (fast) hipLaunchKernelGGL(k_pme_fast, dim3(1), dim3(PME_THREADS), lds, h->cur, P);
            else hipLaunchKernelGGL((k_pme<T, false>), dim3(1), dim3(PME_THREADS), 0, h->cur, P);
        } else hipLaunchKernelGGL((k_pme<T, false>), dim3(1), dim3(PME_THREADS), 0, h->cur, P);
    }
    h->st_launches++;
    HIP_OK(h, hipGetLastError());
    return 0;
}
static int launch_pme(BluesEngine* h, int want_energy) {
    if (!h->pme) return 0;
    return h->precision == 0 ? launch_pme_t<float>(h, want_energy) : launch_pme_t<double>(h, want_energy);
}

static int download_xyz(BluesEngine* h, double* xyz, DBuf<double>* src);
static int sort_and_tile(BluesEngine* h);
static int batch_plan_shape(BluesBatch* B, bool fresh);
static BluesEngine* relayout_many(const std::vector<BluesEngine*>& members, const std::function<bool(BluesEngine*)>& prepare);
static int resolve_xfer(BluesEngine* h) {
    if (!h->xfer_pending) return 0;
    h->xfer_pending = false;
    HIP_OK(h, hipStreamSynchronize(h->xfer_stream));
    const unsigned* out = h->xfer_src ? h->xfer_src : h->h_xfer;
    if (out[0]) { h->e_frozen_valid = false; h->pme_static_valid = false; }
    float worst; memcpy(&worst, &out[1], sizeof worst);
    // tiles are formed from the mobile non-alchemical atoms.  A few wandering i-atoms only stretch their tile's bounding
    // box, and the device notices when that starts to cost (resort_hint); the host re-sorts when an i-atom is far out
    // or a sizeable part of the system has moved
    // (per-atom lists over group images: the groups are tiles of mobile atoms that were neighbours at the sort, and their lists and
    // LDS images are sized for that -- a State from the MD leg, reference blues/simulation.py:1028-1037, arrives with those atoms a few
    // tenths of a nm apart from where they were: re-tile before the lists outgrow their capacity)
    // (fragment lists: the order only decides how many blocks the list builder can skip and how local the force kernel's gathers are;
    // the builder itself asks for a new order when its blocks have spread (resort_hint) -- a State that arrives from another leg
    // every iteration must not cost a host-side re-sort every time)
    // ... only for such a State: a chain's OWN States (the Move's edit at lambda = 0.5, the restore after a rejection) come back every
    // iteration with the atoms a few tenths of a nm from the sort -- re-sorting for those cost 2 x 5 ms per chain and iteration
    const float far = (h->k1_mode == 2 && h->xfer_foreign) ? 0.09f : 1.0f;   // (squared displacement: 0.3 nm / 1 nm)
    h->xfer_foreign = false;
    if (h->sorted_ok && h->k1_mode != 3 && (worst > far || (int)out[2] > h->n / 10)) h->sorted_ok = false;   // (the layout that follows fetches the positions itself: sort_and_tile)
    return 0;
}

// A member whose lists have outgrown the batch's layout shape.  Round 5 re-planned the shape for EVERYBODY (a re-layout of every member:
// seconds for a batch of 1024, the other batches of the device waiting) -- and in the benchmark geometry, one compact group of mobile
// atoms, no other shape holds less, so the whole batch ended on fragment lists for good (7 % slower) because ONE chain's move had thrown
// a few waters outward for the rest of ONE switch (bench.py: the same chain at the same iteration in every run -- its rotation is drawn from
// its own stream).  Round 6: such a member becomes a STRAGGLER: it lays itself out on its own (fragment lists), leaves the shared launches
// (its record is inactive) and is stepped by launches of its own behind them -- ~60 us per step for that one chain, nothing for the others --
// until its atoms fit the batch's shape again (tried at the start of its next switch: a State that arrives from the MD leg or the restore
// after a rejection brings the compact arrangement back).  Only when more than a handful of members straggle does the batch re-plan.
static int relayout(BluesEngine* h);
static bool batch_can_straggle(const BluesBatch* B, const BluesEngine* h) {
    if (h->straggler || h->k1_mode != 2 && h->shape_S == 0) return false;
    int n = 0; for (const BluesEngine* m : B->eng) n += m->straggler;
    return n + 1 <= std::max(1, B->R() / 32) && B->R() > 1;
}
static int make_straggler(BluesBatch* B, BluesEngine* h) {
    h->straggler = true; h->forbid_atom = true; h->shape_S = 0; h->shape_jcap = 0; h->shape_overflow = false;
    B->st_stragglers++;
    if (h->tune.debug_lists) fprintf(stderr, "[straggler] member %d leaves the shared launches at step %d\n", h->batch_index, h->h_step);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = relayout(h);
    B->st_straggle_sec += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}
// ... and back: the batch's shape, if the member's atoms fit it now (its first step of a switch)
static int try_rejoin(BluesBatch* B, BluesEngine* h, const BluesEngine* like) {
    h->forbid_atom = false; h->shape_S = like->S; h->shape_jcap = like->jcap;
    {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = relayout(h);
        B->st_straggle_sec += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rc) return 1;
    }
    if (h->sorted_ok && h->k1_mode == 2 && !h->shape_overflow && h->S == like->S && h->jcap == like->jcap) { h->straggler = false; B->st_rejoined++; return 0; }
    h->straggler = false;   // (make_straggler counts it again: st_stragglers is "times a member left", st_rejoined "times one came back")
    B->st_stragglers--;
    return make_straggler(B, h);
}

static int ensure_sorted(BluesEngine* h) {
    if (resolve_xfer(h)) return 1;
    if (!h->have_positions) E_FAIL(h, "positions have not been set");
    if (!h->sorted_ok && sort_and_tile(h)) return 1;
    // a member of a batch whose re-sorted tiles no longer fit the batch's layout shape (its mobile atoms have spread: a State from
    // a long MD leg): a new shape for everybody NOW -- the next list build would overflow (the stepping loop's own check comes
    // every RESORT_POLL steps, too late for a hand-over)
    if (h->shape_overflow && h->batch && !h->batch->lockstep && !h->batch->replanning) {
        BluesBatch* B = h->batch;
        if (batch_can_straggle(B, h)) return make_straggler(B, h);
        B->replanning = true;
        hipStreamSynchronize(B->stream);
        const int rc = batch_plan_shape(B, false);
        B->replanning = false; B->st_replans++;
        if (rc) E_FAIL(h, "%s", B->err.c_str());
    }
    return 0;
}

// side streams and their fork / join events (a batch's leader with the lane-layout alchemical kernel, graph capture); timing events
static int ensure_side(BluesEngine* h) {
    if (h->s1) return 0;
    HIP_OK(h, hipStreamCreate(&h->s1));
    {   // s2 carries a batch's rebuild kernels (force_pass): a few workgroups on the critical path that must find room on CUs the
        // alchemical kernel has filled -- at equal priority the dispatcher keeps refilling those CUs with that kernel's workgroups
        int least = 0, greatest = 0;
        HIP_OK(h, hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_OK(h, hipStreamCreateWithPriority(&h->s2, hipStreamDefault, greatest));
    }
    if (!h->evFork) HIP_OK(h, hipEventCreateWithFlags(&h->evFork, hipEventDisableTiming));
    HIP_OK(h, hipEventCreateWithFlags(&h->evJ1, hipEventDisableTiming)); HIP_OK(h, hipEventCreateWithFlags(&h->evJ2, hipEventDisableTiming));
    HIP_OK(h, hipEventCreateWithFlags(&h->evA, hipEventDisableTiming)); HIP_OK(h, hipEventCreateWithFlags(&h->evB, hipEventDisableTiming));
    return 0;
}
static int ensure_timing(BluesEngine* h) {
    if (h->ev0) return 0;
    HIP_OK(h, hipEventCreate(&h->ev0)); HIP_OK(h, hipEventCreate(&h->ev1));
    return 0;
}

// one force pass at the current positions: lists (if stale) -> alchemical -> nonbonded -> bonded
static int force_pass(BluesEngine* h, int base_L) {
    if (ensure_sorted(h)) return 1;
    double ls[3], le[3];
    for (int s = 0; s < 3; s++) { const int L = std::min(base_L + s, h->n_lambda); ls[s] = h->tab_ls[L]; le[s] = h->tab_le[L]; }
    // which lambda slots will have their FORCE applied: "H V R O R V H" kicks only at odd lambda indices (after the
    // first H of a step, before the second), so a pass based at an odd index serves V(slot 0) and V(slot 2), one based
    // at an even index (first step, or the step after a Move) only V(slot 1).  Energies are always formed for all three.
    int fmask = 7;
    if (h->split == "HVRORVH" && h->nprop == 1) fmask = (base_L & 1) ? 5 : 2;
    if (h->split == "L") fmask = 1;   // (the MD leg's one kick per step reads slot 0; no lambda, no other slot)
    if (h->tune.slot_mask >= 0) fmask = h->tune.slot_mask & 7;
    h->pass_fmask = fmask;
    // A large batch in per-atom-list mode forks the alchemical kernel onto a side stream after the group lists (which hold its
    // j records): the atoms' own lists are built by a few latency-bound blocks that leave most of the chip idle, and the
    // alchemical kernel fills it.  Joined before finalize.
    const int fork_mode = h->batch ? h->batch->tune.fork : h->tune.fork;   // 2: also with the dense alchemical kernel, joined BEFORE the nonbonded kernel
    const bool fork_env = fork_mode != 0;
    const bool decomposed = !(h->fuse_forces && h->wpb == 4) && !(h->k1_mode == 1 && h->precision == 0 && h->fuse_big);
    // (round 3 kept the dense alchemical kernel off the side stream: a 512-thread workgroup with ~137 KB of LDS per chain cannot share a
    // CU with the nonbonded kernel's, and beside the rebuild's small workgroups it ran at half speed: 516 us per step with every kernel
    // alone against 540)
    const bool fork = fork_env && decomposed && batch_lead(h) && h->k1_mode == 2 && !h->alch.empty() && !h->ctrl_arg;
    // with the dense alchemical kernel only the two SMALL kernels of a pass go to the side stream (alchemical x alchemical block,
    // bonded entries + the next step's noise: 54 us per 1024 chains, 168 / 70 registers, 37 KB / no LDS) -- they run beside the
    // builder of the atoms' lists (85 us, no LDS, 94 registers) and are done before it: 756 -> 716 us per step of 1024 chains.
    // The dense kernel itself beside that builder (fork = 2) gains nothing; beside the GROUP-list builder the small kernels cost
    // it what they saved (round 4, first half).
    // 4 (round 6): the work list and the group lists are two latency-bound kernels of a few hundred workgroups -- 68 us during which the
    // chip idles.  The kernels that need nothing from them start WITH them on the side streams: the two small ones at once, the dense
    // alchemical kernel of the members that do not rebuild as soon as the work list says who they are (all but ~40 of 1024), that of the
    // members that rebuild behind the group lists; all joined before the nonbonded kernel, which keeps the device to itself.
    const bool beside_lists = fork && fork_mode == 4 && h->k2_dense && h->k2_f32;
    const bool small_side = fork && h->k2_dense && fork_mode != 2 && fork_mode != 3 && !beside_lists;   // (3: every alchemical kernel and the bonded entries on the side stream, joined before the sums)
    if (fork && ensure_side(h)) return 1;
    // k2_early (off by default): the alchemical kernel of the members that do NOT rebuild needs nothing from the rebuild and can
    // start as soon as the work list says who they are, with the rebuild kernels on a high-priority stream beside it and the
    // members that rebuild following their group lists.  Measured at R = 512 (round 3): 545-565 us per step against 545-555
    // without -- the step is bound by the sum of the kernels' work, not by their order ("what the timeline says").
    // In the default order the atoms' lists are enqueued BEFORE the alchemical kernel: behind its 12,800 small workgroups the
    // few workgroups of the rebuild waited for room on full CUs (245 us against 107 us alone).
    const bool early = fork && h->s2 && !h->k2_dense && !h->lists_forced && !h->tune.force_lists && (h->batch ? h->batch->tune.k2_early : h->tune.k2_early) != 0;
    int rc = 0;
    bool wait_lists = false;
    // fragment lists (k1_mode 3) in a batch: the alchemical kernel and the bonded entries (+ the next O step's noise) need the
    // alchemical tile's list only -- they run on the side stream beside the fragment-list kernels (a rebuild or a prune of one or two
    // members: a few latency-bound workgroups) and the head of the nonbonded kernel, joined before the sums
    // (a lone chain of a few thousand fragments too: its step is eight launches back to back, and the alchemical tile's list, the
    // alchemical kernel and the bonded entries -- a third of them -- need nothing from the fragment lists)
    const bool fork3 = fork_env && h->k1_mode == 3 && !h->ctrl_arg && (batch_lead(h) || (!batch_dry(h) && h->frag_NI >= 2048));
    if (fork3) {
        if (ensure_side(h)) return 1;
        hipStream_t main_stream = h->cur;
        HIP_OK(h, hipEventRecord(h->evFork, main_stream)); HIP_OK(h, hipStreamWaitEvent(h->s1, h->evFork, 0));
        // (the alchemical tile's list -- work list + builder, 6 + 22 us at R = 16 -- is read by the side stream's kernels only; the list
        // kernels read the flags, nobody writes them before the nonbonded kernel: it goes to the side stream with its readers)
        // (both of its kernels read the flags the nonbonded kernel resets: that kernel waits for evA)
        h->cur = h->s1; rc = launch_lists<float>(h, h->lists_forced, 1); h->cur = main_stream;
        if (rc) return 1;
        HIP_OK(h, hipEventRecord(h->evA, h->s1));
        // (the DENSE alchemical kernel -- a 512-thread workgroup with ~158 KB of LDS per chain -- stays on the main stream, behind the
        // nonbonded kernel: beside it the two ran at a third of their speed (471 against 166 us per 1024 chains); its alchemical x
        // alchemical block and the bonded entries are the small kernels that go to the side)
        h->cur = h->s1; rc = launch_alchemical(h, ls, le, fmask, h->k2_dense ? 3 : 0) || launch_bonded(h, true); h->cur = main_stream;
        if (rc) return 1;
        HIP_OK(h, hipEventRecord(h->evJ1, h->s1));
        // (tried and dropped in round 5: the nonbonded kernel in two launches -- the members whose lists are current on the main stream
        // beside the rebuild / prune of the others on a second side stream, those members' share behind it: 707 against 552 us per
        // step at R = 16, 2,355 against 2,020 at R = 64; the list kernels' workgroups slow the force kernel's more than they hide)
        rc = launch_lists<float>(h, h->lists_forced, 5);
        if (rc) return 1;
        h->lists_forced = false;
        HIP_OK(h, hipStreamWaitEvent(main_stream, h->evA, 0));
        if (launch_nonbonded<float, false>(h)) return 1;
        if (h->k2_dense && launch_alchemical(h, ls, le, fmask, 4)) return 1;
        if (launch_pme(h, 0)) return 1;
        HIP_OK(h, hipStreamWaitEvent(h->cur, h->evJ1, 0));
        if (launch_finalize_deferred(h, le, fmask)) return 1;
        h->pass_valid = true; h->pass_L = base_L; h->st_passes++; h->vel_clean = true; h->acc_cache_valid = false;
        HIP_OK(h, hipGetLastError());
        return 0;
    }
    if (beside_lists) {
        hipStream_t main_stream = h->cur;
#define LISTS(ph) (h->precision == 0 ? launch_lists<float>(h, h->lists_forced, ph) : launch_lists<double>(h, h->lists_forced, ph))
        HIP_OK(h, hipEventRecord(h->evFork, main_stream)); HIP_OK(h, hipStreamWaitEvent(h->s1, h->evFork, 0));
        h->cur = h->s1; rc = launch_alchemical(h, ls, le, fmask, 3) || launch_bonded(h, true); h->cur = main_stream;
        if (rc) return 1;
        if (LISTS(3)) return 1;
        HIP_OK(h, hipEventRecord(h->evA, main_stream)); HIP_OK(h, hipStreamWaitEvent(h->s1, h->evA, 0));
        h->cur = h->s1; rc = launch_alchemical(h, ls, le, fmask, 5); h->cur = main_stream;
        if (rc) return 1;
        HIP_OK(h, hipEventRecord(h->evJ1, h->s1));
        if (LISTS(4)) return 1;
        HIP_OK(h, hipEventRecord(h->evJ2, main_stream)); HIP_OK(h, hipStreamWaitEvent(h->s2, h->evJ2, 0));
        h->cur = h->s2; rc = launch_alchemical(h, ls, le, fmask, 6); h->cur = main_stream;
        if (rc) return 1;
        HIP_OK(h, hipEventRecord(h->evB, h->s2));
        if (LISTS(2)) return 1;
#undef LISTS
        HIP_OK(h, hipStreamWaitEvent(main_stream, h->evJ1, 0)); HIP_OK(h, hipStreamWaitEvent(main_stream, h->evB, 0));
    } else if (fork) {
        hipStream_t main_stream = h->cur;
        hipStream_t rb = early ? h->s2 : main_stream;   // the rebuild's stream (high priority: see engine creation)
#define LISTS(ph) (h->precision == 0 ? launch_lists<float>(h, h->lists_forced, ph) : launch_lists<double>(h, h->lists_forced, ph))
        if (early) { HIP_OK(h, hipEventRecord(h->evFork, main_stream)); HIP_OK(h, hipStreamWaitEvent(rb, h->evFork, 0)); HIP_OK(h, hipStreamWaitEvent(h->s1, h->evFork, 0)); }
        h->cur = rb; rc = LISTS(3); h->cur = main_stream;
        if (rc) return 1;
        if (early) {
            HIP_OK(h, hipEventRecord(h->evA, rb)); HIP_OK(h, hipStreamWaitEvent(h->s1, h->evA, 0));
            h->cur = h->s1; rc = launch_alchemical(h, ls, le, fmask, 1); h->cur = main_stream;
            if (rc) return 1;
        }
        h->cur = rb; rc = LISTS(4); h->cur = main_stream;
        if (rc) return 1;
        HIP_OK(h, hipEventRecord(h->evJ2, rb)); HIP_OK(h, hipStreamWaitEvent(h->s1, h->evJ2, 0));
        h->cur = rb; rc = LISTS(2); h->cur = main_stream;
        if (rc) return 1;
        if (early) { HIP_OK(h, hipEventRecord(h->evB, rb)); wait_lists = true; }
#undef LISTS
        h->cur = h->s1; rc = launch_alchemical(h, ls, le, fmask, small_side ? 3 : (early ? 2 : 0)) || launch_bonded(h, true); h->cur = main_stream;   // (bonded terms and the next O step's noise need no list either)
        if (rc) return 1;
        HIP_OK(h, hipEventRecord(h->evJ1, h->s1));
        if (small_side && launch_alchemical(h, ls, le, fmask, 4)) return 1;   // (the dense kernel follows the atoms' lists on the main stream)
        if (wait_lists) HIP_OK(h, hipStreamWaitEvent(main_stream, h->evB, 0));
    } else {
        rc = h->precision == 0 ? launch_lists<float>(h, h->lists_forced, 0) : launch_lists<double>(h, h->lists_forced, 0);
        if (rc) return 1;
    }
    h->lists_forced = false;
    if (h->fuse_forces && h->wpb == 4) {
        rc = h->precision == 0 ? launch_forces_fused<float>(h, ls, le) : launch_forces_fused<double>(h, ls, le);
        if (rc) return 1;
        if (launch_pme(h, 0)) return 1;
        if (launch_finalize_deferred(h, le, fmask)) return 1;
    } else if (h->k1_mode == 1 && h->precision == 0 && h->fuse_big) {
        if (launch_forces_fused_sub(h, ls, le)) return 1;
        if (launch_pme(h, 0)) return 1;
        if (launch_finalize(h, le, fmask)) return 1;
    } else {
        // (the alchemical x alchemical block stays a launch of its own on the main stream: in the bonded entries' grid it took 49 us against
        // 18 + 13; on a side stream beside the list rebuild, with the bonded entries, the step stayed at 419 us -- the two small kernels
        // overlapped, the group-list builder they ran beside went from 47 to 62 us)
        if (!fork && launch_alchemical(h, ls, le, fmask)) return 1;
        if (fork && fork_mode == 2) HIP_OK(h, hipStreamWaitEvent(h->cur, h->evJ1, 0));   // (the nonbonded kernel has the device to itself)
        rc = h->precision == 0 ? launch_nonbonded<float, false>(h) : launch_nonbonded<double, false>(h);
        if (rc) return 1;
        if (launch_pme(h, 0)) return 1;
        if (fork) { HIP_OK(h, hipStreamWaitEvent(h->cur, h->evJ1, 0)); if (launch_finalize_deferred(h, le, fmask)) return 1; }
        else { if (launch_bonded(h, true)) return 1; if (launch_finalize_deferred(h, le, fmask)) return 1; }
    }
    h->pass_valid = true; h->pass_L = base_L; h->st_passes++; h->vel_clean = true; h->acc_cache_valid = false;
    HIP_OK(h, hipGetLastError());
    return 0;
}

static int check_flags(BluesEngine* h) {
    HIP_OK(h, hipStreamSynchronize(h->stream));
    DevFlags f;
    HIP_OK(h, hipMemcpy(&f, h->d_flags.p, sizeof f, hipMemcpyDeviceToHost));
    if (f.nan_flag || f.list_overflow || f.constraint_fail) {
        // report once and stay usable: BLUES catches the exception, lets the move clean up and carries on with
        // the next iteration (reference blues/simulation.py:1088-1094), exactly as after an OpenMMException
        DevFlags z = f; z.nan_flag = 0; z.list_overflow = 0; z.constraint_fail = 0;
        hipMemcpy(h->d_flags.p, &z, sizeof z, hipMemcpyHostToDevice);
        h->pass_valid = false; h->lists_forced = true;
    }
    if (f.nan_flag) E_FAIL(h, "Particle coordinate is nan");
    if (f.list_overflow && h->tune.debug_lists) {
        std::vector<int> jc, ac; h->d_jcount.download(jc); if (h->d_acount.p) h->d_acount.download(ac);
        int mj = 0, ma = 0; for (int c : jc) mj = std::max(mj, c); for (int c : ac) ma = std::max(ma, c);
        fprintf(stderr, "[overflow] member %d mode %d S %d n_lists %d n_itiles %d jcap %d max jcount %d (lists %zu) acap %d max acount %d skin %.3f skin_m %.3f trig %.3f\n",
                h->batch_index, h->k1_mode, h->S, h->n_lists, h->n_itiles, h->jcap, mj, jc.size(), h->acap, ma, h->skin, h->skin_m, h->trig);
    }
    if (f.list_overflow && h->k1_mode == 3 && (f.list_overflow & ~1)) {
        int mo = -1, mi = -1;
        try { std::vector<int> oc, ic; h->d_ocount.download(oc); h->d_icount.download(ic); mo = mi = 0;
              for (int q = 0; q < h->frag_NI && q < (int)oc.size(); q++) mo = std::max(mo, oc[q]);
              for (int q = 0; q < h->frag_NI && q < (int)ic.size(); q++) mi = std::max(mi, ic[q]); } catch (std::string&) {}
        E_FAIL(h, "neighbour list capacity exceeded (fragment lists:%s%s%s%s; longest outer row %d of %d, inner %d of %d; %d fragments)",
               (f.list_overflow & FR_OVER_OUTER) ? " outer rows" : "", (f.list_overflow & FR_OVER_INNER) ? " inner rows" : "",
               (f.list_overflow & FR_OVER_QUEUE) ? " the builder's candidate queue" : "", (f.list_overflow & FR_OVER_REACH) ? " a fragment's reach against the box" : "",
               mo, h->frag_ocap, mi, h->frag_icap, h->frag_F);
    }
    if (f.list_overflow) E_FAIL(h, "neighbour list capacity exceeded (jcap=%d)", h->jcap);
    if (f.constraint_fail) E_FAIL(h, "constraint solver did not converge (the step is unstable)");
    return 0;
}

// The tiles are formed on the host from a spatial sort; as atoms diffuse they spread and their j-lists grow.  The list
// kernel raises resort_hint when a list nears its capacity; stepping polls it every RESORT_POLL steps (one small
// read-back) and re-sorts from the positions on the device.  Without this a long all-mobile run ends in list_overflow.
#define RESORT_POLL 64
static int relayout(BluesEngine* h);
static int batch_plan_shape(BluesBatch* B, bool fresh);
// Fragment lists: the builder raises resort_hint when a row has reached 90 % of its capacity (kernels_frag.h: finish) as well as when its
// blocks have spread.  Before the re-layout that follows, the rows' lengths say whether the lists want more room than the density estimate
// gave them (a solute-rich region, a denser phase): then the next layout gets a quarter more -- before an entry is ever dropped.
static void frag_grow_caps(BluesEngine* h) {
    if (h->k1_mode != 3 || h->frag_NI <= 0) return;
    std::vector<int> oc, ic;
    try { h->d_ocount.download(oc); h->d_icount.download(ic); } catch (std::string&) { return; }
    int mo = 0, mi = 0;
    for (int q = 0; q < h->frag_NI && q < (int)oc.size(); q++) mo = std::max(mo, oc[q]);
    for (int q = 0; q < h->frag_NI && q < (int)ic.size(); q++) mi = std::max(mi, ic[q]);
    if (h->frag_ocap <= h->frag_F && (100 * mo > 85 * h->frag_ocap || mo > h->frag_ocap - 64)) h->frag_grow_o *= 1.25;   // (a row of F + 1 places holds every fragment)
    if (h->frag_icap <= h->frag_F && (100 * mi > 85 * h->frag_icap || mi > h->frag_icap - 64)) h->frag_grow_i *= 1.25;
    if (h->tune.debug_lists) fprintf(stderr, "[frag caps] member %d: longest outer row %d of %d, inner %d of %d; growth %.2f / %.2f\n", h->batch_index, mo, h->frag_ocap, mi, h->frag_icap, h->frag_grow_o, h->frag_grow_i);
}
// poll_resort_due: 1 = the engine's builder asked for a new order (flag cleared, counted) and nothing is wrong with it, 0 = no, -1 = error
static int poll_resort_due(BluesEngine* h) {
    if (flush_program(h)) return -1;
    if (hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "stream synchronisation failed"; return -1; }
    DevFlags f;
    if (hipMemcpy(&f, h->d_flags.p, sizeof f, hipMemcpyDeviceToHost) != hipSuccess) { h->err = "flag read-back failed"; return -1; }
    if (!f.resort_hint || f.list_overflow || f.nan_flag || f.constraint_fail) return 0;   // errors are reported by check_flags
    if (h->tune.debug_lists) { std::vector<int> jc; h->d_jcount.download(jc); fprintf(stderr, "[resort] member %d step %d jcount[0]=%d hint_count %d jcap %d\n", h->batch_index, h->h_step, jc.empty() ? -1 : jc[0], h->hint_count, h->jcap); }
    f.resort_hint = 0;
    if (hipMemcpy(h->d_flags.p, &f, sizeof f, hipMemcpyHostToDevice) != hipSuccess) { h->err = "flag write-back failed"; return -1; }
    h->st_resorts++;
    frag_grow_caps(h);
    return 1;
}
// Fragment lists of a system in which (nearly) everything moves: the order ages -- the builder's blocks spread as the molecules diffuse,
// its kernels slow down by 1 % per thousand steps, and when the builder finally asks (resort_hint) every member of a batch asks within a few
// hundred steps of the others, one or two per poll, each re-sort 10 ms of host work with the batch waiting: 0.2 s per 16 chains every
// ~6,000 steps (scripts/dev_order_age.py).  A re-sort by AGE instead -- a function of the chain's own step count, so a member still does
// what the lone chain does -- comes due for all members of a batch at the same poll, where the host's cores share them.
#define RESORT_AGE 4096
static bool resort_by_age(const BluesEngine* h) { return h->k1_mode == 3 && h->sorted_ok && 2 * h->mobile.size() > (size_t)h->n && h->steps_since_sort >= RESORT_AGE; }
static int poll_resort(BluesEngine* h) {
    const bool aged = resort_by_age(h);
    const int due = poll_resort_due(h);   // (clears a pending hint and gives the lists more room where they ask: also when the age brings the re-sort anyway)
    if (due < 0) return 1;
    if (aged && !due) h->st_resorts++;
    return (aged || due) ? relayout(h) : 0;
}

// launch geometry of an energy evaluation (partials per kind)
struct EnergyShape { int nw, nbb, nfb; size_t off_frozen; };
static EnergyShape energy_shape(const BluesEngine* h) {
    EnergyShape g;
    const int subs = h->k1_mode == 2 ? 1 : (h->k1_iw != 64 ? 64 / h->k1_iw : 1);   // mode 2: one (LJ, Coulomb) pair per tile
    g.nw = (h->k1_mode == 2 ? h->n_lists : h->n_itiles) * h->npart * subs;
    g.off_frozen = (size_t)std::max(1, h->n_itiles) * h->npart * 2 * subs;
    if (h->k1_mode == 3) { g.nw = h->frag_nwg; g.off_frozen = (size_t)2 * h->frag_nwg; }   // one (LJ, Coulomb) pair per workgroup of the fragment kernel
    int total_terms = 0; for (int ty = 0; ty < T_NTYPES; ty++) total_terms += h->n_terms[ty];
    g.nbb = (total_terms + 255) / 256; g.nfb = (h->n + FROZEN_TILE - 1) / FROZEN_TILE;
    return g;
}

// the launches of an energy evaluation at the current state and alchemical parameters (lock-step capable: in a batch the
// leader's launches cover every member that takes part)
static int energy_launch(BluesEngine* h) {
    if (flush_program(h)) return 1;
    h->fin_pending = false;   // (the slabs of the last pass are about to be overwritten: nothing left to sum)
    if (ensure_sorted(h)) return 1;
    // slot 1 of the alchemical kernel carries lambda_sterics = lambda_electrostatics = 1: the energy at the unmodified potential is
    // what _computeAlchemicalCorrection and _syncStatesMDtoNCMC ask for next (reference blues/simulation.py:1100-1119),
    // and only the alchemical terms differ -- it comes out of the same pass (energy_sum) instead of a second full evaluation
    double ls[3] = {h->cur_ls, 1.0, h->cur_ls}, le[3] = {h->cur_le, 1.0, h->cur_le};
    int rc = h->precision == 0 ? launch_lists<float>(h, h->lists_forced) : launch_lists<double>(h, h->lists_forced);
    h->lists_forced = false;
    if (rc) return 1;
    h->pass_valid = false;  // slabs are about to be overwritten with parameters that are not a regular pass
    if (launch_alchemical(h, ls, le, 1)) return 1;
    rc = h->precision == 0 ? launch_nonbonded<float, true>(h) : launch_nonbonded<double, true>(h);
    if (rc) return 1;
    if (launch_pme(h, 1)) return 1;
    const EnergyShape g = energy_shape(h);
    if (g.nbb > 0) {
        if (batch_lead(h)) hipLaunchKernelGGL(k_bonded_energy_b, dim3(g.nbb, h->batch->R()), dim3(256), 0, h->cur, h->batch->d_core.p);
        else if (!batch_dry(h)) hipLaunchKernelGGL(k_bonded_energy, dim3(g.nbb), dim3(256), 0, h->cur, make_bonded_args(h));
        h->st_launches++;
    }
    HIP_OK(h, hipGetLastError());
    return 0;
}

// the host-side sums over the partials, in a fixed order (the same whether they were downloaded one engine at a time or
// gathered for a whole batch)
// one[2] (optional): the two alchemical terms at lambda_sterics = lambda_electrostatics = 1 (slot 1 of the same pass)
static void energy_sum(BluesEngine* h, const double* enb, const double* eb, int jcount_alch, const double* ep, double e_mesh, double T[BLUES_N_ENERGY_TERMS], double* one = nullptr) {
    const EnergyShape g = energy_shape(h);
    for (int t = 0; t < BLUES_N_ENERGY_TERMS; t++) T[t] = 0.0;
    double e_nb = 0.0; for (int w = 0; w < g.nw; w++) e_nb += enb[2 * w] + enb[2 * w + 1];
    T[3] = e_nb + h->e_frozen[0] + h->e_frozen[1];
    for (int b = 0; b < g.nbb; b++) { T[0] += eb[b * T_NTYPES + T_BOND]; T[1] += eb[b * T_NTYPES + T_ANGLE]; T[2] += eb[b * T_NTYPES + T_TORSION]; T[4] += eb[b * T_NTYPES + T_EXC]; T[7] += eb[b * T_NTYPES + T_RESTR]; T[8] += eb[b * T_NTYPES + T_EWEX]; }
    if (h->pme) { T[8] += e_mesh + h->e_ewald_const; T[9] = h->e_disp; }
    if (!h->alch.empty()) {
        const int nb_env = k2_env_blocks(jcount_alch, h->PA, h->k2_jiter);
        double s[K2_NE] = {0, 0, 0, 0, 0, 0};
        for (int b = 0; b <= h->k2_nblocks_env; b++) { if (b >= nb_env && b != h->k2_nblocks_env) continue; for (int q = 0; q < K2_NE; q++) s[q] += ep[(size_t)b * K2_NP + q]; }
        T[5] = s[1] + s[4]; T[6] = h->cur_le * s[0] + s[5];
        if (one) { one[0] = s[2] + s[4]; one[1] = 1.0 * s[0] + s[5]; }
    } else if (one) { one[0] = 0.0; one[1] = 0.0; }
}

// total at the current parameters and, from the same terms, at lambda = (1, 1): summed in term order, as a direct evaluation at
// those parameters would (so the cached value is the one that evaluation returns)
static void energy_totals(BluesEngine* h, const double T[BLUES_N_ENERGY_TERMS], const double one[2], double* E, double* E_one) {
    *E = 0.0; *E_one = 0.0;
    for (int t = 0; t < BLUES_N_ENERGY_TERMS; t++) { *E += T[t]; *E_one += (t == 5) ? one[0] : (t == 6 ? one[1] : T[t]); }
    (void)h;
}

// full potential energy breakdown at the current state and alchemical parameters (on demand; synchronises)
static int energy_terms(BluesEngine* h, double T[BLUES_N_ENERGY_TERMS]) {
    h->st_energy_evals++;
    if (energy_launch(h)) return 1;
    const EnergyShape g = energy_shape(h);
    if (!h->e_frozen_valid) {
        double* ep = h->d_epart_nb.p + g.off_frozen;
        if (h->precision == 0) {
            hipLaunchKernelGGL(k_frozen_boxes<float>, dim3((g.nfb + 3) / 4), dim3(256), 0, h->stream, h->n, make_nbconst<float>(h), h->d_img_f.p, ep);
            hipLaunchKernelGGL(k_energy_frozen<float>, dim3(g.nfb), dim3(256), 0, h->stream, h->n, make_nbconst<float>(h), h->d_img_f.p, h->d_ex_start.p, h->d_ex_idx.p, ep);
        } else {
            hipLaunchKernelGGL(k_frozen_boxes<double>, dim3((g.nfb + 3) / 4), dim3(256), 0, h->stream, h->n, make_nbconst<double>(h), h->d_img_d.p, ep);
            hipLaunchKernelGGL(k_energy_frozen<double>, dim3(g.nfb), dim3(256), 0, h->stream, h->n, make_nbconst<double>(h), h->d_img_d.p, h->d_ex_start.p, h->d_ex_idx.p, ep);
        }
        h->st_launches += 2;
    }
    if (check_flags(h)) return 1;
    std::vector<double> enb, eb, ep; std::vector<int> jc;
    try {
        h->d_epart_nb.download(enb);
        if (!h->e_frozen_valid) {
            h->e_frozen[0] = h->e_frozen[1] = 0.0;
            for (int b = 0; b < g.nfb; b++) { h->e_frozen[0] += enb[g.off_frozen + 2 * b]; h->e_frozen[1] += enb[g.off_frozen + 2 * b + 1]; }
            h->e_frozen_valid = true;
        }
        if (g.nbb > 0) h->d_epart_b.download(eb);
        if (!h->alch.empty()) { h->d_jcount.download(jc); h->d_e_part.download(ep); }
    } catch (std::string& msg) { E_FAIL(h, "%s", msg.c_str()); }
    double e_mesh = 0.0;
    if (h->pme) HIP_OK(h, hipMemcpy(&e_mesh, h->d_pme_e.p, sizeof e_mesh, hipMemcpyDeviceToHost));
    double one[2], E, E_one;
    energy_sum(h, enb.data(), eb.data(), h->alch.empty() ? 0 : jc[h->n_lists], ep.data(), e_mesh, T, one);
    energy_totals(h, T, one, &E, &E_one);
    h->ecache.put(1.0, 1.0, E_one); h->ecache.put(h->cur_ls, h->cur_le, E);   // (the current parameters last: they win when both are (1, 1))
    return 0;
}

// OpenMM caches the energy of a Context until positions or parameters change; BLUES leans on that (state0 / state1 /
// the alchemical-correction energies of one switch are evaluated at only two distinct states, simulation.py:1056-1119)
static int total_energy(BluesEngine* h, double* E) {
    if (flush_program(h)) return 1;
    if (h->ecache.find(h->cur_ls, h->cur_le, E)) return 0;
    double T[BLUES_N_ENERGY_TERMS];
    if (energy_terms(h, T)) return 1;
    *E = 0.0; for (int t = 0; t < BLUES_N_ENERGY_TERMS; t++) *E += T[t];
    h->ecache.put(h->cur_ls, h->cur_le, *E);
    return 0;
}

static int add_work(BluesEngine* h, double delta) {
    if (flush_program(h)) return 1;
    if (h->batch && h->batch->defer_work) {   // (a batch books the work of all its edited members with one launch: batch_do_steps)
        h->work_pending += delta; h->work_pending_set = true; h->acc_cache_valid = false;
        return 0;
    }
    hipLaunchKernelGGL(k_add_work, dim3(1), dim3(1), 0, h->stream, h->d_acc.p, delta);   // (in stream order: no read-modify-write through the host)
    h->st_launches++; h->acc_cache_valid = false;
    HIP_OK(h, hipGetLastError());
    return 0;
}

static int emit_cm(BluesEngine* h) {
    if (!h->remove_cm) return 0;
    if (h->fast_step && h->vel_clean && h->pass_valid && (h->pass_fmask & 1)) {  // the momentum partials are built from slot-0 forces
        const bool after_finish = h->prog.n == 3 && h->prog.ops[0] == OP_V0 && h->prog.ops[1] == OP_H01 && h->prog.ops[2] == OP_END && h->split == "HVRORVH" && h->nprop == 1 && h->h_step < h->nsteps - 0;
        const bool md_head = h->prog.n == 0 && h->split == "L" && !h->tracing;
        if (after_finish || md_head) return emit(h, OP_CM_PART);
    }
    if (h->int_blocks == 1) return emit(h, OP_CM_BLOCK);
    if (emit(h, OP_CM_REDUCE)) return 1;
    if (flush_program(h)) return 1;
    return emit(h, OP_CM_APPLY);
}

static int need_pass(BluesEngine* h, int lo, int hi) {  // need slots covering L in [lo, hi] of the current x
    if (h->pass_valid && lo - h->pass_L >= 0 && hi - h->pass_L <= 2) return 0;
    if (flush_program(h)) return 1;
    return force_pass(h, lo);
}

static int total_energy(BluesEngine* h, double* E);
static int kinetic_energy_now(BluesEngine* h, double* ke);
static int switching_open(BluesEngine* h);
static int switching_close(BluesEngine* h);
static int load_positions(BluesEngine* h, const double* const src[3], int stride, int n_edit);

static int run_fragment(BluesEngine* h, const std::string& frag) {
    if (h->remove_cm && !h->pass_valid && !frag.empty() && frag[0] == 'L') {
        if (need_pass(h, h->h_lambda_step, h->h_lambda_step)) return 1;  // forces do not depend on velocities: evaluate them first
    }
    if (h->switch_mode == BLUES_SWITCH_NONE && emit_cm(h)) return 1;
    for (size_t ci = 0; ci < frag.size(); ci++) {
        const char c = frag[ci];
        switch (c) {
        case 'V': {
            if (need_pass(h, h->h_lambda_step, h->h_lambda_step)) return 1;
            if (emit(h, OP_V0 + (h->h_lambda_step - h->pass_L))) return 1;
            if (h->sw_bracket_open && switching_close(h)) return 1;
        } break;
        case 'A': {   // first half of a velocity-Verlet step (switching modes only)
            if (h->switch_mode == BLUES_SWITCH_VV && emit_cm(h)) return 1;   // addUpdateContextState, switching.py:939
            if (switching_open(h)) return 1;
            if (need_pass(h, h->h_lambda_step, h->h_lambda_step)) return 1;
            if (emit(h, OP_A0 + (h->h_lambda_step - h->pass_L))) return 1;
            h->pass_valid = false;
        } break;
        case 'R': if (emit(h, OP_R)) return 1; h->pass_valid = false; break;
        case 'O':
            if (h->switch_mode == BLUES_SWITCH_GHMC && ci + 1 < frag.size() && frag[ci + 1] == 'A' && emit_cm(h)) return 1;   // addUpdateContextState ahead of the first randomisation, switching.py:976
            if (emit(h, OP_O)) return 1;
            break;
        case 'L': {
            if (need_pass(h, h->h_lambda_step, h->h_lambda_step)) return 1;
            if (h->h_lambda_step != h->pass_L) { if (flush_program(h)) return 1; if (force_pass(h, h->h_lambda_step)) return 1; }
            if (emit(h, OP_L)) return 1;
            h->pass_valid = false;
        } break;
        case 'H': {
            if (h->h_prop != 1) break;  // reference blues/integrators.py:217
            const int L = h->h_lambda_step;
            if (L + 1 > h->n_lambda) break;
            if (need_pass(h, L, L + 1)) return 1;
            if (emit(h, OP_H01 + (L - h->pass_L))) return 1;
            h->h_lambda_step = L + 1; h->h_lambda = (double)(L + 1) / h->n_lambda;
            h->cur_ls = h->tab_ls[L + 1]; h->cur_le = h->tab_le[L + 1];
        } break;
        default: break;
        }
    }
    return 0;
}

static int splitting_pass(BluesEngine* h) { return run_fragment(h, h->split); }

// ---- host-side energy bookkeeping of the switching integrators (reference blues/switching.py; dead code there, so this is the
// plain synchronous form: one total energy and one kinetic energy per velocity-Verlet step)
static int kinetic_energy_now(BluesEngine* h, double* kinetic) {
    if (flush_program(h)) return 1;
    if (h->ke_cache_valid) { *kinetic = h->ke_cache; return 0; }
    const int nb = (h->n + 255) / 256;
    hipLaunchKernelGGL(k_kinetic, dim3(nb), dim3(256), 0, h->stream, h->n, h->d_mass.p, h->d_v[0].p, h->d_v[1].p, h->d_v[2].p, h->d_scratch.p);
    h->st_launches++;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    std::vector<double> part(nb);
    HIP_OK(h, hipMemcpy(part.data(), h->d_scratch.p, sizeof(double) * nb, hipMemcpyDeviceToHost));
    double ke = 0.0; for (double p : part) ke += p;
    *kinetic = ke; h->ke_cache = ke; h->ke_cache_valid = true;
    return 0;
}

// before the first half of a velocity-Verlet step: Eold = energy + kinetic (switching.py:941-942), or for GHMC
// Eold = kinetic + Epert and the saved state (switching.py:986-989)
static int switching_open(BluesEngine* h) {
    if (h->switch_mode == BLUES_SWITCH_NONE) return 0;
    double ke = 0.0;
    if (h->switch_mode == BLUES_SWITCH_VV) {
        double E;
        if (total_energy(h, &E) || kinetic_energy_now(h, &ke)) return 1;
        h->sw_bracket_E0 = E + ke;
    } else {
        if (kinetic_energy_now(h, &ke)) return 1;
        h->sw_bracket_E0 = ke + h->sw_Epert;
        if (h->sw_saved) { blues_snapshot_release(h->sw_saved); h->sw_saved = nullptr; }
        if (blues_snapshot_capture(h, 3, &h->sw_saved)) return 1;
    }
    h->sw_bracket_open = true;
    return 0;
}

// after the second half: shadow work (switching.py:955-959), or the Metropolis test with momentum flip (switching.py:998-1005)
static int switching_close(BluesEngine* h) {
    h->sw_bracket_open = false;
    double E, ke;
    if (total_energy(h, &E) || kinetic_energy_now(h, &ke)) return 1;
    if (h->switch_mode == BLUES_SWITCH_VV) {
        h->sw_shadow += (E + ke) - h->sw_bracket_E0;
        h->sw_Epert = E;
        return 0;
    }
    uint32_t r[4];
    // keyed on a counter that only ever grows: keyed on ntrials, which reset() zeroes, every switch would have met the same
    // sequence of thresholds u_1, u_2, ... (correlated accept/reject decisions across BLUES iterations)
    philox4x32((uint32_t)h->sw_draw++, 0u, (uint32_t)h->replica * 4u + 2u, 0x47484D43u, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), r);
    const double u = ((double)r[0] + 0.5) * 2.3283064365386963e-10;
    h->sw_accept = (exp(-((ke + E) - h->sw_bracket_E0) / h->kT) - u >= 0.0) ? 1 : 0;   // Lepton step(x): 1 for x >= 0
    if (!h->sw_accept) {
        BluesSnapshot* sn = h->sw_saved;
        const double* src[3] = {sn->x[0], sn->x[1], sn->x[2]};
        if (load_positions(h, src, 1, 0)) return 1;
        h->x_edited = false;   // (not a Move: no external-perturbation work)
        hipLaunchKernelGGL(k_negated_copy3, dim3((h->n + 255) / 256), dim3(256), 0, h->stream, h->n, sn->v[0], sn->v[1], sn->v[2], h->d_v[0].p, h->d_v[1].p, h->d_v[2].p);
        h->st_launches++; h->vel_clean = false; h->ke_cache_valid = false;
        if (total_energy(h, &E)) return 1;   // the energy of the restored positions at the current parameters
    }
    h->sw_Epert = E;   // "Epert = energy" after the step (switching.py:1357; the second randomisation does not move atoms)
    h->sw_naccept += h->sw_accept; h->sw_ntrials++;
    return 0;
}

// ---- hipGraph path for the steady state of "H V R O R V H" (nprop = 1): one unit = [integrate | lists |
// (alchemical || nonbonded || bonded+noise) | finalize]; the three force kernels are forked onto side streams
// so they overlap, and launch gaps shrink to in-graph dependencies.  Step-dependent values come from DevCtrl.
static std::vector<Program> steady_programs(const BluesEngine* h) {
    std::vector<Program> out;
    Program p; p.n = 0;
    auto add = [&](int op) { p.ops[p.n++] = (unsigned char)op; };
    add(OP_V0); add(OP_H01); add(OP_END);
    if (h->remove_cm) {
        if (h->int_blocks == 1) add(OP_CM_BLOCK);
        else { add(OP_CM_REDUCE); out.push_back(p); p.n = 0; add(OP_CM_APPLY); }
    }
    add(OP_H12); add(OP_V2); add(OP_R); add(OP_O); add(OP_R);
    out.push_back(p);
    return out;
}

static int capture_graph(BluesEngine* h) {
    if (h->gexec) { hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
    h->graph_valid = false;
    const Program saved = h->prog; const int saved_trace = h->prog_trace; const unsigned saved_base = h->prog_draw_base;
    const bool saved_nv = h->noise_valid; const unsigned saved_nb = h->noise_draw_base;
    const int64_t saved_launches = h->st_launches;
    h->ctrl_arg = h->d_ctrl.p;
    const std::vector<Program> progs = steady_programs(h);
    const double dummy[3] = {1.0, 1.0, 1.0};
    int rc = 0;
    hipError_t e = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { h->ctrl_arg = nullptr; E_FAIL(h, "hipStreamBeginCapture: %s", hipGetErrorString(e)); }
    for (int u = 0; u < h->graph_units && !rc; u++) {
        h->cur = h->stream;
        for (const Program& p : progs) {
            h->prog = p;
            IntArgs A = make_int_args(h);
            hipLaunchKernelGGL(k_integrate, dim3(h->int_blocks), dim3(h->int_threads), 0, h->cur, A);
        }
        rc |= h->precision == 0 ? launch_lists<float>(h, 0) : launch_lists<double>(h, 0);
        if (h->graph_fork) {
            rc |= ensure_side(h);
            rc |= hipEventRecord(h->evFork, h->stream) != hipSuccess;
            rc |= hipStreamWaitEvent(h->s1, h->evFork, 0) != hipSuccess;
            rc |= hipStreamWaitEvent(h->s2, h->evFork, 0) != hipSuccess;
            h->cur = h->s1; rc |= launch_alchemical(h, dummy, dummy, 7);
            h->cur = h->s2; rc |= launch_bonded(h, true);
            h->cur = h->stream; rc |= h->precision == 0 ? launch_nonbonded<float, false>(h) : launch_nonbonded<double, false>(h);
            rc |= hipEventRecord(h->evJ1, h->s1) != hipSuccess;
            rc |= hipEventRecord(h->evJ2, h->s2) != hipSuccess;
            rc |= hipStreamWaitEvent(h->stream, h->evJ1, 0) != hipSuccess;
            rc |= hipStreamWaitEvent(h->stream, h->evJ2, 0) != hipSuccess;
        } else {
            rc |= launch_alchemical(h, dummy, dummy, 7);
            rc |= h->precision == 0 ? launch_nonbonded<float, false>(h) : launch_nonbonded<double, false>(h);
            rc |= launch_bonded(h, true);
        }
        rc |= launch_finalize(h, dummy);
    }
    hipGraph_t graph = nullptr;
    e = hipStreamEndCapture(h->stream, &graph);
    h->cur = h->stream; h->ctrl_arg = nullptr;
    h->prog = saved; h->prog_trace = saved_trace; h->prog_draw_base = saved_base; h->noise_valid = saved_nv; h->noise_draw_base = saved_nb;
    h->st_launches = saved_launches;
    if (rc || e != hipSuccess || !graph) { if (graph) hipGraphDestroy(graph); E_FAIL(h, "graph capture failed: %s", hipGetErrorString(e)); }
    e = hipGraphInstantiate(&h->gexec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) { h->gexec = nullptr; E_FAIL(h, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
    h->graph_valid = true;
    return 0;
}

// returns the number of whole steps advanced through graph replays (0 if the state is not the steady state)
static int try_graph_steps(BluesEngine* h, int max_steps) {
    if (!h->use_graph || h->split != "HVRORVH" || h->nprop != 1) return 0;
    if (h->h_step < 1 || !h->pass_valid || h->x_edited || h->h_first_step < 1) return 0;
    if (h->prog.n != 3 || h->prog.ops[0] != OP_V0 || h->prog.ops[1] != OP_H01 || h->prog.ops[2] != OP_END) return 0;
    if (h->pass_L != 2 * h->h_step - 1 || h->h_lambda_step != 2 * h->h_step || h->lists_forced) return 0;
    int count = std::min(max_steps, h->nsteps - h->h_step);
    count -= count % h->graph_units;
    if (count <= 0) return 0;
    if (!h->graph_valid && capture_graph(h)) return -1;
    DevCtrl& c = h->host_ctrl;
    c.kint = 0; c.kpass = 0; c.L0 = h->pass_L; c.draw0 = h->h_draw; c.trace0 = h->prog_trace; c.n_lambda = h->n_lambda;
    c.tab_ls = h->d_tab_ls.p; c.tab_le = h->d_tab_le.p;
    if (hipMemcpyAsync(h->d_ctrl.p, &c, sizeof c, hipMemcpyHostToDevice, h->stream) != hipSuccess) { h->err = "ctrl upload failed"; return -1; }
    for (int r = 0; r < count / h->graph_units; r++)
        if (hipGraphLaunch(h->gexec, h->stream) != hipSuccess) { h->err = "hipGraphLaunch failed"; return -1; }
    h->h_step += count; h->steps_since_sort += count; h->h_lambda_step += 2 * count; h->h_lambda = (double)h->h_lambda_step / h->n_lambda;
    h->cur_ls = h->tab_ls[h->h_lambda_step]; h->cur_le = h->tab_le[h->h_lambda_step];
    h->h_draw += (unsigned)count; h->pass_L += 2 * count;
    h->prog_trace = h->h_step - 1; h->prog_draw_base = h->h_draw; h->noise_draw_base = h->h_draw; h->noise_valid = true;
    h->st_passes += count; h->st_launches += (int64_t)count * 6; h->st_graph_steps += count;
    return count;
}

// reference blues/integrators.py:159-209
// One NCMC step = head (everything that may differ between the replicas of a batch: the first-step block and the
// work of an instantaneous Move, which needs a synchronous total energy) + body (the splitting passes; identical
// control flow for replicas in the same mirrored state, so a batch issues it once).
static int step_head(BluesEngine* h) {
    if (h->h_step == 0) {  // first call after construction / reset
        if (flush_program(h)) return 1;
        if (emit(h, OP_PREP)) return 1;
        if (flush_program(h)) return 1;
        h->pass_valid = false;
        HIP_OK(h, hipMemsetAsync(h->d_acc.p, 0, sizeof(DevAccum), h->stream));
        h->acc_cache_valid = false;
        h->h_lambda = 0.0; h->h_lambda_step = 0; h->cur_ls = h->tab_ls[0]; h->cur_le = h->tab_le[0];
        h->h_perturbed = h->h_unperturbed = 0.0; h->unpert_valid = false;
        if (h->switch_mode != BLUES_SWITCH_NONE) {
            // first-step block of the switching integrators (switching.py:1209-1222 / 1341-1348): work reset, the initial
            // alchemical state (table entry 0), its energy, then the propagation that precedes the first perturbation
            h->sw_shadow = 0.0; h->x_edited = false;
            double E;
            if (total_energy(h, &E)) return 1;
            h->sw_Einit = E; h->sw_Epert = E; h->sw_Efinal = 0.0;
            if (run_fragment(h, h->split_first)) return 1;
        }
    }
    if (h->h_step < h->nsteps) {
        if (h->h_first_step < 1) { h->h_first_step = 1; }
        else if (h->x_edited && h->unpert_valid) {  // work done by an instantaneous Move (integrators.py:184-191)
            double E;
            if (total_energy(h, &E)) return 1;
            h->h_perturbed = E;
            if (add_work(h, h->h_perturbed - h->h_unperturbed)) return 1;
        }
        h->x_edited = false; h->unpert_valid = false;
    }
    return 0;
}

static int step_body(BluesEngine* h) {
    if (h->h_step < h->nsteps) {
        if (splitting_pass(h)) return 1;
        if (h->h_lambda > h->prop_min && h->h_lambda <= h->prop_max)
            while (h->h_prop < h->nprop) { h->h_prop++; if (splitting_pass(h)) return 1; }
        if (h->n_H > 0 || h->tracing) {  // END only records the work trace; a plain MD integrator has no protocol work
            if (h->prog_trace >= 0) { if (flush_program(h)) return 1; }
            if (emit(h, OP_END)) return 1;
            h->prog_trace = h->h_step;
        }
        h->h_step++; h->h_prop = 1; h->steps_since_sort++;
    }
    return 0;
}

// reference blues/integrators.py:159-209
static int do_steps(BluesEngine* h, int nsteps) {
    if (ensure_sorted(h)) return 1;
    for (int s = 0; s < nsteps; s++) {
        {
            const int adv = try_graph_steps(h, nsteps - s);
            if (adv < 0) return 1;
            if (adv > 0) { s += adv - 1; continue; }
        }
        if (h->h_step > 0 && h->h_step % RESORT_POLL == 0 && poll_resort(h)) return 1;
        if (step_head(h)) return 1;
        if (step_body(h)) return 1;
    }
    return flush_program(h);
}

// ------------------------------------------------------------------ replica batch: host side
// Everything the control flow of step_body / flush_program reads that can change at run time.  Two members with
// equal signatures issue the same launch sequence with the same launch-to-launch arguments.
struct BatchSig {
    int h_step, h_lambda_step, h_prop, h_first_step, pass_L, prog_n, prog_trace, nprop;
    unsigned h_draw, prog_draw_base, noise_draw_base;
    unsigned char pass_valid, lists_forced, vel_clean, noise_valid, tracing, sorted_ok, pass_valid_for_l, have_positions, fin_pending, fin_mask;
    unsigned char ops[MAX_OPS];
};
static int batch_enter(BluesBatch* B) {
    if (B->entered) return 0;
    for (BluesEngine* m : B->eng) {
        // (a member whose own stream is idle has nothing the batch stream could overtake: between two batch calls that is the rule,
        // and an event record + wait per member was 3-5 ms of every batched boundary call at R = 512)
        if (hipStreamQuery(m->stream) != hipSuccess) {
            (void)hipGetLastError();   // (hipErrorNotReady is not an error)
            if (!m->evFork && hipEventCreateWithFlags(&m->evFork, hipEventDisableTiming) != hipSuccess) { B->err = "hipEventCreate failed"; return 1; }   // (created on first use: blues_engine_create)
            if (hipEventRecord(m->evFork, m->stream) != hipSuccess || hipStreamWaitEvent(B->stream, m->evFork, 0) != hipSuccess) { B->err = "could not order the batch stream after a member's stream"; return 1; }
        }
        m->own_stream = m->stream; m->stream = B->stream; m->cur = B->stream;
    }
    B->entered = true;
    return 0;
}
static void batch_leave(BluesBatch* B) {
    if (!B->entered) return;
    hipStreamSynchronize(B->stream);
    for (BluesEngine* m : B->eng) if (m->own_stream) { m->stream = m->own_stream; m->cur = m->own_stream; m->own_stream = nullptr; }
    B->entered = false;
}

static BatchSig batch_sig(const BluesEngine* h) {
    BatchSig g; memset(&g, 0, sizeof g);
    g.h_step = h->h_step; g.h_lambda_step = h->h_lambda_step; g.h_prop = h->h_prop; g.h_first_step = h->h_first_step; g.pass_L = h->pass_valid ? h->pass_L : 0;
    g.prog_n = h->prog.n; g.prog_trace = h->prog_trace; g.nprop = h->nprop;
    // draw counters enter relative to h_draw: members may have drawn different amounts of noise in the past (the records
    // carry each member's offset to the leader, RepCore.draw_delta), what must agree is where they stand within the step
    g.h_draw = 0; g.prog_draw_base = h->prog.n ? h->h_draw - h->prog_draw_base : 0; g.noise_draw_base = h->noise_valid ? h->h_draw - h->noise_draw_base : 0;
    g.pass_valid = h->pass_valid; g.lists_forced = h->lists_forced; g.vel_clean = h->vel_clean; g.noise_valid = h->noise_valid; g.tracing = h->tracing;
    g.sorted_ok = h->sorted_ok; g.pass_valid_for_l = h->pass_valid_for_l; g.have_positions = h->have_positions;
    g.fin_pending = h->fin_pending; g.fin_mask = g.fin_pending ? (unsigned char)h->fin_mask : 0;   // (what flush_program decides on: members in lock step must agree)
    for (int q = 0; q < h->prog.n; q++) g.ops[q] = h->prog.ops[q];
    return g;
}

// static shape of a member: launch geometry and protocol.  All members must agree.
static bool batch_congruent(const BluesEngine* a, const BluesEngine* b, const char** why) {
#define BC(f) if (a->f != b->f) { *why = #f; return false; }
    BC(device) BC(n) BC(precision) BC(nsteps) BC(nprop) BC(n_lambda) BC(split) BC(remove_cm) BC(dt) BC(gamma) BC(kT) BC(tol) BC(prop_min) BC(prop_max)
    BC(n_itiles) BC(n_tiles) BC(jcap) BC(n_islots) BC(pool_cap) BC(PA) BC(k2_nblocks_env) BC(k2_jiter) BC(seg_len) BC(waves_tile) BC(wpb) BC(npart)
    BC(fuse_forces) BC(fast_step) BC(fuse_big) BC(k2_dense) BC(k2_f32) BC(k1_iw) BC(k1_mode) BC(acap) BC(S) BC(n_lists) BC(n_entries) BC(int_blocks) BC(int_threads) BC(clusters_packed) BC(n_noise) BC(n_rows)
    BC(frag_F) BC(frag_NI) BC(frag_nblk) BC(frag_fpw) BC(frag_nwg) BC(frag_rel)
    BC(cutoff) BC(alpha) BC(sc_alpha) BC(annih_elec) BC(annih_ster) BC(nb_method) BC(pme) BC(pme_K[0]) BC(pme_K[1]) BC(pme_K[2]) BC(pme_order) BC(restr_k) BC(total_mass)
    // (not the box: a MonteCarloBarostat leaves every member in its own; margins, fixed-point scales and PME tables are per member in the records)
#undef BC
    if (a->clusters.size() != b->clusters.size()) { *why = "constraint clusters"; return false; }
    if (a->mobile != b->mobile) { *why = "mobile atoms"; return false; }
    if (a->alch != b->alch) { *why = "alchemical atoms"; return false; }
    if (a->tab_ls != b->tab_ls || a->tab_le != b->tab_le) { *why = "lambda schedule"; return false; }
    for (int ty = 0; ty < T_NTYPES; ty++) if (a->n_terms[ty] != b->n_terms[ty]) { *why = "bonded terms"; return false; }
    return true;
}

// (what batch_congruent compares is fixed at construction or changes with a re-layout, which bumps args_epoch: a member that was
// found congruent with this leader at these epochs is not compared again -- the comparison walks the lambda tables: 3 ms for 1024
// members at the head of every stepping call, energy prefetch and velocity redraw)
static bool batch_congruent_cached(BluesBatch* B, int r, const BluesEngine* lead, const char** why) {
    const int R = B->R();
    if ((int)B->congr_lead.size() != R) { B->congr_lead.assign(R, nullptr); B->congr_le.assign(R, 0); B->congr_me.assign(R, 0); }
    const BluesEngine* m = B->eng[r];
    if (B->congr_lead[r] == lead && B->congr_le[r] == lead->args_epoch && B->congr_me[r] == m->args_epoch) return true;
    if (!batch_congruent(lead, m, why)) return false;
    B->congr_lead[r] = lead; B->congr_le[r] = lead->args_epoch; B->congr_me[r] = m->args_epoch;
    return true;
}

static int batch_refresh_args(BluesBatch* B) {
    bool dirty = false;
    BluesEngine* lead = B->leader ? B->leader : B->eng[0];
    for (int r = 0; r < B->R(); r++) dirty |= B->seen_epoch[r] != B->eng[r]->args_epoch || B->rec_active[r] != (char)(B->active[r] && !B->failed[r] && !B->eng[r]->straggler && !B->eng[r]->solo_now)
                                              || (!B->failed[r] && B->rec_delta[r] != B->eng[r]->h_draw - lead->h_draw);
    if (!dirty) return 0;
    const double one[3] = {1.0, 1.0, 1.0};
    std::vector<RepCore> core(B->R()); std::vector<RepNb<float>> nf; std::vector<RepNb<double>> nd;
    const bool single = B->eng[0]->precision == 0;
    if (single) nf.resize(B->R()); else nd.resize(B->R());
    for (int r = 0; r < B->R(); r++) {
        BluesEngine* h = B->eng[r];
        B->rec_active[r] = (char)(B->active[r] && !B->failed[r] && !h->straggler && !h->solo_now);   // (a straggler's kernels are its own: the shared launches pass it by)
        core[r].active = B->rec_active[r];
        B->rec_delta[r] = h->h_draw - lead->h_draw; core[r].draw_delta = B->rec_delta[r];
        if (single) nf[r].active = B->rec_active[r]; else nd[r].active = B->rec_active[r];
        if (!h->sorted_ok) { core[r].active = 0; if (single) nf[r].active = 0; else nd[r].active = 0; continue; }  // buffers not laid out (its sort failed): never touched
        core[r].al = make_alch_args(h, one, one, 7); core[r].bo = make_bonded_args(h); core[r].fin = make_fin_args(h, one); core[r].in = make_int_args(h);
        core[r].in.work_trace = h->d_trace.p;  // the launch decides whether it is written (IntDyn.tracing)
        core[r].in.n_noise = h->n_noise;
        if (single) { nf[r].L = make_list_args(h); nf[r].nb = make_nb_args<float>(h); nf[r].c = make_nbconst<float>(h); nf[r].img = h->d_img_f.p; nf[r].pme = make_pme_args<float>(h, false); nf[r].fr = make_frag_args(h); }
        else { memset(&nd[r].fr, 0, sizeof nd[r].fr); nd[r].L = make_list_args(h); nd[r].nb = make_nb_args<double>(h); nd[r].c = make_nbconst<double>(h); nd[r].img = h->d_img_d.p; nd[r].pme = make_pme_args<double>(h, false); }
        B->seen_epoch[r] = h->args_epoch;
    }
    // the records may be in use by launches still in flight
    if (hipStreamSynchronize(B->stream) != hipSuccess) { B->err = "stream synchronisation failed"; return 1; }
    try { B->d_core.upload(core); if (single) B->d_nb_f.upload(nf); else B->d_nb_d.upload(nd); } catch (std::string& e) { B->err = e; return 1; }
    return 0;
}

// Total potential and/or kinetic energy of every active member with batched launches and ONE read-back; the values land in
// the members' energy caches, where the per-member getState / setPositions bookkeeping that follows finds them (the
// reference asks each Context separately: simulation.py:908-909).  Sums are formed on the host from the same partials in
// the same order as a lone evaluation.  Members that cannot take part (other alchemical parameters than the rest, a
// pending one-off frozen-frozen energy) are simply left for their own on-demand evaluation.
static int batch_prefetch(BluesBatch* B, int what) {
    const int R = B->R();
    if (R == 0) return 0;
    std::vector<char> live(R), need(R, 0);
    for (int r = 0; r < R; r++) if (B->active[r] && !B->failed[r] && resolve_xfer(B->eng[r])) B->failed[r] = 1;
    {   // A few members whose State arrived far from where their tiles were laid out (a restore after a re-sort at a poll: resolve_xfer) have
        // no layout now.  They are laid out HERE: left as they were, the check below sent ALL members to their own on-demand evaluation --
        // 1024 lone energy evaluations, 0.15-0.25 s, in an iteration that names no layout event (round 6: three such iterations in 40 runs).
        int lacking = 0, have = 0;
        for (int r = 0; r < R; r++) if (B->active[r] && !B->failed[r] && B->eng[r]->have_positions && !B->eng[r]->straggler) { have++; lacking += !B->eng[r]->sorted_ok; }
        if (lacking > 0 && lacking <= std::max(1, have / 8) && !B->lockstep) {
            if (hipStreamSynchronize(B->stream) != hipSuccess) { B->err = "stream synchronisation failed"; return 1; }
            for (int r = 0; r < R; r++) if (B->active[r] && !B->failed[r] && B->eng[r]->have_positions && !B->eng[r]->straggler && !B->eng[r]->sorted_ok) {
                if (ensure_sorted(B->eng[r])) B->failed[r] = 1;
                B->st_relayouts++;
            }
        }
    }
    for (int r = 0; r < R; r++) live[r] = B->active[r] && !B->failed[r] && B->eng[r]->have_positions && B->eng[r]->sorted_ok && !B->eng[r]->straggler;   // (a straggler evaluates on demand, with launches of its own)
    BluesEngine* lead = nullptr;
    for (int r = 0; r < R; r++) if (live[r]) { lead = B->eng[r]; break; }
    if (!lead) return 0;
    for (int r = 0; r < R; r++) if (B->active[r] && !B->failed[r] && !live[r] && !B->eng[r]->straggler) return 0;   // a member without a layout yet: everybody evaluates on demand
    const bool was_entered = B->entered;
    bool ok = true;
    auto pe_pass = [&]() -> int {   // potential energies at the members' CURRENT alchemical parameters
        int n_need = 0, n_live = 0;
        {   // the frozen-frozen constant of the members that have none.  One member (a chain that joined late): its own evaluation.
            // Several -- every chain of a batch whose State came from the MD leg, in every iteration: the frozen atoms were given new
            // positions -- in one launch over a work list; member by member this was a full lone energy evaluation with its host
            // synchronisations each, 0.7 s per 1024 chains and iteration (half of the NCMC leg of a real run)
            std::vector<int> lack;
            for (int r = 0; r < R; r++) if (live[r] && !B->eng[r]->e_frozen_valid) lack.push_back(r);
            bool together = lack.size() > 1;
            for (size_t q = 0; q < lack.size() && together; q++) { const char* why = ""; together = batch_congruent_cached(B, lack[q], lead, &why) && B->eng[lack[q]]->prog.n == 0; }
            if (together) {
                if (batch_enter(B)) return 1;
                B->leader = lead;
                if (batch_refresh_args(B)) { if (!was_entered) batch_leave(B); return 1; }
                const EnergyShape g = energy_shape(lead);
                const int count = (int)lack.size();
                std::vector<double> sums;
                try {
                    if ((int)B->d_frozen_work.n < count) B->d_frozen_work.alloc(R);
                    if ((int)B->d_frozen_sum.n < 2 * count) B->d_frozen_sum.alloc((size_t)2 * R);
                    if (hipMemcpyAsync(B->d_frozen_work.p, lack.data(), sizeof(int) * count, hipMemcpyHostToDevice, B->stream) != hipSuccess) throw std::string("upload of the work list failed");
                    if (lead->precision == 0) {
                        hipLaunchKernelGGL(k_frozen_boxes_b<float>, dim3((g.nfb + 3) / 4, count), dim3(256), 0, B->stream, B->d_nb_f.p, B->d_frozen_work.p, (int)g.off_frozen);
                        hipLaunchKernelGGL(k_energy_frozen_b<float>, dim3(g.nfb, count), dim3(256), 0, B->stream, B->d_nb_f.p, B->d_frozen_work.p, (int)g.off_frozen);
                        hipLaunchKernelGGL(k_sum_frozen_b<float>, dim3((count + 63) / 64), dim3(64), 0, B->stream, B->d_nb_f.p, B->d_frozen_work.p, count, (int)g.off_frozen, g.nfb, B->d_frozen_sum.p);
                    } else {
                        hipLaunchKernelGGL(k_frozen_boxes_b<double>, dim3((g.nfb + 3) / 4, count), dim3(256), 0, B->stream, B->d_nb_d.p, B->d_frozen_work.p, (int)g.off_frozen);
                        hipLaunchKernelGGL(k_energy_frozen_b<double>, dim3(g.nfb, count), dim3(256), 0, B->stream, B->d_nb_d.p, B->d_frozen_work.p, (int)g.off_frozen);
                        hipLaunchKernelGGL(k_sum_frozen_b<double>, dim3((count + 63) / 64), dim3(64), 0, B->stream, B->d_nb_d.p, B->d_frozen_work.p, count, (int)g.off_frozen, g.nfb, B->d_frozen_sum.p);
                    }
                    if (hipStreamSynchronize(B->stream) != hipSuccess) throw std::string("frozen-frozen energy launch failed");
                    sums.resize((size_t)2 * count);
                    if (hipMemcpy(sums.data(), B->d_frozen_sum.p, sizeof(double) * 2 * count, hipMemcpyDeviceToHost) != hipSuccess) throw std::string("download of the frozen-frozen energies failed");
                } catch (std::string& e) { B->err = e; if (!was_entered) batch_leave(B); return 1; }
                for (int q = 0; q < count; q++) { BluesEngine* m = B->eng[lack[q]]; m->e_frozen[0] = sums[2 * q]; m->e_frozen[1] = sums[2 * q + 1]; m->e_frozen_valid = true; m->st_launches++; }
            } else for (int r : lack) {
                double E;
                if (total_energy(B->eng[r], &E)) { B->failed[r] = 1; live[r] = 0; }
            }
        }
        for (int r = 0; r < R; r++) if (live[r]) {
            BluesEngine* m = B->eng[r];
            n_live++;
            need[r] = !m->ecache.find(m->cur_ls, m->cur_le, nullptr);
            n_need += need[r];
        }
        // members that already know their energy simply take part again (same launches, same value): masking them out of the
        // shared launches would cost more than it saves, and ONE such member used to send all the others to their own evaluations
        bool uniform = n_need > 1;
        for (int r = 0; r < R && uniform; r++) if (live[r]) {
            const BluesEngine* m = B->eng[r];
            uniform = m->e_frozen_valid && m->cur_ls == lead->cur_ls && m->cur_le == lead->cur_le && m->lists_forced == lead->lists_forced && m->prog.n == 0;
            const char* why = "";
            uniform = uniform && batch_congruent_cached(B, r, lead, &why);
        }
        if (uniform) {
            if (batch_enter(B)) return 1;
            B->leader = lead;
            if (batch_refresh_args(B)) { if (!was_entered) batch_leave(B); return 1; }
            B->lockstep = true;
            for (int r = 0; r < R && ok; r++) if (live[r]) ok = energy_launch(B->eng[r]) == 0;
            B->lockstep = false;
            const EnergyShape g = energy_shape(lead);
            const int n_nb = 2 * g.nw, n_b = g.nbb * T_NTYPES, n_al = lead->alch.empty() ? 0 : (lead->k2_nblocks_env + 1) * K2_NP, stride = n_nb + n_b + n_al + 2;
            std::vector<double> slab; std::vector<int> hints;
            try {
                if (ok) {
                    if (B->d_gather.n != (size_t)R * stride) B->d_gather.alloc((size_t)R * stride);
                    if ((int)B->d_hints.n != R) B->d_hints.alloc(R);
                    if (lead->precision == 0) hipLaunchKernelGGL(k_gather_energy_parts_b<float>, dim3(R), dim3(256), 0, B->stream, B->d_nb_f.p, B->d_core.p, n_nb, n_b, n_al, stride, B->d_gather.p);
                    else hipLaunchKernelGGL(k_gather_energy_parts_b<double>, dim3(R), dim3(256), 0, B->stream, B->d_nb_d.p, B->d_core.p, n_nb, n_b, n_al, stride, B->d_gather.p);
                    hipLaunchKernelGGL(k_gather_hints_b, dim3((R + 255) / 256), dim3(256), 0, B->stream, B->d_core.p, R, B->d_hints.p);
                    ok = hipStreamSynchronize(B->stream) == hipSuccess;
                    if (ok) { B->d_gather.download(slab); B->d_hints.download(hints); }
                }
            } catch (std::string& e) { B->err = e; ok = false; }
            if (ok) for (int r = 0; r < R; r++) if (live[r] && !(hints[r] & 2)) {   // a member with an error flag keeps no cached value: its own call will report
                BluesEngine* m = B->eng[r];
                const double* o = slab.data() + (size_t)r * stride;
                double T[BLUES_N_ENERGY_TERMS], one[2], E, E_one;
                energy_sum(m, o, o + n_nb, (int)o[n_nb + n_b + n_al], o + n_nb + n_b, o[n_nb + n_b + n_al + 1], T, one);
                energy_totals(m, T, one, &E, &E_one);
                m->ecache.put(1.0, 1.0, E_one); m->ecache.put(m->cur_ls, m->cur_le, E);
            }
            B->st_prefetch_pe++;
        }
        return 0;
    };
    if (what & 1) { if (pe_pass()) return 1; }
    if (ok && (what & 4)) {   // ... and at lambda_sterics = lambda_electrostatics = 1 (what _computeAlchemicalCorrection asks next)
        std::vector<double> sv(2 * R);
        for (int r = 0; r < R; r++) { sv[2 * r] = B->eng[r]->cur_ls; sv[2 * r + 1] = B->eng[r]->cur_le; if (live[r]) { B->eng[r]->cur_ls = 1.0; B->eng[r]->cur_le = 1.0; } }
        const int rc = pe_pass();
        for (int r = 0; r < R; r++) { B->eng[r]->cur_ls = sv[2 * r]; B->eng[r]->cur_le = sv[2 * r + 1]; }
        if (rc) return 1;
    }
    if (ok && (what & 2)) {
        bool any = false;
        for (int r = 0; r < R; r++) if (live[r]) any |= !B->eng[r]->ke_cache_valid;
        if (any) {
            if (batch_enter(B)) return 1;
            for (int r = 0; r < R && ok; r++) if (live[r]) ok = flush_program(B->eng[r]) == 0;   // (nothing pending in the usual call sites)
            B->leader = lead;
            if (ok && batch_refresh_args(B)) ok = false;
            std::vector<double> ke;
            try {
                if (ok) {
                    if (B->d_gather.n < (size_t)7 * R) B->d_gather.alloc((size_t)7 * R);
                    hipLaunchKernelGGL(k_kinetic_b, dim3(R), dim3(256), 0, B->stream, B->d_core.p, B->d_gather.p);
                    ok = hipStreamSynchronize(B->stream) == hipSuccess;
                    if (ok) { ke.resize((size_t)7 * R); ok = hipMemcpy(ke.data(), B->d_gather.p, sizeof(double) * 7 * R, hipMemcpyDeviceToHost) == hipSuccess; }
                }
            } catch (std::string& e) { B->err = e; ok = false; }
            if (ok) for (int r = 0; r < R; r++) if (live[r]) {
                BluesEngine* m = B->eng[r];
                m->ke_cache = ke[r]; m->ke_cache_valid = true;
                static_assert(sizeof(DevAccum) == 6 * sizeof(double), "k_kinetic_b copies six doubles");
                memcpy(&m->acc_cache, &ke[(size_t)R + 6 * r], sizeof(DevAccum)); m->acc_cache_valid = true; m->acc_cache_stamp = m->st_launches;
            }
            B->st_prefetch_ke++;
        }
    }
    if (!was_entered) batch_leave(B);
    if (!ok && B->err.empty()) B->err = "energy prefetch failed: " + lead->err;
    return ok ? 0 : 1;
}

// Advances every member by n_steps.  status[r] != 0: member r raised (message via blues_last_error(engine r)); the
// others are unaffected.  Returns non-zero only for errors of the batch itself (B->err).
static int batch_do_steps(BluesBatch* B, int n_steps, bool tracing, int* status) {
    const int R = B->R();
    // failed[r] doubles as "skip this member": inactive members start out skipped (status stays 0)
    B->failed.assign(R, 0);
    for (int r = 0; r < R; r++) { status[r] = 0; B->eng[r]->tracing = tracing; if (!B->active[r]) B->failed[r] = 1; }
    auto fail = [&](int r) { B->failed[r] = 1; status[r] = 1; };
    for (int r = 0; r < R; r++) if (!B->failed[r] && ensure_sorted(B->eng[r])) fail(r);
    // the leader issues the shared launches: a member in the batch's layout (stragglers issue their own)
    auto pick_leader = [&]() { B->leader = nullptr; for (int r = 0; r < R; r++) if (!B->failed[r] && !B->eng[r]->straggler) { B->leader = B->eng[r]; break; } };
    auto any_straggler = [&]() { for (int r = 0; r < R; r++) if (!B->failed[r] && B->eng[r]->straggler) return true; return false; };
    pick_leader();
    if (B->leader) for (int r = 0; r < R; r++) if (!B->failed[r] && B->eng[r]->straggler && B->eng[r]->h_step == 0 && B->eng[r]->have_positions) {
        // a straggler at the start of a switch: back into the shared launches if its atoms fit the batch's shape again
        if (hipStreamSynchronize(B->stream) != hipSuccess) { B->err = "stream synchronisation failed"; return 1; }
        if (try_rejoin(B, B->eng[r], B->leader)) fail(r);
    }
    if (!B->leader && !any_straggler()) return 0;
    for (int r = 0; r < R; r++) if (!B->failed[r] && !B->eng[r]->straggler) {
        const char* why = "";
        if (!batch_congruent_cached(B, r, B->leader, &why)) { B->err = std::string("replicas of a batch must be congruent; they differ in ") + why; return 1; }
    }
    if (batch_refresh_args(B)) return 1;
    // phase(f): run f on every live member, in lock step when their signatures agree.  Round 6: a FEW members out of step with the rest
    // -- one that was laid out again inside this call (a re-sort at a poll, a straggler that came back: its lists are forced, its last
    // pass is gone) -- take the step on launches of their own, as a straggler does, and the others stay in lock step; one step later the
    // mirrors agree again.  (Until then ONE such member sent all of them through per-member launches for a step: 0.25 s per 1024 chains.)
    auto phase = [&](int (*f)(BluesEngine*)) -> int {
        pick_leader();
        if (!B->leader && !any_straggler()) return 0;
        std::vector<int> odd;
        if (B->leader) {
            auto split = [&](const BluesEngine* ref) {
                odd.clear();
                const BatchSig lead = batch_sig(ref);
                int same = 0;
                for (int r = 0; r < R; r++) if (!B->failed[r] && !B->eng[r]->straggler) {
                    const BatchSig g = batch_sig(B->eng[r]);
                    if (!memcmp(&g, &lead, sizeof g)) { same++; continue; }
                    if (odd.empty() && B->tune.debug_lists) {
#define SIGF(fld) if (g.fld != lead.fld) fprintf(stderr, "[lockstep] member %d differs from member %d in %s: %d against %d (step %d)\n", r, ref->batch_index, #fld, (int)g.fld, (int)lead.fld, g.h_step);
                        SIGF(h_step) SIGF(h_lambda_step) SIGF(h_prop) SIGF(h_first_step) SIGF(pass_L) SIGF(prog_n) SIGF(prog_trace) SIGF(nprop) SIGF(prog_draw_base) SIGF(noise_draw_base)
                        SIGF(pass_valid) SIGF(lists_forced) SIGF(vel_clean) SIGF(noise_valid) SIGF(tracing) SIGF(sorted_ok) SIGF(pass_valid_for_l) SIGF(have_positions) SIGF(fin_pending) SIGF(fin_mask)
#undef SIGF
                    }
                    odd.push_back(r);
                }
                return same;
            };
            int same = split(B->leader);
            if ((int)odd.size() > same) { B->leader = B->eng[odd[0]]; same = split(B->leader); }   // (the first member is the one out of step: the majority leads)
            const bool partial = !odd.empty() && (int)odd.size() <= std::max(1, R / 32);
            if (odd.empty() || partial) {
                for (int r : odd) B->eng[r]->solo_now = true;   // (out of the shared records for this phase: batch_refresh_args)
                int rc = batch_refresh_args(B);
                if (!rc) {
                    B->lockstep = true;
                    for (int r = 0; r < R; r++) if (!B->failed[r] && !B->eng[r]->straggler && !B->eng[r]->solo_now && f(B->eng[r])) {
                        // a host-side failure of the leader leaves the shared launch sequence incomplete: every member is affected
                        if (B->eng[r] == B->leader) { for (int q = 0; q < R; q++) if (!B->failed[q] && !B->eng[q]->straggler && !B->eng[q]->solo_now) { if (q != r) B->eng[q]->err = "batch leader failed: " + B->leader->err; fail(q); } }
                        else fail(r);
                    }
                    B->lockstep = false;
                    B->st_lockstep_steps++;
                    if (partial) B->st_partial_steps++;
                    for (int r : odd) if (!B->failed[r] && f(B->eng[r])) fail(r);
                }
                for (int r : odd) B->eng[r]->solo_now = false;
                if (rc) return 1;
            } else {
                if (batch_refresh_args(B)) return 1;
                for (int r = 0; r < R; r++) if (!B->failed[r] && !B->eng[r]->straggler && f(B->eng[r])) fail(r);
                B->st_fallback_steps++;
            }
        } else if (batch_refresh_args(B)) return 1;   // leader change, a member dropped out, a re-sort: cheap no-op otherwise
        // the stragglers: the same function, launches of their own (behind the shared ones, on the batch's stream)
        for (int r = 0; r < R; r++) if (!B->failed[r] && B->eng[r]->straggler && f(B->eng[r])) fail(r);
        return 0;
    };
    for (int s = 0; s < n_steps; s++) {
        // same cadence as a lone engine (keyed on each member's own step counter, so results stay identical to solo runs);
        // one gathered read-back tells which members (if any) need the per-member treatment
        {
            bool due = false;
            for (int r = 0; r < R; r++) due |= !B->failed[r] && B->eng[r]->h_step > 0 && B->eng[r]->h_step % RESORT_POLL == 0;
            if (due) {
                pick_leader();
                if (!B->leader && !any_straggler()) return 0;
                if (phase(flush_program)) return 1;
                if (batch_refresh_args(B)) return 1;
                std::vector<int> hints;
                try {
                    if ((int)B->d_hints.n != R) B->d_hints.alloc(R);
                    hipLaunchKernelGGL(k_gather_hints_b, dim3((R + 255) / 256), dim3(256), 0, B->stream, B->d_core.p, R, B->d_hints.p);
                    if (hipStreamSynchronize(B->stream) != hipSuccess) { B->err = "stream synchronisation failed"; return 1; }
                    B->d_hints.download(hints);
                } catch (std::string& e) { B->err = e; return 1; }
                {   // the members whose builders asked for a new order: re-sorted on the host's cores together (5-10 ms of host work each)
                    std::vector<BluesEngine*> todo;
                    for (int r = 0; r < R; r++)
                        if (!B->failed[r] && B->eng[r]->h_step > 0 && B->eng[r]->h_step % RESORT_POLL == 0) {
                            const bool aged = resort_by_age(B->eng[r]);
                            if (!aged && !(hints[r] & 1)) continue;
                            // (a member that is due by age may have a hint pending as well: it is cleared -- and the lists given more room
                            // where they ask for it -- here, or the builder's request would cost a second re-sort 64 steps later)
                            const int due = poll_resort_due(B->eng[r]);
                            if (due < 0) { fail(r); continue; }
                            if (aged && !due) B->eng[r]->st_resorts++;
                            if (aged || due) todo.push_back(B->eng[r]);
                        }
                    if (!todo.empty()) {
                        const auto t_rs = std::chrono::steady_clock::now();
                        relayout_many(todo, [](BluesEngine*) { return true; });
                        for (BluesEngine* m : todo) if (m->relayout_failed) fail(m->batch_index);
                        B->st_poll_resorts += (int64_t)todo.size();
                        B->st_resort_sec += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_rs).count();
                    }
                }
                {   // a member whose lists have outgrown the batch's layout shape: it steps on its own from here (make_straggler); a new
                    // shape for everybody only when more than a handful have (rare: a run whose mobile atoms scatter for good)
                    bool over = false;
                    for (int r = 0; r < R; r++) if (!B->failed[r] && B->eng[r]->shape_overflow && !B->eng[r]->straggler) {
                        if (batch_can_straggle(B, B->eng[r])) {
                            if (hipStreamSynchronize(B->stream) != hipSuccess) { B->err = "stream synchronisation failed"; return 1; }
                            if (make_straggler(B, B->eng[r])) fail(r);
                        } else over = true;
                    }
                    if (over) {
                        if (hipStreamSynchronize(B->stream) != hipSuccess) { B->err = "stream synchronisation failed"; return 1; }
                        if (batch_plan_shape(B, false)) return 1;
                        B->st_replans++;
                    }
                }
            }
        }
        {   // the work of an instantaneous Move needs U(x_moved) of every member that was edited: evaluate them together
            int edited = 0;
            for (int r = 0; r < R; r++) if (!B->failed[r]) { const BluesEngine* m = B->eng[r]; edited += m->h_step > 0 && m->h_step < m->nsteps && m->h_first_step >= 1 && m->x_edited && m->unpert_valid; }
            if (edited > 1 && batch_prefetch(B, 1)) return 1;
            B->defer_work = edited > 1;
        }
        for (int r = 0; r < R; r++) if (!B->failed[r] && step_head(B->eng[r])) fail(r);
        if (B->defer_work) {
            // the members' work increments in one launch (member by member it was a 1-thread launch each: 11 ms of host time for
            // 1024 chains at the head of the step after a Move, with the device idle)
            B->defer_work = false;
            B->h_wacc.assign(R, nullptr); B->h_wdelta.assign(R, 0.0);
            int any = 0;
            for (int r = 0; r < R; r++) { BluesEngine* m = B->eng[r]; if (m->work_pending_set) { B->h_wacc[r] = m->d_acc.p; B->h_wdelta[r] = m->work_pending; any++; m->work_pending = 0.0; m->work_pending_set = false; } }
            if (any) {
                try { if ((int)B->d_wacc.n != R) { B->d_wacc.reserve(R); B->d_wdelta.reserve(R); } } catch (std::string& e) { B->err = e; return 1; }
                hipStream_t st = B->stream;
                if (hipMemcpyAsync(B->d_wacc.p, B->h_wacc.data(), sizeof(DevAccum*) * R, hipMemcpyHostToDevice, st) != hipSuccess ||
                    hipMemcpyAsync(B->d_wdelta.p, B->h_wdelta.data(), sizeof(double) * R, hipMemcpyHostToDevice, st) != hipSuccess) { B->err = "upload of the work increments failed"; return 1; }
                hipLaunchKernelGGL(k_add_work_b, dim3((R + 255) / 256), dim3(256), 0, st, reinterpret_cast<DevAccum* const*>(B->d_wacc.p), B->d_wdelta.p, R);
                if (hipGetLastError() != hipSuccess) { B->err = "k_add_work_b launch failed"; return 1; }
            }
        }
        if (phase(step_body)) return 1;
    }
    if (phase(flush_program)) return 1;
    for (int r = 0; r < R; r++) B->eng[r]->tracing = false;
    // error flags: one gathered read-back tells which members (normally none) need their own check_flags
    pick_leader();
    bool gathered = false;
    std::vector<int> hints;
    if ((B->leader || any_straggler()) && !batch_refresh_args(B)) {
        try {
            if ((int)B->d_hints.n != R) B->d_hints.alloc(R);
            hipLaunchKernelGGL(k_gather_hints_b, dim3((R + 255) / 256), dim3(256), 0, B->stream, B->d_core.p, R, B->d_hints.p);
            if (hipStreamSynchronize(B->stream) == hipSuccess) { B->d_hints.download(hints); gathered = (int)hints.size() == R; }
        } catch (std::string&) { gathered = false; }
    }
    B->leader = B->eng[0];
    for (int r = 0; r < R; r++) if (!B->failed[r] && (!gathered || (hints[r] & 2)) && check_flags(B->eng[r])) fail(r);
    return 0;
}

// ------------------------------------------------------------------ C-ABI
extern "C" {

int blues_abi_version(void) { return BLUES_ABI_VERSION; }

void blues_tuning_default(BluesTuning* t) {
    memset(t, 0, sizeof *t);
    t->struct_size = (int32_t)sizeof *t;
    t->prune_margin = -1.0;
    t->k1_mode = -1; t->fuse_forces = -1; t->fuse_big = -1; t->fast_step = -1; t->slot_mask = -1; t->fork = 1; t->k2_dense = -1; t->use_graph = -1; t->graph_fork = -1; t->fuse_finalize = 1;
}
int blues_set_tuning(const BluesTuning* t) {
    if (!t) { blues_tuning_default(&g_tuning); return 0; }
    if (t->struct_size != (int32_t)sizeof(BluesTuning)) { g_create_error = "blues_set_tuning: struct_size does not match this library's BluesTuning (call blues_tuning_default first)"; return 1; }
    g_tuning = *t;
    return 0;
}
int blues_get_tuning(BluesTuning* t) { *t = g_tuning; return 0; }

const char* blues_last_error(const BluesEngine* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

static int create_impl(BluesEngine* h, const BluesSystemDesc* s, const BluesIntegratorDesc* it) {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) if (r != c && s->box[3 * r + c] != 0.0) E_FAIL(h, "only orthorhombic boxes are supported");
    if (s->nonbonded_method != BLUES_NB_PME_DIRECT && s->nonbonded_method != BLUES_NB_PME) E_FAIL(h, "the GPU engine supports periodic systems only (BLUES_NB_PME_DIRECT, BLUES_NB_PME)");
    if (s->nonbonded_method == BLUES_NB_PME) {
        h->pme = true; h->pme_order = s->pme_order; h->disp_corr = s->dispersion_correction;
        for (int d = 0; d < 3; d++) { h->pme_K[d] = s->pme_grid[d]; if (h->pme_K[d] < s->pme_order || h->pme_K[d] > 256) E_FAIL(h, "PME mesh %d along axis %d is outside [order, 256]", h->pme_K[d], d); }
        if (h->pme_order != 5) E_FAIL(h, "PME order %d: the engine implements order 5 (OpenMM's PME_ORDER)", h->pme_order);
    }
    const int n = h->n = s->n_atoms;
    if (n <= 0) E_FAIL(h, "empty system");
    h->box[0] = s->box[0]; h->box[1] = s->box[4]; h->box[2] = s->box[8];
    h->nb_method = BLUES_NB_PME_DIRECT; h->cutoff = s->cutoff;   // (the direct-space kernels are the same under both methods)
    h->alpha = s->ewald_alpha; h->sc_alpha = s->softcore_alpha;
    for (int k = 0; k < 3; k++) if (h->box[k] < 2.0 * h->cutoff) E_FAIL(h, "box edge %g < 2*cutoff", h->box[k]);
    h->tune = g_tuning;
    if (h->tune.assume_batch > 0) h->batch_R = h->tune.assume_batch;
    if (h->tune.skin > 0.0) { h->skin = h->tune.skin; h->skin_fixed = true; }
    derive_margins(h);   // (again once the mobile set is known: sort_and_tile)
    fit_ewald_poly(h->alpha, h->cutoff, &h->ewpoly);
    h->annih_elec = s->annihilate_electrostatics; h->annih_ster = s->annihilate_sterics; h->remove_cm = s->remove_cm_motion;
    h->alch.assign(s->alchemical_atoms, s->alchemical_atoms + s->n_alchemical);
    if (h->alch.size() > 64) E_FAIL(h, "more than 64 alchemical atoms is not supported yet");
    for (int a : h->alch) if (a < 0 || a >= n) E_FAIL(h, "alchemical atom out of range");
    {   // the System's host tables: found by content (FNV-1a over every array they are made from), built once
        static std::mutex mu; static std::map<uint64_t, std::weak_ptr<HostTopology>> known;
        uint64_t key = 1469598103934665603ull;
        auto mix = [&](const void* p, size_t bytes) {   // (FNV-1a over 64-bit words, the tail byte by byte: 2 MB of arrays per engine)
            const unsigned char* c = static_cast<const unsigned char*>(p);
            size_t i = 0;
            for (; i + 8 <= bytes; i += 8) { uint64_t w; memcpy(&w, c + i, 8); key ^= w; key *= 1099511628211ull; key ^= key >> 29; }
            for (; i < bytes; i++) { key ^= c[i]; key *= 1099511628211ull; }
        };
        const int32_t counts[5] = {n, s->n_exclusions, s->n_alchemical, s->n_bonds, s->n_constraints};
        mix(counts, sizeof counts);
        mix(s->mass, sizeof(double) * n); mix(s->charge, sizeof(double) * n); mix(s->sigma, sizeof(double) * n); mix(s->epsilon, sizeof(double) * n);
        mix(s->exclusions, sizeof(int32_t) * 2 * (size_t)s->n_exclusions); mix(s->alchemical_atoms, sizeof(int32_t) * (size_t)s->n_alchemical);
        mix(s->bond_atoms, sizeof(int32_t) * 2 * (size_t)s->n_bonds); mix(s->constraint_atoms, sizeof(int32_t) * 2 * (size_t)s->n_constraints);
        std::lock_guard<std::mutex> lk(mu);
        auto it = known.find(key);
        if (it != known.end()) h->T = it->second.lock();
        if (h->T && ((int)h->T->mass.size() != n || (int)h->T->link_pairs.size() != 2 * (s->n_bonds + s->n_constraints))) h->T.reset();   // (a hash collision: build afresh, share nothing)
        if (!h->T) {
            auto T = std::make_shared<HostTopology>();
            T->mass.assign(s->mass, s->mass + n); T->charge.assign(s->charge, s->charge + n); T->sigma.assign(s->sigma, s->sigma + n); T->eps.assign(s->epsilon, s->epsilon + n);
            T->excl.assign(n, {});
            for (int e = 0; e < s->n_exclusions; e++) {
                const int a = s->exclusions[2 * e], b = s->exclusions[2 * e + 1];
                if (a < 0 || b < 0 || a >= n || b >= n) E_FAIL(h, "exclusion references atom out of range");
                T->excl[a].push_back(b); T->excl[b].push_back(a);
            }
            T->alch_local.assign(n, -1);
            for (size_t a = 0; a < h->alch.size(); a++) T->alch_local[h->alch[a]] = (int)a;
            T->qn = T->charge; for (int a : h->alch) T->qn[a] = 0.0;
            T->link_pairs.assign(s->bond_atoms, s->bond_atoms + 2 * (size_t)s->n_bonds); T->link_pairs.insert(T->link_pairs.end(), s->constraint_atoms, s->constraint_atoms + 2 * (size_t)s->n_constraints);
            h->T = T; known[key] = T;
            for (auto q = known.begin(); q != known.end();) { if (q->second.expired()) q = known.erase(q); else ++q; }
        }
    }
    for (int a : h->alch) for (int p : h->T->excl[a]) if (h->T->alch_local[p] < 0) h->check_env_excl = 1;
    h->total_mass = 0.0;
    for (int i = 0; i < n; i++) if (h->T->mass[i] != 0.0) { h->mobile.push_back(i); h->total_mass += h->T->mass[i]; }
    // integrator
    h->dt = it->timestep; h->temperature = it->temperature; h->gamma = it->collision_rate; h->kT = KB_KJ * it->temperature;
    h->tol = it->constraint_tolerance; h->nsteps = it->nsteps_neq; h->nprop = it->nprop; h->n_lambda = it->n_lambda_steps;
    h->prop_min = it->prop_lambda_min; h->prop_max = it->prop_lambda_max; h->seed = it->seed; h->replica = it->replica; h->precision = it->precision;
    h->switch_mode = it->switching_mode; h->psteps = it->steps_per_propagation;
    if (h->switch_mode != BLUES_SWITCH_NONE) {
        if (h->switch_mode != BLUES_SWITCH_VV && h->switch_mode != BLUES_SWITCH_GHMC) E_FAIL(h, "unknown switching_mode %d", h->switch_mode);
        if (h->psteps < 0 || h->nsteps < 1) E_FAIL(h, "switching integrators need nsteps_neq >= 1 and steps_per_propagation >= 0");
        if (h->switch_mode == BLUES_SWITCH_VV) for (int q = 0; q < h->psteps; q++) h->split_first += "AV";
        else if (h->psteps > 0) h->split_first = "OAVO";
        h->split = "H" + h->split_first;
        // every velocity-Verlet step is a whole timestep: two half kicks, one drift; the two velocity randomisations of a GHMC
        // step use sqrt(b) = exp(-gamma dt / 2), which is the O substep of half a timestep
        h->n_V = 2; h->n_R = 1; h->n_O = 2; h->n_H = 1;
    }
    for (const char* p = h->switch_mode ? "" : it->splitting; *p; p++) {
        if (*p == ' ') continue;
        if (*p != 'R' && *p != 'V' && *p != 'O' && *p != 'H' && *p != 'L') E_FAIL(h, "unsupported splitting token '%c'", *p);
        h->split.push_back(*p);
        if (*p == 'R') h->n_R++; else if (*p == 'V') h->n_V++; else if (*p == 'O') h->n_O++; else if (*p == 'H') h->n_H++; else h->n_L++;
    }
    if (h->n_lambda != h->nsteps * h->n_H) E_FAIL(h, "n_lambda_steps must equal nsteps_neq * (number of H in splitting)");
    h->tab_ls.assign(it->lambda_sterics, it->lambda_sterics + h->n_lambda + 1);
    h->tab_le.assign(it->lambda_electrostatics, it->lambda_electrostatics + h->n_lambda + 1);
    h->prog.n = 0;
    HIP_OK(h, hipSetDevice(h->device));
    { SetupTimer tm_streams(7);
      // An engine's own stream comes from a small per-device pool (a stream costs ~6 ms to create; 512 engines of a rank do not need
      // 512 of them: their per-engine calls are short and, inside a batch, run on the batch's stream anyway).  Side streams, fork /
      // join events and timing events are created when first used (ensure_side / ensure_timing).
      static std::mutex pool_mu; static std::map<int, std::vector<hipStream_t>> pool; static std::map<int, size_t> next;
      std::lock_guard<std::mutex> lk(pool_mu);
      std::vector<hipStream_t>& v = pool[h->device];
      if (h->tune.use_graph > 0) { HIP_OK(h, hipStreamCreate(&h->stream)); h->stream_pooled = false; }   // (a stream under graph capture must see no other engine's work)
      else {
          if (v.size() < 8) { hipStream_t q = nullptr; HIP_OK(h, hipStreamCreate(&q)); v.push_back(q); h->stream = q; }
          else h->stream = v[next[h->device]++ % v.size()];
          h->stream_pooled = true;
      }
    }
    h->cur = h->stream;
    if (h->tune.use_graph >= 0) h->use_graph = h->tune.use_graph != 0;
    if (h->tune.fast_step >= 0) h->fast_step = h->tune.fast_step != 0;
    if (h->tune.fuse_big >= 0) h->fuse_big = h->tune.fuse_big != 0;
    if (h->tune.graph_units > 0) h->graph_units = h->tune.graph_units;
    if (h->tune.graph_fork >= 0) h->graph_fork = h->tune.graph_fork != 0;
    try {
        for (int k = 0; k < 3; k++) { h->d_x[k].alloc(n); h->d_v[k].alloc(n); h->d_xbuild[k].alloc(n); h->d_x_sort[k].alloc(n); }
        h->d_stage.alloc((size_t)3 * n); h->d_xfer_out.alloc(4);
        h->d_mass.upload(h->T->mass); h->d_charge.upload(h->T->charge); h->d_sigma.upload(h->T->sigma); h->d_eps.upload(h->T->eps);
        h->d_flags.alloc(1); h->d_acc.alloc(1); h->d_ctrl.alloc(1); h->d_stamps.alloc(64); h->d_tab_ls.upload(h->tab_ls); h->d_tab_le.upload(h->tab_le); h->d_ftot.alloc((size_t)9 * n); h->d_alch_self.alloc(9 * 64);
        h->d_alch_orig.upload(h->alch); h->d_alch_local.upload(h->T->alch_local);
        // (the protocol-work trace: one slot per step of the switch; an MD integrator has no H step, no protocol work -- and an
        // nsteps of 2^30, which used to be 8 GiB of HBM per MD engine)
        h->d_trace.alloc(it->n_lambda_steps > 0 ? (size_t)std::max(1, h->nsteps) : (size_t)1); h->d_scratch.alloc((size_t)std::max(3 * n, 1024));
    } catch (std::string& e) { E_FAIL(h, "%s", e.c_str()); }
    {
        std::vector<int> mi(n, -1);
        for (size_t m = 0; m < h->mobile.size(); m++) mi[h->mobile[m]] = (int)m;
        try { h->d_mobile_atoms.upload(h->mobile); h->d_mobile_index.upload(mi); h->n_noise = std::max(1, h->n_O + h->n_L);
              h->d_noise.alloc((size_t)h->n_noise * 3 * std::max<size_t>(1, h->mobile.size())); } catch (std::string& e) { E_FAIL(h, "%s", e.c_str()); }
    }
    if (h->pme) {
        for (int i = 0; i < n; i++) if (h->T->mass[i] == 0.0 && h->T->qn[i] != 0.0) h->pme_frozen.push_back(i);
        try { h->d_qn.upload(h->T->qn); h->d_qn_full.upload(h->T->charge); h->d_pme_frozen.upload(h->pme_frozen); } catch (std::string& e) { E_FAIL(h, "%s", e.c_str()); }
        if (h->precision == 0 ? pme_tables<float>(h) : pme_tables<double>(h)) return 1;
    }
    if (build_clusters(h, s)) return 1;
    // (the fragments of kernels_frag.h are cut when a layout first asks for them: sort_and_tile)
    try { if (build_bonded(h, s)) return 1; } catch (std::string& e) { E_FAIL(h, "%s", e.c_str()); }
    return 0;
}

// Diagnostic: every block of the device slabs ends in 256 bytes nobody owns, zero since the slab was made.  A kernel that writes past the
// end of a buffer leaves a mark there.  out[0] blocks checked, out[1] guards with a mark, then for the first three: rounded size of the
// block in bytes, byte offset of the first marked word within the guard.
__global__ void k_check_guards(const unsigned* const* guards, int n, int* out /* [2 + 3 * 2] */) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= n) return;
    const unsigned w = guards[b][lane];
    const unsigned long long bal = __ballot(w != 0u);
    if (bal && lane == 0) {
        const int k = atomicAdd(&out[0], 1);
        if (k < 3) { out[2 + 2 * k] = b; out[3 + 2 * k] = 4 * (int)__builtin_ctzll(bal); }
    }
}
int blues_debug_check_guards(int64_t* out8) {
    if (!out8) return 2;
    for (int k = 0; k < 8; k++) out8[k] = 0;
    DevSlabs& S = dev_slabs();
    std::vector<const unsigned*> g; std::vector<size_t> sz;
    { std::lock_guard<std::mutex> lk(S.mu); for (auto& b : S.blocks) { g.push_back((const unsigned*)(b.first + b.second - 256)); sz.push_back(b.second); } }
    out8[0] = (int64_t)g.size();
    if (g.empty()) return 0;
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    const unsigned** d_g = nullptr; int* d_out = nullptr;
    if (hipMalloc((void**)&d_g, g.size() * sizeof(void*)) != hipSuccess || hipMalloc((void**)&d_out, 8 * sizeof(int)) != hipSuccess) return 1;
    hipMemcpy(d_g, g.data(), g.size() * sizeof(void*), hipMemcpyHostToDevice); hipMemset(d_out, 0, 8 * sizeof(int));
    hipLaunchKernelGGL(k_check_guards, dim3(((int)g.size() + 3) / 4), dim3(256), 0, nullptr, (const unsigned* const*)d_g, (int)g.size(), d_out);
    int h_out[8]; const bool ok = hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost) == hipSuccess;
    hipFree(d_g); hipFree(d_out);
    if (!ok) return 1;
    out8[1] = h_out[0];
    for (int k = 0; k < 3 && k < h_out[0]; k++) { out8[2 + 2 * k] = (int64_t)sz[h_out[2 + 2 * k]]; out8[3 + 2 * k] = h_out[3 + 2 * k]; }
    return 0;
}
int blues_debug_setup_seconds(double* out8) { if (!out8) return 2; for (int k = 0; k < 8; k++) out8[k] = g_setup_sec[k]; return 0; }

int blues_engine_create(const BluesSystemDesc* s, const BluesIntegratorDesc* it, int device, BluesEngine** out) {
    SetupTimer tm_create(0);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_create_error = "no HIP device available: the blues_amd engine has no CPU fallback"; return 2; }
    if (device < 0 || device >= ndev) { g_create_error = "invalid device ordinal"; return 2; }
    BluesEngine* h = new BluesEngine();
    h->device = device;
    int rc;
    try { rc = create_impl(h, s, it); } catch (std::string& e) { h->err = e; rc = 1; }
    if (rc) { g_create_error = h->err; delete h; return rc; }
    *out = h;
    return 0;
}

static int download_xyz(BluesEngine* h, double* xyz, DBuf<double>* src);

// re-derive the tile layout and launch decomposition from the positions currently on the device
static int relayout(BluesEngine* h) {
    if (!h->have_positions) { h->sorted_ok = false; return 0; }
    if (flush_program(h)) return 1;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    std::vector<double>().swap(h->hx);   // (sort_and_tile fetches the positions that are on the device now)
    h->sorted_ok = false;
    return sort_and_tile(h);
}

// A new layout for MANY members of a batch at once (a new shape for everybody: 5-10 ms of host work per member -- sort, exclusion rows,
// image, uploads -- i.e. 8 s for 1024 members one after the other, in the middle of a switch): the members are independent, so the
// host's cores share them.  prepare(m) sets what the new layout is to follow (shape, forbid_atom) and says whether m needs one.
// Host threads for work the members of a batch (or the chains of a rank at set-up) share: the cores this process may use, at most 16,
// or BluesTuning.host_threads where the caller knows better (N ranks on one host: cores / N -- blues_amd/replicas.py sets it).
static unsigned host_pool_threads(size_t items) {
    unsigned nthr = std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
#if defined(__linux__)
    { cpu_set_t set; CPU_ZERO(&set); if (sched_getaffinity(0, sizeof set, &set) == 0) nthr = std::min<unsigned>(nthr, (unsigned)std::max(1, CPU_COUNT(&set))); }
#endif
    if (g_tuning.host_threads > 0) nthr = (unsigned)g_tuning.host_threads;
    return (unsigned)std::max<size_t>(1, std::min<size_t>(nthr, items));
}
// Returns the first member whose re-layout failed (its err says why; relayout_failed is set on every such member), or null.  A
// failure of one member does not keep the others from their turn.
static BluesEngine* relayout_many(const std::vector<BluesEngine*>& members, const std::function<bool(BluesEngine*)>& prepare) {
    std::vector<BluesEngine*> todo;
    for (BluesEngine* m : members) { m->relayout_failed = false; if (prepare(m)) todo.push_back(m); }
    if (todo.empty()) return nullptr;
    const unsigned nthr = host_pool_threads(todo.size());
    std::atomic<size_t> next{0};
    auto work = [&]() {
        hipSetDevice(todo[0]->device);
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= todo.size()) return;
            if (relayout(todo[i])) { todo[i]->relayout_failed = true; todo[i]->sorted_ok = false; }
        }
    };
    if (nthr <= 1) work();
    else { std::vector<std::thread> pool; for (unsigned t = 0; t < nthr; t++) pool.emplace_back(work); for (auto& t : pool) t.join(); }
    for (BluesEngine* m : todo) if (m->relayout_failed) return m;
    return nullptr;
}

// A member of a batch moves to another number of tiles per group list WITHOUT a new sort: the sorted order, the image, the tiles, the
// exclusion rows and every per-slot table are independent of S; what changes is how many group lists there are and which tiles share
// one -- the device rebuilds them at the next force pass (lists_forced).  Possible where the buffers have room for the new number of
// lists (sort_and_tile allocates for the finest shape) and the capacity stays what it is.  Returns false if the member needs a full
// re-layout instead (relayout under shape_S / shape_jcap).
static bool reshape_possible(const BluesEngine* h, int S, int cap) {
    if (!h->sorted_ok || h->k1_mode != 2 || h->jcap != cap || !h->need_table_ok || h->forbid_atom) return false;
    const int nl = (std::max(1, h->n_itiles) + S - 1) / S;
    return nl <= h->lists_alloc && shape_index(S) >= 0;
}
static int reshape_groups(BluesEngine* h, int S, int cap) {
    const int q = shape_index(S);
    h->shape_S = S; h->shape_jcap = cap;
    h->shape_need = q >= 0 ? h->need_by_S[q] : 0.0;
    h->shape_overflow = h->shape_need * 1.1 + 64 > cap;
    S = std::min(S, std::max(1, h->n_itiles));
    if (h->S == S) return 0;   // (it has the shape already: only takes note of it)
    if (flush_program(h)) return 1;
    h->fin_pending = false;    // (the sums of a pass over the old lists are dropped with the pass)
    h->S = S;
    h->n_lists = (std::max(1, h->n_itiles) + S - 1) / S;
    h->n_tiles = h->n_lists + (h->alch.empty() ? 0 : 1);
    h->pool_cap = std::max(1, h->n_tiles) * MASK_QUOTA;
    h->lists_forced = true; h->pass_valid = false; h->graph_valid = false;
    h->args_epoch++;
    h->st_reshapes++;
    return 0;
}

// One layout shape for all members of a batch in the per-atom-list mode (k1_mode 2), where the number of tiles per list and the
// list capacity follow from the geometry: the leader's S (or the next smaller one whose lists fit the LDS), and a capacity
// that holds the largest member's lists with room for the i-atoms to spread.  fresh: start from the members' own choices.
// Round 6: the shape is PLANNED from the members' own tables (need_by_S: what each candidate shape would need where the member's last
// sort found its atoms) and the members move to it without a new sort (reshape_groups) -- a re-plan in the middle of a switch used to
// re-lay every member out up to three times (10 ms of host work each: 3 s for a batch of 1024 with the device idle and, with several
// batches taking turns, the other batches waiting); now it costs a few microseconds per member and one forced list build.
static int batch_plan_shape(BluesBatch* B, bool fresh) {
    if (B->eng.empty()) return 0;
    const auto t_begin = std::chrono::steady_clock::now();
    struct Clock { BluesBatch* B; std::chrono::steady_clock::time_point t0; ~Clock() { B->st_replan_sec += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } clock{B, t_begin};
    auto fail = [&](BluesEngine* m) { B->err = "re-layout for the batch failed: " + m->err; return 1; };
    auto relayout_counted = [&](const std::function<bool(BluesEngine*)>& prepare) -> BluesEngine* {
        return relayout_many(B->eng, [&](BluesEngine* m) { const bool need = prepare(m); if (need) B->st_relayouts++; return need; });
    };
    if (fresh) {
        // (a member that was already laid out on its own for a batch of this size -- BluesTuning.assume_batch -- keeps its layout:
        // re-deriving it would give the same one, 10 ms of host work per member; so does a member that went to fragment lists by itself
        // because no group shape holds its mobile atoms -- that layout does not depend on a shape, and it keeps the flag that says so.
        // The others are laid out on the host's cores together: one after the other, the 2048 engines of the S23k-solute line took 10 s)
        if (BluesEngine* bad = relayout_counted([&](BluesEngine* m) {
                const bool own_frag = m->sorted_ok && m->k1_mode == 3 && m->forbid_atom && m->layout_R == m->batch_R && m->shape_S == 0;
                const bool same = own_frag || (m->sorted_ok && m->layout_R == m->batch_R && m->shape_S == 0 && !m->forbid_atom);
                m->shape_S = 0; m->shape_jcap = 0; if (!own_frag) m->forbid_atom = false;
                return !same; })) return fail(bad);
    }
    BluesEngine* lead = B->eng[0];
    for (BluesEngine* m : B->eng) if (!m->straggler) { lead = m; break; }   // (a member in the batch's layout speaks for it)
    if (lead->k1_mode != 2) {   // the other modes size everything from the topology or the density: congruent by construction
        // (a re-plan in the middle of a run finds the leader here when ITS tiles fit no shape any more -- scattered mobile atoms: it is in
        // the sub-tile layout for want of a shape, and everybody, the leader included, moves on to fragment lists)
        if (BluesEngine* bad = relayout_counted([&](BluesEngine* m) {
                const bool need = m->k1_mode == 2 || (!fresh && !m->forbid_atom && m->k1_mode != 3);
                if (need) m->forbid_atom = true;   // (members that stay as they are keep their flags: their next re-sort must give the same mode again)
                m->straggler = false;              // (everybody is in one layout again)
                return need; })) return fail(bad);
        return 0;
    }
    const int lds_max = NB_JCAP_MAX;
    const double longest = std::max(lead->box[0], std::max(lead->box[1], lead->box[2]));
    int S = lead->S;
    for (int round = 0; round < 12; round++) {
        // the coarsest shape, from S downwards, that holds every member's lists with room to spread (factor 1.3, as a lone engine
        // chooses its own) -- from the members' tables; a member without one (another mode: it joined from a different layout) is laid
        // out under the candidate shape first
        int pick = 0, cap = 0;
        for (int q = 8; q >= 0 && !pick; q--) {
            const int c = SHAPE_CAND[q];
            if (c > S) continue;
            double need = 0.0, rad = 0.0; bool known = true;
            for (BluesEngine* m : B->eng) { if (m->straggler) continue; if (!m->need_table_ok || m->k1_mode != 2) { known = false; continue; } need = std::max(need, m->need_by_S[q]); rad = std::max(rad, m->rad_by_S[q]); }
            if (!known && need == 0.0) { pick = c; cap = lds_max; break; }   // (nobody has a table yet: the sweep below produces them)
            if (2.0 * rad + 1.0 + 0.6 >= longest) continue;                 // (the ghost record's place: sort_and_tile)
            if (need * 1.3 + 64 > lds_max) continue;
            pick = c; cap = need * 1.3 + 64 <= 3328 ? 3328 : lds_max;      // the two capacities of sort_and_tile
        }
        if (!pick) {   // not even one tile per list fits: fragment lists (or the bitmask kernels) for everybody
            if (BluesEngine* bad = relayout_counted([](BluesEngine* m) { m->forbid_atom = true; m->shape_S = 0; m->shape_jcap = 0; m->straggler = false; return true; })) return fail(bad);
            return 0;
        }
        // members that can take the shape as they are move to it in place; the others (another capacity, another mode, no layout)
        // are laid out under it, on the host's cores
        if (BluesEngine* bad = relayout_counted([&](BluesEngine* m) {
                if (m->straggler) return false;      // (it keeps its own layout while the others change shape)
                if (reshape_possible(m, pick, cap)) return false;
                m->shape_S = pick; m->shape_jcap = cap; return true; })) return fail(bad);
        bool again = false;
        for (BluesEngine* m : B->eng) {
            if (m->straggler) continue;
            if (m->k1_mode != 2) { again = true; continue; }            // (its own geometry fits no group shape under this S: a finer one)
            if (m->S != pick || m->shape_S != pick || m->shape_jcap != cap) { if (reshape_groups(m, pick, cap)) return fail(m); }
            again |= m->shape_overflow || m->jcap != cap;
        }
        if (!again) return 0;
        // somebody's lists do not fit after all (a table older than its atoms' spread, or the quantised capacity): the next finer shape
        int next = 0;
        for (int q = 8; q >= 0; q--) if (SHAPE_CAND[q] < pick) { next = SHAPE_CAND[q]; break; }
        if (!next) {
            if (BluesEngine* bad = relayout_counted([](BluesEngine* m) { m->forbid_atom = true; m->shape_S = 0; m->shape_jcap = 0; return true; })) return fail(bad);
            return 0;
        }
        S = next;
    }
    B->err = "internal: the batch's layout shape did not settle";
    return 1;
}

static void batch_detach_all(BluesBatch* B) {
    if (!B) return;
    batch_leave(B);
    for (BluesEngine* m : B->eng) if (m) {
        hipStreamSynchronize(m->stream);
        m->batch = nullptr; m->batch_index = -1; m->batch_R = m->tune.assume_batch > 0 ? m->tune.assume_batch : 1;
        m->shape_S = 0; m->shape_jcap = 0; m->forbid_atom = false; m->pin_jcap = m->pin_seg = m->pin_nw = m->pin_wpb = 0;
        // its own layout again -- when it is next used (ensure_sorted: the layout fetches the positions from the device itself).  Laid out
        // here and now, the end of a run re-sorted every member of every batch, 5 ms each, moments before destroying it.
        if (m->have_positions) { flush_program(m); m->straggler = false; m->sorted_ok = false; m->pass_valid = false; m->fin_pending = false; }
    }
    B->eng.clear(); B->leader = nullptr; B->lockstep = false;
    if (B->stream) { hipStreamDestroy(B->stream); B->stream = nullptr; }
}

int blues_engine_destroy(BluesEngine* h) {
    if (!h) return 0;
    hipSetDevice(h->device);
    if (h->batch) { if (h->batch->stream) hipStreamSynchronize(h->batch->stream); batch_detach_all(h->batch); }  // a batch does not outlive any of its members (and none of its launches the member's buffers)
    hipStreamSynchronize(h->stream);
    for (hipStream_t q : {h->s1, h->s2}) if (q) hipStreamSynchronize(q);
    if (h->sw_saved) { blues_snapshot_release(h->sw_saved); h->sw_saved = nullptr; }
    if (h->h_xfer) { hipHostFree(h->h_xfer); h->h_xfer = nullptr; }
    if (h->h_edit) { hipHostFree(h->h_edit); h->h_edit = nullptr; }
    if (h->ev_edit) { hipEventDestroy(h->ev_edit); h->ev_edit = nullptr; }
#ifdef BLUES_STAMP
    { std::vector<long long> st; h->d_stamps.download(st); static int printed = 0;
      if (printed++ < 2) { fprintf(stderr, "[stamps] last integrate launch, cycles per op:"); for (int i = 1; i < 40 && st[i] > 0; i++) fprintf(stderr, " %lld", st[i] - st[i - 1]); fprintf(stderr, "\n");
        { long long gs[64]; hipMemcpyFromSymbol(gs, HIP_SYMBOL(g_nb_stamps), sizeof gs); fprintf(stderr, "[stamps] dense alchemical kernel, block 0 thread 0 (cycles since entry: staged; pass 1: begun, marked, pairs done; pass 2: begun, marked, pairs done; all passes, end):"); for (int i = 33; i < 42; i++) fprintf(stderr, " %lld", gs[i] - gs[32]); fprintf(stderr, "\n"); }
        fprintf(stderr, "[stamps] last step_default launch, thread 0 (10 ns ticks since entry: gathers issued, alchemical sums, V+rattle, CM, V+rattle, R+shake+rattle, O+rattle, R+shake+rattle, written):"); for (int i = 1; i < 10; i++) fprintf(stderr, " %lld", st[40 + i] - st[40]); fprintf(stderr, " | top: records asked for, entries staged, atom 0, 1, 2, 3 begun:"); for (int i = 10; i < 16; i++) fprintf(stderr, " %lld", st[40 + i] - st[40]); fprintf(stderr, "\n"); } }
#endif
    delete h;
    return 0;
}

static int upload_xyz(BluesEngine* h, const double* xyz, DBuf<double>* dst) {   // one transfer, device-side de-interleave
    HIP_OK(h, hipMemcpyAsync(h->d_stage.p, xyz, sizeof(double) * 3 * h->n, hipMemcpyHostToDevice, h->stream));
    Copy6Args c; memset(&c, 0, sizeof c);
    c.n = h->n; c.count = 3; c.src_stride = 3; c.dst_stride = 1;
    for (int k = 0; k < 3; k++) { c.src[k] = h->d_stage.p + k; c.dst[k] = dst[k].p; }
    hipLaunchKernelGGL(k_copy_arrays, dim3((h->n + 255) / 256), dim3(256), 0, h->stream, c);
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}
static int download_xyz(BluesEngine* h, double* xyz, DBuf<double>* src) {   // device-side interleave, one transfer
    Copy6Args c; memset(&c, 0, sizeof c);
    c.n = h->n; c.count = 3; c.src_stride = 1; c.dst_stride = 3;
    for (int k = 0; k < 3; k++) { c.src[k] = src[k].p; c.dst[k] = h->d_stage.p + k; }
    hipLaunchKernelGGL(k_copy_arrays, dim3((h->n + 255) / 256), dim3(256), 0, h->stream, c);
    HIP_OK(h, hipMemcpyAsync(xyz, h->d_stage.p, sizeof(double) * 3 * h->n, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// Shared tail of setPositions: the new coordinates are already on the device (src, SoA stride 1 or interleaved stride 3).
static int load_positions(BluesEngine* h, const double* const src[3], int stride, int n_edit) {
    LoadPosArgs a; memset(&a, 0, sizeof a);
    a.n = h->n; a.stride = stride;
    for (int k = 0; k < 3; k++) { a.src[k] = src[k]; a.x[k] = h->d_x[k].p; a.x_sort[k] = h->d_x_sort[k].p; }
    a.mass = h->d_mass.p; a.alch_local = h->d_alch_local.p; a.sorted_of_orig = h->d_sorted_of_orig.p;
    a.img_f = h->precision == 0 ? h->d_img_f.p : nullptr; a.img_d = h->precision == 0 ? nullptr : h->d_img_d.p;
    a.box = make_box(h); a.out = h->d_xfer_out.p;
    // (a verdict still pending from an earlier load is kept: the kernels only OR / max / add into it)
    if (!h->xfer_pending) HIP_OK(h, hipMemsetAsync(h->d_xfer_out.p, 0, 4 * sizeof(unsigned), h->stream));
    else if (h->xfer_stream != h->stream) HIP_OK(h, hipStreamSynchronize(h->xfer_stream));
    hipLaunchKernelGGL(k_load_positions, dim3((h->n + 255) / 256), dim3(256), 0, h->stream, a);
    h->st_launches++;
    if (n_edit > 0) {   // the edit list is already in d_edit_idx / d_edit_xyz
        EditPosArgs e; memset(&e, 0, sizeof e);
        e.n_edit = n_edit; e.idx = h->d_edit_idx.p; e.xyz = h->d_edit_xyz.p;
        for (int k = 0; k < 3; k++) { e.x[k] = h->d_x[k].p; e.x_sort[k] = h->d_x_sort[k].p; }
        e.mass = a.mass; e.alch_local = a.alch_local; e.sorted_of_orig = a.sorted_of_orig; e.img_f = a.img_f; e.img_d = a.img_d; e.box = a.box; e.out = a.out;
        hipLaunchKernelGGL(k_edit_positions, dim3((n_edit + 63) / 64), dim3(64), 0, h->stream, e);
        h->st_launches++;
    }
    if (!h->h_xfer && hipHostMalloc((void**)&h->h_xfer, 4 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) E_FAIL(h, "hipHostMalloc failed");
    HIP_OK(h, hipMemcpyAsync(h->h_xfer, h->d_xfer_out.p, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    h->xfer_pending = true; h->xfer_stream = h->stream; h->xfer_src = h->h_xfer;
    h->have_positions = true; h->x_edited = true; h->pass_valid = false; h->fin_pending = false; h->ecache.clear();
    h->lists_forced = true;
    return 0;
}

// work bookkeeping for instantaneous moves: remember U(x_old, lambda) before overwriting (integrators.py:205)
static int before_position_edit(BluesEngine* h) {
    if (flush_program(h)) return 1;
    if (h->switch_mode == BLUES_SWITCH_NONE && h->have_positions && h->h_first_step >= 1 && h->h_step > 0 && h->h_step < h->nsteps && !h->unpert_valid) {
        double E;
        if (total_energy(h, &E)) return 1;
        h->h_unperturbed = E; h->unpert_valid = true;
    }
    return 0;
}

int blues_set_positions(BluesEngine* h, const double* xyz, int32_t n_atoms) {
    if (n_atoms != h->n) E_FAIL(h, "expected %d atoms, got %d", h->n, n_atoms);
    HIP_OK(h, hipSetDevice(h->device));
    if (before_position_edit(h)) return 1;
    // store every constraint cluster as one whole periodic image (a lattice translation of single atoms is
    // physically a no-op); the cluster solves in k_integrate then need no minimum-image arithmetic
    static thread_local std::vector<double> st;   // (staging of one host transfer: per thread, not per engine -- 560 KB x 2048 chains otherwise)
    st.assign(xyz, xyz + 3 * (size_t)h->n);
    for (const HostCluster& c : h->clusters) for (int a = 1; a < 4; a++) if (c.atoms[a] >= 0)
        for (int k = 0; k < 3; k++) {
            double d = st[3 * c.atoms[a] + k] - st[3 * c.atoms[0] + k];
            st[3 * c.atoms[a] + k] -= h->box[k] * std::nearbyint(d / h->box[k]);
        }
    if (!h->sorted_ok) {   // first positions (or a box change pending): lay the tiles out from these coordinates
        HIP_OK(h, hipStreamSynchronize(h->stream));
        h->hx = st;
        if (upload_xyz(h, h->hx.data(), h->d_x)) return 1;
        h->have_positions = true; h->x_edited = true; h->pass_valid = false; h->fin_pending = false; h->ecache.clear(); h->e_frozen_valid = false; h->pme_static_valid = false;
        return sort_and_tile(h);
    }
    // one interleaved transfer; de-interleaving, image refresh and the re-sort statistics happen on the device
    HIP_OK(h, hipMemcpyAsync(h->d_stage.p, st.data(), sizeof(double) * 3 * h->n, hipMemcpyHostToDevice, h->stream));
    const double* src[3] = {h->d_stage.p, h->d_stage.p + 1, h->d_stage.p + 2};
    return load_positions(h, src, 3, 0);
}

int blues_set_velocities(BluesEngine* h, const double* xyz, int32_t n_atoms) {
    if (n_atoms != h->n) E_FAIL(h, "expected %d atoms, got %d", h->n, n_atoms);
    HIP_OK(h, hipSetDevice(h->device));
    if (flush_program(h)) return 1;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    h->vel_clean = false; h->ke_cache_valid = false;
    return upload_xyz(h, xyz, h->d_v);
}

int blues_set_box(BluesEngine* h, const double box[9]) {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) if (r != c && box[3 * r + c] != 0.0) E_FAIL(h, "only orthorhombic boxes are supported");
    if (box[0] == h->box[0] && box[4] == h->box[1] && box[8] == h->box[2]) return 0;
    HIP_OK(h, hipSetDevice(h->device));
    if (flush_program(h)) return 1;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < 3; k++) if (box[4 * k] < 2.0 * h->cutoff) E_FAIL(h, "box edge %g < 2*cutoff", box[4 * k]);   // (before anything is changed: a refused box leaves the engine as it was)
    h->box[0] = box[0]; h->box[1] = box[4]; h->box[2] = box[8];
    h->box_epoch++;
    derive_margins(h);   // (the layout re-derives the skin for the new box)
    h->sorted_ok = false; h->pass_valid = false; h->fin_pending = false; h->e_frozen_valid = false; h->ecache.clear();
    if (h->pme && (h->precision == 0 ? pme_tables<float>(h) : pme_tables<double>(h))) return 1;
    if (h->have_positions) { std::vector<double>().swap(h->hx); return sort_and_tile(h); }
    return 0;
}

int blues_get_positions(BluesEngine* h, double* xyz, int32_t n_atoms) {
    if (n_atoms != h->n) E_FAIL(h, "expected %d atoms, got %d", h->n, n_atoms);
    HIP_OK(h, hipSetDevice(h->device));
    if (flush_program(h)) return 1;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return download_xyz(h, xyz, h->d_x);
}
int blues_get_velocities(BluesEngine* h, double* xyz, int32_t n_atoms) {
    if (n_atoms != h->n) E_FAIL(h, "expected %d atoms, got %d", h->n, n_atoms);
    HIP_OK(h, hipSetDevice(h->device));
    if (flush_program(h)) return 1;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return download_xyz(h, xyz, h->d_v);
}
int blues_get_box(BluesEngine* h, double box[9]) {
    for (int i = 0; i < 9; i++) box[i] = 0.0;
    box[0] = h->box[0]; box[4] = h->box[1]; box[8] = h->box[2];
    return 0;
}

int blues_get_forces(BluesEngine* h, double* out, int32_t n_atoms) {
    if (n_atoms != h->n) E_FAIL(h, "expected %d atoms, got %d", h->n, n_atoms);
    HIP_OK(h, hipSetDevice(h->device));
    if (flush_program(h)) return 1;
    h->fin_pending = false;   // (this call runs a pass and the sums of its own)
    // forces at the current alchemical parameters: evaluate a pass whose slot 0 carries them
    if (ensure_sorted(h)) return 1;
    double ls[3] = {h->cur_ls, h->cur_ls, h->cur_ls}, le[3] = {h->cur_le, h->cur_le, h->cur_le};
    int rc = h->precision == 0 ? launch_lists<float>(h, h->lists_forced) : launch_lists<double>(h, h->lists_forced);
    h->lists_forced = false;
    if (rc) return 1;
    h->pass_valid = false;
    if (launch_alchemical(h, ls, le, 1)) return 1;
    rc = h->precision == 0 ? launch_nonbonded<float, false>(h) : launch_nonbonded<double, false>(h);
    if (rc) return 1;
    if (launch_pme(h, 0)) return 1;
    if (launch_bonded_and_finalize(h, le, false)) return 1;
    IntArgs A = make_int_args(h);
    DBuf<double> tmp;
    try { tmp.alloc((size_t)3 * h->n); } catch (std::string& e) { E_FAIL(h, "%s", e.c_str()); }
    hipLaunchKernelGGL(k_gather_forces, dim3((h->n + 255) / 256), dim3(256), 0, h->stream, A, 0, tmp.p);
    h->st_launches++;
    if (check_flags(h)) return 1;
    HIP_OK(h, hipMemcpy(out, tmp.p, sizeof(double) * 3 * h->n, hipMemcpyDeviceToHost));
#ifdef BLUES_STAMP
    if (h->pme) { long long st[32]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_nb_stamps), sizeof st); fprintf(stderr, "[stamps] pme_fast (cycles since start: cache+region, spread, to X, forward, eterm, backward, gather):"); for (int i = 10; i < 17; i++) fprintf(stderr, " %lld", st[i] - st[9]); fprintf(stderr, "\n"); }
#endif
    return 0;
}

int blues_set_velocities_to_temperature(BluesEngine* h, double temperature, uint64_t seed) {
    HIP_OK(h, hipSetDevice(h->device));
    if (flush_program(h)) return 1;
    hipLaunchKernelGGL(k_maxwell, dim3((h->n + 255) / 256), dim3(256), 0, h->stream, h->n, h->d_mass.p, h->d_v[0].p, h->d_v[1].p, h->d_v[2].p, KB_KJ * temperature, (unsigned long long)seed, (unsigned)h->replica * 4u + 1u);
    h->st_launches++;
    h->vel_clean = false; h->ke_cache_valid = false;
    if (emit(h, OP_RATTLE)) return 1;
    // (no read-back here: a velocity-constraint failure stays flagged on the device and is raised by the next step, which
    // checks the flags anyway -- a synchronous check per chain was 20 us of host time for each of R chains per iteration)
    return flush_program(h);
}

int blues_get_energy(BluesEngine* h, double* potential, double* kinetic) {
    HIP_OK(h, hipSetDevice(h->device));
    if (potential) { if (total_energy(h, potential)) return 1; }
    if (kinetic) {
        if (flush_program(h)) return 1;
        if (h->ke_cache_valid) { *kinetic = h->ke_cache; return 0; }
        const int nb = (h->n + 255) / 256;
        hipLaunchKernelGGL(k_kinetic, dim3(nb), dim3(256), 0, h->stream, h->n, h->d_mass.p, h->d_v[0].p, h->d_v[1].p, h->d_v[2].p, h->d_scratch.p);
        h->st_launches++;
        HIP_OK(h, hipStreamSynchronize(h->stream));
        std::vector<double> part(nb);
        HIP_OK(h, hipMemcpy(part.data(), h->d_scratch.p, sizeof(double) * nb, hipMemcpyDeviceToHost));
        double ke = 0.0; for (double p : part) ke += p;
        *kinetic = ke; h->ke_cache = ke; h->ke_cache_valid = true;
    }
    return 0;
}

int blues_get_energy_at(BluesEngine* h, double lambda_sterics, double lambda_electrostatics, double* potential) {
    if (!potential) E_FAIL(h, "blues_get_energy_at: null result pointer");
    HIP_OK(h, hipSetDevice(h->device));
    if (flush_program(h)) return 1;
    if (h->ecache.find(lambda_sterics, lambda_electrostatics, potential)) return 0;
    const double ls = h->cur_ls, le = h->cur_le;
    h->cur_ls = lambda_sterics; h->cur_le = lambda_electrostatics;
    const int rc = total_energy(h, potential);
    h->cur_ls = ls; h->cur_le = le;
    h->pass_valid = false;   // (the slabs now hold another parameter set)
    return rc;
}

int blues_get_energy_terms(BluesEngine* h, double terms[BLUES_N_ENERGY_TERMS]) {
    HIP_OK(h, hipSetDevice(h->device));
    return energy_terms(h, terms);
}

int blues_step(BluesEngine* h, int32_t n_steps) {
    HIP_OK(h, hipSetDevice(h->device));
    h->tracing = false;
    if (do_steps(h, n_steps)) return 1;
    return check_flags(h);
}

int blues_run_switch(BluesEngine* h, int32_t n_steps, double* work_trace) {
    HIP_OK(h, hipSetDevice(h->device));
    const int first = h->h_step;
    const bool has_work = h->n_H > 0;   // (no H step: the protocol work stays 0, nothing is traced)
    h->tracing = work_trace != nullptr && has_work;
    int rc = do_steps(h, n_steps);
    h->tracing = false;
    if (rc) return 1;
    if (check_flags(h)) return 1;
    if (work_trace && !has_work) { for (int i = 0; i < n_steps; i++) work_trace[i] = 0.0; }
    else if (work_trace) {
        const int done = h->h_step - first;
        if (done > 0) HIP_OK(h, hipMemcpy(work_trace, h->d_trace.p + first, sizeof(double) * done, hipMemcpyDeviceToHost));
        for (int i = std::max(done, 0); i < n_steps; i++) work_trace[i] = done > 0 ? work_trace[done - 1] : 0.0;
    }
    return 0;
}

static int read_acc(BluesEngine* h, DevAccum* a) {
    if (flush_program(h)) return 1;
    if (h->acc_cache_valid && h->acc_cache_stamp == h->st_launches) { *a = h->acc_cache; return 0; }   // nothing was launched since a batch prefetch read them
    HIP_OK(h, hipStreamSynchronize(h->stream));
    HIP_OK(h, hipMemcpy(a, h->d_acc.p, sizeof *a, hipMemcpyDeviceToHost));
    return 0;
}

int blues_get_global(BluesEngine* h, const char* name, double* value) {
    // the host-side mirrors BLUES asks for most come first and need no device call
    if (!strcmp(name, "lambda_sterics")) { *value = h->cur_ls; return 0; }
    if (!strcmp(name, "lambda_electrostatics")) { *value = h->cur_le; return 0; }
    HIP_OK(h, hipSetDevice(h->device));
    std::string k(name);
    DevAccum a;
    if (k == "lambda") *value = h->h_lambda;
    else if (k == "step") *value = h->h_step;
    else if (k == "lambda_step") *value = h->h_lambda_step;
    else if (k == "n_lambda_steps") *value = h->n_lambda;
    else if (k == "nsteps") *value = h->nsteps;
    else if (k == "protocol_work") { if (read_acc(h, &a)) return 1; *value = a.protocol_work; }
    else if (k == "shadow_work") *value = h->switch_mode != BLUES_SWITCH_NONE ? h->sw_shadow : 0.0;
    else if (k == "perturbed_pe") *value = h->h_perturbed;
    else if (k == "unperturbed_pe") *value = h->h_unperturbed;
    else if (k == "first_step") *value = h->h_first_step;
    else if (k == "nprop") *value = h->nprop;
    else if (k == "prop") *value = h->h_prop;
    else if (k == "prop_lambda_min") *value = h->prop_min;
    else if (k == "prop_lambda_max") *value = h->prop_max;
    else if (k == "Enew" || k == "Eold") {
        double E; if (total_energy(h, &E)) return 1;
        if (read_acc(h, &a)) return 1;
        *value = (k == "Enew") ? E : E - a.dE_last;
    }
    else if (k == "heat") *value = 0.0;
    else if (k == "kT") *value = h->kT;
    else if (h->switch_mode != BLUES_SWITCH_NONE && (k == "total_work" || k == "Epert" || k == "initial_energy" || k == "final_energy" || k == "accept" || k == "naccept" || k == "ntrials" || k == "psteps")) {
        // globals of reference blues/switching.py:1062-1080, energies in kJ/mol
        if (k == "total_work") { if (read_acc(h, &a)) return 1; *value = a.protocol_work + h->sw_shadow; }
        else if (k == "Epert") *value = h->sw_Epert;
        else if (k == "initial_energy") *value = h->sw_Einit;
        else if (k == "final_energy") { if (h->h_step == 0) *value = 0.0; else if (h->psteps > 0) *value = h->sw_Epert; else if (total_energy(h, value)) return 1; }
        else if (k == "accept") *value = h->sw_accept;
        else if (k == "naccept") *value = h->sw_naccept;
        else if (k == "ntrials") *value = h->sw_ntrials;
        else *value = h->psteps;
    }
    else if (k == "lambda_sterics") *value = h->cur_ls;
    else if (k == "lambda_electrostatics") *value = h->cur_le;
    else E_FAIL(h, "unknown global variable '%s'", name);
    return 0;
}

int blues_set_global(BluesEngine* h, const char* name, double value) {
    HIP_OK(h, hipSetDevice(h->device));
    std::string k(name);
    if (k == "protocol_work") {
        DevAccum a; if (read_acc(h, &a)) return 1;
        a.protocol_work = value;
        HIP_OK(h, hipMemcpy(h->d_acc.p, &a, sizeof a, hipMemcpyHostToDevice));
        h->acc_cache_valid = false;
    }
    else if (k == "step") h->h_step = (int)value;
    else if (k == "lambda") h->h_lambda = value;
    else if (k == "lambda_step") { h->h_lambda_step = (int)value; h->pass_valid = false; h->fin_pending = false; }
    else if (k == "first_step") h->h_first_step = (int)value;
    else if (k == "prop") h->h_prop = (int)value;
    else if (k == "nprop") h->nprop = (int)value;
    else if (k == "perturbed_pe") h->h_perturbed = value;
    else if (k == "unperturbed_pe") { h->h_unperturbed = value; }
    else if (k == "shadow_work") { if (h->switch_mode != BLUES_SWITCH_NONE) h->sw_shadow = value; else if (value != 0.0) E_FAIL(h, "shadow work is not measured (measure_shadow_work=False)"); }
    else if (h->switch_mode != BLUES_SWITCH_NONE && (k == "total_work" || k == "naccept" || k == "ntrials")) {
        if (k == "naccept") h->sw_naccept = (int)value; else if (k == "ntrials") h->sw_ntrials = (int)value;   // (total_work is derived)
    }
    else if (k == "lambda_sterics") { h->cur_ls = value; h->pass_valid = false; h->fin_pending = false; }
    else if (k == "lambda_electrostatics") { h->cur_le = value; h->pass_valid = false; h->fin_pending = false; }
    else E_FAIL(h, "global variable '%s' cannot be set", name);
    return 0;
}

int blues_reset(BluesEngine* h) {
    HIP_OK(h, hipSetDevice(h->device));
    if (flush_program(h)) return 1;
    h->h_step = 0; h->h_lambda = 0.0; h->h_first_step = 0; h->h_perturbed = 0.0; h->h_unperturbed = 0.0; h->h_prop = 1; h->h_lambda_step = 0;
    h->unpert_valid = false; h->x_edited = false; h->pass_valid = false; h->fin_pending = false;
    HIP_OK(h, hipMemsetAsync(h->d_acc.p, 0, sizeof(DevAccum), h->stream));
    h->acc_cache_valid = false;
    h->sw_shadow = 0.0; h->sw_Einit = 0.0; h->sw_Efinal = 0.0; h->sw_naccept = 0; h->sw_ntrials = 0; h->sw_accept = 0;   // switching.py:1023-1036
    return 0;
}

int blues_get_stats(BluesEngine* h, int64_t stats[BLUES_N_STATS]) {
    for (int i = 0; i < BLUES_N_STATS; i++) stats[i] = 0;
    stats[9] = h->st_resorts; stats[11] = h->st_energy_evals;
    if (h->d_jcount.p && h->sorted_ok) { std::vector<int> jc; hipSetDevice(h->device); hipStreamSynchronize(h->stream); try { h->d_jcount.download(jc); for (int q = 0; q < (int)jc.size() && q < std::max(1, h->n_tiles); q++) stats[8] = std::max<int64_t>(stats[8], jc[q]); } catch (std::string&) {} }
    if (h->tune.debug_lists && h->k1_mode == 2 && h->d_acount.p) {
        std::vector<int> ac; h->d_acount.download(ac); std::vector<int> jc2; h->d_jcount.download(jc2);
        fprintf(stderr, "[lists] S=%d n_lists=%d jcap=%d acap=%d jcount:", h->S, h->n_lists, h->jcap, h->acap); for (int c : jc2) fprintf(stderr, " %d", c);
        fprintf(stderr, "\n[lists] acount:"); for (size_t i = 0; i < ac.size(); i++) fprintf(stderr, " %d", ac[i]); fprintf(stderr, "\n");
    }
    stats[12] = h->k1_mode; stats[13] = h->S;
    if (h->k1_mode == 2 && h->d_acount.p && h->sorted_ok) {
        std::vector<int> ac; hipSetDevice(h->device); hipStreamSynchronize(h->stream);
        try { h->d_acount.download(ac); for (int c : ac) { stats[14] += c; stats[15] += (c + 63) / 64; } } catch (std::string&) {}
    }
    if (h->k1_mode == 3 && h->d_icount.p && h->sorted_ok) {   // fragment lists: [14] / [15] entries and 64-entry chunks of the outer lists, [17] / [18] of the inner ones (the lists the kernel walks)
        std::vector<int> ic, oc; hipSetDevice(h->device); hipStreamSynchronize(h->stream);
        try { h->d_ocount.download(oc); h->d_icount.download(ic); for (int c : oc) { stats[14] += c; stats[15] += (c + 63) / 64; } for (int c : ic) { stats[17] += c; stats[18] += (c + 63) / 64; } } catch (std::string&) {}
    }
    stats[0] = h->st_passes; stats[2] = h->st_launches; stats[3] = h->n_itiles; stats[4] = (int64_t)h->clusters.size(); stats[5] = h->jcap; stats[6] = h->npart; stats[7] = h->seg_len * 1000 + h->wpb;
    if (h->d_flags.p) { DevFlags f; hipSetDevice(h->device); hipStreamSynchronize(h->stream); if (hipMemcpy(&f, h->d_flags.p, sizeof f, hipMemcpyDeviceToHost) == hipSuccess) { stats[1] = f.list_gen; stats[10] = f.builds; stats[16] = f.prunes; } }
    stats[19] = (h->prune_on && h->k1_mode == 2) || (h->k1_mode == 3 && h->frag_m < h->skin);
    stats[21] = h->int_threads;
    stats[20] = h->k2_dense ? (h->k2_f32 ? 1 : 2) : 0;   // (1: the dense form in fp32 pair arithmetic, round 6; 2: its fp64 predecessor, BluesTuning.k2_dense = 2)
    if (stats[19] && h->k1_mode == 2 && h->d_pcount.p && h->sorted_ok) {
        std::vector<int> pc, ta;
        try { h->d_pcount.download(pc); h->d_tile_atoms.download(ta); for (size_t q = 0; q < pc.size() && q < ta.size(); q++) if (ta[q] >= 0) { stats[17] += pc[q]; stats[18] += (pc[q] + 63) / 64; } } catch (std::string&) {}
    }
    return 0;
}

int blues_time_nonbonded(BluesEngine* h, int32_t reps, double* usec) {
    HIP_OK(h, hipSetDevice(h->device));
    if (flush_program(h)) return 1;
    if (ensure_sorted(h)) return 1;
    int rc = h->precision == 0 ? launch_lists<float>(h, 1) : launch_lists<double>(h, 1);
    if (rc) return 1;
    h->pass_valid = false;
    // (with pruned lists: the first launch prunes, the acknowledged state is what the timed launches see -- the steady state
    // between two prunes; blues_batch_time_nonbonded_modes times both kinds of pass)
    for (int w = 0; w < 3; w++) { rc = h->precision == 0 ? launch_nonbonded<float, false>(h) : launch_nonbonded<double, false>(h); if (rc) return 1; }
    if (h->prune_on && h->k1_mode == 2) hipLaunchKernelGGL(k_prune_set, dim3(1), dim3(256), 0, h->stream, (const RepCore*)nullptr, h->d_pneed.p, h->n_islots, 0);
    if (ensure_timing(h)) return 1;
    HIP_OK(h, hipEventRecord(h->ev0, h->stream));
    for (int r = 0; r < reps; r++) { rc = h->precision == 0 ? launch_nonbonded<float, false>(h) : launch_nonbonded<double, false>(h); if (rc) return 1; }
    HIP_OK(h, hipEventRecord(h->ev1, h->stream));
    HIP_OK(h, hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_OK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *usec = 1000.0 * ms / std::max(1, reps);
#ifdef BLUES_STAMP
    { long long st[40]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_nb_stamps), sizeof(long long) * 32); fprintf(stderr, "[stamps] nonbonded_atom (cycles since start):"); for (int i = 17; i < 28; i++) fprintf(stderr, " %lld", st[i] - st[16]); fprintf(stderr, "\n");
      fprintf(stderr, "[stamps] last kernel that stamped 0..7 (pme_fast: cache+region, spread, to X, fwd, eterm, bwd, gather):"); for (int i = 10; i < 17; i++) fprintf(stderr, " %lld", st[i] - st[9]); fprintf(stderr, "\n"); }
#endif
    return check_flags(h);
}

// Audit of the per-atom lists at the current positions (kernels_nb.h: k_audit_atom_lists): out[0] = pairs within the cutoff that
// the nonbonded kernel is responsible for, out[1] = how many of them are in no list (must be 0).  Lists that are due for a
// rebuild are rebuilt first, as the next force pass would.
int blues_audit_lists(BluesEngine* h, int64_t* out) {
    HIP_OK(h, hipSetDevice(h->device));
    if (flush_program(h)) return 1;
    if (ensure_sorted(h)) return 1;
    if ((h->k1_mode != 2 && h->k1_mode != 3) || h->precision != 0) E_FAIL(h, "blues_audit_lists: the engine is not in per-atom-list or fragment-list mode");
    if (h->k1_mode == 2 && h->jcap > 8192) E_FAIL(h, "blues_audit_lists: list capacity %d is beyond the audit kernel's bitmap", h->jcap);
    if (h->k1_mode == 3 && (size_t)3 * ((h->n + 31) / 32) * 4 > 64 * 1024) E_FAIL(h, "blues_audit_lists: %d atoms are beyond the fragment audit's bitmap", h->n);
    hipStream_t st = h->batch && h->batch->entered ? h->batch->stream : h->stream;
    hipStream_t keep = h->cur; h->cur = st;
    BluesBatch* B = h->batch; h->batch = nullptr;   // (the member's own rebuild kernels, whoever leads the batch)
    int rc = launch_lists<float>(h, h->lists_forced);
    h->batch = B; h->cur = keep;
    if (rc) return 1;
    h->lists_forced = false;
    DBuf<unsigned long long> d_out; d_out.alloc(2);
    if (h->k1_mode == 3) hipLaunchKernelGGL(k_audit_frag_lists, dim3(h->frag_NI), dim3(256), (size_t)3 * ((h->n + 31) / 32) * 4, st, make_frag_args(h), make_list_args(h), make_nbconst<float>(h), h->d_img_f.p, d_out.p);
    else hipLaunchKernelGGL(k_audit_atom_lists, dim3(h->n_islots), dim3(256), 0, st, make_nb_args<float>(h), make_list_args(h), make_nbconst<float>(h), h->d_img_f.p, d_out.p);
    HIP_OK(h, hipGetLastError());
    HIP_OK(h, hipStreamSynchronize(st));
    std::vector<unsigned long long> v; d_out.download(v); d_out.release();
    out[0] = (int64_t)v[0]; out[1] = (int64_t)v[1];
    return check_flags(h);
}

// forced neighbour-list rebuild of one engine, timed alone with HIP events (diagnostic: the latency every batched round
// pays whenever some member rebuilds)
int blues_time_list_build(BluesEngine* h, int32_t reps, double* usec) {
    HIP_OK(h, hipSetDevice(h->device));
    if (flush_program(h)) return 1;
    if (ensure_sorted(h)) return 1;
    int rc = 0;
    for (int w = 0; w < 2 && !rc; w++) rc = h->precision == 0 ? launch_lists<float>(h, 1) : launch_lists<double>(h, 1);
    if (rc) return 1;
    if (ensure_timing(h)) return 1;
    HIP_OK(h, hipEventRecord(h->ev0, h->stream));
    for (int r = 0; r < reps && !rc; r++) rc = h->precision == 0 ? launch_lists<float>(h, 1) : launch_lists<double>(h, 1);
    if (rc) return 1;
    HIP_OK(h, hipEventRecord(h->ev1, h->stream));
    HIP_OK(h, hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_OK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *usec = 1000.0 * ms / std::max(1, reps);
    h->pass_valid = false; h->lists_forced = false;
#ifdef BLUES_STAMP
    { long long st[32]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_nb_stamps), sizeof st);
      fprintf(stderr, "[stamps] group lists, block 0 (cycles since its start; box+sphere, scan, barrier, concatenated+counts, image packed):"); for (int i = 1; i < 5; i++) fprintf(stderr, " %lld", st[i] - st[0]); fprintf(stderr, " %lld\n", st[9] - st[0]);
      fprintf(stderr, "[stamps] atoms' lists, block 0 (cycles since its start; bitmap, exclusions found, walk, counts written):"); for (int i = 6; i < 9; i++) fprintf(stderr, " %lld", st[i] - st[5]); fprintf(stderr, " %lld | between the two kernels' starts %lld\n", st[10] - st[5], st[5] - st[0]); }
#endif
    return check_flags(h);
}

// ---- the reciprocal-space mesh energy at the current positions, with the NonbondedForce charges of this (alchemical) System or with
// every atom's own charge (what the non-alchemical System of the MD / alch contexts carries): include/blues_engine.h, blues_mesh_energy
int blues_mesh_energy(BluesEngine* h, int32_t with_alchemical_charges, double* out) {
    if (!out) E_FAIL(h, "mesh energy: no output");
    HIP_OK(h, hipSetDevice(h->device));
    *out = 0.0;
    if (!h->pme) return 0;
    if (flush_program(h)) return 1;
    if (ensure_sorted(h)) return 1;
    h->fin_pending = false; h->pass_valid = false;   // (the launch overwrites the reciprocal-space forces of the last pass)
    if (launch_pme(h, 1 | (with_alchemical_charges ? 2 : 0))) return 1;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    HIP_OK(h, hipMemcpy(out, h->d_pme_e.p, sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int blues_batch_mesh_energy(BluesBatch* B, int32_t with_alchemical_charges, const int32_t* mask, double* out) {
    if (!B || B->eng.empty() || !out) { if (B) B->err = "mesh energy: no batch / no output"; return 2; }
    const int R = B->R();
    if (hipSetDevice(B->eng[0]->device) != hipSuccess) { B->err = "hipSetDevice failed"; return 1; }
    for (int r = 0; r < R; r++) out[r] = 0.0;
    BluesEngine* lead = nullptr;
    B->failed.assign(R, 0);
    std::vector<int> alone;   // (stragglers: their own launches)
    for (int r = 0; r < R; r++) {
        B->active[r] = mask ? (mask[r] != 0) : 1;
        if (!B->active[r]) continue;
        BluesEngine* m = B->eng[r];
        if (flush_program(m) || ensure_sorted(m)) { B->err = "mesh energy: " + m->err; return 1; }
        m->fin_pending = false; m->pass_valid = false;
        if (m->straggler) { alone.push_back(r); B->active[r] = 0; continue; }
        if (!lead) lead = m;
    }
    for (int r : alone) if (blues_mesh_energy(B->eng[r], with_alchemical_charges, &out[r])) { B->err = "mesh energy: " + B->eng[r]->err; for (int q = 0; q < R; q++) B->active[q] = 1; return 1; }
    if (!lead || !lead->pme) { for (int q = 0; q < R; q++) B->active[q] = 1; return 0; }
    for (int r = 0; r < R; r++) if (B->active[r]) { const char* why = ""; if (!batch_congruent_cached(B, r, lead, &why)) { B->err = std::string("mesh energy: members differ in ") + why; return 1; } }
    const bool was_entered = B->entered;
    if (batch_enter(B)) return 1;
    B->leader = lead;
    int rc = batch_refresh_args(B);
    if (!rc) {
        B->lockstep = true;
        for (int r = 0; r < R && !rc; r++) if (B->active[r]) rc = launch_pme(B->eng[r], 1 | (with_alchemical_charges ? 2 : 0));
        B->lockstep = false;
        if (rc) B->err = "mesh energy: " + lead->err;
    }
    if (!rc) {
        try {
            if (B->d_gather.n < (size_t)R) B->d_gather.alloc((size_t)R);
            if (lead->precision == 0) hipLaunchKernelGGL(k_gather_pme_e_b<float>, dim3((R + 255) / 256), dim3(256), 0, B->stream, B->d_nb_f.p, R, B->d_gather.p);
            else hipLaunchKernelGGL(k_gather_pme_e_b<double>, dim3((R + 255) / 256), dim3(256), 0, B->stream, B->d_nb_d.p, R, B->d_gather.p);
            std::vector<double> got(R, 0.0);
            if (hipStreamSynchronize(B->stream) != hipSuccess || hipMemcpy(got.data(), B->d_gather.p, sizeof(double) * R, hipMemcpyDeviceToHost) != hipSuccess) { B->err = "mesh energy: read-back failed"; rc = 1; }
            else for (int r = 0; r < R; r++) if (B->active[r]) out[r] = got[r];
        } catch (std::string& e) { B->err = e; rc = 1; }
    }
    if (!was_entered) batch_leave(B);
    B->leader = B->eng[0];
    for (int r = 0; r < R; r++) B->active[r] = 1;
    return rc;
}

// ---- State snapshots (include/blues_engine.h, "Device-resident State")
int blues_snapshot_capture(BluesEngine* h, int32_t what, BluesSnapshot** out) {
    if (!out || !(what & 3)) E_FAIL(h, "snapshot: nothing requested");
    HIP_OK(h, hipSetDevice(h->device));
    if (flush_program(h)) return 1;
    if ((what & 1) && !h->have_positions) E_FAIL(h, "positions have not been set");
    BluesSnapshot* sn = nullptr;
    if (!h->snap_pool.empty()) { sn = h->snap_pool.back(); h->snap_pool.pop_back(); }
    else {
        sn = new BluesSnapshot(); sn->owner = h; sn->n = h->n;
        if (hipMalloc((void**)&sn->block, sizeof(double) * 6 * (size_t)h->n) != hipSuccess) { delete sn; E_FAIL(h, "hipMalloc failed"); }
        for (int k = 0; k < 3; k++) { sn->x[k] = sn->block + (size_t)k * h->n; sn->v[k] = sn->block + (size_t)(3 + k) * h->n; }
    }
    sn->has_x = what & 1; sn->has_v = (what & 2) != 0;
    Copy6Args c; memset(&c, 0, sizeof c);
    c.n = h->n; c.src_stride = 1; c.dst_stride = 1;
    if (what & 1) for (int k = 0; k < 3; k++) { c.src[c.count] = h->d_x[k].p; c.dst[c.count++] = sn->x[k]; }
    if (what & 2) for (int k = 0; k < 3; k++) { c.src[c.count] = h->d_v[k].p; c.dst[c.count++] = sn->v[k]; }
    hipLaunchKernelGGL(k_copy_arrays, dim3((h->n + 255) / 256), dim3(256), 0, h->stream, c);
    h->st_launches++;
    if (h->acc_cache_valid && h->acc_cache_stamp + 1 == h->st_launches) h->acc_cache_stamp = h->st_launches;   // (a copy of x, v does not touch the accumulators)
    sn->ecache = h->ecache; if (!(what & 1)) sn->ecache.clear();
    sn->box_epoch = h->box_epoch;
    HIP_OK(h, hipGetLastError());
    *out = sn;
    return 0;
}

int blues_snapshot_release(BluesSnapshot* sn) {
    if (!sn) return 0;
    BluesEngine* h = sn->owner;
    if (h->snap_pool.size() < 8) { h->snap_pool.push_back(sn); return 0; }
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);   // the copy that filled it, or a restore reading it, may still be queued
    if (sn->block) hipFree(sn->block);
    delete sn;
    return 0;
}

int blues_snapshot_read(BluesSnapshot* sn, int32_t what, double* out, int32_t n_atoms) {
    BluesEngine* h = sn->owner;
    if (n_atoms != sn->n) E_FAIL(h, "expected %d atoms, got %d", sn->n, n_atoms);
    if (what != 1 && what != 2) E_FAIL(h, "snapshot read: what must be 1 (positions) or 2 (velocities)");
    if ((what == 1 && !sn->has_x) || (what == 2 && !sn->has_v)) E_FAIL(h, "the snapshot does not hold what was asked for");
    HIP_OK(h, hipSetDevice(h->device));
    Copy6Args c; memset(&c, 0, sizeof c);
    c.n = sn->n; c.count = 3; c.src_stride = 1; c.dst_stride = 3;
    for (int k = 0; k < 3; k++) { c.src[k] = what == 1 ? sn->x[k] : sn->v[k]; c.dst[k] = h->d_stage.p + k; }
    hipLaunchKernelGGL(k_copy_arrays, dim3((sn->n + 255) / 256), dim3(256), 0, h->stream, c);
    HIP_OK(h, hipMemcpyAsync(out, h->d_stage.p, sizeof(double) * 3 * sn->n, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}

static int snapshot_usable(BluesEngine* h, const BluesSnapshot* sn) {
    if (!sn) E_FAIL(h, "null snapshot");
    if (sn->n != h->n) E_FAIL(h, "snapshot holds %d atoms, the engine %d", sn->n, h->n);
    if (sn->owner->device != h->device) E_FAIL(h, "snapshot lives on another device");
    return 0;
}

int blues_set_positions_from_snapshot(BluesEngine* h, const BluesSnapshot* sn) {
    if (snapshot_usable(h, sn)) return 1;
    if (!sn->has_x) E_FAIL(h, "the snapshot holds no positions");
    HIP_OK(h, hipSetDevice(h->device));
    if (before_position_edit(h)) return 1;
    if (sn->owner->stream != h->stream) HIP_OK(h, hipStreamSynchronize(sn->owner->stream));   // captured on another engine's stream
    if (!h->sorted_ok) {   // never laid out: take the host route once
        std::vector<double> tmp((size_t)3 * h->n);
        if (blues_snapshot_read(const_cast<BluesSnapshot*>(sn), 1, tmp.data(), h->n)) { h->err = sn->owner->err; return 1; }
        return blues_set_positions(h, tmp.data(), h->n);
    }
    const double* src[3] = {sn->x[0], sn->x[1], sn->x[2]};
    if (load_positions(h, src, 1, 0)) return 1;
    h->xfer_foreign = sn->owner != h;
    // the energy that was known for these positions is known again (OpenMM re-evaluates; same value)
    if (sn->owner == h) { if (sn->box_epoch == h->box_epoch) h->ecache = sn->ecache; }   // (not across a change of the box)
    else if (resolve_xfer(h)) return 1;   // another engine's snapshot: its owner may recycle the buffer on its own stream
    return 0;
}

// d_edit_idx <- idx (skipped when it already holds this list: a Move names the same atoms every iteration) and, when xyz is
// given, d_edit_xyz <- xyz, through pinned staging so that neither copy blocks the host.  The pinned area is reused by the next
// call: a copy still in flight from the previous one is waited for first (same stream, so normally long done).
static int stage_edit_list(BluesEngine* h, const int32_t* idx, int n_idx, const double* xyz) {
    try {
        if ((int)h->d_edit_idx.n < n_idx) { h->d_edit_idx.alloc(n_idx); h->d_edit_xyz.alloc((size_t)3 * n_idx); h->edit_idx_host.clear(); }
    } catch (std::string& e) { E_FAIL(h, "%s", e.c_str()); }
    const size_t need = (sizeof(int) + 3 * sizeof(double)) * (size_t)n_idx;
    if (h->h_edit_cap < need) {
        if (h->h_edit) { hipStreamSynchronize(h->stream); hipHostFree(h->h_edit); h->h_edit = nullptr; h->h_edit_cap = 0; }
        if (hipHostMalloc((void**)&h->h_edit, need, hipHostMallocDefault) != hipSuccess) E_FAIL(h, "hipHostMalloc failed");
        h->h_edit_cap = need;
    } else if (h->ev_edit) HIP_OK(h, hipEventSynchronize(h->ev_edit));
    double* px = reinterpret_cast<double*>(h->h_edit);
    int* pi = reinterpret_cast<int*>(h->h_edit + 3 * sizeof(double) * (size_t)n_idx);
    const bool same = (int)h->edit_idx_host.size() == n_idx && !memcmp(h->edit_idx_host.data(), idx, sizeof(int) * n_idx);
    if (!same) {
        memcpy(pi, idx, sizeof(int) * n_idx);
        HIP_OK(h, hipMemcpyAsync(h->d_edit_idx.p, pi, sizeof(int) * n_idx, hipMemcpyHostToDevice, h->stream));
        h->edit_idx_host.assign(idx, idx + n_idx);
    }
    if (xyz) {
        memcpy(px, xyz, sizeof(double) * 3 * n_idx);
        HIP_OK(h, hipMemcpyAsync(h->d_edit_xyz.p, px, sizeof(double) * 3 * n_idx, hipMemcpyHostToDevice, h->stream));
    }
    if (!same || xyz) {
        if (!h->ev_edit) HIP_OK(h, hipEventCreateWithFlags(&h->ev_edit, hipEventDisableTiming));
        HIP_OK(h, hipEventRecord(h->ev_edit, h->stream));
    }
    return 0;
}

// setPositions with a State's positions in which a Move changed a few atoms (positions[i] = ... on the still device-resident
// array): the unchanged coordinates are restored device to device, the edited ones travel.  Returns 3 (and does nothing)
// when an edited atom shares a constraint cluster with an atom that is not edited -- the cluster would have to be made
// whole with coordinates only the device has; the caller then takes the plain host route.
int blues_set_positions_from_snapshot_edited(BluesEngine* h, const BluesSnapshot* sn, const int32_t* idx, int32_t n_idx, const double* xyz) {
    if (snapshot_usable(h, sn)) return 1;
    if (!sn->has_x) E_FAIL(h, "the snapshot holds no positions");
    if (n_idx <= 0) return blues_set_positions_from_snapshot(h, sn);
    if (!h->sorted_ok) return 3;
    std::vector<int> pos_of(h->n, -1);
    for (int e = 0; e < n_idx; e++) { if (idx[e] < 0 || idx[e] >= h->n) E_FAIL(h, "edited atom %d out of range", idx[e]); pos_of[idx[e]] = e; }
    std::vector<double> ed(xyz, xyz + 3 * (size_t)n_idx);
    for (const HostCluster& c : h->clusters) {
        int inside = 0, total = 0;
        for (int a = 0; a < 4; a++) if (c.atoms[a] >= 0) { total++; inside += pos_of[c.atoms[a]] >= 0; }
        if (inside == 0) continue;
        if (inside != total) return 3;
        for (int a = 1; a < 4; a++) if (c.atoms[a] >= 0)   // one whole periodic image per cluster, as blues_set_positions stores them
            for (int k = 0; k < 3; k++) {
                const double d = ed[3 * pos_of[c.atoms[a]] + k] - ed[3 * pos_of[c.atoms[0]] + k];
                ed[3 * pos_of[c.atoms[a]] + k] -= h->box[k] * std::nearbyint(d / h->box[k]);
            }
    }
    HIP_OK(h, hipSetDevice(h->device));
    if (before_position_edit(h)) return 1;
    if (sn->owner->stream != h->stream) HIP_OK(h, hipStreamSynchronize(sn->owner->stream));
    if (stage_edit_list(h, idx, n_idx, ed.data())) return 1;
    const double* src[3] = {sn->x[0], sn->x[1], sn->x[2]};
    if (load_positions(h, src, 1, n_idx)) return 1;
    if (sn->owner != h && resolve_xfer(h)) return 1;
    return 0;
}

// positions / velocities of a few atoms of a snapshot (a Move that reads positions[indices] of a device-resident State)
int blues_snapshot_read_atoms(BluesSnapshot* sn, int32_t what, const int32_t* idx, int32_t n_idx, double* out) {
    BluesEngine* h = sn->owner;
    if (what != 1 && what != 2) E_FAIL(h, "snapshot read: what must be 1 (positions) or 2 (velocities)");
    if ((what == 1 && !sn->has_x) || (what == 2 && !sn->has_v)) E_FAIL(h, "the snapshot does not hold what was asked for");
    if (n_idx <= 0) return 0;
    for (int e = 0; e < n_idx; e++) if (idx[e] < 0 || idx[e] >= sn->n) E_FAIL(h, "atom %d out of range", idx[e]);
    HIP_OK(h, hipSetDevice(h->device));
    if (stage_edit_list(h, idx, n_idx, nullptr)) return 1;
    double* const* s = what == 1 ? sn->x : sn->v;
    hipLaunchKernelGGL(k_gather_atoms, dim3((n_idx + 63) / 64), dim3(64), 0, h->stream, n_idx, h->d_edit_idx.p, s[0], s[1], s[2], h->d_edit_xyz.p);
    double* px = reinterpret_cast<double*>(h->h_edit);
    HIP_OK(h, hipMemcpyAsync(px, h->d_edit_xyz.p, sizeof(double) * 3 * n_idx, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));   // the one wait of this call
    memcpy(out, px, sizeof(double) * 3 * n_idx);
    return 0;
}

int blues_set_velocities_from_snapshot(BluesEngine* h, const BluesSnapshot* sn) {
    if (snapshot_usable(h, sn)) return 1;
    if (!sn->has_v) E_FAIL(h, "the snapshot holds no velocities");
    HIP_OK(h, hipSetDevice(h->device));
    if (flush_program(h)) return 1;
    if (sn->owner->stream != h->stream) HIP_OK(h, hipStreamSynchronize(sn->owner->stream));
    Copy6Args c; memset(&c, 0, sizeof c);
    c.n = h->n; c.count = 3; c.src_stride = 1; c.dst_stride = 1;
    for (int k = 0; k < 3; k++) { c.src[k] = sn->v[k]; c.dst[k] = h->d_v[k].p; }
    hipLaunchKernelGGL(k_copy_arrays, dim3((h->n + 255) / 256), dim3(256), 0, h->stream, c);
    h->st_launches++; h->vel_clean = false; h->ke_cache_valid = false;
    HIP_OK(h, hipGetLastError());
    if (sn->owner->stream != h->stream) HIP_OK(h, hipStreamSynchronize(h->stream));   // the owner may recycle the buffer on its own stream
    return 0;
}

// ---- replica batch (include/blues_engine.h, "Replica batches")
static thread_local std::string g_batch_create_error;

const char* blues_batch_last_error(const BluesBatch* b) { return b ? b->err.c_str() : g_batch_create_error.c_str(); }

int blues_batch_create(BluesEngine* const* engines, int32_t count, BluesBatch** out) {
    if (count <= 0 || !engines || !out) { g_batch_create_error = "a batch needs at least one engine"; return 2; }
    for (int r = 0; r < count; r++) {
        if (!engines[r]) { g_batch_create_error = "null engine handle"; return 2; }
        if (engines[r]->batch) { g_batch_create_error = "engine already belongs to a batch"; return 2; }
        if (engines[r]->device != engines[0]->device) { g_batch_create_error = "all engines of a batch must live on the same device"; return 2; }
        if (engines[r]->switch_mode != BLUES_SWITCH_NONE) { g_batch_create_error = "the switching integrators (switching_mode != 0) step one engine at a time: their energy bookkeeping is synchronous"; return 2; }
        for (int q = 0; q < r; q++) if (engines[q] == engines[r]) { g_batch_create_error = "duplicate engine handle"; return 2; }
    }
    BluesBatch* B = new BluesBatch();
    hipSetDevice(engines[0]->device);
    if (hipStreamCreate(&B->stream) != hipSuccess) { g_batch_create_error = "hipStreamCreate failed"; delete B; return 1; }
    for (int r = 0; r < count; r++) {
        BluesEngine* m = engines[r];
        if (flush_program(m) || hipStreamSynchronize(m->stream) != hipSuccess) { g_batch_create_error = "could not drain engine stream: " + m->err; batch_detach_all(B); delete B; return 1; }
        B->eng.push_back(m); m->batch = B; m->batch_index = r;
        m->batch_R = std::max((int)count, m->tune.assume_batch);   // (assume_batch: a floor, so that small test batches take the large-batch decomposition too)
        m->use_graph = false;  // graph replays carry per-engine frozen arguments
    }
    if (batch_plan_shape(B, true)) { g_batch_create_error = B->err; batch_detach_all(B); delete B; return 1; }
    if (engines[0]->sorted_ok && engines[0]->k1_mode != 2 && engines[0]->k1_mode != 3)
        for (int r = 0; r < count; r++) { BluesEngine* m = engines[r]; m->pin_jcap = engines[0]->jcap; m->pin_seg = engines[0]->seg_len; m->pin_nw = engines[0]->waves_tile; m->pin_wpb = engines[0]->wpb; }
    // off by default: measured on MI355X at R = 256 it does not pay (with 256 x 276 mobile atoms SOME atom crosses skin/2
    // every ~3 steps, and rebuilding all members at once costs about what one rebuild per step did: 669 vs 678 us/step)
    B->sync_lists = false;
    B->tune = g_tuning;
    B->sync_lists = B->tune.batch_sync_lists != 0;
    try { B->d_req.alloc(1); B->d_work.alloc((size_t)2 * count + 1); B->d_work_frag.alloc((size_t)2 * count + 1); } catch (std::string& e) { g_batch_create_error = e; batch_detach_all(B); delete B; return 1; }
    B->seen_epoch.assign(count, 0); B->failed.assign(count, 0); B->active.assign(count, 1); B->rec_active.assign(count, 0); B->rec_delta.assign(count, 0); B->leader = engines[0];
    *out = B;
    return 0;
}

int blues_batch_destroy(BluesBatch* b) {
    if (b) { k1t_harvest(b, true); for (auto& p : b->k1t_pairs) { if (p.a) hipEventDestroy(p.a); if (p.b) hipEventDestroy(p.b); } b->k1t_pairs.clear(); }
    if (!b) return 0;
    if (!b->eng.empty()) hipSetDevice(b->eng[0]->device);
    // (nothing of this batch may still run when its records and pinned areas go: device buffers from the slabs are released without the
    // implicit device-wide wait every hipFree used to be)
    if (b->stream) hipStreamSynchronize(b->stream);
    for (BluesEngine* m : b->eng) if (m && m->xfer_pending && m->xfer_src != m->h_xfer) resolve_xfer(m);   // (a verdict parked in the batch's buffer)
    batch_detach_all(b);
    if (b->h_arena) { hipHostFree(b->h_arena); b->h_arena = nullptr; }
    if (b->h_xfer_all) { hipHostFree(b->h_xfer_all); b->h_xfer_all = nullptr; }
    delete b;
    return 0;
}

int blues_batch_size(const BluesBatch* b) { return b ? b->R() : 0; }

int blues_batch_set_active(BluesBatch* b, const int32_t* mask) {
    if (!b || b->eng.empty()) { if (b) b->err = "the batch has been dissolved (one of its engines was destroyed)"; return 2; }
    for (int r = 0; r < b->R(); r++) b->active[r] = mask ? (mask[r] != 0) : 1;
    return 0;
}

int blues_batch_step(BluesBatch* b, int32_t n_steps, double* work_trace, int32_t* status) {
    if (!b || b->eng.empty()) { if (b) b->err = "the batch has been dissolved (one of its engines was destroyed)"; return 2; }
    if (hipSetDevice(b->eng[0]->device) != hipSuccess) { b->err = "hipSetDevice failed"; return 1; }
    const int R = b->R();
    std::vector<int> st(R, 0), first(R, 0);
    for (int r = 0; r < R; r++) first[r] = b->eng[r]->h_step;
    if (batch_enter(b)) return 1;
    const bool has_work = b->eng[0]->n_H > 0;
    const int rc = batch_do_steps(b, n_steps, work_trace != nullptr && has_work, st.data());
    batch_leave(b);
    if (status) for (int r = 0; r < R; r++) status[r] = st[r];
    if (rc) return rc;
    if (work_trace && !has_work) { for (size_t i = 0; i < (size_t)R * n_steps; i++) work_trace[i] = 0.0; }
    else if (work_trace) for (int r = 0; r < R; r++) {
        BluesEngine* h = b->eng[r];
        double* wt = work_trace + (size_t)r * n_steps;
        const int done = st[r] ? 0 : h->h_step - first[r];
        if (done > 0 && hipMemcpy(wt, h->d_trace.p + first[r], sizeof(double) * done, hipMemcpyDeviceToHost) != hipSuccess) { b->err = "work trace download failed"; return 1; }
        for (int i = std::max(done, 0); i < n_steps; i++) wt[i] = done > 0 ? wt[done - 1] : 0.0;
    }
    return 0;
}

int blues_batch_prefetch_energies(BluesBatch* b, int32_t what) {
    if (!b || b->eng.empty()) { if (b) b->err = "the batch has been dissolved (one of its engines was destroyed)"; return 2; }
    if (hipSetDevice(b->eng[0]->device) != hipSuccess) { b->err = "hipSetDevice failed"; return 1; }
    b->failed.assign(b->R(), 0);
    return batch_prefetch(b, what);
}

int blues_batch_get_counters(BluesBatch* b, double out[BLUES_N_BATCH_COUNTERS]) {
    if (!b || !out) return 2;
    for (int q = 0; q < BLUES_N_BATCH_COUNTERS; q++) out[q] = 0.0;
    out[0] = (double)b->st_replans; out[1] = b->st_replan_sec; out[2] = (double)b->st_relayouts; out[3] = (double)b->st_poll_resorts; out[4] = b->st_resort_sec;
    for (const BluesEngine* m : b->eng) { out[5] += (double)m->st_reshapes; out[6] += (double)m->st_resorts; }
    const BluesEngine* rep = nullptr;
    for (const BluesEngine* m : b->eng) { if (!m->straggler && !rep) rep = m; out[10] += m->straggler ? 1.0 : 0.0; }
    if (!rep && !b->eng.empty()) rep = b->eng[0];
    if (rep) { out[7] = (double)rep->S; out[8] = (double)rep->jcap; out[9] = (double)rep->k1_mode; }
    out[11] = (double)b->st_stragglers; out[12] = (double)b->st_rejoined; out[13] = b->st_straggle_sec; out[14] = (double)b->st_partial_steps;
    return 0;
}

int blues_batch_get_stats(BluesBatch* b, int64_t stats[4]) {
    stats[0] = b->st_lockstep_steps; stats[1] = b->st_fallback_steps; stats[2] = b->R(); stats[3] = b->st_prefetch_pe;
    return 0;
}

// average duration of one batched nonbonded launch (all replicas), HIP events on the batch's stream
// ---- the plugin boundary for a whole batch.  BLUES takes and hands back whole States of every chain three times per
// iteration, a Move reads and writes a few atoms of each, and the Metropolis step resets every integrator and redraws every
// chain's velocities (reference blues/simulation.py:1028-1187).  Chain by chain that is ~60 C-ABI calls and ~25 small launches
// per chain and iteration; the entry points below do one thing for ALL members with one launch per kernel: the host part of the
// per-member entry point runs in a loop, the kernel arguments of the members travel as an array.
static int batch_arena(BluesBatch* B, size_t bytes) {
    if (B->arena_cap >= bytes) return 0;
    if (B->h_arena) { hipStreamSynchronize(B->stream); hipHostFree(B->h_arena); B->h_arena = nullptr; B->arena_cap = 0; }
    const size_t cap = std::max<size_t>(bytes * 2, 1 << 16);
    if (hipHostMalloc((void**)&B->h_arena, cap, hipHostMallocDefault) != hipSuccess) { B->err = "hipHostMalloc failed"; return 1; }
    try { B->d_arena.alloc(cap); } catch (std::string& e) { B->err = e; return 1; }
    B->arena_cap = cap;
    return 0;
}
static int batch_arena_upload(BluesBatch* B, size_t bytes) {
    if (hipMemcpyAsync(B->d_arena.p, B->h_arena, bytes, hipMemcpyHostToDevice, B->stream) != hipSuccess) { B->err = "argument upload failed"; return 1; }
    return 0;
}
static inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }
// members of a batch need not have the same atom count (non-congruent members step one by one): a launch that covers all of
// them is sized by the largest, and an index must be valid in the smallest
static int batch_max_n(const BluesBatch* B) { int m = 0; for (const BluesEngine* h : B->eng) m = std::max(m, h->n); return m; }
static int batch_min_n(const BluesBatch* B) { int m = B->eng[0]->n; for (const BluesEngine* h : B->eng) m = std::min(m, h->n); return m; }

int blues_batch_snapshot_capture(BluesBatch* B, int32_t what, const int32_t* mask, BluesSnapshot** out) {
    if (!B || !out || !(what & 3)) return 2;
    const int R = B->R();
    BluesEngine* h0 = B->eng[0];
    if (hipSetDevice(h0->device) != hipSuccess) { B->err = "hipSetDevice failed"; return 1; }
    if (batch_arena(B, sizeof(Copy6Args) * R)) return 1;
    for (int r = 0; r < R; r++) if ((!mask || mask[r]) && flush_program(B->eng[r])) { B->err = B->eng[r]->err; return 1; }
    if (batch_enter(B)) return 1;
    Copy6Args* args = reinterpret_cast<Copy6Args*>(B->h_arena);
    for (int r = 0; r < R; r++) {
        BluesEngine* h = B->eng[r];
        Copy6Args& c = args[r]; memset(&c, 0, sizeof c);
        out[r] = nullptr;
        if (mask && !mask[r]) continue;
        // (a failure half way through: the snapshots already taken for earlier members go back to their owners' pools)
        auto undo = [&]() { for (int q = 0; q < r; q++) if (out[q]) { out[q]->owner->snap_pool.push_back(out[q]); out[q] = nullptr; } };
        if ((what & 1) && !h->have_positions) { undo(); B->err = "positions have not been set"; batch_leave(B); return 1; }
        BluesSnapshot* sn = nullptr;
        if (!h->snap_pool.empty()) { sn = h->snap_pool.back(); h->snap_pool.pop_back(); }
        else {
            sn = new BluesSnapshot(); sn->owner = h; sn->n = h->n;
            if (hipMalloc((void**)&sn->block, sizeof(double) * 6 * (size_t)h->n) != hipSuccess) { delete sn; undo(); B->err = "hipMalloc failed"; batch_leave(B); return 1; }
            for (int k = 0; k < 3; k++) { sn->x[k] = sn->block + (size_t)k * h->n; sn->v[k] = sn->block + (size_t)(3 + k) * h->n; }
        }
        sn->has_x = what & 1; sn->has_v = (what & 2) != 0;
        c.n = h->n; c.src_stride = 1; c.dst_stride = 1;
        if (what & 1) for (int k = 0; k < 3; k++) { c.src[c.count] = h->d_x[k].p; c.dst[c.count++] = sn->x[k]; }
        if (what & 2) for (int k = 0; k < 3; k++) { c.src[c.count] = h->d_v[k].p; c.dst[c.count++] = sn->v[k]; }
        sn->ecache = h->ecache; if (!(what & 1)) sn->ecache.clear();
        sn->box_epoch = h->box_epoch;
        h->st_launches++;
        if (h->acc_cache_valid && h->acc_cache_stamp + 1 == h->st_launches) h->acc_cache_stamp = h->st_launches;   // (a copy of x, v does not touch the accumulators)
        out[r] = sn;
    }
    int rc = batch_arena_upload(B, sizeof(Copy6Args) * R);
    if (!rc) hipLaunchKernelGGL(k_copy_arrays_b, dim3((batch_max_n(B) + 255) / 256, R), dim3(256), 0, B->stream, reinterpret_cast<const Copy6Args*>(B->d_arena.p));
    if (!rc && hipGetLastError() != hipSuccess) { B->err = "launch failed"; rc = 1; }
    batch_leave(B);
    return rc;
}

// positions (what bit 0; `edit_*`: a few atoms of every member overwritten on the way, xyz [R][n_idx][3]) and / or velocities
// (bit 1) of every member from its snapshot (null: the member sits out).  Returns 3 without doing anything when some member cannot
// take the device route (never laid out; an edited atom constrained to an unedited one): the caller falls back to member calls.
static int batch_restore_impl(BluesBatch* B, BluesSnapshot* const* snaps, int what, const int32_t* idx, int n_idx, const double* xyz) {
    const int R = B->R();
    BluesEngine* h0 = B->eng[0];
    if (hipSetDevice(h0->device) != hipSuccess) { B->err = "hipSetDevice failed"; return 1; }
    const int n = h0->n;
    std::vector<double> ed;
    if (what & 1) {
        for (int r = 0; r < R; r++) if (snaps[r]) {
            BluesEngine* h = B->eng[r];
            if (snapshot_usable(h, snaps[r]) || !snaps[r]->has_x) { B->err = h->err.empty() ? "the snapshot holds no positions" : h->err; return 1; }
            if (!h->sorted_ok || h->n != n) return 3;
        }
        if (n_idx > 0) {   // whole constraint clusters only, each stored as one periodic image (blues_set_positions_from_snapshot_edited)
            std::vector<int> pos_of(n, -1);
            for (int e = 0; e < n_idx; e++) { if (idx[e] < 0 || idx[e] >= n) { B->err = "edited atom out of range"; return 1; } pos_of[idx[e]] = e; }
            ed.assign(xyz, xyz + (size_t)3 * n_idx * R);
            for (int r = 0; r < R; r++) if (snaps[r]) {
                BluesEngine* h = B->eng[r];
                double* e0 = ed.data() + (size_t)3 * n_idx * r;
                for (const HostCluster& c : h->clusters) {
                    int inside = 0, total = 0;
                    for (int a = 0; a < 4; a++) if (c.atoms[a] >= 0) { total++; inside += pos_of[c.atoms[a]] >= 0; }
                    if (inside == 0) continue;
                    if (inside != total) return 3;
                    for (int a = 1; a < 4; a++) if (c.atoms[a] >= 0)
                        for (int k = 0; k < 3; k++) {
                            const double d = e0[3 * pos_of[c.atoms[a]] + k] - e0[3 * pos_of[c.atoms[0]] + k];
                            e0[3 * pos_of[c.atoms[a]] + k] -= h->box[k] * std::nearbyint(d / h->box[k]);
                        }
                }
            }
        }
    }
    if (what & 2) for (int r = 0; r < R; r++) if (snaps[r] && (snapshot_usable(B->eng[r], snaps[r]) || !snaps[r]->has_v)) { B->err = "the snapshot holds no velocities"; return 1; }
    // (before_position_edit may evaluate an energy of its own: before the members move to the batch stream)
    for (int r = 0; r < R; r++) if (snaps[r]) {
        BluesEngine* h = B->eng[r];
        if ((what & 1) ? before_position_edit(h) : flush_program(h)) { B->err = h->err; return 1; }
        if (snaps[r]->owner->stream != h->stream && !(snaps[r]->owner->batch == B)) hipStreamSynchronize(snaps[r]->owner->stream);   // captured on a stream outside this batch
    }
    const size_t off_load = 0, off_edit = align16(off_load + sizeof(LoadPosArgs) * R), off_copy = align16(off_edit + sizeof(EditPosArgs) * R),
                 off_idx = align16(off_copy + sizeof(Copy6Args) * R), off_xyz = align16(off_idx + sizeof(int) * std::max(1, n_idx)), total = align16(off_xyz + sizeof(double) * 3 * (size_t)std::max(1, n_idx) * R);
    if (batch_arena(B, total)) return 1;
    if (!B->h_xfer_all) { if (hipHostMalloc((void**)&B->h_xfer_all, sizeof(unsigned) * 4 * R, hipHostMallocDefault) != hipSuccess) { B->err = "hipHostMalloc failed"; return 1; } try { B->d_xfer_all.alloc((size_t)4 * R); } catch (std::string& e) { B->err = e; return 1; } }
    if (batch_enter(B)) return 1;
    LoadPosArgs* la = reinterpret_cast<LoadPosArgs*>(B->h_arena + off_load); EditPosArgs* ea = reinterpret_cast<EditPosArgs*>(B->h_arena + off_edit);
    Copy6Args* ca = reinterpret_cast<Copy6Args*>(B->h_arena + off_copy);
    if (n_idx > 0) { memcpy(B->h_arena + off_idx, idx, sizeof(int) * n_idx); memcpy(B->h_arena + off_xyz, ed.data(), sizeof(double) * 3 * (size_t)n_idx * R); }
    unsigned char* dbase = B->d_arena.p;
    for (int r = 0; r < R; r++) {
        BluesEngine* h = B->eng[r];
        LoadPosArgs& a = la[r]; memset(&a, 0, sizeof a); EditPosArgs& e = ea[r]; memset(&e, 0, sizeof e); Copy6Args& c = ca[r]; memset(&c, 0, sizeof c);
        if (!snaps[r]) continue;
        const BluesSnapshot* sn = snaps[r];
        if (what & 1) {
            a.n = h->n; a.stride = 1;
            for (int k = 0; k < 3; k++) { a.src[k] = sn->x[k]; a.x[k] = h->d_x[k].p; a.x_sort[k] = h->d_x_sort[k].p; }
            a.mass = h->d_mass.p; a.alch_local = h->d_alch_local.p; a.sorted_of_orig = h->d_sorted_of_orig.p;
            a.img_f = h->precision == 0 ? h->d_img_f.p : nullptr; a.img_d = h->precision == 0 ? nullptr : h->d_img_d.p;
            a.box = make_box(h); a.out = h->d_xfer_out.p;
            a.keep_out = h->xfer_pending ? 1 : 0;   // (a verdict still pending from an earlier load is kept: the kernels only OR / max / add into it)
            if (n_idx > 0) {
                e.n_edit = n_idx; e.idx = reinterpret_cast<const int*>(dbase + off_idx); e.xyz = reinterpret_cast<const double*>(dbase + off_xyz) + (size_t)3 * n_idx * r;
                for (int k = 0; k < 3; k++) { e.x[k] = h->d_x[k].p; e.x_sort[k] = h->d_x_sort[k].p; }
                e.mass = a.mass; e.alch_local = a.alch_local; e.sorted_of_orig = a.sorted_of_orig; e.img_f = a.img_f; e.img_d = a.img_d; e.box = a.box; e.out = a.out;
            }
        }
        if (what & 2) { c.n = h->n; c.count = 3; c.src_stride = 1; c.dst_stride = 1; for (int k = 0; k < 3; k++) { c.src[k] = sn->v[k]; c.dst[k] = h->d_v[k].p; } }
    }
    int rc = batch_arena_upload(B, total);
    const LoadPosArgs* dla = reinterpret_cast<const LoadPosArgs*>(dbase + off_load);
    if (!rc && (what & 1)) {
        hipLaunchKernelGGL(k_zero_xfer_b, dim3((R + 255) / 256), dim3(256), 0, B->stream, dla, R);
        hipLaunchKernelGGL(k_load_positions_b, dim3((n + 255) / 256, R), dim3(256), 0, B->stream, dla);
        if (n_idx > 0) hipLaunchKernelGGL(k_edit_positions_b, dim3((n_idx + 63) / 64, R), dim3(64), 0, B->stream, reinterpret_cast<const EditPosArgs*>(dbase + off_edit));
        hipLaunchKernelGGL(k_gather_xfer_b, dim3((R + 255) / 256), dim3(256), 0, B->stream, dla, R, B->d_xfer_all.p);
        if (hipMemcpyAsync(B->h_xfer_all, B->d_xfer_all.p, sizeof(unsigned) * 4 * R, hipMemcpyDeviceToHost, B->stream) != hipSuccess) rc = 1;
    }
    if (!rc && (what & 2)) hipLaunchKernelGGL(k_copy_arrays_b, dim3((n + 255) / 256, R), dim3(256), 0, B->stream, reinterpret_cast<const Copy6Args*>(dbase + off_copy));
    if (!rc && hipGetLastError() != hipSuccess) { B->err = "launch failed"; rc = 1; }
    batch_leave(B);   // (drains the batch stream: the verdicts have arrived)
    if (rc) { if (B->err.empty()) B->err = "batched restore failed"; return 1; }
    for (int r = 0; r < R; r++) if (snaps[r]) {
        BluesEngine* h = B->eng[r];
        if (what & 1) {
            h->xfer_pending = true; h->xfer_stream = h->stream; h->xfer_src = B->h_xfer_all + 4 * r; h->xfer_foreign = snaps[r]->owner != h;
            h->have_positions = true; h->x_edited = true; h->pass_valid = false; h->fin_pending = false; h->ecache.clear(); h->lists_forced = true;
            h->st_launches += n_idx > 0 ? 2 : 1;
            if (snaps[r]->owner == h && n_idx == 0 && snaps[r]->box_epoch == h->box_epoch) h->ecache = snaps[r]->ecache;   // the energy that was known for these positions (in this box) is known again
            if (resolve_xfer(h)) { B->err = h->err; return 1; }   // (already on the host: no wait)
        }
        if (what & 2) { h->st_launches++; h->vel_clean = false; h->ke_cache_valid = false; }
    }
    return 0;
}
int blues_batch_restore(BluesBatch* B, BluesSnapshot* const* snaps, int32_t what) {
    if (!B || !snaps || !(what & 3)) return 2;
    return batch_restore_impl(B, snaps, what, nullptr, 0, nullptr);
}
int blues_batch_restore_edited(BluesBatch* B, BluesSnapshot* const* snaps, const int32_t* idx, int32_t n_idx, const double* xyz) {
    if (!B || !snaps || n_idx < 0 || (n_idx > 0 && (!idx || !xyz))) return 2;
    return batch_restore_impl(B, snaps, 1, idx, n_idx, xyz);
}

// positions (what = 1) or velocities (2) of the atoms idx of every member -- from its snapshot, or its live state where
// snaps is NULL -- as out[R][n_idx][3]: one gather, one read-back (positions[atom_indices] of a Move, for all chains)
int blues_batch_read_atoms(BluesBatch* B, BluesSnapshot* const* snaps, int32_t what, const int32_t* idx, int32_t n_idx, double* out) {
    if (!B || !idx || !out || n_idx <= 0 || (what != 1 && what != 2)) return 2;
    const int R = B->R();
    BluesEngine* h0 = B->eng[0];
    if (hipSetDevice(h0->device) != hipSuccess) { B->err = "hipSetDevice failed"; return 1; }
    { const int nmin = batch_min_n(B); for (int e = 0; e < n_idx; e++) if (idx[e] < 0 || idx[e] >= nmin) { B->err = "atom out of range"; return 1; } }
    const size_t off_src = 0, off_idx = align16(sizeof(double*) * 3 * R), off_out = align16(off_idx + sizeof(int) * n_idx), total = off_out + sizeof(double) * 3 * (size_t)n_idx * R;
    if (batch_arena(B, total)) return 1;
    for (int r = 0; r < R; r++) if (!snaps && flush_program(B->eng[r])) { B->err = B->eng[r]->err; return 1; }
    if (batch_enter(B)) return 1;
    const double** src = reinterpret_cast<const double**>(B->h_arena + off_src);
    for (int r = 0; r < R; r++) for (int k = 0; k < 3; k++) {
        const double* p = nullptr;
        if (snaps) { if (snaps[r] && (what == 1 ? snaps[r]->has_x : snaps[r]->has_v)) p = what == 1 ? snaps[r]->x[k] : snaps[r]->v[k]; }
        else p = what == 1 ? B->eng[r]->d_x[k].p : B->eng[r]->d_v[k].p;
        src[3 * r + k] = p;
    }
    memcpy(B->h_arena + off_idx, idx, sizeof(int) * n_idx);
    int rc = batch_arena_upload(B, off_out);
    if (!rc) {
        hipLaunchKernelGGL(k_gather_atoms_b, dim3((n_idx + 63) / 64, R), dim3(64), 0, B->stream, n_idx, reinterpret_cast<const int*>(B->d_arena.p + off_idx),
                           reinterpret_cast<const double* const*>(B->d_arena.p + off_src), reinterpret_cast<double*>(B->d_arena.p + off_out));
        if (hipMemcpyAsync(B->h_arena + off_out, B->d_arena.p + off_out, sizeof(double) * 3 * (size_t)n_idx * R, hipMemcpyDeviceToHost, B->stream) != hipSuccess) rc = 1;
    }
    batch_leave(B);
    if (rc) { if (B->err.empty()) B->err = "batched gather failed"; return 1; }
    memcpy(out, B->h_arena + off_out, sizeof(double) * 3 * (size_t)n_idx * R);
    return 0;
}

// integrator.reset() of every member (mask NULL: all) -- reference blues/integrators.py:240-249, simulation.py:1184
int blues_batch_reset(BluesBatch* B, const int32_t* mask) {
    if (!B) return 2;
    const int R = B->R();
    if (hipSetDevice(B->eng[0]->device) != hipSuccess) { B->err = "hipSetDevice failed"; return 1; }
    if (batch_arena(B, sizeof(DevAccum*) * R)) return 1;
    for (int r = 0; r < R; r++) if ((!mask || mask[r]) && flush_program(B->eng[r])) { B->err = B->eng[r]->err; return 1; }
    if (batch_enter(B)) return 1;
    DevAccum** acc = reinterpret_cast<DevAccum**>(B->h_arena);
    for (int r = 0; r < R; r++) {
        BluesEngine* h = B->eng[r];
        acc[r] = nullptr;
        if (mask && !mask[r]) continue;
        acc[r] = h->d_acc.p;
        h->h_step = 0; h->h_lambda = 0.0; h->h_first_step = 0; h->h_perturbed = 0.0; h->h_unperturbed = 0.0; h->h_prop = 1; h->h_lambda_step = 0;
        h->unpert_valid = false; h->x_edited = false; h->pass_valid = false; h->fin_pending = false; h->acc_cache_valid = false;
    }
    int rc = batch_arena_upload(B, sizeof(DevAccum*) * R);
    if (!rc) hipLaunchKernelGGL(k_zero_acc_b, dim3((R + 255) / 256), dim3(256), 0, B->stream, reinterpret_cast<DevAccum* const*>(B->d_arena.p), R);
    batch_leave(B);
    return rc;
}

// context.setVelocitiesToTemperature(T, seed[r]) of every member (reference blues/simulation.py:1187): one Maxwell-Boltzmann
// launch, then the velocity constraints of all members in lock step
int blues_batch_set_velocities_to_temperature(BluesBatch* B, double temperature, const uint64_t* seeds, const int32_t* mask) {
    if (!B || !seeds) return 2;
    const int R = B->R();
    BluesEngine* h0 = B->eng[0];
    if (hipSetDevice(h0->device) != hipSuccess) { B->err = "hipSetDevice failed"; return 1; }
    if (batch_arena(B, sizeof(MaxwellArgs) * R)) return 1;
    for (int r = 0; r < R; r++) if ((!mask || mask[r]) && flush_program(B->eng[r])) { B->err = B->eng[r]->err; return 1; }
    if (batch_enter(B)) return 1;
    MaxwellArgs* ma = reinterpret_cast<MaxwellArgs*>(B->h_arena);
    for (int r = 0; r < R; r++) {
        BluesEngine* h = B->eng[r];
        MaxwellArgs& a = ma[r]; memset(&a, 0, sizeof a);
        if (mask && !mask[r]) continue;
        a.n = h->n; a.mass = h->d_mass.p; for (int k = 0; k < 3; k++) a.v[k] = h->d_v[k].p;
        a.kT = KB_KJ * temperature; a.seed = (unsigned long long)seeds[r]; a.stream = (unsigned)h->replica * 4u + 1u;
        h->st_launches++; h->vel_clean = false; h->ke_cache_valid = false;
    }
    int rc = batch_arena_upload(B, sizeof(MaxwellArgs) * R);
    if (!rc) hipLaunchKernelGGL(k_maxwell_b, dim3((batch_max_n(B) + 255) / 256, R), dim3(256), 0, B->stream, reinterpret_cast<const MaxwellArgs*>(B->d_arena.p));
    // the velocity constraints: every member queues OP_RATTLE; congruent members flush in lock step (one launch)
    std::vector<char> saved_active = B->active;
    B->failed.assign(R, 0);
    for (int r = 0; r < R; r++) { B->active[r] = (!mask || mask[r]) ? 1 : 0; if (!B->active[r]) B->failed[r] = 1; }
    BluesEngine* lead = nullptr;
    for (int r = 0; r < R && !rc; r++) if (!B->failed[r]) { if (!lead) lead = B->eng[r]; if (ensure_sorted(B->eng[r]) || emit(B->eng[r], OP_RATTLE)) { B->err = B->eng[r]->err; rc = 1; } }
    if (!rc && lead) {
        B->leader = lead;
        bool uniform = true;
        const BatchSig ls = batch_sig(lead);
        for (int r = 0; r < R && uniform; r++) if (!B->failed[r]) { const BatchSig g = batch_sig(B->eng[r]); const char* why = ""; uniform = !memcmp(&g, &ls, sizeof g) && batch_congruent_cached(B, r, lead, &why); }
        if (uniform && batch_refresh_args(B)) rc = 1;
        if (!rc) {
            B->lockstep = uniform;
            for (int r = 0; r < R && !rc; r++) if (!B->failed[r] && flush_program(B->eng[r])) { B->err = B->eng[r]->err; rc = 1; }
            B->lockstep = false;
        }
    }
    B->active = saved_active; B->leader = B->eng[0];
    batch_leave(B);
    return rc;
}

// In-situ timing of the nonbonded force kernel: `every` > 0 brackets every `every`-th lock-step force launch of the stepping
// loop with two HIP events on the stream it is launched on (0: off; statistics are reset either way).
int blues_batch_kernel_timing(BluesBatch* b, int32_t every) {
    if (!b || every < 0) return 2;
    if (hipSetDevice(b->eng[0]->device) != hipSuccess) { b->err = "hipSetDevice failed"; return 1; }
    k1t_harvest(b, true);
    b->k1t_every = every; b->k1t_seen = 0; b->k1t_n = 0; b->k1t_sum_us = 0.0; b->k1t_max_us = 0.0;
    if (every > 0 && b->k1t_pairs.empty()) {
        b->k1t_pairs.resize(32);
        for (auto& p : b->k1t_pairs) if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) { b->err = "hipEventCreate failed"; return 1; }
    }
    return 0;
}
// out[0] = mean duration (us) of the sampled launches, out[1] = how many were sampled, out[2] = the longest of them
int blues_batch_get_kernel_timing(BluesBatch* b, double out[3]) {
    if (!b || !out) return 2;
    if (hipSetDevice(b->eng[0]->device) != hipSuccess) { b->err = "hipSetDevice failed"; return 1; }
    k1t_harvest(b, true);
    out[0] = b->k1t_n ? b->k1t_sum_us / (double)b->k1t_n : 0.0; out[1] = (double)b->k1t_n; out[2] = b->k1t_max_us;
    return 0;
}

// usec[0]: a pass over current pruned lists; usec[1]: a pass that re-derives them (walks the full lists); equal where the
// kernel has no pruned lists.  frac: the share of prune passes among this batch's force passes so far.
int blues_batch_time_nonbonded_modes(BluesBatch* b, int32_t reps, double usec[2], double* prune_fraction) {
    if (!b || b->eng.empty()) return 2;
    BluesEngine* h = b->eng[0];
    if (hipSetDevice(h->device) != hipSuccess) { b->err = "hipSetDevice failed"; return 1; }
    for (BluesEngine* m : b->eng) { if (flush_program(m) || ensure_sorted(m)) { b->err = m->err; return 1; } m->pass_valid = false; }
    {   // history of this batch: prune passes / force passes (before the timing launches add to the counters)
        double np = 0.0, nf = 0.0;
        for (BluesEngine* m : b->eng) {
            DevFlags f; hipStreamSynchronize(m->stream);
            if (hipMemcpy(&f, m->d_flags.p, sizeof f, hipMemcpyDeviceToHost) == hipSuccess) np += f.prunes;
            nf += (double)m->st_passes;
        }
        int n_i = 0; for (int i : h->mobile) if (h->T->alch_local[i] < 0) n_i++;
        if (prune_fraction) *prune_fraction = (h->prune_on && h->k1_mode == 2) ? (nf > 0.0 && n_i > 0 ? std::min(1.0, np / (nf * n_i)) : 1.0) : 0.0;
    }
    b->failed.assign(b->R(), 0); b->active.assign(b->R(), 1);
    if (batch_enter(b)) return 1;
    if (batch_refresh_args(b)) { batch_leave(b); return 1; }
    b->leader = h; b->lockstep = true;
    const bool dual = h->prune_on && h->k1_mode == 2;
    if (ensure_timing(h)) { b->err = h->err; batch_leave(b); return 1; }
    int rc = h->precision == 0 ? launch_lists<float>(h, 1) : launch_lists<double>(h, 1);
    float ms[2] = {0.f, 0.f};
    for (int mode = 0; mode < 2 && !rc; mode++) {   // 0: pruned lists current, 1: stale (every launch prunes)
        for (int w = 0; w < 3 && !rc; w++) rc = h->precision == 0 ? launch_nonbonded<float, false>(h) : launch_nonbonded<double, false>(h);
        if (dual) hipLaunchKernelGGL(k_prune_set, dim3(1, b->R()), dim3(256), 0, h->cur, b->d_core.p, (int*)nullptr, h->n_islots, 2 * mode);   // 2: the kernel leaves the flag up
        if (!rc && hipEventRecord(h->ev0, h->cur) != hipSuccess) rc = 1;
        for (int r = 0; r < reps && !rc; r++) rc = h->precision == 0 ? launch_nonbonded<float, false>(h) : launch_nonbonded<double, false>(h);
        if (!rc && hipEventRecord(h->ev1, h->cur) != hipSuccess) rc = 1;
        if (rc || hipEventSynchronize(h->ev1) != hipSuccess) { rc = 1; break; }
        hipEventElapsedTime(&ms[mode], h->ev0, h->ev1);
    }
    if (dual && !rc) hipLaunchKernelGGL(k_prune_set, dim3(1, b->R()), dim3(256), 0, h->cur, b->d_core.p, (int*)nullptr, h->n_islots, 0);   // (the last launch pruned every list at these positions)
    b->lockstep = false;
    for (BluesEngine* m : b->eng) m->lists_forced = false;
    if (rc) { b->err = "timing launch failed: " + h->err; batch_leave(b); return 1; }
#ifdef BLUES_STAMP
    { long long st[64]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_nb_stamps), sizeof st);
      fprintf(stderr, "[stamps] nonbonded_atom_b, the workgroup that stamped last (10 ns ticks since its entry: slot records, image requested + mobile positions asked for, barrier, then wave 0's turns: walked / reduced ...):");
      for (int i = 17; i < 28; i++) fprintf(stderr, " %lld", st[i] - st[16]);
      fprintf(stderr, " | waves out of atoms at:"); for (int i = 44; i < 60; i++) fprintf(stderr, " %lld", st[i] - st[16]); fprintf(stderr, "\n");
      fprintf(stderr, "[stamps] alchemical_dense32_b, one workgroup (cycles since its entry: image staged; wave 0 done with its first, second atom; waves 0..7 out of atoms; end):");
      for (int i = 33; i < 44; i++) fprintf(stderr, " %lld", st[i] - st[32]); fprintf(stderr, " %lld\n", st[60] - st[32]); }
#endif
    usec[0] = 1000.0 * ms[0] / std::max(1, reps); usec[1] = 1000.0 * ms[1] / std::max(1, reps);
    batch_leave(b);
    for (BluesEngine* m : b->eng) if (check_flags(m)) { b->err = m->err; return 1; }
    return 0;
}

// mean duration of the batched nonbonded launch as this batch has been running it: the two kinds of pass weighted by how
// often each occurred
int blues_batch_time_nonbonded(BluesBatch* b, int32_t reps, double* usec) {
    double u[2], f = 0.0;
    const int rc = blues_batch_time_nonbonded_modes(b, reps, u, &f);
    if (rc) return rc;
    *usec = (1.0 - f) * u[0] + f * u[1];
    return 0;
}

}  // extern "C"
