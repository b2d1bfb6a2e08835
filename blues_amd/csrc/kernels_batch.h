// kernels_batch.h -- replica-batched launches: one launch advances R independent NCMC replicas.
//
// One replica of the flagship system is ~1 MB of state and a few hundred mobile atoms: every kernel of
// its step is latency-bound and occupies a few percent of the MI355X's 256 CUs.  Independent BLUES chains
// (the path shards by replica) therefore share launches: gridDim.y = number of replicas,
// blockIdx.y selects the replica's argument record from an array in HBM, blockIdx.x keeps its meaning
// inside the replica.  The bodies are the very device functions the single-replica kernels run, so a
// batched replica is bitwise identical to the same replica advanced alone.
//
// Replicas of one batch advance in lock step (same step program, same lambda index, same draw counter),
// so everything that changes from launch to launch is common to the batch and travels as ordinary
// kernel arguments (the *Dyn structs); everything that differs between replicas (buffers, Philox stream)
// is static and lives in the per-replica records, refreshed only when a replica re-sorts its atoms.
#pragma once
#include "kernels_integrate.h"
#include "kernels_pme.h"
#include "kernels_frag.h"

template <typename R> struct RepNb {   // precision-dependent part of a replica's argument record
    int active;   // 0: the member sits this launch out (its blocks return at once)
    ListArgs L; NbArgs<R> nb; NbConst<R> c; const typename Img<R>::Atom* img;
    PmeArgs<R> pme;   // reciprocal space (n = 0: none)
    FragArgs fr;      // fragment lists (kernels_frag.h; F = 0: not in that mode)
};
struct RepCore {
    int active;
    unsigned draw_delta;   // this member's O-substep draw counter minus the leader's (constant while they advance in lock step)
    AlchArgs al; BondedArgs bo; FinArgs fin; IntArgs in;
};

// XCD-aware block -> (replica, block-in-replica) map for the kernels that re-read a replica's atoms many times (nonbonded,
// alchemical).  The 8 XCDs of an MI355X each have their own L2 and take workgroups round-robin by linear id, so with a plain
// (x = block, y = replica) grid every replica's tile image would be pulled into all eight L2s.  Launched 1-D with nb*R
// blocks, linear id L runs on XCD L % 8; V = (L % 8) * (total / 8) + L / 8 walks each XCD's share contiguously, and
// replica = V / nb keeps all blocks of a replica on one XCD (R a multiple of 8; otherwise the identity map).
__device__ __forceinline__ void batch_decode(int nb, int R, int& rep, int& bx) {
    const int L = blockIdx.x, total = nb * R;
    const int V = (R & 7) == 0 ? (L & 7) * (total >> 3) + (L >> 3) : L;
    rep = V / nb; bx = V - rep * nb;
}

struct AlchDyn { double ls[3], le[3]; int slot_mask; };
struct BondedDyn { unsigned draw_base; int n_entry_blocks; };
struct FinDyn { double le[3]; int slot_mask; };
struct IntDyn { unsigned draw_base, noise_draw_base; int n_noise, trace_index, tracing; Program prog; double fin_le[3]; int fin_mask; };   // fin_*: the pass whose forces a fused step kernel forms itself

__device__ __forceinline__ void apply_dyn(AlchArgs& A, const AlchDyn& d) {
#pragma unroll
    for (int s = 0; s < 3; s++) { A.ls[s] = d.ls[s]; A.le[s] = d.le[s]; }
    A.slot_mask = d.slot_mask; A.ctrl = nullptr;
}
__device__ __forceinline__ void apply_dyn(BondedArgs& B, const BondedDyn& d, unsigned delta) { B.draw_base = d.draw_base + delta; B.n_entry_blocks = d.n_entry_blocks; B.ctrl = nullptr; }
__device__ __forceinline__ void apply_dyn(FinArgs& F, const FinDyn& d) {
#pragma unroll
    for (int s = 0; s < 3; s++) F.le[s] = d.le[s];
    F.slot_mask = d.slot_mask; F.ctrl = nullptr;
}
__device__ __forceinline__ void apply_dyn(IntArgs& A, const IntDyn& d, unsigned delta) {
    A.draw_base = d.draw_base + delta; A.noise_draw_base = d.noise_draw_base + delta; A.n_noise = d.n_noise; A.trace_index = d.trace_index;
    if (!d.tracing) A.work_trace = nullptr;
    A.ctrl = nullptr;
    A.fin_le[0] = d.fin_le[0]; A.fin_le[1] = d.fin_le[1]; A.fin_le[2] = d.fin_le[2]; A.fin_mask = d.fin_mask;
}

// Work list of the rebuild kernels.  In a large batch a few members rebuild their lists in any given round (each chain trips
// its own displacement trigger).  A grid with one workgroup per (member, tile) then consists almost entirely of workgroups
// that find nothing to do, each a 1024-thread workgroup that wants ~100 KB of LDS and is alone on its CU while it finds out
// (R = 512: 5,120 of them, twenty rounds of dispatch for ~200 with work).  Instead one small kernel writes the ordered list of
// the members that rebuild (work[0] = how many, work[1..] = which) and the rebuild kernels run a SMALL grid whose workgroups
// deal the (member, tile) items of those members among themselves.  The alchemical kernel uses the same list to run in two
// parts (k_alchemical_b): the members that do not rebuild need nothing from the rebuild and start beside it.
template <typename R>
__global__ void __launch_bounds__(LIST_THREADS) k_gather_stale_b(const RepNb<R>* __restrict__ reps, int nrep, int force, int* __restrict__ work) {   // work[1 + nrep + r] = member r rebuilds (0 / 1)
    __shared__ int s_cnt[LIST_WAVES]; __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int r0 = 0; r0 < nrep; r0 += LIST_THREADS) {
        const int r = r0 + tid;
        bool act = false;
        if (r < nrep && reps[r].active) {
            const ListArgs& L = reps[r].L;
            act = force || L.flags->list_gen != L.flags->req_gen || (L.batch_req && *L.batch_req);
        }
        const unsigned long long bal = __ballot(act);
        if (lane == 0) s_cnt[wv] = __popcll(bal);
        __syncthreads();
        int rank = s_base + __popcll(bal & ((1ull << lane) - 1ull));
        for (int w = 0; w < wv; w++) rank += s_cnt[w];
        if (act) work[1 + rank] = r;
        if (r < nrep) work[1 + nrep + r] = act ? 1 : 0;
        __syncthreads();
        if (tid == 0) { int tot = 0; for (int w = 0; w < LIST_WAVES; w++) tot += s_cnt[w]; s_base += tot; }
        __syncthreads();
    }
    if (tid == 0) work[0] = s_base;
}

template <typename R>
__global__ void __launch_bounds__(LIST_THREADS) k_build_lists_b(const RepNb<R>* __restrict__ reps, const int* __restrict__ work, int bpc, int force) {
    const int nact = work[0];
    for (int i = blockIdx.x; i < nact * bpc; i += gridDim.x) {
        const int m = i / bpc;
        const RepNb<R>& rp = reps[work[1 + m]];
        const ListArgs a = rp.L; const NbConst<R> c = rp.c;
        build_lists_body<R>(a, c, rp.img, force, i - m * bpc, bpc);
        __syncthreads();   // (the next item reuses the body's LDS)
    }
}

template <typename R>
__global__ void __launch_bounds__(ATOM_LIST_THREADS) k_build_atom_lists_b(const RepNb<R>* __restrict__ reps, const int* __restrict__ work, int bpc, int force) {
    const int nact = work[0];
    for (int i = blockIdx.x; i < nact * bpc; i += gridDim.x) {
        const int m = i / bpc;
        const RepNb<R>& rp = reps[work[1 + m]];
        const ListArgs a = rp.L; const NbConst<R> c = rp.c;
        build_atom_lists_body<R>(a, c, rp.img, force, i - m * bpc);
        __syncthreads();
    }
}

template <typename R, bool ENERGY, int WPB>
__global__ void __launch_bounds__(WPB * 64) k_nonbonded_b(const RepNb<R>* __restrict__ reps, int nb, int nrep) {
    int rep, bx; batch_decode(nb, nrep, rep, bx);
    const RepNb<R>& rp = reps[rep];
    if (!rp.active) return;
    const NbArgs<R> a = rp.nb; const NbConst<R> c = rp.c;
    nonbonded_body<R, ENERGY, WPB>(a, c, rp.img, bx);
}

template <bool ENERGY, int IW>
__global__ void __launch_bounds__(256) k_nonbonded_sub_b(const RepNb<float>* __restrict__ reps, int nb, int nrep) {
    int rep, bx; batch_decode(nb, nrep, rep, bx);
    const RepNb<float>& rp = reps[rep];
    if (!rp.active) return;
    const NbArgs<float> a = rp.nb; const NbConst<float> c = rp.c;
    nonbonded_sub_body<ENERGY, IW>(a, c, rp.img, bx);
}

template <bool ENERGY>
__global__ void __launch_bounds__(1024) k_nonbonded_atom_b(const RepNb<float>* __restrict__ reps, int nb, int nrep) {
    int rep, bx; batch_decode(nb, nrep, rep, bx);
    const RepNb<float>& rp = reps[rep];
    if (!rp.active) return;
    const NbArgs<float> a = rp.nb; const NbConst<float> c = rp.c;
    nonbonded_atom_body<ENERGY>(a, c, rp.img, bx);
}

// ---- fragment lists (kernels_frag.h): every member's first kernel of a pass, the work list of the members whose lists need
// the second one (stale: rebuild; an atom past the prune trigger: prune), that kernel over the work list, the force kernel
__global__ void __launch_bounds__(FR_THREADS) k_frag_pre_b(const RepNb<float>* __restrict__ reps, int nb, int nrep, int force, int with_boxes) {
    const int rep = blockIdx.x / nb, bx = blockIdx.x - rep * nb;
    const RepNb<float>& rp = reps[rep];
    if (!rp.active) return;
    const FragArgs fa = rp.fr;
    const int blk = bx * (FR_THREADS / 64) + (threadIdx.x >> 6);
    frag_pre_body(fa, rp.img, force, blk);
    if (with_boxes) frag_boxes_body(fa, rp.img, force, blk);   // (k_frag_pre: the rebuild's first kernel rides along)
}
// (the rebuild's first kernel over the same work list: a member that is only there to prune returns at once)
__global__ void __launch_bounds__(FR_THREADS) k_frag_boxes_b(const RepNb<float>* __restrict__ reps, const int* __restrict__ work, int bpc, int force) {
    const int nact = work[0];
    for (int i = blockIdx.x; i < nact * bpc; i += gridDim.x) {
        const int m = i / bpc;
        const RepNb<float>& rp = reps[work[1 + m]];
        const FragArgs fa = rp.fr;
        frag_boxes_body(fa, rp.img, force, (i - m * bpc) * (FR_THREADS / 64) + (threadIdx.x >> 6));
    }
}
__global__ void __launch_bounds__(LIST_THREADS) k_gather_frag_b(const RepNb<float>* __restrict__ reps, int nrep, int force, int* __restrict__ work) {
    __shared__ int s_cnt[LIST_WAVES]; __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int r0 = 0; r0 < nrep; r0 += LIST_THREADS) {
        const int r = r0 + tid;
        bool act = false;
        if (r < nrep && reps[r].active) { const FragArgs& fa = reps[r].fr; act = frag_stale(fa, force) || fa.flags->prune_req != 0; }
        const unsigned long long bal = __ballot(act);
        if (lane == 0) s_cnt[wv] = __popcll(bal);
        __syncthreads();
        int rank = s_base + __popcll(bal & ((1ull << lane) - 1ull));
        for (int w = 0; w < wv; w++) rank += s_cnt[w];
        if (act) work[1 + rank] = r;
        __syncthreads();
        if (tid == 0) { int tot = 0; for (int w = 0; w < LIST_WAVES; w++) tot += s_cnt[w]; s_base += tot; }
        __syncthreads();
    }
    if (tid == 0) work[0] = s_base;
}
__global__ void __launch_bounds__(FR_THREADS) k_frag_lists_b(const RepNb<float>* __restrict__ reps, const int* __restrict__ work, int bpc, int force) {
    const int nact = work[0];
    for (int i = blockIdx.x; i < nact * bpc; i += gridDim.x) {
        const int m = i / bpc;
        const RepNb<float>& rp = reps[work[1 + m]];
        const FragArgs fa = rp.fr;
        frag_lists_body(fa, force, i - m * bpc);
    }
}
template <bool ENERGY, bool REL>
__global__ void __launch_bounds__(FR_THREADS, FR_MIN_WAVES) k_nonbonded_frag_b(const RepNb<float>* __restrict__ reps, int nb, int nrep) {
    int rep, bx; batch_decode(nb, nrep, rep, bx);
    const RepNb<float>& rp = reps[rep];
    if (!rp.active) return;
    const FragArgs fa = rp.fr; const NbConst<float> c = rp.c;
    nonbonded_frag_body<ENERGY, REL>(fa, c, bx);
}

// reciprocal space of every member: one workgroup each (kernels_pme.h)
template <typename R>
__global__ void __launch_bounds__(PME_THREADS) k_pme_b(const RepNb<R>* __restrict__ reps, int want_energy) {
    const RepNb<R>& rp = reps[blockIdx.x];
    if (!rp.active) return;
    PmeArgs<R> P = rp.pme; P.want_energy = want_energy & 1; if (want_energy & 2) P.qn = P.qn_full;
    pme_body<R, false, 5>(P);
}

__global__ void __launch_bounds__(PME_THREADS) k_pme_fast_b(const RepNb<float>* __restrict__ reps, int want_energy) {
    const RepNb<float>& rp = reps[blockIdx.x];
    if (!rp.active) return;
    PmeArgs<float> P = rp.pme; P.want_energy = want_energy & 1; if (want_energy & 2) P.qn = P.qn_full;
    pme_fast_body<5>(P);
}

// the members' mesh energies (PmeArgs::epart) side by side: one read-back for the batch (blues_batch_mesh_energy)
template <typename R>
__global__ void __launch_bounds__(256) k_gather_pme_e_b(const RepNb<R>* __restrict__ reps, int nrep, double* __restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < nrep) out[r] = reps[r].active && reps[r].pme.epart ? reps[r].pme.epart[0] : 0.0;
}

// stale: null = every member; else the work list of the rebuild (k_gather_stale_b): the members that rebuild are left to
// k_alchemical_stale_b, which follows the rebuild of the group lists (the alchemical tile's records are among them)
template <bool FAST, int MASK>
__global__ void __launch_bounds__(256, 3) k_alchemical_b(const RepCore* __restrict__ reps, AlchDyn d, int nb, int nrep, const int* __restrict__ stale) {
    int rep, bx; batch_decode(nb, nrep, rep, bx);
    if (!reps[rep].active) return;
    if (stale && stale[1 + nrep + rep]) return;
    AlchArgs A = reps[rep].al; apply_dyn(A, d);
    alchemical_blocks<FAST, MASK>(A, bx, nb - 1);
}

template <bool FAST, int MASK>
__global__ void __launch_bounds__(256, 3) k_alchemical_stale_b(const RepCore* __restrict__ reps, AlchDyn d, int nb, const int* __restrict__ work) {
    const int nact = work[0];
    for (int i = blockIdx.x; i < nact * nb; i += gridDim.x) {
        const int m = i / nb;
        AlchArgs A = reps[work[1 + m]].al; apply_dyn(A, d);
        alchemical_blocks<FAST, MASK>(A, i - m * nb, nb - 1);
        __syncthreads();   // (the next item reuses the LDS staging)
    }
}

// dense form of the env pairs (kernels_alch.h: alchemical_dense_body): one workgroup per member
template <int MASK>
__global__ void __launch_bounds__(K2D_THREADS) k_alchemical_dense_b(const RepCore* __restrict__ reps, AlchDyn d) {
    const int rep = blockIdx.x;
    if (!reps[rep].active) return;
    AlchArgs A = reps[rep].al; apply_dyn(A, d);
    alchemical_dense_body<MASK>(A);
}

// ... and its fp32 form (round 6, kernels_alch.h: alchemical_dense32_body): two workgroups per CU
// stale / want (BluesTuning.fork = 4): null = every member; else the work list of the rebuild (k_gather_stale_b) and which side of
// it this launch serves -- 0: the members that do not rebuild their lists in this pass (they need nothing from the rebuild and run
// beside it), 1: those that do (behind the builder of the group lists, which leaves the alchemical tile's list)
template <int MASK>
__global__ void __launch_bounds__(K2F_THREADS, 4) k_alchemical_dense32_b(const RepCore* __restrict__ reps, AlchDyn d, const int* __restrict__ stale, int want) {
    const int rep = blockIdx.x;
    if (!reps[rep].active) return;
    if (stale && stale[1 + gridDim.x + rep] != want) return;
    AlchArgs A = reps[rep].al; apply_dyn(A, d);
    alchemical_dense32_body<MASK>(A);
}

__global__ void __launch_bounds__(128) k_bonded_entries_b(const RepCore* __restrict__ reps, BondedDyn d) {
    if (!reps[blockIdx.y].active) return;
    BondedArgs B = reps[blockIdx.y].bo; apply_dyn(B, d, reps[blockIdx.y].draw_delta);
    bonded_entries_body(B, blockIdx.x, 128);
}

template <bool LEAN>
__global__ void __launch_bounds__(256) k_finalize_b(const RepCore* __restrict__ reps, FinDyn d) {
    if (!reps[blockIdx.y].active) return;
    FinArgs F = reps[blockIdx.y].fin; apply_dyn(F, d);
    finalize_body<LEAN>(F);
}

template <typename R>
__global__ void __launch_bounds__(256) k_forces_fused_b(const RepNb<R>* __restrict__ rnb, const RepCore* __restrict__ reps, AlchDyn da, BondedDyn db, int nb1, int nb2) {
    if (!reps[blockIdx.y].active) return;
    const int b = blockIdx.x;
    if (b < nb1) {
        const RepNb<R>& rp = rnb[blockIdx.y];
        const NbArgs<R> a = rp.nb; const NbConst<R> c = rp.c;
        nonbonded_body<R, false, 4>(a, c, rp.img, b);
        return;
    }
    if (b < nb1 + nb2) { AlchArgs A = reps[blockIdx.y].al; apply_dyn(A, da); alchemical_body<sizeof(R) == 4>(A, b - nb1); return; }
    BondedArgs B = reps[blockIdx.y].bo; apply_dyn(B, db, reps[blockIdx.y].draw_delta);
    bonded_entries_body(B, b - nb1 - nb2, 256);
}

template <int IW>
__global__ void __launch_bounds__(256) k_forces_fused_sub_b(const RepNb<float>* __restrict__ rnb, const RepCore* __restrict__ reps, AlchDyn da, BondedDyn db, int nb1, int nb2) {
    if (!reps[blockIdx.y].active) return;
    const int b = blockIdx.x;
    if (b < nb2) { AlchArgs A = reps[blockIdx.y].al; apply_dyn(A, da); alchemical_body<true>(A, b); return; }
    if (b < nb2 + nb1) {
        const RepNb<float>& rp = rnb[blockIdx.y];
        const NbArgs<float> a = rp.nb; const NbConst<float> c = rp.c;
        nonbonded_sub_body<false, IW>(a, c, rp.img, b - nb2);
        return;
    }
    BondedArgs B = reps[blockIdx.y].bo; apply_dyn(B, db, reps[blockIdx.y].draw_delta);
    bonded_entries_body(B, b - nb1 - nb2, 256);
}

__global__ void __launch_bounds__(256) k_integrate_b(const RepCore* __restrict__ reps, IntDyn d) {
    if (!reps[blockIdx.y].active) return;
    IntArgs A = reps[blockIdx.y].in; apply_dyn(A, d, reps[blockIdx.y].draw_delta);
    integrate_body(A, d.prog);
}

template <bool CM, bool FUSED = false>
__global__ void __launch_bounds__(256) k_step_default_b(const RepCore* __restrict__ reps, IntDyn d) {
    if (!reps[blockIdx.y].active) return;
    IntArgs A = reps[blockIdx.y].in; apply_dyn(A, d, reps[blockIdx.y].draw_delta);
    step_default_body<CM, FUSED>(A);
}
// two waves per SIMD (<= 256 registers): the late-load form of the body (kernels_integrate.h); same arithmetic, same bits
template <bool CM>
__global__ void __launch_bounds__(256, 2) k_step_default_late_b(const RepCore* __restrict__ reps, IntDyn d) {
    if (!reps[blockIdx.y].active) return;
    IntArgs A = reps[blockIdx.y].in; apply_dyn(A, d, reps[blockIdx.y].draw_delta);
    step_default_body<CM, true, true>(A);
}

template <bool CM>
__global__ void __launch_bounds__(256) k_step_md_b(const RepCore* __restrict__ reps, IntDyn d) {
    if (!reps[blockIdx.y].active) return;
    IntArgs A = reps[blockIdx.y].in; apply_dyn(A, d, reps[blockIdx.y].draw_delta);
    step_md_body<CM>(A);
}

// one word per member for the host's periodic poll: bit 0 resort_hint, bit 1 any error flag (one read-back for the batch)
__global__ void k_gather_hints_b(const RepCore* __restrict__ reps, int R, int* out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    int w = 0;
    if (reps[r].active) {
        const DevFlags* f = reps[r].in.flags;
        w = (f->resort_hint ? 1 : 0) | ((f->list_overflow | f->nan_flag | f->constraint_fail) ? 2 : 0);
    }
    out[r] = w;
}

// timing harness only (blues_time_nonbonded*): put the pruned lists of every member (or of one lone engine) into the "stale" or
// the "current" state, so that the launches that follow are all prune passes or all pruned-list passes
__global__ void k_prune_set(const RepCore* __restrict__ reps, int* one, int n_islots, int raise) {
    int* f = reps ? reps[blockIdx.y].in.pneed : one;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n_islots; q += gridDim.x * blockDim.x) f[q] = raise;
}

// ---- the plugin boundary for a whole batch (blues_batch_capture / _restore / _read_atoms / _restore_edited / _reset /
// _set_velocities_to_temperature): the per-member kernels of the single-engine entry points, one launch for all members.
// The per-member arguments are plain structs uploaded as an array; a member that sits the call out has n = 0 / count = 0.
__global__ void __launch_bounds__(256) k_copy_arrays_b(const Copy6Args* __restrict__ args) {
    const Copy6Args a = args[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    for (int q = 0; q < a.count; q++) a.dst[q][(size_t)i * a.dst_stride] = a.src[q][(size_t)i * a.src_stride];
}
__global__ void k_zero_xfer_b(const LoadPosArgs* __restrict__ args, int R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R && args[r].n > 0 && !args[r].keep_out) { unsigned* o = args[r].out; o[0] = o[1] = o[2] = o[3] = 0u; }
}
__global__ void __launch_bounds__(256) k_load_positions_b(const LoadPosArgs* __restrict__ args) {
    const LoadPosArgs a = args[blockIdx.y];
    if (a.n <= 0) return;
    load_positions_body(a, blockIdx.x);
}
__global__ void __launch_bounds__(64) k_edit_positions_b(const EditPosArgs* __restrict__ args) {
    const EditPosArgs a = args[blockIdx.y];
    edit_positions_body(a, blockIdx.x);
}
// out[r][e][3] <- the atoms idx[e] of member r's arrays src[3r + k]
__global__ void __launch_bounds__(64) k_gather_atoms_b(int n_idx, const int* __restrict__ idx, const double* const* __restrict__ src, double* out) {
    const int e = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y;
    if (e >= n_idx || !src[3 * r]) return;
    const int i = idx[e];
    double* o = out + ((size_t)r * n_idx + e) * 3;
    o[0] = src[3 * r][i]; o[1] = src[3 * r + 1][i]; o[2] = src[3 * r + 2][i];
}
__global__ void k_gather_xfer_b(const LoadPosArgs* __restrict__ args, int R, unsigned* out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    for (int q = 0; q < 4; q++) out[4 * r + q] = args[r].n > 0 ? args[r].out[q] : 0u;
}
struct MaxwellArgs { int n; const double* mass; double* v[3]; double kT; unsigned long long seed; unsigned stream; };
__global__ void __launch_bounds__(256) k_maxwell_b(const MaxwellArgs* __restrict__ args) {
    const MaxwellArgs a = args[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    double g[3] = {0.0, 0.0, 0.0};
    if (a.mass[i] != 0.0) { gaussians3(a.seed, a.stream, 0u, (unsigned)i, g); const double s = sqrt(a.kT / a.mass[i]); g[0] *= s; g[1] *= s; g[2] *= s; }
    a.v[0][i] = g[0]; a.v[1][i] = g[1]; a.v[2][i] = g[2];
}
// protocol_work += delta of every member that has one (the work of an instantaneous Move, booked at the head of the next step:
// reference blues/integrators.py:184-191) -- one launch instead of one per member
__global__ void k_add_work_b(DevAccum* const* __restrict__ acc, const double* __restrict__ delta, int R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R && acc[r]) acc[r]->protocol_work += delta[r];
}
__global__ void k_zero_acc_b(DevAccum* const* __restrict__ acc, int R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R && acc[r]) { DevAccum z; z.protocol_work = z.dE_last = z.heat = 0.0; z.e_slot[0] = z.e_slot[1] = z.e_slot[2] = 0.0; *acc[r] = z; }
}

__global__ void __launch_bounds__(256) k_bonded_energy_b(const RepCore* __restrict__ reps) {
    if (!reps[blockIdx.y].active) return;
    const BondedArgs B = reps[blockIdx.y].bo;
    bonded_energy_body(B);
}

// kinetic energy of every member: one block each, fixed summation order -> ke[r]
__global__ void __launch_bounds__(256) k_kinetic_b(const RepCore* __restrict__ reps, double* ke) {
    const RepCore& rp = reps[blockIdx.x];
    if (!rp.active) return;
    const IntArgs& A = rp.in;
    double s = 0.0;
    for (int i = threadIdx.x; i < A.n; i += 256) { const double m = A.mass[i]; if (m != 0.0) s += 0.5 * m * (A.v[0][i] * A.v[0][i] + A.v[1][i] * A.v[1][i] + A.v[2][i] * A.v[2][i]); }
    __shared__ double red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) ke[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    // ... and the member's accumulators (protocol work): the Metropolis test reads them next, chain by chain
    if (threadIdx.x < 6) ke[gridDim.x + 6 * blockIdx.x + threadIdx.x] = reinterpret_cast<const double*>(A.acc)[threadIdx.x];
}

// energy partials of every member into one slab [R][stride]: nonbonded partials | bonded partials | alchemical partials |
// the alchemical tile's list length -- one read-back for the whole batch; the host sums them as it does for a lone engine
// The frozen-frozen energy constant of the members that have none (their frozen atoms were given new positions: a State from the MD leg,
// reference blues/simulation.py:1028-1037, does that to every chain in every iteration): blockIdx.y walks the work list, blockIdx.x the
// member's tiles; then one thread per listed member adds its tiles' partials up in tile order (the order of the lone engine's host loop).
template <typename R>
__global__ void __launch_bounds__(256) k_frozen_boxes_b(const RepNb<R>* __restrict__ rnb, const int* __restrict__ work, int off_frozen) {
    const RepNb<R>& rp = rnb[work[blockIdx.y]];
    frozen_boxes_body<R>(rp.L.n, rp.c, rp.img, reinterpret_cast<FrozenBox<R>*>(rp.nb.epart + off_frozen + 2 * ((rp.L.n + FROZEN_TILE - 1) / FROZEN_TILE)), blockIdx.x);
}
template <typename R>
__global__ void __launch_bounds__(256) k_energy_frozen_b(const RepNb<R>* __restrict__ rnb, const int* __restrict__ work, int off_frozen) {
    const RepNb<R>& rp = rnb[work[blockIdx.y]];
    energy_frozen_body<R>(rp.L.n, rp.c, rp.img, rp.L.ex_start, rp.L.ex_idx, rp.nb.epart + off_frozen, blockIdx.x);
}
template <typename R>
__global__ void __launch_bounds__(64) k_sum_frozen_b(const RepNb<R>* __restrict__ rnb, const int* __restrict__ work, int count, int off_frozen, int nfb, double* __restrict__ out) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= count) return;
    const double* ep = rnb[work[q]].nb.epart + off_frozen;
    double a = 0.0, b = 0.0;
    for (int t = 0; t < nfb; t++) { a += ep[2 * t]; b += ep[2 * t + 1]; }
    out[2 * q] = a; out[2 * q + 1] = b;
}

template <typename R>
__global__ void __launch_bounds__(256) k_gather_energy_parts_b(const RepNb<R>* __restrict__ rnb, const RepCore* __restrict__ reps, int n_nb, int n_b, int n_alch, int stride, double* out) {
    const int r = blockIdx.x;
    if (!reps[r].active) return;
    double* o = out + (size_t)r * stride;
    const double* enb = rnb[r].nb.epart; const double* eb = reps[r].bo.epart; const double* ep = reps[r].al.e_part;
    for (int k = threadIdx.x; k < n_nb; k += 256) o[k] = enb[k];
    for (int k = threadIdx.x; k < n_b; k += 256) o[n_nb + k] = eb[k];
    for (int k = threadIdx.x; k < n_alch; k += 256) o[n_nb + n_b + k] = ep[k];
    if (threadIdx.x == 0) o[n_nb + n_b + n_alch] = n_alch > 0 ? (double)*reps[r].al.jcount : 0.0;
    if (threadIdx.x == 1 && stride > n_nb + n_b + n_alch + 1) o[n_nb + n_b + n_alch + 1] = rnb[r].pme.n > 0 ? rnb[r].pme.epart[0] : 0.0;   // mesh energy
}
