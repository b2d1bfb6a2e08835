// kernels_nb.h -- neighbour-tile list build and the direct-space nonbonded kernel (K1).
//
// K1 replaces, for the environment (non-alchemical) atoms, what OpenMM's NonbondedForce
// computes when the CustomIntegrator program of reference blues/integrators.py:159-231
// asks for `f` / `energy`: 12-6 LJ + erfc(alpha r)/r Coulomb inside the cutoff with
// exclusions (openmmtools keeps that force for env x env with the alchemical atoms'
// charge and epsilon zeroed -- SURVEY.md Appendix B).
//
// Work decomposition: an i-tile is 64 mobile atoms (one per lane of a wave64); its j-list
// holds every atom within cutoff+skin of the tile's bounding box.  A work item is
// (i-tile, chunk): the wave stages 64 j-atoms at a time into its private LDS slab and every
// lane walks them (broadcast LDS reads), accumulating the force on its own i-atom in fp64.
// No Newton's-third-law scatter, no atomics: chunk partials go to a slab that the
// integrator kernel sums in a fixed order, so results are bitwise reproducible.
#pragma once
#include "device_common.h"


struct DevFlags {  // device-resident control words
    unsigned list_gen, req_gen;  // neighbour lists are current iff equal
    int list_overflow;           // j-list or mask-pool capacity exceeded
    int constraint_fail;
    int nan_flag;
    int resort_hint;             // a j-list has grown to within 15 % of its capacity: the tiles have spread, re-sort soon
    int builds;                  // list builds executed (own requests, forced ones and batch-synchronised ones)
    int prunes;                  // pruned per-atom lists re-derived so far (one count per atom and prune; nonbonded_atom_body); fragment lists: prunes of the whole chain
    int prune_req;               // fragment lists (kernels_frag.h): an atom has moved half the inner margin since the last prune
};

struct ListArgs {
    int n, n_tiles, n_itiles, jcap, pool_cap;   // n_tiles: list blocks = n_lists (+1 with alchemical atoms)
    const int* tile_atoms;   // [n_tiles*64] sorted atom index or -1
    int* jlist;              // [n_tiles*jcap]
    int* jstage;             // [n_tiles][LIST_WAVES][share] per-wave staging for the ordered compaction (share: see build_lists_body)
    void* alch_jrec;         // AlchJRec[jcap] for the alchemical tile (kernels_alch.h), or null
    const double* p_sigma; const double* p_eps; const double* p_charge;  // caller order
    int* jcount;             // [n_tiles]
    int* batch_slot;         // [n_tiles*(jcap/64)]
    unsigned long long* mask_pool;  // [n_tiles][MASK_QUOTA][64]: a tile's masks live in its own region (no shared counter to reset)
    const int* batch_req;    // replica batches: "some member asked for a rebuild" -> every member rebuilds with it (or null)
    const int* ex_start;     // [n+1] exclusion CSR in sorted index space (self included)
    const int* ex_idx;
    DevFlags* flags;
    // bookkeeping done by the same launch when a rebuild happens
    const double* x[3];      // master positions (caller order)
    double* xbuild[3];
    double* fJ;              // [9*n] alchemical forces on environment atoms, zeroed on rebuild
    int n_fJ;
    // per-atom Verlet lists (nonbonded_atom_body): for every i-slot the LOCAL indices (positions in its tile's j-list) of
    // the atoms within cutoff+skin, exclusions already removed; bit 15 = that j is mobile (energy weight 1/2).  Null: not built.
    unsigned short* alist;   // [n_islots][acap]
    int* acount;             // [n_islots]
    int acap;
    int no_sphere;           // development: bounding-box test only
    int hint_count;          // a list longer than this raises resort_hint (the i-atoms have spread: re-derive the layout)
    int S, n_lists;          // S consecutive i-tiles share one j-list (1 in the bitmask modes); n_lists = ceil(n_itiles / S), the alchemical tile's list comes after them
    // pruned-list mode (NbArgs; null: none): the atom-list build writes every atom's pruned list along with the full one -- it
    // has the distances at hand -- so a rebuild leaves nothing for the nonbonded kernel to re-derive
    // (round 6 measured the other way round -- full lists only, every slot flagged, the force kernel prunes in its next pass: this
    // builder 87.5 -> 73.0 us per step of 1024 chains, the force kernel 222.5 -> 257.4: a workgroup whose chain re-derives every list
    // takes 1.6 x as long, and with one workgroup per CU the launch waits for those)
    int* pneed; unsigned short* plist; int* pcount; unsigned* xprune[3];
    // packed image of every group list (null: not kept): the list's atoms as the nonbonded kernel stages them, {x,y,z,q} and
    // {sigma/2, 2 sqrt(eps)}, written when the list is built -- frozen atoms never change, so the kernel copies it with
    // coalesced loads and refreshes only the MOBILE entries (mlist: pairs (list position, image index)) from the live image
    // Round 4: the stored sigma/2 carries "mobile" in its sign, and entry jcount of a list's image is a GHOST record (no charge,
    // no epsilon, 0.3 ... a few nm from every i-atom of the group) that pads the atoms' lists to whole chunks (nonbonded_atom_body)
    uint4* pimg4; float2* pimg2; int* mlist; int* mcount; int mcap;
    unsigned short* aself;   // [n_islots] every i-atom's own entry in its group's list (NB_ENT form; 0xffff: not there), written with the atoms' lists
    uint4* pimgb;            // [n_lists][jcap] what the builder of the atoms' lists streams: {x, y, z, sorted atom index | mobile << 31}
    const int* sx_row;       // [n_islots][SX_ROW] static per i-slot: the atom's sorted index, its excluded partners, count / min / max (build_atom_lists_body)
};

// LIST_WAVES waves share one tile's scan of all n atoms; each wave keeps LIST_PREFETCH independent loads in flight.
// The scan is a chain of dependent memory round trips (ballot -> running count -> store), so its duration is
// (n / 64 / LIST_WAVES / LIST_PREFETCH) latencies: 4 waves x 1 load took 60-70 us at n = 23,400; 16 x 4 takes ~12.
#define MASK_QUOTA 24   // exclusion-mask tiles per i-tile (j-batches that hold an excluded pair; a handful in practice)
#define LIST_WAVES 16
#ifndef LIST_PREFETCH
#define LIST_PREFETCH 4
#endif
#define LIST_THREADS (LIST_WAVES * 64)
#define ATOM_LIST_WAVES 4                   // waves per block of build_atom_lists_body: a 256-thread block with < 168 registers per
#define ATOM_LIST_THREADS (ATOM_LIST_WAVES * 64)   // lane fits exactly where a workgroup of the alchemical kernel retires
#define ATOM_LIST_U 2                       // atoms per wave (one atom per wave and twice the workgroups: 60 us against 57 at R = 512)
#define ATOM_LIST_PARTS (64 / (ATOM_LIST_WAVES * ATOM_LIST_U))    // its blocks per i-tile
#define ATOM_SLOT(part, wv, u) ((part) * (64 / ATOM_LIST_PARTS) + (wv) + ATOM_LIST_WAVES * (u))
#define LIST_LDS 8192      // j-list entries mirrored in LDS for the exclusion searches (longer lists are searched in HBM)
// per-atom-list mode (nonbonded_atom_body): layout of the dynamic LDS and of a list entry
#define NB_LQ_BYTES 51200   // room for 6400 {sigma/2, 2 sqrt(eps)} records at the bottom of the dynamic LDS; the {x,y,z,q} records follow
#define NB_JCAP_MAX 6400
#define NB_LDS_TAIL 512      // behind the images: the workgroup's atom counter and the energy kernel's per-wave sums; then 8 bytes per i-slot of the group
#define NB_ENT(k) ((unsigned short)((k) << 3))   // list entry of LDS index k
#define NB_IDX(e) ((int)(e) >> 3)

template <typename R>
__device__ __forceinline__ void build_lists_body(const ListArgs& a, const NbConst<R>& c, const typename Img<R>::Atom* __restrict__ img, const int force, const int t, const int nblocks) {
    using sfix = typename Img<R>::sfix;
    using ufix = typename Img<R>::ufix;
    if (!force && a.flags->list_gen == a.flags->req_gen && !(a.batch_req && *a.batch_req)) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // ---- shared bookkeeping: remember where the lists were built, clear alchemical env forces
    if (t >= a.n_tiles) {  // helper blocks: bookkeeping that must not sit on a tile block's critical path
        const int hb = t - a.n_tiles, nh = nblocks - a.n_tiles;
        for (int i = hb * LIST_THREADS + tid; i < a.n; i += nh * LIST_THREADS) {
            a.xbuild[0][i] = a.x[0][i]; a.xbuild[1][i] = a.x[1][i]; a.xbuild[2][i] = a.x[2][i];
        }
        for (int i = hb * LIST_THREADS + tid; i < a.n_fJ; i += nh * LIST_THREADS) a.fJ[i] = 0.0;
        return;
    }

    __shared__ ufix s_cfix[3];
    __shared__ double s_half[3];
    __shared__ int s_wcount[LIST_WAVES];
    __shared__ int s_total;
    __shared__ int s_need[256];
    __shared__ ufix s_ax[64][3];   // alchemical tile: its atoms' positions, for the exact per-atom range test
    __shared__ int s_na;
    __shared__ int s_nmask;
    const bool alch_tile = (t >= a.n_lists);
    // the i-atoms this list serves: tiles [tile0, tile0 + ntile) (the alchemical tile: one tile, index n_itiles)
    const int tile0 = alch_tile ? a.n_itiles : t * a.S;
    const int ntile = alch_tile ? 1 : min(a.S, a.n_itiles - tile0);
    NB_STAMP(t == 0 && tid == 0, 0);
    if (alch_tile && wv == 1) {
        const int ia = a.tile_atoms[tile0 * 64 + lane];
        if (ia >= 0) { s_ax[lane][0] = img[ia].x; s_ax[lane][1] = img[ia].y; s_ax[lane][2] = img[ia].z; }
        const unsigned long long have = __ballot(ia >= 0);
        if (lane == 0) s_na = __popcll(have);   // alchemical atoms fill the tile from slot 0
    }

    // ---- bounding box AND bounding sphere of the list's i-atoms, in fixed point relative to their first atom: a thread per
    // i-slot (ntile <= 16); an atom can only be within range of some i-atom if it is within cutoff+skin of both
    __shared__ double s_lo[LIST_WAVES][3], s_hi[LIST_WAVES][3], s_r2[LIST_WAVES];
    __shared__ double s_mid[3], s_rad;
    {
        const int i0 = a.tile_atoms[tile0 * 64];
        const ufix ref[3] = {img[i0].x, img[i0].y, img[i0].z};
        const int ia = tid < ntile * 64 ? a.tile_atoms[tile0 * 64 + tid] : -1;
        double off[3] = {0.0, 0.0, 0.0};
        if (ia >= 0) { off[0] = (double)(sfix)(img[ia].x - ref[0]) * c.dscale[0]; off[1] = (double)(sfix)(img[ia].y - ref[1]) * c.dscale[1]; off[2] = (double)(sfix)(img[ia].z - ref[2]) * c.dscale[2]; }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double lo = off[k], hi = off[k];   // empty slots sit on the first atom: neutral
            for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o, 64)); hi = fmax(hi, __shfl_xor(hi, o, 64)); }
            if (lane == 0) { s_lo[wv][k] = lo; s_hi[wv][k] = hi; }
        }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) {   // (constant indices: a runtime index into ref[] would put it in scratch memory)
                double lo = s_lo[0][k], hi = s_hi[0][k];
                for (int w = 1; w < LIST_WAVES; w++) { lo = fmin(lo, s_lo[w][k]); hi = fmax(hi, s_hi[w][k]); }
                const double mid = 0.5 * (lo + hi);
                s_mid[k] = mid;
                s_cfix[k] = ref[k] + (ufix)(sfix)llrint(mid / c.dscale[k]);
                s_half[k] = 0.5 * (hi - lo) + 2.0 * c.dscale[k];
            }
        }
        if (tid == 0) { s_total = 0; s_nmask = 0; }
        __syncthreads();
        double r2 = 0.0;
        if (ia >= 0) { const double dx = off[0] - s_mid[0], dy = off[1] - s_mid[1], dz = off[2] - s_mid[2]; r2 = dx * dx + dy * dy + dz * dz; }
        for (int o = 32; o > 0; o >>= 1) r2 = fmax(r2, __shfl_xor(r2, o, 64));
        if (lane == 0) s_r2[wv] = r2;
        __syncthreads();
        if (tid == 0) { double m = 0.0; for (int w = 0; w < LIST_WAVES; w++) m = fmax(m, s_r2[w]); s_rad = sqrt(m) + 4.0 * (c.dscale[0] + c.dscale[1] + c.dscale[2]); }
    }
    __syncthreads();
    const ufix cf[3] = {s_cfix[0], s_cfix[1], s_cfix[2]};
    const double hf[3] = {s_half[0], s_half[1], s_half[2]};

    // ---- ordered stream compaction of every atom within cutoff+skin of the box.  Each wave scans its own
    // contiguous share of the (Hilbert-sorted) atoms with no block barrier inside the loop and stages its
    // hits; the shares are then concatenated in order, so the list stays ascending.
    NB_STAMP(t == 0 && tid == 0, 1);
    int* jl = a.jlist + (size_t)t * a.jcap;
    const int share = (((a.n + LIST_WAVES - 1) / LIST_WAVES) + 63) & ~63;   // a wave cannot stage more than it scans
    int* stage = a.jstage + ((size_t)t * LIST_WAVES + wv) * share;
    const float cfs[3] = {(float)c.dscale[0], (float)c.dscale[1], (float)c.dscale[2]};
    const float hfl[3] = {(float)hf[0] * 1.00001f + 1e-6f, (float)hf[1] * 1.00001f + 1e-6f, (float)hf[2] * 1.00001f + 1e-6f};
    const float rl2f = (float)c.rlist2 * 1.0001f + 1e-5f, rl2m = (float)c.rlist2_m * 1.0001f + 1e-5f;   // frozen / mobile candidates
    const float rsphf = ((float)s_rad * 1.00001f + 1e-6f + sqrtf(rl2f)), rsphm = ((float)s_rad * 1.00001f + 1e-6f + sqrtf(rl2m));
    const float rs2f = a.no_sphere ? __builtin_inff() : rsphf * rsphf * 1.0001f, rs2m = a.no_sphere ? __builtin_inff() : rsphm * rsphm * 1.0001f;
    const int j_end = min(a.n, (wv + 1) * share);
    int wcount = 0;
    for (int base = wv * share; base < j_end; base += 64 * LIST_PREFETCH) {
        ufix px[LIST_PREFETCH], py[LIST_PREFETCH], pz[LIST_PREFETCH]; unsigned fl[LIST_PREFETCH];
#pragma unroll
        for (int u = 0; u < LIST_PREFETCH; u++) {   // all loads of the group are issued before the first is consumed
            const int j = min(base + u * 64 + lane, a.n - 1);
            px[u] = img[j].x; py[u] = img[j].y; pz[u] = img[j].z; fl[u] = img[j].flags;
        }
#pragma unroll
        for (int u = 0; u < LIST_PREFETCH; u++) {
            const int j = base + u * 64 + lane;
            const ufix pj[3] = {px[u], py[u], pz[u]};
            float d2 = 0.0f, o2 = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float o = (float)(sfix)(pj[k] - cf[k]) * cfs[k];
                o2 = fmaf(o, o, o2);
                float d = fabsf(o) - hfl[k];
                d = fmaxf(d, 0.0f);
                d2 = fmaf(d, d, d2);
            }
            const bool mob = (fl[u] & FLAG_MOBILE) != 0;
            const float rl2 = mob ? rl2m : rl2f, rs2 = mob ? rs2m : rs2f;
            bool pass = j < j_end && d2 < rl2 && o2 < rs2 && !(fl[u] & FLAG_ALCH);
            if (alch_tile && pass) {
                // the alchemical kernel spends a thread on every (j, alchemical atom) pair of this list, so the list is
                // made exact: j stays only if it is within cutoff+skin of at least one alchemical atom (the bounding
                // box of a 15-atom ligand plus the margin holds ~1.6x as many atoms as that union of spheres)
                bool near = false;
                for (int q = 0; q < s_na; q++) {
                    float e2 = 0.0f;
#pragma unroll
                    for (int k = 0; k < 3; k++) { const float e = (float)(sfix)(pj[k] - s_ax[q][k]) * cfs[k]; e2 = fmaf(e, e, e2); }
                    near |= e2 < rl2;
                }
                pass = near;
            }
            const unsigned long long bal = __ballot(pass);
            if (pass) stage[wcount + __popcll(bal & ((1ull << lane) - 1ull))] = j;
            wcount += __popcll(bal);
        }
    }
    if (lane == 0) s_wcount[wv] = wcount;
    NB_STAMP(t == 0 && tid == 0, 2);
    __syncthreads();
    NB_STAMP(t == 0 && tid == 0, 3);
    {
        int off = 0;
        for (int w = 0; w < wv; w++) off += s_wcount[w];
        for (int k = lane; k < wcount; k += 64) if (off + k < a.jcap) jl[off + k] = stage[k];
        if (tid == 0) { int tot = 0; for (int w = 0; w < LIST_WAVES; w++) tot += s_wcount[w]; s_total = tot; }
    }
    __threadfence_block();
    __syncthreads();
    int count = s_total;
    if (t == 0 && tid == 0) a.flags->builds++;
    const int cap_eff = (!alch_tile && a.alist && a.pimg4) ? a.jcap - 1 : a.jcap;   // (per-atom-list mode: the image keeps one entry for the ghost)
    if (count > cap_eff) { if (tid == 0) a.flags->list_overflow = 1; count = cap_eff; }
    else if (count > a.hint_count && tid == 0) a.flags->resort_hint = 1;
    if (tid == 0) a.jcount[t] = count;
    const int nb = (count + 63) >> 6;
    const int nbmax = a.jcap >> 6;
    for (int b = tid; b < nbmax; b += LIST_THREADS) { s_need[b] = 0; a.batch_slot[t * nbmax + b] = -1; }
    __syncthreads();
    if (alch_tile) {  // the alchemical kernel checks exclusions itself; give it packed per-entry records
        struct JR { int jo, jsrt; double sig, eps, q; };
        JR* jr = (JR*)a.alch_jrec;
        if (jr) for (int k = tid; k < count; k += LIST_THREADS) {
            const int js = jl[k], jo = img[js].orig;
            JR r; r.jo = jo; r.jsrt = js | ((img[js].flags & FLAG_MOBILE) ? 0x40000000 : 0);   // bit 30: j is mobile (its force is wanted)
            r.sig = a.p_sigma[jo]; r.eps = sqrt(a.p_eps[jo]); r.q = a.p_charge[jo];   // (eps: its square root, see AlchJRec)
            jr[k] = r;
        }
        return;
    }

    NB_STAMP(t == 0 && tid == 0, 4);
    if (a.alist && a.pimg4) {
        if constexpr (sizeof(ufix) == 4) {
            __shared__ int s_mc;
            if (tid == 0) s_mc = 0;
            __syncthreads();
            uint4* g4 = a.pimg4 + (size_t)t * a.jcap; float2* g2 = a.pimg2 + (size_t)t * a.jcap; int* ml = a.mlist + (size_t)t * a.mcap * 2;
            for (int k = tid; k < count; k += LIST_THREADS) {
                const int js = jl[k];
                const typename Img<R>::Atom A = img[js];
                const bool mob = (A.flags & FLAG_MOBILE) != 0;
                g4[k] = make_uint4((unsigned)A.x, (unsigned)A.y, (unsigned)A.z, __float_as_uint((float)A.q));
                a.pimgb[(size_t)t * a.jcap + k] = make_uint4((unsigned)A.x, (unsigned)A.y, (unsigned)A.z, (unsigned)js | (mob ? 0x80000000u : 0u));
                g2[k] = make_float2(mob ? -(float)A.hs : (float)A.hs, (float)A.se);   // (sign bit = mobile; sigma = 0 gives -0.0f, read with |.| and an integer test)
                if (mob) { const int q = atomicAdd(&s_mc, 1); if (q < a.mcap) { ml[2 * q] = k; ml[2 * q + 1] = js; } else a.flags->list_overflow = 1; }   // (order is irrelevant: the entries are only refreshed)
            }
            if (tid == 0) {
                // the ghost: no charge, no epsilon -- its pair force is exactly zero at ANY finite distance -- and placed where the
                // arithmetic stays finite: 0.5 nm outside the group's bounding sphere along the longest box edge, i.e. >= 0.3 nm from
                // every i-atom until the next rebuild (which comes before one has moved 0.2 nm) and a few nm at most.  Needs that edge
                // to exceed the group's diameter + 1 nm, or the point wraps back into the group (host: sort_and_tile only chooses
                // this kernel where it holds with room to spare; a group that has spread that far asks for a new layout)
                const double e0 = c.dscale[0], e1 = c.dscale[1], e2 = c.dscale[2];   // (box edge / 2^32; no runtime index into the argument record)
                const int kl = e0 >= e1 ? (e0 >= e2 ? 0 : 2) : (e1 >= e2 ? 1 : 2);
                const double el = kl == 0 ? e0 : (kl == 1 ? e1 : e2);
                if (2.0 * s_rad + 1.0 >= el * 4294967296.0) a.flags->resort_hint = 1;
                const unsigned off = (unsigned)(long long)((s_rad + 0.5) / el);
                const unsigned gp[3] = {(unsigned)cf[0] + (kl == 0 ? off : 0u), (unsigned)cf[1] + (kl == 1 ? off : 0u), (unsigned)cf[2] + (kl == 2 ? off : 0u)};
                g4[count] = make_uint4(gp[0], gp[1], gp[2], 0u); g2[count] = make_float2(0.0f, 0.0f);
            }
            __syncthreads();
            NB_STAMP(t == 0 && tid == 0, 9);
            if (tid == 0) a.mcount[t] = min(s_mc, a.mcap);
        }
    }
    if (a.alist) return;   // per-atom-list mode: exclusions are dropped when build_atom_lists_body forms the atoms' lists -- no bitmask tiles
    __shared__ int s_jl[LIST_LDS];

    // ---- exclusion bitmasks: one 64x64 bit tile per (i-tile, j-batch) that holds an excluded pair.
    // The binary searches run on a copy of the list in LDS (a dozen dependent steps each: ~100 ns instead of a memory
    // round trip per step) and the positions found in the first pass are kept for the second.
    const bool in_lds = count <= LIST_LDS;
    if (in_lds) for (int k = tid; k < count; k += LIST_THREADS) s_jl[k] = jl[k];
    const int ia = a.tile_atoms[t * 64 + lane];
    int e0 = 0, e1 = 0;
    if (ia >= 0) { e0 = a.ex_start[ia]; e1 = a.ex_start[ia + 1]; }
    __syncthreads();
    auto find = [&](int p) {   // position of sorted index p in the list, or -1
        int lo = 0, hi = count;
        if (in_lds) { while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_jl[mid] < p) lo = mid + 1; else hi = mid; } return (lo < count && s_jl[lo] == p) ? lo : -1; }
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (jl[mid] < p) lo = mid + 1; else hi = mid; }
        return (lo < count && jl[lo] == p) ? lo : -1;
    };
    int found0 = -1, found1 = -1;   // first two entries of this lane/wave (exclusion rows are short)
    {
        int u = 0;
        for (int e = e0 + wv; e < e1; e += LIST_WAVES, u++) {
            const int pos = find(a.ex_idx[e]);
            if (u == 0) found0 = pos; else if (u == 1) found1 = pos;
            if (pos >= 0) s_need[pos >> 6] = 1;
        }
    }
    __syncthreads();
    for (int b = tid; b < nb; b += LIST_THREADS) {
        if (s_need[b]) {
            int slot = atomicAdd(&s_nmask, 1);
            if (slot >= MASK_QUOTA) { a.flags->list_overflow = 1; slot = 0; }
            slot += t * MASK_QUOTA;
            s_need[b] = slot; a.batch_slot[t * nbmax + b] = slot;
        } else s_need[b] = -1;
    }
    __syncthreads();
    for (int b = wv; b < nb; b += LIST_WAVES) {
        const int slot = s_need[b];
        if (slot >= 0) a.mask_pool[(size_t)slot * 64 + lane] = 0ull;
    }
    __threadfence_block();
    __syncthreads();
    {
        int u = 0;
        for (int e = e0 + wv; e < e1; e += LIST_WAVES, u++) {
            const int pos = u == 0 ? found0 : (u == 1 ? found1 : find(a.ex_idx[e]));
            if (pos >= 0) atomicOr(&a.mask_pool[(size_t)s_need[pos >> 6] * 64 + lane], 1ull << (pos & 63));
        }
    }
}

template <typename R>
__global__ void __launch_bounds__(LIST_THREADS) k_build_lists(ListArgs a, NbConst<R> c, const typename Img<R>::Atom* __restrict__ img, int force) {
    build_lists_body<R>(a, c, img, force, blockIdx.x, gridDim.x);
}

// ---- per-atom Verlet lists (second kernel of a rebuild, per-atom-list mode): one block per PART of an i-tile (ATOM_LIST_PARTS
// parts; wave wv of part p serves ATOM_LIST_U slots, ATOM_SLOT).  Every wave walks the list of the tile's group once, 64
// candidates at a time, each lane testing one candidate against the wave's atoms (wave-uniform); hits are compacted in list
// order (ballot + prefix count), so an atom's list is ascending in the local index.
// Round 4: the walk reads ONE 16-byte record per candidate, {x, y, z, sorted atom index | mobile << 31} (pimgb, written by the
// first kernel of the rebuild beside the force kernel's image), ATOM_LIST_PF chunks ahead; the atom's excluded partners come
// as a STATIC row of sorted indices per i-slot (sx_row: the atom itself first, then its partners; laid out by the host at every
// sort) and are matched against the candidates' indices in the chunks whose index range reaches the row's -- no search, no
// bitmap, no LDS.  (Round 3 found every partner's place in the list by a binary search over global memory before the walk --
// a dozen DEPENDENT round trips per atom, two atoms one after the other -- after three more round trips for tile_atoms ->
// image / exclusion row start -> row: 30 of the kernel's 77 us were that prologue, and the walk waited for a load issued one
// chunk earlier at every step.)  The atom's own place in the list falls out of the walk (the candidate whose index is the
// atom's): the force kernel reads the atom's data from its image there.
// A difference of fixed-point coordinates IS the minimum image (the box spans 2^32), so there is no reference point and no wrap.
// The running counts live in vector registers (v_bcnt / v_mbcnt) -- a CU has ONE scalar ALU for its 16 waves.
// The pair kernel tests r < cutoff itself; what must hold here is "within cutoff+skin now" (float, with margin).
#define ATOM_LIST_PF 8      // chunks of 64 candidates in flight per wave (a load from the image the previous kernel wrote takes ~2,000 cycles, a chunk's tests ~300)
#define SX_ROW 32           // ints per i-slot in sx_row: [0] the atom's sorted index (-1: empty slot), [1 .. n-1] its excluded partners, [28] n, [29] min, [30] max of the row
#define SX_MAX 28
template <typename R>
__device__ __forceinline__ void build_atom_lists_body(const ListArgs& a, const NbConst<R>& c, const typename Img<R>::Atom* __restrict__ img, const int force, const int item) {
    using ufix = typename Img<R>::ufix;
    if constexpr (sizeof(ufix) != 4) return;   // (per-atom lists exist in mixed precision only)
    else {
    const int t = item / ATOM_LIST_PARTS, part = item - t * ATOM_LIST_PARTS;
    if (!force && a.flags->list_gen == a.flags->req_gen && !(a.batch_req && *a.batch_req)) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (t >= a.n_itiles) return;
    __builtin_amdgcn_s_setprio(3);   // (a rebuild's few blocks are the critical path to the nonbonded kernel)
    const int l = t / a.S;
    const uint4* gb = a.pimgb + (size_t)l * a.jcap;
    NB_STAMP(t == 0 && tid == 0, 5);
    // ---- first hop: the list's length, the wave's rows; second hop: the atoms' positions
    const int count = __builtin_amdgcn_readfirstlane(a.jcount[l]);
    int rowv[ATOM_LIST_U];
#pragma unroll
    for (int u = 0; u < ATOM_LIST_U; u++) rowv[u] = lane < SX_ROW ? a.sx_row[(size_t)(t * 64 + ATOM_SLOT(part, wv, u)) * SX_ROW + lane] : -1;
    const int nch = (count + 63) >> 6;
    auto fetch = [&](int ch) { return gb[min(ch * 64 + lane, max(count - 1, 0))]; };   // (past the end: the last entry again; masked by k < count)
    uint4 pf[ATOM_LIST_PF];
#pragma unroll
    for (int d = 0; d < ATOM_LIST_PF; d++) pf[d] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int d = 0; d < ATOM_LIST_PF; d++) if (d < nch) pf[d] = fetch(d);
    int ia4[ATOM_LIST_U], nrow4[ATOM_LIST_U], rmin4[ATOM_LIST_U], rmax4[ATOM_LIST_U]; uint32_t xi4[ATOM_LIST_U][3];
#pragma unroll
    for (int u = 0; u < ATOM_LIST_U; u++) {
        ia4[u] = __builtin_amdgcn_readlane(rowv[u], 0); nrow4[u] = __builtin_amdgcn_readlane(rowv[u], 28);
        rmin4[u] = __builtin_amdgcn_readlane(rowv[u], 29); rmax4[u] = __builtin_amdgcn_readlane(rowv[u], 30);
    }
#pragma unroll
    for (int u = 0; u < ATOM_LIST_U; u++) {
        const int iq = ia4[u] >= 0 ? ia4[u] : 0;
        xi4[u][0] = (uint32_t)img[iq].x; xi4[u][1] = (uint32_t)img[iq].y; xi4[u][2] = (uint32_t)img[iq].z;   // (same address in every lane: stays in vector registers)
    }
#pragma unroll
    for (int u = 0; u < ATOM_LIST_U; u++) if (lane == 0 && ia4[u] >= 0) a.aself[t * 64 + ATOM_SLOT(part, wv, u)] = (unsigned short)0xffffu;   // (until the walk meets the atom: same wave, program order)
    NB_STAMP(t == 0 && tid == 0, 6);
    float cfx = (float)c.dscale[0], cfy = (float)c.dscale[1], cfz = (float)c.dscale[2];
    float rl2 = (float)c.rlist2 * 1.0001f + 1e-5f, rl2m = (float)c.rlist2_m * 1.0001f + 1e-5f;
    const bool dual = a.plist != nullptr;
    float rp2 = c.rp2 * 1.0001f + 1e-5f, rp2m = c.rp2_m * 1.0001f + 1e-5f;   // (same safety margin as the full lists)
    asm volatile("" : "+v"(cfx), "+v"(cfy), "+v"(cfz), "+v"(rl2), "+v"(rl2m), "+v"(rp2), "+v"(rp2m));   // (vector-register operands: kernels_nb.h, "Round 4")
    int cntv[ATOM_LIST_U], cntp[ATOM_LIST_U];
    unsigned short* out4[ATOM_LIST_U]; unsigned short* outp4[ATOM_LIST_U];
#pragma unroll
    for (int u = 0; u < ATOM_LIST_U; u++) {
        cntv[u] = cntp[u] = 0;
        out4[u] = a.alist + ((size_t)t * 64 + ATOM_SLOT(part, wv, u)) * a.acap; outp4[u] = dual ? a.plist + ((size_t)t * 64 + ATOM_SLOT(part, wv, u)) * a.acap : nullptr;
    }
    const float INF = __builtin_inff();
    const int acap1 = a.acap - 1;
    NB_STAMP(t == 0 && tid == 0, 7);
    for (int ch0 = 0; ch0 < nch; ch0 += ATOM_LIST_PF) {
#pragma unroll
        for (int d = 0; d < ATOM_LIST_PF; d++) {
            const int ch = ch0 + d;
            if (ch >= nch) break;   // wave-uniform
            const uint4 cur = pf[d];
            if (ch + ATOM_LIST_PF < nch) pf[d] = fetch(ch + ATOM_LIST_PF);   // (in flight while the next chunks are tested)
            const int k = ch * 64 + lane;
            const bool mob = (int)cur.w < 0;
            const int js = (int)(cur.w & 0x7fffffffu);
            const unsigned short ent = NB_ENT(k);   // (the LDS index times 8: nonbonded_atom_body; "mobile" travels as the sign of the image's sigma/2)
            const float kinf = k < count ? 0.0f : INF;
            const float lim = mob ? rl2m : rl2, plim = mob ? rp2m : rp2;
            const int jfirst = __builtin_amdgcn_readfirstlane(js), jlast = __builtin_amdgcn_readlane(js, 63);   // (the list is ascending)
#pragma unroll
            for (int u = 0; u < ATOM_LIST_U; u++) {
                bool excluded = false;
                if (jlast >= rmin4[u] && jfirst <= rmax4[u]) {   // wave-uniform; a handful of chunks per atom
                    for (int e = 0; e < nrow4[u]; e++) excluded |= js == __builtin_amdgcn_readlane(rowv[u], e);
                    if (js == ia4[u] && k < count) a.aself[t * 64 + ATOM_SLOT(part, wv, u)] = ent;   // (the atom itself: its own place in the list)
                }
                const float dx = (float)(int32_t)(cur.x - xi4[u][0]) * cfx, dy = (float)(int32_t)(cur.y - xi4[u][1]) * cfy, dz = (float)(int32_t)(cur.z - xi4[u][2]) * cfz;
                float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx)) + kinf;
                d2 = excluded ? INF : d2;
                const bool pass = d2 < lim && ia4[u] >= 0;   // (an empty slot passes nobody)
                const unsigned long long bal = __ballot(pass);
                const unsigned blo = (unsigned)bal, bhi = (unsigned)(bal >> 32);
                const int pos = cntv[u] + (int)__builtin_amdgcn_mbcnt_hi(bhi, __builtin_amdgcn_mbcnt_lo(blo, 0u));
                // running count on the vector ALU.  gfx950 needs two wait states between a VALU write of an SGPR / VCC (the
                // compare behind the ballot) and a VALU read of it as an operand; the compiler inserts them for its own
                // instructions but not around inline assembly (without the s_nop the counts came out stale)
                int c2;
                asm("s_nop 1\n\tv_bcnt_u32_b32 %0, %1, %3\n\tv_bcnt_u32_b32 %0, %2, %0" : "=&v"(c2) : "s"(blo), "s"(bhi), "v"(cntv[u]));
                cntv[u] = c2;
                if (pass) out4[u][min(pos, acap1)] = ent;   // on overflow (flagged below) the surplus lands on the last entry
                if (dual) {   // (block-uniform) the pruned list: the same entries within cutoff + inner margin, same order
#if defined(NB_PRUNE_MOBILE_BY_BUILD_POSITION)
                    const bool keep = pass && d2 < plim;
#else
                    const bool keep = pass && (mob || d2 < plim);   // (mobile candidates of the full list always stay: nonbonded_atom_body)
#endif
                    const unsigned long long bk = __ballot(keep);
                    const unsigned klo = (unsigned)bk, khi = (unsigned)(bk >> 32);
                    const int ppos = cntp[u] + (int)__builtin_amdgcn_mbcnt_hi(khi, __builtin_amdgcn_mbcnt_lo(klo, 0u));
                    int c3;
                    asm("s_nop 1\n\tv_bcnt_u32_b32 %0, %1, %3\n\tv_bcnt_u32_b32 %0, %2, %0" : "=&v"(c3) : "s"(klo), "s"(khi), "v"(cntp[u]));
                    cntp[u] = c3;
                    if (keep) outp4[u][min(ppos, acap1)] = ent;
                }
            }
        }
    }
    NB_STAMP(t == 0 && tid == 0, 8);
#pragma unroll
    for (int u = 0; u < ATOM_LIST_U; u++) {
        // whole chunks of 64: the force kernel has no "lane holds an entry" mask, the tail of a list points at the group's ghost record
        { const int c0 = min(cntv[u], a.acap), padto = min((c0 + 63) & ~63, a.acap); if (c0 + lane < padto) out4[u][c0 + lane] = NB_ENT(count); }
        if (dual) { const int c0 = min(cntp[u], a.acap), padto = min((c0 + 63) & ~63, a.acap); if (c0 + lane < padto) outp4[u][c0 + lane] = NB_ENT(count); }
        if (lane == 0) {
            if (dual) {   // the pruned list is current as of these positions
                const int sl = t * 64 + ATOM_SLOT(part, wv, u);
                a.pcount[sl] = min(cntp[u], a.acap); a.pneed[sl] = 0;
                a.xprune[0][sl] = xi4[u][0]; a.xprune[1][sl] = xi4[u][1]; a.xprune[2][sl] = xi4[u][2];
            }
            a.acount[t * 64 + ATOM_SLOT(part, wv, u)] = min(cntv[u], a.acap);
            NB_STAMP(t == 0 && tid == 0, 10);
            if (cntv[u] > a.acap) a.flags->list_overflow = 1;
            else if (cntv[u] > a.acap - a.acap / 8) a.flags->resort_hint = 1;   // (a re-sort re-derives the capacities)
        }
    }
    }
}

template <typename R>
__global__ void __launch_bounds__(ATOM_LIST_THREADS) k_build_atom_lists(ListArgs a, NbConst<R> c, const typename Img<R>::Atom* __restrict__ img, int force) {
    build_atom_lists_body<R>(a, c, img, force, blockIdx.x);
}

template <typename R> struct NbArgs {
    int n_itiles, jcap, n_islots;
    int seg_len;     // CH: j-atoms per wave segment (8..64, divides 64)
    int waves_tile;  // NW: waves that share one i-tile (multiple of WPB)
    int npart;       // NW / WPB partial slabs per tile
    const int* tile_atoms;
    const int* jlist;
    const int* jcount;
    const int* batch_slot;
    const unsigned long long* mask_pool;
    double* fpart;   // [npart][3][n_islots]
    double* epart;   // [n_itiles*npart][2] (ENERGY only)
    DevFlags* flags;
    int* batch_req;  // see ListArgs; cleared here (every list block of this pass has read it by now)
    const unsigned short* alist; const int* acount; int acap;   // per-atom lists (ListArgs), nonbonded_atom_body only
    int S, n_lists;                                                // tiles per j-list, number of lists (ListArgs)
    // pruned per-atom lists (null: not in use): the entries of alist within cutoff + inner margin when the atom's list was last
    // pruned; pneed[islot] != 0: stale (ListArgs); xprune[k][islot]: where the atom was then (fixed point, as in the image)
    unsigned short* plist; int* pcount; int* pneed; unsigned* xprune[3];
    const uint4* pimg4; const float2* pimg2; const int* mlist; const int* mcount; int mcap;   // packed group images (ListArgs)
    const unsigned short* aself;   // (ListArgs)
};

// One block = WPB waves working on the SAME i-tile; wave w of the tile walks the j-list segments
// q = w, w+NW, w+2NW, ...; the block's waves are summed through LDS into one partial slab.
template <typename R, bool ENERGY, int WPB>
__device__ __forceinline__ void nonbonded_body(const NbArgs<R>& a, const NbConst<R>& c, const typename Img<R>::Atom* __restrict__ img, const int block_id) {
    using Atom = typename Img<R>::Atom;
    using sfix = typename Img<R>::sfix;
    __shared__ Atom lds[WPB][64];
    __shared__ double red[WPB][ENERGY ? 5 : 3][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (block_id == 0 && threadIdx.x == 0) { a.flags->list_gen = a.flags->req_gen; if (a.batch_req) *a.batch_req = 0; }  // lists are current for this pass
    const int blocks_tile = a.waves_tile / WPB;
    const int t = block_id / blocks_tile, part = block_id - t * blocks_tile;
    if (t >= a.n_itiles) return;  // block-uniform
    const int w = part * WPB + wv;  // wave index within the tile

    const int ia = a.tile_atoms[t * 64 + lane];
    const bool valid = ia >= 0;
    const Atom ai = img[valid ? ia : a.tile_atoms[t * 64]];
    const int count = a.jcount[t];
    const int CH = a.seg_len;
    const int nseg = (count + CH - 1) / CH;
    const int nbmax = a.jcap >> 6;
    const int* jl = a.jlist + (size_t)t * a.jcap;
    double fx = 0.0, fy = 0.0, fz = 0.0, elj = 0.0, ecl = 0.0;
    Atom* my = lds[wv];

    for (int q = w; q < nseg; q += a.waves_tile) {
        const int p0 = q * CH;
        const int nvalid = min(CH, count - p0);
        if (lane < nvalid) my[lane] = img[jl[p0 + lane]];
        const int slot = a.batch_slot[t * nbmax + (p0 >> 6)];
        unsigned long long m = 0ull;
        if (slot >= 0) m = a.mask_pool[(size_t)slot * 64 + lane] >> (p0 & 63);
        if (!valid) m = ~0ull;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS stores have landed (the slab is wave-private)
        for (int k = 0; k < nvalid; k++) {
            const Atom bj = my[k];
            R dx = (R)(sfix)(ai.x - bj.x) * c.scale[0];
            R dy = (R)(sfix)(ai.y - bj.y) * c.scale[1];
            R dz = (R)(sfix)(ai.z - bj.z) * c.scale[2];
            R r2 = dx * dx + dy * dy + dz * dz;
            bool in = (r2 < c.rc2) && !((m >> k) & 1ull);
            if (in) {
                R e1, e2;
                R fs = pair_regular<R>(r2, ai.q * bj.q, ai.hs + bj.hs, ai.se * bj.se, c, &e1, &e2);
                fx += (double)(fs * dx); fy += (double)(fs * dy); fz += (double)(fs * dz);
                if (ENERGY) {
                    double wgt = (bj.flags & FLAG_MOBILE) ? 0.5 : 1.0;
                    elj += wgt * (double)e1; ecl += wgt * (double)e2;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // all reads done before the slab is overwritten
    }
    red[wv][0][lane] = fx; red[wv][1][lane] = fy; red[wv][2][lane] = fz;
    if (ENERGY) { red[wv][3][lane] = elj; red[wv][4][lane] = ecl; }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int u = 1; u < WPB; u++) { fx += red[u][0][lane]; fy += red[u][1][lane]; fz += red[u][2][lane]; if (ENERGY) { elj += red[u][3][lane]; ecl += red[u][4][lane]; } }
        const int slot_i = t * 64 + lane;
        double* fp = a.fpart + (size_t)part * 3 * a.n_islots;
        fp[slot_i] = fx; fp[a.n_islots + slot_i] = fy; fp[2 * a.n_islots + slot_i] = fz;
        if (ENERGY) {
            elj = wave_sum(elj); ecl = wave_sum(ecl);
            if (lane == 0) { a.epart[2 * (t * a.npart + part)] = elj; a.epart[2 * (t * a.npart + part) + 1] = ecl; }
        }
    }
}

template <typename R, bool ENERGY, int WPB>
__global__ void __launch_bounds__(WPB * 64) k_nonbonded(NbArgs<R> a, NbConst<R> c, const typename Img<R>::Atom* __restrict__ img) {
    nonbonded_body<R, ENERGY, WPB>(a, c, img, blockIdx.x);
}

// ---- throughput variant for large i-sets (mixed precision): a wave owns a SUB-tile of IW i-atoms and JL = 64/IW
// j-lanes, lane = (jl, i).  With 64 i-atoms per wave nearly every j has SOME lane in range, so the pair body (70+
// VALU ops) runs for all 64 lanes although only ~23 % of the pairs are inside the cutoff; with 8 i-atoms x 8
// consecutive (Hilbert-sorted) j-atoms many wave-iterations have no pair in range and are skipped by the exec-mask
// branch.  j-batches are staged in a wave-private, bank-conflict-free LDS image ({x,y,z,q} 16 B + {hs,se} 8 B).
template <bool ENERGY, int IW>
__device__ __forceinline__ void nonbonded_sub_body(const NbArgs<float>& a, const NbConst<float>& c, const AtomF* __restrict__ img, const int block_id) {
    constexpr int JL = 64 / IW, SUBS = 64 / IW;
    struct P4 { uint32_t x, y, z; float q; };
    struct P2 { float hs, se; };
    __shared__ P4 lp[4][64];
    __shared__ P2 lq[4][64];
    __shared__ uint32_t lf[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (block_id == 0 && threadIdx.x == 0) { a.flags->list_gen = a.flags->req_gen; if (a.batch_req) *a.batch_req = 0; }
    const int NC = a.waves_tile;
    const int W = block_id * 4 + wv;
    const int t = W / (SUBS * NC);
    if (t >= a.n_itiles) return;  // wave-uniform
    const int rem = W - t * SUBS * NC, sub = rem / NC, ch = rem - sub * NC;
    const int il = lane & (IW - 1), jl = lane / IW;
    const int islot = t * 64 + sub * IW + il;
    const int ia = a.tile_atoms[islot];
    const bool valid = ia >= 0;
    const AtomF ai = img[valid ? ia : a.tile_atoms[t * 64]];
    const int count = a.jcount[t];
    const int nb = (count + 63) >> 6;
    const int nbmax = a.jcap >> 6;
    const int* jlst = a.jlist + (size_t)t * a.jcap;
    double fx = 0.0, fy = 0.0, fz = 0.0, elj = 0.0, ecl = 0.0;
    P4* mp = lp[wv]; P2* mq = lq[wv]; uint32_t* mf = lf[wv];

    for (int b = ch; b < nb; b += NC) {
        const int p0 = b * 64;
        const int nvalid = min(64, count - p0);
        if (lane < nvalid) {
            const AtomF aj = img[jlst[p0 + lane]];
            P4 v4; v4.x = aj.x; v4.y = aj.y; v4.z = aj.z; v4.q = aj.q; mp[lane] = v4;
            P2 v2; v2.hs = aj.hs; v2.se = aj.se; mq[lane] = v2;
            if (ENERGY) mf[lane] = aj.flags;
        }
        const int slot = a.batch_slot[t * nbmax + b];
        unsigned long long m = 0ull;
        if (slot >= 0) m = a.mask_pool[(size_t)slot * 64 + sub * IW + il];
        if (!valid) m = ~0ull;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int ngroups = (nvalid + JL - 1) / JL;
        float bx = 0.0f, by = 0.0f, bz = 0.0f;  // per-batch partial force (at most IW terms per lane), folded into fp64 below
        for (int g = 0; g < ngroups; g++) {
            const int k = g * JL + jl;
            const P4 bj = mp[k];
            float dx = (float)(int32_t)(ai.x - bj.x) * c.scale[0];
            float dy = (float)(int32_t)(ai.y - bj.y) * c.scale[1];
            float dz = (float)(int32_t)(ai.z - bj.z) * c.scale[2];
            float r2 = dx * dx + dy * dy + dz * dz;
            const bool in = (k < nvalid) && (r2 < c.rc2) && !((m >> k) & 1ull);
            if (in) {
                const P2 bp = mq[k];
                float e1, e2;
                float fs = pair_regular<float>(r2, ai.q * bj.q, ai.hs + bp.hs, ai.se * bp.se, c, &e1, &e2);
                bx = fmaf(fs, dx, bx); by = fmaf(fs, dy, by); bz = fmaf(fs, dz, bz);
                if (ENERGY) {
                    const double wgt = (mf[k] & FLAG_MOBILE) ? 0.5 : 1.0;
                    elj += wgt * (double)e1; ecl += wgt * (double)e2;
                }
            }
        }
        fx += (double)bx; fy += (double)by; fz += (double)bz;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // sum over the j-lanes of each i-atom
#pragma unroll
    for (int off = IW; off < 64; off <<= 1) { fx += __shfl_xor(fx, off, 64); fy += __shfl_xor(fy, off, 64); fz += __shfl_xor(fz, off, 64); }
    if (jl == 0) {
        double* fp = a.fpart + (size_t)ch * 3 * a.n_islots;
        fp[islot] = fx; fp[a.n_islots + islot] = fy; fp[2 * a.n_islots + islot] = fz;
    }
    if (ENERGY) {
        elj = wave_sum(elj); ecl = wave_sum(ecl);
        if (lane == 0) { a.epart[2 * W] = elj; a.epart[2 * W + 1] = ecl; }
    }
}

template <bool ENERGY, int IW>
__global__ void __launch_bounds__(256) k_nonbonded_sub(NbArgs<float> a, NbConst<float> c, const AtomF* __restrict__ img) {
    nonbonded_sub_body<ENERGY, IW>(a, c, img, blockIdx.x);
}

// ---- per-atom Verlet lists + LDS-resident tile image (mixed precision; the kernel of the replica-batched benchmark path).
// The sub-tile kernel above tests every i-atom of an 8-atom group against every atom near the 64-atom TILE: at the
// benchmark's density one tested pair in nine is inside the cutoff, and 46 % of its instructions are range checks.  Here
// the tile's j-list is staged into LDS ONCE per workgroup ({x,y,z,q} 16 B + {sigma/2, 2 sqrt(eps)} 8 B per atom), and a wave
// walks the exact neighbour list of ONE i-atom at a time (list built with the tile lists, skin as margin), lane = one
// neighbour, gathered from LDS by its 15-bit local index: 7 tested pairs in 10 are inside the cutoff, exclusions were
// removed when the list was built, and the i-atom lives in scalar registers.  Per-lane fp32 partial sums (at most NB_ATOM_U
// pair terms) are folded into fp64 once per atom; the 64 lanes are then summed in fp64 in a fixed order (bitwise reproducible;
// no atomics).  (Folding every 4 terms, dropping the index select and pairing the chunks were each within noise of 94 us at
// R = 256: with four waves per SIMD the kernel is bound by VALU issue of the pair body itself, ~62 instructions of which three
// are transcendental.)
#define NB_ATOM_U 12   // list entries per lane requested together (768 neighbours per round)
#define NB_ATOM_UA 7   // ... of which this many unconditionally (448 neighbours: the typical pruned list)
#define NB_ATOM_G 4    // chunks per straight-line group of the pair loop (divides NB_ATOM_U)
// Round 4: what a gfx950 SIMD charges for an instruction (scripts/valu_mix.hip, profiles/r04/valu_mix.txt; cycles per wave64
// instruction at four waves per SIMD): v_fma/v_mul/v_add/v_sub_f32, v_add/v_sub_u32, v_and, v_mov with VECTOR-register operands,
// literals or inline constants 2.5-2.8; the same instructions with ANY scalar-register operand 4.2-4.4; v_cvt_f32_i32, every
// shift (v_lshlrev, v_lshl_add, v_bfe), v_max_f32, v_cmp, v_cndmask (e64), DPP adds 4.2-4.4; v_rsq_f32 8.3.  The round-3 pair
// body was 50 instructions of which 23 were of the second kind (the i-atom and every constant lived in scalar registers, the
// fixed-point minimum image cost three conversions, the LDS addresses two shift-adds, the masks two compares and a select):
// ~180 cycles per 64 pairs.  This one is 43 instructions, all but the v_rsq and three conversions of the first kind: ~125 cycles.
//   * the i-atom and every constant (box scale, Ewald polynomial, cutoff) sit in VECTOR registers (laundered through an empty
//     asm so that the compiler does not move them back to scalar registers).  The coordinates stay fixed point: three
//     v_cvt_f32_i32 at the slower rate are what the exact minimum image costs.  (fp32 coordinates relative to the group's
//     centre would be three v_sub_f32 and nothing else, but a group of 5 tiles plus its list margin is wider than half the
//     short edge of the benchmark box: one stored image per list entry is then not every pair's minimum image, and the
//     fixed-point difference, which wraps by itself, is);
//   * a list entry is the j-atom's LDS index times 8 (u16; capacity <= 8191): the byte offset of its {sigma/2, 2 sqrt(eps)}
//     record as it stands and, doubled with one v_add_u32, of its {x, y, z, q} record -- the two arrays start at compile-time
//     offsets of the dynamic LDS (NB_LQ_BYTES), which go into the instructions' offset fields;
//   * "j is mobile" (energy weight 1/2, build-position rule of the prune) is the SIGN of the stored sigma/2, read with |.|;
//   * r < cutoff is a factor clamp((rc^2 - r^2) 2^30) in {0, 1} from one v_fma with the clamp modifier instead of compare +
//     select, and there is no "lane holds an entry" mask at all: every list is padded to whole chunks of 64 with the index
//     of a GHOST record (no charge, no epsilon: zero force at any finite distance; the builder keeps it 0.3 nm or more from
//     every i-atom, behind the last real entry of the group's image).
// Dual lists.  The atoms' lists (alist) are built with the full Verlet margin and are rebuilt rarely; a third of their entries
// sit in the margin, and every one of them costs a full pair evaluation of all 64 lanes.  The kernel therefore walks PRUNED
// lists (plist: the entries within cutoff + a small inner margin) and re-derives them itself, in passing and PER ATOM: the
// integrator raises pneed[islot] when that atom has moved the inner margin since its list was last pruned; for such an atom
// the wave walks the full list -- same arithmetic -- and writes the survivors back in list order (ballot + prefix count).
// No extra kernel, no extra staging of the image.  A hydrogen asks every ~6 steps, a heavy atom every ~20.
// Validity (derive_margins, blues_engine.hip): a pruned list holds every FROZEN candidate within cutoff + m of the atom's
// position at its prune (the atom asks again once it has moved m) and every MOBILE candidate whose position at the chain's list
// BUILD (the packed image keeps it) is within cutoff + trig + m of that position: until the next rebuild the candidate stays
// within `trig` of its build position whatever prunes it goes through itself.  (Round 3's first rule measured mobile
// candidates from their current position and lost pairs at the cutoff: blues_audit_lists.)
template <bool ENERGY>
__device__ __forceinline__ void nonbonded_atom_body(const NbArgs<float>& a, const NbConst<float>& c, const AtomF* __restrict__ img, const int t) {   // t: list (group of S i-tiles)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4))); typedef float f32x4 __attribute__((ext_vector_type(4))); typedef float f32x2 __attribute__((ext_vector_type(2)));   // (plain vector types: loads through an address-space pointer stay loads)
    extern __shared__ __align__(16) unsigned char nb_smem[];   // (the kernel has NO static LDS: the dynamic area then starts at address 0 and a list entry IS an LDS address)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
    const bool stamp_wg = t == 0 && blockIdx.x == (gridDim.x * 5u) / 8u;   // (stamped builds: ONE workgroup of the launch's third quarter writes the phase stamps)
    if (t == 0 && tid == 0) { a.flags->list_gen = a.flags->req_gen; if (a.batch_req) *a.batch_req = 0; }  // lists are current for this pass
    if (t >= a.n_lists) return;
    // Every pointer below arrives inside an argument record read from memory, i.e. as a GENERIC pointer: left like that, each
    // access is a FLAT instruction (64-bit address pair per lane, and counted on LGKMCNT as well as VMCNT, which would tie the
    // LDS pipeline of the pair loop to the list prefetches).  They all point to global memory: say so.
#if defined(__HIP_DEVICE_COMPILE__)
#define G1(T, p) ((__attribute__((address_space(1))) T*)(p))
#define L3(T, p) ((__attribute__((address_space(3))) T*)(p))
#else
#define G1(T, p) ((T*)(p))   // (the host pass only parses this function)
#define L3(T, p) ((T*)(p))
#endif
    const auto lds = L3(unsigned char, nb_smem);
    if (tid == 0 && (unsigned)(size_t)lds != 0u) a.flags->nan_flag = 1;   // (folds away at compile time as long as nobody gives this kernel a __shared__ variable)
    const auto g_alist = G1(const unsigned short, a.alist); const auto g_plist = G1(unsigned short, a.plist);
    const auto g_acount = G1(const int, a.acount); const auto g_pcount = G1(int, a.pcount); const auto g_pneed = G1(int, a.pneed);
    const auto g_tile_atoms = G1(const int, a.tile_atoms); const auto g_img = G1(const AtomF, img);
    const auto g_fpart = G1(double, a.fpart); const auto g_flags = G1(DevFlags, a.flags);
    const auto g_xprune0 = G1(unsigned, a.xprune[0]); const auto g_xprune1 = G1(unsigned, a.xprune[1]); const auto g_xprune2 = G1(unsigned, a.xprune[2]);
    const auto g4 = G1(const u32x4, a.pimg4) + (size_t)t * a.jcap; const auto g2 = G1(const f32x2, a.pimg2) + (size_t)t * a.jcap;
    const bool dual = a.plist != nullptr;
    // (the pruned lists and their counts are addressed as offsets from the full ones: selecting between two POINTERS inside the
    // lambdas below makes the optimiser keep the whole argument record in scratch memory)
    const long plist_off = dual ? (long)(a.plist - a.alist) : 0L, pcount_off = dual ? (long)(a.pcount - a.acount) : 0L;   // (separate allocations: can be gigabytes apart)
    const int count = a.jcount[t];
    const unsigned ghost = (unsigned)NB_ENT(count);
    const int slot0 = t * a.S * 64, nslot = min(a.S, a.n_itiles - t * a.S) * 64;   // the i-slots this list serves
    // An atom's whole list is requested up front (NB_ATOM_U wave-loads of 64 entries, all in flight together) and ONE ATOM
    // AHEAD: the next atom's entries travel while the current atom is computed, the first atom's while the image is staged.
    // ... and so does the atom's own image record
    unsigned entn[NB_ATOM_U]; int cntn = -1; bool stalen = true;
#pragma unroll
    for (int u = 0; u < NB_ATOM_U; u++) entn[u] = 0u;
    // What a wave needs to know about an i-slot -- how long its list is, whether it is due for a prune, where the atom itself sits
    // in the LDS image (a mobile atom is an entry of its own group's list: its record there is the atom's own data) -- is put
    // into LDS ONCE per workgroup while the image is staged: one record of 8 bytes per slot.  Round 3 read it from global memory
    // at the top of every atom's turn: three dependent round trips (tile_atoms -> pneed / image -> count) in front of the list
    // requests, the whole turn of a wave long, and the kernel's duration did not depend on the pair arithmetic at all (an empty
    // pair body took 104 us against 130; lists a quarter as long took the same 130: profiles/r04/README.md).
    const auto s_slot = L3(int, lds + NB_LQ_BYTES + 16 * a.jcap + NB_LDS_TAIL);   // [nslot][2]: {count or -1, self entry | pneed << 16}
    int selfn = 0;
    auto request = [&](int s) {
        cntn = -1;
        if (s < nslot) {
            const int r0 = __builtin_amdgcn_readfirstlane(s_slot[2 * s]), r1 = __builtin_amdgcn_readfirstlane(s_slot[2 * s + 1]);
            cntn = r0; selfn = r1 & 0xffff; stalen = (r1 >> 16) != 0;
            if (cntn >= 0) {
                const auto lst = g_alist + ((long)(slot0 + s) * a.acap + (stalen ? 0L : plist_off));
                // the first NB_ATOM_UA chunks unconditionally (a pruned list is about that long), the others only for a longer list (one
                // wave-uniform branch; a branch per chunk costs a dozen register copies each).  Loads may run past the count inside
                // the atom's own row (acap >= 64 NB_ATOM_U, host): chunks past the padded count are never walked
#if defined(K1X_NOLIST)    // (floor experiment: no list traffic, made-up entries)
#pragma unroll
                for (int u = 0; u < NB_ATOM_U; u++) entn[u] = (unsigned)NB_ENT((lane * 7 + u * 64 + s) % max(count, 1));
#else
#pragma unroll
                for (int u = 0; u < NB_ATOM_UA; u++) entn[u] = (unsigned)lst[u * 64 + lane];
                if (cntn > 64 * NB_ATOM_UA) {
#pragma unroll
                    for (int u = NB_ATOM_UA; u < NB_ATOM_U; u++) entn[u] = (unsigned)lst[u * 64 + lane];
                }
#endif
            }
        }
    };
    // Atoms are handed to the waves on demand (an LDS counter): lists differ in length, an atom that re-derives its pruned
    // list costs half as much again, and a static deal left the workgroup waiting for its unluckiest wave.  An atom's force
    // is computed by one wave in a fixed order whoever takes it: the assignment does not touch the result.
    const auto s_next = L3(int, lds + NB_LQ_BYTES + 16 * a.jcap);        // (behind the images; the host adds NB_LDS_TAIL bytes)
    const auto s_e = L3(double, lds + NB_LQ_BYTES + 16 * a.jcap + 16);   // [16][2], energy kernel only
    if (tid == 0) *s_next = nw;   // (the first nw atoms are dealt statically; the barrier behind the image staging publishes this)
    auto grab = [&]() -> int { int v = 0; if (lane == 0) v = __hip_atomic_fetch_add(s_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); return __builtin_amdgcn_readfirstlane(v); };
    NB_STAMP(stamp_wg && tid == 0, 16);
    // Round 6: the prologue as THREE dependent round trips instead of seven.  A 5-atom group's workgroup took 37 us beside 62 us for
    // the 256-atom one (scripts/r06_sweep_groups.sh: the same launch with 4 + 1 tiles per chain): more than half of a workgroup's time was
    // this prologue -- count -> tile_atoms -> pneed -> acount -> image copy -> barrier -> mcount -> mlist -> live image -> barrier, each a
    // trip to a memory system that 255 other CUs are streaming lists from -- with ONE workgroup per CU, i.e. nothing to overlap it with.
    // Now every load whose address does not depend on another load is issued at once: (1) the list's length, the number of mobile
    // entries, and the five facts of every i-slot; (2) the packed image and the list of mobile entries; (3) their live positions.
    const int mc = G1(const int, a.mcount)[t]; const auto ml = G1(const int, a.mlist) + (size_t)t * a.mcap * 2;
    for (int sl = tid; sl < nslot; sl += blockDim.x) {   // the slot records (one pass: nslot <= 1024)
        const int ia = g_tile_atoms[slot0 + sl];
        const int need = dual ? g_pneed[slot0 + sl] : 1;   // (no pruned lists: every atom walks its full list)
        const int c_full = g_acount[(long)(slot0 + sl)], c_pruned = g_acount[(long)(slot0 + sl) + pcount_off];   // (both lengths: which one counts is known with `need`)
        const unsigned self = G1(const unsigned short, a.aself)[slot0 + sl];
        int r0 = -1, r1 = 0;
        if (ia >= 0) {
            r0 = need ? c_full : c_pruned;
            if (self == 0xffffu) g_flags->list_overflow = 1;   // (the atom is not in its own group's list: only if that list overflowed)
            r1 = (int)(self | ((unsigned)need << 16));
        }
        s_slot[2 * sl] = r0; s_slot[2 * sl + 1] = r1;
    }
    NB_STAMP(stamp_wg && tid == 0, 17);
    {
        // the packed image (frozen entries as the builder left them, the ghost behind them): a straight copy, every load of a
        // thread in flight before its first LDS store; then the few mobile entries from the live image
        constexpr int SU = 7;   // 7 x 1024 threads >= the largest list capacity whose image fits LDS
        const int nthr = blockDim.x;
        // (the mobile entries' places, requested with the image: at most one per thread in the first sweep -- mcap <= the workgroup)
        int mk0 = -1, mjs0 = 0;
        if (tid < mc) { mk0 = ml[2 * tid]; mjs0 = ml[2 * tid + 1]; }
        for (int base = 0; base <= count; base += SU * nthr) {
            u32x4 r4[SU]; f32x2 r2[SU];
#pragma unroll
            for (int u = 0; u < SU; u++) { const int k = base + u * nthr + tid; if (k <= count) { r4[u] = g4[k]; r2[u] = g2[k]; } }
#pragma unroll
            for (int u = 0; u < SU; u++) { const int k = base + u * nthr + tid; if (k <= count) { *L3(u32x4, lds + NB_LQ_BYTES + 16 * k) = r4[u]; *L3(f32x2, lds + 8 * k) = r2[u]; } }
        }
        unsigned mx = 0u, my = 0u, mz = 0u;
        if (mk0 >= 0) { mx = g_img[mjs0].x; my = g_img[mjs0].y; mz = g_img[mjs0].z; }   // (in flight across the barrier)
        __syncthreads();
        request(wv);   // (the first atom's list travels while the mobile entries are refreshed)
        if (mk0 >= 0) { const auto pj = L3(unsigned, lds + NB_LQ_BYTES + 16 * mk0); pj[0] = mx; pj[1] = my; pj[2] = mz; }
        for (int q = tid + blockDim.x; q < mc; q += blockDim.x) {   // (more mobile entries than threads: small workgroups only)
            const int k = ml[2 * q], js = ml[2 * q + 1];
            const auto pj = L3(unsigned, lds + NB_LQ_BYTES + 16 * k);
            pj[0] = g_img[js].x; pj[1] = g_img[js].y; pj[2] = g_img[js].z;
        }
    }
    NB_STAMP(stamp_wg && tid == 0, 18);
    __syncthreads();
    NB_STAMP(stamp_wg && tid == 0, 19);
    // constants of the pair body, in vector registers (see the header)
#define NB_VREG(x) asm volatile("" : "+v"(x))
    float kw[EWALD_POLY_DEG + 1];
#pragma unroll
    for (int k = 0; k <= EWALD_POLY_DEG; k++) { kw[k] = c.ew.c[k]; NB_VREG(kw[k]); }
    float k_wa = c.ew.wa, k_nbig = -1073741824.0f, k_bigrc2 = c.rc2 * 1073741824.0f, k_sx = c.scale[0], k_sy = c.scale[1], k_sz = c.scale[2];
    NB_VREG(k_wa); NB_VREG(k_nbig); NB_VREG(k_bigrc2); NB_VREG(k_sx); NB_VREG(k_sy); NB_VREG(k_sz);
    float k_rp2 = c.rp2, k_rp2m = c.rp2_m;   // (a select between two fields of the argument record becomes an indexed load from a scratch copy of it)
    NB_VREG(k_rp2); NB_VREG(k_rp2m);
    const bool exact = c.ew.exact != 0;   // (kernel argument: wave-uniform)
    double elj = 0.0, ecl = 0.0;
    int s_nxt = wv;
    while (s_nxt < nslot) {
        const int s = s_nxt;
        const int islot = slot0 + s;
        const int cnt = cntn, self = selfn;
        const bool stale = stalen;
        unsigned ent[NB_ATOM_U];
#pragma unroll
        for (int u = 0; u < NB_ATOM_U; u++) ent[u] = entn[u];
        s_nxt = grab();
        request(s_nxt);
        if (cnt < 0) continue;   // wave-uniform; empty slots are never read back (FinRec.atom < 0)
        // the atom itself: its record in the LDS image (same address in every lane: one broadcast read, no scalar round trip)
        const u32x4 me4 = *L3(const u32x4, (size_t)(2u * (unsigned)self + NB_LQ_BYTES)); const f32x2 me2 = *L3(const f32x2, (size_t)(unsigned)self);
        unsigned ix = me4.x, iy = me4.y, iz = me4.z; float iq = __uint_as_float(me4.w), ihs = __builtin_fabsf(me2.x), ise = me2.y;
        NB_VREG(ix); NB_VREG(iy); NB_VREG(iz); NB_VREG(iq); NB_VREG(ihs); NB_VREG(ise);
        const bool pruning = !ENERGY && dual && stale;   // an energy evaluation reads whichever list is valid and writes none
        const auto lst = g_alist + ((long)islot * a.acap + (stale ? 0L : plist_off));
        const auto pout = g_plist + (size_t)islot * a.acap;   // (written only when pruning)
        int pcnt = 0;   // survivors written so far (wave-uniform)
        double fx = 0.0, fy = 0.0, fz = 0.0;
        float bx = 0.0f, by = 0.0f, bz = 0.0f;
        auto fetch = [&](unsigned e, u32x4& q4, f32x2& q2) {
            unsigned e2;   // (the compiler turns e + e into a shift, which issues at half the rate of the add)
            asm("v_add_u32 %0, %1, %1" : "=v"(e2) : "v"(e));
#if defined(K1X_NOGATHER)  // (floor experiment: conflict-free, consecutive addresses)
            e2 = (unsigned)lane * 16u + (e2 & 0x3c00u); const unsigned e1 = (unsigned)lane * 8u + (e & 0x1e00u);
            q4 = *L3(const u32x4, (size_t)(e2 + NB_LQ_BYTES)); q2 = *L3(const f32x2, (size_t)e1);
#else
            q4 = *L3(const u32x4, (size_t)(e2 + NB_LQ_BYTES)); q2 = *L3(const f32x2, (size_t)e);   // (LDS addresses as numbers: the dynamic area starts at 0, checked below)
#endif
        };
        auto pair = [&](unsigned e, const u32x4& q4, const f32x2& q2, auto write_tag, auto exact_tag) {
            constexpr bool W = decltype(write_tag)::value, X = decltype(exact_tag)::value;
#if defined(K1X_NOCVT)     // (floor experiment: wrong numbers, no conversions)
            const float dx = __uint_as_float(ix - q4.x) * k_sx, dy = __uint_as_float(iy - q4.y) * k_sy, dz = __uint_as_float(iz - q4.z) * k_sz;
#else
            const float dx = (float)(int32_t)(ix - q4.x) * k_sx, dy = (float)(int32_t)(iy - q4.y) * k_sy, dz = (float)(int32_t)(iz - q4.z) * k_sz;
#endif
            const float jq = __uint_as_float(q4.w);
#if defined(K1X_EMPTY)     // (floor experiment: gathers and accumulation only)
            if (!ENERGY && !W) { bx = fmaf(jq, dx, bx); by = fmaf(q2.x, dy, by); bz = fmaf(q2.y, dz, bz); return; }
#endif
            const float r2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            if (!ENERGY) {
                const float inv_r = __builtin_amdgcn_rsqf(r2), inv_r2 = inv_r * inv_r;
                const float sig = ihs + __builtin_fabsf(q2.x);
                const float s2 = sig * sig * inv_r2, s6 = s2 * s2 * s2, tt = (ise * q2.y) * s6;
                const float flj = tt * fmaf(12.0f, s6, -6.0f);
                float g;
                if (X) {
                    const float ar = c.alpha * (r2 * inv_r), ex = __expf(-ar * ar);
                    g = fmaf(TWO_OVER_SQRT_PI_F * c.alpha, ex, erfc_scaled_f(ar) * ex * inv_r) * inv_r2;
                } else {
                    const float w = fmaf(r2, k_wa, -1.0f);
                    float T = kw[EWALD_POLY_DEG];
#if defined(K1X_NOPOLY)    // (floor experiment)
                    T = fmaf(T, w, kw[0]);
#else
#pragma unroll
                    for (int k = EWALD_POLY_DEG - 1; k >= 0; k--) T = fmaf(T, w, kw[k]);
#endif
                    g = fmaf(inv_r, inv_r2, -T);
                }
                float fs = fmaf(iq * jq, g, flj * inv_r2);
                float m;   // 1 inside the cutoff, 0 outside (the ghost and the margin entries): (rc^2 - r^2) 2^30 clamped to [0, 1]
                asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(m) : "v"(r2), "v"(k_nbig), "v"(k_bigrc2));
                fs *= m;
                bx = fmaf(fs, dx, bx); by = fmaf(fs, dy, by); bz = fmaf(fs, dz, bz);
            } else if (r2 < c.rc2) {
                float e1, e2;
                pair_regular<float>(r2, iq * jq, ihs + __builtin_fabsf(q2.x), ise * q2.y, c, &e1, &e2);
                const double wgt = __builtin_bit_cast(int, q2.x) < 0 ? 0.5 : 1.0;   // a mobile j meets this pair again from its own list
                elj += wgt * (double)e1; ecl += wgt * (double)e2;
            }
            if (W) {   // prune: keep what lies within cutoff + inner margin, in list order
                const bool mob = __builtin_bit_cast(int, q2.x) < 0;
                // a frozen candidate stays if it is within cutoff + inner margin of where this atom is now; a MOBILE candidate of
                // the full list always stays (NB_KEEP_MOBILE): measuring it from its position at the list build -- the only bound
                // that holds until this atom's next prune (derive_margins) -- needs that position, a dependent global load per
                // chunk in the middle of the walk, and an atom that prunes then takes five normal turns; the mobile region is
                // compact, the rule dropped one mobile candidate in twelve
#if defined(NB_PRUNE_MOBILE_BY_BUILD_POSITION)
                bool keep = r2 < k_rp2;
                if (mob) {
                    const u32x4 pb = g4[NB_IDX(e)];
                    const float ex = (float)(int32_t)(ix - pb.x) * c.scale[0], ey = (float)(int32_t)(iy - pb.y) * c.scale[1], ez = (float)(int32_t)(iz - pb.z) * c.scale[2];
                    keep = ex * ex + ey * ey + ez * ez < k_rp2m;
                }
#else
                bool keep = mob || r2 < k_rp2;
#endif
                keep = keep && e != ghost;   // (the ghost pads lists: it is never an entry in its own right)
                const unsigned long long bal = __ballot(keep);
                if (keep) pout[pcnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = (unsigned short)e;
                pcnt += __popcll(bal);
            }
        };
        // The list is walked in straight-line groups of NB_ATOM_G chunks: all gathers of a group are issued before its first pair
        // is computed (sched_group_barrier pins that order), so the LDS latency is paid once per group.
        // The fp32 partials of a lane hold at most NB_ATOM_U pair terms before they are folded.
        const int nch = __builtin_amdgcn_readfirstlane((cnt + 63) >> 6);   // wave-uniform (and known to be: scalar branches below)
        auto walk = [&](auto write_tag, auto exact_tag) {
#pragma unroll
            for (int u0 = 0; u0 < NB_ATOM_U; u0 += NB_ATOM_G) {
                if (u0 >= nch) break;   // wave-uniform (scalar branch)
                if (u0 + NB_ATOM_G <= nch) {
                    u32x4 q4[NB_ATOM_G]; f32x2 q2[NB_ATOM_G];
#pragma unroll
                    for (int g = 0; g < NB_ATOM_G; g++) fetch(ent[u0 + g], q4[g], q2[g]);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 * NB_ATOM_G, 0);   // the DS reads first
#pragma unroll
                    for (int g = 0; g < NB_ATOM_G; g++) pair(ent[u0 + g], q4[g], q2[g], write_tag, exact_tag);
                } else {
#pragma unroll
                    for (int g = 0; g < NB_ATOM_G - 1; g++) {
                        if (u0 + g >= nch) break;
                        u32x4 q4; f32x2 q2;
                        fetch(ent[u0 + g], q4, q2);
                        pair(ent[u0 + g], q4, q2, write_tag, exact_tag);
                    }
                }
            }
            for (int base = 64 * NB_ATOM_U; base < cnt; base += 64) {   // lists longer than the prefetch window (dense regions); padded like the others
                const unsigned e = (unsigned)lst[base + lane];
                u32x4 q4; f32x2 q2;
                fetch(e, q4, q2);
                pair(e, q4, q2, write_tag, exact_tag);
                fx += (double)bx; fy += (double)by; fz += (double)bz; bx = by = bz = 0.0f;
            }
        };
        if (!pruning) { if (exact) walk(std::false_type{}, std::true_type{}); else walk(std::false_type{}, std::false_type{}); }
        else {
            if (exact) walk(std::true_type{}, std::true_type{}); else walk(std::true_type{}, std::false_type{});
            const int padto = min((pcnt + 63) & ~63, a.acap);
            if (pcnt + lane < padto) pout[pcnt + lane] = (unsigned short)ghost;   // whole chunks: the walk has no "lane holds an entry" mask
            if (lane == 0) {
                g_pcount[islot] = pcnt;   // (a subset of a list that fits acap)
                g_xprune0[islot] = ix; g_xprune1[islot] = iy; g_xprune2[islot] = iz;
                if (dual && !(s_slot[2 * s + 1] >> 17)) g_pneed[islot] = 0;   // (the flag was 1: taken down -- this wave is its only reader in this launch, the integrator raises it in a later one; 2 = kept up by the timing harness)
                __hip_atomic_fetch_add(&g_flags->prunes, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        NB_STAMP(stamp_wg && tid == 0, 20 + min(2 * (s / nw), 6));
        // 64 lanes -> one force: fp32 on DPP, fixed order (bitwise reproducible).  A lane's partial is a few pair terms of at
        // most ~10^3 kJ/mol/nm; six levels of fp32 adds put ~10^-3 kJ/mol/nm on a force of that size, the rounding the
        // per-lane fp32 accumulation carries anyway (tolerance: 10^-5 of the largest force, ~2 10^-2).  Lists past the
        // prefetch window were folded into fp64 per chunk above.
        if (!ENERGY) {
#if defined(K1X_NOREDUCE)  // (floor experiment: no wave sums)
            const float sx = bx, sy = by, sz = bz;
#else
            const float sx = wave_sum_dpp_f32(bx), sy = wave_sum_dpp_f32(by), sz = wave_sum_dpp_f32(bz);
#endif
            if (cnt > 64 * NB_ATOM_U) { fx = wave_sum_dpp(fx); fy = wave_sum_dpp(fy); fz = wave_sum_dpp(fz); }   // (wave-uniform; otherwise they are zero)
            fx += (double)sx; fy += (double)sy; fz += (double)sz;
            if (lane == 0) { g_fpart[islot] = fx; g_fpart[a.n_islots + islot] = fy; g_fpart[2 * a.n_islots + islot] = fz; }
        }
        NB_STAMP(stamp_wg && tid == 0, 21 + min(2 * (s / nw), 6));
    }
#undef G1
#undef NB_VREG
    NB_STAMP(stamp_wg && lane == 0, 44 + min(wv, 15));   // (stamped builds: when each wave ran out of atoms)
    if (ENERGY) {
        elj = wave_sum(elj); ecl = wave_sum(ecl);
        if (lane == 0) { s_e[2 * wv] = elj; s_e[2 * wv + 1] = ecl; }
        __syncthreads();
        if (tid == 0) {
            double s0 = 0.0, s1 = 0.0;
            for (int w = 0; w < nw; w++) { s0 += s_e[2 * w]; s1 += s_e[2 * w + 1]; }
            a.epart[2 * t] = s0; a.epart[2 * t + 1] = s1;
        }
    }
#undef L3
}

// Audit of the per-atom lists (diagnostic; blues_audit_lists).  One block per i-slot: every atom of the SYSTEM within the cutoff
// of the slot's atom at the current positions -- not alchemical, not an excluded partner -- must be an entry of the list the
// nonbonded kernel would walk for it now (the pruned list; the full list if the atom is flagged for a prune).  out[0] counts
// the pairs within the cutoff, out[1] those that are in no list.  This checks the whole chain of margins and triggers
// (derive_margins) against the one thing they exist for.
__global__ void __launch_bounds__(256) k_audit_atom_lists(NbArgs<float> a, ListArgs L, NbConst<float> c, const AtomF* __restrict__ img, unsigned long long* out) {
    __shared__ unsigned s_bits[8192 / 32];   // one bit per group-list entry (jcap <= 8191)
    const int sl = blockIdx.x, tid = threadIdx.x;
    const int ia = a.tile_atoms[sl];
    if (ia < 0) return;
    const int l = (sl >> 6) / a.S;
    const int count = a.jcount[l];
    const int* jl = a.jlist + (size_t)l * a.jcap;
    const bool full = a.plist == nullptr || a.pneed[sl] != 0;
    const unsigned short* lst = (full ? a.alist : a.plist) + (size_t)sl * a.acap;
    const int cnt = full ? a.acount[sl] : a.pcount[sl];
    for (int w = tid; w < (a.jcap + 31) / 32; w += 256) s_bits[w] = 0u;
    __syncthreads();
    for (int e = tid; e < cnt; e += 256) { const int k = NB_IDX(lst[e]); if (k < count) atomicOr(&s_bits[k >> 5], 1u << (k & 31)); }
    __syncthreads();
    const uint32_t xi = img[ia].x, yi = img[ia].y, zi = img[ia].z;
    const int e0 = L.ex_start[ia], e1 = L.ex_start[ia + 1];
    unsigned long long found = 0, missing = 0;
    for (int js = tid; js < L.n; js += 256) {
        if (js == ia || (img[js].flags & FLAG_ALCH)) continue;
        const double dx = (double)(int32_t)(img[js].x - xi) * c.dscale[0], dy = (double)(int32_t)(img[js].y - yi) * c.dscale[1], dz = (double)(int32_t)(img[js].z - zi) * c.dscale[2];
        if (dx * dx + dy * dy + dz * dz >= (double)c.rc2) continue;
        bool excluded = false;
        for (int e = e0; e < e1; e++) excluded |= L.ex_idx[e] == js;
        if (excluded) continue;
        found++;
        int lo = 0, hi = count;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (jl[mid] < js) lo = mid + 1; else hi = mid; }
        if (!(lo < count && jl[lo] == js && ((s_bits[lo >> 5] >> (lo & 31)) & 1u))) missing++;
    }
    if (found) atomicAdd(&out[0], found);
    if (missing) atomicAdd(&out[1], missing);
}

template <bool ENERGY>
__global__ void __launch_bounds__(1024) k_nonbonded_atom(NbArgs<float> a, NbConst<float> c, const AtomF* __restrict__ img) {
    nonbonded_atom_body<ENERGY>(a, c, img, blockIdx.x);
}

// One-off: LJ + Coulomb energy among FROZEN environment atoms (constant while they and the box stay put).
// Block = one tile of 64 consecutive (Hilbert-sorted) atoms, lane = i-atom; its 4 waves walk the later tiles: each lane
// tests one j-atom of the tile against the i-tile's bounding box, and only tiles with an atom inside cutoff range are
// staged in LDS and paired 64 x 64 (about 1 tile in 7 at 23k atoms; the all-pairs loop this replaces took 5.6 ms).
#define FROZEN_TILE 64
// Bounding box of a tile's frozen atoms (centre in fixed point, padded half extents; any = 0: the tile has none): written by
// k_frozen_boxes for every tile, read by the energy kernel to leave out the tiles no atom of which can be in range (6 in 7 at 23k atoms)
// WITHOUT loading them -- every block used to stream the whole image past its tile: 134 MB of L2 reads per chain.
template <typename R> struct FrozenBox { typename Img<R>::ufix c[3]; float h[3]; int any; };
#define FROZEN_BOX_DOUBLES 6    // room per tile behind the kernel's partial sums (sizeof(FrozenBox<double>) = 40)
template <typename R>
__device__ __forceinline__ void frozen_tile_box(const int n, const NbConst<R>& c, const typename Img<R>::Atom* __restrict__ img, const int it, const int lane,
                                                typename Img<R>::ufix cf[3], float hf[3], bool& any, typename Img<R>::Atom& ai, bool& i_on) {
    using sfix = typename Img<R>::sfix;
    using ufix = typename Img<R>::ufix;
    const int i = it * FROZEN_TILE + lane;
    const unsigned skip = FLAG_ALCH | FLAG_MOBILE;
    ai = img[min(i, n - 1)];
    i_on = i < n && !(ai.flags & skip);
    // relative to the tile's first atom (every wave computes it for itself)
    const typename Img<R>::Atom a0 = img[it * FROZEN_TILE];
    const ufix ref[3] = {a0.x, a0.y, a0.z};
    const ufix pi[3] = {ai.x, ai.y, ai.z};
    any = __ballot(i_on) != 0ull;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double off = i_on ? (double)(sfix)(pi[k] - ref[k]) * c.dscale[k] : 0.0;
        double lo = off, hi = off;
        for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o, 64)); hi = fmax(hi, __shfl_xor(hi, o, 64)); }
        cf[k] = ref[k] + (ufix)(sfix)llrint(0.5 * (lo + hi) / c.dscale[k]);
        hf[k] = (float)(0.5 * (hi - lo) + 2.0 * c.dscale[k]) * 1.00001f + 1e-6f;
    }
}
template <typename R>
__device__ __forceinline__ void frozen_boxes_body(const int n, const NbConst<R>& c, const typename Img<R>::Atom* __restrict__ img, FrozenBox<R>* boxes, const int blk) {
    const int lane = threadIdx.x & 63, it = blk * 4 + (threadIdx.x >> 6);
    if (it >= (n + FROZEN_TILE - 1) / FROZEN_TILE) return;
    typename Img<R>::ufix cf[3]; float hf[3]; bool any, i_on; typename Img<R>::Atom ai;
    frozen_tile_box<R>(n, c, img, it, lane, cf, hf, any, ai, i_on);
    if (lane == 0) { FrozenBox<R> B; for (int k = 0; k < 3; k++) { B.c[k] = cf[k]; B.h[k] = hf[k]; } B.any = any ? 1 : 0; boxes[it] = B; }
}

template <typename R>
__device__ __forceinline__ void energy_frozen_body(const int n, const NbConst<R>& c, const typename Img<R>::Atom* __restrict__ img,
                                                   const int* __restrict__ ex_start, const int* __restrict__ ex_idx, double* epart, const int it) {
    using Atom = typename Img<R>::Atom;
    using sfix = typename Img<R>::sfix;
    using ufix = typename Img<R>::ufix;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nt = (n + FROZEN_TILE - 1) / FROZEN_TILE;
    const int i = it * FROZEN_TILE + lane;
    __shared__ Atom lds[4][64];
    __shared__ double s[2][4];
    const unsigned skip = FLAG_ALCH | FLAG_MOBILE;
    const FrozenBox<R>* boxes = reinterpret_cast<const FrozenBox<R>*>(epart + 2 * nt);   // (k_frozen_boxes ran before this kernel)
    Atom ai; bool i_on, any_i;
    ufix cf[3]; float hf[3]; float cs[3];
    frozen_tile_box<R>(n, c, img, it, lane, cf, hf, any_i, ai, i_on);
#pragma unroll
    for (int k = 0; k < 3; k++) cs[k] = (float)c.dscale[k];
    const float rc2f = (float)c.rc2 * 1.0001f + 1e-5f;
    const float rc2b = rc2f * 1.001f + 1e-4f;    // box against box: never tighter than the atom-against-box test below
    const int e0 = i < n ? ex_start[i] : 0, e1 = i < n ? ex_start[i + 1] : 0;
    double elj = 0.0, ecl = 0.0;
    Atom* my = lds[wv];
    // wave wv takes the tiles jt = it + wv, it + wv + 4, ... as before (the order of every lane's sum is unchanged): 64 of its tiles
    // at a time, a lane per tile, box against box
    for (int jb = it + wv; jb < nt; jb += 256) {
        const int jl = jb + 4 * lane;
        bool cand = false;
        if (jl < nt && any_i) {
            const FrozenBox<R> B = boxes[jl];
            float d2 = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; k++) { float d = fabsf((float)(sfix)(B.c[k] - cf[k]) * cs[k]) - hf[k] - B.h[k]; d = fmaxf(d, 0.0f); d2 = fmaf(d, d, d2); }
            cand = B.any && d2 < rc2b;
        }
        unsigned long long todo = __ballot(cand);
      while (todo) {
        const int jt = jb + 4 * (__ffsll((long long)todo) - 1);
        todo &= todo - 1ull;
        const int j = jt * FROZEN_TILE + lane;
        const Atom aj = img[min(j, n - 1)];
        const bool j_on = j < n && !(aj.flags & skip);
        const ufix pj[3] = {aj.x, aj.y, aj.z};
        float d2 = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; k++) { float d = fabsf((float)(sfix)(pj[k] - cf[k]) * cs[k]) - hf[k]; d = fmaxf(d, 0.0f); d2 = fmaf(d, d, d2); }
        const unsigned long long near = __ballot(j_on && d2 < rc2f);
        if (near == 0ull) continue;   // wave-uniform
        my[lane] = aj;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (i_on) {
            for (int k = 0; k < 64; k++) {
                if (!((near >> k) & 1ull)) continue;
                const int jj = jt * FROZEN_TILE + k;
                if (jj <= i) continue;             // each pair once (only matters in the tile's own pass)
                const Atom bj = my[k];
                R dx = (R)(sfix)(ai.x - bj.x) * c.scale[0];
                R dy = (R)(sfix)(ai.y - bj.y) * c.scale[1];
                R dz = (R)(sfix)(ai.z - bj.z) * c.scale[2];
                R r2 = dx * dx + dy * dy + dz * dz;
                if (r2 < c.rc2) {
                    bool ex = false;
                    for (int e = e0; e < e1; e++) ex |= (ex_idx[e] == jj);
                    if (!ex) {
                        R a1, a2;
                        pair_regular<R>(r2, ai.q * bj.q, ai.hs + bj.hs, ai.se * bj.se, c, &a1, &a2);
                        elj += (double)a1; ecl += (double)a2;
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    elj = wave_sum(elj); ecl = wave_sum(ecl);
    if (lane == 0) { s[0][wv] = elj; s[1][wv] = ecl; }
    __syncthreads();
    if (threadIdx.x == 0) {
        epart[2 * it] = s[0][0] + s[0][1] + s[0][2] + s[0][3];
        epart[2 * it + 1] = s[1][0] + s[1][1] + s[1][2] + s[1][3];
    }
}
// (two launches: the boxes of ALL tiles first -- they live behind the nt pairs of partial sums in epart)
template <typename R>
__global__ void __launch_bounds__(256) k_frozen_boxes(int n, NbConst<R> c, const typename Img<R>::Atom* __restrict__ img, double* epart) {
    frozen_boxes_body<R>(n, c, img, reinterpret_cast<FrozenBox<R>*>(epart + 2 * ((n + FROZEN_TILE - 1) / FROZEN_TILE)), blockIdx.x);
}
template <typename R>
__global__ void __launch_bounds__(256) k_energy_frozen(int n, NbConst<R> c, const typename Img<R>::Atom* __restrict__ img,
                                                       const int* __restrict__ ex_start, const int* __restrict__ ex_idx, double* epart) {
    energy_frozen_body<R>(n, c, img, ex_start, ex_idx, epart, blockIdx.x);
}

__device__ inline void to_fixed32(const double p[3], const Box3& box, unsigned u[3]);
__device__ inline void to_fixed(const double p[3], const Box3& box, unsigned long long u[3]);

// Refresh the fixed-point tile image from the fp64 master positions (all atoms).
template <typename R>
__global__ void k_pack_positions(int n, const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
                                 const int* __restrict__ sorted_of_orig, typename Img<R>::Atom* img, Box3 box) {
    using ufix = typename Img<R>::ufix;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double p[3] = {x[i], y[i], z[i]};
    ufix f[3];
    if (sizeof(ufix) == 4) { unsigned u[3]; to_fixed32(p, box, u); f[0] = (ufix)u[0]; f[1] = (ufix)u[1]; f[2] = (ufix)u[2]; }
    else { unsigned long long u[3]; to_fixed(p, box, u); f[0] = (ufix)u[0]; f[1] = (ufix)u[1]; f[2] = (ufix)u[2]; }
    const int s = sorted_of_orig[i];
    img[s].x = f[0]; img[s].y = f[1]; img[s].z = f[2];
}

// 32-bit variant with native conversions: floor(fr*2^32 + 0.5) mod 2^32 == (floor(fr*2^64) + 2^31) >> 32
__device__ inline void to_fixed32(const double p[3], const Box3& box, unsigned u[3]) {
    for (int k = 0; k < 3; k++) {
        double fr = p[k] * box.invL[k]; fr -= floor(fr);
        double t = fr * 4294967296.0 + 0.5;
        if (t >= 4294967296.0) t -= 4294967296.0;
        u[k] = (unsigned)t;
    }
}

__device__ inline void to_fixed(const double p[3], const Box3& box, unsigned long long u[3]) {
    for (int k = 0; k < 3; k++) {
        double fr = p[k] * box.invL[k]; fr -= floor(fr);
        if (fr >= 1.0) fr = 0.0;
        u[k] = (unsigned long long)(fr * 18446744073709551616.0);
    }
}
