// kernels_frag.h -- the direct-space nonbonded path of systems in which EVERY environment atom moves (k1_mode 3): the MD leg of a
// BLUES iteration (reference blues/simulation.py:1189-1213, the unfrozen MD System) and NCMC switches without freeze_radius
// (reference examples/example_water.py, blues/moves.py:846-1083).  Same arithmetic as kernels_nb.h (12-6 LJ + erfc Coulomb inside
// the cutoff, exclusions removed), different decomposition.
//
// Why not the per-atom lists of kernels_nb.h.  There a lane is one (i, j) pair and every pair costs a 24-byte gather from the LDS
// image: with nothing frozen the image of a group is ~5,300 atoms for 320 i-atoms, every pair is met twice, and the launch is bound
// by the LDS gathers (round 4: an EMPTY pair body still takes 65 % of the time).  Why not 64 x 64 tile pairs with the reaction in
// registers (the round-4 review's proposal): counted on the benchmark system (cutoff 1.0 nm, 94 atoms / nm^3, Hilbert order) a
// 64 x 64 tile pair has 6-8 % of its pairs inside the cutoff, 8 x 8 cluster pairs 21-25 %, against 57-70 % here -- the half list
// does not make up for that (scripts/census_clusters.py).
//
// FRAGMENTS.  The environment is cut once, on the host, into fragments of up to three atoms that stay within a bond or two of each
// other -- a rigid water is one fragment -- and the atoms are sorted so that a fragment's atoms are consecutive in the image.  A
// wave takes one i-fragment at a time, a lane one j-FRAGMENT of its neighbour list: one gather of three consecutive 32-byte image
// records (L1 / L2; no LDS image, no staging) feeds NINE pair evaluations, the i-atoms sit in vector registers, the force on the
// three i-atoms is nine wave sums per fragment.  Lists hold fragment pairs with at least one atom pair within the list radius
// (exact test: 57 % of the listed atom pairs are inside the cutoff at a margin of 0.12 nm, 70 % at 0.04), 4 bytes per nine pairs.
// Full lists (every pair from both sides), so no reaction, no scatter, no atomics: a fragment's force is one wave's sum in a
// fixed order -- bitwise reproducible, batch = solo.
//
// DUAL LISTS (as GROMACS prunes): the OUTER list (cutoff + skin, rebuilt when an atom has moved skin / 2 since the build) is
// what the builder makes from scratch -- rarely; the INNER list (cutoff + m) is what the force kernel walks, re-derived from the
// outer one for every fragment at once when some atom has moved m / 2 since the last prune.  Validity: a pair that is not in the
// outer list stays beyond the cutoff until the next rebuild (each partner moves less than skin / 2); a pair of the outer list
// that the prune drops was beyond cutoff + m then and stays beyond the cutoff until the next prune (each moves less than m / 2).
//
// Exclusions and fragments with fewer than three atoms travel as a 9-bit mask in the list entry (bit 3a + b: pair of i-atom a
// with j-atom b counts); the builder takes the masks of bonded fragment pairs from a static table (sp_*), every other entry has
// all nine bits (minus the columns of atoms the j-fragment does not have).  A chunk of 64 entries in which every mask is full
// takes the plain pair body; rows are padded to whole chunks with the ghost record (no charge, no epsilon) under a full mask.
#pragma once
#include "kernels_nb.h"

// One fragment = one 64-byte record, i.e. half a cache line, in image order: what a lane of the force kernel gathers for nine
// pairs.  p[b] = {x, y, z (fixed point), charge * sqrt(k_e)} of atom b (a missing atom: a copy of atom 0 without charge); the
// Lennard-Jones parameters travel as one byte per atom into a table of {sigma / 2, 2 sqrt(eps)} the workgroup keeps in LDS (a
// system has a few dozen atom types).  Positions are refreshed by the first kernel of every pass (frag_pre_body) from the image
// the integrator maintains; everything else is written when the atoms are sorted.
struct FragRec { uint4 p[3]; unsigned types; int sid, islot0, s0; };   // types: t0 | t1 << 8 | t2 << 16 | atoms << 24 | mobile atoms (bit per atom) << 26; static fragment id; i-slot of the first mobile atom; first atom in the sorted image
#define FR_CNT(t) ((int)(((t) >> 24) & 3u))
#define FR_MOB(t) (((t) >> 26) & 7u)
#define FR_TYPES_MAX 64
struct FragBox { unsigned c[3]; float h[3]; float pad[2]; };   // bounding box of a block of 64 consecutive fragments: centre (fixed point), half extents (nm)

#define FR_MASK_SHIFT 23
#define FR_IDX_MASK 0x7fffffu
#define FR_FULL 0x1ffu
#define FR_THREADS 256      // 4 waves: every kernel of this file
#define FR_SP_MAX 64        // bonded partner fragments of one fragment (one lane each in the builder)
#ifndef FR_MIN_WAVES
#define FR_MIN_WAVES 4          // waves per SIMD the force kernel is compiled for (registers: 512 / this)
#endif
#define FR_QCAP 1280        // candidates one fragment's first test may let through (the builder's LDS queue, per wave)
// DevFlags::list_overflow of the fragment-list kernels: which capacity (the other builders write 1)
#define FR_OVER_OUTER 2
#define FR_OVER_INNER 4
#define FR_OVER_QUEUE 8
#define FR_OVER_REACH 16

struct FragArgs {
    int F, nblk, n, ocap, icap, fpw, nwg, n_islots;
    int NI; const int* ifrag;   // the fragments that hold a mobile atom (image order): the i-side of every list; list rows, counts and workgroups go by their ordinal
    FragRec* fimg;              // [F] in sorted order
    const float2* ljtab; int ntypes;   // {sigma / 2, 2 sqrt(eps)} by type
    const int* sp_start;        // [F + 1] by STATIC fragment id: rows of sp_ent
    const int2* sp_ent;         // {partner's static id, mask}: fragment pairs with an excluded atom pair, and the fragment with itself
    FragBox* bb;
    uint4* fpos0;               // [F] every fragment's FIRST atom at the last rebuild, packed: what the builder's first phase streams (16 of a record's 64 bytes)
    unsigned* olist; int* ocount;   // outer lists [F][ocap] (entry: the j-fragment's place in fimg | mask << 23), lengths
    unsigned* ilist; int* icount;   // inner lists [F][icap]
    unsigned* xprune;           // [3][n] by sorted index: where the atoms were at the last prune (fixed point)
    float ro2, ri2, ptrig2;     // (cutoff + skin)^2, (cutoff + m)^2 (both with the float safety of the other builders), (m / 2)^2
    float ro2_f, ri2_f;         // the same for a FROZEN j-fragment: only one partner moves, half the margins do
    float half_min_edge;        // half the shortest box edge (nonbonded_frag_body<., REL>)
    float scale[3];             // box edge / 2^32
    DevFlags* flags; int* batch_req;
    double* fpart; double* epart;
    int count_builds;           // 1: no list block of k_build_lists counts the rebuilds (no alchemical tile)
    int hint_blocks;            // candidate blocks per fragment beyond which the builder asks for a re-sort (the blocks have spread)
};

#if defined(__HIP_DEVICE_COMPILE__)
#define FG1(T, p) ((__attribute__((address_space(1))) T*)(p))
#else
#define FG1(T, p) ((T*)(p))
#endif

__device__ __forceinline__ bool frag_stale(const FragArgs& fa, int force) {
    return force || fa.flags->list_gen != fa.flags->req_gen || (fa.batch_req && *fa.batch_req);
}

// ---- first kernel of a pass: one wave per 64 i-fragments (the fragments that hold a mobile atom).  Their positions from the
// integrator's image into their records; lists current: has one of their atoms moved half the inner margin since the last prune?
__device__ __forceinline__ void frag_pre_body(const FragArgs& fa, const AtomF* __restrict__ img, const int force, const int blk) {
    const int lane = threadIdx.x & 63;
    const int q = blk * 64 + lane;
    if (blk * 64 >= fa.NI) return;
    const bool rebuild = frag_stale(fa, force);
    const bool valid = q < fa.NI;
    FragRec* rec = fa.fimg + fa.ifrag[min(q, fa.NI - 1)];
    const unsigned types = rec->types; const int s0 = rec->s0;
    const int cnt = FR_CNT(types);
    unsigned p[3][3];
#pragma unroll
    for (int b = 0; b < 3; b++) { const AtomF A = img[s0 + min(b, cnt - 1)]; p[b][0] = A.x; p[b][1] = A.y; p[b][2] = A.z; }
    if (valid) {
#pragma unroll
        for (int b = 0; b < 3; b++) { unsigned* w = reinterpret_cast<unsigned*>(&rec->p[b]); w[0] = p[b][0]; w[1] = p[b][1]; w[2] = p[b][2]; }   // (the charge stays)
    }
    if (rebuild) return;   // (the rebuild's own first kernel, frag_boxes_body, sets the reference positions)
    bool moved = false;
#pragma unroll
    for (int b = 0; b < 3; b++) {
        float d2 = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; k++) { const float d = (float)(int32_t)(p[b][k] - fa.xprune[(size_t)k * fa.n + s0 + min(b, cnt - 1)]) * fa.scale[k]; d2 = fmaf(d, d, d2); }
        moved |= d2 > fa.ptrig2 && ((FR_MOB(types) >> min(b, cnt - 1)) & 1u);
    }
    if (__ballot(moved && valid) != 0ull && lane == 0) fa.flags->prune_req = 1;
}

// ---- first kernel of a rebuild: one wave per block of 64 consecutive fragments, ALL of them (frozen ones too: after a load of
// positions everything has moved).  Positions into the records, reference positions of the prune trigger, the block's bounding
// box and its widest fragment.
__device__ __forceinline__ void frag_boxes_body(const FragArgs& fa, const AtomF* __restrict__ img, const int force, const int blk) {
    const int lane = threadIdx.x & 63;
    if (blk >= fa.nblk) return;
    if (!frag_stale(fa, force)) return;
    const int f = blk * 64 + lane;
    const bool valid = f < fa.F;
    FragRec* rec = fa.fimg + min(f, fa.F - 1);
    const unsigned types = rec->types; const int s0 = rec->s0;
    const int cnt = FR_CNT(types);
    unsigned p[3][3];
#pragma unroll
    for (int b = 0; b < 3; b++) { const AtomF A = img[s0 + min(b, cnt - 1)]; p[b][0] = A.x; p[b][1] = A.y; p[b][2] = A.z; }
    if (valid) {
#pragma unroll
        for (int b = 0; b < 3; b++) { unsigned* w = reinterpret_cast<unsigned*>(&rec->p[b]); w[0] = p[b][0]; w[1] = p[b][1]; w[2] = p[b][2]; }
#pragma unroll
        for (int b = 0; b < 3; b++) if (b < cnt) for (int k = 0; k < 3; k++) fa.xprune[(size_t)k * fa.n + s0 + b] = p[b][k];
        fa.fpos0[f] = make_uint4(p[0][0], p[0][1], p[0][2], 0u);
    }
    // bounding box relative to the block's first atom (a fixed-point difference is the minimum image)
    FragBox B;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const unsigned ref = (unsigned)__builtin_amdgcn_readfirstlane((int)p[0][k]);
        float lo = 1e30f, hi = -1e30f;
#pragma unroll
        for (int b = 0; b < 3; b++) { const float o = (float)(int32_t)(p[b][k] - ref) * fa.scale[k]; if (valid) { lo = fminf(lo, o); hi = fmaxf(hi, o); } }
        for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
        B.c[k] = ref + (unsigned)(int32_t)lrintf(0.5f * (lo + hi) / fa.scale[k]);
        B.h[k] = 0.5f * (hi - lo) * 1.0001f + 1e-5f;
    }
    {   // the widest fragment of the block around its first atom (the builder's first, one-distance test is conservative by it)
        float reach = 0.0f;
#pragma unroll
        for (int b = 1; b < 3; b++) {
            const float dx = (float)(int32_t)(p[b][0] - p[0][0]) * fa.scale[0], dy = (float)(int32_t)(p[b][1] - p[0][1]) * fa.scale[1], dz = (float)(int32_t)(p[b][2] - p[0][2]) * fa.scale[2];
            if (valid) reach = fmaxf(reach, sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx))));
        }
        for (int o = 32; o > 0; o >>= 1) reach = fmaxf(reach, __shfl_xor(reach, o, 64));
        B.pad[0] = reach * 1.0001f + 1e-5f; B.pad[1] = 0.0f;
    }
    if (lane == 0) { fa.bb[blk] = B; if (blk == 0 && fa.count_builds) fa.flags->builds++; }
}

// ---- second kernel: one wave per i-fragment.  Lists stale: outer and inner list from scratch -- the blocks whose box reaches
// the fragment, their fragments 64 at a time, exact test (some atom pair within the list radius), ordered compaction, masks from
// the static table.  Lists current but a prune asked for: the inner list from the outer one at the current positions.
__device__ __forceinline__ float frag_d2(const unsigned a[3], const uint4& b, const float sc[3]) {
    const float dx = (float)(int32_t)(a[0] - b.x) * sc[0], dy = (float)(int32_t)(a[1] - b.y) * sc[1], dz = (float)(int32_t)(a[2] - b.z) * sc[2];
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}
__device__ __forceinline__ unsigned frag_colmask(int cnt) { return cnt >= 3 ? FR_FULL : (cnt == 2 ? 0x0DBu : 0x049u); }

__device__ __forceinline__ void frag_lists_body(const FragArgs& fa, const int force, const int item) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int fq = item * (FR_THREADS / 64) + wv;   // the i-fragment's ordinal: its list rows and counts
    if (fq >= fa.NI) return;
    const bool rebuild = frag_stale(fa, force);
    if (!rebuild && !fa.flags->prune_req) return;
    const int f = __builtin_amdgcn_readfirstlane(fa.ifrag[fq]);
    const FragRec me = fa.fimg[f];
    const int cnt_i = __builtin_amdgcn_readfirstlane(FR_CNT(me.types)), s0_i = __builtin_amdgcn_readfirstlane(me.s0), sid_i = __builtin_amdgcn_readfirstlane(me.sid);
    const unsigned mob_i = (unsigned)__builtin_amdgcn_readfirstlane((int)FR_MOB(me.types));   // rows of frozen atoms are nobody's business
    unsigned xi[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++) { xi[a][0] = me.p[a].x; xi[a][1] = me.p[a].y; xi[a][2] = me.p[a].z; }
    const float sc[3] = {fa.scale[0], fa.scale[1], fa.scale[2]};
    unsigned* irow = fa.ilist + (size_t)fq * fa.icap;
    unsigned* orow = fa.olist + (size_t)fq * fa.ocap;
    // rows are padded to whole chunks with the GHOST record (fimg[F]: no charge, no epsilon -- zero force and energy at any finite
    // distance -- at a point no atom sits on) under a full mask: a padded chunk still takes the plain pair body
    const unsigned pad_entry = (unsigned)fa.F | (FR_FULL << FR_MASK_SHIFT);
    int icnt = 0;
    auto push = [&](unsigned* row, int cap, int& count, bool pass, unsigned entry) {
        const unsigned long long bal = __ballot(pass);
        const int pos = count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        if (pass && pos < cap) row[pos] = entry;
        count += __popcll(bal);
    };
    // (list_overflow says which capacity: FR_OVER_*; a row that has reached 90 % of its capacity, or its last chunk, asks for a new layout -- the host gives the
    // lists more room then, blues_engine.hip: frag_grow_caps -- before anything is lost)
    auto finish = [&](unsigned* row, int cap, int* counts, int count, int code) {
        const int c0 = min(count, cap), padto = min((c0 + 63) & ~63, cap);
        if (c0 + lane < padto) row[c0 + lane] = pad_entry;
        if (lane == 0) {
            counts[fq] = c0;
            if (count > cap) atomicOr(&fa.flags->list_overflow, code);
            else if (cap <= fa.F && (10 * count > 9 * cap || count > cap - 64)) fa.flags->resort_hint = 1;   // (90 %, or the row's last chunk is in use; a row of F + 1 places cannot overflow)
        }
    };
    if (!rebuild) {
        // ---- prune: walk the outer list, keep what is within cutoff + inner margin now
        const int ocnt = fa.ocount[fq], nch = (ocnt + 63) >> 6;   // (rows are padded to whole chunks with the ghost entry: never kept)
        unsigned e_cur = 0u, e_nxt = 0u;
        if (nch > 0) e_cur = orow[lane];
        if (nch > 1) e_nxt = orow[64 + lane];
        uint4 pc[3], pn[3]; unsigned tc = 0u, tn = 0u;
        if (nch > 0) { const FragRec* rj = fa.fimg + (e_cur & FR_IDX_MASK); pc[0] = rj->p[0]; pc[1] = rj->p[1]; pc[2] = rj->p[2]; tc = rj->types; }
        for (int u = 0; u < nch; u++) {   // entries two chunks ahead, records one chunk ahead of the tests
            unsigned e_n2 = 0u;
            if (u + 2 < nch) e_n2 = orow[(u + 2) * 64 + lane];
            if (u + 1 < nch) { const FragRec* rj = fa.fimg + (e_nxt & FR_IDX_MASK); pn[0] = rj->p[0]; pn[1] = rj->p[1]; pn[2] = rj->p[2]; tn = rj->types; }
            const unsigned mask = e_cur >> FR_MASK_SHIFT;
            float best = 1e30f;
#pragma unroll
            for (int a = 0; a < 3; a++) if (a < cnt_i && ((mob_i >> a) & 1u)) {
#pragma unroll
                for (int b = 0; b < 3; b++) { const float d2 = frag_d2(xi[a], pc[b], sc); if ((mask >> (3 * a + b)) & 1u) best = fminf(best, d2); }
            }
            // (a pair the mask leaves out never counts, so it need not keep the entry either; the fragment's entry for itself has
            // its live pairs at bond distance; the padding lies beyond ocnt)
            push(irow, fa.icap, icnt, u * 64 + lane < ocnt && best < (FR_MOB(tc) ? fa.ri2 : fa.ri2_f), e_cur);
            e_cur = e_nxt; e_nxt = e_n2; pc[0] = pn[0]; pc[1] = pn[1]; pc[2] = pn[2]; tc = tn;
        }
        finish(irow, fa.icap, fa.icount, icnt, FR_OVER_INNER);
        if (fq == 0 && lane == 0) fa.flags->prunes++;
        if (lane < cnt_i) {
#pragma unroll
            for (int k = 0; k < 3; k++) fa.xprune[(size_t)k * fa.n + s0_i + lane] = lane == 0 ? xi[0][k] : (lane == 1 ? xi[1][k] : xi[2][k]);
        }
        return;
    }
    // ---- rebuild
    // the static row of this fragment: one bonded partner fragment per lane
    const int sp0 = fa.sp_start[sid_i], nsp = __builtin_amdgcn_readfirstlane(fa.sp_start[sid_i + 1] - sp0);
    int2 sp = make_int2(-1, 0);
    if (lane < nsp) sp = fa.sp_ent[sp0 + lane];
    // reach of the fragment around its first atom
    float ri = 0.0f;
#pragma unroll
    for (int a = 1; a < 3; a++) if (a < cnt_i) ri = fmaxf(ri, sqrtf(frag_d2(xi[a], me.p[0], sc)));
    const float reach = sqrtf(fa.ro2) + ri + 1e-4f, reach2 = reach * reach;
    // Two phases.  First: of the blocks whose box reaches the fragment, the fragments whose FIRST atom is within list radius + both
    // fragments' reach (one distance per candidate, 16 bytes of its record), queued in order in LDS: about a third of what the
    // boxes let through.  Second: the exact test (nine distances), masks and the ordered compaction, on full waves of the queue.
    __shared__ int s_queue[FR_THREADS / 64][FR_QCAP];
    int* queue = s_queue[wv];
    int ocnt = 0, nblocks_seen = 0, qn = 0;
    for (int b0 = 0; b0 < fa.nblk; b0 += 64) {
        const int bq = b0 + lane;
        bool near = false; float rblk = 0.0f;
        if (bq < fa.nblk) {
            const FragBox B = fa.bb[bq];
            float d2 = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; k++) { float d = fabsf((float)(int32_t)(B.c[k] - xi[0][k]) * sc[k]) - B.h[k]; d = fmaxf(d, 0.0f); d2 = fmaf(d, d, d2); }
            near = d2 < reach2; rblk = B.pad[0];
        }
        unsigned long long todo = __ballot(near);
        nblocks_seen += __popcll(todo);
        // the blocks that pass, four at a time: their fragments' first atoms requested together, then tested in order
        while (todo) {
            int blk[4]; uint4 P0[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                blk[q] = todo ? __ffsll((long long)todo) - 1 : -1;
                if (blk[q] >= 0) { todo &= todo - 1ull; P0[q] = fa.fpos0[min((b0 + blk[q]) * 64 + lane, fa.F - 1)]; }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (blk[q] < 0) break;   // wave-uniform
                const int jf = (b0 + blk[q]) * 64 + lane;
                const float rb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rblk), blk[q]));
                const float lim = reach + rb;
                const bool cand = jf < fa.F && frag_d2(xi[0], P0[q], sc) < lim * lim;
                const unsigned long long bal = __ballot(cand);
                const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                if (cand && pos < FR_QCAP) queue[pos] = jf;
                qn += __popcll(bal);
            }
        }
    }
    if (qn > FR_QCAP) { if (lane == 0) atomicOr(&fa.flags->list_overflow, FR_OVER_QUEUE); qn = FR_QCAP; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the queue is wave-private: this wave's LDS stores have landed)
    FragRec P;
    if (qn > 0) P = fa.fimg[queue[min(lane, qn - 1)]];
    for (int q0 = 0; q0 < qn; q0 += 64) {
        FragRec Pn;
        const int jf = queue[min(q0 + lane, qn - 1)];
        if (q0 + 64 < qn) Pn = fa.fimg[queue[min(q0 + 64 + lane, qn - 1)]];
        const int cnt_j = FR_CNT(P.types);
        unsigned mask = frag_colmask(cnt_j);
        bool special = false;
        if (nsp > 0) {   // wave-uniform
            for (int e = 0; e < nsp; e++) {
                const int sid_e = __builtin_amdgcn_readlane(sp.x, e), m_e = __builtin_amdgcn_readlane(sp.y, e);
                if (P.sid == sid_e) { mask = (unsigned)m_e; special = true; }
            }
        }
        if (jf == f && !special) mask = 0u;   // the fragment itself without a live internal pair
        float best = 1e30f;
#pragma unroll
        for (int a = 0; a < 3; a++) if (a < cnt_i && ((mob_i >> a) & 1u)) {
#pragma unroll
            for (int b = 0; b < 3; b++) { const float d2 = frag_d2(xi[a], P.p[b], sc); if (b < cnt_j) best = fminf(best, d2); }
        }
        const unsigned entry = (unsigned)jf | (mask << FR_MASK_SHIFT);
        const bool jmob = FR_MOB(P.types) != 0u;   // (a frozen j-fragment: only this fragment's atoms move, half the margins do)
        const bool in_o = q0 + lane < qn && mask != 0u && best < (jmob ? fa.ro2 : fa.ro2_f);
        push(orow, fa.ocap, ocnt, in_o, entry);
        push(irow, fa.icap, icnt, in_o && best < (jmob ? fa.ri2 : fa.ri2_f), entry);
        P = Pn;
    }
    finish(orow, fa.ocap, fa.ocount, ocnt, FR_OVER_OUTER);
    finish(irow, fa.icap, fa.icount, icnt, FR_OVER_INNER);
    if (lane == 0 && nblocks_seen > fa.hint_blocks) fa.flags->resort_hint = 1;
}

// ---- the force / energy kernel.  A wave owns fa.fpw consecutive fragments; per fragment it walks the inner list 64 entries at a
// time: entry -> the j-fragment's record (one 64-byte gather, requested one chunk ahead) -> nine pairs.
#define FR_VREG(x) asm volatile("" : "+v"(x))
// REL: a pair's separation from the j-atom's offset to the fragment's FIRST atom (one fixed-point minimum image per j-atom and
// chunk) plus the i-atom's offset inside its fragment -- valid where cutoff + the fragment's reach stays below half the shortest
// box edge (the host chooses; the kernel checks every fragment).  Otherwise the fixed-point difference per pair.
template <bool ENERGY, bool REL>
__device__ __forceinline__ void nonbonded_frag_body(const FragArgs& fa, const NbConst<float>& c, const int wg) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4))); typedef float f32x2 __attribute__((ext_vector_type(2)));
    __shared__ f32x2 s_lj[FR_TYPES_MAX];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (wg == 0 && tid == 0) { fa.flags->list_gen = fa.flags->req_gen; if (fa.batch_req) *fa.batch_req = 0; fa.flags->prune_req = 0; }   // lists are current for this pass
    if (tid < fa.ntypes) { const float2 t = fa.ljtab[tid]; f32x2 v; v.x = t.x; v.y = t.y; s_lj[tid] = v; }
    __syncthreads();
    const auto g_fimg = FG1(const unsigned char, fa.fimg);
    const auto g_ilist = FG1(const unsigned, fa.ilist); const auto g_icount = FG1(const int, fa.icount);
    const auto g_fpart = FG1(double, fa.fpart);
    // constants of the pair body in vector registers (kernels_nb.h, "Round 4": an instruction with a scalar operand issues at half rate)
    float kw[EWALD_POLY_DEG + 1];
#pragma unroll
    for (int k = 0; k <= EWALD_POLY_DEG; k++) { kw[k] = c.ew.c[k]; FR_VREG(kw[k]); }
    float k_wa = c.ew.wa, k_nbig = -1073741824.0f, k_bigrc2 = c.rc2 * 1073741824.0f, k_sx = c.scale[0], k_sy = c.scale[1], k_sz = c.scale[2], k_far = 4.0f * c.rc2;
    FR_VREG(k_wa); FR_VREG(k_nbig); FR_VREG(k_bigrc2); FR_VREG(k_sx); FR_VREG(k_sy); FR_VREG(k_sz); FR_VREG(k_far);
    const bool exact = c.ew.exact != 0;
    double elj = 0.0, ecl = 0.0;
    const int w = wg * (FR_THREADS / 64) + wv;
    const int q0 = w * fa.fpw, q1 = min(fa.NI, q0 + fa.fpw);
    const auto g_ifrag = FG1(const int, fa.ifrag);
    struct Rec { u32x4 p[3]; unsigned types; };
    auto load_rec = [&](unsigned e, Rec& R) {
        const auto base = g_fimg + (size_t)(e & FR_IDX_MASK) * sizeof(FragRec);
#pragma unroll
        for (int b = 0; b < 3; b++) R.p[b] = *FG1(const u32x4, base + 16 * b);
        R.types = *FG1(const unsigned, base + 48);
    };
    // the head of the NEXT fragment -- which fragment, how long its row, the row's first two chunks of entries -- is requested while
    // the current one computes: a fragment then starts with one round trip (its own record beside the first chunk's records)
    // instead of three in a row
    int f_pf = 0, cnt_pf = 0; unsigned e0_pf = 0u, e1_pf = 0u;
    const bool two_chunks = fa.icap > 64;
    auto prefetch_head = [&](int fqn) {
        f_pf = g_ifrag[fqn]; cnt_pf = g_icount[fqn];
        const auto rown = g_ilist + (size_t)fqn * fa.icap;
        e0_pf = rown[lane];
        if (two_chunks) e1_pf = rown[64 + lane];   // (inside the row whatever its length: used only if the row has a second chunk)
    };
    if (q0 < q1) prefetch_head(q0);
    for (int fq = q0; fq < q1; fq++) {
        const int f = __builtin_amdgcn_readfirstlane(f_pf);
        const int cnt = __builtin_amdgcn_readfirstlane(cnt_pf);
        const int nch = (cnt + 63) >> 6;
        const auto row = g_ilist + (size_t)fq * fa.icap;
        unsigned e_cur = nch > 0 ? e0_pf : 0u, e_nxt = nch > 1 ? e1_pf : 0u;
        if (fq + 1 < q1) prefetch_head(fq + 1);
        // the fragment's own atoms (same address in every lane: stays in vector registers)
        Rec me; load_rec((unsigned)f, me);
        const int islot0 = __builtin_amdgcn_readfirstlane(*FG1(const int, g_fimg + (size_t)f * sizeof(FragRec) + 56));
        const int cnt_i = __builtin_amdgcn_readfirstlane(FR_CNT(me.types));
        const unsigned mob_i = (unsigned)__builtin_amdgcn_readfirstlane((int)FR_MOB(me.types));
        unsigned ix[3], iy[3], iz[3]; float iq[3], ihs[3], ise[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const f32x2 l = s_lj[(me.types >> (8 * a)) & 0xffu];
            ix[a] = me.p[a].x; iy[a] = me.p[a].y; iz[a] = me.p[a].z; iq[a] = __uint_as_float(me.p[a].w); ihs[a] = l.x; ise[a] = l.y;
            FR_VREG(ix[a]); FR_VREG(iy[a]); FR_VREG(iz[a]); FR_VREG(iq[a]); FR_VREG(ihs[a]); FR_VREG(ise[a]);
        }
        // offsets of the fragment's other atoms from its first one (a bond or two: exact to 1e-8 nm in fp32).  A pair's separation
        // is formed as (i0 - j) + (i_a - i0): the first term from the fixed-point difference (the minimum image, once per j-atom and
        // chunk instead of once per pair), the second a small float -- the sum carries an absolute rounding of 6e-8 |i0 - j|, which for
        // a close pair is as small as the per-pair form's because the fragment is small
        float ox[3], oy[3], oz[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            ox[a] = (float)(int32_t)(ix[a] - ix[0]) * k_sx; oy[a] = (float)(int32_t)(iy[a] - iy[0]) * k_sy; oz[a] = (float)(int32_t)(iz[a] - iz[0]) * k_sz;
            FR_VREG(ox[a]); FR_VREG(oy[a]); FR_VREG(oz[a]);
        }
        if (REL) {
            float reach2 = 0.0f;
#pragma unroll
            for (int a = 1; a < 3; a++) reach2 = fmaxf(reach2, fmaf(oz[a], oz[a], fmaf(oy[a], oy[a], ox[a] * ox[a])));
            if (lane == 0 && sqrtf(reach2) + sqrtf(c.rc2) >= fa.half_min_edge) atomicOr(&fa.flags->list_overflow, FR_OVER_REACH);   // (cannot happen for bonded fragments in a box the host let through)
        }
        const bool lj_a[3] = {ise[0] != 0.0f, ise[1] != 0.0f, ise[2] != 0.0f};   // (wave-uniform: a hydrogen of a water has no epsilon, its pairs no 12-6 term)
        Rec cur, nxt;
        if (nch > 0) load_rec(e_cur, cur);
        float fx[3] = {0.0f, 0.0f, 0.0f}, fy[3] = {0.0f, 0.0f, 0.0f}, fz[3] = {0.0f, 0.0f, 0.0f};
        auto pair = [&](int a, const u32x4& jp, float rx, float ry, float rz, float jq, const f32x2& jl, float jw, auto masked_tag, auto exact_tag, auto lj_tag, unsigned bit) {
            constexpr bool M = decltype(masked_tag)::value, X = decltype(exact_tag)::value, LJ = decltype(lj_tag)::value;
            float dx, dy, dz;
            if (REL) { dx = a == 0 ? rx : rx + ox[a]; dy = a == 0 ? ry : ry + oy[a]; dz = a == 0 ? rz : rz + oz[a]; }
            else { dx = (float)(int32_t)(ix[a] - jp.x) * k_sx; dy = (float)(int32_t)(iy[a] - jp.y) * k_sy; dz = (float)(int32_t)(iz[a] - jp.z) * k_sz; }
            float r2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            if (M) r2 = bit ? r2 : k_far;   // a pair the mask leaves out (excluded, the atom itself, a padding entry): beyond the cutoff, finite
            if (!ENERGY) {
                const float inv_r = __builtin_amdgcn_rsqf(r2), inv_r2 = inv_r * inv_r;
                float g;
                if (X) {
                    const float ar = c.alpha * (r2 * inv_r), ex = __expf(-ar * ar);
                    g = fmaf(TWO_OVER_SQRT_PI_F * c.alpha, ex, erfc_scaled_f(ar) * ex * inv_r) * inv_r2;
                } else {
                    const float wq = fmaf(r2, k_wa, -1.0f);
                    float T = kw[EWALD_POLY_DEG];
#pragma unroll
                    for (int k = EWALD_POLY_DEG - 1; k >= 0; k--) T = fmaf(T, wq, kw[k]);
                    g = fmaf(inv_r, inv_r2, -T);
                }
                float fs = (iq[a] * jq) * g;
                if (LJ) {
                    const float sig = ihs[a] + jl.x;
                    const float s2 = sig * sig * inv_r2, s6 = s2 * s2 * s2, tt = (ise[a] * jl.y) * s6;
                    const float flj = tt * fmaf(12.0f, s6, -6.0f);
                    fs = fmaf(flj, inv_r2, fs);
                }
                float m;   // 1 inside the cutoff, 0 outside: (rc^2 - r^2) 2^30 clamped to [0, 1]
                asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(m) : "v"(r2), "v"(k_nbig), "v"(k_bigrc2));
                fs *= m;
                fx[a] = fmaf(fs, dx, fx[a]); fy[a] = fmaf(fs, dy, fy[a]); fz[a] = fmaf(fs, dz, fz[a]);
            } else if (r2 < c.rc2) {
                float e1, e2;
                pair_regular<float>(r2, iq[a] * jq, ihs[a] + jl.x, ise[a] * jl.y, c, &e1, &e2);
                elj += (double)(jw * e1); ecl += (double)(jw * e2);   // jw: 1/2 where the partner is mobile too (that pair is met from both sides), 1 for a frozen one
            }
        };
        auto chunk = [&](const Rec& R, unsigned e, auto masked_tag, auto exact_tag) {
            const unsigned mask = e >> FR_MASK_SHIFT;
            f32x2 jl[3]; float rx[3], ry[3], rz[3], jw[3];
#pragma unroll
            for (int b = 0; b < 3; b++) {
                jl[b] = s_lj[(R.types >> (8 * b)) & 0xffu];
                jw[b] = ENERGY ? (((R.types >> (26 + b)) & 1u) ? 0.5f : 1.0f) : 0.0f;
                if (REL) { rx[b] = (float)(int32_t)(ix[0] - R.p[b].x) * k_sx; ry[b] = (float)(int32_t)(iy[0] - R.p[b].y) * k_sy; rz[b] = (float)(int32_t)(iz[0] - R.p[b].z) * k_sz; }
                else rx[b] = ry[b] = rz[b] = 0.0f;
            }
            const bool ljb[3] = {ENERGY || __ballot(jl[0].y != 0.0f) != 0ull, ENERGY || __ballot(jl[1].y != 0.0f) != 0ull, ENERGY || __ballot(jl[2].y != 0.0f) != 0ull};
#pragma unroll
            for (int a = 0; a < 3; a++) {
                if (a >= cnt_i) break;   // wave-uniform
                if (!((mob_i >> a) & 1u)) continue;   // (a frozen atom of the fragment: nobody reads its force)
                if (lj_a[a]) {
#pragma unroll
                    for (int b = 0; b < 3; b++) {
                        // (wave-uniform: no lane's j-atom b has an epsilon -- the hydrogens of a chunk of waters -- and the pair has no 12-6 term)
                        if (ljb[b]) pair(a, R.p[b], rx[b], ry[b], rz[b], __uint_as_float(R.p[b].w), jl[b], jw[b], masked_tag, exact_tag, std::true_type{}, (mask >> (3 * a + b)) & 1u);
                        else pair(a, R.p[b], rx[b], ry[b], rz[b], __uint_as_float(R.p[b].w), jl[b], jw[b], masked_tag, exact_tag, std::false_type{}, (mask >> (3 * a + b)) & 1u);
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < 3; b++) pair(a, R.p[b], rx[b], ry[b], rz[b], __uint_as_float(R.p[b].w), jl[b], jw[b], masked_tag, exact_tag, std::false_type{}, (mask >> (3 * a + b)) & 1u);
                }
            }
        };
        for (int u = 0; u < nch; u++) {
            unsigned e_n2 = 0u;
            if (u + 2 < nch) e_n2 = row[(u + 2) * 64 + lane];
            if (u + 1 < nch) load_rec(e_nxt, nxt);
            const bool plain = __ballot((e_cur >> FR_MASK_SHIFT) != FR_FULL) == 0ull;
            if (plain) { if (exact) chunk(cur, e_cur, std::false_type{}, std::true_type{}); else chunk(cur, e_cur, std::false_type{}, std::false_type{}); }
            else { if (exact) chunk(cur, e_cur, std::true_type{}, std::true_type{}); else chunk(cur, e_cur, std::true_type{}, std::false_type{}); }
            cur = nxt; e_cur = e_nxt; e_nxt = e_n2;
        }
        if (!ENERGY) {
            // 64 lanes -> one force per i-atom and axis: fp32 on DPP, fixed order (kernels_nb.h: a lane's partial is at most
            // 3 * (icap / 64) pair terms)
#pragma unroll
            for (int a = 0; a < 3; a++) {
                if (a >= cnt_i) break;
                if (!((mob_i >> a) & 1u)) continue;
                const float sx = wave_sum_dpp_f32(fx[a]), sy = wave_sum_dpp_f32(fy[a]), sz = wave_sum_dpp_f32(fz[a]);
                const int isl = islot0 + __popc(mob_i & ((1u << a) - 1u));   // (i-slots go to the mobile atoms, in image order)
                if (lane == 0) { g_fpart[isl] = (double)sx; g_fpart[fa.n_islots + isl] = (double)sy; g_fpart[2 * fa.n_islots + isl] = (double)sz; }
            }
        }
    }
    if (ENERGY) {
        __shared__ double s_e[FR_THREADS / 64][2];
        elj = wave_sum(elj); ecl = wave_sum(ecl);
        if (lane == 0) { s_e[wv][0] = elj; s_e[wv][1] = ecl; }
        __syncthreads();
        if (tid == 0) {
            double a0 = 0.0, a1 = 0.0;
            for (int q = 0; q < FR_THREADS / 64; q++) { a0 += s_e[q][0]; a1 += s_e[q][1]; }
            fa.epart[2 * wg] = a0; fa.epart[2 * wg + 1] = a1;
        }
    }
}
#undef FR_VREG

// with_boxes: the rebuild's first kernel rides along (every environment atom mobile: the two kernels have the same grid, and a
// launch that finds nothing to do still costs its 4-5 us)
__global__ void __launch_bounds__(FR_THREADS) k_frag_pre(FragArgs fa, const AtomF* __restrict__ img, int force, int with_boxes) {
    const int blk = blockIdx.x * (FR_THREADS / 64) + (threadIdx.x >> 6);
    frag_pre_body(fa, img, force, blk);
    if (with_boxes) frag_boxes_body(fa, img, force, blk);
}
__global__ void __launch_bounds__(FR_THREADS) k_frag_boxes(FragArgs fa, const AtomF* __restrict__ img, int force) {
    frag_boxes_body(fa, img, force, blockIdx.x * (FR_THREADS / 64) + (threadIdx.x >> 6));
}
__global__ void __launch_bounds__(FR_THREADS) k_frag_lists(FragArgs fa, int force) {
    frag_lists_body(fa, force, blockIdx.x);
}
template <bool ENERGY, bool REL>
__global__ void __launch_bounds__(FR_THREADS) k_nonbonded_frag(FragArgs fa, NbConst<float> c) {
    nonbonded_frag_body<ENERGY, REL>(fa, c, blockIdx.x);
}

// Audit (diagnostic; blues_audit_lists): one block per i-fragment.  Every atom of the system within the cutoff of one of the
// fragment's atoms at the current positions -- not alchemical, not an excluded partner, not the atom itself -- must be a live
// pair of the list the force kernel walks (the inner list).  out[0]: pairs within the cutoff, out[1]: those no entry covers.
__global__ void __launch_bounds__(256) k_audit_frag_lists(FragArgs fa, ListArgs L, NbConst<float> c, const AtomF* __restrict__ img, unsigned long long* out) {
    extern __shared__ unsigned s_bits[];   // [3][words]: one bit per atom of the system and i-atom
    const int fq = blockIdx.x, tid = threadIdx.x;
    const int f = fa.ifrag[fq];
    const int words = (fa.n + 31) / 32;
    const FragRec m = fa.fimg[f];
    const int cnt_i = FR_CNT(m.types); const unsigned mob_i = FR_MOB(m.types);
    for (int w = tid; w < 3 * words; w += 256) s_bits[w] = 0u;
    __syncthreads();
    const int cnt = fa.icount[fq];
    for (int e = tid; e < cnt; e += 256) {
        const unsigned ent = fa.ilist[(size_t)fq * fa.icap + e];
        const FragRec& rj = fa.fimg[ent & FR_IDX_MASK];
        const int j0 = rj.s0, cnt_j = FR_CNT(rj.types); const unsigned mask = ent >> FR_MASK_SHIFT;
        for (int a = 0; a < 3; a++) for (int b = 0; b < cnt_j; b++) if ((mask >> (3 * a + b)) & 1u) { const int js = j0 + b; atomicOr(&s_bits[a * words + (js >> 5)], 1u << (js & 31)); }
    }
    __syncthreads();
    unsigned long long found = 0, missing = 0;
    for (int a = 0; a < cnt_i; a++) {
        if (!((mob_i >> a) & 1u)) continue;
        const int ia = m.s0 + a;
        const uint32_t xi = img[ia].x, yi = img[ia].y, zi = img[ia].z;
        const int e0 = L.ex_start[ia], e1 = L.ex_start[ia + 1];
        for (int js = tid; js < fa.n; js += 256) {
            if (js == ia || (img[js].flags & FLAG_ALCH)) continue;
            const double dx = (double)(int32_t)(img[js].x - xi) * c.dscale[0], dy = (double)(int32_t)(img[js].y - yi) * c.dscale[1], dz = (double)(int32_t)(img[js].z - zi) * c.dscale[2];
            if (dx * dx + dy * dy + dz * dz >= (double)c.rc2) continue;
            bool excluded = false;
            for (int e = e0; e < e1; e++) excluded |= L.ex_idx[e] == js;
            if (excluded) continue;
            found++;
            if (!((s_bits[a * words + (js >> 5)] >> (js & 31)) & 1u)) missing++;
        }
    }
    if (found) atomicAdd(&out[0], found);
    if (missing) atomicAdd(&out[1], missing);
}
