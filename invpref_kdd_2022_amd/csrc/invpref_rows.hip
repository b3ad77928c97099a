// invpref_rows.hip -- the planned, atomic-free M-step ("row jobs") and its fused Adam.
//
// Why a plan: global float atomics run at ~1.3 TB/s chip-wide on MI355X whatever the schedule and
// LDS float atomics retire about one lane per clock (measured here: an LDS-accumulating variant of
// this kernel spent half its time in ds_add_f32).  The reference's minibatches are static --
// utils.mini_batch (utils.py:12-19) yields the same contiguous, unshuffled slices every epoch -- so
// the scatter pattern of each minibatch is inverted ONCE on the host (plan.py) into row jobs:
//
//   job   = one row of the user tables (or of the item tables) + the minibatch's interactions
//           that touch it, cut into 1, 2, 4, 8 or 16 equal slices, one slice per 16-lane group;
//   round = 16 group slots of one workgroup filled with jobs of one slice count;
//   task  = a few consecutive rounds of one side, run by one 256-thread workgroup.
//
// A group keeps its row's two gradient rows (invariant / env-aware table) in REGISTERS while it
// walks its slice (forward + analytic backward per interaction, partner rows gathered as coalesced
// float4 from L2 / Infinity Cache).  Slices of one row meet through LDS with plain stores and a
// fixed-order sum by the job's leader group, so the big-table gradients are bitwise reproducible.
// The leader then finishes the row: gradient = sum + count * (L2/L1 term), and EITHER stores the
// gradient row (multi-GPU path: all-reduce, then the stand-alone Adam kernel) OR applies Adam on the
// spot and writes p', m', v' (single-GPU path: the gradient never reaches memory).
// Every table row is a job, touched or not (dense-Adam semantics of torch.optim.Adam over
// nn.Embedding(sparse=False), train.py:41: momentum keeps moving untouched rows), so nothing needs
// zeroing.  Parameters are double-buffered (read old, write new): other workgroups still gather
// the old rows.  Each interaction is evaluated twice (once for its user row, once for its item
// row).  The E x D tables (embed_env, classifier) and the loss sums come from the user-side jobs:
// per-group register partials -> workgroup reduction -> float atomics into a few replica slabs
// (~2 KB per workgroup, 256-byte contiguous instructions) folded by rows_finish_kernel.
#include <stdlib.h>
#include <string.h>

#include "kernel_common.hpp"

using namespace invpref;

namespace {

#ifndef REPLICAS
#define REPLICAS 32
#endif
#ifndef ROWS_MIN_WAVES
#define ROWS_MIN_WAVES 3  // workgroups per CU the register allocator must leave room for (4 spills)
#endif

constexpr int kReplicas = REPLICAS;   // replica slabs for the E x D gradients / loss sums
// item tables up to this many rows use the direct hot-row form (HotRows::item_cnt): the finish reads every item
// row speculatively, which is only worth it while that is a few MB
constexpr int kHotDirectMaxItems = 2048;
inline bool hot_direct(const InvPrefTables *tables, const InvPrefRowPlan *plan) {
    return plan->n_hot > 0 && plan->item_hot_count && tables->item_num <= kHotDirectMaxItems;
}
inline size_t hot_scratch_rows(const InvPrefTables *tables, const InvPrefRowPlan *plan) {
    return hot_direct(tables, plan) ? (size_t)tables->item_num : (size_t)plan->n_hot;
}
constexpr int kGroups = 16;     // 16-lane groups per 256-thread workgroup

// one row of the device-side schedule (include/invpref_hip.h: InvPrefAdamSchedule)
struct SchedRow {
    AdamScalars ad;
    float alpha;   // gradient-reversal alpha of the step; NaN: use the one of the call's coefficient block
    float pad;
};
__device__ __forceinline__ const SchedRow *sched_slot_ptr(const int *state, int slot) {
    return reinterpret_cast<const SchedRow *>(state + 16 * slot + 2);
}

// Row-store cache policy, measured on the real loop (ping-pong parameter buffers, 31 different plans,
// tools/kb3.py): plain stores for everything.  The new parameters p' are gathered by the very next
// step and m', v' are re-read by it, so they should stay in the cache hierarchy: nontemporal p'
// stores cost +2 us per step (the next step's gathers miss), nontemporal m'/v' stores +0.5 us.
// (Replaying ONE minibatch without the buffer swap shows the opposite, which is why the loop is the
// benchmark.)  Load hints (nt / sc0 / sc1 on the moment prefetch and the streamed rows) made no difference.
typedef float v4f __attribute__((ext_vector_type(4)));
// build-time knobs for A/B runs: 0 plain, 1 nontemporal
#ifndef ROWS_ST_P
#define ROWS_ST_P 0
#endif
#ifndef ROWS_ST_MV
#define ROWS_ST_MV 0
#endif
template <int MODE>
__device__ __forceinline__ void store4(float *p, float4 r) {
    if (MODE == 0) *reinterpret_cast<float4 *>(p) = r;
    else if (MODE == 2) {   // write-through (sc1): leaves the XCD's L2 at once instead of at the end of the kernel
        v4f val = {r.x, r.y, r.z, r.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(val) : "memory");
    } else if (MODE == 3) {
        v4f val = {r.x, r.y, r.z, r.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(val) : "memory");
    } else {
        v4f val = {r.x, r.y, r.z, r.w};
        __builtin_nontemporal_store(val, reinterpret_cast<v4f *>(p));
    }
}
template <int NC, bool VEC, int MODE = ROWS_ST_MV>
__device__ __forceinline__ void store_row(float *__restrict__ base, int64_t row, int D, int l16, const float4 (&r)[NC]) {
    float *p = base + row * (int64_t)D;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int i0 = (l16 + kRow * c) * 4;
        if (VEC) {
            if (i0 < D) store4<MODE>(p + i0, r[c]);
        } else {
            if (i0 + 0 < D) p[i0 + 0] = r[c].x;
            if (i0 + 1 < D) p[i0 + 1] = r[c].y;
            if (i0 + 2 < D) p[i0 + 2] = r[c].z;
            if (i0 + 3 < D) p[i0 + 3] = r[c].w;
        }
    }
}

__device__ __forceinline__ void f4fma(float4 &acc, float s, float4 a) {
    acc.x = __builtin_fmaf(s, a.x, acc.x); acc.y = __builtin_fmaf(s, a.y, acc.y);
    acc.z = __builtin_fmaf(s, a.z, acc.z); acc.w = __builtin_fmaf(s, a.w, acc.w);
}
__device__ __forceinline__ void f4add(float4 &acc, float4 a) { acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
__device__ __forceinline__ float4 f4sel(bool c, float4 a) { return c ? a : f4zero(); }
__device__ __forceinline__ float4 f4xor_lanes(float4 v, int m) {
    return make_float4(__shfl_xor(v.x, m, 64), __shfl_xor(v.y, m, 64), __shfl_xor(v.z, m, 64), __shfl_xor(v.w, m, 64));
}

// folds (and re-zeroes) the replica slabs: gradients of embed_env / classifier (+ classifier
// regulariser, models.py:211-217), the six loss outputs; then either stores those gradients
// (fused == 0) or applies Adam to the three small tables (fused == 1).
struct SmallTables {
    float *gEv, *gW, *gb;        // fused == 0
    float *nEv, *nW, *nb;        // fused == 1: new parameters
    float *mEv, *mW, *mb, *vEv, *vW, *vb;
};
// item rows whose gradient arrived through atomics (no job of their own): finished here
struct HotRows {
    int n, slab_blocks;          // rows; number of leading blocks that fold the slabs
    const int *rows, *cnt;       // [n] item row id, interactions of that row in this minibatch
    // direct form (small item tables): scratch row = item id, item_cnt[item] = interactions of a hot item (0: not hot);
    // one group per ITEM, so that nothing the group loads depends on an earlier load (n = item_num then)
    const int *item_cnt;
    unsigned long long *stamps;  // diagnostic
    float *scratch;              // [n][2][DP]
    const float *Qi, *Qa;        // current item tables
    float *gQi, *gQa;            // fused == 0: gradient tables
    float *nQi, *nQa, *mQi, *mQa, *vQi, *vQa;   // fused == 1
};

// The finish kernel is one short dependent chain (kernel arguments -> loads -> Adam -> stores) on the critical path of
// every step, so its loads are written to leave in ONE burst: pointers are picked with selects, addresses are clamped
// instead of guarded, and nothing is loaded under a branch (a load under a branch is waited for at the join -- measured:
// the branchy form of this kernel spent 3.7 us of work per step, each `cond ? a[i] : b[i]` a memory round trip of its own).
// Touches every 64-byte line of the kernel-argument segment in ONE burst of scalar loads at the top of a kernel.  The
// compiler fetches arguments lazily, branch by branch; on a cold scalar cache each level is a round trip of its own
// (measured with phase stamps: 1.8 us from the entry of rows_finish_kernel to its first data), after this they all hit.
template <int BYTES>
__device__ __forceinline__ void warm_kernargs() {
    typedef const __attribute__((address_space(4))) unsigned *kptr;
    kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int LINES = (BYTES + 63) / 64;
#pragma unroll
    for (int i0 = 0; i0 < LINES; i0 += 8) {
        unsigned x[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = ka[(i0 + j < LINES ? i0 + j : LINES - 1) * 16];
        asm volatile("" ::"s"(x[0]), "s"(x[1]), "s"(x[2]), "s"(x[3]), "s"(x[4]), "s"(x[5]), "s"(x[6]), "s"(x[7]));
    }
}
// a kernel-argument pointer made resident in scalar registers HERE: without it the compiler fetches each argument
// lazily inside the branch that first needs it (one scalar-cache round trip per branch level, in sequence)
template <typename T>
__device__ __forceinline__ T *pinned(T *p) {
    asm volatile("" ::"s"(p));   // (input only: the pointer keeps its inferred address space)
    return p;
}
template <int NC, bool VEC>
__device__ __forceinline__ void finish_hot_rows(const DevTables &t, const HotRows &h, const StepScalars &k,
                                                int fused, const AdamScalars &ad, int block, bool pure) {
    constexpr int DP = NC * 64;
    const int l16 = threadIdx.x & 15;
    const int i = block * (int)(blockDim.x >> 4) + (int)(threadIdx.x >> 4);
    if (i >= h.n) return;
    const int *item_cnt = pinned(h.item_cnt), *rows = pinned(h.rows), *cnts = pinned(h.cnt);
    const float *qi = pinned(h.Qi), *qa = pinned(h.Qa), *mqi = pinned(h.mQi), *mqa = pinned(h.mQa);
    const float *vqi = pinned(h.vQi), *vqa = pinned(h.vQa);
    float *nqi = pinned(h.nQi), *nqa = pinned(h.nQa), *gqi = pinned(h.gQi), *gqa = pinned(h.gQa);
    float *sc = pinned(h.scratch) + (int64_t)i * 2 * DP;
    int row = i, icnt;
    if (item_cnt) {
        icnt = item_cnt[i];             // direct form: the row is the group's index, nothing below waits for this load
    } else {
        row = rows[i];                  // indexed form: the row loads below depend on this one
        icnt = cnts[i];
    }
    const float *Q[2] = {qi, pure ? qi : qa};                       // (PureMF: the second table is absent; its loads
    const float *M[2] = {fused ? mqi : qi, fused ? (pure ? mqi : mqa) : qi};   //  repeat the first's and are never
    const float *V[2] = {fused ? vqi : qi, fused ? (pure ? vqi : vqa) : qi};   //  used)
    float4 p[2][NC], g[2][NC], m[2][NC], v[2][NC];
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
#pragma unroll
        for (int c = 0; c < NC; c++) g[tt][c] = *reinterpret_cast<const float4 *>(sc + tt * DP + (l16 + kRow * c) * 4);
        if (VEC) {
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const int i0 = (l16 + kRow * c) * 4;
                const int64_t off = (int64_t)row * t.D + (i0 < t.D ? i0 : 0);   // clamped, not guarded
                p[tt][c] = *reinterpret_cast<const float4 *>(Q[tt] + off);
                m[tt][c] = *reinterpret_cast<const float4 *>(M[tt] + off);
                v[tt][c] = *reinterpret_cast<const float4 *>(V[tt] + off);
            }
        } else {
            load_row<NC, VEC>(Q[tt], row, t.D, l16, p[tt]);
            load_row<NC, VEC>(M[tt], row, t.D, l16, m[tt]);
            load_row<NC, VEC>(V[tt], row, t.D, l16, v[tt]);
        }
    }
    // direct form: a group whose item is not a hot row of this minibatch stops here -- its loads were speculative.  The
    // opaque statement keeps them so: without it the compiler sinks the row loads below this test, i.e. behind the
    // round trip of the count they were meant to fly with.
    asm volatile("" : "+v"(icnt)::"memory");
    if (h.stamps) {   // diagnostic: the loads have landed
        __builtin_amdgcn_s_waitcnt(0);
        if (threadIdx.x == 0) h.stamps[blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memrealtime();
    }
    if (item_cnt && icnt == 0) return;
    const float cnt = (float)icnt;
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
        if (pure && tt) break;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            *reinterpret_cast<float4 *>(sc + tt * DP + (l16 + kRow * c) * 4) = f4zero();  // zeroed for the next step
            float4 &gg = g[tt][c];
            const float4 pp = p[tt][c];
            gg.x += cnt * (k.r2 * pp.x + k.r1 * c_sign(pp.x)); gg.y += cnt * (k.r2 * pp.y + k.r1 * c_sign(pp.y));
            gg.z += cnt * (k.r2 * pp.z + k.r1 * c_sign(pp.z)); gg.w += cnt * (k.r2 * pp.w + k.r1 * c_sign(pp.w));
        }
        if (!fused) {
            store_row<NC, VEC>(tt == 0 ? gqi : gqa, row, t.D, l16, g[tt]);
        } else {
#pragma unroll
            for (int c = 0; c < NC; c++) {
                adam1f(p[tt][c].x, g[tt][c].x, m[tt][c].x, v[tt][c].x, ad); adam1f(p[tt][c].y, g[tt][c].y, m[tt][c].y, v[tt][c].y, ad);
                adam1f(p[tt][c].z, g[tt][c].z, m[tt][c].z, v[tt][c].z, ad); adam1f(p[tt][c].w, g[tt][c].w, m[tt][c].w, v[tt][c].w, ad);
            }
            store_row<NC, VEC, ROWS_ST_P>(tt == 0 ? nqi : nqa, row, t.D, l16, p[tt]);
            store_row<NC, VEC>(const_cast<float *>(tt == 0 ? mqi : mqa), row, t.D, l16, m[tt]);
            store_row<NC, VEC>(const_cast<float *>(tt == 0 ? vqi : vqa), row, t.D, l16, v[tt]);
        }
    }
}
// everything one minibatch's finish needs (folding the slabs, the three small tables, the hot rows, the loss
// outputs): the stand-alone rows_finish_kernel takes it as its argument, and a chained launch (invpref_chain_*)
// carries the PREVIOUS step's block inside the next mstep_rows_kernel, whose leading workgroups run it
struct FinishArgs {
    DevTables t;                 // tables the step read (hot rows / small tables: values before the update)
    SmallTables o;
    HotRows hot;
    float *slabs;
    int nslabs, DP, EMAX, nc, vec, fused;
    StepScalars k;
    float l2, l1;
    int64_t Bnorm;
    double inv_B, inv_BD2;       // 1 / Bnorm, 1 / (2 Bnorm D): the loss epilogue multiplies (fp64 divisions are ~0.1 us each,
                                 // at the very end of the step's chain)
    uint32_t flags;
    AdamScalars ad;
    float *losses6;
    unsigned long long *stamps;  // diagnostic (INVPREF_STAMPS): [block][8] s_memrealtime ticks
};
// diagnostic phase stamp of the finish kernel (drains the wave's memory operations first); never executed unless a
// stamp buffer is passed
#define FSTAMP(i)                                                                                         \
    do {                                                                                                  \
        if (f.stamps) {                                                                                   \
            __builtin_amdgcn_s_waitcnt(0);                                                                \
            if (threadIdx.x == 0) f.stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime();      \
        }                                                                                                 \
    } while (0)

// one block of the finish: `block` < hot.slab_blocks folds 64 columns of the slabs with kFinishSubs sub-rows of threads
// (part: [kFinishSubs][64] doubles of LDS, sloss: [kLossSlots]); later blocks finish blockDim/16 hot rows.
// 512 threads: the fold wants many threads with few loads each (one wave per block with 32 loads per thread measured
// 0.5 us slower: a wave issues its loads one after the other), and 512 leave every instance its registers.
#ifndef ROWS_FIN_THREADS
#define ROWS_FIN_THREADS 512
#endif
constexpr int kFinishThreads = ROWS_FIN_THREADS, kFinishSubs = kFinishThreads / 64;
template <int NC, bool VEC>
__device__ __forceinline__ void finish_block(const FinishArgs &f, const AdamScalars &ad, int block, double *part,
                                             double *sloss) {
    const DevTables &t = f.t;
    const SmallTables &o = f.o;
    const HotRows &hot = f.hot;
    const StepScalars &k = f.k;
    const int DP = f.DP, EMAX = f.EMAX, fused = f.fused, nslabs = f.nslabs;
    const uint32_t flags = f.flags;
    float *slabs = f.slabs, *losses6 = f.losses6;
    const float l2 = f.l2, l1 = f.l1;
    FSTAMP(0);
    if (block >= hot.slab_blocks) {
        finish_hot_rows<NC, VEC>(t, hot, k, fused, ad, block - hot.slab_blocks, flags & INVPREF_PURE_MF);
        FSTAMP(3);
        return;
    }
    const int EDP = t.E * DP, slab_len = 2 * EDP + EMAX + kLossSlots;
    const int colx = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int idx = block * 64 + colx;
    const bool pure = flags & INVPREF_PURE_MF;  // no small tables to finish: only the loss sums are folded
    const bool dense = (flags & INVPREF_DENSE_REG) && !(flags & INVPREF_REG_ONLY_EMBED) && !pure;
    const bool last_block = block == hot.slab_blocks - 1;
    // parameter / moments of the output this thread will finish (sub == 0 threads), requested up front: the table is
    // picked with selects and the offset clamped, so the three loads leave together (see finish_hot_rows)
    float pre_p = 0.f, pre_m = 0.f, pre_v = 0.f;
    if (sub == 0 && idx < 2 * EDP + EMAX) {
        const bool isB = idx >= 2 * EDP, isW = !isB && idx >= EDP;
        const int r = isB ? 0 : (isW ? idx - EDP : idx);
        const int e = isB ? idx - 2 * EDP : r / DP, d = isB ? 0 : r - e * DP;
        const bool live = e < t.E && d < t.D && !pure;
        const int off = live ? (isB ? e : e * t.D + d) : 0;
        const float *tb = pinned(t.b), *tW = pinned(t.W), *tEv = pinned(t.Ev);
        const float *mb = pinned(o.mb), *mW = pinned(o.mW), *mEv = pinned(o.mEv);
        const float *vb = pinned(o.vb), *vW = pinned(o.vW), *vEv = pinned(o.vEv);
        pinned(o.nb); pinned(o.nW); pinned(o.nEv); pinned(o.gb); pinned(o.gW); pinned(o.gEv);
        const float *pp = live ? (isB ? tb : (isW ? tW : tEv)) : slabs;
        const float *mp = (live && fused) ? (isB ? mb : (isW ? mW : mEv)) : pp;
        const float *vp = (live && fused) ? (isB ? vb : (isW ? vW : vEv)) : pp;
        pre_p = pp[off]; pre_m = mp[off]; pre_v = vp[off];
    }
    // this thread's column of the replicas sub, sub + kFinishSubs, ...: CH loads in flight together (clamped, not
    // guarded), summed in replica order in fp64; the replicas are re-zeroed at the very end of the block, behind
    // everything that is waited for (a store between the loads and their use is waited for with them)
    constexpr int CH = kReplicas / kFinishSubs > 4 ? kReplicas / kFinishSubs : 4;   // one trip covers all replicas
    const bool mine = idx < slab_len;
    const float *col = slabs + (mine ? idx : 0);
    double acc = 0.0;
    double reg2 = 0.0, reg1 = 0.0;
    for (int s0 = sub; s0 < nslabs; s0 += CH * kFinishSubs) {
        float x[CH];
#pragma unroll
        for (int j = 0; j < CH; j++)
            x[j] = __builtin_nontemporal_load(col + (int64_t)min(s0 + j * kFinishSubs, nslabs - 1) * slab_len);
        if (s0 == sub && last_block && dense && losses6 && sub == 0) {
            // regulariser report of the classifier (models.py:211-217), while the replica loads are in flight
            double w2 = 0, w1 = 0, b2 = 0, b1 = 0;
            for (int i = threadIdx.x; i < t.E * t.D; i += 64) { const double xx = t.W[i]; w2 += xx * xx; w1 += fabs(xx); }
            for (int i = threadIdx.x; i < t.E; i += 64) { const double xx = t.b[i]; b2 += xx * xx; b1 += fabs(xx); }
            reg2 = w2 / ((double)t.D * t.E) + b2 / (double)t.E;
            reg1 = w1 / ((double)t.D * t.E) + b1 / (double)t.E;
            for (int m = 32; m >= 1; m >>= 1) { reg2 += __shfl_xor(reg2, m, 64); reg1 += __shfl_xor(reg1, m, 64); }
        }
#pragma unroll
        for (int j = 0; j < CH; j++) acc += (s0 + j * kFinishSubs < nslabs) ? (double)x[j] : 0.0;
    }
    part[sub * 64 + colx] = acc;
    __syncthreads();
    double v = 0.0;
    if (sub == 0) {
#pragma unroll
        for (int q = 0; q < kFinishSubs; q++) v += part[q * 64 + colx];
    }
    FSTAMP(1);
    if (mine && sub == 0) {
        if (idx < 2 * EDP) {
            const bool isW = idx >= EDP;
            const int r = isW ? idx - EDP : idx;
            const int e = r / DP, d = r - e * DP;
            if (d < t.D && !pure) {
                const int off = e * t.D + d;
                float gv = (float)v;
                float pv = pre_p;
                if (isW && dense) gv += 2.f * l2 / ((float)t.D * (float)t.E) * pv + l1 / ((float)t.D * (float)t.E) * c_sign(pv);
                if (!fused) {
                    (isW ? o.gW : o.gEv)[off] = gv;
                } else {
                    float *mp = (isW ? o.mW : o.mEv) + off, *vp = (isW ? o.vW : o.vEv) + off;
                    float mm = pre_m, vv = pre_v;
                    adam1(pv, gv, mm, vv, ad);
                    (isW ? o.nW : o.nEv)[off] = pv; *mp = mm; *vp = vv;
                }
            }
        } else if (idx < 2 * EDP + EMAX) {
            const int e = idx - 2 * EDP;
            if (e < t.E && !pure) {
                float gv = (float)v, pv = pre_p;
                if (dense) gv += 2.f * l2 / (float)t.E * pv + l1 / (float)t.E * c_sign(pv);
                if (!fused) {
                    o.gb[e] = gv;
                } else {
                    float mm = pre_m, vv = pre_v;
                    adam1(pv, gv, mm, vv, ad);
                    o.nb[e] = pv; o.mb[e] = mm; o.vb[e] = vv;
                }
            }
        } else {
            sloss[idx - 2 * EDP - EMAX] = v;
        }
    }
    FSTAMP(2);
    if (last_block && losses6) {
        __syncthreads();   // (workgroup-uniform condition: the loss sums written above are visible to thread 0)
        if (threadIdx.x == 0) {
            const double Li = sloss[0] * f.inv_B, Le = sloss[1] * f.inv_B, Lc = sloss[2] * f.inv_B;
            const double L2 = sloss[3] * f.inv_BD2 + reg2, L1 = sloss[4] * f.inv_BD2 + reg1;
            // added with fire-and-forget atomics: a plain `+=` would hold the kernel's end back by one more
            // memory round trip (the same single fp32 addition either way)
            atomicAdd(losses6 + 0, (float)Li); atomicAdd(losses6 + 1, (float)Le); atomicAdd(losses6 + 2, (float)Lc);
            atomicAdd(losses6 + 3, (float)L2); atomicAdd(losses6 + 4, (float)L1);
            atomicAdd(losses6 + 5, (float)((double)k.ca * Li + (double)k.cb * Le + (double)k.cc * Lc + (double)l2 * L2 + (double)l1 * L1));
        }
    }
    if (mine)
        for (int s = sub; s < nslabs; s += kFinishSubs) slabs[(int64_t)s * slab_len + idx] = 0.f;  // zeroed for the next step
    FSTAMP(3);
}

struct RowsArgs {
    const int4 *desc;             // [n_rounds][16][2]: see InvPrefRowPlan in include/invpref_hip.h
    int n_rounds, n_item_rounds, rounds_per_task;
    const int *oth[2], *pos[2];   // per side, in that side's sorted order: partner row, position in the minibatch
    const int64_t *envs;          // minibatch base pointers, indexed by pos
    const float *scores, *weights;
    StepScalars k;
    uint32_t flags;
    float *slabs;
    int fused;                    // 0: store gradient rows to g; 1: Adam on the spot -> np / m / v
    float *g[4];                  // Pu, Qi, Pa, Qa gradient tables   (fused == 0)
    float *np[4];                 // new parameter tables              (fused == 1)
    float *m[4], *v[4];           // Adam moments                      (fused == 1)
    AdamScalars ad;
    // rows the minibatch does not touch: no job, just the dense-Adam step (or a zero gradient row),
    // streamed by dedicated workgroups with several rows in flight per group
    const int *stream_rows;       // [n_stream_user + n_stream_item] row ids, user rows first
    int n_stream_user, n_stream_item, rows_per_stream_task, n_job_tasks, n_stream_user_tasks;
    int n_cls;                    // XCD-affine task order: workgroup b runs tasks of class b % n_cls (InvPrefRowPlan)
    int cls[8][8];
    const int *batch_users, *batch_items;   // [n] ids of the minibatch in its own order (dense tasks)
    int n, dense_per_task, n_dense_tasks;
    const int *item_hot_index;    // [item_num]: scratch row of an item whose gradient goes through atomics, or -1
    float *hot_scratch;           // [n_hot][2][DP] gradient accumulators of those rows (zero on entry, re-zeroed by finish)
    int hot_direct;               // the accumulator row of a hot item is its item id (HotRows::item_cnt form)
    int *sched_state;             // optional device int32[32]: two slots {step, base, SchedRow}, see InvPrefAdamSchedule
    const SchedRow *sched_table;  // optional device table of per-step scalars (graph replay)
    int sched_n, sched_slot;
    int stamps_nodrain;           // diagnostic: do not drain memory operations before a stamp
    unsigned long long *stamps;   // diagnostic builds only (INVPREF_STAMPS): [n_tasks][8] s_memrealtime ticks
};

// diagnostic phase stamp: drains the wave's outstanding memory operations first, so the latency of
// a phase is charged to that phase.  Never executed unless a stamp buffer is passed.
#define STAMP(i)                                                                  \
    do {                                                                          \
        if (a.stamps) {                                                           \
            if (!a.stamps_nodrain) __builtin_amdgcn_s_waitcnt(0);                 \
            if (threadIdx.x == 0) a.stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                         \
    } while (0)

struct Sample {
    int oth, ps;
    float y;
};

// ---- forward + analytic backward of ONE interaction on a 16-lane group (M-step arithmetic: hardware
// exp/log/rcp).  Rows in canonical roles: pu/qi invariant user/item rows, pa/qa env-aware rows, ev the env row.
template <int NC, int EMAX>
struct Eval {
    float g_p, g_q, li, le, lcls;
    float gz[EMAX];          // EMAX <= 4: every lane holds all classes
    float gz_lane;           // EMAX > 4: lane l16 of the group holds class l16 (0 beyond E)
    float4 x[NC], gx[NC];   // x = Pu*Qi ; gx = sum_c gz_c W_c
};
template <int NC, int EMAX, bool HOISTW = false>
__device__ __forceinline__ void eval_interaction(Eval<NC, EMAX> &o, const float4 (&pu)[NC], const float4 (&qi)[NC],
                                                 const float4 (&pa)[NC], const float4 (&qa)[NC], const float4 (&ev)[NC],
                                                 const float *sW, const float *sb, int E, int e, float y, float cw_rec,
                                                 float cw_cls, const StepScalars &k, bool implicit, int l16) {
    const float p = dot2<NC>(pu, qi), q = dot3<NC>(pa, qa, ev);
    if (implicit) {
        const float sp = f_sigmoid(p), sq = f_sigmoid(q), sv = sp * sq;
        o.li = f_bce(sp, y);
        o.le = f_bce(sv, y);
        const float d_inv = k.ca * cw_rec * f_dbce(sp, y);
        const float d_env = k.cb * cw_rec * f_dbce(sv, y);
        o.g_p = (d_inv + d_env * sq) * (sp * (1.f - sp));
        o.g_q = d_env * sp * (sq * (1.f - sq));
    } else {
        const float s2 = p + q;
        o.li = (p - y) * (p - y);
        o.le = (s2 - y) * (s2 - y);
        const float d_env = k.cb * cw_rec * 2.f * (s2 - y);
        o.g_p = k.ca * cw_rec * 2.f * (p - y) + d_env;
        o.g_q = d_env;
    }
#pragma unroll
    for (int c = 0; c < NC; c++) o.x[c] = f4mul(pu[c], qi[c]);
    // (measured per shape, tools/kbench.py: the unrolled branch-free loops below win only at E = 8 with two row chunks)
    constexpr bool LANEZ = EMAX > 4 && !(EMAX == 8 && NC == 2);
    if (LANEZ) {
        // Larger classifiers keep ONE class per lane instead of all classes in every lane: the logit of class c is a
        // group-uniform value after the row reduction, lane c keeps it; max / sum of the softmax are 16-lane
        // reductions, one exp per lane instead of E per lane, and the backward fetches gz_c from lane c.  No
        // per-class register arrays, no per-class branches: the class loops run E times with the next class's LDS
        // reads issued early.
        float zmine = -__builtin_inff();
#pragma unroll 2
        for (int c = 0; c < E; c++) {
            float4 wr[NC];
            lds_row<NC>(sW, c, l16, wr);
            const float zc = dot2<NC>(o.x, wr) + sb[c];
            zmine = (l16 == c) ? zc : zmine;
        }
        const float mxl = row16_max(zmine);
        const float ez = l16 < E ? f_exp(zmine - mxl) : 0.f;
        const float rsel = f_rcp(row16_sum(ez));
        const float zel = __shfl(ez, e, 16);
        o.lcls = -f_log(zel * rsel);
        const float gzl = l16 < E ? k.cc * cw_cls * (ez * rsel - (l16 == e ? 1.f : 0.f)) : 0.f;
        o.gz_lane = gzl;
#pragma unroll
        for (int c = 0; c < NC; c++) o.gx[c] = f4zero();
#pragma unroll 2
        for (int c = 0; c < E; c++) {
            const float g = __shfl(gzl, c, 16);
            float4 wr[NC];
            lds_row<NC>(sW, c, l16, wr);
#pragma unroll
            for (int jj = 0; jj < NC; jj++) f4fma(o.gx[jj], g, wr[jj]);
        }
        return;
    }
    float z[EMAX], mx = -__builtin_inff();
    // HOISTW (dense tasks, small classifiers): the W rows are read from LDS ONCE, unconditionally and back to
    // back, and everything after is selects and arithmetic -- with a guarded LDS read per class the E forward
    // dots and the E backward rows each pay their own LDS round trip in sequence.  Rows c >= E of the staged
    // area hold other data: they are masked to zero, their logits to -inf (exp -> 0, gz -> 0).
    constexpr bool HOIST = HOISTW && EMAX * NC <= 4;
    constexpr bool FLAT = !HOIST && EMAX * NC > 4 && EMAX * NC <= 16;   // (beyond that the unrolled loops spill)
    float4 wrow[HOIST ? EMAX : 1][NC];
    if (HOIST) {
#pragma unroll
        for (int c = 0; c < EMAX; c++) {
            float4 t4[NC];
            lds_row<NC>(sW, c, l16, t4);
#pragma unroll
            for (int jj = 0; jj < NC; jj++) wrow[c][jj] = f4sel(c < E, t4[jj]);
        }
#pragma unroll
        for (int c = 0; c < EMAX; c++) {
            const float zc = dot2<NC>(o.x, wrow[c]) + sb[c];
            z[c] = c < E ? zc : -__builtin_inff();
            mx = z[c] > mx ? z[c] : mx;
        }
    } else if (FLAT) {
        // larger classifiers: too many rows to keep, but still no branch per class -- every class's LDS read is
        // unconditional (rows c >= E of the staged area hold other data and are masked), so the reads of the
        // next classes are in flight while this one's dot product is reduced
#pragma unroll
        for (int c = 0; c < EMAX; c++) {
            float4 wr[NC];
            lds_row<NC>(sW, c, l16, wr);
            const float zc = dot2<NC>(o.x, wr) + sb[c];
            z[c] = c < E ? zc : -__builtin_inff();
            mx = z[c] > mx ? z[c] : mx;
        }
    } else {
#pragma unroll
        for (int c = 0; c < EMAX; c++) {
            z[c] = -__builtin_inff();
            if (c < E) {
                float4 wr[NC];
                lds_row<NC>(sW, c, l16, wr);
                z[c] = dot2<NC>(o.x, wr) + sb[c];
                mx = z[c] > mx ? z[c] : mx;
            }
        }
    }
    float se = 0.f, ze = 0.f;
#pragma unroll
    for (int c = 0; c < EMAX; c++) {
        if (HOIST || FLAT) { z[c] = f_exp(z[c] - mx); se += z[c]; }   // exp(-inf) = 0 for the padded classes
        else if (c < E) { z[c] = f_exp(z[c] - mx); se += z[c]; }
    }
#pragma unroll
    for (int c = 0; c < EMAX; c++) ze = (c == e) ? z[c] : ze;
    const float rse = f_rcp(se);
    o.lcls = -f_log(ze * rse);
#pragma unroll
    for (int c = 0; c < NC; c++) o.gx[c] = f4zero();
#pragma unroll
    for (int c = 0; c < EMAX; c++) {
        if (HOIST) {
            o.gz[c] = c < E ? k.cc * cw_cls * (z[c] * rse - (c == e ? 1.f : 0.f)) : 0.f;
#pragma unroll
            for (int jj = 0; jj < NC; jj++) f4fma(o.gx[jj], o.gz[c], wrow[c][jj]);
        } else if (FLAT) {
            o.gz[c] = c < E ? k.cc * cw_cls * (z[c] * rse - (c == e ? 1.f : 0.f)) : 0.f;
            float4 wr[NC];
            lds_row<NC>(sW, c, l16, wr);
#pragma unroll
            for (int jj = 0; jj < NC; jj++) f4fma(o.gx[jj], o.gz[c], f4sel(c < E, wr[jj]));
        } else {
            o.gz[c] = 0.f;
            if (c < E) {
                o.gz[c] = k.cc * cw_cls * (z[c] * rse - (c == e ? 1.f : 0.f));
                float4 wr[NC];
                lds_row<NC>(sW, c, l16, wr);
#pragma unroll
                for (int jj = 0; jj < NC; jj++) f4fma(o.gx[jj], o.gz[c], wr[jj]);
            }
        }
    }
}

// =====================================================================================
// job task: rounds of row jobs of ONE side.  Symmetric in the two sides: a job only produces its own
// row's two gradient rows; everything that is a reduction ACROSS rows (E x D gradients, loss sums,
// regulariser reports, hot-row atomics) belongs to the dense tasks below.
// =====================================================================================
template <int NC, bool VEC, int EMAX, bool USER>
__device__ __forceinline__ void rows_task(const DevTables &t, const RowsArgs &a, const int4 task, float *lds) {
    constexpr int DP = NC * 64;
    constexpr int side = USER ? 0 : 1;
    const int EDP = t.E * DP;
    float *slots = lds;                                  // [16][2][DP] slice partials
    float *sEv = slots + kGroups * 2 * DP, *sW = sEv + EDP, *sb = sW + EDP;   // staged small tables
    // Adam moments of the rows being finished, prefetched by LDS-DMA (no VGPRs held across the
    // interaction loop): [4 waves][m_inv, v_inv, m_env, v_env][NC chunks][64 lanes] float4
    float4 *mv = reinterpret_cast<float4 *>(sb + EMAX);

    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4, wave = threadIdx.x >> 6;
    // (NC = 4 rows would need 64 KB of LDS for the prefetch: those fetch the moments late instead)
    const bool dma = VEC && NC <= 2 && a.fused;
    float4 *mv_wave = mv + wave * 4 * NC * 64;
    const bool implicit = a.flags & INVPREF_IMPLICIT;
    const bool rw_rec = a.flags & INVPREF_REWEIGHT_REC, rw_cls = a.flags & INVPREF_REWEIGHT_CLS;
    // PureMF (INVPREF_PURE_MF): the env-aware tables, embed_env and the classifier are absent -- their
    // rows stay the zeros they are initialised to below and are neither loaded nor stored
    const bool pure = a.flags & INVPREF_PURE_MF;
    StepScalars k = a.k;
    if (a.sched_state) {  // scheduled alpha (train.py:214-217) under graph replay
        const float al = sched_slot_ptr(a.sched_state, a.sched_slot)->alpha;
        if (al == al) k.alpha = al;
    }
    const int *oth_ids = a.oth[side], *pos = a.pos[side];
    const float *T_own_inv = USER ? t.Pu : t.Qi, *T_own_env = USER ? t.Pa : t.Qa;
    const float *T_oth_inv = USER ? t.Qi : t.Pu, *T_oth_env = USER ? t.Qa : t.Pa;
    // Adam scalars of this step: by value, or (graph replay: kernel arguments are frozen) looked up
    // by the device-side step counter that rows_finish_kernel advances
    const AdamScalars ad = a.sched_state ? sched_slot_ptr(a.sched_state, a.sched_slot)->ad : a.ad;

    STAMP(0);
    // the first round's descriptor goes out before anything else: every gather below hangs on it
    int4 d = a.desc[(task.y * kGroups + grp) * 2], d1 = a.desc[(task.y * kGroups + grp) * 2 + 1];
    stage_table(sEv, t.Ev, t.E, t.D, DP);
    stage_table(sW, t.W, t.E, t.D, DP);
    for (int i = threadIdx.x; i < EMAX; i += blockDim.x) sb[i] = (i < t.E && t.b) ? t.b[i] : 0.f;
    STAMP(1);

    for (int r = task.y; r < task.y + task.z; r++) {
        // one 32-byte descriptor per group slot; slices of up to two interactions carry them inline
        // (partner row, position, label), so the gathers below depend on this single load only
        if (r != task.y) { d = a.desc[(r * kGroups + grp) * 2]; d1 = a.desc[(r * kGroups + grp) * 2 + 1]; }
        const int row = d.x, meta = d.y;
        const bool active = row >= 0, leader = meta & 1;
        const int slices = (meta >> 1) & 31, mode = (meta >> 6) & 3;
        const int nsmp = mode == 3 ? d.w - d.z : mode;
        if (r == task.y) STAMP(2);
        auto sample_at = [&](int sidx) {
            Sample sm;
            if (mode == 3) { sm.oth = oth_ids[d.z + sidx]; sm.ps = pos[d.z + sidx]; sm.y = a.scores[sm.ps]; }
            else if (sidx == 0) { sm.oth = d.z; sm.ps = d.w; sm.y = __builtin_bit_cast(float, d1.x); }
            else { sm.oth = d1.y; sm.ps = d1.z; sm.y = __builtin_bit_cast(float, d1.w); }
            return sm;
        };
        // everything that depends only on the descriptor is requested together: own rows, the Adam
        // moments of the row (needed last, LDS-DMA) and the first interaction's partner rows / env / weight
        float4 oi[NC], oe[NC], gi[NC], ge[NC], pi[NC], pe[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) oi[c] = oe[c] = gi[c] = ge[c] = pi[c] = pe[c] = f4zero();
        Sample cur{0, 0, 0.f};
        int e = 0;
        float w = 1.f;
        if (active) {
            load_row<NC, VEC>(T_own_inv, row, t.D, l16, oi);
            if (!pure) load_row<NC, VEC>(T_own_env, row, t.D, l16, oe);
            if (nsmp > 0) {
                cur = sample_at(0);
                load_row<NC, VEC>(T_oth_inv, cur.oth, t.D, l16, pi);
                if (!pure) {
                    load_row<NC, VEC>(T_oth_env, cur.oth, t.D, l16, pe);
                    e = (int)a.envs[cur.ps];
                }
                if (rw_rec || rw_cls) w = a.weights[cur.ps];
            }
        }
        if (dma) {
            // each lane sends its 16-byte piece of the row's four moment rows straight to LDS; the
            // destination of a wave instruction is one contiguous 1 KiB block, lane-major
            const bool mine = active && leader;
#pragma unroll
            for (int tn = 0; tn < 4; tn++) {
                const float *src_tab = (tn & 1) ? a.v[(tn >> 1) * 2 + side] : a.m[(tn >> 1) * 2 + side];
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const int i0 = (l16 + kRow * c) * 4;
                    if (mine && i0 < t.D && !(pure && tn >= 2))
                        __builtin_amdgcn_global_load_lds(
                            (const __attribute__((address_space(1))) void *)(src_tab + (int64_t)row * t.D + i0),
                            (__attribute__((address_space(3))) void *)(mv_wave + (tn * NC + c) * 64), 16, 0, 0);
                }
            }
        }
        if (r == task.y) { __syncthreads(); STAMP(3); }  // staged tables visible (the gathers above are in flight)
        for (int sidx = 0; sidx < nsmp; sidx++) {
            if (sidx > 0) {  // (the first interaction's gathers were issued with the own rows)
                cur = sample_at(sidx);
                load_row<NC, VEC>(T_oth_inv, cur.oth, t.D, l16, pi);
                if (!pure) {
                    load_row<NC, VEC>(T_oth_env, cur.oth, t.D, l16, pe);
                    e = (int)a.envs[cur.ps];
                }
                if (rw_rec || rw_cls) w = a.weights[cur.ps];
            }
            const float cw_rec = (rw_rec ? w : 1.f) * k.invB, cw_cls = (rw_cls ? w : 1.f) * k.invB;
            float4 ev[NC];
            lds_row<NC>(sEv, e, l16, ev);
            Eval<NC, EMAX> o;
            if (USER) eval_interaction<NC, EMAX>(o, oi, pi, oe, pe, ev, sW, sb, t.E, e, cur.y, cw_rec, cw_cls, k, implicit, l16);
            else eval_interaction<NC, EMAX>(o, pi, oi, pe, oe, ev, sW, sb, t.E, e, cur.y, cw_rec, cw_cls, k, implicit, l16);
#pragma unroll
            for (int jj = 0; jj < NC; jj++) {
                float4 gip;
                gip.x = o.g_p - k.alpha * o.gx[jj].x; gip.y = o.g_p - k.alpha * o.gx[jj].y;
                gip.z = o.g_p - k.alpha * o.gx[jj].z; gip.w = o.g_p - k.alpha * o.gx[jj].w;
                f4add(gi[jj], f4mul(gip, pi[jj]));
                f4fma(ge[jj], o.g_q, f4mul(pe[jj], ev[jj]));
            }
        }
        if (r == task.y) STAMP(4);
        // ---- slices of one row meet through LDS: plain stores, fixed-order sum by the leader
        if (slices > 1) {  // same for every slot of a round, idle slots included
            float *mine = slots + grp * 2 * DP;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                *reinterpret_cast<float4 *>(mine + (l16 + kRow * c) * 4) = gi[c];
                *reinterpret_cast<float4 *>(mine + DP + (l16 + kRow * c) * 4) = ge[c];
            }
            __syncthreads();
            if (active && leader) {
#pragma unroll 4
                for (int s = 1; s < slices; s++) {
                    const float *oth_slot = slots + (grp + s) * 2 * DP;
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        f4add(gi[c], *reinterpret_cast<const float4 *>(oth_slot + (l16 + kRow * c) * 4));
                        f4add(ge[c], *reinterpret_cast<const float4 *>(oth_slot + DP + (l16 + kRow * c) * 4));
                    }
                }
            }
            if (r + 1 < task.y + task.z) __syncthreads();  // the slots are rewritten by the next round
        }
        if (r == task.y) STAMP(5);
        // ---- the leader finishes the row
        if (active && leader) {
            const float cnt = (float)(meta >> 8);
            if (cnt != 0.f) {
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    gi[c].x += cnt * (k.r2 * oi[c].x + k.r1 * c_sign(oi[c].x)); gi[c].y += cnt * (k.r2 * oi[c].y + k.r1 * c_sign(oi[c].y));
                    gi[c].z += cnt * (k.r2 * oi[c].z + k.r1 * c_sign(oi[c].z)); gi[c].w += cnt * (k.r2 * oi[c].w + k.r1 * c_sign(oi[c].w));
                    ge[c].x += cnt * (k.r2 * oe[c].x + k.r1 * c_sign(oe[c].x)); ge[c].y += cnt * (k.r2 * oe[c].y + k.r1 * c_sign(oe[c].y));
                    ge[c].z += cnt * (k.r2 * oe[c].z + k.r1 * c_sign(oe[c].z)); ge[c].w += cnt * (k.r2 * oe[c].w + k.r1 * c_sign(oe[c].w));
                }
            }
            if (!a.fused) {
                store_row<NC, VEC>(a.g[side], row, t.D, l16, gi);
                if (!pure) store_row<NC, VEC>(a.g[2 + side], row, t.D, l16, ge);
            } else {
                float4 mi[NC], vi[NC], me[NC], ve[NC];
                if (dma) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the LDS-DMA pieces have landed
                    const int lane = threadIdx.x & 63;
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        mi[c] = mv_wave[(0 * NC + c) * 64 + lane]; vi[c] = mv_wave[(1 * NC + c) * 64 + lane];
                        me[c] = mv_wave[(2 * NC + c) * 64 + lane]; ve[c] = mv_wave[(3 * NC + c) * 64 + lane];
                    }
                } else {
                    load_row<NC, VEC>(a.m[side], row, t.D, l16, mi);
                    load_row<NC, VEC>(a.v[side], row, t.D, l16, vi);
                    if (!pure) {
                        load_row<NC, VEC>(a.m[2 + side], row, t.D, l16, me);
                        load_row<NC, VEC>(a.v[2 + side], row, t.D, l16, ve);
                    }
                }
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    adam1f(oi[c].x, gi[c].x, mi[c].x, vi[c].x, ad); adam1f(oi[c].y, gi[c].y, mi[c].y, vi[c].y, ad);
                    adam1f(oi[c].z, gi[c].z, mi[c].z, vi[c].z, ad); adam1f(oi[c].w, gi[c].w, mi[c].w, vi[c].w, ad);
                    adam1f(oe[c].x, ge[c].x, me[c].x, ve[c].x, ad); adam1f(oe[c].y, ge[c].y, me[c].y, ve[c].y, ad);
                    adam1f(oe[c].z, ge[c].z, me[c].z, ve[c].z, ad); adam1f(oe[c].w, ge[c].w, me[c].w, ve[c].w, ad);
                }
                store_row<NC, VEC, ROWS_ST_P>(a.np[side], row, t.D, l16, oi);
                store_row<NC, VEC>(a.m[side], row, t.D, l16, mi);
                store_row<NC, VEC>(a.v[side], row, t.D, l16, vi);
                if (!pure) {
                    store_row<NC, VEC, ROWS_ST_P>(a.np[2 + side], row, t.D, l16, oe);
                    store_row<NC, VEC>(a.m[2 + side], row, t.D, l16, me);
                    store_row<NC, VEC>(a.v[2 + side], row, t.D, l16, ve);
                }
            }
        }
    }
    STAMP(6);
}

// =====================================================================================
// dense task: a contiguous run of the minibatch's interactions, 16 at a time.  Everything that is a
// reduction ACROSS rows lives here, off the critical path of the row jobs:
//   * E x D gradients of embed_env / classifier (+ bias): every group records x = Pu*Qi,
//     o = g_q*Pa*Qa (+ env regulariser), gz[0..E) and the env id in LDS; after a barrier every thread adds
//     all 16 records into the outputs it OWNS (one column d, a strided set of classes c) held in
//     registers -- an outer-product accumulation on the vector ALU, no atomics, any E*D;
//   * the five loss / regulariser-report sums;
//   * the gradient of HOT item rows (float atomics shaped as 64 contiguous bytes per instruction).
// The workgroup's totals go to one of the replica slabs with float atomics (~2 KB per workgroup).
// =====================================================================================
template <int NC, bool VEC, int EMAX>
__device__ __forceinline__ void dense_task(const DevTables &t, const RowsArgs &a, int s0, int s1, float *lds) {
    constexpr int DP = NC * 64;
    const int EDP = t.E * DP;
    float *sEv = lds, *sW = sEv + EDP, *sb = sW + EDP;
    // records are double-buffered (one barrier per iteration) and the hot-row transpose has its own buffer
    // while that fits comfortably (NC <= 2); NC = 4 uses one record buffer for everything and more barriers
    constexpr bool DBUF = NC <= 2;
    float *rec0 = sb + EMAX;                             // [DBUF ? 2 : 1][16][2][DP]  x, o
    float *recs0 = rec0 + (DBUF ? 2 : 1) * kGroups * 2 * DP;   // [DBUF ? 2 : 1][16][EMAX + 1]  gz[0..EMAX), env id (-1: none)
    float *aL = recs0 + (DBUF ? 2 : 1) * kGroups * (EMAX + 1); // [kLossSlots]
    float *trbuf = aL + kLossSlots;                      // DBUF: [16][2][DP] hot-row transpose buffer
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const bool implicit = a.flags & INVPREF_IMPLICIT;
    const bool rw_rec = a.flags & INVPREF_REWEIGHT_REC, rw_cls = a.flags & INVPREF_REWEIGHT_CLS;
    const bool reg_env = a.flags & INVPREF_REG_ENV_EMBED;
    const bool pure = a.flags & INVPREF_PURE_MF;
    StepScalars k = a.k;
    if (a.sched_state) {
        const float al = sched_slot_ptr(a.sched_state, a.sched_slot)->alpha;
        if (al == al) k.alpha = al;
    }
    STAMP(0);
    // The ids and the four rows of an iteration are requested one phase early.  First iteration: the ids go
    // out before the staging loads (both fly together), the rows before the barrier that makes the staged
    // tables visible.  Next iteration: the ids at the top of this one, the rows right after this iteration's
    // records are written (its rows are dead by then) -- they fly under the barrier and the accumulation.
    const int first_s = s0 + (int)(threadIdx.x >> 4);
    int id_u = 0, id_v = 0, id_e = -1;
    float id_y = 0.f, id_w = 1.f;
    if (first_s < s1) {
        id_u = a.batch_users[first_s]; id_v = a.batch_items[first_s];
        id_e = pure ? 0 : (int)a.envs[first_s];
        id_y = a.scores[first_s];
        id_w = (rw_rec || rw_cls) ? a.weights[first_s] : 1.f;
    }
    stage_table(sEv, t.Ev, t.E, t.D, DP);
    stage_table(sW, t.W, t.E, t.D, DP);
    for (int i = threadIdx.x; i < EMAX; i += blockDim.x) sb[i] = (i < t.E && t.b) ? t.b[i] : 0.f;
    if (threadIdx.x < kLossSlots) aL[threadIdx.x] = 0.f;
    // output ownership: thread -> column d_own, classes cg, cg + CG, ...
    constexpr int CG = (256 / DP) < EMAX ? (256 / DP) : EMAX;
    constexpr int CPT = (EMAX + CG - 1) / CG;
    const int d_own = threadIdx.x % DP, cg = threadIdx.x / DP;
    float dW[CPT], dE[CPT], dB[CPT];
#pragma unroll
    for (int i = 0; i < CPT; i++) dW[i] = dE[i] = dB[i] = 0.f;
    float accLi = 0.f, accLe = 0.f, accLc = 0.f, accL2 = 0.f, accL1 = 0.f;
    int n_u = id_u, n_v = id_v, n_e = id_e, n_hidx = -1;
    float n_y = id_y, n_w = id_w;
    float4 pu[NC], qi[NC], pa[NC], qa[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) pu[c] = qi[c] = pa[c] = qa[c] = f4zero();
    auto fetch_ids = [&](int base) {      // ids / label / weight of the group's interaction of iteration `base`
        const int s = base + grp;
        n_e = -1; n_hidx = -1;
        if (s < s1) {
            n_u = a.batch_users[s]; n_v = a.batch_items[s];
            n_e = pure ? 0 : (int)a.envs[s];
            n_y = a.scores[s];
            n_w = (rw_rec || rw_cls) ? a.weights[s] : 1.f;
        }
    };
    auto fetch_rows = [&]() {             // its four rows (+ the hot-row index of the item)
        if (n_e >= 0) {
            if (a.item_hot_index) n_hidx = a.item_hot_index[n_v];
            load_row<NC, VEC>(t.Pu, n_u, t.D, l16, pu);
            load_row<NC, VEC>(t.Qi, n_v, t.D, l16, qi);
            if (!pure) {
                load_row<NC, VEC>(t.Pa, n_u, t.D, l16, pa);
                load_row<NC, VEC>(t.Qa, n_v, t.D, l16, qa);
            }
        }
    };
    fetch_rows();
    __syncthreads();
    STAMP(1);

    for (int base = s0, it = 0; base < s1; base += kGroups, it++) {
        float *rec = rec0 + (DBUF ? (it & 1) : 0) * kGroups * 2 * DP;
        float *recs = recs0 + (DBUF ? (it & 1) : 0) * kGroups * (EMAX + 1);
        const int e = n_e, hidx = n_hidx;
        const int hrow = a.hot_direct ? n_v : n_hidx;   // accumulator row of a hot item
        const float cur_y = n_y, cur_w = n_w;
        const bool valid = e >= 0;
        const bool more = base + kGroups < s1;
        if (more) fetch_ids(base + kGroups);
        float4 hq[NC], ha[NC];  // hot item row gradients of this interaction
#pragma unroll
        for (int c = 0; c < NC; c++) hq[c] = ha[c] = f4zero();
        if (valid) {
            const float y = cur_y, w = cur_w;
            const float w_rec = rw_rec ? w : 1.f, w_cls = rw_cls ? w : 1.f;
            float4 ev[NC];
            lds_row<NC>(sEv, e, l16, ev);
            Eval<NC, EMAX> o;
            eval_interaction<NC, EMAX, true>(o, pu, qi, pa, qa, ev, sW, sb, t.E, e, y, w_rec * k.invB, w_cls * k.invB, k, implicit, l16);
            // record for the E x D accumulation
#pragma unroll
            for (int jj = 0; jj < NC; jj++) {
                float4 oo = f4mul(pa[jj], qa[jj]);
                oo.x *= o.g_q; oo.y *= o.g_q; oo.z *= o.g_q; oo.w *= o.g_q;
                if (reg_env) {
                    oo.x += 2.f * k.r2 * ev[jj].x + 2.f * k.r1 * c_sign(ev[jj].x);
                    oo.y += 2.f * k.r2 * ev[jj].y + 2.f * k.r1 * c_sign(ev[jj].y);
                    oo.z += 2.f * k.r2 * ev[jj].z + 2.f * k.r1 * c_sign(ev[jj].z);
                    oo.w += 2.f * k.r2 * ev[jj].w + 2.f * k.r1 * c_sign(ev[jj].w);
                }
                *reinterpret_cast<float4 *>(rec + (grp * 2) * DP + (l16 + kRow * jj) * 4) = o.x[jj];
                *reinterpret_cast<float4 *>(rec + (grp * 2 + 1) * DP + (l16 + kRow * jj) * 4) = oo;
                if (hidx >= 0) {
                    float4 gip;
                    gip.x = o.g_p - k.alpha * o.gx[jj].x; gip.y = o.g_p - k.alpha * o.gx[jj].y;
                    gip.z = o.g_p - k.alpha * o.gx[jj].z; gip.w = o.g_p - k.alpha * o.gx[jj].w;
                    hq[jj] = f4mul(gip, pu[jj]);
                    ha[jj] = f4mul(pa[jj], ev[jj]);
                    ha[jj].x *= o.g_q; ha[jj].y *= o.g_q; ha[jj].z *= o.g_q; ha[jj].w *= o.g_q;
                }
                // regulariser REPORTS over the four rows of the interaction (env rows weigh double: 1/(BD) vs 1/(2BD))
                float s2 = pu[jj].x * pu[jj].x + pu[jj].y * pu[jj].y + pu[jj].z * pu[jj].z + pu[jj].w * pu[jj].w;
                s2 += pa[jj].x * pa[jj].x + pa[jj].y * pa[jj].y + pa[jj].z * pa[jj].z + pa[jj].w * pa[jj].w;
                s2 += qi[jj].x * qi[jj].x + qi[jj].y * qi[jj].y + qi[jj].z * qi[jj].z + qi[jj].w * qi[jj].w;
                s2 += qa[jj].x * qa[jj].x + qa[jj].y * qa[jj].y + qa[jj].z * qa[jj].z + qa[jj].w * qa[jj].w;
                float s1 = fabsf(pu[jj].x) + fabsf(pu[jj].y) + fabsf(pu[jj].z) + fabsf(pu[jj].w);
                s1 += fabsf(pa[jj].x) + fabsf(pa[jj].y) + fabsf(pa[jj].z) + fabsf(pa[jj].w);
                s1 += fabsf(qi[jj].x) + fabsf(qi[jj].y) + fabsf(qi[jj].z) + fabsf(qi[jj].w);
                s1 += fabsf(qa[jj].x) + fabsf(qa[jj].y) + fabsf(qa[jj].z) + fabsf(qa[jj].w);
                if (reg_env) {
                    s2 += 2.f * (ev[jj].x * ev[jj].x + ev[jj].y * ev[jj].y + ev[jj].z * ev[jj].z + ev[jj].w * ev[jj].w);
                    s1 += 2.f * (fabsf(ev[jj].x) + fabsf(ev[jj].y) + fabsf(ev[jj].z) + fabsf(ev[jj].w));
                }
                accL2 += s2;
                accL1 += s1;
            }
            if (l16 < EMAX) {
                float gsel = 0.f;
                if (EMAX > 4 && !(EMAX == 8 && NC == 2)) gsel = o.gz_lane;
                else {
#pragma unroll
                    for (int c = 0; c < EMAX; c++) gsel = (l16 == c) ? o.gz[c] : gsel;
                }
                recs[grp * (EMAX + 1) + l16] = gsel;
            }
            if (l16 == 0) { accLi += o.li * w_rec; accLe += o.le * w_rec; accLc += o.lcls * w_cls; }
        }
        if (l16 == 0) recs[grp * (EMAX + 1) + EMAX] = __builtin_bit_cast(float, e);
        if (more) fetch_rows();
        __syncthreads();
        if (threadIdx.x < CG * DP) {
            // Branch-free on purpose (the loads of all 16 records are issued back to back; with per-record
            // branches every record paid two LDS round trips in sequence).  An empty record slot holds env
            // id -1: its stale x / o / gz are masked by selects.  Classes c >= E accumulate garbage that is
            // never stored.
            constexpr int BATCH = CPT <= 2 ? 8 : 2;   // records whose LDS reads are in flight together
#pragma unroll 1
            for (int g0 = 0; g0 < kGroups; g0 += BATCH) {
                int er[BATCH];
                float xr[BATCH], orr[BATCH], gr[BATCH][CPT];
#pragma unroll
                for (int b = 0; b < BATCH; b++) {       // unconditional loads first ...
                    const float *rs = recs + (g0 + b) * (EMAX + 1);
                    er[b] = __builtin_bit_cast(int, rs[EMAX]);
                    xr[b] = rec[((g0 + b) * 2) * DP + d_own];
                    orr[b] = rec[((g0 + b) * 2 + 1) * DP + d_own];
#pragma unroll
                    for (int i = 0; i < CPT; i++) gr[b][i] = rs[cg + CG * i];
                }
#pragma unroll
                for (int b = 0; b < BATCH; b++) {       // ... then selects and arithmetic only
                    const bool ok = er[b] >= 0;
                    const float xv = ok ? xr[b] : 0.f;
#pragma unroll
                    for (int i = 0; i < CPT; i++) {
                        const float gzc = ok ? gr[b][i] : 0.f;
                        dW[i] = __builtin_fmaf(gzc, xv, dW[i]);
                        dE[i] += (cg + CG * i == er[b]) ? orr[b] : 0.f;
                        dB[i] += gzc;
                    }
                }
            }
        }
        if (!DBUF) __syncthreads();
        // hot item rows: shaped atomics through a transpose buffer (NC = 4: the now free record slot)
        if (hidx >= 0) {
            float *tr = (DBUF ? trbuf : rec) + grp * 2 * DP;
            float *dst = a.hot_scratch + (int64_t)hrow * 2 * DP;
#pragma unroll
            for (int jj = 0; jj < NC; jj++) {
                *reinterpret_cast<float4 *>(tr + (l16 + kRow * jj) * 4) = hq[jj];
                *reinterpret_cast<float4 *>(tr + DP + (l16 + kRow * jj) * 4) = ha[jj];
            }
            WAVE_LDS_FENCE();
#pragma unroll
            for (int q4 = 0; q4 < 2 * DP / 16; q4++) {
                const int idx = q4 * 16 + l16;           // 16 lanes -> 16 consecutive floats
                if ((idx & (DP - 1)) < t.D && !(pure && idx >= DP)) atomicAdd(dst + idx, tr[idx]);  // (PureMF: no env-aware half)
            }
        }
        if (!DBUF) __syncthreads();  // the record slots are rewritten by the next iteration
    }
    STAMP(6);
    // ---- this workgroup's totals -> replica slab
    accLi = wave_sum(accLi); accLe = wave_sum(accLe); accLc = wave_sum(accLc);
    accL2 = wave_sum(accL2); accL1 = wave_sum(accL1);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(aL + 0, accLi); atomicAdd(aL + 1, accLe); atomicAdd(aL + 2, accLc);
        atomicAdd(aL + 3, accL2); atomicAdd(aL + 4, accL1);
    }
    __syncthreads();
    const int slab_len = 2 * EDP + EMAX + kLossSlots;
    float *slab = a.slabs + (int64_t)(blockIdx.x % kReplicas) * slab_len;
    if (threadIdx.x < CG * DP) {
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int c = cg + CG * i;
            if (c < t.E && d_own < t.D) {
                if (dE[i] != 0.f) atomicAdd(slab + c * DP + d_own, dE[i]);
                if (dW[i] != 0.f) atomicAdd(slab + EDP + c * DP + d_own, dW[i]);
                if (d_own == 0 && dB[i] != 0.f) atomicAdd(slab + 2 * EDP + c, dB[i]);
            }
        }
    }
    if (threadIdx.x < kLossSlots) { const float x = aL[threadIdx.x]; if (x != 0.f) atomicAdd(slab + 2 * EDP + EMAX + threadIdx.x, x); }
    STAMP(7);
}

// Untouched rows: gradient exactly zero, so m' = m + (1-b1)(0-m), v' = b2 v, p' = p - step*m'/(sqrt(v')/bc+eps)
// (the same adam1f as everywhere, fed g = 0).  Each 16-lane group keeps R = 2 rows of both tables in flight (12 float4
// loads; R = 3 was tried for 96-row tasks: 159 VGPRs, or 40 spilled at the 128 a one-wave step needs); one row at 256 floats.
template <int NC, bool VEC>
__device__ __forceinline__ void stream_task(const DevTables &t, const RowsArgs &a, int side, const int *rows, int n) {
    constexpr int R = NC == 4 ? 1 : 2;   // (four row chunks: two rows in flight are 192 registers -- that alone spilled 250)
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const AdamScalars ad = a.sched_state ? sched_slot_ptr(a.sched_state, a.sched_slot)->ad : a.ad;
    const float *Tinv = side == 0 ? t.Pu : t.Qi, *Tenv = side == 0 ? t.Pa : t.Qa;
    const bool pure = a.flags & INVPREF_PURE_MF;
    for (int i = grp; i < n; i += R * kGroups) {
        int row[R];
        bool on[R];
#pragma unroll
        for (int q = 0; q < R; q++) {
            on[q] = i + q * kGroups < n;
            row[q] = rows[on[q] ? i + q * kGroups : i];
        }
        if (!a.fused) {
            float4 z[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) z[c] = f4zero();
#pragma unroll
            for (int q = 0; q < R; q++) {
                if (!on[q]) continue;
                store_row<NC, VEC>(a.g[side], row[q], t.D, l16, z);
                if (!pure) store_row<NC, VEC>(a.g[2 + side], row[q], t.D, l16, z);
            }
            continue;
        }
        float4 p[2 * R][NC], m[2 * R][NC], v[2 * R][NC];  // {row 0 inv, row 0 env, row 1 inv, ...}
#pragma unroll
        for (int q = 0; q < 2 * R; q++) {
            const int ti = (q & 1) * 2 + side;
            if (!(pure && (q & 1))) {
                load_row<NC, VEC>((q & 1) ? Tenv : Tinv, row[q >> 1], t.D, l16, p[q]);
                load_row<NC, VEC>(a.m[ti], row[q >> 1], t.D, l16, m[q]);
                load_row<NC, VEC>(a.v[ti], row[q >> 1], t.D, l16, v[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 2 * R; q++) {
            const int ti = (q & 1) * 2 + side;
            if (on[q >> 1] && !(pure && (q & 1))) {
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    adam1f(p[q][c].x, 0.f, m[q][c].x, v[q][c].x, ad); adam1f(p[q][c].y, 0.f, m[q][c].y, v[q][c].y, ad);
                    adam1f(p[q][c].z, 0.f, m[q][c].z, v[q][c].z, ad); adam1f(p[q][c].w, 0.f, m[q][c].w, v[q][c].w, ad);
                }
                store_row<NC, VEC, ROWS_ST_P>(a.np[ti], row[q >> 1], t.D, l16, p[q]);
                store_row<NC, VEC>(a.m[ti], row[q >> 1], t.D, l16, m[q]);
                store_row<NC, VEC>(a.v[ti], row[q >> 1], t.D, l16, v[q]);
            }
        }
    }
}

// larger rows / classifiers need more registers than 168: at 3 waves per SIMD they spill hundreds of bytes per lane
// (the MIND-shaped step ran 2.3 ms); with room for 256 registers the same step takes 1.1 ms
#ifndef ROWS_MIN_WAVES_BIG
#define ROWS_MIN_WAVES_BIG 2
#endif
template <int NC, bool VEC, int EMAX>
__global__ __launch_bounds__(256, (NC * EMAX > 4) ? ROWS_MIN_WAVES_BIG : ROWS_MIN_WAVES)
void mstep_rows_kernel(DevTables t, RowsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef ROWS_WARM_MAIN
    warm_kernargs<sizeof(DevTables) + sizeof(RowsArgs)>();
#endif
    // Workgroup b runs the tasks of class c = b % n_cls (XCD-affine order, InvPrefRowPlan), the j-th of them with
    // j = b / n_cls: the class's share of the dense tasks first (they walk several interactions in sequence), then its
    // item jobs, user jobs, streamed user rows, streamed item rows.  Every branch below is workgroup-uniform.
    const int ncls = a.n_cls;
    const int c = (int)blockIdx.x % ncls;
    int j = (int)blockIdx.x / ncls;
    const int nd = a.n_dense_tasks > c ? (a.n_dense_tasks - c + ncls - 1) / ncls : 0;
    if (j < nd) {
        const int s0 = (c + ncls * j) * a.dense_per_task;
        dense_task<NC, VEC, EMAX>(t, a, s0, min(s0 + a.dense_per_task, a.n), lds);
        return;
    }
    j -= nd;
    const int rpt = a.rounds_per_task, spt = a.rows_per_stream_task;
    // (a masked sum over the eight rows instead of a dynamic index -- or of selects, which the optimiser turns back
    //  into one: indexing a by-value kernel argument with a run-time value can make the compiler copy the whole
    //  argument block to scratch memory -- seen with a larger argument block: 1.1 KB per lane, 80 us per launch)
    int q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int on = (c == k) ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 8; i++) q[i] += a.cls[k][i] * on;
    }
    const int i_base = q[0], i_n = q[1], u_base = q[2], u_n = q[3];
    const int su_base = q[4], su_n = q[5], si_base = q[6], si_n = q[7];
    const int ti = (i_n + rpt - 1) / rpt, tu = (u_n + rpt - 1) / rpt;
    if (j < ti) {
        rows_task<NC, VEC, EMAX, false>(t, a, make_int4(1, i_base + j * rpt, min(rpt, i_n - j * rpt), 0), lds);
        return;
    }
    j -= ti;
    if (j < tu) {
        rows_task<NC, VEC, EMAX, true>(t, a, make_int4(0, u_base + j * rpt, min(rpt, u_n - j * rpt), 0), lds);
        return;
    }
    j -= tu;
    const int tsu = (su_n + spt - 1) / spt, tsi = (si_n + spt - 1) / spt;
    if (j < tsu + tsi) {  // untouched rows
        STAMP(0);
        const bool us = j < tsu;
        const int first = (us ? j : j - tsu) * spt;
        stream_task<NC, VEC>(t, a, us ? 0 : 1, a.stream_rows + (us ? su_base : si_base) + first,
                             min(spt, (us ? su_n : si_n) - first));
        STAMP(7);
    }
    // (a class with fewer tasks than the longest one: nothing to do)
}


template <int NC, bool VEC>
__global__ __launch_bounds__(kFinishThreads) void rows_finish_kernel(const FinishArgs f, int *sched_state,
                                                                     const SchedRow *sched_table, int sched_n, int sched_slot) {
    __shared__ double part[kFinishSubs * 64];
    __shared__ double sloss[kLossSlots];
#ifdef ROWS_EMPTY_FINISH   // A/B builds: what a bare kernel boundary costs
    if (f.fused >= 0) return;
#endif
#ifndef ROWS_NO_WARM
    warm_kernargs<sizeof(FinishArgs) + 32>();
#endif
    // the device-side schedule moves on: one thread of the LAST block (a block with nothing else to do, so that the two
    // dependent loads of this look-up are on nobody's chain) fills the OTHER slot with the next step's number and
    // scalars.  Nobody reads that slot before the next launch, so no ordering between blocks is needed.
    // (in the gradient-pass form -- fused == 0 -- the stand-alone Adam kernel that follows is the step's last
    //  launch and moves the schedule on; here the slot is only read, for a scheduled alpha)
    if (blockIdx.x == gridDim.x - 1) {
        if (sched_state && f.fused && threadIdx.x == 0) {
            const int *cur = sched_state + 16 * sched_slot;
            int *nxt = sched_state + 16 * (sched_slot ^ 1);
            const int next = cur[0] + 1, base = cur[1], idx = next - base;
            nxt[0] = next;
            nxt[1] = base;
            if (idx >= 0 && idx < sched_n) *reinterpret_cast<SchedRow *>(nxt + 2) = sched_table[idx];
        }
        return;
    }
    const AdamScalars ad = sched_state ? sched_slot_ptr(sched_state, sched_slot)->ad : f.ad;
    finish_block<NC, VEC>(f, ad, blockIdx.x, part, sloss);
}


void *g_profile_event = nullptr;  // see invpref_set_profile_event()

size_t rows_lds_bytes(int E, int nc, int emax) {
    const size_t DP = (size_t)nc * 64, EDP = (size_t)E * DP;
    // job task: slice slots + staged tables + LDS-DMA moment buffers; dense task: staged tables + records
    // (+ (emax - E) rows: the branch-free classifier loops read EMAX rows of the staged classifier, masked beyond E)
    const size_t job = sizeof(float) * (kGroups * 2 * DP + 2 * EDP + emax + (size_t)(emax - E) * DP) +
                       (nc <= 2 ? 16 * (size_t)(4 * 4 * nc * 64) : 0);
    const size_t nbuf = nc <= 2 ? 2 : 1;
    const size_t dense = sizeof(float) * (2 * EDP + emax + nbuf * (kGroups * 2 * DP + kGroups * (emax + 1)) + kLossSlots +
                                          (nc <= 2 ? kGroups * 2 * DP : 0));
    return job > dense ? job : dense;
}

template <typename K>
int ensure_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// the finish block of one step (see FinishArgs): `t` = the tables the step read, `a` = its launch arguments
FinishArgs make_finish_args(const DevTables &t, const RowsArgs &a, const InvPrefRowPlan *plan, const InvPrefCoefs *coefs,
                            int64_t batch_norm, uint32_t flags, int fused, const InvPrefTables *grads,
                            const InvPrefTables *new_tables, const InvPrefTables *exp_avg, const InvPrefTables *exp_avg_sq,
                            const AdamScalars &ad, float *losses6, float *slabs, float *hot_scratch, int DP, int emax,
                            int nc, int vec) {
    FinishArgs f{};
    f.t = t;
    if (!fused) {
        f.o.gEv = grads->embed_env; f.o.gW = grads->classifier_weight; f.o.gb = grads->classifier_bias;
    } else {
        f.o.nEv = new_tables->embed_env; f.o.nW = new_tables->classifier_weight; f.o.nb = new_tables->classifier_bias;
        f.o.mEv = exp_avg->embed_env; f.o.mW = exp_avg->classifier_weight; f.o.mb = exp_avg->classifier_bias;
        f.o.vEv = exp_avg_sq->embed_env; f.o.vW = exp_avg_sq->classifier_weight; f.o.vb = exp_avg_sq->classifier_bias;
    }
    const int slab_len = 2 * t.E * DP + emax + kLossSlots;
    HotRows &h = f.hot;
    h.n = plan->n_hot; h.slab_blocks = (slab_len + 63) / 64; h.rows = plan->hot_rows; h.cnt = plan->hot_count;
    if (a.hot_direct) { h.n = t.I; h.item_cnt = plan->item_hot_count; }
    h.scratch = hot_scratch;
    h.Qi = t.Qi; h.Qa = t.Qa;
    if (!fused) { h.gQi = a.g[1]; h.gQa = a.g[3]; }
    else { h.nQi = a.np[1]; h.nQa = a.np[3]; h.mQi = a.m[1]; h.mQa = a.m[3]; h.vQi = a.v[1]; h.vQa = a.v[3]; }
    f.slabs = slabs; f.nslabs = kReplicas; f.DP = DP; f.EMAX = emax; f.nc = nc; f.vec = vec; f.fused = fused;
    f.k = a.k; f.l2 = coefs->L2_coe; f.l1 = coefs->L1_coe; f.Bnorm = batch_norm; f.flags = flags; f.ad = ad;
    f.inv_B = 1.0 / (double)batch_norm; f.inv_BD2 = 1.0 / ((double)batch_norm * (double)t.D * 2.0);
    f.losses6 = losses6;
    return f;
}

int launch_rows(const InvPrefTables *tables, const InvPrefRowPlan *plan, const int64_t *envs, const float *scores,
                const float *weights, int64_t batch_norm, const InvPrefCoefs *coefs, uint32_t flags, float *losses6,
                void *workspace, size_t workspace_bytes, hipStream_t st, int fused, const InvPrefTables *grads,
                const InvPrefTables *new_tables, const InvPrefTables *exp_avg, const InvPrefTables *exp_avg_sq,
                const AdamScalars &ad, const InvPrefAdamSchedule *sched = nullptr) {
    const bool pure = flags & INVPREF_PURE_MF;
    int rc = check_tables(tables, pure);
    if (sched && (!sched->state || !sched->table || sched->n <= 0)) return INVPREF_EINVAL;
    if (rc) return rc;
    if (!plan || !coefs || !losses6 || !workspace || (!envs && !pure) || !scores || batch_norm <= 0) return INVPREF_EINVAL;
    if (pure && (flags & (INVPREF_REWEIGHT_CLS | INVPREF_REG_ENV_EMBED))) return INVPREF_EINVAL;
    if (plan->n_rounds < 0 || plan->rounds_per_task <= 0 || plan->n_item_rounds < 0 ||
        plan->n_item_rounds > plan->n_rounds || plan->n_item_rounds % plan->rounds_per_task != 0 || !plan->desc ||
        !plan->other_user || !plan->pos_user || !plan->other_item || !plan->pos_item)
        return INVPREF_EINVAL;
    if ((flags & (INVPREF_REWEIGHT_REC | INVPREF_REWEIGHT_CLS)) && !weights) return INVPREF_EINVAL;
    const DevTables t = dev_tables(tables);
    bool vec = vec_ok(tables);
    RowsArgs a{};
    if (!fused) {
        if ((rc = check_tables(grads, pure))) return rc;
        vec = vec && vec_ok(grads);
        a.g[0] = grads->embed_user_invariant; a.g[1] = grads->embed_item_invariant;
        a.g[2] = grads->embed_user_env_aware; a.g[3] = grads->embed_item_env_aware;
    } else {
        if ((rc = check_tables(new_tables, pure)) || (rc = check_tables(exp_avg, pure)) ||
            (rc = check_tables(exp_avg_sq, pure)))
            return rc;
        vec = vec && vec_ok(new_tables) && vec_ok(exp_avg) && vec_ok(exp_avg_sq);
        const InvPrefTables *src[3] = {new_tables, exp_avg, exp_avg_sq};
        float **dst[3] = {a.np, a.m, a.v};
        for (int i = 0; i < 3; i++) {
            dst[i][0] = src[i]->embed_user_invariant; dst[i][1] = src[i]->embed_item_invariant;
            dst[i][2] = src[i]->embed_user_env_aware; dst[i][3] = src[i]->embed_item_env_aware;
        }
    }
    const int nc = vec ? nc_of(t.D) : 4, emax = emax_of(t.E);
    const int DP = nc * 64, EDP = t.E * DP;
    const int slab_len = 2 * EDP + emax + kLossSlots;
    if (plan->n_hot < 0 || (plan->n_hot > 0 && (!plan->hot_rows || !plan->hot_count || !plan->item_hot_index)))
        return INVPREF_EINVAL;
    if (workspace_bytes < sizeof(float) * ((size_t)slab_len * kReplicas + hot_scratch_rows(tables, plan) * 2 * DP))
        return INVPREF_EWORKSPACE;
    StepScalars k;
    k.ca = coefs->invariant_coe; k.cb = coefs->env_aware_coe; k.cc = coefs->env_coe; k.alpha = coefs->alpha;
    k.invB = 1.0f / (float)batch_norm;
    k.r2 = coefs->L2_coe / ((float)batch_norm * (float)t.D);
    k.r1 = coefs->L1_coe / (2.0f * (float)batch_norm * (float)t.D);
    a.desc = reinterpret_cast<const int4 *>(plan->desc);
    a.n_rounds = plan->n_rounds; a.n_item_rounds = plan->n_item_rounds; a.rounds_per_task = plan->rounds_per_task;
    const int n_job_tasks = (plan->n_rounds + plan->rounds_per_task - 1) / plan->rounds_per_task;
    if (plan->n_stream_user < 0 || plan->n_stream_item < 0 || plan->rows_per_stream_task <= 0 ||
        ((plan->n_stream_user + plan->n_stream_item) > 0 && !plan->stream_rows))
        return INVPREF_EINVAL;
    a.stream_rows = plan->stream_rows; a.n_stream_user = plan->n_stream_user; a.n_stream_item = plan->n_stream_item;
    a.rows_per_stream_task = plan->rows_per_stream_task; a.n_job_tasks = n_job_tasks;
    a.n_stream_user_tasks = (plan->n_stream_user + plan->rows_per_stream_task - 1) / plan->rows_per_stream_task;
    if (plan->dense_per_task <= 0 || (plan->n > 0 && (!plan->batch_users || !plan->batch_items))) return INVPREF_EINVAL;
    a.batch_users = plan->batch_users; a.batch_items = plan->batch_items; a.n = plan->n;
    a.dense_per_task = plan->dense_per_task;
    a.n_dense_tasks = (plan->n + plan->dense_per_task - 1) / plan->dense_per_task;
    // XCD-affine order: the grid holds n_cls interleaved task lists, padded to the longest
    a.n_cls = plan->n_classes > 0 ? plan->n_classes : 1;
    if (a.n_cls > 8) return INVPREF_EINVAL;
    if (plan->n_classes > 0) {
        memcpy(a.cls, plan->cls, sizeof(a.cls));
    } else {  // the plain order as one class
        const int tmp[8] = {0, plan->n_item_rounds, plan->n_item_rounds, plan->n_rounds - plan->n_item_rounds, 0,
                            plan->n_stream_user, plan->n_stream_user, plan->n_stream_item};
        memset(a.cls, 0, sizeof(a.cls));
        memcpy(a.cls[0], tmp, sizeof(tmp));
    }
    int per_class = 0;
    for (int c = 0; c < a.n_cls; c++) {
        const int *q = a.cls[c];
        if (q[0] < 0 || q[1] < 0 || q[2] < 0 || q[3] < 0 || q[4] < 0 || q[5] < 0 || q[6] < 0 || q[7] < 0 ||
            q[0] + q[1] > plan->n_item_rounds || q[2] < plan->n_item_rounds || q[2] + q[3] > plan->n_rounds ||
            q[4] + q[5] > plan->n_stream_user + plan->n_stream_item || q[6] + q[7] > plan->n_stream_user + plan->n_stream_item)
            return INVPREF_EINVAL;
        const int rpt = plan->rounds_per_task, spt = plan->rows_per_stream_task;
        const int nd = a.n_dense_tasks > c ? (a.n_dense_tasks - c + a.n_cls - 1) / a.n_cls : 0;
        const int tot = nd + (q[1] + rpt - 1) / rpt + (q[3] + rpt - 1) / rpt + (q[5] + spt - 1) / spt + (q[7] + spt - 1) / spt;
        per_class = tot > per_class ? tot : per_class;
    }
    const int n_tasks = per_class * a.n_cls;
    if (n_tasks <= 0) return INVPREF_EINVAL;
    a.oth[0] = plan->other_user; a.pos[0] = plan->pos_user;
    a.oth[1] = plan->other_item; a.pos[1] = plan->pos_item;
    a.envs = envs; a.scores = scores; a.weights = weights;
    a.k = k; a.flags = flags; a.slabs = (float *)workspace; a.fused = fused; a.ad = ad;
    a.item_hot_index = plan->n_hot > 0 ? plan->item_hot_index : nullptr;
    a.hot_direct = hot_direct(tables, plan) ? 1 : 0;
    a.hot_scratch = (float *)workspace + (size_t)slab_len * kReplicas;
    a.sched_state = sched ? sched->state : nullptr;
    a.sched_table = sched ? reinterpret_cast<const SchedRow *>(sched->table) : nullptr;
    a.sched_n = sched ? sched->n : 0;
    a.sched_slot = sched ? (sched->slot & 1) : 0;
    // diagnostics: INVPREF_STAMPS=<device pointer, hex> makes the kernel write phase time stamps there
    static const char *stamp_env = getenv("INVPREF_STAMPS");
    a.stamps = stamp_env ? reinterpret_cast<unsigned long long *>(strtoull(stamp_env, nullptr, 16)) : nullptr;
    static const bool nodrain = getenv("INVPREF_STAMPS_NODRAIN") != nullptr;
    a.stamps_nodrain = nodrain;
    const size_t lds = rows_lds_bytes(t.E, nc, emax);
    if (lds > 160 * 1024) return INVPREF_EUNSUPPORTED;
#define CALL(NCV, VECV, EMAXV)                                                                        \
    do {                                                                                              \
        if ((rc = ensure_lds(mstep_rows_kernel<NCV, VECV, EMAXV>, lds))) return rc;                   \
        hipLaunchKernelGGL((mstep_rows_kernel<NCV, VECV, EMAXV>), dim3(n_tasks), dim3(256), lds, st, t, a); \
    } while (0)
    if (!vec) {
        if (emax == 4) CALL(4, false, 4); else if (emax == 8) CALL(4, false, 8); else CALL(4, false, 16);
    } else if (nc == 1) {
        if (emax == 4) CALL(1, true, 4); else if (emax == 8) CALL(1, true, 8); else CALL(1, true, 16);
    } else if (nc == 2) {
        if (emax == 4) CALL(2, true, 4); else if (emax == 8) CALL(2, true, 8); else CALL(2, true, 16);
    } else {
        if (emax == 4) CALL(4, true, 4); else if (emax == 8) CALL(4, true, 8); else CALL(4, true, 16);
    }
#undef CALL
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return (int)err;
    if (g_profile_event) {  // profiling aid (bench.py): time stamp between the main and the finish kernel
        err = hipEventRecord((hipEvent_t)g_profile_event, st);
        if (err != hipSuccess) return (int)err;
    }
    FinishArgs f = make_finish_args(t, a, plan, coefs, batch_norm, flags, fused, grads, new_tables, exp_avg, exp_avg_sq, ad,
                                    losses6, (float *)workspace, a.hot_scratch, DP, emax, nc, (int)vec);
    f.stamps = a.stamps ? a.stamps + 12288 * 8 : nullptr;   // (the stamp buffer's last quarter belongs to the finish blocks)
    f.hot.stamps = f.stamps;
    const int hot_blocks = (f.hot.n + kFinishThreads / 16 - 1) / (kFinishThreads / 16);
#define FIN(NCV, VECV)                                                                                              \
    hipLaunchKernelGGL((rows_finish_kernel<NCV, VECV>), dim3(f.hot.slab_blocks + hot_blocks + 1), dim3(kFinishThreads), 0, st, \
                       f, a.sched_state, a.sched_table, a.sched_n, a.sched_slot)
#ifndef ROWS_NO_FINISH_LAUNCH   // (A/B builds, timing only: the step without its second launch)
    if (!vec) FIN(4, false); else if (nc == 1) FIN(1, true); else if (nc == 2) FIN(2, true); else FIN(4, true);
#endif
#undef FIN
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int invpref_set_profile_event(void *event) {
    g_profile_event = event;
    return 0;
}

size_t invpref_rows_workspace_bytes(const InvPrefTables *tables, const InvPrefRowPlan *plan) {
    if (check_tables(tables, tables && !tables->embed_user_env_aware) || !plan || plan->n_hot < 0) return 0;
    const size_t slab_len = 2 * (size_t)tables->env_num * 256 + 16 + kLossSlots;
    return sizeof(float) * (slab_len * kReplicas + hot_scratch_rows(tables, plan) * 2 * 256);
}

int invpref_mstep_rows_grad_hip(const InvPrefTables *tables, const InvPrefTables *grads, const InvPrefRowPlan *plan,
                                const int64_t *envs, const float *scores, const float *sample_weights,
                                int64_t batch_norm, const InvPrefCoefs *coefs, uint32_t flags, float *losses6,
                                void *workspace, size_t workspace_bytes, void *stream) {
    return launch_rows(tables, plan, envs, scores, sample_weights, batch_norm, coefs, flags, losses6, workspace,
                       workspace_bytes, (hipStream_t)stream, 0, grads, nullptr, nullptr, nullptr, AdamScalars{});
}

int invpref_mstep_rows_grad_sched_hip(const InvPrefTables *tables, const InvPrefTables *grads, const InvPrefRowPlan *plan,
                                      const int64_t *envs, const float *scores, const float *sample_weights,
                                      int64_t batch_norm, const InvPrefCoefs *coefs, uint32_t flags, float *losses6,
                                      const InvPrefAdamSchedule *sched, void *workspace, size_t workspace_bytes,
                                      void *stream) {
    if (!sched) return INVPREF_EINVAL;
    return launch_rows(tables, plan, envs, scores, sample_weights, batch_norm, coefs, flags, losses6, workspace,
                       workspace_bytes, (hipStream_t)stream, 0, grads, nullptr, nullptr, nullptr, AdamScalars{}, sched);
}

int invpref_mstep_rows_adam_hip(const InvPrefTables *tables, const InvPrefTables *new_tables,
                                const InvPrefTables *exp_avg, const InvPrefTables *exp_avg_sq,
                                const InvPrefRowPlan *plan, const int64_t *envs, const float *scores,
                                const float *sample_weights, int64_t batch_norm, const InvPrefCoefs *coefs,
                                uint32_t flags, float *losses6, int64_t step, double lr, double beta1, double beta2,
                                double eps, void *workspace, size_t workspace_bytes, void *stream) {
    if (step < 1) return INVPREF_EINVAL;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamScalars ad;
    ad.step_size = (float)(lr / bc1);
    ad.bc2_sqrt = (float)sqrt(bc2);
    ad.w1 = (float)(1.0 - beta1);
    ad.b2 = (float)beta2;
    ad.w2 = (float)(1.0 - beta2);
    ad.eps = (float)eps;
    return launch_rows(tables, plan, envs, scores, sample_weights, batch_norm, coefs, flags, losses6, workspace,
                       workspace_bytes, (hipStream_t)stream, 1, nullptr, new_tables, exp_avg, exp_avg_sq, ad);
}

/* host helper: the per-step Adam scalars exactly as invpref_adam_hip / invpref_mstep_rows_adam_hip form
 * them from (step, lr, betas, eps); table[i] belongs to step first_step + i. */
int invpref_adam_schedule_fill(float *host_table, int64_t first_step, int64_t n, double lr, double beta1, double beta2,
                               double eps) {
    if (!host_table || first_step < 1 || n < 0) return INVPREF_EINVAL;
    for (int64_t i = 0; i < n; i++) {
        const double step = (double)(first_step + i);
        const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
        float *r = host_table + 8 * i;
        r[0] = (float)(lr / bc1);
        r[1] = (float)sqrt(bc2);
        r[2] = (float)(1.0 - beta1);
        r[3] = (float)beta2;
        r[4] = (float)(1.0 - beta2);
        r[5] = (float)eps;
        r[6] = __builtin_nanf("");   // alpha: the call's coefficient block, unless the caller writes one here
        r[7] = 0.f;
    }
    return 0;
}

int invpref_mstep_rows_adam_sched_hip(const InvPrefTables *tables, const InvPrefTables *new_tables,
                                      const InvPrefTables *exp_avg, const InvPrefTables *exp_avg_sq,
                                      const InvPrefRowPlan *plan, const int64_t *envs, const float *scores,
                                      const float *sample_weights, int64_t batch_norm, const InvPrefCoefs *coefs,
                                      uint32_t flags, float *losses6, const InvPrefAdamSchedule *sched,
                                      void *workspace, size_t workspace_bytes, void *stream) {
    if (!sched) return INVPREF_EINVAL;
    return launch_rows(tables, plan, envs, scores, sample_weights, batch_norm, coefs, flags, losses6, workspace,
                       workspace_bytes, (hipStream_t)stream, 1, nullptr, new_tables, exp_avg, exp_avg_sq, AdamScalars{},
                       sched);
}

}  // extern "C"
