// invpref_step.hip -- the planned M-step as TWO balanced launches with no float atomics anywhere.
//
// The scatter pattern of a minibatch is static (utils.mini_batch, utils.py:12-19 yields the same contiguous,
// unshuffled slices every epoch), so it is inverted once on the host (plan.py) into row jobs:
//
//   job   = one row of the user tables (or of the item tables) + the minibatch's interactions that touch
//           it, cut into 1 .. NG equal slices, one slice per lane group;
//   group = the 16 / 32 / 64 lanes that hold one embedding row, ONE float4 per lane (D <= 64 / 128 / 256):
//           every instance of the kernels has the register profile of the D = 64 one;
//   round = the NG = 256 / lanes group slots of one workgroup, filled with jobs of one slice count;
//   task  = a few consecutive rounds of one side, run by one 256-thread workgroup.
//
// Launch 1, mstep_eval_kernel   = [user jobs | some of the untouched rows].  A user job keeps its two
//   rows and their gradients in registers, gathers the partner item rows ONCE, evaluates every one of its
//   interactions ONCE (forward + analytic backward of train.py:108-153, models.py:307-391) and stores a
//   RECORD {g_p, g_q, env, gz[0..E)} per interaction for the item side; it finishes its own rows on the
//   spot (Adam, or the gradient row for the multi-GPU path).  (Push form, InvPrefRowPlan::push_slot: instead of the
//   record it stores the interaction's two contribution rows to its item's gradient at the item-sorted slot, and
//   launch 2 sums contiguous rows.)  Everything that is a reduction ACROSS rows (gradients of embed_env / classifier,
//   the five loss sums) is accumulated per workgroup -- embed_env's by a read-modify-write of the LDS row the
//   environment names, the classifier's in LDS rows (D <= 64, E <= 4: 154 registers, 50 KB, THREE workgroups per
//   CU), in registers (E <= 8) or through LDS records (E = 16) -- and stored, plain stores, as that workgroup's
//   PARTIAL SLAB.
// Launch 2, mstep_apply_kernel  = [item jobs | the other untouched rows | fold blocks].  An item job
//   gathers the partner user rows + the records and only multiplies and adds (no exp / log / classifier,
//   few registers, several gathers in flight: long item slices are cheap, so there are no "hot rows").
//   The fold blocks sum the partial slabs in a fixed order, apply Adam to embed_env / classifier, write
//   the six loss outputs and move the device-side schedule on.
//
// Rows are fetched 4 x per interaction in total (2 partner rows per side) -- the algorithmic minimum --
// and every sum has a fixed order: the whole step is bitwise reproducible run to run.
// Parameters are double-buffered (read old, write new): launch 2 still gathers the OLD user rows.
#include <stdlib.h>
#include <string.h>

#include "kernel_common.hpp"

using namespace invpref;

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

// one row of the device-side schedule (include/invpref_hip.h: InvPrefAdamSchedule)
struct SchedRow {
    AdamScalars ad;
    float alpha;   // gradient-reversal alpha of the step; NaN: use the one of the call's coefficient block
    float pad;
};
__device__ __forceinline__ const SchedRow *sched_slot_ptr(const int *state, int slot) {
    return reinterpret_cast<const SchedRow *>(state + 16 * slot + 2);
}

// ---- lane-group helpers: a group is LG consecutive lanes of one wave, lane lg holds floats [4 lg, 4 lg + 4).
// Exchanges across the 16-lane rows of a wave use gfx950's v_permlane16_swap / v_permlane32_swap: vector-ALU
// instructions (no LDS pipe, no lgkmcnt wait).  With both operands holding x, the two results are
// {x.row0, x.row0, x.row2, x.row2} / {x.row1, x.row1, x.row3, x.row3} (16) and {x.rows01, x.rows01} / {x.rows23, x.rows23}
// (32): their sum (max) is the xor-16 / xor-32 butterfly step.  Inline assembly on purpose: through
// __builtin_amdgcn_permlane16_swap hipcc (ROCm 7.2) adds the FIRST result to itself (checked on the hardware,
// tools/scratch); the s_nop covers the vector-write -> permlane-read hazard the compiler would otherwise pad.
__device__ __forceinline__ void permlane16_swap(float &a, float &b) {
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void permlane32_swap(float &a, float &b) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float xor16_sum(float x) { float a = x, b = x; permlane16_swap(a, b); return a + b; }
__device__ __forceinline__ float xor32_sum(float x) { float a = x, b = x; permlane32_swap(a, b); return a + b; }
__device__ __forceinline__ float xor16_max(float x) { float a = x, b = x; permlane16_swap(a, b); return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float xor32_max(float x) { float a = x, b = x; permlane32_swap(a, b); return __builtin_fmaxf(a, b); }
template <int LG>
__device__ __forceinline__ float group_sum(float x) {
    x = row16_sum(x);
    if (LG >= 32) x = xor16_sum(x);
    if (LG >= 64) x = xor32_sum(x);
    return x;
}
template <int LG>
__device__ __forceinline__ float group_max(float x) {
    x = row16_max(x);
    if (LG >= 32) x = xor16_max(x);
    if (LG >= 64) x = xor32_max(x);
    return x;
}
__device__ __forceinline__ float wave_sum_valu(float x) { return xor32_sum(xor16_sum(row16_sum(x))); }
// Reduce-scatter butterfly over the low log2(N) lane bits: every lane holds N partial sums v[0..N) (one per class); on
// return lane l holds, as the function's value, the sum over its 2^log2(N) butterfly partners of class (l & (N - 1)).
// Step k keeps the classes whose bit k equals the lane's bit k and trades the others with lane l ^ (1 << k): N - 1
// exchanged values in all, against N * log2(N) for N separate butterflies.  (xor 1, 2: quad_perm; xor 4: row_shl:4 /
// row_shr:4 picked by the lane's bit 2; xor 8: row_ror:8.)
template <int M>
__device__ __forceinline__ float dpp_xor(float x, int lane) {
    if (M == 1) return dpp_move<0xB1>(x);
    if (M == 2) return dpp_move<0x4E>(x);
    if (M == 4) { const float up = dpp_move<0x104>(x), dn = dpp_move<0x114>(x); return (lane & 4) ? dn : up; }
    return dpp_move<0x128>(x);   // M == 8: rotate the row of 16 by 8
}
template <int N, int M = 1>
__device__ __forceinline__ float class_butterfly(const float (&v)[N], int lane) {
    if constexpr (N == 1) {
        return v[0];
    } else {
        float w[N / 2];
        const bool bit = lane & M;
#pragma unroll
        for (int i = 0; i < N / 2; i++) {
            const float keep = bit ? v[2 * i + 1] : v[2 * i], send = bit ? v[2 * i] : v[2 * i + 1];
            w[i] = keep + dpp_xor<M>(send, lane);
        }
        return class_butterfly<N / 2, M * 2>(w, lane);
    }
}
// the remaining lane bits of the group (classes are spread over the low log2(EMAX) bits only)
template <int LG, int EMAX>
__device__ __forceinline__ float group_sum_above(float x, int lane) {
    if (EMAX <= 8) x += dpp_xor<8>(x, lane);
    if (LG >= 32) x = xor16_sum(x);
    if (LG >= 64) x = xor32_sum(x);
    return x;
}
__device__ __forceinline__ float dot4(float4 a, float4 b) {
    float s = a.x * b.x;
    s = __builtin_fmaf(a.y, b.y, s);
    s = __builtin_fmaf(a.z, b.z, s);
    return __builtin_fmaf(a.w, b.w, s);
}
__device__ __forceinline__ void f4fma(float4 &acc, float s, float4 a) {
    acc.x = __builtin_fmaf(s, a.x, acc.x); acc.y = __builtin_fmaf(s, a.y, acc.y);
    acc.z = __builtin_fmaf(s, a.z, acc.z); acc.w = __builtin_fmaf(s, a.w, acc.w);
}
__device__ __forceinline__ void f4add(float4 &acc, float4 a) { acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
__device__ __forceinline__ float4 f4scale(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ float f4sq(float4 a) { return a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w; }
__device__ __forceinline__ float f4abs(float4 a) { return fabsf(a.x) + fabsf(a.y) + fabsf(a.z) + fabsf(a.w); }
// regulariser reports: x^2 + z^2 into a, y^2 + w^2 into b (two independent fma chains); sum of magnitudes as one add chain
// (the |x| is a source modifier of the add)
__device__ __forceinline__ void sq_acc(float &a, float &b, const float4 &v) {
    a = fmaf(v.x, v.x, a);
    b = fmaf(v.y, v.y, b);
    a = fmaf(v.z, v.z, a);
    b = fmaf(v.w, v.w, b);
}
__device__ __forceinline__ float abs_acc(float s, const float4 &v) {
    return (((s + fabsf(v.x)) + fabsf(v.y)) + fabsf(v.z)) + fabsf(v.w);
}
__device__ __forceinline__ float4 reg_term(float4 p, float r2, float r1) {   // r2 p + r1 sign(p)
    return make_float4(r2 * p.x + r1 * c_sign(p.x), r2 * p.y + r1 * c_sign(p.y), r2 * p.z + r1 * c_sign(p.z),
                       r2 * p.w + r1 * c_sign(p.w));
}

// the lane's float4 of row `row` (zero beyond D).  32-bit BYTE offsets (the launcher refuses tables of 2^32 bytes or
// more): a load takes the table's base from scalar registers and ONE vector register of offset instead of a 64-bit
// address pair.  Lanes beyond D read a clamped address and select zero: no branch around the load (a load under a
// branch is waited for at the join).
// FULL (rows of exactly 64 floats on 16 lanes, 16-byte aligned tables): the load and nothing else.  The clamp-and-select
// form below puts an instruction on the loaded value right behind the load; where the next loads sit behind a branch the
// compiler then waits for the data on the spot (found in round 4: the job's own rows, then each gather slot in turn, were
// separate round trips on the step's critical chain).
template <bool VEC, bool FULL = false>
__device__ __forceinline__ float4 row4(const float *__restrict__ base, int row, int D, int lg) {
    if (FULL) return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + ((unsigned)row * 256u + (unsigned)lg * 16u));
    const unsigned i0 = (unsigned)lg * 4u;
    if (VEC) {
        const bool ok = i0 < (unsigned)D;
        const unsigned boff = ((unsigned)row * (unsigned)D + (ok ? i0 : 0u)) * 4u;
        const float4 r = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + boff);
        return make_float4(ok ? r.x : 0.f, ok ? r.y : 0.f, ok ? r.z : 0.f, ok ? r.w : 0.f);
    }
    const float *p = base + (unsigned)row * (unsigned)D;
    float4 r;
    r.x = (i0 + 0 < (unsigned)D) ? p[i0 + 0] : 0.f;
    r.y = (i0 + 1 < (unsigned)D) ? p[i0 + 1] : 0.f;
    r.z = (i0 + 2 < (unsigned)D) ? p[i0 + 2] : 0.f;
    r.w = (i0 + 3 < (unsigned)D) ? p[i0 + 3] : 0.f;
    return r;
}
typedef float v4f __attribute__((ext_vector_type(4)));
// MODE 0: plain store; 1: write-through (sc1: the line leaves the XCD's L2 at once instead of at the end of the kernel)
template <bool VEC, int MODE = 0, bool FULL = false>
__device__ __forceinline__ void put4(float *__restrict__ base, int row, int D, int lg, float4 r) {
    if (FULL) {
        // (ONE 32-bit offset from the table's base: the base stays in scalar registers)
        float4 *dst = reinterpret_cast<float4 *>(reinterpret_cast<char *>(base) + ((unsigned)row * 256u + (unsigned)lg * 16u));
        if (MODE == 0) *dst = r;
        else {
            v4f val = {r.x, r.y, r.z, r.w};
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(val) : "memory");
        }
        return;
    }
    const int i0 = lg * 4;
    if (VEC) {
        const unsigned boff = ((unsigned)row * (unsigned)D + (unsigned)i0) * 4u;
        if (i0 < D) {
            float4 *dst = reinterpret_cast<float4 *>(reinterpret_cast<char *>(base) + boff);
            if (MODE == 0) *dst = r;
            else {
                v4f val = {r.x, r.y, r.z, r.w};
                // (the trailing s_nop: a store of more than 8 bytes followed by a vector write of its data registers is
                //  a hazard the compiler pads for its own stores, not for inline assembly)
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(val) : "memory");
            }
        }
    } else {
        float *p = base + (unsigned)row * (unsigned)D;
        if (i0 + 0 < D) p[i0 + 0] = r.x;
        if (i0 + 1 < D) p[i0 + 1] = r.y;
        if (i0 + 2 < D) p[i0 + 2] = r.z;
        if (i0 + 3 < D) p[i0 + 3] = r.w;
    }
}
// MODE 1: write-through (sc1): the line is on its way to the memory side during the kernel instead of in the end-of-kernel
// write-back of the XCD's L2 -- what a kernel boundary costs grows with the dirty bytes its predecessor leaves (MI355X guide,
// price list "boundary": + B / 6 TB/s)
template <int MODE>
__device__ __forceinline__ void store4(float *dst, float4 r) {
    if (MODE == 0) { *reinterpret_cast<float4 *>(dst) = r; return; }
    v4f val = {r.x, r.y, r.z, r.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(val) : "memory");
}

#ifndef STEP_PUSH_ST
#define STEP_PUSH_ST 0   // (A/B knob, two-launch forms: 1 = launch 1's contribution rows leave as write-through stores)
#endif
__device__ __forceinline__ void adam4(float4 &p, float4 g, float4 &m, float4 &v, const AdamScalars &ad) {
    adam1f(p.x, g.x, m.x, v.x, ad); adam1f(p.y, g.y, m.y, v.y, ad);
    adam1f(p.z, g.z, m.z, v.z, ad); adam1f(p.w, g.w, m.w, v.w, ad);
}

// launch arguments shared by the two kernels (each launch gets its own copy: its rounds, its share of the streamed rows)
struct StepArgs {
    const int4 *desc;             // [rounds][NG][2]: see InvPrefRowPlan in include/invpref_hip.h
    const int *round_iters;       // launch 1: [rounds] longest slice of the round (E > 4 instances: uniform loop count)
    const int4 *ulist;            // launch 1: [n] {item row, position, label bits, slot} sorted by user row
    const int2 *ilist;            // launch 2: [n] {user row, slot} sorted by item row
    const int64_t *envs;          // minibatch base pointers, indexed by position
    const float *weights;
    StepScalars k;
    uint32_t flags;
    int fused;                    // 0: store gradient rows to g; 1: Adam on the spot -> np / m / v
    float *np[4];                 // fused == 1: the new parameter tables (Pu, Qi, Pa, Qa); fused == 0: the GRADIENT tables
                                  // (one set of pointers for both forms: scalar registers are the scarce resource here)
    float *m[4], *v[4];           // Adam moments                      (fused == 1)
    AdamScalars ad;
    const int *stream_rows;       // untouched rows of this launch: row id, bit 30 set for item rows
    int rows_per_stream_task, rounds_per_task;
    int n_cls;                    // XCD-affine task order: workgroup b runs tasks of class b % n_cls (InvPrefRowPlan)
    int cls[8][4];                // per class: first round, rounds, first streamed row, streamed rows
    const int *push_slot;         // push form (InvPrefRowPlan::push_slot): [n] contribution slot of a minibatch position
    float *records;               // pull form: [n][4 + EMAX] records; push form: [n][2][DP] contribution rows
    float *slabs;                 // [launch-1 job tasks][slab_len] partial sums
    int n_rec;                    // wide rows: interactions of the minibatch = index of the spare record / contribution-row pair
    float *slabs_ev;              // wide rows (step_wide.hpp, EVL2): [launch-2 item tasks][EMAX][DP] partial sums of embed_env's gradient
    int *sched_state;             // optional device int32[32]: two slots {step, base, SchedRow}, see InvPrefAdamSchedule
    int sched_slot;
    // (16 unused bytes keep the kernel-argument offsets of rounds 4-5: with the block two pointers shorter hipcc's scalar-register
    //  allocation of mstep_eval_wide_kernel<16, 2, true, 8, false> -- the MovieLens instance, at 100 SGPRs -- spills nine of them
    //  and the dispatch then sets up a private segment; tests/test_kernel_isa.py holds that instance to a segment size of 0)
    const int *rec_slot;          // [n] position -> slot (index in item order): where launch 1 leaves the interaction's record /
                                  // contribution rows (InvPrefRowPlan::rec_slot; push form: the same array as push_slot)
    void *reserved_[1];
    int stamps_nodrain;
    unsigned long long *stamps;   // diagnostic (INVPREF_STAMPS): [workgroup][8] s_memrealtime ticks
};

// diagnostic phase stamp: drains the wave's outstanding memory operations first, so the latency of a phase is charged
// to that phase.  Never executed unless a stamp buffer is passed.
#define STAMP(i)                                                                  \
    do {                                                                          \
        if (a.stamps) {                                                           \
            if (!a.stamps_nodrain) __builtin_amdgcn_s_waitcnt(0);                 \
            if (threadIdx.x == 0) a.stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                         \
    } while (0)

// (which round of a task the per-round stamps 2 .. 5 describe: 0 = its first -- table staging included --, diagnostic builds
//  pass -DSTAMP_ROUND_OFFSET=k for a steady-state round)
#ifndef STAMP_ROUND_OFFSET
#define STAMP_ROUND_OFFSET 0
#endif
#define STAMP_ROUND (STAMP_ROUND_OFFSET < nr ? STAMP_ROUND_OFFSET : nr - 1)
#ifndef STEP_SLOT_ALIAS
#define STEP_SLOT_ALIAS 1
#endif
#ifndef STEP_CLS_QUADS
#define STEP_CLS_QUADS 0   // the four-class classifier of the 16-lane kernels with one class per QUAD of the group (see eval_interaction)
#endif
#ifndef STEP_LDS_DW
#define STEP_LDS_DW 1   // (A/B knob: smallest instance, classifier partial sums accumulated in LDS rows too)
#endif
#ifndef STEP_NO_DMA
#define STEP_NO_DMA 0   // (A/B knob: the smallest instance without the LDS-DMA landing area: 44 KB, three workgroups per CU)
#endif
#ifndef STEP_REG_EMAX
#define STEP_REG_EMAX 8   // (A/B knob: the largest class count whose E x D partial sums live in registers)
#endif
template <int LG, int EMAX>
struct Geo {
    static constexpr int NG = kThreads / LG;          // groups (rows in flight) per workgroup
    static constexpr int DP = 4 * LG;                 // padded row length
    static constexpr int RS = 4 + EMAX;               // floats per record: g_p, g_q, env bits, 0, gz[EMAX]
    static constexpr bool REG = EMAX <= STEP_REG_EMAX;   // E x D partial sums in registers (else: LDS records)
    static constexpr int SLAB = 2 * EMAX * DP + EMAX + kLossSlots;   // dEv | dW | db | loss sums
    // the workgroup's partial sums meet in LDS as RED rows of SLAB floats: one per GROUP for the smallest instance (plain
    // stores, no lane exchanges: the 72 permlane + add pairs of a per-wave pre-reduction sat on the step's critical
    // chain), one per WAVE for the larger ones (their slabs would not fit one per group)
    static constexpr bool DIRECT = LG == 16 && EMAX <= 4;
    static constexpr int RED = DIRECT ? NG : kWaves;
    // E > 4: thread -> column d_own, classes cg, cg + CG, ...
    static constexpr int CG = (kThreads / DP) < EMAX ? (kThreads / DP) : EMAX;
    static constexpr int CPT = (EMAX + CG - 1) / CG;
};

// LDS layout of launch 1 (floats)
template <int LG, int EMAX>
struct EvalLds {
    using G = Geo<LG, EMAX>;
    static constexpr int sEv = 0;                                   // [EMAX][DP]
    static constexpr int sW = sEv + EMAX * G::DP;                   // [EMAX][DP]
    static constexpr int sb = sW + EMAX * G::DP;                    // [EMAX]
    // slice partials [NG][2][DP]: an area of their own, or (ALIAS) inside the LDS-DMA landing area of the group's OWN wave
    // -- the leader takes the row's moments out of it first, same wave, program order -- which is what lets three
    // workgroups of the smallest instance share a CU's 160 KB
    static constexpr bool ALIAS = G::REG && STEP_SLOT_ALIAS;
    static constexpr int scw = sb + EMAX;                           // [EMAX] class weights (INVPREF_WEIGHTS_BY_ENV; else ones)
    static constexpr int slots = scw + EMAX;
    static constexpr int mv = slots + (ALIAS ? 0 : G::NG * 2 * G::DP);   // [4 waves][4][64] float4 LDS-DMA landing area
    static constexpr int red = mv + (STEP_NO_DMA && G::DIRECT ? 0 : kWaves * 4 * 64 * 4);            // REG: [4 waves][SLAB]; else [4 waves][8] loss sums
    static constexpr int rec = red + (G::REG ? G::RED * G::SLAB : kWaves * kLossSlots);   // E > 8: [2][NG][2][DP] x, o
    static constexpr int recs = rec + (G::REG ? 0 : 2 * G::NG * 2 * G::DP);              // E > 4: [2][NG][EMAX + 4] gz, env
    static constexpr int total = recs + (EMAX <= 4 ? 0 : 2 * G::NG * (EMAX + 4));
};

// ---- forward + analytic backward of ONE interaction on a lane group (M-step arithmetic: hardware exp/log/rcp).
template <int LG, int EMAX>
constexpr bool kClsQuads = LG == 16 && EMAX == 4 && STEP_CLS_QUADS;
template <int EMAX>
struct Eval {
    float g_p, g_q, li, le, lcls;
    float gz[EMAX <= 4 ? EMAX : 1];   // E <= 4: every lane holds all classes -- gz[c] of class c, or (STEP_CLS_QUADS, kClsQuads) gz[j] of
                                      // class (lg >> 2) ^ j: the lanes of quad 0 hold them in class order; lcls is then non-zero in
                                      // the quad of the interaction's environment only
    float gz_lane;                    // E > 4: lane lg of the group holds class lg (0 beyond E)
    float4 x, gx;                     // x = Pu*Qi ; gx = sum_c gz_c W_c
};
// KIND -1: implicit / PureMF decided at run time (workgroup-uniform branches); 0 / 1: explicit / implicit InvPref fixed at
// compile time -- the body is then ONE basic block, so that two calls in a row can be interleaved by the scheduler (the
// paired evaluation of csrc/step_alt.hpp)
template <int LG, int EMAX, int KIND = -1>
__device__ __forceinline__ void eval_interaction(Eval<EMAX> &o, float4 pu, float4 qi, float4 pa, float4 qa, float4 ev,
                                                 const float *sW, const float *sb, float *gzs, int E, int e, float y,
                                                 float cw_rec, float cw_cls, const StepScalars &k, bool implicit_rt,
                                                 bool pure_rt, int lg) {
    constexpr int DP = 4 * LG;
    const bool implicit = KIND < 0 ? implicit_rt : (KIND == 1);
    const bool pure = KIND < 0 ? pure_rt : false;
    o.x = f4mul(pu, qi);
    const float p = group_sum<LG>((o.x.x + o.x.y) + (o.x.z + o.x.w));
    const float q = group_sum<LG>(dot4(f4mul(pa, qa), ev));
    if (implicit) {
        const float sp = f_sigmoid(p), sq = f_sigmoid(q), sv = sp * sq;
        // (labels exactly 0 or 1 -- the implicit data, train.py:130-135 -- need one logarithm per loss; wave-uniform test)
        o.li = f_bce_binary(sp, y);
        o.le = f_bce_binary(sv, y);
#ifndef STEP_ASSUME_BINARY
#define STEP_ASSUME_BINARY 0   // (what-if knob, WRONG for labels other than 0 / 1)
#endif
        if (!STEP_ASSUME_BINARY && __builtin_amdgcn_ballot_w64(!(y == 0.0f || y == 1.0f)) != 0) {   // (never with the reference's implicit data)
            o.li = f_bce(sp, y);
            o.le = f_bce(sv, y);
        }
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
    o.gx = f4zero();
    o.lcls = 0.f;
    o.gz_lane = 0.f;
    if (EMAX <= 4) {
#pragma unroll
        for (int c = 0; c < (EMAX <= 4 ? EMAX : 1); c++) o.gz[c] = 0.f;
    }
    if (pure) return;   // PureMF: no classifier
    if constexpr (LG == 16 && EMAX == 4 && STEP_CLS_QUADS) {
        // One class per QUAD of the 16-lane group.  The four partial dot products are reduce-scattered over the quads -- step 1
        // with the quad q ^ 1 (row_half_mirror), step 2 with q ^ 2 (row_ror:8): three exchanged values instead of four full
        // butterflies -- and summed inside the quad; quad q then holds logit q in all four lanes.  Max and sum of the softmax
        // are two exchanges each over the same partners (every lane ends with the same bits: fp addition commutes and the
        // association (e_q + e_q^1) + (e_q^2 + e_q^3) is the same set of pairs in every quad); ONE exponential, reciprocal and
        // logarithm per lane instead of four + 1 + 1.  The class gradients come back by three moves: gz[j] = class q ^ j.
        const int qd = lg >> 2;
        float d[4];
#pragma unroll
        for (int c = 0; c < 4; c++) d[c] = dot4(o.x, *reinterpret_cast<const float4 *>(sW + c * DP + lg * 4));
        const bool odd = qd & 1, hi = qd & 2;
        float ka = odd ? d[1] : d[0], kb = odd ? d[3] : d[2];
        const float ga = odd ? d[0] : d[1], gb = odd ? d[2] : d[3];
        ka += dpp_move<0x141>(ga);
        kb += dpp_move<0x141>(gb);
        float zq = hi ? kb : ka;
        const float gq = hi ? ka : kb;
        zq += dpp_move<0x128>(gq);
        zq += dpp_move<0xB1>(zq);
        zq += dpp_move<0x4E>(zq);
        zq = qd < E ? zq + sb[qd] : -__builtin_inff();
        float mx = __builtin_fmaxf(zq, dpp_move<0x141>(zq));
        mx = __builtin_fmaxf(mx, dpp_move<0x128>(mx));
        const float dz = zq - mx;
        const float ez = f_exp(dz);   // exp(-inf) = 0
        float se = ez + dpp_move<0x141>(ez);
        se += dpp_move<0x128>(se);
        const float rse = f_rcp(se);
        // NLL of log_softmax as the reference forms it (see below); the quad of the interaction's environment reports it
        o.lcls = qd == e ? f_log(se) - dz : 0.f;
        const float g0 = qd < E ? k.cc * cw_cls * (ez * rse - (qd == e ? 1.f : 0.f)) : 0.f;
        const float g1 = dpp_move<0x141>(g0), g2 = dpp_move<0x128>(g0), g3 = dpp_move<0x128>(g1);
        o.gz[0] = g0; o.gz[1] = g1; o.gz[2] = g2; o.gz[3] = g3;
#pragma unroll
        for (int j = 0; j < 4; j++)
            f4fma(o.gx, o.gz[j], *reinterpret_cast<const float4 *>(sW + ((qd ^ j) * DP + lg * 4)));
        return;
    }
    if (EMAX <= 4) {
        // the W rows are read from LDS ONCE, unconditionally and back to back (rows c >= E are staged as zeros);
        // everything after is selects and arithmetic
        float z[EMAX <= 4 ? EMAX : 1], mx = -__builtin_inff();
#pragma unroll
        for (int c = 0; c < (EMAX <= 4 ? EMAX : 1); c++) {
            const float4 wrow = *reinterpret_cast<const float4 *>(sW + c * DP + lg * 4);
            const float zc = group_sum<LG>(dot4(o.x, wrow)) + sb[c];
            z[c] = c < E ? zc : -__builtin_inff();
            mx = z[c] > mx ? z[c] : mx;
        }
        // NLL of log_softmax as the reference forms it (models.py:206-209, train.py:42-44): log(sum exp(z - max)) - (z_e - max).
        // Finite for any finite logits -- log(exp(z_e - max) / sum) is +inf once that exponential underflows, which embeddings
        // grown to a few units reach (seen after 2 000 steps of the Yahoo-shaped run: tools/alt_soak.py)
        float se = 0.f, zr = 0.f;
#pragma unroll
        for (int c = 0; c < (EMAX <= 4 ? EMAX : 1); c++) {
            const float dz = z[c] - mx;
            zr = (c == e) ? dz : zr;
            z[c] = f_exp(dz);   // exp(-inf) = 0
            se += z[c];
        }
        const float rse = f_rcp(se);
        o.lcls = f_log(se) - zr;
#pragma unroll
        for (int c = 0; c < (EMAX <= 4 ? EMAX : 1); c++) {
            o.gz[c] = c < E ? k.cc * cw_cls * (z[c] * rse - (c == e ? 1.f : 0.f)) : 0.f;
            // (read again rather than held across the softmax: 16 registers at the kernel's pressure peak)
            f4fma(o.gx, o.gz[c], *reinterpret_cast<const float4 *>(sW + c * DP + lg * 4));
        }
    }
    // (more than four classes: one class per lane, eval_wide in step_wide.hpp)
}

// stage a small [E][D] table into LDS as [EMAX][DP], zero padded (an absent table, INVPREF_PURE_MF, stages zeros)
__device__ __forceinline__ void stage_small(float *dst, const float *__restrict__ src, int E, int D, int EMAX, int DP) {
    for (int i = threadIdx.x; i < EMAX * DP; i += kThreads) {
        const int e = i / DP, d = i - e * DP;
        dst[i] = (src && e < E && d < D) ? src[e * D + d] : 0.f;
    }
}

// A task's FIRST descriptors through the scalar cache (A/B knob STEP_SCALAR_DESC): the wave's group slots are adjacent,
// so their descriptors are one or two scalar loads; each lane then picks its group's words.  The scalar path is not
// queued behind the launch's vector load burst.  (Plans are written once, before any launch reads them: the scalar
// cache's lack of coherence with vector stores does not matter here.)
#ifndef STEP_SCALAR_DESC
#define STEP_SCALAR_DESC 0
#endif
template <int LG>
__device__ __forceinline__ void first_desc(const int4 *desc, int r0, int grp, int4 &d, int4 &d1) {
    constexpr int NG = kThreads / LG, GW = 64 / LG;   // groups per wave
    if (!STEP_SCALAR_DESC) {
        d = desc[(r0 * NG + grp) * 2];
        d1 = desc[(r0 * NG + grp) * 2 + 1];
        return;
    }
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) v4i *cptr;
    const int wave = threadIdx.x >> 6, gw = (threadIdx.x & 63) / LG;
    const int base = __builtin_amdgcn_readfirstlane((r0 * NG + wave * GW) * 2);
    cptr sd = (cptr)(uintptr_t)desc;
    v4i s[2 * GW];
#pragma unroll
    for (int i = 0; i < 2 * GW; i++) s[i] = sd[base + i];
    d = make_int4(s[0].x, s[0].y, s[0].z, s[0].w); d1 = make_int4(s[1].x, s[1].y, s[1].z, s[1].w);
#pragma unroll
    for (int g = 1; g < GW; g++) {
        const bool me = gw == g;
        d.x = me ? s[2 * g].x : d.x; d.y = me ? s[2 * g].y : d.y; d.z = me ? s[2 * g].z : d.z; d.w = me ? s[2 * g].w : d.w;
        d1.x = me ? s[2 * g + 1].x : d1.x; d1.y = me ? s[2 * g + 1].y : d1.y;
        d1.z = me ? s[2 * g + 1].z : d1.z; d1.w = me ? s[2 * g + 1].w : d1.w;
    }
}

struct USample {
    int oth, ps;
    float y;
};

// =====================================================================================
// launch 1: rounds of USER jobs.  Per interaction: gather the item rows, evaluate, accumulate the user rows'
// gradients in registers, store the record for the item side, accumulate the E x D / loss sums.
// =====================================================================================
#ifndef STEP_STAGE_LATE
#define STEP_STAGE_LATE 1   // (A/B knob, FULL instances: the small tables staged behind the first gathers)
#endif
#ifndef STEP_EVAL_DEPTH
#define STEP_EVAL_DEPTH 3
#endif
#ifndef STEP_SLOT_FROM_LIST
#define STEP_SLOT_FROM_LIST 1   // launch 1 takes an interaction's slot from its list entry (0: always from rec_slot[position], a random read)
#endif
#ifndef STEP_ROW_ST
#define STEP_ROW_ST 1   // write-through stores for the rows the two-launch jobs finish (p', m', v': nothing of them is read again before
                        // the next step; left dirty in L2 they lengthen the kernel boundary).  Round 6, same box: 2^24 interactions at
                        // D = 64 4.95 -> 4.59 ms (0.487 -> 0.524), Yahoo B = N 92.2 -> 90.7 us, MovieLens- / MIND-shaped level (0: plain)
#endif
template <int LG, bool VEC, int EMAX, bool FULL>
__device__ __forceinline__ void user_task(const DevTables &t, const StepArgs &a, int r0, int nr, int slab_index, float *lds) {
    using G = Geo<LG, EMAX>;
    // (rows of more than 64 floats and more than four environments run step_wide.hpp; the branches of this function for
    //  other layouts are compile-time dead)
    static_assert(LG == 16 && EMAX == 4, "user_task: the 16-lane, four-environment instances only");
    static_assert(!kClsQuads<LG, EMAX> || (G::DIRECT && STEP_LDS_DW), "class-per-quad gradients need the classifier's LDS rows");
    // interactions in flight per group (measured: 2 for the D <= 64, E <= 4 instances -- a third slot only costs registers
    // there -- and for E > 8, whose per-interaction barrier paces the groups anyway; 3 for the other larger rows)
    constexpr int UE = !G::REG ? 1 : ((LG == 16 && EMAX <= 4 && STEP_EVAL_DEPTH > 2) ? 2 : STEP_EVAL_DEPTH);
    using L = EvalLds<LG, EMAX>;
    constexpr int NG = G::NG, DP = G::DP, RS = G::RS;
    float *sEv = lds + L::sEv, *sW = lds + L::sW, *sb = lds + L::sb, *scw = lds + L::scw, *slots = lds + L::slots;
    float4 *mv = reinterpret_cast<float4 *>(lds + L::mv);
    float *red = lds + L::red;
    const int lg = threadIdx.x & (LG - 1), grp = threadIdx.x / LG, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4 *mv_wave = mv + wave * 4 * 64;
    const bool implicit = a.flags & INVPREF_IMPLICIT;
    const bool rw_rec = a.flags & INVPREF_REWEIGHT_REC, rw_cls = a.flags & INVPREF_REWEIGHT_CLS;
    const bool by_env = a.flags & INVPREF_WEIGHTS_BY_ENV;   // weight = class_weights[env], staged with the small tables
    const bool reg_env = a.flags & INVPREF_REG_ENV_EMBED;
    const bool pure = a.flags & INVPREF_PURE_MF;   // env-aware tables, embed_env, classifier absent: never touched
    // (E > 8: the LDS-DMA landing area holds the embed_env partial sums instead; the moments are loaded late)
    const bool dma = VEC && a.fused && G::REG && !(STEP_NO_DMA && G::DIRECT);
    const bool push = a.push_slot != nullptr;
    float *sdE = lds + L::mv;   // E > 8: [EMAX][DP] embed_env partial sums of the workgroup, rows indexed by the environment
    StepScalars k = a.k;
    if (a.sched_state) {  // scheduled alpha (train.py:214-217) under graph replay
        const float al = sched_slot_ptr(a.sched_state, a.sched_slot)->alpha;
        if (al == al) k.alpha = al;
    }
    const AdamScalars ad = a.sched_state ? sched_slot_ptr(a.sched_state, a.sched_slot)->ad : a.ad;

    STAMP(0);
    // the first round's descriptor goes out before anything else: every gather below hangs on it
    int4 d, d1;
    first_desc<LG>(a.desc, r0, grp, d, d1);
    if (!STEP_STAGE_LATE || !FULL) {
        stage_small(sEv, t.Ev, t.E, t.D, EMAX, DP);
        stage_small(sW, t.W, t.E, t.D, EMAX, DP);
        if (threadIdx.x < EMAX) sb[threadIdx.x] = (threadIdx.x < t.E && t.b) ? t.b[threadIdx.x] : 0.f;
        if (threadIdx.x < EMAX)   // (a 32-bit offset from the scalar base: an address pair held across the rounds went to scratch memory)
                    scw[threadIdx.x] = (by_env && threadIdx.x < t.E) ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.weights) + (threadIdx.x & (EMAX - 1)) * 4u) : 1.f;
    }
    if (!G::REG)
        for (int i = threadIdx.x; i < EMAX * DP; i += kThreads) sdE[i] = 0.f;
    else {
        // embed_env's partial sums -- in the smallest instance the classifier's too (STEP_LDS_DW) -- are accumulated IN
        // the rows the workgroup's partial sums meet in (below)
        constexpr int ZR = (G::DIRECT && STEP_LDS_DW) ? 2 * EMAX * DP + EMAX : EMAX * DP;
        for (int i = threadIdx.x; i < G::RED * ZR; i += kThreads) red[(i / ZR) * G::SLAB + i % ZR] = 0.f;
    }
    STAMP(1);

    // E x D partial sums: registers (E <= 4: per group, all classes) or thread-owned outputs fed from LDS records
    float4 dW[G::REG ? EMAX : 1];
    float dB[G::REG ? EMAX : 1];
    float oW[G::REG ? 1 : G::CPT], oB1 = 0.f;
#pragma unroll
    for (int c = 0; c < (G::REG ? EMAX : 1); c++) { dW[c] = f4zero(); dB[c] = 0.f; }
#pragma unroll
    for (int i = 0; i < (G::REG ? 1 : G::CPT); i++) oW[i] = 0.f;
    const int d_own = threadIdx.x % DP, cg = threadIdx.x / DP;
    float accLi = 0.f, accLe = 0.f, accLc = 0.f, accL2 = 0.f, accL1 = 0.f;
    int it_total = 0;   // E > 4: parity of the gz scratch / record buffers
    float *slab = a.slabs + (int64_t)slab_index * G::SLAB;   // this workgroup's partial slab

    for (int r = r0; r < r0 + nr; r++) {
        const int4 dd = d, dd1 = d1;   // (this round's copy: what the lambdas below read)
        const int row = dd.x, meta = dd.y;
        const bool active = row >= 0, leader = meta & 1;
        const int slices = (meta >> 1) & 31, mode = (meta >> 6) & 7;
        const int nsmp = active ? (mode == 7 ? dd.w - dd.z : mode) : 0;
        const int iters = G::REG ? nsmp : a.round_iters[r];
        if (r == r0 + STAMP_ROUND) STAMP(2);
        // sample sidx of the slice: inline in the descriptor (up to two) or one 16-byte load from the sorted list.
        // FULL instances fetch it with ONE unconditional load from a selected address (list index clamped into the slice;
        // the inline form read back from the descriptor's own words, 3 ints per interaction from word 2 on): every gather of
        // the task is then issued without a branch around it and without an instruction on loaded data behind it -- a load
        // under a divergent branch is waited for at the join, which made each gather slot a round trip of its own.
        const int s_lo = dd.z, s_hi1 = max(dd.w - 1, dd.z);
        const int *dwords = reinterpret_cast<const int *>(a.desc + (r * NG + grp) * 2);
        auto sample_at = [&](int sidx) {
            USample sm;
            if (FULL) {   // (the refills of the interaction loop: list form; inline slices never get there)
                const int *src = mode == 7 ? reinterpret_cast<const int *>(a.ulist + min(s_lo + sidx, s_hi1)) : dwords + 2 + 3 * min(sidx, 1);
                sm.oth = src[0]; sm.ps = src[1]; sm.y = __builtin_bit_cast(float, src[2]);
            } else if (mode == 7) {
                const int4 q = a.ulist[dd.z + sidx];
                sm.oth = q.x; sm.ps = q.y; sm.y = __builtin_bit_cast(float, q.z);
            } else if (sidx == 0) { sm.oth = dd.z; sm.ps = dd.w; sm.y = __builtin_bit_cast(float, dd1.x); }
            else { sm.oth = dd1.y; sm.ps = dd1.z; sm.y = __builtin_bit_cast(float, dd1.w); }
            return sm;
        };
        // everything that depends only on the descriptor is requested together: own rows, the Adam moments of the
        // row (needed last: LDS-DMA, no registers held across the loop) and the first interactions' rows / env / weight
        float4 oi = f4zero(), oe = f4zero(), gi = f4zero(), ge = f4zero();
        {   // (an idle slot reads row 0 rather than branching around the loads)
            const int rowc = active ? row : 0;
            oi = row4<VEC, FULL>(t.Pu, rowc, t.D, lg);
            if (!pure) oe = row4<VEC, FULL>(t.Pa, rowc, t.D, lg);
        }
        // UE interactions in flight per group, each in a register slot of its own: the slot just consumed is refilled
        // at once (no register copies: a copy of a register that is still being loaded would wait for the load)
        struct Slot {
            float4 qi, qa;
            USample sm;
            int e, cs;
            float w;
        };
        Slot sl[UE];
        USample idn[UE];   // (list form: the ids run UE interactions ahead of the rows)
        auto gather = [&](Slot &q, const USample &sm, int sidx) {
            q.sm = sm;
            q.qi = row4<VEC, FULL>(t.Qi, sm.oth, t.D, lg);
            if (FULL) {   // (32-bit offsets: one address register each, nothing to copy a loaded index into)
                const unsigned pso = (unsigned)sm.ps;
                if (!pure) {
                    q.qa = row4<VEC, FULL>(t.Qa, sm.oth, t.D, lg);
                    q.e = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(a.envs) + pso * 8u);   // low word of the int64 id
                }
                if ((rw_rec || rw_cls) && !by_env) q.w = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.weights) + pso * 4u);
                // the slot: word 3 of the list entry the ids came from (a line that was just read) or, inline form, rec_slot[position]
                // -- one load from a selected address again
                const int *sp = mode == 7 ? reinterpret_cast<const int *>(a.ulist + min(s_lo + sidx, s_hi1)) + 3
                                          : reinterpret_cast<const int *>(reinterpret_cast<const char *>(a.rec_slot) + pso * 4u);
                q.cs = STEP_SLOT_FROM_LIST ? *sp : *reinterpret_cast<const int *>(reinterpret_cast<const char *>(a.rec_slot) + pso * 4u);
                return;
            }
            if (!pure) {
                q.qa = row4<VEC, FULL>(t.Qa, sm.oth, t.D, lg);
                q.e = (int)a.envs[sm.ps];
            }
            if ((rw_rec || rw_cls) && !by_env) q.w = a.weights[sm.ps];
            q.cs = a.rec_slot[sm.ps];
        };
#pragma unroll
        for (int j = 0; j < UE; j++) {
            sl[j].qi = sl[j].qa = f4zero();
            sl[j].sm = USample{0, 0, 0.f};
            sl[j].e = sl[j].cs = 0;
            sl[j].w = 1.f;
            idn[j] = USample{0, 0, 0.f};
        }
        if (FULL) {
            // the first interactions' ids: inline ones straight from the descriptor's registers; listed ones (only if some
            // group of the WAVE has a list: a uniform branch) with every list entry requested BEFORE the first row load
            // hangs on one -- ids, then rows: two round trips, not one per slot
            USample ls[2 * UE];
#pragma unroll
            for (int j = 0; j < 2 * UE; j++) ls[j] = USample{0, 0, 0.f};
            if (__builtin_amdgcn_ballot_w64(mode == 7) != 0) {
#pragma unroll
                for (int j = 0; j < 2 * UE; j++) {
                    // (groups WITHOUT a list ride along in this wave-uniform branch: their words 2 / 3 are an inline
                    //  interaction's ids, not a list range -- entry 0 for them, or the load runs off a small plan's list)
                    const int4 q = a.ulist[mode == 7 ? min(s_lo + j, s_hi1) : 0];
                    ls[j] = USample{q.x, q.y, __builtin_bit_cast(float, q.z)};
                }
            }
#pragma unroll
            for (int j = 0; j < UE; j++) {
                USample sm = ls[j];
                if (mode != 7) sm = j == 0 ? USample{dd.z, dd.w, __builtin_bit_cast(float, dd1.x)} : USample{dd1.y, dd1.z, __builtin_bit_cast(float, dd1.w)};
                gather(sl[j], sm, j);
                idn[j] = ls[UE + j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < UE; j++)
                if (j < nsmp) gather(sl[j], sample_at(j), j);
#pragma unroll
            for (int j = 0; j < UE; j++)
                if (UE + j < nsmp) idn[j] = sample_at(UE + j);
        }
#ifndef STEP_DMA_LATE
        if (dma) {
            // each lane sends its 16-byte piece of the row's four moment rows straight to LDS; the destination of a
            // wave instruction is one contiguous 1 KiB block, lane-major
            const bool mine = active && leader && lg * 4 < t.D;
#pragma unroll
            for (int tn = 0; tn < 4; tn++) {
                const float *src_tab = (tn & 1) ? a.v[(tn >> 1) * 2] : a.m[(tn >> 1) * 2];
                if (mine && !(pure && tn >= 2))
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(src_tab + ((unsigned)row * (unsigned)t.D + (unsigned)lg * 4u)),
                        (__attribute__((address_space(3))) void *)(mv_wave + tn * 64), 16, 0, 0);
            }
        }
#endif
        if (r == r0) {
            if (STEP_STAGE_LATE && FULL) {
                // the two small tables are staged HERE, behind the first round's gathers: their load -> LDS-store loop in
                // front of the descriptor's use put a round trip of its own ahead of the gathers
                stage_small(sEv, t.Ev, t.E, t.D, EMAX, DP);
                stage_small(sW, t.W, t.E, t.D, EMAX, DP);
                if (threadIdx.x < EMAX) sb[threadIdx.x] = (threadIdx.x < t.E && t.b) ? t.b[threadIdx.x] : 0.f;
                if (threadIdx.x < EMAX)   // (a 32-bit offset from the scalar base: an address pair held across the rounds went to scratch memory)
                    scw[threadIdx.x] = (by_env && threadIdx.x < t.E) ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.weights) + (threadIdx.x & (EMAX - 1)) * 4u) : 1.f;
            }
            __syncthreads();   // staged tables visible (the gathers above are in flight)
            STAMP(3);
        }
        // one interaction: evaluate, accumulate the user rows' gradients, store the record, feed the E x D / loss sums
        auto step = [&](const Slot &q, bool has) {
            float *gzs = nullptr;
            if (EMAX > 4) gzs = lds + L::recs + ((it_total & 1) * NG + grp) * (EMAX + 4);
            const int e = q.e;
            if (has) {
                const float wq = by_env ? scw[e] : q.w;
                const float w_rec = rw_rec ? wq : 1.f, w_cls = rw_cls ? wq : 1.f;
                const float4 ev = *reinterpret_cast<const float4 *>(sEv + e * DP + lg * 4);
                Eval<EMAX> o;
                eval_interaction<LG, EMAX>(o, oi, q.qi, oe, q.qa, ev, sW, sb, gzs, t.E, e, q.sm.y, w_rec * k.invB,
                                           w_cls * k.invB, k, implicit, pure, lg);
                float4 gip;
                gip.x = o.g_p - k.alpha * o.gx.x; gip.y = o.g_p - k.alpha * o.gx.y;
                gip.z = o.g_p - k.alpha * o.gx.z; gip.w = o.g_p - k.alpha * o.gx.w;
                f4add(gi, f4mul(gip, q.qi));
                f4fma(ge, o.g_q, f4mul(q.qa, ev));
                if (push) {
                    // push form: the interaction's two contribution rows to its ITEM's gradient, stored at the item-sorted
                    // slot -- launch 2 then sums contiguous rows, no gathers, no classifier
                    float *cr = a.records + (unsigned)q.cs * (unsigned)(2 * DP) + lg * 4;
                    store4<STEP_PUSH_ST>(cr, f4mul(gip, oi));
                    store4<STEP_PUSH_ST>(cr + DP, f4scale(o.g_q, f4mul(oe, ev)));
                } else {
                    // pull form: the record the item side consumes, at the interaction's slot in the item order
                    float *rec_g = a.records + (unsigned)q.cs * (unsigned)RS;
                    if (lg == 0)
                        *reinterpret_cast<float4 *>(rec_g) = make_float4(o.g_p, o.g_q, __builtin_bit_cast(float, e), 0.f);
                    if (EMAX <= 4) {
                        if (lg == 1) *reinterpret_cast<float4 *>(rec_g + 4) =
                            make_float4(o.gz[0], EMAX > 1 ? o.gz[EMAX > 1 ? 1 : 0] : 0.f, EMAX > 2 ? o.gz[EMAX > 2 ? 2 : 0] : 0.f,
                                        EMAX > 3 ? o.gz[EMAX > 3 ? 3 : 0] : 0.f);
                    } else if (lg < EMAX) {
                        rec_g[4 + lg] = o.gz_lane;
                    }
                }
                // o = g_q Pa*Qa (+ env regulariser): the interaction's term of embed_env's gradient
                float4 oo = f4scale(o.g_q, f4mul(oe, q.qa));
                if (reg_env) f4add(oo, reg_term(ev, 2.f * k.r2, 2.f * k.r1));
                if (G::REG) {
                    float gzv[G::REG ? EMAX : 1];   // every class in every lane
                    if (EMAX <= 4) {
#pragma unroll
                        for (int c = 0; c < (G::REG ? EMAX : 1); c++) gzv[c] = o.gz[EMAX <= 4 ? c : 0];
                    } else {
#pragma unroll
                        for (int c4 = 0; c4 < (G::REG ? EMAX : 4); c4 += 4) {
                            const float4 g4 = *reinterpret_cast<const float4 *>(gzs + c4);   // (written by the evaluation)
                            gzv[G::REG ? c4 : 0] = g4.x; gzv[G::REG ? c4 + 1 : 0] = g4.y;
                            gzv[G::REG ? c4 + 2 : 0] = g4.z; gzv[G::REG ? c4 + 3 : 0] = g4.w;
                        }
                    }
#ifndef DBG_NO_EXD
                    if (G::DIRECT && STEP_LDS_DW) {
                        // the classifier's partial sums as well: a read-modify-write of the group's own rows
                        float *mine = red + grp * G::SLAB;
#pragma unroll
                        for (int c = 0; c < (G::REG ? EMAX : 1); c++) {
                            // (kClsQuads: gzv[c] belongs to class (lg >> 2) ^ c -- every lane still visits all four rows)
                            float4 *wr = reinterpret_cast<float4 *>(mine + EMAX * DP + (kClsQuads<LG, EMAX> ? ((lg >> 2) ^ c) : c) * DP + lg * 4);
                            float4 cur = *wr;
                            f4fma(cur, gzv[c], o.x);
                            *wr = cur;
                        }
                        if (lg == 0 && EMAX == 4) {
                            float4 *br = reinterpret_cast<float4 *>(mine + 2 * EMAX * DP);
                            float4 cur = *br;
                            cur.x += gzv[0]; cur.y += gzv[EMAX > 1 ? 1 : 0]; cur.z += gzv[EMAX > 2 ? 2 : 0]; cur.w += gzv[EMAX > 3 ? 3 : 0];
                            *br = cur;
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < (G::REG ? EMAX : 1); c++) {
                            f4fma(dW[c], gzv[c], o.x);
                            dB[c] += gzv[c];
                        }
                    }
                    // embed_env: ONE read-modify-write of the LDS row the interaction's environment names -- a row set per
                    // group (smallest instance) or per wave, whose groups then take turns (in-order LDS operations of one
                    // wave: fixed order, no barrier) -- instead of a select + add per class on registers
                    float4 *erow = reinterpret_cast<float4 *>(red + grp * G::SLAB + e * DP + lg * 4);
                    float4 cur = *erow;
                    f4add(cur, oo);
                    *erow = cur;
#endif
                }
                // regulariser REPORTS over the item rows of the interaction (env rows weigh double: 1/(BD) vs 1/(2BD))
                // (two fma chains for the squares, one |x| add chain for the magnitudes: 16 instructions instead of 28)
                float s2a = 0.f, s2b = 0.f;
                sq_acc(s2a, s2b, q.qi);
                sq_acc(s2a, s2b, q.qa);
                float s2 = s2a + s2b, s1 = abs_acc(abs_acc(0.f, q.qi), q.qa);
                if (reg_env) { s2 += 2.f * f4sq(ev); s1 += 2.f * f4abs(ev); }
                accL2 += s2;
                accL1 += s1;
                if (lg == 0) { accLi += o.li * w_rec; accLe += o.le * w_rec; }
                if (kClsQuads<LG, EMAX> ? (lg & 3) == 0 : lg == 0) accLc += o.lcls * w_cls;
            }
            if (EMAX > 4) it_total++;
        };
#ifdef STEP_DMA_LATE
        // (A/B knob: the moments are requested behind the first gathers' burst instead of inside it)
        __builtin_amdgcn_sched_barrier(0);
        if (dma) {
            // each lane sends its 16-byte piece of the row's four moment rows straight to LDS; the destination of a
            // wave instruction is one contiguous 1 KiB block, lane-major
            const bool mine = active && leader && lg * 4 < t.D;
#pragma unroll
            for (int tn = 0; tn < 4; tn++) {
                const float *src_tab = (tn & 1) ? a.v[(tn >> 1) * 2] : a.m[(tn >> 1) * 2];
                if (mine && !(pure && tn >= 2))
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(src_tab + ((unsigned)row * (unsigned)t.D + (unsigned)lg * 4u)),
                        (__attribute__((address_space(3))) void *)(mv_wave + tn * 64), 16, 0, 0);
            }
        }
#endif
        for (int s = 0; s < iters; s += UE) {
#pragma unroll
            for (int j = 0; j < UE; j++) {
                if (s + j < iters) step(sl[j], s + j < nsmp);   // (E > 8: `iters` is uniform, the barrier inside is too)
                if (FULL || s + UE + j < nsmp) gather(sl[j], idn[j], s + UE + j);
                if (FULL || s + 2 * UE + j < nsmp) idn[j] = sample_at(s + 2 * UE + j);
            }
        }
        if (r == r0 + STAMP_ROUND) STAMP(4);
        // (requested here, behind the interaction loop's register peak: it flies under the slice meet and the row finish)
        if (r + 1 < r0 + nr) { d = a.desc[((r + 1) * NG + grp) * 2]; d1 = a.desc[((r + 1) * NG + grp) * 2 + 1]; }
        const float cnt = (float)(meta >> 9);
        if (active && leader) {   // regulariser reports: the user's rows count once per interaction
            accL2 += cnt * (f4sq(oi) + f4sq(oe));
            accL1 += cnt * (f4abs(oi) + f4abs(oe));
        }
        // The task's LAST round also hands over the workgroup's partial sums (dEv | dW | db | loss sums): they go to LDS
        // in front of the barrier the slices meet at and leave -- plain stores, fixed-order sums -- right behind it, so
        // that the slab's stores fly under the leaders' Adam instead of forming a phase of their own.
        const bool last = r == r0 + nr - 1;
        if (last) {
            if (G::REG && G::DIRECT) {
                // one row per group: every lane stores its own pieces, nothing is exchanged
                accLi = row16_sum(accLi); accLe = row16_sum(accLe); accLc = row16_sum(accLc);
                accL2 = row16_sum(accL2); accL1 = row16_sum(accL1);
                float *mine = red + grp * G::SLAB;
#pragma unroll
                for (int c = 0; c < ((G::REG && !STEP_LDS_DW) ? EMAX : 0); c++) {   // (the embed_env rows are there already)
                    *reinterpret_cast<float4 *>(mine + EMAX * DP + c * DP + lg * 4) = dW[G::REG ? c : 0];
                    if (lg == 0) mine[2 * EMAX * DP + c] = dB[G::REG ? c : 0];
                }
                if (lg == 0) {
                    float *ls = mine + 2 * EMAX * DP + EMAX;
                    ls[0] = accLi; ls[1] = accLe; ls[2] = accLc; ls[3] = accL2; ls[4] = accL1; ls[5] = ls[6] = ls[7] = 0.f;
                }
            } else if (G::REG) {
                accLi = wave_sum_valu(accLi); accLe = wave_sum_valu(accLe); accLc = wave_sum_valu(accLc);
                accL2 = wave_sum_valu(accL2); accL1 = wave_sum_valu(accL1);

                // groups of one wave first (lane exchanges), then the four waves through LDS
                float *mine = red + wave * G::SLAB;
#pragma unroll
                for (int c = 0; c < (G::REG ? EMAX : 1); c++) {
                    float4 &w4 = dW[c];
                    if (LG == 16) {
                        w4.x = xor16_sum(w4.x); w4.y = xor16_sum(w4.y); w4.z = xor16_sum(w4.z); w4.w = xor16_sum(w4.w);
                        dB[c] = xor16_sum(dB[c]);
                    }
                    if (LG <= 32) {
                        w4.x = xor32_sum(w4.x); w4.y = xor32_sum(w4.y); w4.z = xor32_sum(w4.z); w4.w = xor32_sum(w4.w);
                        dB[c] = xor32_sum(dB[c]);
                    }
                    if (lane < LG) *reinterpret_cast<float4 *>(mine + EMAX * DP + c * DP + lane * 4) = w4;
                    if (lane == 0) mine[2 * EMAX * DP + c] = dB[c];
                }
                if (lane == 0) {
                    float *ls = mine + 2 * EMAX * DP + EMAX;
                    ls[0] = accLi; ls[1] = accLe; ls[2] = accLc; ls[3] = accL2; ls[4] = accL1; ls[5] = ls[6] = ls[7] = 0.f;
                }
            } else {
                accLi = wave_sum_valu(accLi); accLe = wave_sum_valu(accLe); accLc = wave_sum_valu(accLc);
                accL2 = wave_sum_valu(accL2); accL1 = wave_sum_valu(accL1);
                if (lane == 0) {
                    float *ls = red + wave * kLossSlots;
                    ls[0] = accLi; ls[1] = accLe; ls[2] = accLc; ls[3] = accL2; ls[4] = accL1; ls[5] = ls[6] = ls[7] = 0.f;
                }
                if (threadIdx.x < EMAX) slab[2 * EMAX * DP + threadIdx.x] = oB1;
                if (cg == 0 && d_own < DP) {        // (this thread's own column of the LDS rows: its own program order)
#pragma unroll 4
                    for (int c = 0; c < EMAX; c++) slab[c * DP + d_own] = sdE[c * DP + d_own];
                }
                if (threadIdx.x < G::CG * DP) {   // thread-owned outputs: straight to the slab
#pragma unroll
                    for (int i = 0; i < (G::REG ? 1 : G::CPT); i++) {
                        const int c = cg + G::CG * i;
                        if (c < EMAX) slab[EMAX * DP + c * DP + d_own] = oW[i];
                    }
                }
            }
        }
        // ---- slices of one row meet through LDS: plain stores, fixed-order sum by the leader
        auto slot_of = [&](int g) {
            constexpr int GW = 64 / LG;   // groups per wave
            return L::ALIAS ? lds + L::mv + (g / GW) * (4 * 64 * 4) + (g % GW) * 2 * DP : slots + g * 2 * DP;
        };
        float4 mi = f4zero(), vi = f4zero(), me = f4zero(), ve = f4zero();
        if (L::ALIAS && dma && active && leader) {   // (the slices' partials overwrite the landing area next)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the LDS-DMA pieces have landed
            mi = mv_wave[0 * 64 + lane]; vi = mv_wave[1 * 64 + lane];
            if (!pure) { me = mv_wave[2 * 64 + lane]; ve = mv_wave[3 * 64 + lane]; }
        }
        // (the reads above and the stores below are ordered ACROSS lanes only by instruction order: keep the compiler from
        //  moving one over the other)
        if (L::ALIAS) asm volatile("" ::: "memory");
        if (slices > 1) {  // same for every slot of a round, idle slots included
            float *mine = slot_of(grp);
            *reinterpret_cast<float4 *>(mine + lg * 4) = gi;
            *reinterpret_cast<float4 *>(mine + DP + lg * 4) = ge;
        }
        if (slices > 1 || last) __syncthreads();
        if (last) {
            if (G::REG) {
                for (int i = threadIdx.x; i < G::SLAB; i += kThreads) {
                    float x[G::RED];
#pragma unroll
                    for (int q = 0; q < G::RED; q++) x[q] = red[q * G::SLAB + i];   // (all reads first, then a fixed-order sum)
                    float sum = x[0];
#pragma unroll
                    for (int q = 1; q < G::RED; q++) sum += x[q];
                    slab[i] = sum;
                }
            } else if (threadIdx.x < kLossSlots) {
                slab[2 * EMAX * DP + EMAX + threadIdx.x] = ((red[threadIdx.x] + red[kLossSlots + threadIdx.x]) +
                                                             red[2 * kLossSlots + threadIdx.x]) + red[3 * kLossSlots + threadIdx.x];
            }
        }
        if (slices > 1) {
            if (active && leader) {
#pragma unroll 4
                for (int s = 1; s < slices; s++) {
                    const float *oth_slot = slot_of(grp + s);
                    f4add(gi, *reinterpret_cast<const float4 *>(oth_slot + lg * 4));
                    f4add(ge, *reinterpret_cast<const float4 *>(oth_slot + DP + lg * 4));
                }
            }
            if (!last) __syncthreads();  // the slots are rewritten by the next round
        }
        if (r == r0 + STAMP_ROUND) STAMP(5);
        // ---- the leader finishes the row
        if (active && leader) {
            if (cnt != 0.f) {
                f4fma(gi, cnt, reg_term(oi, k.r2, k.r1));
                f4fma(ge, cnt, reg_term(oe, k.r2, k.r1));
            }
            if (!a.fused) {
                put4<VEC, 0, FULL>(a.np[0], row, t.D, lg, gi);
                if (!pure) put4<VEC, 0, FULL>(a.np[2], row, t.D, lg, ge);
            } else {
                if (dma && L::ALIAS) {   // (taken out of the landing area above)
                } else if (dma) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the LDS-DMA pieces have landed
                    mi = mv_wave[0 * 64 + lane]; vi = mv_wave[1 * 64 + lane];
                    if (!pure) { me = mv_wave[2 * 64 + lane]; ve = mv_wave[3 * 64 + lane]; }
                } else {
                    mi = row4<VEC, FULL>(a.m[0], row, t.D, lg); vi = row4<VEC, FULL>(a.v[0], row, t.D, lg);
                    if (!pure) { me = row4<VEC, FULL>(a.m[2], row, t.D, lg); ve = row4<VEC, FULL>(a.v[2], row, t.D, lg); }
                }
                adam4(oi, gi, mi, vi, ad);
                put4<VEC, STEP_ROW_ST, FULL>(a.np[0], row, t.D, lg, oi);
                put4<VEC, STEP_ROW_ST, FULL>(a.m[0], row, t.D, lg, mi);
                put4<VEC, STEP_ROW_ST, FULL>(a.v[0], row, t.D, lg, vi);
                if (!pure) {
                    adam4(oe, ge, me, ve, ad);
                    put4<VEC, STEP_ROW_ST, FULL>(a.np[2], row, t.D, lg, oe);
                    put4<VEC, STEP_ROW_ST, FULL>(a.m[2], row, t.D, lg, me);
                    put4<VEC, STEP_ROW_ST, FULL>(a.v[2], row, t.D, lg, ve);
                }
            }
        }
    }
    STAMP(6);
    STAMP(7);
}

// =====================================================================================
// launch 2: rounds of ITEM jobs.  Per interaction: partner user rows + record, multiply-add only.
// =====================================================================================
struct IIn {
    float4 pu, pa, r0;
};
#ifndef STEP_PUSH_DEPTH
#define STEP_PUSH_DEPTH 4
#endif
#ifndef STEP_ITEM_DEPTH
#define STEP_ITEM_DEPTH 2
#endif
template <int LG, bool VEC, int EMAX, bool FULL>
__device__ __forceinline__ void item_task(const DevTables &t, const StepArgs &a, int r0, int nr, float *lds) {
    using G = Geo<LG, EMAX>;
    constexpr int U = EMAX <= 4 ? STEP_ITEM_DEPTH : 2;   // interactions in flight per group
    constexpr int DP = G::DP, RS = G::RS, NG = G::NG;
    float *sEv = lds, *sW = sEv + EMAX * DP, *slots = sW + EMAX * DP;   // [EMAX][DP] x 2, [NG][2][DP]
    float4 *mv = reinterpret_cast<float4 *>(slots + NG * 2 * DP);       // [4 waves][4][64] float4 LDS-DMA landing area
    const int lg = threadIdx.x & (LG - 1), grp = threadIdx.x / LG, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4 *mv_wave = mv + wave * 4 * 64;
    const bool pure = a.flags & INVPREF_PURE_MF;
    const bool dma = VEC && a.fused;
    StepScalars k = a.k;
    if (a.sched_state) {
        const float al = sched_slot_ptr(a.sched_state, a.sched_slot)->alpha;
        if (al == al) k.alpha = al;
    }
    const AdamScalars ad = a.sched_state ? sched_slot_ptr(a.sched_state, a.sched_slot)->ad : a.ad;
    STAMP(0);
    int4 d, d1;
    first_desc<LG>(a.desc, r0, grp, d, d1);
    stage_small(sEv, t.Ev, t.E, t.D, EMAX, DP);
    stage_small(sW, t.W, t.E, t.D, EMAX, DP);
    STAMP(1);
    for (int r = r0; r < r0 + nr; r++) {
        // (the next round's descriptor flies under this round: a further round does not start with a dependent load)
        const int4 dd = d, dd1 = d1;   // (this round's copy: what the lambdas below read)
        if (r + 1 < r0 + nr) { d = a.desc[((r + 1) * NG + grp) * 2]; d1 = a.desc[((r + 1) * NG + grp) * 2 + 1]; }
        const int row = dd.x, meta = dd.y;
        const bool active = row >= 0, leader = meta & 1;
        const int slices = (meta >> 1) & 31, mode = (meta >> 6) & 7;
        const int nsmp = active ? (mode == 7 ? dd.w - dd.z : mode) : 0;
        if (r == r0 + STAMP_ROUND) STAMP(2);
        float4 oi = f4zero(), oe = f4zero(), gi = f4zero(), ge = f4zero();
        {   // (an idle slot reads row 0 rather than branching around the loads)
            const int rowc = active ? row : 0;
            oi = row4<VEC, FULL>(t.Qi, rowc, t.D, lg);
            if (!pure) oe = row4<VEC, FULL>(t.Qa, rowc, t.D, lg);
        }
        // interaction sidx of the slice: (user row, position) inline (up to three) or from the sorted list
        auto ids_at = [&](int sidx) {
            if (mode == 7) return a.ilist[dd.z + sidx];
            if (sidx == 0) return make_int2(dd.z, dd.w);
            if (sidx == 1) return make_int2(dd1.x, dd1.y);
            return make_int2(dd1.z, dd1.w);
        };
        auto fetch = [&](IIn &in, float4 (&gzv)[EMAX / 4], int2 id) {
            in.pu = row4<VEC, FULL>(t.Pu, id.x, t.D, lg);
            const float *rec = a.records + (unsigned)id.y * (unsigned)RS;
            in.r0 = *reinterpret_cast<const float4 *>(rec);
            if (!pure) {
                in.pa = row4<VEC, FULL>(t.Pa, id.x, t.D, lg);
#pragma unroll
                for (int c4 = 0; c4 < EMAX / 4; c4++) gzv[c4] = *reinterpret_cast<const float4 *>(rec + 4 + c4 * 4);
            }
        };
        auto consume = [&](const IIn &in, const float4 (&gzv)[EMAX / 4]) {
            const float g_p = in.r0.x, g_q = in.r0.y;
            const int e = __builtin_bit_cast(int, in.r0.z);
            float4 gx = f4zero();
            if (!pure) {
#pragma unroll
                for (int c4 = 0; c4 < EMAX / 4; c4++) {
                    f4fma(gx, gzv[c4].x, *reinterpret_cast<const float4 *>(sW + (c4 * 4 + 0) * DP + lg * 4));
                    f4fma(gx, gzv[c4].y, *reinterpret_cast<const float4 *>(sW + (c4 * 4 + 1) * DP + lg * 4));
                    f4fma(gx, gzv[c4].z, *reinterpret_cast<const float4 *>(sW + (c4 * 4 + 2) * DP + lg * 4));
                    f4fma(gx, gzv[c4].w, *reinterpret_cast<const float4 *>(sW + (c4 * 4 + 3) * DP + lg * 4));
                }
                const float4 ev = *reinterpret_cast<const float4 *>(sEv + e * DP + lg * 4);
                f4fma(ge, g_q, f4mul(in.pa, ev));
            }
            float4 gip;
            gip.x = g_p - k.alpha * gx.x; gip.y = g_p - k.alpha * gx.y;
            gip.z = g_p - k.alpha * gx.z; gip.w = g_p - k.alpha * gx.w;
            f4add(gi, f4mul(gip, in.pu));
        };
        // U interactions in flight per group: the slot just consumed is refilled at once (no register copies)
        IIn nx[U];
        float4 zn[U][EMAX / 4];
        int2 idn[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
            nx[j].pu = nx[j].pa = nx[j].r0 = f4zero();
#pragma unroll
            for (int c4 = 0; c4 < EMAX / 4; c4++) zn[j][c4] = f4zero();
            idn[j] = make_int2(0, 0);
            if (j < nsmp) fetch(nx[j], zn[j], ids_at(j));
        }
#pragma unroll
        for (int j = 0; j < U; j++)
            if (U + j < nsmp) idn[j] = ids_at(U + j);
        if (dma) {   // the row's Adam moments: needed last, sent straight to LDS (no registers held across the loop)
            const bool mine = active && leader && lg * 4 < t.D;
#pragma unroll
            for (int tn = 0; tn < 4; tn++) {
                const float *src_tab = (tn & 1) ? a.v[(tn >> 1) * 2 + 1] : a.m[(tn >> 1) * 2 + 1];
                if (mine && !(pure && tn >= 2))
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(src_tab + ((unsigned)row * (unsigned)t.D + (unsigned)lg * 4u)),
                        (__attribute__((address_space(3))) void *)(mv_wave + tn * 64), 16, 0, 0);
            }
        }
        if (r == r0) { __syncthreads(); STAMP(3); }  // staged tables visible
        for (int s = 0; s < nsmp; s += U) {
#pragma unroll
            for (int j = 0; j < U; j++) {
                if (s + j < nsmp) consume(nx[j], zn[j]);
                if (s + U + j < nsmp) fetch(nx[j], zn[j], idn[j]);
                if (s + 2 * U + j < nsmp) idn[j] = ids_at(s + 2 * U + j);
            }
        }
        if (r == r0 + STAMP_ROUND) STAMP(4);
        if (slices > 1) {
            float *mine = slots + grp * 2 * DP;
            *reinterpret_cast<float4 *>(mine + lg * 4) = gi;
            *reinterpret_cast<float4 *>(mine + DP + lg * 4) = ge;
            __syncthreads();
            if (active && leader) {
#pragma unroll 4
                for (int s = 1; s < slices; s++) {
                    const float *oth_slot = slots + (grp + s) * 2 * DP;
                    f4add(gi, *reinterpret_cast<const float4 *>(oth_slot + lg * 4));
                    f4add(ge, *reinterpret_cast<const float4 *>(oth_slot + DP + lg * 4));
                }
            }
            if (r + 1 < r0 + nr) __syncthreads();
        }
        if (r == r0 + STAMP_ROUND) STAMP(5);
        if (active && leader) {
            const float cnt = (float)(meta >> 9);
            if (cnt != 0.f) {
                f4fma(gi, cnt, reg_term(oi, k.r2, k.r1));
                f4fma(ge, cnt, reg_term(oe, k.r2, k.r1));
            }
            if (!a.fused) {
                put4<VEC, 0, FULL>(a.np[1], row, t.D, lg, gi);
                if (!pure) put4<VEC, 0, FULL>(a.np[3], row, t.D, lg, ge);
            } else {
                float4 mi = f4zero(), vi = f4zero(), me = f4zero(), ve = f4zero();
                if (dma) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the LDS-DMA pieces have landed
                    mi = mv_wave[0 * 64 + lane]; vi = mv_wave[1 * 64 + lane];
                    if (!pure) { me = mv_wave[2 * 64 + lane]; ve = mv_wave[3 * 64 + lane]; }
                } else {
                    mi = row4<VEC, FULL>(a.m[1], row, t.D, lg); vi = row4<VEC, FULL>(a.v[1], row, t.D, lg);
                    if (!pure) { me = row4<VEC, FULL>(a.m[3], row, t.D, lg); ve = row4<VEC, FULL>(a.v[3], row, t.D, lg); }
                }
                adam4(oi, gi, mi, vi, ad);
                put4<VEC, STEP_ROW_ST, FULL>(a.np[1], row, t.D, lg, oi);
                put4<VEC, STEP_ROW_ST, FULL>(a.m[1], row, t.D, lg, mi);
                put4<VEC, STEP_ROW_ST, FULL>(a.v[1], row, t.D, lg, vi);
                if (!pure) {
                    adam4(oe, ge, me, ve, ad);
                    put4<VEC, STEP_ROW_ST, FULL>(a.np[3], row, t.D, lg, oe);
                    put4<VEC, STEP_ROW_ST, FULL>(a.m[3], row, t.D, lg, me);
                    put4<VEC, STEP_ROW_ST, FULL>(a.v[3], row, t.D, lg, ve);
                }
            }
        }
    }
    STAMP(6);
}

// launch 2, push form: an item job sums the CONTIGUOUS contribution rows launch 1 stored for its row (item-sorted slots
// [a, b) of the slice, read from the descriptor alone): no partner gathers, no records, no classifier rows -- every load of
// a slice of up to PCH interactions leaves in one burst.  Chosen by the plan for minibatches whose contribution rows are a
// small share of the step's bytes (they cost one extra row write + read per interaction and table).
template <int LG, bool VEC, int EMAX, bool FULL>
__device__ __forceinline__ void item_task_push(const DevTables &t, const StepArgs &a, int r0, int nr, float *lds) {
    using G = Geo<LG, EMAX>;
    constexpr int DP = G::DP, NG = G::NG;
    constexpr int PCH = LG == 16 ? STEP_PUSH_DEPTH : 3;   // contribution-row pairs in flight per group
    float *slots = lds;                                                  // [NG][2][DP] slice partials
    float4 *mv = reinterpret_cast<float4 *>(slots + NG * 2 * DP);       // [4 waves][4][64] float4 LDS-DMA landing area
    const int lg = threadIdx.x & (LG - 1), grp = threadIdx.x / LG, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4 *mv_wave = mv + wave * 4 * 64;
    const bool pure = a.flags & INVPREF_PURE_MF;
    const bool dma = VEC && a.fused;
    const StepScalars k = a.k;
    const AdamScalars ad = a.sched_state ? sched_slot_ptr(a.sched_state, a.sched_slot)->ad : a.ad;
    STAMP(0);
    int4 d, d1_unused;
    first_desc<LG>(a.desc, r0, grp, d, d1_unused);
    STAMP(1);
    for (int r = r0; r < r0 + nr; r++) {
        const int4 dd = d;
        if (r + 1 < r0 + nr) d = a.desc[((r + 1) * NG + grp) * 2];
        const int row = dd.x, meta = dd.y;
        const bool active = row >= 0, leader = meta & 1;
        const int slices = (meta >> 1) & 31, mode = (meta >> 6) & 7;
        const int nsmp = (active && mode == 7) ? dd.w - dd.z : 0;
        if (r == r0 + STAMP_ROUND) STAMP(2);
        float4 oi, oe = f4zero(), gi = f4zero(), ge = f4zero();
        {
            const int rowc = active ? row : 0;
            oi = row4<VEC, FULL>(t.Qi, rowc, t.D, lg);
            if (!pure) oe = row4<VEC, FULL>(t.Qa, rowc, t.D, lg);
        }
        const float *base = a.records + (unsigned)dd.z * (unsigned)(2 * DP) + lg * 4;
        // Two register sets of H = PCH / 2 pairs each: while one set's rows are added the other's are in flight, and a set is
        // refilled as soon as it has been added -- a slice of n rows is one round trip + n additions, not n / PCH round trips
        // (the refill used to sit under a divergent branch: waited for at the join, every batch a round trip of its own).
        // Rows are added in slice order either way: the sums are bit for bit those of the one-set form.  Refills are taken
        // while ANY group of the wave still has rows to come (a wave-uniform branch); indices are clamped into the slice.
        constexpr int H = PCH / 2;
        static_assert(PCH % 2 == 0, "two register sets");
        float4 ci[2][H], ce[2][H];
        auto fetch = [&](int set, int s0) {
#pragma unroll
            for (int j = 0; j < H; j++) {
                const int sj = s0 + j < nsmp ? s0 + j : (nsmp > 0 ? nsmp - 1 : 0);
                const float *p = base + (unsigned)sj * (unsigned)(2 * DP);
                ci[set][j] = *reinterpret_cast<const float4 *>(nsmp > 0 ? p : a.records + lg * 4);
                if (!pure) ce[set][j] = *reinterpret_cast<const float4 *>(nsmp > 0 ? p + DP : a.records + DP + lg * 4);
            }
        };
        auto add = [&](int set, int s0) {
#pragma unroll
            for (int j = 0; j < H; j++) {
                const bool has = s0 + j < nsmp;
                f4add(gi, has ? ci[set][j] : f4zero());
                if (!pure) f4add(ge, has ? ce[set][j] : f4zero());
            }
        };
        fetch(0, 0);
        fetch(1, H);
        if (dma) {   // the row's Adam moments: needed last, sent straight to LDS
            const bool mine = active && leader && lg * 4 < t.D;
#pragma unroll
            for (int tn = 0; tn < 4; tn++) {
                const float *src_tab = (tn & 1) ? a.v[(tn >> 1) * 2 + 1] : a.m[(tn >> 1) * 2 + 1];
                if (mine && !(pure && tn >= 2))
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(src_tab + ((unsigned)row * (unsigned)t.D + (unsigned)lg * 4u)),
                        (__attribute__((address_space(3))) void *)(mv_wave + tn * 64), 16, 0, 0);
            }
        }
        if (r == r0 + STAMP_ROUND) STAMP(3);
        int n_wave = nsmp;   // the wave's longest slice
#pragma unroll
        for (int g = 0; g < 64 / LG; g++) n_wave = max(n_wave, __builtin_amdgcn_readlane(nsmp, g * LG));
        for (int s0 = 0; s0 < n_wave; s0 += PCH) {
            add(0, s0);
            if (s0 + PCH < n_wave) fetch(0, s0 + PCH);
            add(1, s0 + H);
            if (s0 + PCH + H < n_wave) fetch(1, s0 + PCH + H);
        }
        if (r == r0 + STAMP_ROUND) STAMP(4);
        if (slices > 1) {
            float *mine = slots + grp * 2 * DP;
            *reinterpret_cast<float4 *>(mine + lg * 4) = gi;
            *reinterpret_cast<float4 *>(mine + DP + lg * 4) = ge;
            __syncthreads();
            if (active && leader) {
#pragma unroll 4
                for (int s = 1; s < slices; s++) {
                    const float *oth_slot = slots + (grp + s) * 2 * DP;
                    f4add(gi, *reinterpret_cast<const float4 *>(oth_slot + lg * 4));
                    f4add(ge, *reinterpret_cast<const float4 *>(oth_slot + DP + lg * 4));
                }
            }
            if (r + 1 < r0 + nr) __syncthreads();
        }
        if (r == r0 + STAMP_ROUND) STAMP(5);
        if (active && leader) {
            const float cnt = (float)(meta >> 9);
            if (cnt != 0.f) {
                f4fma(gi, cnt, reg_term(oi, k.r2, k.r1));
                f4fma(ge, cnt, reg_term(oe, k.r2, k.r1));
            }
            if (!a.fused) {
                put4<VEC, 0, FULL>(a.np[1], row, t.D, lg, gi);
                if (!pure) put4<VEC, 0, FULL>(a.np[3], row, t.D, lg, ge);
            } else {
                float4 mi = f4zero(), vi = f4zero(), me = f4zero(), ve = f4zero();
                if (dma) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the LDS-DMA pieces have landed
                    mi = mv_wave[0 * 64 + lane]; vi = mv_wave[1 * 64 + lane];
                    if (!pure) { me = mv_wave[2 * 64 + lane]; ve = mv_wave[3 * 64 + lane]; }
                } else {
                    mi = row4<VEC, FULL>(a.m[1], row, t.D, lg); vi = row4<VEC, FULL>(a.v[1], row, t.D, lg);
                    if (!pure) { me = row4<VEC, FULL>(a.m[3], row, t.D, lg); ve = row4<VEC, FULL>(a.v[3], row, t.D, lg); }
                }
                adam4(oi, gi, mi, vi, ad);
                put4<VEC, STEP_ROW_ST, FULL>(a.np[1], row, t.D, lg, oi);
                put4<VEC, STEP_ROW_ST, FULL>(a.m[1], row, t.D, lg, mi);
                put4<VEC, STEP_ROW_ST, FULL>(a.v[1], row, t.D, lg, vi);
                if (!pure) {
                    adam4(oe, ge, me, ve, ad);
                    put4<VEC, STEP_ROW_ST, FULL>(a.np[3], row, t.D, lg, oe);
                    put4<VEC, STEP_ROW_ST, FULL>(a.m[3], row, t.D, lg, me);
                    put4<VEC, STEP_ROW_ST, FULL>(a.v[3], row, t.D, lg, ve);
                }
            }
        }
    }
    STAMP(6);
}

// Untouched rows: gradient exactly zero, so m' = m + (1-b1)(0-m), v' = b2 v, p' = p - step*m'/(sqrt(v')/bc+eps)
// (the same adam1f as everywhere, fed g = 0).  Each group keeps R = 2 rows of both tables in flight (12 float4 loads).
// The untouched rows are independent of everything else in a launch and their workgroups are short: started a little
// late, their load burst does not queue in front of the row jobs' first gathers (the launch's critical chain).
// Units: s_sleep counts of 64 clocks (0: no delay).  Measured at the Yahoo shape (tools/ab.sh): 75 = about 2 us in
// launch 1: 21.9 -> 21.3 us per step; longer delays give the gain back.  (Launch 2 in the push form: its item jobs are
// shorter than its stream workgroups, so no delay there.)
#ifndef STEP_STREAM_DELAY1
#define STEP_STREAM_DELAY1 75
#endif
#ifndef STEP_STREAM_DELAY2
#define STEP_STREAM_DELAY2 0
#endif
template <int N>
__device__ __forceinline__ void stream_delay() {
    if (N > 0) __builtin_amdgcn_s_sleep(N > 127 ? 127 : N);
    if (N > 127) __builtin_amdgcn_s_sleep(N - 127 > 127 ? 127 : N - 127);
}
#ifndef STEP_STREAM_ST
#define STEP_STREAM_ST 1   // (A/B knob: 1 = write-through stores for the streamed rows)
#endif
// (Tried in round 3: two register sets with the next rows' loads issued before the current rows' stores, so that a task
//  of several iterations would be one round trip + work -- 128-row tasks in launch 1 ran 23-28 us per step against
//  19.4: the rows per CU, not the iterations' round trips, pace these workgroups.  One iteration per task it is.)
template <int LG, bool VEC, bool FULL>
__device__ __forceinline__ void stream_task(const DevTables &t, const StepArgs &a, const int *rows, int n) {
    constexpr int R = 2, NG = kThreads / LG;
    const int lg = threadIdx.x & (LG - 1), grp = threadIdx.x / LG;
    const AdamScalars ad = a.sched_state ? sched_slot_ptr(a.sched_state, a.sched_slot)->ad : a.ad;
    const bool pure = a.flags & INVPREF_PURE_MF;
    if (!a.fused) {   // gradient form: the untouched rows' gradient is a row of zeros
        for (int i = grp; i < n; i += NG) {
            const int rid = rows[i], side = (rid >> 30) & 1, row = rid & 0x3fffffff;
            put4<VEC, 0, FULL>(side ? a.np[1] : a.np[0], row, t.D, lg, f4zero());
            if (!pure) put4<VEC, 0, FULL>(side ? a.np[3] : a.np[2], row, t.D, lg, f4zero());
        }
        return;
    }
    struct Set {
        int row[R], side[R];
        bool on[R];
        float4 p[2 * R], m[2 * R], v[2 * R];  // {row 0 inv, row 0 env, row 1 inv, ...}
    };
    auto load = [&](Set &S, int i0) {
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int idx = i0 + q * NG;
            S.on[q] = idx < n;
            const int rid = rows[S.on[q] ? idx : 0];      // (clamped: a slot beyond the task re-reads row 0 of it)
            S.side[q] = (rid >> 30) & 1;
            S.row[q] = rid & 0x3fffffff;
        }
#pragma unroll
        for (int q = 0; q < 2 * R; q++) {
            S.p[q] = S.m[q] = S.v[q] = f4zero();
            const int s = S.side[q >> 1];
            if (!(pure && (q & 1))) {
                const float *T = (q & 1) ? (s ? t.Qa : t.Pa) : (s ? t.Qi : t.Pu);
                const float *M = (q & 1) ? (s ? a.m[3] : a.m[2]) : (s ? a.m[1] : a.m[0]);
                const float *V = (q & 1) ? (s ? a.v[3] : a.v[2]) : (s ? a.v[1] : a.v[0]);
                S.p[q] = row4<VEC, FULL>(T, S.row[q >> 1], t.D, lg);
                S.m[q] = row4<VEC, FULL>(M, S.row[q >> 1], t.D, lg);
                S.v[q] = row4<VEC, FULL>(V, S.row[q >> 1], t.D, lg);
            }
        }
    };
    auto finish = [&](Set &S) {
#pragma unroll
        for (int q = 0; q < 2 * R; q++) {
            const int s = S.side[q >> 1];
            if (S.on[q >> 1] && !(pure && (q & 1))) {
                adam4(S.p[q], f4zero(), S.m[q], S.v[q], ad);
                float *NP = (q & 1) ? (s ? a.np[3] : a.np[2]) : (s ? a.np[1] : a.np[0]);
                float *M = (q & 1) ? (s ? a.m[3] : a.m[2]) : (s ? a.m[1] : a.m[0]);
                float *V = (q & 1) ? (s ? a.v[3] : a.v[2]) : (s ? a.v[1] : a.v[0]);
                put4<VEC, STEP_STREAM_ST, FULL>(NP, S.row[q >> 1], t.D, lg, S.p[q]);
                put4<VEC, STEP_STREAM_ST, FULL>(M, S.row[q >> 1], t.D, lg, S.m[q]);
                put4<VEC, STEP_STREAM_ST, FULL>(V, S.row[q >> 1], t.D, lg, S.v[q]);
            }
        }
    };
    const int iters = (n + R * NG - 1) / (R * NG);   // (workgroup-uniform)
    Set A;
    for (int it = 0; it < iters; it++) {
        load(A, grp + it * R * NG);
        finish(A);
    }
}

// ---- the fold: partial slabs -> gradients of embed_env / classifier (+ classifier regulariser, models.py:211-217)
// -> Adam (fused) or gradient store; the six loss outputs; the device-side schedule moves on.
struct FoldArgs {
    float *gEv, *gW, *gb;        // fused == 0
    float *nEv, *nW, *nb;        // fused == 1: new parameters
    float *mEv, *mW, *mb, *vEv, *vW, *vb;
    int n_partials, n_task_wgs, fold_blocks;
    const float *slabs_ev;       // EVL2: embed_env's partial sums come from launch 2's item tasks ...
    int n_partials_ev;           // ... this many slabs of EMAX x DP floats
    float l2, l1;
    double inv_B, inv_BD2;       // 1 / Bnorm, 1 / (2 Bnorm D)
    float *losses6;
    const SchedRow *sched_table;
    int sched_n;
};
constexpr int kFoldCols = 16, kFoldSubs = kThreads / kFoldCols;   // a fold block: 16 columns x 16 sub-rows of partials

template <int DP, int EMAX>
__device__ __forceinline__ void fold_block(const DevTables &t, const StepArgs &a, const FoldArgs &f, int fb, float *lds) {
    constexpr int SLAB = 2 * EMAX * DP + EMAX + kLossSlots, EDP = EMAX * DP;
    double *part = reinterpret_cast<double *>(lds);   // [kFoldSubs][kFoldCols]
    const int colx = threadIdx.x % kFoldCols, sub = threadIdx.x / kFoldCols;
    const int idx = fb * kFoldCols + colx;
    const bool pure = a.flags & INVPREF_PURE_MF;
    const bool dense = (a.flags & INVPREF_DENSE_REG) && !(a.flags & INVPREF_REG_ONLY_EMBED) && !pure;
    const AdamScalars ad = a.sched_state ? sched_slot_ptr(a.sched_state, a.sched_slot)->ad : a.ad;
    const bool mine = idx < SLAB;
    // which output this column is
    const bool isLoss = idx >= 2 * EDP + EMAX, isB = !isLoss && idx >= 2 * EDP, isW = !isLoss && !isB && idx >= EDP;
    const int rr = isW ? idx - EDP : idx;
    const int e = isB ? idx - 2 * EDP : rr / DP, dd = isB ? 0 : rr - e * DP;
    const bool live = mine && !isLoss && e < t.E && dd < t.D && !pure;
    const int off = live ? (isB ? e : e * t.D + dd) : 0;
    // parameter / moments of the output this thread finishes (sub == 0 threads), requested up front
    float pre_p = 0.f, pre_m = 0.f, pre_v = 0.f;
    if (sub == 0 && live) {
        pre_p = (isB ? t.b : (isW ? t.W : t.Ev))[off];
        if (a.fused) {
            pre_m = (isB ? f.mb : (isW ? f.mW : f.mEv))[off];
            pre_v = (isB ? f.vb : (isW ? f.vW : f.vEv))[off];
        }
    }
    // this thread's column of the partials sub, sub + kFoldSubs, ...: 8 loads in flight together (clamped, not
    // guarded), summed in partial order in fp64
    const bool from_ev = f.slabs_ev != nullptr && mine && idx < EDP;   // (embed_env's columns: launch 2's slabs)
    const float *col = from_ev ? f.slabs_ev + idx : a.slabs + (mine ? idx : 0);
    const int np = from_ev ? f.n_partials_ev : f.n_partials;
    const int64_t stride = from_ev ? EDP : SLAB;
    double acc = 0.0;
#ifndef STEP_FOLD_CH
#define STEP_FOLD_CH 32
#endif
    constexpr int CH = STEP_FOLD_CH;   // 32 x 16 sub-rows: up to 512 partials in ONE round trip
    for (int s0 = sub; s0 < np; s0 += CH * kFoldSubs) {
        float x[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) x[j] = col[(int64_t)min(s0 + j * kFoldSubs, np - 1) * stride];
#pragma unroll
        for (int j = 0; j < CH; j++) acc += (s0 + j * kFoldSubs < np) ? (double)x[j] : 0.0;
    }
    part[sub * kFoldCols + colx] = acc;
    __syncthreads();
    if (sub != 0 || !mine) return;
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < kFoldSubs; q++) v += part[q * kFoldCols + colx];
    if (!isLoss) {
        if (!live) return;
        float gv = (float)v, pv = pre_p;
        if (isB) {
            if (dense) gv += 2.f * f.l2 / (float)t.E * pv + f.l1 / (float)t.E * c_sign(pv);
        } else if (isW && dense) {
            gv += 2.f * f.l2 / ((float)t.D * (float)t.E) * pv + f.l1 / ((float)t.D * (float)t.E) * c_sign(pv);
        }
        if (!a.fused) {
            (isB ? f.gb : (isW ? f.gW : f.gEv))[off] = gv;
        } else {
            float mm = pre_m, vv = pre_v;
            adam1(pv, gv, mm, vv, ad);
            (isB ? f.nb : (isW ? f.nW : f.nEv))[off] = pv;
            (isB ? f.mb : (isW ? f.mW : f.mEv))[off] = mm;
            (isB ? f.vb : (isW ? f.vW : f.vEv))[off] = vv;
        }
        return;
    }
    // the loss columns (they are contiguous and, SLAB being a multiple of 4, inside ONE fold block when
    // kFoldCols >= 8 divides their offset -- asserted on the host): lanes share them through LDS
    double *sl = part + kFoldSubs * kFoldCols;
    sl[idx - 2 * EDP - EMAX] = v;
}

__device__ __forceinline__ void fold_losses(const DevTables &t, const StepArgs &a, const FoldArgs &f, float *lds) {
    // called by the block that holds the loss columns, after fold_block and a barrier
    const double *sl = reinterpret_cast<const double *>(lds) + kFoldSubs * kFoldCols;
    const bool pure = a.flags & INVPREF_PURE_MF;
    const bool dense = (a.flags & INVPREF_DENSE_REG) && !(a.flags & INVPREF_REG_ONLY_EMBED) && !pure;
    double reg2 = 0.0, reg1 = 0.0;
    if (threadIdx.x < 64) {
        if (dense) {   // regulariser report of the classifier (models.py:211-217)
            double w2 = 0, w1 = 0, b2 = 0, b1 = 0;
            for (int i = threadIdx.x; i < t.E * t.D; i += 64) { const double xx = t.W[i]; w2 += xx * xx; w1 += fabs(xx); }
            for (int i = threadIdx.x; i < t.E; i += 64) { const double xx = t.b[i]; b2 += xx * xx; b1 += fabs(xx); }
            reg2 = w2 / ((double)t.D * t.E) + b2 / (double)t.E;
            reg1 = w1 / ((double)t.D * t.E) + b1 / (double)t.E;
            for (int m = 32; m >= 1; m >>= 1) { reg2 += __shfl_xor(reg2, m, 64); reg1 += __shfl_xor(reg1, m, 64); }
        }
        if (threadIdx.x == 0 && f.losses6) {
            const StepScalars &k = a.k;
            const double Li = sl[0] * f.inv_B, Le = sl[1] * f.inv_B, Lc = sl[2] * f.inv_B;
            const double L2 = sl[3] * f.inv_BD2 + reg2, L1 = sl[4] * f.inv_BD2 + reg1;
            // ADDED (the caller zeroes; a row-sharded run sums its ranks' partial terms): fire-and-forget atomics from
            // this single thread -- one adder per call, so the result does not depend on any ordering
            atomicAdd(f.losses6 + 0, (float)Li); atomicAdd(f.losses6 + 1, (float)Le); atomicAdd(f.losses6 + 2, (float)Lc);
            atomicAdd(f.losses6 + 3, (float)L2); atomicAdd(f.losses6 + 4, (float)L1);
            atomicAdd(f.losses6 + 5, (float)((double)k.ca * Li + (double)k.cb * Le + (double)k.cc * Lc + (double)f.l2 * L2 + (double)f.l1 * L1));
        }
    }
}

// class c's task counts from the by-value table (a masked sum instead of a dynamic index: indexing a by-value kernel
// argument with a run-time value can make the compiler copy the whole argument block to scratch memory)
__device__ __forceinline__ void class_row(const StepArgs &a, int c, int (&q)[4]) {
    q[0] = q[1] = q[2] = q[3] = 0;
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
        const int on = (c == kk) ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 4; i++) q[i] += a.cls[kk][i] * on;
    }
}

// registers: the instances are held to 4 (E <= 4) / 3 workgroups per CU for launch 1 and 6 / 4 / 3 for launch 2
#ifndef STEP_EVAL_WAVES_SMALL
#define STEP_EVAL_WAVES_SMALL 3   // the smallest instance (D <= 64, E <= 4): 52 KB of LDS, <= 168 registers
#endif
#ifndef STEP_EVAL_WAVES
#define STEP_EVAL_WAVES 2
#endif
#ifndef STEP_APPLY_WAVES
#define STEP_APPLY_WAVES 4
#endif
template <int LG, bool VEC, int EMAX, bool FULL = false>
__global__ __launch_bounds__(kThreads, (LG == 16 && EMAX <= 4) ? STEP_EVAL_WAVES_SMALL : STEP_EVAL_WAVES) void mstep_eval_kernel(DevTables t, StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // Workgroup b runs the tasks of class c = b % n_cls (XCD-affine order, InvPrefRowPlan), the j-th of them with
    // j = b / n_cls: its user jobs first, then its share of the untouched rows.  Every branch is workgroup-uniform.
    const int ncls = a.n_cls;
    const int c = (int)blockIdx.x % ncls;
    int j = (int)blockIdx.x / ncls;
    int q[4];
    class_row(a, c, q);
    const int rpt = a.rounds_per_task, spt = a.rows_per_stream_task;
    const int tj = (q[1] + rpt - 1) / rpt;
    if (j < tj) {
#ifndef DBG_NO_JOBS
        user_task<LG, VEC, EMAX, FULL>(t, a, q[0] + j * rpt, min(rpt, q[1] - j * rpt), q[0] / rpt + j, lds);
#endif
        return;
    }
    j -= tj;
    if (j * spt < q[3]) {
        STAMP(0);
#ifndef DBG_NO_STREAM
        stream_delay<STEP_STREAM_DELAY1>();
        stream_task<LG, VEC, FULL>(t, a, a.stream_rows + q[2] + j * spt, min(spt, q[3] - j * spt));
#endif
        STAMP(7);
    }
}

template <int LG, bool VEC, int EMAX, bool FULL = false>
__global__ __launch_bounds__(kThreads, FULL ? 3 : STEP_APPLY_WAVES) void mstep_apply_kernel(DevTables t, StepArgs a, FoldArgs f) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if ((int)blockIdx.x >= f.n_task_wgs) {
        const int fb = (int)blockIdx.x - f.n_task_wgs;
        if (fb == f.fold_blocks) {
            // the device-side schedule moves on: one thread of the LAST block fills the OTHER slot with the next step's
            // number and scalars.  Nobody reads that slot before the next launch, so no ordering is needed.
            // (gradient-pass form, fused == 0: the stand-alone Adam kernel that follows is the step's last launch and
            //  moves the schedule on; here the slot is only read, for a scheduled alpha)
            if (a.sched_state && a.fused && threadIdx.x == 0) {
                const int *cur = a.sched_state + 16 * a.sched_slot;
                int *nxt = a.sched_state + 16 * (a.sched_slot ^ 1);
                const int next = cur[0] + 1, base = cur[1], idx = next - base;
                nxt[0] = next;
                nxt[1] = base;
                if (idx >= 0 && idx < f.sched_n) *reinterpret_cast<SchedRow *>(nxt + 2) = f.sched_table[idx];
            }
            return;
        }
        STAMP(0);
#ifndef DBG_NO_FOLD
        fold_block<4 * LG, EMAX>(t, a, f, fb, lds);
        constexpr int loss0 = 2 * EMAX * 4 * LG + EMAX;
        if (fb == loss0 / kFoldCols) {   // (workgroup-uniform)
            __syncthreads();
            fold_losses(t, a, f, lds);
        }
#endif
        STAMP(7);
        return;
    }
    const int ncls = a.n_cls;
    const int c = (int)blockIdx.x % ncls;
    int j = (int)blockIdx.x / ncls;
    int q[4];
    class_row(a, c, q);
    const int rpt = a.rounds_per_task, spt = a.rows_per_stream_task;
    const int tj = (q[1] + rpt - 1) / rpt;
    if (j < tj) {
#ifndef DBG_NO_JOBS
        if (a.push_slot) item_task_push<LG, VEC, EMAX, FULL>(t, a, q[0] + j * rpt, min(rpt, q[1] - j * rpt), lds);
        else item_task<LG, VEC, EMAX, FULL>(t, a, q[0] + j * rpt, min(rpt, q[1] - j * rpt), lds);
#endif
        return;
    }
    j -= tj;
    if (j * spt < q[3]) {
        STAMP(0);
#ifndef DBG_NO_STREAM
        stream_delay<STEP_STREAM_DELAY2>();
        stream_task<LG, VEC, FULL>(t, a, a.stream_rows + q[2] + j * spt, min(spt, q[3] - j * spt));
#endif
        STAMP(7);
    }
}

#include "step_wide.hpp"
#include "step_wide_mm.hpp"

// The row layout and the kernel family of a shape.  Rows of up to 64 floats with up to four environments (Yahoo, Coat,
// the PureMF baselines) run the latency-tuned kernels above (16 lanes x 1 float4); everything else the wide ones
// (step_wide.hpp): 16 lanes x 1 or 2 float4 up to 128 floats, 32 lanes x 2 float4 up to 256, class counts 8 or 16.
struct Shape {
    int lg, nc, dp, emax;
    bool wide, evl2;   // evl2: embed_env's outer product runs in launch 2 (pull form only) and a third launch folds
};
inline int emax4_of(int E) { return E <= 4 ? 4 : (E <= 8 ? 8 : 16); }
inline Shape shape_of(int D, int E) {
    Shape s;
    s.lg = D <= 128 ? 16 : 32;
    s.nc = D <= 64 ? 1 : 2;
    s.dp = 4 * s.lg * s.nc;
    const int e4 = emax4_of(E);
    s.wide = !(s.nc == 1 && e4 == 4);
    s.emax = s.wide ? (e4 < 8 ? 8 : e4) : 4;
    s.evl2 = s.lg == 32;
    return s;
}
inline int lanes_of(int D) { return D <= 128 ? 16 : 32; }
inline size_t slab_floats(const Shape &s) { return (size_t)2 * s.emax * s.dp + s.emax + kLossSlots; }
inline size_t eval_lds_bytes(const Shape &s) {
    if (!s.wide) return sizeof(float) * EvalLds<16, 4>::total;
    // (the layout itself: WGeo / WGeo::Img in step_wide.hpp)
    const size_t ng = kThreads / s.lg;
    const size_t live = (size_t)2 * s.emax * s.dp + 32 + ng * 2 * s.dp + 2 * ng * (s.emax + 4);
    const size_t len = (size_t)(s.evl2 ? 1 : 2) * s.emax * s.dp;
    const size_t nimg = 4 * len <= 10240 ? 4 : (2 * len <= 10240 ? 2 : 1);
    const size_t tail = nimg * len > live ? nimg * len : live;
    return sizeof(float) * (tail + kWaves * (s.emax + kLossSlots));
}
inline size_t fold_lds_bytes() { return (kFoldSubs * kFoldCols + kLossSlots) * sizeof(double); }
inline size_t apply_lds_bytes(const Shape &s) {
    const size_t DP = s.dp, NG = kThreads / s.lg;
    const size_t job = (2 * s.emax * DP + NG * 2 * DP + (s.wide ? 0 : kWaves * 4 * 64 * 4)) * sizeof(float);
    return job > fold_lds_bytes() ? job : fold_lds_bytes();
}

template <typename K>
int ensure_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

inline size_t record_floats(const InvPrefRowPlan *plan, const Shape &s) {
    // pull form: a record of 4 + EMAX floats per interaction; push form: two padded contribution rows
    // (+ 1: the spare one an empty slot of a lock-step iteration stores to, step_wide.hpp)
    const size_t per = plan->push_slot ? (size_t)2 * s.dp : (size_t)4 + s.emax;
    return ((size_t)((plan->n > 0 ? plan->n : 0) + 1) * per + 63) & ~(size_t)63;
}
inline size_t workspace_floats(const InvPrefRowPlan *plan, const Shape &s) {
    // records | launch 1's partial slabs | (evl2) launch 2's partial slabs of embed_env's gradient
    const size_t np = (size_t)(plan->n_user_rounds / plan->user_rounds_per_task);
    const size_t ni = (size_t)(plan->n_item_rounds / plan->item_rounds_per_task);
    return record_floats(plan, s) + slab_floats(s) * (np > 0 ? np : 1) +
           (s.evl2 ? (size_t)s.emax * s.dp * (ni > 0 ? ni : 1) : 0);
}

inline int plan_tasks(const InvPrefRowPlan *plan, int launch, int *task_wgs) {
    // workgroups of one launch: the classes' task lists interleaved, padded to the longest
    const int ncls = plan->n_classes > 0 ? plan->n_classes : 1;
    const int rpt = launch == 0 ? plan->user_rounds_per_task : plan->item_rounds_per_task;
    const int spt = (launch == 1 && plan->rows_per_stream_task2 > 0) ? plan->rows_per_stream_task2 : plan->rows_per_stream_task;
    int per_class = 0;
    for (int c = 0; c < ncls; c++) {
        const int32_t *q = plan->cls[c] + 4 * launch;
        const int ns = q[3];
        const int tot = (q[1] + rpt - 1) / rpt + (ns + spt - 1) / spt;
        per_class = tot > per_class ? tot : per_class;
    }
    *task_wgs = per_class * ncls;
    return 0;
}

int launch_step(const InvPrefTables *tables, const InvPrefRowPlan *plan, const int64_t *envs, const float *scores,
                const float *weights, int64_t batch_norm, const InvPrefCoefs *coefs, uint32_t flags, float *losses6,
                void *workspace, size_t workspace_bytes, hipStream_t st, int fused, const InvPrefTables *grads,
                const InvPrefTables *new_tables, const InvPrefTables *exp_avg, const InvPrefTables *exp_avg_sq,
                const AdamScalars &ad, const InvPrefAdamSchedule *sched, void *profile_event) {
    const bool pure = flags & INVPREF_PURE_MF;
    int rc = check_tables(tables, pure);
    if (sched && (!sched->state || !sched->table || sched->n <= 0)) return INVPREF_EINVAL;
    if (rc) return rc;
    if (!plan || !coefs || !losses6 || !workspace || (!envs && !pure) || batch_norm <= 0) return INVPREF_EINVAL;
    (void)scores;   // (the labels travel inside the plan)
    if (pure && (flags & (INVPREF_REWEIGHT_CLS | INVPREF_REG_ENV_EMBED))) return INVPREF_EINVAL;
    if ((flags & (INVPREF_REWEIGHT_REC | INVPREF_REWEIGHT_CLS)) && !weights) return INVPREF_EINVAL;
    if ((flags & INVPREF_WEIGHTS_BY_ENV) && pure) return INVPREF_EINVAL;
    const DevTables t = dev_tables(tables);
    // rows are addressed with 32-bit byte offsets (row4 / put4): every table must stay below 4 GiB
    if ((uint64_t)(t.U > t.I ? t.U : t.I) * (uint64_t)t.D * 4ull >= (1ull << 32)) return INVPREF_EUNSUPPORTED;
    bool vec = vec_ok(tables);
    StepArgs a{};
    if (!fused) {
        if ((rc = check_tables(grads, pure))) return rc;
        vec = vec && vec_ok(grads);
        a.np[0] = grads->embed_user_invariant; a.np[1] = grads->embed_item_invariant;
        a.np[2] = grads->embed_user_env_aware; a.np[3] = grads->embed_item_env_aware;
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
    const Shape shp = shape_of(t.D, t.E);
    const int lg = shp.lg, emax = shp.emax;
    // ---- the plan
    if (plan->lanes_per_group != lg) return INVPREF_EINVAL;   // built for another row layout
    if (plan->n < 0 || plan->n_user_rounds < 0 || plan->n_item_rounds < 0 || plan->user_rounds_per_task <= 0 ||
        plan->item_rounds_per_task <= 0 || plan->n_user_rounds % plan->user_rounds_per_task != 0 ||
        plan->n_item_rounds % plan->item_rounds_per_task != 0 || plan->rows_per_stream_task <= 0 || plan->n_stream < 0 ||
        (plan->n_user_rounds > 0 && (!plan->user_desc || !plan->user_round_iters)) ||
        (plan->n_item_rounds > 0 && !plan->item_desc) || (plan->n > 0 && (!plan->user_list || !plan->item_list || !plan->rec_slot)) ||
        (plan->n_stream > 0 && !plan->stream_rows))
        return INVPREF_EINVAL;
    const int ncls = plan->n_classes > 0 ? plan->n_classes : 1;
    if (ncls > 8) return INVPREF_EINVAL;
    for (int c = 0; c < ncls; c++) {
        const int32_t *q = plan->cls[c];
        for (int i = 0; i < 8; i++) if (q[i] < 0) return INVPREF_EINVAL;
        if (q[0] + q[1] > plan->n_user_rounds || q[4] + q[5] > plan->n_item_rounds || q[2] + q[3] > plan->n_stream ||
            q[6] + q[7] > plan->n_stream || q[0] % plan->user_rounds_per_task || q[4] % plan->item_rounds_per_task)
            return INVPREF_EINVAL;
    }
    const int n_partials = plan->n_user_rounds / plan->user_rounds_per_task;
    const size_t slab = slab_floats(shp);
    const size_t rec_floats = record_floats(plan, shp);
    if (workspace_bytes < sizeof(float) * workspace_floats(plan, shp)) return INVPREF_EWORKSPACE;
    if (shp.evl2 && plan->push_slot) return INVPREF_EINVAL;   // rows on 32 lanes: pull form only (step_wide.hpp)
    StepScalars k;
    k.ca = coefs->invariant_coe; k.cb = coefs->env_aware_coe; k.cc = coefs->env_coe; k.alpha = coefs->alpha;
    k.invB = 1.0f / (float)batch_norm;
    k.r2 = coefs->L2_coe / ((float)batch_norm * (float)t.D);
    k.r1 = coefs->L1_coe / (2.0f * (float)batch_norm * (float)t.D);
    a.envs = envs; a.weights = weights; a.k = k; a.flags = flags; a.fused = fused; a.ad = ad;
    a.rows_per_stream_task = plan->rows_per_stream_task; a.n_cls = ncls;
    a.records = (float *)workspace; a.slabs = (float *)workspace + rec_floats;
    a.slabs_ev = a.slabs + slab * (size_t)(n_partials > 0 ? n_partials : 1);
    a.n_rec = plan->n;
    a.push_slot = plan->push_slot;
    a.rec_slot = plan->push_slot ? plan->push_slot : plan->rec_slot;
    a.sched_state = sched ? sched->state : nullptr;
    a.sched_slot = sched ? (sched->slot & 1) : 0;
    static const char *stamp_env = getenv("INVPREF_STAMPS");   // diagnostics: device pointer (hex) of a stamp buffer
    unsigned long long *stamps = stamp_env ? reinterpret_cast<unsigned long long *>(strtoull(stamp_env, nullptr, 16)) : nullptr;
    static const bool nodrain = getenv("INVPREF_STAMPS_NODRAIN") != nullptr;
    a.stamps_nodrain = nodrain;
    int wg1 = 0, wg2 = 0;
    plan_tasks(plan, 0, &wg1);
    plan_tasks(plan, 1, &wg2);
    const size_t lds1 = eval_lds_bytes(shp), lds2 = apply_lds_bytes(shp);
    if (lds1 > 160 * 1024 || lds2 > 160 * 1024) return INVPREF_EUNSUPPORTED;
    // launch 1
    StepArgs a1 = a;
    a1.desc = reinterpret_cast<const int4 *>(plan->user_desc);
    a1.round_iters = plan->user_round_iters;
    a1.ulist = reinterpret_cast<const int4 *>(plan->user_list);
    a1.rounds_per_task = plan->user_rounds_per_task;
    a1.stream_rows = plan->stream_rows;
    a1.stamps = stamps;
    for (int c = 0; c < 8; c++) for (int i = 0; i < 4; i++) a1.cls[c][i] = c < ncls ? plan->cls[c][i] : 0;
    // launch 2
    StepArgs a2 = a;
    a2.desc = reinterpret_cast<const int4 *>(plan->item_desc);
    a2.ilist = reinterpret_cast<const int2 *>(plan->item_list);
    a2.rounds_per_task = plan->item_rounds_per_task;
    if (plan->rows_per_stream_task2 > 0) a2.rows_per_stream_task = plan->rows_per_stream_task2;
    a2.stream_rows = plan->stream_rows;
    a2.stamps = stamps ? stamps + 8192 * 8 : nullptr;   // (the stamp buffer's second half belongs to launch 2)
    for (int c = 0; c < 8; c++) for (int i = 0; i < 4; i++) a2.cls[c][i] = c < ncls ? plan->cls[c][4 + i] : 0;
    FoldArgs f{};
    if (!fused) {
        f.gEv = grads->embed_env; f.gW = grads->classifier_weight; f.gb = grads->classifier_bias;
    } else {
        f.nEv = new_tables->embed_env; f.nW = new_tables->classifier_weight; f.nb = new_tables->classifier_bias;
        f.mEv = exp_avg->embed_env; f.mW = exp_avg->classifier_weight; f.mb = exp_avg->classifier_bias;
        f.vEv = exp_avg_sq->embed_env; f.vW = exp_avg_sq->classifier_weight; f.vb = exp_avg_sq->classifier_bias;
    }
    f.n_partials = n_partials; f.n_task_wgs = wg2; f.fold_blocks = (int)((slab + kFoldCols - 1) / kFoldCols);
    f.l2 = coefs->L2_coe; f.l1 = coefs->L1_coe;
    f.inv_B = 1.0 / (double)batch_norm; f.inv_BD2 = 1.0 / ((double)batch_norm * (double)t.D * 2.0);
    f.losses6 = losses6;
    f.sched_table = sched ? reinterpret_cast<const SchedRow *>(sched->table) : nullptr;
    f.sched_n = sched ? sched->n : 0;
    const int grid2 = wg2 + f.fold_blocks + 1;
    if (shp.evl2) { f.slabs_ev = a.slabs_ev; f.n_partials_ev = plan->n_item_rounds / plan->item_rounds_per_task; }
    // wide rows / more than four environments (step_wide.hpp).  evl2: launch 2 = the item jobs alone (they produce embed_env's
    // partial slabs), launch 3 = the fold blocks alone
#define CALL_W1(LGV, NCV, VECV, EMAXV, EV2, BYE)                                                                  \
    do {                                                                                                          \
        if ((rc = ensure_lds(mstep_apply_wide_kernel<LGV, NCV, VECV, EMAXV, EV2>, lds2))) return rc;            \
        if (VECV && use_mm) {   /* full rows: the classifier as products over the workgroup's interactions (step_wide_mm.hpp) */ \
            const size_t ldsm = sizeof(float) * MGeo<LGV, NCV, EMAXV, EV2>::total;                                \
            if ((rc = ensure_lds(mstep_eval_mm_kernel<LGV, NCV, EMAXV, EV2, BYE>, ldsm))) return rc;             \
            if (wg1 > 0)                                                                                          \
                hipLaunchKernelGGL((mstep_eval_mm_kernel<LGV, NCV, EMAXV, EV2, BYE>), dim3(wg1), dim3(kThreads), ldsm, st, t, a1); \
        } else {                                                                                                  \
            if ((rc = ensure_lds(mstep_eval_wide_kernel<LGV, NCV, VECV, EMAXV, EV2, BYE>, lds1))) return rc;    \
            if (wg1 > 0)                                                                                          \
                hipLaunchKernelGGL((mstep_eval_wide_kernel<LGV, NCV, VECV, EMAXV, EV2, BYE>), dim3(wg1), dim3(kThreads), lds1, st, t, a1); \
        }                                                                                                         \
        if (profile_event && hipEventRecord((hipEvent_t)profile_event, st) != hipSuccess) return INVPREF_EINVAL; \
        if (!EV2) {                                                                                               \
            hipLaunchKernelGGL((mstep_apply_wide_kernel<LGV, NCV, VECV, EMAXV, EV2>), dim3(grid2), dim3(kThreads), lds2, st, t, a2, f); \
        } else {                                                                                                  \
            if (wg2 > 0)                                                                                          \
                hipLaunchKernelGGL((mstep_apply_wide_kernel<LGV, NCV, VECV, EMAXV, EV2>), dim3(wg2), dim3(kThreads), lds2, st, t, a2, f); \
            FoldArgs f3 = f;                                                                                      \
            f3.n_task_wgs = 0;                                                                                    \
            hipLaunchKernelGGL((mstep_apply_wide_kernel<LGV, NCV, VECV, EMAXV, EV2>), dim3(f.fold_blocks + 1), dim3(kThreads), fold_lds_bytes(), st, t, a2, f3); \
        }                                                                                                         \
    } while (0)
    /* (INVPREF_WEIGHTS_BY_ENV is a compile-time property of the full-row instances: step_wide.hpp) */
#define CALL_W(LGV, NCV, VECV, EMAXV, EV2)                                                                        \
    do {                                                                                                          \
        if (VECV && (flags & INVPREF_WEIGHTS_BY_ENV)) CALL_W1(LGV, NCV, VECV, EMAXV, EV2, (VECV)); else CALL_W1(LGV, NCV, VECV, EMAXV, EV2, false); \
    } while (0)
#define CALL_WE(LGV, NCV, VECV, EV2)                                                               \
    do {                                                                                           \
        if (emax == 8) CALL_W(LGV, NCV, VECV, 8, EV2); else CALL_W(LGV, NCV, VECV, 16, EV2);       \
    } while (0)
    // INVPREF_WIDE_MM=0 (diagnostics / A-B): the per-interaction classifier of step_wide.hpp for full rows too
    // (default: rows on 32 lanes -- D = 256: launch 1 14.4 vs 18.1 ms at 2^24 interactions; rows on 16 lanes measured level
    //  or slower, profiles/r05/EXPERIMENTS.md; INVPREF_WIDE_MM=1 takes the form for every full-row wide instance)
    const char *mm_env = getenv("INVPREF_WIDE_MM");   // (read per call: the tests switch forms inside one process)
    const bool mm_want = mm_env ? mm_env[0] == '1' : shp.lg == 32;
    const bool use_mm = mm_want && !pure && t.b != nullptr && (reinterpret_cast<uintptr_t>(t.W) & 15u) == 0;
    if (shp.wide) {
        // (the wide kernels' vector form is for FULL rows only -- factor_num 64 / 128 / 256: no clamps or selects behind a
        //  load; any other row length takes their element-wise form)
        if (vec && t.D == shp.dp) {
            if (shp.lg == 32) CALL_WE(32, 2, true, true); else if (shp.nc == 2) CALL_WE(16, 2, true, false); else CALL_WE(16, 1, true, false);
        } else {
            if (shp.lg == 32) CALL_WE(32, 2, false, true); else if (shp.nc == 2) CALL_WE(16, 2, false, false); else CALL_WE(16, 1, false, false);
        }
        return (int)hipGetLastError();
    }
#undef CALL_WE
#undef CALL_W
#undef CALL_W1
#define CALL(LGV, VECV, EMAXV)                                                                                  \
    do {                                                                                                        \
        if ((rc = ensure_lds(mstep_eval_kernel<LGV, VECV, EMAXV>, lds1))) return rc;                            \
        if ((rc = ensure_lds(mstep_apply_kernel<LGV, VECV, EMAXV>, lds2))) return rc;                           \
        if (wg1 > 0)                                                                                            \
            hipLaunchKernelGGL((mstep_eval_kernel<LGV, VECV, EMAXV>), dim3(wg1), dim3(kThreads), lds1, st, t, a1); \
        if (profile_event && hipEventRecord((hipEvent_t)profile_event, st) != hipSuccess) return INVPREF_EINVAL; \
        hipLaunchKernelGGL((mstep_apply_kernel<LGV, VECV, EMAXV>), dim3(grid2), dim3(kThreads), lds2, st, t, a2, f); \
    } while (0)
    // rows of exactly 64 floats: loads with nothing behind them (row4<VEC, FULL>); INVPREF_NO_FULL=1 (diagnostics) takes the
    // element-wise-guarded instances instead
    static const bool no_full = getenv("INVPREF_NO_FULL") != nullptr && getenv("INVPREF_NO_FULL")[0] == '1';
    const bool full = vec && t.D == 64 && !no_full;
    if (full) {
        if ((rc = ensure_lds(mstep_eval_kernel<16, true, 4, true>, lds1))) return rc;
        if ((rc = ensure_lds(mstep_apply_kernel<16, true, 4, true>, lds2))) return rc;
        if (wg1 > 0)
            hipLaunchKernelGGL((mstep_eval_kernel<16, true, 4, true>), dim3(wg1), dim3(kThreads), lds1, st, t, a1);
        if (profile_event && hipEventRecord((hipEvent_t)profile_event, st) != hipSuccess) return INVPREF_EINVAL;
        hipLaunchKernelGGL((mstep_apply_kernel<16, true, 4, true>), dim3(grid2), dim3(kThreads), lds2, st, t, a2, f);
    } else if (!vec) {
        CALL(16, false, 4);
    } else {
        CALL(16, true, 4);
    }
#undef CALL
    return (int)hipGetLastError();
}

AdamScalars adam_scalars(int64_t step, double lr, double beta1, double beta2, double eps) {
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamScalars ad;
    ad.step_size = (float)(lr / bc1);
    ad.bc2_sqrt = (float)sqrt(bc2);
    ad.w1 = (float)(1.0 - beta1);
    ad.b2 = (float)beta2;
    ad.w2 = (float)(1.0 - beta2);
    ad.eps = (float)eps;
    return ad;
}

#include "step_alt.hpp"

}  // namespace

extern "C" {

size_t invpref_alt_workspace_bytes(const InvPrefTables *tables, int32_t n_cap, int32_t partials_cap) {
    if (!tables || n_cap < 0 || partials_cap < 0) return 0;
    return alt_flags_offset(n_cap, partials_cap) + kAltTailBytes;
}

size_t invpref_alt_error_offset(const InvPrefTables *tables, int32_t n_cap, int32_t partials_cap) {
    (void)tables;
    return alt_flags_offset(n_cap, partials_cap) + 63 * sizeof(int);
}

int invpref_alt_supported(const InvPrefTables *tables) {
    if (!tables || tables->factor_num <= 0 || tables->env_num <= 0) return 0;
    return tables->factor_num <= 64 && tables->env_num <= 4;
}

int invpref_mstep_alt_hip(const InvPrefTables *tables, const InvPrefTables *exp_avg, const InvPrefTables *exp_avg_sq,
                          const InvPrefAltPlan *plan, const int64_t *envs, const float *sample_weights,
                          int64_t batch_norm, int64_t batch_norm_prev, const InvPrefCoefs *coefs, uint32_t flags,
                          float *losses6_prev, int64_t step, double lr, double beta1, double beta2, double eps,
                          const InvPrefAdamSchedule *sched, void *workspace, size_t workspace_bytes, int32_t n_cap,
                          int32_t partials_cap, int32_t parity, void *stream) {
    return launch_alt(tables, exp_avg, exp_avg_sq, plan, envs, sample_weights, batch_norm, batch_norm_prev, coefs, flags,
                      losses6_prev, step, lr, beta1, beta2, eps, sched, workspace, workspace_bytes, n_cap, partials_cap, parity,
                      (hipStream_t)stream);
}

size_t invpref_rows_workspace_bytes(const InvPrefTables *tables, const InvPrefRowPlan *plan) {
    if (check_tables(tables, tables && !tables->embed_user_env_aware) || !plan || plan->n < 0 || plan->n_user_rounds < 0 ||
        plan->user_rounds_per_task <= 0)
        return 0;
    if (plan->item_rounds_per_task <= 0 || plan->n_item_rounds < 0) return 0;
    const Shape shp = shape_of((int)tables->factor_num, (int)tables->env_num);
    if (plan->lanes_per_group != shp.lg) return 0;
    return sizeof(float) * workspace_floats(plan, shp);
}

int invpref_rows_lanes_per_group(const InvPrefTables *tables) {
    if (!tables || tables->factor_num <= 0 || tables->factor_num > INVPREF_MAX_FACTORS) return INVPREF_EUNSUPPORTED;
    return lanes_of((int)tables->factor_num);
}

int invpref_mstep_rows_grad_hip(const InvPrefTables *tables, const InvPrefTables *grads, const InvPrefRowPlan *plan,
                                const int64_t *envs, const float *scores, const float *sample_weights,
                                int64_t batch_norm, const InvPrefCoefs *coefs, uint32_t flags, float *losses6,
                                void *workspace, size_t workspace_bytes, void *stream) {
    return launch_step(tables, plan, envs, scores, sample_weights, batch_norm, coefs, flags, losses6, workspace,
                       workspace_bytes, (hipStream_t)stream, 0, grads, nullptr, nullptr, nullptr, AdamScalars{}, nullptr,
                       nullptr);
}

int invpref_mstep_rows_grad_sched_hip(const InvPrefTables *tables, const InvPrefTables *grads, const InvPrefRowPlan *plan,
                                      const int64_t *envs, const float *scores, const float *sample_weights,
                                      int64_t batch_norm, const InvPrefCoefs *coefs, uint32_t flags, float *losses6,
                                      const InvPrefAdamSchedule *sched, void *workspace, size_t workspace_bytes,
                                      void *stream) {
    if (!sched) return INVPREF_EINVAL;
    return launch_step(tables, plan, envs, scores, sample_weights, batch_norm, coefs, flags, losses6, workspace,
                       workspace_bytes, (hipStream_t)stream, 0, grads, nullptr, nullptr, nullptr, AdamScalars{}, sched,
                       nullptr);
}

int invpref_mstep_rows_adam_hip(const InvPrefTables *tables, const InvPrefTables *new_tables,
                                const InvPrefTables *exp_avg, const InvPrefTables *exp_avg_sq,
                                const InvPrefRowPlan *plan, const int64_t *envs, const float *scores,
                                const float *sample_weights, int64_t batch_norm, const InvPrefCoefs *coefs,
                                uint32_t flags, float *losses6, int64_t step, double lr, double beta1, double beta2,
                                double eps, void *workspace, size_t workspace_bytes, void *stream) {
    if (step < 1) return INVPREF_EINVAL;
    return launch_step(tables, plan, envs, scores, sample_weights, batch_norm, coefs, flags, losses6, workspace,
                       workspace_bytes, (hipStream_t)stream, 1, nullptr, new_tables, exp_avg, exp_avg_sq,
                       adam_scalars(step, lr, beta1, beta2, eps), nullptr, nullptr);
}

int invpref_mstep_rows_adam_profiled_hip(const InvPrefTables *tables, const InvPrefTables *new_tables,
                                         const InvPrefTables *exp_avg, const InvPrefTables *exp_avg_sq,
                                         const InvPrefRowPlan *plan, const int64_t *envs, const float *scores,
                                         const float *sample_weights, int64_t batch_norm, const InvPrefCoefs *coefs,
                                         uint32_t flags, float *losses6, int64_t step, double lr, double beta1,
                                         double beta2, double eps, void *workspace, size_t workspace_bytes, void *stream,
                                         void *mid_event) {
    if (step < 1) return INVPREF_EINVAL;
    return launch_step(tables, plan, envs, scores, sample_weights, batch_norm, coefs, flags, losses6, workspace,
                       workspace_bytes, (hipStream_t)stream, 1, nullptr, new_tables, exp_avg, exp_avg_sq,
                       adam_scalars(step, lr, beta1, beta2, eps), nullptr, mid_event);
}

/* host helper: the per-step Adam scalars exactly as invpref_adam_hip / invpref_mstep_rows_adam_hip form
 * them from (step, lr, betas, eps); table[i] belongs to step first_step + i. */
int invpref_adam_schedule_fill(float *host_table, int64_t first_step, int64_t n, double lr, double beta1, double beta2,
                               double eps) {
    if (!host_table || first_step < 1 || n < 0) return INVPREF_EINVAL;
    for (int64_t i = 0; i < n; i++) {
        const AdamScalars ad = adam_scalars(first_step + i, lr, beta1, beta2, eps);
        float *r = host_table + 8 * i;
        r[0] = ad.step_size; r[1] = ad.bc2_sqrt; r[2] = ad.w1; r[3] = ad.b2; r[4] = ad.w2; r[5] = ad.eps;
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
    return launch_step(tables, plan, envs, scores, sample_weights, batch_norm, coefs, flags, losses6, workspace,
                       workspace_bytes, (hipStream_t)stream, 1, nullptr, new_tables, exp_avg, exp_avg_sq, AdamScalars{},
                       sched, nullptr);
}

}  // extern "C"

// (diagnostic, not declared in the header) resident workgroups per CU of the smallest launch-1 instance
extern "C" int invpref_debug_eval_occupancy(void) {
    int n = -1;
    const size_t lds = eval_lds_bytes(shape_of(64, 4));
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mstep_eval_kernel<16, true, 4, true>, kThreads, lds) != hipSuccess) return -1;
    return n * 1000 + (int)(lds / 1024);
}
