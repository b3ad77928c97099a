// step_wide.hpp -- the planned M-step for rows of more than 64 floats and / or more than four environments
// (MovieLens: D = 128, E = 8; MIND: D = 256, E = 16).  Included by invpref_step.hip inside its anonymous namespace.
//
// Round 3 gave every row ONE float4 per lane (32 / 64 lanes per row): an interaction then occupied half a wave or a
// whole one, and the per-interaction chain -- row reductions, softmax, the loss terms, some 300 instructions -- was paid
// per wave; the E x D outer products (gradients of the classifier and of embed_env) sat in registers or went through
// LDS records with a barrier per interaction.  Measured: the same rows cost 41 -> 63 -> 204 us on 16 / 32 / 64 lanes.
// Here:
//   * a row lives on 16 lanes x 2 float4 (D <= 128) or 32 lanes x 2 float4 (D <= 256): four (two) interactions
//     share a wave's instruction stream again;
//   * the two outer products  dW += gz (x) x  and  dEv += onehot(env) (x) o  are GEMM-shaped across the wave's
//     interactions and run on the matrix cores: v_mfma_f32_16x16x4_f32, fp32 in, fp32 accumulate -- bit for bit an
//     fmaf chain over the wave's (up to) four interactions, so the step stays bitwise reproducible.  The operands
//     need no data movement: the A operand A[m][k] sits on lane m + 16 k, which IS "class m of the interaction on
//     lane quarter k" in the class-per-lane softmax layout; the B operand B[k][n] on lane n + 16 k is component c of
//     float4 chunk j of that interaction's row -- one MFMA per (chunk, component); the accumulators (4 registers per
//     tile, classes on rows) stay in registers for the whole task and meet in LDS once, at its end.
//     (32 lanes per row: a lane quarter holds HALF a row, so each tile takes the A operand masked to one half.)
//   * D <= 128: both products in launch 1 (64 accumulator registers).  D <= 256: the classifier's in launch 1, embed_env's
//     in launch 2's item jobs (EVL2; they hold Qa[v], gather Pa[u] and read g_q / env from the record: pull form only),
//     whose partial slabs a third, tiny launch folds.
// Arithmetic of one interaction: eval_wide() == eval_interaction() (models.py:307-326, :206-209; train.py:108-153).
#pragma once

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef WIDE_UE
#define WIDE_UE 2      // interactions in flight per group (launch 1)
#endif
#ifndef WIDE_WAVES_16_2_8
#define WIDE_WAVES_16_2_8 2
#endif
#ifndef WIDE_UE_16_2_8
#define WIDE_UE_16_2_8 2
#endif
#ifndef WIDE_U_EVL2_16
#define WIDE_U_EVL2_16 1   // launch 2 of the D <= 256, E = 16 instance (its slots hold 16 class gradients each)
#endif
#ifndef WIDE_FENCE
#define WIDE_FENCE 1   // scheduling fences inside the classifier loops of two-chunk rows (register pressure)
#endif
#ifndef WIDE_STEP_FENCE_MASK
#define WIDE_STEP_FENCE_MASK 0   // what may cross the fences between two lock-step iterations (0x8: the MFMA block)
#endif
#ifndef WIDE_FIRST_FROM_DESC
#define WIDE_FIRST_FROM_DESC 1   // 16-lane rows: a slice's first gather from the descriptor's registers instead of through list_at (32-lane rows have no registers for the descriptor's second half: +3-4 %)
#endif
#ifndef WIDE_PUSH_DEPTH
#define WIDE_PUSH_DEPTH 4   // (measured at the MovieLens shape: 66.7 -> 66.0 us against 2)
#endif
#ifndef WIDE_MFMA_FENCE
#define WIDE_MFMA_FENCE 1   // (A/B knob) a fence between the row updates / stores and the MFMA block
#endif
#ifndef WIDE_FENCE_MASK
// what may still cross a fence of the classifier loops: ALU (0x1 | VALU 0x2 | SALU 0x4 | transcendental 0x400) -- the
// loss chains interleave with the LDS waits -- but no memory instruction: the W-row reads stay where they are
#define WIDE_FENCE_MASK 0x407
#endif

// per-instance launch-1 configuration: workgroups per CU the kernel is compiled for (registers: 512 / waves per SIMD)
// and the interactions each group keeps in flight
template <int LG, int NC, int EMAX>
struct WideCfg {
    static constexpr int WAVES = (LG == 16 && NC == 2 && EMAX == 8) ? WIDE_WAVES_16_2_8 : 2;
    static constexpr int UE = (LG == 16 && NC == 2 && EMAX == 8) ? WIDE_UE_16_2_8 : ((NC == 2 && LG == 16 && EMAX == 16) ? 1 : WIDE_UE);
};

template <int LG, int NC, int EMAX>
struct WGeo {
    static constexpr int NG = kThreads / LG;          // groups (rows in flight) per workgroup
    static constexpr int DP = 4 * LG * NC;            // padded row length
    static constexpr int RS = 4 + EMAX;               // floats per record: g_p, g_q, env bits, 0, gz[EMAX]
    static constexpr int SLAB = 2 * EMAX * DP + EMAX + kLossSlots;   // dEv | dW | db | loss sums
    static constexpr int HALVES = LG / 16;            // 16-lane pieces of a group = MFMA k slots a row spans
    static constexpr int TILES = NC * HALVES;         // column tiles of a table: 16 float4 columns x 16 classes each
    // LDS of launch 1 (floats)
    static constexpr int sEv = 0, sW = EMAX * DP, sb = 2 * EMAX * DP, scw = sb + 16, slots = scw + 16;   // scw: [16] class weights (INVPREF_WEIGHTS_BY_ENV)
    static constexpr int gzs = slots + NG * 2 * DP;                  // [2][NG][EMAX + 4] class gradients of a group
    static constexpr int live_end = gzs + 2 * NG * (EMAX + 4);
    // task end: the waves' accumulator tiles meet in NIMG LDS images laid over everything above (nobody reads it any
    // more); wave w adds into image w % NIMG in turn w / NIMG.  EVL2 = true halves the image (classifier only).
    template <bool EVL2> struct Img {
        static constexpr int LEN = (EVL2 ? 1 : 2) * EMAX * DP;
        static constexpr int N = 4 * LEN <= 10240 ? 4 : (2 * LEN <= 10240 ? 2 : 1);
        static constexpr int tail = (N * LEN > live_end ? N * LEN : live_end);   // [kWaves][EMAX + kLossSlots] db | loss sums
        static constexpr int total = tail + kWaves * (EMAX + kLossSlots);
    };
    // LDS of launch 2: sEv | sW | slots
    static constexpr int apply_total = 2 * EMAX * DP + NG * 2 * DP;
};

// diagnostic build (-DWIDE_DIAG_TRACE, tools/wide_trace.py): shader-clock stamps of the first steps of one wave, written
// behind the phase stamps of the stamp buffer
#ifdef WIDE_DIAG_TRACE
#define WTRACE(tag)                                                                                           \
    do {                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (a.stamps && blockIdx.x == 40 && threadIdx.x == 0 && wtrace_n < 120) {                             \
            a.stamps[100000 + wtrace_n] = ((unsigned long long)(tag) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffffull); \
            wtrace_n++;                                                                                       \
        }                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    } while (0)
// (inside eval_wide: the same stamps through a context handed in by the caller; tags 10 ..)
struct WTraceCtx { unsigned long long *buf; int *n; bool on; };
#define ETRACE(tag)                                                                                           \
    do {                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (tc.on && *tc.n < 120) {                                                                           \
            tc.buf[100000 + *tc.n] = ((unsigned long long)(tag) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffffull); \
            (*tc.n)++;                                                                                        \
        }                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    } while (0)
#define ETRACE_PARAM , WTraceCtx tc
#define ETRACE_ARG , WTraceCtx{a.stamps, &wtrace_n, a.stamps && blockIdx.x == 40 && threadIdx.x == 0}
#else
#define WTRACE(tag) do { } while (0)
#define ETRACE(tag) do { } while (0)
#define ETRACE_PARAM
#define ETRACE_ARG
#endif

// VEC in this file = FULL rows: factor_num == DP (64 / 128 / 256) and 16-byte aligned tables.  A full row is loaded with
// nothing behind the load -- no clamp, no zero-select: an instruction on the loaded value right behind the load makes the
// compiler wait for it on the spot, which turned every prefetch of a lock-step iteration into a blocking round trip.
// Other row lengths take the element-wise form (VEC = false).
template <int LG, int NC, bool VEC>
__device__ __forceinline__ void load_row(float4 (&r)[NC], const float *__restrict__ base, int row, int D, int lg) {
#pragma unroll
    for (int j = 0; j < NC; j++) {
        if (VEC) {
            constexpr unsigned DPB = 16u * LG * NC;   // bytes of a row
            r[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + ((unsigned)row * DPB + 16u * (unsigned)(lg + LG * j)));
        } else {
            r[j] = row4<false>(base, row, D, lg + LG * j);
        }
    }
}
template <int LG, int NC, bool VEC, int MODE = 0>
__device__ __forceinline__ void store_row(float *__restrict__ base, int row, int D, int lg, const float4 (&r)[NC]) {
#pragma unroll
    for (int j = 0; j < NC; j++) {
        if (VEC) {
            constexpr unsigned DPB = 16u * LG * NC;
            float4 *dst = reinterpret_cast<float4 *>(reinterpret_cast<char *>(base) + ((unsigned)row * DPB + 16u * (unsigned)(lg + LG * j)));
            if (MODE == 0) *dst = r[j];
            else {
                v4f val = {r[j].x, r[j].y, r[j].z, r[j].w};
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(val) : "memory");
            }
        } else {
            put4<false, MODE>(base, row, D, lg + LG * j, r[j]);
        }
    }
}
template <int LG, int NC>
__device__ __forceinline__ void lds_row(float4 (&r)[NC], const float *tab, int e, int lg) {
    constexpr int DP = 4 * LG * NC;
#pragma unroll
    for (int j = 0; j < NC; j++) r[j] = *reinterpret_cast<const float4 *>(tab + e * DP + 4 * (lg + LG * j));
}

// ---- forward + analytic backward of ONE interaction on a lane group of LG lanes x NC float4 (M-step arithmetic:
// hardware exp / log / rcp).  One class per lane: lane c of the group (c < E <= 16) ends up with class c's logit.
template <int NC>
struct WEval {
    float g_p, g_q, li, le, lcls, gz_lane, gz_all;
    float4 x[NC], gx[NC];   // x = Pu*Qi ; gx = sum_c gz_c W_c
};
template <int LG, int NC, int EMAX, int C>
__device__ __forceinline__ void gx_classes(float4 (&gx)[NC], float gzl, const float *sW, int lg) {
    if constexpr (C < EMAX) {
        constexpr int DP = 4 * LG * NC;
        if (WIDE_FENCE && NC > 1 && (C & 3) == 0) __builtin_amdgcn_sched_barrier(WIDE_FENCE_MASK);
        const float g = dpp_move<0x150 + C>(gzl);   // row_share:C
#pragma unroll
        for (int j = 0; j < NC; j++) {
            const float4 w = *reinterpret_cast<const float4 *>(sW + C * DP + 4 * (lg + LG * j));
            if (C == 0) gx[j] = f4zero();
            f4fma(gx[j], g, w);
        }
        gx_classes<LG, NC, EMAX, C + 1>(gx, gzl, sW, lg);
    }
}
template <int LG, int NC, int EMAX>
__device__ __forceinline__ void eval_wide(WEval<NC> &o, const float4 (&pu)[NC], const float4 (&qi)[NC],
                                          const float4 (&pa)[NC], const float4 (&qa)[NC], const float4 (&ev)[NC],
                                          const float *sW, const float *sb, float bias_l, float *gzs, bool wr_gzs, int E, int e, float y, float cw_rec,
                                          float cw_cls, const StepScalars &k, bool implicit, bool pure, int lg, bool has ETRACE_PARAM) {
    // `has` = false (an empty slot of a lock-step iteration): the arithmetic runs on the slot's stale -- finite -- rows
    // and every gradient scalar is forced to zero, so that everything downstream contributes nothing; no branch.
    constexpr int DP = 4 * LG * NC;
    float ps = 0.f, qs = 0.f;
#pragma unroll
    for (int j = 0; j < NC; j++) {   // (the first term starts the sum: no 0 + x instruction)
        o.x[j] = f4mul(pu[j], qi[j]);
        const float pj = (o.x[j].x + o.x[j].y) + (o.x[j].z + o.x[j].w), qj = dot4(f4mul(pa[j], qa[j]), ev[j]);
        ps = j ? ps + pj : pj;
        qs = j ? qs + qj : qj;
    }
    const float p = group_sum<LG>(ps);
    const float q = group_sum<LG>(qs);
    ETRACE(10);
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
    if (!has) o.g_p = o.g_q = o.li = o.le = 0.f;
    ETRACE(11);
    if (pure) {   // PureMF: no classifier (the zeros are written on THIS path only: no register moves on the other)
        o.lcls = 0.f;
        o.gz_lane = o.gz_all = 0.f;
#pragma unroll
        for (int j = 0; j < NC; j++) o.gx[j] = f4zero();
        return;
    }
#ifdef WIDE_DIAG_NOCLS
    o.lcls = o.gz_lane = o.gz_all = 0.f;
#pragma unroll
    for (int j = 0; j < NC; j++) o.gx[j] = f4zero();
    return;
#endif
    // all EMAX class dot products per lane (rows c >= E are staged as zeros), then ONE reduce-scatter butterfly: lane l of
    // the group ends up with the logit of class l & (EMAX - 1)
    float part[EMAX];
#pragma unroll
    for (int c = 0; c < EMAX; c++) {
        // (a scheduling fence per four classes: left alone the compiler requests every W row of the loop up front and
        //  the kernel spills; two waves per SIMD cover the LDS latency instead)
        if (WIDE_FENCE && NC > 1 && (c & 3) == 0) __builtin_amdgcn_sched_barrier(WIDE_FENCE_MASK);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NC; j++) {
            const float dj = dot4(o.x[j], *reinterpret_cast<const float4 *>(sW + c * DP + 4 * (lg + LG * j)));
            s = j ? s + dj : dj;
        }
        part[c] = s;
    }
    if (WIDE_FENCE && NC > 1) __builtin_amdgcn_sched_barrier(WIDE_FENCE_MASK);
    ETRACE(12);
    const float zred = group_sum_above<LG, EMAX>(class_butterfly<EMAX>(part, lg), lg);
    // (16-lane rows: the lane's class bias sits in a register; 32-lane rows have none to spare and read it from LDS)
    const float zmine = lg < E ? zred + (LG == 16 ? bias_l : sb[lg & (EMAX - 1)]) : -__builtin_inff();
    const float mxl = group_max<LG>(zmine);
    const float ez = lg < E ? f_exp(zmine - mxl) : 0.f;
    const float rsel = f_rcp(group_sum<LG>(ez));
    const float gzl = (lg < E && has) ? k.cc * cw_cls * (ez * rsel - (lg == e ? 1.f : 0.f)) : 0.f;
    o.gz_lane = gzl;
    // the picked class's loss term stays on ITS lane (the task's loss sums run over every lane)
    // (the reference's log_softmax form, models.py:206-209: finite for any finite logits)
    // (log(sum) = -log(1 / sum): the reciprocal is at hand, no register for the sum itself)
    o.lcls = (lg == e && has) ? -f_log(rsel) - (zmine - mxl) : 0.f;
    if constexpr (LG == 16) {
        // gx = sum_c gz_c W_c, classes in order; gz_c = lane c of the lane's own 16-lane row (DPP row_share: no LDS round
        // trip between the softmax and the backward).  The group's LDS words keep a copy of gz only for the pull record.
        o.gz_all = gzl;
        if (wr_gzs && lg < EMAX) gzs[lg] = gzl;
        ETRACE(13);
        ETRACE(14);
        gx_classes<LG, NC, EMAX, 0>(o.gx, gzl, sW, lg);
    } else {
        // rows on 32 lanes sit at the register limit (every live value more is a spill per round): gz goes through the
        // group's LDS words -- in-order LDS operations of one wave -- to the backward, the record and the MFMA A operand
        if (lg < EMAX) gzs[lg] = gzl;
        ETRACE(13);
        WAVE_LDS_FENCE();
        ETRACE(14);
        o.gz_all = (lg & 15) < EMAX ? gzs[lg & 15] : 0.f;
#pragma unroll
        for (int c4 = 0; c4 < EMAX; c4 += 4) {
            if (WIDE_FENCE && NC > 1) __builtin_amdgcn_sched_barrier(WIDE_FENCE_MASK);
            const float4 g4 = *reinterpret_cast<const float4 *>(gzs + c4);
#pragma unroll
            for (int j = 0; j < NC; j++) {
                const float *wr = sW + c4 * DP + 4 * (lg + LG * j);
                if (c4 == 0) o.gx[j] = f4zero();
                f4fma(o.gx[j], g4.x, *reinterpret_cast<const float4 *>(wr));
                f4fma(o.gx[j], g4.y, *reinterpret_cast<const float4 *>(wr + DP));
                f4fma(o.gx[j], g4.z, *reinterpret_cast<const float4 *>(wr + 2 * DP));
                f4fma(o.gx[j], g4.w, *reinterpret_cast<const float4 *>(wr + 3 * DP));
            }
        }
    }
}

// acc[tile][component] += A (x) B over the wave's lane quarters: tile (j, h) covers the float4 columns
// (lane & 15) + 16 h + LG j; a row on 32 lanes spans two quarters, so the A operand is masked to the quarter's half.
template <int LG, int NC>
__device__ __forceinline__ void outer_mfma(f32x4 (&acc)[NC * (LG / 16)][4], float a, const float4 (&b)[NC], int lane) {
    constexpr int HALVES = LG / 16;
#pragma unroll
    for (int j = 0; j < NC; j++) {
#pragma unroll
        for (int h = 0; h < HALVES; h++) {
            const float ah = (HALVES == 1 || ((lane >> 4) & 1) == h) ? a : 0.f;
            f32x4 (&t)[4] = acc[j * HALVES + h];
            t[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah, b[j].x, t[0], 0, 0, 0);
            t[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah, b[j].y, t[1], 0, 0, 0);
            t[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah, b[j].z, t[2], 0, 0, 0);
            t[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah, b[j].w, t[3], 0, 0, 0);
        }
    }
}
// EMAX = 8 pads the class dimension of a tile to 16 rows: rows 8 .. 15 of the SAME accumulators then take embed_env's
// product (its one-hot A operand moved to lanes 8 .. 15 of the quarter) -- half the accumulator registers
template <int EMAX, bool EVL2>
struct Shared { static constexpr bool value = EMAX == 8 && !EVL2; };

// one wave's accumulator tiles into (first = true: over) the workgroup's [EMAX][DP] LDS image: lane l, register r of a
// tile hold class 4 (l >> 4) + r at float4 column (l & 15) + 16 h + LG j -- the four components are the four MFMAs
template <int LG, int NC, int EMAX, bool SHARED = false>
__device__ __forceinline__ void tiles_to_lds(float *img, const f32x4 (&acc)[NC * (LG / 16)][4], int lane, bool first) {
    // (SHARED: img = embed_env's image, the classifier's follows it: accumulator rows 0 .. 7 are classifier classes,
    //  rows 8 .. 15 environments)
    constexpr int HALVES = LG / 16, DP = 4 * LG * NC;
#pragma unroll
    for (int j = 0; j < NC; j++) {
#pragma unroll
        for (int h = 0; h < HALVES; h++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int cls = 4 * (lane >> 4) + r;
                if (SHARED || cls < EMAX) {
                    const f32x4 (&t)[4] = acc[j * HALVES + h];
                    const int rowi = SHARED ? (cls < 8 ? 8 + cls : cls - 8) : cls;   // image row: [dEv: 8 rows][dW: 8 rows]
                    float4 *dst = reinterpret_cast<float4 *>(img + rowi * DP + 4 * ((lane & 15) + 16 * h + LG * j));
                    float4 v = make_float4(t[0][r], t[1][r], t[2][r], t[3][r]);
                    if (!first) { const float4 cur = *dst; v.x += cur.x; v.y += cur.y; v.z += cur.z; v.w += cur.w; }
                    *dst = v;
                }
            }
        }
    }
}

// =====================================================================================
// launch 1: rounds of USER jobs
// =====================================================================================
// BYENV (full-row instances): INVPREF_WEIGHTS_BY_ENV at COMPILE time -- the weight of an interaction is class_weights[env] from
// LDS and a gather slot carries no weight register (at 250+ registers a run-time switch put loop-carried values into
// scratch memory); the element-wise instances (VEC = false) take the flag at run time.
template <int LG, int NC, bool VEC, int EMAX, bool EVL2, bool BYENV = false>
__device__ __forceinline__ void user_task_wide(const DevTables &t, const StepArgs &a, int r0, int nr, int slab_index, float *lds) {
    using G = WGeo<LG, NC, EMAX>;
    constexpr int NG = G::NG, DP = G::DP, RS = G::RS, TILES = G::TILES;
    // interactions in flight per group: what the instance's register budget allows (WideCfg)
    constexpr int UE = WideCfg<LG, NC, EMAX>::UE;
    using IM = typename G::template Img<EVL2>;
    float *sEv = lds + G::sEv, *sW = lds + G::sW, *sb = lds + G::sb, *scw = lds + G::scw, *slots = lds + G::slots, *tail = lds + IM::tail;
    const int lg = threadIdx.x & (LG - 1), grp = threadIdx.x / LG, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool implicit = a.flags & INVPREF_IMPLICIT;
    const bool rw_rec = a.flags & INVPREF_REWEIGHT_REC, rw_cls = a.flags & INVPREF_REWEIGHT_CLS;
    const bool by_env = BYENV || (!VEC && (a.flags & INVPREF_WEIGHTS_BY_ENV));   // weight = class_weights[env], staged with the small tables
    const bool reg_env = a.flags & INVPREF_REG_ENV_EMBED;
    const bool pure = a.flags & INVPREF_PURE_MF;
    const bool push = a.push_slot != nullptr;
    StepScalars k = a.k;
    if (a.sched_state) {  // scheduled alpha (train.py:214-217) under graph replay
        const float al = sched_slot_ptr(a.sched_state, a.sched_slot)->alpha;
        if (al == al) k.alpha = al;
    }
    const AdamScalars ad = a.sched_state ? sched_slot_ptr(a.sched_state, a.sched_slot)->ad : a.ad;

    STAMP(0);
    // both halves of the descriptor: row | meta | list range or inline interactions | a list slice's leading interaction
    int4 d = a.desc[(r0 * NG + grp) * 2], d2 = a.desc[(r0 * NG + grp) * 2 + 1];
    STAMP(1);

    constexpr bool SH = Shared<EMAX, EVL2>::value;   // one accumulator set for both products
    f32x4 accW[TILES][4], accE[(EVL2 || SH) ? 1 : TILES][4];
#pragma unroll
    for (int i = 0; i < TILES; i++)
#pragma unroll
        for (int c = 0; c < 4; c++) accW[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < ((EVL2 || SH) ? 1 : TILES); i++)
#pragma unroll
        for (int c = 0; c < 4; c++) accE[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dBacc = 0.f;   // lane c of every group: sum of gz_c
    float bias_l = 0.f;
    const bool wr_gzs = !push;   // (the pull record and nothing else reads gz back from the group's LDS words)
    float accLi = 0.f, accLe = 0.f, accLc = 0.f, accL2 = 0.f, accL1 = 0.f;
    int it_total = 0;    // parity of the gz words
    float *slab = a.slabs + (int64_t)slab_index * G::SLAB;
#ifdef WIDE_DIAG_TRACE
    int wtrace_n = 0;
#endif

    for (int r = r0; r < r0 + nr; r++) {
        const int4 dd = d, dd2 = d2;
        const int row = dd.x, meta = dd.y;
        const bool active = row >= 0, leader = meta & 1;
        const int slices = (meta >> 1) & 31, mode = (meta >> 6) & 7;
        const int nsmp = active ? (mode == 7 ? dd.w - dd.z : mode) : 0;
        const int iters = a.round_iters[r];   // the round's longest slice: the loop (and its MFMAs) is workgroup-uniform
        if (r == r0 + STAMP_ROUND) STAMP(2);
        // The slice's interactions come from the sorted list (mode 7) or, up to two of them, from the descriptor itself.
        // Every load of the loop below is UNCONDITIONAL -- list indices are clamped into the slice, so a finished slice
        // re-reads its last interaction (cache hits) and an idle slot reads entry 0 -- because a load under a divergent
        // branch is waited for at the join, and a wait inside a lock-step iteration stops the whole prefetch pipeline.
        // (the inline form is read back from the descriptor's own words -- 3 ints per interaction from word 2 on -- so that
        //  both forms are ONE load from a selected address, with nothing to select behind it)
        const int lo = dd.z, hi1 = max(dd.w - 1, dd.z);
        const int *dwords = reinterpret_cast<const int *>(a.desc + (r * NG + grp) * 2);
        auto list_at = [&](int sidx) {
#ifdef WIDE_DIAG_NOLIST   // (what-if build: no list round trip in front of the gathers -- ids made up from the slice bounds)
            return USample{(lo + sidx) & 1023, min(lo + sidx, hi1), 0.f};
#endif
            const int *src = mode == 7 ? reinterpret_cast<const int *>(a.ulist + min(lo + sidx, hi1)) : dwords + 2 + 3 * min(sidx, 1);
            return USample{src[0], src[1], __builtin_bit_cast(float, src[2])};
        };
        float4 oi[NC], oe[NC], gi[NC], ge[NC];
#pragma unroll
        for (int j = 0; j < NC; j++) oi[j] = oe[j] = gi[j] = ge[j] = f4zero();
        {   // (an idle slot reads row 0 rather than branching around the loads)
            const int rowc = active ? row : 0;
            load_row<LG, NC, VEC>(oi, t.Pu, rowc, t.D, lg);
            if (!pure) load_row<LG, NC, VEC>(oe, t.Pa, rowc, t.D, lg);
        }
        struct Slot {
            float4 qi[NC], qa[NC];
            int e, cs;   // cs: the interaction's slot in the item order (where its record / contribution rows go)
            float y, w;
        };
        Slot sl[UE];
        USample idn[UE];
        auto gather = [&](Slot &q, const USample &sm, int sidx) {
            q.y = sm.y;
#ifdef WIDE_DIAG_HOT   // (what-if build: every gather hits the same few rows -- what the launch costs without gather latency)
            const int oth = sm.oth & 15;
#else
            const int oth = sm.oth;
#endif
            load_row<LG, NC, VEC>(q.qi, t.Qi, oth, t.D, lg);
            const unsigned pso = (unsigned)sm.ps;   // (32-bit offsets: one address register, no 64-bit pair to copy into)
            if (!pure) {
                load_row<LG, NC, VEC>(q.qa, t.Qa, oth, t.D, lg);
                q.e = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(a.envs) + pso * 8u);   // low word of the int64 id
            }
            if ((rw_rec || rw_cls) && !by_env) q.w = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.weights) + pso * 4u);
            // the slot: word 3 of the list entry the ids came from or, inline form, rec_slot[position] -- one load, selected address
            const int *sp = mode == 7 ? reinterpret_cast<const int *>(a.ulist + min(lo + sidx, hi1)) + 3
                                      : reinterpret_cast<const int *>(reinterpret_cast<const char *>(a.rec_slot) + pso * 4u);
            q.cs = STEP_SLOT_FROM_LIST ? *sp : *reinterpret_cast<const int *>(reinterpret_cast<const char *>(a.rec_slot) + pso * 4u);
        };
        // the slice's FIRST interaction is in the descriptor's registers either way (inline form: words 2 .. 4, list form:
        // words 4 .. 6), so its gather leaves as soon as the descriptor is here -- not one list round trip later
        const USample first = mode == 7 ? USample{dd2.x, dd2.y, __builtin_bit_cast(float, dd2.z)}
                                        : USample{dd.z, dd.w, __builtin_bit_cast(float, dd2.x)};
#pragma unroll
        for (int j = 0; j < UE; j++) {
#pragma unroll
            for (int c = 0; c < NC; c++) sl[j].qi[c] = sl[j].qa[c] = f4zero();
            sl[j].e = sl[j].cs = 0;
            sl[j].y = 0.f;
            sl[j].w = 1.f;
            if (WIDE_FIRST_FROM_DESC && LG == 16 && j == 0) gather(sl[j], first, 0);
            else gather(sl[j], list_at(j), j);
        }
#pragma unroll
        for (int j = 0; j < UE; j++) idn[j] = list_at(UE + j);
        if (r == r0) {
            // the two small tables are staged HERE, behind the first round's gathers: a load -> store loop in front of
            // them would put the tables' round trip ahead of the rows' on the task's critical chain
            stage_small(sEv, t.Ev, t.E, t.D, EMAX, DP);
            stage_small(sW, t.W, t.E, t.D, EMAX, DP);
            // (32-bit offsets from the scalar bases: a 64-bit address pair per table, formed at the top of the task and held
            //  across the rounds, went to scratch memory at this register pressure)
            const unsigned so = (threadIdx.x < (unsigned)t.E ? threadIdx.x : 0u) * 4u;
            if (threadIdx.x < 16) sb[threadIdx.x] = (threadIdx.x < t.E && t.b) ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(t.b) + so) : 0.f;
            if (threadIdx.x < 16) scw[threadIdx.x] = (by_env && threadIdx.x < t.E) ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.weights) + so) : 1.f;
            __syncthreads();
            if (LG == 16) bias_l = sb[lg & (EMAX - 1)];   // the lane's class bias, once: an LDS read per interaction sat on the softmax chain
        }
        if (r == r0 + STAMP_ROUND) STAMP(3);

        auto step = [&](const Slot &q, bool has) {
            // branch-free: an empty slot (the round's longest slice sets the trip count) evaluates its stale rows with
            // every gradient scalar forced to zero (eval_wide) and stores nothing
            if (WIDE_FENCE && NC > 1) __builtin_amdgcn_sched_barrier(WIDE_STEP_FENCE_MASK);   // (the unrolled slots' evaluations stay apart)
            float *gzs = lds + G::gzs + ((it_total & 1) * NG + grp) * (EMAX + 4);
            WTRACE(1);
            const int e = q.e;
            const float wq = by_env ? scw[e] : q.w;
            const float w_rec = rw_rec ? wq : 1.f, w_cls = rw_cls ? wq : 1.f;
            float4 ev[NC];
            lds_row<LG, NC>(ev, sEv, e, lg);
#ifdef WIDE_DIAG_TRACE
            {   // (force the slot's rows and the env row to have arrived)
                float probe = q.qi[0].x + q.qa[NC - 1].w + ev[0].x;
                asm volatile("" :: "v"(probe));
            }
            WTRACE(2);
#endif
            WEval<NC> o;
            eval_wide<LG, NC, EMAX>(o, oi, q.qi, oe, q.qa, ev, sW, sb, bias_l, gzs, wr_gzs, t.E, e, q.y, w_rec * k.invB, w_cls * k.invB, k,
                                    implicit, pure, lg, has ETRACE_ARG);
#ifdef WIDE_DIAG_TRACE
            { float probe = o.gx[0].x + o.g_p + o.lcls; asm volatile("" :: "v"(probe)); }
            WTRACE(3);
#endif
            float s2 = 0.f, s1 = 0.f, s2a = 0.f, s2b = 0.f;
            // (an empty slot stores to the spare row / record behind the minibatch's: the loop's stores are unconditional too)
            float *cr = a.records + (unsigned)(has ? q.cs : a.n_rec) * (unsigned)(2 * DP);
            float4 boo[EVL2 ? 1 : NC];
#pragma unroll
            for (int j = 0; j < NC; j++) {
                float4 gip;
                gip.x = o.g_p - k.alpha * o.gx[j].x; gip.y = o.g_p - k.alpha * o.gx[j].y;
                gip.z = o.g_p - k.alpha * o.gx[j].z; gip.w = o.g_p - k.alpha * o.gx[j].w;
                f4add(gi[j], f4mul(gip, q.qi[j]));
                f4fma(ge[j], o.g_q, f4mul(q.qa[j], ev[j]));
#ifdef WIDE_DIAG_NOSTORE
                if (false) {
#else
                if (push) {   // the interaction's two contribution rows to its ITEM's gradient, at the item-sorted slot
#endif
                    store4<STEP_PUSH_ST>(cr + 4 * (lg + LG * j), f4mul(gip, oi[j]));
                    store4<STEP_PUSH_ST>(cr + DP + 4 * (lg + LG * j), f4scale(o.g_q, f4mul(oe[j], ev[j])));
                }
                if constexpr (!EVL2) {
                    // o = g_q Pa*Qa (+ env regulariser): the interaction's term of embed_env's gradient
                    float4 oo = f4scale(o.g_q, f4mul(oe[j], q.qa[j]));
                    if (reg_env && has) f4add(oo, reg_term(ev[j], 2.f * k.r2, 2.f * k.r1));
                    boo[j] = oo;
                }
                // regulariser REPORTS over the item rows of the interaction (env rows weigh double): two fma chains for the
                // squares (packed pairs), one add chain for the magnitudes
                sq_acc(s2a, s2b, q.qi[j]);
                sq_acc(s2a, s2b, q.qa[j]);
                s1 = abs_acc(abs_acc(s1, q.qi[j]), q.qa[j]);
                if (reg_env) { s2 += 2.f * f4sq(ev[j]); s1 += 2.f * f4abs(ev[j]); }
            }
            if (!push) {
                // pull form: the record {g_p, g_q, env, 0, gz[EMAX]} the item side consumes -- one word per lane, every lane
                // (lanes beyond the record repeat its last word), so that the store is one unconditional wave instruction
                float *rec_g = a.records + (unsigned)(has ? q.cs : a.n_rec) * (unsigned)RS;
                // (gz of this interaction was written by the evaluation above: LDS operations of one wave execute in order.  No
                //  asm fence here: its memory clobber made the compiler split the record's store -- +15 % fabric writes)
                const float gzw = pure ? 0.f : gzs[min(max(lg - 4, 0), EMAX - 1)];
                const float val = lg >= 4 ? gzw : (lg == 0 ? o.g_p : (lg == 1 ? o.g_q : (lg == 2 ? __builtin_bit_cast(float, e) : 0.f)));
                rec_g[min(lg, RS - 1)] = val;
                if (RS > LG) {   // (E = 16 on 16 lanes: the record's last words)
                    const int wi = min(lg, RS - LG - 1);
                    rec_g[LG + wi] = pure ? 0.f : gzs[LG - 4 + wi];
                }
            }
            accL2 += has ? s2 + (s2a + s2b) : 0.f;
            accL1 += has ? s1 : 0.f;
            if (lg == 0) { accLi += o.li * w_rec; accLe += o.le * w_rec; }
            accLc += o.lcls * w_cls;   // (nonzero on the picked class's lane only)
            // A operands: class (lane & 15) of this interaction; rows on 32 lanes read it back from the group's words
            const int lc = lane & 15;
            const float a_gz = LG == 16 ? o.gz_lane : (lc < EMAX ? o.gz_all : 0.f);   // (every lane holds gz of class lane & 15)
            if (lg < 16) dBacc += o.gz_lane;
            WTRACE(4);
            if (WIDE_FENCE && WIDE_MFMA_FENCE && NC > 1) __builtin_amdgcn_sched_barrier(0);
#ifndef WIDE_DIAG_NOMFMA
            outer_mfma<LG, NC>(accW, a_gz, o.x, lane);
#endif
#ifndef WIDE_DIAG_NOMFMA
            if constexpr (SH) outer_mfma<LG, NC>(accW, (has && lc == e + 8) ? 1.f : 0.f, boo, lane);
            else if constexpr (!EVL2) outer_mfma<LG, NC>(accE, (has && lc == e) ? 1.f : 0.f, boo, lane);
#endif
            if (WIDE_FENCE && NC > 1) __builtin_amdgcn_sched_barrier(WIDE_STEP_FENCE_MASK);
            WTRACE(5);
            it_total++;
        };
        for (int s = 0; s < iters; s += UE) {
#pragma unroll
            for (int j = 0; j < UE; j++) {
                if (s + j < iters) step(sl[j], s + j < nsmp);
                gather(sl[j], idn[j], s + UE + j);
                idn[j] = list_at(s + 2 * UE + j);
                WTRACE(6);
            }
        }
        if (r == r0 + STAMP_ROUND) STAMP(4);
        if (r + 1 < r0 + nr) {
            d = a.desc[((r + 1) * NG + grp) * 2];
            d2 = a.desc[((r + 1) * NG + grp) * 2 + 1];
        }
        const float cnt = (float)(meta >> 9);
        if (active && leader) {   // regulariser reports: the user's rows count once per interaction
            float s2 = 0.f, s1 = 0.f;
#pragma unroll
            for (int j = 0; j < NC; j++) { s2 += f4sq(oi[j]) + f4sq(oe[j]); s1 += f4abs(oi[j]) + f4abs(oe[j]); }
            accL2 += cnt * s2;
            accL1 += cnt * s1;
        }
        // the row's Adam moments are requested here -- the interaction slots' registers are free again -- and fly under
        // the slices' meeting
        float4 mm[2][NC], vv[2][NC];
#pragma unroll
        for (int tb = 0; tb < 2; tb++)
#pragma unroll
            for (int j = 0; j < NC; j++) mm[tb][j] = vv[tb][j] = f4zero();
        constexpr bool EARLY_MV = LG == 16;   // (rows on 32 lanes: no registers to spare across the meeting)
        if (EARLY_MV && active && leader && a.fused) {
#pragma unroll
            for (int tb = 0; tb < 2; tb++) {
                if (tb == 1 && pure) break;
                load_row<LG, NC, VEC>(mm[tb], a.m[2 * tb], row, t.D, lg);
                load_row<LG, NC, VEC>(vv[tb], a.v[2 * tb], row, t.D, lg);
            }
        }
        // ---- slices of one row meet through LDS: plain stores, fixed-order sum by the leader
        if (slices > 1) {  // same for every slot of a round, idle slots included
            float *mine = slots + grp * 2 * DP;
#pragma unroll
            for (int j = 0; j < NC; j++) {
                *reinterpret_cast<float4 *>(mine + 4 * (lg + LG * j)) = gi[j];
                *reinterpret_cast<float4 *>(mine + DP + 4 * (lg + LG * j)) = ge[j];
            }
            __syncthreads();
            if (active && leader) {
#pragma unroll 2
                for (int s = 1; s < slices; s++) {
                    const float *oth = slots + (grp + s) * 2 * DP;
#pragma unroll
                    for (int j = 0; j < NC; j++) {
                        f4add(gi[j], *reinterpret_cast<const float4 *>(oth + 4 * (lg + LG * j)));
                        f4add(ge[j], *reinterpret_cast<const float4 *>(oth + DP + 4 * (lg + LG * j)));
                    }
                }
            }
            __syncthreads();  // the slots are rewritten by the next round
        }
        if (r == r0 + STAMP_ROUND) STAMP(5);
        // ---- the leader finishes the row
        if (active && leader) {
            if (cnt != 0.f) {
#pragma unroll
                for (int j = 0; j < NC; j++) {
                    f4fma(gi[j], cnt, reg_term(oi[j], k.r2, k.r1));
                    f4fma(ge[j], cnt, reg_term(oe[j], k.r2, k.r1));
                }
            }
            if (!a.fused) {
                store_row<LG, NC, VEC>(a.np[0], row, t.D, lg, gi);
                if (!pure) store_row<LG, NC, VEC>(a.np[2], row, t.D, lg, ge);
            } else {
#pragma unroll
                for (int tb = 0; tb < 2; tb++) {
                    if (tb == 1 && pure) break;
                    if (!EARLY_MV) {
                        load_row<LG, NC, VEC>(mm[tb], a.m[2 * tb], row, t.D, lg);
                        load_row<LG, NC, VEC>(vv[tb], a.v[2 * tb], row, t.D, lg);
                    }
#pragma unroll
                    for (int j = 0; j < NC; j++) adam4(tb ? oe[j] : oi[j], tb ? ge[j] : gi[j], mm[tb][j], vv[tb][j], ad);
                    store_row<LG, NC, VEC>(a.np[2 * tb], row, t.D, lg, tb ? oe : oi);
                    store_row<LG, NC, VEC>(a.m[2 * tb], row, t.D, lg, mm[tb]);
                    store_row<LG, NC, VEC>(a.v[2 * tb], row, t.D, lg, vv[tb]);
                }
            }
        }
    }
    STAMP(6);
    // ---- the task's partial sums: the waves' accumulator tiles meet in NIMG LDS images (wave w adds into image
    // w % NIMG in turn w / NIMG: fixed order), laid over the staged tables and the slots -- nobody reads those any more --
    // and leave, summed, as the workgroup's partial slab
    accLi = wave_sum_valu(accLi); accLe = wave_sum_valu(accLe); accLc = wave_sum_valu(accLc);
    accL2 = wave_sum_valu(accL2); accL1 = wave_sum_valu(accL1);
    if (LG == 16) dBacc = xor16_sum(dBacc);
    dBacc = xor32_sum(dBacc);                 // lanes 0 .. 15 of the wave: class `lane`, summed over the wave's groups
    __syncthreads();
    constexpr int NIMG = IM::N, ILEN = IM::LEN;
    {   // db | loss sums: one line per wave
        float *mine = tail + wave * (EMAX + kLossSlots);
        if (lane < EMAX) mine[lane] = dBacc;
        if (lane == 0) {
            float *ls = mine + EMAX;
            ls[0] = accLi; ls[1] = accLe; ls[2] = accLc; ls[3] = accL2; ls[4] = accL1; ls[5] = ls[6] = ls[7] = 0.f;
        }
    }
#pragma unroll 1
    for (int turn = 0; turn < kWaves / NIMG; turn++) {
        if (wave / NIMG == turn) {
            float *img = lds + (wave % NIMG) * ILEN;        // [dEv: EMAX][DP] | [dW: EMAX][DP]   (EVL2: dW alone)
            if constexpr (SH) tiles_to_lds<LG, NC, EMAX, true>(img, accW, lane, turn == 0);
            else {
                if constexpr (!EVL2) tiles_to_lds<LG, NC, EMAX>(img, accE, lane, turn == 0);
                tiles_to_lds<LG, NC, EMAX>(img + (EVL2 ? 0 : EMAX * DP), accW, lane, turn == 0);
            }
        }
        __syncthreads();
    }
    {
        float *dst = slab + (EVL2 ? EMAX * DP : 0);   // (EVL2: the embed_env part of this slab is never read)
        for (int i = threadIdx.x; i < ILEN / 4; i += kThreads) {
            float4 v = *reinterpret_cast<const float4 *>(lds + 4 * i);
#pragma unroll
            for (int q = 1; q < NIMG; q++) f4add(v, *reinterpret_cast<const float4 *>(lds + q * ILEN + 4 * i));
            *reinterpret_cast<float4 *>(dst + 4 * i) = v;
        }
        if (threadIdx.x < EMAX + kLossSlots) {
            const int i = threadIdx.x;
            slab[2 * EMAX * DP + i] = ((tail[i] + tail[(EMAX + kLossSlots) + i]) + tail[2 * (EMAX + kLossSlots) + i]) +
                                      tail[3 * (EMAX + kLossSlots) + i];
        }
    }
    STAMP(7);
}

// =====================================================================================
// launch 2, pull form: rounds of ITEM jobs -- partner user rows + record, multiply-add only; EVL2: embed_env's outer
// product as well (the job owns Qa[v], gathers Pa[u], the record holds g_q and the environment)
// =====================================================================================
template <int LG, int NC, bool VEC, int EMAX, bool EVL2>
__device__ __forceinline__ void item_task_wide(const DevTables &t, const StepArgs &a, int r0, int nr, int slab_index, float *lds) {
    using G = WGeo<LG, NC, EMAX>;
    constexpr int NG = G::NG, DP = G::DP, RS = G::RS, TILES = G::TILES;
    constexpr int U = (EVL2 && EMAX == 16) ? WIDE_U_EVL2_16 : 2;   // interactions in flight per group
    float *sEv = lds, *sW = sEv + EMAX * DP, *slots = sW + EMAX * DP;
    const int lg = threadIdx.x & (LG - 1), grp = threadIdx.x / LG, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool pure = a.flags & INVPREF_PURE_MF;
    const bool reg_env = a.flags & INVPREF_REG_ENV_EMBED;
    StepScalars k = a.k;
    if (a.sched_state) {
        const float al = sched_slot_ptr(a.sched_state, a.sched_slot)->alpha;
        if (al == al) k.alpha = al;
    }
    const AdamScalars ad = a.sched_state ? sched_slot_ptr(a.sched_state, a.sched_slot)->ad : a.ad;
    STAMP(0);
    int4 d = a.desc[(r0 * NG + grp) * 2];   // (the inline interactions of the descriptor's second half are read back from memory)
    STAMP(1);
    f32x4 accE[EVL2 ? TILES : 1][4];
#pragma unroll
    for (int i = 0; i < (EVL2 ? TILES : 1); i++)
#pragma unroll
        for (int c = 0; c < 4; c++) accE[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < r0 + nr; r++) {
        const int4 dd = d;
        if (r + 1 < r0 + nr) d = a.desc[((r + 1) * NG + grp) * 2];
        const int row = dd.x, meta = dd.y;
        const bool active = row >= 0, leader = meta & 1;
        const int slices = (meta >> 1) & 31, mode = (meta >> 6) & 7;
        const int nsmp = active ? (mode == 7 ? dd.w - dd.z : mode) : 0;
        // EVL2: the loop carries MFMAs, so it runs to the longest slice of the WAVE (two groups of 32 lanes)
        int iters = nsmp;
        if (EVL2) {
            int mx = 0;
#pragma unroll
            for (int g = 0; g < 64 / LG; g++) mx = max(mx, __builtin_amdgcn_readlane(nsmp, g * LG));
            iters = mx;
        }
        if (r == r0 + STAMP_ROUND) STAMP(2);
        // (the job's own rows are needed when the row is finished -- and Qa[v] by embed_env's product, EVL2: they are
        //  requested there, not held across the loop)
        float4 oi[NC], oe[NC], gi[NC], ge[NC];
#pragma unroll
        for (int j = 0; j < NC; j++) oi[j] = oe[j] = gi[j] = ge[j] = f4zero();
        if (EVL2 && !pure) load_row<LG, NC, VEC>(oe, t.Qa, active ? row : 0, t.D, lg);
        // every load of the loop is unconditional (see user_task_wide): list indices clamped into the slice; the inline form
        // (up to three (user row, position) pairs from word 2 of the descriptor) is read back from the descriptor's words
        const int lo = dd.z, hi1 = max(dd.w - 1, dd.z);
        const int2 *dpairs = reinterpret_cast<const int2 *>(a.desc + (r * NG + grp) * 2);
        auto ids_at = [&](int sidx) {
            const int2 *src = mode == 7 ? a.ilist + min(lo + sidx, hi1) : dpairs + 1 + min(sidx, 2);
            return *src;
        };
        struct In {
            float4 pu[NC], pa[NC], r0;
            float4 gz[EMAX / 4];
        };
        auto fetch = [&](In &in, int2 id) {
            load_row<LG, NC, VEC>(in.pu, t.Pu, id.x, t.D, lg);
            const float *rec = a.records + (unsigned)id.y * (unsigned)RS;
            in.r0 = *reinterpret_cast<const float4 *>(rec);
            if (!pure) {
                load_row<LG, NC, VEC>(in.pa, t.Pa, id.x, t.D, lg);
#pragma unroll
                for (int c4 = 0; c4 < EMAX / 4; c4++) in.gz[c4] = *reinterpret_cast<const float4 *>(rec + 4 + c4 * 4);
            }
        };
        auto consume = [&](const In &in, bool has) {
            if (WIDE_FENCE && NC > 1) __builtin_amdgcn_sched_barrier(0);   // (the unrolled slots stay apart)
            float a_one = 0.f;
            float4 boo[EVL2 ? NC : 1];
#pragma unroll
            for (int j = 0; j < (EVL2 ? NC : 1); j++) boo[j] = f4zero();
            if (has) {
                const float g_p = in.r0.x, g_q = in.r0.y;
                const int e = __builtin_bit_cast(int, in.r0.z);
#pragma unroll
                for (int j = 0; j < NC; j++) {
                    float4 gx = f4zero();
                    if (!pure) {
#ifndef WIDE_DIAG_L2_NOGX   // (what-if build: launch 2 without the classifier's backward)
#pragma unroll
                        for (int c4 = 0; c4 < EMAX / 4; c4++) {
                            if (WIDE_FENCE && NC > 1) __builtin_amdgcn_sched_barrier(0);
                            const float *wr = sW + (c4 * 4) * DP + 4 * (lg + LG * j);
                            f4fma(gx, in.gz[c4].x, *reinterpret_cast<const float4 *>(wr));
                            f4fma(gx, in.gz[c4].y, *reinterpret_cast<const float4 *>(wr + DP));
                            f4fma(gx, in.gz[c4].z, *reinterpret_cast<const float4 *>(wr + 2 * DP));
                            f4fma(gx, in.gz[c4].w, *reinterpret_cast<const float4 *>(wr + 3 * DP));
                        }
#endif
                        const float4 ev = *reinterpret_cast<const float4 *>(sEv + e * DP + 4 * (lg + LG * j));
                        f4fma(ge[j], g_q, f4mul(in.pa[j], ev));
                        if (EVL2) {
                            float4 oo = f4scale(g_q, f4mul(in.pa[j], oe[j]));
                            if (reg_env) f4add(oo, reg_term(ev, 2.f * k.r2, 2.f * k.r1));
                            boo[EVL2 ? j : 0] = oo;
                        }
                    }
                    float4 gip;
                    gip.x = g_p - k.alpha * gx.x; gip.y = g_p - k.alpha * gx.y;
                    gip.z = g_p - k.alpha * gx.z; gip.w = g_p - k.alpha * gx.w;
                    f4add(gi[j], f4mul(gip, in.pu[j]));
                }
                a_one = (lane & 15) == e ? 1.f : 0.f;
            }
#ifndef WIDE_DIAG_L2_NOMFMA   // (what-if build: launch 2 without embed_env's outer product)
            if constexpr (EVL2) outer_mfma<LG, NC>(accE, a_one, boo, lane);
#endif
        };
        In nx[U];
        int2 idn[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
#pragma unroll
            for (int c = 0; c < NC; c++) nx[j].pu[c] = nx[j].pa[c] = f4zero();
            nx[j].r0 = f4zero();
#pragma unroll
            for (int c4 = 0; c4 < EMAX / 4; c4++) nx[j].gz[c4] = f4zero();
            fetch(nx[j], ids_at(j));
        }
#pragma unroll
        for (int j = 0; j < U; j++) idn[j] = ids_at(U + j);
        if (r == r0) {   // (staged behind the first round's gathers, see user_task_wide)
            stage_small(sEv, t.Ev, t.E, t.D, EMAX, DP);
            stage_small(sW, t.W, t.E, t.D, EMAX, DP);
            __syncthreads();
        }
        if (r == r0 + STAMP_ROUND) STAMP(3);
        for (int s = 0; s < iters; s += U) {
#pragma unroll
            for (int j = 0; j < U; j++) {
                if (s + j < iters) consume(nx[j], s + j < nsmp);
                fetch(nx[j], idn[j]);
                idn[j] = ids_at(s + 2 * U + j);
            }
        }
        if (r == r0 + STAMP_ROUND) STAMP(4);
        if (slices > 1) {
            float *mine = slots + grp * 2 * DP;
#pragma unroll
            for (int j = 0; j < NC; j++) {
                *reinterpret_cast<float4 *>(mine + 4 * (lg + LG * j)) = gi[j];
                *reinterpret_cast<float4 *>(mine + DP + 4 * (lg + LG * j)) = ge[j];
            }
            __syncthreads();
            if (active && leader) {
#pragma unroll 2
                for (int s = 1; s < slices; s++) {
                    const float *oth = slots + (grp + s) * 2 * DP;
#pragma unroll
                    for (int j = 0; j < NC; j++) {
                        f4add(gi[j], *reinterpret_cast<const float4 *>(oth + 4 * (lg + LG * j)));
                        f4add(ge[j], *reinterpret_cast<const float4 *>(oth + DP + 4 * (lg + LG * j)));
                    }
                }
            }
            if (r + 1 < r0 + nr) __syncthreads();
        }
        if (r == r0 + STAMP_ROUND) STAMP(5);
        if (active && leader) {
            load_row<LG, NC, VEC>(oi, t.Qi, row, t.D, lg);
            if (!EVL2 && !pure) load_row<LG, NC, VEC>(oe, t.Qa, row, t.D, lg);
            const float cnt = (float)(meta >> 9);
            if (cnt != 0.f) {
#pragma unroll
                for (int j = 0; j < NC; j++) {
                    f4fma(gi[j], cnt, reg_term(oi[j], k.r2, k.r1));
                    f4fma(ge[j], cnt, reg_term(oe[j], k.r2, k.r1));
                }
            }
            if (!a.fused) {
                store_row<LG, NC, VEC>(a.np[1], row, t.D, lg, gi);
                if (!pure) store_row<LG, NC, VEC>(a.np[3], row, t.D, lg, ge);
            } else {
#pragma unroll
                for (int tb = 0; tb < 2; tb++) {
                    if (tb == 1 && pure) break;
                    float4 m[NC], v[NC];
                    load_row<LG, NC, VEC>(m, a.m[2 * tb + 1], row, t.D, lg);
                    load_row<LG, NC, VEC>(v, a.v[2 * tb + 1], row, t.D, lg);
#pragma unroll
                    for (int j = 0; j < NC; j++) adam4(tb ? oe[j] : oi[j], tb ? ge[j] : gi[j], m[j], v[j], ad);
                    store_row<LG, NC, VEC>(a.np[2 * tb + 1], row, t.D, lg, tb ? oe : oi);
                    store_row<LG, NC, VEC>(a.m[2 * tb + 1], row, t.D, lg, m);
                    store_row<LG, NC, VEC>(a.v[2 * tb + 1], row, t.D, lg, v);
                }
            }
        }
    }
    STAMP(6);
    if constexpr (EVL2) {
        // the item task's partial sums of embed_env's gradient: the waves' tiles meet over the staged tables, in wave order
        __syncthreads();
        float *img = lds;
#pragma unroll 1
        for (int w = 0; w < kWaves; w++) {
            if (wave == w) tiles_to_lds<LG, NC, EMAX>(img, accE, lane, w == 0);
            __syncthreads();
        }
        float *slab = a.slabs_ev + (int64_t)slab_index * (EMAX * DP);
        for (int i = threadIdx.x; i < EMAX * DP; i += kThreads) slab[i] = img[i];
    }
}

// launch 2, push form: an item job sums the CONTIGUOUS contribution rows launch 1 stored for its row
template <int LG, int NC, bool VEC, int EMAX>
__device__ __forceinline__ void item_task_push_wide(const DevTables &t, const StepArgs &a, int r0, int nr, float *lds) {
    using G = WGeo<LG, NC, EMAX>;
    constexpr int DP = G::DP, NG = G::NG;
    constexpr int PCH = WIDE_PUSH_DEPTH;   // contribution-row pairs in flight per group
    float *slots = lds;      // [NG][2][DP] slice partials
    const int lg = threadIdx.x & (LG - 1), grp = threadIdx.x / LG;
    const bool pure = a.flags & INVPREF_PURE_MF;
    const StepScalars k = a.k;
    const AdamScalars ad = a.sched_state ? sched_slot_ptr(a.sched_state, a.sched_slot)->ad : a.ad;
    STAMP(0);
    int4 d = a.desc[(r0 * NG + grp) * 2];
    STAMP(1);
    for (int r = r0; r < r0 + nr; r++) {
        const int4 dd = d;
        if (r + 1 < r0 + nr) d = a.desc[((r + 1) * NG + grp) * 2];
        const int row = dd.x, meta = dd.y;
        const bool active = row >= 0, leader = meta & 1;
        const int slices = (meta >> 1) & 31, mode = (meta >> 6) & 7;
        const int nsmp = (active && mode == 7) ? dd.w - dd.z : 0;
        if (r == r0 + STAMP_ROUND) STAMP(2);
        float4 oi[NC], oe[NC], gi[NC], ge[NC];
#pragma unroll
        for (int j = 0; j < NC; j++) oi[j] = oe[j] = gi[j] = ge[j] = f4zero();
        {
            const int rowc = active ? row : 0;
            load_row<LG, NC, VEC>(oi, t.Qi, rowc, t.D, lg);
            if (!pure) load_row<LG, NC, VEC>(oe, t.Qa, rowc, t.D, lg);
        }
        const float *base = a.records + (unsigned)dd.z * (unsigned)(2 * DP);
        // two register sets of H = PCH / 2 pairs: one is added while the other is in flight and refilled as soon as it has
        // been added (see item_task_push in invpref_step.hip); rows are added in slice order: same sums bit for bit
        constexpr int H = PCH / 2;
        static_assert(PCH % 2 == 0, "two register sets");
        float4 ci[2][H][NC], ce[2][H][NC];
        auto fetch = [&](int set, int s0) {   // (clamped: no branch around a load)
#pragma unroll
            for (int q = 0; q < H; q++) {
                const int sj = s0 + q < nsmp ? s0 + q : (nsmp > 0 ? nsmp - 1 : 0);
                const float *p = nsmp > 0 ? base + (unsigned)sj * (unsigned)(2 * DP) : a.records;
#pragma unroll
                for (int j = 0; j < NC; j++) {
                    ci[set][q][j] = *reinterpret_cast<const float4 *>(p + 4 * (lg + LG * j));
                    if (!pure) ce[set][q][j] = *reinterpret_cast<const float4 *>(p + DP + 4 * (lg + LG * j));
                }
            }
        };
        auto add = [&](int set, int s0) {
#pragma unroll
            for (int q = 0; q < H; q++) {
                const bool has = s0 + q < nsmp;
#pragma unroll
                for (int j = 0; j < NC; j++) {
                    f4add(gi[j], has ? ci[set][q][j] : f4zero());
                    if (!pure) f4add(ge[j], has ? ce[set][q][j] : f4zero());
                }
            }
        };
        fetch(0, 0);
        fetch(1, H);
        if (r == r0 + STAMP_ROUND) STAMP(3);
        int n_wave = nsmp;   // the wave's longest slice: refills are wave-uniform branches
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
#pragma unroll
            for (int j = 0; j < NC; j++) {
                *reinterpret_cast<float4 *>(mine + 4 * (lg + LG * j)) = gi[j];
                *reinterpret_cast<float4 *>(mine + DP + 4 * (lg + LG * j)) = ge[j];
            }
            __syncthreads();
            if (active && leader) {
#pragma unroll 2
                for (int s = 1; s < slices; s++) {
                    const float *oth = slots + (grp + s) * 2 * DP;
#pragma unroll
                    for (int j = 0; j < NC; j++) {
                        f4add(gi[j], *reinterpret_cast<const float4 *>(oth + 4 * (lg + LG * j)));
                        f4add(ge[j], *reinterpret_cast<const float4 *>(oth + DP + 4 * (lg + LG * j)));
                    }
                }
            }
            if (r + 1 < r0 + nr) __syncthreads();
        }
        if (r == r0 + STAMP_ROUND) STAMP(5);
        if (active && leader) {
            const float cnt = (float)(meta >> 9);
            if (cnt != 0.f) {
#pragma unroll
                for (int j = 0; j < NC; j++) {
                    f4fma(gi[j], cnt, reg_term(oi[j], k.r2, k.r1));
                    f4fma(ge[j], cnt, reg_term(oe[j], k.r2, k.r1));
                }
            }
            if (!a.fused) {
                store_row<LG, NC, VEC>(a.np[1], row, t.D, lg, gi);
                if (!pure) store_row<LG, NC, VEC>(a.np[3], row, t.D, lg, ge);
            } else {
#pragma unroll
                for (int tb = 0; tb < 2; tb++) {
                    if (tb == 1 && pure) break;
                    float4 m[NC], v[NC];
                    load_row<LG, NC, VEC>(m, a.m[2 * tb + 1], row, t.D, lg);
                    load_row<LG, NC, VEC>(v, a.v[2 * tb + 1], row, t.D, lg);
#pragma unroll
                    for (int j = 0; j < NC; j++) adam4(tb ? oe[j] : oi[j], tb ? ge[j] : gi[j], m[j], v[j], ad);
                    store_row<LG, NC, VEC>(a.np[2 * tb + 1], row, t.D, lg, tb ? oe : oi);
                    store_row<LG, NC, VEC>(a.m[2 * tb + 1], row, t.D, lg, m);
                    store_row<LG, NC, VEC>(a.v[2 * tb + 1], row, t.D, lg, v);
                }
            }
        }
    }
    STAMP(6);
}

// untouched rows: the dense-Adam step with a zero gradient (stream_task for rows of NC float4 per lane, one row of both
// tables per group in flight)
template <int LG, int NC, bool VEC, int THREADS = kThreads>
__device__ __forceinline__ void stream_task_wide(const DevTables &t, const StepArgs &a, const int *rows, int n) {
    constexpr int NG = THREADS / LG;
    const int lg = threadIdx.x & (LG - 1), grp = threadIdx.x / LG;
    const AdamScalars ad = a.sched_state ? sched_slot_ptr(a.sched_state, a.sched_slot)->ad : a.ad;
    const bool pure = a.flags & INVPREF_PURE_MF;
    float4 z[NC];
#pragma unroll
    for (int j = 0; j < NC; j++) z[j] = f4zero();
    if (!a.fused) {   // gradient form: the untouched rows' gradient is a row of zeros
        for (int i = grp; i < n; i += NG) {
            const int rid = rows[i], side = (rid >> 30) & 1, row = rid & 0x3fffffff;
            store_row<LG, NC, VEC>(side ? a.np[1] : a.np[0], row, t.D, lg, z);
            if (!pure) store_row<LG, NC, VEC>(side ? a.np[3] : a.np[2], row, t.D, lg, z);
        }
        return;
    }
    for (int i = grp; i < n; i += NG) {   // (workgroup-uniform trip count up to the last, partial, pass)
        const int rid = rows[i], side = (rid >> 30) & 1, row = rid & 0x3fffffff;
        float4 p[2][NC], m[2][NC], v[2][NC];
#pragma unroll
        for (int tb = 0; tb < 2; tb++) {
            if (tb == 1 && pure) break;
            load_row<LG, NC, VEC>(p[tb], tb ? (side ? t.Qa : t.Pa) : (side ? t.Qi : t.Pu), row, t.D, lg);
            load_row<LG, NC, VEC>(m[tb], side ? a.m[2 * tb + 1] : a.m[2 * tb], row, t.D, lg);
            load_row<LG, NC, VEC>(v[tb], side ? a.v[2 * tb + 1] : a.v[2 * tb], row, t.D, lg);
        }
#pragma unroll
        for (int tb = 0; tb < 2; tb++) {
            if (tb == 1 && pure) break;
#pragma unroll
            for (int j = 0; j < NC; j++) adam4(p[tb][j], f4zero(), m[tb][j], v[tb][j], ad);
            store_row<LG, NC, VEC, STEP_STREAM_ST>(side ? a.np[2 * tb + 1] : a.np[2 * tb], row, t.D, lg, p[tb]);
            store_row<LG, NC, VEC, STEP_STREAM_ST>(side ? a.m[2 * tb + 1] : a.m[2 * tb], row, t.D, lg, m[tb]);
            store_row<LG, NC, VEC, STEP_STREAM_ST>(side ? a.v[2 * tb + 1] : a.v[2 * tb], row, t.D, lg, v[tb]);
        }
    }
}

// ---- the two kernels
template <int LG, int NC, bool VEC, int EMAX, bool EVL2, bool BYENV = false>
__global__ __launch_bounds__(kThreads, (WideCfg<LG, NC, EMAX>::WAVES)) void mstep_eval_wide_kernel(DevTables t, StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ncls = a.n_cls;
    const int c = (int)blockIdx.x % ncls;
    int j = (int)blockIdx.x / ncls;
    int q[4];
    class_row(a, c, q);
    const int rpt = a.rounds_per_task, spt = a.rows_per_stream_task;
    const int tj = (q[1] + rpt - 1) / rpt;
    if (j < tj) {
        user_task_wide<LG, NC, VEC, EMAX, EVL2, BYENV>(t, a, q[0] + j * rpt, min(rpt, q[1] - j * rpt), q[0] / rpt + j, lds);
        return;
    }
    j -= tj;
    if (j * spt < q[3]) {
        STAMP(0);
        stream_task_wide<LG, NC, VEC>(t, a, a.stream_rows + q[2] + j * spt, min(spt, q[3] - j * spt));
        STAMP(7);
    }
}

template <int LG, int NC, bool VEC, int EMAX, bool EVL2>
__global__ __launch_bounds__(kThreads, (NC > 1 || EMAX > 8) ? 2 : 3) void mstep_apply_wide_kernel(DevTables t, StepArgs a, FoldArgs f) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if ((int)blockIdx.x >= f.n_task_wgs) {
        const int fb = (int)blockIdx.x - f.n_task_wgs;
        if (fb == f.fold_blocks) {   // the device-side schedule moves on (see mstep_apply_kernel)
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
        constexpr int DP = 4 * LG * NC;
        fold_block<DP, EMAX>(t, a, f, fb, lds);
        constexpr int loss0 = 2 * EMAX * DP + EMAX;
        if (fb == loss0 / kFoldCols) {   // (workgroup-uniform)
            __syncthreads();
            fold_losses(t, a, f, lds);
        }
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
        if (a.push_slot) item_task_push_wide<LG, NC, VEC, EMAX>(t, a, q[0] + j * rpt, min(rpt, q[1] - j * rpt), lds);
        else item_task_wide<LG, NC, VEC, EMAX, EVL2>(t, a, q[0] + j * rpt, min(rpt, q[1] - j * rpt), q[0] / rpt + j, lds);
        return;
    }
    j -= tj;
    if (j * spt < q[3]) {
        STAMP(0);
        stream_task_wide<LG, NC, VEC>(t, a, a.stream_rows + q[2] + j * spt, min(spt, q[3] - j * spt));
        STAMP(7);
    }
}
