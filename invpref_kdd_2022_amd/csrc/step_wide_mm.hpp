// step_wide_mm.hpp -- launch 1 of the planned M-step for FULL wide rows (factor_num = 128 / 256 -- and 64 with more than
// four environments) with the classifier as three small GEMMs over the WORKGROUP's interactions of a lock-step iteration
// (round 5).  Included by invpref_step.hip behind step_wide.hpp, whose plans, records, contribution rows, slabs, launch 2
// and fold it shares unchanged.
//
// step_wide.hpp evaluates the classifier (models.py:206-209: logits Z = W x + b, their backward G = gz W) per
// interaction on the vector ALU: every lane group reads all of W from LDS twice per interaction (D = 256, E = 16: 64
// ds_read_b128 and ~330 of the ~560 vector instructions of a wave's iteration, each read waited for on the spot:
// profiles/r05 what-if build without the classifier: launch 1 18.3 -> 9.0 ms at 2^24 interactions).  Here the NG = 8 / 16
// interactions a workgroup evaluates per iteration form the N (or M, or K) dimension of v_mfma_f32_16x16x4_f32 products,
// and W never sits in LDS: wave w owns the columns [w DP/4, (w + 1) DP/4) of the row dimension and keeps its slices of W
// in registers for the whole task, once per operand layout.
//
//   S1  groups:  x = Pu*Qi, the two row sums, the recommendation losses' chains, g_p, g_q; x (and o = g_q Pa*Qa, the term
//                of embed_env's gradient) to LDS rows, {scale, env, record index, loss weight} of the interaction to META
//   --- barrier
//   S2  waves:   Z_w[class][interaction] = W[class][cols_w] X[interaction][cols_w]   (A = W slice: registers; B = x rows:
//                ds_read_b128; DP/16 MFMAs per wave); the four waves' partial logits to LDS
//   --- barrier
//   S3  waves:   every wave sums the partials and runs the softmax of ALL the iteration's interactions redundantly -- lane
//                (n, k) holds classes 4k .. 4k+3 of interaction n, which IS the backward's A operand; the class gradients
//                gz go to the pull record straight from these registers (64 bytes per interaction) and through a private
//                LDS transpose (no barrier: one wave's LDS operations execute in order) into the A operand of the products
//                over interactions;
//                G[interaction][cols_w] = gz W[.][cols_w]                 (B = W slice: registers) -> LDS rows
//                dW[class][cols_w]   += gz^T X[.][cols_w]                 (accumulators: 4 registers per 16 columns -- the
//                dEv[env][cols_w]    += onehot(env)^T O[.][cols_w]         per-wave tiles of step_wide.hpp took 64)
//   --- barrier
//   S5  groups:  gx from LDS, the rows' gradients, contribution rows, as before.
//
// The accumulators of a wave cover ITS columns for all classes, so the task's partial slab is written straight from
// registers: no LDS images, no meeting of the waves at the end.  fp32 in, fp32 accumulate, fixed orders: a step stays
// bitwise reproducible.  Arithmetic of one interaction == eval_wide() (models.py:307-326, :206-209; train.py:108-153); the
// logits are summed as four chains of DP/4 products instead of one butterfly (float reordering only).
#pragma once

template <int LG, int NC, int EMAX, bool EVL2>
struct MGeo {
    static constexpr int NG = kThreads / LG, DP = 4 * LG * NC, RS = 4 + EMAX;
    static constexpr int DQ = DP / kWaves;   // columns of a wave
    static constexpr int T = DQ / 16;        // its 16-column tiles = the float4 chunks a lane reads of an x row (forward)
    static constexpr int KW = NG / 4;        // k steps of the products over the iteration's interactions
    static constexpr int XS = DP + 4;        // stride of the x / o / gx rows (16 lanes x 16 bytes at this stride: every bank once)
    static constexpr int PS = 20;            // stride of a [16 classes] line of partial logits / class gradients (same reason)
    static constexpr bool SH = EMAX == 8 && !EVL2;   // rows 8 .. 15 of the classifier's accumulators take embed_env's product
    static constexpr int SLAB = 2 * EMAX * DP + EMAX + kLossSlots;   // == WGeo::SLAB: dEv | dW | db | loss sums
    // LDS (floats)
    static constexpr int sEv = 0;
    static constexpr int slots = EMAX * DP;                  // [NG][2][DP] slice partials of a round's rows
    static constexpr int X = slots + NG * 2 * DP;            // [NG][XS]
    static constexpr int OO = X + NG * XS;                   // [NG][XS] (launch-1 embed_env product only)
    static constexpr int GX = OO + (EVL2 ? 0 : NG * XS);     // [NG][XS]
    static constexpr int P = GX + NG * XS;                   // [waves][16 interactions][PS] partial logits
    static constexpr int GZT = P + kWaves * 16 * PS;         // [waves][16 interactions][PS] a wave's private copy of gz
    static constexpr int META = GZT + kWaves * 16 * PS;      // [16 interactions][4]
    static constexpr int CW = META + 64;                     // [16] class weights (INVPREF_WEIGHTS_BY_ENV; else ones)
    static constexpr int tail = CW + 16;                     // [waves][kLossSlots]
    static constexpr int total = tail + kWaves * kLossSlots;
};

template <int T> struct VecT;
template <> struct VecT<1> { typedef float type; };
template <> struct VecT<2> { typedef float2 type; };
template <> struct VecT<4> { typedef float4 type; };
template <int T>
__device__ __forceinline__ void ld_t(float (&v)[T], const float *p) {
    const typename VecT<T>::type r = *reinterpret_cast<const typename VecT<T>::type *>(p);
    const float *f = reinterpret_cast<const float *>(&r);
#pragma unroll
    for (int i = 0; i < T; i++) v[i] = f[i];
}
template <int T>
__device__ __forceinline__ void st_t(float *p, const float (&v)[T]) {
    typename VecT<T>::type r;
    float *f = reinterpret_cast<float *>(&r);
#pragma unroll
    for (int i = 0; i < T; i++) f[i] = v[i];
    *reinterpret_cast<typename VecT<T>::type *>(p) = r;
}
#ifndef WIDE_MM_WAVES
#define WIDE_MM_WAVES 2
#endif
// what-if builds (tools/wide_whatif.sh; numerically wrong): -DMM_DIAG_NOBAR drops the iteration's three barriers,
// -DMM_DIAG_NOMFMA the products, -DMM_DIAG_HOT makes every gather a cache hit
#ifdef MM_DIAG_NOBAR
#define MM_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define MM_BARRIER() __syncthreads()
#endif
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
#ifdef MM_DIAG_NOMFMA
    c[0] += a * b;
    return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

template <int LG, int NC, int EMAX, bool EVL2, bool BYENV = false>
__device__ __forceinline__ void user_task_wide_mm(const DevTables &t, const StepArgs &a, int r0, int nr, int slab_index, float *lds) {
    using G = MGeo<LG, NC, EMAX, EVL2>;
    constexpr bool VEC = true;
    constexpr int NG = G::NG, DP = G::DP, RS = G::RS, T = G::T, KW = G::KW, XS = G::XS, PS = G::PS, DQ = G::DQ;
    constexpr bool SH = G::SH;
    constexpr int UE = WideCfg<LG, NC, EMAX>::UE;
    float *sEv = lds + G::sEv, *slots = lds + G::slots, *sX = lds + G::X, *sO = lds + G::OO, *sG = lds + G::GX;
    float *sP = lds + G::P, *sZ = lds + G::GZT, *sM = lds + G::META, *tail = lds + G::tail;
    const int lg = threadIdx.x & (LG - 1), grp = threadIdx.x / LG, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n16 = lane & 15, kq = lane >> 4;
    const bool implicit = a.flags & INVPREF_IMPLICIT;
    const bool rw_rec = a.flags & INVPREF_REWEIGHT_REC, rw_cls = a.flags & INVPREF_REWEIGHT_CLS;
    constexpr bool by_env = BYENV;   // (compile time, see step_wide.hpp) weight = class_weights[env], staged with embed_env
    float *scw = lds + G::CW;
    const bool reg_env = a.flags & INVPREF_REG_ENV_EMBED;
    const bool push = a.push_slot != nullptr;
    const int E = t.E;
    StepScalars k = a.k;
    if (a.sched_state) {  // scheduled alpha (train.py:214-217) under graph replay
        const float al = sched_slot_ptr(a.sched_state, a.sched_slot)->alpha;
        if (al == al) k.alpha = al;
    }
    const AdamScalars ad = a.sched_state ? sched_slot_ptr(a.sched_state, a.sched_slot)->ad : a.ad;

    STAMP(0);
    int4 d = a.desc[(r0 * NG + grp) * 2], d2 = a.desc[(r0 * NG + grp) * 2 + 1];
    STAMP(1);

    // ---- the wave's slices of the classifier, once per operand layout (registers for the whole task)
    const int base_w = wave * DQ;
    float4 wA[T];      // forward A operand: class n16, columns base_w + 16 q + 4 kq + (0 .. 3)  [k step 4 q + i: component i]
    float wB[T][4];    // backward B operand: class 4 kq + r, column base_w + T n16 + t
    {
        const float *wrow = t.W + min(n16, E - 1) * DP + base_w + 4 * kq;
#pragma unroll
        for (int q = 0; q < T; q++) {
            const float4 w = *reinterpret_cast<const float4 *>(wrow + 16 * q);
            wA[q] = n16 < E ? w : f4zero();
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int c = 4 * kq + r;
            float tmp[T];
            ld_t<T>(tmp, t.W + min(c, E - 1) * DP + base_w + T * n16);
#pragma unroll
            for (int tt = 0; tt < T; tt++) wB[tt][r] = c < E ? tmp[tt] : 0.f;
        }
    }
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};   // the logits' bias enters through wave 0's accumulator
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) bias4[r] = (4 * kq + r < E) ? t.b[min(4 * kq + r, E - 1)] : 0.f;
    }
    f32x4 accW[T], accE[(EVL2 || SH) ? 1 : T];
#pragma unroll
    for (int i = 0; i < T; i++) accW[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < ((EVL2 || SH) ? 1 : T); i++) accE[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dB[4] = {0.f, 0.f, 0.f, 0.f};   // wave 0: sum of gz over the interactions n16 of every iteration, classes 4 kq + r
    float accLi = 0.f, accLe = 0.f, accLc = 0.f, accL2 = 0.f, accL1 = 0.f;
    float *slab = a.slabs + (int64_t)slab_index * G::SLAB;
#ifdef WIDE_DIAG_TRACE
    int wtrace_n = 0;   // (tools/wide_trace.py: shader-clock stamps of workgroup 40's first wave, -DWIDE_DIAG_TRACE builds only)
#endif
    if (threadIdx.x < 64) sM[threadIdx.x] = 0.f;   // (interactions NG .. 15 of the padded products: scale 0 for good)

    for (int r = r0; r < r0 + nr; r++) {
        const int4 dd = d, dd2 = d2;
        const int row = dd.x, meta = dd.y;
        const bool active = row >= 0, leader = meta & 1;
        const int slices = (meta >> 1) & 31, mode = (meta >> 6) & 7;
        const int nsmp = active ? (mode == 7 ? dd.w - dd.z : mode) : 0;
        const int iters = a.round_iters[r];   // the round's longest slice: the loop, its barriers and products are workgroup-uniform
        if (r == r0 + STAMP_ROUND) STAMP(2);
        // (the gather pipeline of step_wide.hpp: every load unconditional, list indices clamped into the slice)
        const int lo = dd.z, hi1 = max(dd.w - 1, dd.z);
        const int *dwords = reinterpret_cast<const int *>(a.desc + (r * NG + grp) * 2);
        auto list_at = [&](int sidx) {
            const int *src = mode == 7 ? reinterpret_cast<const int *>(a.ulist + min(lo + sidx, hi1)) : dwords + 2 + 3 * min(sidx, 1);
            return USample{src[0], src[1], __builtin_bit_cast(float, src[2])};
        };
        float4 oi[NC], oe[NC], gi[NC], ge[NC];
#pragma unroll
        for (int j = 0; j < NC; j++) oi[j] = oe[j] = gi[j] = ge[j] = f4zero();
        {
            const int rowc = active ? row : 0;
            load_row<LG, NC, VEC>(oi, t.Pu, rowc, t.D, lg);
            load_row<LG, NC, VEC>(oe, t.Pa, rowc, t.D, lg);
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
#ifdef MM_DIAG_HOT
            const int oth = sm.oth & 15;
#else
            const int oth = sm.oth;
#endif
            load_row<LG, NC, VEC>(q.qi, t.Qi, oth, t.D, lg);
            const unsigned pso = (unsigned)sm.ps;
            load_row<LG, NC, VEC>(q.qa, t.Qa, oth, t.D, lg);
            q.e = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(a.envs) + pso * 8u);   // low word of the int64 id
            if ((rw_rec || rw_cls) && !by_env) q.w = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.weights) + pso * 4u);
            // the slot: word 3 of the list entry the ids came from or, inline form, rec_slot[position] -- one load, selected address
            const int *sp = mode == 7 ? reinterpret_cast<const int *>(a.ulist + min(lo + sidx, hi1)) + 3
                                      : reinterpret_cast<const int *>(reinterpret_cast<const char *>(a.rec_slot) + pso * 4u);
            q.cs = STEP_SLOT_FROM_LIST ? *sp : *reinterpret_cast<const int *>(reinterpret_cast<const char *>(a.rec_slot) + pso * 4u);
        };
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
        if (r == r0) {   // embed_env's table is staged behind the first round's gathers (see step_wide.hpp)
            stage_small(sEv, t.Ev, t.E, t.D, EMAX, DP);
            if (threadIdx.x < 16)   // (a 32-bit offset from the scalar base, see step_wide.hpp)
                scw[threadIdx.x] = (by_env && threadIdx.x < t.E) ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.weights) + (threadIdx.x & 15u) * 4u) : 1.f;
            __syncthreads();
        }
        if (r == r0 + STAMP_ROUND) STAMP(3);

        auto step = [&](const Slot &q, bool has) {
            // ---- S1: the interaction's recommendation part on its lane group (an empty slot of the lock-step iteration
            // evaluates its stale -- finite -- rows with every gradient scalar forced to zero)
            WTRACE(20);
            const int e = q.e;
            float g_p, g_q;
            {
                float4 ev[NC], x[NC];
                lds_row<LG, NC>(ev, sEv, e, lg);
                float ps = 0.f, qs = 0.f;
#pragma unroll
                for (int j = 0; j < NC; j++) {
                    x[j] = f4mul(oi[j], q.qi[j]);
                    const float pj = (x[j].x + x[j].y) + (x[j].z + x[j].w), qj = dot4(f4mul(oe[j], q.qa[j]), ev[j]);
                    ps = j ? ps + pj : pj;
                    qs = j ? qs + qj : qj;
                    *reinterpret_cast<float4 *>(sX + grp * XS + 4 * (lg + LG * j)) = x[j];
                }
                const float p = group_sum<LG>(ps);
                const float qq = group_sum<LG>(qs);
                // (the class weight is read HERE, with the products above already on their way to LDS: at 256 registers a
                //  value held across them went to scratch memory)
                const float wq = by_env ? scw[e] : q.w;
                const float w_rec = rw_rec ? wq : 1.f, w_cls = rw_cls ? wq : 1.f;
                const float cw_rec = w_rec * k.invB;
                float li, le;
                if (implicit) {
                    const float sp = f_sigmoid(p), sq = f_sigmoid(qq), sv = sp * sq;
                    li = f_bce(sp, q.y);
                    le = f_bce(sv, q.y);
                    const float d_inv = k.ca * cw_rec * f_dbce(sp, q.y);
                    const float d_env = k.cb * cw_rec * f_dbce(sv, q.y);
                    g_p = (d_inv + d_env * sq) * (sp * (1.f - sp));
                    g_q = d_env * sp * (sq * (1.f - sq));
                } else {
                    const float s2 = p + qq;
                    li = (p - q.y) * (p - q.y);
                    le = (s2 - q.y) * (s2 - q.y);
                    const float d_env = k.cb * cw_rec * 2.f * (s2 - q.y);
                    g_p = k.ca * cw_rec * 2.f * (p - q.y) + d_env;
                    g_q = d_env;
                }
                if (!has) g_p = g_q = li = le = 0.f;
                if (lg == 0) {
                    accLi += li * w_rec;
                    accLe += le * w_rec;
                    // what the waves need of this interaction: the class gradients' scale, the environment (an empty slot: none),
                    // the record it writes to (an empty slot: the spare one), the class loss's weight
                    float4 m4;
                    m4.x = has ? k.cc * (w_cls * k.invB) : 0.f;
                    m4.y = __builtin_bit_cast(float, has ? e : -100);
                    m4.z = __builtin_bit_cast(float, has ? q.cs : a.n_rec);
                    m4.w = has ? w_cls : 0.f;
                    *reinterpret_cast<float4 *>(sM + 4 * grp) = m4;
                }
                float s2 = 0.f, s1 = 0.f, s2a = 0.f, s2b = 0.f;
#pragma unroll
                for (int j = 0; j < NC; j++) {
                    if constexpr (!EVL2) {
                        // o = g_q Pa*Qa (+ env regulariser): the interaction's term of embed_env's gradient
                        float4 oo = f4scale(g_q, f4mul(oe[j], q.qa[j]));
                        if (reg_env && has) f4add(oo, reg_term(ev[j], 2.f * k.r2, 2.f * k.r1));
                        *reinterpret_cast<float4 *>(sO + grp * XS + 4 * (lg + LG * j)) = oo;
                    }
                    // regulariser REPORTS over the item rows of the interaction (env rows weigh double)
                    sq_acc(s2a, s2b, q.qi[j]);
                    sq_acc(s2a, s2b, q.qa[j]);
                    s1 = abs_acc(abs_acc(s1, q.qi[j]), q.qa[j]);
                    if (reg_env) { s2 += 2.f * f4sq(ev[j]); s1 += 2.f * f4abs(ev[j]); }
                }
                accL2 += has ? s2 + (s2a + s2b) : 0.f;
                accL1 += has ? s1 : 0.f;
                if (!push) {
                    // pull form: the head {g_p, g_q, env, 0} of the record the item side consumes (the waves add gz[EMAX] below)
                    float *rec_g = a.records + (unsigned)(has ? q.cs : a.n_rec) * (unsigned)RS;
                    const float val = lg == 0 ? g_p : (lg == 1 ? g_q : (lg == 2 ? __builtin_bit_cast(float, e) : 0.f));
                    rec_g[min(lg, 3)] = val;
                }
            }
            WTRACE(23);
            MM_BARRIER();
            WTRACE(24);
            // ---- S2: partial logits of the iteration's interactions over the wave's columns
            {
                // (two accumulator chains: a product does not wait for the one in front of it)
                f32x4 z = bias4, z1 = {0.f, 0.f, 0.f, 0.f};
                const float *xr = sX + (n16 & (NG - 1)) * XS + base_w + 4 * kq;
#pragma unroll
                for (int qd = 0; qd < T; qd++) {
                    const float4 xb = *reinterpret_cast<const float4 *>(xr + 16 * qd);
                    z = mfma4(wA[qd].x, xb.x, z);
                    z1 = mfma4(wA[qd].y, xb.y, z1);
                    z = mfma4(wA[qd].z, xb.z, z);
                    z1 = mfma4(wA[qd].w, xb.w, z1);
                }
                *reinterpret_cast<float4 *>(sP + (wave * 16 + n16) * PS + 4 * kq) =
                    make_float4(z[0] + z1[0], z[1] + z1[1], z[2] + z1[2], z[3] + z1[3]);
            }
            WTRACE(25);
            MM_BARRIER();
            WTRACE(27);
            // ---- S3: softmax of every interaction in every wave (lane (n16, kq): classes 4 kq .. 4 kq + 3 of interaction n16)
            {
                float z[4], gz[4];
                {
                    const float *pr = sP + n16 * PS + 4 * kq;
                    const float4 p0 = *reinterpret_cast<const float4 *>(pr), p1 = *reinterpret_cast<const float4 *>(pr + 16 * PS);
                    const float4 p2 = *reinterpret_cast<const float4 *>(pr + 32 * PS), p3 = *reinterpret_cast<const float4 *>(pr + 48 * PS);
                    z[0] = ((p0.x + p1.x) + p2.x) + p3.x; z[1] = ((p0.y + p1.y) + p2.y) + p3.y;
                    z[2] = ((p0.z + p1.z) + p2.z) + p3.z; z[3] = ((p0.w + p1.w) + p2.w) + p3.w;
                }
                const float4 m4 = *reinterpret_cast<const float4 *>(sM + 4 * n16);
                const float scale = m4.x, wl = m4.w;
                const int en = __builtin_bit_cast(int, m4.y), psn = __builtin_bit_cast(int, m4.z);
                float mx = -__builtin_inff();
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    if (4 * kq + r4 >= E) z[r4] = -__builtin_inff();
                    mx = __builtin_fmaxf(mx, z[r4]);
                }
                mx = xor32_max(xor16_max(mx));
                float ez[4], se = 0.f;
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    ez[r4] = (4 * kq + r4 < E) ? f_exp(z[r4] - mx) : 0.f;
                    se = r4 ? se + ez[r4] : ez[r4];
                }
                se = xor32_sum(xor16_sum(se));
                const float rsel = f_rcp(se);
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) gz[r4] = scale * (ez[r4] * rsel - (4 * kq + r4 == en ? 1.f : 0.f));
                if (wave == 0) {
                    // the class loss (log_softmax form, models.py:206-209) on the lane that holds the interaction's class; db
                    const float ze = (en & 3) == 0 ? z[0] : ((en & 3) == 1 ? z[1] : ((en & 3) == 2 ? z[2] : z[3]));
                    if ((en >> 2) == kq) accLc += wl * (-f_log(rsel) - (ze - mx));
#pragma unroll
                    for (int r4 = 0; r4 < 4; r4++) dB[r4] += gz[r4];
                }
                const float4 gz4 = make_float4(gz[0], gz[1], gz[2], gz[3]);
                if (!push && (n16 & 3) == wave && n16 < NG && 4 * kq < EMAX)
                    *reinterpret_cast<float4 *>(a.records + (unsigned)psn * (unsigned)RS + 4 + 4 * kq) = gz4;
                *reinterpret_cast<float4 *>(sZ + (wave * 16 + n16) * PS + 4 * kq) = gz4;
                WTRACE(21);
                // G[interaction][cols_w] = gz W: k runs over the classes 4 kq + r
                f32x4 g[T];
#pragma unroll
                for (int tt = 0; tt < T; tt++) g[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++)
#pragma unroll
                    for (int tt = 0; tt < T; tt++) g[tt] = mfma4(gz[r4], wB[tt][r4], g[tt]);
                WAVE_LDS_FENCE();
                // dW[class][cols_w] += gz^T X, dEv[env][cols_w] += onehot(env)^T O: k runs over the interactions 4 kk + kq
#pragma unroll
                for (int kk = 0; kk < KW; kk++) {
                    const int ni = 4 * kk + kq;
                    const float ag = sZ[(wave * 16 + ni) * PS + n16];
                    float xb[T];
                    ld_t<T>(xb, sX + ni * XS + base_w + T * n16);
#pragma unroll
                    for (int tt = 0; tt < T; tt++) accW[tt] = mfma4(ag, xb[tt], accW[tt]);
                    if constexpr (!EVL2) {
                        const int eni = __builtin_bit_cast(int, sM[4 * ni + 1]);
                        const float ae = eni == (SH ? n16 - 8 : n16) ? 1.f : 0.f;
                        float ob[T];
                        ld_t<T>(ob, sO + ni * XS + base_w + T * n16);
#pragma unroll
                        for (int tt = 0; tt < T; tt++) {
                            if constexpr (SH) accW[tt] = mfma4(ae, ob[tt], accW[tt]);
                            else accE[tt] = mfma4(ae, ob[tt], accE[tt]);
                        }
                    }
                }
                // (gx leaves last: the products above are all issued before the first of them is waited for)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    const int m = 4 * kq + r4;   // accumulator row = interaction
                    if (m < NG) {
                        float v[T];
#pragma unroll
                        for (int tt = 0; tt < T; tt++) v[tt] = g[tt][r4];
                        st_t<T>(sG + m * XS + base_w + T * n16, v);
                    }
                }
            }
            WTRACE(22);
            MM_BARRIER();
            WTRACE(24);
            // ---- S5: the rows' gradients with gx = gz W from LDS
            {
                float4 ev[NC];
                lds_row<LG, NC>(ev, sEv, e, lg);
                float *cr = a.records + (unsigned)(has ? q.cs : a.n_rec) * (unsigned)(2 * DP);
#pragma unroll
                for (int j = 0; j < NC; j++) {
                    const float4 gx = *reinterpret_cast<const float4 *>(sG + grp * XS + 4 * (lg + LG * j));
                    float4 gip;
                    gip.x = g_p - k.alpha * gx.x; gip.y = g_p - k.alpha * gx.y;
                    gip.z = g_p - k.alpha * gx.z; gip.w = g_p - k.alpha * gx.w;
                    f4add(gi[j], f4mul(gip, q.qi[j]));
                    f4fma(ge[j], g_q, f4mul(q.qa[j], ev[j]));
                    if (push) {   // the interaction's two contribution rows to its ITEM's gradient, at the item-sorted slot
                        store4<STEP_PUSH_ST>(cr + 4 * (lg + LG * j), f4mul(gip, oi[j]));
                        store4<STEP_PUSH_ST>(cr + DP + 4 * (lg + LG * j), f4scale(g_q, f4mul(oe[j], ev[j])));
                    }
                }
            }
        };
        for (int s = 0; s < iters; s += UE) {
#pragma unroll
            for (int j = 0; j < UE; j++) {
                if (s + j < iters) step(sl[j], s + j < nsmp);
                WTRACE(26);
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
        float4 mm[2][NC], vv[2][NC];
#pragma unroll
        for (int tb = 0; tb < 2; tb++)
#pragma unroll
            for (int j = 0; j < NC; j++) mm[tb][j] = vv[tb][j] = f4zero();
        constexpr bool EARLY_MV = LG == 16;   // (the moments fly under the slices' meeting where registers allow)
        if (EARLY_MV && active && leader && a.fused) {
#pragma unroll
            for (int tb = 0; tb < 2; tb++) {
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
                store_row<LG, NC, VEC>(a.np[2], row, t.D, lg, ge);
            } else {
#pragma unroll
                for (int tb = 0; tb < 2; tb++) {
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
    // ---- the task's partial slab: a wave's accumulators cover ITS columns for every class -- stored straight from the
    // registers (accumulator row 4 kq + r, columns base_w + T n16 + t); db from wave 0; the loss sums meet in LDS
    accLi = wave_sum_valu(accLi); accLe = wave_sum_valu(accLe); accLc = wave_sum_valu(accLc);
    accL2 = wave_sum_valu(accL2); accL1 = wave_sum_valu(accL1);
    if (lane == 0) {
        float *ls = tail + wave * kLossSlots;
        ls[0] = accLi; ls[1] = accLe; ls[2] = accLc; ls[3] = accL2; ls[4] = accL1; ls[5] = ls[6] = ls[7] = 0.f;
    }
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) {
        const int c = 4 * kq + r4;
        float v[T];
#pragma unroll
        for (int tt = 0; tt < T; tt++) v[tt] = accW[tt][r4];
        if constexpr (SH) {
            // rows 0 .. 7: classifier classes, rows 8 .. 15: environments
            st_t<T>(slab + (c < 8 ? (EMAX + c) * DP : (c - 8) * DP) + base_w + T * n16, v);
        } else {
            if (c < EMAX) st_t<T>(slab + (EMAX + c) * DP + base_w + T * n16, v);
            if constexpr (!EVL2) {
                float u[T];
#pragma unroll
                for (int tt = 0; tt < T; tt++) u[tt] = accE[tt][r4];
                if (c < EMAX) st_t<T>(slab + c * DP + base_w + T * n16, u);
            }
        }
        if (wave == 0) {
            const float s = row16_sum(dB[r4]);
            if (n16 == 0 && c < EMAX) slab[2 * EMAX * DP + c] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < kLossSlots) {
        const int i = threadIdx.x;
        slab[2 * EMAX * DP + EMAX + i] = ((tail[i] + tail[kLossSlots + i]) + tail[2 * kLossSlots + i]) + tail[3 * kLossSlots + i];
    }
    STAMP(7);
}

template <int LG, int NC, int EMAX, bool EVL2, bool BYENV = false>
__global__ __launch_bounds__(kThreads, WIDE_MM_WAVES) void mstep_eval_mm_kernel(DevTables t, StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ncls = a.n_cls;
    const int c = (int)blockIdx.x % ncls;
    int j = (int)blockIdx.x / ncls;
    int q[4];
    class_row(a, c, q);
    const int rpt = a.rounds_per_task, spt = a.rows_per_stream_task;
    const int tj = (q[1] + rpt - 1) / rpt;
    if (j < tj) {
        user_task_wide_mm<LG, NC, EMAX, EVL2, BYENV>(t, a, q[0] + j * rpt, min(rpt, q[1] - j * rpt), q[0] / rpt + j, lds);
        return;
    }
    j -= tj;
    if (j * spt < q[3]) {
        STAMP(0);
        stream_task_wide<LG, NC, true>(t, a, a.stream_rows + q[2] + j * spt, min(spt, q[3] - j * spt));
        STAMP(7);
    }
}
