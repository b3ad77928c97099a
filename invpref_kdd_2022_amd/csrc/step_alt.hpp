// step_alt.hpp -- ONE launch per optimiser step: the evaluating side alternates (include/invpref_hip.h: InvPrefAltPlan).
// Included by invpref_step.hip (lane-group helpers, Eval, Geo, EvalLds, fold geometry come from there).
//
// Launch c of a run, side S (users for even c, items for odd c), T = the other side:
//   fold blocks (first in the grid): sum launch c-1's partial slabs, Adam on embed_env / classifier IN PLACE with
//       write-through stores, the six loss terms of step c-1, one flag per fold block (value = step number);
//   S's jobs: (i) sum the contribution rows T pushed for the row in launch c-1 (the slices share them and meet in LDS),
//       Adam with step c-1's scalars in registers; (ii) wait for the fold flags, stage the small tables with
//       cache-bypassing loads; (iii) evaluate step c's interactions (forward + analytic backward of train.py:108-153,
//       models.py:307-391 -- symmetric in users and items), push the two contribution rows per interaction for T,
//       accumulate the E x D / loss partial sums; (iv) second Adam, ONE store of p, m, v;
//   S's other rows (no interaction in step c): pending rows (if any) + both Adam updates, streamed.
// T's rows are only read.  Every sum has a fixed order: bitwise reproducible run to run.

// phase stamps of the alt kernels: diagnostic builds only (-DALT_STAMPS; tools/alt_probe.py) -- the stamp pointer and its
// branches cost scalar registers the hot instance does not have
#ifdef ALT_STAMPS
#define ASTAMP(i) STAMP(i)
#else
#define ASTAMP(i) do { } while (0)
#endif

struct AltArgs {
    float *own_p[2];              // S's parameter tables (invariant, env-aware): updated in place
    float *own_m[2], *own_v[2];
    const float *oth_p[2];        // T's tables: read only
    float *Ev, *W, *b;            // small tables + moments: updated in place by the fold blocks
    float *mEv, *mW, *mb, *vEv, *vW, *vb;
    int E, D;
    const int4 *desc;             // [rounds][NG][2]
    const int4 *pend;             // [rounds][NG]
    const int4 *list;             // [n] {partner row, position, label bits, 0}
    const int *push_slot;         // [n]
    const int4 *stream;           // [n_stream] {row, a, b, count}
    int rounds_per_task, rows_per_stream_task, n_cls;
    int cls[8][4];
    const int64_t *envs;
    const float *weights;
    StepScalars k;                // current step
    float r2_prev, r1_prev;       // previous step's regulariser scalars
    uint32_t flags;
    int mode;                     // bit 0: has_prev, bit 1: has_cur
    AdamScalars ad_cur, ad_prev;  // eager form; with a schedule: from the slot
    int gen;                      // eager form: the step number the fold flags carry
    int *sched_state;
    int sched_slot;
    const SchedRow *sched_table;
    int sched_n;
    const float *pend_rows;       // [n_prev][2][DP] contribution rows pushed for S by the previous launch
    float *push_rows;             // [n][2][DP] contribution rows this launch pushes for T
    const float *slabs_prev;      // [n_partials_prev][SLAB]
    int n_partials_prev;
    float *slabs;                 // [job tasks][SLAB]
    int *fold_flags;              // [64]; word 63: error
    unsigned long long *pub;      // [2 * EMAX * DP + EMAX] granules {value, step}: the small tables as the fold blocks publish them
    float l2, l1;
    double inv_B_prev, inv_BD2_prev;
    float *losses_prev;
    int fold_blocks, first_task_block;
    int stamps_nodrain;
    unsigned long long *stamps;
};

__device__ __forceinline__ float ld_sc1(const float *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_sc1(const int *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(float *p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(int *p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifndef ALT_POLL_MAX
#define ALT_POLL_MAX (1 << 17)    // polls before a waiting workgroup gives up (sets the error word): ~30 ms
#endif
#ifndef ALT_TEST_BAD_TAG
#define ALT_TEST_BAD_TAG 0        // (test builds only, tests/test_alt_gpu.py: the job workgroups wait for a tag nobody publishes,
                                  //  so every wait times out -- with a small ALT_POLL_MAX -- and the managers must raise)
#endif
#ifndef ALT_STREAM_DELAY
#define ALT_STREAM_DELAY 40       // s_sleep units of 64 clocks in front of a stream task (evaluating launches): ~1 us.  Round 5 measured
                                  // no gain from it (16.0 vs 15.95 us); with the pushes written through and two pending pairs in flight
                                  // (round 6) the rows without a job DO queue in front of the jobs' first burst: 14.08 -> 13.97 us
#endif
#ifndef ALT_PEND_COND
#define ALT_PEND_COND 0           // (A/B knob: 1 = the first pending-row loads only for waves that have pending rows -- measured
                                  //  slower, 17.0 vs 16.1 us per step: the wave-uniform branch needs the ranges before any load)
#endif
#ifndef ALT_EARLY_STAGE
#define ALT_EARLY_STAGE 0         // (A/B knob: 1 = the small tables' granules requested in front of the previous step's update;
                                  //  18 registers across that phase: needs ALT_PEND_DEPTH 2 to stay free of scratch)
#endif
#ifndef ALT_PUSH_ST
#define ALT_PUSH_ST 1             // the contribution rows pushed for the other side leave as write-through stores (0: plain) -- they are
                                  // read by the NEXT launch on other XCDs, and dirty bytes left in L2 lengthen the kernel boundary
                                  // (round 6, same box: 14.69 -> 14.33 us per launch; write-through rows p / m / v: 14.9, slower)
#endif
#ifndef ALT_SLAB_ST
#define ALT_SLAB_ST 1             // the partial slab of a job workgroup too (read by the next launch's fold blocks)
#endif
#ifndef ALT_ROW_ST
#define ALT_ROW_ST 0              // (A/B knob) 1: the rows a job finishes (p, m, v) leave as write-through stores -- measured SLOWER
                                  // here (14.9 vs 14.7 us per launch): the other side gathers them in the very next launch
#endif
#ifndef ALT_MV_ST
#define ALT_MV_ST ALT_ROW_ST      // (A/B knob) 1: the Adam moments of the rows a job finishes leave as write-through stores (nobody
                                  // reads them before the row's own job two launches on)
#endif
#ifndef ALT_PEND_DEPTH
#define ALT_PEND_DEPTH 2          // pending contribution-row pairs in flight per group (two register sets).  Round 5: 4 (16.26 vs 16.43
                                  // us); round 6, the pushed rows now written through (they come from the Infinity Cache, not from a
                                  // neighbour's L2) and the first burst the launch's bottleneck: 2 is faster, 14.23 -> 14.08 us
#endif

// The small tables travel from the fold blocks to the job workgroups of the SAME launch as data-tagged granules: the fold
// thread that finishes an entry stores {value, step number} as ONE 8-byte write-through store, wave 0 of a job workgroup
// loads its nine granules in one burst with cache-bypassing loads and accepts them when every tag carries this step's
// number -- one round trip once the fold is through (a separate flag costs two: poll, then loads), no ordering needed
// between the stores (MI355X guide, inter-workgroup visibility: R2 granules, observed untorn for naturally aligned 8-byte
// sc1 stores on gfx950).  Entries outside the tables (classes >= env_num, columns >= factor_num) are never published and
// never waited for.  A run's first launch (no fold) zeroes the tags, so a stale granule of an earlier run cannot carry
// one of this run's step numbers.
struct AltStage {
    unsigned long long g[9];
    float cw;   // INVPREF_WEIGHTS_BY_ENV: class weight of environment `lane` (lanes < env_num of wave 0), requested with the granules
};
__device__ __forceinline__ void alt_stage_lanes(const AltArgs &a, int lane, bool (&on)[4], bool &onb) {
    constexpr int DP = 64;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int idx = lane + 64 * i, e = idx / DP, d = idx - e * DP;
        on[i] = e < a.E && d < a.D;
    }
    onb = lane < a.E;
}
#define ALT_GRANULE_LOADS                                                                                                     \
    "global_load_dwordx2 %0, %9, %11 sc1\n\t"                                                                              \
    "global_load_dwordx2 %1, %9, %11 offset:512 sc1\n\t"                                                                   \
    "global_load_dwordx2 %2, %9, %11 offset:1024 sc1\n\t"                                                                  \
    "global_load_dwordx2 %3, %9, %11 offset:1536 sc1\n\t"                                                                  \
    "global_load_dwordx2 %4, %9, %11 offset:2048 sc1\n\t"                                                                  \
    "global_load_dwordx2 %5, %9, %11 offset:2560 sc1\n\t"                                                                  \
    "global_load_dwordx2 %6, %9, %11 offset:3072 sc1\n\t"                                                                  \
    "global_load_dwordx2 %7, %9, %11 offset:3584 sc1\n\t"                                                                  \
    "global_load_dwordx2 %8, %10, %11 sc1"
// all nine loads in ONE burst, the base pointer in scalar registers; no wait: they fly under the previous step's update
__device__ __forceinline__ void alt_stage_issue(const AltArgs &a, AltStage &x) {
    constexpr int EDP = 4 * 64;
    const int lane = threadIdx.x & 63;
    const unsigned o0 = (unsigned)lane * 8u, ob = (unsigned)(2 * EDP + (lane < a.E ? lane : 0)) * 8u;
    asm volatile(ALT_GRANULE_LOADS
                 : "=&v"(x.g[0]), "=&v"(x.g[1]), "=&v"(x.g[2]), "=&v"(x.g[3]), "=&v"(x.g[4]), "=&v"(x.g[5]), "=&v"(x.g[6]),
                   "=&v"(x.g[7]), "=&v"(x.g[8])
                 : "v"(o0), "v"(ob), "s"(a.pub)
                 : "memory");
    // (the class weights are final before the launch -- stat_envs runs between runs -- and ride in the same burst)
    x.cw = ((a.flags & INVPREF_WEIGHTS_BY_ENV) && lane < a.E) ? a.weights[lane] : 1.f;
}
__device__ __forceinline__ void alt_stage_finish(const AltArgs &a, AltStage &x, int gen, float *sEv, float *sW, float *sb, float *scw) {
    constexpr int EMAX = 4, EDP = 4 * 64;
    const int lane = threadIdx.x & 63;
    bool on[4], onb;
    alt_stage_lanes(a, lane, on, onb);
    if (ALT_TEST_BAD_TAG) gen ^= 0x40000000;
    const unsigned o0 = (unsigned)lane * 8u, ob = (unsigned)(2 * EDP + (onb ? lane : 0)) * 8u;
    // (the loaded registers pass THROUGH the wait: nothing the compiler schedules can read them before it)
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(x.g[0]), "+v"(x.g[1]), "+v"(x.g[2]), "+v"(x.g[3]), "+v"(x.g[4]), "+v"(x.g[5]), "+v"(x.g[6]), "+v"(x.g[7]),
                   "+v"(x.g[8])
                 :
                 : "memory");
    int polls = 0;
    bool timed_out = false;
    for (;;) {
        bool ok = !onb || (int)(x.g[8] >> 32) == gen;
#pragma unroll
        for (int i = 0; i < 4; i++)
            ok = ok && (!on[i] || ((int)(x.g[i] >> 32) == gen && (int)(x.g[4 + i] >> 32) == gen));
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
        if (++polls > ALT_POLL_MAX) {
            // gave up: the error word is STICKY (only AltWorkspace.reset_error() clears it; the managers read it with every
            // loss read-back and raise), and the tables staged below are poisoned with NaN instead of whatever stale granules
            // the last poll returned -- a run that timed out cannot train on quietly on wrong small tables
            if (lane == 0) st_sc1(a.fold_flags + 63, 1);
            timed_out = true;
            break;
        }
        __builtin_amdgcn_s_sleep(4);
        asm volatile(ALT_GRANULE_LOADS "\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(x.g[0]), "=&v"(x.g[1]), "=&v"(x.g[2]), "=&v"(x.g[3]), "=&v"(x.g[4]), "=&v"(x.g[5]), "=&v"(x.g[6]),
                       "=&v"(x.g[7]), "=&v"(x.g[8])
                     : "v"(o0), "v"(ob), "s"(a.pub)
                     : "memory");
    }
    if (timed_out) {
#pragma unroll
        for (int i = 0; i < 9; i++) x.g[i] = 0x7fc00000ull;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        sEv[lane + 64 * i] = on[i] ? __builtin_bit_cast(float, (unsigned)x.g[i]) : 0.f;
        sW[lane + 64 * i] = on[i] ? __builtin_bit_cast(float, (unsigned)x.g[4 + i]) : 0.f;
    }
    if (lane < EMAX) sb[lane] = onb ? __builtin_bit_cast(float, (unsigned)x.g[8]) : 0.f;
    if (lane < EMAX) scw[lane] = x.cw;
}
// a run's first launch: the tables are final in memory (the previous run's flush launch ended before this one began)
__device__ __forceinline__ void alt_stage_plain(const AltArgs &a, float *sEv, float *sW, float *sb, float *scw, bool pure) {
    constexpr int EMAX = 4, DP = 64;
    const int lane = threadIdx.x & 63;
    bool on[4], onb;
    alt_stage_lanes(a, lane, on, onb);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int idx = lane + 64 * i, e = idx / DP, d = idx - e * DP;
        const bool o = on[i] && !pure;
        sEv[idx] = o ? a.Ev[e * a.D + d] : 0.f;
        sW[idx] = o ? a.W[e * a.D + d] : 0.f;
    }
    if (lane < EMAX) sb[lane] = (onb && !pure) ? a.b[lane] : 0.f;
    if (lane < EMAX) scw[lane] = ((a.flags & INVPREF_WEIGHTS_BY_ENV) && onb && !pure) ? a.weights[lane] : 1.f;
}

// geometry of a job workgroup of THREADS threads: THREADS / 16 group slots (16 or 32: a hot row's interactions spread over up
// to 32 slices -- the evaluation is bound by vector-instruction issue, ~340 instructions per interaction, so a slice's
// length is what a hot row costs), one partial slab row per group, a 4 KB landing / slot area per wave
template <int THREADS>
struct AltGeo {
    static constexpr int NG = THREADS / 16, WAVES = THREADS / 64, DP = 64, EMAX = 4;
    static constexpr int SLAB = 2 * EMAX * DP + EMAX + kLossSlots;
    static constexpr int sEv = 0, sW = sEv + EMAX * DP, sb = sW + EMAX * DP;
    static constexpr int scw = sb + EMAX;                     // [EMAX] class weights (INVPREF_WEIGHTS_BY_ENV; else ones)
    static constexpr int mv = scw + EMAX;                     // [WAVES][4][64] float4
    static constexpr int red = mv + WAVES * 4 * 64 * 4;       // [NG][SLAB]
    static constexpr int total = red + NG * SLAB;
};

__device__ __forceinline__ void adam4_prev(float4 &p, float4 g, float4 &m, float4 &v, const AdamScalars &ad, float2 prev2) {
    AdamScalars ap = ad;
    ap.step_size = prev2.x; ap.bc2_sqrt = prev2.y;
    adam4(p, g, m, v, ap);
}

// =====================================================================================
// rounds of jobs of the evaluating side
// =====================================================================================
template <bool VEC, bool FULL, int MODE, int THREADS>
__device__ __forceinline__ void alt_task(const AltArgs &a, int r0, int slab_index, float *lds,
                                         const AdamScalars &ad_cur, float2 prev2, const StepScalars &k, int gen) {
    constexpr int LG = 16, EMAX = 4, UE = 2;
    using G = AltGeo<THREADS>;
    using L = AltGeo<THREADS>;
    static_assert(STEP_LDS_DW && !STEP_NO_DMA && EvalLds<16, 4>::total == AltGeo<256>::total, "alt_task: the default smallest-instance layout");
    constexpr int NG = G::NG, DP = G::DP;
    float *sEv = lds + L::sEv, *sW = lds + L::sW, *sb = lds + L::sb, *scw = lds + L::scw;
    float4 *mv = reinterpret_cast<float4 *>(lds + L::mv);
    float *red = lds + L::red;
    const int lg = threadIdx.x & (LG - 1), grp = threadIdx.x / LG, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4 *mv_wave = mv + wave * 4 * 64;
    const bool implicit = a.flags & INVPREF_IMPLICIT;
    const bool rw_rec = a.flags & INVPREF_REWEIGHT_REC, rw_cls = a.flags & INVPREF_REWEIGHT_CLS;
    const bool by_env = a.flags & INVPREF_WEIGHTS_BY_ENV;   // weight = class_weights[env], staged in LDS: no per-interaction load
    const bool reg_env = a.flags & INVPREF_REG_ENV_EMBED;
    const bool pure = a.flags & INVPREF_PURE_MF;
    const bool dense = (a.flags & INVPREF_DENSE_REG) && !(a.flags & INVPREF_REG_ONLY_EMBED) && !pure;
    constexpr bool has_prev = MODE & 1, has_cur = MODE & 2;
    const int D = a.D;
    float *own0 = a.own_p[0], *own1 = a.own_p[1];
    const float *oth0 = a.oth_p[0], *oth1 = a.oth_p[1];

    ASTAMP(0);
    int4 d = a.desc[(r0 * NG + grp) * 2], d1 = a.desc[(r0 * NG + grp) * 2 + 1];
    // (the round's pending ranges travel with its descriptor: one round trip, not two)
    int4 pd = make_int4(0, 0, 0, 0);
    if (has_prev) pd = a.pend[r0 * NG + grp];
    if (has_cur) {
        constexpr int ZR4 = (2 * EMAX * DP + EMAX) / 4;   // (dEv | dW | db: 516 floats, SLAB = 524: both multiples of 4)
        static_assert((2 * EMAX * DP + EMAX) % 4 == 0 && G::SLAB % 4 == 0, "16-byte zeroing");
        for (int i = threadIdx.x; i < NG * ZR4; i += THREADS)
            *reinterpret_cast<float4 *>(red + (i / ZR4) * G::SLAB + (i % ZR4) * 4) = f4zero();
    }
    ASTAMP(1);
    float accLi = 0.f, accLe = 0.f, accLc = 0.f, accL2 = 0.f, accL1 = 0.f;
    float *slab = a.slabs + (int64_t)slab_index * G::SLAB;
    auto slot_of = [&](int g) {
        constexpr int GW = 64 / LG;
        return lds + L::mv + (g / GW) * (4 * 64 * 4) + (g % GW) * 2 * DP;
    };

    {   // ONE round per task (InvPrefAltPlan.rounds_per_task == 1: no round loop, nothing carried across the interaction loop)
        constexpr int nr = 1;
        const int r = r0;
        const int4 dd = d, dd1 = d1;
        const int4 pdd = pd;
        const int row = dd.x, meta = dd.y;
        const bool active = row >= 0, leader = meta & 1;
        const int slices_f = (meta >> 1) & 31, mode = (meta >> 6) & 7;
        const int slices = slices_f ? slices_f : 32;   // (32 slices, rounds of 32 slots: the field holds 0)
        const bool rpend = has_prev && (meta < 0);            // bit 31: some row of the round has pending rows
        const int cnt_i = (meta >> 9) & 0x3fffff;
        const int nsmp = (active && has_cur) ? (mode == 7 ? dd.w - dd.z : mode) : 0;
        if (r == r0 + STAMP_ROUND) ASTAMP(2);
        const int s_lo = dd.z, s_hi1 = max(dd.w - 1, dd.z);
        const int *dwords = reinterpret_cast<const int *>(a.desc + (r * NG + grp) * 2);
        auto sample_at = [&](int sidx) {
            USample sm;
            if (FULL) {
                const int *src = mode == 7 ? reinterpret_cast<const int *>(a.list + min(s_lo + sidx, s_hi1)) : dwords + 2 + 3 * min(sidx, 1);
                sm.oth = src[0]; sm.ps = src[1]; sm.y = __builtin_bit_cast(float, src[2]);
            } else if (mode == 7) {
                const int4 q = a.list[dd.z + sidx];
                sm.oth = q.x; sm.ps = q.y; sm.y = __builtin_bit_cast(float, q.z);
            } else if (sidx == 0) { sm.oth = dd.z; sm.ps = dd.w; sm.y = __builtin_bit_cast(float, dd1.x); }
            else { sm.oth = dd1.y; sm.ps = dd1.z; sm.y = __builtin_bit_cast(float, dd1.w); }
            return sm;
        };
        // ---- everything that depends only on the descriptor goes out together: own rows + moments, the first pending
        // rows, the first interactions' partner rows
        const int rowc = active ? row : 0;
        float4 oi = row4<VEC, FULL>(own0, rowc, D, lg), oe = f4zero();
        float4 mi = row4<VEC, FULL>(a.own_m[0], rowc, D, lg), vi = row4<VEC, FULL>(a.own_v[0], rowc, D, lg);
        float4 me = f4zero(), ve = f4zero();
        if (!pure) {
            oe = row4<VEC, FULL>(own1, rowc, D, lg);
            me = row4<VEC, FULL>(a.own_m[1], rowc, D, lg); ve = row4<VEC, FULL>(a.own_v[1], rowc, D, lg);
        }
        // pending contribution rows of this slice: contiguous pairs [pa, pb), two register sets
        constexpr int H = ALT_PEND_DEPTH / 2;
        const int npend = (active && has_prev) ? pdd.y - pdd.x : 0;
        const float *pbase = a.pend_rows + (unsigned)(npend > 0 ? pdd.x : 0) * (unsigned)(2 * DP) + lg * 4;
        float4 ci[2][H], ce[2][H];
        auto pfetch = [&](int set, int s0) {
#pragma unroll
            for (int j = 0; j < H; j++) {
                const int sj = s0 + j < npend ? s0 + j : (npend > 0 ? npend - 1 : 0);
                const float *p = pbase + (unsigned)sj * (unsigned)(2 * DP);
                ci[set][j] = *reinterpret_cast<const float4 *>(p);
                ce[set][j] = pure ? f4zero() : *reinterpret_cast<const float4 *>(p + DP);
            }
        };
        float4 gpi = f4zero(), gpe = f4zero();
        auto padd = [&](int set, int s0) {
#pragma unroll
            for (int j = 0; j < H; j++) {
                const bool has = s0 + j < npend;
                f4add(gpi, has ? ci[set][j] : f4zero());
                if (!pure) f4add(gpe, has ? ce[set][j] : f4zero());
            }
        };
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int j = 0; j < H; j++) ci[s][j] = ce[s][j] = f4zero();
        // (wave-uniform branches: a wave whose slices have nothing pending -- most user-side waves -- adds no loads to the
        //  launch's first burst, and the second register set only goes out for slices of more than H pairs)
        int n_wave = 0;
        if (rpend) {
            n_wave = npend;
#pragma unroll
            for (int g = 0; g < 64 / LG; g++) n_wave = max(n_wave, __builtin_amdgcn_readlane(npend, g * LG));
#if ALT_PEND_COND
            if (n_wave > 0) pfetch(0, 0);
            if (n_wave > H) pfetch(1, H);
#else
            pfetch(0, 0);
            pfetch(1, H);
#endif
        }

        struct Slot {
            float4 qi, qa;
            USample sm;
            int e, cs;
            float w;
        };
        Slot sl[UE];
        USample idn[UE];
        auto gather = [&](Slot &q, const USample &sm) {
            q.sm = sm;
            q.qi = row4<VEC, FULL>(oth0, sm.oth, D, lg);
            if (FULL) {
                const unsigned pso = (unsigned)sm.ps;
                if (!pure) {
                    q.qa = row4<VEC, FULL>(oth1, sm.oth, D, lg);
                    q.e = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(a.envs) + pso * 8u);
                }
                if ((rw_rec || rw_cls) && !by_env) q.w = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.weights) + pso * 4u);
                q.cs = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(a.push_slot) + pso * 4u);
                return;
            }
            if (!pure) {
                q.qa = row4<VEC, FULL>(oth1, sm.oth, D, lg);
                q.e = (int)a.envs[sm.ps];
            }
            if ((rw_rec || rw_cls) && !by_env) q.w = a.weights[sm.ps];
            q.cs = a.push_slot[sm.ps];
        };
#pragma unroll
        for (int j = 0; j < UE; j++) {
            sl[j].qi = sl[j].qa = f4zero();
            sl[j].sm = USample{0, 0, 0.f};
            sl[j].e = sl[j].cs = 0;
            sl[j].w = 1.f;
            idn[j] = USample{0, 0, 0.f};
        }
        if (has_cur) {   // (workgroup-uniform)
            if (FULL) {
                USample ls[2 * UE];
#pragma unroll
                for (int j = 0; j < 2 * UE; j++) ls[j] = USample{0, 0, 0.f};
                if (__builtin_amdgcn_ballot_w64(mode == 7) != 0) {
#pragma unroll
                    for (int j = 0; j < 2 * UE; j++) {
                        const int4 q = a.list[mode == 7 ? min(s_lo + j, s_hi1) : 0];
                        ls[j] = USample{q.x, q.y, __builtin_bit_cast(float, q.z)};
                    }
                }
#pragma unroll
                for (int j = 0; j < UE; j++) {
                    USample sm = ls[j];
                    if (mode != 7) sm = j == 0 ? USample{dd.z, dd.w, __builtin_bit_cast(float, dd1.x)} : USample{dd1.y, dd1.z, __builtin_bit_cast(float, dd1.w)};
                    if (mode == 0 || !active) sm = USample{0, 0, 0.f};   // (a job without interactions: entry 0's rows, unused)
                    gather(sl[j], sm);
                    idn[j] = ls[UE + j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < UE; j++)
                    if (j < nsmp) gather(sl[j], sample_at(j));
#pragma unroll
                for (int j = 0; j < UE; j++)
                    if (UE + j < nsmp) idn[j] = sample_at(UE + j);
            }
        }
        // ---- (ii, first half) this step's small tables are requested as soon as this wave's first burst is in -- by then the
        // fold blocks have usually published them (they end ~5 us into a launch, the burst arrives ~6 us in) -- and fly under
        // the previous step's update; a request that comes too early is repeated at the point of use
        AltStage stg;
#pragma unroll
        for (int i = 0; i < 9; i++) stg.g[i] = 0ull;
        stg.cw = 1.f;
#if ALT_EARLY_STAGE
        if (has_cur && has_prev && !pure && wave == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            alt_stage_issue(a, stg);
        }
#endif
        // ---- (i) the previous step's update of this row
        if (has_prev) {
            if (rpend) {
                for (int s0 = 0; s0 < n_wave; s0 += 2 * H) {
                    padd(0, s0);
                    if (s0 + 2 * H < n_wave) pfetch(0, s0 + 2 * H);
                    padd(1, s0 + H);
                    if (s0 + 3 * H < n_wave) pfetch(1, s0 + 3 * H);
                }
                if (slices > 1) {   // the slices' shares meet: every slice forms the same fixed-order total
                    float *mine = slot_of(grp);
                    *reinterpret_cast<float4 *>(mine + lg * 4) = gpi;
                    *reinterpret_cast<float4 *>(mine + DP + lg * 4) = gpe;
                    __syncthreads();
                    const int lead = grp & ~(slices - 1);
                    if (VEC && slices >= 8) {
                        // a hot row (8 .. 32 slices): the LEADER sums the shares -- eight slices' LDS reads in flight at a
                        // time -- and hands the total back through its own slot (every slice reading every share is
                        // slices^2 reads: a third of a 32-slice workgroup's LDS time)
                        if (active && leader) {
                            gpi = gpe = f4zero();
#pragma nounroll
                            for (int s = 0; s < slices; s += 8) {
                                float4 xi[8], xe[8];
#pragma unroll
                                for (int j = 0; j < 8; j++) {
                                    xi[j] = *reinterpret_cast<const float4 *>(slot_of(lead + s + j) + lg * 4);
                                    xe[j] = *reinterpret_cast<const float4 *>(slot_of(lead + s + j) + DP + lg * 4);
                                }
#pragma unroll
                                for (int j = 0; j < 8; j++) { f4add(gpi, xi[j]); f4add(gpe, xe[j]); }
                            }
                            *reinterpret_cast<float4 *>(mine + lg * 4) = gpi;
                            *reinterpret_cast<float4 *>(mine + DP + lg * 4) = gpe;
                        }
                        __syncthreads();
                        gpi = *reinterpret_cast<const float4 *>(slot_of(lead) + lg * 4);
                        gpe = *reinterpret_cast<const float4 *>(slot_of(lead) + DP + lg * 4);
                    } else if (VEC && slices == 4) {   // (the four slices' LDS reads in flight together)
                        float4 xi[4], xe[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            xi[j] = *reinterpret_cast<const float4 *>(slot_of(lead + j) + lg * 4);
                            xe[j] = *reinterpret_cast<const float4 *>(slot_of(lead + j) + DP + lg * 4);
                        }
                        gpi = gpe = f4zero();
#pragma unroll
                        for (int j = 0; j < 4; j++) { f4add(gpi, xi[j]); f4add(gpe, xe[j]); }
                    } else {   // (two slices; the element-wise instances, short of registers, one slice at a time)
                        gpi = gpe = f4zero();
#pragma nounroll
                        for (int s = 0; s < slices; s++) {
                            f4add(gpi, *reinterpret_cast<const float4 *>(slot_of(lead + s) + lg * 4));
                            f4add(gpe, *reinterpret_cast<const float4 *>(slot_of(lead + s) + DP + lg * 4));
                        }
                    }
                    __syncthreads();   // (the moments are parked over the slots next)
                }
            }
            const float cp = (float)pdd.z;
            if (cp != 0.f) {
                f4fma(gpi, cp, reg_term(oi, a.r2_prev, a.r1_prev));
                f4fma(gpe, cp, reg_term(oe, a.r2_prev, a.r1_prev));
            }
            adam4_prev(oi, gpi, mi, vi, ad_cur, prev2);
            if (!pure) adam4_prev(oe, gpe, me, ve, ad_cur, prev2);
        }
        if (r == r0 + STAMP_ROUND) ASTAMP(3);
        if (!has_cur) {
            // a flush launch: the row is finished here
            if (active && leader) {
                put4<VEC, 0, FULL>(own0, row, D, lg, oi);
                put4<VEC, 0, FULL>(a.own_m[0], row, D, lg, mi);
                put4<VEC, 0, FULL>(a.own_v[0], row, D, lg, vi);
                if (!pure) {
                    put4<VEC, 0, FULL>(own1, row, D, lg, oe);
                    put4<VEC, 0, FULL>(a.own_m[1], row, D, lg, me);
                    put4<VEC, 0, FULL>(a.own_v[1], row, D, lg, ve);
                }
            }
            return;
        }
        // the moments are needed again when the row is finished: parked in the lane's own words of the wave's landing area
        mv_wave[0 * 64 + lane] = mi; mv_wave[1 * 64 + lane] = vi;
        if (!pure) { mv_wave[2 * 64 + lane] = me; mv_wave[3 * 64 + lane] = ve; }
        // ---- (ii) the small tables of this step
        if (r == r0) {
            if (wave == 0) {
#if !ALT_EARLY_STAGE
                if (has_prev && !pure) alt_stage_issue(a, stg);
#endif
                if (has_prev && !pure) alt_stage_finish(a, stg, gen, sEv, sW, sb, scw);
                else alt_stage_plain(a, sEv, sW, sb, scw, pure);
            }
            __syncthreads();
            ASTAMP(4);
        }
        // ---- (iii) one interaction: evaluate, accumulate the own rows' gradients, push the partner's contribution rows
        float4 gi = f4zero(), ge = f4zero();
        auto step = [&](const Slot &q, bool has) {
            const int e = q.e;
            if (has) {
                const float wq = by_env ? scw[e] : q.w;
                const float w_rec = rw_rec ? wq : 1.f, w_cls = rw_cls ? wq : 1.f;
                const float4 ev = *reinterpret_cast<const float4 *>(sEv + e * DP + lg * 4);
                Eval<EMAX> o;
#ifndef ALT_EVAL_KIND
#define ALT_EVAL_KIND -1   // (what-if knob: 1 = the implicit InvPref evaluation fixed at compile time -- one basic block)
#endif
                eval_interaction<LG, EMAX, ALT_EVAL_KIND>(o, oi, q.qi, oe, q.qa, ev, sW, sb, nullptr, a.E, e, q.sm.y, w_rec * k.invB,
                                           w_cls * k.invB, k, implicit, pure, lg);
                float4 gip;
                gip.x = o.g_p - k.alpha * o.gx.x; gip.y = o.g_p - k.alpha * o.gx.y;
                gip.z = o.g_p - k.alpha * o.gx.z; gip.w = o.g_p - k.alpha * o.gx.w;
                f4add(gi, f4mul(gip, q.qi));
                f4fma(ge, o.g_q, f4mul(q.qa, ev));
                float *cr = a.push_rows + (unsigned)q.cs * (unsigned)(2 * DP) + lg * 4;
                store4<ALT_PUSH_ST>(cr, f4mul(gip, oi));
                store4<ALT_PUSH_ST>(cr + DP, f4scale(o.g_q, f4mul(oe, ev)));
                float4 oo = f4scale(o.g_q, f4mul(oe, q.qa));
                if (reg_env) f4add(oo, reg_term(ev, 2.f * k.r2, 2.f * k.r1));
                float *mine = red + grp * G::SLAB;
#pragma unroll
                for (int c = 0; c < EMAX; c++) {
                    float4 *wr = reinterpret_cast<float4 *>(mine + EMAX * DP + (kClsQuads<LG, EMAX> ? ((lg >> 2) ^ c) : c) * DP + lg * 4);
                    float4 cur = *wr;
                    f4fma(cur, o.gz[c], o.x);
                    *wr = cur;
                }
                if (lg == 0) {
                    float4 *br = reinterpret_cast<float4 *>(mine + 2 * EMAX * DP);
                    float4 cur = *br;
                    cur.x += o.gz[0]; cur.y += o.gz[1]; cur.z += o.gz[2]; cur.w += o.gz[3];
                    *br = cur;
                }
                float4 *erow = reinterpret_cast<float4 *>(mine + e * DP + lg * 4);
                float4 cur = *erow;
                f4add(cur, oo);
                *erow = cur;
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
        };
        // (Tried in round 5: the slice's interactions evaluated two at a time in one basic block at two workgroups per CU, so
        //  that both chains interleave -- item-side launches 17.4 us against 16.7: one wave's evaluation is bound by vector
        //  instruction ISSUE, ~340 VALU instructions per interaction, not by dependent latencies; nothing to interleave)
        for (int s = 0; s < nsmp; s += UE) {
#pragma unroll
            for (int j = 0; j < UE; j++) {
                if (s + j < nsmp) step(sl[j], true);
                if (FULL || s + UE + j < nsmp) gather(sl[j], idn[j]);
                if (FULL || s + 2 * UE + j < nsmp) idn[j] = sample_at(s + 2 * UE + j);
            }
        }
        if (r == r0 + STAMP_ROUND) ASTAMP(5);
        const float cnt = (float)cnt_i;
        if (active && leader) {
            accL2 += cnt * (f4sq(oi) + f4sq(oe));
            accL1 += cnt * (f4abs(oi) + f4abs(oe));
        }
        const bool last = r == r0 + nr - 1;
        if (last) {
            accLi = row16_sum(accLi); accLe = row16_sum(accLe); accLc = row16_sum(accLc);
            accL2 = row16_sum(accL2); accL1 = row16_sum(accL1);
            if (lg == 0) {
                float *ls = red + grp * G::SLAB + 2 * EMAX * DP + EMAX;
                ls[0] = accLi; ls[1] = accLe; ls[2] = accLc; ls[3] = accL2; ls[4] = accL1;
                ls[5] = ls[6] = ls[7] = 0.f;
            }
            if (dense && slab_index == 0 && wave == 0) {
                // report of the classifier regulariser (models.py:211-217; the tables of THIS step's forward, still staged):
                // ||W||^2 / (D E) + ||b||^2 / E, the same with |.|  -- by the task that owns slab 0, into group 0's loss slots
                // (group 0 is on this wave: its own stores above come first in program order)
                const float4 w4 = *reinterpret_cast<const float4 *>(sW + lane * 4);
                float w2 = f4sq(w4), w1 = f4abs(w4);
                const float bb = lane < EMAX ? sb[lane] : 0.f;
                float b2 = bb * bb, b1 = fabsf(bb);
                w2 = wave_sum_valu(w2); w1 = wave_sum_valu(w1); b2 = wave_sum_valu(b2); b1 = wave_sum_valu(b1);
                if (lane == 0) {
                    float *ls = red + 2 * EMAX * DP + EMAX;
                    ls[5] = w2 / ((float)a.D * (float)a.E) + b2 / (float)a.E;
                    ls[6] = w1 / ((float)a.D * (float)a.E) + b1 / (float)a.E;
                }
            }
        }
        // ---- the slices of a row meet (the leader takes the parked moments out first: same wave, program order)
        if (active && leader) {
            mi = mv_wave[0 * 64 + lane]; vi = mv_wave[1 * 64 + lane];
            if (!pure) { me = mv_wave[2 * 64 + lane]; ve = mv_wave[3 * 64 + lane]; }
        }
        asm volatile("" ::: "memory");
        if (slices > 1) {
            float *mine = slot_of(grp);
            *reinterpret_cast<float4 *>(mine + lg * 4) = gi;
            *reinterpret_cast<float4 *>(mine + DP + lg * 4) = ge;
        }
        if (slices > 1 || last) __syncthreads();
        if (last) {
            // (the groups' rows summed in group order, four columns per thread: 16-byte LDS reads, eight rows in flight)
            static_assert(G::SLAB % 4 == 0 && G::SLAB / 4 <= THREADS && NG % 8 == 0, "slab sum: one float4 column set per thread");
            if ((int)threadIdx.x < G::SLAB / 4) {
                const float *col = red + threadIdx.x * 4;
                float4 sum = *reinterpret_cast<const float4 *>(col);
#pragma unroll
                for (int q0 = 0; q0 < NG; q0 += 8) {
                    float4 x[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) x[j] = *reinterpret_cast<const float4 *>(col + (q0 + j) * G::SLAB);
#pragma unroll
                    for (int j = (q0 == 0 ? 1 : 0); j < 8; j++) f4add(sum, x[j]);
                }
                store4<ALT_SLAB_ST>(slab + threadIdx.x * 4, sum);
            }
        }
        if (slices > 1) {
            if (active && leader) {
                // (slices is a power of two: the other slices' partials in batches of up to four -- eight LDS reads in flight,
                //  then the adds in slice order; one read + add at a time made a 16-slice meet 2 us of LDS latencies)
                if (VEC && slices >= 8) {
                    float4 xi[8], xe[8];
#pragma unroll
                    for (int j = 1; j < 8; j++) {
                        xi[j] = *reinterpret_cast<const float4 *>(slot_of(grp + j) + lg * 4);
                        xe[j] = *reinterpret_cast<const float4 *>(slot_of(grp + j) + DP + lg * 4);
                    }
#pragma unroll
                    for (int j = 1; j < 8; j++) { f4add(gi, xi[j]); f4add(ge, xe[j]); }
#pragma nounroll
                    for (int s = 8; s < slices; s += 8) {
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            xi[j] = *reinterpret_cast<const float4 *>(slot_of(grp + s + j) + lg * 4);
                            xe[j] = *reinterpret_cast<const float4 *>(slot_of(grp + s + j) + DP + lg * 4);
                        }
#pragma unroll
                        for (int j = 0; j < 8; j++) { f4add(gi, xi[j]); f4add(ge, xe[j]); }
                    }
                } else if (VEC && slices == 4) {
                    float4 xi[4], xe[4];
#pragma unroll
                    for (int j = 1; j < 4; j++) {
                        xi[j] = *reinterpret_cast<const float4 *>(slot_of(grp + j) + lg * 4);
                        xe[j] = *reinterpret_cast<const float4 *>(slot_of(grp + j) + DP + lg * 4);
                    }
#pragma unroll
                    for (int j = 1; j < 4; j++) { f4add(gi, xi[j]); f4add(ge, xe[j]); }
                } else {
#pragma nounroll
                    for (int s = 1; s < slices; s++) {
                        f4add(gi, *reinterpret_cast<const float4 *>(slot_of(grp + s) + lg * 4));
                        f4add(ge, *reinterpret_cast<const float4 *>(slot_of(grp + s) + DP + lg * 4));
                    }
                }
            }
            if (!last) __syncthreads();
        }
        if (r == r0 + STAMP_ROUND) ASTAMP(6);
        // ---- (iv) the leader finishes the row: this step's update, one store of p, m, v
        if (active && leader) {
            if (cnt != 0.f) {
                f4fma(gi, cnt, reg_term(oi, k.r2, k.r1));
                f4fma(ge, cnt, reg_term(oe, k.r2, k.r1));
            }
            adam4(oi, gi, mi, vi, ad_cur);
            put4<VEC, ALT_ROW_ST, FULL>(own0, row, D, lg, oi);
            put4<VEC, ALT_MV_ST, FULL>(a.own_m[0], row, D, lg, mi);
            put4<VEC, ALT_MV_ST, FULL>(a.own_v[0], row, D, lg, vi);
            if (!pure) {
                adam4(oe, ge, me, ve, ad_cur);
                put4<VEC, ALT_ROW_ST, FULL>(own1, row, D, lg, oe);
                put4<VEC, ALT_MV_ST, FULL>(a.own_m[1], row, D, lg, me);
                put4<VEC, ALT_MV_ST, FULL>(a.own_v[1], row, D, lg, ve);
            }
        }
    }
    ASTAMP(7);
}

// =====================================================================================
// rows of the evaluating side without a job: pending rows (few), then both updates -- streamed, 2 rows per group
// =====================================================================================
template <bool VEC, bool FULL, int MODE, int THREADS>
__device__ __forceinline__ void alt_stream(const AltArgs &a, const int4 *rows, int n, const AdamScalars &ad_cur, float2 prev2) {
    constexpr int R = 2, LG = 16, NG = THREADS / LG, DP = 64, H = 2;
    const int lg = threadIdx.x & (LG - 1), grp = threadIdx.x / LG;
    const bool pure = a.flags & INVPREF_PURE_MF;
    constexpr bool has_prev = MODE & 1, has_cur = MODE & 2;
    const int D = a.D;
    const int iters = (n + R * NG - 1) / (R * NG);
    for (int it = 0; it < iters; it++) {
        int4 e[R];
        bool on[R];
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int idx = grp + (it * R + q) * NG;
            on[q] = idx < n;
            e[q] = rows[on[q] ? idx : 0];
#ifdef ALT_DIAG_SKIP_UNTOUCHED   // (what-if build, WRONG results: rows without pending pairs cost nothing -- the bound of a deferred Adam)
            if (on[q] && e[q].z - e[q].y <= 0) { on[q] = false; e[q] = rows[0]; }
#endif
        }
        float4 p[2 * R], m[2 * R], v[2 * R], ci[R][H], ce[R][H];
        int npend[R];
#pragma unroll
        for (int q = 0; q < 2 * R; q++) {
            p[q] = m[q] = v[q] = f4zero();
            if (!(pure && (q & 1))) {
#if defined(ALT_STREAM_NT)   // (A/B knob: the streamed rows are read once per two launches -- non-temporal loads)
                if (FULL) {
                    typedef float nt4 __attribute__((ext_vector_type(4)));
                    const unsigned off = (unsigned)e[q >> 1].x * 256u + (unsigned)lg * 16u;
                    const nt4 pp = __builtin_nontemporal_load(reinterpret_cast<const nt4 *>(reinterpret_cast<const char *>(a.own_p[q & 1]) + off));
                    const nt4 mm = __builtin_nontemporal_load(reinterpret_cast<const nt4 *>(reinterpret_cast<const char *>(a.own_m[q & 1]) + off));
                    const nt4 vv = __builtin_nontemporal_load(reinterpret_cast<const nt4 *>(reinterpret_cast<const char *>(a.own_v[q & 1]) + off));
                    p[q] = make_float4(pp.x, pp.y, pp.z, pp.w); m[q] = make_float4(mm.x, mm.y, mm.z, mm.w); v[q] = make_float4(vv.x, vv.y, vv.z, vv.w);
                } else
#endif
                {
                p[q] = row4<VEC, FULL>(a.own_p[q & 1], e[q >> 1].x, D, lg);
                m[q] = row4<VEC, FULL>(a.own_m[q & 1], e[q >> 1].x, D, lg);
                v[q] = row4<VEC, FULL>(a.own_v[q & 1], e[q >> 1].x, D, lg);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < R; q++) {
            npend[q] = (on[q] && has_prev) ? e[q].z - e[q].y : 0;
#pragma unroll
            for (int j = 0; j < H; j++) ci[q][j] = ce[q][j] = f4zero();
        }
        int n_wave = 0;
        if (has_prev) {
#pragma unroll
            for (int q = 0; q < R; q++) {
                const float *pb = a.pend_rows + (unsigned)(npend[q] > 0 ? e[q].y : 0) * (unsigned)(2 * DP) + lg * 4;
#pragma unroll
                for (int j = 0; j < H; j++) {
                    const int sj = j < npend[q] ? j : (npend[q] > 0 ? npend[q] - 1 : 0);
                    ci[q][j] = *reinterpret_cast<const float4 *>(pb + (unsigned)sj * (unsigned)(2 * DP));
                    if (!pure) ce[q][j] = *reinterpret_cast<const float4 *>(pb + (unsigned)sj * (unsigned)(2 * DP) + DP);
                }
#pragma unroll
                for (int g = 0; g < 64 / LG; g++) n_wave = max(n_wave, __builtin_amdgcn_readlane(npend[q], g * LG));
            }
        }
#pragma unroll
        for (int q = 0; q < R; q++) {
            float4 gi = f4zero(), ge = f4zero();
            if (has_prev) {
#pragma unroll
                for (int j = 0; j < H; j++) {
                    f4add(gi, j < npend[q] ? ci[q][j] : f4zero());
                    if (!pure) f4add(ge, j < npend[q] ? ce[q][j] : f4zero());
                }
                // (rows with more pending pairs than the first burst: rare -- the plan turns long ones into jobs)
                const float *pb = a.pend_rows + (unsigned)(npend[q] > 0 ? e[q].y : 0) * (unsigned)(2 * DP) + lg * 4;
                for (int s0 = H; s0 < n_wave; s0 += H) {
                    float4 xi[H], xe[H];
#pragma unroll
                    for (int j = 0; j < H; j++) {
                        const int sj = s0 + j < npend[q] ? s0 + j : (npend[q] > 0 ? npend[q] - 1 : 0);
                        xi[j] = *reinterpret_cast<const float4 *>(pb + (unsigned)sj * (unsigned)(2 * DP));
                        xe[j] = pure ? f4zero() : *reinterpret_cast<const float4 *>(pb + (unsigned)sj * (unsigned)(2 * DP) + DP);
                    }
#pragma unroll
                    for (int j = 0; j < H; j++) {
                        f4add(gi, s0 + j < npend[q] ? xi[j] : f4zero());
                        if (!pure) f4add(ge, s0 + j < npend[q] ? xe[j] : f4zero());
                    }
                }
                const float cp = (float)e[q].w;
                if (cp != 0.f) {
                    f4fma(gi, cp, reg_term(p[2 * q], a.r2_prev, a.r1_prev));
                    f4fma(ge, cp, reg_term(p[2 * q + 1], a.r2_prev, a.r1_prev));
                }
                adam4_prev(p[2 * q], gi, m[2 * q], v[2 * q], ad_cur, prev2);
                if (!pure) adam4_prev(p[2 * q + 1], ge, m[2 * q + 1], v[2 * q + 1], ad_cur, prev2);
            }
            if (has_cur) {
                adam4(p[2 * q], f4zero(), m[2 * q], v[2 * q], ad_cur);
                if (!pure) adam4(p[2 * q + 1], f4zero(), m[2 * q + 1], v[2 * q + 1], ad_cur);
            }
            if (on[q]) {
#pragma unroll
                for (int tb = 0; tb < 2; tb++) {
                    if (pure && tb) continue;
                    put4<VEC, STEP_STREAM_ST, FULL>(a.own_p[tb], e[q].x, D, lg, p[2 * q + tb]);
                    put4<VEC, STEP_STREAM_ST, FULL>(a.own_m[tb], e[q].x, D, lg, m[2 * q + tb]);
                    put4<VEC, STEP_STREAM_ST, FULL>(a.own_v[tb], e[q].x, D, lg, v[2 * q + tb]);
                }
            }
        }
    }
}

// =====================================================================================
// fold block fb: 16 columns of the previous launch's partial slabs -> gradient of embed_env / classifier -> Adam IN PLACE
// (write-through stores), the six loss outputs of the previous step, the block's flag
// =====================================================================================
template <int THREADS>
__device__ __forceinline__ void alt_fold_block(const AltArgs &a, int fb, float *lds, const AdamScalars &ad, int gen) {
    constexpr int kFoldSubs = THREADS / kFoldCols;   // (shadows the 256-thread constant: more sub-rows, fewer loads per thread)
    constexpr int DP = 64, EMAX = 4, SLAB = 2 * EMAX * DP + EMAX + kLossSlots, EDP = EMAX * DP;
    double *part = reinterpret_cast<double *>(lds);   // [kFoldSubs][kFoldCols] (+ kLossSlots)
    const int colx = threadIdx.x % kFoldCols, sub = threadIdx.x / kFoldCols;
    const int idx = fb * kFoldCols + colx;
    const bool pure = a.flags & INVPREF_PURE_MF;
    const bool dense = (a.flags & INVPREF_DENSE_REG) && !(a.flags & INVPREF_REG_ONLY_EMBED) && !pure;
    const bool mine = idx < SLAB;
    const bool isLoss = idx >= 2 * EDP + EMAX, isB = !isLoss && idx >= 2 * EDP, isW = !isLoss && !isB && idx >= EDP;
    const int rr = isW ? idx - EDP : idx;
    const int e = isB ? idx - 2 * EDP : rr / DP, dd = isB ? 0 : rr - e * DP;
    const bool live = mine && !isLoss && e < a.E && dd < a.D && !pure;
    const int off = live ? (isB ? e : e * a.D + dd) : 0;
    float pre_p = 0.f, pre_m = 0.f, pre_v = 0.f;
    if (sub == 0 && live) {
        pre_p = (isB ? a.b : (isW ? a.W : a.Ev))[off];
        pre_m = (isB ? a.mb : (isW ? a.mW : a.mEv))[off];
        pre_v = (isB ? a.vb : (isW ? a.vW : a.vEv))[off];
    }
    const float *col = a.slabs_prev + (mine ? idx : 0);
    const int np = a.n_partials_prev;
    double acc = 0.0;
    constexpr int CH = STEP_FOLD_CH;
    for (int s0 = sub; s0 < np; s0 += CH * kFoldSubs) {
        float x[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) x[j] = col[(int64_t)min(s0 + j * kFoldSubs, np - 1) * SLAB];
#pragma unroll
        for (int j = 0; j < CH; j++) acc += (s0 + j * kFoldSubs < np) ? (double)x[j] : 0.0;
    }
    part[sub * kFoldCols + colx] = acc;
    __syncthreads();
    if (sub == 0 && mine) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < kFoldSubs; q++) v += part[q * kFoldCols + colx];
        if (!isLoss) {
            if (live) {
                float gv = (float)v, pv = pre_p;
                if (isB) {
                    if (dense) gv += 2.f * a.l2 / (float)a.E * pv + a.l1 / (float)a.E * c_sign(pv);
                } else if (isW && dense) {
                    gv += 2.f * a.l2 / ((float)a.D * (float)a.E) * pv + a.l1 / ((float)a.D * (float)a.E) * c_sign(pv);
                }
                float mm = pre_m, vv = pre_v;
                adam1(pv, gv, mm, vv, ad);
                (isB ? a.b : (isW ? a.W : a.Ev))[off] = pv;          // (the next launch's fold reads it: plain)
                __hip_atomic_store(a.pub + idx, ((unsigned long long)(unsigned)gen << 32) | (unsigned long long)__builtin_bit_cast(unsigned, pv),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // {value, step}: one 8-byte write-through store
                (isB ? a.mb : (isW ? a.mW : a.mEv))[off] = mm;
                (isB ? a.vb : (isW ? a.vW : a.vEv))[off] = vv;
            }
        } else {
            double *sl = part + kFoldSubs * kFoldCols;
            sl[idx - 2 * EDP - EMAX] = v;
        }
    }
    __syncthreads();   // (the loss columns meet in LDS)
    constexpr int loss0 = 2 * EDP + EMAX;
    if (fb == loss0 / kFoldCols && threadIdx.x == 0 && a.losses_prev) {
        const double *sl = part + kFoldSubs * kFoldCols;
        const StepScalars &k = a.k;
        const double Li = sl[0] * a.inv_B_prev, Le = sl[1] * a.inv_B_prev, Lc = sl[2] * a.inv_B_prev;
        const double L2 = sl[3] * a.inv_BD2_prev + (dense ? sl[5] : 0.0), L1 = sl[4] * a.inv_BD2_prev + (dense ? sl[6] : 0.0);
        atomicAdd(a.losses_prev + 0, (float)Li); atomicAdd(a.losses_prev + 1, (float)Le); atomicAdd(a.losses_prev + 2, (float)Lc);
        atomicAdd(a.losses_prev + 3, (float)L2); atomicAdd(a.losses_prev + 4, (float)L1);
        atomicAdd(a.losses_prev + 5, (float)((double)k.ca * Li + (double)k.cb * Le + (double)k.cc * Lc + (double)a.l2 * L2 + (double)a.l1 * L1));
    }
}

template <bool VEC, bool FULL, int MODE, int THREADS = 256>
__global__ __launch_bounds__(THREADS, THREADS == 256 ? 3 : 1) void mstep_alt_kernel(AltArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // (the previous step's Adam scalars differ from this step's in step_size / bc2_sqrt only: two scalars, not a second set)
    AdamScalars ad_cur = a.ad_cur;
    float2 prev2 = make_float2(a.ad_prev.step_size, a.ad_prev.bc2_sqrt);
    StepScalars k = a.k;
    int gen = a.gen;
    if (a.sched_state) {
        const int *cur = a.sched_state + 16 * a.sched_slot;
        const SchedRow *sr = reinterpret_cast<const SchedRow *>(cur + 2);
        gen = cur[0];
        ad_cur = sr->ad;
        // (a flush launch runs in the slot of the step it finishes: that row IS the "previous" step)
        prev2 = (MODE & 2) ? make_float2(__builtin_bit_cast(float, cur[10]), __builtin_bit_cast(float, cur[11]))
                           : make_float2(sr->ad.step_size, sr->ad.bc2_sqrt);
        const float al = sr->alpha;
        if (al == al) k.alpha = al;
    }
    const int b = (int)blockIdx.x;
    if (b < a.first_task_block) {
        if (!(MODE & 1) && b == 0 && threadIdx.x < 63) {
            // a run starts (no fold in this launch, nobody polls): every flag word back to zero, so that no stale flag of an
            // earlier run can look like one of this run's step numbers.  (In the kernel, not a hipMemsetAsync in front of
            // it: a captured memset node replayed garbage into these words on ROCm 7.0.2 -- tools/alt_soak.py)
            // Word 63, the error word, is NOT touched: a time-out stays visible until the host has seen it.
            a.fold_flags[threadIdx.x] = 0;
        }
        if (!(MODE & 1) && b == 0) {
            constexpr int NPUB = 2 * 4 * 64 + 4;
            for (int i = threadIdx.x; i < NPUB; i += THREADS) a.pub[i] = 0ull;
        }
        if (b < a.fold_blocks) {
            if (MODE & 1) {
                ASTAMP(0);
                { AdamScalars adp = ad_cur; adp.step_size = prev2.x; adp.bc2_sqrt = prev2.y; alt_fold_block<THREADS>(a, b, lds, adp, gen); }
                ASTAMP(7);
            }
        } else if (b == a.fold_blocks) {
            // the device-side schedule moves on (nobody reads the other slot before the next launch)
            if (a.sched_state && (MODE & 2) && threadIdx.x == 0) {
                const int *cur = a.sched_state + 16 * a.sched_slot;
                int *nxt = a.sched_state + 16 * (a.sched_slot ^ 1);
                const int next = cur[0] + 1, base = cur[1], idx = next - base;
                nxt[0] = next;
                nxt[1] = base;
                if (idx >= 0 && idx < a.sched_n) *reinterpret_cast<SchedRow *>(nxt + 2) = a.sched_table[idx];
                nxt[10] = cur[2];   // this step's step_size / bc2_sqrt: the next launch's "previous step"
                nxt[11] = cur[3];
            }
        }
        return;
    }
    const int tb = b - a.first_task_block;
    const int ncls = a.n_cls;
    const int c = tb % ncls;
    int j = tb / ncls;
    int q[4] = {0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
        const int on = (c == kk) ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 4; i++) q[i] += a.cls[kk][i] * on;
    }
    const int rpt = a.rounds_per_task, spt = a.rows_per_stream_task;
    const int tj = (q[1] + rpt - 1) / rpt;
    if (j < tj) {
        alt_task<VEC, FULL, MODE, THREADS>(a, q[0] + j, q[0] + j, lds, ad_cur, prev2, k, gen);
        return;
    }
    j -= tj;
    if (j * spt < q[3]) {
        ASTAMP(0);
        // (started a little late: the rows without a job are off the launch's critical chain, and their load burst would
        //  queue in front of the jobs' first gathers and the fold blocks' slab reads -- as in the two-launch form)
        if (MODE & 2) stream_delay<ALT_STREAM_DELAY>();
        alt_stream<VEC, FULL, MODE, THREADS>(a, a.stream + q[2] + j * spt, min(spt, q[3] - j * spt), ad_cur, prev2);
        ASTAMP(7);
    }
}

// ---- host side
inline size_t alt_half_floats(int n_cap, int partials_cap) {
    const size_t rows = ((size_t)(n_cap > 0 ? n_cap : 0) + 1) * 2 * 64;
    const size_t slabs = (size_t)(partials_cap > 0 ? partials_cap : 1) * (2 * 4 * 64 + 4 + kLossSlots);
    return (rows + slabs + 63) & ~(size_t)63;
}
constexpr size_t kAltTailBytes = 64 * sizeof(int) + ((2 * 4 * 64 + 4) * 8 + 255) / 256 * 256;   // fold flags | published granules
inline size_t alt_flags_offset(int n_cap, int partials_cap) { return 2 * alt_half_floats(n_cap, partials_cap) * sizeof(float); }

int launch_alt(const InvPrefTables *tables, const InvPrefTables *exp_avg, const InvPrefTables *exp_avg_sq,
               const InvPrefAltPlan *plan, const int64_t *envs, const float *weights, int64_t batch_norm,
               int64_t batch_norm_prev, const InvPrefCoefs *coefs, uint32_t flags, float *losses6_prev, int64_t step, double lr,
               double beta1, double beta2, double eps, const InvPrefAdamSchedule *sched, void *workspace,
               size_t workspace_bytes, int n_cap, int partials_cap, int parity, hipStream_t st) {
    const bool pure = flags & INVPREF_PURE_MF;
    int rc;
    if ((rc = check_tables(tables, pure)) || (rc = check_tables(exp_avg, pure)) || (rc = check_tables(exp_avg_sq, pure))) return rc;
    if (!plan || !coefs || !workspace) return INVPREF_EINVAL;
    if (sched && (!sched->state || !sched->table || sched->n <= 0)) return INVPREF_EINVAL;
    const int D = (int)tables->factor_num, E = (int)tables->env_num;
    if (D > 64 || E > 4) return INVPREF_EUNSUPPORTED;
    const bool has_prev = plan->has_prev != 0, has_cur = plan->has_cur != 0;
    if (!has_prev && !has_cur) return INVPREF_EINVAL;
    if (has_cur && (batch_norm <= 0 || (!envs && !pure) || plan->n <= 0)) return INVPREF_EINVAL;
    if (has_prev && batch_norm_prev <= 0) return INVPREF_EINVAL;
    if (!sched && step < 1) return INVPREF_EINVAL;
    if (pure && (flags & (INVPREF_REWEIGHT_CLS | INVPREF_REG_ENV_EMBED))) return INVPREF_EINVAL;
    if (has_cur && (flags & (INVPREF_REWEIGHT_REC | INVPREF_REWEIGHT_CLS)) && !weights) return INVPREF_EINVAL;
    if ((flags & INVPREF_WEIGHTS_BY_ENV) && pure) return INVPREF_EINVAL;
    if ((uint64_t)(tables->user_num > tables->item_num ? tables->user_num : tables->item_num) * (uint64_t)D * 4ull >= (1ull << 32))
        return INVPREF_EUNSUPPORTED;
    if (plan->lanes_per_group != 16 || (plan->slots_per_round != 16 && plan->slots_per_round != 32) || plan->side < 0 || plan->side > 1 || plan->n < 0 || plan->n_prev < 0 ||
        plan->n_rounds < 0 || plan->rounds_per_task != 1 || plan->n_rounds % plan->rounds_per_task != 0 ||
        plan->n_stream < 0 || plan->rows_per_stream_task <= 0 || (plan->n_rounds > 0 && (!plan->desc || !plan->pend)) ||
        (plan->n > 0 && (!plan->list || !plan->push_slot)) || (plan->n_stream > 0 && !plan->stream) ||
        plan->n > n_cap || plan->n_prev > n_cap || plan->n_partials_prev < 0 || plan->n_partials_prev > partials_cap ||
        (has_cur && plan->n_rounds > partials_cap) || (parity != 0 && parity != 1))   // (a flush launch writes no slabs)
        return INVPREF_EINVAL;
    const int ncls = plan->n_classes > 0 ? plan->n_classes : 1;
    if (ncls > 8) return INVPREF_EINVAL;
    int per_class = 0;
    for (int c = 0; c < ncls; c++) {
        const int32_t *q = plan->cls[c];
        for (int i = 0; i < 4; i++) if (q[i] < 0) return INVPREF_EINVAL;
        if (q[0] + q[1] > plan->n_rounds || q[2] + q[3] > plan->n_stream || q[0] % plan->rounds_per_task) return INVPREF_EINVAL;
        const int tot = (q[1] + plan->rounds_per_task - 1) / plan->rounds_per_task +
                        (q[3] + plan->rows_per_stream_task - 1) / plan->rows_per_stream_task;
        per_class = tot > per_class ? tot : per_class;
    }
    if (workspace_bytes < alt_flags_offset(n_cap, partials_cap) + kAltTailBytes) return INVPREF_EWORKSPACE;
    const size_t half = alt_half_floats(n_cap, partials_cap);
    const size_t rows_floats = ((size_t)n_cap + 1) * 2 * 64;
    float *ws = (float *)workspace;
    AltArgs a{};
    const int s = plan->side;
    float *P[4] = {tables->embed_user_invariant, tables->embed_item_invariant, tables->embed_user_env_aware, tables->embed_item_env_aware};
    float *M[4] = {exp_avg->embed_user_invariant, exp_avg->embed_item_invariant, exp_avg->embed_user_env_aware, exp_avg->embed_item_env_aware};
    float *V[4] = {exp_avg_sq->embed_user_invariant, exp_avg_sq->embed_item_invariant, exp_avg_sq->embed_user_env_aware, exp_avg_sq->embed_item_env_aware};
    for (int tb = 0; tb < 2; tb++) {
        a.own_p[tb] = P[2 * tb + s]; a.own_m[tb] = M[2 * tb + s]; a.own_v[tb] = V[2 * tb + s];
        a.oth_p[tb] = P[2 * tb + (1 - s)];
    }
    a.Ev = tables->embed_env; a.W = tables->classifier_weight; a.b = tables->classifier_bias;
    a.mEv = exp_avg->embed_env; a.mW = exp_avg->classifier_weight; a.mb = exp_avg->classifier_bias;
    a.vEv = exp_avg_sq->embed_env; a.vW = exp_avg_sq->classifier_weight; a.vb = exp_avg_sq->classifier_bias;
    a.E = E; a.D = D;
    a.desc = reinterpret_cast<const int4 *>(plan->desc);
    a.pend = reinterpret_cast<const int4 *>(plan->pend);
    a.list = reinterpret_cast<const int4 *>(plan->list);
    a.push_slot = plan->push_slot;
    a.stream = reinterpret_cast<const int4 *>(plan->stream);
    a.rounds_per_task = plan->rounds_per_task; a.rows_per_stream_task = plan->rows_per_stream_task; a.n_cls = ncls;
    for (int c = 0; c < 8; c++) for (int i = 0; i < 4; i++) a.cls[c][i] = c < ncls ? plan->cls[c][i] : 0;
    a.envs = envs; a.weights = weights;
    StepScalars k{};
    k.ca = coefs->invariant_coe; k.cb = coefs->env_aware_coe; k.cc = coefs->env_coe; k.alpha = coefs->alpha;
    if (has_cur) {
        k.invB = 1.0f / (float)batch_norm;
        k.r2 = coefs->L2_coe / ((float)batch_norm * (float)D);
        k.r1 = coefs->L1_coe / (2.0f * (float)batch_norm * (float)D);
    }
    a.k = k;
    if (has_prev) {
        a.r2_prev = coefs->L2_coe / ((float)batch_norm_prev * (float)D);
        a.r1_prev = coefs->L1_coe / (2.0f * (float)batch_norm_prev * (float)D);
        a.inv_B_prev = 1.0 / (double)batch_norm_prev;
        a.inv_BD2_prev = 1.0 / ((double)batch_norm_prev * (double)D * 2.0);
    }
    a.flags = flags;
    a.mode = (has_prev ? 1 : 0) | (has_cur ? 2 : 0);
    if (!sched) {
        // has_cur: `step` is the current step, the previous one is step - 1; a flush finishes `step` itself
        a.ad_cur = adam_scalars(step, lr, beta1, beta2, eps);
        a.ad_prev = has_cur ? adam_scalars(step > 1 ? step - 1 : 1, lr, beta1, beta2, eps) : a.ad_cur;
        a.gen = (int)step;
    } else {
        a.sched_state = sched->state; a.sched_slot = sched->slot & 1;
        a.sched_table = reinterpret_cast<const SchedRow *>(sched->table); a.sched_n = sched->n;
    }
    float *mine = ws + (size_t)parity * half, *other = ws + (size_t)(1 - parity) * half;
    a.push_rows = mine; a.slabs = mine + rows_floats;
    a.pend_rows = other; a.slabs_prev = other + rows_floats;
    a.n_partials_prev = plan->n_partials_prev;
    a.fold_flags = reinterpret_cast<int *>(reinterpret_cast<char *>(workspace) + alt_flags_offset(n_cap, partials_cap));
    a.pub = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(workspace) + alt_flags_offset(n_cap, partials_cap) + 64 * sizeof(int));
    a.l2 = coefs->L2_coe; a.l1 = coefs->L1_coe;
    a.losses_prev = has_prev ? losses6_prev : nullptr;
    constexpr int SLAB = 2 * 4 * 64 + 4 + kLossSlots;
    a.fold_blocks = (SLAB + kFoldCols - 1) / kFoldCols;
    if (a.fold_blocks > 62) return INVPREF_EUNSUPPORTED;
    a.first_task_block = (a.fold_blocks + 1 + 7) & ~7;   // (a multiple of the XCD count: task block tb keeps class tb % 8 on XCD b % 8)
    static const char *stamp_env = getenv("INVPREF_STAMPS");
    a.stamps = stamp_env ? reinterpret_cast<unsigned long long *>(strtoull(stamp_env, nullptr, 16)) : nullptr;
    static const bool nodrain = getenv("INVPREF_STAMPS_NODRAIN") != nullptr;
    a.stamps_nodrain = nodrain;
    const int slots = plan->slots_per_round;
    const size_t lds_job = sizeof(float) * (slots == 32 ? AltGeo<512>::total : AltGeo<256>::total);
    const size_t lds_fold = ((size_t)(slots * 16 / kFoldCols) * kFoldCols + kLossSlots) * sizeof(double);
    const size_t lds = lds_job > lds_fold ? lds_job : lds_fold;
    const int grid = a.first_task_block + per_class * ncls;
    const bool vec = vec_ok(tables) && vec_ok(exp_avg) && vec_ok(exp_avg_sq);
    static const bool no_full = getenv("INVPREF_NO_FULL") != nullptr && getenv("INVPREF_NO_FULL")[0] == '1';
    const bool full = vec && D == 64 && !no_full;
#define CALL_ALT_M(VECV, FULLV, MODEV, THR)                                                                      \
    do {                                                                                                         \
        if ((rc = ensure_lds(mstep_alt_kernel<VECV, FULLV, MODEV, THR>, lds))) return rc;                        \
        hipLaunchKernelGGL((mstep_alt_kernel<VECV, FULLV, MODEV, THR>), dim3(grid), dim3(THR), lds, st, a);      \
    } while (0)
#define CALL_ALT_T(VECV, FULLV, THR)                                                 \
    do {                                                                             \
        if (a.mode == 3) CALL_ALT_M(VECV, FULLV, 3, THR);                            \
        else if (a.mode == 2) CALL_ALT_M(VECV, FULLV, 2, THR);                       \
        else CALL_ALT_M(VECV, FULLV, 1, THR);                                        \
    } while (0)
#define CALL_ALT(VECV, FULLV)                                                        \
    do {                                                                             \
        if (slots == 32) CALL_ALT_T(VECV, FULLV, 512); else CALL_ALT_T(VECV, FULLV, 256); \
    } while (0)
    if (full) CALL_ALT(true, true);
    else if (vec) CALL_ALT(true, false);
    else CALL_ALT(false, false);
#undef CALL_ALT
#undef CALL_ALT_T
#undef CALL_ALT_M
    return (int)hipGetLastError();
}
