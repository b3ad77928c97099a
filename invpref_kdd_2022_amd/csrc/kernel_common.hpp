// kernel_common.hpp -- device helpers shared by the InvPref kernels (row layout, gathers, dots).
//
// Work layout shared by every per-interaction kernel: one interaction is owned by a 16-lane row
// of a wavefront (4 interactions per wave64).  Lane l of the row holds the float4 chunks
// l, l+16, l+32, ... of each gathered embedding row, so a row of D=64 floats is ONE coalesced
// 256-byte global_load_dwordx4 across the 16 lanes, dot products are a per-lane fma chain plus a
// 4-step DPP butterfly (canon_math.hpp), and no LDS is needed for the reductions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/invpref_hip.h"
#include "canon_math.hpp"

// Ordering of LDS traffic between the lanes of ONE wavefront (a lane reads what another lane of the same
// wave wrote): LDS operations of a wave execute in issue order, so it is enough to keep the compiler from
// reordering across this point and to drain the LDS counter.  A wavefront-scope __builtin_amdgcn_fence also
// waits for every outstanding GLOBAL operation (vmcnt(0)) -- with float atomics in flight that is
// microseconds per fence (measured: 2.5 us per interaction in the user-side path).
#define WAVE_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

namespace invpref {

constexpr int kRow = 16;          // lanes per interaction
constexpr int kLossSlots = 8;     // 5 used

struct DevTables {
    const float *Pu, *Qi, *Pa, *Qa, *Ev, *W, *b;
    int U, I, E, D;
};
struct DevGrads {
    float *Pu, *Qi, *Pa, *Qa, *Ev, *W, *b;
};

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }

// gather one embedding row into the lane's NC chunks (zero beyond D)
template <int NC, bool VEC>
__device__ __forceinline__ void load_row(const float *__restrict__ base, int64_t row, int D, int l16, float4 (&r)[NC]) {
    const float *p = base + row * (int64_t)D;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int i0 = (l16 + kRow * c) * 4;
        if (VEC) {
            r[c] = (i0 < D) ? *reinterpret_cast<const float4 *>(p + i0) : f4zero();
        } else {
            r[c].x = (i0 + 0 < D) ? p[i0 + 0] : 0.f;
            r[c].y = (i0 + 1 < D) ? p[i0 + 1] : 0.f;
            r[c].z = (i0 + 2 < D) ? p[i0 + 2] : 0.f;
            r[c].w = (i0 + 3 < D) ? p[i0 + 3] : 0.f;
        }
    }
}
// a row of an LDS-resident [E][DP] table (DP = NC*64, zero padded)
template <int NC>
__device__ __forceinline__ void lds_row(const float *tab, int e, int l16, float4 (&r)[NC]) {
#pragma unroll
    for (int c = 0; c < NC; c++) r[c] = *reinterpret_cast<const float4 *>(tab + e * (NC * 64) + (l16 + kRow * c) * 4);
}

template <int NC>
__device__ __forceinline__ float dot2(const float4 (&a)[NC], const float4 (&b)[NC]) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        s = __builtin_fmaf(a[c].x, b[c].x, s);
        s = __builtin_fmaf(a[c].y, b[c].y, s);
        s = __builtin_fmaf(a[c].z, b[c].z, s);
        s = __builtin_fmaf(a[c].w, b[c].w, s);
    }
    return row16_sum(s);
}
template <int NC>
__device__ __forceinline__ float dot3(const float4 (&a)[NC], const float4 (&b)[NC], const float4 (&cc)[NC]) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        s = __builtin_fmaf(a[c].x * b[c].x, cc[c].x, s);
        s = __builtin_fmaf(a[c].y * b[c].y, cc[c].y, s);
        s = __builtin_fmaf(a[c].z * b[c].z, cc[c].z, s);
        s = __builtin_fmaf(a[c].w * b[c].w, cc[c].w, s);
    }
    return row16_sum(s);
}

// stage a small [E][D] table into LDS as [E][DP] zero padded (an absent table, INVPREF_PURE_MF, stages zeros)
__device__ __forceinline__ void stage_table(float *dst, const float *__restrict__ src, int E, int D, int DP) {
    for (int i = threadIdx.x; i < E * DP; i += blockDim.x) {
        const int e = i / DP, d = i - e * DP;
        dst[i] = (src && d < D) ? src[e * D + d] : 0.f;
    }
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}



struct StepScalars {
    float ca, cb, cc, alpha, invB, r2, r1;
};

struct AdamScalars {
    float step_size, bc2_sqrt, w1, b2, w2, eps;
};
__device__ __forceinline__ void adam1(float &p, float g, float &m, float &v, const AdamScalars &a) {
    m = m + a.w1 * (g - m);
    v = v * a.b2 + (a.w2 * g) * g;
    const float denom = __builtin_sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p + ((-a.step_size) * m) / denom;
}

// the same update with the hardware sqrt / reciprocal units (~1 ulp each) instead of the IEEE
// sequences: used where Adam is fused behind the M-step and instruction count, not bandwidth, is
// the price (the stand-alone adam_kernel keeps the exact form; it is bandwidth-bound anyway)
__device__ __forceinline__ void adam1f(float &p, float g, float &m, float &v, const AdamScalars &a) {
    m = m + a.w1 * (g - m);
    v = v * a.b2 + (a.w2 * g) * g;
    const float denom = __builtin_amdgcn_sqrtf(v) * __builtin_amdgcn_rcpf(a.bc2_sqrt) + a.eps;
    p = p + ((-a.step_size) * m) * __builtin_amdgcn_rcpf(denom);
}

// host-side helpers
inline int check_tables(const InvPrefTables *t, bool pure_mf = false) {
    if (!t) return INVPREF_EINVAL;
    if (t->user_num <= 0 || t->item_num <= 0 || t->env_num <= 0 || t->factor_num <= 0) return INVPREF_EINVAL;
    if (t->factor_num > INVPREF_MAX_FACTORS || t->env_num > INVPREF_MAX_ENVS) return INVPREF_EUNSUPPORTED;
    if (!t->embed_user_invariant || !t->embed_item_invariant) return INVPREF_EINVAL;
    if (pure_mf) return t->env_num == 1 ? 0 : INVPREF_EINVAL;  // INVPREF_PURE_MF: the other five tables are ignored
    if (!t->embed_user_env_aware || !t->embed_item_env_aware || !t->embed_env || !t->classifier_weight ||
        !t->classifier_bias)
        return INVPREF_EINVAL;
    return 0;
}
inline DevTables dev_tables(const InvPrefTables *t) {
    return DevTables{t->embed_user_invariant, t->embed_item_invariant, t->embed_user_env_aware, t->embed_item_env_aware,
                     t->embed_env, t->classifier_weight, t->classifier_bias,
                     (int)t->user_num, (int)t->item_num, (int)t->env_num, (int)t->factor_num};
}
inline DevGrads dev_grads(const InvPrefTables *t) {
    return DevGrads{t->embed_user_invariant, t->embed_item_invariant, t->embed_user_env_aware, t->embed_item_env_aware,
                    t->embed_env, t->classifier_weight, t->classifier_bias};
}
inline bool vec_ok(const InvPrefTables *t) {
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    return (t->factor_num % 4 == 0) && al(t->embed_user_invariant) && al(t->embed_item_invariant) &&
           al(t->embed_user_env_aware) && al(t->embed_item_env_aware);  // (null pointers count as aligned)
}
inline int nc_of(int D) { return D <= 64 ? 1 : (D <= 128 ? 2 : 4); }
inline int emax_of(int E) { return E <= 4 ? 4 : (E <= 8 ? 8 : 16); }

}  // namespace invpref
