// invpref_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the InvPref hot path and their
// C-ABI launchers (include/invpref_hip.h).  No torch types, no CPU fallback.
//
// Work layout shared by every per-interaction kernel: one interaction is owned by a 16-lane row
// of a wavefront (4 interactions per wave64).  Lane l of the row holds the float4 chunks
// l, l+16, l+32, ... of each gathered embedding row, so a row of D=64 floats is ONE coalesced
// 256-byte global_load_dwordx4 across the 16 lanes, dot products are a per-lane fma chain plus a
// 4-step DPP butterfly (canon_math.hpp), and no LDS is needed for the reductions.  The E x D
// environment table and classifier live in LDS for the whole workgroup.
#include "kernel_common.hpp"

using namespace invpref;

namespace {

// =====================================================================================
// forward  (models.py:307-326 / :448-467)
// =====================================================================================
template <int NC, bool VEC, int EMAX>
__global__ __launch_bounds__(256) void forward_kernel(DevTables t, const int64_t *__restrict__ users,
                                                      const int64_t *__restrict__ items,
                                                      const int64_t *__restrict__ envs, int64_t B, uint32_t flags,
                                                      float *__restrict__ inv, float *__restrict__ envaware,
                                                      float *__restrict__ envout) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int DP = NC * 64;
    float *sEv = lds, *sW = lds + t.E * DP, *sb = sW + t.E * DP;
    stage_table(sEv, t.Ev, t.E, t.D, DP);
    stage_table(sW, t.W, t.E, t.D, DP);
    for (int i = threadIdx.x; i < t.E; i += blockDim.x) sb[i] = t.b[i];
    __syncthreads();
    const int l16 = threadIdx.x & 15;
    const int64_t rows_per_block = blockDim.x / kRow;
    const bool implicit = flags & INVPREF_IMPLICIT;
    for (int64_t s = blockIdx.x * rows_per_block + (threadIdx.x >> 4); s < B; s += gridDim.x * rows_per_block) {
        const int64_t u = users[s], v = items[s];
        const int e = (int)envs[s];
        float4 pu[NC], qi[NC], pa[NC], qa[NC], ev[NC];
        load_row<NC, VEC>(t.Pu, u, t.D, l16, pu);
        load_row<NC, VEC>(t.Qi, v, t.D, l16, qi);
        load_row<NC, VEC>(t.Pa, u, t.D, l16, pa);
        load_row<NC, VEC>(t.Qa, v, t.D, l16, qa);
        lds_row<NC>(sEv, e, l16, ev);
        const float p = dot2<NC>(pu, qi), q = dot3<NC>(pa, qa, ev);
        float s_inv, s_env;
        if (implicit) { const float sp = c_sigmoid(p); s_inv = sp; s_env = sp * c_sigmoid(q); }
        else { s_inv = p; s_env = p + q; }
        float4 x[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) x[c] = f4mul(pu[c], qi[c]);
        float z[EMAX], mx = -__builtin_inff();
#pragma unroll
        for (int c = 0; c < EMAX; c++) {
            if (c < t.E) {
                float4 w[NC];
                lds_row<NC>(sW, c, l16, w);
                z[c] = dot2<NC>(x, w) + sb[c];
                mx = z[c] > mx ? z[c] : mx;
            }
        }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < EMAX; c++) if (c < t.E) se += c_exp(z[c] - mx);
        const float lse = c_log(se);
        if (l16 == 0) { inv[s] = s_inv; envaware[s] = s_env; }
#pragma unroll
        for (int c = 0; c < EMAX; c++) if (c < t.E && l16 == (c & 15)) envout[s * t.E + c] = (z[c] - mx) - lse;
    }
}

// =====================================================================================
// M-step gradient, scatter-add by global float atomics (works on any minibatch, no plan)
//   train.py:94-156; analytic backward of SURVEY.md §8 a7'
// Per workgroup: gEv/gW/gb and the five loss sums are accumulated in LDS and written as one
// partial slab; mstep_finish_kernel folds the slabs (no same-address global atomics).
// =====================================================================================

// UPSTREAM = true turns the same kernel into the plain backward of forward(): the per-sample
// upstream gradients (d invariant_score, d env_aware_score, d env_outputs) come from memory instead
// of from the fused losses, and no regulariser / loss sums are formed.
struct Upstream {
    const float *d_inv, *d_env, *d_out;  // [B], [B], [B,E]; any may be null (= zeros)
};
// DCOL (large E*D): the E x D partials are not accumulated with LDS atomics (one lane per clock) but by
// "records": every 16-lane group stores its interaction's x = Pu*Qi, o = g_q*Pa*Qa(+reg) and gz[0..E) in
// LDS, and after a barrier every thread adds all records into the outputs it owns (one column d, a fixed
// set of classes c) held in registers -- an outer-product accumulation on the vector ALU.
template <int NC, bool VEC, int EMAX, bool UPSTREAM, bool DCOL>
__global__ __launch_bounds__(512) void mstep_atomic_kernel(DevTables t, DevGrads g, const int64_t *__restrict__ users,
                                                           const int64_t *__restrict__ items,
                                                           const int64_t *__restrict__ envs,
                                                           const float *__restrict__ scores,
                                                           const float *__restrict__ weights, int64_t B,
                                                           StepScalars k, uint32_t flags, float *__restrict__ slabs,
                                                           Upstream up) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int DP = NC * 64;
    const int EDP = t.E * DP;
    const int G = blockDim.x / kRow;
    // region R holds the aEv|aW accumulators; in DCOL mode the interaction records [G][2][DP] live there
    // until the very end (the accumulators are registers then), so R = max(2*EDP, G*2*DP) floats
    const int R = DCOL ? (2 * EDP > G * 2 * DP ? 2 * EDP : G * 2 * DP) : 2 * EDP;
    float *sEv = lds, *sW = sEv + EDP, *aEv = sW + EDP, *aW = aEv + EDP;
    float *sb = aEv + R, *ab = sb + EMAX, *aL = ab + EMAX;  // aL[kLossSlots]
    float *rec = aEv;                                     // [G][2][DP]
    float *recs = aL + kLossSlots;                        // [G][EMAX + 1]  gz[0..EMAX), env id (or -1: no record)
    const int slab_len = 2 * EDP + EMAX + kLossSlots;
    stage_table(sEv, t.Ev, t.E, t.D, DP);
    stage_table(sW, t.W, t.E, t.D, DP);
    for (int i = threadIdx.x; i < 2 * EDP; i += blockDim.x) aEv[i] = 0.f;  // aEv and aW are adjacent
    for (int i = threadIdx.x; i < EMAX; i += blockDim.x) { sb[i] = (i < t.E) ? t.b[i] : 0.f; ab[i] = 0.f; }
    if (threadIdx.x < kLossSlots) aL[threadIdx.x] = 0.f;
    __syncthreads();

    const int l16 = threadIdx.x & 15;
    const int64_t rows_per_block = blockDim.x / kRow;
    const bool implicit = flags & INVPREF_IMPLICIT;
    const bool rw_rec = flags & INVPREF_REWEIGHT_REC, rw_cls = flags & INVPREF_REWEIGHT_CLS;
    const bool reg_env = flags & INVPREF_REG_ENV_EMBED;
    const bool no_grad = flags & INVPREF_NO_GRAD;
    float accLi = 0.f, accLe = 0.f, accLc = 0.f, accL2 = 0.f, accL1 = 0.f;
    // DCOL ownership: thread -> column d_own and classes [c_lo, c_lo + CPT)
    constexpr int CG = (256 / DP) < EMAX ? (256 / DP) : EMAX;   // class groups (block of 256 threads)
    constexpr int CPT = (EMAX + CG - 1) / CG;
    const int d_own = threadIdx.x % DP, c_lo = (threadIdx.x / DP) * CPT;
    float dW[DCOL ? CPT : 1], dE[DCOL ? CPT : 1], dB[DCOL ? CPT : 1];
#pragma unroll
    for (int i = 0; i < (DCOL ? CPT : 1); i++) dW[i] = dE[i] = dB[i] = 0.f;
    const int grp_id = threadIdx.x >> 4;
    // every workgroup runs the same number of iterations (the scatter below is a whole-wave affair and
    // the DCOL accumulation has barriers); a group past the end of the minibatch just contributes nothing
    const int64_t n_iter = (B + (int64_t)gridDim.x * rows_per_block - 1) / ((int64_t)gridDim.x * rows_per_block);
    // scatter staging: per wave 4 gradient rows + their 4 destination row ids
    float *scat = recs + (DCOL ? G * (EMAX + 1) : 0);
    long long *scat_ids = reinterpret_cast<long long *>(scat + (blockDim.x >> 6) * 4 * DP);
    const int wv = threadIdx.x >> 6, gw = (threadIdx.x >> 4) & 3, lane = threadIdx.x & 63;

    for (int64_t it = 0, s = blockIdx.x * rows_per_block + (threadIdx.x >> 4); it < n_iter;
         it++, s += gridDim.x * rows_per_block) {
      const bool valid = s < B;
      if (DCOL && l16 == 0) recs[grp_id * (EMAX + 1) + EMAX] = __builtin_bit_cast(float, -1);
      // the four embedding-row gradients (+ per-occurrence L2/L1 terms) of this group's interaction
      float4 gPu4[NC], gQi4[NC], gPa4[NC], gQa4[NC];
#pragma unroll
      for (int j = 0; j < NC; j++) gPu4[j] = gQi4[j] = gPa4[j] = gQa4[j] = f4zero();
      long long su = -1, sv = -1;
      if (valid) {
        const int64_t u = users[s], v = items[s];
        const int e = (int)envs[s];
        const float y = UPSTREAM ? 0.f : scores[s];
        const float w = (!UPSTREAM && (rw_rec || rw_cls)) ? weights[s] : 1.f;
        const float w_rec = rw_rec ? w : 1.f, w_cls = rw_cls ? w : 1.f;
        const float cw_rec = w_rec * k.invB, cw_cls = w_cls * k.invB;
        float4 pu[NC], qi[NC], pa[NC], qa[NC], ev[NC];
        load_row<NC, VEC>(t.Pu, u, t.D, l16, pu);
        load_row<NC, VEC>(t.Qi, v, t.D, l16, qi);
        load_row<NC, VEC>(t.Pa, u, t.D, l16, pa);
        load_row<NC, VEC>(t.Qa, v, t.D, l16, qa);
        lds_row<NC>(sEv, e, l16, ev);
        const float p = dot2<NC>(pu, qi), q = dot3<NC>(pa, qa, ev);
        float g_p, g_q, li = 0.f, le = 0.f;
        if (UPSTREAM) {
            const float ui = up.d_inv ? up.d_inv[s] : 0.f, ue = up.d_env ? up.d_env[s] : 0.f;
            if (implicit) {
                const float sp = c_sigmoid(p), sq = c_sigmoid(q);
                g_p = (ui + ue * sq) * (sp * (1.f - sp));
                g_q = ue * sp * (sq * (1.f - sq));
            } else {
                g_p = ui + ue;
                g_q = ue;
            }
        } else if (implicit) {
            const float sp = c_sigmoid(p), sq = c_sigmoid(q), sv = sp * sq;
            li = c_bce(sp, y);
            le = c_bce(sv, y);
            const float d_inv = k.ca * cw_rec * c_dbce(sp, y);
            const float d_env = k.cb * cw_rec * c_dbce(sv, y);
            g_p = (d_inv + d_env * sq) * (sp * (1.f - sp));
            g_q = d_env * sp * (sq * (1.f - sq));
        } else {
            const float s2 = p + q;
            li = (p - y) * (p - y);
            le = (s2 - y) * (s2 - y);
            const float d_env = k.cb * cw_rec * 2.f * (s2 - y);
            g_p = k.ca * cw_rec * 2.f * (p - y) + d_env;
            g_q = d_env;
        }
        // classifier on x = Pu*Qi
        float4 x[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) x[c] = f4mul(pu[c], qi[c]);
        float z[EMAX], mx = -__builtin_inff();
#pragma unroll
        for (int c = 0; c < EMAX; c++) {
            z[c] = -__builtin_inff();
            if (c < t.E) {
                float4 wr[NC];
                lds_row<NC>(sW, c, l16, wr);
                z[c] = dot2<NC>(x, wr) + sb[c];
                mx = z[c] > mx ? z[c] : mx;
            }
        }
        float se = 0.f, zr = 0.f;
#pragma unroll
        for (int c = 0; c < EMAX; c++) {
            if (c < t.E) { const float dz = z[c] - mx; zr = (c == e) ? dz : zr; z[c] = c_exp(dz); se += z[c]; }
        }
        // z[c] now holds exp(z-mx); NLL of log_softmax as the reference forms it (models.py:206-209): finite for finite logits
        const float lcls = UPSTREAM ? 0.f : c_log(se) - zr;
        const float rse = 1.f / se;
        float usum = 0.f;  // sum_c d_out[c]  (log_softmax backward: dz = d_out - softmax * sum d_out)
        if (UPSTREAM && up.d_out) {
#pragma unroll
            for (int c = 0; c < EMAX; c++) if (c < t.E) usum += up.d_out[s * t.E + c];
        }
        float4 gx[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) gx[c] = f4zero();
#pragma unroll
        for (int c = 0; c < EMAX; c++) {
            if (c < t.E) {
                float gz;
                if (UPSTREAM) gz = up.d_out ? (up.d_out[s * t.E + c] - (z[c] * rse) * usum) : 0.f;
                else gz = k.cc * cw_cls * (z[c] * rse - (c == e ? 1.f : 0.f));
                float4 wr[NC];
                lds_row<NC>(sW, c, l16, wr);
#pragma unroll
                for (int j = 0; j < NC; j++) {
                    gx[j].x = __builtin_fmaf(gz, wr[j].x, gx[j].x);
                    gx[j].y = __builtin_fmaf(gz, wr[j].y, gx[j].y);
                    gx[j].z = __builtin_fmaf(gz, wr[j].z, gx[j].z);
                    gx[j].w = __builtin_fmaf(gz, wr[j].w, gx[j].w);
                    if (!no_grad && !DCOL) {
                        float *dst = aW + c * DP + (l16 + kRow * j) * 4;
                        atomicAdd(dst + 0, gz * x[j].x);
                        atomicAdd(dst + 1, gz * x[j].y);
                        atomicAdd(dst + 2, gz * x[j].z);
                        atomicAdd(dst + 3, gz * x[j].w);
                    }
                }
                if (l16 == 0 && !no_grad) {
                    if (DCOL) recs[grp_id * (EMAX + 1) + c] = gz;
                    else atomicAdd(ab + c, gz);
                }
            }
        }
        su = u; sv = v;
#pragma unroll
        for (int j = 0; j < NC; j++) {
            const int i0 = (l16 + kRow * j) * 4;
            if (i0 < t.D && !no_grad) {
                float4 gip, o;
                gip.x = g_p - k.alpha * gx[j].x; gip.y = g_p - k.alpha * gx[j].y;
                gip.z = g_p - k.alpha * gx[j].z; gip.w = g_p - k.alpha * gx[j].w;
#define REG(a) (k.r2 * (a) + k.r1 * c_sign(a))
                // the four gradient rows are kept for the shaped scatter below
                gPu4[j] = make_float4(gip.x * qi[j].x + REG(pu[j].x), gip.y * qi[j].y + REG(pu[j].y),
                                      gip.z * qi[j].z + REG(pu[j].z), gip.w * qi[j].w + REG(pu[j].w));
                gQi4[j] = make_float4(gip.x * pu[j].x + REG(qi[j].x), gip.y * pu[j].y + REG(qi[j].y),
                                      gip.z * pu[j].z + REG(qi[j].z), gip.w * pu[j].w + REG(qi[j].w));
                gPa4[j] = make_float4(g_q * (qa[j].x * ev[j].x) + REG(pa[j].x), g_q * (qa[j].y * ev[j].y) + REG(pa[j].y),
                                      g_q * (qa[j].z * ev[j].z) + REG(pa[j].z), g_q * (qa[j].w * ev[j].w) + REG(pa[j].w));
                gQa4[j] = make_float4(g_q * (pa[j].x * ev[j].x) + REG(qa[j].x), g_q * (pa[j].y * ev[j].y) + REG(qa[j].y),
                                      g_q * (pa[j].z * ev[j].z) + REG(qa[j].z), g_q * (pa[j].w * ev[j].w) + REG(qa[j].w));
#undef REG
                o = make_float4(g_q * (pa[j].x * qa[j].x), g_q * (pa[j].y * qa[j].y), g_q * (pa[j].z * qa[j].z),
                                g_q * (pa[j].w * qa[j].w));
                if (reg_env) {
                    o.x += 2.f * k.r2 * ev[j].x + 2.f * k.r1 * c_sign(ev[j].x);
                    o.y += 2.f * k.r2 * ev[j].y + 2.f * k.r1 * c_sign(ev[j].y);
                    o.z += 2.f * k.r2 * ev[j].z + 2.f * k.r1 * c_sign(ev[j].z);
                    o.w += 2.f * k.r2 * ev[j].w + 2.f * k.r1 * c_sign(ev[j].w);
                }
                if (DCOL) {
                    *reinterpret_cast<float4 *>(rec + (grp_id * 2) * DP + i0) = x[j];
                    *reinterpret_cast<float4 *>(rec + (grp_id * 2 + 1) * DP + i0) = o;
                } else {
                    float *dst = aEv + e * DP + i0;
                    atomicAdd(dst + 0, o.x); atomicAdd(dst + 1, o.y); atomicAdd(dst + 2, o.z); atomicAdd(dst + 3, o.w);
                }
            }
            if (UPSTREAM) continue;
            // regulariser REPORTS: users+items weigh 1/(2BD), env rows 1/(BD) -> count env terms twice
            float s2 = pu[j].x * pu[j].x + pu[j].y * pu[j].y + pu[j].z * pu[j].z + pu[j].w * pu[j].w;
            s2 += pa[j].x * pa[j].x + pa[j].y * pa[j].y + pa[j].z * pa[j].z + pa[j].w * pa[j].w;
            s2 += qi[j].x * qi[j].x + qi[j].y * qi[j].y + qi[j].z * qi[j].z + qi[j].w * qi[j].w;
            s2 += qa[j].x * qa[j].x + qa[j].y * qa[j].y + qa[j].z * qa[j].z + qa[j].w * qa[j].w;
            float s1 = fabsf(pu[j].x) + fabsf(pu[j].y) + fabsf(pu[j].z) + fabsf(pu[j].w);
            s1 += fabsf(pa[j].x) + fabsf(pa[j].y) + fabsf(pa[j].z) + fabsf(pa[j].w);
            s1 += fabsf(qi[j].x) + fabsf(qi[j].y) + fabsf(qi[j].z) + fabsf(qi[j].w);
            s1 += fabsf(qa[j].x) + fabsf(qa[j].y) + fabsf(qa[j].z) + fabsf(qa[j].w);
            if (reg_env) {
                s2 += 2.f * (ev[j].x * ev[j].x + ev[j].y * ev[j].y + ev[j].z * ev[j].z + ev[j].w * ev[j].w);
                s1 += 2.f * (fabsf(ev[j].x) + fabsf(ev[j].y) + fabsf(ev[j].z) + fabsf(ev[j].w));
            }
            accL2 += s2;
            accL1 += s1;
        }
        if (l16 == 0) { accLi += li * w_rec; accLe += le * w_rec; accLc += lcls * w_cls; }
        if (DCOL && l16 == 0 && !no_grad) recs[grp_id * (EMAX + 1) + EMAX] = __builtin_bit_cast(float, e);
      }  // valid
      // ---- scatter-add by global float atomics, shaped: one table at a time the wave's four gradient rows
      // go through LDS and every instruction adds 256 contiguous bytes of ONE row (the shape the atomic unit
      // runs at full rate with, MI355X_MICROARCH.md); a lane-per-float4 scatter touches 16-byte strides
      if (!no_grad) {
          float *sc = scat + wv * 4 * DP;
          long long *ids = scat_ids + wv * 4;
          auto scatter_table = [&](const float4 (&G4)[NC], long long rid, float *base) {
#pragma unroll
              for (int j = 0; j < NC; j++) *reinterpret_cast<float4 *>(sc + gw * DP + (l16 + kRow * j) * 4) = G4[j];
              if (l16 == 0) ids[gw] = rid;
              WAVE_LDS_FENCE();
              __builtin_amdgcn_wave_barrier();
#pragma unroll
              for (int g4 = 0; g4 < 4; g4++) {
                  const long long r = ids[g4];
                  if (r >= 0) {
#pragma unroll
                      for (int kk = 0; kk < NC; kk++) {
                          const int idx = kk * 64 + lane;
                          if (idx < t.D) atomicAdd(base + r * (long long)t.D + idx, sc[g4 * DP + idx]);
                      }
                  }
              }
              WAVE_LDS_FENCE();
              __builtin_amdgcn_wave_barrier();
          };
          scatter_table(gPu4, su, g.Pu);
          scatter_table(gQi4, sv, g.Qi);
          scatter_table(gPa4, su, g.Pa);
          scatter_table(gQa4, sv, g.Qa);
      }
      if (DCOL) {
          __syncthreads();
          if (threadIdx.x < CG * DP) {
#pragma unroll 1
              for (int g2 = 0; g2 < G; g2++) {
                  const float *rs = recs + g2 * (EMAX + 1);
                  const int er = __builtin_bit_cast(int, rs[EMAX]);
                  if (er < 0) continue;
                  const float xv = rec[(g2 * 2) * DP + d_own], ov = rec[(g2 * 2 + 1) * DP + d_own];
#pragma unroll
                  for (int i = 0; i < CPT; i++) {
                      const int c = c_lo + i;
                      if (c < t.E) {
                          const float gzc = rs[c];
                          dW[i] = __builtin_fmaf(gzc, xv, dW[i]);
                          dE[i] += (c == er) ? ov : 0.f;
                          dB[i] += gzc;
                      }
                  }
              }
          }
          __syncthreads();
      }
    }
    if (DCOL) {  // the owned outputs go where the LDS-atomic path keeps its accumulators
        for (int i = threadIdx.x; i < 2 * EDP; i += blockDim.x) aEv[i] = 0.f;   // (padding columns stay zero)
        __syncthreads();
        if (threadIdx.x < CG * DP) {
#pragma unroll
            for (int i = 0; i < CPT; i++) {
                const int c = c_lo + i;
                if (c < t.E) {
                    aEv[c * DP + d_own] = dE[i];
                    aW[c * DP + d_own] = dW[i];
                    if (d_own == 0) ab[c] = dB[i];
                }
            }
        }
    }
    // workgroup reduction of the loss sums
    accLi = wave_sum(accLi); accLe = wave_sum(accLe); accLc = wave_sum(accLc);
    accL2 = wave_sum(accL2); accL1 = wave_sum(accL1);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(aL + 0, accLi); atomicAdd(aL + 1, accLe); atomicAdd(aL + 2, accLc);
        atomicAdd(aL + 3, accL2); atomicAdd(aL + 4, accL1);
    }
    __syncthreads();
    float *slab = slabs + (int64_t)blockIdx.x * slab_len;
    for (int i = threadIdx.x; i < 2 * EDP; i += blockDim.x) slab[i] = aEv[i];
    for (int i = threadIdx.x; i < EMAX + kLossSlots; i += blockDim.x) slab[2 * EDP + i] = ab[i];  // ab then aL
}

// folds the per-workgroup slabs into grads.Ev / W / b and the six loss outputs, adds the
// classifier regulariser (models.py:211-217) when INVPREF_DENSE_REG is set.
// grid: ceil(slab_len/64) blocks of 1024 threads: 64 slab elements x 16 slab-subsets per block.
template <int DUMMY>
__global__ __launch_bounds__(1024) void mstep_finish_kernel(DevTables t, DevGrads g, const float *__restrict__ slabs,
                                                            int nslabs, int DP, int EMAX, StepScalars k,
                                                            float l2, float l1, int64_t Bnorm, uint32_t flags,
                                                            float *__restrict__ losses6) {
    __shared__ double part[16][64];
    __shared__ double sloss[kLossSlots];
    __shared__ double sreg[2];
    const int EDP = t.E * DP, slab_len = 2 * EDP + EMAX + kLossSlots;
    const int col = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + col;
    double acc = 0.0;
    if (idx < slab_len)
        for (int s = sub; s < nslabs; s += 16) acc += (double)slabs[(int64_t)s * slab_len + idx];
    part[sub][col] = acc;
    if (threadIdx.x < 2) sreg[threadIdx.x] = 0.0;
    __syncthreads();
    const bool dense = (flags & INVPREF_DENSE_REG) && !(flags & INVPREF_REG_ONLY_EMBED);
    const bool no_grad = flags & INVPREF_NO_GRAD;
    const bool last_block = blockIdx.x == gridDim.x - 1;  // owns the bias + loss tail (asserted by the launcher)
    if (sub == 0 && idx < slab_len) {
        double v = 0.0;
        for (int s = 0; s < 16; s++) v += part[s][col];
        if (idx < 2 * EDP) {
            const bool isW = idx >= EDP;
            const int r = isW ? idx - EDP : idx;
            const int e = r / DP, d = r - e * DP;
            if (d < t.D && !no_grad) {
                float add = (float)v;
                if (isW && dense) {
                    const float wv = t.W[e * t.D + d];
                    add += 2.f * l2 / ((float)t.D * (float)t.E) * wv + l1 / ((float)t.D * (float)t.E) * c_sign(wv);
                }
                float *dst = (isW ? g.W : g.Ev) + e * t.D + d;
                *dst += add;
            }
        } else if (idx < 2 * EDP + EMAX) {
            const int e = idx - 2 * EDP;
            if (e < t.E && !no_grad) {
                float add = (float)v;
                if (dense) { const float bv = t.b[e]; add += 2.f * l2 / (float)t.E * bv + l1 / (float)t.E * c_sign(bv); }
                g.b[e] += add;
            }
        } else {
            sloss[idx - 2 * EDP - EMAX] = v;
        }
    }
    __syncthreads();
    if (last_block && losses6) {
        // classifier regulariser report (tiny: E*D + E terms), by the first wave
        if (dense && threadIdx.x < 64) {
            double w2 = 0, w1 = 0, b2 = 0, b1 = 0;
            for (int i = threadIdx.x; i < t.E * t.D; i += 64) { const double x = t.W[i]; w2 += x * x; w1 += fabs(x); }
            for (int i = threadIdx.x; i < t.E; i += 64) { const double x = t.b[i]; b2 += x * x; b1 += fabs(x); }
            double r2v = w2 / ((double)t.D * t.E) + b2 / (double)t.E, r1v = w1 / ((double)t.D * t.E) + b1 / (double)t.E;
            for (int m = 32; m >= 1; m >>= 1) { r2v += __shfl_xor(r2v, m, 64); r1v += __shfl_xor(r1v, m, 64); }
            if (threadIdx.x == 0) { sreg[0] = r2v; sreg[1] = r1v; }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double Bn = (double)Bnorm, BD2 = Bn * (double)t.D * 2.0;
            const double Li = sloss[0] / Bn, Le = sloss[1] / Bn, Lc = sloss[2] / Bn;
            const double L2 = sloss[3] / BD2 + sreg[0], L1 = sloss[4] / BD2 + sreg[1];
            losses6[0] += (float)Li; losses6[1] += (float)Le; losses6[2] += (float)Lc;
            losses6[3] += (float)L2; losses6[4] += (float)L1;
            losses6[5] += (float)((double)k.ca * Li + (double)k.cb * Le + (double)k.cc * Lc + (double)l2 * L2 + (double)l1 * L1);
        }
    }
}

// =====================================================================================
// dense Adam over a flat buffer (torch.optim.Adam single-tensor rule; train.py:41,155-157)
// =====================================================================================
__device__ __forceinline__ void adam_scalar_at(float *p, float *g, float *m, float *v, int64_t i, const AdamScalars &a,
                                               int zero_grad) {
    float pp = p[i], mm = m[i], vv = v[i];
    adam1(pp, g[i], mm, vv, a);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (zero_grad) g[i] = 0.f;
}
// `head` (0..3, or n for buffers whose addresses are misaligned differently): leading elements handled one
// float at a time so that the float4 body starts on a 16-byte boundary of all four buffers (a user-sharded
// rank's row range of a table whose factor_num is not a multiple of 4 starts anywhere)
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, int64_t n, AdamScalars a, int zero_grad,
                                                   int64_t head) {
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
    if (head >= 4) {  // no common alignment: scalar everywhere
        for (int64_t i = tid; i < n; i += nthr) adam_scalar_at(p, g, m, v, i, a, zero_grad);
        return;
    }
    const int64_t n4 = (n - head) >> 2;
    float4 *p4 = reinterpret_cast<float4 *>(p + head), *g4 = reinterpret_cast<float4 *>(g + head);
    float4 *m4 = reinterpret_cast<float4 *>(m + head), *v4 = reinterpret_cast<float4 *>(v + head);
    for (int64_t i = tid; i < n4; i += nthr) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        adam1(pp.x, gg.x, mm.x, vv.x, a); adam1(pp.y, gg.y, mm.y, vv.y, a);
        adam1(pp.z, gg.z, mm.z, vv.z, a); adam1(pp.w, gg.w, mm.w, vv.w, a);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        if (zero_grad) g4[i] = f4zero();
    }
    if (blockIdx.x == 0) {
        if ((int64_t)threadIdx.x < head) adam_scalar_at(p, g, m, v, threadIdx.x, a, zero_grad);
        const int64_t t0 = head + (n4 << 2);
        if (threadIdx.x >= 4 && t0 + (threadIdx.x - 4) < n && threadIdx.x < 8)
            adam_scalar_at(p, g, m, v, t0 + (threadIdx.x - 4), a, zero_grad);
    }
}

// the same rule over up to four pieces of the flat buffers in ONE launch (user-sharded runs: a rank updates
// its own rows of the two user tables + the shared tail); offsets and lengths are multiples of 4 floats
struct AdamRanges {
    int64_t off4[4], end4[4];   // piece r covers float4 indices [off4[r], off4[r] + len4[r]); end4 = running total
    int n;
};
// one row of the device-side schedule, as laid out in include/invpref_hip.h (InvPrefAdamSchedule)
struct SchedRowK {
    AdamScalars ad;
    float alpha, pad;
};
// sched_state != nullptr (HIP-graph replay: kernel arguments are frozen): the Adam scalars are read from slot
// `sched_slot` of the device-side schedule, and one thread fills the other slot for the step after this one
// (in the gradient-pass + stand-alone-Adam sequence THIS kernel is the step's last, so it moves the schedule on).
__global__ __launch_bounds__(256) void adam_ranges_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                          float *__restrict__ v, AdamRanges r, AdamScalars a, int zero_grad,
                                                          int *sched_state, const SchedRowK *sched_table, int sched_n,
                                                          int sched_slot) {
    if (sched_state) {
        a = reinterpret_cast<const SchedRowK *>(sched_state + 16 * sched_slot + 2)->ad;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const int *cur = sched_state + 16 * sched_slot;
            int *nxt = sched_state + 16 * (sched_slot ^ 1);
            const int next = cur[0] + 1, base = cur[1], idx = next - base;
            nxt[0] = next;
            nxt[1] = base;
            if (idx >= 0 && idx < sched_n) *reinterpret_cast<SchedRowK *>(nxt + 2) = sched_table[idx];
        }
    }
    const int64_t total = r.end4[r.n - 1];
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
        int q = 0;
        while (j >= r.end4[q]) q++;
        const int64_t i = r.off4[q] + (j - (q ? r.end4[q - 1] : 0));
        float4 pp = reinterpret_cast<float4 *>(p)[i], gg = reinterpret_cast<float4 *>(g)[i];
        float4 mm = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
        adam1(pp.x, gg.x, mm.x, vv.x, a); adam1(pp.y, gg.y, mm.y, vv.y, a);
        adam1(pp.z, gg.z, mm.z, vv.z, a); adam1(pp.w, gg.w, mm.w, vv.w, a);
        reinterpret_cast<float4 *>(p)[i] = pp; reinterpret_cast<float4 *>(m)[i] = mm;
        reinterpret_cast<float4 *>(v)[i] = vv;
        if (zero_grad) reinterpret_cast<float4 *>(g)[i] = f4zero();
    }
}

// =====================================================================================
// E-step  (train.py:169-202, :235-259): argmin_e dist_e, lowest index on ties
// =====================================================================================
// ---- cluster_use_random_sort (train.py:86-92, :192-196): the reference builds the E! x E table of all permutations of
// [1e-10, 1e-11, ...] (itertools.permutations order) and adds row np.random.randint(0, E!) to every interaction's
// distances.  Here only the drawn INDEX travels to the device (1, 4 or 8 bytes per interaction); one lane per
// interaction unranks it -- factorial-base digits, then the d-th element still available, position by position -- into
// sixteen 4-bit element numbers, which the E-step's lanes turn into their tie-break term.
constexpr int kEpsTableMaxE = 7;
struct EpsBase { float v[INVPREF_MAX_ENVS]; };
struct Factorials { unsigned long long f[INVPREF_MAX_ENVS + 1]; };   // f[k] = k!, k = 0 .. 16 (16! = 2.1e13 fits; eps_unrank_kernel clamps to f[E] - 1)
__device__ __forceinline__ unsigned long long unrank_packed(unsigned long long r, int E, const Factorials &fac) {
    unsigned avail = (1u << E) - 1u;
    unsigned long long out = 0;
#pragma unroll
    for (int pos = 0; pos < INVPREF_MAX_ENVS; pos++) {
        if (pos < E) {
            unsigned long long f = 1;   // (E - 1 - pos)!  -- a masked select: no run-time index into the by-value table
#pragma unroll
            for (int k = 0; k < INVPREF_MAX_ENVS; k++) f = (k == E - 1 - pos) ? fac.f[k] : f;
            const unsigned d = (unsigned)(r / f);
            r -= (unsigned long long)d * f;
            // the d-th (0-based) element still available, in increasing order
            unsigned seen = 0, pick = 0;
#pragma unroll
            for (int b = 0; b < INVPREF_MAX_ENVS; b++) {
                const bool on = (avail >> b) & 1u;
                pick = (on && seen == d) ? (unsigned)b : pick;
                seen += on ? 1u : 0u;
            }
            avail &= ~(1u << pick);
            out |= (unsigned long long)pick << (4 * pos);
        }
    }
    return out;
}
// the same row for the LDS table of the E-step's workgroups (E <= 7, r < 5 040): 32-bit arithmetic, the factorial-base digits
// from the low end (c_k = r % (k + 1), r /= k + 1: divisions by compile-time constants) instead of 64-bit divisions by run-time
// factorials -- the table is rebuilt by every workgroup in front of its first pass, and 24 lanes running ~3 000 instructions each
// were 5 us of every workgroup's life (round 5)
__device__ __forceinline__ unsigned unrank_packed_small(unsigned r, int E) {
    unsigned c[8];
    c[0] = 0;
#pragma unroll
    for (int k = 1; k < 8; k++) { c[k] = r % (unsigned)(k + 1); r /= (unsigned)(k + 1); }
    unsigned avail = (1u << E) - 1u, out = 0;
#pragma unroll
    for (int pos = 0; pos < kEpsTableMaxE; pos++) {
        if (pos < E) {
            unsigned d = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) d = (k == E - 1 - pos) ? c[k] : d;
            unsigned seen = 0, pick = 0;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const bool on = (avail >> b) & 1u;
                pick = (on && seen == d) ? (unsigned)b : pick;
                seen += on ? 1u : 0u;
            }
            avail &= ~(1u << pick);
            out |= pick << (4 * pos);
        }
    }
    return out;
}
template <typename IT>
__global__ __launch_bounds__(256) void eps_unrank_kernel(const IT *__restrict__ idx, int64_t N, int E, Factorials fac,
                                                         unsigned long long *__restrict__ packed) {
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < N; s += (int64_t)gridDim.x * blockDim.x)
        // (an index outside [0, E!) -- the reference's np.random.randint cannot draw one -- is clamped, not followed past the table)
        packed[s] = unrank_packed(min((unsigned long long)idx[s], fac.f[E] - 1ull), E, fac);
}
// up to seven environments (7! = 5 040 rows): the E-step's workgroups unrank EVERY row of the permutation table into LDS
// once (4 bytes each) and look an interaction's row up there -- no unranking launch, no packed rows through memory
// NARROW: every big table is under 2^32 bytes (any reference configuration: MIND's are 51 MB), so a row's address is its
// table's base + one 32-bit offset (id x D in a 32-bit multiply) instead of a 64-bit product per gathered row; the ids
// arrive as the reference's int64 (LongTensor) either way.
// The stat_envs half of an E-step (train.py:268-280: counts, class weights; + cluster()'s diff_num, train.py:255) as the
// EPILOGUE of the assignment kernel (invpref_estep_fused_hip): every workgroup publishes its count slab with
// write-through stores, drains them and takes a ticket; the workgroup whose ticket is the last one folds all the slabs.  The
// hand-off is the guide's measured form (MI355X_MICROARCH.md, inter-workgroup visibility, first row of the sc1 table): sc1
// stores by ONE wave of the producer, its s_waitcnt vmcnt(0), an agent-scope atomic add by a lane of that wave, the adder
// whose add came last reads everything with sc1 loads behind a workgroup barrier.
// The ticket is SHARDED: 2 048 workgroups that finish together would queue on one word for ~25 us (one address takes ~88
// returning atomics per microsecond); workgroup b takes a ticket of shard b % kTicketShards (each on a 128-byte line of its own),
// the workgroup that completes a shard takes a ticket of the top word, the one that completes that folds.
#ifndef ESTEP_IDX_BULK
#define ESTEP_IDX_BULK 1
#endif
#ifndef ESTEP_SHARD_COUNTERS
#define ESTEP_SHARD_COUNTERS 1   // the fused epilogue's counts: 1 = integer atomics into the shard's counters (round 6, late: the fused kernel
                                 // 51.5 -> 42.2 us plain, 62.7 -> 51.7 with the tie-break, same box); 0 = a slab per workgroup, folded by the last one
#endif
constexpr int kTicketShards = 32;   // (INVPREF_ESTEP_STATE_INTS = 32 + 32 * kTicketShards)
struct EstepFin {
    int *state;          // device int32[kEstepStateInts], ZERO before the first call and left zero: [0] top ticket, [1] ring
                         // position, [32 + 32 s] ticket of shard s; NULL: no epilogue
    int64_t *ring;       // [ring_cap][E + 1] {counts[0..E), diff} of E-step number `ring position` (mod ring_cap); may be NULL
    int ring_cap;
    int64_t *counts;     // [E] (may be NULL)
    int64_t *diff;       // [1] (may be NULL)
    float *class_w;      // [E] min(count + 1, N - 1) / N   (train.py:274-277; may be NULL)
    const unsigned *perm_table;   // E <= 7: the E! packed permutation rows, built once (invpref_perm_table_fill); NULL: unranked here
};
__device__ __forceinline__ int ld_sc1_i(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1_i(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#ifndef ESTEP_NUM_SGPR
#define ESTEP_NUM_SGPR 80   // 256-thread workgroups per CU are also capped by the scalar registers: <= 80 -> 8, 82-96 -> 7, 98+ -> 6
                            // (MI355X guide, residency); left alone the compiler takes 100 and a quarter of the 2 048 workgroups
                            // of a Yahoo-sized E-step waits for a second residency
#endif
template <int NC, bool VEC, bool NARROW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(ESTEP_NUM_SGPR))) void estep_assign_kernel(DevTables t, const int64_t *__restrict__ users,
                                                           const int64_t *__restrict__ items,
                                                           const float *__restrict__ scores, int64_t N, uint32_t flags,
                                                           const float *__restrict__ eps_rows,
                                                           const unsigned long long *__restrict__ eps_packed, EpsBase eps_base,
                                                           const void *__restrict__ eps_index, int eps_index_bytes,
                                                           int eps_rows_n, Factorials fac,
                                                           const int64_t *old_envs, int64_t *new_envs,
                                                           int *slabs, EstepFin fin) {
    // (old_envs and new_envs may be the SAME buffer -- cluster() updates the assignments in place -- so neither
    //  is __restrict__; element s is read into a register before it is written)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int DP = NC * 64;
    float *sEv = lds;
    int *cnt = reinterpret_cast<int *>(lds + t.E * DP);  // [E + 1]
    float *sbase = lds + t.E * DP + INVPREF_MAX_ENVS + 1;  // [16] the tie-break vector (train.py:86-92), eps_packed form
    stage_table(sEv, t.Ev, t.E, t.D, DP);
    for (int i = threadIdx.x; i <= t.E; i += blockDim.x) cnt[i] = 0;
    unsigned *stab = reinterpret_cast<unsigned *>(sbase + INVPREF_MAX_ENVS);   // [E!] packed permutation rows (E <= 7)
    if (eps_packed || eps_index) {
#pragma unroll
        for (int k = 0; k < INVPREF_MAX_ENVS; k++)   // (static indices: a by-value argument indexed at run time goes to scratch)
            if (threadIdx.x == k) sbase[k] = eps_base.v[k];
    }
    // A workgroup walks ONE contiguous block of interactions, 16 per pass (any split gives the same per-interaction results).
    const int64_t rows_per_block = blockDim.x / kRow;
    const int64_t chunk = ((N + gridDim.x - 1) / gridDim.x + rows_per_block - 1) / rows_per_block * rows_per_block;
    const int64_t s_begin = blockIdx.x * chunk, s_end = min(N, s_begin + chunk);
    // (E <= 7) the permutation INDEX of an interaction is fetched by its own lane group, one pass ahead of its use: a byte (or
    // word) per interaction, adjacent for the wave's four groups -- from device memory or, as the managers hand it over, straight
    // from PINNED HOST memory: the link's latency sits under a pass of arithmetic instead of in front of the workgroup's first
    // (round 5: the staged form -- the workgroup's whole index range fetched and spread into LDS before the first pass -- cost
    // 7 us of a 43 us kernel with device-resident indices and 12 us from pinned memory)
    auto load_idx = [&](int64_t sx) -> unsigned {
        return eps_index_bytes == 1 ? (unsigned)reinterpret_cast<const uint8_t *>(eps_index)[sx]
                                    : (unsigned)reinterpret_cast<const int32_t *>(eps_index)[sx];
    };
    if (eps_index) {
        // (the table is the same for every workgroup of every launch: handed over ready-made by the fused entry point -- one
        //  coalesced load of 96 bytes at four environments instead of ~1 000 instructions on wave 0 while the others wait)
        if (fin.perm_table)
            for (int i = threadIdx.x; i < eps_rows_n; i += blockDim.x) stab[i] = fin.perm_table[i];
        else
            for (int i = threadIdx.x; i < eps_rows_n; i += blockDim.x) stab[i] = unrank_packed_small((unsigned)i, t.E);
    }
    __syncthreads();
    const int l16 = threadIdx.x & 15;
    const bool implicit = flags & INVPREF_IMPLICIT;
    // The tie-break terms are tiny (the reference's 1e-10, 1e-11, ...): added to a distance d >= 2^26 max|eps| (6.7e-3) they
    // are below a quarter of d's last place and the sum rounds back to d, bit for bit.  An interaction whose SMALLEST
    // distance is that large therefore needs no permutation row at all -- its three dependent LDS look-ups are skipped
    // (most interactions; the others take the look-ups as before: same sums, same argmin).
    float eps_thr = 0.f;
    if (eps_packed || eps_index) {
        float mx = 0.f;
#pragma unroll
        for (int k = 0; k < INVPREF_MAX_ENVS; k++) mx = (k < t.E) ? __builtin_fmaxf(mx, __builtin_fabsf(eps_base.v[k])) : mx;
        eps_thr = mx * 67108864.f * 1.001f;
    }
    const int64_t s_first = s_begin + (threadIdx.x >> 4);
    // One-byte indices (E <= 5, the managers' form), ESTEP_IDX_BULK: a WAVE fetches the bytes of its four groups for sixteen passes
    // with ONE load -- lane 4 k + g holds the byte of group g in pass k: four 64-byte lines instead of sixteen requests of four
    // useful bytes each -- and a group picks its pass's byte out of the wave's registers (v_readlane: indifferent to lanes that
    // have left the loop).  From pinned host memory the kernel is bound by the NUMBER of link reads, not by their latency.
    const bool idx_bulk = ESTEP_IDX_BULK && eps_index && eps_index_bytes == 1;
    const int lane64 = threadIdx.x & 63, grp_w = lane64 >> 4;
    auto load_bulk = [&](int blk) -> unsigned {
        const int64_t sx = s_begin + ((threadIdx.x >> 6) * 4 + (lane64 & 3)) + rows_per_block * ((int64_t)blk * 16 + (lane64 >> 2));
        return sx < s_end ? (unsigned)reinterpret_cast<const uint8_t *>(eps_index)[sx] : 0u;
    };
    unsigned bulk_cur = idx_bulk ? load_bulk(0) : 0u, bulk_next = 0u;
    unsigned idx_cur = (eps_index && !idx_bulk && s_first < s_end) ? load_idx(s_first) : 0u;
    int pass = 0;
    for (int64_t s = s_first; s < s_end; s += rows_per_block, pass++) {
        unsigned idx_next = 0u;
        if (idx_bulk) {
            const int pk = __builtin_amdgcn_readfirstlane(pass);
            if ((pk & 15) == 0) {
                if (pk) bulk_cur = bulk_next;
                bulk_next = load_bulk((pk >> 4) + 1);
            }
            const int src = (pk & 15) * 4;
            const unsigned i0 = __builtin_amdgcn_readlane(bulk_cur, src), i1 = __builtin_amdgcn_readlane(bulk_cur, src + 1);
            const unsigned i2 = __builtin_amdgcn_readlane(bulk_cur, src + 2), i3 = __builtin_amdgcn_readlane(bulk_cur, src + 3);
            idx_cur = grp_w == 0 ? i0 : (grp_w == 1 ? i1 : (grp_w == 2 ? i2 : i3));
        } else if (eps_index && s + rows_per_block < s_end) {
            idx_next = load_idx(s + rows_per_block);
        }
        const int64_t u = users[s], v = items[s];
        const float y = scores[s];
        float4 pu[NC], qi[NC], pa[NC], qa[NC];
        if (NARROW) {
            const unsigned uo = (unsigned)u * (unsigned)t.D, vo = (unsigned)v * (unsigned)t.D;
            load_row<NC, VEC>(t.Pu + uo, 0, t.D, l16, pu);
            load_row<NC, VEC>(t.Qi + vo, 0, t.D, l16, qi);
            load_row<NC, VEC>(t.Pa + uo, 0, t.D, l16, pa);
            load_row<NC, VEC>(t.Qa + vo, 0, t.D, l16, qa);
        } else {
            load_row<NC, VEC>(t.Pu, u, t.D, l16, pu);
            load_row<NC, VEC>(t.Qi, v, t.D, l16, qi);
            load_row<NC, VEC>(t.Pa, u, t.D, l16, pa);
            load_row<NC, VEC>(t.Qa, v, t.D, l16, qa);
        }
        const float p = dot2<NC>(pu, qi);
        const float sp = implicit ? c_sigmoid(p) : p;
        // One environment per LANE: q_e is a group-uniform value after the row reduction and lane e keeps it; the
        // sigmoid / BCE chain (canonical, ~100 VALU instructions -- the kernel is ALU-bound, VALUBusy 97 %) then
        // runs ONCE for all environments instead of once per environment, each lane on its own q.  Same operations on
        // the same operands as the sequential form, so the distances are bit-identical.
        float qmine = 0.f;
        for (int c = 0; c < t.E; c++) {
            float4 ev[NC];
            lds_row<NC>(sEv, c, l16, ev);
            const float q = dot3<NC>(pa, qa, ev);
            qmine = (l16 == c) ? q : qmine;
        }
        float dist;
        if (implicit) {
            // labels of the implicit data are 0 or 1: ONE canonical logarithm per distance instead of two (same float, see
            // canon_math.hpp: c_bce_binary); a wave that meets any other label takes the definition
            const float sv = sp * c_sigmoid(qmine);
            const bool binary = __builtin_amdgcn_ballot_w64(!(y == 0.0f || y == 1.0f)) == 0;
            dist = binary ? c_bce_binary(sv, y) : c_bce(sv, y);
        } else { const float r = (p + qmine) - y; dist = r * r; }
        if (eps_rows && l16 < t.E) dist = dist + eps_rows[s * t.E + l16];
        // train.py:192-196 with the row's permutation unranked on the device: position l16 of permutation row idx[s] of
        // the tie-break vector is element (packed >> 4 l16) & 15 of it (eps_unrank_kernel)
        if (eps_packed || eps_index) {
            // (NaN distances are ignored by the minimum and stay NaN under the add: nothing to look up for them either)
            const bool need = row16_min(l16 < t.E ? dist : __builtin_inff()) < eps_thr;
            if (need) {
                if (eps_packed && l16 < t.E) dist = dist + sbase[(eps_packed[s] >> (4 * l16)) & 15ull];
                if (eps_index && l16 < t.E)   // (E <= 7: the row looked up in the workgroup's LDS table)
                    dist = dist + sbase[(stab[min(idx_cur, (unsigned)(eps_rows_n - 1))] >> (4 * l16)) & 15u];   // (clamped: see eps_unrank_kernel)
            }
        }
        // argmin with the lowest index among equal minima (torch.argmin; the sequential `dist < best` scan)
        // A NaN distance wins, the first one if there are several (torch.argmin's LessOrNan; only reachable with
        // NaN parameters) -- row16_min ignores NaNs, so the lowest NaN lane is found separately.
        dist = l16 < t.E ? dist : __builtin_inff();
        const float dmin = row16_min(dist);
        const int bi_num = (int)row16_min(dist == dmin ? (float)l16 : 99.f);
        const int bi_nan = (int)row16_min(dist != dist ? (float)l16 : 99.f);
        const int bi = bi_nan < 99 ? bi_nan : bi_num;   // (bi_num == 99 only when every distance is NaN)
        if (l16 == 0) {
            const int64_t was = old_envs ? old_envs[s] : (int64_t)bi;
            const bool changed = was != (int64_t)bi;
            new_envs[s] = bi;
            atomicAdd(cnt + bi, 1);
            if (changed) atomicAdd(cnt + t.E, 1);
        }
        if (!idx_bulk) idx_cur = idx_next;
    }
    __syncthreads();
    if (!fin.state) {
        for (int i = threadIdx.x; i <= t.E; i += blockDim.x) slabs[(int64_t)blockIdx.x * (t.E + 1) + i] = cnt[i];
        return;
    }
    // ---- fused epilogue (E + 1 <= 17: the slab is stored by lanes of wave 0 alone)
    __shared__ int s_last;
    __shared__ long long s_tot[INVPREF_MAX_ENVS + 1];
    if (threadIdx.x < 64) {
#if ESTEP_SHARD_COUNTERS
        // (the workgroup's counts are ADDED into its shard's counters, words 1 .. E + 1 of the shard's 128-byte line, in front of its ticket)
        if ((int)threadIdx.x <= t.E) {
            const int S0 = min(kTicketShards, (int)gridDim.x);
            __hip_atomic_fetch_add(fin.state + 32 + 32 * ((int)blockIdx.x % S0) + 1 + (int)threadIdx.x, cnt[threadIdx.x], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
#else
        if ((int)threadIdx.x <= t.E) st_sc1_i(slabs + (int64_t)blockIdx.x * (t.E + 1) + threadIdx.x, cnt[threadIdx.x]);
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
            const int S = min(kTicketShards, (int)gridDim.x), sh = (int)blockIdx.x % S;
            const int members = ((int)gridDim.x - sh + S - 1) / S;          // workgroups b with b % S == sh
            int *tk = fin.state + 32 + 32 * sh;
            int last = 0;
            if (__hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
                st_sc1_i(tk, 0);                                            // (this launch is through with the shard)
                last = __hip_atomic_fetch_add(fin.state, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S - 1;
            }
            s_last = last;
        }
    }
    for (int i = threadIdx.x; i <= t.E; i += blockDim.x) s_tot[i] = 0;
    __syncthreads();
    if (!s_last) return;
    {
        const int E1 = t.E + 1, nsl = (int)gridDim.x;
        // (write-through stores -> cache-bypassing loads.  The slabs are ONE contiguous int array [workgroups][E + 1]: thread t of
        //  the first `per` = 256 - 256 % (E + 1) threads reads elements t, t + per, ... -- coalesced, and all of ONE class, t %
        //  (E + 1) -- twenty loads in flight, one accumulator; then one LDS atomic per thread.  The fold runs in one workgroup per
        //  launch but its registers are every workgroup's: with 17 per-thread 64-bit accumulators the Yahoo instance went from
        //  57 to 74 registers, 8 -> 6 waves per SIMD; every thread adding every count through LDS atomics on the same five words
        //  took 138 us; one class at a time (a round trip each) 53 us)
        const int per = (int)blockDim.x - (int)blockDim.x % E1, total = nsl * E1;
#if ESTEP_SHARD_COUNTERS
        // (every workgroup has ADDED its counts to its shard's counters -- integer atomics: order-free, exact -- in front of its
        //  ticket: shards x (E + 1) words, ONE round trip of cache-bypassing loads, instead of every workgroup's slab in passes of
        //  twenty loads.  The words go back to zero for the next launch)
        (void)per; (void)total;
        const int S0 = min(kTicketShards, nsl);
        for (int i = threadIdx.x; i < S0 * E1; i += blockDim.x) {
            int *w = fin.state + 32 + 32 * (i / E1) + 1 + i % E1;
            const int x = ld_sc1_i(w);
            st_sc1_i(w, 0);
            atomicAdd((unsigned long long *)&s_tot[i % E1], (unsigned long long)(long long)x);
        }
#else
        if ((int)threadIdx.x < per) {
            long long a = 0;
            for (int i0 = threadIdx.x; i0 < total; i0 += 20 * per) {
                int x[20];
#pragma unroll
                for (int k = 0; k < 20; k++) {
                    const int i = i0 + k * per;
                    x[k] = i < total ? ld_sc1_i(slabs + i) : 0;
                }
#pragma unroll
                for (int k = 0; k < 20; k++) a += x[k];
            }
            atomicAdd((unsigned long long *)&s_tot[(int)threadIdx.x % E1], (unsigned long long)a);
        }
#endif
        __syncthreads();
        int64_t *row = nullptr;
        if (fin.ring) row = fin.ring + (int64_t)((unsigned)fin.state[1] % (unsigned)fin.ring_cap) * E1;
        if ((int)threadIdx.x < t.E) {
            const long long c = s_tot[threadIdx.x];
            const long long r = (c + 1 < N - 1) ? c + 1 : N - 1;
            if (fin.counts) fin.counts[threadIdx.x] = c;
            if (fin.class_w) fin.class_w[threadIdx.x] = (float)((double)r / (double)N);
            if (row) row[threadIdx.x] = c;
        }
        if (threadIdx.x == 0) {
            if (fin.diff) *fin.diff = s_tot[t.E];
            if (row) row[t.E] = s_tot[t.E];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (fin.ring) fin.state[1] = fin.state[1] + 1;
            st_sc1_i(fin.state, 0);   // the ticket is back at zero for the next launch (stream order: it starts after this one ends)
        }
    }
}

// per-workgroup env histogram (for stat_envs alone)
__global__ __launch_bounds__(256) void env_hist_kernel(const int64_t *__restrict__ envs, int64_t N, int E,
                                                       int *__restrict__ slabs) {
    __shared__ int cnt[INVPREF_MAX_ENVS + 1];
    for (int i = threadIdx.x; i <= E; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < N; s += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(cnt + (int)envs[s], 1);
    __syncthreads();
    for (int i = threadIdx.x; i <= E; i += blockDim.x) slabs[(int64_t)blockIdx.x * (E + 1) + i] = cnt[i];
}

// stat_envs (train.py:268-280): every workgroup folds the count slabs (tiny), forms
// class_w[e] = min(cnt+1, N-1)/N and then gathers sample_weights for its share of the rows.
__global__ __launch_bounds__(256) void stat_envs_kernel(const int64_t *__restrict__ envs, int64_t N, int E,
                                                        const int *__restrict__ slabs, int nslabs,
                                                        int64_t *__restrict__ counts, int64_t *__restrict__ diff,
                                                        float *__restrict__ class_w, float *__restrict__ sample_w) {
    __shared__ long long tot[INVPREF_MAX_ENVS + 1];
    __shared__ float cw[INVPREF_MAX_ENVS];
    for (int i = threadIdx.x; i <= E; i += blockDim.x) tot[i] = 0;
    __syncthreads();
    // (every count of a thread's slabs is requested before the first is summed: one round trip for the fold, not one per
    //  class -- the per-class loop took 10.9 us of a 250 154-row call)
    long long acc[INVPREF_MAX_ENVS + 1];
#pragma unroll
    for (int c = 0; c <= INVPREF_MAX_ENVS; c++) acc[c] = 0;
    for (int s = threadIdx.x; s < nslabs; s += blockDim.x) {
        const int *row = slabs + (int64_t)s * (E + 1);
#pragma unroll
        for (int c = 0; c <= INVPREF_MAX_ENVS; c++) acc[c] += (c <= E) ? row[min(c, E)] : 0;
    }
#pragma unroll
    for (int c = 0; c <= INVPREF_MAX_ENVS; c++) {
        if (c <= E) {   // (uniform)
            long long a = acc[c];
            for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
            if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)&tot[c], (unsigned long long)a);
        }
    }
    __syncthreads();
    if (threadIdx.x < E) {
        const long long c = tot[threadIdx.x];
        const long long r = (c + 1 < N - 1) ? c + 1 : N - 1;
        const float w = (float)((double)r / (double)N);
        cw[threadIdx.x] = w;
        if (blockIdx.x == 0) {
            counts[threadIdx.x] = c;
            if (class_w) class_w[threadIdx.x] = w;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && diff) *diff = tot[E];
    __syncthreads();
    if (sample_w) {
        // four consecutive rows per thread and pass, every load of the pass requested before the first look-up (a row per
        // thread and pass left one dependent load -> store chain per iteration on four waves per CU)
        const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
        for (int64_t s0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; s0 < N; s0 += stride) {
            int e[4];
#pragma unroll
            for (int k = 0; k < 4; k++) e[k] = (int)envs[min(s0 + k, N - 1)];
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (s0 + k < N) sample_w[s0 + k] = cw[e[k]];
        }
    }
}

// The same fold as estep_assign_kernel's epilogue, as ONE workgroup of its own behind the assignment kernel (INVPREF_ESTEP_FOLD=kernel:
// a kernel boundary + two round trips of one workgroup, against the epilogue's chain of store drain -> shard ticket -> top
// ticket -> fold on the assignment kernel's own tail; measured per E-step in profiles/r06).
__global__ __launch_bounds__(256) void estep_fold_kernel(const int *__restrict__ slabs, int nslabs, int E, int64_t N, EstepFin fin) {
    __shared__ long long s_tot[INVPREF_MAX_ENVS + 1];
    const int E1 = E + 1, total = nslabs * E1;
    for (int i = threadIdx.x; i <= E; i += blockDim.x) s_tot[i] = 0;
    __syncthreads();
    const int per = (int)blockDim.x - (int)blockDim.x % E1;
    if ((int)threadIdx.x < per) {
        long long a = 0;
        for (int i0 = threadIdx.x; i0 < total; i0 += 20 * per) {
            int x[20];
#pragma unroll
            for (int k = 0; k < 20; k++) {
                const int i = i0 + k * per;
                x[k] = i < total ? slabs[i] : 0;
            }
#pragma unroll
            for (int k = 0; k < 20; k++) a += x[k];
        }
        atomicAdd((unsigned long long *)&s_tot[(int)threadIdx.x % E1], (unsigned long long)a);
    }
    __syncthreads();
    int64_t *row = nullptr;
    if (fin.ring) row = fin.ring + (int64_t)((unsigned)fin.state[1] % (unsigned)fin.ring_cap) * E1;
    if ((int)threadIdx.x < E) {
        const long long c = s_tot[threadIdx.x];
        const long long r = (c + 1 < N - 1) ? c + 1 : N - 1;
        if (fin.counts) fin.counts[threadIdx.x] = c;
        if (fin.class_w) fin.class_w[threadIdx.x] = (float)((double)r / (double)N);
        if (row) row[threadIdx.x] = c;
    }
    if (threadIdx.x == 0) {
        if (fin.diff) *fin.diff = s_tot[E];
        if (row) row[E] = s_tot[E];
    }
    __syncthreads();
    if (threadIdx.x == 0 && fin.ring) fin.state[1] = fin.state[1] + 1;
}

// =====================================================================================
// predict (models.py:393-407): scores[n, I] = f(Pu[users[n]] . Qi[i]) for ALL items, f = sigmoid
// (implicit) or identity.  One 16-lane row keeps its user row in registers and sweeps the item
// table (L2/Infinity-Cache resident); 16 results are collected across the row and stored as one
// 64-byte segment.
// =====================================================================================
template <int NC, bool VEC>
__global__ __launch_bounds__(256) void predict_kernel(const float *__restrict__ Pu, const float *__restrict__ Qi,
                                                      const int64_t *__restrict__ users, int64_t n, int I, int D,
                                                      int apply_sigmoid, float *__restrict__ out, int per_slice) {
    const int l16 = threadIdx.x & 15;
    const int64_t row = blockIdx.x * (int64_t)(blockDim.x / kRow) + (threadIdx.x >> 4);
    if (row >= n) return;
    float4 pu[NC];
    load_row<NC, VEC>(Pu, users[row], D, l16, pu);
    float *o = out + row * (int64_t)I;
    const int i_lo = blockIdx.y * per_slice;       // (a multiple of 16: the 64-byte result segments stay aligned)
    int i_hi = i_lo + per_slice;
    i_hi = i_hi < I ? i_hi : I;
    for (int i0 = i_lo; i0 < i_hi; i0 += 16) {
        float res = 0.f;
#pragma unroll 4
        for (int j = 0; j < 16; j++) {
            const int i = i0 + j;
            if (i < i_hi) {
                float4 qi[NC];
                load_row<NC, VEC>(Qi, i, D, l16, qi);
                float p = dot2<NC>(pu, qi);
                if (apply_sigmoid) p = c_sigmoid(p);
                res = (j == l16) ? p : res;
            }
        }
        if (i0 + l16 < i_hi) o[i0 + l16] = res;
    }
}

// The same score matrix on the MATRIX CORES ("the one genuinely dense contraction"; models.py:393-407,
// evaluate.py:88-92) for rows of 64 / 128 / 256 floats -- bit for bit the values of predict_kernel, forward() and the oracle:
// the canonical dot product gives element i to slot (i >> 2) & 15, runs ONE fma chain per slot over its elements in
// increasing i and sums the 16 slots pairwise (xor 1, 2, 4, 8).  v_mfma_f32_16x16x4_f32 accumulates its four k terms as an fmaf
// chain in k order, so ONE MFMA per (64-float chunk c, slot s) with k = the four consecutive elements 64 c + 4 s + k, chained
// over c in the slot's own accumulator tile, IS the slot's chain -- for 16 users x 16 items at a time -- and the 16 slot tiles are
// then added pairwise in the butterfly's order.  A workgroup = 64 users (one 16-user tile per wave, its rows held as MFMA A
// operands for the whole sweep: 16 DC registers) x a range of items; a 16-item tile is staged in LDS once for the four waves
// (row stride 64 DC + 4 floats: the B operand's lanes -- item n, k -- fall on 64 different banks), double-buffered, one barrier
// per tile.  Lane l of a wave then holds C[4 (l >> 4) + r][l & 15]: users 4 (l >> 4) + r, item l & 15 -- 64-byte row segments.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int DC>
__global__ __launch_bounds__(256, 2) void predict_mm_kernel(const float *__restrict__ Pu, const float *__restrict__ Qi,
                                                            const int64_t *__restrict__ users, int64_t n, int I,
                                                            int apply_sigmoid, float *__restrict__ out, int steps_per) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int D = 64 * DC, RS = D + 4, TILE = 16 * RS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, k = lane >> 4;
    // ---- A operands: user tile of this wave, a[c][s] = Pu[user m][64 c + 4 s + k]
    const int64_t urow = (int64_t)blockIdx.x * 64 + wave * 16 + m;
    const int64_t uid = users[urow < n ? urow : n - 1];
    const float *pu = Pu + uid * (int64_t)D + k;
    float a[DC][16];
#pragma unroll
    for (int c = 0; c < DC; c++)
#pragma unroll
        for (int s = 0; s < 16; s++) a[c][s] = pu[64 * c + 4 * s];
    // ---- item tiles [t0, t1) of 16 items each
    const int tiles = (I + 15) / 16;
    const int t0 = (int)blockIdx.y * steps_per, t1 = min(tiles, t0 + steps_per);
    if (t0 >= t1) return;
    // staging: 16 rows x D floats = 4 D float4; thread th moves float4 number th + 256 j: row (th + 256 j) / (D / 4)
    constexpr int F4 = D / 4, PER = 16 * F4 / 256;   // float4 per thread and tile: DC
    static_assert(PER == DC, "staging: DC float4 per thread");
    // (thread th moves float4 number th + 256 j of a tile: row r_j, float4 q_j of the row -- fixed per thread)
    int src_off[PER], dst_off[PER], rr[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const int f = threadIdx.x + 256 * j;
        rr[j] = f / F4;
        src_off[j] = 4 * (f - rr[j] * F4);
        dst_off[j] = rr[j] * RS + src_off[j];
    }
    float4 st[PER];
    // every load and LDS store of the loop is unconditional (a load under a branch is waited for at the join, and a register
    // array written under a branch goes to scratch memory): the tile after the last is the last one again
#pragma unroll
    for (int j = 0; j < PER; j++)
        st[j] = *reinterpret_cast<const float4 *>(Qi + (int64_t)min(t0 * 16 + rr[j], I - 1) * D + src_off[j]);
#pragma unroll
    for (int j = 0; j < PER; j++) *reinterpret_cast<float4 *>(lds + dst_off[j]) = st[j];
    __syncthreads();
    for (int t = t0; t < t1; t++) {
        const int buf = (t - t0) & 1;
        const int tn = min(t + 1, t1 - 1);
#pragma unroll
        for (int j = 0; j < PER; j++)
            st[j] = *reinterpret_cast<const float4 *>(Qi + (int64_t)min(tn * 16 + rr[j], I - 1) * D + src_off[j]);
        const float *bt = lds + buf * TILE + m * RS + k;      // B[k][n = m]: item m of the tile, element 64 c + 4 s + k
        f32x4_t acc[16];
#pragma unroll
        for (int s = 0; s < 16; s++) acc[s] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < DC; c++)
#pragma unroll
            for (int s = 0; s < 16; s++)
                acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][s], bt[64 * c + 4 * s], acc[s], 0, 0, 0);
        // the 16 slots, pairwise in the butterfly's order (xor 1, 2, 4, 8)
#pragma unroll
        for (int s = 0; s < 16; s += 2) acc[s] = acc[s] + acc[s + 1];
#pragma unroll
        for (int s = 0; s < 16; s += 4) acc[s] = acc[s] + acc[s + 2];
#pragma unroll
        for (int s = 0; s < 16; s += 8) acc[s] = acc[s] + acc[s + 4];
        acc[0] = acc[0] + acc[8];
        const int item = t * 16 + m;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float p = acc[0][r];
            if (apply_sigmoid) p = c_sigmoid(p);
            const int64_t row = (int64_t)blockIdx.x * 64 + wave * 16 + 4 * k + r;
            if (row < n && item < I) out[row * (int64_t)I + item] = p;
        }
#pragma unroll
        for (int j = 0; j < PER; j++) *reinterpret_cast<float4 *>(lds + (buf ^ 1) * TILE + dst_off[j]) = st[j];
        __syncthreads();
    }
}

// class/sample weights from already-global counts (multi-GPU: counts were all-reduced)
__global__ __launch_bounds__(256) void sample_weights_kernel(const int64_t *__restrict__ envs, int64_t N_local,
                                                             const int64_t *__restrict__ counts, int64_t N_total, int E,
                                                             float *__restrict__ class_w, float *__restrict__ sample_w) {
    __shared__ float cw[INVPREF_MAX_ENVS];
    if (threadIdx.x < E) {
        const long long c = counts[threadIdx.x];
        const long long r = (c + 1 < N_total - 1) ? c + 1 : N_total - 1;
        const float w = (float)((double)r / (double)N_total);
        cw[threadIdx.x] = w;
        if (blockIdx.x == 0 && class_w) class_w[threadIdx.x] = w;
    }
    __syncthreads();
    if (sample_w)
        for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < N_local; s += (int64_t)gridDim.x * blockDim.x)
            sample_w[s] = cw[(int)envs[s]];
}

// ------------------------------------------------------------------ host helpers
constexpr int kMstepThreads = 512, kMstepMaxBlocks = 512;
constexpr int kEstepThreads = 256, kEstepMaxBlocks = 2048;
inline int mstep_blocks(int64_t B) {
    int64_t nb = (B + (kMstepThreads / kRow) - 1) / (kMstepThreads / kRow);
    return (int)(nb < 1 ? 1 : (nb > kMstepMaxBlocks ? kMstepMaxBlocks : nb));
}
inline int estep_blocks(int64_t N) {
    int64_t nb = (N + (kEstepThreads / kRow) - 1) / (kEstepThreads / kRow);
    static const int cap_env = std::getenv("INVPREF_ESTEP_BLOCKS") ? atoi(std::getenv("INVPREF_ESTEP_BLOCKS")) : 0;   // (A/B knob)
    const int cap = cap_env > 0 && cap_env < kEstepMaxBlocks ? cap_env : kEstepMaxBlocks;
    return (int)(nb < 1 ? 1 : (nb > cap ? cap : nb));
}

// dispatch over (NC, VEC, EMAX): VEC=false only exists at NC=4 (any D <= 256)
#define DISPATCH_NVE(NCV, VECV, EMAXV, CALL)                                             \
    do {                                                                                  \
        if (!(VECV)) {                                                                    \
            if ((EMAXV) == 4) { CALL(4, false, 4); } else if ((EMAXV) == 8) { CALL(4, false, 8); } else { CALL(4, false, 16); } \
        } else if ((NCV) == 1) {                                                          \
            if ((EMAXV) == 4) { CALL(1, true, 4); } else if ((EMAXV) == 8) { CALL(1, true, 8); } else { CALL(1, true, 16); } \
        } else if ((NCV) == 2) {                                                          \
            if ((EMAXV) == 4) { CALL(2, true, 4); } else if ((EMAXV) == 8) { CALL(2, true, 8); } else { CALL(2, true, 16); } \
        } else {                                                                          \
            if ((EMAXV) == 4) { CALL(4, true, 4); } else if ((EMAXV) == 8) { CALL(4, true, 8); } else { CALL(4, true, 16); } \
        }                                                                                 \
    } while (0)

}  // namespace

// ===================================================================================== C ABI
// ---- packed exchange of a row-sharded step: the rows of the flat gradient the GLOBAL minibatch touches --
// every other row of the four big tables is zero on every rank -- and the flat buffer's tail (the small tables) are copied
// into one contiguous buffer for the all-reduce (UNPACK = false) and back (UNPACK = true).  HBM-bound copies: one float4
// per lane, consecutive lanes on consecutive float4 of a row.
template <bool UNPACK, bool VEC>
__global__ __launch_bounds__(256) void pack_rows_kernel(float *__restrict__ flat, const int64_t *__restrict__ row_offsets,
                                                        int64_t n_rows, int32_t D, int64_t tail_offset, int64_t tail_len,
                                                        float *__restrict__ packed) {
    constexpr int W = VEC ? 4 : 1;
    const int64_t per_row = D / W, body = n_rows * per_row, total = body + (tail_len + W - 1) / W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float *f, *q;
        int cnt = W;
        if (i < body) {
            const int64_t r = i / per_row, c = i - r * per_row;
            f = flat + row_offsets[r] + c * W;
            q = packed + r * (int64_t)D + c * W;
        } else {
            const int64_t c = (i - body) * W;
            f = flat + tail_offset + c;
            q = packed + n_rows * (int64_t)D + c;
            if (tail_len - c < W) cnt = (int)(tail_len - c);
        }
        if (VEC && cnt == 4) {
            if (UNPACK) *reinterpret_cast<float4 *>(f) = *reinterpret_cast<const float4 *>(q);
            else *reinterpret_cast<float4 *>(q) = *reinterpret_cast<const float4 *>(f);
        } else {
            for (int k = 0; k < cnt; k++) {
                if (UNPACK) f[k] = q[k];
                else q[k] = f[k];
            }
        }
    }
}

static int pack_rows_launch(bool unpack, float *flat, const int64_t *row_offsets, int64_t n_rows, int32_t D,
                            int64_t tail_offset, int64_t tail_len, float *packed, int vec_ok, void *stream) {
    if (!flat || !packed || n_rows < 0 || D < 1 || tail_len < 0 || tail_offset < 0 || (n_rows > 0 && !row_offsets))
        return INVPREF_EINVAL;
    if (n_rows == 0 && tail_len == 0) return 0;
    // float4 form: rows of whole float4 starting on 16-byte boundaries (the caller vouches for the row offsets: vec_ok)
    const uintptr_t af = reinterpret_cast<uintptr_t>(flat), aq = reinterpret_cast<uintptr_t>(packed);
    const bool vec = vec_ok && D % 4 == 0 && tail_offset % 4 == 0 && ((af | aq) & 15u) == 0;
    const int64_t items = n_rows * (D / (vec ? 4 : 1)) + (tail_len + (vec ? 3 : 0)) / (vec ? 4 : 1);
    int64_t nb = (items + 255) / 256;
    nb = nb < 1 ? 1 : (nb > 4096 ? 4096 : nb);
#define PK(U, V)                                                                                                         \
    hipLaunchKernelGGL((pack_rows_kernel<U, V>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, flat, row_offsets, \
                       n_rows, D, tail_offset, tail_len, packed)
    if (unpack) { if (vec) PK(true, true); else PK(true, false); }
    else { if (vec) PK(false, true); else PK(false, false); }
#undef PK
    return (int)hipGetLastError();
}

extern "C" {

int invpref_abi_version(void) { return INVPREF_ABI_VERSION; }

int invpref_device_name(char *buf, size_t len) {
    if (!buf || len == 0) return INVPREF_EINVAL;
    int dev = 0;
    hipError_t err = hipGetDevice(&dev);
    if (err != hipSuccess) return (int)err;
    hipDeviceProp_t prop;
    err = hipGetDeviceProperties(&prop, dev);
    if (err != hipSuccess) return (int)err;
    snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

int invpref_forward_hip(const InvPrefTables *tables, const int64_t *users, const int64_t *items, const int64_t *envs,
                        int64_t B, uint32_t flags, float *invariant_score, float *env_aware_score, float *env_outputs,
                        void *stream) {
    int rc = check_tables(tables);
    if (rc) return rc;
    if (B < 0 || (B > 0 && (!users || !items || !envs || !invariant_score || !env_aware_score || !env_outputs)))
        return INVPREF_EINVAL;
    if (B == 0) return 0;
    const DevTables t = dev_tables(tables);
    const bool vec = vec_ok(tables);
    const int nc = vec ? nc_of(t.D) : 4, emax = emax_of(t.E);
    const int nb = estep_blocks(B);
    const size_t lds = sizeof(float) * (2 * (size_t)t.E * nc * 64 + t.E);
    hipStream_t st = (hipStream_t)stream;
#define CALL(NCV, VECV, EMAXV)                                                                               \
    hipLaunchKernelGGL((forward_kernel<NCV, VECV, EMAXV>), dim3(nb), dim3(256), lds, st, t, users, items, envs, B, \
                       flags, invariant_score, env_aware_score, env_outputs)
    DISPATCH_NVE(nc, vec, emax, CALL);
#undef CALL
    return (int)hipGetLastError();
}

size_t invpref_mstep_workspace_bytes(const InvPrefTables *tables, int64_t B) {
    if (check_tables(tables)) return 0;
    const int nc = 4;  // upper bound independent of alignment
    const size_t slab_len = 2 * (size_t)tables->env_num * nc * 64 + 16 + kLossSlots;
    (void)B;
    return sizeof(float) * slab_len * kMstepMaxBlocks;
}

int invpref_mstep_grad_hip(const InvPrefTables *tables, const InvPrefTables *grads, const int64_t *users,
                           const int64_t *items, const int64_t *envs, const float *scores, const float *sample_weights,
                           int64_t B, int64_t batch_norm, const InvPrefCoefs *coefs, uint32_t flags, float *losses6,
                           void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_tables(tables);
    if (rc) return rc;
    rc = check_tables(grads);
    if (rc) return rc;
    if (!coefs || !losses6 || !workspace || B < 0 || batch_norm < B || batch_norm <= 0) return INVPREF_EINVAL;
    if (B > 0 && (!users || !items || !envs || !scores)) return INVPREF_EINVAL;
    if ((flags & (INVPREF_REWEIGHT_REC | INVPREF_REWEIGHT_CLS)) && B > 0 && !sample_weights) return INVPREF_EINVAL;
    if (workspace_bytes < invpref_mstep_workspace_bytes(tables, B)) return INVPREF_EWORKSPACE;
    const DevTables t = dev_tables(tables);
    const DevGrads g = dev_grads(grads);
    const bool vec = vec_ok(tables) && vec_ok(grads);
    const int nc = vec ? nc_of(t.D) : 4, emax = emax_of(t.E);
    const int DP = nc * 64, EDP = t.E * DP;
    const int nb = mstep_blocks(B);
    StepScalars k;
    k.ca = coefs->invariant_coe; k.cb = coefs->env_aware_coe; k.cc = coefs->env_coe; k.alpha = coefs->alpha;
    k.invB = 1.0f / (float)batch_norm;
    k.r2 = coefs->L2_coe / ((float)batch_norm * (float)t.D);
    k.r1 = coefs->L1_coe / (2.0f * (float)batch_norm * (float)t.D);
    const bool dcol = nc * emax > 4;
    const size_t G = (dcol ? 256 : kMstepThreads) / kRow;
    const size_t R = dcol ? (2 * (size_t)EDP > G * 2 * DP ? 2 * (size_t)EDP : G * 2 * DP) : 2 * (size_t)EDP;
    const size_t nwaves = (dcol ? 256 : kMstepThreads) / 64;
    const size_t lds = sizeof(float) * (2 * (size_t)EDP + R + 2 * emax + kLossSlots + (dcol ? G * (emax + 1) : 0) +
                                        nwaves * 4 * DP) + 8 * nwaves * 4 + 8;
    hipStream_t st = (hipStream_t)stream;
    float *slabs = (float *)workspace;
#define CALL(NCV, VECV, EMAXV)                                                                                       \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mstep_atomic_kernel<NCV, VECV, EMAXV, false, (NCV * EMAXV > 4)>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((mstep_atomic_kernel<NCV, VECV, EMAXV, false, (NCV * EMAXV > 4)>), dim3(nb), dim3((NCV * EMAXV > 4) ? 256 : kMstepThreads), lds, st, t, g, users, \
                       items, envs, scores, sample_weights, B, k, flags, slabs, Upstream{nullptr, nullptr, nullptr})
    DISPATCH_NVE(nc, vec, emax, CALL);
#undef CALL
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return (int)err;
    const int slab_len = 2 * EDP + emax + kLossSlots;
    // the loss tail (kLossSlots) must sit entirely inside the last 64-column block
    const int nfb = (slab_len + 63) / 64;
    if ((nfb - 1) * 64 > 2 * EDP + emax) return INVPREF_EUNSUPPORTED;
    hipLaunchKernelGGL((mstep_finish_kernel<0>), dim3(nfb), dim3(1024), 0, st, t, g, slabs, nb, DP, emax, k,
                       coefs->L2_coe, coefs->L1_coe, batch_norm, flags, losses6);
    return (int)hipGetLastError();
}

int invpref_backward_hip(const InvPrefTables *tables, const InvPrefTables *grads, const int64_t *users,
                         const int64_t *items, const int64_t *envs, int64_t B, uint32_t flags, float alpha,
                         const float *d_invariant_score, const float *d_env_aware_score, const float *d_env_outputs,
                         void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_tables(tables);
    if (rc) return rc;
    rc = check_tables(grads);
    if (rc) return rc;
    if (!workspace || B < 0 || (B > 0 && (!users || !items || !envs))) return INVPREF_EINVAL;
    if (workspace_bytes < invpref_mstep_workspace_bytes(tables, B)) return INVPREF_EWORKSPACE;
    if (B == 0) return 0;
    const DevTables t = dev_tables(tables);
    const DevGrads g = dev_grads(grads);
    const bool vec = vec_ok(tables) && vec_ok(grads);
    const int nc = vec ? nc_of(t.D) : 4, emax = emax_of(t.E);
    const int DP = nc * 64, EDP = t.E * DP;
    const int nb = mstep_blocks(B);
    StepScalars k{};
    k.alpha = alpha;
    k.invB = 1.f;
    const uint32_t fl = flags & INVPREF_IMPLICIT;
    const bool dcol = nc * emax > 4;
    const size_t G = (dcol ? 256 : kMstepThreads) / kRow;
    const size_t R = dcol ? (2 * (size_t)EDP > G * 2 * DP ? 2 * (size_t)EDP : G * 2 * DP) : 2 * (size_t)EDP;
    const size_t nwaves = (dcol ? 256 : kMstepThreads) / 64;
    const size_t lds = sizeof(float) * (2 * (size_t)EDP + R + 2 * emax + kLossSlots + (dcol ? G * (emax + 1) : 0) +
                                        nwaves * 4 * DP) + 8 * nwaves * 4 + 8;
    hipStream_t st = (hipStream_t)stream;
    float *slabs = (float *)workspace;
    const Upstream up{d_invariant_score, d_env_aware_score, d_env_outputs};
#define CALL(NCV, VECV, EMAXV)                                                                                            \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mstep_atomic_kernel<NCV, VECV, EMAXV, true, (NCV * EMAXV > 4)>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((mstep_atomic_kernel<NCV, VECV, EMAXV, true, (NCV * EMAXV > 4)>), dim3(nb), dim3((NCV * EMAXV > 4) ? 256 : kMstepThreads), lds, st, t, g, users, \
                       items, envs, (const float *)nullptr, (const float *)nullptr, B, k, fl, slabs, up)
    DISPATCH_NVE(nc, vec, emax, CALL);
#undef CALL
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return (int)err;
    const int slab_len = 2 * EDP + emax + kLossSlots;
    const int nfb = (slab_len + 63) / 64;
    hipLaunchKernelGGL((mstep_finish_kernel<0>), dim3(nfb), dim3(1024), 0, st, t, g, slabs, nb, DP, emax, k, 0.f, 0.f,
                       (int64_t)1, fl, (float *)nullptr);
    return (int)hipGetLastError();
}

int invpref_predict_hip(const float *user_table, const float *item_table, const int64_t *users, int64_t n_users,
                        int64_t item_num, int64_t factor_num, int apply_sigmoid, float *out, void *stream) {
    if (!user_table || !item_table || !out || n_users < 0 || item_num <= 0 || factor_num <= 0) return INVPREF_EINVAL;
    if (factor_num > INVPREF_MAX_FACTORS) return INVPREF_EUNSUPPORTED;
    if (n_users == 0) return 0;
    if (!users) return INVPREF_EINVAL;
    const bool vec = (factor_num % 4 == 0) && !((reinterpret_cast<uintptr_t>(user_table) | reinterpret_cast<uintptr_t>(item_table)) & 15u);
    const int nc = vec ? nc_of((int)factor_num) : 4;
    const int64_t rows_per_block = 256 / kRow;
    const unsigned gx = (unsigned)((n_users + rows_per_block - 1) / rows_per_block);
    unsigned gy = 1;  // split the item sweep when there are few users, to fill the chip
    while ((int64_t)gx * gy < 1024 && gy * 64 < (unsigned)item_num) gy *= 2;
    // item slices start on multiples of 16 for the 64-byte result segments: the slice length is rounded UP to one (round 6:
    // it used to fall back to ONE slice whenever I / gy was not a multiple of 16 -- MIND's 51 283 items with 256 test users per
    // call ran on 16 workgroups, 4.9 s per evaluate())
    int per = (int)((item_num + gy - 1) / gy);
    per = (per + 15) / 16 * 16;
    gy = (unsigned)((item_num + per - 1) / per);
    hipStream_t st = (hipStream_t)stream;
    // full 64-float chunks (factor_num 64 / 128 / 256) on 16-byte-aligned tables: the dense contraction on the matrix cores
    // (predict_mm_kernel: fp32 MFMA, the canonical summation order); INVPREF_PREDICT_MM=0 keeps the vector-ALU sweep
    static const bool mm_off = std::getenv("INVPREF_PREDICT_MM") != nullptr && std::getenv("INVPREF_PREDICT_MM")[0] == '0';
    if (vec && !mm_off && (factor_num == 64 || factor_num == 128 || factor_num == 256) && n_users >= 16 && item_num >= 16) {
        const unsigned ux = (unsigned)((n_users + 63) / 64);
        // item groups: enough workgroups for two per CU, at least eight 16-item steps each
        unsigned ig = (unsigned)((512 + ux - 1) / ux);
        const unsigned steps_total = (unsigned)((item_num + 15) / 16);
        if (ig > (steps_total + 7) / 8) ig = (steps_total + 7) / 8;
        if (ig < 1) ig = 1;
        const int steps_per = (int)((steps_total + ig - 1) / ig);
        ig = (steps_total + steps_per - 1) / steps_per;
#define MCALL(DCV)                                                                                                   \
        do {                                                                                                         \
            const size_t lds = sizeof(float) * 2 * 16 * (64 * DCV + 4);                                              \
            hipLaunchKernelGGL((predict_mm_kernel<DCV>), dim3(ux, ig), dim3(256), lds, st, user_table, item_table, users, n_users, \
                               (int)item_num, apply_sigmoid, out, steps_per);                                        \
        } while (0)
        if (factor_num == 64) MCALL(1); else if (factor_num == 128) MCALL(2); else MCALL(4);
#undef MCALL
        return (int)hipGetLastError();
    }
#define PCALL(NCV, VECV)                                                                                   \
    hipLaunchKernelGGL((predict_kernel<NCV, VECV>), dim3(gx, gy), dim3(256), 0, st, user_table, item_table, users, \
                       n_users, (int)item_num, (int)factor_num, apply_sigmoid, out, per)
    if (!vec) { PCALL(4, false); } else if (nc == 1) { PCALL(1, true); } else if (nc == 2) { PCALL(2, true); } else { PCALL(4, true); }
#undef PCALL
    return (int)hipGetLastError();
}

int invpref_adam_hip(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, int64_t step, double lr,
                     double beta1, double beta2, double eps, int zero_grad, void *stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return INVPREF_EINVAL;
    if (n == 0) return 0;
    const uintptr_t ap = reinterpret_cast<uintptr_t>(param), ag = reinterpret_cast<uintptr_t>(grad);
    const uintptr_t am = reinterpret_cast<uintptr_t>(exp_avg), av = reinterpret_cast<uintptr_t>(exp_avg_sq);
    if ((ap | ag | am | av) & 3u) return INVPREF_EINVAL;   // not even float-aligned
    // float4 body from the first common 16-byte boundary; buffers misaligned differently go one float at a time
    int64_t head = ((16u - (ap & 15u)) & 15u) >> 2;
    if ((ag & 15u) != (ap & 15u) || (am & 15u) != (ap & 15u) || (av & 15u) != (ap & 15u)) head = n > 4 ? n : 4;
    if (head < 4 && head > n) head = n;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamScalars a;
    a.step_size = (float)(lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.w1 = (float)(1.0 - beta1);
    a.b2 = (float)beta2;
    a.w2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    int64_t nb = (((head >= 4 ? n : n >> 2)) + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, a, zero_grad, head);
    return (int)hipGetLastError();
}

int invpref_pack_rows_hip(const float *flat, const int64_t *row_offsets, int64_t n_rows, int32_t D, int64_t tail_offset,
                          int64_t tail_len, float *packed, int vec_ok, void *stream) {
    return pack_rows_launch(false, const_cast<float *>(flat), row_offsets, n_rows, D, tail_offset, tail_len, packed, vec_ok,
                            stream);
}

int invpref_unpack_rows_hip(float *flat, const int64_t *row_offsets, int64_t n_rows, int32_t D, int64_t tail_offset,
                            int64_t tail_len, const float *packed, int vec_ok, void *stream) {
    return pack_rows_launch(true, flat, row_offsets, n_rows, D, tail_offset, tail_len, const_cast<float *>(packed), vec_ok,
                            stream);
}

static int adam_ranges_launch(float *param, float *grad, float *exp_avg, float *exp_avg_sq, const int64_t *offsets,
                              const int64_t *lengths, int32_t n_ranges, const AdamScalars &a, int zero_grad,
                              const InvPrefAdamSchedule *sched, void *stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !offsets || !lengths || n_ranges < 1 || n_ranges > 4)
        return INVPREF_EINVAL;
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15u)
        return INVPREF_EINVAL;
    if (sched && (!sched->state || !sched->table || sched->n <= 0)) return INVPREF_EINVAL;
    AdamRanges r{};
    int64_t total = 0;
    for (int i = 0; i < n_ranges; i++) {
        if (offsets[i] < 0 || lengths[i] <= 0 || (offsets[i] & 3) || (lengths[i] & 3)) return INVPREF_EINVAL;
        r.off4[i] = offsets[i] >> 2;
        total += lengths[i] >> 2;
        r.end4[i] = total;
    }
    r.n = n_ranges;
    int64_t nb = (total + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(adam_ranges_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, r, a, zero_grad, sched ? sched->state : nullptr,
                       sched ? reinterpret_cast<const SchedRowK *>(sched->table) : nullptr, sched ? sched->n : 0,
                       sched ? (sched->slot & 1) : 0);
    return (int)hipGetLastError();
}

int invpref_adam_ranges_hip(float *param, float *grad, float *exp_avg, float *exp_avg_sq, const int64_t *offsets,
                            const int64_t *lengths, int32_t n_ranges, int64_t step, double lr, double beta1, double beta2,
                            double eps, int zero_grad, void *stream) {
    if (step < 1) return INVPREF_EINVAL;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamScalars a;
    a.step_size = (float)(lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.w1 = (float)(1.0 - beta1);
    a.b2 = (float)beta2;
    a.w2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    return adam_ranges_launch(param, grad, exp_avg, exp_avg_sq, offsets, lengths, n_ranges, a, zero_grad, nullptr, stream);
}

int invpref_adam_ranges_sched_hip(float *param, float *grad, float *exp_avg, float *exp_avg_sq, const int64_t *offsets,
                                  const int64_t *lengths, int32_t n_ranges, const InvPrefAdamSchedule *sched,
                                  int zero_grad, void *stream) {
    if (!sched) return INVPREF_EINVAL;
    return adam_ranges_launch(param, grad, exp_avg, exp_avg_sq, offsets, lengths, n_ranges, AdamScalars{}, zero_grad,
                              sched, stream);
}

size_t invpref_estep_workspace_bytes(const InvPrefTables *tables, int64_t N) {
    // count slabs of the workgroups | (invpref_estep_perm_hip) one packed permutation per interaction
    const int64_t E = tables ? tables->env_num : INVPREF_MAX_ENVS;
    return sizeof(int) * (size_t)(E + 1) * kEstepMaxBlocks + sizeof(unsigned long long) * (size_t)(N > 0 ? N : 0);
}

int invpref_stat_envs_hip(const int64_t *envs, int64_t N, int64_t env_num, int64_t *counts, float *class_weights,
                          float *sample_weights, void *workspace, size_t workspace_bytes, void *stream) {
    if (!envs || !counts || !workspace || N <= 0 || env_num <= 0) return INVPREF_EINVAL;
    if (env_num > INVPREF_MAX_ENVS) return INVPREF_EUNSUPPORTED;
    if (workspace_bytes < sizeof(int) * (size_t)(env_num + 1) * kEstepMaxBlocks) return INVPREF_EWORKSPACE;
    int64_t nb = (N + 255) / 256;
    if (nb > kEstepMaxBlocks) nb = kEstepMaxBlocks;
    hipStream_t st = (hipStream_t)stream;
    int *slabs = (int *)workspace;
    hipLaunchKernelGGL(env_hist_kernel, dim3((unsigned)nb), dim3(256), 0, st, envs, N, (int)env_num, slabs);
    hipLaunchKernelGGL(stat_envs_kernel, dim3((unsigned)nb), dim3(256), 0, st, envs, N, (int)env_num, slabs, (int)nb,
                       counts, (int64_t *)nullptr, class_weights, sample_weights);
    return (int)hipGetLastError();
}

int invpref_sample_weights_hip(const int64_t *envs, int64_t N_local, const int64_t *counts, int64_t N_total,
                               int64_t env_num, float *class_weights, float *sample_weights, void *stream) {
    if (!counts || N_local < 0 || N_total <= 0 || env_num <= 0 || (N_local > 0 && sample_weights && !envs))
        return INVPREF_EINVAL;
    if (env_num > INVPREF_MAX_ENVS) return INVPREF_EUNSUPPORTED;
    int64_t nb = (N_local + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > kEstepMaxBlocks) nb = kEstepMaxBlocks;
    hipLaunchKernelGGL(sample_weights_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, envs, N_local, counts,
                       N_total, (int)env_num, class_weights, sample_weights);
    return (int)hipGetLastError();
}

static int estep_launch(const InvPrefTables *tables, const int64_t *users, const int64_t *items, const float *scores,
                        int64_t N, uint32_t flags, const float *eps_rows, const void *perm_index, int index_bytes,
                        const float *eps_base_host, const int64_t *old_envs, int64_t *new_envs, int64_t *counts,
                        int64_t *diff, float *class_weights, float *sample_weights, void *workspace,
                        size_t workspace_bytes, void *stream, const EstepFin *fused = nullptr) {
    int rc = check_tables(tables);
    if (rc) return rc;
    if (N <= 0 || !users || !items || !scores || !new_envs || !workspace) return INVPREF_EINVAL;
    if (!fused && (!counts || !diff)) return INVPREF_EINVAL;
    if (workspace_bytes < invpref_estep_workspace_bytes(tables, N)) return INVPREF_EWORKSPACE;
    const DevTables t = dev_tables(tables);
    const bool vec = vec_ok(tables);
    const int nc = vec ? nc_of(t.D) : 4;
    const int nb = estep_blocks(N);
    const size_t lds = sizeof(float) * ((size_t)t.E * nc * 64 + INVPREF_MAX_ENVS + 1 + INVPREF_MAX_ENVS);
    hipStream_t st = (hipStream_t)stream;
    int *slabs = (int *)workspace;
    unsigned long long *eps_packed = nullptr;
    EpsBase eps_base{};
    Factorials fac{};
    fac.f[0] = 1;
    for (int k = 1; k <= INVPREF_MAX_ENVS; k++) fac.f[k] = fac.f[k - 1] * (unsigned long long)k;
    const void *eps_index = nullptr;   // E <= 7: looked up in the kernel's own LDS table
    int eps_rows_n = 0;
    size_t lds_extra = 0;
    if (perm_index) {
        if (eps_rows || !eps_base_host || (index_bytes != 1 && index_bytes != 4 && index_bytes != 8)) return INVPREF_EINVAL;
        // (an index type must be able to hold E! - 1)
        if ((index_bytes == 1 && t.E > 5) || (index_bytes == 4 && t.E > 12)) return INVPREF_EINVAL;
        for (int k = 0; k < t.E; k++) eps_base.v[k] = eps_base_host[k];
    }
    // the LDS-table form (up to seven environments: the E! packed rows fit LDS at any interaction count -- the indices
    // themselves are fetched pass by pass); otherwise the unrank-first form below
    bool table_form = perm_index && t.E <= kEpsTableMaxE && index_bytes != 8;
    if (table_form) {
        lds_extra = sizeof(unsigned) * (size_t)fac.f[t.E];
        if (lds + lds_extra > 64 * 1024) { table_form = false; lds_extra = 0; }
    }
    if (table_form) {
        eps_index = perm_index;
        eps_rows_n = (int)fac.f[t.E];
    } else if (perm_index) {
        eps_packed = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(workspace) +
                                                           sizeof(int) * (size_t)(t.E + 1) * kEstepMaxBlocks);
        int64_t ub = (N + 255) / 256;
        if (ub > 4096) ub = 4096;
        if (index_bytes == 1)
            hipLaunchKernelGGL(eps_unrank_kernel<uint8_t>, dim3((unsigned)ub), dim3(256), 0, st, (const uint8_t *)perm_index, N, t.E, fac, eps_packed);
        else if (index_bytes == 4)
            hipLaunchKernelGGL(eps_unrank_kernel<int32_t>, dim3((unsigned)ub), dim3(256), 0, st, (const int32_t *)perm_index, N, t.E, fac, eps_packed);
        else
            hipLaunchKernelGGL(eps_unrank_kernel<int64_t>, dim3((unsigned)ub), dim3(256), 0, st, (const int64_t *)perm_index, N, t.E, fac, eps_packed);
    }
    EstepFin fin{};
    // INVPREF_ESTEP_FOLD=epilogue (default) | kernel: where the fused entry point folds the count slabs (see estep_fold_kernel)
    static const bool fold_kernel = std::getenv("INVPREF_ESTEP_FOLD") != nullptr && std::getenv("INVPREF_ESTEP_FOLD")[0] == 'k';
    if (fused) {
        fin = *fused;
        if (!table_form) fin.perm_table = nullptr;
        if (fold_kernel) fin.state = nullptr;          // (the assignment kernel stores plain slabs and takes no ticket)
    }
#define ECALL1(NCV, VECV, NARV)                                                                                   \
    hipLaunchKernelGGL((estep_assign_kernel<NCV, VECV, NARV>), dim3(nb), dim3(kEstepThreads), lds + lds_extra, st, t, users, items, \
                       scores, N, flags, eps_rows, eps_packed, eps_base, eps_index, index_bytes, eps_rows_n, fac, old_envs, \
                       new_envs, slabs, fin)
#define ECALL(NCV, VECV) do { if (narrow) ECALL1(NCV, VECV, true); else ECALL1(NCV, VECV, false); } while (0)
    // (INVPREF_ESTEP_OFFSETS64=1: the 64-bit form regardless -- the only way a test reaches it short of a 4 GB table)
    static const bool force64 = std::getenv("INVPREF_ESTEP_OFFSETS64") != nullptr && std::getenv("INVPREF_ESTEP_OFFSETS64")[0] == '1';
    const bool narrow = !force64 && (uint64_t)std::max(t.U, t.I) * (uint64_t)t.D * 4u < (1ull << 32);
    if (!vec) { ECALL(4, false); } else if (nc == 1) { ECALL(1, true); } else if (nc == 2) { ECALL(2, true); } else { ECALL(4, true); }
#undef ECALL
#undef ECALL1
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return (int)err;
    if (fused && fold_kernel) {
        hipLaunchKernelGGL(estep_fold_kernel, dim3(1), dim3(256), 0, st, slabs, nb, t.E, N, *fused);
        return (int)hipGetLastError();
    }
    if (fused) return 0;   // (counts, diff and class weights came out of the kernel's epilogue; no sample-weight array)
    // every workgroup of stat_envs folds ALL the count slabs for itself before it gathers its share of the sample weights:
    // at most 256 of them (one per CU, a grid-stride share of rows each) -- a thousand workgroups read 40 MB of slabs for
    // 1 MB of weights (round 5: 10.9 us at the Yahoo shape)
    int64_t nb2 = (N + 255) / 256;
    if (nb2 > 256) nb2 = 256;
    hipLaunchKernelGGL(stat_envs_kernel, dim3((unsigned)nb2), dim3(256), 0, st, new_envs, N, t.E, slabs, nb, counts, diff,
                       class_weights, sample_weights);
    return (int)hipGetLastError();
}

int invpref_estep_hip(const InvPrefTables *tables, const int64_t *users, const int64_t *items, const float *scores,
                      int64_t N, uint32_t flags, const float *eps_rows, const int64_t *old_envs, int64_t *new_envs,
                      int64_t *counts, int64_t *diff, float *class_weights, float *sample_weights, void *workspace,
                      size_t workspace_bytes, void *stream) {
    return estep_launch(tables, users, items, scores, N, flags, eps_rows, nullptr, 0, nullptr, old_envs, new_envs, counts,
                        diff, class_weights, sample_weights, workspace, workspace_bytes, stream);
}

int invpref_estep_perm_hip(const InvPrefTables *tables, const int64_t *users, const int64_t *items, const float *scores,
                           int64_t N, uint32_t flags, const void *perm_index, int index_bytes, const float *eps_base,
                           const int64_t *old_envs, int64_t *new_envs, int64_t *counts, int64_t *diff,
                           float *class_weights, float *sample_weights, void *workspace, size_t workspace_bytes,
                           void *stream) {
    if (!perm_index) return INVPREF_EINVAL;
    return estep_launch(tables, users, items, scores, N, flags, nullptr, perm_index, index_bytes, eps_base, old_envs,
                        new_envs, counts, diff, class_weights, sample_weights, workspace, workspace_bytes, stream);
}

/* host helper: the E! packed permutation rows of train.py:86-92 in itertools.permutations order -- row r, position pos:
 * element number (table[r] >> 4 pos) & 15 of the tie-break vector -- for up to seven environments (5 040 rows). */
int invpref_perm_table_fill(int32_t env_num, uint32_t *host_table) {
    if (!host_table || env_num < 1 || env_num > kEpsTableMaxE) return INVPREF_EINVAL;
    int rows = 1;
    for (int k = 2; k <= env_num; k++) rows *= k;
    for (int r = 0; r < rows; r++) {
        unsigned avail = (1u << env_num) - 1u, out = 0;
        int rem = r, f = rows;
        for (int pos = 0; pos < env_num; pos++) {
            f /= (env_num - pos);
            const int d = rem / f;
            rem -= d * f;
            int seen = 0, pick = 0;
            for (int b = 0; b < env_num; b++)
                if ((avail >> b) & 1u) { if (seen == d) pick = b; seen++; }
            avail &= ~(1u << pick);
            out |= (unsigned)pick << (4 * pos);
        }
        host_table[r] = out;
    }
    return rows;
}

int invpref_estep_fused_hip(const InvPrefTables *tables, const int64_t *users, const int64_t *items, const float *scores,
                            int64_t N, uint32_t flags, const void *perm_index, int index_bytes, const float *eps_base,
                            const uint32_t *perm_table, int64_t *envs, int32_t *state, int64_t *ring, int32_t ring_cap,
                            int64_t *counts, int64_t *diff, float *class_weights, void *workspace, size_t workspace_bytes,
                            void *stream) {
    if (!state || !envs || (ring && ring_cap <= 0)) return INVPREF_EINVAL;   // (state: INVPREF_ESTEP_STATE_INTS int32, zeroed once)
    EstepFin fin{};
    fin.state = state; fin.ring = ring; fin.ring_cap = ring_cap; fin.counts = counts; fin.diff = diff;
    fin.class_w = class_weights; fin.perm_table = perm_table;
    return estep_launch(tables, users, items, scores, N, flags, nullptr, perm_index, perm_index ? index_bytes : 0, eps_base, envs,
                        envs, counts, diff, class_weights, nullptr, workspace, workspace_bytes, stream, &fin);
}

}  // extern "C"
