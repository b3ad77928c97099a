// invpref_eval.hip -- evaluation kernels: the part of evaluate.py that runs
// per test user on the rating matrix -- train-item masking, item-pool highlighting, top-k selection and
// the hit labels (evaluate.py:88-112, :11-19) -- and the error sums of the explicit evaluator
// (evaluate.py:199-210).  The rating matrix itself comes from predict_kernel (invpref_kernels.hip).
#include "kernel_common.hpp"

using namespace invpref;

namespace {

constexpr int kMaxK = 64;

// One wavefront per test user.  The user's row of the rating matrix is staged in LDS, masked
// (row[i] = -1024 for the user's train items, evaluate.py:101) and highlighted (row[i] += 1024 for
// the user's item pool, evaluate.py:111); then K argmax passes pick the top-k items in descending
// score order, lowest item id first among equal scores, and each pick is looked up in the user's
// sorted ground-truth list.
__global__ __launch_bounds__(256) void topk_mask_kernel(const float *__restrict__ ratings, int64_t n_users, int n_items,
                                                        const int *__restrict__ mask_ptr, const int *__restrict__ mask_items,
                                                        const int *__restrict__ hl_ptr, const int *__restrict__ hl_items,
                                                        const int *__restrict__ gt_ptr, const int *__restrict__ gt_items,
                                                        int K, int *__restrict__ out_items, float *__restrict__ out_hits) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t u = blockIdx.x * (int64_t)(blockDim.x >> 6) + wave;
    if (u >= n_users) return;
    float *row = lds + (size_t)wave * n_items;
    const float *src = ratings + u * (int64_t)n_items;
    for (int i = lane; i < n_items; i += 64) row[i] = src[i];
    WAVE_LDS_FENCE();
    __builtin_amdgcn_wave_barrier();
    for (int j = mask_ptr[u] + lane; j < mask_ptr[u + 1]; j += 64) row[mask_items[j]] = -1024.0f;
    WAVE_LDS_FENCE();
    __builtin_amdgcn_wave_barrier();
    if (hl_ptr)
        for (int j = hl_ptr[u] + lane; j < hl_ptr[u + 1]; j += 64) row[hl_items[j]] += 1024.0f;
    WAVE_LDS_FENCE();
    __builtin_amdgcn_wave_barrier();
    const int g0 = gt_ptr[u], g1 = gt_ptr[u + 1];
    for (int k = 0; k < K; k++) {
        float best = -__builtin_inff();
        int bi = 0x7fffffff;
        for (int i = lane; i < n_items; i += 64) {
            const float v = row[i];
            if (v > best) { best = v; bi = i; }  // ascending i per lane: the first maximum is the lowest index
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ov = __shfl_xor(best, m, 64);
            const int oi = __shfl_xor(bi, m, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) {
            int lo = g0, hi = g1;  // binary search in the sorted ground-truth list
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (gt_items[mid] < bi) lo = mid + 1; else hi = mid; }
            out_items[u * K + k] = bi;
            out_hits[u * K + k] = (lo < g1 && gt_items[lo] == bi) ? 1.0f : 0.0f;
            if (bi < n_items) row[bi] = -__builtin_inff();
        }
        WAVE_LDS_FENCE();
        __builtin_amdgcn_wave_barrier();
    }
}

// Item counts beyond what four LDS-staged rows allow (16 * n_items bytes <= 160 KB, i.e. n_items <= 10 240): one
// 256-thread workgroup per test user; the rating row stays in global memory (L2-resident: it is re-read once per
// pick) and LDS holds three bit sets over the items -- train-item mask, item-pool highlight, already picked --
// so the same arithmetic (-1024 for a masked item, += 1024 for a highlighted one, picked items out of the race)
// is applied on the fly.  MIND has 51 283 items (19 KB of bit sets).
__global__ __launch_bounds__(256) void topk_mask_big_kernel(const float *__restrict__ ratings, int64_t n_users, int n_items,
                                                            const int *__restrict__ mask_ptr, const int *__restrict__ mask_items,
                                                            const int *__restrict__ hl_ptr, const int *__restrict__ hl_items,
                                                            const int *__restrict__ gt_ptr, const int *__restrict__ gt_items,
                                                            int K, int *__restrict__ out_items, float *__restrict__ out_hits) {
    extern __shared__ __attribute__((aligned(16))) unsigned bits[];
    __shared__ float wbest[4];
    __shared__ int wbi[4];
    const int words = (n_items + 31) >> 5;
    unsigned *bm = bits, *bh = bits + words, *bp = bits + 2 * words;
    const int64_t u = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 3 * words; i += blockDim.x) bits[i] = 0u;
    __syncthreads();
    for (int j = mask_ptr[u] + threadIdx.x; j < mask_ptr[u + 1]; j += blockDim.x)
        atomicOr(bm + (mask_items[j] >> 5), 1u << (mask_items[j] & 31));
    if (hl_ptr)
        for (int j = hl_ptr[u] + threadIdx.x; j < hl_ptr[u + 1]; j += blockDim.x)
            atomicOr(bh + (hl_items[j] >> 5), 1u << (hl_items[j] & 31));
    __syncthreads();
    const float *src = ratings + u * (int64_t)n_items;
    const int g0 = gt_ptr[u], g1 = gt_ptr[u + 1];
    for (int k = 0; k < K; k++) {
        float best = -__builtin_inff();
        int bi = 0x7fffffff;
        for (int i = threadIdx.x; i < n_items; i += blockDim.x) {
            const unsigned w = (unsigned)i >> 5, b = 1u << (i & 31);
            float v = (bm[w] & b) ? -1024.0f : src[i];
            if (bh[w] & b) v += 1024.0f;
            if (bp[w] & b) v = -__builtin_inff();
            if (v > best) { best = v; bi = i; }  // ascending i per thread: the first maximum is the lowest index
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ov = __shfl_xor(best, m, 64);
            const int oi = __shfl_xor(bi, m, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { wbest[wave] = best; wbi[wave] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; w++)
                if (wbest[w] > best || (wbest[w] == best && wbi[w] < bi)) { best = wbest[w]; bi = wbi[w]; }
            int lo = g0, hi = g1;  // binary search in the sorted ground-truth list
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (gt_items[mid] < bi) lo = mid + 1; else hi = mid; }
            out_items[u * K + k] = bi;
            out_hits[u * K + k] = (lo < g1 && gt_items[lo] == bi) ? 1.0f : 0.0f;
            if (bi < n_items) bp[bi >> 5] |= 1u << (bi & 31);
        }
        __syncthreads();
    }
}

// The same selection for large item counts WITHOUT k passes over the row (round 6: MIND's 51 283 items x top-40 was 40 sweeps
// of a 205 KB row per test user -- 0.58 s of a 0.60 s evaluate()): a RADIX SELECT finds the k-th best (value, id) pair in three
// histogram passes over the row -- 11 + 11 + 10 bits of the value's order-preserving key -- plus, only when more items tie with
// the k-th value than are needed, two passes over the ids of the tied items; one more pass collects the k winners, which one
// wave ranks by (value descending, id ascending) -- exactly the order of k argmax passes with the lowest id first among equal
// scores -- and looks up in the ground-truth list.  Same masking arithmetic as above: -1024 for a train item, += 1024 for
// an item of the pool.
__device__ __forceinline__ unsigned order_key(float v) {
    v = v + 0.0f;                                   // -0 -> +0: equal values, equal keys
    if (v != v) return 0u;                          // a NaN score is never picked (`v > best` is false for it above)
    const unsigned u = __builtin_bit_cast(unsigned, v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ __launch_bounds__(256) void topk_select_kernel(const float *__restrict__ ratings, int64_t n_users, int n_items,
                                                          const int *__restrict__ mask_ptr, const int *__restrict__ mask_items,
                                                          const int *__restrict__ hl_ptr, const int *__restrict__ hl_items,
                                                          const int *__restrict__ gt_ptr, const int *__restrict__ gt_items,
                                                          int K, int *__restrict__ out_items, float *__restrict__ out_hits) {
    extern __shared__ __attribute__((aligned(16))) unsigned bits[];
    __shared__ int hist[2048];
    __shared__ unsigned s_sel;
    __shared__ int s_need, s_n;
    __shared__ unsigned ckey[kMaxK];
    __shared__ int cid[kMaxK];
    const int words = (n_items + 31) >> 5;
    unsigned *bm = bits, *bh = bits + words;
    const int64_t u = blockIdx.x;
    for (int i = threadIdx.x; i < 2 * words; i += blockDim.x) bits[i] = 0u;
    if (threadIdx.x == 0) { s_need = K; s_n = 0; }
    __syncthreads();
    for (int j = mask_ptr[u] + threadIdx.x; j < mask_ptr[u + 1]; j += blockDim.x)
        atomicOr(bm + (mask_items[j] >> 5), 1u << (mask_items[j] & 31));
    if (hl_ptr)
        for (int j = hl_ptr[u] + threadIdx.x; j < hl_ptr[u + 1]; j += blockDim.x)
            atomicOr(bh + (hl_items[j] >> 5), 1u << (hl_items[j] & 31));
    __syncthreads();
    const float *src = ratings + u * (int64_t)n_items;
    auto key_of = [&](int i) {
        const unsigned w = (unsigned)i >> 5, b = 1u << (i & 31);
        float v = (bm[w] & b) ? -1024.0f : src[i];
        if (bh[w] & b) v += 1024.0f;
        return order_key(v);
    };
    // one radix pass: among the items `live` accepts, the histogram of digit(i); thread 0 then walks the bins from the top
    // (descending = true) or the bottom until the s_need-th item falls into a bin: s_sel = that bin, s_need = its rank inside
    auto pass = [&](int nbins, bool descending, auto live, auto digit) {
        for (int b = threadIdx.x; b < nbins; b += blockDim.x) hist[b] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n_items; i += blockDim.x) {
            const unsigned k = key_of(i);
            if (live(i, k)) atomicAdd(&hist[digit(i, k)], 1);
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            // (one wave: lane l owns nbins / 64 consecutive bins in walk order; a prefix over the lanes finds the lane, the lane
            //  its bin)
            const int per = nbins / 64, lane = threadIdx.x;
            int mine = 0;
            for (int j = 0; j < per; j++) {
                const int pos = lane * per + j;
                mine += hist[descending ? nbins - 1 - pos : pos];
            }
            int incl = mine;
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
            const int before = incl - mine, need = s_need;
            if (before < need && need <= incl) {
                int cum = before;
                for (int j = 0; j < per; j++) {
                    const int pos = lane * per + j, b = descending ? nbins - 1 - pos : pos, h = hist[b];
                    if (cum + h >= need) { s_sel = (unsigned)b; s_need = need - cum; s_n = h; break; }
                    cum += h;
                }
            }
        }
        __syncthreads();
    };
    unsigned pre = 0;      // the key's bits fixed so far
    pass(2048, true, [&](int, unsigned) { return true; }, [&](int, unsigned k) { return k >> 21; });
    pre = s_sel << 21;
    __syncthreads();
    pass(2048, true, [&](int, unsigned k) { return (k >> 21) == (pre >> 21); }, [&](int, unsigned k) { return (k >> 10) & 2047u; });
    pre |= s_sel << 10;
    __syncthreads();
    pass(1024, true, [&](int, unsigned k) { return (k >> 10) == (pre >> 10); }, [&](int, unsigned k) { return k & 1023u; });
    const unsigned T = pre | s_sel;     // the k-th best value's key; s_need of the s_n items that carry it are taken
    int id_T = 0x7fffffff;              // ... the ones with id <= id_T
    const bool tie = s_need < s_n;
    __syncthreads();
    if (tie) {                          // (workgroup-uniform) the s_need LOWEST ids among the tied items: 10 + 10 bits of the id
        pass(1024, false, [&](int, unsigned k) { return k == T; }, [&](int i, unsigned) { return (unsigned)i >> 10; });
        const unsigned hi = s_sel;
        __syncthreads();
        pass(1024, false, [&](int i, unsigned k) { return k == T && ((unsigned)i >> 10) == hi; }, [&](int i, unsigned) { return (unsigned)i & 1023u; });
        id_T = (int)((hi << 10) | s_sel);
        __syncthreads();
    }
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n_items; i += blockDim.x) {
        const unsigned k = key_of(i);
        if (k > T || (k == T && i <= id_T)) {
            const int at = atomicAdd(&s_n, 1);
            if (at < kMaxK) { ckey[at] = k; cid[at] = i; }
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {             // rank the K winners: (key descending, id ascending)
        const int lane = threadIdx.x, n = min(s_n, K);
        const unsigned mk = lane < n ? ckey[lane] : 0u;
        const int mi = lane < n ? cid[lane] : 0x7fffffff;
        int rank = 0;
        for (int j = 0; j < n; j++) {
            const unsigned ok = ckey[j];
            const int oi = cid[j];
            rank += (ok > mk || (ok == mk && oi < mi)) ? 1 : 0;
        }
        if (lane < n) {
            const int g0 = gt_ptr[u], g1 = gt_ptr[u + 1];
            int lo = g0, hi = g1;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (gt_items[mid] < mi) lo = mid + 1; else hi = mid; }
            out_items[u * K + rank] = mi;
            out_hits[u * K + rank] = (lo < g1 && gt_items[lo] == mi) ? 1.0f : 0.0f;
        }
    }
}

// sum (a-b)^2 and sum |a-b| in double (evaluate.py:199-203: nn.MSELoss / nn.L1Loss over all test pairs)
__global__ __launch_bounds__(256) void err_sums_kernel(const float *__restrict__ a, const float *__restrict__ b, int64_t n,
                                                       double *__restrict__ out2) {
    double s2 = 0.0, s1 = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i];
        s2 += (double)(d * d);
        s1 += (double)fabsf(d);
    }
    for (int m = 32; m >= 1; m >>= 1) { s2 += __shfl_xor(s2, m, 64); s1 += __shfl_xor(s1, m, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(out2, s2); atomicAdd(out2 + 1, s1); }
}

// ---- static_pop (train.py:509-571).  Accumulators per env: [0] interactions, [1] sum user_cnt, [2] sum item_cnt
// (integers, exact), then doubles: [3] sum user_norm, [4] sum item_norm, [5] sum norm*norm; distinct users:
// [6] count, [7] sum user_cnt, [8] sum user_norm; distinct items: [9] count, [10] sum item_cnt, [11] sum item_norm.
constexpr int kPopAcc = 12;
__device__ __forceinline__ void acc_i(unsigned long long *a, long long v) { atomicAdd(a, (unsigned long long)v); }

__global__ __launch_bounds__(256) void pop_interactions_kernel(const int64_t *__restrict__ users, const int64_t *__restrict__ items,
                                                               const int64_t *__restrict__ envs, int64_t n, int U, int I, int E,
                                                               const int64_t *__restrict__ ucnt, const int64_t *__restrict__ icnt,
                                                               const double *__restrict__ un, const double *__restrict__ in,
                                                               unsigned long long *__restrict__ acc, unsigned char *__restrict__ fu,
                                                               unsigned char *__restrict__ fi) {
    __shared__ unsigned long long si[INVPREF_MAX_ENVS][3];
    __shared__ double sd[INVPREF_MAX_ENVS][3];
    for (int t = threadIdx.x; t < INVPREF_MAX_ENVS * 3; t += blockDim.x) { si[t / 3][t % 3] = 0; sd[t / 3][t % 3] = 0.0; }
    __syncthreads();
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)envs[j];
        if (e < 0 || e >= E) continue;
        const int64_t u = users[j], i = items[j];
        const double a = un[u], b = in[i];
        acc_i(&si[e][0], 1); acc_i(&si[e][1], ucnt[u]); acc_i(&si[e][2], icnt[i]);
        atomicAdd(&sd[e][0], a); atomicAdd(&sd[e][1], b); atomicAdd(&sd[e][2], a * b);
        fu[(int64_t)e * U + u] = 1;   // presence marks: same value from every writer
        fi[(int64_t)e * I + i] = 1;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < E * 3; t += blockDim.x) {
        const int e = t / 3, q = t % 3;
        if (si[e][q]) atomicAdd(acc + e * kPopAcc + q, si[e][q]);
        if (sd[e][q] != 0.0) atomicAdd(reinterpret_cast<double *>(acc + e * kPopAcc + 3 + q), sd[e][q]);
    }
}

// distinct users (side 0) / items (side 1) of every env: one thread per (env, row) presence mark
__global__ __launch_bounds__(256) void pop_distinct_kernel(const unsigned char *__restrict__ flags, int rows, int E,
                                                           const int64_t *__restrict__ cnt, const double *__restrict__ norm,
                                                           unsigned long long *__restrict__ acc, int base) {
    __shared__ unsigned long long si[INVPREF_MAX_ENVS][2];
    __shared__ double sd[INVPREF_MAX_ENVS];
    for (int t = threadIdx.x; t < INVPREF_MAX_ENVS; t += blockDim.x) { si[t][0] = si[t][1] = 0; sd[t] = 0.0; }
    __syncthreads();
    const int64_t total = (int64_t)E * rows;
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
        if (!flags[j]) continue;
        const int e = (int)(j / rows), r = (int)(j - (int64_t)e * rows);
        acc_i(&si[e][0], 1); acc_i(&si[e][1], cnt[r]);
        atomicAdd(&sd[e], norm[r]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        if (si[e][0]) {
            atomicAdd(acc + e * kPopAcc + base, si[e][0]);
            atomicAdd(acc + e * kPopAcc + base + 1, si[e][1]);
            atomicAdd(reinterpret_cast<double *>(acc + e * kPopAcc + base + 2), sd[e]);
        }
    }
}

__global__ void pop_means_kernel(const unsigned long long *__restrict__ acc, int E, double *__restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const unsigned long long *a = acc + e * kPopAcc;
    const double *d = reinterpret_cast<const double *>(a);
    const double n = (double)a[0], nu = (double)a[6], ni = (double)a[9];
    double *o = out + e * 10;
    o[0] = (double)a[1] / n; o[1] = (double)a[2] / n;      // 0/0 = NaN: np.mean of an empty selection
    o[2] = d[3] / n; o[3] = d[4] / n;
    o[4] = (double)a[7] / nu; o[5] = (double)a[10] / ni;
    o[6] = d[8] / nu; o[7] = d[11] / ni;
    o[8] = (double)(a[1] + a[2]) / n; o[9] = d[5] / n;
}

}  // namespace

extern "C" {

size_t invpref_static_pop_workspace_bytes(int64_t user_num, int64_t item_num, int64_t env_num) {
    if (user_num <= 0 || item_num <= 0 || env_num <= 0 || env_num > INVPREF_MAX_ENVS) return 0;
    return sizeof(unsigned long long) * kPopAcc * (size_t)env_num + (size_t)env_num * (size_t)(user_num + item_num);
}

int invpref_static_pop_hip(const int64_t *users, const int64_t *items, const int64_t *envs, int64_t n, int64_t user_num,
                           int64_t item_num, int64_t env_num, const int64_t *user_cnt, const int64_t *item_cnt,
                           const double *user_cnt_norm, const double *item_cnt_norm, double *out, void *workspace,
                           size_t workspace_bytes, void *stream) {
    if (n < 0 || user_num <= 0 || item_num <= 0 || env_num <= 0 || !user_cnt || !item_cnt || !user_cnt_norm ||
        !item_cnt_norm || !out || !workspace || (n > 0 && (!users || !items || !envs)))
        return INVPREF_EINVAL;
    if (env_num > INVPREF_MAX_ENVS || user_num > INT32_MAX || item_num > INT32_MAX) return INVPREF_EUNSUPPORTED;
    const size_t need = invpref_static_pop_workspace_bytes(user_num, item_num, env_num);
    if (workspace_bytes < need) return INVPREF_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(workspace, 0, need, st);
    if (e != hipSuccess) return (int)e;
    auto *acc = reinterpret_cast<unsigned long long *>(workspace);
    auto *fu = reinterpret_cast<unsigned char *>(acc + kPopAcc * env_num);
    auto *fi = fu + (size_t)env_num * (size_t)user_num;
    const int E = (int)env_num, U = (int)user_num, I = (int)item_num;
    if (n > 0) {
        const unsigned nb = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
        hipLaunchKernelGGL(pop_interactions_kernel, dim3(nb), dim3(256), 0, st, users, items, envs, n, U, I, E, user_cnt,
                           item_cnt, user_cnt_norm, item_cnt_norm, acc, fu, fi);
        const unsigned nu = (unsigned)std::min<int64_t>(((int64_t)E * U + 255) / 256, 2048);
        hipLaunchKernelGGL(pop_distinct_kernel, dim3(nu), dim3(256), 0, st, fu, U, E, user_cnt, user_cnt_norm, acc, 6);
        const unsigned ni = (unsigned)std::min<int64_t>(((int64_t)E * I + 255) / 256, 2048);
        hipLaunchKernelGGL(pop_distinct_kernel, dim3(ni), dim3(256), 0, st, fi, I, E, item_cnt, item_cnt_norm, acc, 9);
    }
    hipLaunchKernelGGL(pop_means_kernel, dim3(1), dim3(64), 0, st, acc, E, out);
    return (int)hipGetLastError();
}

int invpref_eval_topk_hip(const float *ratings, int64_t n_users, int64_t n_items, const int32_t *mask_ptr,
                          const int32_t *mask_items, const int32_t *highlight_ptr, const int32_t *highlight_items,
                          const int32_t *truth_ptr, const int32_t *truth_items, int32_t k, int32_t *out_items,
                          float *out_hits, void *stream) {
    if (!ratings || !mask_ptr || !truth_ptr || !out_items || !out_hits || n_users < 0 || n_items <= 0 || k <= 0)
        return INVPREF_EINVAL;
    if (k > kMaxK || k > n_items || n_items > 400000) return INVPREF_EUNSUPPORTED;
    if (n_users == 0) return 0;
    const size_t lds = sizeof(float) * 4 * (size_t)n_items;
    // INVPREF_TOPK_SELECT=0: the k-pass kernels for every size (A/B, tests); default: the radix select beyond 4 096 items
    static const bool sel_off = getenv("INVPREF_TOPK_SELECT") != nullptr && getenv("INVPREF_TOPK_SELECT")[0] == '0';
    static const bool sel_all = getenv("INVPREF_TOPK_SELECT") != nullptr && getenv("INVPREF_TOPK_SELECT")[0] == '2';
    if (!sel_off && (n_items > 4096 || sel_all) && n_items <= (1 << 20)) {
        const size_t lds_sel = sizeof(unsigned) * 2 * (((size_t)n_items + 31) / 32);
        if (lds_sel > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(topk_select_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sel);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(topk_select_kernel, dim3((unsigned)n_users), dim3(256), lds_sel, (hipStream_t)stream, ratings, n_users,
                           (int)n_items, mask_ptr, mask_items, highlight_ptr, highlight_items, truth_ptr, truth_items, (int)k,
                           out_items, out_hits);
        return (int)hipGetLastError();
    }
    if (lds > 160 * 1024) {   // the four staged rows exceed the CU's LDS: bit-set form, one workgroup per user
        const size_t lds_bits = sizeof(unsigned) * 3 * (((size_t)n_items + 31) / 32);
        if (lds_bits > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(topk_mask_big_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bits);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(topk_mask_big_kernel, dim3((unsigned)n_users), dim3(256), lds_bits, (hipStream_t)stream, ratings,
                           n_users, (int)n_items, mask_ptr, mask_items, highlight_ptr, highlight_items, truth_ptr,
                           truth_items, (int)k, out_items, out_hits);
        return (int)hipGetLastError();
    }
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(topk_mask_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const unsigned nb = (unsigned)((n_users + 3) / 4);
    hipLaunchKernelGGL(topk_mask_kernel, dim3(nb), dim3(256), lds, (hipStream_t)stream, ratings, n_users, (int)n_items,
                       mask_ptr, mask_items, highlight_ptr, highlight_items, truth_ptr, truth_items, (int)k, out_items,
                       out_hits);
    return (int)hipGetLastError();
}

int invpref_eval_error_sums_hip(const float *pred, const float *target, int64_t n, double *out2, void *stream) {
    if (!pred || !target || !out2 || n < 0) return INVPREF_EINVAL;
    hipError_t e = hipMemsetAsync(out2, 0, 2 * sizeof(double), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    if (n == 0) return 0;
    int64_t nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(err_sums_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, pred, target, n, out2);
    return (int)hipGetLastError();
}

}  // extern "C"
