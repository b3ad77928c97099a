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

}  // namespace

extern "C" {

int invpref_eval_topk_hip(const float *ratings, int64_t n_users, int64_t n_items, const int32_t *mask_ptr,
                          const int32_t *mask_items, const int32_t *highlight_ptr, const int32_t *highlight_items,
                          const int32_t *truth_ptr, const int32_t *truth_items, int32_t k, int32_t *out_items,
                          float *out_hits, void *stream) {
    if (!ratings || !mask_ptr || !truth_ptr || !out_items || !out_hits || n_users < 0 || n_items <= 0 || k <= 0)
        return INVPREF_EINVAL;
    if (k > kMaxK || k > n_items || n_items > 36 * 1024) return INVPREF_EUNSUPPORTED;  // 4 rows of <= 36K floats in LDS
    if (n_users == 0) return 0;
    const size_t lds = sizeof(float) * 4 * (size_t)n_items;
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
