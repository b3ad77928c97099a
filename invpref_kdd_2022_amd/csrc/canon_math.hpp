// canon_math.hpp -- the "canonical arithmetic" of DESIGN.md §3, device side.
//
// Everything here is built from IEEE fp32 add / mul / fma / div / sqrt, v_rndne and integer bit
// moves only, written out operation by operation (compile with -ffp-contract=off), so that the
// same definition evaluated anywhere else gives the same bits.  That is what makes the E-step
// environment assignment reproducible bit for bit.
//
//   c_exp / c_log : Cody-Waite range reduction + Horner polynomial (cephes single-precision
//                   coefficient sets), ~1-2 ulp
//   c_log1p       : log(1+x) * x/((1+x)-1)
//   row dot       : a row of D floats lives on 16 lanes, lane l owning the float4 chunks
//                   l, l+16, l+32, ... ; each lane runs one fma chain over its elements in
//                   increasing index, then the 16 partials are combined by an xor butterfly in the
//                   order 1, 2, 4, 8
//   sigmoid       : 1/(1+exp(-x))         (aten sigmoid)
//   bce           : (y-1)*max(log1p(-s),-100) - y*max(log(s),-100)   (aten binary_cross_entropy)
#pragma once
#include <hip/hip_runtime.h>

namespace invpref {

__device__ __forceinline__ float bits_to_float(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned float_to_bits(float f) { return __builtin_bit_cast(unsigned, f); }

__device__ __forceinline__ float pow2i(int n) { return bits_to_float((unsigned)(n + 127) << 23); }

__device__ __forceinline__ float c_exp(float x) {
    float n = __builtin_rintf(x * 1.44269504f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r);
    y = y + 1.0f;
    int ni = (int)n;
    int h = ni >> 1;
    float res = (y * pow2i(h)) * pow2i(ni - h);
    res = (x > 88.72283f) ? __builtin_inff() : res;
    res = (x < -87.33654f) ? 0.0f : res;
    return res;
}

__device__ __forceinline__ float c_log(float x) {
    float xs = x;
    int eadj = 0;
    if (x < 1.17549435e-38f) { xs = x * 8388608.0f; eadj = -23; }
    unsigned u = float_to_bits(xs);
    int e = (int)((u >> 23) & 0xffu) - 126 + eadj;
    float m = bits_to_float((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781f) { e -= 1; m = (m + m) - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = __builtin_fmaf(p, m, -1.1514610310e-1f);
    p = __builtin_fmaf(p, m, 1.1676998740e-1f);
    p = __builtin_fmaf(p, m, -1.2420140846e-1f);
    p = __builtin_fmaf(p, m, 1.4249322787e-1f);
    p = __builtin_fmaf(p, m, -1.6668057665e-1f);
    p = __builtin_fmaf(p, m, 2.0000714765e-1f);
    p = __builtin_fmaf(p, m, -2.4999993993e-1f);
    p = __builtin_fmaf(p, m, 3.3333331174e-1f);
    float y = (m * z) * p;
    float fe = (float)e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = m + y;
    r = __builtin_fmaf(fe, 0.693359375f, r);
    // special cases, same precedence as the definition: NaN, <0, ==0, +inf
    r = (x == __builtin_inff()) ? x : r;
    r = (x == 0.0f) ? -__builtin_inff() : r;
    r = (x < 0.0f) ? __builtin_nanf("") : r;
    r = (x != x) ? x : r;
    return r;
}

__device__ __forceinline__ float c_log1p(float x) {
    float u = 1.0f + x;
    float r = c_log(u) * (x / (u - 1.0f));
    return (u == 1.0f) ? x : r;
}

__device__ __forceinline__ float c_sigmoid(float x) { return 1.0f / (1.0f + c_exp(-x)); }

__device__ __forceinline__ float c_bce(float s, float y) {
    float a = c_log1p(-s);
    a = a > -100.0f ? a : -100.0f;
    float b = c_log(s);
    b = b > -100.0f ? b : -100.0f;
    return (y - 1.0f) * a - y * b;
}

// The same value for labels that are exactly 0 or 1 (the implicit path's data, train.py:130-135) with ONE logarithm: for y = 1 the
// definition is (0 * a) - b = -b, for y = 0 it is -a - (0 * b) = -a -- a and b are finite after their clamps (a NaN logarithm
// clamps to -100), so the dropped product is a signed zero and the result is the same float (up to the sign of a zero result,
// which no comparison or sum can see).  The caller checks the labels (a wave-uniform test) and keeps c_bce for anything else.
__device__ __forceinline__ float c_bce_binary(float s, float y) {
    const float x = -s, u = 1.0f + x;
    const bool one = y == 1.0f;
    float l = c_log(one ? s : u);
    const float l1p = (u == 1.0f) ? x : l * (x / (u - 1.0f));     // c_log1p(-s) from log(1 - s)
    l = one ? l : l1p;
    l = l > -100.0f ? l : -100.0f;
    return -l;
}

// aten binary_cross_entropy_backward: (s-y)/max((1-s)*s, 1e-12)
__device__ __forceinline__ float c_dbce(float s, float y) {
    float d = (1.0f - s) * s;
    d = d > 1e-12f ? d : 1e-12f;
    return (s - y) / d;
}

__device__ __forceinline__ float c_sign(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }

// ---- M-step arithmetic: hardware transcendental units (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp).
// The M-step is held to 1e-5 relative on the losses (north_star), not to bit equality -- float sums
// over a minibatch reorder anyway -- so it does not pay for the canonical polynomials: a sigmoid is
// 4 instructions here against ~40.  (The E-step and forward() keep the canonical forms.)
__device__ __forceinline__ float f_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float f_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504f); }
__device__ __forceinline__ float f_log(float x) { return __builtin_amdgcn_logf(x) * 0.693147181f; }
__device__ __forceinline__ float f_sigmoid(float x) { return f_rcp(1.0f + f_exp(-x)); }
// -log(s) and -log(1-s) with the reference's clamp at 100 (aten binary_cross_entropy)
__device__ __forceinline__ float f_bce(float s, float y) {
    float a = f_log(1.0f - s);
    a = a > -100.0f ? a : -100.0f;
    float b = f_log(s);
    b = b > -100.0f ? b : -100.0f;
    return (y - 1.0f) * a - y * b;
}
// (labels exactly 0 or 1: one logarithm, see c_bce_binary)
__device__ __forceinline__ float f_bce_binary(float s, float y) {
    float l = f_log(y == 1.0f ? s : 1.0f - s);
    l = l > -100.0f ? l : -100.0f;
    return -l;
}
__device__ __forceinline__ float f_dbce(float s, float y) {
    float d = (1.0f - s) * s;
    d = d > 1e-12f ? d : 1e-12f;
    return (s - y) * f_rcp(d);
}

// ---- lane exchanges inside a 16-lane row (DPP; a "row" of the DPP unit is exactly 16 lanes)
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// xor-butterfly sum over the 16 lanes of a row, order 1,2,4,8.  quad_perm[1,0,3,2] is xor 1,
// quad_perm[2,3,0,1] is xor 2; once the four lanes of every quad agree, row_half_mirror (l <-> 7-l)
// delivers the value of lane l^4, and once every 8 agree row_mirror (l <-> 15-l) that of l^8.
__device__ __forceinline__ float row16_sum(float x) {
    x = x + dpp_move<0xB1>(x);
    x = x + dpp_move<0x4E>(x);
    x = x + dpp_move<0x141>(x);
    x = x + dpp_move<0x140>(x);
    return x;
}
// the same butterfly with min (E-step argmin over one environment per lane)
__device__ __forceinline__ float row16_min(float x) {
    x = __builtin_fminf(x, dpp_move<0xB1>(x));
    x = __builtin_fminf(x, dpp_move<0x4E>(x));
    x = __builtin_fminf(x, dpp_move<0x141>(x));
    x = __builtin_fminf(x, dpp_move<0x140>(x));
    return x;
}
// the same butterfly with max (M-step softmax over one class per lane)
__device__ __forceinline__ float row16_max(float x) {
    x = __builtin_fmaxf(x, dpp_move<0xB1>(x));
    x = __builtin_fmaxf(x, dpp_move<0x4E>(x));
    x = __builtin_fmaxf(x, dpp_move<0x141>(x));
    x = __builtin_fmaxf(x, dpp_move<0x140>(x));
    return x;
}

}  // namespace invpref
