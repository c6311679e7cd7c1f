// sjd_mlp_epilogue.h -- the element arithmetic of F3 (SiLU(gate) * up on a gate|up projection), shared by the two kernels that apply it:
// f3_silu_mul (sjd_glue.hip, on the split-K partial planes of G1) and g1_gateup_silu (sjd_gemm.hip, as the epilogue of the projection).
// Both must give the SAME bits (tests/test_gpu_glue.py::test_g1_gateup_silu_matches_g1_then_f3), so nothing here is left to the
// instruction selector: no contraction inside these functions, and the fp16 conversion is an explicit v_cvt_f16_f32 -- fp32 product, then
// the conversion, as the framework ops they replace do (reference modeling_chameleon.py:59-73, 193-195).
#pragma once
#include "../../include/sjd_hip.h"

template <int DT> struct SjdAct;
template <> struct SjdAct<SJD_DTYPE_BF16> {
    static __device__ __forceinline__ unsigned short from_f(float x)
    {
        unsigned u = __float_as_uint(x);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (unsigned short)(u >> 16);
    }
    static __device__ __forceinline__ float to_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
};
template <> struct SjdAct<SJD_DTYPE_F16> {
    // the conversion is an opaque instruction: `(_Float16)(a * b)` is otherwise selected as v_fma_mixlo_f16 in one kernel and as
    // v_mul_f32 + v_cvt_f16_f32 in the other (measured: f3_silu_mul 16 mix instructions, g1_gateup_silu none, results one fp16 ulp apart)
    static __device__ __forceinline__ unsigned short from_f(float x)
    {
        unsigned r;
        asm("v_cvt_f16_f32 %0, %1" : "=v"(r) : "v"(x));
        return (unsigned short)(r & 0xffffu);
    }
    static __device__ __forceinline__ float to_f(unsigned short h) { return (float)(*reinterpret_cast<_Float16 *>(&h)); }
};

// gsum / usum: the summed fp32 partials of the gate / up column; r: the row scale of the folded RMSNorm (1 when there is none).
// -> the activation-dtype bits of silu(dtype(gsum r)) * dtype(usum r), SiLU rounded to the activation dtype before the product.
template <int DT>
__device__ __forceinline__ unsigned short sjd_silu_mul_elem(float gsum, float usum, float r)
{
#pragma clang fp contract(off)
    const float gp = gsum * r, up = usum * r;
    const float gv = SjdAct<DT>::to_f(SjdAct<DT>::from_f(gp));
    const float uv = SjdAct<DT>::to_f(SjdAct<DT>::from_f(up));
    const float sl = gv / (1.0f + __expf(-gv));
    const float sv = SjdAct<DT>::to_f(SjdAct<DT>::from_f(sl));
    const float pr = sv * uv;
    return SjdAct<DT>::from_f(pr);
}

// the same on values that are already in the activation dtype (F3 on a materialised gate|up tensor)
template <int DT>
__device__ __forceinline__ unsigned short sjd_silu_mul_elem_rounded(float gv, float uv)
{
#pragma clang fp contract(off)
    const float sl = gv / (1.0f + __expf(-gv));
    const float sv = SjdAct<DT>::to_f(SjdAct<DT>::from_f(sl));
    const float pr = sv * uv;
    return SjdAct<DT>::from_f(pr);
}

// ---- F1r's element arithmetic (residual add of a projection's summed fp32 partials, then the row statistics), shared by f1r_residual_sumsq
// (sjd_glue.hip, its own graph node) and the reducing epilogue of g1_skinny_gemm (sjd_gemm.hip, round 3): same bits from both
// (tests/test_gpu_glue.py::test_g1_reduce_epilogue_matches_g1_then_f1r).  hx: the residual stream element (already in the activation
// dtype), d: the projection's fp32 sum over the K chunks in chunk order.  -> bits of dtype(hx + dtype(d)): the projection output rounds
// to the activation dtype, the residual add rounds again (reference modeling_chameleon.py:637, 643).
template <int DT>
__device__ __forceinline__ unsigned short sjd_residual_elem(float hx, float d)
{
#pragma clang fp contract(off)
    const float dj = SjdAct<DT>::to_f(SjdAct<DT>::from_f(d));
    const float s = hx + dj;
    return SjdAct<DT>::from_f(s);
}

// ss + x0^2 + x1^2 + x2^2 + x3^2, left to right, products and sums rounded separately
__device__ __forceinline__ float sjd_sumsq4(float ss, float x0, float x1, float x2, float x3)
{
#pragma clang fp contract(off)
    const float p0 = x0 * x0, p1 = x1 * x1, p2 = x2 * x2, p3 = x3 * x3;
    ss = ss + p0; ss = ss + p1; ss = ss + p2; ss = ss + p3;
    return ss;
}
