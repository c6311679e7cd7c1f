// sjd_attention_dsplit_fp8.h -- included by sjd_attention.hip behind k1_partial_fp8 (its fp8 helpers)
#pragma once

// k1_dsplit for an fp8 (e4m3) KV cache (BASELINE config 5): the arithmetic of k1_partial_fp8 (K bytes straight into the fp8 MFMA's A operand,
// P scaled by 256 and rounded to fp8, V through a per-wave LDS tile and ds_read_b64_tr_b8), the decomposition of k1_dsplit.  A key row is 128
// bytes, a workgroup's V slice 32 of them: every CU pulls 160 bytes per key instead of 256 -- the column split stays ahead of the key split
// up to twice the context length of the 16-bit cache.
template <int DT, int D, int NW, int DS>
__global__ __launch_bounds__(64 * NW) void k1_dsplit_fp8(
    const unsigned short *__restrict__ q, const unsigned char *__restrict__ kc, const unsigned char *__restrict__ vc,
    const sjd_iter_params *__restrict__ params, const int *__restrict__ key_start, unsigned short *__restrict__ out,
    int n_rows, int H, int H_kv, int S_max, int kv_len_arg, int n_chunks, int B, float k_scale, float v_scale)
{
    constexpr int KP = D / 64;                // 16-byte K pieces per key row and lane group
    constexpr int DW = D / DS;                // output columns (= V bytes per key) of this workgroup
    constexpr int DB = DW / 16;
    constexpr int VROW = DW + 16;             // padded LDS row in bytes
    constexpr float PSCALE = 256.0f;
    constexpr int V_BYTES = NW * K1_KT * VROW;
    constexpr int R_BYTES = NW * K1_ROWS * (DW + K1_RPAD + 2) * 4;
    __shared__ __attribute__((aligned(16))) unsigned char arena[V_BYTES > R_BYTES ? V_BYTES : R_BYTES];
    float (*red_o)[K1_ROWS][DW + K1_RPAD] = reinterpret_cast<float (*)[K1_ROWS][DW + K1_RPAD]>(arena);
    float (*red_ml)[K1_ROWS][2] = reinterpret_cast<float (*)[K1_ROWS][2]>(arena + NW * K1_ROWS * (DW + K1_RPAD) * 4);

    const int n_pairs = n_chunks * H * B;
    int pair, dq;
    if ((n_pairs & 7) == 0) {
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
        dq = slot % DS;
        pair = (slot / DS) * 8 + xcd;
    } else {
        dq = blockIdx.x % DS;
        pair = blockIdx.x / DS;
    }
    const int chunk = pair % n_chunks, head = (pair / n_chunks) % H, b = pair / (n_chunks * H);
    const int G = H / H_kv, hkv = head / G;
    int kv_base, n_total, kstart;
    k1_entry(params, key_start, b, kv_len_arg, n_rows, kv_base, n_total, kstart);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    unsigned char *vl = arena + (size_t)w * K1_KT * VROW;
    const int row0 = chunk * K1_ROWS;
    const int n_c = min(K1_ROWS, n_total - row0);
    const int kv_len = kv_base + row0;
    const int total = kv_len + max(n_c, 0);
    const float scale = rsqrtf((float)D) * k_scale;
    const int t_lo = kstart / K1_KT, t_hi = (total + K1_KT - 1) / K1_KT;

    long qf[KP][2], qfl[KP][2];           // (qfl: the residual operand, K1_FP8_HILO)                           // Q as fp8, in the byte order of the K pieces (k1_partial_fp8)
    {
        const bool rv = (c < n_c);
        const unsigned short *qp = q + (((size_t)b * n_rows + (row0 + (rv ? c : 0))) * H + head) * D + 16 * g;
#pragma unroll
        for (int p = 0; p < KP; ++p) {
            unsigned wds[4] = {0u, 0u, 0u, 0u}, wdl[4] = {0u, 0u, 0u, 0u};
            if (rv) {
                const u32x4 lo = *reinterpret_cast<const u32x4 *>(qp + 64 * p), hi = *reinterpret_cast<const u32x4 *>(qp + 64 * p + 8);
                float f[16];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f[2 * i] = k1_to_f32<DT>((unsigned short)(lo[i] & 0xffffu)); f[2 * i + 1] = k1_to_f32<DT>((unsigned short)(lo[i] >> 16));
                    f[8 + 2 * i] = k1_to_f32<DT>((unsigned short)(hi[i] & 0xffffu)); f[8 + 2 * i + 1] = k1_to_f32<DT>((unsigned short)(hi[i] >> 16));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) pack4_fp8_hilo(f[4 * i], f[4 * i + 1], f[4 * i + 2], f[4 * i + 3], wds[i], wdl[i]);
            }
            qf[p][0] = as_long(wds[0], wds[1]);
            qf[p][1] = as_long(wds[2], wds[3]);
            qfl[p][0] = as_long(wdl[0], wdl[1]);
            qfl[p][1] = as_long(wdl[2], wdl[3]);
        }
    }
    const unsigned char *kbase = kc + ((size_t)b * H_kv + hkv) * (size_t)S_max * D;
    const unsigned char *vbase = vc + ((size_t)b * H_kv + hkv) * (size_t)S_max * D + dq * DW;

    float m_run = -INFINITY, l_run = 0.0f;
    f32x4 o_acc[DB], o_lo[DB];            // (o_lo: the products with P's residual operand, K1_FP8_HILO)
#pragma unroll
    for (int db = 0; db < DB; ++db) { o_acc[db] = f32x4{0.f, 0.f, 0.f, 0.f}; o_lo[db] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    constexpr int LPR = DW / 16;              // lanes per V slice row
    static_assert(K1_KT * LPR == 64, "one 16-byte piece per lane covers the V slice tile");
    u32x4 kreg[2][KP], kn[2][KP], vstage, vstage2;
    auto load_tile = [&](int t, u32x4 (&kd)[2][KP], u32x4 &vd) {
        const unsigned char *kt = kbase + (size_t)(t * K1_KT) * D;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int p = 0; p < KP; ++p)
                kd[kb][p] = *reinterpret_cast<const u32x4 *>(kt + (size_t)(16 * kb + c) * D + 64 * p + 16 * g);
        vd = *reinterpret_cast<const u32x4 *>(vbase + (size_t)(t * K1_KT + lane / LPR) * D + 16 * (lane % LPR));
    };
    auto store_v = [&](int t, const u32x4 &vd) {          // bytes of keys >= total may decode to NaN: zeroed while staging
        const bool live = (t * K1_KT + lane / LPR) < total;
        *reinterpret_cast<u32x4 *>(vl + (lane / LPR) * VROW + 16 * (lane % LPR)) = live ? vd : u32x4{0u, 0u, 0u, 0u};
    };
    const int jrow = c >> 1;
    const unsigned char *vrd = vl + (16 * (jrow >> 2) + 4 * g + (jrow & 3)) * VROW + 8 * (c & 1);
    auto compute_tile = [&](int t) {
        f32x4 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            st[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#if K1_FP8_HILO
            f32x4 sl = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
#pragma unroll
            for (int p = 0; p < KP; ++p) {
                st[kb] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(as_long(kreg[kb][p][0], kreg[kb][p][1]), qf[p][0], st[kb], 0, 0, 0);
                st[kb] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(as_long(kreg[kb][p][2], kreg[kb][p][3]), qf[p][1], st[kb], 0, 0, 0);
#if K1_FP8_HILO
                sl = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(as_long(kreg[kb][p][0], kreg[kb][p][1]), qfl[p][0], sl, 0, 0, 0);
                sl = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(as_long(kreg[kb][p][2], kreg[kb][p][3]), qfl[p][1], sl, 0, 0, 0);
#endif
            }
#if K1_FP8_HILO
            st[kb] += sl * (1.0f / K1_LO_SCALE);
#endif
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = t * K1_KT + 16 * kb + 4 * g + r;
                const bool vis = (key >= kstart) && (key <= kv_len + c) && (key < total);
                const float sv = vis ? st[kb][r] * scale : -INFINITY;
                st[kb][r] = sv;
                mx = fmaxf(mx, sv);
            }
        mx = k1r_max_across_groups(mx);
        const float m_new = fmaxf(m_run, mx);
        const float m_safe = (m_new == -INFINITY) ? 0.0f : m_new;
        const float alpha = __expf(m_run - m_safe);
        float rs = 0.0f, pv[8];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(st[kb][r] - m_safe);
                rs += e;
                pv[4 * kb + r] = e * PSCALE;
            }
        rs = k1r_sum_across_groups(rs);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#if K1_FP8_HILO
        unsigned ph0, pl0, ph1, pl1;
        pack4_fp8_hilo(pv[0], pv[1], pv[2], pv[3], ph0, pl0);
        pack4_fp8_hilo(pv[4], pv[5], pv[6], pv[7], ph1, pl1);
        const long pfrag = as_long(ph0, ph1), pfrag_lo = as_long(pl0, pl1);
#else
        const long pfrag = as_long(pack4_fp8(pv[0], pv[1], pv[2], pv[3]), pack4_fp8(pv[4], pv[5], pv[6], pv[7]));
#endif
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const i32x2 vv = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2 *)(vrd + 16 * db));
            f32x4 acc = o_acc[db];
            acc[0] *= alpha; acc[1] *= alpha; acc[2] *= alpha; acc[3] *= alpha;
            o_acc[db] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(as_long((unsigned)vv[0], (unsigned)vv[1]), pfrag, acc, 0, 0, 0);
#if K1_FP8_HILO
            f32x4 al = o_lo[db];
            al[0] *= alpha; al[1] *= alpha; al[2] *= alpha; al[3] *= alpha;
            o_lo[db] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(as_long((unsigned)vv[0], (unsigned)vv[1]), pfrag_lo, al, 0, 0, 0);
#endif
        }
    };
    auto adopt_next = [&](int tn, const u32x4 &vd) {
        store_v(tn, vd);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int p = 0; p < KP; ++p) kreg[kb][p] = kn[kb][p];
    };
    int t = t_lo + w;
    if (t < t_hi) {
        load_tile(t, kreg, vstage);
        if (t + NW < t_hi) {
            load_tile(t + NW, kn, vstage2);
            store_v(t, vstage);
            compute_tile(t);
            adopt_next(t + NW, vstage2);
            t += NW;
        } else {
            store_v(t, vstage);
        }
    }
    for (; t < t_hi; t += NW) {
        const int tn = t + NW;
        const bool has_next = tn < t_hi;
        if (has_next) load_tile(tn, kn, vstage);
        compute_tile(t);
        if (has_next) adopt_next(tn, vstage);
    }
    __syncthreads();
    const float oscale = v_scale / PSCALE;
    if (g == 0) { red_ml[w][c][0] = m_run; red_ml[w][c][1] = l_run; }
#pragma unroll
    for (int db = 0; db < DB; ++db) {
#if K1_FP8_HILO
        o_acc[db] += o_lo[db] * (1.0f / K1_LO_SCALE);
#endif
        *reinterpret_cast<f32x4 *>(&red_o[w][c][16 * db + 4 * g]) = o_acc[db] * oscale;
    }
    __syncthreads();
    constexpr int D4 = DW / 4;
    for (int u = threadIdx.x; u < K1_ROWS * D4; u += 64 * NW) {
        const int row = u / D4, d = (u % D4) * 4;
        const int grow = row0 + row;
        if (grow >= n_rows) continue;
        float mk[NW], lk[NW];
        float4 ok[NW];
#pragma unroll
        for (int kp = 0; kp < NW; ++kp) {
            const float2 v = *reinterpret_cast<const float2 *>(&red_ml[kp][row][0]);
            ok[kp] = *reinterpret_cast<const float4 *>(&red_o[kp][row][d]);
            mk[kp] = v.x;
            lk[kp] = v.y;
        }
        float M = -INFINITY;
#pragma unroll
        for (int kp = 0; kp < NW; ++kp) M = fmaxf(M, mk[kp]);
        const float Ms = (M == -INFINITY) ? 0.0f : M;
        float L = 0.f, O0 = 0.f, O1 = 0.f, O2 = 0.f, O3 = 0.f;
#pragma unroll
        for (int kp = 0; kp < NW; ++kp) {
            const float wgt = __expf(mk[kp] - Ms);
            L += wgt * lk[kp];
            O0 += wgt * ok[kp].x; O1 += wgt * ok[kp].y; O2 += wgt * ok[kp].z; O3 += wgt * ok[kp].w;
        }
        const float inv = L > 0.f ? 1.0f / L : 0.0f;
        uint2 pk{0u, 0u};
        if (grow < n_total) {
            pk.x = (unsigned)Frag<DT>::cvt(O0 * inv) | ((unsigned)Frag<DT>::cvt(O1 * inv) << 16);
            pk.y = (unsigned)Frag<DT>::cvt(O2 * inv) | ((unsigned)Frag<DT>::cvt(O3 * inv) << 16);
        }
        *reinterpret_cast<uint2 *>(out + (((size_t)b * n_rows + grow) * H + head) * D + dq * DW + d) = pk;
    }
}
