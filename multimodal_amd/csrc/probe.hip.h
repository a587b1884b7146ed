// Hardware self-checks for the layout facts the fused kernels rely on
// (gfx950): MFMA 32x32x16 f16 operand/accumulator maps, ds_read_b64_tr_b16
// addressing, "accumulator tile as the next MFMA's B operand" k-permutation,
// and global_load_lds lane-linear destination.  Exact small-integer data.
#pragma once
#include <vector>

#include "mfma.hip.h"

namespace klnmf {

// a padded 64-column image row for the transposed-read probe (144 B = 36 dwords)
constexpr int kHRow = 64 + 8, kHRowB = kHRow * 2;
// two ds_read_b64_tr_b16: each gives this lane one column of a 4-row x 16-col block
__device__ __forceinline__ opx8 tr_pair(const KL_LDS unsigned char *p0, const KL_LDS unsigned char *p1) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((KL_LDS s16x4 *)p0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((KL_LDS s16x4 *)p1);
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(opx8, v);
}
// `rounds` x 8 KiB from global to LDS with global_load_lds_dwordx4 (LDS image == global image; destination is the
// wave-uniform base + lane * 16)
__device__ __forceinline__ void glds_copy(const unsigned char *gsrc, KL_LDS unsigned char *ldst, int rounds, int tid) {
    const int wave_base = (tid & ~63) * 16;
    for (int r = 0; r < rounds; ++r)
        __builtin_amdgcn_global_load_lds((const KL_GLB void *)(gsrc + r * kGldsRound + tid * 16),
                                         (KL_LDS void *)(ldst + r * kGldsRound + wave_base), 16, 0, 0);
}

// P1: D = A[32x16] . B[16x32] through the documented fragment maps.
KL_GLOBAL void k_probe_mfma(const float *A, const float *B, float *D) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    opx8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = (opnd_t)A[r * 16 + 8 * h + j];
        b[j] = (opnd_t)B[(8 * h + j) * 32 + r];
    }
    f32x16 d;
#pragma unroll
    for (int e = 0; e < 16; ++e) d[e] = 0.f;
    d = KL_MFMA_BUILTIN(a, b, d, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        D[row * 32 + r] = d[e];
    }
}

// P2: transposed LDS read with the row-pass addressing (image rows = k index).
KL_GLOBAL void k_probe_tr(short *out, int s, int u) {
    __shared__ __attribute__((aligned(16))) unsigned char img[64 * kHRowB];
    const int l = threadIdx.x;
    for (int e = l; e < 64 * 64; e += 64) {                      // logical (row, col) -> permuted image
        const int row = e / 64, col = e % 64;
        ((short *)img)[row * kHRow + h_col_perm(col)] = (short)(row * 128 + col);
    }
    __syncthreads();
    const int h = l >> 5, i16 = l & 15, tq = i16 >> 2, tp = i16 & 3, half = (l >> 4) & 1;
    const int off = (8 * h + tq) * kHRowB + h_col_perm(16 * half + 4 * tp) * 2;
    const KL_LDS unsigned char *p = (const KL_LDS unsigned char *)img + off + (16 * s) * kHRowB + (32 * u) * 2;
    opx8 v = tr_pair(p, p + 4 * kHRowB);
    s16x8 w = __builtin_bit_cast(s16x8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) out[l * 8 + j] = w[j];
    // MFMA-2 row fragment of the same image: lane (r, h), k-step s: element j must be logical
    // column 32u + 16s + 8(j>>2) + 4h + (j&3) of row r
    const int r = l & 31;
    const KL_LDS unsigned char *p2 = (const KL_LDS unsigned char *)img + r * kHRowB + 16 * h + (32 * u) * 2 + 32 * s;
    s16x8 w2 = __builtin_bit_cast(s16x8, *(const KL_LDS opx8 *)p2);
#pragma unroll
    for (int j = 0; j < 8; ++j) out[512 + l * 8 + j] = w2[j];
}

// P3: X = A0.B0 (32x32, K=16) kept in the accumulator, converted to bf16 and fed
// as the B operand of Y = A2[32x32] . X with the permuted k order.
KL_GLOBAL void k_probe_chain(const float *A0, const float *B0, const float *A2, float *Y) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    opx8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = (opnd_t)A0[r * 16 + 8 * h + j];
        b[j] = (opnd_t)B0[(8 * h + j) * 32 + r];
    }
    f32x16 x;
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = 0.f;
    x = KL_MFMA_BUILTIN(a, b, x, 0, 0, 0);
    float q[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) q[e] = x[e];
    const opx8 b0 = pack8(q), b1 = pack8(q + 8);
    f32x16 y;
#pragma unroll
    for (int e = 0; e < 16; ++e) y[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        opx8 a2;
#pragma unroll
        for (int j = 0; j < 8; ++j) a2[j] = (opnd_t)A2[r * 32 + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)];
        y = KL_MFMA_BUILTIN(a2, s == 0 ? b0 : b1, y, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        Y[row * 32 + r] = y[e];
    }
}

// P4: global_load_lds round trip (2 rounds of 8 KiB, 512 threads).
KL_GLOBAL __launch_bounds__(kThreads) void k_probe_glds(const unsigned char *src, unsigned char *dst) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    KL_LDS unsigned char *smem = (KL_LDS unsigned char *)smem_raw;
    glds_copy(src, smem, 2, threadIdx.x);
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * kGldsRound / 16; e += kThreads)
        ((u32x4 *)dst)[e] = ((const KL_LDS u32x4 *)smem)[e];
}

inline int run_probes() {
    int failed = 0;
    auto chk = [](hipError_t e) {
        if (e != hipSuccess) throw std::runtime_error(std::string("probe: ") + hipGetErrorString(e));
    };
    // deterministic small integers
    std::vector<float> A(32 * 16), B(16 * 32), A2(32 * 32);
    for (int i = 0; i < 32 * 16; ++i) A[i] = (float)((i * 7 + 3) % 5 - 2);
    for (int i = 0; i < 16 * 32; ++i) B[i] = (float)((i * 11 + 1) % 7 - 3);
    for (int i = 0; i < 32 * 32; ++i) A2[i] = (float)((i * 13 + 5) % 5 - 2);
    std::vector<float> X(32 * 32, 0.f), Yref(32 * 32, 0.f);
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float s = 0;
            for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + j];
            X[i * 32 + j] = s;
        }
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float s = 0;
            for (int k = 0; k < 32; ++k) s += A2[i * 32 + k] * X[k * 32 + j];
            Yref[i * 32 + j] = s;
        }
    float *dA, *dB, *dA2, *dD, *dY;
    chk(hipMalloc((void **)&dA, A.size() * 4));
    chk(hipMalloc((void **)&dB, B.size() * 4));
    chk(hipMalloc((void **)&dA2, A2.size() * 4));
    chk(hipMalloc((void **)&dD, 32 * 32 * 4));
    chk(hipMalloc((void **)&dY, 32 * 32 * 4));
    chk(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    chk(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    chk(hipMemcpy(dA2, A2.data(), A2.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> D(32 * 32), Y(32 * 32);
    hipLaunchKernelGGL(k_probe_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    chk(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < 32 * 32; ++i)
        if (D[i] != X[i]) { failed |= 1; break; }
    hipLaunchKernelGGL(k_probe_chain, dim3(1), dim3(64), 0, 0, dA, dB, dA2, dY);
    chk(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < 32 * 32; ++i)
        if (Y[i] != Yref[i]) { failed |= 4; break; }

    short *dT;
    chk(hipMalloc((void **)&dT, 2 * 64 * 8 * 2));
    std::vector<short> T(2 * 64 * 8);
    for (int s = 0; s < 2; ++s)
        for (int u = 0; u < 2; ++u) {
            hipLaunchKernelGGL(k_probe_tr, dim3(1), dim3(64), 0, 0, dT, s, u);
            chk(hipMemcpy(T.data(), dT, T.size() * 2, hipMemcpyDeviceToHost));
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int row = 16 * s + 8 * (l >> 5) + j, col = 32 * u + (l & 31);
                    if (T[l * 8 + j] != (short)(row * 128 + col)) failed |= 2;
                    const int row2 = l & 31, col2 = 32 * u + 16 * s + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
                    if (T[512 + l * 8 + j] != (short)(row2 * 128 + col2)) failed |= 16;
                }
        }

    unsigned char *dS, *dO;
    const int nb = 2 * kGldsRound;
    chk(hipMalloc((void **)&dS, nb));
    chk(hipMalloc((void **)&dO, nb));
    std::vector<unsigned char> S(nb), O(nb);
    for (int i = 0; i < nb; ++i) S[i] = (unsigned char)((i * 31 + (i >> 8)) & 0xff);
    chk(hipMemcpy(dS, S.data(), nb, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe_glds, dim3(1), dim3(kThreads), nb, 0, dS, dO);
    chk(hipMemcpy(O.data(), dO, nb, hipMemcpyDeviceToHost));
    if (std::memcmp(S.data(), O.data(), nb) != 0) failed |= 8;

    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dA2); (void)hipFree(dD); (void)hipFree(dY);
    (void)hipFree(dT); (void)hipFree(dS); (void)hipFree(dO);
    return failed;
}

}  // namespace klnmf
