// Layout probe for v_mfma_f64_16x16x4_f64 on gfx950 (operands: one double per lane; result: 4 doubles per lane).
// Assumed: A[i][k] in lane i + 16k, B[k][j] in lane j + 16k.  Prints, for every lane and result register, the (i, j) it holds,
// and checks the k placement of both operands.   hipcc --offload-arch=gfx950 -O2 mfma_f64_probe.hip -o mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) double d4;
__global__ void probe(double *out, int K0) {
    const int l = threadIdx.x, i = l & 15, k = l >> 4;
    // D[i][j] = sum_k A[i][k] B[k][j];  A[i][k] = (k == K0) * (i + 1),  B[k][j] = (k == K0) * (j + 1) * 100
    const double a = (k == K0) ? (double)(i + 1) : 0.0;
    const double b = (k == K0) ? (double)((l & 15) + 1) * 100.0 : 0.0;
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
    double *d; hipMalloc(&d, 64 * 4 * 8);
    double h[256];
    int ok = 1;
    for (int K0 = 0; K0 < 4; ++K0) {
        probe<<<1, 64>>>(d, K0);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const long v = (long)h[l * 4 + r];
                const int j = (int)(v / 100 % 100 == 0 ? 0 : 0);
                (void)j;
                // v = (i+1) * (j+1) * 100 ; candidate: j = l % 16, i = (l / 16) + 4 * r
                const int jc = l & 15, ic = (l >> 4) + 4 * r;
                if (v != (long)(ic + 1) * (jc + 1) * 100) { ok = 0; if (K0 == 0) printf("lane %d reg %d: value %ld (candidate i=%d j=%d expects %ld)\n", l, r, v, ic, jc, (long)(ic + 1) * (jc + 1) * 100); }
            }
    }
    printf(ok ? "LAYOUT OK: D reg r of lane l = D[(l/16)+4r][l%%16]; A[i][k] lane i+16k; B[k][j] lane j+16k\n" : "LAYOUT MISMATCH\n");
    return ok ? 0 : 1;
}
