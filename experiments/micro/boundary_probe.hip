// What does a dependent kernel boundary cost on this part, and what makes it cost more?  Chains of trivial kernels on one stream,
// 256 workgroups each, timed over 2000 launches with one event pair around the whole chain (round 6: the per-launch timeline of a
// fit iteration shows 6 us in front of some launches and 0 in front of others -- profiles/r06_timelines_shard.txt).
//   hipcc --offload-arch=gfx950 -O3 -o boundary_probe boundary_probe.hip && ./boundary_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int LDS, int SCRATCH>
__global__ void k(float *out, int spin) {
    __shared__ float s[LDS / 4 > 0 ? LDS / 4 : 1];
    float acc = (float)threadIdx.x;
    float priv[SCRATCH > 0 ? SCRATCH : 1];
    if (SCRATCH > 0) { for (int i = 0; i < SCRATCH; ++i) priv[i] = acc + i; }
    for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 1.f;
    if (LDS > 0) { s[threadIdx.x % (LDS / 4)] = acc; __syncthreads(); acc += s[(threadIdx.x * 7) % (LDS / 4)]; }
    if (SCRATCH > 0) acc += priv[(int)acc & (SCRATCH - 1)];
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
// a kernel that leaves `bytes` of dirty lines behind
__global__ void kw(float4 *buf, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) buf[i] = float4{1.f, 2.f, 3.f, 4.f};
}

int main() {
    float *out; CHK(hipMalloc(&out, 1 << 20));
    float4 *big; const size_t big_bytes = 64u << 20; CHK(hipMalloc(&big, big_bytes));
    hipStream_t st; CHK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    std::vector<hipEvent_t> evs(8192);
    for (auto &e : evs) CHK(hipEventCreate(&e));
    const int N = 2000, G = 256;
    auto run = [&](const char *name, auto body) -> int {
        for (int w = 0; w < 50; ++w) body(w);
        CHK(hipStreamSynchronize(st));
        CHK(hipEventRecord(e0, st));
        for (int i = 0; i < N; ++i) body(i);
        CHK(hipEventRecord(e1, st));
        CHK(hipStreamSynchronize(st));
        float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-72s %7.2f us per step\n", name, 1e3 * ms / N);
        return 0;
    };
    const int spin = 200;
#define L(LDSB, SCR, TH) hipLaunchKernelGGL((k<LDSB, SCR>), dim3(G), dim3(TH), 0, st, out, spin)
    run("1 launch: lds 0, 256 threads", [&](int) { L(0, 0, 256); });
    run("2 launches: lds 0 | lds 0", [&](int) { L(0, 0, 256); L(0, 0, 256); });
    run("2 launches: lds 0 | lds 64K", [&](int) { L(0, 0, 256); L(65536, 0, 256); });
    run("2 launches: lds 64K | lds 64K", [&](int) { L(65536, 0, 256); L(65536, 0, 256); });
    run("2 launches: lds 8K | lds 128K", [&](int) { L(8192, 0, 256); L(131072, 0, 256); });
    run("2 launches: 256 threads | 1024 threads", [&](int) { L(0, 0, 256); L(0, 0, 1024); });
    run("2 launches: no scratch | scratch", [&](int) { L(0, 0, 256); L(0, 64, 256); });
    run("2 launches: scratch | scratch", [&](int) { L(0, 64, 256); L(0, 64, 256); });
    run("2 launches + 1 event record between", [&](int i) { L(0, 0, 256); hipEventRecord(evs[i % 8192], st); L(0, 0, 256); });
    run("2 launches + event record after each", [&](int i) { L(0, 0, 256); hipEventRecord(evs[(2 * i) % 8192], st); L(0, 0, 256); hipEventRecord(evs[(2 * i + 1) % 8192], st); });
    run("6 launches (one iteration's count), no events", [&](int) { for (int j = 0; j < 6; ++j) L(0, 0, 256); });
    run("6 launches, 4 event records among them", [&](int i) { hipEventRecord(evs[(4 * i) % 8192], st); L(0, 0, 256); hipEventRecord(evs[(4 * i + 1) % 8192], st); L(0, 0, 256);
                                                                hipEventRecord(evs[(4 * i + 2) % 8192], st); L(0, 0, 256); L(0, 0, 256); hipEventRecord(evs[(4 * i + 3) % 8192], st); L(0, 0, 256); L(0, 0, 256); });
    run("writer of 64 MB alone", [&](int) { hipLaunchKernelGGL(kw, dim3(1024), dim3(256), 0, st, big, big_bytes / 16); });
    run("writer of 64 MB | trivial", [&](int) { hipLaunchKernelGGL(kw, dim3(1024), dim3(256), 0, st, big, big_bytes / 16); L(0, 0, 256); });
    run("writer of 8 MB alone", [&](int) { hipLaunchKernelGGL(kw, dim3(1024), dim3(256), 0, st, big, (8u << 20) / 16); });
    run("writer of 8 MB | trivial", [&](int) { hipLaunchKernelGGL(kw, dim3(1024), dim3(256), 0, st, big, (8u << 20) / 16); L(0, 0, 256); });
    return 0;
}
