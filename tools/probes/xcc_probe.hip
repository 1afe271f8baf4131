// xcc_probe.hip -- which XCD does workgroup b of a launch run on?  Records HW_REG_XCC_ID and the start / end s_memtime of every workgroup
// for GEMM-like launches (512 threads, 64 / 96 / 128 KiB of LDS, a spin of `work` cycles) so the XCD-aware tile remap of gemm_pipe_kernel.h
// (assumes workgroup b -> XCD b % 8) can be checked against the hardware, alone and with a second stream competing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

template <int LDS_KB>
__global__ void __launch_bounds__(512) probe(int* xcc, unsigned long long* t0, unsigned long long* t1, int work) {
    __shared__ char lds[LDS_KB * 1024];
    if (threadIdx.x == 0) lds[0] = 1;
    const unsigned long long s = __builtin_amdgcn_s_memtime();
    unsigned long long e = s;
    while ((long long)(e - s) < work) e = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        int id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = id & 0xf; t0[blockIdx.x] = s; t1[blockIdx.x] = e + lds[0];
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1024, work = argc > 2 ? atoi(argv[2]) : 20000, threads = argc > 3 ? atoi(argv[3]) : 512;
    const int two = argc > 4 ? atoi(argv[4]) : 0;
    int* xcc; unsigned long long *t0, *t1;
    hipMalloc(&xcc, n * 4 * 2); hipMalloc(&t0, n * 8 * 2); hipMalloc(&t1, n * 8 * 2);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    for (int rep = 0; rep < 2; ++rep) {
        probe<96><<<n, threads, 0, s1>>>(xcc, t0, t1, work);
        if (two) probe<64><<<n, 256, 0, s2>>>(xcc + n, t0 + n, t1 + n, work);
        hipDeviceSynchronize();
    }
    std::vector<int> h(n * 2); std::vector<unsigned long long> a(n * 2), b(n * 2);
    hipMemcpy(h.data(), xcc, n * 4 * 2, hipMemcpyDeviceToHost); hipMemcpy(a.data(), t0, n * 8 * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), t1, n * 8 * 2, hipMemcpyDeviceToHost);
    for (int k = 0; k <= two; ++k) {
        int match = 0; int hist[8] = {0};
        unsigned long long base = a[k * n]; for (int i = 0; i < n; ++i) if (a[k * n + i] < base) base = a[k * n + i];
        for (int i = 0; i < n; ++i) { match += h[k * n + i] == i % 8; hist[h[k * n + i] & 7]++; }
        printf("launch %d: %d workgroups, XCC_ID == b %% 8 for %d (%.1f %%); per-XCD counts:", k, n, match, 100.0 * match / n);
        for (int x = 0; x < 8; ++x) printf(" %d", hist[x]);
        printf("\n first 48 (b: xcc @start-cycles/100):");
        for (int i = 0; i < 48 && i < n; ++i) printf(" %d:%d@%llu", i, h[k * n + i], (a[k * n + i] - base) / 100);
        printf("\n later  (b: xcc @start):");
        for (int i = n / 2; i < n / 2 + 32 && i < n; ++i) printf(" %d:%d@%llu", i, h[k * n + i], (a[k * n + i] - base) / 100);
        printf("\n");
    }
    return 0;
}
