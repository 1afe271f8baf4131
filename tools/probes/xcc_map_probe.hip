// xcc_map_probe.hip -- which XCD does workgroup i of a launch run on?  (standalone: hipcc --offload-arch=gfx950 -O2 -o xcc_map_probe xcc_map_probe.hip)
// The GEMM's tile -> workgroup remap (gemm_pipe_kernel.h: "XCD-aware bijective remap") assumes workgroup i runs on XCD i % 8, so that an XCD's L2 sees a
// contiguous block of tiles.  That holds for ONE launch on an idle GPU; the step runs four lanes' launches at once.  Cases:
//   0: one stream;  1: four streams launching the same kernel concurrently;  2: a stream created with a CU mask (hipExtStreamCreateWithCUMask) of the
//   first 32 / 64 / 128 mask bits -- which XCDs / CUs do its workgroups land on, and in what order?
// Every workgroup records {XCC_ID, HW_ID} and spins ~20 us so that launches really overlap.  Prints one JSON line per case.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void probe(unsigned* out, long spin) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const long t0 = __builtin_amdgcn_s_memtime();
    while ((long)__builtin_amdgcn_s_memtime() - t0 < spin) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

static void report(const char* name, const std::vector<unsigned>& h, int nwg) {
    int match = 0, hist[16] = {0};
    for (int i = 0; i < nwg; ++i) { const int x = h[2 * i] & 15; ++hist[x]; match += (x == i % 8); }
    printf("{\"case\": \"%s\", \"workgroups\": %d, \"xcc_eq_id_mod_8\": %d, \"per_xcc\": [", name, nwg, match);
    for (int x = 0; x < 8; ++x) printf("%d%s", hist[x], x < 7 ? ", " : "");
    printf("], \"first_24_xcc\": [");
    for (int i = 0; i < 24 && i < nwg; ++i) printf("%u%s", h[2 * i] & 15, i < 23 ? ", " : "");
    printf("], \"first_8_hw_id\": [");
    for (int i = 0; i < 8 && i < nwg; ++i) printf("%u%s", h[2 * i + 1], i < 7 ? ", " : "");
    printf("]}\n");
}

int main() {
    const int nwg = 1024, nstream = 4;
    const long spin = 2000;                       // s_memtime ticks (100 MHz): 20 us
    unsigned* d[nstream];
    for (int s = 0; s < nstream; ++s) CHECK(hipMalloc(&d[s], nwg * 8));
    std::vector<unsigned> h(nwg * 2);
    hipStream_t st[nstream];
    for (int s = 0; s < nstream; ++s) CHECK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
    // case 0
    probe<<<nwg, 256, 0, st[0]>>>(d[0], spin);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), d[0], nwg * 8, hipMemcpyDeviceToHost));
    report("one stream", h, nwg);
    // case 1: three rounds on four streams, report the last round of every stream
    for (int r = 0; r < 3; ++r)
        for (int s = 0; s < nstream; ++s) probe<<<nwg, 256, 0, st[s]>>>(d[s], spin);
    CHECK(hipDeviceSynchronize());
    for (int s = 0; s < nstream; ++s) {
        CHECK(hipMemcpy(h.data(), d[s], nwg * 8, hipMemcpyDeviceToHost));
        char name[64]; snprintf(name, sizeof name, "four concurrent streams, stream %d", s);
        report(name, h, nwg);
    }
    // case 2: CU-masked streams
    for (int bits : {32, 64, 128}) {
        uint32_t mask[8] = {0};
        for (int b = 0; b < bits; ++b) mask[b / 32] |= 1u << (b % 32);
        hipStream_t ms;
        hipError_t e = hipExtStreamCreateWithCUMask(&ms, 8, mask);
        if (e != hipSuccess) { printf("{\"case\": \"cu mask %d bits\", \"error\": \"%s\"}\n", bits, hipGetErrorString(e)); continue; }
        probe<<<nwg, 256, 0, ms>>>(d[0], spin);
        CHECK(hipStreamSynchronize(ms));
        CHECK(hipMemcpy(h.data(), d[0], nwg * 8, hipMemcpyDeviceToHost));
        char name[64]; snprintf(name, sizeof name, "cu mask: first %d bits", bits);
        report(name, h, nwg);
        // distinct (xcc, cu) pairs
        int seen[16][64] = {{0}}, distinct = 0;
        for (int i = 0; i < nwg; ++i) { const int x = h[2 * i] & 15, cu = (h[2 * i + 1] >> 8) & 15, sh = (h[2 * i + 1] >> 12) & 1, se = (h[2 * i + 1] >> 13) & 7;
            const int k = (se * 2 + sh) * 16 + cu; if (k < 64 && !seen[x][k]) { seen[x][k] = 1; ++distinct; } }
        printf("{\"case\": \"cu mask: first %d bits\", \"distinct_xcc_se_sh_cu\": %d}\n", bits, distinct);
        CHECK(hipStreamDestroy(ms));
    }
    // case 3: stride-8 patterns (bit b enabled iff b % 8 in the set) -- the layout diffusion_pipe_amd/hip.py cu_mask_for_xcds assumes: do the workgroups stay on those XCDs?
    for (unsigned set : {0x03u, 0x0cu, 0x0fu, 0x01u}) {
        uint32_t mask[8] = {0};
        for (int b = 0; b < 256; ++b) if ((set >> (b % 8)) & 1) mask[b / 32] |= 1u << (b % 32);
        hipStream_t ms;
        hipError_t e = hipExtStreamCreateWithCUMask(&ms, 8, mask);
        if (e != hipSuccess) { printf("{\"case\": \"stride-8 mask 0x%02x\", \"error\": \"%s\"}\n", set, hipGetErrorString(e)); continue; }
        probe<<<nwg, 256, 0, ms>>>(d[0], spin);
        CHECK(hipStreamSynchronize(ms));
        CHECK(hipMemcpy(h.data(), d[0], nwg * 8, hipMemcpyDeviceToHost));
        char name[64]; snprintf(name, sizeof name, "stride-8 mask, XCD set 0x%02x", set);
        report(name, h, nwg);
        CHECK(hipStreamDestroy(ms));
    }
    return 0;
}
