// gemm_timeline.hip -- per-phase cycle breakdown of gemm_pipe_kernel (s_memtime stamps compiled in with DPIPE_TIMELINE): for a few SDXL
// shapes, HBM-cold operands, prints the workgroup-averaged cycles of: prologue issue, and per K-step: wait for the DMA of the step
// (s_waitcnt vmcnt), barrier, issue of the refill, fragment reads + MFMAs; then the epilogue.  100 MHz s_memtime ticks -> ns = ticks * 10.
#define DPIPE_TIMELINE 1
#include "../../diffusion_pipe_amd/csrc/gemm_pipe.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>

namespace dpipe {
void set_last_error(const char* msg) { fprintf(stderr, "error: %s\n", msg); }
int option(int, int dflt) { return dflt; }
int check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return (int)e; }
}

int main(int argc, char** argv) {
    struct Case { int ta, tb, M, N, K, tile, sk; };
    std::vector<Case> cases = {{0, 1, 1024, 1280, 1280, 64, 1}, {0, 1, 1024, 1280, 1280, 65, 1}, {0, 1, 1024, 10240, 1280, 128, 1}, {0, 1, 1024, 10240, 1280, 130, 1},
                               {0, 0, 1024, 1280, 1280, 64, 1}, {0, 0, 1024, 1280, 1280, 65, 1}, {1, 0, 1280, 1280, 1024, 65, 1}, {1, 0, 10240, 1280, 1024, 129, 1},
                               {1, 0, 10240, 1280, 1024, 130, 1}, {0, 0, 1024, 5120, 1280, 130, 1}, {0, 1, 1024, 10240, 1280, 257, 1}, {0, 1, 1024, 10240, 1280, 129, 1}, {0, 1, 8192, 8192, 8192, 257, 1}};
    const size_t arena_bytes = 2ul << 30;
    char* arena; hipMalloc(&arena, arena_bytes); hipMemset(arena, 0x11, arena_bytes);
    void* ws; hipMalloc(&ws, 4096 + 640 * 65536); hipMemset(ws, 0, 4096 + 640 * 65536);
    unsigned long long* tl; hipMalloc(&tl, 8 * 64 * 8192);
    size_t off = 0;
    auto take = [&](size_t n) { n = (n + 255) & ~255ul; if (off + n > arena_bytes) off = 0; char* p = arena + off; off += n; return p; };
    for (auto c : cases) {
        for (int rep = 0; rep < 3; ++rep) {
            GemmParams p{};
            const bool a_mc = c.ta, b_mc = !c.tb;
            p.M = c.M; p.N = c.N; p.K = c.K;
            p.lda = a_mc ? c.M : c.K; p.ldb = b_mc ? c.N : c.K; p.ldc = c.N;
            p.A = take((size_t)c.M * c.K * 2); p.B = take((size_t)c.N * c.K * 2); p.C = take((size_t)c.M * c.N * 2);
            p.batch_inner = 1; p.alpha = 1.f; p.timeline = tl;
            hipMemset(tl, 0, 8 * 64 * 8192);
            int rc = 0;
            if (!gemm_pipe_try(p, c.ta, c.tb, 1, ws, 4096 + 640 * 65536, c.sk, c.tile, 0, &rc) || rc) { printf("launch failed\n"); continue; }
            hipDeviceSynchronize();
            if (rep < 2) continue;
            const int nwg = p.tiles_m * p.tiles_n * p.splitk;
            std::vector<unsigned long long> h((size_t)nwg * 64);
            hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost);
            const int nk = p.ksteps_per_split < 15 ? p.ksteps_per_split : 15;
            double pro = 0, epi = 0, tot = 0, wait[16] = {0}, bar[16] = {0}, iss[16] = {0}, comp[16] = {0};
            int cnt = 0;
            for (int w = 0; w < nwg; ++w) {
                const unsigned long long* s = &h[(size_t)w * 64];
                if (!s[0] || !s[3]) continue;
                ++cnt; pro += s[1] - s[0]; epi += s[3] - s[2]; tot += s[3] - s[0];
                unsigned long long prev = s[1];
                for (int it = 0; it < nk; ++it) {
                    wait[it] += s[4 + 4 * it] - prev; bar[it] += s[5 + 4 * it] - s[4 + 4 * it]; iss[it] += s[6 + 4 * it] - s[5 + 4 * it]; comp[it] += s[7 + 4 * it] - s[6 + 4 * it];
                    prev = s[7 + 4 * it];
                }
            }
            printf("ta=%d tb=%d %dx%dx%d tile=%d splitk=%d: %d WGs, ksteps/WG=%d | total %.0f ns  prologue %.0f  epilogue %.0f\n", c.ta, c.tb, c.M, c.N, c.K, c.tile, p.splitk, nwg,
                   p.ksteps_per_split, tot / cnt * 10, pro / cnt * 10, epi / cnt * 10);
            printf("   k-step:   "); for (int it = 0; it < nk; ++it) printf("%6d", it); printf("\n   wait ns:  "); for (int it = 0; it < nk; ++it) printf("%6.0f", wait[it] / cnt * 10);
            printf("\n   barrier:  "); for (int it = 0; it < nk; ++it) printf("%6.0f", bar[it] / cnt * 10);
            printf("\n   issue:    "); for (int it = 0; it < nk; ++it) printf("%6.0f", iss[it] / cnt * 10);
            printf("\n   compute:  "); for (int it = 0; it < nk; ++it) printf("%6.0f", comp[it] / cnt * 10);
            printf("\n");
        }
    }
    return 0;
}
