// l2_feed_probe.hip -- how many bytes per second can all 256 CUs pull through their vector-memory path when the data sits in L2 / Infinity Cache / HBM?
// The GEMM's 128^2 tile needs 1 byte from L2 per 64 FLOPs (32 KiB of operand panels per 2.1 MFLOP K-step), the 64^2 tile 1 per 32, the 256^2 tile 1 per 128:
// the aggregate L2 -> CU feed rate times those intensities is a roofline of its own next to the MFMA peak.
//
// Every workgroup (256 threads) streams a window of `window` bytes `iters` times; windows of workgroups on the same XCD coincide when `shared` = 1 (operand panels
// shared by the tiles of an XCD) and are disjoint otherwise.  Modes: 0 = global_load_dwordx4 into VGPRs (contiguous 1 KiB per wave instruction), 1 = the GEMM's
// LDS-DMA (buffer_load_dwordx4 ... lds), contiguous; 2 = LDS-DMA with the GEMM's K-contiguous panel pattern (8 rows x 128 B per wave instruction, row pitch 2 560 B).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/l2_feed_probe.hip -o tools/probes/l2_feed_probe && tools/probes/l2_feed_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;

template <int MODE>
__global__ void __launch_bounds__(256) feed_kernel(const char* base, long window, long wg_stride, int iters, int inflight, unsigned* sink, int rows_per_instr = 8, int pitch = 2560) {
    __shared__ __attribute__((aligned(1024))) char lds[64 * 1024];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // workgroups b, b + 8, ... run on one XCD: give them the same window when wg_stride is applied per XCD
    const char* win = base + (wg_stride < 0 ? (long)blockIdx.x * (-wg_stride) : (long)(blockIdx.x % 8) * wg_stride);
    unsigned acc = 0;
    if constexpr (MODE == 0) {
        const uint4* p = reinterpret_cast<const uint4*>(win);
        const long n16 = window / 16;
        for (int it = 0; it < iters; ++it) {
            for (long i = threadIdx.x; i < n16; i += 256 * 8) {
                uint4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (i + u * 256 < n16) ? p[i + u * 256] : make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].w;
            }
        }
    } else {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(win), (short)0, (int)window, 0x00020000);
        // one "piece" = one wave instruction = 1 KiB; a workgroup issues `inflight` KiB (<= 64) before it waits, like a ring of K-steps
        const long pieces = window / 1024;
        // R rows of 1024 / R bytes at row pitch P: piece q covers rows R (q / cpr) .. + R at byte column (1024 / R) (q % cpr) of a [rows][P bytes] panel; the wave walks
        // its pieces q = wid, wid + 4, ... with an incremental (row block, column) counter -- no division in the loop
        const int bpr = 1024 / rows_per_instr, cpr = pitch / bpr, lpr = bpr / 16;
        const unsigned lane_off = (unsigned)((lane / lpr) * pitch + (lane % lpr) * 16);
        const unsigned rblk_step = (unsigned)(rows_per_instr * pitch);
        for (int it = 0; it < iters; ++it) {
            int kc = wid % cpr; unsigned rbase = (unsigned)(wid / cpr) * rblk_step;
            for (long q0 = 0; q0 < pieces; q0 += inflight) {
                for (int j = wid; j < inflight && q0 + j < pieces; j += 4) {
                    unsigned off;
                    if constexpr (MODE == 1) off = (unsigned)((q0 + j) * 1024 + lane * 16);
                    else {
                        off = rbase + (unsigned)(kc * bpr) + lane_off;
                        kc += 4; while (kc >= cpr) { kc -= cpr; rbase += rblk_step; }
                    }
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(lds + (j & 63) * 1024), 16, off, 0, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            acc ^= reinterpret_cast<unsigned*>(lds)[threadIdx.x];
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const size_t arena_bytes = 3ul << 30;
    char* arena; hipMalloc(&arena, arena_bytes); hipMemset(arena, 0x5a, arena_bytes);
    unsigned* sink; hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Case { const char* name; long window; long wg_stride; int wgs; };
    // wg_stride >= 0: the 8 XCDs read 8 windows (every workgroup of an XCD the same one); < 0: every workgroup its own window (|stride| apart)
    const Case cases[] = {
        {"L2-resident, shared per XCD: 1 MiB window", 1l << 20, 1l << 20, 1024},
        {"L2-resident, shared per XCD: 2.5 MiB window", 2560l << 10, 2560l << 10, 1024},
        {"Infinity-Cache-resident: 16 MiB window per XCD", 16l << 20, 16l << 20, 1024},
        {"private 640 KiB windows (1024 workgroups, 640 MiB: HBM stream)", 640l << 10, -(640l << 10), 1024},
        {"private 2.5 MiB windows (1024 workgroups, 2.5 GiB: HBM stream)", 2560l << 10, -(2560l << 10), 1024},
    };
    for (const Case& c : cases) {
        for (int mode = 0; mode < 3; ++mode) {
            for (int inflight : {16, 32, 64}) {
                if (mode == 0 && inflight != 16) continue;
                const long bytes_once = c.window * c.wgs;
                const int iters = (int)(((c.wg_stride < 0 ? 2l : 8l) << 30) / bytes_once) + 1;     // 8 GiB of traffic per measurement (2 for the HBM streams)
                auto launch = [&](int its) {
                    if (mode == 0) feed_kernel<0><<<c.wgs, 256>>>(arena, c.window, c.wg_stride, its, inflight, sink);
                    else if (mode == 1) feed_kernel<1><<<c.wgs, 256>>>(arena, c.window, c.wg_stride, its, inflight, sink);
                    else feed_kernel<2><<<c.wgs, 256>>>(arena, c.window, c.wg_stride, its, inflight, sink);
                };
                launch(1); hipDeviceSynchronize();
                hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                const double tbs = (double)bytes_once * iters / (ms * 1e-3) / 1e12;
                printf("{\"case\": \"%s\", \"mode\": \"%s\", \"kib_in_flight_per_wg\": %d, \"workgroups\": %d, \"TB_per_s\": %.2f, \"ms\": %.3f}\n", c.name,
                       mode == 0 ? "global_load_dwordx4 -> VGPR" : mode == 1 ? "LDS-DMA contiguous" : "LDS-DMA 8 rows x 128 B, pitch 2560 B", mode == 0 ? 32 : inflight, c.wgs, tbs, ms);
                fflush(stdout);
            }
        }
    }
    // which property of the row-gather pattern costs the rate?  L2-resident 2.5 MiB window shared per XCD, 64 KiB in flight per workgroup
    struct Pat { int rows, pitch; };
    const Pat pats[] = {{8, 128}, {8, 2560}, {8, 2688}, {8, 4096}, {8, 10240}, {8, 20480}, {4, 256}, {4, 2560}, {4, 10240}, {2, 512}, {2, 2560}, {16, 64}, {16, 2560}, {1, 1024}};
    for (const Pat& pt : pats) {
        const long window = 2560l << 10;
        const long bytes_once = window * 1024;
        const int iters = 4;
        feed_kernel<2><<<1024, 256>>>(arena, window, window, 1, 64, sink, pt.rows, pt.pitch); hipDeviceSynchronize();
        hipEventRecord(e0); feed_kernel<2><<<1024, 256>>>(arena, window, window, iters, 64, sink, pt.rows, pt.pitch); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("{\"case\": \"pattern sweep, L2-resident 2.5 MiB window per XCD, LDS-DMA, 64 KiB in flight\", \"rows_per_wave_instruction\": %d, \"bytes_per_row\": %d, \"row_pitch\": %d, \"TB_per_s\": %.2f}\n",
               pt.rows, 1024 / pt.rows, pt.pitch, (double)bytes_once * iters / (ms * 1e-3) / 1e12);
        fflush(stdout);
    }
    return 0;
}
