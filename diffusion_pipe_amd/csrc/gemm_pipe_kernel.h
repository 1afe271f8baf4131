// gemm_pipe_kernel.h -- the bf16 hot-path GEMM of the training step: LDS-DMA pipelined MFMA kernel with in-launch split-K
// (shared by gemm_pipe.hip = plain GEMM dispatch and conv_pipe.hip = implicit-GEMM convolution: the CONV template modes).
//
//   C[z] = act(alpha * op(A[z]) . op(B[z]) + bias) (+ C[z])        (same contract as dpipe_gemm, bf16 operands)
//
// Why a second kernel: at micro-batch 1 the SDXL / DiT linears are small (M = 77 .. 4096 tokens, N, K = 640 .. 10240):
// a 128 x 128 tiling yields 10 .. 100 workgroups for 256 CUs and every workgroup walks K serially, so the generic
// register-staged kernel (gemm.hip) is bound by one global-load latency per K-step, not by MFMA or HBM.  This kernel
//   * streams operand tiles global -> LDS with `buffer_load_dwordx4 ... lds` (no staging VGPRs, no ds_write pass) into
//     a ring of STAGES x 32 KiB buffers, keeps STAGES-1 K-steps in flight across ONE raw s_barrier per K-step and
//     drains them with counted `s_waitcnt vmcnt(N)` (never 0 in steady state);
//   * lays the LDS images out bank-conflict-free by permuting the per-lane SOURCE address (the DMA destination is
//     lane-linear): K-contiguous operands as [128 rows][8 x 16 B] with chunk ^= (row >> 1) & 7 for ds_read_b128,
//     MN-contiguous operands (dgrad's W, wgrad's dy and x) as [64 k-rows][4 x 64 B] with granule ^= krow & 3 for
//     ds_read_b64_tr_b16 -- so no transposed copy of any operand is ever materialised;
//   * uses the buffer descriptor's bounds check for ragged tiles (out-of-range rows read as zero);
//   * splits K over up to 16 workgroups per tile when the tile count alone cannot fill the chip: each slice stores an
//     fp32 slab in MFMA-native order (coalesced 16-B stores), publishes it with an agent-scope release + ticket counter,
//     and the last arriver acquires, sums the slabs in slice order (deterministic) and runs the epilogue;
//   * computes C^T tiles (operands swapped in the MFMA) so each lane owns 4 consecutive N elements of one row:
//     8-byte bf16 / 16-byte fp32 stores and vector bias loads instead of 2-byte scatter.
// Workgroup = 256 threads = 4 waves (2 x 2), 64 x 64 per wave, v_mfma_f32_32x32x16_bf16, fp32 accumulate.
#pragma once
#include "gemm_internal.h"
#include "lds_dma_tiles.h"

using namespace dpipe;

namespace dpipe_pipe {   // named: a kernel template argument may not have internal linkage (its host stub would be dropped)

constexpr int BK = 64;
constexpr int COUNTER_BYTES = 4096;
// Split-K slab publish: 1 = write-through (sc1) slab stores + ticket, no agent-scope release fence (the shipped form); 0 = plain stores + release
// fence (round 2's form, kept for A/B: build a second library with -DDPIPE_SLAB_WT=0 and load it through DPIPE_HIP_LIB, tools/README.md)
#ifndef DPIPE_SLAB_WT
#define DPIPE_SLAB_WT 1
#endif

// Cycle stamps for tools/probes/gemm_timeline.hip (compiled only there): wave 0 of every workgroup writes s_memtime at the phase
// boundaries of the kernel into timeline[blockIdx.x * 64 + slot].
#ifdef DPIPE_TIMELINE
#define TL_STAMP(slot) do { if (threadIdx.x == 0 && blockIdx.y == 0 && (slot) < 64) \
    reinterpret_cast<unsigned long long*>(p.timeline)[(long)blockIdx.x * 64 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TL_STAMP(slot) do { } while (0)
#endif

using namespace dpipe_tiles;   // LDS image formats of the DMA'd operand tiles (lds_dma_tiles.h)

// Tile geometry.  T64: 64 x 64 tile, 4 waves (2 x 2, 32 x 32 each), 4-deep ring of 16 KiB stages -> 2 workgroups per CU;
// the small-problem configuration (4 x the workgroups of T128: aggregate L1/L2 bandwidth of more CUs is what bounds a
// GEMM whose whole operand set is a few MB).  T128: 128 x 128 tile, 8 waves (2 x 4, 64 x 32 each) = 2 waves per SIMD so
// one wave's DMA issue (60..180 cycles per 1 KiB piece, MI355X_MICROARCH.md) hides under the other's MFMAs; 3 x 32 KiB.
template <int BM_, int BN_, int WM_, int WN_, int STAGES_, int BK_ = 64, int VS_ = 0> struct Tile {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, STAGES = STAGES_, BK = BK_;   // BK: k extent of a ring stage (64, or 32 for the 4-deep 256^2 ring)
    static constexpr int VS = VS_;                                                             // register stages in front of a 2-deep LDS ring (0 = the LDS-DMA ring)
    static constexpr int NW = WM * WN, NT = NW * 64;
    static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;       // 32x32 MFMA tiles per wave
    static constexpr int IMG_A = BM * BK * 2, IMG_B = BN * BK * 2, STAGE_BYTES = IMG_A + IMG_B;
    static constexpr int PA = IMG_A / 1024 / NW, PB = IMG_B / 1024 / NW;   // 1 KiB DMA pieces per wave per operand
    static constexpr int NLOAD = PA + PB;
    static constexpr int SLAB_F4 = BM * BN / 4 + BM / 4;               // accumulators + one row of fused column sums
    static constexpr int ACC_F4 = TM * TN * 4;                        // float4 per thread in a slab
    static_assert(IMG_A % (1024 * NW) == 0 && IMG_B % (1024 * NW) == 0, "pieces must split evenly over the waves");
};
using T64 = Tile<64, 64, 2, 2, 4>;
using T128 = Tile<128, 128, 2, 4, 3>;
using T128R2 = Tile<128, 128, 2, 4, 2>;     // 2-deep ring, 64 KiB -> 2 workgroups per CU: one tile's epilogue / prologue overlaps the other's K-loop.
                                            // Measured (tools/kernel_timing.py cold): 1.3-1.7x over T128 once there are >= 256 tiles and an
                                            // MN-contiguous operand (dgrad / wgrad), 1.2x at 8192^3; slower with few tiles (no second workgroup).
// Tiles built, measured and removed again (the measurements stay in profiles/; HISTORY.md section 4 has the numbers): T128S5 (5-deep ring = all of the LDS: whole-kernel
// time unchanged, 52.9 vs 51.5 us), T64 on 6- / 8-deep rings (round 4: list and step unchanged), T64S3 (3-deep 48 KiB ring: never dispatched), T128N64 (skinny M <= 128 with
// deep split-K: 56 vs 39 ms per step over those launches), T256 (256 x 128: T128R2 beats it), FOUR-wave tiles of every shape (15 .. 55 % slower: nothing covers a lone
// wave's waits; round 5's 4-wave 256^2 register-staged tile: 2 - 4 x slower), T128Q3 / Q4 (occupancy-style 4-wave 128^2 on half K-steps: -11 % on one shape, -2 % on
// the step), the 8-wave 128^2 tile on half-K-step rings (within +-5 %, behind in the step).
using T256K = Tile<256, 256, 2, 4, 4, 32>;  // the 256 x 256 tile on a 4-deep ring of HALF K-steps (32 k, 32 KiB each; the same 128 KiB of LDS): three half steps in flight instead of
                                            // one whole step -- the 2-deep ring parks every wave ~1 100 cycles per K-step at vmcnt (timeline probe), its refill can only be issued
                                            // once the whole previous step has been consumed
using T256S = Tile<256, 256, 2, 4, 2>;      // 256 x 256, 8 waves of 128 x 64 (128 accumulator VGPRs), 2 x 64 KiB: twice the MFMA work per DMA'd
                                            // byte of the 128^2 tile -- the large-GEMM configuration (DiT-sized linears: Flux / Wan / HunyuanVideo)

// ---- REGISTER-STAGED tile (round 5; VS > 0): global -> VGPR -> LDS instead of the LDS-DMA ring.  What four rounds of measurements asked for: the DMA
// ring can only hold as many bytes in flight as the LDS has room to land (2 x 64 KiB per CU = ~15 B / clk / CU at the ~4 500-cycle loaded latency of a piece), the
// register file is not full.  A K-step is fetched with `buffer_load_dwordx4` into one of VS register sets (NLOAD x 4 VGPRs each) VS + 1 K-steps ahead of its use and
// written to the free slot of a 2-deep LDS ring one K-step ahead (`ds_write_b128`, lane-linear = the DMA's own destination pattern, so the images, the source swizzles
// and every fragment read are unchanged): VS K-steps are in flight per workgroup whatever the LDS size.  The price is the ds_write pass (13 LDS cycles per 1 KiB piece).
// Measured (profiles/r5b_gemm_ledger_register_staged_tiles.jsonl, the step's >= 9 GFLOP descriptors, HBM-cold, us per launch: T128 / T128R2 / this tile / hipBLASLt):
//   [1024, 10240] <- 1280 NT 60.3 / 58.9 / 49.6 / 35.6      [4096, 5120] <- 640 NT 68.3 / 57.7 / 52.7 / 45.5      [1024, 3840] <- 1280 NT 19.6 / 26.6 / 18.9 / 18.6
//   [4096, 2560] <- 640 NN 35.5 / 32.4 / 27.8 / 26.5         [4096, 640] <- 5120 NN 61.3 / 80.1 / 50.3 / 49.7      [4096, 640] <- 1920 NN 26.2 / 33.6 / 22.5 / 28.3
// i.e. -3 .. -19 % wherever A is K-contiguous (forward / dgrad) and the tile runs unsplit; the wgrad layouts (A MN-contiguous: 146 - 150 VGPRs, one workgroup per CU) lose
// 5 - 15 % and stay on the DMA ring.  Also built and measured in the same pass, removed again: FOUR register sets (156 - 202 VGPRs = one workgroup per CU: within +-3 % of
// two sets except [1024, 10240] <- 1280, 51.8) and the 256^2 tile as FOUR waves of 128 x 128 (hipBLASLt's shape for these problems: 256 accumulator registers, one wave per
// SIMD), register-staged and on the DMA ring: 2 - 4 x SLOWER than the 8-wave tiles (118.8 / 114.8 us on [1024, 10240] <- 1280) -- with one wave per SIMD nothing hides the
// compiler's read-then-wait fragment schedule, the same finding as round 3's 4-wave tiles; that shape needs a hand-scheduled instruction stream, not this template.
using T128V = Tile<128, 128, 2, 4, 2, 64, 2>;     // the 8-wave 128^2 tile, 64 KiB of LDS, 2 K-steps (64 KiB) in flight in registers per workgroup, <= 128 VGPRs (two workgroups
                                                  // per CU); tile code 132, tile_hint 12000 + S

// sum of the 8 bf16 of an MFMA operand fragment (fp32)
__device__ __forceinline__ float frag_sum(bf16x8_t f) {
    const uint4 u = __builtin_bit_cast(uint4, f);
    return (__uint_as_float(u.x << 16) + __uint_as_float(u.x & 0xffff0000u)) + (__uint_as_float(u.y << 16) + __uint_as_float(u.y & 0xffff0000u)) +
           (__uint_as_float(u.z << 16) + __uint_as_float(u.z & 0xffff0000u)) + (__uint_as_float(u.w << 16) + __uint_as_float(u.w & 0xffff0000u));
}

// counted wait for this wave's LDS-DMA: `ahead` later K-steps (NLOAD DMA instructions each) may stay in flight
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int NLOAD, int MAXAHEAD> __device__ __forceinline__ void wait_dma_ahead(int ahead) {
    static_assert(NLOAD == 4 || NLOAD == 6 || NLOAD == 8 || NLOAD == 12 || NLOAD == 16, "DMA pieces per wave per K-step");
    static_assert(MAXAHEAD >= 1 && MAXAHEAD <= 7 && NLOAD * MAXAHEAD <= 63, "look-ahead of the ring vs the vmcnt field");
    switch (ahead) {       // (ahead <= MAXAHEAD = STAGES - 2 by construction; the unreachable cases fold away)
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<NLOAD>(); break;
    case 2: if constexpr (MAXAHEAD >= 2) { wait_vmcnt<NLOAD * 2>(); break; }
    case 3: if constexpr (MAXAHEAD >= 3) { wait_vmcnt<NLOAD * 3>(); break; }
    case 4: if constexpr (MAXAHEAD >= 4) { wait_vmcnt<NLOAD * 4>(); break; }
    case 5: if constexpr (MAXAHEAD >= 5) { wait_vmcnt<NLOAD * 5>(); break; }
    case 6: if constexpr (MAXAHEAD >= 6) { wait_vmcnt<NLOAD * 6>(); break; }
    default: wait_vmcnt<NLOAD * MAXAHEAD>(); break;
    }
}

// CONV = 0: plain GEMM.  CONV = 1: implicit-GEMM convolution, the A operand's ROWS are gathered pixels of an NHWC tensor (forward: rows =
// output pixels, k = (tap, input channel); dgrad: rows = input pixels, k = (tap, output channel), B = the weight read MN-contiguous) --
// a row that falls into the zero padding gets a DMA source offset beyond the buffer extent, which the buffer bounds check turns into
// zeros, so no im2col matrix and no padded copy ever exists.  CONV = 2: wgrad, one GEMM per tap (grid.y): the B operand's K-ROWS are
// the gathered pixels (k = output pixel, n = input channel), A = dy read MN-contiguous.
// The body of one workgroup: `orig` = its index among the problem's tiles_m * tiles_n * splitk workgroups (the launch's blockIdx.x, or the workgroup's index
// inside its problem's share of a GROUPED launch), `bz` = batch index / CONV 2 tap (blockIdx.y), `lds` = the launch's one LDS array (STAGES * STAGE_BYTES, 1 KiB
// aligned).  Always inlined: the array's address space reaches the ds_read / DMA instructions through the inliner.
template <int BM_, int BN_, int WM_, int WN_, int STAGES_, bool A_MC, bool B_MC, int CONV = 0, int BKT = 64, int VS = 0>
__device__ __forceinline__ void gemm_pipe_body(const GemmParams& p, char* const lds, const int orig, const int bz) {
    using TL = Tile<BM_, BN_, WM_, WN_, STAGES_, BKT, VS>;
    static_assert(VS == 0 || (STAGES_ == 2 && CONV == 0 && BKT == 64), "register-staged tile: plain GEMM on a 2-deep LDS ring of whole K-steps (the convolution form lost: conv_pipe.hip)");
    constexpr int BM = TL::BM, BN = TL::BN, STAGES = TL::STAGES, TM = TL::TM, TN = TL::TN, NLOAD = TL::NLOAD;
    constexpr int BK = BKT, KS = BKT / 16;          // (shadows the namespace default) k extent of a stage, 16-wide k-slices per stage
    static_assert(CONV == 0 || BKT == 64, "the convolution gathers are written for 64-channel K-steps");
    static_assert(STAGES >= 2 && STAGES <= 8, "ring depth");
    TL_STAMP(0);

    // XCD-aware bijective remap (consecutive ids round-robin over the 8 XCDs): each XCD owns a contiguous run of
    // (tile, slice) pairs; slices of one tile are adjacent, so a tile's slabs stay in one L2.
    const int nt = p.tiles_m * p.tiles_n;
    const int nwg = nt * p.splitk;
    const int xq = nwg / 8, xr = nwg % 8, xcd = orig % 8;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + orig / 8;
    const int split = wg % p.splitk, tile = wg / p.splitk;
    // grouped rasterisation: consecutive tiles walk 8 tile-rows before moving one tile-column over, so the ~32 tiles an
    // XCD works on at any moment form an 8 x 4 block sharing 8 + 4 operand panels (instead of 32 + 1): ~2.7x less
    // L2 miss traffic once the operands outgrow the 4 MiB L2.
    // (round 5: the group height follows the run an XCD owns -- ~sqrt(tiles per XCD), 8 at most: a run of 10 tiles as 3 x 3.3 touches 7 operand panels, as 8 x 1.25 ten)
    const int GR = p.gr;
    const int gsz = GR * p.tiles_n;
    const int grp = tile / gsz, first_m = grp * GR;
    const int rows_in = min(p.tiles_m - first_m, GR);
    const int tile_m = first_m + (tile - grp * gsz) % rows_in, tile_n = (tile - grp * gsz) / rows_in;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int z = bz;
    const long zo = z / p.batch_inner, zi = z % p.batch_inner;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + zo * p.sAo + zi * p.sAi;
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B) + zo * p.sBo + zi * p.sBi;
    // Buffer extents for the bounds check (out-of-range dwords read as zero).  An MN-contiguous operand's last row is
    // rounded up to whole 16-byte pieces: the check is per dword, and an odd MN count would otherwise zero the last
    // element of the last K-row (its row pitch is a multiple of 8 elements, so those bytes exist; they only feed
    // output indices >= M / N, which are never stored).
    const unsigned a_bytes = CONV == 1 ? (unsigned)(p.cg.a_ext * 2)
                                       : (unsigned)((A_MC ? (long)(p.K - 1) * p.lda + ((p.M + 7) & ~7) : (long)(p.M - 1) * p.lda + p.K) * 2);
    const unsigned b_bytes = (CONV == 2 || (CONV == 1 && B_MC)) ? (unsigned)(p.cg.b_ext * 2)
                                       : (unsigned)((B_MC ? (long)(p.K - 1) * p.ldb + ((p.N + 7) & ~7) : (long)(p.N - 1) * p.ldb + p.K) * 2);
    // (hipcc's host pass drops a kernel's launch stub -- silently -- when a target builtin is called with type-dependent
    // operands: the DMA builtin below only ever sees non-dependent locals, hence the ISSUE_STAGE macro)
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), (short)0, (int)a_bytes, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B), (short)0, (int)b_bytes, 0x00020000);

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm0 = (wid / TL::WN) * (BM / TL::WM), wn0 = (wid % TL::WN) * (BN / TL::WN);

    const int kbeg = split * p.ksteps_per_split;
    const int kend = min(p.ksteps, kbeg + p.ksteps_per_split);
    const int nk = kend - kbeg;

    // per-lane source offsets of this wave's DMA pieces (piece q = j * NW + wid), advanced by one K-step per issue
    const unsigned stepA = A_MC ? (unsigned)(BK * p.lda * 2) : (unsigned)(BK * 2);
    const unsigned stepB = B_MC ? (unsigned)(BK * p.ldb * 2) : (unsigned)(BK * 2);
    unsigned voA[TL::PA], voB[TL::PB];
#pragma unroll
    for (int j = 0; j < TL::PA; ++j) voA[j] = dma_voffset<A_MC, BM, BKT>(j * TL::NW + wid, lane, m0, p.lda) + (CONV == 1 ? 0u : (unsigned)kbeg * stepA);
#pragma unroll
    for (int j = 0; j < TL::PB; ++j)
        voB[j] = dma_voffset<B_MC, BN, BKT>(j * TL::NW + wid, lane, n0, p.ldb) + ((CONV == 2 || (CONV == 1 && B_MC)) ? 0u : (unsigned)kbeg * stepB);

    // ---- implicit-GEMM convolution: gathered-pixel state of this lane's DMA pieces
    const ConvGeom& cg = p.cg;
    constexpr unsigned OOB = 0x80000000u;            // >= every buffer extent (< 2^31 bytes): the bounds check returns zeros
    int c_kc = 0, c_kx = 0, c_ky = 0;                // (tap (ky, kx), 64-channel chunk) of the next K-step to issue (CONV 1); CONV 2: this launch's tap
    int gy[CONV == 2 ? TL::PB : TL::PA], gx[CONV == 2 ? TL::PB : TL::PA];     // pixel (y, x) of the piece's row (CONV 1) / k-row (CONV 2)
    unsigned gb[CONV == 2 ? TL::PB : TL::PA];                                 // its image's first pixel index in the gathered tensor (b * src_h * src_w)
    unsigned gch[CONV == 2 ? TL::PB : TL::PA];                                // byte offset inside a pixel's channel row (16-byte chunk / n-chunk)
    int adv_x = 0, adv_y = 0;
    if constexpr (CONV == 1) {
        const int t0 = kbeg / cg.cchunks;
        c_kc = kbeg - t0 * cg.cchunks; c_ky = t0 / cg.kw; c_kx = t0 - c_ky * cg.kw;
        const int plane = cg.rows_h * cg.rows_w;
#pragma unroll
        for (int j = 0; j < TL::PA; ++j) {
            const int row = 8 * (j * TL::NW + wid) + (lane >> 3);
            const int lc = (lane & 7) ^ ((row >> 1) & 7);
            const int m = m0 + row;
            const int b = m / plane, r = m - b * plane;
            const int y = r / cg.rows_w;
            gy[j] = m < p.M ? y : -(1 << 24);       // rows past M: every tap is out of range
            gx[j] = r - y * cg.rows_w;
            gb[j] = (unsigned)(b * cg.src_h * cg.src_w);
            gch[j] = (unsigned)(lc * 16);
        }
    }
    if constexpr (CONV == 2) {
        const int t = cg.tap0 + bz;
        c_ky = t / cg.kw; c_kx = t - c_ky * cg.kw;
        adv_y = BK / cg.rows_w; adv_x = BK - adv_y * cg.rows_w;
        const int plane = cg.rows_h * cg.rows_w;
        constexpr int CPR = BN / 8, G = BN / 32;
#pragma unroll
        for (int j = 0; j < TL::PB; ++j) {
            const int q = j * TL::NW + wid;
            const int krow = q * (64 / CPR) + lane / CPR;
            const int pc = lane % CPR;
            const int f = G >= 4 ? (krow & 3) : ((krow >> 1) & 1);
            const int lc = (((pc >> 2) ^ f) << 2) | (pc & 3);
            const int pix = kbeg * BK + krow;
            const int b = pix / plane, r = pix - b * plane;
            gy[j] = r / cg.rows_w; gx[j] = r - gy[j] * cg.rows_w;
            gb[j] = (unsigned)b;
            gch[j] = (unsigned)((n0 + lc * 8) * 2);
        }
    }
    // forward-type gather: enumerated pixel (y, x), tap (ky, kx) -> source pixel of the (nearest-up-sampled) tensor, or out of range
    auto gather_fwd = [&](int y, int x, int& sy, int& sx) -> bool {
        sy = (y << cg.stride_log2) + c_ky - cg.pad; sx = (x << cg.stride_log2) + c_kx - cg.pad;
        const bool ok = (unsigned)sy < (unsigned)(cg.src_h << cg.ups_log2) && (unsigned)sx < (unsigned)(cg.src_w << cg.ups_log2);
        sy >>= cg.ups_log2; sx >>= cg.ups_log2;
        return ok;
    };
    auto conv_off_a = [&](int j) -> unsigned {      // CONV 1
        int sy, sx; bool ok;
        if (!cg.flip) ok = gather_fwd(gy[j], gx[j], sy, sx);
        else {
            const int ny = gy[j] + cg.pad - c_ky, nx = gx[j] + cg.pad - c_kx;
            ok = (ny | nx) >= 0 && ((ny | nx) & ((1 << cg.stride_log2) - 1)) == 0;
            sy = ny >> cg.stride_log2; sx = nx >> cg.stride_log2;
            ok = ok && sy < cg.src_h && sx < cg.src_w;
        }
        return ok ? (unsigned)(((long)(gb[j] + (unsigned)(sy * cg.src_w + sx)) * p.lda + c_kc * BK) * 2) + gch[j] : OOB;
    };
    auto conv_off_b = [&](int j) -> unsigned {      // CONV 2
        int sy, sx;
        const bool ok = gather_fwd(gy[j], gx[j], sy, sx);
        return ok ? (unsigned)(((long)((gb[j] * (unsigned)cg.src_h + (unsigned)sy) * (unsigned)cg.src_w + (unsigned)sx) * p.ldb) * 2) + gch[j] : OOB;
    };
    // One K-step of this wave's DMA pieces into ring buffer `buf` (piece j of an operand lands at LDS byte
    // (j * NW + wid) * 1024 of its image); advances the per-lane source offsets by one K-step.
#define ISSUE_RANGE(buf, J0, J1)                                                                                              \
    do {                                                                                                                      \
        char* base_ = lds + (buf) * TL::STAGE_BYTES + wid * 1024;                                                             \
        const unsigned kofsB_ = (CONV == 1 && B_MC) ? (unsigned)(((long)c_kc * BK * p.ldb + (long)(c_ky * cg.kw + c_kx) * cg.b_tap_stride) * 2) : 0u; \
        _Pragma("unroll") for (int j = (J0); j < (J1); ++j) {                                                                 \
            if (j < TL::PA) {                                                                                                 \
                char* dst_ = base_ + j * TL::NW * 1024;                                                                       \
                unsigned off_;   /* non-dependent builtin operands */                                                        \
                if constexpr (CONV == 1) off_ = conv_off_a(j); else { off_ = voA[j]; voA[j] += stepA; }                       \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)dst_, 16, off_, 0, 0, 0);                          \
            } else {                                                                                                          \
                char* dst_ = base_ + TL::IMG_A + (j - TL::PA) * TL::NW * 1024;                                                \
                unsigned off_;                                                                                                \
                if constexpr (CONV == 2) off_ = conv_off_b(j - TL::PA);                                                       \
                else if constexpr (CONV == 1 && B_MC) off_ = voB[j - TL::PA] + kofsB_;                                        \
                else { off_ = voB[j - TL::PA]; voB[j - TL::PA] += stepB; }                                                    \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)dst_, 16, off_, 0, 0, 0);                          \
            }                                                                                                                 \
        }                                                                                                                     \
        if ((J1) == NLOAD) {          /* the whole K-step is issued: move the gather state on by one K-step */               \
            if constexpr (CONV == 1) {                                                                                        \
                if (++c_kc == cg.cchunks) { c_kc = 0; if (++c_kx == cg.kw) { c_kx = 0; ++c_ky; } }                            \
            }                                                                                                                 \
            if constexpr (CONV == 2) {                                                                                        \
                _Pragma("unroll") for (int j = 0; j < TL::PB; ++j) {                                                          \
                    gx[j] += adv_x; gy[j] += adv_y;                                                                           \
                    if (gx[j] >= cg.rows_w) { gx[j] -= cg.rows_w; ++gy[j]; }                                                  \
                    while (gy[j] >= cg.rows_h) { gy[j] -= cg.rows_h; ++gb[j]; }                                               \
                }                                                                                                             \
            }                                                                                                                 \
        }                                                                                                                     \
    } while (0)
#define ISSUE_STAGE(buf) ISSUE_RANGE(buf, 0, NLOAD)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fused bias gradient of a wgrad GEMM (A = dy stored [tokens][features]): colsum[m] = sum_k A[k][m], taken from the A
    // fragments the MFMAs consume anyway -- by the waves of the tile column 0 workgroups that own distinct m rows
    float csum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) csum[i] = 0.f;
    const bool colsum_here = A_MC && p.colsum != nullptr && tile_n == 0 && (CONV != 2 || bz == 0);   // CONV 2: every tap sees the same dy
    const bool do_colsum = colsum_here && (wid % TL::WN) == 0;

    // ---- register-staged feed (VS > 0): set r of rg holds one K-step's pieces of this wave on their way global -> LDS
    // The steady-state loop carries NO branch around a load or a ds_write: hipcc's s_waitcnt insertion gives up counting at control-flow joins (the first version, with
    // `if (reload)` around the loads, waited vmcnt(0) ahead of every ds_write -- every load in flight drained, i.e. no pipeline).  A K-step past the end is therefore
    // still "loaded", from an offset beyond the buffer extent (the bounds check returns zeros without a memory request), and its ds_write lands in the slot nobody reads.
    u32x4_t rg[VS > 0 ? VS : 1][NLOAD];
    constexpr unsigned VS_OOB = 0x80000000u;
    // (same rule as the DMA builtin: the buffer builtins only ever see non-dependent locals)
#define VS_LOAD(r, J0, J1, LIVE)                                                                                              \
    do {                                                                                                                      \
        _Pragma("unroll") for (int j = (J0); j < (J1); ++j) {                                                                 \
            if (j < TL::PA) { const unsigned off_ = (LIVE) ? voA[j] : VS_OOB; voA[j] += stepA; rg[r][j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off_, 0, 0); }            \
            else { const unsigned off_ = (LIVE) ? voB[j - TL::PA] : VS_OOB; voB[j - TL::PA] += stepB; rg[r][j] = __builtin_amdgcn_raw_buffer_load_b128(rsB, off_, 0, 0); }     \
        }                                                                                                                     \
    } while (0)
    // piece j of an operand -> LDS byte (j * NW + wid) * 1024 + 16 * lane of its image: the LDS-DMA's own lane-linear destination, so the image formats, the source
    // swizzles (dma_voffset) and every fragment read are the DMA ring's
#define VS_WRITE(r, buf, J0, J1)                                                                                              \
    do {                                                                                                                      \
        char* base_ = lds + (buf) * TL::STAGE_BYTES + wid * 1024 + lane * 16;                                                 \
        _Pragma("unroll") for (int j = (J0); j < (J1); ++j) {                                                                 \
            char* dst_ = base_ + (j < TL::PA ? j * TL::NW * 1024 : TL::IMG_A + (j - TL::PA) * TL::NW * 1024);                 \
            *reinterpret_cast<u32x4_t*>(dst_) = rg[r][j];                                                                     \
        }                                                                                                                     \
    } while (0)

    // ---- prologue: STAGES - 1 K-steps in flight (VS > 0: K-steps 0 .. VS - 1 into the register sets, step 0 on into LDS slot 0, step VS behind it)
    if constexpr (VS == 0) {
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < nk) ISSUE_STAGE(s);
    } else {
#pragma unroll
        for (int r = 0; r < VS; ++r) VS_LOAD(r, 0, NLOAD, r < nk);
        VS_WRITE(0, 0, 0, NLOAD);
        VS_LOAD(0, 0, NLOAD, VS < nk);
    }

    TL_STAMP(1);
    int cur = 0, nxt = STAGES - 1;
    constexpr int UNR = VS > 0 ? VS : 1;       // VS > 0: the loop is unrolled by VS so that every iteration names its register set statically
    for (int it0 = 0; it0 < nk; it0 += UNR) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int it = it0 + u;      // (VS > 1: the last group may run past nk -- no break: a branch inside the unrolled group makes hipcc's s_waitcnt insertion fall back to
                                     //  conservative counts (ISA checked: vmcnt 15 / 11 / 4 / 3 over the four copies instead of 15 everywhere).  K-steps past nk were "loaded"
                                     //  from beyond the buffer extent, i.e. are zeros in registers and LDS: their MFMAs add nothing; at most VS - 1 dead K-steps per tile)
        constexpr int RSET_BASE = 1;           // iteration it writes K-step it + 1 (register set (it + 1) % VS) into the other LDS slot and reloads that set with step it + 1 + VS
        const int rset = VS > 0 ? (u + RSET_BASE) % UNR : 0;
        // retire this wave's DMA of K-step `it` (later steps stay in flight), then one barrier: every wave's share of
        // step `it` has landed AND every wave has finished reading buffer `nxt` (it computed step it-1 from it).
        // (VS > 0: this wave's ds_writes of step `it` -- issued one iteration ago -- have completed)
        if constexpr (VS == 0) wait_dma_ahead<NLOAD, (STAGES > 2 ? STAGES - 2 : 1)>(min(nk - it - 1, STAGES - 2));
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TL_STAMP(4 + 4 * it);
        __builtin_amdgcn_s_barrier();
        TL_STAMP(5 + 4 * it);
        const bool refill = VS == 0 && it + STAGES - 1 < nk;
        const bool reload = VS > 0 && it + 1 + VS < nk;        // VS > 0: K-step it + 1 moves registers -> LDS during this iteration, K-step it + 1 + VS (if any) takes its registers
        TL_STAMP(6 + 4 * it);
        const char* imgA = lds + cur * TL::STAGE_BYTES;
        const char* imgB = imgA + TL::IMG_A;
        if constexpr (TM * TN < 8) {
            // all fragment reads of the K-step are issued up front: the MFMAs of k-slice ks start as soon as their
            // fragments land while the later slices are still in flight (counted lgkmcnt by the compiler)
            bf16x8_t fa[KS][TM], fb[KS][TN];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[ks][i] = read_frag<A_MC, BM, BKT>(imgA, wm0 + i * 32, ks, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[ks][j] = read_frag<B_MC, BN, BKT>(imgB, wn0 + j * 32, ks, lane);
            }
            if (do_colsum) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int i = 0; i < TM; ++i) csum[i] += frag_sum(fa[ks][i]);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                // the refill of ring buffer `nxt` is issued a quarter at a time BETWEEN the k-slices' MFMAs: a DMA piece costs the wave
                // 75 .. 85 issue cycles (timeline probe), which now run while the matrix pipe works instead of ahead of the whole K-step
                if constexpr (VS == 0) { if (refill) ISSUE_RANGE(nxt, ks * NLOAD / KS, (ks + 1) * NLOAD / KS); }
                else {
                    VS_WRITE(rset, nxt, ks * NLOAD / KS, (ks + 1) * NLOAD / KS);
                    VS_LOAD(rset, ks * NLOAD / KS, (ks + 1) * NLOAD / KS, reload);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)   // operands swapped: D[row = n][col = m]
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8_mfma, fb[ks][j]), __builtin_bit_cast(bf16x8_mfma, fa[ks][i]), acc[i][j], 0, 0, 0);
            }
        } else {
            // 8 MFMA tiles per wave (256^2 configuration, 128 accumulator VGPRs): fragments are double-buffered per k-slice --
            // slice ks + 1 is read while the 8 MFMAs of slice ks run
            bf16x8_t fa[2][TM], fb[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[0][i] = read_frag<A_MC, BM, BKT>(imgA, wm0 + i * 32, 0, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[0][j] = read_frag<B_MC, BN, BKT>(imgB, wn0 + j * 32, 0, lane);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                // this wave's share of the next K-step's DMA is spread over the first two k-slices: a piece costs the issuing wave
                // 60 .. 180 cycles (MI355X_MICROARCH.md), which now falls into the shadow of the other wave's MFMAs instead of
                // both waves of a SIMD issuing all their pieces right after the barrier
                if constexpr (VS == 0) {
                    if (refill && ks < 2) ISSUE_RANGE(nxt, ks * NLOAD / 2, (ks + 1) * NLOAD / 2);   // first half of the K-step: the second half is the landing window
                }
                if (ks + 1 < KS) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[(ks + 1) & 1][i] = read_frag<A_MC, BM, BKT>(imgA, wm0 + i * 32, ks + 1, lane);
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[(ks + 1) & 1][j] = read_frag<B_MC, BN, BKT>(imgB, wn0 + j * 32, ks + 1, lane);
                }
                if constexpr (VS > 0) {      // register-staged: a quarter of the pieces per k-slice, behind the slice's fragment reads (LDS operations retire in order)
                    VS_WRITE(rset, nxt, ks * NLOAD / KS, (ks + 1) * NLOAD / KS);
                    VS_LOAD(rset, ks * NLOAD / KS, (ks + 1) * NLOAD / KS, reload);
                }
                if (do_colsum) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) csum[i] += frag_sum(fa[ks & 1][i]);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8_mfma, fb[ks & 1][j]), __builtin_bit_cast(bf16x8_mfma, fa[ks & 1][i]), acc[i][j], 0, 0, 0);
            }
        }
        TL_STAMP(7 + 4 * it);
        cur = (cur + 1 == STAGES) ? 0 : cur + 1;
        nxt = (nxt + 1 == STAGES) ? 0 : nxt + 1;
      }
    }
    TL_STAMP(2);

#undef ISSUE_STAGE
#undef ISSUE_RANGE
#undef VS_LOAD
#undef VS_WRITE
    // ---- split-K: publish this slice's slab; the last arriver of the tile reduces all slabs in slice order
    if (p.splitk > 1) {
        float4* slab0 = reinterpret_cast<float4*>(p.slabs) + ((long)(z * nt + tile) * p.splitk) * TL::SLAB_F4;
        float4* mine = slab0 + (long)split * TL::SLAB_F4;
#if DPIPE_SLAB_WT
        // write-through publish (cdna_hip_programming.md section 6 Guideline 16, recipe R1): the slab goes out as 16-byte sc1 stores (they leave the XCD's L2 for
        // memory as they retire), every storing wave drains its own vmcnt, then ONE lane draws the ticket -- no agent-scope release, i.e. no buffer_wbl2 scan of the
        // L2 per slice (the price list's publish-large row: 3.0 vs 8.2 us for a 64 KB slab per workgroup)
        const auto rsS = __builtin_amdgcn_make_buffer_rsrc(mine, (short)0, (int)(TL::SLAB_F4 * 16), 0x00020000);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u32x4_t v;
                    v[0] = __float_as_uint(acc[i][j][4 * q]); v[1] = __float_as_uint(acc[i][j][4 * q + 1]);
                    v[2] = __float_as_uint(acc[i][j][4 * q + 2]); v[3] = __float_as_uint(acc[i][j][4 * q + 3]);
                    // per-lane part in voffset (one VGPR for all stores), the piece's constant offset in soffset (an SGPR: it exceeds the 12-bit immediate)
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsS, (int)threadIdx.x * 16, ((i * TN + j) * 4 + q) * TL::NT * 16, 16);      // aux 16 = sc1
                }
        if (do_colsum) {
            float* crow = reinterpret_cast<float*>(mine + TL::SLAB_F4 - BM / 4);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float v = csum[i] + __shfl_xor(csum[i], 32, 64);
                if (lane < 32) __hip_atomic_store(&crow[wm0 + i * 32 + lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // <= 8 bytes: an sc1 store
            }
        }
#else
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    mine[((i * TN + j) * 4 + q) * TL::NT + threadIdx.x] =
                        make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
        if (do_colsum) {
            float* crow = reinterpret_cast<float*>(mine + TL::SLAB_F4 - BM / 4);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float v = csum[i] + __shfl_xor(csum[i], 32, 64);
                if (lane < 32) crow[wm0 + i * 32 + lane] = v;
            }
        }
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // all slab stores of this workgroup issued and waited for
        int* flag = reinterpret_cast<int*>(lds);               // the one LDS array doubles as the broadcast word
        if (threadIdx.x == 0) {
#if !DPIPE_SLAB_WT
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            const int ticket = __hip_atomic_fetch_add(&p.counters[z * nt + tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = (ticket == p.splitk - 1);
        }
        __syncthreads();
        if (!*flag) return;
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(&p.counters[z * nt + tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        // the slices' slabs, summed in slice order (deterministic).  Round 6: SLB slices per pass with every read of the pass in flight before the first add (the
        // one-slice-per-iteration loop was a dependent round trip per slice: 4 - 8 of them in the last arriver of a 77-token linear); a pass past the last slice
        // re-reads the last slab and skips the add.  SLB = 4 on the 64^2 tile (the split tiles of the step are almost all 64^2: 4 x 16 VGPRs of reads).
        constexpr int SLB = TL::ACC_F4 <= 4 ? 4 : 1;      // (two slices per pass on the 128^2 tiles: 104 -> 108 VGPRs, a register allocation step the lanes feel -- left at one)
        if constexpr (SLB == 1) {
            for (int s = 0; s < p.splitk; ++s) {
                const float4* sl = slab0 + (long)s * TL::SLAB_F4;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 v = sl[((i * TN + j) * 4 + q) * TL::NT + threadIdx.x];
                            acc[i][j][4 * q] += v.x; acc[i][j][4 * q + 1] += v.y; acc[i][j][4 * q + 2] += v.z; acc[i][j][4 * q + 3] += v.w;
                        }
            }
        } else {
            for (int s = 0; s < p.splitk; s += SLB) {
                float4 sv[SLB][TM * TN * 4];
#pragma unroll
                for (int t = 0; t < SLB; ++t) {
                    const int ss = s + t < p.splitk ? s + t : p.splitk - 1;
                    const float4* sl = slab0 + (long)ss * TL::SLAB_F4;
#pragma unroll
                    for (int x = 0; x < TM * TN * 4; ++x) sv[t][x] = sl[x * TL::NT + threadIdx.x];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < SLB; ++t) {
                    if (s + t < p.splitk) {
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const float4 v = sv[t][(i * TN + j) * 4 + q];
                                    acc[i][j][4 * q] += v.x; acc[i][j][4 * q + 1] += v.y; acc[i][j][4 * q + 2] += v.z; acc[i][j][4 * q + 3] += v.w;
                                }
                    }
                }
            }
        }
    }

    // ---- fused column sums: final value = this workgroup's (split-K: the slices' slab rows, in slice order)
    if (colsum_here) {
        if (p.splitk > 1) {
            if (threadIdx.x < BM) {
                const float* crow0 = reinterpret_cast<const float*>(reinterpret_cast<float4*>(p.slabs) + ((long)(z * nt + tile) * p.splitk) * TL::SLAB_F4 +
                                                                    TL::SLAB_F4 - BM / 4);
                float v = 0.f;
                for (int s = 0; s < p.splitk; ++s) v += crow0[(long)s * TL::SLAB_F4 * 4 + threadIdx.x];
                const int m = m0 + threadIdx.x;
                if (m < p.M) {
                    bf16_t* dst = reinterpret_cast<bf16_t*>(p.colsum) + (long)z * p.M + m;
                    *dst = f32_to_bf16(p.colsum_acc ? bf16_to_f32(*dst) + v : v);
                }
            }
        } else if (do_colsum) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float v = csum[i] + __shfl_xor(csum[i], 32, 64);
                const int m = m0 + wm0 + i * 32 + lane;
                if (lane < 32 && m < p.M) {
                    bf16_t* dst = reinterpret_cast<bf16_t*>(p.colsum) + (long)z * p.M + m;
                    *dst = f32_to_bf16(p.colsum_acc ? bf16_to_f32(*dst) + v : v);
                }
            }
        }
    }

    // ---- epilogue.  D layout of the swapped 32x32 MFMA: m = lane & 31, n = (e & 3) + 8 (e >> 2) + 4 (lane >> 5):
    // for each q = e >> 2 the lane holds 4 consecutive n of one output row.
    const long coff = zo * p.sCo + zi * p.sCi;
    const int h = lane >> 5;
    // Batched epilogue (round 6), taken by the bf16-output launches of the 64^2 / 128^2 tiles that READ something in the epilogue (a bias, a residual, the old C of an
    // accumulating launch) and whose rows allow 8-byte accesses (N % 4 == 0: the 4 columns a lane owns are inside or outside together): the reads of a 32 x 32
    // accumulator tile -- 4 lane groups x up to 4 operands x 8 bytes -- are all issued before the first is used, and NO store is issued before the last read has
    // returned (gfx9 counts stores in vmcnt: a read behind a store waits for the store's acknowledgement as well).  The general form below wraps every read in a
    // wave-uniform `if` (operand present? vector legal?), which hipcc compiles to a branch with `s_waitcnt vmcnt(0)` behind each load: per lane 8 groups x (bias +
    // residual + old C) dependent round trips, each behind the previous group's store (ISA: profiles/r6n_gemm_epilogue_isa.txt).  An absent operand reads the zero
    // pad (branch-free pointer select), rows past M are clamped to M - 1 and columns past N to 0 for the reads and skipped for the stores; the arithmetic order
    // (alpha, bias hi, bias lo, activation, residual, old C) is the general form's, so results are bit-identical.  A launch that reads nothing keeps the general
    // form: its stores are fire-and-forget, and a first version that sent it through here (16 pad reads per tile ahead of the stores) cost every plain launch
    // 1 - 4 us (profiles/r6n_gemm_ledger_epilogue_v1_{base,new}.jsonl: the ledger +4 %).
    if (TM * TN <= 2 && !p.out_f32 && p.vecA >= 2 && (p.N & 3) == 0 && (!p.bias || p.vecB >= 2) && (!p.residual || p.vecA >= 3) &&
        (p.bias || p.residual || p.accumulate)) {
        // GEGLU backward in this epilogue (DPIPE_ACT_GEGLU_BWD | activation; plain GEMMs only): the GEMM is the dgrad of the Linear BEHIND a GEGLU, acc = dy [M, N];
        // `residual` = the GEGLU's input h [M, 2 N] (value | gate halves), C = dh [M, 2 N]:  dh[m, n] = dy * act(gate),  dh[m, N + n] = dy * value * act'(gate).
        // dy never reaches memory and the separate geglu_bwd pass (3 reads + 2 writes of [M, N]-sized tensors, the most expensive element-wise kernel of the SDXL
        // step under the lanes: profiles/r6p_ablation_census_four_lane_step.jsonl) is gone.  The gate vectors travel in the `l` slots, the second result in outw.
        const bool gg = CONV == 0 && (p.act & ACT_GEGLU_BWD) != 0;
        const int iact = p.act & 15;
        const bool hb = p.bias != nullptr, hl = CONV != 0 && hb && p.bias_lo != 0, hr = p.residual != nullptr, ha = p.accumulate != 0;     // (hi / lo bias pairs: convolutions only)
        const uint2* pad = reinterpret_cast<const uint2*>(g_param_pad);
        const bf16_t* biasp = reinterpret_cast<const bf16_t*>(p.bias);
        const bf16_t* resp = reinterpret_cast<const bf16_t*>(p.residual);
        bf16_t* cp = reinterpret_cast<bf16_t*>(p.C);
        const int brows = p.bias_rows ? p.bias_rows : 0x7fffffff;       // one bias row for every output row: row index 0
        constexpr int U = TM * TN <= 2 ? TM * TN : 1;
        struct EpiOps { uint2 b[4], l[4], r[4], o[4]; };
        uint2 outv[U][4], outw[CONV == 0 ? U : 1][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = u / TN, j = u % TN;
            const int m = m0 + wm0 + i * 32 + (lane & 31);
            const int mm = m < p.M ? m : p.M - 1;
            const long brow = (long)(mm / brows) * p.N;
            const int nb = n0 + wn0 + j * 32 + 4 * h;                  // group q covers columns nb + 8 q .. + 3
            const long crow = coff + (long)mm * p.ldc, rrow = coff + (long)mm * p.ldr;
            EpiOps e;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = nb + 8 * q;
                const int nn = n < p.N ? n : 0;
                e.b[q] = *(hb ? reinterpret_cast<const uint2*>(biasp + brow + nn) : pad);
                if constexpr (CONV != 0) e.l[q] = *(hl ? reinterpret_cast<const uint2*>(biasp + brow + nn + p.bias_lo) : pad);
                else e.l[q] = *(gg ? reinterpret_cast<const uint2*>(resp + rrow + p.N + nn) : pad);
                e.r[q] = *(hr ? reinterpret_cast<const uint2*>(resp + rrow + nn) : pad);
                e.o[q] = *(ha ? reinterpret_cast<const uint2*>(cp + crow + nn) : pad);
            }
            __builtin_amdgcn_sched_barrier(0);          // the reads above stay above: the machine scheduler would sink them next to their uses
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = p.alpha * acc[i][j][4 * q + r];
                if (hb) {
                    v[0] += __uint_as_float(e.b[q].x << 16); v[1] += __uint_as_float(e.b[q].x & 0xffff0000u);
                    v[2] += __uint_as_float(e.b[q].y << 16); v[3] += __uint_as_float(e.b[q].y & 0xffff0000u);
                }
                if constexpr (CONV != 0) if (hl) {
                    v[0] += __uint_as_float(e.l[q].x << 16); v[1] += __uint_as_float(e.l[q].x & 0xffff0000u);
                    v[2] += __uint_as_float(e.l[q].y << 16); v[3] += __uint_as_float(e.l[q].y & 0xffff0000u);
                }
                if constexpr (CONV == 0) if (gg) {
                    float vl[4], gt[4], w[4];
                    vl[0] = __uint_as_float(e.r[q].x << 16); vl[1] = __uint_as_float(e.r[q].x & 0xffff0000u);
                    vl[2] = __uint_as_float(e.r[q].y << 16); vl[3] = __uint_as_float(e.r[q].y & 0xffff0000u);
                    gt[0] = __uint_as_float(e.l[q].x << 16); gt[1] = __uint_as_float(e.l[q].x & 0xffff0000u);
                    gt[2] = __uint_as_float(e.l[q].y << 16); gt[3] = __uint_as_float(e.l[q].y & 0xffff0000u);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { w[r] = v[r] * vl[r] * epilogue_act_grad(gt[r], iact); v[r] = v[r] * epilogue_act(gt[r], iact); }
                    outv[u][q] = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                    outw[u][q] = make_uint2(pack_bf16x2(w[0], w[1]), pack_bf16x2(w[2], w[3]));
                    continue;
                }
                if (iact != ACT_NONE) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = epilogue_act(v[r], iact);
                }
                if (hr) {
                    v[0] += __uint_as_float(e.r[q].x << 16); v[1] += __uint_as_float(e.r[q].x & 0xffff0000u);
                    v[2] += __uint_as_float(e.r[q].y << 16); v[3] += __uint_as_float(e.r[q].y & 0xffff0000u);
                }
                if (ha) {
                    v[0] += __uint_as_float(e.o[q].x << 16); v[1] += __uint_as_float(e.o[q].x & 0xffff0000u);
                    v[2] += __uint_as_float(e.o[q].y << 16); v[3] += __uint_as_float(e.o[q].y & 0xffff0000u);
                }
                outv[u][q] = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = u / TN, j = u % TN;
            const int m = m0 + wm0 + i * 32 + (lane & 31);
            const int nb = n0 + wn0 + j * 32 + 4 * h;
            bf16_t* crow = cp + coff + (long)m * p.ldc;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (m < p.M && nb + 8 * q < p.N) {
                    *reinterpret_cast<uint2*>(crow + nb + 8 * q) = outv[u][q];
                    if constexpr (CONV == 0) if (gg) *reinterpret_cast<uint2*>(crow + p.N + nb + 8 * q) = outw[u][q];
                }
        }
        TL_STAMP(3);
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm0 + i * 32 + (lane & 31);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn0 + j * 32 + 8 * q + 4 * h;
                if (n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = p.alpha * acc[i][j][4 * q + r];
                const bool full = n + 3 < p.N;
                if (p.bias) {
                    const bf16_t* bp = reinterpret_cast<const bf16_t*>(p.bias) + (p.bias_rows ? (long)(m / p.bias_rows) * p.N : 0L) + n;
                    if (full && p.vecB >= 2) {
                        const uint2 bv = *reinterpret_cast<const uint2*>(bp);
                        v[0] += __uint_as_float(bv.x << 16); v[1] += __uint_as_float(bv.x & 0xffff0000u);
                        v[2] += __uint_as_float(bv.y << 16); v[3] += __uint_as_float(bv.y & 0xffff0000u);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += bf16_to_f32(bp[r]);
                    }
                    if (p.bias_lo) {                       // the LO set of a hi / lo bias pair (an fp32 per-channel addend: DPIPE_CONV_BIAS_HILO; N % 4 == 0)
                        const uint2 lv = *reinterpret_cast<const uint2*>(bp + p.bias_lo);
                        v[0] += __uint_as_float(lv.x << 16); v[1] += __uint_as_float(lv.x & 0xffff0000u);
                        v[2] += __uint_as_float(lv.y << 16); v[3] += __uint_as_float(lv.y & 0xffff0000u);
                    }
                }
                if (p.act != ACT_NONE) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = epilogue_act(v[r], p.act);
                }
                const long idx = coff + (long)m * p.ldc + n;
                if (p.residual) {
                    const long ridx = coff + (long)m * p.ldr + n;
                    if (p.out_f32) {
                        const float* rp = reinterpret_cast<const float*>(p.residual) + ridx;
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += rp[r];
                    } else {
                        const bf16_t* rp = reinterpret_cast<const bf16_t*>(p.residual) + ridx;
                        if (full && p.vecA >= 3) {
                            const uint2 rv = *reinterpret_cast<const uint2*>(rp);
                            v[0] += __uint_as_float(rv.x << 16); v[1] += __uint_as_float(rv.x & 0xffff0000u);
                            v[2] += __uint_as_float(rv.y << 16); v[3] += __uint_as_float(rv.y & 0xffff0000u);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += bf16_to_f32(rp[r]);
                        }
                    }
                }
                if (p.out_f32) {
                    float* c = reinterpret_cast<float*>(p.C) + idx;
                    if (full && p.vecA >= 2) {
                        float4 o = make_float4(v[0], v[1], v[2], v[3]);
                        if (p.accumulate) { const float4 old = *reinterpret_cast<const float4*>(c); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
                        *reinterpret_cast<float4*>(c) = o;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (n + r < p.N) c[r] = p.accumulate ? c[r] + v[r] : v[r];
                    }
                } else {
                    bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + idx;
                    if (full && p.vecA >= 2) {
                        if (p.accumulate) {
                            const uint2 old = *reinterpret_cast<const uint2*>(c);
                            v[0] += __uint_as_float(old.x << 16); v[1] += __uint_as_float(old.x & 0xffff0000u);
                            v[2] += __uint_as_float(old.y << 16); v[3] += __uint_as_float(old.y & 0xffff0000u);
                        }
                        *reinterpret_cast<uint2*>(c) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n + r < p.N) c[r] = f32_to_bf16(p.accumulate ? bf16_to_f32(c[r]) + v[r] : v[r]);
                    }
                }
            }
    }
    TL_STAMP(3);
}

template <int BM_, int BN_, int WM_, int WN_, int STAGES_, bool A_MC, bool B_MC, int CONV = 0, int BKT = 64, int VS = 0>
__global__ void __launch_bounds__(WM_ * WN_ * 64, (VS > 0 && !A_MC) ? 2 : 1) gemm_pipe_kernel(const GemmParams p) {     // register-staged 128^2: two workgroups per CU (<= 128 VGPRs)
    using TL = Tile<BM_, BN_, WM_, WN_, STAGES_, BKT, VS>;
    __shared__ __attribute__((aligned(1024))) char lds[TL::STAGES * TL::STAGE_BYTES];
    gemm_pipe_body<BM_, BN_, WM_, WN_, STAGES_, A_MC, B_MC, CONV, BKT, VS>(p, lds, (int)blockIdx.x, (int)blockIdx.y);
}

// ---- grouped launch: up to GROUP_MAX INDEPENDENT plain GEMMs (batch 1, one tile geometry, any mix of operand layouts) as ONE kernel launch.  Each workgroup
// finds its problem from the table of first-workgroup indices (a problem's share starts at a multiple of 8, so its workgroups keep the XCD round-robin the body's
// remap assumes; the padding workgroups exit at once) and runs the ordinary body on it.  What it buys: the dgrad and wgrad of a Linear (same dy, nothing else
// in common) fill the chip together instead of as two half-empty launches with a kernel boundary in between; the launch list of a micro-batch shrinks by a third.
constexpr int GROUP_MAX = 4;
struct GemmGroup {
    GemmParams p[GROUP_MAX];
    int start[GROUP_MAX];        // first workgroup of problem i (multiple of 8)
    int nwg[GROUP_MAX];          // its workgroups: tiles_m * tiles_n * splitk
    int mode[GROUP_MAX];         // 2 * a_mc + b_mc (+ 4: the problem was planned onto the register-staged form of the tile, T128V next to T128R2 -- same geometry, same LDS)
    int n;
};

template <int BM_, int BN_, int WM_, int WN_, int STAGES_>
__global__ void __launch_bounds__(WM_ * WN_ * 64, (BM_ == 128 && STAGES_ == 2) ? 2 : 1) gemm_pipe_group_kernel(const GemmGroup g) {
    using TL = Tile<BM_, BN_, WM_, WN_, STAGES_, 64>;
    __shared__ __attribute__((aligned(1024))) char lds[TL::STAGES * TL::STAGE_BYTES];
    const int b = (int)blockIdx.x;
    int i = 0;
#pragma unroll
    for (int j = 1; j < GROUP_MAX; ++j)
        if (j < g.n && b >= g.start[j]) i = j;
    const int local = b - g.start[i];
    if (local >= g.nwg[i]) return;
    const GemmParams& p = g.p[i];
    if constexpr (BM_ == 128 && STAGES_ == 2) {      // T128R2's group also carries problems planned onto T128V (K-contiguous A only)
        if (g.mode[i] == 4) { gemm_pipe_body<BM_, BN_, WM_, WN_, STAGES_, false, false, 0, 64, T128V::VS>(p, lds, local, 0); return; }
        if (g.mode[i] == 5) { gemm_pipe_body<BM_, BN_, WM_, WN_, STAGES_, false, true, 0, 64, T128V::VS>(p, lds, local, 0); return; }
    }
    switch (g.mode[i]) {
    case 0: gemm_pipe_body<BM_, BN_, WM_, WN_, STAGES_, false, false>(p, lds, local, 0); break;
    case 1: gemm_pipe_body<BM_, BN_, WM_, WN_, STAGES_, false, true>(p, lds, local, 0); break;
    case 2: gemm_pipe_body<BM_, BN_, WM_, WN_, STAGES_, true, false>(p, lds, local, 0); break;
    default: gemm_pipe_body<BM_, BN_, WM_, WN_, STAGES_, true, true>(p, lds, local, 0); break;
    }
}

template <typename TL>
int launch_pipe_group(const GemmGroup& g, int total_wg, hipStream_t s) {
    gemm_pipe_group_kernel<TL::BM, TL::BN, TL::WM, TL::WN, TL::STAGES><<<dim3((unsigned)total_wg), TL::NT, 0, s>>>(g);
    return check_launch("dpipe_gemm_group");
}

template <typename TL, int CONV = 0>
int launch_pipe(const GemmParams& p, bool a_mc, bool b_mc, int batch, hipStream_t s) {
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n * p.splitk), (unsigned)batch);
#define DPIPE_PIPE_LAUNCH(AM, BMC) gemm_pipe_kernel<TL::BM, TL::BN, TL::WM, TL::WN, TL::STAGES, AM, BMC, CONV, TL::BK, TL::VS><<<grid, TL::NT, 0, s>>>(p)
    if constexpr (CONV == 1) {          // A rows gathered: A is K-contiguous; B = weight K-contiguous (forward) or MN-contiguous (dgrad)
        if (b_mc) DPIPE_PIPE_LAUNCH(false, true); else DPIPE_PIPE_LAUNCH(false, false);
        return check_launch("dpipe_conv2d");
    }
    if constexpr (CONV == 2) { DPIPE_PIPE_LAUNCH(true, true); return check_launch("dpipe_conv2d_wgrad"); }
    if (!a_mc && !b_mc) DPIPE_PIPE_LAUNCH(false, false);
    else if (!a_mc && b_mc) DPIPE_PIPE_LAUNCH(false, true);
    else if (a_mc && b_mc) DPIPE_PIPE_LAUNCH(true, true);
    else DPIPE_PIPE_LAUNCH(true, false);
#undef DPIPE_PIPE_LAUNCH
    return check_launch("dpipe_gemm(pipe)");
}

}  // namespace dpipe_pipe
