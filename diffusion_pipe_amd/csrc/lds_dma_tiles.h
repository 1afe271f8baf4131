// lds_dma_tiles.h -- the two LDS image formats of operand tiles streamed global -> LDS with `buffer_load_dwordx4 ... lds`, shared by the
// pipelined GEMM / implicit-GEMM convolution (gemm_pipe_kernel.h) and the flash-attention forward (attention.hip).
//
// The DMA destination is lane-linear (lane L of 1 KiB piece q lands at LDS byte q * 1024 + 16 L), so a bank-conflict-free layout is made by
// permuting the per-lane SOURCE address:
//   * K-contiguous image [ROWS mn-rows][128 B = 64 k]: 16-byte chunk ^= (row >> 1) & 7, read with ds_read_b128 (one MFMA fragment per read);
//   * MN-contiguous image [64 k-rows][ROWS * 2 B]: 64-byte granule ^= f(k-row), read with ds_read_b64_tr_b16 (hardware transpose) -- the
//     operand is consumed transposed without a transposed copy ever existing.
#pragma once
#include "dpipe_common.h"

namespace dpipe_tiles {
using namespace dpipe;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) bf16x4_t lds_bf16x4_t;

// Byte offset (from the operand base of this batch) of the 16 bytes lane `lane` fetches for DMA piece q of an image of
// ROWS mn-rows at K-step 0.  The DMA writes lane L of piece q to LDS byte (q * 1024 + 16 L); the logical chunk fetched
// is the inverse of the read-side swizzle.
template <bool MC, int ROWS, int KT = 64>
__device__ __forceinline__ unsigned dma_voffset(int q, int lane, int mn0, long ld) {
    static_assert(KT == 64 || KT == 32, "k extent of one image");
    if (!MC && KT == 64) {   // image [ROWS mn-rows][128 B]: physical 16-B chunk pc of row holds logical chunk pc ^ ((row >> 1) & 7)
        const int row = 8 * q + (lane >> 3);
        const int pc = lane & 7;
        const int lc = pc ^ ((row >> 1) & 7);
        return (unsigned)(((long)(mn0 + row) * ld + lc * 8) * 2);
    } else if (!MC) {        // KT = 32: image [ROWS mn-rows][64 B] (a piece = 16 rows): rows r, r + 4, r + 8, r + 12 share a bank group -> chunk ^= (row >> 2) & 3
        const int row = 16 * q + (lane >> 2);
        const int pc = lane & 3;
        const int lc = pc ^ ((row >> 2) & 3);
        return (unsigned)(((long)(mn0 + row) * ld + lc * 8) * 2);
    } else {                 // (KT k-rows: the same format, KT / 64 as many pieces)     // image [64 k-rows][ROWS * 2 B]: physical 64-B granule pg of a k-row holds logical granule pg ^ f(krow)
        constexpr int CPR = ROWS / 8;            // 16-B chunks per k-row (16 or 8)
        constexpr int G = ROWS / 32;             // 64-B granules per k-row (4 or 2)
        const int krow = q * (64 / CPR) + lane / CPR;
        const int pc = lane % CPR;
        const int f = G >= 4 ? (krow & 3) : ((krow >> 1) & 1);
        const int lc = (((pc >> 2) ^ f) << 2) | (pc & 3);
        return (unsigned)(((long)krow * ld + mn0 + lc * 8) * 2);
    }
}

// MFMA operand fragment (32 mn-rows x 16 k): lane (i = lane & 31, h = lane >> 5) gets k = 16 ks + 8 h .. + 8 of row mn + i.
template <bool MC, int ROWS, int KT = 64>
__device__ __forceinline__ bf16x8_t read_frag(const char* img, int mn, int ks, int lane) {
    if (!MC && KT == 32) {
        const int l31 = lane & 31;
        const int x = (2 * ks + (lane >> 5)) ^ ((l31 >> 2) & 3);
        return *reinterpret_cast<const bf16x8_t*>(img + mn * 64 + l31 * 64 + (x << 4));
    } else if (!MC) {
        // mn is a multiple of 32, so the swizzle term (row >> 1) & 7 depends on the lane only and 2 ks + h == (2 ks) ^ h:
        // the lane part of the address is one of 4 values (per ks) shared by every fragment of both operands; mn * 128
        // is wave-uniform / an immediate offset
        const int l31 = lane & 31;
        const int x = (lane >> 5) ^ ((l31 >> 1) & 7);
        const int lane_off = l31 * 128 + ((x ^ (2 * ks)) << 4);
        return *reinterpret_cast<const bf16x8_t*>(img + mn * 128 + lane_off);
    } else {
        // ds_read_b64_tr_b16: in a 16-lane group lane t supplies the address of 4 contiguous bf16 of k-row (t >> 2) at
        // columns 4 (t & 3) of a [4][16] block and receives column t of it (the 4 k values of one mn index).
        constexpr int RB = ROWS * 2, G = ROWS / 32;
        const int t = lane & 15, g = lane >> 4;
        const int krow = 16 * ks + 8 * (g >> 1) + (t >> 2);
        const int col_b = (16 * (g & 1) + 4 * (t & 3)) * 2;          // byte column inside the 64-B granule
        const int f = G >= 4 ? (krow & 3) : ((krow >> 1) & 1);       // identical for krow + 4
        const char* p = img + krow * RB + ((mn >> 5) ^ f) * 64 + col_b;
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(p));
        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(p + 4 * RB));
        bf16x8_t out;
        out[0] = lo[0]; out[1] = lo[1]; out[2] = lo[2]; out[3] = lo[3];
        out[4] = hi[0]; out[5] = hi[1]; out[6] = hi[2]; out[7] = hi[3];
        return out;
    }
}


// The same MN-contiguous image read for an MFMA whose k index must follow the ROW SET a lane holds in a 32 x 32 fp32 accumulator
// (entries 8u .. 8u + 7 of lane (i, h) are rows {4h + 0..3, 8 + 4h + 0..3} of the 16-row block u): k-slot j of lane half h <-> k-row
// r0 + 4h + j (j < 4), r0 + 8 + 4h + (j - 4).  With it a score tile can be fed back as the B operand straight from its accumulator
// registers (flash attention: O^T += V^T . P^T).  Same bank behaviour as read_frag<true>: the two half-waves differ by 4 (not 8) k-rows.
template <int ROWS>
__device__ __forceinline__ bf16x8_t read_frag_tr_acc(const char* img, int mn, int r0, int lane) {
    constexpr int RB = ROWS * 2, G = ROWS / 32;
    const int t = lane & 15, g = lane >> 4;
    const int krow = r0 + 4 * (g >> 1) + (t >> 2);
    const int col_b = (16 * (g & 1) + 4 * (t & 3)) * 2;
    const int f = G >= 4 ? (krow & 3) : ((krow >> 1) & 1);           // identical for krow + 8
    const char* p = img + krow * RB + ((mn >> 5) ^ f) * 64 + col_b;
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(p));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(p + 8 * RB));
    bf16x8_t out;
    out[0] = lo[0]; out[1] = lo[1]; out[2] = lo[2]; out[3] = lo[3];
    out[4] = hi[0]; out[5] = hi[1]; out[6] = hi[2]; out[7] = hi[3];
    return out;
}

}  // namespace dpipe_tiles
