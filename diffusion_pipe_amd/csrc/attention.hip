// attention.hip -- K4: flash-style scaled-dot-product attention, forward + backward, bf16 on the CDNA4 matrix cores.
//
// Replaces flash_attn_varlen_func / SDPA of models/wan/attention.py:91-122,159-174, the diffusers attention
// processors behind models/sdxl.py:797-865 and the joint attention of utils/patches.py:325-340.
//
// Layout: q [B, Sq, H, D], k/v [B, Sk, H, D], o [B, Sq, H, D] (head dim contiguous, D in {64, 128});
// lse [B, H, Sq] fp32 (natural log).  Non-causal; optional per-batch valid key count.
//
// MI355X design.  Everything is built from v_mfma_f32_32x32x16_bf16 with the *transposed* score tile
// S^T = K . Q^T, so that each lane owns one query column of the 32x32 accumulator: running max / sum and the
// rescale of O^T are lane-local (one cross-lane exchange with lane^32 per row reduction), and the exponentiated
// tile is already in MFMA B-operand order for O^T += V^T . P^T -- no LDS round trip, no permutes.  V^T (and K^T,
// Q^T, dO^T in the backward) fragments come from row-major LDS tiles through the hardware transpose read
// ds_read_b64_tr_b16.  K/V tiles are staged global -> registers -> LDS with 16-byte accesses, the next tile's
// loads being issued before the current tile's MFMAs.  Backward = three kernels without atomics (deterministic):
// delta = rowsum(dO * O); a query-outer kernel for dQ; a key-outer kernel for dK and dV.
#include <cstdlib>
#include "dpipe_common.h"
#include "lds_dma_tiles.h"
#include "../../include/dpipe_hip.h"

using namespace dpipe;
using namespace dpipe_tiles;

namespace {

struct AttnParams {
    const bf16_t *q, *k, *v, *o, *dout;
    bf16_t *out, *dq, *dk, *dv;
    float* lse; float* delta; const int* kv_len;
    int B, H, Sq, Sk;
    long q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh, o_sb, o_ss, o_sh;
    long do_sb, do_ss, do_sh, dq_sb, dq_ss, dq_sh, dk_sb, dk_ss, dk_sh, dv_sb, dv_ss, dv_sh;
    float scale;
    int causal;
    int qsplit;            // dK/dV kernel: query tiles are split over `qsplit` workgroups per key block (short-key cross attention)
    float* part;           // fp32 partials [B][H][qsplit][2][Sk][D] when qsplit > 1
    float* out32;          // forward, optional: O once more in fp32, [B][Sq][H][D] contiguous (unrounded accumulator / l) -- kept for the backward's delta
    const float* o32;      // backward, optional: that tensor; delta = rowsum(dO . O) then uses it instead of the bf16 O
};

// delta = rowsum(dO . O) is subtracted from dP = dO . V^T element by element.  When the value rows share a large common component (V_j = c + v_j: activations
// behind a LayerNorm usually do) both are ~ dO . c and the difference is what matters: the 2^-9 rounding of a bf16 O then lands on dS, dQ and dK amplified by
// |c| / |v_j| (measured on the full-size SDXL step: 8 % of sum|g| on the to_q / to_k weights of the deepest self-attentions; a host simulation with |c| = 3 |v_j|:
// dQ 25 % off with the bf16 O, 1.4 % with the fp32 one).  The forward therefore stores O a second time in fp32 for the backward (optional: NULL = bf16 O).

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ f32x16 mfma16(bf16x8_t a, bf16x8_t b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_mfma, a), __builtin_bit_cast(bf16x8_mfma, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    return z;
}
__device__ __forceinline__ bf16x8_t zero_frag() {
    bf16x8_t z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = 0;
    return z;
}
// 8 consecutive accumulator entries (e0 .. e0+7) -> bf16 MFMA operand
__device__ __forceinline__ bf16x8_t pack_frag(const f32x16& s, int e0) {
    uint4 u;
    u.x = pack_bf16x2(s[e0 + 0], s[e0 + 1]); u.y = pack_bf16x2(s[e0 + 2], s[e0 + 3]);
    u.z = pack_bf16x2(s[e0 + 4], s[e0 + 5]); u.w = pack_bf16x2(s[e0 + 6], s[e0 + 7]);
    return __builtin_bit_cast(bf16x8_t, u);
}

// ---- LDS tiles: `ROWS` rows of D bf16, row stride RS bytes ------------------------------------------------------
// A/B fragment with the K index along the row (contiguous): lane (i, h) reads 8 bf16 at (row0 + i, col + 8h).
template <int RS>
__device__ __forceinline__ bf16x8_t frag_rows(const char* tile, int row0, int col, int lane) {
    return *reinterpret_cast<const bf16x8_t*>(tile + (row0 + (lane & 31)) * RS + (col + 8 * (lane >> 5)) * 2);
}
// A fragment of the TRANSPOSED tile: MFMA row = tile column (col0 + i), MFMA k = tile rows
// {r0 + 4h + 0..3, r0 + 8 + 4h + 0..3} -- exactly the row set a lane holds in entries 8u..8u+7 of a 32x32 accumulator.
template <int RS>
__device__ __forceinline__ bf16x8_t frag_cols_tr(const char* tile, int r0, int col0, int lane) {
    const int t = lane & 15, g = lane >> 4;
    const char* p = tile + (r0 + 4 * (g >> 1) + (t >> 2)) * RS + (col0 + 16 * (g & 1) + 4 * (t & 3)) * 2;
    bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(p));
    bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(p + 8 * RS));
    bf16x8_t out;
    out[0] = lo[0]; out[1] = lo[1]; out[2] = lo[2]; out[3] = lo[3];
    out[4] = hi[0]; out[5] = hi[1]; out[6] = hi[2]; out[7] = hi[3];
    return out;
}

template <int N> struct RowRegs { uint4 v[N]; };

// Stage a [ROWS][D] tile: rows row0.. of a [*, D] matrix with row stride `ss` elements; rows >= limit read as zero.
template <int D, int ROWS, int NT>
__device__ __forceinline__ void load_rows(RowRegs<ROWS * (D / 8) / NT>& r, const bf16_t* __restrict__ base, long ss, int row0, int limit) {
    constexpr int CPR = D / 8;
#pragma unroll
    for (int i = 0; i < ROWS * CPR / NT; ++i) {
        const int v = threadIdx.x + i * NT;
        const int row = v / CPR, c = v % CPR;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (row0 + row < limit) val = *reinterpret_cast<const uint4*>(base + (long)(row0 + row) * ss + c * 8);
        r.v[i] = val;
    }
}
template <int D, int ROWS, int NT, int RS>
__device__ __forceinline__ void store_rows(const RowRegs<ROWS * (D / 8) / NT>& r, char* tile) {
    constexpr int CPR = D / 8;
#pragma unroll
    for (int i = 0; i < ROWS * CPR / NT; ++i) {
        const int v = threadIdx.x + i * NT;
        *reinterpret_cast<uint4*>(tile + (v / CPR) * RS + (v % CPR) * 16) = r.v[i];
    }
}

// ================================================================================================ forward
template <int D, int NW>
__global__ void __launch_bounds__(NW * 64, D == 64 ? 2 : 1) attn_fwd_kernel(const AttnParams p) {
    constexpr int NT = NW * 64, KT = 64;
    constexpr int KRS = 2 * D + 16;   // K tile: row reads (ds_read_b128), conflict-free with a 16 B pad
    constexpr int VRS = 2 * D + 64;   // V tile: transpose reads, 4 rows x 64 B windows tile the 256 B bank row
    __shared__ __attribute__((aligned(16))) char lds[KT * KRS + KT * VRS];
    char* Kl = lds; char* Vl = lds + KT * KRS;

    const int b = blockIdx.z, hh = blockIdx.y;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const int qrow = blockIdx.x * 32 * NW + wid * 32 + i;
    const int kvl = p.kv_len ? min(p.kv_len[b], p.Sk) : p.Sk;
    const bf16_t* Q = p.q + b * p.q_sb + hh * p.q_sh;
    const bf16_t* K = p.k + b * p.k_sb + hh * p.k_sh;
    const bf16_t* V = p.v + b * p.v_sb + hh * p.v_sh;
    const float sl2 = p.scale * LOG2E;

    bf16x8_t qf[D / 16];
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks)
        qf[ks] = qrow < p.Sq ? *reinterpret_cast<const bf16x8_t*>(Q + (long)qrow * p.q_ss + 16 * ks + 8 * h) : zero_frag();

    f32x16 oacc[D / 32];
#pragma unroll
    for (int db = 0; db < D / 32; ++db) oacc[db] = zero16();
    float m = -INFINITY, l = 0.f;

    const int nkt = (kvl + KT - 1) / KT;
    RowRegs<KT * (D / 8) / NT> rk, rv;
    load_rows<D, KT, NT>(rk, K, p.k_ss, 0, kvl);
    load_rows<D, KT, NT>(rv, V, p.v_ss, 0, kvl);

    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        store_rows<D, KT, NT, KRS>(rk, Kl);
        store_rows<D, KT, NT, VRS>(rv, Vl);
        __syncthreads();
        if (kt + 1 < nkt) {
            load_rows<D, KT, NT>(rk, K, p.k_ss, (kt + 1) * KT, kvl);
            load_rows<D, KT, NT>(rv, V, p.v_ss, (kt + 1) * KT, kvl);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int key0 = kt * KT + 32 * kb;
            if (key0 >= kvl) break;                     // wave-uniform: nothing valid in this 32-key block
            if (p.causal && key0 > qrow - i + 31) break; // whole block above the diagonal for every row of the wave
            f32x16 s = zero16();
#pragma unroll
            for (int ks = 0; ks < D / 16; ++ks) s = mfma16(frag_rows<KRS>(Kl, 32 * kb, 16 * ks, lane), qf[ks], s);
            float mx = -INFINITY;
            // wave-uniform: every key of this 32-key block is valid for every query row of the wave -> no per-element masking
            const bool full = key0 + 32 <= kvl && (!p.causal || key0 + 31 <= qrow - i);
            if (full) {
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float v = s[e] * sl2; s[e] = v; mx = fmaxf(mx, v); }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const bool ok = key < kvl && (!p.causal || key <= qrow);
                    const float v = ok ? s[e] * sl2 : -INFINITY;
                    s[e] = v; mx = fmaxf(mx, v);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m, mx);              // finite: key0 < kvl guarantees one valid key per row
            float rs = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float pe = __builtin_amdgcn_exp2f(s[e] - m_new); s[e] = pe; rs += pe; }
            rs += __shfl_xor(rs, 32, 64);
            if (__all(m_new == m)) {                       // no row of the wave raised its maximum: alpha == 1 exactly, skip the rescale
                l += rs;
            } else {
                const float alpha = __builtin_amdgcn_exp2f(m - m_new);
                l = l * alpha + rs; m = m_new;
#pragma unroll
                for (int db = 0; db < D / 32; ++db)
#pragma unroll
                    for (int e = 0; e < 16; ++e) oacc[db][e] *= alpha;
            }
            const bf16x8_t pf0 = pack_frag(s, 0), pf1 = pack_frag(s, 8);
#pragma unroll
            for (int db = 0; db < D / 32; ++db) {
                oacc[db] = mfma16(frag_cols_tr<VRS>(Vl, 32 * kb, 32 * db, lane), pf0, oacc[db]);
                oacc[db] = mfma16(frag_cols_tr<VRS>(Vl, 32 * kb + 16, 32 * db, lane), pf1, oacc[db]);
            }
        }
    }

    if (qrow < p.Sq) {
        const float inv = l > 0.f ? 1.f / l : 0.f;          // a sample without valid keys (kv_len[b] == 0): O = 0, lse = +inf -> P = exp(s - lse) = 0 in the backward kernels
        bf16_t* O = p.out + b * p.o_sb + hh * p.o_sh + (long)qrow * p.o_ss;
#pragma unroll
        for (int db = 0; db < D / 32; ++db)
#pragma unroll
            for (int eg = 0; eg < 4; ++eg) {
                uint2 w;
                w.x = pack_bf16x2(oacc[db][4 * eg] * inv, oacc[db][4 * eg + 1] * inv);
                w.y = pack_bf16x2(oacc[db][4 * eg + 2] * inv, oacc[db][4 * eg + 3] * inv);
                *reinterpret_cast<uint2*>(O + 32 * db + 8 * eg + 4 * h) = w;
            }
        if (p.out32) {
            float* O32 = p.out32 + (((long)b * p.Sq + qrow) * p.H + hh) * D;
#pragma unroll
            for (int db = 0; db < D / 32; ++db)
#pragma unroll
                for (int eg = 0; eg < 4; ++eg)
                    *reinterpret_cast<float4*>(O32 + 32 * db + 8 * eg + 4 * h) =
                        make_float4(oacc[db][4 * eg] * inv, oacc[db][4 * eg + 1] * inv, oacc[db][4 * eg + 2] * inv, oacc[db][4 * eg + 3] * inv);
        }
        if (h == 0) p.lse[((long)b * p.H + hh) * p.Sq + qrow] = l > 0.f ? (m + __builtin_amdgcn_logf(l)) * LN2 : INFINITY;   // v_log_f32 = log2
    }
}

// ================================================================================================ forward, long sequences
// 64 query rows per wave (two 32-column score blocks sharing every K / V fragment read: the 32-row kernel above reads 32 KiB of LDS per
// wave per 64-key tile for 32 MFMAs = the CU's whole 128 B/clk once four waves run), 256 per workgroup; K / V tiles go global -> LDS by
// `buffer_load_dwordx4 ... lds` into a two-deep ring in the GEMM's two image formats (lds_dma_tiles.h: K as K-contiguous images read with
// ds_read_b128, V as an MN-contiguous image read transposed with ds_read_b64_tr_b16), one barrier per tile, the next tile's DMA in flight
// under the current tile's MFMAs -- no staging VGPRs, no ds_write pass.  Keys past kv_len come back as zeros from the buffer bounds check.
// Within a tile both 32-key blocks' score MFMAs are issued before the first softmax, so exponentials overlap matrix work of the same wave.
// (Round 5 negative results, profiles/r5b_attention_ring_qb_negative.jsonl, head dim 64, in-graph us forward / backward, S = 1024 x 20 heads and S = 4096 x 10 heads:
//  two-deep ring (this kernel) 26.1 / 76.3 and 110.2 / 267.8; a THREE-deep ring with counted vmcnt in the forward and dQ kernels 27.5 / 75.8 and 114.9 / 269.3; the same in
//  the one-pass dK / dV kernel (99 KiB: one workgroup per CU) 83.4 and 450.9 backward; 64 query rows per wave (QB = 2: K / V fragments shared by two score blocks, 202 VGPRs)
//  as 4-wave 256-row workgroups 39.0 and 126.6 forward, as 2-wave 128-row workgroups 38.2 and 183.5.  The vmcnt(0) wait of the two-deep ring is not DMA latency a deeper
//  ring could hide (a tile's 16 KiB land well inside one tile of compute; the wait share the PMC pass saw is then the per-tile barrier itself: unmeasured), and
//  halving the LDS reads per MFMA costs more in occupancy than it saves.  All variants removed again.)
template <int D, int QB, int NW>
__global__ void __launch_bounds__(NW * 64, (D == 64 && NW == 4) ? 2 : 1) attn_fwd_dma_kernel(const AttnParams p) {
    constexpr int QW = 32 * QB, QWG = QW * NW;          // query rows per wave / per workgroup
    constexpr int KT = 64, NKS = D / 16, NDB = D / 32;
    constexpr int IMG = KT * D * 2, STAGE = 2 * IMG;      // K image(s) then V image
    constexpr int PP = IMG / 1024 / NW;                   // 1 KiB DMA pieces per wave per operand per tile
    __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];

    // XCD-aware order (consecutive ids round-robin over the 8 XCDs): an XCD works through a contiguous run of (head, query block) pairs, so the
    // query blocks of one head -- which stream the same K / V -- share one L2
    const int nq = (p.Sq + QWG - 1) / QWG;
    const int total = nq * p.H * p.B;
    const int orig = blockIdx.x, xq = total / 8, xr = total % 8, xcd = orig % 8;
    const int L = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + orig / 8;
    const int qblk = L % nq, hh = (L / nq) % p.H, b = L / (nq * p.H);

    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = lane & 31, h = lane >> 5;
    const int q0w = qblk * QWG + wid * QW;                // first query row of this wave
    const int kvl = p.kv_len ? min(p.kv_len[b], p.Sk) : p.Sk;
    const bf16_t* Q = p.q + b * p.q_sb + hh * p.q_sh;
    const bf16_t* K = p.k + b * p.k_sb + hh * p.k_sh;
    const bf16_t* V = p.v + b * p.v_sb + hh * p.v_sh;
    const float sl2 = p.scale * LOG2E;

    bf16x8_t qf[QB][NKS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = q0w + 32 * qb + i;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            qf[qb][ks] = qrow < p.Sq ? *reinterpret_cast<const bf16x8_t*>(Q + (long)qrow * p.q_ss + 16 * ks + 8 * h) : zero_frag();
    }
    f32x16 oacc[QB][NDB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int db = 0; db < NDB; ++db) oacc[qb][db] = zero16();
    float m[QB], l[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { m[qb] = -INFINITY; l[qb] = 0.f; }

    int nkt = (kvl + KT - 1) / KT;
    if (p.causal) nkt = min(nkt, min(qblk * QWG + QWG - 1, p.Sq - 1) / KT + 1);      // tiles at or below the diagonal of the workgroup's last row

    // rows >= kvl lie beyond the extent (row offsets grow with the row): the bounds check zero-fills them
    const unsigned k_bytes = kvl > 0 ? (unsigned)(((long)(kvl - 1) * p.k_ss + D) * 2) : 0u;
    const unsigned v_bytes = kvl > 0 ? (unsigned)(((long)(kvl - 1) * p.v_ss + D) * 2) : 0u;
    const auto rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(K), (short)0, (int)k_bytes, 0x00020000);
    const auto rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(V), (short)0, (int)v_bytes, 0x00020000);
    unsigned voK[PP], voV[PP];
#pragma unroll
    for (int j = 0; j < PP; ++j) {
        const int P = j * NW + wid;                         // piece P of the tile: K slice P / 8 (64 head-dim columns each), rows 8 (P % 8) ..
        voK[j] = dma_voffset<false, KT>(P % 8, lane, 0, p.k_ss) + (unsigned)(P / 8) * 128u;
        voV[j] = dma_voffset<true, D>(P, lane, 0, p.v_ss);
    }
    const unsigned stepK = (unsigned)(KT * p.k_ss * 2), stepV = (unsigned)(KT * p.v_ss * 2);
    // (the DMA builtin must only see non-dependent operands: see gemm_pipe_kernel.h)
#define ISSUE_TILE(buf)                                                                                                   \
    do {                                                                                                                  \
        char* base_ = lds + (buf) * STAGE + wid * 1024;                                                                   \
        _Pragma("unroll") for (int j = 0; j < PP; ++j) {                                                                  \
            char* dk_ = base_ + j * NW * 1024;                                                                                 \
            const unsigned ok_ = voK[j]; voK[j] += stepK;                                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_void_t*)dk_, 16, ok_, 0, 0, 0);                            \
        }                                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < PP; ++j) {                                                                  \
            char* dv_ = base_ + IMG + j * NW * 1024;                                                                           \
            const unsigned ov_ = voV[j]; voV[j] += stepV;                                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_void_t*)dv_, 16, ov_, 0, 0, 0);                            \
        }                                                                                                                 \
    } while (0)

    if (nkt > 0) ISSUE_TILE(0);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile kt (the only DMA in flight here)
        __builtin_amdgcn_s_barrier();                      // every wave's pieces landed; everyone is done reading the other buffer
        if (kt + 1 < nkt) ISSUE_TILE((kt + 1) & 1);
        const char* Kl = lds + (kt & 1) * STAGE;
        const char* Vl = Kl + IMG;
        const int key00 = kt * KT;
        if (p.causal && key00 > q0w + QW - 1) continue;        // wave-uniform: the whole tile is above the diagonal for every row of the wave

        // scores of a 32-key block x the wave's query blocks (K fragments shared by the query blocks)
#define SCORES(kb)                                                                                                        \
        do {                                                                                                              \
            _Pragma("unroll") for (int qb = 0; qb < QB; ++qb) sc[qb] = zero16();                                      \
            _Pragma("unroll") for (int ks = 0; ks < NKS; ++ks) {                                                          \
                const bf16x8_t kf = read_frag<false, KT>(Kl + (ks / 4) * (KT * 128), 32 * (kb), ks & 3, lane);            \
                _Pragma("unroll") for (int qb = 0; qb < QB; ++qb) sc[qb] = mfma16(kf, qf[qb][ks], sc[qb]);        \
            }                                                                                                             \
        } while (0)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int key0 = key00 + 32 * kb;
            if (key0 >= kvl) break;                        // workgroup-uniform: nothing valid in this block
            if (p.causal && key0 > q0w + QW - 1) break;
            f32x16 sc[QB];
            SCORES(kb);
            bf16x8_t pf[QB][2];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                f32x16& s = sc[qb];
                const int qrow = q0w + 32 * qb + i;
                float mx = -INFINITY;
                const bool full = key0 + 32 <= kvl && (!p.causal || key0 + 31 <= q0w + 32 * qb);
                if (full) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) { const float v = s[e] * sl2; s[e] = v; mx = fmaxf(mx, v); }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                        const bool ok = key < kvl && (!p.causal || key <= qrow);
                        const float v = ok ? s[e] * sl2 : -INFINITY;
                        s[e] = v; mx = fmaxf(mx, v);
                    }
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m[qb], mx);      // finite from the first block on (key 0 is valid for every row)
                float rs = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float pe = __builtin_amdgcn_exp2f(s[e] - m_new); s[e] = pe; rs += pe; }
                rs += __shfl_xor(rs, 32, 64);
                if (__all(m_new == m[qb])) {
                    l[qb] += rs;
                } else {
                    const float alpha = __builtin_amdgcn_exp2f(m[qb] - m_new);
                    l[qb] = l[qb] * alpha + rs; m[qb] = m_new;
#pragma unroll
                    for (int db = 0; db < NDB; ++db)
#pragma unroll
                        for (int e = 0; e < 16; ++e) oacc[qb][db][e] *= alpha;
                }
                pf[qb][0] = pack_frag(s, 0); pf[qb][1] = pack_frag(s, 8);
            }
            // ---- O^T += V^T . P^T (V fragments shared by the two query blocks)
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const bf16x8_t v0 = read_frag_tr_acc<D>(Vl, 32 * db, 32 * kb, lane);
                const bf16x8_t v1 = read_frag_tr_acc<D>(Vl, 32 * db, 32 * kb + 16, lane);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) oacc[qb][db] = mfma16(v0, pf[qb][0], oacc[qb][db]);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) oacc[qb][db] = mfma16(v1, pf[qb][1], oacc[qb][db]);
            }
        }
    }
#undef ISSUE_TILE
#undef SCORES

#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = q0w + 32 * qb + i;
        if (qrow < p.Sq) {
            const float inv = l[qb] > 0.f ? 1.f / l[qb] : 0.f;      // no valid key: O = 0, lse = +inf (see attn_fwd)
            bf16_t* O = p.out + b * p.o_sb + hh * p.o_sh + (long)qrow * p.o_ss;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int eg = 0; eg < 4; ++eg) {
                    uint2 w;
                    w.x = pack_bf16x2(oacc[qb][db][4 * eg] * inv, oacc[qb][db][4 * eg + 1] * inv);
                    w.y = pack_bf16x2(oacc[qb][db][4 * eg + 2] * inv, oacc[qb][db][4 * eg + 3] * inv);
                    *reinterpret_cast<uint2*>(O + 32 * db + 8 * eg + 4 * h) = w;
                }
            if (p.out32) {
                float* O32 = p.out32 + (((long)b * p.Sq + qrow) * p.H + hh) * D;
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int eg = 0; eg < 4; ++eg)
                        *reinterpret_cast<float4*>(O32 + 32 * db + 8 * eg + 4 * h) = make_float4(oacc[qb][db][4 * eg] * inv, oacc[qb][db][4 * eg + 1] * inv,
                                                                                                 oacc[qb][db][4 * eg + 2] * inv, oacc[qb][db][4 * eg + 3] * inv);
            }
            if (h == 0) p.lse[((long)b * p.H + hh) * p.Sq + qrow] = l[qb] > 0.f ? (m[qb] + __builtin_amdgcn_logf(l[qb])) * LN2 : INFINITY;
        }
    }
}

// ================================================================================================ backward: delta
// delta[b, h, q] = sum_d dO[b, q, h, d] * O[b, q, h, d]      16 lanes per row
template <int D>
__global__ void __launch_bounds__(256) attn_delta_kernel(const AttnParams p) {
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int sub = threadIdx.x & 15;
    const long total = (long)p.B * p.H * p.Sq;
    const bool live = row < total;
    const long r = live ? row : 0;
    const int q = (int)(r % p.Sq); const int hh = (int)((r / p.Sq) % p.H); const int b = (int)(r / ((long)p.Sq * p.H));
    const bf16_t* o = p.o + b * p.o_sb + hh * p.o_sh + (long)q * p.o_ss;
    const bf16_t* d = p.dout + b * p.do_sb + hh * p.do_sh + (long)q * p.do_ss;
    float acc = 0.f;
    constexpr int PER = D / 16;   // 8 (one uint4) or 4 (one uint2)
    if (p.o32) {                  // the forward's unrounded O (see AttnParams)
        const float* o32 = p.o32 + (((long)b * p.Sq + q) * p.H + hh) * D + sub * PER;
#pragma unroll
        for (int j = 0; j < PER; ++j) acc += o32[j] * bf16_to_f32(d[sub * PER + j]);
    } else if (PER == 8) {
        Vec16<bf16_t> vo, vd; vo.load(o + sub * 8); vd.load(d + sub * 8);
        float fo[8], fd[8]; vo.unpack(fo); vd.unpack(fd);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += fo[j] * fd[j];
    } else {
        const uint2 uo = *reinterpret_cast<const uint2*>(o + sub * 4), ud = *reinterpret_cast<const uint2*>(d + sub * 4);
        acc += __uint_as_float(uo.x << 16) * __uint_as_float(ud.x << 16) + __uint_as_float(uo.x & 0xffff0000u) * __uint_as_float(ud.x & 0xffff0000u);
        acc += __uint_as_float(uo.y << 16) * __uint_as_float(ud.y << 16) + __uint_as_float(uo.y & 0xffff0000u) * __uint_as_float(ud.y & 0xffff0000u);
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (live && sub == 0) p.delta[r] = acc;   // delta is [B, H, Sq] contiguous and r enumerates it in that order
}

// ================================================================================================ backward: dQ
// Query-outer.  dQ^T[d][q] += K^T[d][key] . dS^T[key][q],   dS^T = P^T o (dP^T - delta[q]),  dP^T = V . dO^T
template <int D, int NW>
__global__ void __launch_bounds__(NW * 64, D == 64 ? 2 : 1) attn_bwd_dq_kernel(const AttnParams p) {
    constexpr int NT = NW * 64, KT = 64;
    constexpr int KRS = 2 * D + 16, VRS = 2 * D + 16;
    __shared__ __attribute__((aligned(16))) char lds[KT * KRS + KT * VRS];
    char* Kl = lds; char* Vl = lds + KT * KRS;

    const int b = blockIdx.z, hh = blockIdx.y;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const int qrow = blockIdx.x * 32 * NW + wid * 32 + i;
    const bool qlive = qrow < p.Sq;
    const int kvl = p.kv_len ? min(p.kv_len[b], p.Sk) : p.Sk;
    const bf16_t* Q = p.q + b * p.q_sb + hh * p.q_sh;
    const bf16_t* K = p.k + b * p.k_sb + hh * p.k_sh;
    const bf16_t* V = p.v + b * p.v_sb + hh * p.v_sh;
    const bf16_t* DO = p.dout + b * p.do_sb + hh * p.do_sh;
    const float sl2 = p.scale * LOG2E;

    bf16x8_t qf[D / 16], dof[D / 16];
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
        qf[ks] = qlive ? *reinterpret_cast<const bf16x8_t*>(Q + (long)qrow * p.q_ss + 16 * ks + 8 * h) : zero_frag();
        dof[ks] = qlive ? *reinterpret_cast<const bf16x8_t*>(DO + (long)qrow * p.do_ss + 16 * ks + 8 * h) : zero_frag();
    }
    const long stat = ((long)b * p.H + hh) * p.Sq + (qlive ? qrow : 0);
    const float lse2 = qlive ? p.lse[stat] * LOG2E : INFINITY;
    const float dl = qlive ? p.delta[stat] : 0.f;

    f32x16 dqacc[D / 32];
#pragma unroll
    for (int db = 0; db < D / 32; ++db) dqacc[db] = zero16();

    const int nkt = (kvl + KT - 1) / KT;
    RowRegs<KT * (D / 8) / NT> rk, rv;
    load_rows<D, KT, NT>(rk, K, p.k_ss, 0, kvl);
    load_rows<D, KT, NT>(rv, V, p.v_ss, 0, kvl);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        store_rows<D, KT, NT, KRS>(rk, Kl);
        store_rows<D, KT, NT, VRS>(rv, Vl);
        __syncthreads();
        if (kt + 1 < nkt) {
            load_rows<D, KT, NT>(rk, K, p.k_ss, (kt + 1) * KT, kvl);
            load_rows<D, KT, NT>(rv, V, p.v_ss, (kt + 1) * KT, kvl);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int key0 = kt * KT + 32 * kb;
            if (key0 >= kvl) break;
            if (p.causal && key0 > qrow - i + 31) break;
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < D / 16; ++ks) {
                s = mfma16(frag_rows<KRS>(Kl, 32 * kb, 16 * ks, lane), qf[ks], s);
                dp = mfma16(frag_rows<VRS>(Vl, 32 * kb, 16 * ks, lane), dof[ks], dp);
            }
            if (key0 + 32 <= kvl && (!p.causal || key0 + 31 <= qrow - i)) {      // wave-uniform: block fully valid
#pragma unroll
                for (int e = 0; e < 16; ++e) s[e] = __builtin_amdgcn_exp2f(s[e] * sl2 - lse2) * (dp[e] - dl);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const bool ok = key < kvl && (!p.causal || key <= qrow);
                    const float pe = ok ? __builtin_amdgcn_exp2f(s[e] * sl2 - lse2) : 0.f;
                    s[e] = pe * (dp[e] - dl);
                }
            }
            const bf16x8_t ds0 = pack_frag(s, 0), ds1 = pack_frag(s, 8);
#pragma unroll
            for (int db = 0; db < D / 32; ++db) {
                dqacc[db] = mfma16(frag_cols_tr<KRS>(Kl, 32 * kb, 32 * db, lane), ds0, dqacc[db]);
                dqacc[db] = mfma16(frag_cols_tr<KRS>(Kl, 32 * kb + 16, 32 * db, lane), ds1, dqacc[db]);
            }
        }
    }
    if (qlive) {
        bf16_t* DQ = p.dq + b * p.dq_sb + hh * p.dq_sh + (long)qrow * p.dq_ss;
#pragma unroll
        for (int db = 0; db < D / 32; ++db)
#pragma unroll
            for (int eg = 0; eg < 4; ++eg) {
                uint2 w;
                w.x = pack_bf16x2(dqacc[db][4 * eg] * p.scale, dqacc[db][4 * eg + 1] * p.scale);
                w.y = pack_bf16x2(dqacc[db][4 * eg + 2] * p.scale, dqacc[db][4 * eg + 3] * p.scale);
                *reinterpret_cast<uint2*>(DQ + 32 * db + 8 * eg + 4 * h) = w;
            }
    }
}

// ================================================================================================ backward: dQ, LDS-DMA form
// Same computation as attn_bwd_dq_kernel, tiles streamed like the forward: per 64-key tile the K tile lands twice -- as K-contiguous image(s) for
// S^T = K . Q^T and as the MN-contiguous image whose transposed read feeds dQ^T += K^T . dS^T without bank conflicts -- plus V K-contiguous for
// dP^T = V . dO^T; two-deep ring (6 images), one barrier per tile.
template <int D, int NW>
__global__ void __launch_bounds__(NW * 64, (D == 64 && NW == 4) ? 2 : 1) attn_bwd_dq_dma_kernel(const AttnParams p) {
    constexpr int QWG = 32 * NW, KT = 64, NKS = D / 16, NDB = D / 32;
    constexpr int IMG = KT * D * 2, STAGE = 3 * IMG, PP = IMG / 1024 / NW;
    __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];

    const int nq = (p.Sq + QWG - 1) / QWG;
    const int total = nq * p.H * p.B;
    const int orig = blockIdx.x, xq = total / 8, xr = total % 8, xcd = orig % 8;
    const int L = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + orig / 8;
    const int qblk = L % nq, hh = (L / nq) % p.H, b = L / (nq * p.H);

    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = lane & 31, h = lane >> 5;
    const int q0w = qblk * QWG + wid * 32;
    const int qrow = q0w + i;
    const bool qlive = qrow < p.Sq;
    const int kvl = p.kv_len ? min(p.kv_len[b], p.Sk) : p.Sk;
    const bf16_t* Q = p.q + b * p.q_sb + hh * p.q_sh;
    const bf16_t* K = p.k + b * p.k_sb + hh * p.k_sh;
    const bf16_t* V = p.v + b * p.v_sb + hh * p.v_sh;
    const bf16_t* DO = p.dout + b * p.do_sb + hh * p.do_sh;
    const float sl2 = p.scale * LOG2E;

    bf16x8_t qf[NKS], dof[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        qf[ks] = qlive ? *reinterpret_cast<const bf16x8_t*>(Q + (long)qrow * p.q_ss + 16 * ks + 8 * h) : zero_frag();
        dof[ks] = qlive ? *reinterpret_cast<const bf16x8_t*>(DO + (long)qrow * p.do_ss + 16 * ks + 8 * h) : zero_frag();
    }
    const long stat = ((long)b * p.H + hh) * p.Sq + (qlive ? qrow : 0);
    const float lse2 = qlive ? p.lse[stat] * LOG2E : INFINITY;
    // delta = rowsum(dO * O), fused here (the lane pair (i, h) holds the whole dO row): one launch less per attention backward; written for the dK kernels
    float dl = 0.f;
    if (qlive && p.o32) {         // the forward's unrounded O (see AttnParams)
        const float* O32 = p.o32 + (((long)b * p.Sq + qrow) * p.H + hh) * D;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const float4 a = *reinterpret_cast<const float4*>(O32 + 16 * ks + 8 * h), c = *reinterpret_cast<const float4*>(O32 + 16 * ks + 8 * h + 4);
            const float of[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += of[e] * bf16_to_f32((bf16_t)dof[ks][e]);
        }
    } else if (qlive) {
        const bf16_t* O = p.o + b * p.o_sb + hh * p.o_sh + (long)qrow * p.o_ss;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const bf16x8_t of = *reinterpret_cast<const bf16x8_t*>(O + 16 * ks + 8 * h);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += bf16_to_f32((bf16_t)of[e]) * bf16_to_f32((bf16_t)dof[ks][e]);
        }
    }
    dl += __shfl_xor(dl, 32, 64);
    if (qlive && h == 0) p.delta[stat] = dl;

    f32x16 dqacc[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) dqacc[db] = zero16();

    int nkt = (kvl + KT - 1) / KT;
    if (p.causal) nkt = min(nkt, min(qblk * QWG + QWG - 1, p.Sq - 1) / KT + 1);

    const unsigned k_bytes = kvl > 0 ? (unsigned)(((long)(kvl - 1) * p.k_ss + D) * 2) : 0u;
    const unsigned v_bytes = kvl > 0 ? (unsigned)(((long)(kvl - 1) * p.v_ss + D) * 2) : 0u;
    const auto rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(K), (short)0, (int)k_bytes, 0x00020000);
    const auto rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(V), (short)0, (int)v_bytes, 0x00020000);
    unsigned voKc[PP], voKm[PP], voVc[PP];
#pragma unroll
    for (int j = 0; j < PP; ++j) {
        const int P = j * NW + wid;
        voKc[j] = dma_voffset<false, KT>(P % 8, lane, 0, p.k_ss) + (unsigned)(P / 8) * 128u;
        voKm[j] = dma_voffset<true, D>(P, lane, 0, p.k_ss);
        voVc[j] = dma_voffset<false, KT>(P % 8, lane, 0, p.v_ss) + (unsigned)(P / 8) * 128u;
    }
    const unsigned stepK = (unsigned)(KT * p.k_ss * 2), stepV = (unsigned)(KT * p.v_ss * 2);
#define ISSUE_TILE(buf)                                                                                                   \
    do {                                                                                                                  \
        char* base_ = lds + (buf) * STAGE + wid * 1024;                                                                   \
        _Pragma("unroll") for (int j = 0; j < PP; ++j) {                                                                  \
            char* d0_ = base_ + j * NW * 1024;                                                                            \
            char* d1_ = d0_ + IMG;                                                                                        \
            char* d2_ = d0_ + 2 * IMG;                                                                                    \
            const unsigned o0_ = voKc[j], o1_ = voKm[j], o2_ = voVc[j];                                                   \
            voKc[j] += stepK; voKm[j] += stepK; voVc[j] += stepV;                                                         \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_void_t*)d0_, 16, o0_, 0, 0, 0);                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_void_t*)d1_, 16, o1_, 0, 0, 0);                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_void_t*)d2_, 16, o2_, 0, 0, 0);                            \
        }                                                                                                                 \
    } while (0)

    if (nkt > 0) ISSUE_TILE(0);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nkt) ISSUE_TILE((kt + 1) & 1);
        const char* Kc = lds + (kt & 1) * STAGE;
        const char* Km = Kc + IMG;
        const char* Vc = Kc + 2 * IMG;
        if (p.causal && kt * KT > q0w + 31) continue;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int key0 = kt * KT + 32 * kb;
            if (key0 >= kvl) break;
            if (p.causal && key0 > q0w + 31) break;
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                s = mfma16(read_frag<false, KT>(Kc + (ks / 4) * (KT * 128), 32 * kb, ks & 3, lane), qf[ks], s);
                dp = mfma16(read_frag<false, KT>(Vc + (ks / 4) * (KT * 128), 32 * kb, ks & 3, lane), dof[ks], dp);
            }
            if (key0 + 32 <= kvl && (!p.causal || key0 + 31 <= q0w)) {      // wave-uniform: block fully valid
#pragma unroll
                for (int e = 0; e < 16; ++e) s[e] = __builtin_amdgcn_exp2f(s[e] * sl2 - lse2) * (dp[e] - dl);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const bool ok = key < kvl && (!p.causal || key <= qrow);
                    const float pe = ok ? __builtin_amdgcn_exp2f(s[e] * sl2 - lse2) : 0.f;
                    s[e] = pe * (dp[e] - dl);
                }
            }
            const bf16x8_t ds0 = pack_frag(s, 0), ds1 = pack_frag(s, 8);
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                dqacc[db] = mfma16(read_frag_tr_acc<D>(Km, 32 * db, 32 * kb, lane), ds0, dqacc[db]);
                dqacc[db] = mfma16(read_frag_tr_acc<D>(Km, 32 * db, 32 * kb + 16, lane), ds1, dqacc[db]);
            }
        }
    }
#undef ISSUE_TILE
    if (qlive) {
        bf16_t* DQ = p.dq + b * p.dq_sb + hh * p.dq_sh + (long)qrow * p.dq_ss;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int eg = 0; eg < 4; ++eg) {
                uint2 w;
                w.x = pack_bf16x2(dqacc[db][4 * eg] * p.scale, dqacc[db][4 * eg + 1] * p.scale);
                w.y = pack_bf16x2(dqacc[db][4 * eg + 2] * p.scale, dqacc[db][4 * eg + 3] * p.scale);
                *reinterpret_cast<uint2*>(DQ + 32 * db + 8 * eg + 4 * h) = w;
            }
    }
}

// ================================================================================================ backward: dK, dV
// Key-outer.  S[q][key] = Q . K^T (lane owns one key column);  dV^T[d][key] += dO^T[d][q] . P[q][key];
// dK^T[d][key] += Q^T[d][q] . dS[q][key]
// MODE 0: dK and dV in one pass.  MODE 1 / 2: dV only / dK only -- at head dim 128 the one-pass kernel holds K, V fragments and two accumulators
// (> 256 registers: one wave per SIMD); the two half kernels fit two waves per SIMD with 8 waves per workgroup, and recomputing S once more
// (+25 % MFMA work) costs less than the idle matrix pipe of the single resident wave.
template <int D, int NW, int MODE = 0>
__global__ void __launch_bounds__(NW * 64, D == 64 ? 2 : 1) attn_bwd_dkv_kernel(const AttnParams p) {
    constexpr bool DO_K = MODE != 1, DO_V = MODE != 2;
    constexpr int NT = NW * 64, QT = 64;
    constexpr int QRS = 2 * D + 16;
    __shared__ __attribute__((aligned(16))) char lds[2 * QT * QRS + 2 * QT * 4];
    char* Ql = lds; char* DOl = lds + QT * QRS;
    float* lse_l = reinterpret_cast<float*>(lds + 2 * QT * QRS);
    float* dl_l = lse_l + QT;

    const int b = blockIdx.z, hh = blockIdx.y;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const int kblock = blockIdx.x / p.qsplit, qs = blockIdx.x % p.qsplit;
    const int key = kblock * 32 * NW + wid * 32 + i;
    const int kvl = p.kv_len ? min(p.kv_len[b], p.Sk) : p.Sk;
    const bool klive = key < kvl;
    const bf16_t* Q = p.q + b * p.q_sb + hh * p.q_sh;
    const bf16_t* K = p.k + b * p.k_sb + hh * p.k_sh;
    const bf16_t* V = p.v + b * p.v_sb + hh * p.v_sh;
    const bf16_t* DO = p.dout + b * p.do_sb + hh * p.do_sh;
    const float* LSE = p.lse + ((long)b * p.H + hh) * p.Sq;
    const float* DL = p.delta + ((long)b * p.H + hh) * p.Sq;
    const float sl2 = p.scale * LOG2E;

    bf16x8_t kf[D / 16], vf[D / 16];
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
        kf[ks] = klive ? *reinterpret_cast<const bf16x8_t*>(K + (long)key * p.k_ss + 16 * ks + 8 * h) : zero_frag();
        vf[ks] = klive ? *reinterpret_cast<const bf16x8_t*>(V + (long)key * p.v_ss + 16 * ks + 8 * h) : zero_frag();
    }
    f32x16 dkacc[D / 32], dvacc[D / 32];
#pragma unroll
    for (int db = 0; db < D / 32; ++db) { dkacc[db] = zero16(); dvacc[db] = zero16(); }

    const int nqt_all = (p.Sq + QT - 1) / QT;
    const int per = (nqt_all + p.qsplit - 1) / p.qsplit;
    const int qt0 = qs * per, nqt = min(nqt_all, qt0 + per);      // this workgroup's query tiles [qt0, nqt)
    RowRegs<QT * (D / 8) / NT> rq, rdo;
    load_rows<D, QT, NT>(rq, Q, p.q_ss, qt0 * QT, p.Sq);
    load_rows<D, QT, NT>(rdo, DO, p.do_ss, qt0 * QT, p.Sq);
    float r_lse = 0.f, r_dl = 0.f;
    if (threadIdx.x < QT) {
        const int q = qt0 * QT + threadIdx.x;
        r_lse = q < p.Sq ? LSE[q] * LOG2E : INFINITY; r_dl = q < p.Sq ? DL[q] : 0.f;
    }
    for (int qt = qt0; qt < nqt; ++qt) {
        __syncthreads();
        store_rows<D, QT, NT, QRS>(rq, Ql);
        store_rows<D, QT, NT, QRS>(rdo, DOl);
        if (threadIdx.x < QT) { lse_l[threadIdx.x] = r_lse; dl_l[threadIdx.x] = r_dl; }
        __syncthreads();
        if (qt + 1 < nqt) {
            load_rows<D, QT, NT>(rq, Q, p.q_ss, (qt + 1) * QT, p.Sq);
            load_rows<D, QT, NT>(rdo, DO, p.do_ss, (qt + 1) * QT, p.Sq);
            if (threadIdx.x < QT) {
                const int q = (qt + 1) * QT + threadIdx.x;
                r_lse = q < p.Sq ? LSE[q] * LOG2E : INFINITY; r_dl = q < p.Sq ? DL[q] : 0.f;
            }
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (qt * QT + 32 * qb >= p.Sq) break;
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < D / 16; ++ks) {
                s = mfma16(frag_rows<QRS>(Ql, 32 * qb, 16 * ks, lane), kf[ks], s);
                if constexpr (DO_K) dp = mfma16(frag_rows<QRS>(DOl, 32 * qb, 16 * ks, lane), vf[ks], dp);
            }
            f32x16 pr;
            // wave-uniform: all 32 keys of the wave are valid (padding queries carry lse = +inf -> p = 0) and nothing is causal
            const bool full = !p.causal && (key - i) + 32 <= kvl;
#pragma unroll
            for (int eg = 0; eg < 4; ++eg) {
                const float4 l4 = *reinterpret_cast<const float4*>(&lse_l[32 * qb + 8 * eg + 4 * h]);
                const float4 d4 = *reinterpret_cast<const float4*>(&dl_l[32 * qb + 8 * eg + 4 * h]);
                const float ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * eg + j;
                    float pe;
                    if (full) {
                        pe = __builtin_amdgcn_exp2f(s[e] * sl2 - ll[j]);
                    } else {
                        const int qq = qt * QT + 32 * qb + 8 * eg + 4 * h + j;
                        const bool ok = klive && (!p.causal || key <= qq);
                        pe = ok ? __builtin_amdgcn_exp2f(s[e] * sl2 - ll[j]) : 0.f;
                    }
                    pr[e] = pe; s[e] = pe * (dp[e] - dd[j]);
                }
            }
            const bf16x8_t p0 = pack_frag(pr, 0), p1 = pack_frag(pr, 8), ds0 = pack_frag(s, 0), ds1 = pack_frag(s, 8);
#pragma unroll
            for (int db = 0; db < D / 32; ++db) {
                if constexpr (DO_V) {
                    dvacc[db] = mfma16(frag_cols_tr<QRS>(DOl, 32 * qb, 32 * db, lane), p0, dvacc[db]);
                    dvacc[db] = mfma16(frag_cols_tr<QRS>(DOl, 32 * qb + 16, 32 * db, lane), p1, dvacc[db]);
                }
                if constexpr (DO_K) {
                    dkacc[db] = mfma16(frag_cols_tr<QRS>(Ql, 32 * qb, 32 * db, lane), ds0, dkacc[db]);
                    dkacc[db] = mfma16(frag_cols_tr<QRS>(Ql, 32 * qb + 16, 32 * db, lane), ds1, dkacc[db]);
                }
            }
        }
    }
    if (p.qsplit > 1) {
        if (key < p.Sk) {            // fp32 partial of this query slice; attn_dkv_reduce_kernel sums the slices in order
            float* PK = p.part + ((((long)b * p.H + hh) * p.qsplit + qs) * 2) * p.Sk * D + (long)key * D;
            float* PV = PK + (long)p.Sk * D;
#pragma unroll
            for (int db = 0; db < D / 32; ++db)
#pragma unroll
                for (int eg = 0; eg < 4; ++eg) {
                    if constexpr (DO_K) *reinterpret_cast<float4*>(PK + 32 * db + 8 * eg + 4 * h) =
                        make_float4(dkacc[db][4 * eg], dkacc[db][4 * eg + 1], dkacc[db][4 * eg + 2], dkacc[db][4 * eg + 3]);
                    if constexpr (DO_V) *reinterpret_cast<float4*>(PV + 32 * db + 8 * eg + 4 * h) =
                        make_float4(dvacc[db][4 * eg], dvacc[db][4 * eg + 1], dvacc[db][4 * eg + 2], dvacc[db][4 * eg + 3]);
                }
        }
        return;
    }
    if (key < p.Sk) {
        bf16_t* DK = p.dk + b * p.dk_sb + hh * p.dk_sh + (long)key * p.dk_ss;
        bf16_t* DV = p.dv + b * p.dv_sb + hh * p.dv_sh + (long)key * p.dv_ss;
#pragma unroll
        for (int db = 0; db < D / 32; ++db)
#pragma unroll
            for (int eg = 0; eg < 4; ++eg) {
                uint2 wk, wv;
                wk.x = pack_bf16x2(dkacc[db][4 * eg] * p.scale, dkacc[db][4 * eg + 1] * p.scale);
                wk.y = pack_bf16x2(dkacc[db][4 * eg + 2] * p.scale, dkacc[db][4 * eg + 3] * p.scale);
                wv.x = pack_bf16x2(dvacc[db][4 * eg], dvacc[db][4 * eg + 1]);
                wv.y = pack_bf16x2(dvacc[db][4 * eg + 2], dvacc[db][4 * eg + 3]);
                if constexpr (DO_K) *reinterpret_cast<uint2*>(DK + 32 * db + 8 * eg + 4 * h) = wk;
                if constexpr (DO_V) *reinterpret_cast<uint2*>(DV + 32 * db + 8 * eg + 4 * h) = wv;
            }
    }
}

// ================================================================================================ backward: dK, dV, LDS-DMA form
// Same computation and MODEs as attn_bwd_dkv_kernel; per 64-query tile the ring stage holds Q K-contiguous (S = Q . K^T), and per MODE dO
// MN-contiguous (dV^T += dO^T . P), dO K-contiguous (dP = dO . V^T), Q MN-contiguous (dK^T += Q^T . dS), plus the tile's lse / delta rows
// (256-byte DMA pieces).  Query rows past Sq arrive as zeros (Q, dO, lse, delta): P = 1 there but dO = 0 and dS = 0, so they add nothing.
template <int D, int NW, int MODE>
__global__ void __launch_bounds__(NW * 64, (D == 64 && NW == 4) ? 2 : 1) attn_bwd_dkv_dma_kernel(const AttnParams p) {
    constexpr bool DO_K = MODE != 1, DO_V = MODE != 2;
    constexpr int QT = 64, NKS = D / 16, NDB = D / 32;
    constexpr int IMG = QT * D * 2, NIMG = 1 + (DO_V ? 1 : 0) + (DO_K ? 2 : 0), STAGE = NIMG * IMG + 1024, PP = IMG / 1024 / NW;
    constexpr int O_DOM = IMG, O_DOC = (1 + (DO_V ? 1 : 0)) * IMG, O_QM = O_DOC + IMG, O_STAT = NIMG * IMG;
    __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];

    const int b = blockIdx.z, hh = blockIdx.y;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = lane & 31, h = lane >> 5;
    const int kblock = blockIdx.x / p.qsplit, qs = blockIdx.x % p.qsplit;
    const int key = kblock * 32 * NW + wid * 32 + i;
    const int kvl = p.kv_len ? min(p.kv_len[b], p.Sk) : p.Sk;
    const bool klive = key < kvl;
    const bf16_t* Q = p.q + b * p.q_sb + hh * p.q_sh;
    const bf16_t* K = p.k + b * p.k_sb + hh * p.k_sh;
    const bf16_t* V = p.v + b * p.v_sb + hh * p.v_sh;
    const bf16_t* DO = p.dout + b * p.do_sb + hh * p.do_sh;
    const float* LSE = p.lse + ((long)b * p.H + hh) * p.Sq;
    const float* DL = p.delta + ((long)b * p.H + hh) * p.Sq;
    const float sl2 = p.scale * LOG2E;

    bf16x8_t kf[NKS], vf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        kf[ks] = klive ? *reinterpret_cast<const bf16x8_t*>(K + (long)key * p.k_ss + 16 * ks + 8 * h) : zero_frag();
        vf[ks] = klive ? *reinterpret_cast<const bf16x8_t*>(V + (long)key * p.v_ss + 16 * ks + 8 * h) : zero_frag();
    }
    f32x16 dkacc[NDB], dvacc[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) { dkacc[db] = zero16(); dvacc[db] = zero16(); }

    const int nqt_all = (p.Sq + QT - 1) / QT;
    const int per = (nqt_all + p.qsplit - 1) / p.qsplit;
    const int qt0 = qs * per, nqt = min(nqt_all, qt0 + per);      // this workgroup's query tiles [qt0, nqt)

    const unsigned q_bytes = (unsigned)(((long)(p.Sq - 1) * p.q_ss + D) * 2), do_bytes = (unsigned)(((long)(p.Sq - 1) * p.do_ss + D) * 2);
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Q), (short)0, (int)q_bytes, 0x00020000);
    const auto rsDO = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(DO), (short)0, (int)do_bytes, 0x00020000);
    const auto rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(LSE), (short)0, p.Sq * 4, 0x00020000);
    const auto rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DL), (short)0, p.Sq * 4, 0x00020000);
    unsigned voQc[PP], voQm[PP], voDc[PP], voDm[PP];
#pragma unroll
    for (int j = 0; j < PP; ++j) {
        const int P = j * NW + wid;
        voQc[j] = dma_voffset<false, QT>(P % 8, lane, qt0 * QT, p.q_ss) + (unsigned)(P / 8) * 128u;
        voQm[j] = dma_voffset<true, D>(P, lane, 0, p.q_ss) + (unsigned)((long)qt0 * QT * p.q_ss * 2);
        voDc[j] = dma_voffset<false, QT>(P % 8, lane, qt0 * QT, p.do_ss) + (unsigned)(P / 8) * 128u;
        voDm[j] = dma_voffset<true, D>(P, lane, 0, p.do_ss) + (unsigned)((long)qt0 * QT * p.do_ss * 2);
    }
    const unsigned stepQ = (unsigned)(QT * p.q_ss * 2), stepD = (unsigned)(QT * p.do_ss * 2);
    unsigned voS = (unsigned)((qt0 * QT + lane) * 4);
#define ISSUE_TILE(buf)                                                                                                   \
    do {                                                                                                                  \
        char* base_ = lds + (buf) * STAGE + wid * 1024;                                                                   \
        _Pragma("unroll") for (int j = 0; j < PP; ++j) {                                                                  \
            char* d0_ = base_ + j * NW * 1024;                                                                            \
            const unsigned o0_ = voQc[j]; voQc[j] += stepQ;                                                               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_void_t*)d0_, 16, o0_, 0, 0, 0);                            \
            if constexpr (DO_V) {                                                                                         \
                char* d1_ = d0_ + O_DOM;                                                                                  \
                const unsigned o1_ = voDm[j]; voDm[j] += stepD;                                                           \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsDO, (lds_void_t*)d1_, 16, o1_, 0, 0, 0);                       \
            }                                                                                                             \
            if constexpr (DO_K) {                                                                                         \
                char* d2_ = d0_ + O_DOC;                                                                                  \
                char* d3_ = d0_ + O_QM;                                                                                   \
                const unsigned o2_ = voDc[j], o3_ = voQm[j]; voDc[j] += stepD; voQm[j] += stepQ;                          \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsDO, (lds_void_t*)d2_, 16, o2_, 0, 0, 0);                       \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_void_t*)d3_, 16, o3_, 0, 0, 0);                        \
            }                                                                                                             \
        }                                                                                                                 \
        if (wid == 0) {                                                                                                   \
            char* ds_ = lds + (buf) * STAGE + O_STAT;                                                                     \
            const unsigned os_ = voS;                                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsL, (lds_void_t*)ds_, 4, os_, 0, 0, 0);                             \
        }                                                                                                                 \
        if (DO_K && wid == 1) {                                                                                           \
            char* dd_ = lds + (buf) * STAGE + O_STAT + 256;                                                               \
            const unsigned od_ = voS;                                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (lds_void_t*)dd_, 4, od_, 0, 0, 0);                             \
        }                                                                                                                 \
        voS += QT * 4;                                                                                                    \
    } while (0)

    if (qt0 < nqt) ISSUE_TILE(0);
    for (int qt = qt0; qt < nqt; ++qt) {
        const int buf = (qt - qt0) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (qt + 1 < nqt) ISSUE_TILE(buf ^ 1);
        const char* Qc = lds + buf * STAGE;
        const char* DOm = Qc + O_DOM;
        const char* DOc = Qc + O_DOC;
        const char* Qm = Qc + O_QM;
        const float* lse_l = reinterpret_cast<const float*>(Qc + O_STAT);
        const float* dl_l = lse_l + 64;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (qt * QT + 32 * qb >= p.Sq) break;
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                s = mfma16(read_frag<false, QT>(Qc + (ks / 4) * (QT * 128), 32 * qb, ks & 3, lane), kf[ks], s);
                if constexpr (DO_K) dp = mfma16(read_frag<false, QT>(DOc + (ks / 4) * (QT * 128), 32 * qb, ks & 3, lane), vf[ks], dp);
            }
            f32x16 pr;
            // wave-uniform: all 32 keys of the wave are valid and nothing is causal (padding queries: see the header)
            const bool full = !p.causal && (key - i) + 32 <= kvl;
#pragma unroll
            for (int eg = 0; eg < 4; ++eg) {
                const float4 l4 = *reinterpret_cast<const float4*>(&lse_l[32 * qb + 8 * eg + 4 * h]);
                float4 d4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (DO_K) d4 = *reinterpret_cast<const float4*>(&dl_l[32 * qb + 8 * eg + 4 * h]);
                const float ll[4] = {l4.x * LOG2E, l4.y * LOG2E, l4.z * LOG2E, l4.w * LOG2E}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * eg + j;
                    float pe;
                    if (full) {
                        pe = __builtin_amdgcn_exp2f(s[e] * sl2 - ll[j]);
                    } else {
                        const int qq = qt * QT + 32 * qb + 8 * eg + 4 * h + j;
                        const bool ok = klive && qq < p.Sq && (!p.causal || key <= qq);
                        pe = ok ? __builtin_amdgcn_exp2f(s[e] * sl2 - ll[j]) : 0.f;
                    }
                    pr[e] = pe; s[e] = pe * (dp[e] - dd[j]);
                }
            }
            const bf16x8_t p0 = pack_frag(pr, 0), p1 = pack_frag(pr, 8), ds0 = pack_frag(s, 0), ds1 = pack_frag(s, 8);
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                if constexpr (DO_V) {
                    dvacc[db] = mfma16(read_frag_tr_acc<D>(DOm, 32 * db, 32 * qb, lane), p0, dvacc[db]);
                    dvacc[db] = mfma16(read_frag_tr_acc<D>(DOm, 32 * db, 32 * qb + 16, lane), p1, dvacc[db]);
                }
                if constexpr (DO_K) {
                    dkacc[db] = mfma16(read_frag_tr_acc<D>(Qm, 32 * db, 32 * qb, lane), ds0, dkacc[db]);
                    dkacc[db] = mfma16(read_frag_tr_acc<D>(Qm, 32 * db, 32 * qb + 16, lane), ds1, dkacc[db]);
                }
            }
        }
    }
#undef ISSUE_TILE
    if (p.qsplit > 1) {
        if (key < p.Sk) {            // fp32 partial of this query slice; attn_dkv_reduce_kernel sums the slices in order
            float* PK = p.part + ((((long)b * p.H + hh) * p.qsplit + qs) * 2) * p.Sk * D + (long)key * D;
            float* PV = PK + (long)p.Sk * D;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int eg = 0; eg < 4; ++eg) {
                    if constexpr (DO_K) *reinterpret_cast<float4*>(PK + 32 * db + 8 * eg + 4 * h) =
                        make_float4(dkacc[db][4 * eg], dkacc[db][4 * eg + 1], dkacc[db][4 * eg + 2], dkacc[db][4 * eg + 3]);
                    if constexpr (DO_V) *reinterpret_cast<float4*>(PV + 32 * db + 8 * eg + 4 * h) =
                        make_float4(dvacc[db][4 * eg], dvacc[db][4 * eg + 1], dvacc[db][4 * eg + 2], dvacc[db][4 * eg + 3]);
                }
        }
        return;
    }
    if (key < p.Sk) {
        bf16_t* DK = p.dk + b * p.dk_sb + hh * p.dk_sh + (long)key * p.dk_ss;
        bf16_t* DV = p.dv + b * p.dv_sb + hh * p.dv_sh + (long)key * p.dv_ss;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int eg = 0; eg < 4; ++eg) {
                uint2 wk, wv;
                wk.x = pack_bf16x2(dkacc[db][4 * eg] * p.scale, dkacc[db][4 * eg + 1] * p.scale);
                wk.y = pack_bf16x2(dkacc[db][4 * eg + 2] * p.scale, dkacc[db][4 * eg + 3] * p.scale);
                wv.x = pack_bf16x2(dvacc[db][4 * eg], dvacc[db][4 * eg + 1]);
                wv.y = pack_bf16x2(dvacc[db][4 * eg + 2], dvacc[db][4 * eg + 3]);
                if constexpr (DO_K) *reinterpret_cast<uint2*>(DK + 32 * db + 8 * eg + 4 * h) = wk;
                if constexpr (DO_V) *reinterpret_cast<uint2*>(DV + 32 * db + 8 * eg + 4 * h) = wv;
            }
    }
}

// dK / dV = sum over query slices of the fp32 partials (slice order: deterministic), one thread per 4 head-dim elements
template <int D>
__global__ void __launch_bounds__(256) attn_dkv_reduce_kernel(const AttnParams p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)p.B * p.H * p.Sk * (D / 4);
    if (idx >= total) return;
    const int d4 = (int)(idx % (D / 4)); const int key = (int)((idx / (D / 4)) % p.Sk);
    const int hh = (int)((idx / ((long)(D / 4) * p.Sk)) % p.H); const int b = (int)(idx / ((long)(D / 4) * p.Sk * p.H));
    const float* base = p.part + (((long)b * p.H + hh) * p.qsplit * 2) * p.Sk * D + (long)key * D + 4 * d4;
    float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f), v4 = k4;
    // (round 6) eight slices' reads in flight per pass: the one-slice-per-iteration loop was a dependent round trip per slice (12 of them for the 77-key cross
    // attention of a 1 024-token block); a pass past the last slice re-reads the last one and skips the add.  Slice order unchanged: same sums.
    const long slice = 2L * p.Sk * D;
    for (int s0 = 0; s0 < p.qsplit; s0 += 8) {
        float4 a[8], c[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int s = s0 + t < p.qsplit ? s0 + t : p.qsplit - 1;
            a[t] = *reinterpret_cast<const float4*>(base + s * slice);
            c[t] = *reinterpret_cast<const float4*>(base + s * slice + (long)p.Sk * D);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (s0 + t < p.qsplit) {
                k4.x += a[t].x; k4.y += a[t].y; k4.z += a[t].z; k4.w += a[t].w; v4.x += c[t].x; v4.y += c[t].y; v4.z += c[t].z; v4.w += c[t].w;
            }
        }
    }
    bf16_t* DK = p.dk + b * p.dk_sb + hh * p.dk_sh + (long)key * p.dk_ss + 4 * d4;
    bf16_t* DV = p.dv + b * p.dv_sb + hh * p.dv_sh + (long)key * p.dv_ss + 4 * d4;
    *reinterpret_cast<uint2*>(DK) = make_uint2(pack_bf16x2(k4.x * p.scale, k4.y * p.scale), pack_bf16x2(k4.z * p.scale, k4.w * p.scale));
    *reinterpret_cast<uint2*>(DV) = make_uint2(pack_bf16x2(v4.x, v4.y), pack_bf16x2(v4.z, v4.w));
}

// Query-slice count of the dK/dV kernel: fill ~256 workgroups when the key blocks x heads alone cannot (cross attention
// to 77 text tokens: one key block per head), at least one 64-query tile per slice.
int dkv_qsplit(int B, int H, int Sq, int Sk, int NW) {
    const long wgs = (long)cdiv(Sk, 32 * NW) * H * B;
    const int nqt = (int)cdiv(Sq, 64);
    long qs = wgs >= 192 ? 1 : 256 / wgs;
    if (qs > nqt) qs = nqt;
    if (qs > 64) qs = 64;
    return qs < 1 ? 1 : (int)qs;
}

bool strides_ok(const void* ptr, long sb, long ss, long sh) {
    return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0 && sb % 8 == 0 && ss % 8 == 0 && sh % 8 == 0;
}

}  // namespace

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

extern "C" {

int dpipe_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int* kv_len, int B, int H,
                   int Sq, int Sk, int D, long q_sb, long q_ss, long q_sh, long k_sb, long k_ss, long k_sh, long v_sb,
                   long v_ss, long v_sh, long o_sb, long o_ss, long o_sh, float scale, int causal, float* o_f32, void* stream) {
    if (!q || !k || !v || !o || !lse || B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0) { set_last_error("dpipe_attn_fwd: bad argument"); return DPIPE_ERR_ARG; }
    if (D != 64 && D != 128) { set_last_error("dpipe_attn_fwd: head dim must be 64 or 128"); return DPIPE_ERR_UNSUPPORTED; }
    if (ablated(ABL_ATTN)) return DPIPE_OK;                                        // (debug switch: runtime.hip)
    if (!strides_ok(q, q_sb, q_ss, q_sh) || !strides_ok(k, k_sb, k_ss, k_sh) || !strides_ok(v, v_sb, v_ss, v_sh) || !strides_ok(o, o_sb, o_ss, o_sh)) {
        set_last_error("dpipe_attn_fwd: tensors must be 16-byte aligned with strides multiple of 8 elements"); return DPIPE_ERR_ARG; }
    AttnParams p = {};
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.out = (bf16_t*)o; p.lse = lse; p.kv_len = kv_len;
    p.B = B; p.H = H; p.Sq = Sq; p.Sk = Sk; p.scale = scale; p.causal = causal;
    p.q_sb = q_sb; p.q_ss = q_ss; p.q_sh = q_sh; p.k_sb = k_sb; p.k_ss = k_ss; p.k_sh = k_sh;
    p.v_sb = v_sb; p.v_ss = v_ss; p.v_sh = v_sh; p.o_sb = o_sb; p.o_ss = o_ss; p.o_sh = o_sh;
    if (o_f32 && (reinterpret_cast<uintptr_t>(o_f32) & 15)) { set_last_error("dpipe_attn_fwd: o_f32 must be 16-byte aligned"); return DPIPE_ERR_ARG; }
    p.out32 = o_f32;
    // The LDS-DMA kernel: 8 waves x 32 query rows (two waves per SIMD: one wave's softmax runs under the other's MFMAs) once 256-row workgroups alone
    // fill most of the chip, else 4 waves x 32 rows (twice the workgroups, 2 - 3 of them per CU).  Measured (tools/kernel_timing.py attn, profiles/):
    // head dim 128 at 4.6k / 9.2k / 61k tokens 737 / 914 / 990 TFLOP/s (the register-staged kernel below: 385 / 448 / 474); SDXL's 1024 x 1024 x 20 heads
    // 23.8 us (34.0).  The K / V extents must fit the 32-bit buffer offsets.  DPIPE_ATTN_FWD_DMA = 0 selects the register-staged kernel (A/B timing).
    const int dma_mode = option(DPIPE_OPT_ATTN_FWD_DMA, 1);
    const bool fits = ((long)(Sk - 1) * k_ss + D) * 2 < (1l << 31) && ((long)(Sk - 1) * v_ss + D) * 2 < (1l << 31);
    if (fits && dma_mode != 0) {
        hipStream_t st = STREAM(stream);
        const long wg256 = (long)cdiv(Sq, 256) * H * B, wg128 = (long)cdiv(Sq, 128) * H * B;
        // 8-wave form from 192 256-row workgroups on.  (A 256-row workgroup streams its head's K / V once for twice the queries of a 128-row one; taking it from 16 / 64
        // workgroups on under concurrent lanes was neutral -- 22.10 / 22.14 vs 22.06 images/s, profiles/r4m_* -- and the option that selected it is gone)
        if (wg256 >= 192 && Sq >= 256) {
            if (D == 64) attn_fwd_dma_kernel<64, 1, 8><<<(unsigned)wg256, 512, 0, st>>>(p);
            else attn_fwd_dma_kernel<128, 1, 8><<<(unsigned)wg256, 512, 0, st>>>(p);
        } else {
            if (D == 64) attn_fwd_dma_kernel<64, 1, 4><<<(unsigned)wg128, 256, 0, st>>>(p);
            else attn_fwd_dma_kernel<128, 1, 4><<<(unsigned)wg128, 256, 0, st>>>(p);
        }
        return check_launch("dpipe_attn_fwd");
    }
    // 128 queries per workgroup (4 waves) when that alone fills the chip, else 64 (2 waves): twice the workgroups
    const bool small = cdiv(Sq, 128) * H * B < 256;
    if (small) {
        dim3 grid((unsigned)cdiv(Sq, 64), (unsigned)H, (unsigned)B);
        if (D == 64) attn_fwd_kernel<64, 2><<<grid, 128, 0, STREAM(stream)>>>(p);
        else attn_fwd_kernel<128, 2><<<grid, 128, 0, STREAM(stream)>>>(p);
    } else {
        dim3 grid((unsigned)cdiv(Sq, 128), (unsigned)H, (unsigned)B);
        if (D == 64) attn_fwd_kernel<64, 4><<<grid, 256, 0, STREAM(stream)>>>(p);
        else attn_fwd_kernel<128, 4><<<grid, 256, 0, STREAM(stream)>>>(p);
    }
    return check_launch("dpipe_attn_fwd");
}

long dpipe_attn_bwd_partial_floats(int B, int H, int Sq, int Sk, int D) {
    const int qs = dkv_qsplit(B, H, Sq, Sk, 4);
    return qs > 1 ? (long)B * H * qs * 2 * Sk * D : 0;
}

int dpipe_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                   float* delta, void* dq, void* dk, void* dv, const int* kv_len, int B, int H, int Sq, int Sk, int D,
                   long q_sb, long q_ss, long q_sh, long k_sb, long k_ss, long k_sh, long v_sb, long v_ss, long v_sh,
                   long o_sb, long o_ss, long o_sh, long do_sb, long do_ss, long do_sh, long dq_sb, long dq_ss,
                   long dq_sh, long dk_sb, long dk_ss, long dk_sh, long dv_sb, long dv_ss, long dv_sh, float scale,
                   int causal, float* dkv_partial, long dkv_partial_floats, const float* o_f32, void* stream) {
    if (!q || !k || !v || !o || !dout || !lse || !delta || !dq || !dk || !dv || B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0) {
        set_last_error("dpipe_attn_bwd: bad argument"); return DPIPE_ERR_ARG; }
    if (D != 64 && D != 128) { set_last_error("dpipe_attn_bwd: head dim must be 64 or 128"); return DPIPE_ERR_UNSUPPORTED; }
    if (ablated(ABL_ATTN)) return DPIPE_OK;
    if (!strides_ok(q, q_sb, q_ss, q_sh) || !strides_ok(k, k_sb, k_ss, k_sh) || !strides_ok(v, v_sb, v_ss, v_sh) || !strides_ok(o, o_sb, o_ss, o_sh) ||
        !strides_ok(dout, do_sb, do_ss, do_sh) || !strides_ok(dq, dq_sb, dq_ss, dq_sh) || !strides_ok(dk, dk_sb, dk_ss, dk_sh) || !strides_ok(dv, dv_sb, dv_ss, dv_sh)) {
        set_last_error("dpipe_attn_bwd: tensors must be 16-byte aligned with strides multiple of 8 elements"); return DPIPE_ERR_ARG; }
    AttnParams p = {};
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (const bf16_t*)o; p.dout = (const bf16_t*)dout;
    p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.lse = const_cast<float*>(lse); p.delta = delta; p.kv_len = kv_len;
    p.B = B; p.H = H; p.Sq = Sq; p.Sk = Sk; p.scale = scale; p.causal = causal;
    p.q_sb = q_sb; p.q_ss = q_ss; p.q_sh = q_sh; p.k_sb = k_sb; p.k_ss = k_ss; p.k_sh = k_sh;
    p.v_sb = v_sb; p.v_ss = v_ss; p.v_sh = v_sh; p.o_sb = o_sb; p.o_ss = o_ss; p.o_sh = o_sh;
    p.do_sb = do_sb; p.do_ss = do_ss; p.do_sh = do_sh; p.dq_sb = dq_sb; p.dq_ss = dq_ss; p.dq_sh = dq_sh;
    p.dk_sb = dk_sb; p.dk_ss = dk_ss; p.dk_sh = dk_sh; p.dv_sb = dv_sb; p.dv_ss = dv_ss; p.dv_sh = dv_sh;
    if (o_f32 && (reinterpret_cast<uintptr_t>(o_f32) & 15)) { set_last_error("dpipe_attn_bwd: o_f32 must be 16-byte aligned"); return DPIPE_ERR_ARG; }
    p.o32 = o_f32;
    hipStream_t s = STREAM(stream);
    constexpr int NW = 4;
    const long rows = (long)B * H * Sq;
    p.qsplit = dkv_qsplit(B, H, Sq, Sk, NW);
    if (p.qsplit > 1 && (!dkv_partial || dkv_partial_floats < (long)B * H * p.qsplit * 2 * Sk * D)) p.qsplit = 1;   // no workspace: unsplit
    p.part = dkv_partial;
    const bool small_q = cdiv(Sq, 128) * H * B < 256;
    dim3 gq((unsigned)cdiv(Sq, small_q ? 64 : 128), (unsigned)H, (unsigned)B), gk((unsigned)(cdiv(Sk, 32 * NW) * p.qsplit), (unsigned)H, (unsigned)B);
    const unsigned gred = (unsigned)cdiv((long)B * H * Sk * (D / 4), 256);
    // head dim 128: the 4-wave dQ kernel needs > 256 registers (one wave per SIMD); 8 waves x 32 rows fit 2 per SIMD (220 VGPRs) -- DPIPE_ATTN_DQ8 = 0 for A/B timing
    const bool dq8_on = option(DPIPE_OPT_ATTN_DQ8, 1) != 0;
    const bool dq8 = dq8_on && (long)cdiv(Sq, 256) * H * B >= 192 && Sq >= 256;
    dim3 gq8((unsigned)cdiv(Sq, 256), (unsigned)H, (unsigned)B);
    // LDS-DMA forms of the backward kernels (K / V extents within the 32-bit buffer offsets) -- DPIPE_ATTN_BWD_DMA = 0 for A/B timing
    const bool bwd_dma_on = option(DPIPE_OPT_ATTN_BWD_DMA, 1) != 0;
    const bool bwd_dma = bwd_dma_on && ((long)(Sk - 1) * k_ss + D) * 2 < (1l << 31) && ((long)(Sk - 1) * v_ss + D) * 2 < (1l << 31);
    const long wg256 = (long)cdiv(Sq, 256) * H * B, wg128 = (long)cdiv(Sq, 128) * H * B;
    const bool bwd_dma_q = bwd_dma_on && ((long)(Sq - 1) * q_ss + D) * 2 < (1l << 31) && ((long)(Sq - 1) * do_ss + D) * 2 < (1l << 31);   // the dK / dV kernels stream Q and dO
    // head dim 128, long key sequences: dV and dK as two 8-wave kernels (two waves per SIMD) -- DPIPE_ATTN_DKV_SPLIT = 0 for A/B timing
    const bool split_on = option(DPIPE_OPT_ATTN_DKV_SPLIT, 1) != 0;
    const bool dkv_split = split_on && p.qsplit == 1 && (long)cdiv(Sk, 256) * H * B >= 192;
    dim3 gk8((unsigned)cdiv(Sk, 256), (unsigned)H, (unsigned)B);
#define ATTN_BWD(DD) do { \
        if (!bwd_dma) attn_delta_kernel<DD><<<(unsigned)cdiv(rows, 16), 256, 0, s>>>(p);   /* the LDS-DMA dQ kernel computes delta itself */ \
        if (bwd_dma && dq8) attn_bwd_dq_dma_kernel<DD, 8><<<(unsigned)wg256, 512, 0, s>>>(p); \
        else if (bwd_dma) attn_bwd_dq_dma_kernel<DD, 4><<<(unsigned)wg128, 256, 0, s>>>(p); \
        else if (small_q) attn_bwd_dq_kernel<DD, 2><<<gq, 128, 0, s>>>(p); \
        else if (DD == 128 && dq8) attn_bwd_dq_kernel<DD, 8><<<gq8, 512, 0, s>>>(p); \
        else attn_bwd_dq_kernel<DD, 4><<<gq, 256, 0, s>>>(p); \
        if (DD == 128 && dkv_split && bwd_dma_q) { attn_bwd_dkv_dma_kernel<DD, 8, 1><<<gk8, 512, 0, s>>>(p); attn_bwd_dkv_dma_kernel<DD, 8, 2><<<gk8, 512, 0, s>>>(p); } \
        else if (DD == 128 && dkv_split) { attn_bwd_dkv_kernel<DD, 8, 1><<<gk8, 512, 0, s>>>(p); attn_bwd_dkv_kernel<DD, 8, 2><<<gk8, 512, 0, s>>>(p); } \
        else if (bwd_dma_q) attn_bwd_dkv_dma_kernel<DD, NW, 0><<<gk, NW * 64, 0, s>>>(p); \
        else attn_bwd_dkv_kernel<DD, NW><<<gk, NW * 64, 0, s>>>(p); \
        if (p.qsplit > 1) attn_dkv_reduce_kernel<DD><<<gred, 256, 0, s>>>(p); } while (0)
    if (D == 64) ATTN_BWD(64); else ATTN_BWD(128);
#undef ATTN_BWD
    return check_launch("dpipe_attn_bwd");
}

}  // extern "C"
