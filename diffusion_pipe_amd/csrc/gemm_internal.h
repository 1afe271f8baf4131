// gemm_internal.h -- argument block shared by the two bf16/fp32 GEMM back ends of dpipe_gemm (gemm.hip: generic
// register-staged kernel; gemm_pipe.hip: LDS-DMA pipelined bf16 kernel with in-launch split-K).
#pragma once
#include "dpipe_common.h"

namespace dpipe {

// Implicit-GEMM convolution geometry (gemm_pipe_kernel.h CONV modes; NHWC bf16 tensors).  "rows" = the pixel grid the GEMM enumerates
// (CONV 1: the M rows; CONV 2: the K rows); "src" = the tensor whose pixels are gathered.
struct ConvGeom {
    int rows_h, rows_w;          // pixel grid enumerated: row r -> (b, y, x) = (r / (rows_h * rows_w), ...)
    int src_h, src_w;            // spatial size of the gathered tensor (before a fused nearest up-sampling)
    int kw, taps;                // kernel width, kh * kw
    int cchunks;                 // CONV 1: channels of the gathered tensor / 64 (k-steps per tap)
    int stride_log2, ups_log2;   // conv stride 1 / 2; fused nearest up-sampling of the source 1x / 2x
    int pad;
    int flip;                    // CONV 1 dgrad: source = (y + pad - ky) / stride when divisible, else zero
    int tap0;                    // CONV 2: first tap of this launch (grid.y indexes taps)
    long b_tap_stride;           // CONV 1 dgrad: element offset of tap t inside a weight row (= Cin); B k-row pitch = ldb
    long a_ext, b_ext;           // element extents of the gathered / irregularly addressed operands (buffer bounds)
};

struct GemmParams {
    const void* A; const void* B; void* C; const void* bias;
    int M, N, K;
    long lda, ldb, ldc;
    long sAo, sAi, sBo, sBi, sCo, sCi;  // outer / inner batch strides (elements)
    int batch_inner;                      // batch index z -> (z / batch_inner, z % batch_inner)
    float alpha;
    int act, accumulate, out_f32;
    int vecA, vecB;                       // 16-byte global loads legal for A / B
    int tiles_m, tiles_n;
    int gr;                               // pipelined kernel: tile-rows a run of consecutive tiles walks before it moves one tile-column over (rasterisation group height)
    // split-K (pipelined kernel only)
    int splitk, ksteps, ksteps_per_split;
    float* slabs; int* counters;
    // fused epilogue extras
    const void* residual; long ldr;       // C += residual[m, n] (C's dtype, row pitch ldr, batch strides of C)
    void* colsum; int colsum_acc;          // pipelined TN kernel only: colsum[m] (+)= sum_k A[k][m]  (bias gradient of a wgrad GEMM), operand dtype
    int bias_rows;                         // 0: one bias row for every output row; r > 0: output row m takes bias row m / r (a per-sample bias of a convolution: r = Ho Wo), pitch N
    long bias_lo;                          // 0, or the element offset from a bias row to its LO row: the bias is a bf16 hi / lo pair of an fp32 addend, both are added (round 6)
    ConvGeom cg;                           // CONV modes only
#ifdef DPIPE_TIMELINE
    void* timeline;                        // tools/probes/gemm_timeline.hip only
#endif
};

enum { ACT_NONE = 0, ACT_GELU_TANH = 1, ACT_GELU_ERF = 2, ACT_SILU = 3, ACT_QUICK_GELU = 4,
       ACT_GEGLU_BWD = 16 };     // flag on top of an activation code (DPIPE_ACT_GEGLU_BWD): the GEGLU backward rides this dgrad GEMM's epilogue, see gemm_pipe_kernel.h
__device__ __forceinline__ float epilogue_act(float x, int act) {
    switch (act) {
    case ACT_GELU_TANH: { const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x); return 0.5f * x * (1.f + tanhf(u)); }
    case ACT_GELU_ERF: { float E; return x * gelu_erf_cdf(x, E); }
    case ACT_SILU: return x / (1.f + __expf(-x));
    case ACT_QUICK_GELU: return x / (1.f + __expf(-1.702f * x));
    default: return x;
    }
}

// d act / dx (the formulas of elementwise.hip's act_bwd)
__device__ __forceinline__ float epilogue_act_grad(float x, int act) {
    switch (act) {
    case ACT_GELU_TANH: {
        const float k0 = 0.7978845608028654f, k1 = 0.044715f;
        const float u = k0 * (x + k1 * x * x * x);
        const float th = tanhf(u);
        const float du = k0 * (1.f + 3.f * k1 * x * x);
        return 0.5f * (1.f + th) + 0.5f * x * (1.f - th * th) * du;
    }
    case ACT_GELU_ERF: {
        float E;
        const float cdf = gelu_erf_cdf(x, E);
        return cdf + x * (0.3989422804014327f * E);
    }
    case ACT_SILU: { const float s = 1.f / (1.f + __expf(-x)); return s * (1.f + x * (1.f - s)); }
    case ACT_QUICK_GELU: { const float s = 1.f / (1.f + __expf(-1.702f * x)); return s * (1.f + 1.702f * x * (1.f - s)); }
    default: return 1.f;
    }
}

// Pipelined bf16 path.  Returns true when it launched the GEMM (*rc_out = launch status), false when the problem is
// not eligible (caller falls through to the generic kernel).  `ws` = [counters: 4 KiB][fp32 slabs ...], zero-initialised
// once by the host and private to one stream; NULL disables split-K.
bool gemm_pipe_try(GemmParams& p, int transA, int transB, int batch, void* ws, long ws_bytes, int force_splitk, int force_tile,
                   hipStream_t s, int* rc_out);
// Tile / split-K choice shared by the GEMM and the convolution front ends: fills p.tiles_*, p.ksteps*, p.splitk, p.counters, p.slabs,
// p.vecA / p.vecB (epilogue vector flags) and returns the tile code (64, 128, 129 = 128^2 2-deep ring, 256 = 256 x 128, 257 = 256^2, 63).
// `allow_vs` = false: never the register-staged tile (132) -- the convolution front end, whose gathered-row instances of it lost their measurement (conv_pipe.hip), keeps its own 128 / 129 ring choice
int gemm_pipe_plan(GemmParams& p, bool a_mc, bool b_mc, int batch, void* ws, long ws_bytes, int force_splitk, int force_tile, int counter_base = 0, long slab_base = 0, bool allow_vs = true);
// n independent plain bf16 GEMMs (batch 1, all eligible for the pipelined kernel: the caller checked gemm_pipe_eligible) as few launches as possible; *launches_out = launches issued
// tiles_out != NULL: plan only (no launch): the tile code and split factor the launch would use, per problem (dpipe_gemm_group_plan)
int gemm_pipe_group(GemmParams* ps, const int* transA, const int* transB, int n, void* ws, long ws_bytes, hipStream_t s, int* launches_out, int* tiles_out = nullptr, int* splitk_out = nullptr);
bool gemm_pipe_eligible(const GemmParams& p, int transA, int transB);

}  // namespace dpipe
