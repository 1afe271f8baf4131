// groupnorm.hip -- GroupNorm (+ fused SiLU) forward / backward on NCHW activations (gfx950, wave64).
//
// SDXL's UNet (diffusers ResnetBlock2D / Transformer2DModel behind models/sdxl.py:797-865) normalises every convolution
// input with nn.GroupNorm(32, C) and, in the resnets, feeds it through SiLU.  In NCHW a group (n, g) is ONE contiguous run
// of L = (C / G) * H * W elements, so GroupNorm is a row normalisation over long rows with a per-channel affine:
//
//   forward   stats  : grid (N*G, SPLIT)   partial (sum, sum of squares) of one chunk of a group           -> workspace
//             apply  : grid (N*C, HSPLIT)  every block folds its group's partials (a handful of floats from L2), then
//                                          y = act((x - mean) * rstd * gamma[c] + beta[c]);  mean / rstd saved per group
//   backward  stats  : grid (N*C, HSPLIT)  per channel chunk: s1 = sum dz * xhat, s2 = sum dz   (dz = dy * act'(z))
//             apply  : grid (N*C, HSPLIT)  folds the group's per-channel sums (A = sum gamma s1 / L, B = sum gamma s2 / L),
//                                          dx = rstd * (dz * gamma[c] - B - xhat * A);  block 0 of a channel also writes
//                                          dgamma[c] (+)= sum_n s1, dbeta[c] (+)= sum_n s2
//
// 4 launches with 16-byte accesses and hundreds of workgroups, instead of ATen's RowwiseMoments (one workgroup per group:
// 32 workgroups, 68 us at 1024^2) + ~10 small kernels + separate SiLU forward / backward.  Statistics in fp32
// (E[x^2] - mean^2, clamped at 0), deterministic (no atomics).  Bound: HBM (3 passes forward, 5 backward).
#include "dpipe_common.h"
#include "../../include/dpipe_hip.h"

using namespace dpipe;

namespace {

constexpr int NB = 256;

__device__ __forceinline__ float silu_f(float z) { return z / (1.f + __expf(-z)); }
__device__ __forceinline__ float silu_grad(float z) { const float s = 1.f / (1.f + __expf(-z)); return s * (1.f + z * (1.f - s)); }

// chunk [c0, c1) of `len` elements split into `parts`, vector aligned
__device__ __forceinline__ void chunk_of(long len, int parts, int part, int V, long& c0, long& c1) {
    const long per = cdiv(cdiv(len, parts), V) * V;
    c0 = min(len, per * part); c1 = min(len, c0 + per);
}

template <typename T>
__global__ void __launch_bounds__(NB) gn_stats_kernel(const T* __restrict__ x, float* __restrict__ ws, long L, int split) {
    constexpr int V = Elem<T>::VEC;
    __shared__ float smem[16];
    const long row = blockIdx.x;
    long c0, c1; chunk_of(L, split, blockIdx.y, V, c0, c1);
    const T* xr = x + row * L;
    float s = 0.f, q = 0.f;
    for (long i = c0 + (long)threadIdx.x * V; i < c1; i += (long)NB * V) {
        Vec16<T> v; v.load(xr + i);
        float f[V]; v.unpack(f);
#pragma unroll
        for (int j = 0; j < V; ++j) { s += f[j]; q += f[j] * f[j]; }
    }
    s = block_sum(s, smem); q = block_sum(q, smem);
    if (threadIdx.x == 0) { ws[(row * split + blockIdx.y) * 2] = s; ws[(row * split + blockIdx.y) * 2 + 1] = q; }
}

template <typename T, typename W, int ACT>
__global__ void __launch_bounds__(NB) gn_apply_kernel(const T* __restrict__ x, const W* __restrict__ gamma, const W* __restrict__ beta,
                                                      const float* __restrict__ ws, T* __restrict__ y, float* __restrict__ mean_out,
                                                      float* __restrict__ rstd_out, int C, long HW, int G, int split, int hsplit, float eps) {
    constexpr int V = Elem<T>::VEC;
    const long nc = blockIdx.x;                  // n * C + c
    const int c = (int)(nc % C); const long n = nc / C;
    const int cpg = C / G;
    const long row = n * G + c / cpg;
    const long L = (long)cpg * HW;
    float s = 0.f, q = 0.f;
    for (int k = 0; k < split; ++k) { s += ws[(row * split + k) * 2]; q += ws[(row * split + k) * 2 + 1]; }   // same order in every block
    const float mean = s / (float)L;
    const float rstd = rsqrtf(fmaxf(q / (float)L - mean * mean, 0.f) + eps);
    if (threadIdx.x == 0 && blockIdx.y == 0 && c % cpg == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    const float ga = gamma ? Elem<W>::to_f(gamma[c]) * rstd : rstd;
    const float be = (beta ? Elem<W>::to_f(beta[c]) : 0.f) - mean * ga;
    long c0, c1; chunk_of(HW, hsplit, blockIdx.y, V, c0, c1);
    const T* xr = x + nc * HW; T* yr = y + nc * HW;
    for (long i = c0 + (long)threadIdx.x * V; i < c1; i += (long)NB * V) {
        Vec16<T> v; v.load(xr + i);
        float f[V]; v.unpack(f);
#pragma unroll
        for (int j = 0; j < V; ++j) { const float z = f[j] * ga + be; f[j] = ACT ? silu_f(z) : z; }
        Vec16<T> o; o.pack(f); o.store(yr + i);
    }
}

template <typename T, typename W, int ACT>
__global__ void __launch_bounds__(NB) gn_bwd_stats_kernel(const T* __restrict__ x, const T* __restrict__ dy, const W* __restrict__ gamma,
                                                          const W* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          float* __restrict__ ws, int C, long HW, int G, int hsplit) {
    constexpr int V = Elem<T>::VEC;
    __shared__ float smem[16];
    const long nc = blockIdx.x;
    const int c = (int)(nc % C); const long n = nc / C;
    const long row = n * G + c / (C / G);
    const float mu = mean[row], rs = rstd[row];
    const float ga = gamma ? Elem<W>::to_f(gamma[c]) : 1.f, be = beta ? Elem<W>::to_f(beta[c]) : 0.f;
    long c0, c1; chunk_of(HW, hsplit, blockIdx.y, V, c0, c1);
    const T* xr = x + nc * HW; const T* gr = dy + nc * HW;
    float s1 = 0.f, s2 = 0.f;
    for (long i = c0 + (long)threadIdx.x * V; i < c1; i += (long)NB * V) {
        Vec16<T> vx, vg; vx.load(xr + i); vg.load(gr + i);
        float fx[V], fg[V]; vx.unpack(fx); vg.unpack(fg);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float xh = (fx[j] - mu) * rs;
            const float dz = ACT ? fg[j] * silu_grad(xh * ga + be) : fg[j];
            s1 += dz * xh; s2 += dz;
        }
    }
    s1 = block_sum(s1, smem); s2 = block_sum(s2, smem);
    if (threadIdx.x == 0) { ws[(nc * hsplit + blockIdx.y) * 2] = s1; ws[(nc * hsplit + blockIdx.y) * 2 + 1] = s2; }
}

template <typename T, typename W, int ACT>
__global__ void __launch_bounds__(NB) gn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, const W* __restrict__ gamma,
                                                          const W* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ ws, T* __restrict__ dx, const T* __restrict__ dx_add, W* __restrict__ dgamma,
                                                          W* __restrict__ dbeta, long N, int C, long HW, int G, int hsplit, int accumulate) {
    constexpr int V = Elem<T>::VEC;
    __shared__ float smem[16];
    const long nc = blockIdx.x;
    const int c = (int)(nc % C); const long n = nc / C;
    const int cpg = C / G; const int g = c / cpg;
    const long row = n * G + g;
    const float mu = mean[row], rs = rstd[row];
    // group sums A = sum_c' gamma s1 / L, B = sum_c' gamma s2 / L over the cpg * hsplit partial pairs of this (n, g)
    float a = 0.f, b = 0.f;
    for (int k = threadIdx.x; k < cpg * hsplit; k += NB) {
        const int cc = g * cpg + k / hsplit;
        const float gg = gamma ? Elem<W>::to_f(gamma[cc]) : 1.f;
        const long o = ((n * C + cc) * hsplit + k % hsplit) * 2;
        a += gg * ws[o]; b += gg * ws[o + 1];
    }
    a = block_sum(a, smem); b = block_sum(b, smem);
    const float invL = 1.f / (float)((long)cpg * HW);
    const float A = a * invL, B = b * invL;
    const float ga = gamma ? Elem<W>::to_f(gamma[c]) : 1.f, be = beta ? Elem<W>::to_f(beta[c]) : 0.f;
    long c0, c1; chunk_of(HW, hsplit, blockIdx.y, V, c0, c1);
    const T* xr = x + nc * HW; const T* gr = dy + nc * HW; T* dr = dx + nc * HW;
    for (long i = c0 + (long)threadIdx.x * V; i < c1; i += (long)NB * V) {
        Vec16<T> vx, vg; vx.load(xr + i); vg.load(gr + i);
        float fx[V], fg[V]; vx.unpack(fx); vg.unpack(fg);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float xh = (fx[j] - mu) * rs;
            const float dz = ACT ? fg[j] * silu_grad(xh * ga + be) : fg[j];
            fx[j] = rs * (dz * ga - B - xh * A);
        }
        if (dx_add) {          // gradient of the branch that bypasses the norm (resnet skip / shortcut), added in the same pass
            Vec16<T> va; va.load(dx_add + nc * HW + i);
            float fa[V]; va.unpack(fa);
#pragma unroll
            for (int j = 0; j < V; ++j) fx[j] += fa[j];
        }
        Vec16<T> o; o.pack(fx); o.store(dr + i);
    }
    // parameter gradients: one thread per channel (the n == 0, chunk 0 block), summed over samples and chunks in order
    if (n == 0 && blockIdx.y == 0 && threadIdx.x == 0 && (dgamma || dbeta)) {
        float s1 = 0.f, s2 = 0.f;
        for (long nn = 0; nn < N; ++nn)
            for (int k = 0; k < hsplit; ++k) { const long o = ((nn * C + c) * hsplit + k) * 2; s1 += ws[o]; s2 += ws[o + 1]; }
        if (dgamma) dgamma[c] = Elem<W>::from_f(accumulate ? Elem<W>::to_f(dgamma[c]) + s1 : s1);
        if (dbeta) dbeta[c] = Elem<W>::from_f(accumulate ? Elem<W>::to_f(dbeta[c]) + s2 : s2);
    }
}

int pick_split(long len, int V) {            // ~16 K elements per workgroup
    long s = len / 16384; if (s < 1) s = 1; if (s > 16) s = 16;
    while (s > 1 && cdiv(cdiv(len, s), V) * V * (s - 1) >= len) --s;      // no empty chunks
    return (int)s;
}

}  // namespace

#define STREAM(s) reinterpret_cast<hipStream_t>(s)
#define BAD(msg) do { set_last_error(msg); return DPIPE_ERR_ARG; } while (0)

extern "C" {

long dpipe_groupnorm_workspace_floats(long N, int C, long HW, int G) {
    const long L = (long)(C / G) * HW;
    long need = 0;
    for (int V = 4; V <= 8; V += 4) {            // either element width
        const long fwd = N * G * pick_split(L, V) * 2, bwd = N * C * pick_split(HW, V) * 2;
        need = need > fwd ? need : fwd; need = need > bwd ? need : bwd;
    }
    return need;
}

int dpipe_groupnorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd, float* workspace, long N,
                        int C, long HW, int G, float eps, int act, int dtype, int wdtype, void* stream) {
    const int V = dtype == DPIPE_BF16 ? 8 : 4;
    if (!x || !y || !mean || !rstd || !workspace || N <= 0 || C <= 0 || G <= 0 || C % G != 0 || HW <= 0 || HW % V != 0) BAD("dpipe_groupnorm_fwd: bad argument");
    if (act != DPIPE_ACT_NONE && act != DPIPE_ACT_SILU) BAD("dpipe_groupnorm_fwd: activation must be none or silu");
    hipStream_t s = STREAM(stream);
    const long L = (long)(C / G) * HW;
    const int split = pick_split(L, V), hsplit = pick_split(HW, V);
    dim3 g1((unsigned)(N * G), split), g2((unsigned)(N * C), hsplit);
#define GN_FWD(TT, WW) do { \
        gn_stats_kernel<TT><<<g1, NB, 0, s>>>((const TT*)x, workspace, L, split); \
        if (act) gn_apply_kernel<TT, WW, 1><<<g2, NB, 0, s>>>((const TT*)x, (const WW*)gamma, (const WW*)beta, workspace, (TT*)y, mean, rstd, C, HW, G, split, hsplit, eps); \
        else gn_apply_kernel<TT, WW, 0><<<g2, NB, 0, s>>>((const TT*)x, (const WW*)gamma, (const WW*)beta, workspace, (TT*)y, mean, rstd, C, HW, G, split, hsplit, eps); } while (0)
    if (dtype == DPIPE_BF16 && wdtype == DPIPE_BF16) GN_FWD(bf16_t, bf16_t);
    else if (dtype == DPIPE_BF16 && wdtype == DPIPE_F32) GN_FWD(bf16_t, float);
    else if (dtype == DPIPE_F32 && wdtype == DPIPE_F32) GN_FWD(float, float);
    else { set_last_error("dpipe_groupnorm_fwd: dtype combination"); return DPIPE_ERR_UNSUPPORTED; }
#undef GN_FWD
    return check_launch("dpipe_groupnorm_fwd");
}

int dpipe_groupnorm_bwd(const void* x, const void* dy, const void* gamma, const void* beta, const float* mean, const float* rstd, void* dx,
                        void* dgamma, void* dbeta, float* workspace, long N, int C, long HW, int G, int act, int dtype, int wdtype,
                        int accumulate_params, const void* dx_add, void* stream) {
    const int V = dtype == DPIPE_BF16 ? 8 : 4;
    if (!x || !dy || !mean || !rstd || !dx || !workspace || N <= 0 || C <= 0 || G <= 0 || C % G != 0 || HW <= 0 || HW % V != 0) BAD("dpipe_groupnorm_bwd: bad argument");
    if (act != DPIPE_ACT_NONE && act != DPIPE_ACT_SILU) BAD("dpipe_groupnorm_bwd: activation must be none or silu");
    hipStream_t s = STREAM(stream);
    const int hsplit = pick_split(HW, V);
    dim3 g2((unsigned)(N * C), hsplit);
#define GN_BWD(TT, WW, AA) do { \
        gn_bwd_stats_kernel<TT, WW, AA><<<g2, NB, 0, s>>>((const TT*)x, (const TT*)dy, (const WW*)gamma, (const WW*)beta, mean, rstd, workspace, C, HW, G, hsplit); \
        gn_bwd_apply_kernel<TT, WW, AA><<<g2, NB, 0, s>>>((const TT*)x, (const TT*)dy, (const WW*)gamma, (const WW*)beta, mean, rstd, workspace, (TT*)dx, (const TT*)dx_add, \
                                                           (WW*)dgamma, (WW*)dbeta, N, C, HW, G, hsplit, accumulate_params); } while (0)
#define GN_BWD_ACT(TT, WW) do { if (act) GN_BWD(TT, WW, 1); else GN_BWD(TT, WW, 0); } while (0)
    if (dtype == DPIPE_BF16 && wdtype == DPIPE_BF16) GN_BWD_ACT(bf16_t, bf16_t);
    else if (dtype == DPIPE_BF16 && wdtype == DPIPE_F32) GN_BWD_ACT(bf16_t, float);
    else if (dtype == DPIPE_F32 && wdtype == DPIPE_F32) GN_BWD_ACT(float, float);
    else { set_last_error("dpipe_groupnorm_bwd: dtype combination"); return DPIPE_ERR_UNSUPPORTED; }
#undef GN_BWD_ACT
#undef GN_BWD
    return check_launch("dpipe_groupnorm_bwd");
}

}  // extern "C"
