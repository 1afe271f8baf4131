// groupnorm_nhwc.hip -- GroupNorm (+ fused SiLU) forward / backward on NHWC (token-major [N, HW, C]) activations (gfx950, wave64).
//
// The UNet of the SDXL step (diffusers ResnetBlock2D / Transformer2DModel behind models/sdxl.py:797-865) runs channels-last here: the
// convolutions are implicit GEMMs over [pixels, channels] (conv_pipe.hip) and the transformer blocks are token-major anyway, so
// nn.GroupNorm(32, C) sees [N, HW, C] with the C / G channels of a group contiguous inside every pixel row.  Every kernel gives a
// thread a FIXED 16-byte channel vector and walks rows, so per-channel quantities (gamma, beta, the group's mean / rstd) live in
// registers and per-channel sums fall out of the row walk:
//
//   forward   partial : grid (row chunks, N)  per-channel (sum x, sum x^2) of one chunk of rows                     -> workspace
//             final   : grid (N * G)          mean / rstd of a group from the chunks' channel sums (fixed order)
//             apply   : grid (row chunks, N)  y = act((x - mean) * rstd * gamma[c] + beta[c])
//   backward  partial : grid (row chunks, N)  per-channel s1 = sum dz, s2 = sum dz * xhat   (dz = dy * act'(z))     -> workspace
//             final   : grid (G)              per (n, group): A = sum_c gamma s1, B = sum_c gamma s2; dgamma[c] (+)= sum_n s2,
//                                             dbeta[c] (+)= sum_n s1 (n in order: deterministic)
//             apply   : grid (row chunks, N)  dx = rstd * (gamma dz - A / m - xhat B / m),  m = HW * C / G
//
// fp32 statistics (E[x^2] - mean^2 clamped at 0, as groupnorm.hip), no atomics.  Bound: HBM -- forward 2 reads + 1 write of the
// tensor, backward 4 reads + 1 write; algorithmic bytes per element 3 x / 5 x the element size.
#include "dpipe_common.h"
#include "../../include/dpipe_hip.h"

using namespace dpipe;

namespace {

constexpr int NB = 256;
constexpr int MAXVI = 4;      // channel vectors per thread: C <= 256 * 4 * VEC

__device__ __forceinline__ float silu_f(float z) { return z / (1.f + __expf(-z)); }
__device__ __forceinline__ float silu_grad(float z) { const float s = 1.f / (1.f + __expf(-z)); return s * (1.f + z * (1.f - s)); }

// Thread layout of the row-walking kernels: `vpb` threads across channel vectors, `rl` = NB / vpb row lanes.
struct Lay {
    int nvec, vpb, rl, vi;
};
__device__ __forceinline__ Lay layout(int C, int V) {
    Lay l; l.nvec = C / V; l.vpb = min(l.nvec, NB); l.rl = NB / l.vpb; l.vi = (l.nvec + l.vpb - 1) / l.vpb;
    return l;
}

// cross-row-lane reduction of per-thread channel sums through LDS: smem[rl][vpb * V] floats per quantity; the first row lane returns totals
template <int V>
__device__ __forceinline__ void lane_reduce(float* acc, const Lay& l, int tv, int rlane, float* smem) {
    // acc: V floats of this thread's channel vector
    if (l.rl == 1) return;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < V; ++j) smem[(rlane * l.vpb + tv) * V + j] = acc[j];
    __syncthreads();
    if (rlane == 0) {
        for (int r = 1; r < l.rl; ++r)
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += smem[(r * l.vpb + tv) * V + j];
    }
}

template <typename T>
__global__ void __launch_bounds__(NB) gn_nhwc_partial_kernel(const T* __restrict__ x, float* __restrict__ ws, int C, long HW, int rows_per_block, int chunks) {
    constexpr int V = Elem<T>::VEC;
    __shared__ float smem[NB * V];
    const Lay l = layout(C, V);
    const int tv = threadIdx.x % l.vpb, rlane = threadIdx.x / l.vpb;
    const long n = blockIdx.y;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(HW, r0 + rows_per_block);
    const bool live = rlane < l.rl;
    for (int i = 0; i < l.vi; ++i) {
        const int v = tv + i * l.vpb;
        float s[V], q[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { s[j] = 0.f; q[j] = 0.f; }
        if (live && v < l.nvec) {
            const T* xp = x + (n * HW) * C + (long)v * V;
            long r = r0 + rlane;
            for (; r + 3L * l.rl < r1; r += 4L * l.rl) {            // 4 independent 16-byte loads in flight per thread
                Vec16<T> a[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u].load(xp + (r + (long)u * l.rl) * C);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float f[V]; a[u].unpack(f);
#pragma unroll
                    for (int j = 0; j < V; ++j) { s[j] += f[j]; q[j] += f[j] * f[j]; }
                }
            }
            for (; r < r1; r += l.rl) {
                Vec16<T> a; a.load(xp + r * C);
                float f[V]; a.unpack(f);
#pragma unroll
                for (int j = 0; j < V; ++j) { s[j] += f[j]; q[j] += f[j] * f[j]; }
            }
        }
        lane_reduce<V>(s, l, tv, rlane, smem);
        lane_reduce<V>(q, l, tv, rlane, smem);
        if (live && rlane == 0 && v < l.nvec) {
            float* w = ws + ((n * chunks + blockIdx.x) * 2) * C + (long)v * V;
#pragma unroll
            for (int j = 0; j < V; ++j) { w[j] = s[j]; w[C + j] = q[j]; }
        }
    }
}

__global__ void __launch_bounds__(NB) gn_nhwc_final_kernel(const float* __restrict__ ws, float* __restrict__ mean, float* __restrict__ rstd,
                                                           int C, long HW, int G, int chunks, float eps) {
    __shared__ float smem[16];
    const long row = blockIdx.x;                 // n * G + g
    const long n = row / G; const int g = (int)(row % G);
    const int cpg = C / G;
    float s = 0.f, q = 0.f;
#pragma unroll 8                                                         // independent loads: 8 in flight per thread instead of one L2 round trip per iteration
    for (int idx = threadIdx.x; idx < chunks * cpg; idx += NB) {      // the group's sums need no per-channel separation; fixed thread -> index map
        const int k = idx / cpg, c = g * cpg + idx % cpg;
        s += ws[((n * chunks + k) * 2) * C + c]; q += ws[((n * chunks + k) * 2 + 1) * C + c];
    }
    s = block_sum(s, smem); q = block_sum(q, smem);
    if (threadIdx.x == 0) {
        const float m = (float)cpg * (float)HW;
        const float mu = s / m;
        mean[row] = mu; rstd[row] = rsqrtf(fmaxf(q / m - mu * mu, 0.f) + eps);
    }
}

template <typename T, typename W, int ACT>
__global__ void __launch_bounds__(NB) gn_nhwc_apply_kernel(const T* __restrict__ x, const W* __restrict__ gamma, const W* __restrict__ beta,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ y,
                                                           int C, long HW, int G, int rows_per_block) {
    constexpr int V = Elem<T>::VEC;
    const Lay l = layout(C, V);
    const int tv = threadIdx.x % l.vpb, rlane = threadIdx.x / l.vpb;
    if (rlane >= l.rl) return;
    const long n = blockIdx.y;
    const int cpg = C / G;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(HW, r0 + rows_per_block);
    for (int i = 0; i < l.vi; ++i) {
        const int v = tv + i * l.vpb;
        if (v >= l.nvec) break;
        float ga[V], be[V];
        PVec<W, V> pg, pb; pg.load(gamma, (long)v * V); pb.load(beta, (long)v * V);      // 16-byte, branch-free operand reads (dpipe_common.h, PVec)
        pg.unpack(ga); pb.unpack(be);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int c = v * V + j;
            const long row = n * G + c / cpg;
            const float rs = rstd[row];
            ga[j] = (gamma ? ga[j] : 1.f) * rs;
            be[j] = (beta ? be[j] : 0.f) - mean[row] * ga[j];
        }
        const T* xp = x + (n * HW) * C + (long)v * V;
        T* yp = y + (n * HW) * C + (long)v * V;
        long r = r0 + rlane;
        for (; r + 3L * l.rl < r1; r += 4L * l.rl) {
            Vec16<T> a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u].load(xp + (r + (long)u * l.rl) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[V]; a[u].unpack(f);
#pragma unroll
                for (int j = 0; j < V; ++j) { const float z = f[j] * ga[j] + be[j]; f[j] = ACT ? silu_f(z) : z; }
                Vec16<T> o; o.pack(f); o.store(yp + (r + (long)u * l.rl) * C);
            }
        }
        for (; r < r1; r += l.rl) {
            Vec16<T> a; a.load(xp + r * C);
            float f[V]; a.unpack(f);
#pragma unroll
            for (int j = 0; j < V; ++j) { const float z = f[j] * ga[j] + be[j]; f[j] = ACT ? silu_f(z) : z; }
            Vec16<T> o; o.pack(f); o.store(yp + r * C);
        }
    }
}

template <typename T, typename W, int ACT>
__global__ void __launch_bounds__(NB) gn_nhwc_bwd_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy, const W* __restrict__ gamma,
                                                                 const W* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 float* __restrict__ ws, int C, long HW, int G, int rows_per_block, int chunks) {
    constexpr int V = Elem<T>::VEC;
    __shared__ float smem[NB * V];
    const Lay l = layout(C, V);
    const int tv = threadIdx.x % l.vpb, rlane = threadIdx.x / l.vpb;
    const long n = blockIdx.y;
    const int cpg = C / G;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(HW, r0 + rows_per_block);
    const bool live = rlane < l.rl;
    for (int i = 0; i < l.vi; ++i) {
        const int v = tv + i * l.vpb;
        float s1[V], s2[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
        if (live && v < l.nvec) {
            float mu[V], rs[V], ga[V], be[V];
            PVec<W, V> pg, pb; pg.load(gamma, (long)v * V); pb.load(beta, (long)v * V);
            pg.unpack(ga); pb.unpack(be);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const int c = v * V + j;
                const long row = n * G + c / cpg;
                mu[j] = mean[row]; rs[j] = rstd[row];
                ga[j] = gamma ? ga[j] : 1.f; be[j] = beta ? be[j] : 0.f;
            }
            const T* xp = x + (n * HW) * C + (long)v * V;
            const T* gp = dy + (n * HW) * C + (long)v * V;
            long r = r0 + rlane;
            for (; r + (long)l.rl < r1; r += 2L * l.rl) {            // 2 rows x 2 tensors = 4 loads in flight per thread
                Vec16<T> a[2], b[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) { a[u].load(xp + (r + (long)u * l.rl) * C); b[u].load(gp + (r + (long)u * l.rl) * C); }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float fx[V], fg[V]; a[u].unpack(fx); b[u].unpack(fg);
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        const float xh = (fx[j] - mu[j]) * rs[j];
                        const float dz = ACT ? fg[j] * silu_grad(xh * ga[j] + be[j]) : fg[j];
                        s1[j] += dz; s2[j] += dz * xh;
                    }
                }
            }
            for (; r < r1; r += l.rl) {
                Vec16<T> a, b; a.load(xp + r * C); b.load(gp + r * C);
                float fx[V], fg[V]; a.unpack(fx); b.unpack(fg);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float xh = (fx[j] - mu[j]) * rs[j];
                    const float dz = ACT ? fg[j] * silu_grad(xh * ga[j] + be[j]) : fg[j];
                    s1[j] += dz; s2[j] += dz * xh;
                }
            }
        }
        lane_reduce<V>(s1, l, tv, rlane, smem);
        lane_reduce<V>(s2, l, tv, rlane, smem);
        if (live && rlane == 0 && v < l.nvec) {
            float* w = ws + ((n * chunks + blockIdx.x) * 2) * C + (long)v * V;
#pragma unroll
            for (int j = 0; j < V; ++j) { w[j] = s1[j]; w[C + j] = s2[j]; }
        }
    }
}

// one block per group: per image the group's (A, B) and, summed over the images in order, dgamma / dbeta of its channels.
// Thread t -> (channel i = t % cpgp, chunk lane t / cpgp); chunk lanes fold through LDS.  Groups of up to 256 channels.
template <typename W>
__global__ void __launch_bounds__(NB) gn_nhwc_bwd_final_kernel(const float* __restrict__ ws, const W* __restrict__ gamma, float* __restrict__ ab,
                                                               W* __restrict__ dgamma, W* __restrict__ dbeta, long N, int C, int G, int chunks, int accumulate) {
    __shared__ float red[16];
    __shared__ float part[2][NB];
    const int g = blockIdx.x;
    const int cpg = C / G;
    int cpgp = 1; while (cpgp < cpg) cpgp <<= 1;
    const int kls = NB / cpgp;
    const int i = threadIdx.x % cpgp, kl = threadIdx.x / cpgp;
    const bool on = i < cpg;
    const int c = g * cpg + (on ? i : 0);
    const float ga = on ? (gamma ? Elem<W>::to_f(gamma[c]) : 1.f) : 0.f;
    float dgs = 0.f, dbs = 0.f;
    for (long n = 0; n < N; ++n) {
        float s1 = 0.f, s2 = 0.f;
        if (on) {
#pragma unroll 8
            for (int k = kl; k < chunks; k += kls) { s1 += ws[((n * chunks + k) * 2) * C + c]; s2 += ws[((n * chunks + k) * 2 + 1) * C + c]; }
        }
        __syncthreads();
        part[0][threadIdx.x] = s1; part[1][threadIdx.x] = s2;
        __syncthreads();
        float t1 = 0.f, t2 = 0.f;
        if (kl == 0 && on)
            for (int r = 0; r < kls; ++r) { t1 += part[0][r * cpgp + i]; t2 += part[1][r * cpgp + i]; }
        dbs += t1; dgs += t2;
        const float a = block_sum(ga * t1, red), b = block_sum(ga * t2, red);
        if (threadIdx.x == 0) { ab[(n * G + g) * 2] = a; ab[(n * G + g) * 2 + 1] = b; }
    }
    if (kl == 0 && on) {
        if (dgamma) dgamma[c] = Elem<W>::from_f(accumulate ? Elem<W>::to_f(dgamma[c]) + dgs : dgs);
        if (dbeta) dbeta[c] = Elem<W>::from_f(accumulate ? Elem<W>::to_f(dbeta[c]) + dbs : dbs);
    }
}

template <typename T, typename W, int ACT>
__global__ void __launch_bounds__(NB) gn_nhwc_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, const W* __restrict__ gamma,
                                                               const W* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ ab, T* __restrict__ dx, const T* __restrict__ dadd, int C, long HW, int G,
                                                               int rows_per_block) {
    constexpr int V = Elem<T>::VEC;
    const Lay l = layout(C, V);
    const int tv = threadIdx.x % l.vpb, rlane = threadIdx.x / l.vpb;
    if (rlane >= l.rl) return;
    const long n = blockIdx.y;
    const int cpg = C / G;
    const float inv_m = 1.f / ((float)cpg * (float)HW);
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(HW, r0 + rows_per_block);
    for (int i = 0; i < l.vi; ++i) {
        const int v = tv + i * l.vpb;
        if (v >= l.nvec) break;
        float mu[V], rs[V], ga[V], be[V], ca[V], cb[V];
        PVec<W, V> pg, pb; pg.load(gamma, (long)v * V); pb.load(beta, (long)v * V);
        pg.unpack(ga); pb.unpack(be);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int c = v * V + j;
            const long row = n * G + c / cpg;
            mu[j] = mean[row]; rs[j] = rstd[row];
            ga[j] = gamma ? ga[j] : 1.f; be[j] = beta ? be[j] : 0.f;
            ca[j] = ab[row * 2] * inv_m; cb[j] = ab[row * 2 + 1] * inv_m;
        }
        const T* xp = x + (n * HW) * C + (long)v * V;
        const T* gp = dy + (n * HW) * C + (long)v * V;
        T* op = dx + (n * HW) * C + (long)v * V;
        // gradient that reached x around the norm (residual branch): read with the x / dy loads of its row, branch-free (absent: the zero pad, row pitch 0)
        const bool ha = dadd != nullptr;
        const T* ap = ha ? dadd + (n * HW) * C + (long)v * V : reinterpret_cast<const T*>(g_param_pad);
        const long apitch = ha ? C : 0;
        long r = r0 + rlane;
        for (; r + (long)l.rl < r1; r += 2L * l.rl) {
            Vec16<T> a[2], b[2], e[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { a[u].load(xp + (r + (long)u * l.rl) * C); b[u].load(gp + (r + (long)u * l.rl) * C); e[u].load(ap + (r + (long)u * l.rl) * apitch); }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float fx[V], fg[V], fe[V]; a[u].unpack(fx); b[u].unpack(fg); e[u].unpack(fe);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float xh = (fx[j] - mu[j]) * rs[j];
                    const float dz = ACT ? fg[j] * silu_grad(xh * ga[j] + be[j]) : fg[j];
                    fx[j] = rs[j] * (ga[j] * dz - ca[j] - xh * cb[j]) + (ha ? fe[j] : 0.f);
                }
                Vec16<T> o; o.pack(fx); o.store(op + (r + (long)u * l.rl) * C);
            }
        }
        for (; r < r1; r += l.rl) {
            Vec16<T> a, b, e; a.load(xp + r * C); b.load(gp + r * C); e.load(ap + r * apitch);
            float fx[V], fg[V], fe[V]; a.unpack(fx); b.unpack(fg); e.unpack(fe);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float xh = (fx[j] - mu[j]) * rs[j];
                const float dz = ACT ? fg[j] * silu_grad(xh * ga[j] + be[j]) : fg[j];
                fx[j] = rs[j] * (ga[j] * dz - ca[j] - xh * cb[j]) + (ha ? fe[j] : 0.f);
            }
            Vec16<T> o; o.pack(fx); o.store(op + r * C);
        }
    }
}

// rows per block: ~512 blocks over the tensor (the two "final" kernels walk the chunk list), at least 8 rows per row lane (4 loads in flight)
void plan(int C, long HW, long N, int V, int& rows_per_block, int& chunks) {
    const int nvec = C / V, vpb = nvec < NB ? nvec : NB, rl = NB / vpb;
    long rpb = cdiv(HW * N, 512);
    if (rpb < 8L * rl) rpb = 8L * rl;
    if (rpb > HW) rpb = HW;
    rows_per_block = (int)rpb; chunks = (int)cdiv(HW, rpb);
}

}  // namespace

#define BAD(msg) do { set_last_error(msg); return DPIPE_ERR_ARG; } while (0)

extern "C" {

long dpipe_groupnorm_nhwc_workspace_floats(long N, int C, long HW, int G) {
    int rpb, chunks; plan(C, HW, N, 4, rpb, chunks);      // fp32 vectors give the larger chunk count
    int rpb2, chunks2; plan(C, HW, N, 8, rpb2, chunks2);
    const long ch = chunks > chunks2 ? chunks : chunks2;
    return N * ch * 2 * C + N * G * 2;
}

int dpipe_groupnorm_nhwc_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd, float* workspace,
                             long N, int C, long HW, int G, float eps, int act, int dtype, int wdtype, void* stream) {
    if (!x || !y || !mean || !rstd || !workspace || N <= 0 || C <= 0 || HW <= 0 || G <= 0 || C % G) BAD("dpipe_groupnorm_nhwc_fwd: bad argument");
    if (ablated(ABL_GN)) return DPIPE_OK;
    const int V = dtype == DPIPE_BF16 ? 8 : 4;
    if ((dtype != DPIPE_BF16 && dtype != DPIPE_F32) || (wdtype != DPIPE_BF16 && wdtype != DPIPE_F32)) BAD("dpipe_groupnorm_nhwc_fwd: dtype");
    if (C % V || C / V > NB * MAXVI || ((uintptr_t)x & 15) || ((uintptr_t)y & 15) || !aligned16(gamma) || !aligned16(beta)) BAD("dpipe_groupnorm_nhwc_fwd: C must be a multiple of the 16-byte vector (<= 8192 channels), 16-byte aligned tensors");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rpb, chunks; plan(C, HW, N, V, rpb, chunks);
    dim3 grid(chunks, (unsigned)N);
#define GN_FWD(TT, WW, AA) do { \
        gn_nhwc_partial_kernel<TT><<<grid, NB, 0, s>>>((const TT*)x, workspace, C, HW, rpb, chunks); \
        gn_nhwc_final_kernel<<<(unsigned)(N * G), NB, 0, s>>>(workspace, mean, rstd, C, HW, G, chunks, eps); \
        gn_nhwc_apply_kernel<TT, WW, AA><<<grid, NB, 0, s>>>((const TT*)x, (const WW*)gamma, (const WW*)beta, mean, rstd, (TT*)y, C, HW, G, rpb); } while (0)
#define GN_FWD_ACT(TT, WW) do { if (act == DPIPE_ACT_SILU) GN_FWD(TT, WW, 1); else if (act == DPIPE_ACT_NONE) GN_FWD(TT, WW, 0); else BAD("dpipe_groupnorm_nhwc_fwd: act"); } while (0)
    if (dtype == DPIPE_BF16 && wdtype == DPIPE_BF16) GN_FWD_ACT(bf16_t, bf16_t);
    else if (dtype == DPIPE_BF16) GN_FWD_ACT(bf16_t, float);
    else if (wdtype == DPIPE_F32) GN_FWD_ACT(float, float);
    else BAD("dpipe_groupnorm_nhwc_fwd: fp32 activations need fp32 parameters");
    return check_launch("dpipe_groupnorm_nhwc_fwd");
}

int dpipe_groupnorm_nhwc_bwd(const void* x, const void* dy, const void* gamma, const void* beta, const float* mean, const float* rstd,
                             void* dx, void* dgamma, void* dbeta, float* workspace, long N, int C, long HW, int G, int act, int dtype, int wdtype,
                             int accumulate_params, const void* dx_add, void* stream) {
    if (!x || !dy || !dx || !mean || !rstd || !workspace || N <= 0 || C <= 0 || HW <= 0 || G <= 0 || C % G) BAD("dpipe_groupnorm_nhwc_bwd: bad argument");
    if (ablated(ABL_GN)) return DPIPE_OK;
    const int V = dtype == DPIPE_BF16 ? 8 : 4;
    if ((dtype != DPIPE_BF16 && dtype != DPIPE_F32) || (wdtype != DPIPE_BF16 && wdtype != DPIPE_F32)) BAD("dpipe_groupnorm_nhwc_bwd: dtype");
    if (C % V || C / V > NB * MAXVI || C / G > NB || ((uintptr_t)x & 15) || ((uintptr_t)dy & 15) || ((uintptr_t)dx & 15) || !aligned16(gamma) || !aligned16(beta) || !aligned16(dx_add)) BAD("dpipe_groupnorm_nhwc_bwd: alignment / channel count");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rpb, chunks; plan(C, HW, N, V, rpb, chunks);
    dim3 grid(chunks, (unsigned)N);
    float* ab = workspace + N * (long)chunks * 2 * C;
#define GN_BWD(TT, WW, AA) do { \
        gn_nhwc_bwd_partial_kernel<TT, WW, AA><<<grid, NB, 0, s>>>((const TT*)x, (const TT*)dy, (const WW*)gamma, (const WW*)beta, mean, rstd, workspace, C, HW, G, rpb, chunks); \
        gn_nhwc_bwd_final_kernel<WW><<<(unsigned)G, NB, 0, s>>>(workspace, (const WW*)gamma, ab, (WW*)dgamma, (WW*)dbeta, N, C, G, chunks, accumulate_params); \
        gn_nhwc_bwd_apply_kernel<TT, WW, AA><<<grid, NB, 0, s>>>((const TT*)x, (const TT*)dy, (const WW*)gamma, (const WW*)beta, mean, rstd, ab, (TT*)dx, (const TT*)dx_add, C, HW, G, rpb); } while (0)
#define GN_BWD_ACT(TT, WW) do { if (act == DPIPE_ACT_SILU) GN_BWD(TT, WW, 1); else if (act == DPIPE_ACT_NONE) GN_BWD(TT, WW, 0); else BAD("dpipe_groupnorm_nhwc_bwd: act"); } while (0)
    if (dtype == DPIPE_BF16 && wdtype == DPIPE_BF16) GN_BWD_ACT(bf16_t, bf16_t);
    else if (dtype == DPIPE_BF16) GN_BWD_ACT(bf16_t, float);
    else if (wdtype == DPIPE_F32) GN_BWD_ACT(float, float);
    else BAD("dpipe_groupnorm_nhwc_bwd: fp32 activations need fp32 parameters");
    return check_launch("dpipe_groupnorm_nhwc_bwd");
}

}  // extern "C"
