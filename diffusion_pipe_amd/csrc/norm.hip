// norm.hip -- row-wise normalisation / rotary / softmax kernels (gfx950, wave64).
//
//   K2  RMSNorm fwd+bwd                 (models/wan/model.py:70-86; per-head form: hunyuan_image_modeling.py:98-103)
//   K5  LayerNorm (+affine) + AdaLN modulate fwd+bwd  (models/wan/model.py:89-99,295-309)
//   K3  RoPE fwd+bwd (complex multiply on interleaved or split pairs)  (models/wan/model.py:40-67)
//       row softmax fwd+bwd (used only by the fp32 unfused attention parity path)
//       batched 2-D transpose (GEMM fallback for MN-contiguous operands)
//
// Row kernels: LPR lanes cooperate on one row (LPR = 16 for head-dim rows, 64 otherwise),
// reductions by wavefront shuffles, statistics in fp32, 16-byte accesses.
#include "dpipe_common.h"
#include "../../include/dpipe_hip.h"
#include <stdlib.h>

using namespace dpipe;

namespace {

constexpr int NB = 256;  // threads per block

template <int LPR> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int LPR> __device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ------------------------------------------------------------------ RMSNorm
// y = (T)(x * rsqrt(mean(x^2) + eps)) * w        rstd saved per row for backward
template <typename T, typename W, int LPR>
__global__ void __launch_bounds__(NB) rmsnorm_fwd_kernel(const T* __restrict__ x, const W* __restrict__ w, T* __restrict__ y,
                                                         float* __restrict__ rstd_out, long rows, int cols, float eps) {
    constexpr int V = Elem<T>::VEC;
    constexpr int RPB = NB / LPR;
    const int sub = threadIdx.x % LPR;
    const long row = (long)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live = row < rows;
    const T* xr = x + (live ? row : 0) * (long)cols;
    float ss = 0.f;
    for (int c = sub * V; c < cols; c += LPR * V) {
        Vec16<T> v; v.load(xr + c);
        float f[V]; v.unpack(f);
#pragma unroll
        for (int j = 0; j < V; ++j) ss += f[j] * f[j];
    }
    ss = group_sum<LPR>(ss);
    const float rstd = rsqrtf(ss / (float)cols + eps);
    if (!live) return;
    if (sub == 0 && rstd_out) rstd_out[row] = rstd;
    T* yr = y + row * (long)cols;
    for (int c = sub * V; c < cols; c += LPR * V) {
        Vec16<T> v; v.load(xr + c);
        PVec<W, V> pw; pw.load(w, c);                                  // 16-byte, branch-free weight read (dpipe_common.h, PVec)
        float f[V], fw[V]; v.unpack(f); pw.unpack(fw);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float n = Elem<T>::to_f(Elem<T>::from_f(f[j] * rstd));  // ".type_as(x)" rounding point of the reference
            f[j] = w ? n * fw[j] : n;
        }
        v.pack(f); v.store(yr + c);
    }
}
// dx = rstd * (g*w - xhat * mean(g*w*xhat)),  xhat = x*rstd
template <typename T, typename W, int LPR>
__global__ void __launch_bounds__(NB) rmsnorm_bwd_dx_kernel(const T* __restrict__ x, const W* __restrict__ w, const T* __restrict__ gy,
                                                            const float* __restrict__ rstd_in, T* __restrict__ gx, long rows, int cols) {
    constexpr int V = Elem<T>::VEC;
    constexpr int RPB = NB / LPR;
    const int sub = threadIdx.x % LPR;
    const long row = (long)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live = row < rows;
    const long r = live ? row : 0;
    const T* xr = x + r * (long)cols; const T* gr = gy + r * (long)cols;
    const float rstd = rstd_in[r];
    float dot = 0.f;
    for (int c = sub * V; c < cols; c += LPR * V) {
        Vec16<T> vx, vg; vx.load(xr + c); vg.load(gr + c);
        PVec<W, V> pw; pw.load(w, c);
        float fx[V], fg[V], fw[V]; vx.unpack(fx); vg.unpack(fg); pw.unpack(fw);
#pragma unroll
        for (int j = 0; j < V; ++j) dot += fg[j] * (w ? fw[j] : 1.f) * fx[j] * rstd;
    }
    dot = group_sum<LPR>(dot) / (float)cols;
    if (!live) return;
    T* o = gx + row * (long)cols;
    for (int c = sub * V; c < cols; c += LPR * V) {
        Vec16<T> vx, vg; vx.load(xr + c); vg.load(gr + c);
        PVec<W, V> pw; pw.load(w, c);
        float fx[V], fg[V], fw[V]; vx.unpack(fx); vg.unpack(fg); pw.unpack(fw);
#pragma unroll
        for (int j = 0; j < V; ++j) fg[j] = rstd * (fg[j] * (w ? fw[j] : 1.f) - fx[j] * rstd * dot);
        vg.pack(fg); vg.store(o + c);
    }
}
// ------------------------------------------------------------------ K2 + K3 fused: RMSNorm -> RoPE in ONE pass (models/wan/model.py:124-125,139-140: q = rope(norm_q(q(x)));
// per-head form: hunyuan_image_modeling.py:181-190, diffusers FluxAttnProcessor): y = rot_s(type_as(x * rstd) * w) -- the normalised, weighted row never round-trips
// through memory (and is not rounded to the storage type before the rotation).  A row is one normalisation group: the whole token (G = 1, cols = H D: Wan) or one head
// (G = H, cols = D: Flux / HunyuanVideo); its token is s = (row / G) % S, the rotation (interleaved pairs (2i, 2i + 1), angle table row s + tok_off, column (c % D) / 2)
// applies to tokens s < rope_tokens (HunyuanVideo's single-stream blocks rotate the image tokens of an [image ; text] sequence only).  Input rows may be strided views
// of a fused QKV projection (token pitch x_ts elements, heads D apart); outputs and gradients are dense.
struct RopeGeom { const float* cs; const float* sn; long S; int G; int D; long tok_off; long rope_tokens; long x_ts; };

template <typename T, typename W, int LPR>
__global__ void __launch_bounds__(NB) rmsnorm_rope_fwd_kernel(const T* __restrict__ x, const W* __restrict__ w, T* __restrict__ y, float* __restrict__ rstd_out,
                                                              long rows, int cols, float eps, const RopeGeom rg) {
    constexpr int V = Elem<T>::VEC;
    constexpr int RPB = NB / LPR;
    const int sub = threadIdx.x % LPR;
    const long row = (long)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live = row < rows;
    const long r = live ? row : 0;
    const long tok = r / rg.G;
    const T* xr = x + tok * rg.x_ts + (r - tok * rg.G) * (long)cols;
    float ss = 0.f;
    for (int c = sub * V; c < cols; c += LPR * V) {
        Vec16<T> v; v.load(xr + c);
        float f[V]; v.unpack(f);
#pragma unroll
        for (int j = 0; j < V; ++j) ss += f[j] * f[j];
    }
    ss = group_sum<LPR>(ss);
    const float rstd = rsqrtf(ss / (float)cols + eps);
    if (!live) return;
    if (sub == 0 && rstd_out) rstd_out[row] = rstd;
    const long s = tok % rg.S;
    const bool rot = s < rg.rope_tokens;
    const float* cs = rg.cs + (s + rg.tok_off) * (rg.D / 2);
    const float* sn = rg.sn + (s + rg.tok_off) * (rg.D / 2);
    T* yr = y + row * (long)cols;
    for (int c = sub * V; c < cols; c += LPR * V) {
        Vec16<T> v; v.load(xr + c);
        PVec<W, V> pw; pw.load(w, c);
        float f[V], fw[V]; v.unpack(f); pw.unpack(fw);
        const int cd = c % rg.D;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float n = Elem<T>::to_f(Elem<T>::from_f(f[j] * rstd));  // ".type_as(x)" rounding point of the reference norm
            f[j] = w ? n * fw[j] : n;
        }
        if (rot) {
#pragma unroll
            for (int j = 0; j < V; j += 2) {
                const float co = cs[(cd + j) / 2], si = sn[(cd + j) / 2];
                const float a = f[j], b = f[j + 1];
                f[j] = a * co - b * si; f[j + 1] = a * si + b * co;
            }
        }
        v.pack(f); v.store(yr + c);
    }
}
// g_t = rot_s^T(gy) (inverse rotation), then the RMSNorm backward on it: dx = rstd * (g_t w - xhat * mean(g_t w xhat))
template <typename T, typename W, int LPR>
__global__ void __launch_bounds__(NB) rmsnorm_rope_bwd_dx_kernel(const T* __restrict__ x, const W* __restrict__ w, const T* __restrict__ gy,
                                                                 const float* __restrict__ rstd_in, T* __restrict__ gx, long rows, int cols, const RopeGeom rg) {
    constexpr int V = Elem<T>::VEC;
    constexpr int RPB = NB / LPR;
    const int sub = threadIdx.x % LPR;
    const long row = (long)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live = row < rows;
    const long r = live ? row : 0;
    const long tok = r / rg.G;
    const T* xr = x + tok * rg.x_ts + (r - tok * rg.G) * (long)cols;
    const T* gr = gy + r * (long)cols;
    const float rstd = rstd_in[r];
    const long s = tok % rg.S;
    const bool rot = s < rg.rope_tokens;
    const float* cs = rg.cs + (s + rg.tok_off) * (rg.D / 2);
    const float* sn = rg.sn + (s + rg.tok_off) * (rg.D / 2);
    auto unrotate = [&](float* fg, int c) {
        if (!rot) return;
        const int cd = c % rg.D;
#pragma unroll
        for (int j = 0; j < V; j += 2) {
            const float co = cs[(cd + j) / 2], si = sn[(cd + j) / 2];
            const float a = fg[j], b = fg[j + 1];
            fg[j] = a * co + b * si; fg[j + 1] = b * co - a * si;
        }
    };
    float dot = 0.f;
    for (int c = sub * V; c < cols; c += LPR * V) {
        Vec16<T> vx, vg; vx.load(xr + c); vg.load(gr + c);
        PVec<W, V> pw; pw.load(w, c);
        float fx[V], fg[V], fw[V]; vx.unpack(fx); vg.unpack(fg); pw.unpack(fw);
        unrotate(fg, c);
#pragma unroll
        for (int j = 0; j < V; ++j) dot += fg[j] * (w ? fw[j] : 1.f) * fx[j] * rstd;
    }
    dot = group_sum<LPR>(dot) / (float)cols;
    if (!live) return;
    T* o = gx + row * (long)cols;
    for (int c = sub * V; c < cols; c += LPR * V) {
        Vec16<T> vx, vg; vx.load(xr + c); vg.load(gr + c);
        PVec<W, V> pw; pw.load(w, c);
        float fx[V], fg[V], fw[V]; vx.unpack(fx); vg.unpack(fg); pw.unpack(fw);
        unrotate(fg, c);
#pragma unroll
        for (int j = 0; j < V; ++j) fg[j] = rstd * (fg[j] * (w ? fw[j] : 1.f) - fx[j] * rstd * dot);
        vg.pack(fg); vg.store(o + c);
    }
}
// dw partials: partial[slab][c] = sum over the slab's rows of rot^T(gy)[r, c] * x[r, c] * rstd[r]   (block / slab geometry of colreduce_kernel below)
template <typename T>
__global__ void __launch_bounds__(NB) rmsnorm_rope_dw_kernel(const T* __restrict__ x, const T* __restrict__ gy, const float* __restrict__ rstd, long rows, int cols,
                                                             int slabs, float* __restrict__ p0, const RopeGeom rg) {
    constexpr int V = Elem<T>::VEC;
    constexpr int CT = 32, RT = 8;
    __shared__ float red[RT][CT * V];
    const int ct = threadIdx.x % CT, rl = threadIdx.x / CT;
    const int c = (blockIdx.x * CT + ct) * V;
    const bool live = c < cols;
    const long rps = cdiv(rows, slabs);
    const long r0 = blockIdx.y * rps, r1 = min(r0 + rps, rows);
    float a0[V];
#pragma unroll
    for (int j = 0; j < V; ++j) a0[j] = 0.f;
    if (live) {
        const int cd = c % rg.D;
#pragma unroll 4
        for (long row = r0 + rl; row < r1; row += RT) {
            const long tok = row / rg.G;
            Vec16<T> vx, vg; vx.load(x + tok * rg.x_ts + (row - tok * rg.G) * (long)cols + c); vg.load(gy + row * (long)cols + c);
            float fx[V], fg[V]; vx.unpack(fx); vg.unpack(fg);
            const long s = tok % rg.S;
            if (s < rg.rope_tokens) {
                const float* cs = rg.cs + (s + rg.tok_off) * (rg.D / 2);
                const float* sn = rg.sn + (s + rg.tok_off) * (rg.D / 2);
#pragma unroll
                for (int j = 0; j < V; j += 2) {
                    const float co = cs[(cd + j) / 2], si = sn[(cd + j) / 2];
                    const float a = fg[j], b = fg[j + 1];
                    fg[j] = a * co + b * si; fg[j + 1] = b * co - a * si;
                }
            }
            const float rs = rstd[row];
#pragma unroll
            for (int j = 0; j < V; ++j) a0[j] += fg[j] * Elem<T>::to_f(Elem<T>::from_f(fx[j] * rs));      // the weight multiplies the ROUNDED normalised value in the forward
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) red[rl][ct * V + j] = a0[j];
    __syncthreads();
    const int col_in = threadIdx.x;
    if (col_in < CT * V) {
        const int cc = blockIdx.x * CT * V + col_in;
        if (cc < cols) {
            float s0 = 0.f;
#pragma unroll
            for (int r = 0; r < RT; ++r) s0 += red[r][col_in];
            p0[(long)blockIdx.y * cols + cc] = s0;
        }
    }
}

// column reductions over a slab of rows: partial[slab][c] = sum_r f(r, c).  grid = (colblocks, slabs, groups)
// MODE 0: RMSNorm dw   = sum gy * x * rstd
// MODE 1: LN (dgamma, dbeta) = (sum dn * xhat, sum dn)      with dn = gy * (1 + scale)
// MODE 2: AdaLN (dscale, dshift) = (sum gy * n, sum gy)      n = xhat*gamma + beta
// MODE 3: plain column sum of x (bias gradients); gy / statistics unused
// Block = 32 column vectors (512 B of a row: whole cache lines) x 8 row lanes; a slab is ~32 rows so a [1024, 1280]
// operand launches 5 x 8 blocks of 128 rows, 16 rows per thread (unrolled 4x for loads in flight); row lanes combine through LDS.
constexpr int CR_CT = 32, CR_RT = 8;
template <typename T, typename W, typename M, int MODE>
__global__ void __launch_bounds__(NB) colreduce_kernel(const T* __restrict__ x, const T* __restrict__ gy, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const W* __restrict__ gamma, const W* __restrict__ beta,
                                                       const M* __restrict__ scale, long rows_per_group, int cols, long ld, int slabs,
                                                       float* __restrict__ p0, float* __restrict__ p1) {
    constexpr int V = Elem<T>::VEC;
    __shared__ float red[2][CR_RT][CR_CT * V];
    const int ct = threadIdx.x % CR_CT, rl = threadIdx.x / CR_CT;
    const int c = (blockIdx.x * CR_CT + ct) * V;
    const bool live = c < cols;
    const long grp = blockIdx.z;
    const long rps = cdiv(rows_per_group, slabs);
    const long r0 = blockIdx.y * rps, r1 = min(r0 + rps, rows_per_group);
    float a0[V], a1[V], fgam[V], fbet[V], fsc[V];
    {
        const long cs = live ? c : 0;                                   // 16-byte, branch-free operand reads (dpipe_common.h, PVec)
        PVec<W, V> pg, pb; PVec<M, V> ps; pg.load(gamma, cs); pb.load(beta, cs); ps.load(scale, grp * cols + cs);
        pg.unpack(fgam); pb.unpack(fbet); ps.unpack(fsc);
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
        a0[j] = 0.f; a1[j] = 0.f;
        fgam[j] = gamma ? fgam[j] : 1.f;
        fbet[j] = beta ? fbet[j] : 0.f;
        fsc[j] = scale ? 1.f + fsc[j] : 1.f;
    }
    if (live) {
#pragma unroll 4
        for (long r = r0 + rl; r < r1; r += CR_RT) {
            const long row = grp * rows_per_group + r;
            Vec16<T> vx; vx.load(x + row * ld + c);
            float fx[V]; vx.unpack(fx);
            if (MODE == 3) {
#pragma unroll
                for (int j = 0; j < V; ++j) a0[j] += fx[j];
            } else {
                Vec16<T> vg; vg.load(gy + row * ld + c);
                float fg[V]; vg.unpack(fg);
                const float mu = mean ? mean[row] : 0.f, rs = rstd[row];
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float xhat = (fx[j] - mu) * rs;
                    if (MODE == 0) { a0[j] += fg[j] * xhat; }
                    else if (MODE == 1) { const float dn = fg[j] * fsc[j]; a0[j] += dn * xhat; a1[j] += dn; }
                    else { a0[j] += fg[j] * (xhat * fgam[j] + fbet[j]); a1[j] += fg[j]; }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
        red[0][rl][ct * V + j] = a0[j];
        if (MODE == 1 || MODE == 2) red[1][rl][ct * V + j] = a1[j];
    }
    __syncthreads();
    // CR_CT * V columns, one per thread (bf16: 256 columns = 256 threads; fp32: 128 columns)
    const int col_in = threadIdx.x;
    if (col_in < CR_CT * V) {
        const int cc = blockIdx.x * CR_CT * V + col_in;
        if (cc < cols) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int r = 0; r < CR_RT; ++r) { s0 += red[0][r][col_in]; if (MODE == 1 || MODE == 2) s1 += red[1][r][col_in]; }
            const long o = (grp * slabs + blockIdx.y) * cols + cc;
            p0[o] = s0;
            if (MODE == 1 || MODE == 2) p1[o] = s1;
        }
    }
}
// two reductions that share a shape (dgamma + dbeta, dscale + dshift) in one launch; out1 may be null
template <typename O>
__global__ void __launch_bounds__(NB) slabsum2_kernel(const float* __restrict__ p0, const float* __restrict__ p1, O* __restrict__ out0,
                                                      O* __restrict__ out1, long groups, int cols, int slabs, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups * cols) return;
    const long g = i / cols; const int c = (int)(i - g * cols);
    float a = 0.f, b = 0.f;
#pragma unroll 8
    for (int s = 0; s < slabs; ++s) {
        a += p0[(g * slabs + s) * cols + c];
        if (out1) b += p1[(g * slabs + s) * cols + c];
    }
    if (accumulate) { a += Elem<O>::to_f(out0[i]); if (out1) b += Elem<O>::to_f(out1[i]); }
    out0[i] = Elem<O>::from_f(a);
    if (out1) out1[i] = Elem<O>::from_f(b);
}
// the same over MANY slabs (the fused LayerNorm backward leaves rows / 16 partial rows): 32 columns x 8 slab lanes per block, each lane walks every 8th slab
// with 8 loads in flight, the lanes combine through LDS -- deterministic (fixed lane / slab order)
template <typename O>
__global__ void __launch_bounds__(NB) slabsum2p_kernel(const float* __restrict__ p0, const float* __restrict__ p1, O* __restrict__ out0,
                                                       O* __restrict__ out1, int cols, int slabs, int accumulate) {
    __shared__ float red[2][8][32];
    const int ct = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + ct;
    const long g = blockIdx.y;
    float a = 0.f, b = 0.f;
    if (c < cols) {
#pragma unroll 16
        for (int s = sl; s < slabs; s += 8) {                 // (16 x 2 loads in flight per lane: 256 slabs in two round trips)
            a += p0[(g * slabs + s) * cols + c];
            b += p1[(g * slabs + s) * cols + c];
        }
    }
    red[0][sl][ct] = a; red[1][sl][ct] = b;
    __syncthreads();
    if (threadIdx.x < 32 && c < cols) {
        float u = 0.f, v = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) { u += red[0][r][ct]; v += red[1][r][ct]; }
        const long i = g * cols + c;
        if (accumulate) { u += Elem<O>::to_f(out0[i]); if (out1) v += Elem<O>::to_f(out1[i]); }
        out0[i] = Elem<O>::from_f(u);
        if (out1) out1[i] = Elem<O>::from_f(v);
    }
}
template <typename O>
__global__ void __launch_bounds__(NB) slabsum_kernel(const float* __restrict__ partial, O* __restrict__ out, long groups, int cols, int slabs, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups * cols) return;
    const long g = i / cols; const int c = (int)(i - g * cols);
    float a = 0.f;
#pragma unroll 8
    for (int s = 0; s < slabs; ++s) a += partial[(g * slabs + s) * cols + c];     // 8 independent loads in flight
    if (accumulate) a += Elem<O>::to_f(out[i]);
    out[i] = Elem<O>::from_f(a);
}

// ------------------------------------------------- LayerNorm + AdaLN modulate
// n = (x - mean) * rstd [* gamma + beta] ;  y = n * (1 + scale[b]) + shift[b]
//
// Round 6: the per-column operands (gamma, beta, scale, shift) are read as 16-byte vectors through PVec -- BRANCH-FREE, an absent operand reads a zero
// pad -- and every load of a row (x, the operands, the statistics) is issued before the first use.  The previous form `gamma ? to_f(gamma[c + j]) : 1.f`
// per element compiled to a branch around every 2-byte load with `s_waitcnt vmcnt(0)` behind it: 160 DEPENDENT round trips per wave in the forward, 128 - 640
// in the backward (ISA counts in profiles/r6m_norm_isa_before_after.txt); the kernels were bound by that chain, not by HBM.
// NK > 0: the row fits NK vectors per lane (cols <= LPR * V * NK, NK <= 4: every SDXL width) and stays in registers across the passes; lanes past the end
// of the row read column 0 and are masked.  NK = 0: any width, three passes over the row (L2-resident re-reads).
template <typename T, typename W, typename M, int LPR, int NK>
__global__ void __launch_bounds__(NB) lnmod_fwd_kernel(const T* __restrict__ x, const W* __restrict__ gamma, const W* __restrict__ beta,
                                                       const M* __restrict__ scale, const M* __restrict__ shift, T* __restrict__ y,
                                                       float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                       long rows, int cols, long rows_per_mod, float eps) {
    constexpr int V = Elem<T>::VEC;
    constexpr int RPB = NB / LPR;
    const int sub = threadIdx.x % LPR;
    const long row = (long)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live = row < rows;
    const long r = live ? row : 0;
    const T* xr = x + r * (long)cols;
    T* yr = y + r * (long)cols;
    const long mb = (r / rows_per_mod) * (long)cols;
    const bool hg = gamma != nullptr, hb = beta != nullptr, hs = scale != nullptr, hh = shift != nullptr;
    auto affine = [&](float* f, const PVec<W, V>& pg, const PVec<W, V>& pb, const PVec<M, V>& ps, const PVec<M, V>& ph, float mu, float rstd) {
        float g[V], bt[V], sc[V], sh[V];
        pg.unpack(g); pb.unpack(bt); ps.unpack(sc); ph.unpack(sh);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float n = (f[j] - mu) * rstd;
            n = hg ? n * g[j] + (hb ? bt[j] : 0.f) : n;
            n = hs ? n * (1.f + sc[j]) : n;
            n = hh ? n + sh[j] : n;
            f[j] = n;
        }
    };
    if constexpr (NK > 0) {
        Vec16<T> rv[NK]; PVec<W, V> pg[NK], pb[NK]; PVec<M, V> ps[NK], ph[NK];
        bool in[NK]; int cc[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int c = (sub + k * LPR) * V;
            in[k] = c < cols; cc[k] = in[k] ? c : 0;
            rv[k].load(xr + cc[k]);
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) { pg[k].load(gamma, cc[k]); pb[k].load(beta, cc[k]); ps[k].load(scale, mb + cc[k]); ph[k].load(shift, mb + cc[k]); }
        __builtin_amdgcn_sched_barrier(0);   // every load of the row is in flight before the first use (the machine scheduler would otherwise delay some behind the arithmetic)
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            float f[V]; rv[k].unpack(f);
#pragma unroll
            for (int j = 0; j < V; ++j) s += in[k] ? f[j] : 0.f;
        }
        const float mu = group_sum<LPR>(s) / (float)cols;
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            float f[V]; rv[k].unpack(f);
#pragma unroll
            for (int j = 0; j < V; ++j) { const float d = f[j] - mu; ss += in[k] ? d * d : 0.f; }
        }
        const float rstd = rsqrtf(group_sum<LPR>(ss) / (float)cols + eps);
        if (!live) return;
        if (sub == 0) { if (mean_out) mean_out[row] = mu; if (rstd_out) rstd_out[row] = rstd; }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            float f[V]; rv[k].unpack(f);
            affine(f, pg[k], pb[k], ps[k], ph[k], mu, rstd);
            rv[k].pack(f);
            if (in[k]) rv[k].store(yr + cc[k]);
        }
    } else {
        float s = 0.f;
        for (int c = sub * V; c < cols; c += LPR * V) {
            Vec16<T> v; v.load(xr + c); float f[V]; v.unpack(f);
#pragma unroll
            for (int j = 0; j < V; ++j) s += f[j];
        }
        const float mu = group_sum<LPR>(s) / (float)cols;
        float ss = 0.f;
        for (int c = sub * V; c < cols; c += LPR * V) {
            Vec16<T> v; v.load(xr + c); float f[V]; v.unpack(f);
#pragma unroll
            for (int j = 0; j < V; ++j) { const float d = f[j] - mu; ss += d * d; }
        }
        const float rstd = rsqrtf(group_sum<LPR>(ss) / (float)cols + eps);
        if (!live) return;
        if (sub == 0) { if (mean_out) mean_out[row] = mu; if (rstd_out) rstd_out[row] = rstd; }
        for (int c = sub * V; c < cols; c += LPR * V) {
            Vec16<T> v; PVec<W, V> pg, pb; PVec<M, V> ps, ph;
            v.load(xr + c); pg.load(gamma, c); pb.load(beta, c); ps.load(scale, mb + c); ph.load(shift, mb + c);
            float f[V]; v.unpack(f);
            affine(f, pg, pb, ps, ph, mu, rstd);
            v.pack(f); v.store(yr + c);
        }
    }
}
// dxhat = gy * (1+scale) * gamma ;  dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)) (+ gadd: the gradient that reached x
// through the residual branch around this norm, added here instead of by a separate elementwise kernel)
//
// FUSE != 0 (NK > 0 only: rows that fit the register cache): the per-column parameter-gradient sums ride this pass as well -- every lane keeps
// fp32 accumulators for the columns it owns over the RW rows its wave walks, the block's waves combine through LDS and ONE partial row per block goes
// to the workspace (p0 / p1, [groups][slabs][cols]); slabsum2p then finishes.  The separate column-reduction launch, which re-read x and gy, is gone:
//   FUSE 1: (dgamma, dbeta):  p0 += dn * xhat, p1 += dn         with dn = gy * (1 + scale)
//   FUSE 2: (dscale, dshift): p0 += gy * (xhat * gamma + beta), p1 += gy
// Operand loads: see lnmod_fwd_kernel (vectors, branch-free, all issued ahead of the first use; gamma / beta once per block, the rest once per row).
template <typename T, typename W, typename M, int LPR, int NK, int FUSE = 0, int RW = 1>
__global__ void __launch_bounds__(NB) lnmod_bwd_dx_kernel(const T* __restrict__ x, const T* __restrict__ gy, const W* __restrict__ gamma,
                                                          const M* __restrict__ scale, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          T* __restrict__ gx, const T* __restrict__ gadd, long rows, int cols, long rows_per_mod,
                                                          const W* __restrict__ beta = nullptr, float* __restrict__ p0 = nullptr, float* __restrict__ p1 = nullptr,
                                                          int slabs = 1) {
    static_assert(FUSE == 0 || NK > 0, "the fused column partials need the register-cached row");
    constexpr int V = Elem<T>::VEC;
    constexpr int RPB = NB / LPR;
    constexpr int NA = NK > 0 ? NK : 1;
    const int sub = threadIdx.x % LPR;
    const int wave = threadIdx.x / LPR;
    const bool hg = gamma != nullptr, hs = scale != nullptr, hb = beta != nullptr, ha = gadd != nullptr;
    float a0[FUSE ? NA : 1][V], a1[FUSE ? NA : 1][V];
    if (FUSE) {
#pragma unroll
        for (int k = 0; k < NA; ++k)
#pragma unroll
            for (int j = 0; j < V; ++j) { a0[k][j] = 0.f; a1[k][j] = 0.f; }
    }
    if constexpr (NK > 0) {
        bool in[NK]; int cc[NK];
        PVec<W, V> pg[NK], pb[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int c = (sub + k * LPR) * V;
            in[k] = c < cols; cc[k] = in[k] ? c : 0;
            pg[k].load(gamma, cc[k]);
            if (FUSE == 2) pb[k].load(beta, cc[k]);
        }
#pragma unroll
        for (int t = 0; t < RW; ++t) {
            const long row = ((long)blockIdx.x * RW + t) * RPB + wave;
            const bool live = row < rows;
            const long r = live ? row : 0;
            const T* xr = x + r * (long)cols; const T* gr = gy + r * (long)cols;
            const long mb = (r / rows_per_mod) * (long)cols;
            Vec16<T> rx[NK], rg[NK], ra[NK]; PVec<M, V> ps[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) { rx[k].load(xr + cc[k]); rg[k].load(gr + cc[k]); }
            const float mu = mean[r], rs = rstd[r];
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                ps[k].load(scale, mb + cc[k]);
                ra[k].raw = *(ha ? reinterpret_cast<const uint4*>(gadd + r * (long)cols + cc[k]) : &g_param_pad[0]);
            }
            __builtin_amdgcn_sched_barrier(0);   // (see lnmod_fwd_kernel)
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                float fx[V], fg[V], g[V], sc[V]; rx[k].unpack(fx); rg[k].unpack(fg); pg[k].unpack(g); ps[k].unpack(sc);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float d = fg[j] * (hs ? 1.f + sc[j] : 1.f) * (hg ? g[j] : 1.f);
                    s1 += in[k] ? d : 0.f; s2 += in[k] ? d * (fx[j] - mu) * rs : 0.f;
                }
            }
            // gadd is `const __restrict__`: its loads may move across anything, and LLVM sinks them behind the reductions (a second dependent round trip).
            // A register use here pins them ahead; x / gy, issued before them, were waited for above, so this adds no wait of its own.
#pragma unroll
            for (int k = 0; k < NK; ++k) asm volatile("" : "+v"(ra[k].raw.x), "+v"(ra[k].raw.y), "+v"(ra[k].raw.z), "+v"(ra[k].raw.w));
            s1 = group_sum<LPR>(s1) / (float)cols;
            s2 = group_sum<LPR>(s2) / (float)cols;
            if (!live) continue;                       // (wave-uniform: a wave owns whole rows)
            T* o = gx + row * (long)cols;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                float fx[V], fg[V], g[V], sc[V], bt[V], fa[V];
                rx[k].unpack(fx); rg[k].unpack(fg); pg[k].unpack(g); ps[k].unpack(sc); ra[k].unpack(fa);
                if (FUSE == 2) pb[k].unpack(bt);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float scv = hs ? 1.f + sc[j] : 1.f;
                    const float gm = hg ? g[j] : 1.f;
                    const float xhat = (fx[j] - mu) * rs;
                    if (FUSE == 1) { const float dn = fg[j] * scv; a0[FUSE ? k : 0][j] += in[k] ? dn * xhat : 0.f; a1[FUSE ? k : 0][j] += in[k] ? dn : 0.f; }
                    if (FUSE == 2) { a0[FUSE ? k : 0][j] += in[k] ? fg[j] * (xhat * gm + (hb ? bt[j] : 0.f)) : 0.f; a1[FUSE ? k : 0][j] += in[k] ? fg[j] : 0.f; }
                    const float d = fg[j] * scv * gm;
                    fg[j] = rs * (d - s1 - xhat * s2) + (ha ? fa[j] : 0.f);
                }
                rg[k].pack(fg);
                if (in[k]) rg[k].store(o + cc[k]);
            }
        }
    } else {
        const long row = (long)blockIdx.x * RPB + wave;
        const bool live = row < rows;
        const long r = live ? row : 0;
        const T* xr = x + r * (long)cols; const T* gr = gy + r * (long)cols;
        const float mu = mean[r], rs = rstd[r];
        const long mb = (r / rows_per_mod) * (long)cols;
        float s1 = 0.f, s2 = 0.f;
        for (int c = sub * V; c < cols; c += LPR * V) {
            Vec16<T> vx, vg; PVec<W, V> pg; PVec<M, V> ps;
            vx.load(xr + c); vg.load(gr + c); pg.load(gamma, c); ps.load(scale, mb + c);
            float fx[V], fg[V], g[V], sc[V]; vx.unpack(fx); vg.unpack(fg); pg.unpack(g); ps.unpack(sc);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float d = fg[j] * (hs ? 1.f + sc[j] : 1.f) * (hg ? g[j] : 1.f);
                s1 += d; s2 += d * (fx[j] - mu) * rs;
            }
        }
        s1 = group_sum<LPR>(s1) / (float)cols;
        s2 = group_sum<LPR>(s2) / (float)cols;
        if (!live) return;
        T* o = gx + row * (long)cols;
        for (int c = sub * V; c < cols; c += LPR * V) {
            Vec16<T> vx, vg, va; PVec<W, V> pg; PVec<M, V> ps;
            vx.load(xr + c); vg.load(gr + c); pg.load(gamma, c); ps.load(scale, mb + c);
            va.raw = *(ha ? reinterpret_cast<const uint4*>(gadd + row * (long)cols + c) : &g_param_pad[0]);
            float fx[V], fg[V], g[V], sc[V], fa[V]; vx.unpack(fx); vg.unpack(fg); pg.unpack(g); ps.unpack(sc); va.unpack(fa);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float xhat = (fx[j] - mu) * rs;
                const float d = fg[j] * (hs ? 1.f + sc[j] : 1.f) * (hg ? g[j] : 1.f);
                fg[j] = rs * (d - s1 - xhat * s2) + (ha ? fa[j] : 0.f);
            }
            vg.pack(fg); vg.store(o + c);
        }
    }
    if (FUSE) {
        // block partial: the RPB waves' accumulators, one LPR * V column chunk at a time through LDS -> workspace row of this block's slab
        __shared__ float red[2][RPB][LPR * V];
        const long row0 = (long)blockIdx.x * RW * RPB;
        const long grp = row0 / rows_per_mod;
        const long slab = (row0 - grp * rows_per_mod) / (RW * RPB);
        float* o0 = p0 + (grp * slabs + slab) * (long)cols;
        float* o1 = p1 + (grp * slabs + slab) * (long)cols;
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            if (k) __syncthreads();
#pragma unroll
            for (int j = 0; j < V; ++j) { red[0][wave][sub * V + j] = a0[k][j]; red[1][wave][sub * V + j] = a1[k][j]; }
            __syncthreads();
            for (int cc2 = threadIdx.x; cc2 < LPR * V; cc2 += NB) {
                const int c = k * LPR * V + cc2;
                if (c < cols) {
                    float u0 = 0.f, u1 = 0.f;
#pragma unroll
                    for (int w = 0; w < RPB; ++w) { u0 += red[0][w][cc2]; u1 += red[1][w][cc2]; }
                    o0[c] = u0; o1[c] = u1;
                }
            }
        }
    }
}

// --------------------------------------------------------------------- RoPE
// x: [B, S, H, D]; cos/sin: [S, D/2] fp32.  interleaved=1: pairs (2i, 2i+1) (Wan view_as_complex);
// interleaved=0: pairs (i, i + D/2) (rotate_half).  conj=1 applies the inverse rotation (backward).
template <typename T>
__global__ void __launch_bounds__(NB) rope_kernel(const T* __restrict__ x, const float* __restrict__ cs, const float* __restrict__ sn,
                                                  T* __restrict__ y, long B, long S, long H, int D, int interleaved, int conj) {
    constexpr int V = Elem<T>::VEC;
    const int half = D / 2;
    if (interleaved) {
        const int dv = D / V;
        const long total = B * S * H * dv;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const int c = (int)(i % dv) * V;
            const long row = i / dv;            // (b*S + s)*H + h
            const long s = (row / H) % S;
            Vec16<T> v; v.load(x + row * D + c);
            float f[V]; v.unpack(f);
#pragma unroll
            for (int j = 0; j < V; j += 2) {
                const float co = cs[s * half + (c + j) / 2];
                float si = sn[s * half + (c + j) / 2]; if (conj) si = -si;
                const float a = f[j], b = f[j + 1];
                f[j] = a * co - b * si; f[j + 1] = a * si + b * co;
            }
            v.pack(f); v.store(y + row * D + c);
        }
    } else {
        const int hv = half / V;
        const long total = B * S * H * hv;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const int c = (int)(i % hv) * V;
            const long row = i / hv;
            const long s = (row / H) % S;
            Vec16<T> va, vb; va.load(x + row * D + c); vb.load(x + row * D + half + c);
            float fa[V], fb[V]; va.unpack(fa); vb.unpack(fb);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float co = cs[s * half + c + j];
                float si = sn[s * half + c + j]; if (conj) si = -si;
                const float a = fa[j], b = fb[j];
                fa[j] = a * co - b * si; fb[j] = a * si + b * co;
            }
            va.pack(fa); va.store(y + row * D + c);
            vb.pack(fb); vb.store(y + row * D + half + c);
        }
    }
}

// ------------------------------------------------------------- row softmax
// y = softmax(scale * x) over the last dim; cols valid entries per row, ld = row stride.
template <typename T>
__global__ void __launch_bounds__(NB) softmax_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long rows, int cols_all, long ld, float scale, int causal_rows) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (NB / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + row * ld; T* yr = y + row * ld;
    // causal_rows > 0: rows come in matrices of `causal_rows` rows; row r of a matrix attends to columns <= r
    const int cols = causal_rows > 0 ? min(cols_all, (int)(row % causal_rows) + 1) : cols_all;
    for (int c = cols + lane; c < cols_all; c += 64) yr[c] = Elem<T>::from_f(0.f);
    float m = -INFINITY;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, Elem<T>::to_f(xr[c]) * scale);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += __expf(Elem<T>::to_f(xr[c]) * scale - m);
    s = wave_sum(s);
    const float inv = 1.f / s;
    for (int c = lane; c < cols; c += 64) yr[c] = Elem<T>::from_f(__expf(Elem<T>::to_f(xr[c]) * scale - m) * inv);
}
// gx = scale * y * (gy - sum(gy * y))
template <typename T>
__global__ void __launch_bounds__(NB) softmax_bwd_kernel(const T* __restrict__ y, const T* __restrict__ gy, T* __restrict__ gx, long rows, int cols, long ld, float scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (NB / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* yr = y + row * ld; const T* gr = gy + row * ld; T* o = gx + row * ld;
    float d = 0.f;
    for (int c = lane; c < cols; c += 64) d += Elem<T>::to_f(yr[c]) * Elem<T>::to_f(gr[c]);
    d = wave_sum(d);
    for (int c = lane; c < cols; c += 64) o[c] = Elem<T>::from_f(scale * Elem<T>::to_f(yr[c]) * (Elem<T>::to_f(gr[c]) - d));
}

// --------------------------------------------------------- batched transpose
// in: [batch][R][C] (row stride ldi) -> out: [batch][C][R] (row stride ldo); 64x64 tiles through LDS.
template <typename T>
__global__ void __launch_bounds__(NB) transpose_kernel(const T* __restrict__ in, T* __restrict__ out, int R, int C, long ldi, long ldo,
                                                       long bsi, long bso) {
    __shared__ T tile[64][65];
    const T* ib = in + blockIdx.z * bsi; T* ob = out + blockIdx.z * bso;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 4 rows per pass
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        if (r < R && c < C) tile[i][tx] = ib[(long)r * ldi + c];
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (r < R && c < C) ob[(long)c * ldo + r] = tile[tx][i];
    }
}

template <typename T> struct Tag { using type = T; };

}  // namespace

#define STREAM(s) reinterpret_cast<hipStream_t>(s)
#define BAD(msg) do { set_last_error(msg); return DPIPE_ERR_ARG; } while (0)

static inline int pick_lpr(int cols, int vec) { return (cols / vec) <= 16 ? 16 : 64; }
// A/B switch (environment DPIPE_LNMOD_FUSE=0): LayerNorm backward with the separate column-reduction launch instead of the fused partials
static inline bool lnmod_fuse_enabled() { static const int on = [] { const char* e = getenv("DPIPE_LNMOD_FUSE"); return e ? atoi(e) : 1; }(); return on != 0; }

// dispatch helper over (T, W) where W in {T, float}
#define DISPATCH_TW(dtype, wdtype, ...)                                         \
    if (dtype == DPIPE_BF16 && wdtype == DPIPE_BF16) { using T = bf16_t; using W = bf16_t; __VA_ARGS__; } \
    else if (dtype == DPIPE_BF16 && wdtype == DPIPE_F32) { using T = bf16_t; using W = float; __VA_ARGS__; } \
    else if (dtype == DPIPE_F32 && wdtype == DPIPE_F32) { using T = float; using W = float; __VA_ARGS__; } \
    else { set_last_error("unsupported dtype combination"); return DPIPE_ERR_UNSUPPORTED; }

extern "C" {

int dpipe_norm_slabs(long rows_per_group) {
    long s = rows_per_group / 128; if (s < 1) s = 1; if (s > 64) s = 64; return (int)s;   // few slabs: the second stage walks them serially
}

int dpipe_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, long rows, int cols, float eps, int dtype, int wdtype, void* stream) {
    const int V = dtype == DPIPE_BF16 ? 8 : 4;
    if (!x || !y || rows <= 0 || cols <= 0 || (cols % V) != 0 || !aligned16(w)) BAD("dpipe_rmsnorm_fwd: cols must be a multiple of the 16-byte vector, the weight 16-byte aligned");
    hipStream_t s = STREAM(stream);
    const int lpr = pick_lpr(cols, V);
    const unsigned grid = (unsigned)cdiv(rows, NB / lpr);
    DISPATCH_TW(dtype, wdtype, {
        if (lpr == 16) rmsnorm_fwd_kernel<T, W, 16><<<grid, NB, 0, s>>>((const T*)x, (const W*)w, (T*)y, rstd, rows, cols, eps);
        else rmsnorm_fwd_kernel<T, W, 64><<<grid, NB, 0, s>>>((const T*)x, (const W*)w, (T*)y, rstd, rows, cols, eps);
    })
    return check_launch("dpipe_rmsnorm_fwd");
}

// workspace: slabs * cols floats (only when dw != null)
int dpipe_rmsnorm_bwd(const void* x, const void* w, const void* gy, const float* rstd, void* gx, void* dw, float* workspace,
                      long rows, int cols, int dtype, int wdtype, int accumulate_params, void* stream) {
    const int V = dtype == DPIPE_BF16 ? 8 : 4;
    if (!x || !gy || !rstd || !gx || rows <= 0 || cols <= 0 || (cols % V) != 0 || !aligned16(w)) BAD("dpipe_rmsnorm_bwd: bad argument");
    if (dw && !workspace) BAD("dpipe_rmsnorm_bwd: workspace required for dw");
    hipStream_t s = STREAM(stream);
    const int lpr = pick_lpr(cols, V);
    const unsigned grid = (unsigned)cdiv(rows, NB / lpr);
    const int slabs = dpipe_norm_slabs(rows);
    DISPATCH_TW(dtype, wdtype, {
        if (lpr == 16) rmsnorm_bwd_dx_kernel<T, W, 16><<<grid, NB, 0, s>>>((const T*)x, (const W*)w, (const T*)gy, rstd, (T*)gx, rows, cols);
        else rmsnorm_bwd_dx_kernel<T, W, 64><<<grid, NB, 0, s>>>((const T*)x, (const W*)w, (const T*)gy, rstd, (T*)gx, rows, cols);
        if (dw) {
            dim3 g2((unsigned)cdiv(cols / V, CR_CT), slabs, 1);
            colreduce_kernel<T, W, float, 0><<<g2, NB, 0, s>>>((const T*)x, (const T*)gy, nullptr, rstd, nullptr, nullptr, nullptr, rows, cols, cols, slabs, workspace, nullptr);
            slabsum_kernel<W><<<(unsigned)cdiv(cols, NB), NB, 0, s>>>(workspace, (W*)dw, 1, cols, slabs, accumulate_params);
        }
    })
    return check_launch("dpipe_rmsnorm_bwd");
}

static bool rope_geom(RopeGeom& rg, const float* cos_t, const float* sin_t, long rows, int cols, int head_dim, long S, int groups, long token_offset, long rope_tokens,
                      long x_token_stride) {
    if (!cos_t || !sin_t || head_dim <= 0 || (head_dim % 2) || S <= 0 || groups <= 0 || rows % groups || (rows / groups) % S) return false;
    if (groups == 1 ? (cols % head_dim) != 0 : cols != head_dim) return false;
    if (x_token_stride < (long)groups * cols || token_offset < 0) return false;
    rg.cs = cos_t; rg.sn = sin_t; rg.S = S; rg.G = groups; rg.D = head_dim; rg.tok_off = token_offset; rg.rope_tokens = rope_tokens < 0 ? S : rope_tokens; rg.x_ts = x_token_stride;
    return true;
}

int dpipe_rmsnorm_rope_fwd(const void* x, const void* w, const float* cos_t, const float* sin_t, void* y, float* rstd, long rows, int cols, int head_dim, long S,
                           int groups_per_token, long token_offset, long rope_tokens, long x_token_stride, float eps, int dtype, int wdtype, void* stream) {
    const int V = dtype == DPIPE_BF16 ? 8 : 4;
    RopeGeom rg;
    if (!x || !y || rows <= 0 || cols <= 0 || (cols % V) != 0 || (head_dim % V) != 0 || (x_token_stride % V) != 0 || !aligned16(w) ||
        !rope_geom(rg, cos_t, sin_t, rows, cols, head_dim, S, groups_per_token, token_offset, rope_tokens, x_token_stride)) BAD("dpipe_rmsnorm_rope_fwd: bad argument");
    hipStream_t s = STREAM(stream);
    const int lpr = pick_lpr(cols, V);
    const unsigned grid = (unsigned)cdiv(rows, NB / lpr);
    DISPATCH_TW(dtype, wdtype, {
        if (lpr == 16) rmsnorm_rope_fwd_kernel<T, W, 16><<<grid, NB, 0, s>>>((const T*)x, (const W*)w, (T*)y, rstd, rows, cols, eps, rg);
        else rmsnorm_rope_fwd_kernel<T, W, 64><<<grid, NB, 0, s>>>((const T*)x, (const W*)w, (T*)y, rstd, rows, cols, eps, rg);
    })
    return check_launch("dpipe_rmsnorm_rope_fwd");
}

// workspace: dpipe_norm_slabs(rows) * cols floats (only when dw != null)
int dpipe_rmsnorm_rope_bwd(const void* x, const void* w, const void* gy, const float* rstd, const float* cos_t, const float* sin_t, void* gx, void* dw, float* workspace,
                           long rows, int cols, int head_dim, long S, int groups_per_token, long token_offset, long rope_tokens, long x_token_stride, int dtype,
                           int wdtype, int accumulate_params, void* stream) {
    const int V = dtype == DPIPE_BF16 ? 8 : 4;
    RopeGeom rg;
    if (!x || !gy || !rstd || !gx || rows <= 0 || cols <= 0 || (cols % V) != 0 || (head_dim % V) != 0 || (x_token_stride % V) != 0 || !aligned16(w) ||
        !rope_geom(rg, cos_t, sin_t, rows, cols, head_dim, S, groups_per_token, token_offset, rope_tokens, x_token_stride)) BAD("dpipe_rmsnorm_rope_bwd: bad argument");
    if (dw && !workspace) BAD("dpipe_rmsnorm_rope_bwd: workspace required for dw");
    hipStream_t s = STREAM(stream);
    const int lpr = pick_lpr(cols, V);
    const unsigned grid = (unsigned)cdiv(rows, NB / lpr);
    const int slabs = dpipe_norm_slabs(rows);
    DISPATCH_TW(dtype, wdtype, {
        if (lpr == 16) rmsnorm_rope_bwd_dx_kernel<T, W, 16><<<grid, NB, 0, s>>>((const T*)x, (const W*)w, (const T*)gy, rstd, (T*)gx, rows, cols, rg);
        else rmsnorm_rope_bwd_dx_kernel<T, W, 64><<<grid, NB, 0, s>>>((const T*)x, (const W*)w, (const T*)gy, rstd, (T*)gx, rows, cols, rg);
        if (dw) {
            dim3 g2((unsigned)cdiv(cols / V, 32), slabs, 1);
            rmsnorm_rope_dw_kernel<T><<<g2, NB, 0, s>>>((const T*)x, (const T*)gy, rstd, rows, cols, slabs, workspace, rg);
            slabsum_kernel<W><<<(unsigned)cdiv(cols, NB), NB, 0, s>>>(workspace, (W*)dw, 1, cols, slabs, accumulate_params);
        }
    })
    return check_launch("dpipe_rmsnorm_rope_bwd");
}

int dpipe_lnmod_fwd(const void* x, const void* gamma, const void* beta, const void* scale, const void* shift, void* y,
                    float* mean, float* rstd, long rows, int cols, long rows_per_mod, float eps,
                    int dtype, int wdtype, int mdtype, void* stream) {
    const int V = dtype == DPIPE_BF16 ? 8 : 4;
    if (!x || !y || rows <= 0 || cols <= 0 || (cols % V) != 0 || rows_per_mod <= 0) BAD("dpipe_lnmod_fwd: bad argument");
    if (ablated(ABL_LN)) return DPIPE_OK;
    hipStream_t s = STREAM(stream);
    if (!aligned16(x) || !aligned16(y) || !aligned16(gamma) || !aligned16(beta) || !aligned16(scale) || !aligned16(shift)) BAD("dpipe_lnmod_fwd: operands must be 16-byte aligned");
    const unsigned grid = (unsigned)cdiv(rows, NB / 64);
    const int nk = cols <= 64 * V * 4 ? (int)cdiv(cols, 64 * V) : 0;      // vectors per lane when the row stays in registers (see lnmod_fwd_kernel)
#define LNFWD_K(TT, WW, MM, KK) lnmod_fwd_kernel<TT, WW, MM, 64, KK><<<grid, NB, 0, s>>>((const TT*)x, (const WW*)gamma, (const WW*)beta, (const MM*)scale, (const MM*)shift, (TT*)y, mean, rstd, rows, cols, rows_per_mod, eps)
#define LNFWD(TT, WW, MM) do { switch (nk) { case 1: LNFWD_K(TT, WW, MM, 1); break; case 2: LNFWD_K(TT, WW, MM, 2); break; case 3: LNFWD_K(TT, WW, MM, 3); break; \
    case 4: LNFWD_K(TT, WW, MM, 4); break; default: LNFWD_K(TT, WW, MM, 0); } } while (0)
    if (dtype == DPIPE_BF16) {
        if (wdtype == DPIPE_BF16 && mdtype == DPIPE_BF16) LNFWD(bf16_t, bf16_t, bf16_t);
        else if (wdtype == DPIPE_BF16) LNFWD(bf16_t, bf16_t, float);
        else if (mdtype == DPIPE_BF16) LNFWD(bf16_t, float, bf16_t);
        else LNFWD(bf16_t, float, float);
    } else if (dtype == DPIPE_F32 && wdtype == DPIPE_F32 && mdtype == DPIPE_F32) LNFWD(float, float, float);
    else { set_last_error("dpipe_lnmod_fwd: dtype combination"); return DPIPE_ERR_UNSUPPORTED; }
#undef LNFWD
#undef LNFWD_K
    return check_launch("dpipe_lnmod_fwd");
}

// workspace: 2 * max(slabs(rows) , groups*slabs(rows_per_mod)) * cols floats
int dpipe_lnmod_bwd(const void* x, const void* gy, const void* gamma, const void* beta, const void* scale,
                    const float* mean, const float* rstd, void* gx, void* dgamma, void* dbeta, void* dscale, void* dshift,
                    float* workspace, long rows, int cols, long rows_per_mod, int dtype, int wdtype, int mdtype, int accumulate_params,
                    const void* gx_add, void* stream) {
    const int V = dtype == DPIPE_BF16 ? 8 : 4;
    if (!x || !gy || !mean || !rstd || !gx || rows <= 0 || cols <= 0 || (cols % V) != 0 || rows_per_mod <= 0 || (rows % rows_per_mod) != 0)
        BAD("dpipe_lnmod_bwd: bad argument");
    if (ablated(ABL_LN)) return DPIPE_OK;
    if ((dgamma || dscale) && !workspace) BAD("dpipe_lnmod_bwd: workspace required");
    if (!aligned16(x) || !aligned16(gy) || !aligned16(gx) || !aligned16(gx_add) || !aligned16(gamma) || !aligned16(beta) || !aligned16(scale))
        BAD("dpipe_lnmod_bwd: operands must be 16-byte aligned");
    const int nk = cols <= 64 * V * 4 ? (int)cdiv(cols, 64 * V) : 0;
    hipStream_t s = STREAM(stream);
    const unsigned grid = (unsigned)cdiv(rows, NB / 64);
    const long groups = rows / rows_per_mod;
    const int slabs_all = dpipe_norm_slabs(rows), slabs_mod = dpipe_norm_slabs(rows_per_mod);
    // fused parameter-gradient partials (see lnmod_bwd_dx_kernel): rows that fit the register cache, exactly one pair of column sums wanted, and modulation
    // groups made of whole blocks.
    const int vec_cols = 64 * V * 4;
    // rows per wave: 1 (4 rows per block) up to 2 048 rows -- the kernel is latency-bound and wants the blocks; 4 beyond (measured on MI355X, tools/norm_timing.py:
    // [1024, 1280] backward 35.8 us at 4 rows per wave vs 24.4 us unfused; [4096, 640] 33.3 vs 34.3).  DPIPE_LNMOD_RW = 1 / 4 forces one for A/B.
    static const int rw_env = [] { const char* e = getenv("DPIPE_LNMOD_RW"); return e ? atoi(e) : 0; }();
    // (round 6, after the operand-load rewrite: 4 rows per wave from 1 024 rows on measured level with this policy -- 345.6 / 347.0 vs 346.8 / 346.2 ms per step, call r6z;
    //  344.9 vs 345.7 / 346.0 in call r6t, profiles/r6t_bench_lnmod_rows_per_wave.jsonl -- the threshold stays)
    const int rw = rw_env == 1 || rw_env == 4 ? rw_env : (rows_per_mod > 2048 ? 4 : 1);
    const bool fuse_ok = cols <= vec_cols && ((dgamma != nullptr) != (dscale != nullptr)) && (groups == 1 || rows_per_mod % (4 * rw) == 0) && lnmod_fuse_enabled();
    const int fslabs = fuse_ok ? (int)cdiv(rows_per_mod, 4 * rw) : 0;
#define LNBWD_FUSED_K(TT, WW, MM, FU, RWW, KK) \
    lnmod_bwd_dx_kernel<TT, WW, MM, 64, KK, FU, RWW><<<(unsigned)cdiv(rows, 4 * RWW), NB, 0, s>>>((const TT*)x, (const TT*)gy, (const WW*)gamma, (const MM*)scale, mean, rstd, (TT*)gx, \
        (const TT*)gx_add, rows, cols, rows_per_mod, (const WW*)beta, p0, p1, fslabs)
#define LNBWD_FUSED(TT, WW, MM, FU, RWW) do { \
    float* p0 = workspace; float* p1 = workspace + groups * fslabs * (long)cols; \
    switch (nk) { case 1: LNBWD_FUSED_K(TT, WW, MM, FU, RWW, 1); break; case 2: LNBWD_FUSED_K(TT, WW, MM, FU, RWW, 2); break; \
                  case 3: LNBWD_FUSED_K(TT, WW, MM, FU, RWW, 3); break; default: LNBWD_FUSED_K(TT, WW, MM, FU, RWW, 4); } \
    if (FU == 1) slabsum2p_kernel<WW><<<dim3((unsigned)cdiv(cols, 32), 1), NB, 0, s>>>(p0, p1, (WW*)dgamma, (WW*)dbeta, cols, (int)(groups * fslabs), accumulate_params); \
    else slabsum2p_kernel<MM><<<dim3((unsigned)cdiv(cols, 32), (unsigned)groups), NB, 0, s>>>(p0, p1, (MM*)dscale, (MM*)dshift, cols, fslabs, 0); \
    } while (0)
#define LNBWD_DX_K(TT, WW, MM, KK) lnmod_bwd_dx_kernel<TT, WW, MM, 64, KK><<<grid, NB, 0, s>>>((const TT*)x, (const TT*)gy, (const WW*)gamma, (const MM*)scale, mean, rstd, (TT*)gx, (const TT*)gx_add, rows, cols, rows_per_mod)
#define LNBWD(TT, WW, MM) do { \
    if (fuse_ok && dgamma) { if (rw == 4) LNBWD_FUSED(TT, WW, MM, 1, 4); else LNBWD_FUSED(TT, WW, MM, 1, 1); break; } \
    if (fuse_ok && dscale) { if (rw == 4) LNBWD_FUSED(TT, WW, MM, 2, 4); else LNBWD_FUSED(TT, WW, MM, 2, 1); break; } \
    switch (nk) { case 1: LNBWD_DX_K(TT, WW, MM, 1); break; case 2: LNBWD_DX_K(TT, WW, MM, 2); break; case 3: LNBWD_DX_K(TT, WW, MM, 3); break; \
                  case 4: LNBWD_DX_K(TT, WW, MM, 4); break; default: LNBWD_DX_K(TT, WW, MM, 0); } \
    if (dscale) { \
        float* p0 = workspace; float* p1 = workspace + groups * slabs_mod * (long)cols; \
        dim3 g2((unsigned)cdiv(cols / V, CR_CT), slabs_mod, (unsigned)groups); \
        colreduce_kernel<TT, WW, MM, 2><<<g2, NB, 0, s>>>((const TT*)x, (const TT*)gy, mean, rstd, (const WW*)gamma, (const WW*)beta, (const MM*)scale, rows_per_mod, cols, cols, slabs_mod, p0, p1); \
        slabsum2_kernel<MM><<<(unsigned)cdiv(groups * cols, NB), NB, 0, s>>>(p0, p1, (MM*)dscale, (MM*)dshift, groups, cols, slabs_mod, 0); \
    } \
    if (dgamma) { \
        /* (dgamma, dbeta) need dn = gy*(1+scale[b]) which varies per modulation group: reduce per group, then over groups */ \
        float* p0 = workspace; float* p1 = workspace + groups * slabs_mod * (long)cols; \
        dim3 g2((unsigned)cdiv(cols / V, CR_CT), slabs_mod, (unsigned)groups); \
        colreduce_kernel<TT, WW, MM, 1><<<g2, NB, 0, s>>>((const TT*)x, (const TT*)gy, mean, rstd, (const WW*)gamma, (const WW*)beta, (const MM*)scale, rows_per_mod, cols, cols, slabs_mod, p0, p1); \
        slabsum2_kernel<WW><<<(unsigned)cdiv(cols, NB), NB, 0, s>>>(p0, p1, (WW*)dgamma, (WW*)dbeta, 1, cols, (int)(groups * slabs_mod), accumulate_params); \
    } } while (0)
    (void)slabs_all;
    if (dtype == DPIPE_BF16) {
        if (wdtype == DPIPE_BF16 && mdtype == DPIPE_BF16) LNBWD(bf16_t, bf16_t, bf16_t);
        else if (wdtype == DPIPE_BF16) LNBWD(bf16_t, bf16_t, float);
        else if (mdtype == DPIPE_BF16) LNBWD(bf16_t, float, bf16_t);
        else LNBWD(bf16_t, float, float);
    } else if (dtype == DPIPE_F32 && wdtype == DPIPE_F32 && mdtype == DPIPE_F32) LNBWD(float, float, float);
    else { set_last_error("dpipe_lnmod_bwd: dtype combination"); return DPIPE_ERR_UNSUPPORTED; }
#undef LNBWD
#undef LNBWD_FUSED
#undef LNBWD_FUSED_K
#undef LNBWD_DX_K
    return check_launch("dpipe_lnmod_bwd");
}

int dpipe_lnmod_workspace_floats(long rows, int cols, long rows_per_mod) {
    long groups = rows / rows_per_mod;
    long slabs = dpipe_norm_slabs(rows_per_mod);
    const long fused = (rows_per_mod + 3) / 4;          // the fused backward writes one partial row per 4 (or 16) rows of a group (ragged tail block included)
    if (cols <= 2048 && fused > slabs) slabs = fused;     // (the fused path exists for rows that fit the register cache only)
    return (int)(2 * groups * slabs * (long)cols);
}

// out[c] (+)= sum_r x[r, c]: bias gradients of nn.Linear (db = column sums of dy).  workspace: dpipe_norm_slabs(rows) * cols floats.
int dpipe_colsum(const void* x, long rows, int cols, long ld, void* out, float* workspace, int dtype, int out_dtype, int accumulate,
                 void* stream) {
    const int V = dtype == DPIPE_BF16 ? 8 : 4;
    if (!x || !out || !workspace || rows <= 0 || cols <= 0 || (cols % V) != 0 || (ld % V) != 0) BAD("dpipe_colsum: bad argument");
    hipStream_t s = STREAM(stream);
    const int slabs = dpipe_norm_slabs(rows);
    DISPATCH_TW(dtype, out_dtype, {
        dim3 g2((unsigned)cdiv(cols / V, CR_CT), slabs, 1);
        colreduce_kernel<T, W, float, 3><<<g2, NB, 0, s>>>((const T*)x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, rows, cols, ld, slabs, workspace, nullptr);
        slabsum_kernel<W><<<(unsigned)cdiv(cols, NB), NB, 0, s>>>(workspace, (W*)out, 1, cols, slabs, accumulate);
    });
    return check_launch("dpipe_colsum");
}

int dpipe_rope(const void* x, const float* cos_t, const float* sin_t, void* y, long B, long S, long H, int D,
               int interleaved, int conj, int dtype, void* stream) {
    const int V = dtype == DPIPE_BF16 ? 8 : 4;
    if (!x || !cos_t || !sin_t || !y || B <= 0 || S <= 0 || H <= 0 || D <= 0) BAD("dpipe_rope: bad argument");
    if (interleaved ? (D % V) != 0 : ((D / 2) % V) != 0) BAD("dpipe_rope: head dim not vectorisable");
    const long work = B * S * H * (interleaved ? D / V : (D / 2) / V);
    hipStream_t s = STREAM(stream);
    if (dtype == DPIPE_BF16) rope_kernel<bf16_t><<<stream_grid(work, NB), NB, 0, s>>>((const bf16_t*)x, cos_t, sin_t, (bf16_t*)y, B, S, H, D, interleaved, conj);
    else if (dtype == DPIPE_F32) rope_kernel<float><<<stream_grid(work, NB), NB, 0, s>>>((const float*)x, cos_t, sin_t, (float*)y, B, S, H, D, interleaved, conj);
    else { set_last_error("dpipe_rope: dtype"); return DPIPE_ERR_UNSUPPORTED; }
    return check_launch("dpipe_rope");
}

int dpipe_softmax_fwd(const void* x, void* y, long rows, int cols, long ld, float scale, int causal_rows, int dtype, void* stream) {
    if (!x || !y || rows <= 0 || cols <= 0 || ld < cols) BAD("dpipe_softmax_fwd: bad argument");
    hipStream_t s = STREAM(stream);
    const unsigned grid = (unsigned)cdiv(rows, NB / 64);
    if (dtype == DPIPE_BF16) softmax_fwd_kernel<bf16_t><<<grid, NB, 0, s>>>((const bf16_t*)x, (bf16_t*)y, rows, cols, ld, scale, causal_rows);
    else if (dtype == DPIPE_F32) softmax_fwd_kernel<float><<<grid, NB, 0, s>>>((const float*)x, (float*)y, rows, cols, ld, scale, causal_rows);
    else { set_last_error("dpipe_softmax_fwd: dtype"); return DPIPE_ERR_UNSUPPORTED; }
    return check_launch("dpipe_softmax_fwd");
}

int dpipe_softmax_bwd(const void* y, const void* gy, void* gx, long rows, int cols, long ld, float scale, int dtype, void* stream) {
    if (!y || !gy || !gx || rows <= 0 || cols <= 0 || ld < cols) BAD("dpipe_softmax_bwd: bad argument");
    hipStream_t s = STREAM(stream);
    const unsigned grid = (unsigned)cdiv(rows, NB / 64);
    if (dtype == DPIPE_BF16) softmax_bwd_kernel<bf16_t><<<grid, NB, 0, s>>>((const bf16_t*)y, (const bf16_t*)gy, (bf16_t*)gx, rows, cols, ld, scale);
    else if (dtype == DPIPE_F32) softmax_bwd_kernel<float><<<grid, NB, 0, s>>>((const float*)y, (const float*)gy, (float*)gx, rows, cols, ld, scale);
    else { set_last_error("dpipe_softmax_bwd: dtype"); return DPIPE_ERR_UNSUPPORTED; }
    return check_launch("dpipe_softmax_bwd");
}

int dpipe_transpose(const void* in, void* out, int R, int C, long ldi, long ldo, long batch_stride_in, long batch_stride_out, int batch,
                    int dtype, void* stream) {
    if (!in || !out || R <= 0 || C <= 0 || batch <= 0 || ldi < C || ldo < R) BAD("dpipe_transpose: bad argument");
    dim3 grid((unsigned)cdiv(C, 64), (unsigned)cdiv(R, 64), (unsigned)batch);
    hipStream_t s = STREAM(stream);
    if (dtype == DPIPE_BF16) transpose_kernel<bf16_t><<<grid, NB, 0, s>>>((const bf16_t*)in, (bf16_t*)out, R, C, ldi, ldo, batch_stride_in, batch_stride_out);
    else if (dtype == DPIPE_F32) transpose_kernel<float><<<grid, NB, 0, s>>>((const float*)in, (float*)out, R, C, ldi, ldo, batch_stride_in, batch_stride_out);
    else { set_last_error("dpipe_transpose: dtype"); return DPIPE_ERR_UNSUPPORTED; }
    return check_launch("dpipe_transpose");
}

}  // extern "C"
