// elementwise.hip -- HBM-bound kernels of the training hot path (gfx950).
//
//   K9  loss            : fused (out-target)^2 * mask -> per-row mean -> weighted mean, fwd + bwd
//                         (reference: models/base.py:418-436, models/sdxl.py:632-651)
//   K5  gated residual  : out = x + y * gate[b, :]            (models/wan/model.py:301,308)
//   K6  activations     : gelu(tanh|erf), silu, geglu fwd + bwd (models/wan/model.py:270-272)
//   K7  timestep embed  : sinusoidal_embedding_1d             (models/wan/model.py:15-25)
//   K8  flow-match prep : x_t = (1-t) x1 + t x0, target = x0 - x1 (models/flux.py:368-372)
//   K10 grad-norm/clip  : multi-tensor sum of squares + in-place scale (utils/patches.py:175-246)
//
// All kernels: 16-byte vector accesses, grid-stride over <=2048 blocks, fp32 math,
// deterministic two-stage reductions (no float atomics).
#include "dpipe_common.h"
#include "../../include/dpipe_hip.h"

using namespace dpipe;

namespace {

constexpr int EW_BLOCK = 256;

// ------------------------------------------------------------------ loss (K9)
enum { LOSS_MSE = 0, LOSS_HUBER = 1, LOSS_SMOOTH_L1 = 2 };

__device__ __forceinline__ float loss_elem(float d, int kind, float p) {
    if (kind == LOSS_MSE) return d * d;
    float a = fabsf(d);
    if (kind == LOSS_HUBER) return a <= p ? 0.5f * d * d : p * (a - 0.5f * p);
    // smooth_l1(beta=p); beta == 0 degenerates to L1 like torch
    if (p == 0.f) return a;
    return a < p ? 0.5f * d * d / p : a - 0.5f * p;
}
__device__ __forceinline__ float loss_elem_grad(float d, int kind, float p) {
    if (kind == LOSS_MSE) return 2.f * d;
    float a = fabsf(d);
    float s = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    if (kind == LOSS_HUBER) return a <= p ? d : p * s;
    if (p == 0.f) return s;
    return a < p ? d / p : s;
}

// grid = (chunks, rows). partials[row * chunks + chunk] = sum over the chunk of elem loss * mask
template <typename T>
__global__ void __launch_bounds__(EW_BLOCK) loss_fwd_partial_kernel(
    const T* __restrict__ out, const float* __restrict__ target, const float* __restrict__ mask,
    long cols, int kind, float param, float* __restrict__ partials) {
    __shared__ float smem[16];
    constexpr int V = Elem<T>::VEC;
    const long row = blockIdx.y;
    const T* o = out + row * cols;
    const float* t = target + row * cols;
    const float* m = mask ? mask + row * cols : nullptr;
    float acc = 0.f;
    const bool vec_ok = (cols % V) == 0;  // rows stay 16-byte aligned
    if (vec_ok) {
        const long nv = cols / V;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
            Vec16<T> vo; vo.load(o + i * V);
            float fo[V]; vo.unpack(fo);
#pragma unroll
            for (int j = 0; j < V; j += 4) {
                float4 ft = *reinterpret_cast<const float4*>(t + i * V + j);
                float tt[4] = {ft.x, ft.y, ft.z, ft.w};
                float mm[4] = {1.f, 1.f, 1.f, 1.f};
                if (m) { float4 fm = *reinterpret_cast<const float4*>(m + i * V + j); mm[0] = fm.x; mm[1] = fm.y; mm[2] = fm.z; mm[3] = fm.w; }
#pragma unroll
                for (int k = 0; k < 4; ++k) acc += loss_elem(fo[j + k] - tt[k], kind, param) * mm[k];
            }
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < cols; i += (long)gridDim.x * blockDim.x) {
            float d = Elem<T>::to_f(o[i]) - t[i];
            acc += loss_elem(d, kind, param) * (m ? m[i] : 1.f);
        }
    }
    float s = block_sum(acc, smem);
    if (threadIdx.x == 0) partials[row * gridDim.x + blockIdx.x] = s;
}

// one block: loss = (1/rows) * sum_r w_r * (1/cols) * sum_c partials[r][c]
__global__ void __launch_bounds__(EW_BLOCK) loss_fwd_final_kernel(
    const float* __restrict__ partials, const float* __restrict__ row_weight, int rows, int chunks,
    long cols, float* __restrict__ loss, float* __restrict__ row_loss) {
    __shared__ float smem[16];
    float total = 0.f;
    for (int r = 0; r < rows; ++r) {
        float a = 0.f;
        for (int c = threadIdx.x; c < chunks; c += blockDim.x) a += partials[(long)r * chunks + c];
        float s = block_sum(a, smem) / (float)cols;
        if (row_loss && threadIdx.x == 0) row_loss[r] = s;
        total += s * (row_weight ? row_weight[r] : 1.f);
    }
    if (threadIdx.x == 0) loss[0] = total / (float)rows;
}

template <typename T>
__global__ void __launch_bounds__(EW_BLOCK) loss_bwd_kernel(
    const T* __restrict__ out, const float* __restrict__ target, const float* __restrict__ mask,
    const float* __restrict__ row_weight, const float* __restrict__ grad_loss, long rows, long cols,
    int kind, float param, T* __restrict__ grad_out) {
    constexpr int V = Elem<T>::VEC;
    const long row = blockIdx.y;
    const float g = grad_loss[0] * (row_weight ? row_weight[row] : 1.f) / ((float)rows * (float)cols);
    const T* o = out + row * cols;
    const float* t = target + row * cols;
    const float* m = mask ? mask + row * cols : nullptr;
    T* go = grad_out + row * cols;
    if ((cols % V) == 0) {
        const long nv = cols / V;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
            Vec16<T> vo; vo.load(o + i * V);
            float fo[V], r[V]; vo.unpack(fo);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float d = fo[j] - t[i * V + j];
                r[j] = g * loss_elem_grad(d, kind, param) * (m ? m[i * V + j] : 1.f);
            }
            Vec16<T> vr; vr.pack(r); vr.store(go + i * V);
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < cols; i += (long)gridDim.x * blockDim.x) {
            float d = Elem<T>::to_f(o[i]) - t[i];
            go[i] = Elem<T>::from_f(g * loss_elem_grad(d, kind, param) * (m ? m[i] : 1.f));
        }
    }
}

// ------------------------------------------------------------- activations (K6)
enum { ACT_NONE = 0, ACT_GELU_TANH = 1, ACT_GELU_ERF = 2, ACT_SILU = 3, ACT_QUICK_GELU = 4 };

__device__ __forceinline__ float act_fwd(float x, int act) {
    switch (act) {
    case ACT_GELU_TANH: {
        const float k0 = 0.7978845608028654f, k1 = 0.044715f;
        float u = k0 * (x + k1 * x * x * x);
        return 0.5f * x * (1.f + tanhf(u));
    }
    case ACT_GELU_ERF: { float E; return x * gelu_erf_cdf(x, E); }
    case ACT_SILU: return x / (1.f + __expf(-x));
    case ACT_QUICK_GELU: return x / (1.f + __expf(-1.702f * x));       // HF CLIP quick_gelu: x * sigmoid(1.702 x)
    default: return x;
    }
}
__device__ __forceinline__ float act_bwd(float x, int act) {  // d act / dx
    switch (act) {
    case ACT_GELU_TANH: {
        const float k0 = 0.7978845608028654f, k1 = 0.044715f;
        float u = k0 * (x + k1 * x * x * x);
        float th = tanhf(u);
        float du = k0 * (1.f + 3.f * k1 * x * x);
        return 0.5f * (1.f + th) + 0.5f * x * (1.f - th * th) * du;
    }
    case ACT_GELU_ERF: {
        float E;
        const float cdf = gelu_erf_cdf(x, E);
        return cdf + x * (0.3989422804014327f * E);
    }
    case ACT_SILU: {
        float s = 1.f / (1.f + __expf(-x));
        return s * (1.f + x * (1.f - s));
    }
    case ACT_QUICK_GELU: {
        float s = 1.f / (1.f + __expf(-1.702f * x));
        return s * (1.f + 1.702f * x * (1.f - s));
    }
    default: return 1.f;
    }
}

// Adjoint of diffusers' nearest 2x Upsample2D on a channels-last tensor: dst[b, y, x, :] = sum of the 2 x 2 block src[b, 2y .. 2y+1, 2x .. 2x+1, :] (fp32 accumulate, one
// rounding).  Replaces `dxu.view(B, H, 2, W, 2, C).sum(dim=(2, 4))` in the convolution dgrad (round 5: the last ATen REDUCTION inside the hipGraph-replayed SDXL step --
// an ATen reduction returned garbage under replay on the stacked path; this one had not misbehaved, it simply no longer has the chance).
template <typename T>
__global__ void __launch_bounds__(EW_BLOCK) upsample2x_adjoint_kernel(const T* __restrict__ src, T* __restrict__ dst, long npix, int H, int W, int C) {
    constexpr int V = Elem<T>::VEC;
    const int cv = C / V;
    const long total = npix * cv;
    const long row = (long)2 * W * C;                 // one source pixel row
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / cv; const int c0 = (int)(i - pix * cv) * V;
        const long b = pix / ((long)H * W); const long r = pix - b * (long)H * W;
        const long y = r / W, x = r - y * W;
        const T* s0 = src + ((b * 2 * H + 2 * y) * 2 * W + 2 * x) * C + c0;
        Vec16<T> a, bb, c, d;
        a.load(s0); bb.load(s0 + C); c.load(s0 + row); d.load(s0 + row + C);
        float fa[V], fb[V], fc[V], fd[V];
        a.unpack(fa); bb.unpack(fb); c.unpack(fc); d.unpack(fd);
#pragma unroll
        for (int j = 0; j < V; ++j) fa[j] = (fa[j] + fb[j]) + (fc[j] + fd[j]);
        a.pack(fa); a.store(dst + pix * C + c0);
    }
}

template <typename T, int ACTC>
__global__ void __launch_bounds__(EW_BLOCK) act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long n, int act) {
    constexpr int V = Elem<T>::VEC;
    const long nv = n / V;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
        Vec16<T> v; v.load(x + i * V);
        float f[V]; v.unpack(f);
#pragma unroll
        for (int j = 0; j < V; ++j) f[j] = act_fwd(f[j], ACTC);
        v.pack(f); v.store(y + i * V);
    }
    for (long i = nv * V + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = Elem<T>::from_f(act_fwd(Elem<T>::to_f(x[i]), ACTC));
}
template <typename T, int ACTC>
__global__ void __launch_bounds__(EW_BLOCK) act_bwd_kernel(const T* __restrict__ x, const T* __restrict__ gy, T* __restrict__ gx, long n, int act) {
    constexpr int V = Elem<T>::VEC;
    const long nv = n / V;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
        Vec16<T> vx, vg; vx.load(x + i * V); vg.load(gy + i * V);
        float fx[V], fg[V]; vx.unpack(fx); vg.unpack(fg);
#pragma unroll
        for (int j = 0; j < V; ++j) fg[j] *= act_bwd(fx[j], ACTC);
        vg.pack(fg); vg.store(gx + i * V);
    }
    for (long i = nv * V + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        gx[i] = Elem<T>::from_f(Elem<T>::to_f(gy[i]) * act_bwd(Elem<T>::to_f(x[i]), ACTC));
}

// GEGLU (diffusers GEGLU: h, gate = proj.chunk(2, -1); h * gelu(gate)).  x: [rows, 2H], y: [rows, H]
// (round 6: the activation is a TEMPLATE parameter of the act / GEGLU kernels -- with a runtime code the 5-way switch sat inside the unrolled element loop, 9 scalar
//  branches per element: 165 in geglu_bwd; the launchers dispatch.  The kernels keep the `act` argument for the launch signature only.)
template <typename T, int ACTC>
__global__ void __launch_bounds__(EW_BLOCK) geglu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long rows, long H, int act) {
    constexpr int V = Elem<T>::VEC;
    const long hv = H / V, total = rows * hv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / hv, c = (i - r * hv) * V;
        Vec16<T> vh, vg; vh.load(x + r * 2 * H + c); vg.load(x + r * 2 * H + H + c);
        float fh[V], fg[V]; vh.unpack(fh); vg.unpack(fg);
#pragma unroll
        for (int j = 0; j < V; ++j) fh[j] *= act_fwd(fg[j], ACTC);
        vh.pack(fh); vh.store(y + r * H + c);
    }
}
template <typename T, int ACTC>
__global__ void __launch_bounds__(EW_BLOCK) geglu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ gy, T* __restrict__ gx, long rows, long H, int act) {
    constexpr int V = Elem<T>::VEC;
    const long hv = H / V, total = rows * hv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / hv, c = (i - r * hv) * V;
        Vec16<T> vh, vg, vy; vh.load(x + r * 2 * H + c); vg.load(x + r * 2 * H + H + c); vy.load(gy + r * H + c);
        float fh[V], fg[V], fy[V], dh[V], dg[V]; vh.unpack(fh); vg.unpack(fg); vy.unpack(fy);
#pragma unroll
        for (int j = 0; j < V; ++j) { dh[j] = fy[j] * act_fwd(fg[j], ACTC); dg[j] = fy[j] * fh[j] * act_bwd(fg[j], ACTC); }
        vh.pack(dh); vh.store(gx + r * 2 * H + c);
        vg.pack(dg); vg.store(gx + r * 2 * H + H + c);
    }
}

// ------------------------------------------------------- precise row linear (round 6): fp32 per-channel addends
// A handful of rows [R, K] of fp32 values (the SDXL time embedding `emb`, one row per sample) enter MFMA GEMMs whose results are added to EVERY pixel of a channel
// (diffusers ResnetBlock2D: conv1(h) + time_emb_proj(silu(emb))[:, :, None, None]).  A bf16 rounding of such a row is not noise: it shifts a whole channel, and the
// gradient norm of the network follows the mean of the residual.  So the row keeps fp32 accuracy through the bf16 MFMA path: the (activated)
// operand is split into a bf16 hi / lo row pair (hi + lo = value to 2^-17), the GEMM runs over 2R rows with an fp32 result, and the two result rows are added
// together with the biases in fp32 -- the result leaves either as fp32 rows or again as a hi / lo bf16 pair (a convolution's bias operand, DPIPE_CONV_BIAS_HILO).
__global__ void __launch_bounds__(EW_BLOCK) rowsplit_fwd_kernel(const float* __restrict__ x, bf16_t* __restrict__ hl, long n, int act) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = act_fwd(x[i], act);
        const bf16_t hi = f32_to_bf16(v);
        hl[i] = hi;
        hl[n + i] = f32_to_bf16(v - bf16_to_f32(hi));
    }
}
// dx = d(hi + lo) * act'(x): the gradient of either row of the pair IS the gradient of their sum
__global__ void __launch_bounds__(EW_BLOCK) rowsplit_bwd_kernel(const bf16_t* __restrict__ ds, const float* __restrict__ x, float* __restrict__ dx, long n, int act) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dx[i] = bf16_to_f32(ds[i]) * act_bwd(x[i], act);
}
// y[r, c] = g[r, c] + g[R + r, c] + bias[c] + extra[c]  ->  out32[r, c] and / or the pair out_hl[r, c], out_hl[R N + r N + c]
__global__ void __launch_bounds__(EW_BLOCK) rowcombine_fwd_kernel(const float* __restrict__ g, const bf16_t* __restrict__ bias, const bf16_t* __restrict__ extra,
                                                                  float* __restrict__ out32, bf16_t* __restrict__ out_hl, int R, int N) {
    const long n = (long)R * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % N);
        float v = g[i] + g[n + i];
        if (bias) v += bf16_to_f32(bias[c]);
        if (extra) v += bf16_to_f32(extra[c]);
        if (out32) out32[i] = v;
        if (out_hl) {
            const bf16_t hi = f32_to_bf16(v);
            out_hl[i] = hi;
            out_hl[n + i] = f32_to_bf16(v - bf16_to_f32(hi));
        }
    }
}
// gy[r, c] = gy[R + r, c] = bf16(gout[r, c]) (both GEMM result rows of a pair receive the gradient of their sum);  dbias[c] (+)= sum_r gout[r, c], likewise dextra
template <typename G>
__global__ void __launch_bounds__(EW_BLOCK) rowcombine_bwd_kernel(const G* __restrict__ gout, bf16_t* __restrict__ gy, bf16_t* __restrict__ dbias, int dbias_acc,
                                                                  bf16_t* __restrict__ dextra, int dextra_acc, int R, int N) {
    const long n = (long)R * N;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < N; c += gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) {
            const float v = Elem<G>::to_f(gout[(long)r * N + c]);
            const bf16_t b = f32_to_bf16(v);
            gy[(long)r * N + c] = b; gy[n + (long)r * N + c] = b;
            s += v;
        }
        if (dbias) dbias[c] = f32_to_bf16(dbias_acc ? bf16_to_f32(dbias[c]) + s : s);
        if (dextra) dextra[c] = f32_to_bf16(dextra_acc ? bf16_to_f32(dextra[c]) + s : s);
    }
}

// ------------------------------------------------------- gated residual (K5)
// out[r, :] = x[r, :] + y[r, :] * gate[r / rows_per_gate, :]   (gate may be null => plain add)
template <typename T, typename G>
__global__ void __launch_bounds__(EW_BLOCK) gated_residual_fwd_kernel(
    const T* __restrict__ x, const T* __restrict__ y, const G* __restrict__ gate, T* __restrict__ out,
    long rows, long D, long rows_per_gate) {
    constexpr int V = Elem<T>::VEC;
    const long dv = D / V, total = rows * dv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / dv, c = (i - r * dv) * V;
        Vec16<T> vx, vy; vx.load(x + r * D + c); vy.load(y + r * D + c);
        float fx[V], fy[V]; vx.unpack(fx); vy.unpack(fy);
        PVec<G, V> pg; pg.load(gate, (r / rows_per_gate) * D + c);          // 16-byte, branch-free gate read (dpipe_common.h, PVec)
        float fgt[V]; pg.unpack(fgt);
#pragma unroll
        for (int j = 0; j < V; ++j) fx[j] += fy[j] * (gate ? fgt[j] : 1.f);
        vx.pack(fx); vx.store(out + r * D + c);
    }
}
// dy = g * gate ; dgate partial sums are produced per (batch, row-slab) then reduced by dgate_final.
// grid = (D / (V*64) rounded, slabs, batches)
template <typename T, typename G>
__global__ void __launch_bounds__(EW_BLOCK) gated_residual_bwd_kernel(
    const T* __restrict__ gout, const T* __restrict__ y, const G* __restrict__ gate,
    T* __restrict__ gy, float* __restrict__ dgate_partial, long rows_per_gate, long D, int slabs) {
    constexpr int V = Elem<T>::VEC;
    const long b = blockIdx.z;
    const long col = ((long)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (col >= D) return;
    const long rows_per_slab = cdiv(rows_per_gate, slabs);
    const long r0 = blockIdx.y * rows_per_slab;
    const long r1 = min(r0 + rows_per_slab, rows_per_gate);
    float fg[V];
#pragma unroll
    for (int j = 0; j < V; ++j) fg[j] = Elem<G>::to_f(gate[b * D + col + j]);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (long r = r0; r < r1; ++r) {
        const long off = (b * rows_per_gate + r) * D + col;
        Vec16<T> vgo, vy; vgo.load(gout + off); vy.load(y + off);
        float fgo[V], fy[V], o[V]; vgo.unpack(fgo); vy.unpack(fy);
#pragma unroll
        for (int j = 0; j < V; ++j) { o[j] = fgo[j] * fg[j]; acc[j] += fgo[j] * fy[j]; }
        Vec16<T> vo; vo.pack(o); vo.store(gy + off);
    }
#pragma unroll
    for (int j = 0; j < V; ++j) dgate_partial[((long)b * slabs + blockIdx.y) * D + col + j] = acc[j];
}
template <typename G>
__global__ void __launch_bounds__(EW_BLOCK) slab_reduce_kernel(const float* __restrict__ partial, G* __restrict__ out, long batches, long D, int slabs) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batches * D) return;
    const long b = i / D, c = i - b * D;
    float a = 0.f;
    for (int s = 0; s < slabs; ++s) a += partial[(b * slabs + s) * D + c];
    out[i] = Elem<G>::from_f(a);
}

// ---------------------------------------------------- timestep embedding (K7)
// out[i, :] = [cos(t_i * f_k) | sin(t_i * f_k)]  (flip=0, Wan) or [sin | cos] (flip=1),
// f_k = max_period^(-k / (half - downscale_shift))
__global__ void sinusoidal_embed_kernel(const float* __restrict__ t, float* __restrict__ out, long n, int dim,
                                        float max_period, int sin_first, float downscale_shift, float scale) {
    const int half = dim / 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * half) return;
    const long r = i / half; const int k = (int)(i - r * half);
    const float freq = expf(-logf(max_period) * (float)k / ((float)half - downscale_shift));
    const float a = scale * t[r] * freq;
    const float c = cosf(a), s = sinf(a);
    out[r * dim + k] = sin_first ? s : c;
    out[r * dim + half + k] = sin_first ? c : s;
}

// ---------------------------------------------------- flow-matching prep (K8)
// x_t = (1 - t_b) * x1 + t_b * x0 ; target = x0 - x1      (x1 = latents, x0 = noise)
__global__ void __launch_bounds__(EW_BLOCK) flow_match_prep_kernel(const float* __restrict__ x1, const float* __restrict__ x0,
                                                                  const float* __restrict__ t, float* __restrict__ xt,
                                                                  float* __restrict__ target, long per_sample, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float tb = t[i / per_sample];
        const float a = x1[i], b = x0[i];
        xt[i] = (1.f - tb) * a + tb * b;
        target[i] = b - a;
    }
}

// ------------------------------------------------ multi-tensor grad norm (K10)
// The host uploads a chunk table: chunk c covers elements [chunk_off[c], chunk_off[c] + chunk_len[c])
// of tensor ptrs[chunk_tensor[c]].  One block per chunk.
template <typename T>
__global__ void __launch_bounds__(EW_BLOCK) multi_sumsq_kernel(const void* const* __restrict__ ptrs, const int* __restrict__ chunk_tensor,
                                                              const long* __restrict__ chunk_off, const int* __restrict__ chunk_len,
                                                              float* __restrict__ partials) {
    __shared__ float smem[16];
    constexpr int V = Elem<T>::VEC;
    const int c = blockIdx.x;
    const T* p = reinterpret_cast<const T*>(ptrs[chunk_tensor[c]]) + chunk_off[c];
    const int len = chunk_len[c];
    float acc = 0.f;
    const bool aligned = ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
    int start = 0;
    if (aligned) {
        const int nv = len / V;
        for (int i = threadIdx.x; i < nv; i += blockDim.x) {
            Vec16<T> v; v.load(p + (long)i * V);
            float f[V]; v.unpack(f);
#pragma unroll
            for (int j = 0; j < V; ++j) acc += f[j] * f[j];
        }
        start = nv * V;
    }
    for (int i = start + threadIdx.x; i < len; i += blockDim.x) { float f = Elem<T>::to_f(p[i]); acc += f * f; }
    float s = block_sum(acc, smem);
    if (threadIdx.x == 0) partials[c] = s;
}
__global__ void __launch_bounds__(EW_BLOCK) sum_partials_kernel(const float* __restrict__ partials, int n, float* __restrict__ out) {
    __shared__ float smem[16];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += partials[i];
    float s = block_sum(a, smem);
    if (threadIdx.x == 0) out[0] = s;
}
// grads *= min(1, max_norm / (sqrt(total_sumsq) + 1e-6))   (utils/patches.py:240-245)
template <typename T>
__global__ void __launch_bounds__(EW_BLOCK) multi_clip_scale_kernel(void* const* __restrict__ ptrs, const int* __restrict__ chunk_tensor,
                                                                   const long* __restrict__ chunk_off, const int* __restrict__ chunk_len,
                                                                   const float* __restrict__ total_sumsq, float max_norm) {
    constexpr int V = Elem<T>::VEC;
    const float norm = sqrtf(total_sumsq[0]);
    const float coef = fminf(1.f, max_norm / (norm + 1e-6f));
    if (coef >= 1.f) return;  // multiplying by exactly 1.0 is the identity
    const int c = blockIdx.x;
    T* p = reinterpret_cast<T*>(ptrs[chunk_tensor[c]]) + chunk_off[c];
    const int len = chunk_len[c];
    const bool aligned = ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
    int start = 0;
    if (aligned) {
        const int nv = len / V;
        for (int i = threadIdx.x; i < nv; i += blockDim.x) {
            Vec16<T> v; v.load(p + (long)i * V);
            float f[V]; v.unpack(f);
#pragma unroll
            for (int j = 0; j < V; ++j) f[j] *= coef;
            v.pack(f); v.store(p + (long)i * V);
        }
        start = nv * V;
    }
    for (int i = start + threadIdx.x; i < len; i += blockDim.x) p[i] = Elem<T>::from_f(Elem<T>::to_f(p[i]) * coef);
}

}  // namespace

#define ACT_SWITCH(act, LAUNCH) switch (act) { case ACT_GELU_TANH: { LAUNCH(ACT_GELU_TANH); } break; case ACT_GELU_ERF: { LAUNCH(ACT_GELU_ERF); } break; \
    case ACT_SILU: { LAUNCH(ACT_SILU); } break; case ACT_QUICK_GELU: { LAUNCH(ACT_QUICK_GELU); } break; default: { LAUNCH(ACT_NONE); } }
#define STREAM(s) reinterpret_cast<hipStream_t>(s)

extern "C" {

int dpipe_loss_workspace_floats(long rows, long cols) {
    long chunks = cdiv(cols, (long)EW_BLOCK * 8 * 4);
    if (chunks < 1) chunks = 1;
    if (chunks > 1024) chunks = 1024;
    return (int)(rows * chunks);
}

int dpipe_loss_fwd(const void* out, int dtype, const float* target, const float* mask, const float* row_weight,
                   long rows, long cols, int kind, float param, float* workspace, float* loss, float* row_loss, void* stream) {
    if (!out || !target || !workspace || !loss || rows <= 0 || cols <= 0) { set_last_error("dpipe_loss_fwd: bad argument"); return DPIPE_ERR_ARG; }
    int chunks = dpipe_loss_workspace_floats(1, cols);
    dim3 grid(chunks, (unsigned)rows);
    if (dtype == DPIPE_BF16)
        loss_fwd_partial_kernel<bf16_t><<<grid, EW_BLOCK, 0, STREAM(stream)>>>((const bf16_t*)out, target, mask, cols, kind, param, workspace);
    else if (dtype == DPIPE_F32)
        loss_fwd_partial_kernel<float><<<grid, EW_BLOCK, 0, STREAM(stream)>>>((const float*)out, target, mask, cols, kind, param, workspace);
    else { set_last_error("dpipe_loss_fwd: dtype"); return DPIPE_ERR_UNSUPPORTED; }
    loss_fwd_final_kernel<<<1, EW_BLOCK, 0, STREAM(stream)>>>(workspace, row_weight, (int)rows, chunks, cols, loss, row_loss);
    return check_launch("dpipe_loss_fwd");
}

int dpipe_loss_bwd(const void* out, int dtype, const float* target, const float* mask, const float* row_weight,
                   const float* grad_loss, long rows, long cols, int kind, float param, void* grad_out, void* stream) {
    if (!out || !target || !grad_loss || !grad_out || rows <= 0 || cols <= 0) { set_last_error("dpipe_loss_bwd: bad argument"); return DPIPE_ERR_ARG; }
    long per = (long)EW_BLOCK * (dtype == DPIPE_BF16 ? 8 : 4);
    long gx = cdiv(cols, per); if (gx > 1024) gx = 1024; if (gx < 1) gx = 1;
    dim3 grid((unsigned)gx, (unsigned)rows);
    if (dtype == DPIPE_BF16)
        loss_bwd_kernel<bf16_t><<<grid, EW_BLOCK, 0, STREAM(stream)>>>((const bf16_t*)out, target, mask, row_weight, grad_loss, rows, cols, kind, param, (bf16_t*)grad_out);
    else if (dtype == DPIPE_F32)
        loss_bwd_kernel<float><<<grid, EW_BLOCK, 0, STREAM(stream)>>>((const float*)out, target, mask, row_weight, grad_loss, rows, cols, kind, param, (float*)grad_out);
    else { set_last_error("dpipe_loss_bwd: dtype"); return DPIPE_ERR_UNSUPPORTED; }
    return check_launch("dpipe_loss_bwd");
}

int dpipe_upsample2x_adjoint(const void* src, void* dst, int B, int H, int W, int C, int dtype, void* stream) {
    const int V = dtype == DPIPE_BF16 ? 8 : 4;
    if (!src || !dst || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % V) != 0 || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15)) {
        set_last_error("dpipe_upsample2x_adjoint: needs 16-byte aligned tensors and C a multiple of the 16-byte vector"); return DPIPE_ERR_ARG; }
    const long npix = (long)B * H * W;
    const int grid = stream_grid(npix * (C / V), EW_BLOCK);
    if (dtype == DPIPE_BF16) upsample2x_adjoint_kernel<bf16_t><<<grid, EW_BLOCK, 0, STREAM(stream)>>>((const bf16_t*)src, (bf16_t*)dst, npix, H, W, C);
    else if (dtype == DPIPE_F32) upsample2x_adjoint_kernel<float><<<grid, EW_BLOCK, 0, STREAM(stream)>>>((const float*)src, (float*)dst, npix, H, W, C);
    else { set_last_error("dpipe_upsample2x_adjoint: dtype"); return DPIPE_ERR_UNSUPPORTED; }
    return check_launch("dpipe_upsample2x_adjoint");
}

int dpipe_act_fwd(const void* x, void* y, long n, int dtype, int act, void* stream) {
    if (!x || !y || n < 0) { set_last_error("dpipe_act_fwd: bad argument"); return DPIPE_ERR_ARG; }
    if (n == 0 || ablated(ABL_EW)) return DPIPE_OK;
#define L_BF(A) act_fwd_kernel<bf16_t, A><<<stream_grid(n / 8 + 1, EW_BLOCK), EW_BLOCK, 0, STREAM(stream)>>>((const bf16_t*)x, (bf16_t*)y, n, act)
#define L_F32(A) act_fwd_kernel<float, A><<<stream_grid(n / 4 + 1, EW_BLOCK), EW_BLOCK, 0, STREAM(stream)>>>((const float*)x, (float*)y, n, act)
    if (dtype == DPIPE_BF16) ACT_SWITCH(act, L_BF)
    else if (dtype == DPIPE_F32) ACT_SWITCH(act, L_F32)
    else { set_last_error("dpipe_act_fwd: dtype"); return DPIPE_ERR_UNSUPPORTED; }
#undef L_BF
#undef L_F32
    return check_launch("dpipe_act_fwd");
}

int dpipe_act_bwd(const void* x, const void* gy, void* gx, long n, int dtype, int act, void* stream) {
    if (!x || !gy || !gx || n < 0) { set_last_error("dpipe_act_bwd: bad argument"); return DPIPE_ERR_ARG; }
    if (n == 0 || ablated(ABL_EW)) return DPIPE_OK;
#define L_BF(A) act_bwd_kernel<bf16_t, A><<<stream_grid(n / 8 + 1, EW_BLOCK), EW_BLOCK, 0, STREAM(stream)>>>((const bf16_t*)x, (const bf16_t*)gy, (bf16_t*)gx, n, act)
#define L_F32(A) act_bwd_kernel<float, A><<<stream_grid(n / 4 + 1, EW_BLOCK), EW_BLOCK, 0, STREAM(stream)>>>((const float*)x, (const float*)gy, (float*)gx, n, act)
    if (dtype == DPIPE_BF16) ACT_SWITCH(act, L_BF)
    else if (dtype == DPIPE_F32) ACT_SWITCH(act, L_F32)
    else { set_last_error("dpipe_act_bwd: dtype"); return DPIPE_ERR_UNSUPPORTED; }
#undef L_BF
#undef L_F32
    return check_launch("dpipe_act_bwd");
}

int dpipe_rowsplit_fwd(const float* x, void* hl, long n, int act, void* stream) {
    if (!x || !hl || n <= 0) { set_last_error("dpipe_rowsplit_fwd: bad argument"); return DPIPE_ERR_ARG; }
    rowsplit_fwd_kernel<<<stream_grid(n, EW_BLOCK), EW_BLOCK, 0, STREAM(stream)>>>(x, (bf16_t*)hl, n, act);
    return check_launch("dpipe_rowsplit_fwd");
}

int dpipe_rowsplit_bwd(const void* ds, const float* x, float* dx, long n, int act, void* stream) {
    if (!ds || !x || !dx || n <= 0) { set_last_error("dpipe_rowsplit_bwd: bad argument"); return DPIPE_ERR_ARG; }
    rowsplit_bwd_kernel<<<stream_grid(n, EW_BLOCK), EW_BLOCK, 0, STREAM(stream)>>>((const bf16_t*)ds, x, dx, n, act);
    return check_launch("dpipe_rowsplit_bwd");
}

int dpipe_rowcombine_fwd(const float* g, const void* bias, const void* extra, float* out32, void* out_hl, int R, int N, void* stream) {
    if (!g || (!out32 && !out_hl) || R <= 0 || N <= 0) { set_last_error("dpipe_rowcombine_fwd: bad argument"); return DPIPE_ERR_ARG; }
    rowcombine_fwd_kernel<<<stream_grid((long)R * N, EW_BLOCK), EW_BLOCK, 0, STREAM(stream)>>>(g, (const bf16_t*)bias, (const bf16_t*)extra, out32, (bf16_t*)out_hl, R, N);
    return check_launch("dpipe_rowcombine_fwd");
}

int dpipe_rowcombine_bwd(const void* gout, int gout_dtype, void* gy, void* dbias, int dbias_accumulate, void* dextra, int dextra_accumulate, int R, int N, void* stream) {
    if (!gout || !gy || R <= 0 || N <= 0) { set_last_error("dpipe_rowcombine_bwd: bad argument"); return DPIPE_ERR_ARG; }
    const int grid = stream_grid(N, EW_BLOCK);
    if (gout_dtype == DPIPE_F32) rowcombine_bwd_kernel<float><<<grid, EW_BLOCK, 0, STREAM(stream)>>>((const float*)gout, (bf16_t*)gy, (bf16_t*)dbias, dbias_accumulate, (bf16_t*)dextra, dextra_accumulate, R, N);
    else if (gout_dtype == DPIPE_BF16) rowcombine_bwd_kernel<bf16_t><<<grid, EW_BLOCK, 0, STREAM(stream)>>>((const bf16_t*)gout, (bf16_t*)gy, (bf16_t*)dbias, dbias_accumulate, (bf16_t*)dextra, dextra_accumulate, R, N);
    else { set_last_error("dpipe_rowcombine_bwd: dtype"); return DPIPE_ERR_UNSUPPORTED; }
    return check_launch("dpipe_rowcombine_bwd");
}

int dpipe_geglu_fwd(const void* x, void* y, long rows, long H, int dtype, int act, void* stream) {
    const int V = dtype == DPIPE_BF16 ? 8 : 4;
    if (!x || !y || rows <= 0 || H <= 0 || (H % V) != 0) { set_last_error("dpipe_geglu_fwd: H must be a multiple of the 16-byte vector"); return DPIPE_ERR_ARG; }
    if (ablated(ABL_EW)) return DPIPE_OK;
    int grid = stream_grid(rows * (H / V), EW_BLOCK);
#define L_BF(A) geglu_fwd_kernel<bf16_t, A><<<grid, EW_BLOCK, 0, STREAM(stream)>>>((const bf16_t*)x, (bf16_t*)y, rows, H, act)
#define L_F32(A) geglu_fwd_kernel<float, A><<<grid, EW_BLOCK, 0, STREAM(stream)>>>((const float*)x, (float*)y, rows, H, act)
    if (dtype == DPIPE_BF16) ACT_SWITCH(act, L_BF)
    else ACT_SWITCH(act, L_F32)
#undef L_BF
#undef L_F32
    return check_launch("dpipe_geglu_fwd");
}

int dpipe_geglu_bwd(const void* x, const void* gy, void* gx, long rows, long H, int dtype, int act, void* stream) {
    const int V = dtype == DPIPE_BF16 ? 8 : 4;
    if (!x || !gy || !gx || rows <= 0 || H <= 0 || (H % V) != 0) { set_last_error("dpipe_geglu_bwd: bad argument"); return DPIPE_ERR_ARG; }
    if (ablated(ABL_EW)) return DPIPE_OK;
    int grid = stream_grid(rows * (H / V), EW_BLOCK);
#define L_BF(A) geglu_bwd_kernel<bf16_t, A><<<grid, EW_BLOCK, 0, STREAM(stream)>>>((const bf16_t*)x, (const bf16_t*)gy, (bf16_t*)gx, rows, H, act)
#define L_F32(A) geglu_bwd_kernel<float, A><<<grid, EW_BLOCK, 0, STREAM(stream)>>>((const float*)x, (const float*)gy, (float*)gx, rows, H, act)
    if (dtype == DPIPE_BF16) ACT_SWITCH(act, L_BF)
    else ACT_SWITCH(act, L_F32)
#undef L_BF
#undef L_F32
    return check_launch("dpipe_geglu_bwd");
}

int dpipe_gated_residual_fwd(const void* x, const void* y, const void* gate, void* out, long rows, long D, long rows_per_gate,
                             int dtype, int gate_dtype, void* stream) {
    const int V = dtype == DPIPE_BF16 ? 8 : 4;
    if (!x || !y || !out || rows <= 0 || D <= 0 || (D % V) != 0 || rows_per_gate <= 0 || !aligned16(gate)) { set_last_error("dpipe_gated_residual_fwd: bad argument (or a gate that is not 16-byte aligned)"); return DPIPE_ERR_ARG; }
    int grid = stream_grid(rows * (D / V), EW_BLOCK);
    hipStream_t s = STREAM(stream);
    if (dtype == DPIPE_BF16 && gate_dtype == DPIPE_BF16) gated_residual_fwd_kernel<bf16_t, bf16_t><<<grid, EW_BLOCK, 0, s>>>((const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)gate, (bf16_t*)out, rows, D, rows_per_gate);
    else if (dtype == DPIPE_BF16) gated_residual_fwd_kernel<bf16_t, float><<<grid, EW_BLOCK, 0, s>>>((const bf16_t*)x, (const bf16_t*)y, (const float*)gate, (bf16_t*)out, rows, D, rows_per_gate);
    else if (gate_dtype == DPIPE_F32) gated_residual_fwd_kernel<float, float><<<grid, EW_BLOCK, 0, s>>>((const float*)x, (const float*)y, (const float*)gate, (float*)out, rows, D, rows_per_gate);
    else { set_last_error("dpipe_gated_residual_fwd: fp32 data with bf16 gate unsupported"); return DPIPE_ERR_UNSUPPORTED; }
    return check_launch("dpipe_gated_residual_fwd");
}

int dpipe_gated_residual_slabs(long rows_per_gate) {
    long s = rows_per_gate / 256; if (s < 1) s = 1; if (s > 64) s = 64; return (int)s;
}

int dpipe_gated_residual_bwd(const void* gout, const void* y, const void* gate, void* gy, void* dgate, float* workspace,
                             long batches, long rows_per_gate, long D, int dtype, int gate_dtype, void* stream) {
    const int V = dtype == DPIPE_BF16 ? 8 : 4;
    if (!gout || !y || !gate || !gy || !dgate || !workspace || batches <= 0 || rows_per_gate <= 0 || (D % V) != 0) { set_last_error("dpipe_gated_residual_bwd: bad argument"); return DPIPE_ERR_ARG; }
    const int slabs = dpipe_gated_residual_slabs(rows_per_gate);
    dim3 grid((unsigned)cdiv(D / V, EW_BLOCK), slabs, (unsigned)batches);
    hipStream_t s = STREAM(stream);
    int rgrid = (int)cdiv(batches * D, EW_BLOCK);
    if (dtype == DPIPE_BF16 && gate_dtype == DPIPE_BF16) {
        gated_residual_bwd_kernel<bf16_t, bf16_t><<<grid, EW_BLOCK, 0, s>>>((const bf16_t*)gout, (const bf16_t*)y, (const bf16_t*)gate, (bf16_t*)gy, workspace, rows_per_gate, D, slabs);
        slab_reduce_kernel<bf16_t><<<rgrid, EW_BLOCK, 0, s>>>(workspace, (bf16_t*)dgate, batches, D, slabs);
    } else if (dtype == DPIPE_BF16) {
        gated_residual_bwd_kernel<bf16_t, float><<<grid, EW_BLOCK, 0, s>>>((const bf16_t*)gout, (const bf16_t*)y, (const float*)gate, (bf16_t*)gy, workspace, rows_per_gate, D, slabs);
        slab_reduce_kernel<float><<<rgrid, EW_BLOCK, 0, s>>>(workspace, (float*)dgate, batches, D, slabs);
    } else if (gate_dtype == DPIPE_F32) {
        gated_residual_bwd_kernel<float, float><<<grid, EW_BLOCK, 0, s>>>((const float*)gout, (const float*)y, (const float*)gate, (float*)gy, workspace, rows_per_gate, D, slabs);
        slab_reduce_kernel<float><<<rgrid, EW_BLOCK, 0, s>>>(workspace, (float*)dgate, batches, D, slabs);
    } else { set_last_error("dpipe_gated_residual_bwd: dtype combination"); return DPIPE_ERR_UNSUPPORTED; }
    return check_launch("dpipe_gated_residual_bwd");
}

int dpipe_sinusoidal_embed(const float* t, float* out, long n, int dim, float max_period, int sin_first, float downscale_shift, float scale, void* stream) {
    if (!t || !out || n <= 0 || dim <= 0 || (dim & 1)) { set_last_error("dpipe_sinusoidal_embed: dim must be even"); return DPIPE_ERR_ARG; }
    long total = n * (dim / 2);
    sinusoidal_embed_kernel<<<(unsigned)cdiv(total, 256), 256, 0, STREAM(stream)>>>(t, out, n, dim, max_period, sin_first, downscale_shift, scale);
    return check_launch("dpipe_sinusoidal_embed");
}

int dpipe_flow_match_prep(const float* x1, const float* x0, const float* t, float* xt, float* target, long batch, long per_sample, void* stream) {
    if (!x1 || !x0 || !t || !xt || !target || batch <= 0 || per_sample <= 0) { set_last_error("dpipe_flow_match_prep: bad argument"); return DPIPE_ERR_ARG; }
    long n = batch * per_sample;
    flow_match_prep_kernel<<<stream_grid(n, EW_BLOCK), EW_BLOCK, 0, STREAM(stream)>>>(x1, x0, t, xt, target, per_sample, n);
    return check_launch("dpipe_flow_match_prep");
}

int dpipe_multi_sumsq(const void* const* ptrs, const int* chunk_tensor, const long* chunk_off, const int* chunk_len, int nchunks,
                      int dtype, float* partials, float* out_sumsq, void* stream) {
    if (!partials || !out_sumsq || nchunks < 0) { set_last_error("dpipe_multi_sumsq: bad argument"); return DPIPE_ERR_ARG; }
    hipStream_t s = STREAM(stream);
    if (nchunks > 0) {
        if (!ptrs || !chunk_tensor || !chunk_off || !chunk_len) { set_last_error("dpipe_multi_sumsq: null table"); return DPIPE_ERR_ARG; }
        if (dtype == DPIPE_BF16) multi_sumsq_kernel<bf16_t><<<nchunks, EW_BLOCK, 0, s>>>(ptrs, chunk_tensor, chunk_off, chunk_len, partials);
        else if (dtype == DPIPE_F32) multi_sumsq_kernel<float><<<nchunks, EW_BLOCK, 0, s>>>(ptrs, chunk_tensor, chunk_off, chunk_len, partials);
        else { set_last_error("dpipe_multi_sumsq: dtype"); return DPIPE_ERR_UNSUPPORTED; }
    }
    sum_partials_kernel<<<1, EW_BLOCK, 0, s>>>(partials, nchunks, out_sumsq);
    return check_launch("dpipe_multi_sumsq");
}

int dpipe_multi_clip_scale(void* const* ptrs, const int* chunk_tensor, const long* chunk_off, const int* chunk_len, int nchunks,
                           int dtype, const float* total_sumsq, float max_norm, void* stream) {
    if (nchunks <= 0) return DPIPE_OK;
    if (!ptrs || !chunk_tensor || !chunk_off || !chunk_len || !total_sumsq) { set_last_error("dpipe_multi_clip_scale: bad argument"); return DPIPE_ERR_ARG; }
    hipStream_t s = STREAM(stream);
    if (dtype == DPIPE_BF16) multi_clip_scale_kernel<bf16_t><<<nchunks, EW_BLOCK, 0, s>>>(ptrs, chunk_tensor, chunk_off, chunk_len, total_sumsq, max_norm);
    else if (dtype == DPIPE_F32) multi_clip_scale_kernel<float><<<nchunks, EW_BLOCK, 0, s>>>(ptrs, chunk_tensor, chunk_off, chunk_len, total_sumsq, max_norm);
    else { set_last_error("dpipe_multi_clip_scale: dtype"); return DPIPE_ERR_UNSUPPORTED; }
    return check_launch("dpipe_multi_clip_scale");
}

}  // extern "C"
