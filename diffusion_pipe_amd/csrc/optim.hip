// optim.hip -- the step-end pass of train_batch as two HBM-bound multi-tensor kernels (SURVEY.md 8(f) row 4, 8(d)'s "HBM
// side-term"): the reference runs clip_grad_norm_ (utils/patches.py:175-246) and torch.optim.AdamW (train.py:672-678) as
// separate passes over the gradients; with concurrent micro-batch lanes the engine additionally sums the lanes' gradient
// accumulators.  Here
//   pass 1  adamw_sumsq_kernel : sum_p ( sum_lanes g )^2            -> fp32 partial per chunk  (reads L x 2 B / element)
//   pass 2  adamw_step_kernel  : g = clip_coef * sum_lanes g ; AdamW update of (p, m, v) in fp32 ; lanes zeroed
//                                (reads (L + 3) x 2 B, writes (L + 3) x 2 B per element, nothing else touches the gradients)
// Chunk table as in elementwise.hip: chunk c covers [chunk_off[c], + chunk_len[c]) of tensor chunk_tensor[c]; pointer tables are
// int64 device arrays, gradients laid out [tensor][lane].  One block per chunk, 16-byte vector accesses.
// Update rule = torch's fused AdamW functor in fp32 opmath:
//   p *= 1 - lr * wd ; m = lerp(m, g, 1 - b1) ; v = b2 v + (1 - b2) g^2 ; p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
#include "dpipe_common.h"
#include "../../include/dpipe_hip.h"

using namespace dpipe;

namespace {

constexpr int OPT_BLOCK = 256;
constexpr int MAX_LANES = 8;

template <typename T>
__global__ void __launch_bounds__(OPT_BLOCK) adamw_sumsq_kernel(const void* const* __restrict__ g_ptrs, int lanes, const int* __restrict__ chunk_tensor,
                                                               const long* __restrict__ chunk_off, const int* __restrict__ chunk_len,
                                                               float* __restrict__ partials) {
    __shared__ float smem[16];
    constexpr int V = Elem<T>::VEC;
    const int c = blockIdx.x, t = chunk_tensor[c], len = chunk_len[c];
    const long off = chunk_off[c];
    const T* g[MAX_LANES];
    bool aligned = true;
    for (int l = 0; l < lanes; ++l) {
        g[l] = reinterpret_cast<const T*>(g_ptrs[(long)t * lanes + l]) + off;
        aligned = aligned && ((reinterpret_cast<uintptr_t>(g[l]) & 15) == 0);
    }
    float acc = 0.f;
    int start = 0;
    if (aligned) {
        const int nv = len / V;
        for (int i = threadIdx.x; i < nv; i += blockDim.x) {
            float s[V];
#pragma unroll
            for (int j = 0; j < V; ++j) s[j] = 0.f;
            for (int l = 0; l < lanes; ++l) {
                Vec16<T> v; v.load(g[l] + (long)i * V);
                float f[V]; v.unpack(f);
#pragma unroll
                for (int j = 0; j < V; ++j) s[j] += f[j];
            }
#pragma unroll
            for (int j = 0; j < V; ++j) acc += s[j] * s[j];
        }
        start = nv * V;
    }
    for (int i = start + threadIdx.x; i < len; i += blockDim.x) {
        float s = 0.f;
        for (int l = 0; l < lanes; ++l) s += Elem<T>::to_f(g[l][i]);
        acc += s * s;
    }
    const float r = block_sum(acc, smem);
    if (threadIdx.x == 0) partials[c] = r;
}

__global__ void __launch_bounds__(OPT_BLOCK) adamw_sum_partials_kernel(const float* __restrict__ partials, int n, float* __restrict__ out, int accumulate) {
    __shared__ float smem[16];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += partials[i];
    const float s = block_sum(a, smem);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + s : s;
}

struct AdamHyper {
    float lr, beta1, beta2, eps, weight_decay, bias_c1, sqrt_bias_c2, max_norm;
};

__device__ __forceinline__ void adamw_update(float& p, float& m, float& v, float g, const AdamHyper& h) {
    p *= 1.f - h.lr * h.weight_decay;
    m = m + (1.f - h.beta1) * (g - m);                      // lerp(m, g, 1 - beta1)
    v = h.beta2 * v + (1.f - h.beta2) * g * g;
    const float denom = sqrtf(v) / h.sqrt_bias_c2 + h.eps;
    p -= (h.lr / h.bias_c1) * m / denom;
}

// Kahan (compensated) application of an update to a low-precision parameter, the reference's sequence with every intermediate rounded to the
// parameter dtype (optimizers/generic_optim.py:486-497): shift += update; old = p; p += shift; shift += old - p.  In: p = the fp32 result of the
// plain update of `old`; out: p = the stored parameter value, shift = the new compensation.
template <typename T>
__device__ __forceinline__ void kahan_apply(float& p, float& shift, float old) {
    auto rnd = [](float x) { return Elem<T>::to_f(Elem<T>::from_f(x)); };
    const float s = rnd(shift + rnd(p - old));
    const float pn = rnd(old + s);
    shift = rnd(s + rnd(old - pn));
    p = pn;
}

template <typename T>
__global__ void __launch_bounds__(OPT_BLOCK) adamw_step_kernel(void* const* __restrict__ p_ptrs, void* const* __restrict__ m_ptrs, void* const* __restrict__ v_ptrs,
                                                              void* const* __restrict__ s_ptrs, void* const* __restrict__ g_ptrs, int lanes, const int* __restrict__ chunk_tensor,
                                                              const long* __restrict__ chunk_off, const int* __restrict__ chunk_len,
                                                              const float* __restrict__ total_sumsq, AdamHyper h, int zero_grads) {
    constexpr int V = Elem<T>::VEC;
    float coef = 1.f;
    if (total_sumsq != nullptr && h.max_norm > 0.f) coef = fminf(1.f, h.max_norm / (sqrtf(total_sumsq[0]) + 1e-6f));   // utils/patches.py:240-245
    const int c = blockIdx.x, t = chunk_tensor[c], len = chunk_len[c];
    const long off = chunk_off[c];
    T* p = reinterpret_cast<T*>(p_ptrs[t]) + off;
    T* m = reinterpret_cast<T*>(m_ptrs[t]) + off;
    T* v = reinterpret_cast<T*>(v_ptrs[t]) + off;
    T* sh = s_ptrs ? reinterpret_cast<T*>(s_ptrs[t]) + off : nullptr;      // Kahan compensation buffer (optimizers/generic_optim.py:486-497), parameter dtype
    T* g[MAX_LANES];
    bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(sh)) & 15) == 0;
    for (int l = 0; l < lanes; ++l) {
        g[l] = reinterpret_cast<T*>(g_ptrs[(long)t * lanes + l]) + off;
        aligned = aligned && ((reinterpret_cast<uintptr_t>(g[l]) & 15) == 0);
    }
    int start = 0;
    if (aligned) {
        const int nv = len / V;
        Vec16<T> zero; zero.raw = make_uint4(0u, 0u, 0u, 0u);
        for (int i = threadIdx.x; i < nv; i += blockDim.x) {
            const long e = (long)i * V;
            float gs[V];
#pragma unroll
            for (int j = 0; j < V; ++j) gs[j] = 0.f;
            for (int l = 0; l < lanes; ++l) {
                Vec16<T> gv; gv.load(g[l] + e);
                float f[V]; gv.unpack(f);
#pragma unroll
                for (int j = 0; j < V; ++j) gs[j] += f[j];
                if (zero_grads) zero.store(g[l] + e);
            }
            Vec16<T> pv, mv, vv;
            pv.load(p + e); mv.load(m + e); vv.load(v + e);
            float pf[V], mf[V], vf[V];
            pv.unpack(pf); mv.unpack(mf); vv.unpack(vf);
            if (sh) {
                Vec16<T> sv; sv.load(sh + e);
                float sf[V]; sv.unpack(sf);
#pragma unroll
                for (int j = 0; j < V; ++j) { const float old = pf[j]; adamw_update(pf[j], mf[j], vf[j], gs[j] * coef, h); kahan_apply<T>(pf[j], sf[j], old); }
                sv.pack(sf); sv.store(sh + e);
            } else {
#pragma unroll
                for (int j = 0; j < V; ++j) adamw_update(pf[j], mf[j], vf[j], gs[j] * coef, h);
            }
            pv.pack(pf); mv.pack(mf); vv.pack(vf);
            pv.store(p + e); mv.store(m + e); vv.store(v + e);
        }
        start = nv * V;
    }
    for (int i = start + threadIdx.x; i < len; i += blockDim.x) {
        float gs = 0.f;
        for (int l = 0; l < lanes; ++l) {
            gs += Elem<T>::to_f(g[l][i]);
            if (zero_grads) g[l][i] = Elem<T>::from_f(0.f);
        }
        float pf = Elem<T>::to_f(p[i]), mf = Elem<T>::to_f(m[i]), vf = Elem<T>::to_f(v[i]);
        const float old = pf;
        adamw_update(pf, mf, vf, gs * coef, h);
        if (sh) { float sf = Elem<T>::to_f(sh[i]); kahan_apply<T>(pf, sf, old); sh[i] = Elem<T>::from_f(sf); }
        p[i] = Elem<T>::from_f(pf); m[i] = Elem<T>::from_f(mf); v[i] = Elem<T>::from_f(vf);
    }
}

}  // namespace

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

extern "C" {

int dpipe_adamw_sumsq(const void* const* g_ptrs, int lanes, const int* chunk_tensor, const long* chunk_off, const int* chunk_len, int nchunks,
                      int dtype, float* partials, float* out_sumsq, int accumulate, void* stream) {
    if (!partials || !out_sumsq || nchunks < 0 || lanes < 1 || lanes > MAX_LANES) { set_last_error("dpipe_adamw_sumsq: bad argument"); return DPIPE_ERR_ARG; }
    hipStream_t s = STREAM(stream);
    if (nchunks > 0) {
        if (!g_ptrs || !chunk_tensor || !chunk_off || !chunk_len) { set_last_error("dpipe_adamw_sumsq: null table"); return DPIPE_ERR_ARG; }
        if (dtype == DPIPE_BF16) adamw_sumsq_kernel<bf16_t><<<nchunks, OPT_BLOCK, 0, s>>>(g_ptrs, lanes, chunk_tensor, chunk_off, chunk_len, partials);
        else if (dtype == DPIPE_F32) adamw_sumsq_kernel<float><<<nchunks, OPT_BLOCK, 0, s>>>(g_ptrs, lanes, chunk_tensor, chunk_off, chunk_len, partials);
        else { set_last_error("dpipe_adamw_sumsq: dtype"); return DPIPE_ERR_UNSUPPORTED; }
    }
    adamw_sum_partials_kernel<<<1, OPT_BLOCK, 0, s>>>(partials, nchunks, out_sumsq, accumulate);
    return check_launch("dpipe_adamw_sumsq");
}

static int adamw_step_impl(void* const* p_ptrs, void* const* m_ptrs, void* const* v_ptrs, void* const* s_ptrs, void* const* g_ptrs, int lanes, const int* chunk_tensor,
                           const long* chunk_off, const int* chunk_len, int nchunks, int dtype, float lr, float beta1, float beta2, float eps,
                           float weight_decay, float bias_correction1, float bias_correction2, const float* total_sumsq, float max_norm,
                           int zero_grads, void* stream) {
    if (nchunks <= 0) return DPIPE_OK;
    if (!p_ptrs || !m_ptrs || !v_ptrs || !g_ptrs || !chunk_tensor || !chunk_off || !chunk_len || lanes < 1 || lanes > MAX_LANES ||
        bias_correction1 <= 0.f || bias_correction2 <= 0.f) {
        set_last_error("dpipe_adamw_step: bad argument"); return DPIPE_ERR_ARG;
    }
    AdamHyper h{lr, beta1, beta2, eps, weight_decay, bias_correction1, sqrtf(bias_correction2), max_norm};
    hipStream_t s = STREAM(stream);
    if (dtype == DPIPE_BF16)
        adamw_step_kernel<bf16_t><<<nchunks, OPT_BLOCK, 0, s>>>(p_ptrs, m_ptrs, v_ptrs, s_ptrs, g_ptrs, lanes, chunk_tensor, chunk_off, chunk_len, total_sumsq, h, zero_grads);
    else if (dtype == DPIPE_F32)
        adamw_step_kernel<float><<<nchunks, OPT_BLOCK, 0, s>>>(p_ptrs, m_ptrs, v_ptrs, s_ptrs, g_ptrs, lanes, chunk_tensor, chunk_off, chunk_len, total_sumsq, h, zero_grads);
    else { set_last_error("dpipe_adamw_step: dtype"); return DPIPE_ERR_UNSUPPORTED; }
    return check_launch("dpipe_adamw_step");
}

int dpipe_adamw_step(void* const* p_ptrs, void* const* m_ptrs, void* const* v_ptrs, void* const* g_ptrs, int lanes, const int* chunk_tensor,
                     const long* chunk_off, const int* chunk_len, int nchunks, int dtype, float lr, float beta1, float beta2, float eps,
                     float weight_decay, float bias_correction1, float bias_correction2, const float* total_sumsq, float max_norm,
                     int zero_grads, void* stream) {
    return adamw_step_impl(p_ptrs, m_ptrs, v_ptrs, nullptr, g_ptrs, lanes, chunk_tensor, chunk_off, chunk_len, nchunks, dtype, lr, beta1, beta2, eps, weight_decay,
                           bias_correction1, bias_correction2, total_sumsq, max_norm, zero_grads, stream);
}

int dpipe_adamw_step_kahan(void* const* p_ptrs, void* const* m_ptrs, void* const* v_ptrs, void* const* shift_ptrs, void* const* g_ptrs, int lanes,
                           const int* chunk_tensor, const long* chunk_off, const int* chunk_len, int nchunks, int dtype, float lr, float beta1, float beta2,
                           float eps, float weight_decay, float bias_correction1, float bias_correction2, const float* total_sumsq, float max_norm,
                           int zero_grads, void* stream) {
    if (nchunks > 0 && !shift_ptrs) { set_last_error("dpipe_adamw_step_kahan: null shift table"); return DPIPE_ERR_ARG; }
    return adamw_step_impl(p_ptrs, m_ptrs, v_ptrs, shift_ptrs, g_ptrs, lanes, chunk_tensor, chunk_off, chunk_len, nchunks, dtype, lr, beta1, beta2, eps, weight_decay,
                           bias_correction1, bias_correction2, total_sumsq, max_norm, zero_grads, stream);
}

}  // extern "C"
