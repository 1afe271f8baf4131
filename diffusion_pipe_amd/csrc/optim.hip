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

// LN > 0: the lane count is static -- every read of a 16-byte element group (LN gradient vectors, p, m, v, the Kahan buffer) is issued before the first is
// used (round 6: the runtime lane loop waited for each lane's load before it issued the next, five dependent round trips per group with 16 bytes per thread in
// flight -- 4.5 - 4.7 TB/s; the pointers come from tables, i.e. these are flat loads, which count on both wait counters).  LN = 0: any lane count, the old loop.
template <typename T, int LN>
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
            Vec16<T> pv, mv, vv, sv;
            if constexpr (LN > 0) {
                Vec16<T> gv[LN];
#pragma unroll
                for (int l = 0; l < LN; ++l) gv[l].load(g[l] + e);
                pv.load(p + e); mv.load(m + e); vv.load(v + e);
                sv.load(sh ? sh + e : p + e);                      // (no Kahan buffer: a second read of the parameter vector, branch-free)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int l = 0; l < LN; ++l) {
                    float f[V]; gv[l].unpack(f);
#pragma unroll
                    for (int j = 0; j < V; ++j) gs[j] += f[j];
                }
                if (zero_grads) {
#pragma unroll
                    for (int l = 0; l < LN; ++l) zero.store(g[l] + e);
                }
            } else {
                for (int l = 0; l < lanes; ++l) {
                    Vec16<T> gv; gv.load(g[l] + e);
                    float f[V]; gv.unpack(f);
#pragma unroll
                    for (int j = 0; j < V; ++j) gs[j] += f[j];
                    if (zero_grads) zero.store(g[l] + e);
                }
                pv.load(p + e); mv.load(m + e); vv.load(v + e);
                if (sh) sv.load(sh + e);
            }
            float pf[V], mf[V], vf[V];
            pv.unpack(pf); mv.unpack(mf); vv.unpack(vf);
            if (sh) {
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

// ------------------------------------------------------------------------------------------------ 8-bit block-wise AdamW
// The reference's `adamw8bit` / `adamw8bitkahan` optimizers (train.py:673-686 -> bitsandbytes.optim.AdamW8bit, optimizers/adamw_8bit.py): moments stored as
// uint8 codes of a 256-entry dynamic map plus one fp32 absmax per 256 elements.  bitsandbytes is absent here, so this follows the library's published
// algorithm as restated in oracle/adam8bit_ref.py (parity unpinned).  One workgroup = one 256-element block per iteration, one element per thread:
// dequantise -> moments -> block absmax (wave max + LDS) -> parameter update (+ decoupled decay) -> requantise to the nearest code.
// `shift` != NULL: the reference's Kahan variant -- the library kernel updates the shift buffer instead of the parameter (weight decay included), then
// p' = p + shift, shift' = shift + (p - p') in the parameter's dtype.
__device__ __forceinline__ int nearest_code(const float* q, float x) {
    int lo = 0, hi = 255;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int mid = (lo + hi) >> 1;
        if (q[mid] <= x) lo = mid; else hi = mid;
    }
    return (q[hi] - x) < (x - q[lo]) ? hi : lo;
}

// ---- direct quantiser (round 5).  The library's maps are not arbitrary tables: `create_dynamic_map` lays 7 decades 10^-6 .. 10^0 side by side, decade i holding
// 2^i (signed map) / 2^(i+1) (unsigned) EVENLY spaced values in (0.1, 1) x 10^(i-6), plus 0 and 1 (mirrored negatives in the signed map).  So the index of the code
// nearest to x follows from the decade (one v_log_f32, a multiply-add, a floor), one multiply-add and a round (the slot) -- right to within +-1 across decade borders and
// rounding, whatever the last-ulp behaviour of the hardware logarithm (a misjudged decade lands on the neighbouring decade's end slot, which is the adjacent code) -- and
// the exact answer, ties included, is the nearest of the three map entries around the guess: 3 INDEPENDENT LDS reads instead of the binary search's 9 dependent ones per
// moment per element.  Oracle-side emulation incl. a perturbed logarithm: tests/test_optim_cpu.py.  Whether a launch's maps ARE the dynamic maps is checked by the
// workgroup itself (one closed-form comparison per thread while it stages the map into LDS); any other table keeps the search.
// Second pass of round 5 (the kernel had become ALU-bound: ~150 VALU operations per element against 10 bytes): the maps sit in LDS with one sentinel either side, so the
// three candidates need no index clamps; the per-decade constants come from a 7-entry LDS table instead of two select chains; x = moment / absmax is one multiply by the
// block's reciprocal (the exact quotient is kept for blocks whose absmax is below 1e-30, where the reciprocal would overflow); the sign rule of the first moment reduces,
// on the dynamic map, to "code 127 (zero) for a negative moment becomes 126"; and the bf16 parameter update takes v_sqrt_f32 / v_rcp_f32 (1 ulp each; the library's own
// kernel divides with __fdividef) -- fp32 parameters keep the correctly rounded quotient and root.
struct Q8Lds {
    float q1[258], q2[258];      // [k + 1] = map[k]; [0] = -3e38, [257] = +3e38
    float4 dec1[8], dec2[8];     // decade i: {lo = 0.1 x 10^(i-6), slots_i / (0.9 x 10^(i-6)), slots_i - 1, bits of the int (first position of the decade)}: slot = (|x| - lo) * scale - 0.5
};
__device__ __forceinline__ float dyn_map_value(int idx, bool is_signed) {
    int pos; float sgn = 1.f;
    if (is_signed) { if (idx == 127) return 0.f; if (idx == 255) return 1.f; pos = idx < 127 ? 126 - idx : idx - 128; sgn = idx < 127 ? -1.f : 1.f; }
    else { if (idx == 0) return 0.f; if (idx == 255) return 1.f; pos = idx - 1; }
    const int base = is_signed ? 1 : 2;
    const int i = 31 - __clz(pos + base) - (is_signed ? 0 : 1);        // decade: its first slot sits at pos = nper - base
    const int nper = is_signed ? (1 << i) : (2 << i);
    const int j = pos + base - nper;
    float scale = 1e-6f;
    for (int d = 0; d < i; ++d) scale *= 10.f;
    return sgn * scale * (0.1f + 0.9f * ((float)j + 0.5f) / (float)nper);
}
// true when the staged map equals the dynamic map to 1e-5 relative (every thread of the 256-thread workgroup checks its own entry)
__device__ __forceinline__ bool is_dynamic_map(float mine, bool is_signed) {
    const float want = dyn_map_value((int)threadIdx.x, is_signed);
    return __syncthreads_and(fabsf(mine - want) <= 1e-5f * fabsf(want) + 1e-12f) != 0;
}
// stages both maps and the decade tables (256 threads); returns whether both are the library's dynamic maps.  Ends with the barriers of is_dynamic_map.
__device__ __forceinline__ bool stage_maps(Q8Lds& L, const float* __restrict__ qmap1, const float* __restrict__ qmap2) {
    const int t = threadIdx.x;
    const float m1 = qmap1[t], m2 = qmap2[t];
    L.q1[t + 1] = m1; L.q2[t + 1] = m2;
    if (t == 0) { L.q1[0] = -3e38f; L.q2[0] = -3e38f; L.q1[257] = 3e38f; L.q2[257] = 3e38f; }
    if (t < 8) {
        const int i = t < 7 ? t : 6;
        const float inv = i == 0 ? 1e6f : i == 1 ? 1e5f : i == 2 ? 1e4f : i == 3 ? 1e3f : i == 4 ? 1e2f : i == 5 ? 1e1f : 1e0f;
        const float lo = 0.1f / inv;
        L.dec1[t] = make_float4(lo, (float)(1 << i) * inv * (1.f / 0.9f), (float)((1 << i) - 1), __int_as_float((1 << i) - 1));
        L.dec2[t] = make_float4(lo, (float)(2 << i) * inv * (1.f / 0.9f), (float)((2 << i) - 1), __int_as_float((2 << i) - 2));
    }
    return is_dynamic_map(m1, true) & is_dynamic_map(m2, false);
}
// qp = the padded map (qp[k + 1] = map[k]), dec = its decade table
template <bool SIGNED>
__device__ __forceinline__ int direct_code(const float* qp, const float4* dec, float x) {
    const float a = fabsf(x);
    const int i = (int)__builtin_amdgcn_fmed3f(floorf(fmaf(__log2f(a), 0.30103f, 7.f)), 0.f, 6.f);      // decade [10^(i-7), 10^(i-6)): a = 0 -> -inf -> 0
    const float4 d = dec[i];
    const int j = (int)rintf(__builtin_amdgcn_fmed3f(fmaf(a - d.x, d.y, -0.5f), 0.f, d.z));
    const int pos = __float_as_int(d.w) + j;                  // slots_i - base + j
    const int k = SIGNED ? (x >= 0.f ? 128 + pos : 126 - pos) : 1 + pos;
    const float* c = qp + k;                                  // c[0 .. 2] = map[k - 1 .. k + 1] (sentinels past either end)
    const float dm = fabsf(c[0] - x), d0 = fabsf(c[1] - x), dp = fabsf(c[2] - x);
    int best = k - 1; float db = dm;
    if (d0 < db) { best = k; db = d0; }       // ties keep the lower index, like the search's `(q[hi] - x) < (x - q[lo]) ? hi : lo`
    if (dp < db) best = k + 1;
    return best;
}
// x = moment / absmax of its block (r = 1 / absmax): the exact quotient where the reciprocal would overflow
__device__ __forceinline__ float block_scaled(float x, float amax, float r) {
    if (__builtin_expect(amax < 1e-30f, 0)) return amax > 0.f ? x / amax : 0.f;
    return x * r;
}
// the two codes of one element from x1 = m / absmax1, x2 = v / absmax2
__device__ __forceinline__ void codes_of(const Q8Lds& L, bool dyn, float m, float x1, float x2, int& n1, int& n2) {
    if (dyn) {
        n1 = direct_code<true>(L.q1, L.dec1, x1);
        n2 = direct_code<false>(L.q2, L.dec2, x2);
        if (n1 == 127 && __float_as_int(m) < 0) n1 = 126;      // the first moment keeps its sign through quantisation (on this map only zero can lose it)
    } else {
        n1 = nearest_code(L.q1 + 1, x1);
        n2 = nearest_code(L.q2 + 1, x2);
        if (signbit(L.q1[n1 + 1]) != signbit(m)) n1 = m > 0.f ? min(n1 + 1, 255) : max(n1 - 1, 0);
    }
}
// step_size * m / (sqrt(v) + correction2 * eps) without the step size
template <typename T>
__device__ __forceinline__ float update_ratio(float m, float v, float c2eps) {
    if constexpr (sizeof(T) == 2) return m * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + c2eps);
    else return m / (sqrtf(v) + c2eps);
}

template <typename T>
__global__ void __launch_bounds__(256) adamw8bit_kernel(T* __restrict__ p, const T* __restrict__ g, uint8_t* __restrict__ c1, uint8_t* __restrict__ c2,
                                                        float* __restrict__ absmax1, float* __restrict__ absmax2, const float* __restrict__ qmap1,
                                                        const float* __restrict__ qmap2, T* __restrict__ shift, long n, float beta1, float beta2, float eps,
                                                        float lr, float weight_decay, float step_size, float correction2, float gnorm_scale) {
    __shared__ Q8Lds L;
    __shared__ float red[2][4];
    const bool dyn = stage_maps(L, qmap1, qmap2);      // (two barriers: the maps are staged)
    const float c2eps = correction2 * eps;
    const long nblocks = (n + 255) / 256;
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const long i = blk * 256 + threadIdx.x;
        const bool live = i < n;
        const float gv = live ? Elem<T>::to_f(g[i]) : 0.f;
        const bool fin = isfinite(gv);
        float m = 0.f, v = 0.f;
        if (live && fin) {
            const float gs = gv * gnorm_scale;
            m = L.q1[c1[i] + 1] * absmax1[blk]; v = L.q2[c2[i] + 1] * absmax2[blk];
            m = m * beta1 + (1.f - beta1) * gs;
            v = v * beta2 + (1.f - beta2) * gs * gs;
        }
        float a1 = fabsf(m), a2 = v;      // v >= 0
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a1 = fmaxf(a1, __shfl_xor(a1, o, 64)); a2 = fmaxf(a2, __shfl_xor(a2, o, 64)); }
        __syncthreads();                   // every thread has read absmax[blk] and the previous iteration's red[]
        if (lane == 0) { red[0][wid] = a1; red[1][wid] = a2; }
        __syncthreads();
        const float new1 = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
        const float new2 = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
        if (threadIdx.x == 0) { absmax1[blk] = new1; absmax2[blk] = new2; }
        if (!live) continue;
        T* target = shift ? shift : p;
        float t = Elem<T>::to_f(target[i]);
        if (fin) {
            t = Elem<T>::to_f(Elem<T>::from_f(t + step_size * update_ratio<T>(m, v, c2eps)));
            if (weight_decay > 0.f) t = Elem<T>::to_f(Elem<T>::from_f(t * (1.f - lr * weight_decay)));
        }
        if (shift) {
            const float pv = Elem<T>::to_f(p[i]);
            const float pn = Elem<T>::to_f(Elem<T>::from_f(pv + t));
            const float diff = Elem<T>::to_f(Elem<T>::from_f(pv - pn));
            p[i] = Elem<T>::from_f(pn);
            shift[i] = Elem<T>::from_f(t + diff);
        } else {
            p[i] = Elem<T>::from_f(t);
        }
        int k1, k2;
        codes_of(L, dyn, m, block_scaled(m, new1, 1.f / new1), block_scaled(v, new2, 1.f / new2), k1, k2);
        c1[i] = (uint8_t)k1; c2[i] = (uint8_t)k2;
    }
}


// The same update as adamw8bit_kernel for MANY tensors in one launch, 8 consecutive elements per thread: a 256-element quantisation block is 32 lanes x 8
// elements (16-byte parameter / gradient / shift accesses, 8-byte code accesses), its absmax a 5-step shuffle inside the half-wave -- no LDS round trip, no
// barrier in the loop.  One workgroup = one 2 048-element chunk (8 quantisation blocks) of one tensor; chunk table as in adamw_step_kernel.  Element-wise
// arithmetic identical to adamw8bit_kernel (same order, same roundings), so the two produce the same codes and parameters.
template <typename T>
__global__ void __launch_bounds__(256) adamw8bit_multi_kernel(void* const* __restrict__ p_ptrs, void* const* __restrict__ g_ptrs, void* const* __restrict__ c1_ptrs,
                                                              void* const* __restrict__ c2_ptrs, void* const* __restrict__ a1_ptrs, void* const* __restrict__ a2_ptrs,
                                                              void* const* __restrict__ s_ptrs, const long* __restrict__ sizes, const int* __restrict__ chunk_tensor,
                                                              const long* __restrict__ chunk_off, const float* __restrict__ qmap1, const float* __restrict__ qmap2,
                                                              float beta1, float beta2, float eps, float lr, float weight_decay, float step_size, float correction2,
                                                              float gnorm_scale, int nchunks) {
    __shared__ Q8Lds L;
    const bool dyn = stage_maps(L, qmap1, qmap2);      // (two barriers: the maps are staged)
    const float c2eps = correction2 * eps;
    constexpr int E = 8;
    // grid-stride over the chunk table (round 5): a workgroup stages the two maps once and walks ~nchunks / gridDim chunks -- with one chunk per workgroup (296 k workgroups
    // for the SDXL UNet) every 20 KiB of traffic paid a map load, two barriers and a cold start of its own
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int ti = chunk_tensor[c];
    const long off = chunk_off[c], n = sizes[ti];
    T* p = reinterpret_cast<T*>(p_ptrs[ti]);
    const T* g = reinterpret_cast<const T*>(g_ptrs[ti]);
    uint8_t* c1 = reinterpret_cast<uint8_t*>(c1_ptrs[ti]);
    uint8_t* c2 = reinterpret_cast<uint8_t*>(c2_ptrs[ti]);
    float* absmax1 = reinterpret_cast<float*>(a1_ptrs[ti]);
    float* absmax2 = reinterpret_cast<float*>(a2_ptrs[ti]);
    T* shift = s_ptrs ? reinterpret_cast<T*>(s_ptrs[ti]) : nullptr;
    const long i0 = off + (long)threadIdx.x * E;
    const bool any = i0 < n;                                  // lanes past the tensor's end stay in the half-wave shuffles with neutral values
    const long blk = (any ? i0 : n - 1) >> 8;
    const bool full = i0 + E <= n && (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(c1) & 7) == 0 && (reinterpret_cast<uintptr_t>(c2) & 7) == 0 && (!shift || (reinterpret_cast<uintptr_t>(shift) & 15) == 0);
    float gv[E], tv[E], pv[E];
    uint8_t k1[E], k2[E];
    bool live[E];
    // (round 6) the block's two absmax values are read ahead of the branchy load phase, and the bf16 full-vector path issues all of its reads (gradient, target,
    // parameter, both code words) back to back with nothing conditional in between: these are flat loads through table pointers, and a load behind a branch waits for
    // everything in flight (the pattern tools/isa_census.py found in the norm kernels)
    const float am1 = absmax1[blk], am2 = absmax2[blk];
    uint2 u1pre = make_uint2(0u, 0u), u2pre = make_uint2(0u, 0u);
    if (full) {
        if constexpr (sizeof(T) == 2) {
            Vec16<T> vg, vt, vp;
            vg.load(g + i0); vt.load((shift ? shift : p) + i0); vp.load(p + i0);          // (no Kahan buffer: the parameter vector twice, the second read an L1 hit)
            u1pre = *reinterpret_cast<const uint2*>(c1 + i0); u2pre = *reinterpret_cast<const uint2*>(c2 + i0);
            __builtin_amdgcn_sched_barrier(0);
            vg.unpack(gv); vt.unpack(tv); vp.unpack(pv);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                Vec16<T> vg; vg.load(g + i0 + 4 * h); vg.unpack(gv + 4 * h);
                Vec16<T> vt; vt.load((shift ? shift : p) + i0 + 4 * h); vt.unpack(tv + 4 * h);
                if (shift) { Vec16<T> vp; vp.load(p + i0 + 4 * h); vp.unpack(pv + 4 * h); }
            }
        }
        const uint2 u1 = sizeof(T) == 2 ? u1pre : *reinterpret_cast<const uint2*>(c1 + i0), u2 = sizeof(T) == 2 ? u2pre : *reinterpret_cast<const uint2*>(c2 + i0);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            k1[e] = (uint8_t)(((e < 4 ? u1.x : u1.y) >> (8 * (e & 3))) & 0xff);
            k2[e] = (uint8_t)(((e < 4 ? u2.x : u2.y) >> (8 * (e & 3))) & 0xff);
            live[e] = true;
        }
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            live[e] = any && i0 + e < n;
            const long i = live[e] ? i0 + e : n - 1;
            gv[e] = live[e] ? Elem<T>::to_f(g[i]) : 0.f;
            tv[e] = Elem<T>::to_f((shift ? shift : p)[i]);
            pv[e] = shift ? Elem<T>::to_f(p[i]) : 0.f;
            k1[e] = c1[i]; k2[e] = c2[i];
        }
    }
    const float wdf = weight_decay > 0.f ? 1.f - lr * weight_decay : 1.f;       // (x 1 and a second rounding of an already rounded value change nothing)
    float m[E], v[E];
    bool fin[E];
    float a1 = 0.f, a2 = 0.f;
    // no branch inside the element loops of the usual case (finite gradients, the library's maps): the eight elements' LDS reads and arithmetic interleave
#pragma unroll
    for (int e = 0; e < E; ++e) {
        fin[e] = isfinite(gv[e]);
        const float gs = gv[e] * gnorm_scale;
        const float mq = L.q1[k1[e] + 1] * am1, vq = L.q2[k2[e] + 1] * am2;
        const float mn = mq * beta1 + (1.f - beta1) * gs;
        const float vn = vq * beta2 + (1.f - beta2) * gs * gs;
        const bool ok = live[e] && fin[e];
        m[e] = ok ? mn : 0.f; v[e] = ok ? vn : 0.f;
        a1 = fmaxf(a1, fabsf(m[e])); a2 = fmaxf(a2, v[e]);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { a1 = fmaxf(a1, __shfl_xor(a1, o, 64)); a2 = fmaxf(a2, __shfl_xor(a2, o, 64)); }   // the 32 lanes of one quantisation block
    if ((threadIdx.x & 31) == 0 && any) { absmax1[blk] = a1; absmax2[blk] = a2; }
    float outp[E], outs[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        float tn = Elem<T>::to_f(Elem<T>::from_f(tv[e] + step_size * update_ratio<T>(m[e], v[e], c2eps)));
        tn = Elem<T>::to_f(Elem<T>::from_f(tn * wdf));
        const float t = fin[e] ? tn : tv[e];
        if (shift) {
            const float pn = Elem<T>::to_f(Elem<T>::from_f(pv[e] + t));
            const float diff = Elem<T>::to_f(Elem<T>::from_f(pv[e] - pn));
            outp[e] = pn; outs[e] = t + diff;
        } else {
            outp[e] = t; outs[e] = 0.f;
        }
    }
    const float r1 = 1.f / a1, r2 = 1.f / a2;
    float x1[E], x2[E];
    if (__builtin_expect((a1 < 1e-30f) | (a2 < 1e-30f), 0)) {
#pragma unroll
        for (int e = 0; e < E; ++e) { x1[e] = block_scaled(m[e], a1, r1); x2[e] = block_scaled(v[e], a2, r2); }
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) { x1[e] = m[e] * r1; x2[e] = v[e] * r2; }
    }
    if (dyn) {
#pragma unroll
        for (int e = 0; e < E; ++e) { int n1, n2; codes_of(L, true, m[e], x1[e], x2[e], n1, n2); k1[e] = (uint8_t)n1; k2[e] = (uint8_t)n2; }
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) { int n1, n2; codes_of(L, false, m[e], x1[e], x2[e], n1, n2); k1[e] = (uint8_t)n1; k2[e] = (uint8_t)n2; }
    }
    if (full) {
        if constexpr (sizeof(T) == 2) {
            Vec16<T> vo; vo.pack(outp); vo.store(p + i0);
            if (shift) { Vec16<T> vs; vs.pack(outs); vs.store(shift + i0); }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                Vec16<T> vo; vo.pack(outp + 4 * h); vo.store(p + i0 + 4 * h);
                if (shift) { Vec16<T> vs; vs.pack(outs + 4 * h); vs.store(shift + i0 + 4 * h); }
            }
        }
        uint2 u1, u2;
        u1.x = k1[0] | (k1[1] << 8) | (k1[2] << 16) | ((unsigned)k1[3] << 24); u1.y = k1[4] | (k1[5] << 8) | (k1[6] << 16) | ((unsigned)k1[7] << 24);
        u2.x = k2[0] | (k2[1] << 8) | (k2[2] << 16) | ((unsigned)k2[3] << 24); u2.y = k2[4] | (k2[5] << 8) | (k2[6] << 16) | ((unsigned)k2[7] << 24);
        *reinterpret_cast<uint2*>(c1 + i0) = u1; *reinterpret_cast<uint2*>(c2 + i0) = u2;
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (!live[e]) continue;
            p[i0 + e] = Elem<T>::from_f(outp[e]);
            if (shift) shift[i0 + e] = Elem<T>::from_f(outs[e]);
            c1[i0 + e] = k1[e]; c2[i0 + e] = k2[e];
        }
    }
    }
}

}  // namespace

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

extern "C" {

int dpipe_adamw_sumsq(const void* const* g_ptrs, int lanes, const int* chunk_tensor, const long* chunk_off, const int* chunk_len, int nchunks,
                      int dtype, float* partials, float* out_sumsq, int accumulate, void* stream) {
    if (!partials || !out_sumsq || nchunks < 0 || lanes < 1 || lanes > MAX_LANES) { set_last_error("dpipe_adamw_sumsq: bad argument"); return DPIPE_ERR_ARG; }
    if (ablated(ABL_STEP)) return DPIPE_OK;                                        // (debug switch: runtime.hip)
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
    if (nchunks <= 0 || ablated(ABL_STEP)) return DPIPE_OK;
    if (!p_ptrs || !m_ptrs || !v_ptrs || !g_ptrs || !chunk_tensor || !chunk_off || !chunk_len || lanes < 1 || lanes > MAX_LANES ||
        bias_correction1 <= 0.f || bias_correction2 <= 0.f) {
        set_last_error("dpipe_adamw_step: bad argument"); return DPIPE_ERR_ARG;
    }
    AdamHyper h{lr, beta1, beta2, eps, weight_decay, bias_correction1, sqrtf(bias_correction2), max_norm};
    hipStream_t s = STREAM(stream);
    // DPIPE_ADAMW_STATIC_LANES=1: the static-lane form.  OFF: measured level with the runtime loop -- 343.8 / 343.2 vs 344.6 ms per step in the bench (r6q), and in isolation
    // (tools/optim_timing.py, r6x, two pairs) 2.65 / 2.39 vs 2.34 / 2.39 ms at one lane, 3.49 / 3.93 vs 3.41 / 3.47 at three: the step end already streams at 5.2 - 5.3 TB/s with
    // three or more lanes' reads per group, more reads in flight per thread buy nothing.
    static const bool static_lanes = [] { const char* e = getenv("DPIPE_ADAMW_STATIC_LANES"); return e && atoi(e) != 0; }();
#define ADAMW_STEP(TT, LL) adamw_step_kernel<TT, LL><<<nchunks, OPT_BLOCK, 0, s>>>(p_ptrs, m_ptrs, v_ptrs, s_ptrs, g_ptrs, lanes, chunk_tensor, chunk_off, chunk_len, total_sumsq, h, zero_grads)
#define ADAMW_STEP_L(TT) do { switch (static_lanes ? lanes : 0) { case 1: ADAMW_STEP(TT, 1); break; case 2: ADAMW_STEP(TT, 2); break; case 3: ADAMW_STEP(TT, 3); break; \
    case 4: ADAMW_STEP(TT, 4); break; case 6: ADAMW_STEP(TT, 6); break; case 8: ADAMW_STEP(TT, 8); break; default: ADAMW_STEP(TT, 0); } } while (0)
    if (dtype == DPIPE_BF16) ADAMW_STEP_L(bf16_t);
    else if (dtype == DPIPE_F32) ADAMW_STEP_L(float);
    else { set_last_error("dpipe_adamw_step: dtype"); return DPIPE_ERR_UNSUPPORTED; }
#undef ADAMW_STEP_L
#undef ADAMW_STEP
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

int dpipe_adamw8bit_step(void* p, const void* g, void* state1, void* state2, float* absmax1, float* absmax2, const float* qmap1, const float* qmap2,
                         void* shift, long n, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float gnorm_scale, int dtype,
                         void* stream) {
    if (!p || !g || !state1 || !state2 || !absmax1 || !absmax2 || !qmap1 || !qmap2 || n <= 0 || step < 1) { set_last_error("dpipe_adamw8bit_step: bad argument"); return DPIPE_ERR_ARG; }
    const float correction1 = 1.f - powf(beta1, (float)step), correction2 = sqrtf(1.f - powf(beta2, (float)step));
    const float step_size = -lr * correction2 / correction1;
    const long nblocks = (n + 255) / 256;
    const unsigned grid = (unsigned)(nblocks < 4096 ? nblocks : 4096);
    hipStream_t s = STREAM(stream);
    if (dtype == DPIPE_BF16)
        adamw8bit_kernel<bf16_t><<<grid, 256, 0, s>>>((bf16_t*)p, (const bf16_t*)g, (uint8_t*)state1, (uint8_t*)state2, absmax1, absmax2, qmap1, qmap2, (bf16_t*)shift, n,
                                                      beta1, beta2, eps, lr, weight_decay, step_size, correction2, gnorm_scale);
    else if (dtype == DPIPE_F32)
        adamw8bit_kernel<float><<<grid, 256, 0, s>>>((float*)p, (const float*)g, (uint8_t*)state1, (uint8_t*)state2, absmax1, absmax2, qmap1, qmap2, (float*)shift, n,
                                                     beta1, beta2, eps, lr, weight_decay, step_size, correction2, gnorm_scale);
    else { set_last_error("dpipe_adamw8bit_step: dtype"); return DPIPE_ERR_UNSUPPORTED; }
    return check_launch("dpipe_adamw8bit_step");
}

int dpipe_adamw8bit_multi(void* const* p_ptrs, void* const* g_ptrs, void* const* state1_ptrs, void* const* state2_ptrs, void* const* absmax1_ptrs,
                          void* const* absmax2_ptrs, void* const* shift_ptrs, const long* sizes, const int* chunk_tensor, const long* chunk_off, int nchunks,
                          const float* qmap1, const float* qmap2, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float gnorm_scale,
                          int dtype, void* stream) {
    if (nchunks <= 0) return DPIPE_OK;
    if (!p_ptrs || !g_ptrs || !state1_ptrs || !state2_ptrs || !absmax1_ptrs || !absmax2_ptrs || !sizes || !chunk_tensor || !chunk_off || !qmap1 || !qmap2 || step < 1) {
        set_last_error("dpipe_adamw8bit_multi: bad argument"); return DPIPE_ERR_ARG;
    }
    const float correction1 = 1.f - powf(beta1, (float)step), correction2 = sqrtf(1.f - powf(beta2, (float)step));
    const float step_size = -lr * correction2 / correction1;
    hipStream_t s = STREAM(stream);
    const int grid = nchunks < 8192 ? nchunks : 8192;          // 256 CUs x 7 resident workgroups x ~4.5: every workgroup walks a few dozen chunks at SDXL size
    if (dtype == DPIPE_BF16)
        adamw8bit_multi_kernel<bf16_t><<<grid, 256, 0, s>>>(p_ptrs, g_ptrs, state1_ptrs, state2_ptrs, absmax1_ptrs, absmax2_ptrs, shift_ptrs, sizes, chunk_tensor, chunk_off,
                                                            qmap1, qmap2, beta1, beta2, eps, lr, weight_decay, step_size, correction2, gnorm_scale, nchunks);
    else if (dtype == DPIPE_F32)
        adamw8bit_multi_kernel<float><<<grid, 256, 0, s>>>(p_ptrs, g_ptrs, state1_ptrs, state2_ptrs, absmax1_ptrs, absmax2_ptrs, shift_ptrs, sizes, chunk_tensor, chunk_off,
                                                           qmap1, qmap2, beta1, beta2, eps, lr, weight_decay, step_size, correction2, gnorm_scale, nchunks);
    else { set_last_error("dpipe_adamw8bit_multi: dtype"); return DPIPE_ERR_UNSUPPORTED; }
    return check_launch("dpipe_adamw8bit_multi");
}

}  // extern "C"
