// dpipe_common.h -- shared device helpers for the gfx950 (CDNA4, wave64) kernels.
// Everything here is written for MI355X only: 64-lane wavefronts, 16-byte
// vector global accesses, fp32 accumulation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DPIPE_BF16 0
#define DPIPE_F32 1

#define DPIPE_OK 0
#define DPIPE_ERR_ARG -1
#define DPIPE_ERR_UNSUPPORTED -2

namespace dpipe {

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment, 4 VGPRs
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_mfma;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;   // 16-byte payload of a raw buffer store
typedef __attribute__((ext_vector_type(16))) float f32x16;

void set_last_error(const char* msg);
int option(int id, int dflt);          // runtime.hip: dpipe_set_option value, else the option's environment variable, else dflt
int check_launch(const char* what);
// debug ablation switches (runtime.hip): classes 1 GEMM, 2 convolution, 4 attention, 8 LayerNorm / RMSNorm, 16 GroupNorm, 32 element-wise, 64 step end
enum { ABL_GEMM = 1, ABL_CONV = 2, ABL_ATTN = 4, ABL_LN = 8, ABL_GN = 16, ABL_EW = 32, ABL_STEP = 64 };
bool ablated(int cls);
int ablate_gemm_kdiv();

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    return __uint_as_float(((uint32_t)v) << 16);
}
// fp32 -> bf16, round-to-nearest-even, NaN preserved (torch's cast).  Written as a native conversion so hipcc emits the
// gfx950 hardware instruction (one v_cvt_pk_bf16_f32 per PAIR) instead of ~5 integer VALU operations per value: the
// attention kernels are VALU-issue bound (SQ_ACTIVE_INST_ANY 58 % of wave cycles, profiles/), half of it conversions.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_native;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2_native v;
    v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu);
}

// Phi(x) = 0.5 (1 + erf(x / sqrt 2)) of the exact ("erf") GELU and E = exp(-x^2 / 2), branch-free (round 6): Abramowitz-Stegun 7.1.26 with the complementary
// form on the negative side (no cancellation), one v_exp_f32 and one v_rcp_f32 -- |Phi error| <= 3.0e-7, |gelu error| <= 4.3e-7, |gelu' error| <= 3.2e-7 over
// [-12, 12] in fp32 (tests/test_optim_cpu.py::test_gelu_erf_approximation; torch's own fp32 erf GELU is 1.2e-6 off the fp64 value).  ocml's erff costs ~3 x the
// instructions and a divergent branch per element: the GEGLU kernels were VALU-bound on it next to 50 MB of traffic (165 branches in geglu_bwd's ISA).
__device__ __forceinline__ float gelu_erf_cdf(float x, float& E) {
    const float z = fabsf(x) * 0.7071067811865476f;
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * z);
    E = __expf(-0.5f * x * x);
    const float P = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float half = 0.5f * P * E;
    return x >= 0.f ? 1.f - half : half;
}

// Element traits so that HBM-bound kernels can be instantiated for bf16 and fp32.
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int VEC = 4;  // elements per 16-byte access
    __device__ static __forceinline__ float to_f(float v) { return v; }
    __device__ static __forceinline__ float from_f(float v) { return v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float to_f(bf16_t v) { return bf16_to_f32(v); }
    __device__ static __forceinline__ bf16_t from_f(float v) { return f32_to_bf16(v); }
};

// 16-byte vector of T: load, convert to fp32 lanes, store back.
template <typename T> struct Vec16 {
    static constexpr int N = Elem<T>::VEC;
    uint4 raw;
    __device__ __forceinline__ void load(const T* p) { raw = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void store(T* p) const { *reinterpret_cast<uint4*>(p) = raw; }
    __device__ __forceinline__ void unpack(float* f) const;
    __device__ __forceinline__ void pack(const float* f);
};
template <> __device__ __forceinline__ void Vec16<float>::unpack(float* f) const {
    f[0] = __uint_as_float(raw.x); f[1] = __uint_as_float(raw.y);
    f[2] = __uint_as_float(raw.z); f[3] = __uint_as_float(raw.w);
}
template <> __device__ __forceinline__ void Vec16<float>::pack(const float* f) {
    raw.x = __float_as_uint(f[0]); raw.y = __float_as_uint(f[1]);
    raw.z = __float_as_uint(f[2]); raw.w = __float_as_uint(f[3]);
}
template <> __device__ __forceinline__ void Vec16<bf16_t>::unpack(float* f) const {
    f[0] = __uint_as_float(raw.x << 16); f[1] = __uint_as_float(raw.x & 0xffff0000u);
    f[2] = __uint_as_float(raw.y << 16); f[3] = __uint_as_float(raw.y & 0xffff0000u);
    f[4] = __uint_as_float(raw.z << 16); f[5] = __uint_as_float(raw.z & 0xffff0000u);
    f[6] = __uint_as_float(raw.w << 16); f[7] = __uint_as_float(raw.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void Vec16<bf16_t>::pack(const float* f) {
    raw.x = pack_bf16x2(f[0], f[1]); raw.y = pack_bf16x2(f[2], f[3]);
    raw.z = pack_bf16x2(f[4], f[5]); raw.w = pack_bf16x2(f[6], f[7]);
}

// V consecutive values of a per-column operand (norm weights, modulation rows, biases) as fp32, read by 16-byte loads and BRANCH-FREE: an absent operand
// (null pointer) reads a zero pad and the caller selects the identity value per element.  Why this exists (round 6): `w ? to_f(w[c + j]) : 1.f` per element makes
// hipcc branch around every 2-byte load and wait `vmcnt(0)` behind each one -- a chain of V dependent round trips per vector (cdna_hip_programming.md section 5, trap (c)).
// `p + idx` must be 16-byte aligned (the launchers check the base pointers; idx is a multiple of V).
static __device__ uint4 g_param_pad[2];   // zero-initialised; NOT const: a constant-address-space object would turn the pointer select in PVec::load into flat loads
template <typename W, int V> struct PVec {
    static_assert(V * sizeof(W) == 16 || V * sizeof(W) == 32, "PVec: 8 bf16, 4 fp32 or 8 fp32 values");
    static constexpr int NR = V * (int)sizeof(W) / 16;
    uint4 raw[NR];
    __device__ __forceinline__ void load(const W* p, long idx) {
        const uint4* q = p ? reinterpret_cast<const uint4*>(p + idx) : g_param_pad;
#pragma unroll
        for (int r = 0; r < NR; ++r) raw[r] = q[r];
    }
    __device__ __forceinline__ void unpack(float* f) const {
        if constexpr (sizeof(W) == 4) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                f[4 * r + 0] = __uint_as_float(raw[r].x); f[4 * r + 1] = __uint_as_float(raw[r].y);
                f[4 * r + 2] = __uint_as_float(raw[r].z); f[4 * r + 3] = __uint_as_float(raw[r].w);
            }
        } else {
            f[0] = __uint_as_float(raw[0].x << 16); f[1] = __uint_as_float(raw[0].x & 0xffff0000u);
            f[2] = __uint_as_float(raw[0].y << 16); f[3] = __uint_as_float(raw[0].y & 0xffff0000u);
            f[4] = __uint_as_float(raw[0].z << 16); f[5] = __uint_as_float(raw[0].z & 0xffff0000u);
            f[6] = __uint_as_float(raw[0].w << 16); f[7] = __uint_as_float(raw[0].w & 0xffff0000u);
        }
    }
};
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- wave64 / block reductions -------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// Block-wide sum; every thread receives the result. `smem` must hold >= 16 floats.
// blockDim.x must be a multiple of 64 and <= 1024.
__device__ __forceinline__ float block_sum(float v, float* smem) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_sum(v);
    __syncthreads();  // protect smem reuse between consecutive calls
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += smem[i];  // fixed order: deterministic
    return r;
}
__device__ __forceinline__ float block_max(float v, float* smem) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    float r = smem[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, smem[i]);
    return r;
}

__host__ __device__ __forceinline__ long cdiv(long a, long b) { return (a + b - 1) / b; }

// Grid size for HBM-bound grid-stride kernels: enough blocks to cover the 256 CUs
// eight times over (guide: cap ~2048 blocks and stride the rest).
static inline int stream_grid(long work_items, int block) {
    long g = cdiv(work_items, block);
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (int)g;
}

}  // namespace dpipe
