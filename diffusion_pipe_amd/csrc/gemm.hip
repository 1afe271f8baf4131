// gemm.hip -- K1/K6 dense contractions on the CDNA4 matrix cores (gfx950).
//
//   C[b] = act(alpha * op(A[b]) . op(B[b]) + bias) (+ C[b] when accumulate)
//
// op(A) is M x K, op(B) is K x N.  transA = 0: A stored [M][K] (K contiguous);  transA = 1: A stored [K][M].
//                                  transB = 1: B stored [N][K] (K contiguous);  transB = 0: B stored [K][N].
// This covers the three contractions of a Linear layer (reference call sites: models/wan/model.py:120-122,
// 138-142,270-272; diffusers attention/FF projections of models/sdxl.py:797-865):
//   forward  y  = x . W^T      (transA=0, transB=1)
//   dgrad    dx = dy . W       (transA=0, transB=0)
//   wgrad    dW = dy^T . x     (transA=1, transB=0)
//
// Design (MI355X): 256 threads = 4 wavefronts in a 2x2 grid, each wave owns a (BM/2)x(BN/2) sub-tile built from
// 32x32 MFMA tiles (v_mfma_f32_32x32x16_bf16, fp32 accumulate; v_mfma_f32_32x32x2_f32 for the exact-fp32 parity
// path).  Operand tiles are staged global -> registers -> LDS with 16-byte accesses; the next K-tile's global loads
// are issued before the current tile's MFMAs (issue-early / write-late), so HBM latency hides under the matrix
// pipe.  K-contiguous operands sit in LDS as [mn][k] rows padded by 16 B (conflict-free ds_read_b128); operands
// that are MN-contiguous in memory are stored as [k][mn] rows padded by 64 B and consumed with the LDS
// transpose-read ds_read_b64_tr_b16, so no transposed copy of dy / x / W is ever materialised in HBM.
// Workgroup ids are remapped so that each XCD (private 4 MiB L2) works on a contiguous band of tiles.
#include "gemm_internal.h"
#include "../../include/dpipe_hip.h"

using namespace dpipe;

namespace {

typedef __attribute__((address_space(3))) bf16x4_t lds_bf16x4_t;

template <typename T> struct GemmCfg;
template <> struct GemmCfg<bf16_t> { static constexpr int BK = 64; static constexpr int VEC = 8; };
template <> struct GemmCfg<float> { static constexpr int BK = 32; static constexpr int VEC = 4; };

// Tile of `ROWS` MN-rows.  TRANS=false: LDS image [ROWS][BK] (+16 B pad).  TRANS=true: [BK][ROWS] (+64 B pad).
template <typename T, int ROWS, bool TRANS> struct TileGeom {
    static constexpr int BK = GemmCfg<T>::BK;
    static constexpr int VEC = GemmCfg<T>::VEC;
    static constexpr int ROW_BYTES = TRANS ? ROWS * (int)sizeof(T) + 64 : BK * (int)sizeof(T) + 16;
    static constexpr int NROWS = TRANS ? BK : ROWS;
    static constexpr int BYTES = ROW_BYTES * NROWS;
    static constexpr int VEC_PER_ROW = TRANS ? ROWS / VEC : BK / VEC;
    static constexpr int NVEC = NROWS * VEC_PER_ROW;
    static constexpr int PER_THREAD = NVEC / 256;
    static_assert(NVEC % 256 == 0, "tile must split evenly over 256 threads");
};

// Load this thread's share of an operand tile into registers (zero-filled outside the matrix).
//   base: operand base for this batch;  mn0: first MN index of the tile;  k0: first K index of the tile
template <int N> struct TileRegs { uint4 v[N]; };

template <typename T, int ROWS, bool TRANS>
__device__ __forceinline__ void load_tile(TileRegs<TileGeom<T, ROWS, TRANS>::PER_THREAD>& regs, const T* __restrict__ base,
                                          long ld, int mn0, int k0, int MN, int K, bool vec_ok) {
    using G = TileGeom<T, ROWS, TRANS>;
    constexpr int V = G::VEC;
#pragma unroll
    for (int i = 0; i < G::PER_THREAD; ++i) {
        const int v = threadIdx.x + i * 256;
        const int r = v / G::VEC_PER_ROW, c = (v % G::VEC_PER_ROW) * V;
        // TRANS: r indexes k, c indexes mn.  else: r indexes mn, c indexes k.
        const int mn = mn0 + (TRANS ? c : r), k = k0 + (TRANS ? r : c);
        const long row = TRANS ? k : mn, col = TRANS ? mn : k;
        const int row_lim = TRANS ? K : MN, col_lim = TRANS ? MN : K;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (row < row_lim) {
            if (vec_ok && col + V <= col_lim) {
                val = *reinterpret_cast<const uint4*>(base + row * ld + col);
            } else if (col < col_lim) {
                alignas(16) T tmp[V];
#pragma unroll
                for (int j = 0; j < V; ++j) tmp[j] = (col + j < col_lim) ? base[row * ld + col + j] : (T)0;
                val = *reinterpret_cast<uint4*>(tmp);
            }
        }
        regs.v[i] = val;
    }
}
template <typename T, int ROWS, bool TRANS>
__device__ __forceinline__ void store_tile(const TileRegs<TileGeom<T, ROWS, TRANS>::PER_THREAD>& regs, char* lds) {
    using G = TileGeom<T, ROWS, TRANS>;
#pragma unroll
    for (int i = 0; i < G::PER_THREAD; ++i) {
        const int v = threadIdx.x + i * 256;
        const int r = v / G::VEC_PER_ROW, c = v % G::VEC_PER_ROW;
        *reinterpret_cast<uint4*>(lds + r * G::ROW_BYTES + c * 16) = regs.v[i];
    }
}

// bf16 fragment for one 32x32x16 MFMA: lane (i = lane & 31, h = lane >> 5) gets k = kk + 8h .. +8 of MN-row (mn + i).
template <int ROWS, bool TRANS>
__device__ __forceinline__ bf16x8_t read_frag_bf16(const char* lds, int mn, int kk, int lane) {
    using G = TileGeom<bf16_t, ROWS, TRANS>;
    if (!TRANS) {
        return *reinterpret_cast<const bf16x8_t*>(lds + (mn + (lane & 31)) * G::ROW_BYTES + (kk + 8 * (lane >> 5)) * 2);
    } else {
        // ds_read_b64_tr_b16: within a 16-lane group each lane supplies the address of 4 contiguous bf16 of one
        // k-row (lane t: row t>>2, columns 4*(t&3)..+3 of a [4][16] block) and receives column t of that block.
        const int t = lane & 15, g = lane >> 4;
        const int krow = kk + 8 * (g >> 1) + (t >> 2);
        const int col = mn + 16 * (g & 1) + 4 * (t & 3);
        const char* p = lds + krow * G::ROW_BYTES + col * 2;
        bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(p));
        bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(p + 4 * G::ROW_BYTES));
        bf16x8_t out;
        out[0] = lo[0]; out[1] = lo[1]; out[2] = lo[2]; out[3] = lo[3];
        out[4] = hi[0]; out[5] = hi[1]; out[6] = hi[2]; out[7] = hi[3];
        return out;
    }
}
// fp32 operand for one 32x32x2 MFMA: lane (i, h) gets element (mn + i, kk + h).
template <int ROWS, bool TRANS>
__device__ __forceinline__ float read_frag_f32(const char* lds, int mn, int kk, int lane) {
    using G = TileGeom<float, ROWS, TRANS>;
    const int i = lane & 31, h = lane >> 5;
    if (!TRANS) return *reinterpret_cast<const float*>(lds + (mn + i) * G::ROW_BYTES + (kk + h) * 4);
    return *reinterpret_cast<const float*>(lds + (kk + h) * G::ROW_BYTES + (mn + i) * 4);
}

template <typename T, int BM, int BN, bool TA, bool TB>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmParams p) {
    using GA = TileGeom<T, BM, TA>;
    using GB = TileGeom<T, BN, !TB>;   // B is "transposed in memory" when stored [K][N] (transB = 0)
    constexpr int BK = GemmCfg<T>::BK;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    __shared__ __attribute__((aligned(16))) char lds[GA::BYTES + GB::BYTES];
    char* ldsA = lds; char* ldsB = lds + GA::BYTES;

    // XCD-aware, bijective remap of the workgroup id: consecutive ids (round-robin over the 8 XCDs) are
    // regrouped so that each XCD owns a contiguous run of tiles; runs walk tiles_m fastest so that the B panel
    // of a tile column stays hot in that XCD's L2.
    const int nwg = p.tiles_m * p.tiles_n;
    const int orig = blockIdx.x;
    const int q = nwg / 8, r = nwg % 8, xcd = orig % 8;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
    const int tile_m = wg % p.tiles_m, tile_n = wg / p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int z = blockIdx.y;
    const long zo = z / p.batch_inner, zi = z % p.batch_inner;
    const T* A = reinterpret_cast<const T*>(p.A) + zo * p.sAo + zi * p.sAi;
    const T* B = reinterpret_cast<const T*>(p.B) + zo * p.sBo + zi * p.sBi;

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm0 = (wid >> 1) * WM, wn0 = (wid & 1) * WN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    TileRegs<GA::PER_THREAD> ra; TileRegs<GB::PER_THREAD> rb;
    const int nk = (p.K + BK - 1) / BK;
    load_tile<T, BM, TA>(ra, A, p.lda, m0, 0, p.M, p.K, p.vecA);
    load_tile<T, BN, !TB>(rb, B, p.ldb, n0, 0, p.N, p.K, p.vecB);

    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                       // every wave finished reading the previous tile
        store_tile<T, BM, TA>(ra, ldsA);
        store_tile<T, BN, !TB>(rb, ldsB);
        __syncthreads();
        if (kt + 1 < nk) {                     // issue next tile's HBM loads; they land while the MFMAs run
            load_tile<T, BM, TA>(ra, A, p.lda, m0, (kt + 1) * BK, p.M, p.K, p.vecA);
            load_tile<T, BN, !TB>(rb, B, p.ldb, n0, (kt + 1) * BK, p.N, p.K, p.vecB);
        }
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 16) {
                bf16x8_t fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = read_frag_bf16<BM, TA>(ldsA, wm0 + i * 32, kk, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = read_frag_bf16<BN, !TB>(ldsB, wn0 + j * 32, kk, lane);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8_mfma, fa[i]), __builtin_bit_cast(bf16x8_mfma, fb[j]), acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll 4
            for (int kk = 0; kk < BK; kk += 2) {
                float fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = read_frag_f32<BM, TA>(ldsA, wm0 + i * 32, kk, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = read_frag_f32<BN, !TB>(ldsB, wn0 + j * 32, kk, lane);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    // Epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5).
    const long coff = zo * p.sCo + zi * p.sCi;
    const int col_in = lane & 31, row_in = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + j * 32 + col_in;
            if (n >= p.N) continue;
            float bv = 0.f;
            if (p.bias) bv = Elem<T>::to_f(reinterpret_cast<const T*>(p.bias)[n]);   // bias has the operand dtype
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + row_in;
                if (m >= p.M) continue;
                float v = epilogue_act(p.alpha * acc[i][j][e] + bv, p.act);
                const long idx = coff + (long)m * p.ldc + n;
                if (p.residual) {
                    const long ridx = coff + (long)m * p.ldr + n;
                    v += (p.out_f32 || sizeof(T) == 4) ? reinterpret_cast<const float*>(p.residual)[ridx] : bf16_to_f32(reinterpret_cast<const bf16_t*>(p.residual)[ridx]);
                }
                if (p.out_f32 || sizeof(T) == 4) {
                    float* c = reinterpret_cast<float*>(p.C);
                    if (p.accumulate) v += c[idx];
                    c[idx] = v;
                } else {
                    bf16_t* c = reinterpret_cast<bf16_t*>(p.C);
                    if (p.accumulate) v += bf16_to_f32(c[idx]);
                    c[idx] = f32_to_bf16(v);
                }
            }
        }
}

template <typename T, int BM, int BN>
int launch_cfg(GemmParams& p, int transA, int transB, int batch, hipStream_t s) {
    p.tiles_m = (p.M + BM - 1) / BM; p.tiles_n = (p.N + BN - 1) / BN;
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)batch);
    if (!transA && transB) gemm_kernel<T, BM, BN, false, true><<<grid, 256, 0, s>>>(p);
    else if (!transA && !transB) gemm_kernel<T, BM, BN, false, false><<<grid, 256, 0, s>>>(p);
    else if (transA && !transB) gemm_kernel<T, BM, BN, true, false><<<grid, 256, 0, s>>>(p);
    else gemm_kernel<T, BM, BN, true, true><<<grid, 256, 0, s>>>(p);
    return check_launch("dpipe_gemm");
}

// ---- micro-probe used by the GPU tests to pin the ds_read_b64_tr_b16 semantics this file relies on -------------
__global__ void tr16_probe_kernel(const short* __restrict__ in, short* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) short tile[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 64) tile[i] = in[i];
    __syncthreads();
    bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(&tile[threadIdx.x * 4]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}

}  // namespace

extern "C" {

int dpipe_gemm_ex(int dtype, int transA, int transB, int M, int N, int K,
                  const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                  int batch_outer, int batch_inner,
                  long strideA_outer, long strideA_inner, long strideB_outer, long strideB_inner,
                  long strideC_outer, long strideC_inner,
                  const void* bias, int act, float alpha, int accumulate, int out_f32, int tile_hint,
                  void* splitk_ws, long splitk_ws_bytes, const void* residual, long ldr, void* colsum, int colsum_accumulate,
                  void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || batch_outer <= 0 || batch_inner <= 0) { set_last_error("dpipe_gemm: bad argument"); return DPIPE_ERR_ARG; }
    if (dtype != DPIPE_BF16 && dtype != DPIPE_F32) { set_last_error("dpipe_gemm: dtype"); return DPIPE_ERR_UNSUPPORTED; }
    if (ablated(ABL_GEMM)) return DPIPE_OK;                                        // (debug switch: runtime.hip)
    if (ablate_gemm_kdiv() > 1 && K >= 128) { K = (K / ablate_gemm_kdiv() + 63) / 64 * 64; }
    const int V = dtype == DPIPE_BF16 ? 8 : 4;
    GemmParams p;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.sAo = strideA_outer; p.sAi = strideA_inner; p.sBo = strideB_outer; p.sBi = strideB_inner; p.sCo = strideC_outer; p.sCi = strideC_inner;
    p.batch_inner = batch_inner; p.alpha = alpha; p.act = act; p.accumulate = accumulate; p.out_f32 = out_f32;
    p.splitk = 1; p.ksteps = 0; p.ksteps_per_split = 0; p.slabs = nullptr; p.counters = nullptr;
    p.residual = residual; p.ldr = ldr; p.colsum = colsum; p.colsum_acc = colsum_accumulate; p.bias_rows = 0; p.bias_lo = 0;
    if (residual && residual == C && !accumulate) { set_last_error("dpipe_gemm: residual may not alias C"); return DPIPE_ERR_ARG; }
    const int batch = batch_outer * batch_inner;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // tile_hint: 0 = auto (pipelined bf16 kernel when eligible, else generic); 64 / 128 = generic kernel with that tile;
    // 1000 + S = pipelined kernel, S K-slices forced (S = 0: its own choice), 2000 + S / 3000 + S = the same with the
    // 64 x 64 / 128 x 128 tile forced -- an error when the problem is not eligible.
    if (dtype == DPIPE_BF16 && (tile_hint == 0 || tile_hint >= 1000)) {
        int rc = 0;
        // tile_hint >= 1000: the pipelined kernel, hint % 1000 = forced split-K (0 / 1000: automatic).  2000: 64^2, 3000: 128^2 3-deep ring, 4000: 128^2 2-deep ring,
        // 7000: 256^2, 9000: 256^2 on half K-steps, 12000: 128^2 register-staged.  (5000 / 6000 / 8000 / 10000 / 11000 named tiles that were removed in round 5: refused.)
        const int th = tile_hint / 1000;
        if (th == 5 || th == 6 || th == 8 || th == 10 || th == 11 || th > 12) { set_last_error("dpipe_gemm: tile_hint names a tile configuration that no longer exists"); return DPIPE_ERR_UNSUPPORTED; }
        const int force_tile = th == 12 ? 132 : th == 9 ? 258 : th == 7 ? 257 : th == 4 ? 129 : th == 3 ? 128 : th == 2 ? 64 : 0;
        const int force_s = tile_hint >= 1000 ? tile_hint % 1000 : 0;
        if (gemm_pipe_try(p, transA, transB, batch, splitk_ws, splitk_ws_bytes, force_s, force_tile, s, &rc)) return rc;
        if (tile_hint >= 1000) { set_last_error("dpipe_gemm: problem not eligible for the pipelined kernel"); return DPIPE_ERR_UNSUPPORTED; }
    }
    if (tile_hint >= 1000) { set_last_error("dpipe_gemm: pipelined kernel is bf16 only"); return DPIPE_ERR_UNSUPPORTED; }
    if (act & ACT_GEGLU_BWD) { set_last_error("dpipe_gemm: DPIPE_ACT_GEGLU_BWD needs the pipelined bf16 kernel (aligned h / dh, N % 4 == 0, no bias, no accumulation)"); return DPIPE_ERR_UNSUPPORTED; }
    if (colsum) { set_last_error("dpipe_gemm: fused column sum needs the pipelined kernel with a K-major A operand (transA = 1)"); return DPIPE_ERR_UNSUPPORTED; }
    // 16-byte loads need an aligned base and vector-multiple strides; ragged edges are handled per vector in load_tile.
    auto vec_ok = [&](const void* ptr, long ld, long so, long si) {
        return ((reinterpret_cast<uintptr_t>(ptr) & 15) == 0) && (ld % V == 0) && (so % V == 0) && (si % V == 0);
    };
    p.vecA = vec_ok(A, lda, strideA_outer, strideA_inner);
    p.vecB = vec_ok(B, ldb, strideB_outer, strideB_inner);
    // Tile choice: 128x128 when it yields at least ~one wave of workgroups over the 256 CUs, else 64x64.
    long t128 = (long)((M + 127) / 128) * ((N + 127) / 128) * batch;
    bool big = tile_hint == 128 || (tile_hint == 0 && t128 >= 192);
    if (dtype == DPIPE_BF16) return big ? launch_cfg<bf16_t, 128, 128>(p, transA, transB, batch, s) : launch_cfg<bf16_t, 64, 64>(p, transA, transB, batch, s);
    return big ? launch_cfg<float, 128, 128>(p, transA, transB, batch, s) : launch_cfg<float, 64, 64>(p, transA, transB, batch, s);
}

int dpipe_gemm_group(const dpipe_gemm_desc* descs, int n, void* splitk_ws, long splitk_ws_bytes, int* launches_out, void* stream) {
    if (!descs || n <= 0 || n > 16) { set_last_error("dpipe_gemm_group: 1 <= n <= 16 descriptors"); return DPIPE_ERR_ARG; }
    if (ablated(ABL_GEMM)) { if (launches_out) *launches_out = 0; return DPIPE_OK; }      // (debug switch: runtime.hip)
    GemmParams ps[16];
    int ta[16], tb[16];
    bool pipe[16];
    int npipe = 0;
    for (int i = 0; i < n; ++i) {
        const dpipe_gemm_desc& d = descs[i];
        if (!d.A || !d.B || !d.C || d.M <= 0 || d.N <= 0 || d.K <= 0) { set_last_error("dpipe_gemm_group: bad argument"); return DPIPE_ERR_ARG; }
        if (d.dtype != DPIPE_BF16 && d.dtype != DPIPE_F32) { set_last_error("dpipe_gemm_group: dtype"); return DPIPE_ERR_UNSUPPORTED; }
        if (d.residual && d.residual == d.C && !d.accumulate) { set_last_error("dpipe_gemm_group: residual may not alias C"); return DPIPE_ERR_ARG; }
        GemmParams p;
        p.A = d.A; p.B = d.B; p.C = d.C; p.bias = d.bias; p.M = d.M; p.N = d.N; p.K = d.K; p.lda = d.lda;
        if (ablate_gemm_kdiv() > 1 && p.K >= 128) p.K = (p.K / ablate_gemm_kdiv() + 63) / 64 * 64; p.ldb = d.ldb; p.ldc = d.ldc;
        p.sAo = p.sAi = p.sBo = p.sBi = p.sCo = p.sCi = 0;
        p.batch_inner = 1; p.alpha = d.alpha; p.act = d.act; p.accumulate = d.accumulate; p.out_f32 = d.out_f32;
        p.splitk = 1; p.ksteps = 0; p.ksteps_per_split = 0; p.slabs = nullptr; p.counters = nullptr;
        p.residual = d.residual; p.ldr = d.ldr; p.colsum = d.colsum; p.colsum_acc = d.colsum_accumulate; p.bias_rows = 0; p.bias_lo = 0;
        pipe[i] = d.dtype == DPIPE_BF16 && gemm_pipe_eligible(p, d.transA, d.transB);
        if (d.colsum && !pipe[i]) { set_last_error("dpipe_gemm_group: fused column sum needs the pipelined kernel with a K-major A operand (transA = 1)"); return DPIPE_ERR_UNSUPPORTED; }
        if (pipe[i]) { ps[npipe] = p; ta[npipe] = d.transA; tb[npipe] = d.transB; ++npipe; }
    }
    int launches = 0;
    if (npipe > 0) {
        const int rc = gemm_pipe_group(ps, ta, tb, npipe, splitk_ws, splitk_ws_bytes, reinterpret_cast<hipStream_t>(stream), &launches);
        if (rc != DPIPE_OK) return rc;
    }
    for (int i = 0; i < n; ++i) {          // fp32 / unaligned / ragged-K problems: the generic kernel, one launch each
        if (pipe[i]) continue;
        const dpipe_gemm_desc& d = descs[i];
        const int rc = dpipe_gemm_ex(d.dtype, d.transA, d.transB, d.M, d.N, d.K, d.A, d.lda, d.B, d.ldb, d.C, d.ldc, 1, 1, 0, 0, 0, 0, 0, 0, d.bias, d.act, d.alpha,
                                     d.accumulate, d.out_f32, 0, nullptr, 0, d.residual, d.ldr, nullptr, 0, stream);
        if (rc != DPIPE_OK) return rc;
        ++launches;
    }
    if (launches_out) *launches_out = launches;
    return DPIPE_OK;
}

int dpipe_gemm_group_plan(const dpipe_gemm_desc* descs, int n, long splitk_ws_bytes, int* tiles_out, int* splitk_out, int* launches_out) {
    if (!descs || !tiles_out || n <= 0 || n > 16) { set_last_error("dpipe_gemm_group_plan: 1 <= n <= 16 descriptors and a tiles_out array"); return DPIPE_ERR_ARG; }
    GemmParams ps[16];
    int ta[16], tb[16];
    for (int i = 0; i < n; ++i) {
        const dpipe_gemm_desc& d = descs[i];
        GemmParams p;
        p.A = d.A; p.B = d.B; p.C = d.C; p.bias = d.bias; p.M = d.M; p.N = d.N; p.K = d.K; p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
        p.sAo = p.sAi = p.sBo = p.sBi = p.sCo = p.sCi = 0;
        p.batch_inner = 1; p.alpha = d.alpha; p.act = d.act; p.accumulate = d.accumulate; p.out_f32 = d.out_f32;
        p.splitk = 1; p.ksteps = 0; p.ksteps_per_split = 0; p.slabs = nullptr; p.counters = nullptr;
        p.residual = d.residual; p.ldr = d.ldr; p.colsum = d.colsum; p.colsum_acc = d.colsum_accumulate; p.bias_rows = 0; p.bias_lo = 0;
        if (d.dtype != DPIPE_BF16 || !gemm_pipe_eligible(p, d.transA, d.transB)) { set_last_error("dpipe_gemm_group_plan: a problem is not eligible for the pipelined kernel"); return DPIPE_ERR_UNSUPPORTED; }
        ps[i] = p; ta[i] = d.transA; tb[i] = d.transB;
    }
    // planning only does pointer arithmetic on the workspace: any non-null base of the caller's size gives the caller's plan
    return gemm_pipe_group(ps, ta, tb, n, reinterpret_cast<void*>(0x1000), splitk_ws_bytes, nullptr, launches_out, tiles_out, splitk_out);
}

int dpipe_gemm(int dtype, int transA, int transB, int M, int N, int K,
               const void* A, long lda, const void* B, long ldb, void* C, long ldc,
               int batch_outer, int batch_inner,
               long strideA_outer, long strideA_inner, long strideB_outer, long strideB_inner,
               long strideC_outer, long strideC_inner,
               const void* bias, int act, float alpha, int accumulate, int out_f32, int tile_hint, void* stream) {
    return dpipe_gemm_ex(dtype, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, batch_outer, batch_inner, strideA_outer, strideA_inner,
                         strideB_outer, strideB_inner, strideC_outer, strideC_inner, bias, act, alpha, accumulate, out_f32, tile_hint,
                         nullptr, 0, nullptr, 0, nullptr, 0, stream);
}

int dpipe_tr16_probe(const void* in256_i16, void* out256_i16, void* stream) {
    if (!in256_i16 || !out256_i16) { set_last_error("dpipe_tr16_probe: null"); return DPIPE_ERR_ARG; }
    tr16_probe_kernel<<<1, 64, 0, reinterpret_cast<hipStream_t>(stream)>>>((const short*)in256_i16, (short*)out256_i16);
    return check_launch("dpipe_tr16_probe");
}

}  // extern "C"
