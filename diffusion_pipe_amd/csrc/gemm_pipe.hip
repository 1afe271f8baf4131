// gemm_pipe.hip -- dispatch of the plain bf16 GEMM onto the LDS-DMA pipelined kernel (gemm_pipe_kernel.h): eligibility, tile / split-K choice.
#include <stdlib.h>
#include "gemm_pipe_kernel.h"
#include "../../include/dpipe_hip.h"


using namespace dpipe_pipe;

namespace dpipe {

// the largest instantiations live in translation units of their own (build wall time)
int gemm_pipe_launch_group(int geom, const GemmGroup& g, int total_wg, hipStream_t s);      // gemm_pipe_group.hip
int gemm_pipe_launch_256(int tile, const GemmParams& p, bool a_mc, bool b_mc, int batch, hipStream_t s);      // gemm_pipe_256.hip

// eligibility for the LDS-DMA kernel: 16-byte DMA pieces, K-contiguous operands need whole K-steps, 32-bit buffer offsets
static bool pipe_eligible(const GemmParams& p, bool a_mc, bool b_mc) {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al16(p.A) || !al16(p.B)) return false;
    if (p.lda % 8 || p.ldb % 8 || p.sAo % 8 || p.sAi % 8 || p.sBo % 8 || p.sBi % 8) return false;
    if ((!a_mc || !b_mc) && (p.K % BK) != 0) return false;
    if (p.colsum && !a_mc) return false;                      // fused column sums read the K-major A image only
    if (p.act & ACT_GEGLU_BWD) {                               // the GEGLU-backward epilogue exists in the batched bf16 form only (gemm_pipe_kernel.h)
        const auto al8 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 7) == 0; };
        if (!p.residual || p.bias || p.accumulate || p.out_f32 || (p.N & 3) || (p.ldr & 3) || (p.ldc & 3) || (p.sCo & 3) || (p.sCi & 3) || !al8(p.residual) || !al8(p.C)) return false;
    }
    const long kpad = (long)((p.K + BK - 1) / BK) * BK, mpad = (long)((p.M + 255) / 256) * 256, npad = (long)((p.N + 255) / 256) * 256;
    const long ext_a = a_mc ? kpad * p.lda + mpad : mpad * p.lda + kpad;
    const long ext_b = b_mc ? kpad * p.ldb + npad : npad * p.ldb + kpad;
    return ext_a * 2 < (1L << 31) && ext_b * 2 < (1L << 31);
}

static int launch_by_tile(int tile, const GemmParams& p, bool a_mc, bool b_mc, int batch, hipStream_t s) {
    switch (tile) {
    case 257: case 258: return gemm_pipe_launch_256(tile, p, a_mc, b_mc, batch, s);       // (gemm_pipe_256.hip)
    case 129: return launch_pipe<T128R2>(p, a_mc, b_mc, batch, s);
    case 128: return launch_pipe<T128>(p, a_mc, b_mc, batch, s);
    case 132: return launch_pipe<T128V>(p, a_mc, b_mc, batch, s);
    default: return launch_pipe<T64>(p, a_mc, b_mc, batch, s);
    }
}

bool gemm_pipe_eligible(const GemmParams& p, int transA, int transB) { return pipe_eligible(p, transA != 0, transB == 0); }

bool gemm_pipe_try(GemmParams& p, int transA, int transB, int batch, void* ws, long ws_bytes, int force_splitk, int force_tile,
                   hipStream_t s, int* rc_out) {
    const bool a_mc = transA != 0, b_mc = transB == 0;
    if (!pipe_eligible(p, a_mc, b_mc)) return false;
    const int tile = gemm_pipe_plan(p, a_mc, b_mc, batch, ws, ws_bytes, force_splitk, force_tile);
    *rc_out = launch_by_tile(tile, p, a_mc, b_mc, batch, s);
    return true;
}

// Grouped launch of n independent plain GEMMs (dpipe_gemm_group).  Each problem is planned exactly as a single launch would plan it (same tile, same split-K:
// bit-identical results), on a private share of the split-K workspace; problems that land on the same groupable tile geometry (64^2 4-deep, 128^2 2-deep,
// 128^2 3-deep) leave as ONE launch of up to GROUP_MAX problems, the workgroups of the problem with the longest K walk first; a dgrad / wgrad pair whose
// geometries differ is re-planned onto the 64^2 tile when that keeps it one launch (`pair_unify`).  Everything else goes out as single launches.
int gemm_pipe_group(GemmParams* ps, const int* transA, const int* transB, int n, void* ws, long ws_bytes, hipStream_t s, int* launches_out, int* tiles_out, int* splitk_out) {
    const bool dry = tiles_out != nullptr;                 // dpipe_gemm_group_plan: plan only, report the tile code / split factor of every problem, launch nothing
    struct Plan { int tile; bool a_mc, b_mc; long tiles; int counter_base; long slab_base; };
    Plan pl[16];
    if (n > 16) { set_last_error("dpipe_gemm_group: at most 16 problems"); return DPIPE_ERR_ARG; }
    auto slab_bytes_of = [](int tile) { const long bm = (tile == 128 || tile == 129 || tile == 132) ? 128 : 64; return (bm * bm + bm) * 4L; };
    int force_of[16] = {};
    auto plan_all = [&]() {
        int cbase = 0; long sbase = 0;
        for (int i = 0; i < n; ++i) {
            const int force = force_of[i];
            GemmParams& p = ps[i];
            p.splitk = 1; p.ksteps = 0; p.ksteps_per_split = 0; p.slabs = nullptr; p.counters = nullptr;
            pl[i].a_mc = transA[i] != 0; pl[i].b_mc = transB[i] == 0;
            pl[i].counter_base = cbase; pl[i].slab_base = sbase;
            pl[i].tile = gemm_pipe_plan(p, pl[i].a_mc, pl[i].b_mc, 1, ws, ws_bytes, 0, force, cbase, sbase);
            pl[i].tiles = (long)p.tiles_m * p.tiles_n;
            if (p.splitk > 1) {
                cbase += (int)pl[i].tiles;
                sbase += pl[i].tiles * p.splitk * slab_bytes_of(pl[i].tile);
                sbase = (sbase + 255) & ~255L;
            }
        }
    };
    plan_all();
    auto groupable = [](int tile) { return tile == 64 || tile == 129 || tile == 128 || tile == 132; };
    auto geom = [](int tile) { return tile == 132 ? 129 : tile; };        // the register-staged tile shares T128R2's geometry and LDS: one grouped launch carries both
    static const bool pair_unify = [] { const char* e = getenv("DPIPE_GEMM_GROUP_UNIFY"); return !e || atoi(e) != 0; }();
    if (pair_unify && n == 2 && geom(pl[0].tile) != geom(pl[1].tile) && groupable(pl[0].tile) && groupable(pl[1].tile)) {
        // (round 6) a register-staged member (132: T128R2's geometry) next to a partner on the 3-deep 128^2 ring (128: single-lane engines, 128 .. 255 tiles)
        // is re-planned onto its partner's ring -- both keep 128^2 tiles and leave as one T128 launch, as before the register-staged rule existed; only a pair that
        // mixes a 64^2 member with a 128^2 one is unified onto 64^2 tiles
        const int vs = pl[0].tile == 132 ? 0 : pl[1].tile == 132 ? 1 : -1;
        if (vs >= 0 && pl[1 - vs].tile == 128) force_of[vs] = 128;
        else force_of[0] = force_of[1] = 64;
        plan_all();
    }
    bool done[16] = {};
    int launches = 0, rc = DPIPE_OK;
    for (int i = 0; i < n && rc == DPIPE_OK; ++i) {
        if (done[i]) continue;
        int members[GROUP_MAX], m = 0;
        members[m++] = i;
        if (groupable(pl[i].tile))
            for (int j = i + 1; j < n && m < GROUP_MAX; ++j)
                if (!done[j] && geom(pl[j].tile) == geom(pl[i].tile)) members[m++] = j;
        for (int k = 0; k < m; ++k) done[members[k]] = true;
        ++launches;
        if (dry) continue;
        if (m == 1) { rc = launch_by_tile(pl[i].tile, ps[i], pl[i].a_mc, pl[i].b_mc, 1, s); continue; }
        // longest K walk first: its workgroups start in the first round, the short ones fill the tail
        for (int a = 1; a < m; ++a)
            for (int b = a; b > 0 && ps[members[b]].ksteps_per_split > ps[members[b - 1]].ksteps_per_split; --b) { const int t = members[b]; members[b] = members[b - 1]; members[b - 1] = t; }
        GemmGroup g;
        g.n = m;
        int at = 0;
        for (int k = 0; k < GROUP_MAX; ++k) {
            if (k < m) {
                const int id = members[k];
                g.p[k] = ps[id]; g.mode[k] = 2 * (int)pl[id].a_mc + (int)pl[id].b_mc + (pl[id].tile == 132 ? 4 : 0);
                g.start[k] = at; g.nwg[k] = (int)(pl[id].tiles * ps[id].splitk);
                at += (g.nwg[k] + 7) & ~7;
            } else { g.p[k] = ps[members[0]]; g.mode[k] = 0; g.start[k] = at; g.nwg[k] = 0; }
        }
        rc = gemm_pipe_launch_group(geom(pl[i].tile), g, at, s);                                // (gemm_pipe_group.hip)
    }
    if (launches_out) *launches_out = launches;
    if (dry) for (int i = 0; i < n; ++i) { tiles_out[i] = pl[i].tile; if (splitk_out) splitk_out[i] = ps[i].splitk; }
    return rc;
}

int gemm_pipe_plan(GemmParams& p, bool a_mc, bool b_mc, int batch, void* ws, long ws_bytes, int force_splitk, int force_tile, int counter_base, long slab_base, bool allow_vs) {
    p.ksteps = (p.K + BK - 1) / BK;
    const long tiles128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * batch;
    const long tiles64 = (long)((p.M + 63) / 64) * ((p.N + 63) / 64) * batch;
    // tile (measured on the SDXL shapes, tools/kernel_timing.py): 128 x 128 from ~half a wave of workgroups on, or when a
    // long K can be split three ways over few tiles; else 64 x 64 (4 x the workgroups, 2 resident per CU)
    const bool long_k = p.ksteps >= 48;
    // DPIPE_OPT_GEMM_BIG_TILES (default 128): fewest 128^2 tiles for which the 128^2 tile is chosen.  The default is the isolated-launch optimum (a launch of < 128
    // tiles fills more CUs as 4 x as many 64^2 tiles); with several micro-batch lanes replaying at once the chip is full either way and what counts is the CU time a
    // launch takes: a 64^2 tile moves twice the operand bytes per FLOP through the CU's global -> LDS path, which is what saturates -- the engine
    // lowers the threshold when it runs >= 2 lanes
    // (a lowered threshold only applies where the 128^2 tiling covers the problem about as tightly as the 64^2 one -- padded area within 1.25 x: the rank-32 LoRA
    //  projections, N = 32, keep 64^2 tiles; M = 77 pads to 128 rows either way)
    const long area128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * 16384, area64 = (long)((p.M + 63) / 64) * ((p.N + 63) / 64) * 4096;
    const int big_min = (area128 * 4 <= area64 * 5) ? option(DPIPE_OPT_GEMM_BIG_TILES, 128) : 128;
    const bool big = force_tile ? (force_tile >= 128) : (tiles128 >= big_min || (long_k && tiles128 >= (big_min < 48 ? big_min : 48)));
    // 256 x 256 (T256S) for DiT-sized forward / dgrad GEMMs: K-contiguous or mixed operands, >= 64 K-steps to amortise the
    // un-overlapped prologue / epilogue of the one resident workgroup, >= half a wave of 256^2 tiles.  Measured
    // (tools/kernel_timing.py large): +9 .. +21 % over T128R2 there (8192^3: 1.27 vs 1.08 PFLOP/s), -2 .. -6 % at K = 3072, and
    // wgrad (both operands MN-contiguous) stays faster on T128R2
    const long tiles256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256) * batch;
    // ... and on the 4-deep ring of half K-steps (T256K) when B is MN-contiguous (dgrad): its DMA pieces are whole k-rows either way, and three half steps in
    // flight shorten the vmcnt wait: +5 .. +6 % (9216 x 5120 x 13824 NN: 996 vs 940 TFLOP/s).  A K-contiguous operand pays for half steps with 64-byte pieces
    // (half a cache line per row per step) and twice the barriers: NT is 8 % slower on T256K and stays on T256S (8192^3: 1 292 vs 1 184)
    if (force_tile == 0 && !a_mc && p.ksteps >= 64 && tiles256 >= 128 && !(p.act & ACT_GEGLU_BWD)) force_tile = b_mc ? 258 : 257;       // (the GEGLU epilogue: 64^2 / 128^2 tiles)
    // (Round 4 negative result, profiles/r4h_bench_tile_policy.jsonl: the 256^2 tile for SDXL-sized forward / dgrad problems under four lanes -- from 16 / 48 tiles and 16 K-steps on --
    //  19.62 / 20.72 images/s against 21.10 with the 128^2 rule below, same box: one 128 KiB workgroup per CU leaves no room for another lane's workgroup.)
    // (Removed in round 5 after losing twice: the occupancy-style 4-wave tile for short-K forwards (DPIPE_GEMM_Q3: single-stream list +0.7 %, four-lane step -0.8 %) and the
    //  skinny 128 x 64 tile for M <= 128 (DPIPE_GEMM_SKINNY: 56 vs 39 ms per step over those launches).)
    const int bm = (force_tile == 257 || force_tile == 258) ? 256 : big ? 128 : 64, bn = bm;
    p.tiles_m = (p.M + bm - 1) / bm; p.tiles_n = (p.N + bn - 1) / bn;
    const long tiles = (long)p.tiles_m * p.tiles_n * batch;
    const long slab_bytes = ((long)bm * bn + bm) * 4;
    int S = 1;
    if (ws && ws_bytes > COUNTER_BYTES) {
        if (force_splitk > 0) S = force_splitk;
        else if (!big && tiles <= 128 && p.ksteps >= 32) {   // few, long tiles split
            // up to 4 slices; 8 once a slice would still walk >= 32 K-steps (the batched context K / V dgrad [77, 2048] <- [77, 25600]: 69.8 -> 44.9 us)
            S = (int)(512 / tiles); const int cap = p.ksteps / 8; if (S > cap) S = cap; const int smax = p.ksteps >= 256 ? 8 : 4; if (S > smax) S = smax;
        } else if (!big && tiles <= 64 && p.ksteps >= 16) {
            // <= 64 tiles of 16 .. 31 K-steps (the 77-token and 1-token linears at K = 1 280 / 1 024): four slices of >= 4 K-steps -- with write-through
            // slabs a slice no longer pays an L2 write-back scan (profiles/r3_gemm_desc_ledger.jsonl: NN 15.1 -> 9.2 us, NT 11.1 -> 10.4 us at [77, 1280, 1280])
            S = 4; const int cap = p.ksteps / 4; if (S > cap) S = cap;
        } else if (big && long_k && tiles <= 128) {
            S = (int)(256 / tiles); if (S > 3) S = 3; if (S < 2) S = 2;      // one round of workgroups: 80 tiles x 3, 120 x 2 (measured: tools/kernel_timing.py, tools/conv_timing.py)
        } else if (big && long_k && tiles < 256) {
            // 129 .. 255 tiles: split only when every slice keeps >= 48 K-steps ([4096,5120] x [5120,640], 160 tiles x 80 steps, runs best unsplit: 54 vs 67 us;
            // the 3x3 convolutions' 135 .. 225 tiles x 180 .. 256 steps gain 25 .. 45 % from 2 - 3 slices)
            S = (int)(512 / tiles); const int cap = p.ksteps / 48; if (S > cap) S = cap; if (S > 3) S = 3;
        }
        // (Round 4: capping the automatic split factor at 1 / 2 under four lanes -- 20.90 / 21.14 vs 21.09 images/s, same box -- buys nothing: removed again.)
        if (S < 1) S = 1;
        if (S > p.ksteps) S = p.ksteps;
        // (counter_base, slab_base: the share of the workspace problems planned earlier into the same grouped launch already own)
        const long max_slabs = (ws_bytes - COUNTER_BYTES - slab_base) / slab_bytes;
        if (tiles * S > max_slabs) S = (int)(max_slabs / tiles);
        if (counter_base + tiles > COUNTER_BYTES / 4 || S < 1) S = 1;
    }
    p.ksteps_per_split = (p.ksteps + S - 1) / S;
    p.splitk = (p.ksteps + p.ksteps_per_split - 1) / p.ksteps_per_split;
    {   // rasterisation group height: an XCD owns a contiguous run of R = tiles / 8 tiles (the slices of a tile are adjacent); walking g tile-rows before moving a tile-column
        // over, the run touches ~g + R / g operand panels -- least at g = sqrt(R).  DPIPE_GEMM_GR=8 restores the fixed height of rounds 1 - 4 (A/B)
        static const int gr_fixed = [] { const char* e = getenv("DPIPE_GEMM_GR"); return e ? atoi(e) : 0; }();
        const long run = (long)p.tiles_m * p.tiles_n / 8;
        int g = 1;
        while ((long)(g + 1) * (g + 1) <= run) ++g;
        if ((long)g * (g + 1) < run) ++g;          // round to nearest
        p.gr = gr_fixed > 0 ? gr_fixed : (g < 1 ? 1 : g > 8 ? 8 : g);
    }
    p.counters = ws ? reinterpret_cast<int*>(ws) + counter_base : nullptr;
    p.slabs = ws ? reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + COUNTER_BYTES + slab_base) : nullptr;
    // vector epilogue flags (re-using the generic kernel's fields): vecA >= 2 -> C rows allow 4-element vector accesses,
    // vecB >= 2 -> bias allows 8-byte loads
    const int celt = p.out_f32 ? 4 : 2;
    const bool c_ok = (reinterpret_cast<uintptr_t>(p.C) % (4 * celt) == 0) && p.ldc % 4 == 0 && p.sCo % 4 == 0 && p.sCi % 4 == 0;
    p.vecA = c_ok ? 2 : 0;
    if (c_ok && p.residual && reinterpret_cast<uintptr_t>(p.residual) % 8 == 0 && p.ldr % 4 == 0) p.vecA = 3;   // 8-byte residual loads too
    p.vecB = (p.bias && reinterpret_cast<uintptr_t>(p.bias) % 8 == 0) ? 2 : 0;
    if (force_tile == 258) { p.ksteps *= 2; p.ksteps_per_split *= 2; return force_tile; }      // T256K counts K in 32-wide half steps (slices keep their K ranges)
    if (force_tile == 257 || force_tile == 132) return force_tile;
    // Register-staged 128^2 tile (T128V, round 5) wherever A is K-contiguous (forward NT, dgrad NN) and a workgroup walks >= 8 K-steps: -3 .. -19 % against the better of
    // the two DMA rings on every such descriptor of the SDXL step (gemm_pipe_kernel.h has the table), split tiles included ([1024, 1280] <- 10240 NN in three slices 42.6 vs
    // 49.7 us, <- 3840 23.4 vs 25.2: profiles/r5c_gemm_desc_ledger_register_staged_and_gr.jsonl); the wgrad layouts stay on the DMA rings.  DPIPE_GEMM_VS=0: rule off (A/B).
    static const bool vs_rule = [] { const char* e = getenv("DPIPE_GEMM_VS"); return !e || atoi(e) != 0; }();
    if (vs_rule && allow_vs && force_tile == 0 && big && !a_mc && p.ksteps_per_split >= 8 && batch == 1) return 132;
    if (force_tile == 129 || (force_tile == 0 && big && tiles128 >= 256 && (a_mc || b_mc || tiles128 >= 1024))) return 129;
    // DPIPE_OPT_GEMM_SHALLOW: the 128^2 tile on its 2-deep 64 KiB ring everywhere -- slower launches in isolation (step list: 23.5 vs 22.0 us average), but a 64 KiB
    // footprint lets a workgroup of ANOTHER micro-batch lane share the CU: 19.34 vs 18.93 images/s with 3 lanes (profiles/r3e_bench_variants.jsonl).  The engine sets
    // it when it replays >= 2 lanes.  (The 64^2 tile's 3-deep variant never paid and is gone.)
    if (option(DPIPE_OPT_GEMM_SHALLOW, 0) != 0 && force_tile == 0 && big) return 129;
    return big ? 128 : 64;
}

}  // namespace dpipe
