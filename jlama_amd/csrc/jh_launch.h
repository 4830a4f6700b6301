// jh_launch.h -- launch planners of the GEMV / GEMM kernels (templates: every translation unit instantiates what it launches).
#pragma once
#include "jh_host.h"
#include "jh_t16.h"
#include "jh_bf16r.h"

// (R, NB, PIPE) instantiations of gemv_i8q4_kernel
#define JH_GEMV_COMBOS(X)                                                                                         \
    X(1, 1, 0) X(1, 2, 0) X(1, 4, 0) X(1, 7, 0) X(2, 1, 0) X(2, 2, 0) X(2, 4, 0) X(2, 7, 0) X(4, 1, 0) X(4, 2, 0)  \
    X(4, 4, 0) X(8, 1, 0) X(8, 2, 0) X(14, 2, 0)                                                                  \
    X(2, 2, 1) X(4, 2, 1) X(2, 4, 1) X(4, 1, 1) X(8, 1, 1) X(2, 7, 1) X(1, 0, 1) X(2, 0, 1) X(4, 0, 1)

template <int PRO, int EPI>
int launch_gemv_i8q4_combo(const GemvParams& p, int R, int NB, int PIPE, int grid, int threads, hipStream_t st) {
    const size_t lds = lds_bytes_i8(p.K);
#define X(RV, NBV, PV)                                                                                          \
    if (R == RV && NB == NBV && PIPE == PV) {                                                                   \
        if constexpr (!(EPI == EPI_SILU_MUL && ((RV) & 1))) {                                                   \
            JHCHK(allow_lds(gemv_i8q4_kernel<PRO, EPI, RV, NBV, PV>, lds));                                     \
            hipLaunchKernelGGL((gemv_i8q4_kernel<PRO, EPI, RV, NBV, PV>), dim3(grid), dim3(threads), lds, st, p); \
            g_last_gemv_grid = grid;                                                                            \
            HIPCHK(hipGetLastError());                                                                          \
            return JH_OK;                                                                                       \
        }                                                                                                       \
    }
    JH_GEMV_COMBOS(X)
#undef X
    return set_err(JH_ERR_INVALID, "gemv: no kernel instantiation for R=" + std::to_string(R) + " NB=" + std::to_string(NB) +
                                       " PIPE=" + std::to_string(PIPE));
}

// Plan a GEMV launch.  Goal (tools/membw.hip calibration): every CU gets one workgroup whose waves each hold
// R*NB >= ~8 independent 16-byte loads, ALL issued before the activation prologue (PIPE=0).  When a wave's share of
// rows does not fit in registers, fall back to the software-pipelined loop (PIPE=1).
template <int PRO, int EPI>
int launch_gemv_i8q4(const GemvParams& p, LaunchCfg cfg, hipStream_t st) {
    const int total = (EPI == EPI_SILU_MUL) ? 2 * p.nrows : p.nrows;
    const int nb = nb_for(p.K);
    const int cu = g_cu_count;
    auto divides = [&](int R) {
        if (EPI == EPI_SILU_MUL) return (R % 2 == 0) && (p.nrows % (R / 2) == 0);
        return p.nrows % R == 0;
    };
    int R = cfg.R, pipe = cfg.pipe, waves = cfg.waves;
    if (nb == 0) {
        pipe = 1;
        if (R <= 0) R = 2;
        while (R > 1 && !divides(R)) R >>= 1;
        if (EPI == EPI_SILU_MUL && R < 2) return set_err(JH_ERR_INVALID, "gate/up GEMV needs an even hidden length");
        if (waves <= 0) waves = 8;
    } else if (R <= 0 || pipe < 0) {
        // Measured on MI355X (tools/gemv_lab2.hip, gate/up-sized matrix): issuing a wave's whole weight stream before the
        // prologue ("single shot", PIPE=0) only helps when the prologue is trivial -- a CU keeps ~32-64 KB of loads in
        // flight, so a wave that queues more than that is still ISSUING loads when its prologue should already run.
        // The software-pipelined loop with ~8 KB per wave prefetched (R*NB = 4..8) is the better default.
        if (pipe < 0) {
            // small/medium GEMVs (<= ~200 KB of weights per CU) run best single-shot, the big gate/up one pipelined
            const double bytes_per_cu = (double)total * p.K * 0.625 / cu;
            pipe = bytes_per_cu <= 200e3 ? 0 : 1;
        }
        if (R <= 0) {
            if (pipe >= 1) R = (nb == 1) ? 4 : 2;
            else {
                const int need = (total + cu * 16 - 1) / (cu * 16);
                static const int oneshot[] = {1, 2, 4, 8, 14};
                R = 1;
                for (int r : oneshot) {
                    const bool inst = (r == 1 || r == 2) || (r == 4 && nb <= 4) || (r == 8 && nb <= 2) || (r == 14 && nb == 2);
                    if (inst && r >= need && r * nb <= 28 && divides(r)) { R = r; break; }
                }
            }
        }
        while (R > 2 && !divides(R)) R >>= 1;
    }
    const int ngroups = total / R;
    int grid, threads;
    if (pipe == 0) {
        // VGPR budget: 1024-thread blocks are capped at 128 registers
        int wmax = (R * nb <= 4) ? 16 : 8;   // must match gemv_i8q4_kernel __launch_bounds__
        if (waves <= 0) {
            waves = (ngroups + cu - 1) / cu;
            if (waves < 4) waves = 4;
            if (waves > wmax) waves = wmax;
        }
        if (waves > wmax) waves = wmax;
        grid = (ngroups + waves - 1) / waves;
    } else {
        const int wmax = (nb > 0 && R * nb <= 4) ? 16 : 8;   // must match gemv_i8q4_kernel __launch_bounds__
        if (waves <= 0) waves = 8;
        if (waves > wmax) waves = wmax;
        grid = (ngroups + waves - 1) / waves;
        const int cap = cfg.grid_cap > 0 ? cfg.grid_cap : cu;
        if (grid > cap) grid = cap;
    }
    if (grid < 1) grid = 1;
    threads = waves * 64;
    return launch_gemv_i8q4_combo<PRO, EPI>(p, R, nb, pipe, grid, threads, st);
}

template <int PRO, int EPI, bool ARGMAX>
int launch_gemv_bf16(const GemvParams& p, int grid_cap, int* grid_out, hipStream_t st) {
    const size_t lds = lds_bytes_bf(p.K);
    const int total = (EPI == EPI_SILU_MUL) ? 2 * p.nrows : p.nrows;
    constexpr int R = (EPI == EPI_SILU_MUL) ? 2 : 1;   // 8 x 16-byte loads per row already keep a wave's queue full
    const int ngroups = total / R, waves = 8;
    int grid = (ngroups + waves - 1) / waves;
    if (grid > grid_cap) grid = grid_cap;
    if (grid < 1) grid = 1;
    if (grid_out) *grid_out = grid;
    JHCHK(allow_lds(gemv_bf16_kernel<PRO, EPI, R, ARGMAX>, lds));
    hipLaunchKernelGGL((gemv_bf16_kernel<PRO, EPI, R, ARGMAX>), dim3(grid), dim3(waves * 64), lds, st, p);
    HIPCHK(hipGetLastError());
    return JH_OK;
}

// GEMM planners of the prompt path and Tier 1 (gemm_launch.hip)
int launch_splitk_reduce(const float* ws, int S, int m, int n, int n0, float* c, int ldc, int roffset, const float* resid, hipStream_t st);
int launch_gemm_q8q4_mfma(const MfmaQ4Params& g, hipStream_t st, bool tiled = false, float* ws = nullptr, size_t ws_bytes = 0);
int launch_gemm_bf16_mfma(const MfmaGemmParams& g, hipStream_t st);
int launch_gemm_bf16_tile(const MfmaBf16TileParams& g, hipStream_t st);

template <int PRO, int R>
int launch_gemv_f32q4_r(const GemvParams& p, int grid, int threads, hipStream_t st) {
    const size_t lds = lds_bytes_f32(p.K);
    const int nb = nb_for(p.K);
    if (nb == 1) {
        JHCHK(allow_lds(gemv_f32q4_kernel<PRO, R, 1>, lds));
        hipLaunchKernelGGL((gemv_f32q4_kernel<PRO, R, 1>), dim3(grid), dim3(threads), lds, st, p);
    } else if (nb == 2) {
        JHCHK(allow_lds(gemv_f32q4_kernel<PRO, R, 2>, lds));
        hipLaunchKernelGGL((gemv_f32q4_kernel<PRO, R, 2>), dim3(grid), dim3(threads), lds, st, p);
    } else {
        JHCHK(allow_lds(gemv_f32q4_kernel<PRO, R, 0>, lds));
        hipLaunchKernelGGL((gemv_f32q4_kernel<PRO, R, 0>), dim3(grid), dim3(threads), lds, st, p);
    }
    HIPCHK(hipGetLastError());
    return JH_OK;
}
template <int PRO>
int launch_gemv_f32q4(const GemvParams& p, LaunchCfg cfg, int* grid_out, hipStream_t st) {
    int R = cfg.R <= 0 ? 4 : cfg.R;
    if (R > 4) R = 4;
    while (R > 1 && p.nrows % R) R >>= 1;
    const int ngroups = p.nrows / R;
    int grid = (ngroups + cfg.waves - 1) / cfg.waves;
    if (grid > cfg.grid_cap) grid = cfg.grid_cap;
    if (grid < 1) grid = 1;
    if (grid_out) *grid_out = grid;
    if (R == 4) return launch_gemv_f32q4_r<PRO, 4>(p, grid, cfg.waves * 64, st);
    if (R == 2) return launch_gemv_f32q4_r<PRO, 2>(p, grid, cfg.waves * 64, st);
    return launch_gemv_f32q4_r<PRO, 1>(p, grid, cfg.waves * 64, st);
}

// ---- reference-order launchers (jh_p16.h).  A wave serves 4 weight rows ("row quad"); the plan gives every CU the same
// number of row quads: one 512-thread workgroup per CU, `tw` of its 8 waves own `per` row quads each (the others help
// with the activation prologue only -- a 16-lane row per chain caps the useful waves at rows / 4).
inline P16Plan p16_plan(int nrows, int wgs_per_cu, int nwaves = 8) {
    const int nq = (nrows + 3) / 4;
    int grid = g_cu_count * wgs_per_cu;
    if (grid > nq) grid = nq;
    if (grid < 1) grid = 1;
    const int q_wg = (nq + grid - 1) / grid;
    const int per = (q_wg + nwaves - 1) / nwaves;
    const int tw = (q_wg + per - 1) / per;
    return P16Plan{grid, per, tw};
}
// prefetch depth D = groups (16 Q blocks) in flight per lane: the largest of {8, 7, 4, 2, 1} that divides the groups of a row, so
// that a pass ends exactly at the end of a ring block (K = 4096: 8 = the whole row; 14336: 7; 2048: 4)
inline int p16_depth_for(int K, int want) {
    const int G = (K / QB + 15) / 16;
    static const int ds[] = {8, 7, 4, 2, 1};
    for (int d : ds)
        if (d <= want && G % d == 0) return d;
    return 1;
}
// UM = 8-element units of the activation row per thread, all held in registers (no load loop in the kernel): 2 (K <= 8192) / 4 for
// the RMSNorm prologues, 4 (K <= 16384) / 8 for the plain-quantize ones
template <int PRO, int EPI, int D>
int launch_gemv_i8q4_p16_d(const GemvParams& p, const P16Plan& pl, bool wide, hipStream_t st) {
    const size_t lds = lds_bytes_p16(p.K);
    constexpr int UM_LO = (PRO == PRO_RMS_Q8) ? 2 : 4;
    (void)wide;
    if (p.nrows % 4) return set_err(JH_ERR_UNSUPPORTED, "reference-order GEMV: rows must be whole quads (a multiple of 4)");
    if (p.K <= UM_LO * 4096) {
        JHCHK(allow_lds((gemv_i8q4_p16_kernel<PRO, EPI, D, UM_LO>), lds));
        hipLaunchKernelGGL((gemv_i8q4_p16_kernel<PRO, EPI, D, UM_LO>), dim3(pl.grid), dim3(P16_THREADS), lds, st, p, pl.per, pl.tw);
    } else if (p.K <= 2 * UM_LO * 4096) {
        JHCHK(allow_lds((gemv_i8q4_p16_kernel<PRO, EPI, D, 2 * UM_LO>), lds));
        hipLaunchKernelGGL((gemv_i8q4_p16_kernel<PRO, EPI, D, 2 * UM_LO>), dim3(pl.grid), dim3(P16_THREADS), lds, st, p, pl.per, pl.tw);
    } else {
        return set_err(JH_ERR_UNSUPPORTED, "reference-order GEMV: K = " + std::to_string(p.K) + " exceeds the register-resident activation row");
    }
    HIPCHK(hipGetLastError());
    g_last_gemv_grid = pl.grid;
    return JH_OK;
}
template <int PRO, int EPI>
int launch_gemv_i8q4_p16(const GemvParams& p, int depth, hipStream_t st) {
    // more than 8 row quads per CU (gate|up): two workgroups per CU, so that every SIMD has 3-4 waves to issue from -- the kernel
    // is as much VALU- as HBM-bound, and a wave alone issues one instruction per ~4 cycles
    const P16Plan pl = p16_plan(p.nrows, 1, 8);
    const bool wide = false;
    // ring depth by bytes in flight per CU (tw waves x D KiB): ~32 KiB is what a CU sustains; deeper rings only cost registers
    // (measured: q|k|v and gate|up with 6-7 task waves 4 > 8, the o- and down-projections with 4 task waves 8 / 7 > 4)
    if (pl.tw >= 6 && depth > 4) depth = 4;
    switch (p16_depth_for(p.K, depth)) {
        case 8: return launch_gemv_i8q4_p16_d<PRO, EPI, 8>(p, pl, wide, st);
        case 7: return launch_gemv_i8q4_p16_d<PRO, EPI, 7>(p, pl, wide, st);
        case 4: return launch_gemv_i8q4_p16_d<PRO, EPI, 4>(p, pl, wide, st);
        case 2: return launch_gemv_i8q4_p16_d<PRO, EPI, 2>(p, pl, wide, st);
        default: return launch_gemv_i8q4_p16_d<PRO, EPI, 1>(p, pl, wide, st);
    }
}
template <int PRO, int D>
int launch_gemv_f32q4_p16_d(const GemvParams& p, const P16Plan& pl, hipStream_t st) {
    const size_t lds = lds_bytes_f32_p16(p.K);
    if (p.K <= 8192) {
        JHCHK(allow_lds((gemv_f32q4_p16_kernel<PRO, D, 2>), lds));
        hipLaunchKernelGGL((gemv_f32q4_p16_kernel<PRO, D, 2>), dim3(pl.grid), dim3(P16_THREADS), lds, st, p, pl.per, pl.tw);
    } else if (p.K <= 16384) {
        JHCHK(allow_lds((gemv_f32q4_p16_kernel<PRO, D, 4>), lds));
        hipLaunchKernelGGL((gemv_f32q4_p16_kernel<PRO, D, 4>), dim3(pl.grid), dim3(P16_THREADS), lds, st, p, pl.per, pl.tw);
    } else {
        return set_err(JH_ERR_UNSUPPORTED, "reference-order LM head: K exceeds the register-resident activation row");
    }
    HIPCHK(hipGetLastError());
    return JH_OK;
}
template <int PRO>
int launch_gemv_f32q4_p16(const GemvParams& p, int* grid_out, hipStream_t st) {
    const P16Plan pl = p16_plan(p.nrows, 2);   // two workgroups per CU; the argmax partial buffers hold 4096 entries
    if (grid_out) *grid_out = pl.grid;
    switch (p16_depth_for(p.K, 8)) {
        case 8: return launch_gemv_f32q4_p16_d<PRO, 8>(p, pl, st);
        case 7: return launch_gemv_f32q4_p16_d<PRO, 7>(p, pl, st);
        case 4: return launch_gemv_f32q4_p16_d<PRO, 4>(p, pl, st);
        case 2: return launch_gemv_f32q4_p16_d<PRO, 2>(p, pl, st);
        default: return launch_gemv_f32q4_p16_d<PRO, 1>(p, pl, st);
    }
}

// ---- reference-order GEMV on the integer MFMA (jh_t16.h): one wave per 16-row tile, one 512-thread workgroup per CU
// K % 256 == 0 (whole q steps of 4 blocks, an even number of them), K <= 8192 (register-resident activation row at 512 threads)
inline bool t16_shape_ok(int K) { return K % 256 == 0 && K <= 8192 && lds_bytes_t16(K) <= 150 * 1024; }
template <int PRO, int EPI>
int launch_gemv_t16(const GemvParams& p, hipStream_t st) {
    constexpr int NT = 512;
    const int ntiles = (EPI == EPI_SILU_MUL) ? p.nrows / 8 : p.nrows / 16;
    const int nq = p.K / QB / 4;
    int cus = g_cu_count < ntiles ? g_cu_count : ntiles;
    if (cus < 1) cus = 1;
    const int t_cu = (ntiles + cus - 1) / cus;
    const int tpw = (t_cu + 7) / 8;
    const int aw = (t_cu + tpw - 1) / tpw;
    const int grid = (ntiles + aw * tpw - 1) / (aw * tpw);
    const size_t lds = lds_bytes_t16(p.K);
#define JH_T16_LAUNCH(DV, UMV)                                                                                     \
    do {                                                                                                           \
        JHCHK(allow_lds((gemv_t16_kernel<PRO, EPI, DV, UMV, NT>), lds));                                           \
        hipLaunchKernelGGL((gemv_t16_kernel<PRO, EPI, DV, UMV, NT>), dim3(grid), dim3(NT), lds, st, p, tpw, aw);   \
    } while (0)
    if (nq % 4 == 0) {   // 4 q steps (4 KiB + 1 KiB of scales) in flight per wave; 8 measured slower (gate|up 14.2 -> 15.4 us, 186 VGPRs: round 5)
        if (p.K <= 4096) JH_T16_LAUNCH(4, 1); else JH_T16_LAUNCH(4, 2);
    } else {
        if (p.K <= 4096) JH_T16_LAUNCH(2, 1); else JH_T16_LAUNCH(2, 2);
    }
#undef JH_T16_LAUNCH
    HIPCHK(hipGetLastError());
    g_last_gemv_grid = grid;
    return JH_OK;
}
// ---- reference-order GEMV of a dense BF16 weight (jh_bf16r.h): same work split as the p16 kernels
template <int PRO, int EPI, bool ARGMAX>
int launch_gemv_bf16r(const GemvParams& p, int* grid_out, hipStream_t st) {
    if (p.K % 32 || p.K > 32768) return set_err(JH_ERR_UNSUPPORTED, "reference-order BF16 GEMV: K must be a multiple of 32, at most 32768");
    const P16Plan pl = p16_plan(p.nrows, ARGMAX ? 2 : 1);   // LM head: two workgroups per CU (argmax partial buffers hold 4096 entries)
    if (grid_out) *grid_out = pl.grid;
    const int G = (p.K + BF16R_GROUP - 1) / BF16R_GROUP;
    const size_t lds = lds_bytes_bfr(p.K);
#define JH_BFR(DV, UMV)                                                                                                          \
    do {                                                                                                                         \
        JHCHK(allow_lds((gemv_bf16r_kernel<PRO, EPI, ARGMAX, DV, UMV>), lds));                                                   \
        hipLaunchKernelGGL((gemv_bf16r_kernel<PRO, EPI, ARGMAX, DV, UMV>), dim3(pl.grid), dim3(P16_THREADS), lds, st, p, pl.per, pl.tw); \
    } while (0)
    // ring depth by bytes in flight per CU (task waves x D KiB): few-row matrices (o, down: 4 task waves per CU) take 16 groups per wave
    const int deep_opt = opt_int("JH_BF16R_DEEP", 16);      // upper bound of the ring depth (groups of 256 B per lane row in flight)
    const bool deep = G % 16 == 0 && deep_opt >= 16;
#define JH_BFR_D(UMV)                                                                                                            \
    do {                                                                                                                         \
        if (deep) JH_BFR(16, UMV);   /* measured on Mistral-7B: 8 -> 16 groups per lane row in flight +3.6 % tok/s, 28 / 32 slower (registers) */ \
        else if (G % 8 == 0) JH_BFR(8, UMV); else if (G % 4 == 0) JH_BFR(4, UMV); else if (G % 2 == 0) JH_BFR(2, UMV); else JH_BFR(1, UMV); \
    } while (0)
    if (p.K <= 8192) JH_BFR_D(2); else if (p.K <= 16384) JH_BFR_D(4); else JH_BFR_D(8);
#undef JH_BFR_D
#undef JH_BFR
    HIPCHK(hipGetLastError());
    g_last_gemv_grid = pl.grid;
    return JH_OK;
}

// Instantiated once, in gemv_*.hip: every other translation unit links against them instead of compiling the kernels again.
#ifndef JH_LAUNCH_INSTANTIATE
extern template int launch_gemv_i8q4<PRO_Q8, EPI_RESID>(const GemvParams&, LaunchCfg, hipStream_t);
extern template int launch_gemv_i8q4<PRO_Q8, EPI_SILU_MUL>(const GemvParams&, LaunchCfg, hipStream_t);
extern template int launch_gemv_i8q4<PRO_Q8, EPI_STORE>(const GemvParams&, LaunchCfg, hipStream_t);
extern template int launch_gemv_i8q4<PRO_QUANT_Q8, EPI_RESID>(const GemvParams&, LaunchCfg, hipStream_t);
extern template int launch_gemv_i8q4<PRO_QUANT_Q8, EPI_STORE>(const GemvParams&, LaunchCfg, hipStream_t);
extern template int launch_gemv_i8q4<PRO_QUANT_Q8, EPI_TP>(const GemvParams&, LaunchCfg, hipStream_t);
extern template int launch_gemv_i8q4<PRO_RMS_Q8, EPI_SILU_MUL>(const GemvParams&, LaunchCfg, hipStream_t);
extern template int launch_gemv_i8q4<PRO_RMS_Q8, EPI_STORE>(const GemvParams&, LaunchCfg, hipStream_t);
extern template int launch_gemv_f32q4<PRO_F32>(const GemvParams&, LaunchCfg, int*, hipStream_t);
extern template int launch_gemv_f32q4<PRO_RMS_F32>(const GemvParams&, LaunchCfg, int*, hipStream_t);
extern template int launch_gemv_i8q4_p16<PRO_QUANT_Q8, EPI_RESID>(const GemvParams&, int, hipStream_t);
extern template int launch_gemv_i8q4_p16<PRO_QUANT_Q8, EPI_STORE>(const GemvParams&, int, hipStream_t);
extern template int launch_gemv_i8q4_p16<PRO_QUANT_Q8, EPI_TP>(const GemvParams&, int, hipStream_t);
extern template int launch_gemv_i8q4_p16<PRO_RMS_Q8, EPI_SILU_MUL>(const GemvParams&, int, hipStream_t);
extern template int launch_gemv_i8q4_p16<PRO_RMS_Q8, EPI_STORE>(const GemvParams&, int, hipStream_t);
extern template int launch_gemv_f32q4_p16<PRO_RMS_F32>(const GemvParams&, int*, hipStream_t);
extern template int launch_gemv_t16<PRO_RMS_Q8, EPI_SILU_MUL>(const GemvParams&, hipStream_t);
extern template int launch_gemv_bf16<PROB_QUANT_BF16, EPI_RESID, false>(const GemvParams&, int, int*, hipStream_t);
extern template int launch_gemv_bf16<PROB_QUANT_BF16, EPI_STORE, false>(const GemvParams&, int, int*, hipStream_t);
extern template int launch_gemv_bf16<PROB_RMS_BF16, EPI_SILU_MUL, false>(const GemvParams&, int, int*, hipStream_t);
extern template int launch_gemv_bf16<PROB_RMS_BF16, EPI_STORE, false>(const GemvParams&, int, int*, hipStream_t);
extern template int launch_gemv_bf16<PROB_RMS_F32, EPI_STORE, true>(const GemvParams&, int, int*, hipStream_t);
extern template int launch_gemv_bf16r<PROB_QUANT_BF16, EPI_RESID, false>(const GemvParams&, int*, hipStream_t);
extern template int launch_gemv_bf16r<PROB_QUANT_BF16, EPI_STORE, false>(const GemvParams&, int*, hipStream_t);
extern template int launch_gemv_bf16r<PROB_RMS_BF16, EPI_SILU_MUL, false>(const GemvParams&, int*, hipStream_t);
extern template int launch_gemv_bf16r<PROB_RMS_BF16, EPI_STORE, false>(const GemvParams&, int*, hipStream_t);
extern template int launch_gemv_bf16r<PROB_RMS_F32, EPI_STORE, true>(const GemvParams&, int*, hipStream_t);
#endif
