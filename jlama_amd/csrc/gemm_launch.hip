// gemm_launch.hip -- part of libjlamahip.so (C ABI: include/jlama_hip.h).  Launch planners of the MFMA GEMMs (I8 x Q4, BF16) and their split-K reduce: the one place their kernels are instantiated.
#include "jh_host.h"
#include "jh_launch.h"

template <int MT>
int launch_gemm_q8q4_mfma_mt(const MfmaQ4Params& g, hipStream_t st) {
    const int tiles = g.n / 32;
    const int nblk = g.k / QB;
    // split-K so that small N still fills the chip: one workgroup per 32-column tile, KSPLIT waves each
    const int ksplit = (tiles >= g_cu_count * 4 || nblk < 8) ? 1 : (tiles >= g_cu_count * 2 || nblk < 16 ? 2 : 4);
    const size_t stage = (size_t)MT * 32 * MQ_ASTRIDE + 2 * (size_t)MT * 32 * 4;
    const size_t red = ksplit > 1 ? (size_t)ksplit * MT * 16 * 64 * 4 : 0;
    size_t lds = (size_t)ksplit * 2 * stage;
    if (red > lds) lds = red;
    if (ksplit == 4) { JHCHK(allow_lds(gemm_q8q4_mfma_kernel<MT, 4>, lds)); hipLaunchKernelGGL((gemm_q8q4_mfma_kernel<MT, 4>), dim3(tiles), dim3(256), lds, st, g); }
    else if (ksplit == 2) { JHCHK(allow_lds(gemm_q8q4_mfma_kernel<MT, 2>, lds)); hipLaunchKernelGGL((gemm_q8q4_mfma_kernel<MT, 2>), dim3(tiles), dim3(128), lds, st, g); }
    else { JHCHK(allow_lds(gemm_q8q4_mfma_kernel<MT, 1>, lds)); hipLaunchKernelGGL((gemm_q8q4_mfma_kernel<MT, 1>), dim3(tiles), dim3(64), lds, st, g); }
    HIPCHK(hipGetLastError());
    return JH_OK;
}
template <int S, bool TILED, int CW>
int launch_gemm_q8q4_tile(const MfmaQ4Params& g, int mtiles, hipStream_t st) {
    const int cgroups = g.n / (32 * CW), gg = (cgroups + 7) / 8;
    const size_t lds_scales = (size_t)(g.k / QB) * 32 * 4, lds_red = S > 1 ? (size_t)CW * S * 16 * 64 * 4 : 0;
    const size_t lds = lds_scales > lds_red ? lds_scales : lds_red;
    JHCHK(allow_lds((gemm_q8q4_tile_kernel<S, TILED, CW>), lds));
    hipLaunchKernelGGL((gemm_q8q4_tile_kernel<S, TILED, CW>), dim3(8 * mtiles * gg), dim3(S * CW * 64), lds, st, g, mtiles);
    HIPCHK(hipGetLastError());
    return JH_OK;
}

// gemm_q8q4_lds_kernel: one row tile x CW*CT column tiles per workgroup, A staged through LDS; K split over grid.y when the
// output alone does not fill the chip (partials in ws, summed in ascending K order by splitk_reduce_kernel)
// second pass of a split-K GEMM (partials summed in ascending K order): four columns per thread where the shapes allow (the
// one-float-per-thread form spent 19 us on the 2 x 14.8 MB of the BF16 gate|up GEMM at 129 rows: 64-bit divisions per element)
int launch_splitk_reduce(const float* ws, int S, int m, int n, int n0, float* c, int ldc, int roffset, const float* resid, hipStream_t st) {
    const int coff = n0 - roffset;
    if (n % 4 == 0 && ldc % 4 == 0 && coff % 4 == 0 && ((uintptr_t)ws | (uintptr_t)c | (uintptr_t)resid) % 16 == 0) {
        hipLaunchKernelGGL(splitk_reduce4_kernel, dim3((unsigned)((n / 4 + 255) / 256), (unsigned)m), dim3(256), 0, st, (const f32x4*)ws, S, m, n / 4,
                           (f32x4*)c, ldc / 4, coff / 4, (const f32x4*)resid);
    } else {
        const size_t tot = (size_t)m * n;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, ws, S, m, n, n0, c, ldc, roffset, resid);
    }
    HIPCHK(hipGetLastError());
    return JH_OK;
}

template <int CW, int CT, int S, bool PK = false>
int launch_gemm_q8q4_lds(const MfmaQ4Params& g, int mtiles, float* ws, size_t ws_bytes, hipStream_t st) {
    const int nblk = g.k / QB;
    const int cgroups = g.n / (32 * CW * CT), gg = (cgroups + 7) / 8;
    int Z = 1;
    const int z_env = opt_int("JH_GEMM_Z", 0);
    while (Z < 8 && (long long)mtiles * cgroups * Z < (long long)g_cu_count * 2 && nblk % (16 * S * Z) == 0 &&
           ws && (size_t)(2 * Z) * g.m * g.n * 4 <= ws_bytes) Z *= 2;
    if (z_env > 0 && nblk % (8 * S * z_env) == 0 && (z_env == 1 || (ws && (size_t)z_env * g.m * g.n * 4 <= ws_bytes))) Z = z_env;
    const int nbz = nblk / Z;
    size_t lds = (size_t)nbz * 128 + (size_t)S * 2 * 4 * 1024;
    const size_t red = S > 1 ? (size_t)CW * S * CT * 16 * 64 * 4 : 0;
    if (red > lds) lds = red;
    JHCHK(allow_lds((gemm_q8q4_lds_kernel<CW, CT, S, PK>), lds));
    hipLaunchKernelGGL((gemm_q8q4_lds_kernel<CW, CT, S, PK>), dim3(8 * mtiles * gg, Z), dim3(CW * S * 64), lds, st, g, mtiles, nbz, Z > 1 ? ws : nullptr);
    HIPCHK(hipGetLastError());
    if (Z > 1) JHCHK(launch_splitk_reduce((const float*)ws, Z, g.m, g.n, g.n0, g.c, g.ldc, g.roffset, g.resid, st));
    return JH_OK;
}

// tiled = both operands already in MFMA order (see gemm_q8q4_tile_kernel)
int launch_gemm_q8q4_mfma(const MfmaQ4Params& g, hipStream_t st, bool tiled, float* ws, size_t ws_bytes) {
    const int mt = (g.m + 31) / 32;
    const int nblk = g.k / QB;
    if (nblk % 8 == 0 && (size_t)nblk * 128 <= 150 * 1024 && (tiled || (g.lda % 16) == 0) && true) {
        // one 32x32 output tile per wave; split K over S waves until the chip has >= ~8 waves per CU; CW column tiles per
        // workgroup share the A tile through L1 (S*CW <= 8 waves: several workgroups per CU keep the CUs evenly loaded)
        // gemm_q8q4_lds_kernel (A through LDS, K split over the waves of a workgroup) for the GEMMs with enough work per column
        // group -- gate|up and down: 71 vs 78 us and 45 vs 56 us at M = 129, 96 vs 102 and 54 vs 67 at M = 256 -- the tile kernel
        // below for q|k|v and the o-projection (21 vs 30 us).  JH_GEMM_LDS = 0 / 1 forces one of them.
        const int lds_env = opt_int("JH_GEMM_LDS", -1);   // read per call: the tests flip it within one process
        const bool lds_auto = (long long)(g.n / 32) * nblk >= (long long)896 * 64;
        if (tiled && (lds_env > 0 || (lds_env < 0 && lds_auto))) {
            const int cw_l = opt_int("JH_GEMM_LDS_CW", 4), ct_l = opt_int("JH_GEMM_LDS_CT", 1), s_l = opt_int("JH_GEMM_LDS_S", 2);
            int CWL = cw_l, CTL = ct_l > 2 ? 2 : ct_l, SL = s_l;
            while (CTL > 1 && g.n % (32 * CWL * CTL)) CTL >>= 1;
            while (CWL > 1 && g.n % (32 * CWL * CTL)) CWL >>= 1;
            while (SL > 1 && nblk % (8 * SL)) SL >>= 1;
            while (SL > 1 && (size_t)nblk * 128 + (size_t)SL * 8192 > 150 * 1024) SL >>= 1;   // scale slice + the A rings of the SL K slices
            const bool lds_fits = (size_t)nblk * 128 + (size_t)SL * 8192 <= 150 * 1024;
            if (!lds_fits) CWL = -1;   // no instantiation below matches: the tile kernel takes it
            if (CWL == 4 && CTL == 1 && SL == 2 && opt_int("JH_GEMM_LDS_PK", 0)) return launch_gemm_q8q4_lds<4, 1, 2, true>(g, mt, ws, ws_bytes, st);
#define JH_LDS(CV, TV, SV) if (CWL == CV && CTL == TV && SL == SV) return launch_gemm_q8q4_lds<CV, TV, SV>(g, mt, ws, ws_bytes, st);
            JH_LDS(4, 1, 1) JH_LDS(4, 1, 2) JH_LDS(4, 1, 4) JH_LDS(2, 1, 2) JH_LDS(2, 1, 4) JH_LDS(2, 1, 8) JH_LDS(4, 2, 1) JH_LDS(4, 2, 2) JH_LDS(2, 2, 2)
            JH_LDS(1, 1, 4) JH_LDS(1, 1, 8) JH_LDS(2, 1, 1) JH_LDS(1, 1, 1) JH_LDS(1, 1, 2)
#undef JH_LDS
        }
        const long long tiles = (long long)mt * (g.n / 32);
        int S = 1;
        while (S < 8 && tiles * S < (long long)g_cu_count * 8 && nblk % (16 * S) == 0) S *= 2;   // nblk/S stays a multiple of 8
        const int cw_env = opt_int("JH_GEMM_CW", 0), s_env = opt_int("JH_GEMM_S", 0);
        if (s_env > 0 && nblk % (8 * s_env) == 0) S = s_env;
        int CW = 1;
        if (tiled) {
            CW = S <= 4 ? 4 : 2;                       // S*CW <= 16 waves
            if (cw_env > 0 && cw_env * S <= 16) CW = cw_env;
            while (CW > 1 && (g.n % (32 * CW)) != 0) CW >>= 1;
        }
#define JH_TILE(SV, CV) if (S == SV && CW == CV) return tiled ? launch_gemm_q8q4_tile<SV, true, CV>(g, mt, st) : launch_gemm_q8q4_tile<SV, false, 1>(g, mt, st);
        JH_TILE(1, 1) JH_TILE(2, 1) JH_TILE(4, 1) JH_TILE(8, 1) JH_TILE(1, 2) JH_TILE(2, 2) JH_TILE(4, 2) JH_TILE(1, 4) JH_TILE(2, 4) JH_TILE(8, 2) JH_TILE(1, 8) JH_TILE(4, 4)
#undef JH_TILE
        return set_err(JH_ERR_INVALID, "tile GEMM: no instantiation for this (S, CW)");
    }
    if (tiled) return set_err(JH_ERR_UNSUPPORTED, "tiled I8xQ4 GEMM needs K % 256 == 0");
    switch (mt) {
        case 1: return launch_gemm_q8q4_mfma_mt<1>(g, st);
        case 2: return launch_gemm_q8q4_mfma_mt<2>(g, st);
        case 3: case 4: return launch_gemm_q8q4_mfma_mt<4>(g, st);
        case 5: case 6: return launch_gemm_q8q4_mfma_mt<6>(g, st);
        // M > 192 with K % 256 != 0: the all-of-M-per-wave kernel would need 8 accumulator tiles and spills (156 VGPRs
        // to scratch when it existed); such shapes take the generic kernel instead
        default: return set_err(JH_ERR_UNSUPPORTED, "I8xQ4 MFMA GEMM: M > 192 needs K % 256 == 0");
    }
}

template <int MT>
int launch_gemm_bf16_mfma_mt(const MfmaGemmParams& g0, hipStream_t st) {
    MfmaGemmParams g = g0;
    const int tiles = g.n / 32;
    // 4 column tiles (waves) per workgroup share one staged A slice (A is 6x the W bytes per slice at M=192, so a
    // single-wave workgroup spends its time re-staging A); parallelism comes from splitting K over workgroup rows
    // (partials to the caller's workspace, splitk_reduce_kernel adds them in K order)
    const int waves = tiles >= 4 ? 4 : (tiles >= 2 ? 2 : 1);
    const int grid = (tiles + waves - 1) / waves;
    int S = 1;
    if (g.ws) {
        const int nslices = g.k / MG_KS;
        while (S < 16 && grid * S < g_cu_count * 3 && nslices % (2 * S) == 0 && nslices / (2 * S) >= 2 &&
               (size_t)(2 * S) * g.n <= (size_t)8 * 16384 && (g.n <= 8192 || 2 * S * g.m <= 640)) S *= 2;   // wide outputs: the reduce pass costs S*M*N*8 bytes
    }
    g.nsplit = S;
    const size_t lds = (size_t)2 * MT * 32 * MG_ASTRIDE;
    if (waves == 4) { JHCHK(allow_lds(gemm_bf16_mfma_kernel<MT, 4>, lds)); hipLaunchKernelGGL((gemm_bf16_mfma_kernel<MT, 4>), dim3(grid, S), dim3(256), lds, st, g); }
    else if (waves == 2) { JHCHK(allow_lds(gemm_bf16_mfma_kernel<MT, 2>, lds)); hipLaunchKernelGGL((gemm_bf16_mfma_kernel<MT, 2>), dim3(grid, S), dim3(128), lds, st, g); }
    else { JHCHK(allow_lds(gemm_bf16_mfma_kernel<MT, 1>, lds)); hipLaunchKernelGGL((gemm_bf16_mfma_kernel<MT, 1>), dim3(grid, S), dim3(64), lds, st, g); }
    HIPCHK(hipGetLastError());
    if (S > 1) JHCHK(launch_splitk_reduce((const float*)g.ws, S, g.m, g.n, g.n0, g.c, g.ldc, g.roffset, g.resid, st));
    return JH_OK;
}

int launch_gemm_bf16_mfma(const MfmaGemmParams& g, hipStream_t st) {
    const int mt = (g.m + 31) / 32;
    switch (mt) {
        case 1: return launch_gemm_bf16_mfma_mt<1>(g, st);
        case 2: return launch_gemm_bf16_mfma_mt<2>(g, st);
        case 3: case 4: return launch_gemm_bf16_mfma_mt<4>(g, st);
        case 5: case 6: return launch_gemm_bf16_mfma_mt<6>(g, st);
        default: return launch_gemm_bf16_mfma_mt<8>(g, st);
    }
}

template <int MT, int CWB>
int launch_gemm_bf16_tile_mc(MfmaBf16TileParams g, int S, hipStream_t st) {
    const int tiles = g.n / 32, grid = (tiles + CWB - 1) / CWB;
    g.nsplit = S;
    // A through LDS (gemm_bf16_lds_kernel) when the K range of a workgroup row divides into double chunks of 4 slices
    const int lds_env = opt_int("JH_BF16_LDS", 1);
    if (lds_env && (tiles % CWB) == 0 && ((g.k / 16 / S) % 8) == 0) {
        const size_t lds = (size_t)2 * MT * 4 * 1024;
        JHCHK(allow_lds((gemm_bf16_lds_kernel<MT, CWB>), lds));
        hipLaunchKernelGGL((gemm_bf16_lds_kernel<MT, CWB>), dim3(grid, S), dim3(CWB * 64), lds, st, g);
    } else {
        hipLaunchKernelGGL((gemm_bf16_tile_kernel<MT, CWB>), dim3(grid, S), dim3(CWB * 64), 0, st, g);
    }
    HIPCHK(hipGetLastError());
    if (S > 1) JHCHK(launch_splitk_reduce((const float*)g.ws, S, g.m, g.n, 0, g.c, g.ldc, 0, g.resid, st));
    return JH_OK;
}
// gemm_bf16_cw2_kernel: two column tiles per MFMA wave, A staged by a loader wave (jh_kernels.h).  128 columns per workgroup; K is
// split (partials + reduce pass) only until the launch covers the chip.
template <int MT>
int launch_gemm_bf16_cw2(MfmaBf16TileParams g, int S, hipStream_t st) {
    constexpr int PW = 4;                                    // 4 chunks x 4 slices x 2 KiB = 32 KiB of weights in flight per MFMA wave
    g.nsplit = S;
    const size_t lds = (size_t)2 * MT * 4 * 1024;
    const int knock = opt_int("JH_BF16_CW2", 1);            // 11 / 12: knock-outs of the A / the weight stream (tools/gemm_bench.py; results wrong)
    if (knock == 11 && MT == 5) {
        JHCHK(allow_lds((gemm_bf16_cw2_kernel<5, PW, 1>), lds));
        hipLaunchKernelGGL((gemm_bf16_cw2_kernel<5, PW, 1>), dim3(g.n / 128, S), dim3(BF16_CW2_WAVES * 64), lds, st, g);
    } else if (knock == 12 && MT == 5) {
        JHCHK(allow_lds((gemm_bf16_cw2_kernel<5, PW, 2>), lds));
        hipLaunchKernelGGL((gemm_bf16_cw2_kernel<5, PW, 2>), dim3(g.n / 128, S), dim3(BF16_CW2_WAVES * 64), lds, st, g);
    } else {
        JHCHK(allow_lds((gemm_bf16_cw2_kernel<MT, PW>), lds));
        hipLaunchKernelGGL((gemm_bf16_cw2_kernel<MT, PW>), dim3(g.n / 128, S), dim3(BF16_CW2_WAVES * 64), lds, st, g);
    }
    HIPCHK(hipGetLastError());
    if (S > 1) JHCHK(launch_splitk_reduce((const float*)g.ws, S, g.m, g.n, 0, g.c, g.ldc, 0, g.resid, st));
    return JH_OK;
}
// gemm_bf16_w8_kernel: 8 MFMA waves (4 column tiles x 2 K halves) + an LDS-DMA loader wave per workgroup; K additionally split over
// grid.y (partials + reduce pass) until the launch covers the chip
template <int MT>
int launch_gemm_bf16_w8(MfmaBf16TileParams g, int S, hipStream_t st) {
    g.nsplit = S;
    const size_t lds = (size_t)BF16_W8_NBUF * MT * 8 * 1024;   // NBUF buffers x 2 halves x MT x 4 slices x 1 KiB (the reduce area reuses MT x 16 KiB of it)
    JHCHK(allow_lds((gemm_bf16_w8_kernel<MT>), lds));
    hipLaunchKernelGGL((gemm_bf16_w8_kernel<MT>), dim3(g.n / 128, S), dim3(BF16_W8_WAVES * 64), lds, st, g);
    HIPCHK(hipGetLastError());
    if (S > 1) JHCHK(launch_splitk_reduce((const float*)g.ws, S, g.m, g.n, 0, g.c, g.ldc, 0, g.resid, st));
    return JH_OK;
}
// both operands in MFMA order (gemm_bf16_tile_kernel); n % 32 == 0, k % 16 == 0
int launch_gemm_bf16_tile(const MfmaBf16TileParams& g, hipStream_t st) {
    const int mt = (g.m + 31) / 32, tiles = g.n / 32, nks = g.k / 16;
    // prompt-sized M (2..6 row tiles) and whole 128-column groups: the 8-MFMA-wave kernel (round 6).  JH_BF16_W8=0 restores the
    // round-5 dispatch below for comparisons.
    if (opt_int("JH_BF16_W8", 1) && mt >= 2 && mt <= 6 && g.n % 128 == 0 && nks % 16 == 0) {
        int S = 1;
        auto fits = [&](int s2) { return nks % (16 * s2) == 0 && (s2 == 1 || (g.ws && (size_t)s2 * g.n <= (size_t)8 * 16384 && (g.n <= 8192 || s2 * g.m <= 512))); };
        while (S < 16 && fits(2 * S) && (g.n / 128) * 2 * S <= g_cu_count) S *= 2;   // one 9-wave workgroup per CU
        const int s_env3 = opt_int("JH_BF16_S", 0);
        if (s_env3 > 0 && fits(s_env3)) S = s_env3;
        if (fits(S)) {
            switch (mt) {
                case 2: return launch_gemm_bf16_w8<2>(g, S, st);
                case 3: return launch_gemm_bf16_w8<3>(g, S, st);
                case 4: return launch_gemm_bf16_w8<4>(g, S, st);
                case 5: return launch_gemm_bf16_w8<5>(g, S, st);
                default: return launch_gemm_bf16_w8<6>(g, S, st);
            }
        }
    }
    // prompt-sized M (2..6 row tiles), whole 128-column groups and enough of them to cover the chip WITHOUT splitting K (gate|up of an
    // 8B-class model: 224 groups): the loader-wave kernel, no reduce pass.  Everywhere else it measured equal or slower than the
    // LDS kernel below with its K split (profiles/r05c_*): JH_BF16_CW2=2 forces it (with a K split) for comparisons.
    const int cw2 = opt_int("JH_BF16_CW2", 1);
    if (cw2 && mt >= 2 && mt <= 6 && g.n % 128 == 0 && nks % 16 == 0 && (cw2 >= 2 || (g.n / 128) * 4 >= g_cu_count * 3)) {
        int S = 1;
        const int s_env2 = opt_int("JH_BF16_S", 0);
        if (g.ws && cw2 >= 2) {
            auto fits = [&](int s2) { return nks % (16 * s2) == 0 && (size_t)s2 * g.n <= (size_t)8 * 16384 && (g.n <= 8192 || s2 * g.m <= 512); };
            while (S < 16 && fits(2 * S) && (g.n / 128) * 2 * S <= g_cu_count) S *= 2;   // one workgroup per CU at most (3 waves of ~400 registers)
            if (s_env2 > 0 && fits(s_env2)) S = s_env2;
        }
        switch (mt) {
            case 2: return launch_gemm_bf16_cw2<2>(g, S, st);
            case 3: return launch_gemm_bf16_cw2<3>(g, S, st);
            case 4: return launch_gemm_bf16_cw2<4>(g, S, st);
            case 5: return launch_gemm_bf16_cw2<5>(g, S, st);
            default: return launch_gemm_bf16_cw2<6>(g, S, st);
        }
    }
    // waves per workgroup (column tiles sharing the A fragments through L1) vs workgroups: want >= ~2 workgroups per CU
    // before splitting K, because the split's reduce pass moves S*M*N*8 bytes
    const int cwb_env = opt_int("JH_BF16_CWB", 0), s_env = opt_int("JH_BF16_S", 0);
    int cwb = 8;
    while (cwb > 1 && ((tiles % cwb) != 0 || tiles / cwb < g_cu_count * 2)) cwb >>= 1;
    if (cwb < 4 && tiles % 4 == 0 && nks >= 512) cwb = 4;      // long K, few tiles: measured best (tools/gemm_bench.py)
    if (cwb < 2 && tiles % 2 == 0) cwb = 2;
    if (cwb_env > 0 && tiles % cwb_env == 0) cwb = cwb_env;
    int S = 1;
    const bool lds_kernel = opt_int("JH_BF16_LDS", 1) != 0 && nks % 8 == 0;
    if (lds_kernel && cwb_env <= 0) {
        // gemm_bf16_lds_kernel (tools/bf16_exp.sh sweeps, profiles/r02i_*): the waves of a workgroup share the staged A chunk, so
        // 4 column tiles per workgroup (8 when the A chunk is 7-8 row tiles); K split until the launch has a workgroup per CU and
        // either 1.5 per CU or <= 64 k slices per workgroup
        cwb = (mt >= 7 && tiles % 8 == 0) ? 8 : (tiles % 4 == 0 ? 4 : (tiles % 2 == 0 ? 2 : 1));
        if (g.ws) {
            auto fits = [&](int s2) {
                return nks % (8 * s2) == 0 && (size_t)s2 * g.n <= (size_t)8 * 16384 && (g.n <= 8192 || s2 * g.m <= 512);
            };
            while (S < 16 && fits(2 * S)) {
                const int wgs = (tiles / cwb) * S;
                if (wgs >= g_cu_count && (2 * wgs >= 3 * g_cu_count || nks / S <= 64)) break;
                S *= 2;
            }
        }
    } else if (g.ws) {
        while (S < 16 && (tiles / cwb) * S < g_cu_count * 2 && nks % (4 * S) == 0 && nks / (2 * S) >= 8 &&   // nks/S stays even
               (size_t)(2 * S) * g.n <= (size_t)8 * 16384 && (g.n <= 8192 || 2 * S * g.m <= 640)) S *= 2;
    }
    if (s_env > 0 && g.ws && nks % (2 * s_env) == 0 && (size_t)s_env * g.n <= (size_t)8 * 16384) S = s_env;
    if ((nks / S) % 2) return set_err(JH_ERR_UNSUPPORTED, "tiled BF16 GEMM needs K % 32 == 0");
#define JH_BT(MV) { if (cwb == 8) return launch_gemm_bf16_tile_mc<MV, 8>(g, S, st); if (cwb == 4) return launch_gemm_bf16_tile_mc<MV, 4>(g, S, st); \
                    if (cwb == 2) return launch_gemm_bf16_tile_mc<MV, 2>(g, S, st); return launch_gemm_bf16_tile_mc<MV, 1>(g, S, st); }
    switch (mt) {
        case 1: JH_BT(1)
        case 2: JH_BT(2)
        case 3: JH_BT(3)
        case 4: JH_BT(4)
        case 5: JH_BT(5)
        case 6: JH_BT(6)
        default: JH_BT(8)
    }
#undef JH_BT
}

