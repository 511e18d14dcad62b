// Winograd F(2x2,3x3) weight gradient on v_mfma_f32_32x32x2_f32 (gfx950).
//
// Same operator as conv_wgrad_wino_kernel (wmd_conv_bwd.hip): autograd of ConvBlock / Conv3x3 + upsample + concat + pad with
// respect to the filter (the reference gets it from torch.autograd, KITTI/trainer.py:211, NYUv2/train.py:327),
//     dU_xi[co, ci] = sum_tiles dM_xi[tile, co] * V_xi[tile, ci],   dM = A dY A^T,   V = B^T d B,   dg = G^T dU G,
// same partial layout ([split][16][Cout*Cin] + bias sums; wgrad_wino_reduce_kernel finishes it), same LDS-DMA gather of dz
// rows and of the padded / upsampled / concatenated patch rows.  What changes is the work around the matrix pipe, as in
// conv_wino32_kernel: a wave owns a 32 x 32 (out x in channel) block of dU for 8 of the 16 transformed positions (8 x 16
// accumulator registers), the K index of the 32x32x2 MFMA walks two tiles of a row.  A lane holds one channel (l & 31) and
// one tile (l >> 5) for both operands: 2 + 6 ds_read_b64 and 15 packed adds (round 5; ~33 scalar ones before) feed 8 MFMAs of 64 cycles -- no weight fragments at
// all on this side -- against 36 reads + 76 adds per 32 MFMAs of 32 cycles in the 16x16x4 form, which runs at 25-36 % of the
// pipe.  Block = WCO x WCI slabs x the two position halves; the halves of a slab pair share a SIMD.
// For the 2x-upsampled operand B^T d B vanishes on transformed row 2 / column 2 (see conv_wino32_kernel): waves whose 32 input
// channels all belong to it issue 5 / 4 MFMAs per K-step instead of 8, and the reduce kernel sees zeros there.
#include <algorithm>
#include <type_traits>
#include <utility>
#include "wmd_conv_common.h"

namespace wmd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <class F, int... I>
__device__ __forceinline__ void wg_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void wg_static_for(F&& f) {
    wg_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// position ownership of the two halves: identical to conv_wino32_kernel (w32_owns)
__host__ __device__ constexpr bool wg32_owns(int hf, int xi) {
    const int r = xi / 4, c = xi % 4;
    const bool h0 = r == 0 || ((r == 1 || r == 2) && c < 2);
    return hf == 0 ? h0 : !h0;
}
__host__ __device__ constexpr bool wg32_up_reaches(int xi) { return xi / 4 != 2 && xi % 4 != 2; }
// the p-th position (ascending) a half works on; UP: only the positions an upsampled operand reaches
__host__ __device__ constexpr int wg32_nth(int hf, bool up, int p) {
    int n = 0;
    for (int xi = 0; xi < 16; ++xi)
        if (wg32_owns(hf, xi) && (!up || wg32_up_reaches(xi))) {
            if (n == p) return xi;
            ++n;
        }
    return -1;
}
__host__ __device__ constexpr int wg32_slot(int hf, int xi) {
    int n = 0;
    for (int x = 0; x < xi; ++x) n += wg32_owns(hf, x) ? 1 : 0;
    return n;
}

template <int TH, int TW, int WCO, int WCI>
struct Wg32Tile {
    static constexpr int NW = WCO * WCI * 2, NT = NW * 64;
    static constexpr int PH = TH + 2, PW = TW + 2;
    static constexpr int NPIX = TH * TW, NPATCH = PH * PW;
    static constexpr int COT = WCO * 32, CIT = WCI * 32;
    static constexpr int TXW = TW / 2, NT2 = (TH / 2) * TXW, KS = NT2 / 2;
    // row strides: even with an odd half, so the 32 lanes of an 8-byte read group (one channel each) fall on 32 distinct
    // even banks of the 64
    static constexpr int SA = ((NPIX + 1) / 4) * 4 + 2, SB = ((NPATCH + 1) / 4) * 4 + 2;
    static constexpr int PA = (NPIX + 63) / 64, PB = (NPATCH + 63) / 64;
    static constexpr int BUF = COT * SA + CIT * SB;
    static constexpr int LDS_FLOATS = 2 * BUF;
    static constexpr int ROWS_A = COT / NW, ROWS_B = CIT / NW;      // rows a wave stages per pixel tile
    static constexpr int NPIECES = ROWS_A * PA + ROWS_B * PB;
    static_assert(TH % 2 == 0 && TW % 4 == 0, "whole 2x2 tiles; the two tiles of a K-step stay in one tile row");
    static_assert(NT2 % 2 == 0, "whole K-steps");
    static_assert(COT % NW == 0 && CIT % NW == 0, "rows are dealt evenly to the waves");
    static_assert(SA >= NPIX && SB >= NPATCH && SA % 4 == 2 && SB % 4 == 2, "row strides");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS");
};

// (the 4-wave blocks -- one slab pair -- carry 56 staging pieces per wave and tile next to their 128 accumulators: 299 registers, one
//  wave per SIMD; asking for two made hipcc warn "desired occupancy 2, final 1" for code that is the same either way)
template <int TH, int TW, int WCO, int WCI>
__global__ __launch_bounds__(WCO* WCI * 128, (WCO * WCI >= 4 ? 2 : 1)) void conv_wgrad_wino32_kernel(const WgradKArgs a) {
    using T = Wg32Tile<TH, TW, WCO, WCI>;
    constexpr int PW = T::PW, SA = T::SA, SB = T::SB, NPIX = T::NPIX, NW = T::NW;
    constexpr unsigned kOOB = 0x80000000u;
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // waves w and w + WCO*WCI are the two halves of slab pair w: dealt cyclically to the SIMDs they share one (8-wave blocks)
    const int pair = wave % (WCO * WCI), hf = wave / (WCO * WCI);
    const int wco = pair % WCO, wci = pair / WCO;
    const int co0 = blockIdx.y * T::COT, ci0 = blockIdx.x * T::CIT;
    const int split = blockIdx.z;
    const int H = a.H, W = a.W;
    const size_t plane = (size_t)H * W, plane1 = (size_t)a.H1 * a.W1;
    const unsigned pbz = (unsigned)(plane * 4), pb1 = (unsigned)(plane1 * 4);

    const int per = (a.ntiles + a.nsplit - 1) / a.nsplit;
    const int t_begin = split * per, t_end = min(t_begin + per, a.ntiles);

    // this lane's fixed positions inside a tile: output pixels (dz rows) and patch positions
    int azy[T::PA], azx[T::PA], apy[T::PB], apx[T::PB];
#pragma unroll
    for (int i = 0; i < T::PA; ++i) {
        const int p = i * 64 + lane;
        azy[i] = p < NPIX ? p / TW : -(1 << 20);
        azx[i] = p % TW;
    }
#pragma unroll
    for (int i = 0; i < T::PB; ++i) {
        const int p = i * 64 + lane;
        apy[i] = p < T::NPATCH ? p / PW - 1 : -(1 << 20);
        apx[i] = p % PW - 1;
    }
    auto fold = [&](int g, int n, int& ok) {   // padded coordinate -> source coordinate, branch-free
        const int refl = g < 0 ? -g : (g >= n ? 2 * n - 2 - g : g);
        const int clam = min(max(g, 0), n - 1);
        ok &= (int)(a.pad_mode != WMD_PAD_ZERO) | (int)(g == clam);
        const int r = a.pad_mode == WMD_PAD_REFLECT ? refl : clam;
        return min(max(r, 0), n - 1);
    };

    // per-tile staging state: descriptors of the tile's image and this lane's byte offsets
    struct TileSrc {
        __amdgpu_buffer_rsrc_t rz, r1, r2;
        unsigned oz[T::PA], o1[T::PB], o2[T::PB];
    };
    auto tile_src = [&](int tile) {
        TileSrc ts;
        int t = tile;
        const int tx = t % a.tiles_x;
        t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        const int b = t / a.tiles_y;
        const int y0 = ty * TH, x0 = tx * TW;
        ts.rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dz + (size_t)b * a.Cout * plane), 0, (int)(a.Cout * plane * 4), 0x00020000);
        ts.r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x1 + (size_t)b * a.C1 * plane1), 0, (int)(a.C1 * plane1 * 4), 0x00020000);
        ts.r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x2 ? a.x2 + (size_t)b * a.C2 * plane : a.x1), 0,
                                                  (int)(a.C2 * plane * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < T::PA; ++i) {
            const int oy = y0 + azy[i], ox = x0 + azx[i];
            ts.oz[i] = (oy >= 0 && oy < H && ox < W) ? (unsigned)(oy * W + ox) * 4u : kOOB;
        }
#pragma unroll
        for (int i = 0; i < T::PB; ++i) {
            const int gy0 = y0 + apy[i], gx0 = x0 + apx[i];
            int ok = (int)(gy0 >= -1) & (int)(gy0 <= H) & (int)(gx0 <= W);
            const int gy = fold(gy0, H, ok), gx = fold(gx0, W, ok);
            ts.o2[i] = ok ? (unsigned)(gy * W + gx) * 4u : kOOB;
            ts.o1[i] = ok ? (unsigned)((a.up1 == 2 ? gy >> 1 : gy) * a.W1 + (a.up1 == 2 ? gx >> 1 : gx)) * 4u : kOOB;
        }
        return ts;
    };
    // piece q of this wave's share of a tile: rows wave, wave + NW, ... of the dz block, then of the patch block.
    // A row beyond Cout / Cin is staged from the last valid channel instead of being zeroed per lane: it only feeds accumulator rows /
    // columns (and bias sums) that are never stored.  `wave_t` is the wave index laundered once per tile: left to itself the compiler
    // hoists every piece's scalar channel offset out of the tile loop, runs out of scalar registers and fetches them back with
    // v_readlane -- vector-ALU instructions in the MFMA stream, four per piece.
    // (Round 5 also built per-tile LDS tables of the folded patch rows / columns, filled by PH + PW threads two tiles ahead, in place of
    //  every lane folding its own coordinates in tile_src: slower, 0.237 -> 0.246 ms per step for <2,40,2,2> -- the table reads sit
    //  between a tile's barrier and its first LDS-DMA issue.)
    auto stage_piece = [&](const TileSrc& ts, float* bufA, int q, int wave_t) {
        if (q < T::ROWS_A * T::PA) {
            const int j = q / T::PA, i = q % T::PA;
            const int c = wave_t + j * NW, co = co0 + c;
            const unsigned so = (unsigned)min(co, a.Cout - 1) * pbz;
            if ((i + 1) * 64 <= NPIX || i * 64 + lane < NPIX)   // partial last piece: exec-masked
                lds_dma4(ts.rz, (lds_ptr_t)(bufA + c * SA + i * 64), ts.oz[i], so);
        } else {
            const int q2 = q - T::ROWS_A * T::PA;
            const int j = q2 / T::PB, i = q2 % T::PB;
            const int c = wave_t + j * NW, ci = min(ci0 + c, a.Cin - 1);
            const bool from1 = ci < a.C1;   // wave-uniform
            const unsigned so = from1 ? (unsigned)ci * pb1 : (unsigned)(ci - a.C1) * pbz;
            // (a wave-uniform branch, not a select: selecting between the two descriptors of the struct by address would
            // turn the whole struct into an LDS-resident array)
            if ((i + 1) * 64 <= T::NPATCH || i * 64 + lane < T::NPATCH) {
                lds_ptr_t d = (lds_ptr_t)(bufA + T::COT * SA + c * SB + i * 64);
                if (from1) lds_dma4(ts.r1, d, ts.o1[i], so);
                else lds_dma4(ts.r2, d, ts.o2[i], so);
            }
        }
    };

    // lane-parallel bias sums of the dz rows this wave stages (reduced over the lanes once, at the end)
    float bs[T::ROWS_A];
#pragma unroll
    for (int j = 0; j < T::ROWS_A; ++j) bs[j] = 0.f;

    if (t_begin < t_end) {
        const TileSrc ts0 = tile_src(t_begin);
        wg_static_for<T::NPIECES>([&](auto qc) { stage_piece(ts0, lds, decltype(qc)::value, wave); });
    }
    __syncthreads();

    // the slab's 32 input channels all come from the 2x-upsampled x1: transformed row 2 / column 2 of V are identically zero
    const bool up_slab = a.up1 == 2 && ci0 + (wci + 1) * 32 <= a.C1;

    auto run = [&](auto hf_tag) {
        constexpr int HF = decltype(hf_tag)::value;
        f32x16 acc[8];
#pragma unroll
        for (int o = 0; o < 8; ++o)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[o][r] = 0.f;

        // One pixel tile: KS K-steps (two tiles each) of NP MFMAs.  The operand reads of K-step s + 1 and the next tile's
        // DMA pieces are issued in the shadow of the MFMAs of K-step s (pinned with sched_barrier).
        auto tile_body = [&](int tile, auto up_tag, auto next_tag) {
            constexpr bool UP = decltype(up_tag)::value;
            constexpr bool NEXT = decltype(next_tag)::value;
            const int buf = (tile - t_begin) & 1;
            const float* ldsA = lds + buf * T::BUF;
            const float* ldsB = ldsA + T::COT * SA;
            float* nbuf = lds + (buf ^ 1) * T::BUF;
            TileSrc tsn;
            if constexpr (NEXT) tsn = tile_src(tile + 1);
            int wave_t = wave;
            asm volatile("" : "+s"(wave_t));   // (see stage_piece)
            // lane = (channel l & 31, tile l >> 5 of the K-step): the tile's column offset lives in the base pointers
            const float* pa = ldsA + (wco * 32 + (lane & 31)) * SA + 2 * (lane >> 5);
            const float* pb = ldsB + (wci * 32 + (lane & 31)) * SB + 2 * (lane >> 5);
            // Operand transforms in packed fp32 on register pairs along the pixel column (wmd_conv_common.h, w32_pk_*): fp32 MFMA
            // and the vector ALU are the same hardware, and this loop used to issue ~33 scalar VALU instructions per 8 MFMAs.
            // dM = A dY A^T's minus signs on positions (r,3), r < 3, and (3,c), c < 3, move into V's column mix, which produces
            // either sign in the same instruction: every product is the same bits as before.  15 packed instructions per K-step.
            // Volatile LDS-space reads: one ds_read_b64 with an immediate offset each (no v_add_u32 for a ds_read2 base), and only
            // the three patch rows this half's positions use.
            typedef const volatile __attribute__((address_space(3))) f32x2* lds_cv2_t;
            typedef const volatile __attribute__((address_space(3))) float* lds_cv1_t;
            const lds_cv1_t pa3 = (lds_cv1_t)pa, pb3 = (lds_cv1_t)pb;
            f32x2 dzp[2][2], xp[2][8];   // [slot][row] of dY, [slot][2 row + half] of the patch
            auto fetch = [&](int ks) {
                const int t0 = ks * 2, trow = t0 / T::TXW, tcol = t0 % T::TXW, sl = ks & 1;
#pragma unroll
                for (int r = 0; r < 2; ++r) dzp[sl][r] = *reinterpret_cast<lds_cv2_t>(pa3 + (2 * trow + r) * TW + 2 * tcol);
#pragma unroll
                for (int e = 2 * HF; e < 2 * HF + 6; ++e)
                    xp[sl][e] = *reinterpret_cast<lds_cv2_t>(pb3 + (2 * trow + (e >> 1)) * PW + 2 * tcol + 2 * (e & 1));
            };
            constexpr int NPIECES = NEXT ? T::NPIECES : 0;
            constexpr int NP = UP ? (HF == 0 ? 5 : 4) : 8;
            constexpr int S = T::KS * NP;
            fetch(0);
            wg_static_for<T::KS>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value;
                constexpr int sl = ks & 1;
                if constexpr (ks + 1 < T::KS) fetch(ks + 1);
                // dM = A dY A^T,  A = [[1,0],[1,1],[1,-1],[0,-1]], up to the signs V takes over
                float dm[16];   // (elements of the pair results: no instruction of their own)
                f32x2 vp[8];
                {
                    const f32x2 d0 = dzp[sl][0], d1 = dzp[sl][1];
                    const f32x2 sm = w32_pk_addn(d0, d1), df = w32_pk_subn(d0, d1);   // rows 1, 2 of A dY
                    if constexpr (HF == 0) {   // row 0 (all columns), (1,0) (1,1) (2,0) (2,1)
                        const f32x2 pm = w32_pk_pm(d0);   // (d00 + d01, d00 - d01)
                        dm[0] = d0[0], dm[1] = pm[0], dm[2] = pm[1];
                        dm[3] = d0[1];                    // + d01: V carries the minus
                        dm[4] = sm[0], dm[5] = sm[0] + sm[1];
                        dm[8] = df[0], dm[9] = df[0] + df[1];
                    } else {                   // (1,2) (1,3) (2,2) (2,3), row 3 (all columns)
                        const f32x2 pm = w32_pk_pm(d1);   // (d10 + d11, d10 - d11)
                        dm[6] = sm[0] - sm[1], dm[7] = sm[1];     // (1,3): + s1, V carries the minus
                        dm[10] = df[0] - df[1], dm[11] = df[1];   // (2,3) likewise
                        dm[12] = d1[0], dm[13] = pm[0], dm[14] = pm[1];   // row 3, columns 0-2: V carries the minus
                        dm[15] = d1[1];
                    }
                }
                {   // V = B^T d B: rows mix as whole pairs (rows 0-2 / 1-3), then the column mix of each owned row pair
                    f32x2 tp[8];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if constexpr (HF == 0) tp[0 + h] = w32_pk_sub(xp[sl][0 + h], xp[sl][4 + h]);
                        tp[2 + h] = w32_pk_add(xp[sl][2 + h], xp[sl][4 + h]);
                        tp[4 + h] = w32_pk_sub(xp[sl][4 + h], xp[sl][2 + h]);
                        if constexpr (HF == 1) tp[6 + h] = w32_pk_sub(xp[sl][2 + h], xp[sl][6 + h]);
                    }
                    if constexpr (HF == 0) {
                        vp[0] = w32_pk_lo(tp[0], tp[1]);
                        vp[1] = w32_pk_hi_f(tp[0], tp[1]);      // (0,2), -(0,3)
                        vp[2] = w32_pk_lo(tp[2], tp[3]);
                        vp[4] = w32_pk_lo(tp[4], tp[5]);
                    } else {
                        vp[3] = w32_pk_hi_f(tp[2], tp[3]);      // (1,2), -(1,3)
                        vp[5] = w32_pk_hi_f(tp[4], tp[5]);      // (2,2), -(2,3)
                        vp[6] = w32_pk_lo_ff(tp[6], tp[7]);     // -(3,0), -(3,1)
                        vp[7] = w32_pk_hi_fl(tp[6], tp[7]);     // -(3,2), (3,3)
                    }
                }
                wg_static_for<NP>([&](auto pc) {
                    constexpr int p = decltype(pc)::value;
                    constexpr int xi = wg32_nth(HF, UP, p);
                    if constexpr (NEXT) {
                        // the next tile's pieces q with q * S / NPIECES == s, s = ks * NP + p
                        constexpr int s = ks * NP + p;
                        constexpr int q0 = (s * NPIECES + S - 1) / S, q1 = ((s + 1) * NPIECES + S - 1) / S;
                        constexpr int qb = q0 < NPIECES ? q0 : NPIECES, qe = q1 < NPIECES ? q1 : NPIECES;
                        wg_static_for<qe - qb>([&](auto qc) { stage_piece(tsn, nbuf, qb + decltype(qc)::value, wave_t); });
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc[wg32_slot(HF, xi)] = __builtin_amdgcn_mfma_f32_32x32x2f32(dm[xi], vp[xi >> 1][xi & 1], acc[wg32_slot(HF, xi)], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            if (a.want_bias && blockIdx.x == 0) {
                // rows this wave staged; one value per lane and piece
#pragma unroll
                for (int j = 0; j < T::ROWS_A; ++j)
#pragma unroll
                    for (int i = 0; i < T::PA; ++i)
                        if ((i + 1) * 64 <= NPIX || i * 64 + lane < NPIX) bs[j] += ldsA[(wave + j * NW) * SA + i * 64 + lane];
            }
            __syncthreads();
        };
        int tile = t_begin;
        if (up_slab) {
            for (; tile + 1 < t_end; ++tile) tile_body(tile, std::true_type{}, std::true_type{});
            if (tile < t_end) tile_body(tile, std::true_type{}, std::false_type{});
        } else {
            for (; tile + 1 < t_end; ++tile) tile_body(tile, std::false_type{}, std::true_type{});
            if (tile < t_end) tile_body(tile, std::false_type{}, std::false_type{});
        }

        // partial [split][16][Cout*Cin] (+ [Cout] bias sums): D row = out channel (reg&3) + 8 (reg>>2) + 4 (lane>>5),
        // D column = input channel lane & 31.  Positions an upsampled slab never touches are written as the zeros they are.
        const size_t nwc = (size_t)a.Cout * a.Cin;
        float* out = a.partial + (size_t)split * (16 * nwc + a.Cout);
        const int ci = ci0 + wci * 32 + (lane & 31);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            if (!wg32_owns(HF, xi)) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < a.Cout && ci < a.Cin) out[(size_t)xi * nwc + (size_t)co * a.Cin + ci] = acc[wg32_slot(HF, xi)][r];
            }
        }
    };
    if (hf == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});

    if (a.want_bias && blockIdx.x == 0) {
        const size_t nwc = (size_t)a.Cout * a.Cin;
        float* out = a.partial + (size_t)split * (16 * nwc + a.Cout);
#pragma unroll
        for (int j = 0; j < T::ROWS_A; ++j) {
            float s = bs[j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            const int co = co0 + wave + j * NW;
            if (lane == 0 && co < a.Cout) out[16 * nwc + co] = s;
        }
    }
}

template <int TH, int TW, int WCO, int WCI>
void launch_wgrad_wino32(const WgradKArgs& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((conv_wgrad_wino32_kernel<TH, TW, WCO, WCI>), grid, dim3(WCO * WCI * 128), 0, s, a);
}

#define WMD_WG32_INST(TH, TW, WCO, WCI) template void launch_wgrad_wino32<TH, TW, WCO, WCI>(const WgradKArgs&, dim3, hipStream_t);
#include "wmd_conv_wgrad32_table.inc"
#undef WMD_WG32_INST

}  // namespace wmd
