// Two chained GEMMs per pixel for the wide levels of the inference wavelet heads (C = 64, 128, 256;
// KITTI/networks/decoders/depth_decoder.py:108-136):
//
//   mid_s = LeakyReLU(W1_s x + b1_s)                    s in {+,-},  1x1, C -> C
//   t_s   = W3'_s mid_s                                 the 3x3 regrouped as 27 tap-partial 1x1 outputs (row co*9+tap)
//
// (wmd_head_shiftsum_fwd then gathers the nine shifted taps, applies bias / sigmoid / combine and the Haar synthesis.)
// Same contract, same weight images and the same 54- (81-) plane output as the FUSE form of conv_fwd_kernel it replaces;
// what changes is the orientation of the products.  There the pixels were the MFMA rows: the first product left a lane with
// 4 pixels of ONE channel, the wrong shape for the second product's operand, so `mid` made a round trip through LDS, the
// out-channel rows were split over the waves and only two of them ran the second GEMM.  Here the WEIGHTS are the A operand:
//
//   GEMM 1   D[row = out channel][col = pixel] += W1[row][k] * x[k][col]
//            lane l ends up with mid[16m + 4(l>>4) + i][pixel l&15], i = 0..3, for every row tile m of its slice
//   GEMM 2   that register IS the B operand of the second product for the K-step whose four K-lanes are the channels
//            16m + 4g + i (g = 0..3): the sum over channels is order-free, so the K-steps simply run over (m, i) and the A
//            operand is W3'[tap row][16m + 4g + i] -- a gather from the ordinary packed image (fragment 4m + g, lane
//            16 i + (l & 15)).  No LDS traffic, no barrier, every wave takes part.
//
// A wave owns 16*NT pixels and C/RS mid channels of one side (blockIdx.y); RS > 1 (C = 256, whose level has only 5 760
// pixels) splits the channels -- i.e. the second product's reduction -- over RS waves, whose 32 x 16 partial tiles meet in
// LDS.  W1 and x are staged through LDS in K-chunks (register-prefetched double buffer, one barrier per chunk): the PG pixel
// groups of a block share the weight chunk.  x rows are padded to a stride == 16 (mod 32) floats: the four K-lanes of a
// fragment read hit disjoint banks.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include "wmd_internal.h"

#ifndef WMD_CHAIN_PD256
#define WMD_CHAIN_PD256 4   // chunks requested ahead, in registers (development: -DWMD_CHAIN_PD256=1 = rounds 2-5)
#endif
#ifndef WMD_CHAIN_PD128
#define WMD_CHAIN_PD128 1
#endif

namespace wmd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <class F, int... I>
__device__ __forceinline__ void hc_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void hc_static_for(F&& f) {
    hc_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains vmcnt -- i.e. every global load in
// flight, including the chunks requested AHEAD: with it a deeper prefetch changes nothing (measured: PD 1 / 2 / 4 = 33.7 / 32.9 /
// 33.9 us) and a lone block still takes 16 x 0.94 us for 16 x 0.33 us of MFMA work.  The staged chunks travel global -> registers
// (the compiler waits for exactly the registers it consumes) -> LDS, so lgkmcnt(0) is all the barrier needs.
__device__ __forceinline__ void hc_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

struct HeadChainArgs {
    const float* x;      // [B, C, plane]
    const float* wp1;    // packed [2C, C, 1, 1]
    const float* bias1;  // [2C] or null
    const float* wp2;    // two packed [27, C, 1, 1] images
    float* t;            // [B, t_planes, plane]
    int plane, tiles, t_planes;
    int C;               // (read by the merged launch only: which body a block range runs)
    float slope;
    const uint8_t* run_mask;   // optional [B, plane]: a block whose pixel run holds no set byte returns (wmd_head_fused_args.run_mask)
    // the coarsest level's low-pass chain (C = 256 only; wmd_head_fused_args.ll_wp1) rides in the blocks of side 0: W1_ll
    // [C/4, C] = four more row tiles of the first product over the SAME staged x chunks (one per channel slice), W3'_ll = one
    // [27 -> 32, C/4] image with nine rows in use -> planes 54..62
    const float* ll_wp1;
    const float* ll_bias1;
    const float* ll_wp2;
    // training forward (wmd_head_fused_args.mid_out): the LeakyReLU outputs also go to mid_out [B, mid_ct, plane], this side's
    // channels from mid_off[side] (the low-pass chain's from mid_off[2])
    float* mid_out;
    int mid_ct, mid_off[3];
};

template <int C, int RS, int PG, int NT, int KC>
struct HeadChainTile {
    static constexpr int NW = RS * PG, NTH = NW * 64;
    static constexpr int PXB = PG * NT * 16;     // pixels of a block
    static constexpr int MTS = C / 16;           // row tiles of one side
    static constexpr int MT = MTS / RS;          // ... of one wave
    static constexpr int KS = C / 4;             // K-steps of GEMM 1
    static constexpr int NCH = KS / KC;          // staged chunks
    static constexpr int XS = PXB % 32 == 0 ? PXB + 16 : PXB + 32;   // x row stride in LDS: == 16 (mod 32) for PXB = 32, 48, 64, 128
    static constexpr int WF = MTS * KC * 64;     // floats of a weight chunk
    static constexpr int XF = KC * 4 * XS;       // floats of an x chunk
    static constexpr int NBUF = NCH > 1 ? 2 : 1;
    static constexpr int WV = (WF / 4 + NTH - 1) / NTH;          // float4 per thread and weight chunk (last one may be partial: 12-wave blocks)
    static constexpr int XV = (KC * PXB + NTH - 1) / NTH;        // float4 per thread and x chunk (last one may be partial)
    static constexpr int RED = RS > 1 ? NW * 3 * NT * 256 : 0;   // floats of the cross-wave reduction (aliases the chunks; third tile: the low-pass rows)
    static constexpr int WFL = (C == 256 && RS == 4) ? RS * KC * 64 : 0;   // the low-pass chain's part of a weight chunk (C = 256 only): one row tile per channel slice
    static constexpr int STAGE = NBUF * (WF + WFL + XF);
    static constexpr int LDS_FLOATS = STAGE > RED ? STAGE : RED;
    static_assert(C % 16 == 0 && MTS % RS == 0 && KS % KC == 0, "whole tiles");
    static_assert(XS % 32 == 16 && PXB % 4 == 0, "bank-conflict-free x rows");
};

// (LLX: the block also stages the low-pass chain's row tiles of the chunk -- WFL floats, one 16-byte piece for the first WFL/4
//  threads.  The staged pieces are native vectors (f32x4), not HIP's float4 struct: with register SETS indexed by the prefetch
//  distance the struct copies became memcpys into a stack object that the optimiser no longer promoted -- scratch.)
template <class T, int KC, bool LLX>
__device__ __forceinline__ void chain_fetch(f32x4 (&wreg)[T::WV], f32x4 (&xreg)[T::XV], f32x4& lreg, const float* __restrict__ w1,
                                            const float* __restrict__ wl, const float* __restrict__ xb, int c, int tid, int pix0, int plane) {
    if constexpr (LLX) {
        static_assert(T::WFL > 0 && T::WFL / 4 <= T::NTH, "one 16-byte piece per thread covers the low-pass part of a chunk");
        const int f = tid * 4, m = f / (KC * 64), rem = f % (KC * 64);
        if (f < T::WFL) lreg = *reinterpret_cast<const f32x4*>(wl + ((size_t)m * T::KS + c * KC) * 64 + rem);
    }
#pragma unroll
    for (int v = 0; v < T::WV; ++v) {
        const int f = (v * T::NTH + tid) * 4;                     // float index inside the chunk: [row tile][KC*64]
        const int m = f / (KC * 64), rem = f % (KC * 64);
        f32x4 wv = {0.f, 0.f, 0.f, 0.f};
        if (T::WF % (4 * T::NTH) == 0 || f < T::WF) wv = *reinterpret_cast<const f32x4*>(w1 + ((size_t)m * T::KS + c * KC) * 64 + rem);
        wreg[v] = wv;
    }
#pragma unroll
    for (int v = 0; v < T::XV; ++v) {
        const int q = v * T::NTH + tid;                           // piece index: [channel of the chunk][PXB / 4]
        const int chl = q / (T::PXB / 4), px = pix0 + (q % (T::PXB / 4)) * 4;
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (q < KC * T::PXB && px < plane)                        // plane % 4 == 0 (host check): a piece never straddles the end
            val = *reinterpret_cast<const f32x4*>(xb + (size_t)(c * KC * 4 + chl) * plane + px);
        xreg[v] = val;
    }
}

template <class T, int KC, bool LLX>
__device__ __forceinline__ void chain_commit(const f32x4 (&wreg)[T::WV], const f32x4 (&xreg)[T::XV], const f32x4& lreg, float* ws, int tid) {
    float* xs = ws + T::WF + T::WFL;
    if constexpr (LLX) {
        if (tid * 4 < T::WFL) *reinterpret_cast<f32x4*>(ws + T::WF + tid * 4) = lreg;
    }
#pragma unroll
    for (int v = 0; v < T::WV; ++v)
        if (T::WF % (4 * T::NTH) == 0 || (v * T::NTH + tid) * 4 < T::WF) *reinterpret_cast<f32x4*>(ws + (v * T::NTH + tid) * 4) = wreg[v];
#pragma unroll
    for (int v = 0; v < T::XV; ++v) {
        const int q = v * T::NTH + tid;
        if (q < KC * T::PXB) *reinterpret_cast<f32x4*>(xs + (q / (T::PXB / 4)) * T::XS + (q % (T::PXB / 4)) * 4) = xreg[v];
    }
}

// One block's work.  LLX = the block also carries the low-pass chain (side 0 of the C = 256 launch): compiled as a second body
// so that every size stays a compile-time constant of its body.
template <int C, int RS, int PG, int NT, int KC, bool LLX, int PD>
__device__ __forceinline__ void head_chain_body(const HeadChainArgs& a, float* lds, const int bx, const int side) {
    using T = HeadChainTile<C, RS, PG, NT, KC>;
    constexpr int MT = T::MT, KS = T::KS, XS = T::XS, PXB = T::PXB;
    constexpr int MTX = MT + (LLX ? 1 : 0);   // row tiles of the first product per wave
    static_assert(!LLX || (C / 4) / 16 == RS, "the low-pass chain: C/4 mid channels = one row tile per channel slice");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = wave % RS, pg = wave / RS;   // channel slice, pixel group
    const int g = lane >> 4, lc = lane & 15;
    const int tile = bx % a.tiles, b = bx / a.tiles;   // (bx, side): blockIdx.x / .y of a level's own launch, or dealt by the merged one
    const int pix0 = tile * PXB, plane = a.plane;
    if (a.run_mask) {   // block-sparse levels: nothing downstream reads the tap-partials of a run without an active pixel
        static_assert(PXB <= T::NTH, "one mask byte per thread covers the block's pixel run");
        __shared__ int run_any[T::NW];
        const bool mine = tid < PXB && pix0 + tid < plane && a.run_mask[(size_t)b * plane + pix0 + tid] != 0;
        const bool wave_any = __builtin_amdgcn_ballot_w64(mine) != 0;
        if (lane == 0) run_any[wave] = wave_any ? 1 : 0;
        __syncthreads();
        int all = 0;
#pragma unroll
        for (int w = 0; w < T::NW; ++w) all |= run_any[w];
        if (all == 0) return;
    }
    const float* xb = a.x + (size_t)b * C * plane;
    const float* w1 = a.wp1 + (size_t)side * T::MTS * KS * 64;   // fragments [row tile][K-step][64 lanes]
    const float* w2 = a.wp2 + (size_t)side * 2 * KS * 64;        // fragments [tap-row tile 0..1][K-step][64 lanes]

    // ---- staging: chunk c = K-steps [c*KC, (c+1)*KC) of every row tile of W1_side, and channels [c*KC*4, +KC*4) of x ----
    // Round 6: the chunks are requested PD deep IN REGISTERS (chunk k PD iterations before its K-steps run, moved to the LDS double
    // buffer one iteration before), behind barriers that do not drain the loads in flight (hc_barrier).
    static_assert(PD >= 1 && (T::NCH % PD == 0 || T::NCH == 1), "the chunk loop is unrolled by the prefetch distance");
    f32x4 wreg[PD][T::WV], xreg[PD][T::XV], lreg[PD];
#pragma unroll
    for (int d = 0; d < PD; ++d) lreg[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int BUF = T::WF + T::WFL + T::XF;

    // ---- GEMM 1: acc[m][n] = rows 16(r*MT+m) .. +15 of W1_side x, pixels of group pg*NT + n (LLX: acc[MT] = row tile r of
    //      the low-pass chain's W1) -------------------------------------------------------------------------------------
    f32x4 acc[MTX][NT];
#pragma unroll
    for (int m = 0; m < MTX; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    chain_fetch<T, KC, LLX>(wreg[0], xreg[0], lreg[0], w1, a.ll_wp1, xb, 0, tid, pix0, plane);
    chain_commit<T, KC, LLX>(wreg[0], xreg[0], lreg[0], lds, tid);
    hc_static_for<PD>([&](auto kc) __attribute__((always_inline)) {      // chunks 1 .. PD -> register sets 1 .. PD-1, 0
        constexpr int k = decltype(kc)::value + 1;
        if (k < T::NCH) chain_fetch<T, KC, LLX>(wreg[k % PD], xreg[k % PD], lreg[k % PD], w1, a.ll_wp1, xb, k, tid, pix0, plane);
    });
    hc_barrier();
    for (int c0 = 0; c0 < T::NCH; c0 += PD) {
        hc_static_for<PD>([&](auto dc) __attribute__((always_inline)) {
            constexpr int d = decltype(dc)::value, dn = (d + 1) % PD;   // register sets of chunk c (free: committed an iteration ago) and c + 1
            const int c = c0 + d;                                       // (NCH % PD == 0: c < NCH)
            const float* ws = lds + (c & (T::NBUF - 1)) * BUF;
            const float* xs = ws + T::WF + T::WFL;
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) {
                float xf[NT], wf[MTX];
#pragma unroll
                for (int n = 0; n < NT; ++n) xf[n] = xs[(kk * 4 + g) * XS + (pg * NT + n) * 16 + lc];
#pragma unroll
                for (int m = 0; m < MT; ++m) wf[m] = ws[((r * MT + m) * KC + kk) * 64 + lane];
                if constexpr (LLX) wf[MT] = ws[T::WF + (r * KC + kk) * 64 + lane];
#pragma unroll
                for (int m = 0; m < MTX; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[m], xf[n], acc[m][n], 0, 0, 0);
            }
            if (c + 1 < T::NCH) {
                chain_commit<T, KC, LLX>(wreg[dn], xreg[dn], lreg[dn], lds + ((c + 1) & (T::NBUF - 1)) * BUF, tid);
                if (c + 1 + PD < T::NCH)      // ... and set dn is free for chunk c + 1 + PD
                    chain_fetch<T, KC, LLX>(wreg[dn], xreg[dn], lreg[dn], w1, a.ll_wp1, xb, c + 1 + PD, tid, pix0, plane);
                hc_barrier();
            }
        });
    }

    // ---- bias + LeakyReLU in registers, then GEMM 2 straight from them ---------------------------------------------------
    f32x4 acc2[2][NT];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc2[j][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mg = r * MT + m;                                   // row tile within the side
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias1) {
            const float4 b4 = *reinterpret_cast<const float4*>(a.bias1 + side * C + mg * 16 + g * 4);
            bv[0] = b4.x, bv[1] = b4.y, bv[2] = b4.z, bv[3] = b4.w;
        }
        float w2f[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) w2f[j][i] = w2[((size_t)j * KS + 4 * mg + g) * 64 + i * 16 + lc];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float mid = act_apply(acc[m][n][i] + bv[i], WMD_ACT_LEAKY, a.slope);
                if (a.mid_out) {   // lane = (channel 16 mg + 4 g + i, pixel lc of group n): 64-byte runs of four channel rows
                    const int px = pix0 + (pg * NT + n) * 16 + lc;
                    if (px < plane) a.mid_out[((size_t)b * a.mid_ct + a.mid_off[side] + mg * 16 + g * 4 + i) * plane + px] = mid;
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) acc2[j][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2f[j][i], mid, acc2[j][n], 0, 0, 0);
            }
        }
    }

    // ---- the low-pass chain's second product (LLX): mid_ll = LeakyReLU(acc[MT] + b1_ll) is the B operand of the K-steps
    //      (r, i) of W3'_ll (16 K-steps: fragment 4 r + g of tap-row tile 0; rows 9..15 of the tile are zero) -----------------
    f32x4 acc2l[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc2l[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (LLX) {
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.ll_bias1) {
            const float4 b4 = *reinterpret_cast<const float4*>(a.ll_bias1 + r * 16 + g * 4);
            bv[0] = b4.x, bv[1] = b4.y, bv[2] = b4.z, bv[3] = b4.w;
        }
        float w2l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w2l[i] = a.ll_wp2[((size_t)4 * r + g) * 64 + i * 16 + lc];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float mid = act_apply(acc[MT][n][i] + bv[i], WMD_ACT_LEAKY, a.slope);
                if (a.mid_out) {
                    const int px = pix0 + (pg * NT + n) * 16 + lc;
                    if (px < plane) a.mid_out[((size_t)b * a.mid_ct + a.mid_off[2] + r * 16 + g * 4 + i) * plane + px] = mid;
                }
                acc2l[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2l[i], mid, acc2l[n], 0, 0, 0);
            }
    }

    // ---- the channel slices of a pixel group meet in LDS (RS > 1), fixed order r = 0 .. RS-1 ------------------------------
    if constexpr (RS > 1) {
        constexpr int NJ = LLX ? 3 : 2;   // reduction tiles per pixel group: the two tap-row tiles (+ the low-pass rows)
        __syncthreads();   // every wave is done with the staged chunks
        float* red = lds + (size_t)wave * 3 * NT * 256;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const f32x4 v = j < 2 ? acc2[j < 2 ? j : 0][n] : acc2l[n];
                *reinterpret_cast<float4*>(red + ((j * NT + n) * 64 + lane) * 4) = make_float4(v[0], v[1], v[2], v[3]);
            }
        __syncthreads();
        if (r != 0) return;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                float4 s = *reinterpret_cast<const float4*>(lds + (size_t)(pg * RS) * 3 * NT * 256 + ((j * NT + n) * 64 + lane) * 4);
#pragma unroll
                for (int rr = 1; rr < RS; ++rr) {
                    const float4 p = *reinterpret_cast<const float4*>(lds + (size_t)(pg * RS + rr) * 3 * NT * 256 + ((j * NT + n) * 64 + lane) * 4);
                    s.x += p.x, s.y += p.y, s.z += p.z, s.w += p.w;
                }
                if (j < 2) acc2[j < 2 ? j : 0][n] = f32x4{s.x, s.y, s.z, s.w};
                else acc2l[n] = f32x4{s.x, s.y, s.z, s.w};
            }
    }

    // ---- store: lane = (tap rows 16 j + 4 g + i, pixel lc of group n) -----------------------------------------------------
    float* tb = a.t + ((size_t)b * a.t_planes + side * 27) * plane;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int px = pix0 + (pg * NT + n) * 16 + lc;
        if (px >= plane) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = j * 16 + g * 4 + i;
                if (row < 27) tb[(size_t)row * plane + px] = acc2[j][n][i];
            }
        if constexpr (LLX) {   // rows 0..8 of the low-pass tile -> planes 54..62
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = g * 4 + i;
                if (row < 9) tb[(size_t)(54 + row) * plane + px] = acc2l[n][i];
            }
        }
    }
}

// (second launch bound = waves per SIMD: the 8- and 12-wave C = 256 blocks must not be throttled below 3 per SIMD by the staged
//  register sets)
template <int C, int RS, int PG, int NT, int KC, int PD, bool LLC = false>
__global__ __launch_bounds__(RS* PG * 64, (RS * PG >= 8 ? 3 : 1)) void head_chain_kernel(const HeadChainArgs a) {
    using T = HeadChainTile<C, RS, PG, NT, KC>;
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS];
    if constexpr (LLC) {
        if (blockIdx.y == 0) {
            head_chain_body<C, RS, PG, NT, KC, true, PD>(a, lds, blockIdx.x, 0);
            return;
        }
    }
    head_chain_body<C, RS, PG, NT, KC, false, PD>(a, lds, blockIdx.x, blockIdx.y);
}

// ---- round 6: the chained GEMMs of SEVERAL levels in ONE launch -------------------------------------------------------------
// The coarse levels are too small to balance 256 CUs alone (5 760 / 23 040 / 92 160 pixels at config 2: block counts of 360 / 720 /
// 1440 leave CUs idle through whole block times, profiles/r06_notes.md section 6), and a replayed graph runs its nodes one after
// the other.  The heads of a level read only that level's trunk activation, so the dense decoder may postpone them: after
// `upconv(2,1)` ONE launch runs the first stage of levels 4, 3 and 2 as consecutive block ranges (largest blocks first) and the
// machine balances them against each other.  256-thread blocks throughout: the C = 256 level as 16-pixel blocks (RS = 4 waves
// share a pixel group's channels).  Same per-pixel arithmetic as the per-level launches: bit-identical planes.
struct HeadChainMulti {
    HeadChainArgs lv[3];
    int first[4];   // block ranges: level k owns [first[k], first[k + 1]) = (its B * tiles blocks of side 0, then of side 1)
    int n;
};
constexpr int head_chain_multi_lds() {
    int m = HeadChainTile<256, 4, 1, 1, 4>::LDS_FLOATS;
    if (HeadChainTile<128, 1, 4, 1, 8>::LDS_FLOATS > m) m = HeadChainTile<128, 1, 4, 1, 8>::LDS_FLOATS;
    if (HeadChainTile<64, 1, 4, 2, 16>::LDS_FLOATS > m) m = HeadChainTile<64, 1, 4, 2, 16>::LDS_FLOATS;
    return m;
}
__global__ __launch_bounds__(256, 3) void head_chain_multi_kernel(const HeadChainMulti m) {
    __shared__ __attribute__((aligned(16))) float lds[head_chain_multi_lds()];
    int k = 0;
    while (k + 1 < m.n && (int)blockIdx.x >= m.first[k + 1]) ++k;
    const HeadChainArgs& a = m.lv[k];
    const int v = (int)blockIdx.x - m.first[k], per_side = (m.first[k + 1] - m.first[k]) >> 1;
    const int side = v >= per_side ? 1 : 0, bx = v - side * per_side;
    if (a.C == 256) {
        if (side == 0 && a.ll_wp1) head_chain_body<256, 4, 1, 1, 4, true, WMD_CHAIN_PD256>(a, lds, bx, 0);
        else head_chain_body<256, 4, 1, 1, 4, false, WMD_CHAIN_PD256>(a, lds, bx, side);
    } else if (a.C == 128) {
        head_chain_body<128, 1, 4, 1, 8, false, WMD_CHAIN_PD128>(a, lds, bx, side);
    } else {
        head_chain_body<64, 1, 4, 2, 16, false, 1>(a, lds, bx, side);
    }
}

template <int C, int RS, int PG, int NT, int KC, int PD>
static void launch_chain(const HeadChainArgs& a, int B, hipStream_t s) {
    using T = HeadChainTile<C, RS, PG, NT, KC>;
    HeadChainArgs k = a;
    k.tiles = (a.plane + T::PXB - 1) / T::PXB;
    if constexpr (C == 256 && RS == 4) {
        if (a.ll_wp1) {
            hipLaunchKernelGGL((head_chain_kernel<C, RS, PG, NT, KC, PD, true>), dim3((unsigned)(B * k.tiles), 2), dim3(T::NTH), 0, s, k);
            return;
        }
    }
    hipLaunchKernelGGL((head_chain_kernel<C, RS, PG, NT, KC, PD>), dim3((unsigned)(B * k.tiles), 2), dim3(T::NTH), 0, s, k);
}

// -> true when the chained form took the launch (C = 64 / 128 / 256, image planes of a multiple of 4 pixels; WMD_HEAD_CHAIN=0
// keeps the FUSE form of conv_fwd_kernel)
static HeadChainArgs head_chain_args(const wmd_head_fused_args* g, int t_planes, long plane, bool& with_ll) {
    HeadChainArgs a;
    a.x = g->x;
    a.wp1 = g->wp1;
    a.bias1 = g->bias1;
    a.wp2 = g->wp2;
    a.t = g->t;
    a.plane = (int)plane;
    a.tiles = 0;
    a.t_planes = t_planes;
    a.slope = g->slope;
    a.run_mask = g->run_mask;
    with_ll = g->ll_wp1 && g->ll_wp2 && g->C == 256 && t_planes == 81 && !g->run_mask;
    a.ll_wp1 = with_ll ? g->ll_wp1 : nullptr;
    a.ll_bias1 = with_ll ? g->ll_bias1 : nullptr;
    a.ll_wp2 = with_ll ? g->ll_wp2 : nullptr;
    a.mid_out = g->mid_out;
    a.mid_ct = g->mid_ct;
    a.mid_off[0] = g->mid_off_p;
    a.mid_off[1] = g->mid_off_n;
    a.mid_off[2] = g->mid_off_ll;
    a.C = g->C;
    return a;
}

int head_chain_launch(const wmd_head_fused_args* g, int t_planes, hipStream_t s) {
    static const bool on = [] {
        const char* e = getenv("WMD_HEAD_CHAIN");
        return !(e && atoi(e) == 0);
    }();
    const long plane = (long)g->H * g->W;
    if (!on || (plane & 3) || plane > (1L << 28) || (g->C != 64 && g->C != 128 && g->C != 256)) return 0;
    bool with_ll = false;
    HeadChainArgs a = head_chain_args(g, t_planes, plane, with_ll);
    const double pix = (double)g->B * plane;
    ProfScope prof("head_chain_kernel", 2.0 * pix * (2.0 * g->C * g->C + 54.0 * g->C), 4.0 * pix * (g->C + 54), s);
    // (pixel-tile / wave-count / chunk variants -- 64 to 256 pixels, 2 to 16 waves, channel split 1 / 2 / 4 / 8 -- all measured
    //  within +-2 us of these)
    if (g->C == 64)
        launch_chain<64, 1, 4, 2, 16, 1>(a, g->B, s);    // 128 pixels x all 64 channels per wave quartet, one chunk
    else if (g->C == 128)
        launch_chain<128, 1, 4, 1, 8, WMD_CHAIN_PD128>(a, g->B, s);    // 64 pixels, 4 chunks of 32 channels
    else {
        // 32 pixels (8 waves: 4 share a pixel group's 256 channels), 16 chunks -- or 48 pixels (12 waves) when that brings the launch down
        // to one block per CU: the level has so few pixels that whole blocks per CU is what its time is made of (config 2, batch 12:
        // 360 blocks = two on 104 CUs, one on 152 -> 240 blocks, one each; profiles/r06_notes.md section 6)
        static const int pg_force = [] {
            const char* e = getenv("WMD_HEAD_CHAIN_PG256");
            return e ? atoi(e) : 0;
        }();
        const long b32 = (long)g->B * ((plane + 31) / 32) * 2, b48 = (long)g->B * ((plane + 47) / 48) * 2;
        const long cost32 = ((b32 + kNumCU - 1) / kNumCU) * 2, cost48 = ((b48 + kNumCU - 1) / kNumCU) * 3;
        if (pg_force == 3 || (pg_force == 0 && cost48 < cost32)) launch_chain<256, 4, 3, 1, 4, WMD_CHAIN_PD256>(a, g->B, s);
        else launch_chain<256, 4, 2, 1, 4, WMD_CHAIN_PD256>(a, g->B, s);
    }
    return with_ll ? 2 : 1;
}

// -> levels taken (n) or 0: the merged launch needs every level to be a chained width with planes of a multiple of 4 pixels, no
// run mask and no training outputs (WMD_HEAD_CHAIN_MULTI=0 off)
int head_chain_multi_launch(const wmd_head_fused_args* g, int n, hipStream_t s) {
    static const bool on = [] {
        const char* e = getenv("WMD_HEAD_CHAIN_MULTI");
        const char* c = getenv("WMD_HEAD_CHAIN");
        return !(e && atoi(e) == 0) && !(c && atoi(c) == 0);
    }();
    if (!on || n < 1 || n > 3) return 0;
    HeadChainMulti m;
    m.n = n;
    m.first[0] = 0;
    double flops = 0, bytes = 0;
    for (int k = 0; k < n; ++k) {
        const long plane = (long)g[k].H * g[k].W;
        const int t_planes = g[k].t_planes ? g[k].t_planes : 54;
        if ((plane & 3) || plane > (1L << 28) || (g[k].C != 64 && g[k].C != 128 && g[k].C != 256) || g[k].run_mask || g[k].mid_out || g[k].chain != 0)
            return 0;
        if (g[k].ll_wp1 && (g[k].C != 256 || t_planes != 81 || !g[k].ll_wp2)) return 0;
        bool with_ll = false;
        m.lv[k] = head_chain_args(&g[k], t_planes, plane, with_ll);
        const int pxb = g[k].C == 256 ? 16 : (g[k].C == 128 ? 64 : 128);
        m.lv[k].tiles = (int)((plane + pxb - 1) / pxb);
        const long blocks = 2L * g[k].B * m.lv[k].tiles;
        if (m.first[k] + blocks > (1L << 30)) return 0;
        m.first[k + 1] = m.first[k] + (int)blocks;
        const double pix = (double)g[k].B * plane;
        flops += 2.0 * pix * (2.0 * g[k].C * g[k].C + 54.0 * g[k].C);
        bytes += 4.0 * pix * (g[k].C + 54);
    }
    ProfScope prof("head_chain_multi_kernel", flops, bytes, s);
    hipLaunchKernelGGL(head_chain_multi_kernel, dim3((unsigned)m.first[n]), dim3(256), 0, s, m);
    return n;
}

}  // namespace wmd
