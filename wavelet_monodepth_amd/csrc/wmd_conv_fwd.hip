// Dense decoder convolutions on gfx950: implicit GEMM on v_mfma_f32_16x16x4_f32.
//
//   out[b,co,y,x] = act( bias[co] + sum_{ci,ky,kx} W[co,ci,ky,kx] * P[b,ci,y+ky-1,x+kx-1] )
//
// where P is the *virtual* padded input: channels [0,C1) come from x1 (optionally nearest-upsampled
// x2), channels [C1,C1+C2) from the skip tensor x2, and the 1-pixel border follows the reference's
// ReflectionPad2d / ReplicationPad2d / ZeroPad2d.  Neither the upsample, the concat nor the pad is
// ever materialised (reference: KITTI/layers.py:120-173,233-236; depth_decoder.py:145-150;
// NYUv2/networks/layers.py:11-32,57-67).
//
// GEMM view: M = Cout, N = pixels of a TH x TW tile of one image, K = Cin*k*k.
//   B (pixels)   : a CK-channel halo patch gathered by LDS-DMA (buffer_load_dword ... lds) into a position-linear
//                  LDS image, double-buffered, one barrier per CK channels; every lane reads its B element with
//                  ds_read_b32 at a compile-time offset (tap and channel are immediates).
//   A (weights)  : the block's slice of the fragment-ordered global image (wmd_conv_pack_weights), copied into LDS
//                  by 16-byte LDS-DMA with the same chunk; one ds_read_b32 per 16co x 4ci x tap fragment.
//   D            : MR x NR accumulator fragments (16x16) per wave; lane (l&15) = out-channel, (l>>4)*4+r = pixel
//                  (MFMA roles swapped, see below) -> 16-byte stores.
// fp32 MFMA issues once per 32 cycles per SIMD; one A read + NR B reads feed MR*NR MFMAs, so the matrix pipe is the
// only unit that can saturate: the kernel is MFMA-bound by construction (measured 0.62-0.73 MFMA-busy).
// Split-K (coarse levels) keeps a second, HBM-bound reduce launch: finishing the tile in the last-arriving block
// needs agent-scope release/acquire fences, which write back / invalidate the XCD's whole L2 on gfx950 -- measured
// 2-6x SLOWER than the two-launch form (every other block loses its weight and halo lines), so it was dropped.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "wmd_conv_common.h"

namespace wmd {

template <int TH, int TW, int MR, int NR, int WM, int WN, int CK, int TAPS, int NBUF = 2>
struct ConvTile {
    static constexpr int NT = WM * WN * 64;
    static constexpr int HALO = (TAPS == 9) ? 1 : 0;
    static constexpr int PH = TH + 2 * HALO;
    static constexpr int PW = TW + 2 * HALO;
    static constexpr int NPOSITIONS = PH * PW;
    // per-channel LDS stride: >= NPOSITIONS and == 16 (mod 32) so that the two k-lanes of a
    // 32-lane ds_read_b32 group (lanes 0-15 / 16-31) fall on disjoint bank halves
    static constexpr int PS = ((NPOSITIONS - 16 + 31) / 32) * 32 + 16;
    static constexpr int NPOS = (NPOSITIONS + NT - 1) / NT;
    static constexpr int NPIX = TH * TW;
    // weight tile of one chunk: (WM*MR) out-channel tiles x (CK/4) K-steps x TAPS fragments of 64 floats, in the
    // packed image's own order (a straight copy of WM*MR contiguous global runs)
    static constexpr int A_RUN = (CK / 4) * TAPS * 64;
    static constexpr int A_FLOATS = WM * MR * A_RUN;
    static constexpr int B_FLOATS = CK * PS;
    static constexpr int NAV = (A_FLOATS / 4 + NT - 1) / NT;  // 16-byte weight pieces per thread and chunk
    static constexpr int LDS_FLOATS = NBUF * (B_FLOATS + A_FLOATS);  // NBUF = 1: the whole reduction is one chunk
    static_assert(WN * NR * 16 >= NPIX, "tile has more pixels than MFMA columns");
    static_assert(CK % 4 == 0, "CK must be a whole number of 4-channel K-steps");
    static_assert(A_FLOATS % 4 == 0, "the weight tile is moved in 16-byte pieces (a partial last wave is exec-masked)");
    static_assert(TW % 4 == 0, "four consecutive pixels of an accumulator row must not straddle image rows");
};

// (Round 3's PRE instantiations -- the encoder edge: act(v * scale[c] + shift[c]) applied to every operand between its LDS
//  read and its MFMA -- were measured slower than activating the map in a pass of its own and running the tuned kernel
//  (R18: 75.4 vs 66.9 us, R50: 432.9 vs 170.5 us) and are gone; layers.DeferredActivation activates explicitly.)
template <int TH, int TW, int MR, int NR, int WM, int WN, int CK, int TAPS, bool FUSE = false, int NBUF = 2>
__global__ __launch_bounds__(WM* WN * 64) void conv_fwd_kernel(const ConvKArgs a) {
    using T = ConvTile<TH, TW, MR, NR, WM, WN, CK, TAPS, NBUF>;
    constexpr int NT = T::NT, HALO = T::HALO, PW = T::PW, PS = T::PS, NPOS = T::NPOS;
    constexpr int KSTEPS = CK / 4;

    constexpr int MID_FLOATS = FUSE ? WM * MR * 16 * T::PS : 0;
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS > MID_FLOATS ? T::LDS_FLOATS : MID_FLOATS];
    float* ldsA = lds + NBUF * T::B_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM;
    const int wn = wave / WM;

    // Work item = (pixel tile, out-channel slab), slab fastest.  Workgroups are dealt to the 8 XCDs round-robin and every
    // XCD has its own L2: giving an XCD one contiguous run of the item sequence puts the slabs of a tile -- which gather
    // the SAME input patch -- on the same XCD back to back, so only the first of them fetches it from HBM (a 2-D grid
    // launched all tiles of slab 0 before any tile of slab 1: Cout = 64 / 128 layers with 32-channel blocks moved 1.95x
    // their algorithmic bytes), and neighbouring tiles share halo rows and 128-byte lines in that L2 as before.
    int t, by;
    if (a.cob > 0) {
        const int item = xcd_contiguous(blockIdx.x, gridDim.x);
        t = item / a.cob;
        by = item - t * a.cob;
    } else {
        t = xcd_contiguous(blockIdx.x, gridDim.x);
        by = blockIdx.y;
    }
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int b = t / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int ks = blockIdx.z;
    const int H = a.H, W = a.W;

    // ---- staging geometry: each thread owns NPOS patch positions for every channel ----------
    // Staging is LDS-DMA (buffer_load ... lds): a wavefront instruction gathers 64 arbitrary global dwords
    // (descriptor base + wave-uniform SGPR channel offset + 32-bit per-lane byte offset) straight into 64
    // CONSECUTIVE LDS dwords — the patch is position-linear, so no VGPR round trip, no ds_write, no select.
    // A position that must read zero (zero padding, tile overhang, dgrad extension, channel tail) gets an
    // out-of-range byte offset: the descriptor's range check makes the load return 0.
    constexpr unsigned kOOB = 0x80000000u;  // >= any num_records (< 2^31) and cannot wrap when an SGPR offset is added
    unsigned ob1[NPOS], ob2[NPOS];
#pragma unroll
    for (int i = 0; i < NPOS; ++i) {
        const int p = tid + i * NT;
        const int py = p / PW, px = p % PW;
        int gy = y0 + py - HALO, gx = x0 + px - HALO;
        bool ok = p < T::NPOSITIONS;
        if (HALO) {
            ok = pad_coord(gy, H, a.pad_mode) && ok;
            ok = pad_coord(gx, W, a.pad_mode) && ok;
        }
        ok = ok && gy < H && gx < W && gy >= 0 && gx >= 0;  // tile overhang (H % TH != 0)
        gy = min(max(gy, 0), H - 1);
        gx = min(max(gx, 0), W - 1);
        ob2[i] = ok ? (unsigned)(gy * W + gx) * 4u : kOOB;
        int sy = gy - a.shift1, sx = gx - a.shift1;
        if (a.up1 == 2) {
            sy = gy >> 1;
            sx = gx >> 1;
        }
        const bool ok1 = ok && sy >= 0 && sx >= 0 && sy < a.H1 && sx < a.W1;
        ob1[i] = ok1 ? (unsigned)(sy * a.W1 + sx) * 4u : kOOB;
    }

    const size_t plane1 = (size_t)a.H1 * a.W1, plane2 = (size_t)H * W;
    const float* x1b = a.x1 + (size_t)b * a.C1 * plane1;
    const float* x2b = a.x2 ? a.x2 + (size_t)b * a.C2 * plane2 : a.x1;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x1b), 0, (int)(a.C1 * plane1 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x2b), 0, (int)(a.C2 * plane2 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.wp), 0, (int)((size_t)a.ncot * a.nci4 * TAPS * 64 * 4), 0x00020000);
    const unsigned pb1 = (unsigned)(plane1 * 4), pb2 = (unsigned)(plane2 * 4);

    // A (weights): the packed image keeps, for every out-channel tile, the chunk's (K-step, tap) fragments
    // contiguous, so the block's weight tile is WM*MR straight runs of A_RUN floats, DMA'd as 16-byte pieces.
    // (Loading the fragments from global in every wave costs ~30 % of the MFMA rate: tools/probes/mfma_probe.hip.)
    unsigned aoff[T::NAV];
#pragma unroll
    for (int v = 0; v < T::NAV; ++v) {
        const int e = tid + v * NT;  // 16-byte piece index inside the block's weight tile
        const int run = (e * 4) / T::A_RUN, rem = (e * 4) % T::A_RUN;
        const int cot = min(by * WM * MR + run, a.ncot - 1);
        aoff[v] = (unsigned)(((size_t)cot * a.nci4 * TAPS * 64 + rem) * 4);
    }

    // DMA of one chunk (patch + weights) into LDS buffer `buf`, as NPIECES wave-instructions per wave; completion =
    // vmcnt(0) + barrier.  `q` is a compile-time constant wherever this is called (fully unrolled loops).
    constexpr int NPB = CK * NPOS, NPIECES = NPB + T::NAV;
    auto stage_piece = [&](int chunk, int buf, int q) {
        if (q < NPB) {
            const int j = q / NPOS, i = q % NPOS;
            const int ci = chunk * CK + j;  // wave-uniform
            const bool from_x1 = ci < a.C1;
            const bool chan_ok = ci < a.Cin;
            const unsigned soff = from_x1 ? (unsigned)ci * pb1 : (unsigned)max(ci - a.C1, 0) * pb2;
            if (NPOS * NT == T::NPOSITIONS || tid + i * NT < T::NPOSITIONS) {  // partial last wave: exec-masked
                const unsigned vo = chan_ok ? (from_x1 ? ob1[i] : ob2[i]) : kOOB;
                lds_ptr_t d = (lds_ptr_t)(lds + buf * T::B_FLOATS + wave * 64 + j * PS + i * NT);  // + lane*4 bytes by the hardware
                if (from_x1) lds_dma4(r1, d, vo, soff);
                else lds_dma4(r2, d, vo, soff);
            }
        } else {
            const int v = q - NPB;
            const unsigned soffA = (unsigned)chunk * (unsigned)(T::A_RUN * 4);  // K-steps of a chunk are contiguous
            if (T::NAV * NT * 4 == T::A_FLOATS || (tid + v * NT) * 4 < T::A_FLOATS)  // partial last wave: exec-masked
                lds_dma16(rw, (lds_ptr_t)(ldsA + buf * T::A_FLOATS + wave * 256 + v * NT * 4), aoff[v], soffA);
        }
    };

    // ---- operand fragments: both from LDS at immediate offsets ------------------------------------
    // MFMA roles are swapped w.r.t. the GEMM view: the 16 PIXELS are the row operand and the 16 OUT CHANNELS
    // the column operand, so a lane's 4 accumulator registers are 4 consecutive pixels of ONE channel ->
    // 16-byte stores in the epilogue.
    int boff[NR];
#pragma unroll
    for (int n = 0; n < NR; ++n) {
        int q = (wn * NR + n) * 16 + (lane & 15);
        q = min(q, T::NPIX - 1);
        boff[n] = (q / TW) * PW + (q % TW) + (lane >> 4) * PS;
    }
    const int a_lane = (wm * MR) * T::A_RUN + lane;

    f32x4 acc[MR][NR];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NR; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int c_begin = ks * a.chunks_per_split;
    const int c_end = min(c_begin + a.chunks_per_split, a.nchunks);

    if (c_begin < c_end) {
#pragma unroll
        for (int q = 0; q < NPIECES; ++q) stage_piece(c_begin, 0, q);
    }
    // epilogue operands requested now: a global load at the start of the epilogue is an exposed round trip per block
    float bias_v[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const int co = FUSE ? (int)(by * (WM * MR * 16)) + (wm * MR + m) * 16 + (lane & 15)
                            : ((by * WM + wm) * MR + m) * 16 + (lane & 15);
        bias_v[m] = (a.bias && (FUSE || co < a.Cout)) ? a.bias[co] : 0.f;
    }
    __syncthreads();  // (hipcc drains the DMA with vmcnt(0) ahead of the barrier)

    // One chunk of the reduction.  Software pipeline over its S = KSTEPS*TAPS (K-step, tap) groups: the fragments of
    // group s+D are requested before the MFMAs of group s are issued, so a ds_read has D groups (>= 160 MFMA cycles
    // each for NR = 5) to land; with PREFETCH the next chunk's DMA pieces are dealt out one or two per group over
    // the first two thirds of the chunk, in the shadow of the MFMAs, instead of as one burst that stalls the wave
    // (an LDS-DMA piece costs 60-180 issue cycles, MI355X_MICROARCH.md).  sched_barrier pins that order (left
    // alone, the scheduler sinks the reads next to their uses and hoists the DMA into one block).
    auto chunk_body = [&](int c, auto prefetch) {
        constexpr bool PREFETCH = decltype(prefetch)::value;
        constexpr int S = KSTEPS * TAPS, D = S >= 3 ? 2 : 1, RS = D + 1;
        constexpr int SP = (S * 2) / 3 > 0 ? (S * 2) / 3 : 1;
        const int buf = (c - c_begin) & 1;
        const float* bsrc = lds + buf * T::B_FLOATS;
        const float* asrc = ldsA + buf * T::A_FLOATS + a_lane;
        float pf[RS][NR], wf[RS][MR];
        auto fetch = [&](int s) {
            const int kk = s / TAPS, tp = s % TAPS, slot = s % RS;
            const int ky = (TAPS == 9) ? tp / 3 : 0, kx = (TAPS == 9) ? tp % 3 : 0;
#pragma unroll
            for (int m = 0; m < MR; ++m) wf[slot][m] = asrc[m * T::A_RUN + (kk * TAPS + tp) * 64];
#pragma unroll
            for (int n = 0; n < NR; ++n) pf[slot][n] = bsrc[boff[n] + kk * 4 * PS + ky * PW + kx];
        };
#pragma unroll
        for (int s = 0; s < D && s < S; ++s) fetch(s);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if constexpr (PREFETCH) {
#pragma unroll
                for (int q = 0; q < NPIECES; ++q)
                    if (q * SP / NPIECES == s) stage_piece(c + 1, buf ^ 1, q);
            }
            if (s + D < S) fetch(s + D);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int n = 0; n < NR; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf[s % RS][n], wf[s % RS][m], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();  // next buffer landed (vmcnt(0) precedes the barrier), this one is released
    };
    if constexpr (NBUF == 2) {
        for (int c = c_begin; c + 1 < c_end; ++c) chunk_body(c, std::true_type{});
        if (c_begin < c_end) chunk_body(c_end - 1, std::false_type{});
    } else {
        for (int c = c_begin; c < c_end; ++c) chunk_body(c, std::false_type{});
    }

    if constexpr (FUSE) {
        // ---- fused wavelet head: mid = LeakyReLU(acc + b1) stays on chip; t = W3' * mid --------------------
        // The block holds ALL CO_T mid channels of one side (blockIdx.y) for its pixels.  W3' is the 3x3 filter
        // regrouped as 27 "tap-partial" 1x1 outputs (row co*9+tap); the spatial shift-sum over the 9 taps
        // (+ bias, sigmoid, combine, IDWT) is done by head_shiftsum_kernel on the 54-plane result.
        static_assert(!FUSE || TAPS == 1, "the fused head is a 1x1 chain");
        constexpr int CO_T = WM * MR * 16, PS2 = T::PS, R2W = WM >= 2 ? 1 : 2;
        // the two 16-row tiles of tap-partials go to wave rows 0 and 1 (both to row 0 when WM == 1)
        float* mid = lds;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int chl = (wm * MR + m) * 16 + (lane & 15);
            const float bv = bias_v[m];
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                const int q = (wn * NR + n) * 16 + (lane >> 4) * 4;
                float4 v;
                v.x = act_apply(acc[m][n][0] + bv, a.act, a.slope);
                v.y = act_apply(acc[m][n][1] + bv, a.act, a.slope);
                v.z = act_apply(acc[m][n][2] + bv, a.act, a.slope);
                v.w = act_apply(acc[m][n][3] + bv, a.act, a.slope);
                *reinterpret_cast<float4*>(mid + chl * PS2 + q) = v;
            }
        }
        __syncthreads();
        if (WM > 2 && wm >= 2) return;
        f32x4 acc2[R2W][NR];
#pragma unroll
        for (int j = 0; j < R2W; ++j)
#pragma unroll
            for (int n = 0; n < NR; ++n) acc2[j][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int NCI4_2 = CO_T / 4;   // CO_T is a multiple of 16
        const float* w2 = a.wp2 + (size_t)by * (2 * NCI4_2 * 64) + lane;
        const float* msrc = mid + (lane >> 4) * PS2 + (lane & 15);
        // the tap-partial weights come straight from L2 (they do not fit beside `mid`): a ring of QD K-steps of fragments
        // keeps QD global loads per lane in flight so that none of them is waited for in the MFMA stream
        constexpr int QD = NCI4_2 < 8 ? NCI4_2 : 8;
        float wq[QD][R2W];
#pragma unroll
        for (int q = 0; q < QD; ++q)
#pragma unroll
            for (int j = 0; j < R2W; ++j) wq[q][j] = w2[((WM == 1 ? j : wm) * NCI4_2 + q) * 64];
#pragma unroll
        for (int k4 = 0; k4 < NCI4_2; ++k4) {
            float pf[NR], wf[R2W];
#pragma unroll
            for (int j = 0; j < R2W; ++j) wf[j] = wq[k4 % QD][j];
            if (k4 + QD < NCI4_2) {
#pragma unroll
                for (int j = 0; j < R2W; ++j) wq[k4 % QD][j] = w2[((WM == 1 ? j : wm) * NCI4_2 + k4 + QD) * 64];
            }
#pragma unroll
            for (int n = 0; n < NR; ++n) pf[n] = msrc[k4 * 4 * PS2 + (wn * NR + n) * 16];
#pragma unroll
            for (int j = 0; j < R2W; ++j)
#pragma unroll
                for (int n = 0; n < NR; ++n) acc2[j][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf[n], wf[j], acc2[j][n], 0, 0, 0);
        }
        const bool vec = (W & 3) == 0;
#pragma unroll
        for (int j = 0; j < R2W; ++j) {
            const int r2 = (WM == 1 ? j : wm) * 16 + (lane & 15);
            if (r2 >= 27) continue;
            float* tb = a.t + ((size_t)b * a.t_ctot + a.t_row0 + by * 27 + r2) * plane2;
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                const int ox = x0 + (wn * NR + n) * 16 + (lane >> 4) * 4;
                if (ox >= W) continue;
                if (vec && ox + 3 < W) {
                    *reinterpret_cast<float4*>(tb + ox) = make_float4(acc2[j][n][0], acc2[j][n][1], acc2[j][n][2], acc2[j][n][3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ox + r < W) tb[ox + r] = acc2[j][n][r];
                }
            }
        }
        return;
    }

    // ---- epilogue: lane = (channel lane&15, pixel quad lane>>4) ----------------------------------------
    const bool final_out = (a.ksplit == 1);
    float* ybase = a.y + ((size_t)ks * a.B + b) * a.Cout * plane2;
    const bool vec_ok = (W & 3) == 0;
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const int co = ((by * WM + wm) * MR + m) * 16 + (lane & 15);
        const float bv = final_out ? bias_v[m] : 0.f;
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            const int q = (wn * NR + n) * 16 + (lane >> 4) * 4;  // first of this lane's 4 consecutive pixels
            const int oy = y0 + q / TW, ox = x0 + q % TW;
            if (co >= a.Cout || q >= T::NPIX || oy >= H || ox >= W) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[m][n][r];
                if (final_out) v[r] = act_apply(v[r] + bv, a.act, a.slope);
            }
            float* dst = ybase + (size_t)co * plane2 + (size_t)oy * W + ox;
            if (final_out && a.gate) {
                const float* gp = a.gate + ((size_t)b * a.Cout + co) * plane2 + (size_t)oy * W + ox;
                if (vec_ok && ox + 3 < W) {
                    const float4 gv = *reinterpret_cast<const float4*>(gp);
                    v[0] *= act_deriv(gv.x, a.gate_act, a.gate_slope);
                    v[1] *= act_deriv(gv.y, a.gate_act, a.gate_slope);
                    v[2] *= act_deriv(gv.z, a.gate_act, a.gate_slope);
                    v[3] *= act_deriv(gv.w, a.gate_act, a.gate_slope);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ox + r < W) v[r] *= act_deriv(gp[r], a.gate_act, a.gate_slope);
                }
            }
            if (vec_ok && ox + 3 < W) {
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (ox + r < W) dst[r] = v[r];
            }
        }
    }
}

// split-K second stage: y = act(bias + sum_s partial[s])
__global__ void conv_splitk_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                          float* __restrict__ y, size_t n, size_t plane, int Cout, int ksplit,
                                          int act, float slope, const float* __restrict__ gate, int gate_act, float gate_slope,
                                          const uint8_t* __restrict__ out_mask) {
    // out_mask [B,plane] (block-sparse execution): an inactive pixel is 0 -- its partial sums are never read (a skipped tile left
    // its workspace slots unwritten).
    // the ksplit partial loads of an element are independent: issue them four at a time (a plain `v += partial[...]` loop
    // with a run-time trip count waits for every round trip in turn), summing in the fixed order s = 0 .. ksplit-1
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        int s = 0;
        if (out_mask && out_mask[i / (plane * Cout) * plane + i % plane] == 0) {
            y[i] = 0.f;
            continue;
        }
        for (; s + 4 <= ksplit; s += 4) {
            const float p0 = partial[(size_t)s * n + i], p1 = partial[(size_t)(s + 1) * n + i];
            const float p2 = partial[(size_t)(s + 2) * n + i], p3 = partial[(size_t)(s + 3) * n + i];
            v = (((v + p0) + p1) + p2) + p3;
        }
        for (; s < ksplit; ++s) v += partial[(size_t)s * n + i];
        if (bias) v += bias[(i / plane) % Cout];
        v = act_apply(v, act, slope);
        if (gate) v *= act_deriv(gate[i], gate_act, gate_slope);
        y[i] = v;
    }
}

// Second pass of a work-list launch (wmd_conv_args.out_tiles): a block = (listed tile, group of kRedCh out channels) sums the ks
// partial planes of its outputs in the fixed order s = 0 .. ks-1 (all ks loads of an element in flight together: 16-byte pieces
// along the row), adds the bias, applies the activation and the out-mask select.  ks comes from the same device-side rule as in
// the convolution (list_ksplit); ks == 1 means the convolution already wrote the final values and every block returns at once.
// (The first version gave a block a whole tile x Cout: 64 dependent rounds of ks loads per thread, 114-126 us per launch at
// one frame against 13-20 us for the convolution itself.)
constexpr int kRedCh = 8;
__global__ __launch_bounds__(256) void conv_splitk_reduce_list_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                                                      float* __restrict__ y, const int* __restrict__ tile_list,
                                                                      const int* __restrict__ tile_count, const uint8_t* __restrict__ out_mask,
                                                                      int B, int Cout, int H, int W, int TH, int TW, int tiles_x, int tiles_y,
                                                                      int cob, int nchunks, int ksmax, int slots, int act, float slope) {
    const int n_act = list_total(tile_count, B);
    if ((int)blockIdx.x >= n_act) return;
    int ks, cps;
    list_ksplit(n_act * cob, nchunks, ksmax, slots, ks, cps);
    if (ks == 1) return;
    int t = list_entry(tile_list, tile_count, B, tiles_x * tiles_y, blockIdx.x);
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW, c0 = blockIdx.y * kRedCh;
    const size_t plane = (size_t)H * W, n = (size_t)B * Cout * plane;
    const int vpr = TW / 4, per_ch = TH * vpr;   // 16-byte pieces per tile row / per channel (TW % 4 == 0: the list kernels' tiles)
    const bool vec = (W & 3) == 0;
    for (int e = threadIdx.x; e < kRedCh * per_ch; e += 256) {
        const int c = c0 + e / per_ch, r = e % per_ch;
        const int yy = y0 + r / vpr, xx = x0 + (r % vpr) * 4;
        if (c >= Cout || yy >= H || xx >= W) continue;
        const size_t i = ((size_t)b * Cout + c) * plane + (size_t)yy * W + xx;
        const float bv = bias ? bias[c] : 0.f;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (vec) {   // xx % 4 == 0 and W % 4 == 0: the piece is aligned and inside the row
            float4 p[16];
#pragma unroll
            for (int s = 0; s < 16; ++s)
                if (s < ks) p[s] = *reinterpret_cast<const float4*>(partial + (size_t)s * n + i);
#pragma unroll
            for (int s = 0; s < 16; ++s)
                if (s < ks) {
                    v[0] += p[s].x;
                    v[1] += p[s].y;
                    v[2] += p[s].z;
                    v[3] += p[s].w;
                }
        } else {
            for (int s = 0; s < ks; ++s)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (xx + k < W) v[k] += partial[(size_t)s * n + i + k];
        }
        uint8_t mv[4] = {1, 1, 1, 1};
        if (out_mask) {
#pragma unroll
            for (int k = 0; k < 4; ++k) mv[k] = out_mask[(size_t)b * plane + (size_t)yy * W + min(xx + k, W - 1)];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = act_apply(v[k] + bv, act, slope);
            if (!mv[k]) v[k] = 0.f;
        }
        if (vec) {
            *reinterpret_cast<float4*>(y + i) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (xx + k < W) y[i + k] = v[k];
        }
    }
}

// ================================================================================================
// Winograd F(2x2, 3x3) form of the same convolution: 2.25x fewer MFMAs per output.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A          d: 4x4 input patch of a 2x2 output tile, g: 3x3 filter
//
// In the transformed domain every one of the 16 positions xi is an independent GEMM over the input channels,
//   M_xi[tile, co] += V_xi[tile, ci] * U_xi[ci, co],
// so the kernel keeps the staging of the direct kernel (the same LDS-DMA gather of the padded / upsampled /
// concatenated halo patch, the same fragment-ordered weight image -- now with 16 "taps" = transformed positions, see
// conv_pack_wino_kernel) and changes what happens between LDS and the matrix pipe:
//   * a lane owns one tile (l & 15) and one channel of the K-step (l >> 4): it reads the tile's 4x4 patch from LDS at
//     immediate offsets (16 ds_read_b32), computes B^T d B in registers (32 adds) and so holds the row operand of all
//     16 positions -- no transformed-input buffer, no extra barrier;
//   * a wave owns 16 tiles x MRW out-channel tiles x all 16 positions (16*MRW accumulator fragments), so the output
//     transform A^T M A is lane-local as well: a lane's accumulator registers are 4 consecutive tiles of one channel;
//   * bias + activation + 2 x (2 x 16-byte) stores per fragment row.
// Block = WM x WN waves: WN tile groups (16 tiles = 64 output pixels each), WM * MRW * 16 out channels.
// ================================================================================================
template <int TH, int TW, int MRW, int WM, int WN, int CK>
struct WinoTile {
    static constexpr int NT = WM * WN * 64;
    static constexpr int PH = TH + 2, PW = TW + 2;
    static constexpr int NPOSITIONS = PH * PW;
    // per-channel LDS stride: odd (== 1 mod 32), so that the two k-lanes of a 32-lane ds_read_b32 group -- whose tile
    // origins are 2 floats apart -- fall on even and on odd banks
    static constexpr int PS = ((NPOSITIONS - 1 + 31) / 32) * 32 + 1;
    static constexpr int NPOS = (NPOSITIONS + NT - 1) / NT;
    static constexpr int TXW = TW / 2, TYH = TH / 2, NTILES = TXW * TYH;
    static constexpr int A_RUN = (CK / 4) * 16 * 64;
    static constexpr int A_FLOATS = WM * MRW * A_RUN;
    static constexpr int B_FLOATS = ((CK * PS + 3) / 4) * 4;   // keeps the weight tile 16-byte aligned
    static constexpr int NAV = (A_FLOATS / 4 + NT - 1) / NT;
    static constexpr int LDS_FLOATS = 2 * (B_FLOATS + A_FLOATS);
    static_assert(TH % 2 == 0 && TW % 8 == 0, "whole 2x2 tiles; four consecutive tiles of a lane stay in one tile row");
    static_assert(WN * 16 >= NTILES, "more tiles than MFMA rows");
    static_assert(CK % 4 == 0, "CK must be a whole number of 4-channel K-steps");
};

// MASKED: the block-sparse instantiation (in_mask / out_mask of the sparse decoders); the dense instantiation carries none
// of that code (with run-time tests alone the dense trunk layers lost 4-6 %: measured by bisection, round 2)
template <int TH, int TW, int MRW, int WM, int WN, int CK, bool MASKED = false>
__global__ __launch_bounds__(WM* WN * 64, 2) void conv_wino_kernel(const ConvKArgs a) {   // <= 256 registers: 2 blocks per CU
    using T = WinoTile<TH, TW, MRW, WM, WN, CK>;
    constexpr int NT = T::NT, PW = T::PW, PS = T::PS, NPOS = T::NPOS, KSTEPS = CK / 4;
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS];
    float* ldsA = lds + 2 * T::B_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM;
    const int wn = wave / WM;

    // Work item = (pixel tile, out-channel slab), slab fastest.  Workgroups are dealt to the 8 XCDs round-robin and every
    // XCD has its own L2: giving an XCD one contiguous run of the item sequence puts the slabs of a tile -- which gather
    // the SAME input patch -- on the same XCD back to back, so only the first of them fetches it from HBM (a 2-D grid
    // launched all tiles of slab 0 before any tile of slab 1: Cout = 64 / 128 layers with 32-channel blocks moved 1.95x
    // their algorithmic bytes), and neighbouring tiles share halo rows and 128-byte lines in that L2 as before.
    int t, by;
    if (a.cob > 0) {
        const int item = xcd_contiguous(blockIdx.x, gridDim.x);
        t = item / a.cob;
        by = item - t * a.cob;
    } else {
        t = xcd_contiguous(blockIdx.x, gridDim.x);
        by = blockIdx.y;
    }
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int b = t / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int ks = blockIdx.z;
    const int H = a.H, W = a.W;
    if (MASKED && a.out_mask) {   // block-sparse: nothing to do for a tile without active output pixels (its outputs stay zero)
        int any = 0;
        for (int i = tid; i < TH * TW; i += (int)blockDim.x) {
            const int yy = y0 + i / TW, xx = x0 + i % TW;
            if (yy < H && xx < W) any |= a.out_mask[(size_t)b * H * W + (size_t)yy * W + xx];
        }
        // a ballot per wave + flags in LDS (not __syncthreads_or: the device library's work-group reduction it pulls in
        // slowed conv_wino32_kernel by 45 % -- even its launches without masks)
        __shared__ int tile_flags[16];
        const bool wave_any = __builtin_amdgcn_ballot_w64(any != 0) != 0;
        if ((tid & 63) == 0) tile_flags[tid >> 6] = wave_any ? 1 : 0;
        __syncthreads();
        int all = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) all |= tile_flags[w];
        if (__builtin_amdgcn_readfirstlane(all) == 0) return;
    }

    // ---- staging geometry (identical to conv_fwd_kernel: see the comments there) -------------------------------
    constexpr unsigned kOOB = 0x80000000u;
    unsigned ob1[NPOS], ob2[NPOS];
#pragma unroll
    for (int i = 0; i < NPOS; ++i) {
        const int p = tid + i * NT;
        const int py = p / PW, px = p % PW;
        int gy = y0 + py - 1, gx = x0 + px - 1;
        bool ok = p < T::NPOSITIONS;
        ok = pad_coord(gy, H, a.pad_mode) && ok;
        ok = pad_coord(gx, W, a.pad_mode) && ok;
        ok = ok && gy < H && gx < W && gy >= 0 && gx >= 0;
        gy = min(max(gy, 0), H - 1);
        gx = min(max(gx, 0), W - 1);
        if (MASKED && a.in_mask) ok = ok && a.in_mask[(size_t)b * H * W + gy * W + gx] != 0;   // sparse support of the virtual input
        ob2[i] = ok ? (unsigned)(gy * W + gx) * 4u : kOOB;
        int sy = gy - a.shift1, sx = gx - a.shift1;
        if (a.up1 == 2) {
            sy = gy >> 1;
            sx = gx >> 1;
        }
        const bool ok1 = ok && sy >= 0 && sx >= 0 && sy < a.H1 && sx < a.W1;
        ob1[i] = ok1 ? (unsigned)(sy * a.W1 + sx) * 4u : kOOB;
    }
    const size_t plane1 = (size_t)a.H1 * a.W1, plane2 = (size_t)H * W;
    const float* x1b = a.x1 + (size_t)b * a.C1 * plane1;
    const float* x2b = a.x2 ? a.x2 + (size_t)b * a.C2 * plane2 : a.x1;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x1b), 0, (int)(a.C1 * plane1 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x2b), 0, (int)(a.C2 * plane2 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.wp), 0, (int)((size_t)a.ncot * a.nci4 * 16 * 64 * 4), 0x00020000);
    const unsigned pb1 = (unsigned)(plane1 * 4), pb2 = (unsigned)(plane2 * 4);
    unsigned aoff[T::NAV];
#pragma unroll
    for (int v = 0; v < T::NAV; ++v) {
        const int e = tid + v * NT;
        const int run = (e * 4) / T::A_RUN, rem = (e * 4) % T::A_RUN;
        const int cot = min(by * WM * MRW + run, a.ncot - 1);
        aoff[v] = (unsigned)(((size_t)cot * a.nci4 * 16 * 64 + rem) * 4);
    }
    constexpr int NPB = CK * NPOS, NPIECES = NPB + T::NAV;
    // Where a chunk's CK channels come from is decided ONCE per chunk (wave-uniform, scalar unit): a chunk that lies entirely
    // in one source tensor -- the usual case, C1 and C2 multiples of CK -- selects that tensor's descriptor, per-lane
    // offsets and plane stride up front, and every piece is then descriptor + (base + j * stride): no per-piece channel
    // arithmetic, comparisons or branches in the MFMA stream (measured with SQ_INSTS_SALU: 4.75 scalar instructions per
    // MFMA before, see profiles/r02_wino_counters.md).  A chunk that straddles the C1 boundary or the channel tail takes the
    // generic per-channel path.
    struct ChunkSrc {
        bool pure, in1;
        unsigned base, step;
        unsigned ob[NPOS];
    };
    auto chunk_src = [&](int chunk) {
        ChunkSrc cs;
        const int ci0 = chunk * CK;
        cs.in1 = ci0 + CK <= a.C1;
        cs.pure = cs.in1 || (ci0 >= a.C1 && ci0 + CK <= a.Cin);
        cs.base = cs.in1 ? (unsigned)ci0 * pb1 : (unsigned)max(ci0 - a.C1, 0) * pb2;
        cs.step = cs.in1 ? pb1 : pb2;
#pragma unroll
        for (int i = 0; i < NPOS; ++i) cs.ob[i] = cs.in1 ? ob1[i] : ob2[i];
        return cs;
    };
    auto stage_piece = [&](int chunk, int buf, int q, const ChunkSrc& cs) {
        if (q < NPB) {
            const int j = q / NPOS, i = q % NPOS;
            if (NPOS * NT == T::NPOSITIONS || tid + i * NT < T::NPOSITIONS) {
                lds_ptr_t d = (lds_ptr_t)(lds + buf * T::B_FLOATS + wave * 64 + j * PS + i * NT);
                if (cs.pure) {
                    const unsigned soff = cs.base + (unsigned)j * cs.step;
                    if (cs.in1) lds_dma4(r1, d, cs.ob[i], soff);
                    else lds_dma4(r2, d, cs.ob[i], soff);
                } else {
                    const int ci = chunk * CK + j;
                    const bool from_x1 = ci < a.C1;
                    const bool chan_ok = ci < a.Cin;
                    const unsigned soff = from_x1 ? (unsigned)ci * pb1 : (unsigned)max(ci - a.C1, 0) * pb2;
                    const unsigned vo = chan_ok ? (from_x1 ? ob1[i] : ob2[i]) : kOOB;
                    if (from_x1) lds_dma4(r1, d, vo, soff);
                    else lds_dma4(r2, d, vo, soff);
                }
            }
        } else {
            const int v = q - NPB;
            const unsigned soffA = (unsigned)chunk * (unsigned)(T::A_RUN * 4);
            if (T::NAV * NT * 4 == T::A_FLOATS || (tid + v * NT) * 4 < T::A_FLOATS)
                lds_dma16(rw, (lds_ptr_t)(ldsA + buf * T::A_FLOATS + wave * 256 + v * NT * 4), aoff[v], soffA);
        }
    };

    // ---- operand addressing -------------------------------------------------------------------------------------
    // this lane's tile (row operand): tile slot wn*16 + (l & 15) -> patch origin (2 ty, 2 tx); channel lane l >> 4
    const int tslot = min(wn * 16 + (lane & 15), T::NTILES - 1);
    const int boff = (tslot / T::TXW) * 2 * PW + (tslot % T::TXW) * 2 + (lane >> 4) * PS;
    const int a_lane = (wm * MRW) * T::A_RUN + lane;

    f32x4 acc[16][MRW];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int m = 0; m < MRW; ++m) acc[xi][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int c_begin = ks * a.chunks_per_split;
    const int c_end = min(c_begin + a.chunks_per_split, a.nchunks);
    if (c_begin < c_end) {
        const ChunkSrc cs0 = chunk_src(c_begin);
#pragma unroll
        for (int q = 0; q < NPIECES; ++q) stage_piece(c_begin, 0, q, cs0);
    }
    float bias_v[MRW];   // requested now, used in the epilogue (see conv_fwd_kernel)
#pragma unroll
    for (int m = 0; m < MRW; ++m) {
        const int co = ((by * WM + wm) * MRW + m) * 16 + (lane & 15);
        bias_v[m] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
    }
    __syncthreads();

    // One chunk = S = KSTEPS*16 groups (K-step, position) of MRW MFMAs each.  Explicitly software-pipelined and pinned
    // with sched_barrier like the direct kernel: the weight fragments of group s+D, the 4x4 patch of the next K-step
    // (16 ds_reads at group 2) and its transform (32 adds at group 10) are issued in the shadow of earlier MFMAs; the
    // next chunk's DMA pieces are dealt out over the first two thirds of the groups.
    auto chunk_body = [&](int c, auto prefetch) {
        constexpr bool PREFETCH = decltype(prefetch)::value;
        constexpr int S = KSTEPS * 16, D = 4, RS = D + 1, SP = (S * 2) / 3;
        const int buf = (c - c_begin) & 1;
        const float* bsrc = lds + buf * T::B_FLOATS + boff;
        const float* asrc = ldsA + buf * T::A_FLOATS + a_lane;
        float d[16], v[2][16], wf[RS][MRW];
        ChunkSrc csn;
        if constexpr (PREFETCH) csn = chunk_src(c + 1);
        auto fetch_patch = [&](int kk) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) d[r * 4 + cc] = bsrc[kk * 4 * PS + r * PW + cc];
        };
        auto transform = [&](int kk) {   // V = B^T d B
            float tr[16];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                tr[0 * 4 + cc] = d[0 * 4 + cc] - d[2 * 4 + cc];
                tr[1 * 4 + cc] = d[1 * 4 + cc] + d[2 * 4 + cc];
                tr[2 * 4 + cc] = d[2 * 4 + cc] - d[1 * 4 + cc];
                tr[3 * 4 + cc] = d[1 * 4 + cc] - d[3 * 4 + cc];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[kk & 1][r * 4 + 0] = tr[r * 4 + 0] - tr[r * 4 + 2];
                v[kk & 1][r * 4 + 1] = tr[r * 4 + 1] + tr[r * 4 + 2];
                v[kk & 1][r * 4 + 2] = tr[r * 4 + 2] - tr[r * 4 + 1];
                v[kk & 1][r * 4 + 3] = tr[r * 4 + 1] - tr[r * 4 + 3];
            }
        };
        auto fetch_u = [&](int s2) {
#pragma unroll
            for (int m = 0; m < MRW; ++m) wf[s2 % RS][m] = asrc[m * T::A_RUN + s2 * 64];
        };
        fetch_patch(0);
#pragma unroll
        for (int s2 = 0; s2 < D && s2 < S; ++s2) fetch_u(s2);
        transform(0);
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) {
            const int kk = s2 / 16, xi = s2 % 16;
            if (s2 + D < S) fetch_u(s2 + D);
            if (xi == 2 && kk + 1 < KSTEPS) fetch_patch(kk + 1);
            if (xi == 10 && kk + 1 < KSTEPS) transform(kk + 1);
            if constexpr (PREFETCH) {
#pragma unroll
                for (int q = 0; q < NPIECES; ++q)
                    if (q * SP / NPIECES == s2) stage_piece(c + 1, buf ^ 1, q, csn);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MRW; ++m)
                acc[xi][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[kk & 1][xi], wf[s2 % RS][m], acc[xi][m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    for (int c = c_begin; c + 1 < c_end; ++c) chunk_body(c, std::true_type{});
    if (c_begin < c_end) chunk_body(c_end - 1, std::false_type{});

    // ---- epilogue: Y = A^T M A per (channel, tile); lane = (channel l & 15, tiles (l >> 4) * 4 + r) ---------------
    const bool final_out = (a.ksplit == 1);
    float* ybase = a.y + ((size_t)ks * a.B + b) * a.Cout * plane2;
    const int tfirst = wn * 16 + (lane >> 4) * 4;          // first of this lane's four consecutive tiles (same tile row)
    const int oy = y0 + (tfirst / T::TXW) * 2, ox = x0 + (tfirst % T::TXW) * 2;
    const bool vec_ok = (W & 3) == 0;
#pragma unroll
    for (int m = 0; m < MRW; ++m) {
        const int co = ((by * WM + wm) * MRW + m) * 16 + (lane & 15);
        if (co >= a.Cout || tfirst >= T::NTILES || oy >= H || ox >= W) continue;
        const float bv = final_out ? bias_v[m] : 0.f;
        float yrow[2][8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float tt[2][4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                tt[0][cc] = acc[0 * 4 + cc][m][r] + acc[1 * 4 + cc][m][r] + acc[2 * 4 + cc][m][r];
                tt[1][cc] = acc[1 * 4 + cc][m][r] - acc[2 * 4 + cc][m][r] - acc[3 * 4 + cc][m][r];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                yrow[j][2 * r + 0] = tt[j][0] + tt[j][1] + tt[j][2];
                yrow[j][2 * r + 1] = tt[j][1] - tt[j][2] - tt[j][3];
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (oy + j >= H) continue;
            float* dst = ybase + (size_t)co * plane2 + (size_t)(oy + j) * W + ox;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (final_out) yrow[j][e] = act_apply(yrow[j][e] + bv, a.act, a.slope);
            if (MASKED && a.out_mask) {
                const uint8_t* mp = a.out_mask + (size_t)b * plane2 + (size_t)(oy + j) * W + ox;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (ox + e < W && mp[e] == 0) yrow[j][e] = 0.f;
            }
            if (vec_ok && ox + 7 < W) {
                *reinterpret_cast<float4*>(dst) = make_float4(yrow[j][0], yrow[j][1], yrow[j][2], yrow[j][3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(yrow[j][4], yrow[j][5], yrow[j][6], yrow[j][7]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (ox + e < W) dst[e] = yrow[j][e];
            }
        }
    }
}

// Winograd weight image: the layout of conv_pack_kernel with 16 "taps" = transformed positions xi = 4a + b,
//   U[a][b] = sum_{i,j} G[a][i] g[i][j] G[b][j],   G = [[1,0,0],[1/2,1/2,1/2],[1/2,-1/2,1/2],[0,0,1]].
__global__ void conv_pack_wino_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int ncot, int nci4,
                                      int dgrad) {
    const size_t total = (size_t)ncot * nci4 * 16 * 64;
    const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int l = i & 63;
        size_t r = i >> 6;
        const int xi = r % 16;
        r /= 16;
        const int ci4 = r % nci4;
        const int cot = r / nci4;
        const int m = cot * 16 + (l & 15);
        const int k = ci4 * 4 + (l >> 4);
        float g[9];
        bool ok;
        if (!dgrad) {
            ok = m < Cout && k < Cin;
            for (int t = 0; t < 9; ++t) g[t] = ok ? w[((size_t)m * Cin + k) * 9 + t] : 0.f;
        } else {
            ok = m < Cin && k < Cout;
            for (int t = 0; t < 9; ++t) g[t] = ok ? w[((size_t)k * Cin + m) * 9 + (8 - t)] : 0.f;
        }
        const int pa = xi / 4, pb = xi % 4;
        float s = 0.f;
        for (int ii = 0; ii < 3; ++ii)
            for (int jj = 0; jj < 3; ++jj) s += G[pa][ii] * g[ii * 3 + jj] * G[pb][jj];
        wp[i] = s;
    }
}

// weights [Cout,Cin,k,k] -> fragment image [ncot][nci4][taps][64]; lane l holds
// W[cot*16 + (l&15)][ci4*4 + (l>>4)][tap], zero outside.
__global__ void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int taps,
                                 int ncot, int nci4, int dgrad) {
    const size_t total = (size_t)ncot * nci4 * taps * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int l = i & 63;
        size_t r = i >> 6;
        const int tap = r % taps;
        r /= taps;
        const int ci4 = r % nci4;
        const int cot = r / nci4;
        const int m = cot * 16 + (l & 15);  // GEMM row
        const int k = ci4 * 4 + (l >> 4);   // GEMM reduction channel
        float v = 0.f;
        if (!dgrad) {
            if (m < Cout && k < Cin) v = w[((size_t)m * Cin + k) * taps + tap];
        } else {
            // data gradient: rows are input channels, reduction runs over output channels, taps flipped
            if (m < Cin && k < Cout) v = w[((size_t)k * Cin + m) * taps + (taps - 1 - tap)];
        }
        wp[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// host side: configuration table + selection
// ------------------------------------------------------------------------------------------------
struct ConvCfg {
    int TH, TW, MR, NR, WM, WN, CK, TAPS;
    int lds_bytes;
    void (*launch)(const ConvKArgs&, dim3, hipStream_t);
    const char* name;
};

template <int TH, int TW, int MR, int NR, int WM, int WN, int CK, int TAPS>
static void launch_cfg(const ConvKArgs& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((conv_fwd_kernel<TH, TW, MR, NR, WM, WN, CK, TAPS>), grid, dim3(WM * WN * 64), 0, s, a);
}

#define WMD_CFG(TH, TW, MR, NR, WM, WN, CK, TAPS)                                                       \
    ConvCfg {                                                                                           \
        TH, TW, MR, NR, WM, WN, CK, TAPS, (int)sizeof(float) * ConvTile<TH, TW, MR, NR, WM, WN, CK, TAPS>::LDS_FLOATS, \
            &launch_cfg<TH, TW, MR, NR, WM, WN, CK, TAPS>,                                              \
            "conv_fwd_kernel<" #TH "," #TW "," #MR "," #NR "," #WM "," #WN "," #CK "," #TAPS ">"         \
    }

template <int TH, int TW, int MRW, int WM, int WN, int CK>
static void launch_wino(const ConvKArgs& a, dim3 grid, hipStream_t s) {
    if (a.in_mask || a.out_mask) hipLaunchKernelGGL((conv_wino_kernel<TH, TW, MRW, WM, WN, CK, true>), grid, dim3(WM * WN * 64), 0, s, a);
    else hipLaunchKernelGGL((conv_wino_kernel<TH, TW, MRW, WM, WN, CK, false>), grid, dim3(WM * WN * 64), 0, s, a);
}
// Winograd entries reuse the table's fields: MR = out-channel tiles per wave, NR = 1 (a wave owns one group of 16
// tiles = 64 pixels), TAPS = 16 transformed positions (this is what marks them).
#define WMD_WINO(TH, TW, MRW, WM, WN, CK)                                                               \
    ConvCfg {                                                                                           \
        TH, TW, MRW, 1, WM, WN, CK, 16, (int)sizeof(float) * WinoTile<TH, TW, MRW, WM, WN, CK>::LDS_FLOATS, \
            &launch_wino<TH, TW, MRW, WM, WN, CK>, "conv_wino_kernel<" #TH "," #TW "," #MRW "," #WM "," #WN "," #CK ">" \
    }

// conv_wino32_kernel (wmd_conv_wino32.hip) entries: TAPS = 17 marks the family (Winograd weight image, 3x3 semantics);
// MR = 2 (a block's slab = 32 out channels), NR = 1 and WN = waves / WM so that the planner's block-shape arithmetic holds
#define WMD_W32_INST(TH, TW, WN, CK)                                                                                   \
    ConvCfg{TH, TW, 2, 1, 1, (WN) * 2, CK, 17,                                                                         \
            (int)sizeof(float) * (W32Tile<TH, TW, WN, CK>::LDS_FLOATS + W32Tile<TH, TW, WN, CK>::TAB_FLOATS),          \
            &launch_wino32<TH, TW, WN, CK>, "conv_wino32_kernel<" #TH "," #TW "," #WN "," #CK ">"},

// conv_wino32q_kernel (wmd_conv_wino32q.hip, round 5) entries: TAPS = 18 marks the family (quarter-position waves: pure
// channel chunking required; MASKED / LIST instantiations serve the block-sparse levels); MR = 2 (a block's slab = 32 out channels), WN = its four waves
#define WMD_W32Q_INST(TH, TW, CK)                                                                                      \
    ConvCfg{TH, TW, 2, 1, 1, 4, CK, 18,                                                                                \
            (int)sizeof(float) * (W32QTile<TH, TW, CK>::LDS_FLOATS + W32QTile<TH, TW, CK>::TAB_FLOATS),                \
            &launch_wino32q<TH, TW, CK>, "conv_wino32q_kernel<" #TH "," #TW "," #CK ">"},

static const ConvCfg kCfgs[] = {
    // 3x3, 32-wide rows (W % 32 == 0: 160/320, 1024-wide pyramids)
    WMD_CFG(16, 32, 2, 8, 1, 4, 8, 9),  // co32  x 512px
    WMD_CFG(8, 32, 2, 8, 1, 2, 8, 9),   // co32  x 256px, 2 waves
    WMD_CFG(8, 32, 4, 4, 1, 4, 16, 9),  // co64  x 256px, 16-channel chunks
    WMD_CFG(8, 32, 4, 8, 1, 2, 8, 9),   // co64  x 256px, 2 waves of 64x128
    WMD_CFG(8, 32, 2, 4, 1, 4, 8, 9),   // co32  x 256px
    WMD_CFG(8, 32, 2, 4, 1, 4, 16, 9),  // co32  x 256px, 16-channel chunks
    WMD_CFG(16, 32, 2, 8, 1, 4, 16, 9), // co32  x 512px, 16-channel chunks
    WMD_CFG(8, 32, 4, 4, 1, 4, 8, 9),   // co64  x 256px
    WMD_CFG(4, 32, 4, 4, 2, 2, 8, 9),   // co128 x 128px
    WMD_CFG(4, 32, 4, 2, 1, 4, 8, 9),   // co64  x 128px
    WMD_CFG(2, 32, 4, 2, 2, 2, 8, 9),   // co128 x 64px
    WMD_CFG(5, 32, 4, 5, 2, 2, 8, 9),   // co128 x 160px  (H % 5: 10x32 coarsest level of 1024x320)
    WMD_CFG(4, 32, 2, 2, 1, 4, 8, 9),   // co32  x 128px (32 KB of LDS: 5 blocks per CU)
    WMD_CFG(8, 32, 2, 4, 1, 4, 4, 9),   // co32  x 256px, 4-channel chunks (21 KB)
    WMD_CFG(4, 32, 4, 2, 1, 4, 4, 9),   // co64  x 128px, 4-channel chunks (25 KB)
    WMD_CFG(2, 40, 1, 5, 2, 1, 8, 9),   // co32  x 80px, 2 waves: enough blocks on 12x40 / 24x80 without split-K
    WMD_CFG(2, 40, 2, 5, 2, 1, 8, 9),   // co64  x 80px, 2 waves
    WMD_CFG(4, 40, 1, 5, 2, 2, 8, 9),   // co32  x 160px, 4 waves
    WMD_CFG(4, 40, 2, 5, 2, 2, 4, 9),   // co64  x 160px, 4-channel chunks
    WMD_CFG(4, 40, 4, 5, 2, 2, 4, 9),   // co128 x 160px, 4-channel chunks
    WMD_CFG(4, 32, 2, 4, 1, 2, 8, 9),   // co32  x 128px, 2 waves
    WMD_CFG(8, 32, 4, 4, 2, 4, 8, 9),   // co128 x 256px, 8 waves
    WMD_CFG(8, 32, 2, 4, 2, 4, 8, 9),   // co64  x 256px, 8 waves
    WMD_CFG(4, 32, 2, 4, 2, 2, 8, 9),   // co64  x 128px, 2x2 waves
    // 3x3, 40-wide rows (W = 40/80/160/320)
    WMD_CFG(8, 40, 4, 5, 1, 4, 8, 9),   // co64  x 320px
    WMD_CFG(4, 40, 2, 5, 2, 2, 8, 9),   // co64  x 160px, 2x2 waves
    WMD_CFG(4, 40, 4, 5, 2, 2, 8, 9),   // co128 x 160px
    WMD_CFG(4, 40, 4, 5, 1, 2, 8, 9),   // co64  x 160px
    WMD_CFG(2, 40, 4, 5, 1, 1, 8, 9),   // co64  x 80px, single wave
    WMD_CFG(3, 40, 4, 4, 1, 2, 8, 9),   // co64  x 120px (H = 15/30/60 ...)
    // 3x3, 20-wide rows (coarsest 640-wide level, NYUv2 15x20)
    WMD_CFG(6, 20, 4, 4, 2, 2, 8, 9),   // co128 x 120px
    WMD_CFG(6, 20, 2, 4, 2, 2, 8, 9),   // co64  x 120px (48 KB: 3 blocks per CU)
    WMD_CFG(6, 20, 2, 4, 1, 2, 8, 9),   // co32  x 120px, 2 waves
    WMD_CFG(3, 20, 4, 4, 1, 1, 8, 9),   // co64  x 60px, single wave
    WMD_CFG(5, 20, 4, 7, 1, 1, 8, 9),   // co64  x 100px, single wave
    // generic small tile (any W)
    WMD_CFG(4, 16, 4, 4, 1, 1, 8, 9),   // co64 x 64px
    WMD_CFG(8, 16, 2, 4, 1, 2, 8, 9),   // co32 x 128px
    // Winograd F(2x2,3x3): needs wmd_conv_args.wp_wino
    WMD_WINO(8, 32, 2, 1, 4, 8),    // co32 x 256px (64 tiles)
    WMD_WINO(8, 16, 2, 2, 2, 8),    // co64 x 128px (32 tiles)
    WMD_WINO(16, 16, 2, 1, 4, 8),   // co32 x 256px, square tile
    WMD_WINO(4, 32, 2, 2, 2, 8),    // co64 x 128px, one tile row pair
    WMD_WINO(4, 64, 2, 1, 4, 8),    // co32 x 256px, 64-wide
    WMD_WINO(8, 32, 2, 2, 4, 8),    // co64 x 256px, 8 waves
    WMD_WINO(6, 40, 2, 1, 4, 8),    // co32 x 240px (60 of 64 tile slots): 40-wide rows, H % 6 == 0
    WMD_WINO(6, 40, 2, 2, 4, 8),    // co64 x 240px, 8 waves
    WMD_WINO(4, 40, 2, 1, 3, 8),    // co32 x 160px (40 of 48 tile slots), 3 waves
    WMD_WINO(4, 40, 2, 2, 3, 8),    // co64 x 160px, 6 waves
    WMD_WINO(8, 16, 2, 4, 2, 8),    // co128 x 128px, 8 waves
    WMD_WINO(8, 32, 1, 2, 4, 8),    // co32 x 256px, 8 waves with one out-channel tile each (<= 128 registers: 4 waves / SIMD)
    WMD_WINO(8, 16, 1, 4, 2, 8),    // co64 x 128px, 8 waves, one out-channel tile each
    WMD_WINO(8, 16, 1, 2, 2, 8),    // co32 x 128px, 4 waves, one out-channel tile each
    WMD_WINO(4, 32, 1, 2, 2, 8),    // co32 x 128px, 4 waves; a tile group = one row of 16 tiles: no LDS bank conflicts
    WMD_WINO(4, 32, 1, 4, 2, 8),    // co64 x 128px, 8 waves
    WMD_WINO(6, 40, 1, 2, 4, 8),    // co32 x 240px, 8 waves
    WMD_WINO(8, 32, 1, 1, 4, 8),    // co16 x 256px, 4 waves: small blocks, many independent phases per CU
    WMD_WINO(8, 16, 1, 1, 2, 8),    // co16 x 128px, 2 waves
    WMD_WINO(16, 32, 1, 1, 8, 8),   // co16 x 512px, 8 waves
    WMD_WINO(16, 32, 1, 2, 8, 8),   // co32 x 512px, 16 waves
    // Winograd on 32x32x2 MFMAs, two position halves per tile group
#include "wmd_conv_wino32_table.inc"
    // Winograd on 32x32x2 MFMAs, four quarter-position waves per tile group (three blocks per CU)
#include "wmd_conv_wino32q_table.inc"
    // 1x1 on the flattened image (TH = 1)
    WMD_CFG(1, 256, 4, 4, 1, 4, 32, 1),  // co64  x 256px
    WMD_CFG(1, 256, 2, 4, 1, 4, 32, 1),  // co32  x 256px
    WMD_CFG(1, 128, 4, 4, 2, 2, 32, 1),  // co128 x 128px
    WMD_CFG(1, 64, 4, 4, 1, 1, 32, 1),   // co64  x 64px
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

struct ConvPlan {
    const ConvCfg* cfg;
    bool ticket;   // ksplit > 1 is finished inside the convolution (32x32x2 kernels) instead of by the second-stage kernel
    int ksplit, chunks_per_split, nchunks, tiles_x, tiles_y;
    int H, W;  // problem as seen by the kernel (1x1 is flattened)
    size_t workspace_floats;
};

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}


// Does table entry c run a work list (wmd_conv_args.out_tiles) on its own tile shape?  ONE predicate for the planner and for
// wmd_conv_list_tile_supported (the two used to disagree on shapes only the quarter family can serve).  Both 32x32x2
// families have LIST instantiations; where both serve a shape the quarter-position kernel takes it unless WMD_LIST_FAMILY=17
// (forced = an explicit table index was asked for: either twin may run).
static bool cfg_serves_list(const ConvCfg& c, bool forced) {
    static const int list_family = env_int("WMD_LIST_FAMILY", 18);
    const bool has17 = c.TAPS == 17 && wino32_has_list(c.TH, c.TW, c.WN / 2, c.CK);
    const bool has18 = c.TAPS == 18 && wino32q_has_list(c.TH, c.TW, c.CK);
    if (!has17 && !has18) return false;
    if (forced) return true;
    if (has17 && list_family == 18 && wino32q_has_list(c.TH, c.TW, c.CK)) return false;   // its quarter twin follows in the table
    if (has18 && list_family == 17 && (wino32_has_list(c.TH, c.TW, 1, c.CK) || wino32_has_list(c.TH, c.TW, 2, c.CK))) return false;
    return true;
}

static bool plan_conv(const wmd_conv_args* g, ConvPlan* plan, bool have_ws, size_t ws_floats) {
    const int taps = g->ksize == 3 ? 9 : 1;
    const int Cin = g->C1 + g->C2;
    const int H = taps == 9 ? g->H : 1;
    const int W = taps == 9 ? g->W : g->H * g->W;
    const int ncot = (g->Cout + 15) / 16;
    const int force = g->tune_cfg > 0 ? g->tune_cfg - 1 : env_int("WMD_CONV_CFG", -1);
    // tune_ksplit: k > 0 forces a k-way split, finished inside the convolution where the kernel can (32x32x2 families); k < 0 forces
    // |k| slices summed by the second-stage kernel (rounds 1-5; offered to the kernels that have both forms only)
    const int force_raw = g->tune_ksplit != 0 ? g->tune_ksplit : env_int("WMD_CONV_KSPLIT", 0);
    const int force_ks = force_raw < 0 ? -force_raw : force_raw;
    static const bool ticket_on = env_int("WMD_SPLITK_TICKET", 1) != 0;
    double best = 1e300;
    bool found = false;
    plan->ticket = false;
    for (int i = 0; i < kNumCfgs; ++i) {
        const ConvCfg& c = kCfgs[i];
        const bool wino = c.TAPS >= 16;
        if (wino ? (taps != 9 || !g->wp_wino) : c.TAPS != taps) continue;
        if ((g->in_mask || g->out_mask) && !wino) continue;   // block-sparse execution lives in the Winograd kernels
        if (g->gate && wino) continue;                        // the output gate lives in the direct kernels (and the split-K reduce)
        if (c.TAPS == 18 && (Cin % c.CK || (g->C2 > 0 && g->C1 % c.CK) || (g->in_mask && g->up1 == 2 && !g->in_mask_2x2)))
            continue;                                         // quarter-position kernel: flattened staging of pure layers only (wino32_pure)
        if (g->out_tiles) {   // work-list form: the LIST instantiations of the 32x32x2 kernels on the list's own tile shape
            // (both families have one for 8 x 16 tiles: the quarter-position kernel unless WMD_LIST_FAMILY=17)
            if (!cfg_serves_list(c, force >= 0) || c.TH != g->out_tile_h || c.TW != g->out_tile_w) continue;
            if (Cin % c.CK || (g->C2 > 0 && g->C1 % c.CK) || (g->in_mask && g->up1 == 2 && !g->in_mask_2x2)) continue;   // wino32_pure
        }
        if (force >= 0 && force != i) continue;
        const bool can_ticket = c.TAPS >= 17 && !g->gate && (double)g->Cout * g->H * g->W * 4.0 < 2147483647.0;   // (32-bit buffer offsets inside a frame)
        if (force_raw < 0 && (!can_ticket || force_ks < 2)) continue;   // "-k" names the second-stage form of a kernel that has both
        const bool ticket = can_ticket && ticket_on && force_raw >= 0;
        const int tiles_x = (W + c.TW - 1) / c.TW, tiles_y = (H + c.TH - 1) / c.TH;
        const long tiles = (long)g->B * tiles_x * tiles_y;
        const int cob = (ncot + c.WM * c.MR - 1) / (c.WM * c.MR);
        const long blocks = tiles * cob;
        const int nchunks = (Cin + c.CK - 1) / c.CK;
        const int waves = c.WM * c.WN;
        // blocks resident per CU: LDS (160 KiB) and ~2 waves/SIMD of these register-heavy kernels
        int bpc = std::min(160 * 1024 / std::max(c.lds_bytes, 1), std::max(1, (c.TAPS == 18 ? 12 : 8) / waves));
        bpc = std::max(bpc, 1);
        const double block_macs_per_chunk = (double)(c.WM * c.MR * 16) * (c.WN * c.NR * 16) * c.CK * c.TAPS;   // MFMA work (Winograd: 16 positions x 16 tiles)
        if (g->out_tiles) {
            // the device picks the split (list_ksplit) up to ksmax slices = gridDim.z: as many as could matter if every
            // slice of every possible item had to find a workgroup slot (a batch of frames never splits)
            // (measured at one frame: leaving the 8- and 12-chunk layers of the finest level unsplit -- no second pass -- cost
            //  20.4 + 24.7 us against 12.4 + 4.9 + 16.4 + 6.7 us split and reduced: the split stays on offer wherever it fits)
            const int ksmax = (int)std::max<long>(1, std::min<long>(std::min(nchunks / 2, 16), 2L * kNumCU * bpc / std::max<long>(blocks, 1)));
            const size_t need = ksmax > 1 ? (size_t)ksmax * g->B * g->Cout * g->H * g->W : 0;
            if (ksmax > 1 && (!have_ws || need > ws_floats)) continue;
            found = true;
            plan->cfg = &c;
            plan->ticket = ticket;
            plan->ksplit = ksmax;
            plan->chunks_per_split = nchunks;
            plan->nchunks = nchunks;
            plan->tiles_x = tiles_x;
            plan->tiles_y = tiles_y;
            plan->H = H;
            plan->W = W;
            plan->workspace_floats = need;
            break;
        }
        for (int ks = 1; ks <= 32; ++ks) {
            if (force_ks > 0 ? ks != force_ks : (ks & (ks - 1)) != 0 || ks > 16) continue;  // model: powers of two
            if (ks > 1 && (!have_ws || nchunks < ks)) continue;
            const int cps = (nchunks + ks - 1) / ks;
            const int ks_eff = (nchunks + cps - 1) / cps;
            if (ks > 1 && (size_t)ks_eff * g->B * g->Cout * g->H * g->W > ws_floats) continue;
            const long nblk = blocks * ks_eff;
            const long rounds = (nblk + (long)kNumCU * bpc - 1) / ((long)kNumCU * bpc);
            // waves sharing one CU's four matrix pipes in a full round
            const double conc = std::min<double>(bpc, (double)nblk / kNumCU) * waves;
            const double rate = 128.0 * std::min(1.0, conc / 4.0);  // MAC / clk / CU
            double per_cu_blocks = (double)rounds * std::min<double>(bpc, std::max(1.0, (double)nblk / kNumCU / rounds));
            double cycles = per_cu_blocks * block_macs_per_chunk * cps / std::max(rate, 1.0);
            cycles += 3000.0 * rounds;            // prologue/epilogue per block round
            if (c.TAPS == 16) cycles *= 1.3;      // transforms + 1.5 LDS reads per MFMA: measured, not modelled
            if (c.TAPS >= 17) cycles *= 0.8;      // 32x32x2 forms: fewer MFMAs on upsampled operands, lighter issue stream
            if (ks_eff > 1) cycles += ticket ? 2500.0 : 6000.0;     // the last block's finish / the reduce pass launch
            if (cycles < best) {
                best = cycles;
                found = true;
                plan->cfg = &c;
                plan->ticket = ticket;
                plan->ksplit = ks_eff;
                plan->chunks_per_split = cps;
                plan->nchunks = nchunks;
                plan->tiles_x = tiles_x;
                plan->tiles_y = tiles_y;
                plan->H = H;
                plan->W = W;
                plan->workspace_floats = ks_eff > 1 ? (size_t)ks_eff * g->B * g->Cout * g->H * g->W : 0;
            }
        }
    }
    return found;
}

}  // namespace wmd

using namespace wmd;

extern "C" int wmd_head_fused_fwd(const wmd_head_fused_args* g, void* stream);

extern "C" int wmd_head_fused_multi_fwd(const wmd_head_fused_args* levels, int n_levels, void* stream) {
    if (!levels || n_levels < 1 || n_levels > 3) return fail(WMD_ERR_BAD_ARG, "wmd_head_fused_multi_fwd: 1..3 levels (got %d)", n_levels);
    for (int k = 0; k < n_levels; ++k) {
        const wmd_head_fused_args* g = &levels[k];
        if (!g->x || !g->wp1 || !g->wp2 || !g->t) return fail(WMD_ERR_BAD_ARG, "wmd_head_fused_multi_fwd: level %d: null tensor pointer", k);
        if (g->B <= 0 || g->H <= 0 || g->W <= 0) return fail(WMD_ERR_BAD_SHAPE, "wmd_head_fused_multi_fwd: level %d: B=%d H=%d W=%d", k, g->B, g->H, g->W);
    }
    if (head_chain_multi_launch(levels, n_levels, (hipStream_t)stream)) return check_launch("head_chain_multi_kernel");
    for (int k = 0; k < n_levels; ++k) {      // a level the merged launch cannot take: the per-level launches, same planes
        const int st = wmd_head_fused_fwd(&levels[k], stream);
        if (st) return st;
    }
    return WMD_OK;
}

extern "C" int wmd_head_fused_fwd(const wmd_head_fused_args* g, void* stream) {
    if (!g) return fail(WMD_ERR_BAD_ARG, "wmd_head_fused_fwd: null args");
    if (!g->x || !g->wp1 || !g->wp2 || !g->t) return fail(WMD_ERR_BAD_ARG, "wmd_head_fused_fwd: null tensor pointer");
    if (g->B <= 0 || g->H <= 0 || g->W <= 0) return fail(WMD_ERR_BAD_SHAPE, "wmd_head_fused_fwd: B=%d H=%d W=%d", g->B, g->H, g->W);
    if (g->C != 32 && g->C != 64 && g->C != 128 && g->C != 256)
        return fail(WMD_ERR_UNSUPPORTED, "wmd_head_fused_fwd: C=%d (32, 64, 128 or 256; other widths run unfused)", g->C);
    if (g->chain != 0 && g->chain != 1) return fail(WMD_ERR_BAD_ARG, "wmd_head_fused_fwd: chain=%d", g->chain);
    if (g->t_planes != 0 && g->t_planes != 54 && g->t_planes != 81) return fail(WMD_ERR_BAD_ARG, "wmd_head_fused_fwd: t_planes=%d", g->t_planes);
    const int t_planes = g->t_planes ? g->t_planes : 54;
    if (g->chain == 1 && (g->C != 256 || t_planes != 81))
        return fail(WMD_ERR_UNSUPPORTED, "wmd_head_fused_fwd: the low-pass chain needs C = 256 and an 81-plane t (C=%d, t_planes=%d)", g->C, t_planes);
    const bool want_ll = g->chain == 0 && g->ll_wp1 && g->ll_wp2;
    if (want_ll && (g->C != 256 || t_planes != 81))
        return fail(WMD_ERR_UNSUPPORTED, "wmd_head_fused_fwd: the low-pass chain needs C = 256 and an 81-plane t (C=%d, t_planes=%d)", g->C, t_planes);
    if (g->mid_out && (g->chain != 0 || g->mid_ct <= 0 || g->mid_off_p < 0 || g->mid_off_n < 0 || g->mid_off_p + g->C > g->mid_ct ||
                       g->mid_off_n + g->C > g->mid_ct || (want_ll && (g->mid_off_ll < 0 || g->mid_off_ll + g->C / 4 > g->mid_ct))))
        return fail(WMD_ERR_BAD_ARG, "wmd_head_fused_fwd: mid_out needs chain = 0 and channel offsets inside mid_ct = %d", g->mid_ct);
    if (g->chain == 0) {
        const int took = head_chain_launch(g, t_planes, (hipStream_t)stream);
        int st = took ? check_launch("head_chain_kernel") : WMD_OK;
        if (st) return st;
        if (g->mid_out && (!took || (want_ll && took != 2)))
            return fail(WMD_ERR_UNSUPPORTED, "wmd_head_fused_fwd: mid_out is written by the chained kernel only (C = 64, 128, 256, H*W %% 4 == 0)");
        if (want_ll && took != 2) {   // the chained kernel did not take the low-pass chain along: a launch of its own
            wmd_head_fused_args l = *g;
            l.wp1 = g->ll_wp1;
            l.bias1 = g->ll_bias1;
            l.wp2 = g->ll_wp2;
            l.chain = 1;
            l.run_mask = nullptr;
            l.ll_wp1 = l.ll_wp2 = l.ll_bias1 = nullptr;
            st = wmd_head_fused_fwd(&l, stream);
            if (st) return st;
        }
        if (took) return WMD_OK;
    }
    const int rows = g->chain == 1 ? g->C / 4 : 2 * g->C;   // stacked mid channels of this launch
    ConvKArgs a;
    memset(&a, 0, sizeof(a));
    a.x1 = g->x;
    a.wp = g->wp1;
    a.bias = g->bias1;
    a.B = g->B;
    a.H = 1;
    a.W = g->H * g->W;   // 1x1: the image is a flat pixel vector
    a.H1 = 1;
    a.W1 = a.W;
    a.C1 = g->C;
    a.Cin = g->C;
    a.Cout = rows;
    a.up1 = 1;
    a.act = WMD_ACT_LEAKY;
    a.slope = g->slope;
    a.nci4 = ((g->C + 15) / 16) * 4;
    a.ncot = rows / 16;
    a.nchunks = g->C >= 128 ? g->C / 32 : 1;   // C = 32 / 64: the whole reduction is one LDS-resident chunk
    a.ksplit = 1;
    a.chunks_per_split = a.nchunks;
    a.tiles_y = 1;
    a.wp2 = g->wp2;
    a.t = g->t;
    a.t_ctot = t_planes;
    a.t_row0 = g->chain == 1 ? 54 : 0;
    hipStream_t s = (hipStream_t)stream;
    const double pix = (double)g->B * g->H * g->W;
    if (g->chain == 1) {
        // 64 mid channels = one 16-row tile per wave of a 4-wave block; the second GEMM (K = 64) gives rows 0..8
        ProfScope prof("conv_fwd_kernel<fused LL head>", 2.0 * pix * (g->C * (g->C / 4.0) + 9.0 * (g->C / 4.0)), 4.0 * pix * (g->C + 9), s);
        a.tiles_x = (a.W + 31) / 32;
        hipLaunchKernelGGL((conv_fwd_kernel<1, 32, 1, 2, 4, 1, 32, 1, true>), dim3(g->B * a.tiles_x, 1), dim3(256), 0, s, a);
        return check_launch("conv_fwd_kernel<fused LL head>");
    }
    ProfScope prof("conv_fwd_kernel<fused head>", 2.0 * pix * (2.0 * g->C * g->C + 54.0 * g->C),
                   4.0 * pix * (g->C + 54), s);
    if (g->C == 32) {
        a.tiles_x = (a.W + 255) / 256;
        hipLaunchKernelGGL((conv_fwd_kernel<1, 256, 2, 4, 1, 4, 32, 1, true, 1>), dim3(g->B * a.tiles_x, 2), dim3(256), 0, s, a);
    } else if (g->C == 64) {
        a.tiles_x = (a.W + 127) / 128;
        hipLaunchKernelGGL((conv_fwd_kernel<1, 128, 2, 4, 2, 2, 64, 1, true, 1>), dim3(g->B * a.tiles_x, 2), dim3(256), 0, s, a);
    } else if (g->C == 128) {
        // coarse levels have few pixels: 64- / 32-pixel tiles (53 / 78 KB of LDS: 3 / 2 blocks per CU, 720 / 360 blocks) beat
        // 128- / 64-pixel tiles by 20 % / 8 % although every weight fragment then feeds half as many MFMAs
        a.tiles_x = (a.W + 63) / 64;
        hipLaunchKernelGGL((conv_fwd_kernel<1, 64, 4, 2, 2, 2, 32, 1, true>), dim3(g->B * a.tiles_x, 2), dim3(256), 0, s, a);
    } else {
        a.tiles_x = (a.W + 31) / 32;
        hipLaunchKernelGGL((conv_fwd_kernel<1, 32, 4, 2, 4, 1, 32, 1, true>), dim3(g->B * a.tiles_x, 2), dim3(256), 0, s, a);
    }
    return check_launch("conv_fwd_kernel<fused head>");
}

extern "C" int wmd_conv_num_configs(void) { return kNumCfgs; }
extern "C" const char* wmd_conv_config_name(int i) { return (i >= 0 && i < kNumCfgs) ? kCfgs[i].name : nullptr; }

extern "C" size_t wmd_conv_packed_weight_floats(int Cout, int Cin, int ksize) {
    const int taps = ksize == 3 ? 9 : 1;
    // rows and reduction are both padded to 16, so the forward and the dgrad image have the same size
    const size_t ncot = (Cout + 15) / 16;
    const size_t nci4 = ((Cin + 15) / 16) * 4;
    return ncot * nci4 * taps * 64;
}

static int pack_common(const float* w, float* wp, int Cout, int Cin, int ksize, int dgrad, void* stream) {
    if (!w || !wp) return fail(WMD_ERR_BAD_ARG, "wmd_conv_pack_weights: null pointer");
    if (Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3))
        return fail(WMD_ERR_BAD_SHAPE, "wmd_conv_pack_weights: Cout=%d Cin=%d ksize=%d", Cout, Cin, ksize);
    const int taps = ksize == 3 ? 9 : 1;
    const int rows = dgrad ? Cin : Cout, red = dgrad ? Cout : Cin;
    const int ncot = (rows + 15) / 16, nci4 = ((red + 15) / 16) * 4;
    const size_t total = (size_t)ncot * nci4 * taps * 64;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    ProfScope prof("conv_pack_kernel", 0.0, 8.0 * total, (hipStream_t)stream);
    hipLaunchKernelGGL(conv_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wp, Cout, Cin, taps, ncot,
                       nci4, dgrad);
    return check_launch("conv_pack_kernel");
}

extern "C" int wmd_conv_pack_weights(const float* w, float* wp, int Cout, int Cin, int ksize, void* stream) {
    return pack_common(w, wp, Cout, Cin, ksize, 0, stream);
}
extern "C" int wmd_conv_pack_weights_dgrad(const float* w, float* wp, int Cout, int Cin, int ksize, void* stream) {
    return pack_common(w, wp, Cout, Cin, ksize, 1, stream);
}

extern "C" size_t wmd_conv_packed_weight_floats_wino(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    const size_t ncot = (Cout + 15) / 16;
    const size_t nci4 = ((Cin + 15) / 16) * 4;
    return ncot * nci4 * 16 * 64;
}

extern "C" int wmd_conv_pack_weights_wino(const float* w, float* wp, int Cout, int Cin, int dgrad, void* stream) {
    if (!w || !wp) return fail(WMD_ERR_BAD_ARG, "wmd_conv_pack_weights_wino: null pointer");
    if (Cout <= 0 || Cin <= 0) return fail(WMD_ERR_BAD_SHAPE, "wmd_conv_pack_weights_wino: Cout=%d Cin=%d", Cout, Cin);
    const int rows = dgrad ? Cin : Cout, red = dgrad ? Cout : Cin;
    const int ncot = (rows + 15) / 16, nci4 = ((red + 15) / 16) * 4;
    const size_t total = (size_t)ncot * nci4 * 16 * 64;
    ProfScope prof("conv_pack_wino_kernel", 0.0, 8.0 * total, (hipStream_t)stream);
    hipLaunchKernelGGL(conv_pack_wino_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, w,
                       wp, Cout, Cin, ncot, nci4, dgrad ? 1 : 0);
    return check_launch("conv_pack_wino_kernel");
}

// ------------------------------------------------------------------------------------------------
// every weight image of a set of filters in one launch (a training step repacks all of them after each optimizer update:
// as ~50 separate 5-10 us launches that was 9 % of the decoder's forward + backward).
// One workgroup = one 16 x 16 (Cout x Cin) tile of one filter, all taps: the 16 rows are 16*taps contiguous floats each,
// staged once in LDS, and every requested image of that tile -- forward and data-gradient fragment order, direct and
// Winograd -- is a contiguous 4*taps*64 (4*16*64) float run written from there.  Same arithmetic, in the same order, as
// conv_pack_kernel / conv_pack_wino_kernel: the images are bit-identical.
// ------------------------------------------------------------------------------------------------
constexpr int kPackManyMax = 40;
struct PackManyItem {
    const float* w;
    float *fwd, *dgrad, *wfwd, *wdgrad;
    int Cout, Cin, taps, tile0;   // tile0 = index of this item's first tile in the launch
};
struct PackManyArgs {
    int n, tiles;
    PackManyItem it[kPackManyMax];
};

__global__ __launch_bounds__(256) void conv_pack_many_kernel(const PackManyArgs a) {
    __shared__ float s[16 * 144];
    const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const int tid = threadIdx.x;
    for (int tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
        int k = 0;
        while (k + 1 < a.n && a.it[k + 1].tile0 <= tile) ++k;
        const PackManyItem& it = a.it[k];
        const int taps = it.taps, Cout = it.Cout, Cin = it.Cin;
        const int nb = (Cin + 15) / 16, na = (Cout + 15) / 16;
        const int ta = (tile - it.tile0) / nb, tb = (tile - it.tile0) % nb;   // Cout tile, Cin tile
        const int row_len = 16 * taps;
        __syncthreads();
        {
            const int r = tid >> 4, m = ta * 16 + r;
            const float* src = it.w + ((size_t)m * Cin + tb * 16) * taps;
            const int valid = m < Cout ? (min(16, Cin - tb * 16)) * taps : 0;
            for (int c = tid & 15; c < row_len; c += 16) s[r * 144 + c] = c < valid ? src[c] : 0.f;
        }
        __syncthreads();
        const size_t fbase = ((size_t)ta * nb * 4 + 4 * tb) * 64;   // x taps (or x 16): first float of the tile's run, forward order
        const size_t dbase = ((size_t)tb * na * 4 + 4 * ta) * 64;   // data-gradient order: rows = Cin tile, reduction = Cout tile
        const int n_direct = 4 * taps * 64;
        if (it.fwd)
            for (int i = tid; i < n_direct; i += 256) {
                const int l = i & 63, r = i >> 6, tap = r % taps, q = r / taps;
                it.fwd[fbase * taps + i] = s[(l & 15) * 144 + (q * 4 + (l >> 4)) * taps + tap];
            }
        if (it.dgrad)
            for (int i = tid; i < n_direct; i += 256) {
                const int l = i & 63, r = i >> 6, tap = r % taps, q = r / taps;
                it.dgrad[dbase * taps + i] = s[(q * 4 + (l >> 4)) * 144 + (l & 15) * taps + (taps - 1 - tap)];
            }
        if (taps == 9 && (it.wfwd || it.wdgrad))
            for (int i = tid; i < 4 * 16 * 64; i += 256) {
                const int l = i & 63, r = i >> 6, xi = r & 15, q = r >> 4;
                const int pa = xi >> 2, pb = xi & 3;
                if (it.wfwd) {
                    const float* g = &s[(l & 15) * 144 + (q * 4 + (l >> 4)) * 9];
                    float u = 0.f;
                    for (int ii = 0; ii < 3; ++ii)
                        for (int jj = 0; jj < 3; ++jj) u += G[pa][ii] * g[ii * 3 + jj] * G[pb][jj];
                    it.wfwd[fbase * 16 + i] = u;
                }
                if (it.wdgrad) {
                    const float* g = &s[(q * 4 + (l >> 4)) * 144 + (l & 15) * 9];
                    float u = 0.f;
                    for (int ii = 0; ii < 3; ++ii)
                        for (int jj = 0; jj < 3; ++jj) u += G[pa][ii] * g[8 - (ii * 3 + jj)] * G[pb][jj];
                    it.wdgrad[dbase * 16 + i] = u;
                }
            }
    }
}

extern "C" int wmd_conv_pack_many(const wmd_pack_item* items, int n, void* stream) {
    if (n < 0 || (n && !items)) return fail(WMD_ERR_BAD_ARG, "wmd_conv_pack_many: null items");
    for (int i = 0; i < n; ++i) {
        const wmd_pack_item& p = items[i];
        if (!p.w) return fail(WMD_ERR_BAD_ARG, "wmd_conv_pack_many: item %d has no weights", i);
        if (p.Cout <= 0 || p.Cin <= 0 || (p.ksize != 1 && p.ksize != 3))
            return fail(WMD_ERR_BAD_SHAPE, "wmd_conv_pack_many: item %d Cout=%d Cin=%d ksize=%d", i, p.Cout, p.Cin, p.ksize);
        if (p.ksize != 3 && (p.wino_fwd || p.wino_dgrad))
            return fail(WMD_ERR_BAD_ARG, "wmd_conv_pack_many: item %d asks for a Winograd image of a %dx%d filter", i, p.ksize, p.ksize);
    }
    for (int i0 = 0; i0 < n; i0 += kPackManyMax) {
        PackManyArgs a;
        a.n = std::min(kPackManyMax, n - i0);
        int tiles = 0;
        double floats = 0;
        for (int i = 0; i < a.n; ++i) {
            const wmd_pack_item& p = items[i0 + i];
            a.it[i] = PackManyItem{p.w, p.fwd, p.dgrad, p.wino_fwd, p.wino_dgrad, p.Cout, p.Cin, p.ksize == 3 ? 9 : 1, tiles};
            tiles += ((p.Cout + 15) / 16) * ((p.Cin + 15) / 16);
            const double img = (double)wmd_conv_packed_weight_floats(p.Cout, p.Cin, p.ksize);
            floats += (double)p.Cout * p.Cin * (p.ksize == 3 ? 9 : 1) + img * ((p.fwd != nullptr) + (p.dgrad != nullptr)) +
                      img / 9 * 16 * ((p.wino_fwd != nullptr) + (p.wino_dgrad != nullptr));
        }
        a.tiles = tiles;
        ProfScope prof("conv_pack_many_kernel", 0.0, 4.0 * floats, (hipStream_t)stream);
        hipLaunchKernelGGL(conv_pack_many_kernel, dim3(std::min(tiles, 8192)), dim3(256), 0, (hipStream_t)stream, a);
        int rc = check_launch("conv_pack_many_kernel");
        if (rc) return rc;
    }
    return WMD_OK;
}

extern "C" int wmd_conv_list_tile_supported(int tile_h, int tile_w) {
    for (int i = 0; i < kNumCfgs; ++i) {
        const ConvCfg& c = kCfgs[i];
        if (c.TH == tile_h && c.TW == tile_w && cfg_serves_list(c, false)) return 1;
    }
    return 0;
}

static int validate_conv(const wmd_conv_args* g, const char* who) {
    if (!g) return fail(WMD_ERR_BAD_ARG, "%s: null args", who);
    if (!g->x1 || !g->wp || !g->y) return fail(WMD_ERR_BAD_ARG, "%s: null tensor pointer", who);
    if (g->C2 > 0 && !g->x2) return fail(WMD_ERR_BAD_ARG, "%s: C2=%d but x2 is null", who, g->C2);
    if (g->B <= 0 || g->H <= 0 || g->W <= 0 || g->C1 <= 0 || g->C2 < 0 || g->Cout <= 0)
        return fail(WMD_ERR_BAD_SHAPE, "%s: B=%d H=%d W=%d C1=%d C2=%d Cout=%d", who, g->B, g->H, g->W, g->C1, g->C2,
                    g->Cout);
    if (g->ksize != 1 && g->ksize != 3) return fail(WMD_ERR_UNSUPPORTED, "%s: ksize=%d", who, g->ksize);
    if (g->up1 != 1 && g->up1 != 2) return fail(WMD_ERR_BAD_ARG, "%s: up1=%d", who, g->up1);
    if (g->up1 == 2 && ((g->H | g->W) & 1)) return fail(WMD_ERR_BAD_SHAPE, "%s: up1=2 needs even H,W", who);
    if (g->pad_mode < 0 || g->pad_mode > 2) return fail(WMD_ERR_BAD_ARG, "%s: pad_mode=%d", who, g->pad_mode);
    if (g->act < 0 || g->act > 3) return fail(WMD_ERR_BAD_ARG, "%s: act=%d", who, g->act);
    // ReflectionPad2d(1) requires every padded dimension to be >= 2 (torch raises otherwise)
    if (g->ksize == 3 && g->pad_mode == WMD_PAD_REFLECT && (g->H < 2 || g->W < 2))
        return fail(WMD_ERR_BAD_SHAPE, "%s: reflect padding needs H,W >= 2 (got %dx%d)", who, g->H, g->W);
    if (g->out_tiles) {   // work-list form
        if (!g->out_tile_count || !g->out_mask || g->ksize != 3 || !g->wp_wino || g->gate)
            return fail(WMD_ERR_BAD_ARG, "%s: a tile list needs its count, out_mask, a 3x3 layer with wp_wino and no gate", who);
        if (!wmd_conv_list_tile_supported(g->out_tile_h, g->out_tile_w))
            return fail(WMD_ERR_UNSUPPORTED, "%s: no work-list kernel for %dx%d tiles", who, g->out_tile_h, g->out_tile_w);
    }
    return WMD_OK;
}

extern "C" size_t wmd_conv_fwd_workspace_floats(const wmd_conv_args* g) {
    if (!g || g->B <= 0 || g->Cout <= 0) return 0;
    ConvPlan plan;
    if (!plan_conv(g, &plan, true, (size_t)-1)) return 0;
    return plan.workspace_floats;
}

namespace wmd {
int run_conv(const wmd_conv_args* g, int shift1, int H1, int W1, void* stream);
}

extern "C" int wmd_conv_fwd(const wmd_conv_args* g, void* stream) {
    int st = validate_conv(g, "wmd_conv_fwd");
    if (st) return st;
    return run_conv(g, 0, g->H / g->up1, g->W / g->up1, stream);
}

// The ticket ring of the in-kernel split-K finish: 1 Mi counters, zero-filled once (outside any capture), handed out in
// consecutive regions.
static int* ticket_region(size_t items, hipStream_t stream) {
    constexpr size_t kRing = 1u << 20;
    static int* ring = nullptr;
    static size_t next = 0;
    static bool failed = false;
    if (items == 0 || items > kRing || failed) return nullptr;
    if (!ring) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return nullptr;   // first use inside a capture: no allocation here (the decoders warm up eagerly first)
        }
        if (hipMalloc(&ring, kRing * sizeof(int)) != hipSuccess || hipMemset(ring, 0, kRing * sizeof(int)) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess) {
            (void)hipGetLastError();
            ring = nullptr;
            failed = true;
            return nullptr;
        }
    }
    if (next + items > kRing) next = 0;
    int* r = ring + next;
    next += items;
    return r;
}

// Shared by the forward pass and by the data-gradient pass (which feeds dz through the same kernel with
// transposed/flipped weights, shift1 = 1 and an (H+2) x (W+2) logical extent).
int wmd::run_conv(const wmd_conv_args* g, int shift1, int H1, int W1, void* stream) {
    int st = WMD_OK;
    // buffer descriptors address one image of a source tensor with 32-bit byte offsets below 2^31
    const double lim = 2147483647.0;
    if ((double)g->C1 * H1 * W1 * 4 > lim || (double)g->C2 * g->H * g->W * 4 > lim ||
        (double)wmd_conv_packed_weight_floats(g->Cout, g->C1 + g->C2, g->ksize) * 4 > lim ||
        (double)g->Cout * g->H * g->W * 4 > lim)
        return fail(WMD_ERR_UNSUPPORTED, "wmd_conv: a per-image tensor slice or the weight image exceeds 2 GiB");
    ConvPlan plan;
    if (!plan_conv(g, &plan, g->workspace != nullptr, g->workspace_floats)) return fail(WMD_ERR_UNSUPPORTED, "wmd_conv_fwd: no kernel configuration");
    const ConvCfg& c = *plan.cfg;
    const bool wino = c.TAPS >= 16;           // Winograd configuration (16: 16x16x4 kernels, 17: 32x32x2 kernels): transformed weight image, 3x3 semantics
    const int taps = wino ? 9 : c.TAPS;
    ConvKArgs a;
    memset(&a, 0, sizeof(a));
    a.x1 = g->x1;
    a.x2 = g->C2 > 0 ? g->x2 : nullptr;
    a.wp = wino ? g->wp_wino : g->wp;
    a.bias = g->bias;
    a.B = g->B;
    a.H = plan.H;
    a.W = plan.W;
    a.C1 = g->C1;
    a.C2 = g->C2;
    a.Cin = g->C1 + g->C2;
    a.Cout = g->Cout;
    a.up1 = taps == 9 ? g->up1 : 1;
    if (taps == 1 && g->up1 == 2) return fail(WMD_ERR_UNSUPPORTED, "wmd_conv_fwd: 1x1 with upsampled input");
    a.shift1 = shift1;
    a.H1 = taps == 9 ? H1 : 1;
    a.W1 = taps == 9 ? W1 : H1 * W1;
    a.pad_mode = g->pad_mode;
    a.act = g->act;
    a.slope = g->slope;
    a.tiles_x = plan.tiles_x;
    a.tiles_y = plan.tiles_y;
    a.nci4 = ((a.Cin + 15) / 16) * 4;
    a.ncot = (g->Cout + 15) / 16;
    a.nchunks = plan.nchunks;
    a.ksplit = plan.ksplit;
    a.chunks_per_split = plan.chunks_per_split;
    a.y = plan.ksplit > 1 ? g->workspace : g->y;
    // split-K finished inside the convolution: one self-resetting ticket counter per (tile, slab) item from the library's ring
    // (the last arriver of an item re-arms its counter, so a region is all zero whenever no launch is using it; consecutive
    // launches take consecutive regions, so launches in flight on different streams -- and the nodes of a captured graph, whose
    // region is part of the captured arguments -- never share one).  No ring (allocation refused, or the first use would fall
    // inside a stream capture): the second-stage kernels run as before.
    bool ticket = plan.ticket && plan.ksplit > 1;
    if (ticket) {
        const int cob_t = (a.ncot + c.WM * c.MR - 1) / (c.WM * c.MR);
        const size_t items = (size_t)g->B * plan.tiles_x * plan.tiles_y * cob_t;
        a.tickets = ticket_region(items, (hipStream_t)stream);
        if (!a.tickets) ticket = false;
        else a.y_final = g->y;
    }
    static const int no_x4 = env_int("WMD_X4", 1) == 0;
    a.no_x4 = no_x4;
    static const int st_coalesce = env_int("WMD_W32_COALESCE", 1);
    a.st_coalesce = st_coalesce;
#ifdef WMD_STAMPS
    if (const char* e = getenv("WMD_DBG_PTR")) a.dbg = (unsigned long long*)strtoull(e, nullptr, 0);
    a.dbg_mode = env_int("WMD_DBG_MODE", 0);
#endif
    const bool list = g->out_tiles != nullptr;
    if (list) {
        a.tile_list = g->out_tiles;
        a.tile_count = g->out_tile_count;
        a.ksmax = plan.ksplit;
        a.list_slots = kNumCU * std::max(1, std::min(160 * 1024 / std::max(c.lds_bytes, 1), std::max(1, (c.TAPS == 18 ? 12 : 8) / (c.WM * c.WN))));
        a.y_final = g->y;
    }
    if (ticket) a.y_final = g->y;
    a.gate = g->gate;
    a.gate_act = g->gate_act;
    a.gate_slope = g->gate_slope;
    a.in_mask = g->in_mask;
    a.out_mask = g->out_mask;
    a.in_mask_2x2 = g->in_mask_2x2;
    if ((g->out_mask || g->in_mask) && !wino)
        return fail(WMD_ERR_UNSUPPORTED, "wmd_conv: masks (block-sparse execution) need a 3x3 layer and the Winograd weight image (wp_wino)");
    const int cob = (a.ncot + c.WM * c.MR - 1) / (c.WM * c.MR);
    // Item order: slabs of a tile back to back on one XCD (they share the gathered input patch) while the whole weight image
    // stays resident in an XCD's 4 MB L2; for the coarse layers, whose weights are larger than that and whose patches are
    // small, the tiles of one slab run together instead (they share that slab's weights) -- the 2-D grid.
    const double wbytes = (double)a.ncot * 16 * a.nci4 * 4 * (wino ? 16 : taps) * 4;
    const bool tile_major = list || (cob > 1 && wbytes <= 3.0 * 1024 * 1024);   // (list items are always (tile, slab) pairs)
    a.cob = tile_major ? cob : 0;
    // the coarse layers' 2-D grid: a slab per XCD when the slabs divide over the eight XCDs, every z-slice starts on XCD 0 and the
    // weight image outweighs the input maps (they are then what each XCD re-reads; WMD_XCD_SLAB=0/1 forces it off / on)
    {
        static const int xs = env_int("WMD_XCD_SLAB", -1);
        const size_t tiles_n = (size_t)g->B * plan.tiles_x * plan.tiles_y;
        const double in_bytes = 4.0 * g->B * ((double)g->C1 * H1 * W1 + (double)g->C2 * g->H * g->W);
        const bool fits = !tile_major && c.TAPS >= 17 && cob % 8 == 0 && (tiles_n * cob) % 8 == 0;
        a.xcd_slab = fits && (xs >= 0 ? xs != 0 : wbytes > in_bytes) ? 1 : 0;
    }
    dim3 grid((unsigned)((size_t)g->B * plan.tiles_x * plan.tiles_y * (tile_major ? cob : 1)), tile_major ? 1u : (unsigned)cob,
              (unsigned)plan.ksplit);
    if (env_int("WMD_CONV_VERBOSE", 0))
        // (sig = the autotuner's problem signature: profiles key the counters of a launch by it, see tools/make_profile_summary.py)
        fprintf(stderr, "[wmd] conv %dx%d C%d+%d->%d k%d: cfg %s grid %u,%u,%u block %d sig conv|%d|%d|%d|%d|%d|%d|%d|%d%s\n", g->H, g->W, g->C1,
                g->C2, g->Cout, g->ksize, c.name, grid.x, grid.y, grid.z, c.WM * c.WN * 64, g->B, g->H, g->W, g->C1, g->up1, g->C2, g->Cout,
                g->ksize, shift1 ? "|dgrad" : "");
    const double pix = (double)g->B * g->H * g->W;
    {
        // algorithmic work: 2*Cin*k*k*Cout FLOP per output pixel; bytes: inputs + weights + outputs once
        ProfScope prof(c.name, 2.0 * a.Cin * taps * g->Cout * pix,
                       4.0 * (pix * g->C1 / (a.up1 * a.up1) + pix * g->C2 + (double)a.Cin * taps * g->Cout + pix * g->Cout),
                       (hipStream_t)stream);
        if (c.TAPS == 16) prof.mfma(2.0 * a.Cin * 9 * g->Cout * pix / 2.25);
        if (c.TAPS >= 17) {   // 32x32x2 MFMAs of 4096 FLOP: per block its tile groups x chunks x K-steps x positions
            const int n_up = (wino32_pure(a, c.CK) && a.up1 == 2) ? a.C1 / c.CK : 0;
            const int groups = c.TAPS == 17 ? c.WN / 2 : 1;
            prof.mfma(4096.0 * grid.x * grid.y * groups * (c.CK / 2) * (9.0 * n_up + 16.0 * (plan.nchunks - n_up)));
        }
        c.launch(a, grid, (hipStream_t)stream);
    }
    st = check_launch("conv_fwd_kernel");
    if (st) return st;
    if (ticket) return st;   // the last K-slice block of every item has written the final values
    if (list) {
        if (plan.ksplit > 1) {   // (ksmax = 1: the device can never split, no second pass)
            const int ntiles = g->B * plan.tiles_x * plan.tiles_y;
            ProfScope prof("conv_splitk_reduce_list_kernel", 0.0, 0.0, (hipStream_t)stream);
            hipLaunchKernelGGL(conv_splitk_reduce_list_kernel, dim3(ntiles, (g->Cout + kRedCh - 1) / kRedCh), dim3(256), 0, (hipStream_t)stream, g->workspace, g->bias, g->y,
                               g->out_tiles, g->out_tile_count, g->out_mask, g->B, g->Cout, g->H, g->W, c.TH, c.TW, plan.tiles_x,
                               plan.tiles_y, cob, plan.nchunks, plan.ksplit, a.list_slots, g->act, g->slope);
            st = check_launch("conv_splitk_reduce_list_kernel");
        }
        return st;
    }
    if (plan.ksplit > 1) {
        const size_t n = (size_t)g->B * g->Cout * g->H * g->W;
        const int blocks = (int)std::min<size_t>((n + 255) / 256, 2048);
        ProfScope prof("conv_splitk_reduce_kernel", (double)n * plan.ksplit, 4.0 * n * (plan.ksplit + 1), (hipStream_t)stream);
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g->workspace,
                           g->bias, g->y, n, (size_t)g->H * g->W, g->Cout, plan.ksplit, g->act, g->slope, g->gate, g->gate_act,
                           g->gate_slope, g->out_mask);
        st = check_launch("conv_splitk_reduce_kernel");
    }
    return st;
}
