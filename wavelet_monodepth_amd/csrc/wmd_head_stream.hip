// Streaming form of the finest level's wavelet heads (round 6; C = 32, inference):  same operator and the same contract as
// head_level_kernel (wmd_head_level.hip; KITTI/networks/decoders/depth_decoder.py:108-136,164-166)
//
//   mid_s = LeakyReLU(W1_s x + b1_s)            s in {+,-},  1x1, C -> C
//   h_s   = b3_s + Conv3x3_reflect(mid_s; W3_s) C -> 3            (as 27 tap-partial rows t_s = W3'_s mid_s + a 9-tap shift-sum)
//   yh    = 2^(s-1) sigmoid(h_+) - 2^(s-1) sigmoid(h_-);  out = HaarIDWT(yl, yh);  disp = clamp(out * disp_scale, 0, 1)
//
// but cut differently, because of what round 5 measured on that kernel (58 us for 30 us of MFMA work at its own 1.6x halo):
// five barriers per 4 x 40 tile with nothing overlapping them, `mid` through LDS, and a 6 x 42 patch per 160 pixels.
//
//   * A block owns a STRIP SEGMENT: 32 columns x TH rows (TH chosen by the host, 48 at 96 x 320 x 12 = one block per CU) and walks
//     its (TH + 2) x 34 halo patch as ONE flattened sequence of positions, 128 per step: the halo costs 34/32 x (TH+2)/TH = 1.1x.
//   * Eight GEMM waves (two per SIMD), each alone with 16 positions of the step: the wave stages ITS OWN x slice by LDS-DMA (32
//     channels x 16 positions, laid out as the MFMA B fragments it reads back: no barrier, no bank conflict, three buffers deep;
//     reflect padding folded incrementally: the lane's patch coordinates advance by a compare-and-carry per step), runs the
//     first product with the WEIGHTS as the A operand -- D[mid channel][position] -- so that its accumulator registers ARE the
//     A operand of the second product (the head_chain_kernel trick: K-steps run over (m, i), the W3' fragment is gathered to
//     match), applies bias (as the accumulators' initial value) + LeakyReLU in registers (4 v_mul + 4 v_max per tile in asm:
//     fmaxf costs a canonicalising third), and leaves D2[position][tap row] as 16-byte LDS writes into a RING of tap-partial
//     planes.  `mid` never touches LDS; all weights live in registers (64 + 16).
//   * Four EPILOGUE waves (one per SIMD) run one step behind on the ring: 32 anchors x 2 sides per wave, 27 LDS reads per lane
//     from three row bases (compile-time offsets), bias, sigmoid, the two sides meet through a lane shuffle, Haar butterfly, clamp,
//     stores.
//   * ONE barrier per step; GEMM of step s overlaps the epilogue of step s-1 and the DMA of step s+2.
// Every vector instruction of a GEMM wave comes out of the matrix rate (fp32 MFMA and the vector ALU are one resource): a step is
// 64 MFMAs + 60 VALU + 58 SALU + 16 LDS + 8 DMA instructions (329 before the diet).  Measured (profiles/r06_notes.md section 2):
// 64.5 (head_level_kernel) -> 57.1 (two 6-wave blocks per CU) -> 55.3 (diet) -> 49.8 us (one 12-wave block per CU: two GEMM
// waves per SIMD hide each other's LDS / barrier waits, the epilogue waves load the four SIMDs evenly).
// LDS: ring 54 planes x 356 floats (2 steps + the reach of an anchor, 70 positions; plane stride == 4 mod 32: the 8 lanes of a
// ds_write_b128 service group hit 32 distinct banks) + 3 x 8 x 2 KB of x slices = 126 KB -> one block per CU.
// yh_mask / mid_out / sig_* (sparse and training forms) and non-reflect padding stay on head_level_kernel.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "wmd_internal.h"
#include "wmd_head_shiftsum.h"

namespace wmd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void hs_dma4(__amdgpu_buffer_rsrc_t r, lds_ptr_t dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 4, voff, soff, 0, 0);
}

#ifndef WMD_HS_NG
#define WMD_HS_NG 8      // GEMM waves per block: 8 (+ 4 epilogue waves) = one 12-wave block per CU, two GEMM waves and one epilogue wave per SIMD; 4 = rounds 6a (two 6-wave blocks per CU)
#endif
constexpr int HS_C = 32, HS_TW = 32, HS_PW = HS_TW + 2;
constexpr int HS_NG = WMD_HS_NG, HS_NE = HS_NG / 2;             // GEMM waves, epilogue waves
constexpr int HS_S = HS_NG * 16;                                // positions per step
constexpr int HS_L = 2 * HS_PW + 2;                             // an anchor's taps reach L positions ahead
constexpr int HS_RING = ((2 * HS_S + HS_L + 15) / 16) * 16;     // live span: the step being written + the step being read + L
constexpr int HS_TS = ((HS_RING + 20 - 4 + 31) / 32) * 32 + 4;   // plane stride: >= RING + 4 (mirror of slots 0..3) + 16 spare, == 4 (mod 32)
constexpr int HS_XW = HS_C * 16;                                // dwords of one wave's x slice of one step
static_assert(HS_TS >= HS_RING + 4 && HS_TS % 32 == 4, "ring plane stride");
static_assert(HS_NE * 32 == HS_S, "two sides of 32 anchors per epilogue wave cover a step");

// The completions of the coarser levels (up to three, coarse to fine, each twice the size of the one before; the finest is half
// this level's size) as a PYRAMID inside this kernel: level k's synthesis output is level k+1's low-pass input pixel for pixel, and
// the last one's is THIS level's (wmd_head_level_args.yl).  A unit's 32 x TH pixels nest over 16 x TH/2, 8 x TH/4, 4 x TH/8 pixels of
// the coarser levels (TH a multiple of 8), so its four epilogue waves complete those regions first -- shiftsum_pixel, the arithmetic
// of head_shiftsum_chain_kernel: same bits -- handing the low-pass tiles down through LDS, the last one (32 x TH) being what the
// level's own epilogue reads instead of global memory.  Removes the completion launch of the dense decoder (21.7 us of latency-bound
// gathers at config 2) at the price of the GEMM waves waiting for the pyramid behind their first step.
struct HeadStreamPyr {
    wmd_head_shiftsum_args lv[3];
    int n;   // 0: no pyramid (a.yl from memory)
};
constexpr int HS_PYR_TH_MAX = 64;   // 16 x TH/2 pixels of the finest coarse level <= the GEMM waves' 512 lanes: one pixel per lane

struct HeadStreamGeom {
    int strips, segs, TH, nunits;
    int dbg;   // -DHS_DBG builds only (timing experiments, results wrong): 1 no epilogue work, 2 no DMA, 4 no MFMAs, 8 no ring stores, 16 no global stores
};
#ifdef HS_DBG
#define HS_DBG_ON(bit) (gm.dbg & (bit))
#else
#define HS_DBG_ON(bit) false
#endif

// LDS writes of this wave are complete, then the workgroup barrier (no vmcnt wait: a GEMM wave's LDS-DMA of the step after
// next stays in flight across it -- the x slices are wave-private, the compiler tracks their arrival per buffer)
__device__ __forceinline__ void hs_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// The GEMM waves' LDS traffic goes through inline asm: the compiler's wait-count pass treats every LDS access that follows an
// LDS-DMA as a possible reader of the DMA's destination and puts `s_waitcnt vmcnt(0)` in front of it (reads of the staged slice
// AND the ring writes) -- which would drain the prefetch of the step after next in every step.  Here the waves' own accounting
// applies: DMA instructions complete in issue order, so `vmcnt(N)` with N = the instructions of the newer slices in flight.
__device__ __forceinline__ unsigned hs_lds_addr(const float* p) {
    return (unsigned)(size_t)(lds_ptr_t)const_cast<float*>(p);
}
template <int NEWER>   // the eight B fragments of a staged slice: lane-linear dwords, 256 bytes apart
__device__ __forceinline__ void hs_read_slice(unsigned addr, float (&xf)[8]) {
    asm volatile(
        "s_waitcnt vmcnt(%9)\n\t"
        "ds_read_b32 %0, %8\n\t"
        "ds_read_b32 %1, %8 offset:256\n\t"
        "ds_read_b32 %2, %8 offset:512\n\t"
        "ds_read_b32 %3, %8 offset:768\n\t"
        "ds_read_b32 %4, %8 offset:1024\n\t"
        "ds_read_b32 %5, %8 offset:1280\n\t"
        "ds_read_b32 %6, %8 offset:1536\n\t"
        "ds_read_b32 %7, %8 offset:1792\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(xf[0]), "=&v"(xf[1]), "=&v"(xf[2]), "=&v"(xf[3]), "=&v"(xf[4]), "=&v"(xf[5]), "=&v"(xf[6]), "=&v"(xf[7])
        : "v"(addr), "n"(NEWER)
        : "memory");
}
// Four 16-byte ring stores of a step (tap-row tiles j = 0, 1 of both sides) in ONE asm block that takes the accumulators as
// operands, so that every second-product MFMA is issued before it -- the hazard recognizer does not protect an MFMA result
// consumed by inline asm (an 8-pass MFMA's destination may be read by an LDS instruction 11 wait states later at the earliest;
// without the s_nop the stores picked up stale registers) -- and the scheduler cannot slip an MFMA between the stores.
template <int OFF>
__device__ __forceinline__ void hs_write_tiles(unsigned a0, unsigned a1, const f32x4& p0, const f32x4& p1, const f32x4& n0, const f32x4& n1) {
    asm volatile(
        "s_nop 15\n\ts_nop 3\n\t"
        "ds_write_b128 %0, %2\n\t"
        "ds_write_b128 %1, %3\n\t"
        "ds_write_b128 %0, %4 offset:%6\n\t"
        "ds_write_b128 %1, %5 offset:%6"
        : : "v"(a0), "v"(a1), "v"(p0), "v"(p1), "v"(n0), "v"(n1), "n"(OFF) : "memory");
}

// LeakyReLU of one accumulator tile, max(v, slope v) for 0 <= slope <= 1, as 4 v_mul + 4 v_max: `fmaxf` costs a third instruction
// per value (the compiler canonicalises the MFMA result with `v_max x, x` first), and every vector instruction of a GEMM wave comes
// out of the matrix rate.  FIRST = the block directly behind the first product: the MFMA results it reads need their wait states
// (the hazard recognizer does not look into asm); every block ends with the wait states an MFMA needs before it may read `o`.
template <bool FIRST>
__device__ __forceinline__ void hs_leaky4(const f32x4& v, float slope, float (&o)[4]) {
    if constexpr (FIRST) asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
    asm volatile(
        "v_mul_f32 %0, %8, %4\n\t"
        "v_mul_f32 %1, %8, %5\n\t"
        "v_mul_f32 %2, %8, %6\n\t"
        "v_mul_f32 %3, %8, %7\n\t"
        "v_max_f32 %0, %0, %4\n\t"
        "v_max_f32 %1, %1, %5\n\t"
        "v_max_f32 %2, %2, %6\n\t"
        "v_max_f32 %3, %3, %7\n\t"
        "s_nop 1"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "s"(slope));
}

// The pyramid of one unit, run by ALL waves of the block before the unit's pipeline starts.  Every level's gathers are issued at once
// on disjoint lanes -- the finest coarse level (16 x TH/2 pixels) on the GEMM waves' 512 lanes, the one or two above it (8 x TH/4,
// 4 x TH/8) side by side on the epilogue waves' 256 -- so the memory round trip of ~60 gathers per pixel is paid ONCE; only the
// low-pass hand-over (one LDS value in, four out per pixel, a butterfly) is sequential, one block barrier per level.  (First version:
// one phase per level on the epilogue waves' lanes, loads and all: 18.5 us per unit -- as much as the launch it replaced.)
__device__ __forceinline__ void hs_pyramid(const HeadStreamPyr& pyr, int b, int y0, int x0, int TH, int tid, float (*pyr_low)[16 * (HS_PYR_TH_MAX / 2)],
                                           float* yl_tile) {
    constexpr int NGL = HS_NG * 64;
    static_assert((HS_TW / 2) * (HS_PYR_TH_MAX / 2) <= HS_NG * 64 && (HS_TW / 4) * (HS_PYR_TH_MAX / 4) + (HS_TW / 8) * (HS_PYR_TH_MAX / 8) <= HS_NE * 64, "one pixel per lane");
    const int n = pyr.n;
    const bool gemm_lane = tid < NGL;
    // job of this lane: (level, pixel index inside the unit's region of that level)
    int lvl = -1, idx = 0;
    if (gemm_lane) {
        lvl = n - 1;
        idx = tid;
    } else {
        int et = tid - NGL, base = 0;
        for (int k = n - 2; k >= 0; --k) {      // level n-2 first (the larger region), then n-3
            const int sh = n - k, cnt = (HS_TW >> sh) * (TH >> sh);
            if (lvl < 0 && et >= base && et < base + cnt) lvl = k, idx = et - base;
            base += cnt;
        }
    }
    float yh[3], l_own = 0.f;
    int py = 0, px = 0;
    bool live = false, own = false;
    if (lvl >= 0) {
        const int sh = n - lvl, rw = HS_TW >> sh, rh = TH >> sh;
        if (idx < rw * rh) {
            py = idx / rw, px = idx - py * rw;
            const int yy = (y0 >> sh) + py, xx = (x0 >> sh) + px;
            live = yy < pyr.lv[lvl].H && xx < pyr.lv[lvl].W;
            if (live) own = shiftsum_gather(pyr.lv[lvl], (size_t)b, yy, xx, yh, l_own);
        }
    }
    // hand-over, coarse to fine: level k reads the tile level k - 1 left, finishes, leaves its own 2 x 2 per pixel
    for (int k = 0; k < n; ++k) {
        if (lvl == k && live) {
            const int sh = n - k, rw = HS_TW >> sh;
            const float* lp = pyr_low[(k + 1) & 1];
            float* nl = k + 1 < n ? pyr_low[k & 1] : yl_tile;
            const int ns = rw * 2;
            const float l = own ? l_own : (k > 0 ? lp[py * rw + px] : 0.f);
            float v[4];
            shiftsum_finish(pyr.lv[k], (size_t)b, (y0 >> sh) + py, (x0 >> sh) + px, yh, l, v);
            nl[(2 * py) * ns + 2 * px] = v[0];
            nl[(2 * py) * ns + 2 * px + 1] = v[1];
            nl[(2 * py + 1) * ns + 2 * px] = v[2];
            nl[(2 * py + 1) * ns + 2 * px + 1] = v[3];
        }
        hs_barrier();
    }
}

// (second launch bound = waves per SIMD: two 6-wave blocks per CU -> 3, i.e. <= 168 VGPRs)
// PYR: the instantiation that carries the pyramid (its mere presence in the code costs the plain launch 7 us of 49: two kernels)
template <bool PYR>
__global__ __launch_bounds__((HS_NG + HS_NE) * 64, 3) void head_stream_kernel(const wmd_head_level_args a, const HeadStreamGeom gm, const HeadStreamPyr pyr_) {
    struct NoPyr { int n; };
    const std::conditional_t<PYR, HeadStreamPyr, NoPyr> pyr = [&] { if constexpr (PYR) return pyr_; else return NoPyr{0}; }();
    constexpr int C = HS_C, PW = HS_PW, TW = HS_TW, S = HS_S, L = HS_L, RING = HS_RING, TS = HS_TS, XW = HS_XW;
    constexpr int MR = C / 16, KS = C / 4;
    __shared__ __attribute__((aligned(16))) float ts[54 * TS];
    __shared__ __attribute__((aligned(16))) float xs0[HS_NG * XW];
    __shared__ __attribute__((aligned(16))) float xs1[HS_NG * XW];
    __shared__ __attribute__((aligned(16))) float xs2[HS_NG * XW];
    __shared__ float pyr_low[2][PYR ? 16 * (HS_PYR_TH_MAX / 2) : 1];   // low-pass tiles handed down the pyramid: 8 x TH/4, then 16 x TH/2
    __shared__ float yl_tile[PYR ? 32 * HS_PYR_TH_MAX : 1];            // ... and the last one: this level's low-pass input, 32 x TH

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, lc = lane & 15;
    const int H = a.H, W = a.W;
    const size_t plane = (size_t)H * W;
    const int upb = gm.strips * gm.segs;   // units per frame

    if (wave < HS_NG) {
        // ================================================ GEMM waves =========================================================
        // weight fragments, once per block (blocks are persistent).  First product: A = W1 (row = mid channel 16m + lc, K-lane g =
        // input channel 4k + g): the ordinary packed image.  Second product: B = W3' for the K-step (m, i) whose K-lane g is mid
        // channel 16m + 4g + i -- fragment 4m + g of the ordinary image, lane 16 i + lc.
        float w1f[2][MR][KS], w2f[2][2][KS];
        f32x4 b1v[2][MR];
        auto load_weights = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int sd = 0; sd < 2; ++sd) {
                const float* w1 = a.wp1 + (size_t)sd * MR * KS * 64 + lane;
                const float* w2 = a.wp2 + (size_t)sd * 2 * KS * 64;
#pragma unroll
                for (int m = 0; m < MR; ++m)
#pragma unroll
                    for (int k = 0; k < KS; ++k) w1f[sd][m][k] = w1[(m * KS + k) * 64];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int m = 0; m < MR; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i) w2f[sd][j][m * 4 + i] = w2[((size_t)j * KS + 4 * m + g) * 64 + i * 16 + lc];
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (a.bias1) b4 = *reinterpret_cast<const float4*>(a.bias1 + sd * C + m * 16 + g * 4);
                    b1v[sd][m] = f32x4{b4.x, b4.y, b4.z, b4.w};
                }
            }
        };
        if constexpr (!PYR) load_weights();      // (with a pyramid: behind it, unit by unit -- its gathers need the registers)
        const float slope = a.slope;

        for (int unit = blockIdx.x; unit < gm.nunits; unit += gridDim.x) {   // (PYR: one unit per block -- HS_UNIT_END -- see the launch)
            const int b = unit / upb, rem = unit - b * upb;
            const int seg = rem / gm.strips, strip = rem - seg * gm.strips;
            const int x0 = strip * TW, y0 = seg * gm.TH;
            const int th = min(gm.TH, H - y0);
            const int npu = (th + 2) * PW, nsteps = (npu + S - 1) / S;
            const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a.x + (size_t)b * C * plane), 0, (int)(C * plane * 4), 0x00020000);
            const unsigned chq = (unsigned)(plane * 16);   // byte stride of four channels

            // stage this wave's slice of the NEXT step in sequence: 8 LDS-DMA instructions of [4 channels][16 positions].  The calls
            // come strictly in step order, so the lane's patch coordinates advance incrementally ((py, px) += S positions: one
            // compare-and-carry instead of a division per step), and the padding is the heads' reflect padding folded by min / abs
            // (other modes stay on head_level_kernel): ~18 vector instructions per step where the generic form had ~45 and a dozen
            // scalar branches -- every one of them comes out of the matrix rate (fp32 MFMA and the vector ALU are one resource).
            int ipy = (wave * 16 + lc) / PW, ipx = (wave * 16 + lc) % PW;
            const unsigned gch = (unsigned)g * (unsigned)(plane * 4);
            auto issue = [&](float* buf) {
                const int gy = y0 - 1 + ipy, gx = x0 - 1 + ipx;
                const int ay = gy < 0 ? -gy : gy, ax = gx < 0 ? -gx : gx;
                const int ry = min(ay, 2 * H - 2 - ay), rx = min(ax, 2 * W - 2 - ax);     // reflect: -1 -> 1, n -> n - 2
                const bool ok = ipy < th + 2 && rx >= 0;                                   // beyond the patch / strip overhang -> 0
                const unsigned off = ok ? gch + (unsigned)(ry * W + rx) * 4u : 0x80000000u;
                float* dst = buf + wave * XW;
#pragma unroll
                for (int q = 0; q < KS; ++q) hs_dma4(rx_, (lds_ptr_t)(dst + q * 64), off, (unsigned)q * chq);
                ipx += S % PW;
                ipy += S / PW;
                if (ipx >= PW) ipx -= PW, ipy += 1;
            };

            // ring write addresses of this lane (bytes): tap row lc (j = 0) and 16 + lc (j = 1) at slot 4g of the group; the j = 1
            // lanes of rows 27..31 (zero weights, nobody reads them) go to the 16 spare floats behind a plane's ring instead of
            // being masked out: no branch in the store sequence
            const unsigned wa0 = hs_lds_addr(ts) + (unsigned)((lc * TS + g * 4) * 4);
            const unsigned wa1 = hs_lds_addr(ts) + (unsigned)(lc < 11 ? ((16 + lc) * TS + g * 4) * 4 : (lc * TS + RING + 4 + g * 4) * 4);
            const unsigned wm1 = lc < 11 ? 4u : 0u;
            static_assert(HS_TS - (HS_RING + 4) >= 16, "spare floats behind the ring for the masked lanes");

            auto step = [&](int s, const float* cur, float* nxt2) {
                float xf[KS];
#ifdef HS_VM0
                hs_read_slice<0>(hs_lds_addr(cur + wave * XW + lane), xf);
#else
                if (s + 1 < nsteps) hs_read_slice<KS>(hs_lds_addr(cur + wave * XW + lane), xf);   // the next slice's 8 may stay in flight
                else hs_read_slice<0>(hs_lds_addr(cur + wave * XW + lane), xf);
#endif
                if (s + 2 < nsteps && !HS_DBG_ON(2)) issue(nxt2);
                f32x4 acc[2][MR];
#pragma unroll
                for (int sd = 0; sd < 2; ++sd)
#pragma unroll
                    for (int m = 0; m < MR; ++m) acc[sd][m] = b1v[sd][m];
                if (!HS_DBG_ON(4)) {
#pragma unroll
                for (int k = 0; k < KS; ++k)
#pragma unroll
                    for (int sd = 0; sd < 2; ++sd)
#pragma unroll
                        for (int m = 0; m < MR; ++m)
                            acc[sd][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1f[sd][m][k], xf[k], acc[sd][m], 0, 0, 0);
                }
                f32x4 acc2[2][2];
#pragma unroll
                for (int sd = 0; sd < 2; ++sd)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc2[sd][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (!HS_DBG_ON(4)) {
                    float mid[2][MR][4];
#pragma unroll
                    for (int sd = 0; sd < 2; ++sd)
#pragma unroll
                        for (int m = 0; m < MR; ++m) {
                            if (sd == 0 && m == 0) hs_leaky4<true>(acc[sd][m], slope, mid[sd][m]);
                            else hs_leaky4<false>(acc[sd][m], slope, mid[sd][m]);
                        }
#pragma unroll
                    for (int m = 0; m < MR; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int sd = 0; sd < 2; ++sd)
#pragma unroll
                                for (int j = 0; j < 2; ++j)
                                    acc2[sd][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(mid[sd][m][i], w2f[sd][j][m * 4 + i], acc2[sd][j], 0, 0, 0);
                }
                // D2[position 4g + i][tap row 16j + lc] -> ring plane (side, row), 4 consecutive slots
                const int slot0 = (s * S + wave * 16) % RING;
#ifdef HS_CWRITE
#pragma unroll
                for (int sd = 0; sd < 2; ++sd)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int row = j * 16 + lc;
                        if (row < 27) {
                            float* d = ts + (sd * 27 + row) * TS + slot0 + g * 4;
                            const float4 v = make_float4(acc2[sd][j][0], acc2[sd][j][1], acc2[sd][j][2], acc2[sd][j][3]);
                            *reinterpret_cast<float4*>(d) = v;
                            if (slot0 == 0 && g == 0) *reinterpret_cast<float4*>(d + RING) = v;
                        }
                    }
#else
                const unsigned a0 = wa0 + (unsigned)slot0 * 4u, a1 = wa1 + (unsigned)slot0 * wm1;
                if (!HS_DBG_ON(8)) hs_write_tiles<27 * TS * 4>(a0, a1, acc2[0][0], acc2[0][1], acc2[1][0], acc2[1][1]);
                if (slot0 == 0) {   // (uniform) slots 0..3 are mirrored behind the ring's end: a row base at slot RING-1 reads on to RING+1
                    const unsigned am0 = g == 0 ? a0 + RING * 4 : wa0 + (RING + 4) * 4;      // other lanes: the spare floats again
                    const unsigned am1 = (g == 0 && lc < 11) ? a1 + RING * 4 : wa1 + (lc < 11 ? (RING + 4) * 4 : 0);
                    hs_write_tiles<27 * TS * 4>(am0, am1, acc2[0][0], acc2[0][1], acc2[1][0], acc2[1][1]);
                }
#endif
                hs_barrier();
            };

            issue(xs0);
            if (nsteps > 1) issue(xs1);
            if constexpr (PYR) {   // behind the first two slices' DMA (they land meanwhile); ends in pyr.n block barriers
                hs_pyramid(pyr, b, y0, x0, gm.TH, tid, pyr_low, yl_tile);
                load_weights();
            }
            for (int s = 0; s < nsteps; s += 3) {
                step(s, xs0, xs2);
                if (s + 1 < nsteps) step(s + 1, xs1, xs0);
                if (s + 2 < nsteps) step(s + 2, xs2, xs1);
            }
            hs_barrier();   // the epilogue of the last step has read the ring: the next unit may overwrite it
            if constexpr (PYR) break;
        }
    } else {
        // ============================================== epilogue waves =======================================================
        const int e = wave - HS_NG;
        const int side = lane >> 5, al = lane & 31;
        float b3v[3];
        {
            const float* b3 = side == 0 ? a.bias_p : a.bias_n;
#pragma unroll
            for (int co = 0; co < 3; ++co) b3v[co] = b3 ? b3[co] : 0.f;
        }
        const float* tsl = ts + side * 27 * TS;

        for (int unit = blockIdx.x; unit < gm.nunits; unit += gridDim.x) {   // (PYR: one unit per block -- HS_UNIT_END -- see the launch)
            const int b = unit / upb, rem = unit - b * upb;
            const int seg = rem / gm.strips, strip = rem - seg * gm.strips;
            const int x0 = strip * TW, y0 = seg * gm.TH;
            const int th = min(gm.TH, H - y0);
            const int npu = (th + 2) * PW, nsteps = (npu + S - 1) / S;

            if constexpr (PYR) hs_pyramid(pyr, b, y0, x0, gm.TH, tid, pyr_low, yl_tile);
            hs_barrier();   // step 0 has no finished anchors
            for (int s = 1; s <= nsteps; ++s) {
                // anchors of this interval: [(s-1) S - L, s S - L) -- every tap of theirs was written in steps <= s-1
                const int abase = (s - 1) * S - L + e * 32;
                const int an = abase + al;
                const int ac = max(an, 0);
                const int oy = ac / PW, ox = ac - oy * PW;
                const bool valid = an >= 0 && oy < th && ox < TW && x0 + ox < W;
                if (__builtin_amdgcn_ballot_w64(valid) != 0 && !HS_DBG_ON(1)) {
                    const int y = y0 + oy, x = x0 + ox;
                    const bool writer = valid && side == 0;
                    // (the global low-pass value is requested here, unconditionally as far as the pyramid is concerned, and consumed
                    //  after the gathers: behind a select on pyr.n the compiler issued it late and every step waited for it)
                    const float yl_g = (writer && a.yl) ? a.yl[(size_t)b * plane + (size_t)y * W + x] : 0.f;
                    float yl_v = yl_g;
                    if constexpr (PYR) yl_v = yl_tile[min(oy, HS_PYR_TH_MAX - 1) * TW + min(ox, TW - 1)];
                    const int sb = ((abase % RING) + RING) % RING;          // (uniform)
                    int r0 = sb + al;
                    r0 -= r0 >= RING ? RING : 0;
                    int r1 = r0 + PW;
                    r1 -= r1 >= RING ? RING : 0;
                    int r2 = r1 + PW;
                    r2 -= r2 >= RING ? RING : 0;
                    const float* rb[3] = {tsl + r0, tsl + r1, tsl + r2};
                    float sg[3];
#pragma unroll
                    for (int co = 0; co < 3; ++co) {
                        float h = b3v[co];
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx) h += rb[dy][(co * 9 + dy * 3 + dx) * TS + dx];
                        sg[co] = __builtin_amdgcn_rcpf(1.f + __expf(-h));   // v_exp_f32 / v_rcp_f32 (1 ulp each): ~2e-7 on a sigmoid, 8 instead of ~60 instructions
                    }
                    float yh[3];
#pragma unroll
                    for (int co = 0; co < 3; ++co) {
                        const float other = __shfl_xor(sg[co], 32);   // the - side's sigmoid arrives in the + side's lane
                        yh[co] = a.scale * sg[co] - a.scale * other;
                    }
                    if (writer && !HS_DBG_ON(16)) {
                        const size_t px = (size_t)y * W + x;
#pragma unroll
                        for (int co = 0; co < 3; ++co) a.yh[((size_t)b * 3 + co) * plane + px] = yh[co];
                        if ((a.yl || PYR) && a.out) {
                            const float l = yl_v;
                            float v[4] = {(l + yh[0] + yh[1] + yh[2]) * 0.5f, (l + yh[0] - yh[1] - yh[2]) * 0.5f,
                                          (l - yh[0] + yh[1] - yh[2]) * 0.5f, (l - yh[0] - yh[1] + yh[2]) * 0.5f};
                            const size_t dst = (size_t)b * 4 * plane + (size_t)(2 * y) * (2 * W) + 2 * x;
                            *reinterpret_cast<float2*>(a.out + dst) = make_float2(v[0], v[1]);
                            *reinterpret_cast<float2*>(a.out + dst + 2 * W) = make_float2(v[2], v[3]);
                            if (a.disp) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    v[k] *= a.disp_scale;
                                    if (a.clamp01) v[k] = fminf(fmaxf(v[k], 0.f), 1.f);
                                }
                                *reinterpret_cast<float2*>(a.disp + dst) = make_float2(v[0], v[1]);
                                *reinterpret_cast<float2*>(a.disp + dst + 2 * W) = make_float2(v[2], v[3]);
                            }
                        }
                    }
                }
                if (s < nsteps) hs_barrier();
            }
            hs_barrier();   // unit end (pairs with the GEMM waves')
            if constexpr (PYR) break;
        }
    }
}

// segment height: the host's model of the launch -- a CU that holds n blocks at a time needs f(n) x the steps of one block,
// and two co-resident blocks gain little over running one after the other (measured at 96 x 320 x 12: 480 units of 24 rows
// 62.0 us, 240 units of 48 rows 57.5 us: the matrix pipe and the vector ALU are one resource, a second block's waves mostly
// queue behind the first's) -- so prefer one block per CU as long as that fills the chip; ties -> the taller segment (less halo)
static int head_stream_pick_th(int B, int H, int strips, bool pyramid, double* cost_out = nullptr) {
    static const int forced = [] {
        const char* e = getenv("WMD_HEAD_STREAM_TH");
        return e ? atoi(e) : 0;
    }();
    if (forced > 0) return pyramid ? std::min(((std::min(forced, H) + 7) / 8) * 8, HS_PYR_TH_MAX) : std::min(forced, H);
    double best = 1e300;
    int best_th = std::min(H, 24);
    for (int segs = 1; segs <= (H + 3) / 4; ++segs) {
        int th = (H + segs - 1) / segs;                              // equal segments (the last one may be shorter)
        if (pyramid) th = std::min(((th + 7) / 8) * 8, HS_PYR_TH_MAX);   // the coarser levels' regions nest in multiples of 8 rows
        if ((H + th - 1) / th != segs) continue;
        const long units = (long)B * strips * segs;
        const long per_cu = (units + kNumCU - 1) / kNumCU;
        const double steps = ((th + 2) * HS_PW + HS_S - 1) / HS_S + 4;
        const double f = HS_NG > 4 ? (double)per_cu : (double)(per_cu / 2) * 1.8 + (double)(per_cu % 2);   // pairs of co-resident blocks, then a lone one
        const double c = steps * f;
        if (c < best - 1e-9) best = c, best_th = th;
    }
    if (cost_out) *cost_out = best;
    return best_th;
}

// Does the pyramid pay for a level of this size?  Its segments are at most 64 rows (one pixel of the finest coarse level per GEMM-wave
// lane), so a tall level may need more units than CUs where the plain launch needs one per CU (1024x320, batch 8: 384 units of 56
// rows against 256 of 80: 1.297 vs 1.279 ms for the whole forward) -- then the completion stays a launch of its own.
int head_stream_pyramid_pays(int B, int H, int W) {
    const int strips = (W + HS_TW - 1) / HS_TW;
    double plain = 0, pyr = 0;
    head_stream_pick_th(B, H, strips, false, &plain);
    head_stream_pick_th(B, H, strips, true, &pyr);
    return pyr <= plain * 1.05;
}

// -> 1 when the streaming kernel took the launch (C = 32, no sparse / training outputs, 0 <= slope <= 1; WMD_HEAD_STREAM=0 off)
int head_stream_launch(const wmd_head_level_args* g, const wmd_head_shiftsum_args* coarse, int n_coarse, hipStream_t s) {
    static const bool on = [] {
        const char* e = getenv("WMD_HEAD_STREAM");
        return !(e && atoi(e) == 0);
    }();
    if (!on || g->C != HS_C || g->yh_mask || g->mid_out || g->sig_p || g->sig_n || g->pad_mode != WMD_PAD_REFLECT) return 0;
    if (!(g->slope >= 0.f && g->slope <= 1.f)) return 0;
    // small maps stay on the one-shot tile kernel: a streaming block needs ~5 steps to fill and drain its pipeline (2 x 12 x 40:
    // 17.1 vs 13.5 us, one 96 x 320 frame: 17.2 vs 15.2 us)
    static const long min_pixels = [] {
        const char* e = getenv("WMD_HEAD_STREAM_MIN_PIXELS");
        return e ? atol(e) : 0L;
    }();
    if ((long)g->B * g->H * g->W < min_pixels) return 0;
    HeadStreamGeom gm;
    gm.strips = (g->W + HS_TW - 1) / HS_TW;
    HeadStreamPyr pyr;
    pyr.n = 0;
    if (n_coarse > 0) {      // (shapes validated by wmd_head_level_pyramid_fwd)
        if (n_coarse > 3 || (g->H & 7) || g->H > 8 * 4096) return 0;
        for (int k = 0; k < n_coarse; ++k) pyr.lv[k] = coarse[k];
        pyr.n = n_coarse;
    }
    for (int k = pyr.n; k < 3; ++k) pyr.lv[k] = pyr.lv[0];
    gm.TH = head_stream_pick_th(g->B, g->H, gm.strips, pyr.n > 0);
    gm.segs = (g->H + gm.TH - 1) / gm.TH;
    const long nunits = (long)g->B * gm.strips * gm.segs;
    if (nunits > (1L << 30)) return 0;
    gm.nunits = (int)nunits;
    gm.dbg = 0;
#ifdef HS_DBG
    gm.dbg = getenv("WMD_HS_DBG") ? atoi(getenv("WMD_HS_DBG")) : 0;
#endif
    const double pix = (double)g->B * g->H * g->W;
    ProfScope prof("head_stream_kernel", 2.0 * pix * (2.0 * g->C * g->C + 54.0 * g->C),
                   4.0 * pix * (g->C + 3 + (g->out ? (g->disp ? 9 : 5) : 0)), s);
    // what the matrix pipe executes: every patch position of every unit through 2 x (C x C + 32 x C) MACs
    {
        double pos = 0;
        for (int sg = 0; sg < gm.segs; ++sg) {
            const int th = std::min(gm.TH, g->H - sg * gm.TH);
            pos += (double)(((th + 2) * HS_PW + HS_S - 1) / HS_S) * HS_S;
        }
        prof.mfma(2.0 * pos * g->B * gm.strips * 2.0 * (g->C * g->C + 32.0 * g->C));
    }
    // the pyramid instantiation takes ONE unit per block: its three argument blocks must not stay live across the step loops of a
    // persistent block (132 SGPR spills to VGPR lanes, v_readlane in the hot loops: 7 us of 49)
    const dim3 grid((unsigned)(pyr.n > 0 ? gm.nunits : std::min(gm.nunits, (HS_NG > 4 ? 1 : 2) * kNumCU)));
    if (pyr.n > 0) hipLaunchKernelGGL(head_stream_kernel<true>, grid, dim3((HS_NG + HS_NE) * 64), 0, s, *g, gm, pyr);
    else hipLaunchKernelGGL(head_stream_kernel<false>, grid, dim3((HS_NG + HS_NE) * 64), 0, s, *g, gm, pyr);
    return 1;
}

}  // namespace wmd
