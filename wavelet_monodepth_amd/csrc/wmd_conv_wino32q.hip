// Winograd F(2x2,3x3) trunk convolution on v_mfma_f32_32x32x2_f32, QUARTER-position waves (round 5, gfx950).
//
// Same operator, same algebra, same weight image and the same flattened LDS-DMA staging as conv_wino32_kernel
// (wmd_conv_wino32.hip: ConvBlock / Conv3x3 + nearest upsample + skip concat + pad of the reference, KITTI/layers.py:120-161,
// 233-236; depth_decoder.py:145-150).  What changes is how the work is cut into waves, because of what round 5's cycle stamps
// showed about that kernel (profiles/r05_notes.md): with 128 accumulators a wave needs 200 registers = two waves per SIMD = two
// blocks per CU; the matrix pipe runs at 0.90 while both are in their main loops, at 0.645 when one of them is alone (it cannot
// hide its own LDS-DMA issue) and idles through every prologue / epilogue the two blocks spend together.
//
//   * A wave owns ONE transformed row of the 4 x 4 position grid: 32 tiles x 32 out channels x 4 positions = 64 accumulator
//     registers.  Four such waves (the "quarters") make a tile group = a block of 256 threads on a 128-pixel tile: <= 168
//     registers and 46 KB of LDS, so THREE blocks share a CU and every SIMD holds three waves of three different blocks -- one
//     block's prologue, epilogue and DMA issue run under the other two's MFMAs.
//   * A quarter needs only the two patch rows its transformed row is made of (B^T row r: d0-d2, d1+d2, d2-d1, d1-d3): 4
//     ds_read_b64 + 8 adds + 4 ds_read_b32 feed 4 MFMAs of 64 cycles -- the same ~4 instructions per MFMA as the half-position
//     kernel.  On the upsampled operand (staged at its own resolution, 9 of 16 positions) rows 0, 1 and 3 carry three
//     positions each and row 2 none: that quarter only stages and waits on those chunks.
//   * Y = A^T M A row by row: quarter r reduces its row to P_r[b] = sum_c A^T[b][c] M[r][c] (2 values per tile), all four trade
//     them through LDS, and Y[0] = P0 + P1 + P2, Y[1] = P1 - P2 - P3 are finished by quarters (0, 1) and (2, 3), eight tiles of a
//     lane each; the output leaves through whole 128-byte lines like conv_wino32_kernel's (LDS transpose inside the wave).
// Contract: PURE channel chunking is required (every CK-chunk inside one source tensor; no generic gather).  Masks and work
// lists ARE supported -- the MASKED and LIST instantiations below serve the block-sparse levels (input mask in the dword gather
// offsets, tile-activity test, output select; 8x16 list tile with the device-chosen K split) and plan_conv offers TAPS == 18 to
// in_mask / out_mask / out_tiles launches; the only masked case declined is an upsampled input mask without the 2x2 promise
// (wmd_conv_args.in_mask_2x2 == 0), which returns WMD_ERR_UNSUPPORTED when forced.
#include <algorithm>
#include <type_traits>
#include <utility>
#include "wmd_conv_common.h"

namespace wmd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <class F, int... I>
__device__ __forceinline__ void q_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void q_static_for(F&& f) {
    q_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// positions of transformed row R an operand reaches: all four columns, or columns 0, 1, 3 of rows 0, 1, 3 (upsampled operand)
__host__ __device__ constexpr int q_count(int R, bool up) { return up ? (R == 2 ? 0 : 3) : 4; }
__host__ __device__ constexpr int q_col(bool up, int p) { return up ? (p == 2 ? 3 : p) : p; }


// MASKED: block-sparse execution (wmd_conv_args.in_mask / out_mask: the sparse decoders' levels) -- input positions outside in_mask
//   gather the out-of-range offset (read 0), a tile without an out_mask pixel walks an empty chunk range and stores nothing, stored
//   pixels outside out_mask are 0; dword staging only (the mask is per position).  LIST (with MASKED): the block's tile comes out of
//   the per-frame work lists and the K split is chosen on the device (wmd_conv_args.out_tiles), as in conv_wino32_kernel.
template <int TH, int TW, int CK, bool MASKED = false, bool LIST = false>
__global__ __launch_bounds__(256, 3) void conv_wino32q_kernel(const ConvKArgs a) {
    using T = W32QTile<TH, TW, CK>;
    constexpr int NT = T::NT, PWS = T::PWS, PSF = T::PSF, PWL = T::PWL, PSL = T::PSL, KW = T::KW, TXB = T::TXB;
    static_assert(!LIST || MASKED, "the work-list form is a masked instantiation");
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS + T::TAB_FLOATS + 8];   // + one tile-activity flag per wave

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int qr = __builtin_amdgcn_readfirstlane(tid >> 6);   // this wave's transformed row

    int t, by;
    int tick;   // this block's (pixel tile, slab) item: index of its split-K ticket counter
    int ks_n = a.ksplit, cps = a.chunks_per_split;   // LIST: chosen below from the list's length
    if constexpr (LIST) {
        const int n_items = list_total(a.tile_count, a.B) * a.cob;
        if ((int)blockIdx.x >= n_items) return;
        list_ksplit(n_items, a.nchunks, a.ksmax, a.list_slots, ks_n, cps);
        if ((int)blockIdx.z >= ks_n) return;
        const int item = xcd_contiguous(blockIdx.x, n_items);
        const int ti = item / a.cob;
        by = item - ti * a.cob;
        t = list_entry(a.tile_list, a.tile_count, a.B, a.tiles_x * a.tiles_y, ti);
        tick = item;
    } else if (a.cob > 0) {   // (pixel tile, out-channel slab) items, slab fastest, one contiguous run per XCD
        const int item = xcd_contiguous(blockIdx.x, gridDim.x);
        t = item / a.cob;
        by = item - t * a.cob;
        tick = item;
    } else if (a.xcd_slab) {
        // 2-D grid, one out-channel slab per XCD (round 6): workgroup L of a z-slice runs on XCD L % 8 (tools/probes/xcc_probe.hip), so
        // slab L % nslab with nslab a multiple of 8 keeps a slab's 1/nslab of the weight image in ONE L2 instead of streaming the whole
        // image into all eight (counters: L0 / L1 of config 2 read inputs + 8 x 8.4 MB of weights per launch, profiles/r06_notes.md)
        const int L = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y;
        by = L % (int)gridDim.y;
        t = L / (int)gridDim.y;
        tick = t * (int)gridDim.y + by;
    } else {
        t = xcd_contiguous(blockIdx.x, gridDim.x);
        by = blockIdx.y;
        tick = t * (int)gridDim.y + by;
    }
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int b = t / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int ks = blockIdx.z;
    const int H = a.H, W = a.W;
    const bool upl = a.up1 == 2;   // structured low-resolution path of the upsampled operand
    // block-sparse: a tile without active output pixels keeps its zeros (empty chunk range + no stores; see conv_wino32_kernel for
    // why not an early return)
    bool skip = false;
    if (MASKED && !LIST && a.out_mask) {
        int any = 0;
        for (int i = tid; i < TH * TW; i += NT) {
            const int yy = y0 + i / TW, xx = x0 + i % TW;
            if (yy < H && xx < W) any |= a.out_mask[(size_t)b * H * W + (size_t)yy * W + xx];
        }
        int* flags = reinterpret_cast<int*>(lds + T::LDS_FLOATS + T::TAB_FLOATS);
        const bool wave_any = __builtin_amdgcn_ballot_w64(any != 0) != 0;
        if (lane == 0) flags[qr] = wave_any ? 1 : 0;
        __syncthreads();
        skip = __builtin_amdgcn_readfirstlane(flags[0] | flags[1] | flags[2] | flags[3]) == 0;
    }

    // ---- staging geometry (as conv_wino32_kernel's flattened staging) ------------------------------------------------
    constexpr unsigned kOOB = 0x80000000u;
    const size_t plane1 = (size_t)a.H1 * a.W1, plane2 = (size_t)H * W;
    const unsigned pb1 = (unsigned)(plane1 * 4), pb2 = (unsigned)(plane2 * 4);
    auto fold = [&](int g, int n, int& ok) {
        const int refl = g < 0 ? -g : (g >= n ? 2 * n - 2 - g : g);
        const int clam = min(max(g, 0), n - 1);
        ok &= (int)(a.pad_mode != WMD_PAD_ZERO) | (int)(g == clam);
        const int r = a.pad_mode == WMD_PAD_REFLECT ? refl : clam;
        return min(max(r, 0), n - 1);
    };
    constexpr int NPF = (CK * PSF + NT - 1) / NT, NPL = (CK * PSL + NT - 1) / NT;
    constexpr bool X4 = T::X4OK && !MASKED;
    constexpr int NPF4 = (CK * PSF / 4 + NT - 1) / NT, NPL4 = (CK * PSL / 4 + NT - 1) / NT;
    bool x4 = false;
    unsigned obF[NPF], obL[NPL];
    int* tab = reinterpret_cast<int*>(lds + T::LDS_FLOATS);
    constexpr int PH = T::PH, PHL = T::PHL;
    const int Hs_g = upl ? H : a.H1, Ws_g = upl ? W : a.W1, sh_g = upl ? 0 : a.shift1;
    {
        int tt = tid;
        if (tt < PH + PWS) {
            const bool row = tt < PH;
            const int g0 = row ? y0 + tt - 1 : x0 + (tt - PH) - 1, n = row ? H : W, ns = row ? Hs_g : Ws_g;
            int ok = (int)(g0 <= n);
            const int g = fold(g0, n, ok) - sh_g;
            ok &= (int)(g >= 0) & (int)(g < ns);
            tab[tt] = ok ? (row ? g * Ws_g * 4 : g * 4) : -1;
        } else if (tt < PH + PWS + PHL + PWL) {
            tt -= PH + PWS;
            const bool row = tt < PHL;
            const int s0 = row ? (y0 >> 1) - 1 + tt : (x0 >> 1) - 1 + (tt - PHL), n = row ? a.H1 : a.W1;
            const int sc = min(max(s0, 0), n - 1);
            const int ok = (int)(s0 <= n) & ((int)(a.pad_mode != WMD_PAD_ZERO) | (int)(sc == s0));
            tab[PH + PWS + tt] = ok ? (row ? sc * a.W1 * 4 : sc * 4) : -1;
        }
    }
    __syncthreads();
    const unsigned pbs = upl ? pb2 : pb1;
    if constexpr (X4) x4 = x0 - 1 - sh_g >= 0 && x0 + TW - sh_g < Ws_g && !a.no_x4;
    if (X4 && x4) {
        constexpr int GF = T::GF, GL = T::GL;
#pragma unroll
        for (int i = 0; i < NPF4; ++i) {
            const unsigned e = tid + i * NT, ch = e / (PH * GF), rem = e - __umul24(ch, PH * GF);
            const unsigned py = rem / GF, j = rem - __umul24(py, GF);
            const int r = tab[min(py, (unsigned)PH - 1)];
            const bool ok = ch < CK && r >= 0;
            obF[i] = ok ? ch * pbs + (unsigned)r + (unsigned)(x0 - 1 - sh_g + 4 * (int)j) * 4u : kOOB;
        }
#pragma unroll
        for (int i = 0; i < NPL4; ++i) {
            const unsigned e = tid + i * NT, ch = e / (PHL * GL), rem = e - __umul24(ch, PHL * GL);
            const unsigned py = rem / GL, j = rem - __umul24(py, GL);
            const int r = tab[PH + PWS + min(py, (unsigned)PHL - 1)];
            const bool ok = ch < CK && r >= 0;
            obL[i] = ok ? ch * pb1 + (unsigned)r + (unsigned)((x0 >> 1) - 1 + 4 * (int)j) * 4u : kOOB;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const unsigned e = tid + i * NT, ch = e / PSF, pos = e - __umul24(ch, PSF);
            const unsigned py = pos / PWS, px = pos - __umul24(py, PWS);
            const int r = tab[py], c = tab[PH + px];
            bool ok = ch < CK && (r | c) >= 0;
            // the full-resolution geometry is the mask's own (pure layers): the folded pixel offset indexes it directly
            if (MASKED && a.in_mask && ok) ok = a.in_mask[(size_t)b * plane2 + ((unsigned)(r + c) >> 2)] != 0;
            obF[i] = ok ? ch * pbs + (unsigned)(r + c) : kOOB;
        }
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const unsigned e = tid + i * NT, ch = e / PSL, pos = e - __umul24(ch, PSL);
            const unsigned py = pos / PWL, px = pos - __umul24(py, PWL);
            const int r = tab[PH + PWS + py], c = tab[PH + PWS + PHL + px];
            bool ok = ch < CK && (r | c) >= 0;
            if (MASKED && a.in_mask && upl && ok) {   // 2x2-constant mask: source pixel (sy, sx) is masked like (2 sy, 2 sx)
                const unsigned sidx = (unsigned)(r + c) >> 2, sy = sidx / (unsigned)a.W1, sx = sidx - sy * (unsigned)a.W1;
                ok = a.in_mask[(size_t)b * plane2 + (size_t)(2 * sy) * W + 2 * sx] != 0;
            }
            obL[i] = ok ? ch * pb1 + (unsigned)(r + c) : kOOB;
        }
    }
    const float* x1b = a.x1 + (size_t)b * a.C1 * plane1;
    const float* x2b = a.x2 ? a.x2 + (size_t)b * a.C2 * plane2 : a.x1;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x1b), 0, (int)(a.C1 * plane1 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x2b), 0, (int)(a.C2 * plane2 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.wp), 0, (int)((size_t)a.ncot * a.nci4 * 16 * 64 * 4), 0x00020000);
    unsigned aoff[T::NAV];
#pragma unroll
    for (int v = 0; v < T::NAV; ++v) {
        const int e = tid + v * NT;
        const int run = e / (CK * 64), rem = e % (CK * 64);
        const int cot = min(by * 2 + run, a.ncot - 1);
        aoff[v] = e < 2 * CK * 64 ? (unsigned)(((size_t)cot * a.nci4 * 16 * 64 + (size_t)rem * 4) * 4) : kOOB;
    }
#ifdef WMD_STAMPS
    const int c_begin_dbg = blockIdx.z * a.chunks_per_split;
#endif
    auto is_up = [&](int chunk) { return upl && (chunk + 1) * CK <= a.C1; };
    struct ChunkSrc {
        __amdgpu_buffer_rsrc_t r;
        unsigned base;
    };
    auto chunk_src = [&](int chunk) {
        ChunkSrc cs;
        const int ci0 = chunk * CK;
        const bool in1 = ci0 < a.C1;
        cs.r = in1 ? r1 : r2;
        cs.base = in1 ? (unsigned)ci0 * pb1 : (unsigned)(ci0 - a.C1) * pb2;
        return cs;
    };
    auto stage_weight_piece = [&](int chunk, float* bufp, int v) {
#ifdef WMD_STAMPS
        if ((a.dbg_mode & 4) && chunk > c_begin_dbg + 1) return;   // timing experiment: no weight traffic after the first two chunks (results wrong)
#endif
        const unsigned soffA = (unsigned)chunk * (unsigned)(T::RUN * 4);
        const int e0 = qr * 64 + v * NT;
        if (T::NAV * NT == 2 * CK * 64 || e0 < 2 * CK * 64)
            lds_dma16(rw, (lds_ptr_t)(bufp + T::B_FLOATS + (e0 / (CK * 64)) * T::RUN_LDS + (e0 % (CK * 64)) * 4), aoff[v], soffA);
    };
    auto stage_full_piece = [&](int chunk, float* bufp, int q, const ChunkSrc& cs) {
#ifdef WMD_STAMPS
        if ((a.dbg_mode & 8) && q < NPF && chunk > c_begin_dbg + 1) return;   // timing experiment: no patch traffic either
#endif
        if (q < NPF) {
            if (X4 && x4) {
                if (q < NPF4 && ((q + 1) * NT * 4 <= CK * PSF || (qr * 64 + q * NT) * 4 < CK * PSF))
                    lds_dma16(cs.r, (lds_ptr_t)(bufp + (qr * 64 + q * NT) * 4), obF[q], cs.base);
            } else if ((q + 1) * NT <= CK * PSF || qr * 64 + q * NT < CK * PSF)
                lds_dma4(cs.r, (lds_ptr_t)(bufp + qr * 64 + q * NT), obF[q], cs.base);
        } else {
            stage_weight_piece(chunk, bufp, q - NPF);
        }
    };
    auto stage_up_piece = [&](int chunk, float* bufp, int q) {
#ifdef WMD_STAMPS
        if ((a.dbg_mode & 8) && q < NPL && chunk > c_begin_dbg + 1) return;
#endif
        if (q < NPL) {
            if (X4 && x4) {
                if (q < NPL4 && ((q + 1) * NT * 4 <= CK * PSL || (qr * 64 + q * NT) * 4 < CK * PSL))
                    lds_dma16(r1, (lds_ptr_t)(bufp + (qr * 64 + q * NT) * 4), obL[q], (unsigned)(chunk * CK) * pb1);
            } else if ((q + 1) * NT <= CK * PSL || qr * 64 + q * NT < CK * PSL)
                lds_dma4(r1, (lds_ptr_t)(bufp + qr * 64 + q * NT), obL[q], (unsigned)(chunk * CK) * pb1);
        } else {
            stage_weight_piece(chunk, bufp, q - NPL);
        }
    };

    const int c_begin = ks * cps;
    const int c_end = skip ? c_begin : min(c_begin + cps, a.nchunks);
    if (c_begin < c_end) {
        if (is_up(c_begin)) {
            q_static_for<NPL + T::NAV>([&](auto qc) { stage_up_piece(c_begin, lds, decltype(qc)::value); });
        } else {
            const ChunkSrc cs0 = chunk_src(c_begin);
            q_static_for<NPF + T::NAV>([&](auto qc) { stage_full_piece(c_begin, lds, decltype(qc)::value, cs0); });
        }
    }
    const int co = by * 32 + (lane & 31);
    const float bias_v = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;

    // ---- operand addressing -------------------------------------------------------------------------------------
    const int tslot = min(lane & 31, T::NTILES - 1);
    const int tyy = tslot / TXB, txx = tslot % TXB;
    const int pbF = tyy * 2 * PWS + txx * 2 + (lane >> 5) * PSF;
    const int pbL = tyy * PWL + txx + (lane >> 5) * PSL;
    const int wbase = T::B_FLOATS + ((lane & 31) >> 4) * T::RUN_LDS + (lane & 15) + 16 * (lane >> 5);
    __syncthreads();

    // Everything from here on is compiled once per quarter (the owned row is a compile-time property).
    auto run = [&](auto r_tag) {
        constexpr int R = decltype(r_tag)::value;
        // patch rows the transformed row R is made of, and the signs: tr = s0 * d[ra] + s1 * d[rb]
        constexpr int RA = R == 0 ? 0 : 1, RB = R == 3 ? 3 : 2;          // rows (0,2) (1,2) (1,2) (1,3)
        // low-resolution source rows of the upsampled operand: row 0 = s0 - s1, row 1 = s1 + s1, row 3 = s1 - s2
        constexpr int LA = R == 0 ? 0 : 1, LB = R == 3 ? 2 : 1;
        f32x16 acc[4];
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[o][r] = 0.f;

        auto chunk_body = [&](int c, auto up_tag, auto next_tag) {
            constexpr bool UP = decltype(up_tag)::value;
            constexpr int NEXT = decltype(next_tag)::value;
            constexpr int NP = q_count(R, UP);
            constexpr int S = KW * NP, D = 3, RS = D + 1;
            constexpr int SP = S > 0 ? ((S * 2) / 3 > 0 ? (S * 2) / 3 : 1) : 1;
            constexpr int NPIECES = NEXT == 0 ? 0 : (NEXT == 2 ? NPL : NPF) + T::NAV;
            const int buf = (c - c_begin) & 1;
            const float* bufp = lds + buf * T::BUF_FLOATS;
            float* nbufp = lds + (buf ^ 1) * T::BUF_FLOATS;
            const float* psrc = bufp + (UP ? pbL : pbF);
            const float* wsrc = bufp + wbase;
            ChunkSrc csn;
            if constexpr (NEXT == 1) csn = chunk_src(c + 1);
            if constexpr (S == 0) {   // the row the upsampled operand never reaches: this wave only stages the next chunk
                q_static_for<NPIECES>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    if constexpr (NEXT == 2) stage_up_piece(c + 1, nbufp, q);
                    else if constexpr (NEXT == 1) stage_full_piece(c + 1, nbufp, q, csn);
                });
            } else {
                // packed fp32 like conv_wino32_kernel (wmd_conv_common.h, w32_pk_*): pairs along patch columns
                f32x2 pa[2], pb[2], vp[2][2];
                float ca = 0.f, cb = 0.f, wf[RS];   // UP: the third source column
                auto fetch_patch = [&](int kk) {
#ifdef WMD_STAMPS
                    if ((a.dbg_mode & 64) && kk > 0) return;   // timing experiment: no patch reads after the first K-step
#endif
                    // volatile LDS-space reads: one ds_read with an immediate offset each, no v_add_u32 for a ds_read2 base
                    // (conv_wino32_kernel has the measurements)
                    typedef const volatile __attribute__((address_space(3))) float* lds_cv1_t;
                    typedef const volatile __attribute__((address_space(3))) f32x2* lds_cv2_t;
                    const lds_cv1_t psrc3 = (lds_cv1_t)psrc;
                    if constexpr (UP) {
                        const lds_cv1_t ra = psrc3 + kk * 2 * PSL + LA * PWL;
                        const lds_cv1_t rb = psrc3 + kk * 2 * PSL + LB * PWL;
                        pa[0] = f32x2{ra[0], ra[1]};
                        ca = ra[2];
                        if constexpr (R != 1) {
                            pb[0] = f32x2{rb[0], rb[1]};
                            cb = rb[2];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            pa[e] = *reinterpret_cast<lds_cv2_t>(psrc3 + kk * 2 * PSF + RA * PWS + e * 2);
                            pb[e] = *reinterpret_cast<lds_cv2_t>(psrc3 + kk * 2 * PSF + RB * PWS + e * 2);
                        }
                    }
                };
                auto transform = [&](int kk) {   // tr = row R of B^T d, then V = tr B on the columns the operand reaches
                    f32x2* vv = vp[kk & 1];      // column c is vv[c / 2][c % 2]
#ifdef WMD_STAMPS
                    if (a.dbg_mode & 16) {   // timing experiment: no transform arithmetic (results wrong)
                        vv[0] = pa[0], vv[1] = pb[0];
                        return;
                    }
#endif
                    if constexpr (UP) {
                        const f32x2 t01 = R == 1 ? w32_pk_add(pa[0], pa[0]) : w32_pk_sub(pa[0], pb[0]);
                        const float t2 = R == 1 ? ca + ca : ca - cb;
                        vv[0] = w32_pk_up(t01);
                        vv[1][1] = t01[1] - t2;
                    } else {
                        f32x2 t[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e)
                            t[e] = R == 1 ? w32_pk_add(pa[e], pb[e]) : (R == 2 ? w32_pk_sub(pb[e], pa[e]) : w32_pk_sub(pa[e], pb[e]));
                        vv[0] = w32_pk_lo(t[0], t[1]);
                        vv[1] = w32_pk_hi(t[0], t[1]);
                    }
                };
                auto fetch_u = [&](int s2) {
#ifdef WMD_STAMPS
                    if ((a.dbg_mode & 32) && s2 >= RS) return;   // timing experiment: no weight-fragment reads after the first ring fill
#endif
                    const int kk = s2 / NP, xi = R * 4 + q_col(UP, s2 % NP);
                    wf[s2 % RS] = wsrc[((kk >> 1) * 16 + xi) * 64 + (kk & 1) * 32];
                };
                fetch_patch(0);
#pragma unroll
                for (int s2 = 0; s2 < D && s2 < S; ++s2) fetch_u(s2);
                transform(0);
                q_static_for<S>([&](auto s2c) {
                    constexpr int s2 = decltype(s2c)::value;
                    constexpr int kk = s2 / NP, p = s2 % NP;
                    constexpr int col = q_col(UP, p);
                    if constexpr (s2 + D < S) fetch_u(s2 + D);
                    if constexpr (kk + 1 < KW) {
                        if constexpr (p == 0) fetch_patch(kk + 1);
                        if constexpr (p == NP - 1) transform(kk + 1);
                    }
                    if constexpr (NEXT != 0) {
                        constexpr int q0 = (s2 * NPIECES + SP - 1) / SP, q1 = ((s2 + 1) * NPIECES + SP - 1) / SP;
                        constexpr int qb = q0 < NPIECES ? q0 : NPIECES, qe = q1 < NPIECES ? q1 : NPIECES;
                        q_static_for<qe - qb>([&](auto qc) {
                            constexpr int q = qb + decltype(qc)::value;
                            if constexpr (NEXT == 2) stage_up_piece(c + 1, nbufp, q);
                            else stage_full_piece(c + 1, nbufp, q, csn);
                        });
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc[col] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[kk & 1][col >> 1][col & 1], wf[s2 % RS], acc[col], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            __syncthreads();   // next buffer landed (vmcnt(0) precedes the barrier), this one is released
        };
        const int up_end = min(c_end, max(c_begin, upl ? a.C1 / CK : 0));
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        int c = c_begin;
        for (; c + 1 < up_end; ++c) chunk_body(c, std::true_type{}, I2{});
        if (c < up_end) {
            if (up_end < c_end) chunk_body(c, std::true_type{}, I1{});
            else chunk_body(c, std::true_type{}, I0{});
            ++c;
        }
        for (; c + 1 < c_end; ++c) chunk_body(c, std::false_type{}, I1{});
        if (c < c_end) chunk_body(c, std::false_type{}, I0{});

        // ---- epilogue: P_R[g][b] = sum_c A^T[b][c] M[R][c] (A^T = [[1,1,1,0],[0,1,-1,-1]]), traded through LDS -----------------
        // accumulator register g = 4q + r of lane l: tile slot 8q + 4(l >> 5) + r, out channel l & 31
        // (on register pairs (g, g + 1), neighbouring accumulator registers: packed sums, 8-byte LDS traffic -- see conv_wino32_kernel)
        float* xch = lds + lane * 2;   // [quarter][g / 2][b][lane][2]: 4 x 32 x 64 floats = 32 KB of the (dead) staging buffers
#pragma unroll
        for (int gp = 0; gp < 8; ++gp) {
            auto M = [&](int c) { return f32x2{acc[c][2 * gp], acc[c][2 * gp + 1]}; };
            *reinterpret_cast<f32x2*>(xch + ((R * 8 + gp) * 2 + 0) * 128) = (M(0) + M(1)) + M(2);
            *reinterpret_cast<f32x2*>(xch + ((R * 8 + gp) * 2 + 1) * 128) = (M(1) - M(2)) - M(3);
        }
        __syncthreads();
        // quarter R finishes output row A = R >> 1 of the tiles g in [8 (R & 1), +8): Y[0] = P0 + P1 + P2, Y[1] = P1 - P2 - P3
        constexpr int A = R >> 1, G0 = 8 * (R & 1);
        f32x2 y2[4][2];   // [(g - G0) / 2][b]
#pragma unroll
        for (int gi = 0; gi < 4; ++gi)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                auto P = [&](int rr) { return *reinterpret_cast<const f32x2*>(xch + ((rr * 8 + G0 / 2 + gi) * 2 + bb) * 128); };
                y2[gi][bb] = A == 0 ? (P(0) + P(1)) + P(2) : (P(1) - P(2)) - P(3);
            }
        __syncthreads();   // every quarter has read what it needs: the exchange area is free for the output transposes

        const bool final_out = (ks_n == 1);
        float* ybase = (LIST && final_out) ? a.y_final + (size_t)b * a.Cout * plane2 : a.y + ((size_t)ks * a.B + b) * a.Cout * plane2;
        const bool vec_ok = (W & 3) == 0;
        const bool wt = a.tickets != nullptr && !final_out;
        const __amdgpu_buffer_rsrc_t ry = agent_rsrc(ybase, (size_t)a.Cout * plane2 * 4);   // this slice's frame (write-through stores)
        // lane (co = l & 31, h = l >> 5) holds, for q2 = 0, 1 and r = 0..3: tile slot 8 (2 (R & 1) + q2) + 4 h + r, output row A,
        // 2 pixels.  16 slots x 2 pixels x 32 channels = 4 KB per wave: [co][8 pieces of two tiles] through the wave's own
        // quarter of the exchange area, then [8 channels][128 B] per store instruction (see conv_wino32_kernel::store_lines).
        auto store_lines = [&](auto act_tag) {
            constexpr int ACT = decltype(act_tag)::value;
            float* tb = lds + R * (32 * 32);
            const int col = lane & 31, hh = lane >> 5;
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    f32x2 o0 = y2[2 * q2 + pr][0], o1 = y2[2 * q2 + pr][1];   // pixels 0 / 1 of the slot pair
                    if constexpr (ACT >= 0) {
                        o0 = w32_act2<(ACT < 0 ? 0 : ACT)>(o0 + f32x2{bias_v, bias_v}, a.slope);
                        o1 = w32_act2<(ACT < 0 ? 0 : ACT)>(o1 + f32x2{bias_v, bias_v}, a.slope);
                    }
                    const int pc = 4 * q2 + 2 * hh + pr;     // piece 0..7: slots 2 pc, 2 pc + 1 of this quarter's sixteen
                    *reinterpret_cast<float4*>(tb + col * 32 + ((pc ^ (col & 7)) * 4)) = make_float4(o0[0], o1[0], o0[1], o1[1]);
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cc = 8 * i + (lane >> 3), pc = lane & 7;
                const float4 v = *reinterpret_cast<const float4*>(tb + cc * 32 + ((pc ^ (cc & 7)) * 4));
                const int ts = 16 * (R & 1) + 2 * pc;
                const int oy = y0 + (ts / TXB) * 2 + A, ox = x0 + (ts % TXB) * 2;
                const int cg = by * 32 + cc;
                if (cg < a.Cout && ts < T::NTILES && oy < H && ox < W && !skip) {
                    float* dst = ybase + (size_t)cg * plane2 + (size_t)oy * W + ox;
                    float o[4] = {v.x, v.y, v.z, v.w};
                    if (MASKED && a.out_mask) {   // branch-free: clamped byte loads + selects (elements past W are never stored)
                        const uint8_t* mp = a.out_mask + (size_t)b * plane2 + (size_t)oy * W;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = mp[min(ox + e, W - 1)] ? o[e] : 0.f;
                    }
                    if (wt) {   // split-K partial of a launch that finishes in-kernel: write-through (splitk_ticket_finish)
                        const unsigned ob = (unsigned)(((size_t)cg * plane2 + (size_t)oy * W + ox) * 4);
                        if (vec_ok) st16_agent(ry, ob, make_float4(o[0], o[1], o[2], o[3]));
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (ox + e < W) st4_agent(ry, ob + 4 * e, o[e]);
                        }
                    } else if (vec_ok) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (ox + e < W) dst[e] = o[e];
                    }
                }
            }
        };
        const int act_sel = final_out ? a.act : -1;
        if (act_sel < 0) store_lines(std::integral_constant<int, -1>{});
        else if (act_sel == WMD_ACT_ELU) store_lines(std::integral_constant<int, WMD_ACT_ELU>{});
        else if (act_sel == WMD_ACT_LEAKY) store_lines(std::integral_constant<int, WMD_ACT_LEAKY>{});
        else if (act_sel == WMD_ACT_SIGMOID) store_lines(std::integral_constant<int, WMD_ACT_SIGMOID>{});
        else store_lines(std::integral_constant<int, WMD_ACT_NONE>{});
        if (wt)   // (uniform) the last K-slice block of this (tile, slab) to arrive sums the slices and writes the final tile
            splitk_ticket_finish<256, TH, TW, 32, MASKED>(a, reinterpret_cast<int*>(lds + T::LDS_FLOATS + T::TAB_FLOATS) + 4, tick, ks_n, b, y0, x0, by * 32);
    };
    if (qr == 0) run(std::integral_constant<int, 0>{});
    else if (qr == 1) run(std::integral_constant<int, 1>{});
    else if (qr == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 3>{});
}

template <int TH, int TW, int CK>
void launch_wino32q(const ConvKArgs& a, dim3 grid, hipStream_t s) {
    if constexpr (wino32q_has_list(TH, TW, CK)) {
        if (a.tile_list) {
            hipLaunchKernelGGL((conv_wino32q_kernel<TH, TW, CK, true, true>), grid, dim3(256), 0, s, a);
            return;
        }
    }
    if (a.in_mask || a.out_mask) hipLaunchKernelGGL((conv_wino32q_kernel<TH, TW, CK, true, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_wino32q_kernel<TH, TW, CK>), grid, dim3(256), 0, s, a);
}

#define WMD_W32Q_INST(TH, TW, CK) template void launch_wino32q<TH, TW, CK>(const ConvKArgs&, dim3, hipStream_t);
#include "wmd_conv_wino32q_table.inc"
#undef WMD_W32Q_INST

}  // namespace wmd
