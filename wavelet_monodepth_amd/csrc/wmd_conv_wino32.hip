// Winograd F(2x2,3x3) trunk convolution on v_mfma_f32_32x32x2_f32 (gfx950).
//
// Same operator as conv_wino_kernel (wmd_conv_fwd.hip): ConvBlock / Conv3x3 + nearest upsample + skip concat + pad of the
// reference (KITTI/layers.py:120-161,233-236; depth_decoder.py:145-150), Y = A^T [(G g G^T) (.) (B^T d B)] A, the same
// fragment-ordered Winograd weight image (wmd_conv_pack_weights_wino) and the same LDS-DMA gather of the virtual padded /
// upsampled / concatenated input.  What changes is the shape of the work around the matrix pipe -- round 2's counters showed
// the 16x16x4 kernel issue-bound at 11.7 instructions per 32-cycle MFMA (profiles/r02_wino_counters.md):
//
//   * MFMA 32x32x2: a wave owns 32 tiles (MFMA rows) x 32 out channels (columns) x 8 of the 16 transformed positions = 8 x 16
//     accumulator registers; the two waves of a tile group ("halves") share a SIMD's matrix pipe.  A lane holds ONE tile and
//     ONE of the K-step's two channels: 6 ds_read_b64 of its patch + 18 adds (its half of B^T d B) + 8 ds_read_b32 (weight
//     fragments) feed 8 MFMAs of 64 cycles each -- 4 instructions per 64-cycle MFMA instead of 11.7 per 32.
//   * The 2x-nearest-upsampled operand (upconv(i,1): decoder channels first, depth_decoder.py:146) is staged at its OWN
//     resolution.  A 4x4 patch of an upsampled map has rows 1,2 and columns 1,2 pairwise equal (the same source pixel), so
//     B^T d B vanishes identically on transformed row 2 and column 2: 9 of the 16 positions remain, computed from the 3x3
//     low-resolution neighbourhood -- 7/16 of the MFMAs of those channels, 3/4 of their LDS-DMA traffic and the >>1 gather
//     disappear.  Exact: d1 + d2 = 2 s1 and d2 - d1 = 0 hold bitwise.  The halves own 5 and 4 of the 9 (8 and 8 of the 16).
//   * Y = A^T M A is linear in M, so each half transforms the positions it owns into a partial 2x2 output; the halves trade
//     one output row through LDS and each finishes (bias, activation, 16-byte stores) the row it keeps.
// Weight fragments come from the ordinary 16x16x4 image: the B operand of K-step k2 (channels 2 k2, 2 k2 + 1) for column
// j = l & 31 is element (co16 = j & 15, ci4 = 2 (k2 & 1) + (l >> 5)) of fragment [j >> 4][k2 >> 1][xi] -- a per-lane base plus
// immediates; the two 16-channel runs are staged 16 floats apart (mod 32) so a 32-lane read group covers all 32 banks.
#include <algorithm>
#include <type_traits>
#include <utility>
#include "wmd_conv_common.h"

namespace wmd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// compile-time loop: f(integral_constant<int, I>) for I = 0 .. N-1, expanded by the front end (a `#pragma unroll` loop over
// the MFMA groups with the piece test inside exceeds the optimizer's full-unroll budget, and a rolled loop would index the
// accumulators at run time)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Which half owns transformed position xi = 4 r + c.  Half 0: row 0 and the left 2x2 of rows 1-2; half 1: the rest.  Of the
// nine positions an upsampled operand reaches (rows / columns 0, 1, 3) half 0 holds five, half 1 four.
__host__ __device__ constexpr bool w32_owns(int hf, int xi) {
    const int r = xi / 4, c = xi % 4;
    const bool h0 = r == 0 || ((r == 1 || r == 2) && c < 2);
    return hf == 0 ? h0 : !h0;
}
__host__ __device__ constexpr bool w32_up_reaches(int xi) { return xi / 4 != 2 && xi % 4 != 2; }
// the p-th owned position (ascending); UP: only the positions an upsampled operand reaches
__host__ __device__ constexpr int w32_nth(int hf, bool up, int p) {
    int n = 0;
    for (int xi = 0; xi < 16; ++xi)
        if (w32_owns(hf, xi) && (!up || w32_up_reaches(xi))) {
            if (n == p) return xi;
            ++n;
        }
    return -1;
}
__host__ __device__ constexpr int w32_count(int hf, bool up) {
    int n = 0;
    for (int xi = 0; xi < 16; ++xi) n += (w32_owns(hf, xi) && (!up || w32_up_reaches(xi))) ? 1 : 0;
    return n;
}
__host__ __device__ constexpr int w32_slot(int hf, int xi) {   // accumulator index of an owned position
    int n = 0;
    for (int x = 0; x < xi; ++x) n += w32_owns(hf, x) ? 1 : 0;
    return n;
}
// A^T = [[1,1,1,0],[0,1,-1,-1]]
__host__ __device__ constexpr int w32_at(int a, int r) { return a == 0 ? (r < 3 ? 1 : 0) : (r == 0 ? 0 : (r == 1 ? 1 : -1)); }

// Activation with the kind as a compile-time constant (w32_act / w32_act2, wmd_conv_common.h): the epilogue selects ONE instantiation
// of its store loop per launch (a run-time switch per element put ~100 scalar branches into every wave's epilogue: 12 k -> 5.5 k
// cycles on L14).

// GENERIC = false: every chunk of CK channels lies inside one source tensor, no masks (every trunk layer of the decoders).
//   The chunk's patch is staged as ONE flattened [channel][position] run: every LDS-DMA instruction moves 64 consecutive
//   dwords of it from per-lane offsets that are fixed for the whole layer (channel x plane + gathered position), the chunk is
//   a scalar base -- no per-piece predication, channel arithmetic or branches in the MFMA stream; the upsampled operand takes
//   the structured low-resolution path.
// GENERIC = true: ragged channel counts, chunks that straddle the concat boundary, block-sparse masks (in_mask / out_mask of
//   the sparse decoders): per-channel pieces with a per-piece source choice, upsampling through the >>1 gather.
// MASKED: the instantiation carries the block-sparse mask code (always with GENERIC; the flattened staging has a dense and a
// masked instantiation -- the mask code in the epilogue cost the dense launches 2-3 % when they shared one).
// LIST (with MASKED, !GENERIC): the work-list form -- the block's tile comes out of a compacted list of active tiles and the
//   K split is chosen on the device from the list's length (wmd_conv_args.out_tiles).  A block beyond the list returns at once.
template <int TH, int TW, int WN, int CK, bool GENERIC, bool MASKED, bool LIST = false>
__global__ __launch_bounds__(WN * 128, 2) void conv_wino32_kernel(const ConvKArgs a) {
    using T = W32Tile<TH, TW, WN, CK>;
    constexpr int NT = T::NT, PWS = T::PWS, PSF = T::PSF, PWL = T::PWL, PSL = T::PSL;
    constexpr int KW = T::KW, TXB = T::TXB;
    static_assert(MASKED || !GENERIC, "the generic instantiation carries the mask code");
    static_assert(!LIST || (MASKED && !GENERIC), "the work-list form is a masked, flattened-staging instantiation");
    // masks are tested in the prologue (input: folded into the gather offsets) and the epilogue only
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS + T::TAB_FLOATS + 16];   // + one tile-activity flag per wave

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef WMD_STAMPS
    unsigned long long stamp_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    WMD_STAMP(0);
    // waves w and w + WN are the two halves of tile group w % WN; a workgroup's waves are dealt to the four SIMDs cyclically,
    // so for WN = 4 the halves of a group share a SIMD (their MFMA counts on an upsampled chunk, 5 + 4, add up evenly)
    const int wn = wave % WN;
    const int hf = wave / WN;

    int t, by;
    int tick;   // this block's (pixel tile, slab) item: index of its split-K ticket counter (splitk_ticket_finish)
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
    } else if (a.cob > 0) {   // (pixel tile, out-channel slab) items, slab fastest, one contiguous run per XCD (see conv_fwd_kernel)
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
    WMD_STAMP_AFTER(1, y0 + x0 + H);   // kernel arguments have arrived, the item is decoded
    // Block-sparse: a tile without active output pixels keeps its zeros.  It does not `return` here: an early exit ahead of
    // the pipelined body costs the DENSE launches of this same instantiation 45 % (L14: 110 -> 160 us, A/B builds on one box;
    // the epilogue's mask code costs nothing) -- the skipped block walks an empty chunk range and stores nothing instead
    // (prologue + an epilogue over zeros: a few thousand cycles against ~50 k for a computed tile).
    bool skip = false;
    if (MASKED && !LIST && a.out_mask) {
        int any = 0;
        for (int i = tid; i < TH * TW; i += NT) {
            const int yy = y0 + i / TW, xx = x0 + i % TW;
            if (yy < H && xx < W) any |= a.out_mask[(size_t)b * H * W + (size_t)yy * W + xx];
        }
        // (a ballot per wave + flags in LDS; __syncthreads_or pulls in the device library's work-group reduction)
        int* flags = reinterpret_cast<int*>(lds + T::LDS_FLOATS + T::TAB_FLOATS);
        const bool wave_any = __builtin_amdgcn_ballot_w64(any != 0) != 0;
        if (lane == 0) flags[wave] = wave_any ? 1 : 0;
        __syncthreads();
        int all = 0;
#pragma unroll
        for (int w = 0; w < T::NW; ++w) all |= flags[w];
        skip = __builtin_amdgcn_readfirstlane(all) == 0;
    }
    const bool upl = !GENERIC && a.up1 == 2;   // structured low-resolution path of the upsampled operand

    // ---- staging geometry ---------------------------------------------------------------------------------------
    constexpr unsigned kOOB = 0x80000000u;   // >= any num_records: the descriptor's range check returns 0
    const size_t plane1 = (size_t)a.H1 * a.W1, plane2 = (size_t)H * W;
    const unsigned pb1 = (unsigned)(plane1 * 4), pb2 = (unsigned)(plane2 * 4);
    // Padded coordinate g in [-1, n] -> source coordinate, branch-free (the pad mode is uniform, everything is a select): every
    // block of a launch computes its offsets at the same moment, so this prologue is not hidden by anything, and the branchy
    // form (pad_coord's switch per coordinate, short-circuit tests) cost 16 k cycles per block on L14 (cycle stamps)
    auto fold = [&](int g, int n, int& ok) {
        const int refl = g < 0 ? -g : (g >= n ? 2 * n - 2 - g : g);
        const int clam = min(max(g, 0), n - 1);
        ok &= (int)(a.pad_mode != WMD_PAD_ZERO) | (int)(g == clam);
        const int r = a.pad_mode == WMD_PAD_REFLECT ? refl : clam;
        return min(max(r, 0), n - 1);      // tile overhang beyond the pad ring: any in-range pixel (never stored)
    };
    // byte offset of full-resolution patch position p inside one channel plane of x2 (o2) / of x1 (o1): generic layers
    auto full_pos = [&](int p, unsigned& o1, unsigned& o2) {
        const int py = p / PWS, px = p - py * PWS;
        const int gy0 = y0 + py - 1, gx0 = x0 + px - 1;
        int ok = (int)(p < PSF) & (int)(gy0 <= H) & (int)(gx0 <= W);
        const int gy = fold(gy0, H, ok), gx = fold(gx0, W, ok);
        if (MASKED && a.in_mask) ok &= (int)(a.in_mask[(size_t)b * H * W + gy * W + gx] != 0);
        o2 = ok ? (unsigned)(gy * W + gx) * 4u : kOOB;
        const int sy = a.up1 == 2 ? gy >> 1 : gy - a.shift1, sx = a.up1 == 2 ? gx >> 1 : gx - a.shift1;
        const int ok1 = ok & (int)(sy >= 0) & (int)(sx >= 0) & (int)(sy < a.H1) & (int)(sx < a.W1);
        o1 = ok1 ? (unsigned)(sy * a.W1 + sx) * 4u : kOOB;
    };
    // pure layers: flattened [channel][position] runs of a chunk, element tid + i * NT
    constexpr int NPF = GENERIC ? 1 : (CK * PSF + NT - 1) / NT, NPL = GENERIC ? 1 : (CK * PSL + NT - 1) / NT;
    // 16-byte pieces (dense flattened staging, tiles whose patch columns all lie inside the source rows): groups per thread
    constexpr bool X4 = !GENERIC && !MASKED && T::X4OK;
    constexpr int NPF4 = (CK * PSF / 4 + NT - 1) / NT, NPL4 = (CK * PSL / 4 + NT - 1) / NT;
    bool x4 = false;
    // generic layers: positions tid + i * NT of one channel
    constexpr int NPOSF = GENERIC ? T::NPOSF : 1;
    unsigned obF[NPF], obL[NPL], ob1[NPOSF], ob2[NPOSF];
    if constexpr (!GENERIC) {
        // The full-resolution chunks of a pure layer all have one geometry: the skip tensor's when x1 is upsampled (x1 then
        // takes the low-resolution path), else x1's own (a skip tensor beside a non-upsampled x1 has the same H x W).  The
        // pad / bounds logic is separable: PH + PWS (+ PHL + PWL) threads fold one patch row or column each into a byte
        // offset (or -1: reads zero) in a corner of LDS, and every flattened element is then channel * plane + row + column:
        // ~20 instructions per element instead of ~80.
        // Low-resolution halo patch of the upsampled operand: position (py, px) is source pixel (y0/2 - 1 + py, x0/2 - 1 + px).
        // The full-resolution pad ring (row -1 / row H) maps onto it as: reflect (-1 -> 1, H -> H-2) and replicate both land in
        // the border source pixel, zero padding stays zero; rows beyond the ring (tile overhang) are never used by a stored output.
        int* tab = reinterpret_cast<int*>(lds + T::LDS_FLOATS);
        constexpr int PH = T::PH, PHL = T::PHL;
        const bool up_g = a.up1 == 2;
        const int Hs_g = up_g ? H : a.H1, Ws_g = up_g ? W : a.W1, sh_g = up_g ? 0 : a.shift1;
        // folded byte offset of full-resolution patch row / column t (-1: reads zero), and of the low-resolution patch's
        auto full_rc = [&](bool row, int t) {
            const int g0 = row ? y0 + t - 1 : x0 + t - 1, n = row ? H : W, ns = row ? Hs_g : Ws_g;
            int ok = (int)(g0 <= n);
            const int g = fold(g0, n, ok) - sh_g;
            ok &= (int)(g >= 0) & (int)(g < ns);
            return ok ? (row ? g * Ws_g * 4 : g * 4) : -1;
        };
        auto low_rc = [&](bool row, int t) {
            const int s0 = row ? (y0 >> 1) - 1 + t : (x0 >> 1) - 1 + t, n = row ? a.H1 : a.W1;
            const int sc = min(max(s0, 0), n - 1);
            const int ok = (int)(s0 <= n) & ((int)(a.pad_mode != WMD_PAD_ZERO) | (int)(sc == s0));
            return ok ? (row ? sc * a.W1 * 4 : sc * 4) : -1;
        };
        // (Round 5 measured the alternative -- every element folding its own row and column, no LDS table, no barrier: 40 VALU
        //  instructions per element instead of two table reads.  Slower: offsets phase 3.4 k -> 4.4 k cycles on L14, 2.8 k -> 6.5 k on
        //  the 40-wide tiles; a young wave's VALU instructions get the issue slots its SIMD partner's main loop leaves over, so the
        //  prologue's price is its VALU count.  s_setprio 1 through the prologue and / or the epilogue: no change either.
        //  profiles/r05_stamps.txt, profiles/r05_notes.md.)
        {
            int t = tid;
            if (t < PH + PWS) tab[t] = full_rc(t < PH, t < PH ? t : t - PH);
            else if (t < PH + PWS + PHL + PWL) {
                t -= PH + PWS;
                tab[PH + PWS + t] = low_rc(t < PHL, t < PHL ? t : t - PHL);
            }
        }
        __syncthreads();
        WMD_STAMP_AFTER(2, 0);   // row / column tables in LDS
        const unsigned pbs = a.up1 == 2 ? pb2 : pb1;
        if constexpr (X4) {
            // every used patch column x0 - 1 .. x0 + TW (minus the shift of a data-gradient launch) inside the source row: no
            // column is folded, a row of the patch is one contiguous run of the source row (the upsampled operand's low-resolution
            // patch likewise: (x0 >> 1) - 1 .. (x0 + TW) >> 1).  Border tiles keep the dword pieces with their per-column folds.
            const bool up = a.up1 == 2;
            const int Ws = up ? W : a.W1, sh = up ? 0 : a.shift1;
            x4 = x0 - 1 - sh >= 0 && x0 + TW - sh < Ws && !a.no_x4;
        }
        if (X4 && x4) {
            const bool up = a.up1 == 2;
            const int sh = up ? 0 : a.shift1;
            constexpr int PH_ = T::PH, GF = T::GF, GL = T::GL, PHL_ = T::PHL;
#pragma unroll
            for (int i = 0; i < NPF4; ++i) {
                const unsigned e = tid + i * NT, ch = e / (PH_ * GF), rem = e - __umul24(ch, PH_ * GF);
                const unsigned py = rem / GF, j = rem - __umul24(py, GF);
                const int r = tab[min(py, (unsigned)PH_ - 1)];
                const bool ok = ch < CK && r >= 0;
                obF[i] = ok ? ch * pbs + (unsigned)r + (unsigned)(x0 - 1 - sh + 4 * (int)j) * 4u : kOOB;
            }
#pragma unroll
            for (int i = 0; i < NPL4; ++i) {
                const unsigned e = tid + i * NT, ch = e / (PHL_ * GL), rem = e - __umul24(ch, PHL_ * GL);
                const unsigned py = rem / GL, j = rem - __umul24(py, GL);
                const int r = tab[PH + PWS + min(py, (unsigned)PHL_ - 1)];
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
            // block-sparse input support: the full-resolution geometry is the mask's own (wino32_pure), so the folded pixel
            // offset indexes it directly; a masked position gathers from the out-of-range offset like a zero-padded one
            if (MASKED && a.in_mask && ok) ok = a.in_mask[(size_t)b * plane2 + ((unsigned)(r + c) >> 2)] != 0;
            obF[i] = ok ? ch * pbs + (unsigned)(r + c) : kOOB;
        }
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const unsigned e = tid + i * NT, ch = e / PSL, pos = e - __umul24(ch, PSL);
            const unsigned py = pos / PWL, px = pos - __umul24(py, PWL);
            const int r = tab[PH + PWS + py], c = tab[PH + PWS + PHL + px];
            bool ok = ch < CK && (r | c) >= 0;
            if (MASKED && a.in_mask && upl && ok) {   // 2x2-constant mask (wino32_pure): source pixel (sy, sx) is masked like (2sy, 2sx)
                const unsigned sidx = (unsigned)(r + c) >> 2, sy = sidx / (unsigned)a.W1, sx = sidx - sy * (unsigned)a.W1;
                ok = a.in_mask[(size_t)b * plane2 + (size_t)(2 * sy) * W + 2 * sx] != 0;
            }
            obL[i] = ok ? ch * pb1 + (unsigned)(r + c) : kOOB;
        }
        }   // dword pieces
    } else {
#pragma unroll
        for (int i = 0; i < NPOSF; ++i) full_pos(tid + i * NT, ob1[i], ob2[i]);
    }
    WMD_STAMP_AFTER(3, 0);   // per-lane gather offsets
    const float* x1b = a.x1 + (size_t)b * a.C1 * plane1;
    const float* x2b = a.x2 ? a.x2 + (size_t)b * a.C2 * plane2 : a.x1;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x1b), 0, (int)(a.C1 * plane1 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x2b), 0, (int)(a.C2 * plane2 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.wp), 0, (int)((size_t)a.ncot * a.nci4 * 16 * 64 * 4), 0x00020000);
    // weights: the slab's two 16-channel runs of a chunk as 16-byte pieces; piece e -> run e / (CK*64), slot e % (CK*64)
    unsigned aoff[T::NAV];
#pragma unroll
    for (int v = 0; v < T::NAV; ++v) {
        const int e = tid + v * NT;
        const int run = e / (CK * 64), rem = e % (CK * 64);
        const int cot = min(by * 2 + run, a.ncot - 1);
        aoff[v] = e < 2 * CK * 64 ? (unsigned)(((size_t)cot * a.nci4 * 16 * 64 + (size_t)rem * 4) * 4) : kOOB;
    }

#ifdef WMD_STAMPS
    const int c_begin_dbg = LIST ? 0 : (int)blockIdx.z * a.chunks_per_split;
#endif
    // Chunk kinds: UP = CK channels of the upsampled x1, staged at low resolution (pure layers only); FULL = anything else.
    auto is_up = [&](int chunk) { return upl && (chunk + 1) * CK <= a.C1; };
    // wave-instructions per chunk and wave: pure layers move whole 64-dword runs (the tail lanes of the last one carry the
    // out-of-range offset and write zeros into the buffer's padding), generic layers one channel's positions at a time
    constexpr int NPB_F = GENERIC ? CK * NPOSF : NPF, NPB_L = NPL;
    struct ChunkSrc {   // pure layers: the chunk's descriptor and scalar byte base, resolved once per chunk
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
        const int e0 = wave * 64 + v * NT;   // first piece of this wave-instruction (a multiple of 64: inside one run)
        if (T::NAV * NT == 2 * CK * 64 || e0 < 2 * CK * 64)   // wave-uniform: whole wave-instructions only
            lds_dma16(rw, (lds_ptr_t)(bufp + T::B_FLOATS + (e0 / (CK * 64)) * T::RUN_LDS + (e0 % (CK * 64)) * 4), aoff[v], soffA);
    };
    auto stage_full_piece = [&](int chunk, float* bufp, int q, const ChunkSrc& cs) {
#ifdef WMD_STAMPS
        if ((a.dbg_mode & 8) && q < NPB_F && chunk > c_begin_dbg + 1) return;   // timing experiment: no patch traffic either
#endif
        if (q < NPB_F) {
            if constexpr (!GENERIC) {
                if (X4 && x4) {      // block-uniform: 64 lanes x 16 bytes = 256 consecutive dwords of the chunk's run per instruction
                    if (q < NPF4 && ((q + 1) * NT * 4 <= CK * PSF || (wave * 64 + q * NT) * 4 < CK * PSF))
                        lds_dma16(cs.r, (lds_ptr_t)(bufp + (wave * 64 + q * NT) * 4), obF[q], cs.base);
                } else if ((q + 1) * NT <= CK * PSF || wave * 64 + q * NT < CK * PSF)   // wave-uniform: skip wholly empty runs
                    lds_dma4(cs.r, (lds_ptr_t)(bufp + wave * 64 + q * NT), obF[q], cs.base);
            } else {
                const int j = q / NPOSF, i = q % NPOSF;
                if ((i + 1) * NT <= PSF || tid + i * NT < PSF) {
                    lds_ptr_t d = (lds_ptr_t)(bufp + wave * 64 + j * PSF + i * NT);
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
            stage_weight_piece(chunk, bufp, q - NPB_F);
        }
    };
    auto stage_up_piece = [&](int chunk, float* bufp, int q) {
#ifdef WMD_STAMPS
        if ((a.dbg_mode & 8) && q < NPB_L && chunk > c_begin_dbg + 1) return;
#endif
        if (q < NPB_L) {
            if (X4 && x4) {
                if (q < NPL4 && ((q + 1) * NT * 4 <= CK * PSL || (wave * 64 + q * NT) * 4 < CK * PSL))
                    lds_dma16(r1, (lds_ptr_t)(bufp + (wave * 64 + q * NT) * 4), obL[q], (unsigned)(chunk * CK) * pb1);
            } else if ((q + 1) * NT <= CK * PSL || wave * 64 + q * NT < CK * PSL)
                lds_dma4(r1, (lds_ptr_t)(bufp + wave * 64 + q * NT), obL[q], (unsigned)(chunk * CK) * pb1);
        } else {
            stage_weight_piece(chunk, bufp, q - NPB_L);
        }
    };

    const int c_begin = ks * cps;
    const int c_end = skip ? c_begin : min(c_begin + cps, a.nchunks);
    WMD_STAMP(4);   // descriptors, weight offsets
    if (c_begin < c_end) {
        if (is_up(c_begin)) {
            static_for<NPB_L + T::NAV>([&](auto qc) { stage_up_piece(c_begin, lds, decltype(qc)::value); });
        } else {
            const ChunkSrc cs0 = chunk_src(c_begin);
            static_for<NPB_F + T::NAV>([&](auto qc) { stage_full_piece(c_begin, lds, decltype(qc)::value, cs0); });
        }
    }
    const int co = by * 32 + (lane & 31);
    const float bias_v = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;   // requested now, used in the epilogue

    // ---- operand addressing -------------------------------------------------------------------------------------
    const int tslot = min(wn * 32 + (lane & 31), T::NTILES - 1);
    const int tyy = tslot / TXB, txx = tslot % TXB;
    const int pbF = tyy * 2 * PWS + txx * 2 + (lane >> 5) * PSF;
    const int pbL = tyy * PWL + txx + (lane >> 5) * PSL;
    const int wbase = T::B_FLOATS + ((lane & 31) >> 4) * T::RUN_LDS + (lane & 15) + 16 * (lane >> 5);
    WMD_STAMP(5);   // first chunk issued
    __syncthreads();
    WMD_STAMP(6);   // first chunk landed

    // Everything from here on is compiled once per half (the set of owned positions is a compile-time property).
    auto run = [&](auto hf_tag) {
        constexpr int HF = decltype(hf_tag)::value;
        f32x16 acc[8];
#pragma unroll
        for (int o = 0; o < 8; ++o)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[o][r] = 0.f;

        // One chunk of this wave: KW K-steps x NP owned positions, one MFMA each.  Software pipeline pinned with sched_barrier:
        // the weight fragment of MFMA s + D, the next K-step's patch (first gap), its transform (third and fourth gap) and the
        // next chunk's DMA pieces (dealt over the first two thirds) are issued in the shadow of earlier MFMAs -- this wave's
        // and those of the other wave on the SIMD.
        //   UP: 0 = full-resolution chunk, 1 = low-resolution chunk of the upsampled operand
        //   NEXT: 0 = last chunk, 1 = the next chunk is FULL, 2 = the next chunk is UP
        auto chunk_body = [&](int c, auto up_tag, auto next_tag) {
            constexpr bool UP = decltype(up_tag)::value;
            constexpr int NEXT = decltype(next_tag)::value;
            constexpr int NP = w32_count(HF, UP);
            constexpr int S = KW * NP, D = 3, RS = D + 1;
                        // (round 5, A/B builds: dealing the pieces over the first third or half instead moved nothing -- bench 0.601 / 0.601 / 0.599 ms)
            constexpr int SP = (S * 2) / 3 > 0 ? (S * 2) / 3 : 1;
            constexpr int NPIECES = NEXT == 0 ? 0 : (NEXT == 2 ? NPB_L : NPB_F) + T::NAV;
            const int buf = (c - c_begin) & 1;
            const float* bufp = lds + buf * T::BUF_FLOATS;
            float* nbufp = lds + (buf ^ 1) * T::BUF_FLOATS;
            const float* psrc = bufp + (UP ? pbL : pbF);
            const float* wsrc = bufp + wbase;
            ChunkSrc csn;
            if constexpr (NEXT == 1) csn = chunk_src(c + 1);
            // The input transform in packed fp32 (v_pk_add_f32, two adds per lane per instruction): fp32 MFMA and the vector ALU are
            // the same hardware, every VALU instruction between two MFMAs comes out of the matrix rate
            // (tools/probes/mfma32_issue_probe.hip: one packed add costs what one plain add costs).  Pairs run along patch columns.
            //   FULL: dp[2 r + h] = (d[r][2h], d[r][2h+1]) straight from the 8-byte LDS reads; rows mix as whole pairs, the
            //         column mix of a row is two instructions with half selects (w32_pk_*), owned pairs only
            //   UP:   the 3x3 source pixels; pairs (d[r][0], d[r][1]) + the third column as scalars
            // volatile LDS-space reads: one ds_read with an immediate offset each.  Left alone the compiler pairs them into
            // ds_read2 forms whose 8-bit offsets do not reach a K-step's rows and spends a v_add_u32 per pair on a new base --
            // and a VALU instruction costs the matrix pipe several times what a second ds_read does (mainloop_replica_probe).
            typedef const volatile __attribute__((address_space(3))) float* lds_cv1_t;
            typedef const volatile __attribute__((address_space(3))) f32x2* lds_cv2_t;
            const lds_cv1_t psrc3 = (lds_cv1_t)psrc;
            f32x2 dp[8], tp[8], vp[2][8];
            float d2[3], wf[RS];
            auto fetch_patch = [&](int kk) {   // only the rows this half's positions depend on (a volatile read is never dropped)
                if constexpr (UP) {   // half 0: source rows 0, 1; half 1: rows 1, 2
#pragma unroll
                    for (int r = HF; r < HF + 2; ++r) {
                        const lds_cv1_t pr = psrc3 + kk * 2 * PSL + r * PWL;
                        dp[r] = f32x2{pr[0], pr[1]};
                        d2[r] = pr[2];
                    }
                } else {              // half 0: patch rows 0, 1, 2; half 1: rows 1, 2, 3
#pragma unroll
                    for (int e = 2 * HF; e < 2 * HF + 6; ++e)
                        dp[e] = *reinterpret_cast<lds_cv2_t>(psrc3 + kk * 2 * PSF + (e >> 1) * PWS + (e & 1) * 2);
                }
            };
            auto transform_rows = [&]() {   // tr = B^T d  (UP: rows 0, 1, 3 from the 3x3 source pixels, whichever the half uses)
                if constexpr (UP) {
                    if constexpr (HF == 0) {
                        tp[0] = w32_pk_sub(dp[0], dp[1]);  // tr[0][0..1]
                        tp[1] = w32_pk_add(dp[1], dp[1]);  // tr[1][0..1]
                        tp[2][0] = d2[0] - d2[1];         // tr[0][2]
                    } else {
                        tp[0] = w32_pk_sub(dp[1], dp[2]);  // tr[3][0..1]
                        tp[1][0] = dp[1][1] - d2[1];      // (tr[1][1] - tr[1][2]) / 2
                        tp[2][0] = d2[1] - d2[2];         // tr[3][2]
                    }
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if constexpr (HF == 0) tp[0 + h] = w32_pk_sub(dp[0 + h], dp[4 + h]);
                        tp[2 + h] = w32_pk_add(dp[2 + h], dp[4 + h]);
                        tp[4 + h] = w32_pk_sub(dp[4 + h], dp[2 + h]);
                        if constexpr (HF == 1) tp[6 + h] = w32_pk_sub(dp[2 + h], dp[6 + h]);
                    }
                }
            };
            auto transform_cols = [&](int kk) {   // V = tr B, owned positions only; position 4 r + c is vp[..][2 r + c / 2][c % 2]
                f32x2* vv = vp[kk & 1];
                if constexpr (UP) {
                    if constexpr (HF == 0) {   // (0,0) (0,1) (0,3) (1,0) (1,1)
                        vv[0] = w32_pk_up(tp[0]);
                        vv[1][1] = tp[0][1] - tp[2][0];
                        vv[2] = w32_pk_up(tp[1]);
                    } else {                   // (1,3) (3,0) (3,1) (3,3)
                        vv[3][1] = tp[1][0] + tp[1][0];
                        vv[6] = w32_pk_up(tp[0]);
                        vv[7][1] = tp[0][1] - tp[2][0];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (w32_owns(HF, r * 4 + 0)) vv[2 * r] = w32_pk_lo(tp[2 * r], tp[2 * r + 1]);
                        if (w32_owns(HF, r * 4 + 2)) vv[2 * r + 1] = w32_pk_hi(tp[2 * r], tp[2 * r + 1]);
                    }
                }
            };
            auto fetch_u = [&](int s2) {
                const int kk = s2 / NP, xi = w32_nth(HF, UP, s2 % NP);
                wf[s2 % RS] = wsrc[((kk >> 1) * 16 + xi) * 64 + (kk & 1) * 32];
            };
            // cold start of the chunk: first patch, first fragments, first transform
            fetch_patch(0);
#pragma unroll
            for (int s2 = 0; s2 < D && s2 < S; ++s2) fetch_u(s2);
            transform_rows();
            transform_cols(0);
            static_for<S>([&](auto s2c) {
                constexpr int s2 = decltype(s2c)::value;
                constexpr int kk = s2 / NP, p = s2 % NP;
                constexpr int xi = w32_nth(HF, UP, p);
                if constexpr (s2 + D < S) fetch_u(s2 + D);
                if constexpr (kk + 1 < KW) {
                    if constexpr (p == 0) fetch_patch(kk + 1);
                    if constexpr (p == 2) transform_rows();
                    if constexpr (p == 3) transform_cols(kk + 1);
                }
                if constexpr (NEXT != 0) {
                    // pieces q with q * SP / NPIECES == s2, i.e. q in [ceil(s2 * NPIECES / SP), ceil((s2 + 1) * NPIECES / SP))
                    constexpr int q0 = (s2 * NPIECES + SP - 1) / SP, q1 = ((s2 + 1) * NPIECES + SP - 1) / SP;
                    constexpr int qb = q0 < NPIECES ? q0 : NPIECES, qe = q1 < NPIECES ? q1 : NPIECES;
                    static_for<qe - qb>([&](auto qc) {
                        constexpr int q = qb + decltype(qc)::value;
                        if constexpr (NEXT == 2) stage_up_piece(c + 1, nbufp, q);
                        else stage_full_piece(c + 1, nbufp, q, csn);
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[w32_slot(HF, xi)] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[kk & 1][xi >> 1][xi & 1], wf[s2 % RS], acc[w32_slot(HF, xi)], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            __syncthreads();   // next buffer landed (vmcnt(0) precedes the barrier), this one is released
        };
        // Two plain loops (upsampled chunks first, then the rest) with peeled last iterations: one body per loop keeps the
        // accumulators in place across the back edge (a single loop that switches between body variants makes the register
        // allocator rotate the 128 accumulator registers through copies and spills).
        const int up_end = min(c_end, max(c_begin, upl ? a.C1 / CK : 0));
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        int c = c_begin;
        if constexpr (!GENERIC) {
            for (; c + 1 < up_end; ++c) chunk_body(c, std::true_type{}, I2{});
            if (c < up_end) {
                if (up_end < c_end) chunk_body(c, std::true_type{}, I1{});
                else chunk_body(c, std::true_type{}, I0{});
                ++c;
            }
        }
        for (; c + 1 < c_end; ++c) chunk_body(c, std::false_type{}, I1{});
        if (c < c_end) chunk_body(c, std::false_type{}, I0{});

        WMD_STAMP(7);   // main loop done
        // ---- epilogue -------------------------------------------------------------------------------------------------
        // Accumulator register g = 4q + r of lane l belongs to tile slot wn*32 + 8q + 4(l >> 5) + r and out channel l & 31.
        // Partial outputs of this half: Y[a][b] += A^T[a][r] A^T[b][c] M[r][c] over the owned (r, c).  Half hf keeps output
        // row a = hf and hands row 1 - hf to the other half: 32 values per lane through LDS (the staging buffers are free).
        // Everything on register pairs (g, g + 1) -- neighbouring accumulator registers of one position -- so that the sums are
        // v_pk_add_f32: the epilogue's VALU instructions are paid by the SIMD's other wave like the main loop's (per tile and wave
        // ~450 scalar instructions before, more than the main loop of a 12-chunk layer issues).  The row sums are factored
        // (A^T's rows applied along c first): 12 packed sums per pair instead of 28 scalar ones.
        f32x2 kp[8][2], gv[8][2];   // [g / 2][b]: kept row, given row
#pragma unroll
        for (int gp = 0; gp < 8; ++gp) {
            auto M = [&](int xi) { return f32x2{acc[w32_slot(HF, xi)][2 * gp], acc[w32_slot(HF, xi)][2 * gp + 1]}; };
            if constexpr (HF == 0) {   // row 0 (all columns), (1,0) (1,1) (2,0) (2,1)
                const f32x2 b0 = (M(0) + M(1)) + M(2), b1 = (M(1) - M(2)) - M(3);
                const f32x2 p0 = M(4) + M(5), p1 = M(5), q0 = M(8) + M(9), q1 = M(9);
                kp[gp][0] = (b0 + p0) + q0;   // Y[0][0]
                kp[gp][1] = (b1 + p1) + q1;   // Y[0][1]
                gv[gp][0] = p0 - q0;          // Y[1][0]
                gv[gp][1] = p1 - q1;          // Y[1][1]
            } else {                   // (1,2) (1,3) (2,2) (2,3), row 3 (all columns)
                const f32x2 s1 = M(6) + M(7), s2 = M(10) + M(11);
                const f32x2 b0 = (M(12) + M(13)) + M(14), b1 = (M(13) - M(14)) - M(15);
                gv[gp][0] = M(6) + M(10);            // Y[0][0]
                gv[gp][1] = -s1 - s2;                // Y[0][1]
                kp[gp][0] = (M(6) - M(10)) - b0;     // Y[1][0]
                kp[gp][1] = (s2 - s1) - b1;          // Y[1][1]
            }
        }
        WMD_STAMP(8);   // output transform
        float* xch = lds + (size_t)wn * (2 * 32 * 64) + lane * 2;   // [wn][sender half][pair j = 2 (g / 2) + b][lane][2]
#pragma unroll
        for (int j = 0; j < 16; ++j) *reinterpret_cast<f32x2*>(xch + (HF * 16 + j) * 128) = gv[j >> 1][j & 1];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) kp[j >> 1][j & 1] += *reinterpret_cast<const f32x2*>(xch + ((1 - HF) * 16 + j) * 128);

        WMD_STAMP(9);   // halves exchanged
        const bool final_out = (ks_n == 1);
        float* ybase = (LIST && final_out) ? a.y_final + (size_t)b * a.Cout * plane2 : a.y + ((size_t)ks * a.B + b) * a.Cout * plane2;
        const bool vec_ok = (W & 3) == 0;
        const bool lines = !MASKED && a.st_coalesce != 0 && vec_ok;
        const bool wt = a.tickets != nullptr && !final_out;   // split-K partial of a launch that finishes in-kernel: write-through stores
        const __amdgpu_buffer_rsrc_t ry = agent_rsrc(ybase, (size_t)a.Cout * plane2 * 4);
        // bias + activation in place (ACT < 0: split-K partial sums, neither), then one of the two store forms
        auto finish = [&](auto act_tag) {
            constexpr int ACT = decltype(act_tag)::value;
            if constexpr (ACT >= 0) {
                const f32x2 bias2 = f32x2{bias_v, bias_v};
#pragma unroll
                for (int gp = 0; gp < 8; ++gp)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) kp[gp][bb] = w32_act2<(ACT < 0 ? 0 : ACT)>(kp[gp][bb] + bias2, a.slope);
            }
            auto keep = [&](int g, int bb) { return kp[g >> 1][bb][g & 1]; };   // output value of accumulator register g, pixel bb
            if (!lines) {
                // a lane stores 8-pixel runs of ONE out channel (masked launches, widths that are not a multiple of 4)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int tfirst = wn * 32 + 8 * q + 4 * (lane >> 5);
                    const int oy = y0 + (tfirst / TXB) * 2 + HF, ox = x0 + (tfirst % TXB) * 2;
                    if (co >= a.Cout || tfirst >= T::NTILES || oy >= H || ox >= W || skip) continue;
                    float* dst = ybase + (size_t)co * plane2 + (size_t)oy * W + ox;
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = keep(4 * q + (e >> 1), e & 1);
                    if (MASKED && a.out_mask) {   // branch-free: clamped byte loads + selects (elements past W are never stored)
                        const uint8_t* mp = a.out_mask + (size_t)b * plane2 + (size_t)oy * W;
                        uint8_t mv[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) mv[e] = mp[min(ox + e, W - 1)];
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = mv[e] ? o[e] : 0.f;
                    }
#ifdef WMD_STAMPS
                    if ((a.dbg_mode & 1) && o[0] == o[0]) continue;
#endif
                    if (wt) {
                        const unsigned ob = (unsigned)(((size_t)co * plane2 + (size_t)oy * W + ox) * 4);
                        if (vec_ok && ox + 7 < W) {
                            st16_agent(ry, ob, make_float4(o[0], o[1], o[2], o[3]));
                            st16_agent(ry, ob + 16, make_float4(o[4], o[5], o[6], o[7]));
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (ox + e < W) st4_agent(ry, ob + 4 * e, o[e]);
                        }
                    } else if (vec_ok && ox + 7 < W) {
                        *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                        *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (ox + e < W) dst[e] = o[e];
                    }
                }
                return;
            }
            // Round 5: the kept output row leaves through whole 128-byte lines.  A lane holds 8-pixel runs of ONE out channel, so a
            // store instruction of the form above touches 64 different lines with 16 bytes each (cycle stamps: 4.2 k cycles from the
            // exchange to the last store issued).  Here the wave first transposes its 32 channels x 32 tile slots x 2 pixels through the
            // 8 KB of the exchange area only it has read ([co][16 pieces of two tiles], pieces XOR-swizzled by the channel so that the
            // 8-lane write groups and the 16-lane read groups each cover distinct banks) and then stores [4 channels][256 bytes] per
            // instruction.  LDS operations of one wave execute in order: no barrier.
            float* tb = lds + (size_t)wn * (2 * 32 * 64) + (size_t)(1 - HF) * (32 * 64);
            const int col = lane & 31, hh = lane >> 5;
            const int swz_w = ((col & 3) << 2) | ((col >> 2) & 3);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int g = 4 * q + 2 * pr, pc = 4 * q + 2 * hh + pr;
                    *reinterpret_cast<float4*>(tb + col * 64 + ((pc ^ swz_w) * 4)) = make_float4(keep(g, 0), keep(g, 1), keep(g + 1, 0), keep(g + 1, 1));
                }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = 4 * i + (lane >> 4), pc = lane & 15;
                const int swz_r = ((c & 3) << 2) | ((c >> 2) & 3);
                const float4 v = *reinterpret_cast<const float4*>(tb + c * 64 + ((pc ^ swz_r) * 4));
                const int ts = wn * 32 + 2 * pc;
                const int oy = y0 + (ts / TXB) * 2 + HF, ox = x0 + (ts % TXB) * 2;
                const int cg = by * 32 + c;
#ifdef WMD_STAMPS
                if ((a.dbg_mode & 1) && v.x == v.x) continue;
#endif
                if (cg < a.Cout && ts < T::NTILES && oy < H && ox < W) {
                    const size_t o4 = (size_t)cg * plane2 + (size_t)oy * W + ox;
                    if (wt) st16_agent(ry, (unsigned)(o4 * 4), v);
                    else *reinterpret_cast<float4*>(ybase + o4) = v;
                }
            }
        };
        int act_sel = final_out ? a.act : -1;
#ifdef WMD_STAMPS
        if ((a.dbg_mode & 2) && final_out) act_sel = WMD_ACT_NONE;
#endif
        if (act_sel < 0) finish(std::integral_constant<int, -1>{});
        else if (act_sel == WMD_ACT_ELU) finish(std::integral_constant<int, WMD_ACT_ELU>{});
        else if (act_sel == WMD_ACT_LEAKY) finish(std::integral_constant<int, WMD_ACT_LEAKY>{});
        else if (act_sel == WMD_ACT_SIGMOID) finish(std::integral_constant<int, WMD_ACT_SIGMOID>{});
        else finish(std::integral_constant<int, WMD_ACT_NONE>{});
        WMD_STAMP(10);   // stores issued
        if (wt)   // (uniform) the last K-slice block of this (tile, slab) to arrive sums the slices and writes the final tile
            splitk_ticket_finish<WN * 128, TH, TW, 32, MASKED>(a, reinterpret_cast<int*>(lds + T::LDS_FLOATS + T::TAB_FLOATS) + 8, tick, ks_n, b, y0, x0, by * 32);
#ifdef WMD_STAMPS
        if (a.dbg && tid == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            stamp_[11] = xcc;
            const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            for (int k = 0; k < 12; ++k) a.dbg[blk * 12 + k] = stamp_[k];
        }
#endif
    };
    if (hf == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
}

// ---- launchers (the configuration table lives in wmd_conv_fwd.hip) -----------------------------------------------
template <int TH, int TW, int WN, int CK>
void launch_wino32(const ConvKArgs& a, dim3 grid, hipStream_t s) {
    const bool masked = a.in_mask || a.out_mask;
    if constexpr (wino32_has_list(TH, TW, WN, CK)) {
        if (a.tile_list) {   // (the planner only offers list launches to pure layers on these tile shapes)
            hipLaunchKernelGGL((conv_wino32_kernel<TH, TW, WN, CK, false, true, true>), grid, dim3(WN * 128), 0, s, a);
            return;
        }
    }
    if (!wino32_pure(a, CK)) hipLaunchKernelGGL((conv_wino32_kernel<TH, TW, WN, CK, true, true>), grid, dim3(WN * 128), 0, s, a);
    else if (masked) hipLaunchKernelGGL((conv_wino32_kernel<TH, TW, WN, CK, false, true>), grid, dim3(WN * 128), 0, s, a);
    else hipLaunchKernelGGL((conv_wino32_kernel<TH, TW, WN, CK, false, false>), grid, dim3(WN * 128), 0, s, a);
}

#define WMD_W32_INST(TH, TW, WN, CK) template void launch_wino32<TH, TW, WN, CK>(const ConvKArgs&, dim3, hipStream_t);
#ifdef WMD_W32_DEV_TABLE   // development: compile-check one shape in seconds instead of the table in minutes
WMD_W32_INST(8, 32, 2, 8)
#else
#include "wmd_conv_wino32_table.inc"
#endif
#undef WMD_W32_INST

}  // namespace wmd
