// Threshold-gated sparse decoder path (batch 1) for gfx950.
//
// Reference: SparseDepthWaveProgressiveDecoder.forward (KITTI/networks/decoders/depth_decoder.py:292-428),
// SparseDecoderWave.forward (NYUv2/networks/decoders/densedepth_decoder.py:271-409) and the sparse_* helpers
// (KITTI/layers.py:337-507).  There every level is a storm of boolean-mask gathers, index_puts and small
// matmuls with a host sync (`.sum()` -> `arange(numel)`) per index map.  Here:
//   minmax_kernel            global min/max of the LL plane                      (depth_decoder.py:308)
//   mask_threshold_kernel    max_b |yh_b| > ratio * range                        (:308-309)
//   mask_dilate_multi_kernel every MaxPool2d(3|5)(upsample?) variant at once     (:311-319)
//   mask_compact_multi_kernel raster-order stream compaction: 16 flags per lane packed into a bit field + __popc, wavefront
//                            __shfl_up prefix sums (a lane's bit field plays the role of a 16-wide ballot), LDS scan over the 16 wavefronts,
//                            one workgroup per mask, count left on the device     (layers.py:371-389)
//   sparse_conv_kernel       gather-GEMM on fp32 MFMA over the compacted active pixels, fused
//                            select/upsample/concat/pad/bias/activation/scatter   (layers.py:337-507)
// Activations stay dense and zero-initialised; "not in the input mask => reads 0" is a mask test after the
// coordinate padding, which is exactly what padding the index map does in the reference (layers.py:444).
#include <algorithm>
#include <cstring>
#include "wmd_internal.h"

namespace wmd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void minmax_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out2) {
    x += (size_t)blockIdx.x * n;      // blockIdx.x = frame of a batch: x [B,n], out2 [B,2]
    out2 += 2 * blockIdx.x;
    float lo = INFINITY, hi = -INFINITY;
    for (size_t i = threadIdx.x; i < n; i += 1024) {
        const float v = x[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    __shared__ float slo[16], shi[16];
    if ((threadIdx.x & 63) == 0) {
        slo[threadIdx.x >> 6] = lo;
        shi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) {
            lo = fminf(lo, slo[w]);
            hi = fmaxf(hi, shi[w]);
        }
        out2[0] = lo;
        out2[1] = hi;
    }
}

__global__ void mask_threshold_kernel(const float* __restrict__ yh, const float* __restrict__ minmax, float ratio,
                                      uint8_t* __restrict__ mask, int npix) {
    const float thr = (minmax[1] - minmax[0]) * ratio;  // fp32, like the reference's 0-dim tensor arithmetic
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const float m = fmaxf(fmaxf(fabsf(yh[i]), fabsf(yh[npix + i])), fabsf(yh[2 * npix + i]));
        mask[i] = m > thr ? 1 : 0;
    }
}

// One tiled kernel behind wmd_mask_dilate_multi and wmd_mask_level: a block owns a 16x16 cell tile of the COARSE grid, puts the
// base mask of the tile + a 3-cell halo into LDS (FROM_MASK: copied from the given mask; else thresholded from the three
// coefficient bands, one evaluation per cell) and writes every spec's pixels of its tile from there -- a dilated pixel is the
// OR of the coarse cells its clamped window covers (out-of-range taps never win: MaxPool2d pads with -inf).  The untiled
// forms re-evaluated up to 25 cells x 3 bands per output pixel and spec (2 M loads per 640x192 frame at the finest level:
// 54 us per level at 12 frames).  Optional counts: the set pixels of a spec are added to spec.nnz[frame * nnz_stride] (one
// atomic per wavefront and pass), which is all the block-sparse decoders need of a compaction.
constexpr int ML_T = 16, ML_HALO = 3, ML_P = ML_T + 2 * ML_HALO;
// work-list form (wmd_mask_level_lists): per spec the tile list / count column, per launch the scratch + ring protocol
struct MaskListSpec {
    int tile_h, tile_w;    // 0: no list
    int32_t* tile_list;
    int32_t* tile_count;
    int count;             // k > 0: pixel count column k - 1
    const uint8_t* and_mask;   // optional: out = dilation AND and_mask (wmd_level_spec.and_mask)
};
struct MaskLevelKArgs {
    const float* mm;       // optional [B,2] precomputed (min, max) of every frame's yl: skips the in-block reduction
    const float* yl;
    const float* yh;
    const uint8_t* mask0;  // FROM_MASK: the base mask [B,h,w]
    float ratio;
    int n_yl, h, w, n, tiles_x;
    wmd_dilate_spec s[8];
    // ---- work-list form ----
    unsigned* range_keys;  // optional [B][2] ordered keys of (min, max), consumed and re-armed
    int32_t* scratch;      // non-null = work-list form
    int32_t* ring;
    int ring_slots, slot_ints, counts_off, ncounts, advance, B;
    MaskListSpec ls[8];
};
// scratch layout (int32, all zero at rest): [0] ticket of the frames, [1] forward sequence number; per frame f at
// kMlFrame0 + f * kMlFrameInts: [0] ticket of the frame's blocks, [1 + column] pixel-count accumulators, [16 + spec] tile
// accumulators.  Everything is per frame so that no address sees more than a few hundred atomics per launch: one ticket for
// the whole launch serialised 2 160 blocks at ~14 ns each (38 us at 12 frames).
constexpr int kMlTicket = 0, kMlSeq = 1, kMlFrame0 = 16, kMlFrameInts = 32, kMlFTicket = 0, kMlFCnt0 = 1, kMlFTile0 = 16;

__device__ __forceinline__ float key_to_float(unsigned k) {   // inverse of wmd_head_shiftsum_args.range_keys' encoding
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// One spec over one tile: a pixel's window covers coarse rows cy0..cy1 (at most NC) and columns cx0..cx1; the OR over the
// window is the OR of the rows' bit words masked to the column range -- NC LDS words per pixel, no inner loops (a wave64
// VALU instruction takes four cycles: the per-cell loop of the first tiled version spent 8 k cycles per wavefront).
// (tbits: bit k = work-list tile k of the block's region holds a set pixel; tile_h = 0: no list)
template <int NC>
__device__ __forceinline__ int mask_tile_spec(const wmd_dilate_spec& sp, const unsigned* rows, int f, int h, int w, int ty0, int tx0,
                                              int tile_h, int tile_w, unsigned& tbits, const uint8_t* and_mask) {
    const int up = sp.up, r = sp.radius, sh = up == 2 ? 1 : 0;
    const int H = h * up, W = w * up, lts = 4 + sh;     // log2 of the tile side in output pixels
    uint8_t* out = sp.out + (size_t)f * H * W;
    if (and_mask) and_mask += (size_t)f * H * W;
    int cnt = 0;
    const int ntx_l = tile_w ? (1 << lts) / tile_w : 1;
    for (int p = threadIdx.x; p < (1 << (2 * lts)); p += 256) {
        const int y = ty0 * up + (p >> lts), x = tx0 * up + (p & ((1 << lts) - 1));
        if (y < H && x < W) {
            const int cy0 = (max(y - r, 0) >> sh) - ty0 + ML_HALO, cy1 = (min(y + r, H - 1) >> sh) - ty0 + ML_HALO;
            const int cx0 = (max(x - r, 0) >> sh) - tx0 + ML_HALO, cx1 = (min(x + r, W - 1) >> sh) - tx0 + ML_HALO;
            unsigned acc = 0;
#pragma unroll
            for (int k = 0; k < NC; ++k) acc |= rows[min(cy0 + k, cy1)];
            uint8_t v = ((acc >> cx0) & ((2u << (cx1 - cx0)) - 1u)) != 0;
            if (and_mask && and_mask[(size_t)y * W + x] == 0) v = 0;
            out[(size_t)y * W + x] = v;
            cnt += v;
            if (tile_h && v) tbits |= 1u << (((p >> lts) / tile_h) * ntx_l + (p & ((1 << lts) - 1)) / tile_w);
        }
    }
    return cnt;
}

template <bool FROM_MASK>
__global__ __launch_bounds__(256) void mask_level_kernel(const MaskLevelKArgs a) {
    // (`a` is never written: a modified by-value argument struct becomes a private copy, and the run-time index a.s[blockIdx.z]
    // into a private copy is scratch memory -- every wavefront then waits for a scratch allocation at dispatch)
    // blockIdx.y = frame of the batch: its own LL plane, coefficients, range and masks
    const int f = blockIdx.y, h = a.h, w = a.w, npix = h * w;
    const int ty0 = (blockIdx.x / a.tiles_x) * ML_T, tx0 = (blockIdx.x % a.tiles_x) * ML_T;
    __shared__ unsigned rows[ML_P];
    float thr = 0.f;
    const float* yl = FROM_MASK ? nullptr : a.yl + (size_t)f * a.n_yl;
    const float* yh = FROM_MASK ? nullptr : a.yh + (size_t)f * 3 * npix;
    const uint8_t* mask0 = FROM_MASK ? a.mask0 + (size_t)f * npix : nullptr;
    if constexpr (!FROM_MASK) {
        float lo, hi;
        if (a.range_keys) {  // the range the previous level's head epilogue left behind (re-armed by this launch's last block)
            lo = key_to_float(a.range_keys[2 * f]);
            hi = key_to_float(a.range_keys[2 * f + 1]);
        } else if (a.mm) {   // batched decode: every block re-reducing its frame's whole LL plane was 40 us per level at 12 frames
            lo = a.mm[2 * f];
            hi = a.mm[2 * f + 1];
        } else {
            lo = INFINITY, hi = -INFINITY;
            // eight independent loads per round (clamped index, the duplicate of the last element changes neither min nor max):
            // a rolled one-load-per-iteration loop would be one memory round trip per 256 values
            for (int i = threadIdx.x; i < a.n_yl; i += 256 * 8) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = yl[min(i + k * 256, a.n_yl - 1)];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    lo = fminf(lo, v[k]);
                    hi = fmaxf(hi, v[k]);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                lo = fminf(lo, __shfl_xor(lo, o));
                hi = fmaxf(hi, __shfl_xor(hi, o));
            }
            __shared__ float slo[4], shi[4];
            if ((threadIdx.x & 63) == 0) {
                slo[threadIdx.x >> 6] = lo;
                shi[threadIdx.x >> 6] = hi;
            }
            __syncthreads();
            lo = fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3]));
            hi = fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));
        }
        thr = (hi - lo) * a.ratio;   // same fp32 expression as mask_threshold_kernel
    }
    // the base tile as one bit row per coarse row (bit cx = cell (cy, cx) of the haloed tile)
    if (threadIdx.x < ML_P) rows[threadIdx.x] = 0;
    __syncthreads();
    for (int c = threadIdx.x; c < ML_P * ML_P; c += 256) {
        const int ry = c / ML_P, rx = c % ML_P;
        const int cy = ty0 - ML_HALO + ry, cx = tx0 - ML_HALO + rx;
        bool v = false;
        if (cy >= 0 && cy < h && cx >= 0 && cx < w) {
            const int p = cy * w + cx;
            if constexpr (FROM_MASK) v = mask0[p] != 0;
            else v = fmaxf(fmaxf(fabsf(yh[p]), fabsf(yh[npix + p])), fabsf(yh[2 * npix + p])) > thr;
        }
        if (v) atomicOr(&rows[ry], 1u << rx);
    }
    __syncthreads();
    // blockIdx.z = spec: the specs of a tile run side by side (each block rebuilds the small base tile; at batch 1 a level
    // has 3 to 30 tiles and the launch is a latency chain, not a throughput problem)
    const wmd_dilate_spec sp = a.s[blockIdx.z];
    const int lth = a.scratch ? a.ls[blockIdx.z].tile_h : 0, ltw = a.scratch ? a.ls[blockIdx.z].tile_w : 0;
    const uint8_t* andm = a.scratch ? a.ls[blockIdx.z].and_mask : nullptr;
    unsigned tbits = 0;
    int cnt = 0;
    switch (sp.up == 2 ? sp.radius + 1 : 2 * sp.radius + 1) {   // coarse rows a window can span (clamped re-reads are harmless)
        case 1: cnt = mask_tile_spec<1>(sp, rows, f, h, w, ty0, tx0, lth, ltw, tbits, andm); break;
        case 2: cnt = mask_tile_spec<2>(sp, rows, f, h, w, ty0, tx0, lth, ltw, tbits, andm); break;
        case 3: cnt = mask_tile_spec<3>(sp, rows, f, h, w, ty0, tx0, lth, ltw, tbits, andm); break;
        case 4:
        case 5: cnt = mask_tile_spec<5>(sp, rows, f, h, w, ty0, tx0, lth, ltw, tbits, andm); break;
        default: cnt = mask_tile_spec<7>(sp, rows, f, h, w, ty0, tx0, lth, ltw, tbits, andm); break;
    }
    if (a.scratch) {
        // ---- work-list form: the block's pixel count and active-tile bits -> accumulators; the last block publishes --------
        __shared__ int s_cnt, s_last;
        __shared__ unsigned s_bits;
        if (threadIdx.x == 0) {
            s_cnt = 0;
            s_bits = 0;
        }
        __syncthreads();
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            cnt += __shfl_xor(cnt, o);
            tbits |= (unsigned)__shfl_xor((int)tbits, o);
        }
        if ((threadIdx.x & 63) == 0) {
            if (cnt) atomicAdd(&s_cnt, cnt);
            if (tbits) atomicOr(&s_bits, tbits);
        }
        __syncthreads();
        const MaskListSpec ls = a.ls[blockIdx.z];
        const unsigned bits = s_bits;
        int32_t* fs = a.scratch + kMlFrame0 + (size_t)f * kMlFrameInts;    // this frame's accumulators
        const int up_ = sp.up;
        const int tiles_xg = ls.tile_w ? (w * up_ + ls.tile_w - 1) / ls.tile_w : 0, tiles_yg = ls.tile_h ? (h * up_ + ls.tile_h - 1) / ls.tile_h : 0;
        // thread 0's two accumulator updates travel together (one round trip instead of two): the pixel count and the reservation
        // of a run of the frame's list; both returned values are awaited before the ticket is drawn (below)
        int old = 0;
        __shared__ int s_base;
        if (threadIdx.x == 0) {
            if (ls.count && s_cnt) old = atomicAdd(&fs[kMlFCnt0 + ls.count - 1], s_cnt);
            if (ls.tile_list && bits) s_base = atomicAdd(&fs[kMlFTile0 + blockIdx.z], __popc(bits));
        }
        if (ls.tile_list && bits) {   // bit k's entry sits at popcount(bits below k) of the reserved run
            __syncthreads();
            const int k = threadIdx.x;
            if (k < 32 && ((bits >> k) & 1u)) {
                const int R = ML_T * up_, ntx_l = R / ls.tile_w;
                const int gty = (ty0 * up_) / ls.tile_h + k / ntx_l, gtx = (tx0 * up_) / ls.tile_w + k % ntx_l;
                ls.tile_list[(size_t)f * tiles_yg * tiles_xg + s_base + __popc(bits & ((1u << k) - 1u))] = (f * tiles_yg + gty) * tiles_xg + gtx;
            }
        }
        if (threadIdx.x == 0) {
            // Ordering without a fence: the accumulators are only ever touched by device-scope atomics (performed at the level
            // that is coherent across the XCDs), so all the ticket needs is that THIS thread's accumulator updates have been
            // performed before it draws -- their returned values are awaited (the empty asm consumes them).  A __threadfence()
            // here is a release at agent scope = a write-back of this XCD's L2, dirty with the masks just written.  The list
            // entries and masks are plain stores for LATER launches and need no ordering inside this one.
            asm volatile("" ::"v"(old) : "memory");
            s_last = atomicAdd(&fs[kMlFTicket], 1) == (int)(gridDim.x * gridDim.z) - 1;
        }
        __syncthreads();
        if (s_last) {          // every other block of the frame has drawn its ticket: the frame's accumulators are final
            const int seq = __hip_atomic_load(&a.scratch[kMlSeq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int32_t* slot = a.ring ? a.ring + (size_t)(seq % max(a.ring_slots, 1)) * a.slot_ints : nullptr;
            if (threadIdx.x < a.ncounts) {
                const int v = atomicExch(&fs[kMlFCnt0 + threadIdx.x], 0);
                if (slot) slot[1 + a.counts_off + f * a.ncounts + threadIdx.x] = v;
            }
            if (threadIdx.x == 0) {   // (uniform index: a per-thread index into the by-value argument block would become scratch memory)
                for (int i = 0; i < a.n; ++i)
                    if (a.ls[i].tile_list) a.ls[i].tile_count[f] = atomicExch(&fs[kMlFTile0 + i], 0);
                if (a.range_keys) {
                    a.range_keys[2 * f] = 0xFFFFFFFFu;
                    a.range_keys[2 * f + 1] = 0u;
                }
                atomicExch(&fs[kMlFTicket], 0);
                if (atomicAdd(&a.scratch[kMlTicket], 1) == a.B - 1) {   // the last frame: the launch is complete
                    atomicExch(&a.scratch[kMlTicket], 0);
                    if (a.advance) {
                        if (slot) slot[0] = seq + 1;
                        atomicExch(&a.scratch[kMlSeq], seq + 1);
                    }
                }
            }
        }
        return;
    }
    if (sp.nnz) {   // one atomic per block: a frame's blocks all add to the same word
        __shared__ int blk_cnt;
        if (threadIdx.x == 0) blk_cnt = 0;
        __syncthreads();
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&blk_cnt, cnt);
        __syncthreads();
        if (threadIdx.x == 0 && blk_cnt) atomicAdd(sp.nnz + (size_t)f * sp.nnz_stride, blk_cnt);
    }
}

struct CompactKArgs {
    wmd_compact_spec s[8];
};

// One workgroup (16 wavefronts) per mask.  A thread takes 16 consecutive flags (one 16-byte load), packs them into a
// bit field and counts them; counts are scanned inside the wavefront by shuffles and across the 16 wavefronts through
// LDS (two barriers per 16 384 pixels); a thread then writes its own pixels in ascending order => raster order.
__global__ __launch_bounds__(1024) void mask_compact_multi_kernel(const CompactKArgs a) {
    wmd_compact_spec sp = a.s[blockIdx.x];
    // blockIdx.y = frame of the batch: masks / lists are [B][npix], the counts [B][n masks]
    sp.mask += (size_t)blockIdx.y * sp.npix;
    sp.coords += (size_t)blockIdx.y * sp.npix;
    sp.nnz += (size_t)blockIdx.y * gridDim.x;
    __shared__ int wave_tot[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec = (reinterpret_cast<uintptr_t>(sp.mask) & 15) == 0;
    int running = 0;   // identical in every thread
    for (int base = 0; base < sp.npix; base += 1024 * 16) {
        const int i0 = base + threadIdx.x * 16;
        unsigned bits = 0;
        if (vec && i0 + 16 <= sp.npix) {
            const uint4 v = *reinterpret_cast<const uint4*>(sp.mask + i0);
            const unsigned wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned t = wd[k];
                t |= t >> 4;
                t |= t >> 2;
                t |= t >> 1;   // bit 0 of every byte = (byte != 0)
                bits |= ((t & 1u) | ((t >> 7) & 2u) | ((t >> 14) & 4u) | ((t >> 21) & 8u)) << (4 * k);
            }
        } else {
            for (int k = 0; k < 16; ++k)
                if (i0 + k < sp.npix && sp.mask[i0 + k] != 0) bits |= 1u << k;
        }
        const int cnt = __popc(bits);
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int before = 0, tot = 0;
#pragma unroll
        for (int wv = 0; wv < 16; ++wv) {
            const int c = wave_tot[wv];
            before += wv < wave ? c : 0;
            tot += c;
        }
        int off = running + before + incl - cnt;
        while (bits) {
            const int k = __ffs(bits) - 1;
            sp.coords[off++] = i0 + k;
            bits &= bits - 1;
        }
        running += tot;
        __syncthreads();   // wave_tot is rewritten by the next round
    }
    if (threadIdx.x == 0) *sp.nnz = running;
}

// ------------------------------------------------------------------------------------------------
struct SparseKArgs {
    wmd_sparse_conv_args g;
    int nci4, ncot, W1;
    int split_waves;   // (tiles x K-slices) that must stay in flight before a wave takes a longer K range
    size_t plane, plane1;
    int nnz_stride;    // ints between the counts of consecutive frames (batched decode: blockIdx.z = frame)
};

// WK wavefronts share one (16-pixel tile, MR out-channel tiles) task and split the input-channel loop between them;
// partial accumulators are combined through LDS by wave 0.  At batch 1 a launch has a few dozen to a few hundred
// active tiles, far fewer than the GPU has SIMDs, so the kernel is a dependent chain of memory round trips, not a
// throughput problem.  The chain is kept at two trips: (1) count + pixel-list entry, (2) everything else at once --
// the input-mask bytes, the gathers of UN K-steps x TAPS (addresses are clamped coordinates and never wait for the
// mask: liveness is applied to the loaded value) and the weight fragments -- then MFMAs, the LDS reduction and the
// store (bias requested up front).
//
// ROWS (3x3 only, W >= 3 and W1 >= 3): the three taps of a kernel row are neighbours in memory, so a lane fetches them
// with ONE 12-byte load of a 3-wide window (base column clamped so that the window stays inside the row and still
// covers the padded tap columns; for the 2x-upsampled source the three taps fall on <= 2 adjacent coarse columns) and
// picks each tap by a 2-bit index -- a third of the gather instructions, which is what a CU runs out of first
// (every scattered 64-lane dword gather costs the texture path some tens of cycles).
struct __attribute__((packed, aligned(4))) Window3 {
    float v[3];
};

template <int MR, int TAPS, bool DUAL, int WK, int UN, bool ROWS>
__global__ __launch_bounds__(64 * WK) void sparse_conv_kernel(const SparseKArgs a0) {
    static_assert(!ROWS || TAPS == 9, "row windows are a 3x3 feature");
    // batched decode: every frame has its own activations, masks, pixel list and count; the weights are shared
    SparseKArgs a = a0;
    {
        const size_t b = blockIdx.z;
        wmd_sparse_conv_args& gm = a.g;
        gm.x1 += b * gm.C1tot * a.plane1;
        if (gm.x2) gm.x2 += b * gm.C2 * a.plane;
        if (gm.in_mask) gm.in_mask += b * a.plane;
        gm.out_coords += b * gm.max_out;
        gm.out_nnz += b * a.nnz_stride;
        gm.y += b * gm.Cout * a.plane;
    }
    const wmd_sparse_conv_args& g = a.g;
    const int lane = threadIdx.x & 63;
    const int wk = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kq = lane >> 4;
    // How the WK wavefronts of a block are used depends on the pixel count, which only the device knows: with few tiles
    // all WK waves share one tile and split K (S = WK, latency-bound chain as short as possible); with many tiles the
    // machine is full anyway and a wave that walks a longer K range for its own tile amortises the fixed trips and skips
    // the LDS reduction (S = 1: WK tiles per block).  The pixel-list entry of every candidate mapping is requested
    // together with the count, not after it; entries past the count are stale, so dead lanes borrow a live pixel.
    constexpr int NS = WK == 8 ? 4 : (WK == 4 ? 3 : 1);   // S in {WK, WK/2, ..., 1}
    int praw_s[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        const int S_ = WK >> q;
        praw_s[q] = g.out_coords[min(((int)blockIdx.x * (WK / S_) + wk / S_) * 16 + j, g.max_out - 1)];
    }
    const int nnz = min(*g.out_nnz, g.max_out);
    const int ntile = (nnz + 15) >> 4;
    int sq = 0;   // S = WK >> sq
#pragma unroll
    for (int q = 1; q < NS; ++q)
        if (ntile * (WK >> q) >= a.split_waves) sq = q;   // merge K slices only while enough wavefronts stay busy
    const int S = WK >> sq;
    const int tpb = WK / S;
    const int ks = wk % S;                               // this wave's K slice
    const int Cin = g.C1 + g.C2;
    const int cot0 = blockIdx.y * MR;
    __shared__ f32x4 red[WK > 1 ? WK : 1][DUAL ? 2 * MR : MR][64];

    float bias_v[MR][4], bias2_v[DUAL ? MR : 1][4];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = (cot0 + m) * 16 + kq * 4 + r;
            bias_v[m][r] = (g.bias && co < g.Cout) ? g.bias[co] : 0.f;
            if (DUAL) bias2_v[m][r] = (g.bias2 && co < g.Cout) ? g.bias2[co] : 0.f;
        }

    // The grid is sized for the machine, not for the pixel capacity: a block walks the task sequence tb = blockIdx.x,
    // blockIdx.x + gridDim.x, ... while tasks remain (device-side count).  With a capacity grid a 10 % dense level spent its
    // time launching and retiring idle blocks (12 frames x 1920 tile slots x out-channel groups: 87 us per call for ~10 us
    // of work).  The first task still uses the pixel-list entries requested together with the count.
    for (int tb = blockIdx.x; tb * tpb * 16 < nnz; tb += gridDim.x) {
    const int tile = tb * tpb + wk / S;
    const bool tile_ok = tile * 16 < nnz;                // wave-uniform; an idle wave still joins the barriers below
    int praw;
    if (tb == (int)blockIdx.x) {
        praw = praw_s[0];
#pragma unroll
        for (int q = 1; q < NS; ++q) praw = sq == q ? praw_s[q] : praw;
    } else {
        praw = g.out_coords[min(tile * 16 + j, g.max_out - 1)];
    }
    const int pidx = tile * 16 + j;
    const bool px_ok = pidx < nnz;
    const int plive = __shfl(praw, lane & 48);           // executed by every lane; lane 16k holds pixel tile*16 (< nnz)
    const int p = tile_ok ? (px_ok ? praw : plive) : 0;
    const int oy = p / g.W, ox = p % g.W;

    // neighbour coordinates through the coordinate padding (layers.py:439-453); the input-mask test only gates the value
    int o1[TAPS], o2[TAPS];
    bool live[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        int gy = oy + (TAPS == 9 ? t / 3 - 1 : 0), gx = ox + (TAPS == 9 ? t % 3 - 1 : 0);
        bool ok = true;
        if (TAPS == 9) {
            ok = pad_coord(gy, g.H, g.pad_mode) && ok;
            ok = pad_coord(gx, g.W, g.pad_mode) && ok;
        }
        gy = min(max(gy, 0), g.H - 1);
        gx = min(max(gx, 0), g.W - 1);
        o2[t] = gy * g.W + gx;
        o1[t] = (gy / g.up1) * a.W1 + gx / g.up1;
        live[t] = ok;
    }
    uint8_t mk[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) mk[t] = g.in_mask ? g.in_mask[o2[t]] : (uint8_t)1;
    // ROWS: window origin per kernel row for both sources + the column index of each tap inside the window
    int w1off[3], w2off[3];
    unsigned ix1 = 0, ix2 = 0;
    if constexpr (ROWS) {
        const int cx0 = o2[0] % g.W, cx1 = o2[1] % g.W, cx2 = o2[2] % g.W;   // padded + clamped tap columns
        const int base2 = min(min(cx0, min(cx1, cx2)), g.W - 3);
        const int dx0 = cx0 / g.up1, dx1 = cx1 / g.up1, dx2 = cx2 / g.up1;
        const int base1 = min(min(dx0, min(dx1, dx2)), a.W1 - 3);
        ix2 = (unsigned)(cx0 - base2) | (unsigned)(cx1 - base2) << 2 | (unsigned)(cx2 - base2) << 4;
        ix1 = (unsigned)(dx0 - base1) | (unsigned)(dx1 - base1) << 2 | (unsigned)(dx2 - base1) << 4;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int cy = o2[3 * r] / g.W;
            w2off[r] = cy * g.W + base2;
            w1off[r] = (cy / g.up1) * a.W1 + base1;
        }
    }

    f32x4 acc[MR], acc2[DUAL ? MR : 1];
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (DUAL) {
#pragma unroll
        for (int m = 0; m < MR; ++m) acc2[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float* wa[MR];
    const float* wa2[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const int cot = min(cot0 + m, a.ncot - 1);
        wa[m] = g.wp + (size_t)cot * a.nci4 * (TAPS * 64) + lane;
        wa2[m] = DUAL ? g.wp2 + (size_t)cot * a.nci4 * (TAPS * 64) + lane : nullptr;
    }

    const int nci4 = (Cin + 3) / 4;
    for (int cb = ks; cb < (tile_ok ? nci4 : 0); cb += S * UN) {
        float bv[UN][TAPS], bv2[DUAL ? UN : 1][TAPS];
        float av[UN][TAPS][MR], av2[DUAL ? UN : 1][TAPS][DUAL ? MR : 1];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int ci4 = min(cb + u * S, nci4 - 1);           // past the end: recomputed with a zeroed operand
            const bool step_ok = cb + u * S < nci4;
            const int ci = ci4 * 4 + kq;
            const bool from1 = ci < g.C1;
            const bool ch_ok = step_ok && ci < Cin;
            const int c1 = min(ci, g.C1 - 1), c2 = min(max(ci - g.C1, 0), max(g.C2 - 1, 0));
            const float* s1 = g.x1 + (size_t)(g.c1_off + c1) * a.plane1;
            const float* s1b = DUAL ? g.x1 + (size_t)(g.c1_off2 + c1) * a.plane1 : nullptr;
            const float* s2 = g.x2 ? g.x2 + (size_t)c2 * a.plane : s1;
            const bool use1 = from1 || !g.x2;
            if constexpr (ROWS) {
                const unsigned ix = use1 ? ix1 : ix2;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const Window3 wv = *reinterpret_cast<const Window3*>(use1 ? s1 + w1off[r] : s2 + w2off[r]);
                    Window3 wv2;
                    if (DUAL) wv2 = *reinterpret_cast<const Window3*>(s1b + w1off[r]);
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        const unsigned sel = (ix >> (2 * d)) & 3u;
                        bv[u][3 * r + d] = sel == 0 ? wv.v[0] : (sel == 1 ? wv.v[1] : wv.v[2]);
                        if (DUAL) bv2[u][3 * r + d] = sel == 0 ? wv2.v[0] : (sel == 1 ? wv2.v[1] : wv2.v[2]);
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const float* src = use1 ? s1 + o1[t] : s2 + o2[t];   // one load per tap whichever the source
                    bv[u][t] = *src;
                    if (DUAL) bv2[u][t] = s1b[o1[t]];
                }
            }
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    av[u][t][m] = wa[m][(size_t)(ci4 * TAPS + t) * 64];
                    if (DUAL) av2[u][t][m] = wa2[m][(size_t)(ci4 * TAPS + t) * 64];
                }
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const bool on = ch_ok && live[t] && mk[t] != 0 && (from1 || g.x2 != nullptr);
                bv[u][t] = on ? bv[u][t] : 0.f;
                if (DUAL) bv2[u][t] = (on && from1) ? bv2[u][t] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][t][m], bv[u][t], acc[m], 0, 0, 0);
                    if (DUAL) acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av2[u][t][m], bv2[u][t], acc2[m], 0, 0, 0);
                }
    }

    if (WK > 1 && S > 1) {   // S is block-uniform
        if (ks > 0) {
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                red[wk][m][lane] = acc[m];
                if (DUAL) red[wk][MR + m][lane] = acc2[m];
            }
        }
        __syncthreads();
        if (ks == 0) {
            for (int w = 1; w < S; ++w)
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    acc[m] += red[wk + w][m][lane];
                    if (DUAL) acc2[m] += red[wk + w][MR + m][lane];
                }
        }
    }
    if (ks == 0 && tile_ok && px_ok) {
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = (cot0 + m) * 16 + kq * 4 + r;
                if (co < g.Cout) {
                    float v = acc[m][r] + bias_v[m][r];
                    v = g.out_scale * act_apply(v, g.act, g.slope);
                    if (DUAL) {
                        float u = acc2[m][r] + bias2_v[m][r];
                        v = v - g.out_scale * act_apply(u, g.act, g.slope);
                    }
                    g.y[(size_t)co * a.plane + p] = v;
                }
            }
    }
    if (WK > 1 && S > 1) __syncthreads();   // `red` is rewritten by the next task
    }   // task loop
}

}  // namespace wmd

using namespace wmd;

extern "C" int wmd_minmax(const float* x, size_t n, float* out2, void* stream) {
    if (!x || !out2) return fail(WMD_ERR_BAD_ARG, "wmd_minmax: null pointer");
    if (n == 0) return fail(WMD_ERR_BAD_SHAPE, "wmd_minmax: empty input (torch.max raises too)");
    ProfScope prof("minmax_kernel", (double)n, 4.0 * n, (hipStream_t)stream);
    hipLaunchKernelGGL(minmax_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, out2);
    return check_launch("minmax_kernel");
}

extern "C" int wmd_mask_threshold(const float* yh, const float* minmax, float thresh_ratio, uint8_t* mask, int h, int w,
                                  void* stream) {
    if (!yh || !minmax || !mask) return fail(WMD_ERR_BAD_ARG, "wmd_mask_threshold: null pointer");
    if (h <= 0 || w <= 0) return fail(WMD_ERR_BAD_SHAPE, "wmd_mask_threshold: h=%d w=%d", h, w);
    const int npix = h * w;
    ProfScope prof("mask_threshold_kernel", 3.0 * npix, 13.0 * npix, (hipStream_t)stream);
    hipLaunchKernelGGL(mask_threshold_kernel, dim3(std::min((npix + 255) / 256, 1024)), dim3(256), 0, (hipStream_t)stream,
                       yh, minmax, thresh_ratio, mask, npix);
    return check_launch("mask_threshold_kernel");
}

extern "C" int wmd_mask_dilate_multi(const uint8_t* mask, int h, int w, const wmd_dilate_spec* specs, int n, void* stream) {
    return wmd_mask_dilate_multi_b(mask, 1, h, w, specs, n, stream);
}

static int mask_specs_ok(const char* who, const wmd_dilate_spec* specs, int n, int* max_up) {
    *max_up = 1;
    for (int i = 0; i < n; ++i) {
        if (!specs[i].out || (specs[i].up != 1 && specs[i].up != 2) || specs[i].radius < 0 || specs[i].radius > 3)
            return fail(WMD_ERR_BAD_ARG, "%s: spec %d (up=%d radius=%d)", who, i, specs[i].up, specs[i].radius);
        *max_up = std::max(*max_up, specs[i].up);
    }
    return WMD_OK;
}

extern "C" int wmd_mask_dilate_multi_b(const uint8_t* mask, int B, int h, int w, const wmd_dilate_spec* specs, int n, void* stream) {
    if (!mask || !specs) return fail(WMD_ERR_BAD_ARG, "wmd_mask_dilate_multi: null pointer");
    if (B <= 0 || B > 65535 || h <= 0 || w <= 0 || n <= 0 || n > 8)
        return fail(WMD_ERR_BAD_SHAPE, "wmd_mask_dilate_multi: B=%d h=%d w=%d n=%d", B, h, w, n);
    int up;
    if (int st = mask_specs_ok("wmd_mask_dilate_multi", specs, n, &up)) return st;
    MaskLevelKArgs a;
    memset(&a, 0, sizeof(a));
    a.mask0 = mask;
    a.h = h;
    a.w = w;
    a.n = n;
    a.tiles_x = (w + ML_T - 1) / ML_T;
    for (int i = 0; i < n; ++i) a.s[i] = specs[i];
    const double maxpix = (double)h * up * w * up;
    ProfScope prof("mask_dilate_multi_kernel", 25.0 * maxpix * n * B, 2.0 * maxpix * n * B, (hipStream_t)stream);
    hipLaunchKernelGGL(mask_level_kernel<true>, dim3(a.tiles_x * ((h + ML_T - 1) / ML_T), B, n), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("mask_dilate_multi_kernel");
}

extern "C" int wmd_mask_level(const float* yl, size_t n_yl, const float* yh, float thresh_ratio, int h, int w,
                              const wmd_dilate_spec* specs, int n, void* stream) {
    return wmd_mask_level_b(yl, n_yl, yh, thresh_ratio, 1, h, w, specs, n, nullptr, stream);
}

extern "C" int wmd_mask_level_b(const float* yl, size_t n_yl, const float* yh, float thresh_ratio, int B, int h, int w,
                                const wmd_dilate_spec* specs, int n, float* minmax_scratch, void* stream) {
    if (!yl || !yh || !specs) return fail(WMD_ERR_BAD_ARG, "wmd_mask_level: null pointer");
    if (B <= 0 || B > 65535) return fail(WMD_ERR_BAD_SHAPE, "wmd_mask_level: B=%d", B);
    if (n_yl == 0 || n_yl > (size_t)1 << 24) return fail(WMD_ERR_BAD_SHAPE, "wmd_mask_level: n_yl=%zu", n_yl);
    if (h <= 0 || w <= 0 || n <= 0 || n > 8) return fail(WMD_ERR_BAD_SHAPE, "wmd_mask_level: h=%d w=%d n=%d", h, w, n);
    int up;
    if (int st = mask_specs_ok("wmd_mask_level", specs, n, &up)) return st;
    MaskLevelKArgs a;
    memset(&a, 0, sizeof(a));
    if (B > 1 && minmax_scratch) {   // one min/max launch per batch instead of a whole-plane reduction in every block of every frame
        hipLaunchKernelGGL(minmax_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, yl, n_yl, minmax_scratch);
        a.mm = minmax_scratch;
    }
    a.yl = yl;
    a.yh = yh;
    a.ratio = thresh_ratio;
    a.n_yl = (int)n_yl;
    a.h = h;
    a.w = w;
    a.n = n;
    a.tiles_x = (w + ML_T - 1) / ML_T;
    for (int i = 0; i < n; ++i) a.s[i] = specs[i];
    const double maxpix = (double)h * up * w * up;
    ProfScope prof("mask_level_kernel", 25.0 * maxpix * n * B, (2.0 * maxpix * n + 16.0 * h * w) * B, (hipStream_t)stream);
    hipLaunchKernelGGL(mask_level_kernel<false>, dim3(a.tiles_x * ((h + ML_T - 1) / ML_T), B, n), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("mask_level_kernel");
}

extern "C" size_t wmd_mask_level_scratch_ints(int B) { return B > 0 ? (size_t)kMlFrame0 + (size_t)kMlFrameInts * B : 0; }

extern "C" int wmd_mask_level_lists(const wmd_mask_level_args* g, void* stream) {
    if (!g || !g->specs || !g->scratch) return fail(WMD_ERR_BAD_ARG, "wmd_mask_level_lists: null pointer");
    if (g->B <= 0 || g->B > 65535 || g->h <= 0 || g->w <= 0 || g->n <= 0 || g->n > 8)
        return fail(WMD_ERR_BAD_SHAPE, "wmd_mask_level_lists: B=%d h=%d w=%d n=%d", g->B, g->h, g->w, g->n);
    if (!g->mask0 && (!g->yl || !g->yh)) return fail(WMD_ERR_BAD_ARG, "wmd_mask_level_lists: neither a base mask nor yl / yh");
    if (!g->mask0 && (g->n_yl == 0 || g->n_yl > (size_t)1 << 24)) return fail(WMD_ERR_BAD_SHAPE, "wmd_mask_level_lists: n_yl=%zu", g->n_yl);
    if (g->ncounts < 0 || g->ncounts > 8) return fail(WMD_ERR_BAD_ARG, "wmd_mask_level_lists: ncounts=%d", g->ncounts);
    if (g->ncounts > 0 && g->counts &&
        (g->ring_slots <= 0 || g->counts_off < 0 || 1 + g->counts_off + g->B * g->ncounts > g->slot_ints))
        return fail(WMD_ERR_BAD_ARG, "wmd_mask_level_lists: counts do not fit a ring slot (off %d + %d x %d > %d ints)", g->counts_off,
                    g->B, g->ncounts, g->slot_ints - 1);
    MaskLevelKArgs a;
    memset(&a, 0, sizeof(a));
    wmd_dilate_spec plain[8];
    for (int i = 0; i < g->n; ++i) {
        const wmd_level_spec& sp = g->specs[i];
        plain[i] = wmd_dilate_spec{sp.up, sp.radius, sp.out, nullptr, 0};
        if (sp.count < 0 || sp.count > g->ncounts) return fail(WMD_ERR_BAD_ARG, "wmd_mask_level_lists: spec %d count column %d", i, sp.count);
        if (sp.tile_h || sp.tile_w) {
            const int R = ML_T * sp.up;
            if (sp.tile_h <= 0 || sp.tile_w <= 0 || R % sp.tile_h || R % sp.tile_w || (R / sp.tile_h) * (R / sp.tile_w) > 32)
                return fail(WMD_ERR_UNSUPPORTED, "wmd_mask_level_lists: spec %d: %dx%d tiles do not nest in a %dx%d region (<= 32 of them)",
                            i, sp.tile_h, sp.tile_w, R, R);
            if (!sp.tile_list || !sp.tile_count) return fail(WMD_ERR_BAD_ARG, "wmd_mask_level_lists: spec %d asks for a tile list without buffers", i);
        }
        a.ls[i] = MaskListSpec{sp.tile_h, sp.tile_w, sp.tile_list, sp.tile_count, sp.count, sp.and_mask};
    }
    int up;
    if (int st = mask_specs_ok("wmd_mask_level_lists", plain, g->n, &up)) return st;
    for (int i = 0; i < g->n; ++i) a.s[i] = plain[i];
    a.mask0 = g->mask0;
    a.yl = g->yl;
    a.yh = g->yh;
    a.mm = g->minmax;
    a.range_keys = g->range_keys;   // (an injected mask does not read them, but its last block re-arms them all the same)
    a.ratio = g->thresh_ratio;
    a.n_yl = (int)g->n_yl;
    a.h = g->h;
    a.w = g->w;
    a.n = g->n;
    a.tiles_x = (g->w + ML_T - 1) / ML_T;
    a.scratch = g->scratch;
    a.ring = g->ncounts > 0 ? g->counts : nullptr;
    a.ring_slots = g->ring_slots;
    a.slot_ints = g->slot_ints;
    a.counts_off = g->counts_off;
    a.ncounts = g->ncounts;
    a.advance = g->advance;
    a.B = g->B;
    const double maxpix = (double)g->h * up * g->w * up;
    ProfScope prof("mask_level_kernel", 25.0 * maxpix * g->n * g->B, (2.0 * maxpix * g->n + 16.0 * g->h * g->w) * g->B, (hipStream_t)stream);
    const dim3 grid(a.tiles_x * ((g->h + ML_T - 1) / ML_T), g->B, g->n);
    if (g->mask0) {
        hipLaunchKernelGGL(mask_level_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL(mask_level_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    }
    return check_launch("mask_level_kernel");
}

extern "C" int wmd_mask_compact_multi(const wmd_compact_spec* specs, int n, void* stream) {
    return wmd_mask_compact_multi_b(specs, n, 1, stream);
}

extern "C" int wmd_mask_compact_multi_b(const wmd_compact_spec* specs, int n, int B, void* stream) {
    if (!specs) return fail(WMD_ERR_BAD_ARG, "wmd_mask_compact_multi: null pointer");
    if (n <= 0 || n > 8 || B <= 0 || B > 65535) return fail(WMD_ERR_BAD_SHAPE, "wmd_mask_compact_multi: n=%d B=%d", n, B);
    CompactKArgs a;
    double px = 0;
    for (int i = 0; i < n; ++i) {
        if (!specs[i].mask || !specs[i].coords || !specs[i].nnz || specs[i].npix <= 0)
            return fail(WMD_ERR_BAD_ARG, "wmd_mask_compact_multi: spec %d", i);
        a.s[i] = specs[i];
        px += specs[i].npix;
    }
    ProfScope prof("mask_compact_multi_kernel", px * B, 5.0 * px * B, (hipStream_t)stream);
    hipLaunchKernelGGL(mask_compact_multi_kernel, dim3(n, B), dim3(1024), 0, (hipStream_t)stream, a);
    return check_launch("mask_compact_multi_kernel");
}

extern "C" int wmd_sparse_conv(const wmd_sparse_conv_args* g, void* stream) {
    if (!g) return fail(WMD_ERR_BAD_ARG, "wmd_sparse_conv: null args");
    if (!g->x1 || !g->out_coords || !g->out_nnz || !g->wp || !g->y)
        return fail(WMD_ERR_BAD_ARG, "wmd_sparse_conv: null pointer");
    if (g->C2 > 0 && !g->x2) return fail(WMD_ERR_BAD_ARG, "wmd_sparse_conv: C2=%d but x2 is null", g->C2);
    if (g->H <= 0 || g->W <= 0 || g->C1 <= 0 || g->C2 < 0 || g->Cout <= 0 || g->max_out < 0)
        return fail(WMD_ERR_BAD_SHAPE, "wmd_sparse_conv: H=%d W=%d C1=%d C2=%d Cout=%d", g->H, g->W, g->C1, g->C2, g->Cout);
    if (g->ksize != 1 && g->ksize != 3) return fail(WMD_ERR_UNSUPPORTED, "wmd_sparse_conv: ksize=%d", g->ksize);
    if (g->up1 != 1 && g->up1 != 2) return fail(WMD_ERR_BAD_ARG, "wmd_sparse_conv: up1=%d", g->up1);
    if (g->up1 == 2 && ((g->H | g->W) & 1))
        return fail(WMD_ERR_BAD_SHAPE, "wmd_sparse_conv: a 2x-upsampled source needs even H, W (got %dx%d)", g->H, g->W);
    if (g->ksize == 3 && g->pad_mode == WMD_PAD_REFLECT && (g->H < 2 || g->W < 2))
        return fail(WMD_ERR_BAD_SHAPE, "wmd_sparse_conv: reflection padding needs H, W >= 2 (got %dx%d)", g->H, g->W);
    if (g->c1_off < 0 || g->c1_off + g->C1 > g->C1tot) return fail(WMD_ERR_BAD_ARG, "wmd_sparse_conv: channel slice");
    if (g->wp2 && (g->Cout > 16 || g->c1_off2 < 0 || g->c1_off2 + g->C1 > g->C1tot || g->C2 != 0))
        return fail(WMD_ERR_BAD_ARG, "wmd_sparse_conv: dual-head mode needs Cout <= 16, C2 == 0 and a valid slice");
    if (g->pad_mode < 0 || g->pad_mode > 2 || g->act < 0 || g->act > 3) return fail(WMD_ERR_BAD_ARG, "wmd_sparse_conv: enum");
    if (g->max_out == 0) return WMD_OK;
    SparseKArgs a;
    a.g = *g;
    const int Cin = g->C1 + g->C2;
    a.nci4 = ((Cin + 15) / 16) * 4;
    a.ncot = (g->Cout + 15) / 16;
    a.W1 = g->W / g->up1;
    a.plane = (size_t)g->H * g->W;
    a.plane1 = (size_t)(g->H / g->up1) * a.W1;
    const int B = g->B > 1 ? g->B : 1;
    if (B > 65535) return fail(WMD_ERR_BAD_SHAPE, "wmd_sparse_conv: B=%d", B);
    a.nnz_stride = B > 1 ? g->nnz_stride : 0;
    static const int split_waves = [] { const char* e = getenv("WMD_SPARSE_SPLIT_WAVES"); return e ? atoi(e) : 2048; }();
    if (g->split_waves < 0) return fail(WMD_ERR_BAD_ARG, "wmd_sparse_conv: split_waves=%d", g->split_waves);
    a.split_waves = g->split_waves > 0 ? g->split_waves : split_waves;
    hipStream_t s = (hipStream_t)stream;
    const int tiles = (g->max_out + 15) / 16;
    const int taps = g->ksize == 3 ? 9 : 1;
    char pname[96] = "sparse_conv_kernel";
    if (g_prof_on && getenv("WMD_SPARSE_PROF_SHAPES"))   // development: one profile line per layer shape
        snprintf(pname, sizeof(pname), "sparse_conv_kernel C%d+%d->%d k%d %dx%d%s", g->C1, g->C2, g->Cout, g->ksize, g->H, g->W,
                 g->wp2 ? " dual" : "");
    ProfScope prof(pname, 0.0, 0.0, s);
    // MR out-channel tiles per block: as few as keeps the grid near the machine size -- a sparse launch has far fewer
    // pixel tiles than the GPU has SIMDs, so out-channel tiles go to separate blocks (each re-gathers the same few
    // pixels out of L2) until the capacity grid reaches ~512 blocks (measured: tools/sparse_microbench.py); a full-density fine level keeps MR large and
    // gathers once.  UN K-steps per wave are in flight together: the whole K-slice of a wave when registers allow.
    static const int mr_force = [] { const char* e = getenv("WMD_SPARSE_MR"); return e ? atoi(e) : 0; }();
    int MR = 1;
    while (MR < 4 && MR < a.ncot && (long)tiles * B * ((a.ncot + MR - 1) / MR) > 512) MR *= 2;
    if (mr_force == 1 || mr_force == 2 || mr_force == 4) MR = std::min(mr_force, a.ncot >= 4 ? 4 : a.ncot >= 2 ? 2 : 1);
    // grid.x: enough blocks to fill the machine a few times over, never more than the capacity (a block walks further tasks
    // itself); WMD_SPARSE_GRID overrides the total block target (development)
    static const int grid_target = [] { const char* e = getenv("WMD_SPARSE_GRID"); return e ? atoi(e) : 2048; }();
#define WMD_SPARSE_LAUNCH(MR_, TAPS_, DUAL_, WK_, UN_, ROWS_)                                                          \
    hipLaunchKernelGGL((sparse_conv_kernel<MR_, TAPS_, DUAL_, WK_, UN_, ROWS_>),                                       \
                       dim3(std::max(1, std::min(tiles, grid_target / (((a.ncot + MR_ - 1) / MR_) * B))), (a.ncot + MR_ - 1) / MR_, B), \
                       dim3(64 * WK_), 0, s, a)
    static const int rows_off = [] { const char* e = getenv("WMD_SPARSE_ROWS"); return e && atoi(e) == 0; }();
    if (taps == 9 && g->W >= 3 && a.W1 >= 3 && !rows_off) {
        if (g->wp2) WMD_SPARSE_LAUNCH(1, 9, true, 8, 2, true);
        else if (MR == 4) WMD_SPARSE_LAUNCH(4, 9, false, 8, 2, true);
        else if (MR == 2) WMD_SPARSE_LAUNCH(2, 9, false, 8, 2, true);
        else WMD_SPARSE_LAUNCH(1, 9, false, 8, 2, true);
    } else if (taps == 9) {   // maps narrower than a window
        if (g->wp2) WMD_SPARSE_LAUNCH(1, 9, true, 8, 2, false);
        else if (MR == 4) WMD_SPARSE_LAUNCH(4, 9, false, 8, 2, false);
        else if (MR == 2) WMD_SPARSE_LAUNCH(2, 9, false, 8, 2, false);
        else WMD_SPARSE_LAUNCH(1, 9, false, 8, 2, false);
    } else {
        if (g->wp2) WMD_SPARSE_LAUNCH(1, 1, true, 4, 8, false);
        else if (MR == 4) WMD_SPARSE_LAUNCH(4, 1, false, 4, 8, false);
        else if (MR == 2) WMD_SPARSE_LAUNCH(2, 1, false, 4, 8, false);
        else WMD_SPARSE_LAUNCH(1, 1, false, 4, 8, false);
    }
#undef WMD_SPARSE_LAUNCH
    return check_launch("sparse_conv_kernel");
}
