// Shared by the dense-decoder convolution kernels (wmd_conv_fwd.hip: direct implicit GEMM + Winograd on 16x16x4 MFMAs;
// wmd_conv_wino32.hip: Winograd on 32x32x2 MFMAs): kernel argument block, XCD-aware work order, LDS-DMA wrappers.
#pragma once
#include "wmd_internal.h"

namespace wmd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Epilogue activation of the 32x32x2 Winograd kernels, scalar and on register pairs (v_pk_add_f32 / v_pk_mul_f32 where the
// operation is one; the exponentials and the selects stay per element).  One arithmetic for both so that every store path of
// every instantiation (dense / masked / work-list) produces the same bits.  ELU through v_exp_f32: |error| <= 1.2e-7 absolute
// (the cancellation in e^v - 1 near 0 costs RELATIVE accuracy of values that are themselves < 1e-3; the trunk's tolerance is
// relative to the tensor's scale).
template <int ACT>
__device__ __forceinline__ float w32_act(float v, float slope) {
    if constexpr (ACT == WMD_ACT_ELU) return v > 0.f ? v : __builtin_amdgcn_exp2f(v * 1.442695041f) - 1.f;
    else if constexpr (ACT == WMD_ACT_LEAKY) return v > 0.f ? v : v * slope;
    else if constexpr (ACT == WMD_ACT_SIGMOID) return 1.f / (1.f + __builtin_amdgcn_exp2f(v * -1.442695041f));
    else return v;
}
template <int ACT>
__device__ __forceinline__ f32x2 w32_act2(f32x2 v, float slope) {
    if constexpr (ACT == WMD_ACT_ELU) {
        const f32x2 t = v * 1.442695041f;
        const f32x2 e = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} - 1.f;
        return f32x2{v[0] > 0.f ? v[0] : e[0], v[1] > 0.f ? v[1] : e[1]};
    } else if constexpr (ACT == WMD_ACT_LEAKY) {
        const f32x2 t = v * slope;
        return f32x2{v[0] > 0.f ? v[0] : t[0], v[1] > 0.f ? v[1] : t[1]};
    } else if constexpr (ACT == WMD_ACT_SIGMOID) {
        return f32x2{w32_act<ACT>(v[0], slope), w32_act<ACT>(v[1], slope)};
    } else return v;
}

// Column mix of the Winograd F(2x2,3x3) input transform, V = tr B, on register pairs: one v_pk_add_f32 with half selects per
// output pair (the compiler does not fold a shuffle into op_sel for packed fp32 and would spend a v_mov per half).
// a = (t0, t1), b = (t2, t3) of one transformed row:
//   w32_pk_lo -> (t0 - t2, t1 + t2)      w32_pk_hi -> (t2 - t1, t1 - t3)
// and for the rows of an upsampled operand (three source columns, a = (t0, t1)):  w32_pk_up -> (t0 - t1, t1 + t1).
// Every result is the same IEEE operation the scalar form performs (a negated operand, a sum in the other order).
// w32_pk_add / w32_pk_sub: plain pair sums, also as inline asm -- the compiler's post-RA peephole UNPACKS a v_pk_add_f32 that follows
// an MFMA into two v_add_f32 (it assumes they hide in the MFMA's shadow; here they are what the matrix pipe waits for).
// An MFMA must not read a VGPR in the two wait states after a VALU wrote it.  The compiler's hazard recognizer pads its own VALU
// instructions (s_nop) but does not look into inline asm, and it is free to place an asm statement right in front of the MFMA that
// consumes its result (seen: the last chunk of an all-upsampled layer, conv_wino32q_kernel, wrong sums).  The column-mix helpers --
// the ones whose results are MFMA operands -- therefore carry their own s_nop 1 (two cycles against 64 per MFMA).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ f32x2 w32_pk_add(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 w32_pk_sub(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 w32_pk_lo(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 w32_pk_hi(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 w32_pk_up(f32x2 a) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,0]\n\ts_nop 1" : "=v"(r) : "v"(a));
    return r;
}
// Variants for the weight-gradient kernel (conv_wgrad_wino32_kernel), every one with its own s_nop 1 (all are MFMA operands):
//   w32_pk_addn / w32_pk_subn: a + b, a - b          w32_pk_pm: (a.lo + a.hi, a.lo - a.hi)
//   column mixes with the sign of some results flipped (the matching dM operand is used without ITS minus sign: the product is the same
//   bits): w32_pk_hi_f -> (t2 - t1, t3 - t1),  w32_pk_lo_ff -> (t2 - t0, -t1 - t2),  w32_pk_hi_fl -> (t1 - t2, t1 - t3)
__device__ __forceinline__ f32x2 w32_pk_addn(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 w32_pk_subn(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 w32_pk_pm(f32x2 a) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]\n\ts_nop 1" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ f32x2 w32_pk_hi_f(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[1,0]\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 w32_pk_lo_ff(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,1]\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 w32_pk_hi_fl(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#else
__device__ __forceinline__ f32x2 w32_pk_add(f32x2 a, f32x2 b) { return a + b; }
__device__ __forceinline__ f32x2 w32_pk_sub(f32x2 a, f32x2 b) { return a - b; }
__device__ __forceinline__ f32x2 w32_pk_lo(f32x2 a, f32x2 b) { return f32x2{a[0] - b[0], a[1] + b[0]}; }
__device__ __forceinline__ f32x2 w32_pk_hi(f32x2 a, f32x2 b) { return f32x2{b[0] - a[1], a[1] - b[1]}; }
__device__ __forceinline__ f32x2 w32_pk_up(f32x2 a) { return f32x2{a[0] - a[1], a[1] + a[1]}; }
__device__ __forceinline__ f32x2 w32_pk_addn(f32x2 a, f32x2 b) { return a + b; }
__device__ __forceinline__ f32x2 w32_pk_subn(f32x2 a, f32x2 b) { return a - b; }
__device__ __forceinline__ f32x2 w32_pk_pm(f32x2 a) { return f32x2{a[0] + a[1], a[0] - a[1]}; }
__device__ __forceinline__ f32x2 w32_pk_hi_f(f32x2 a, f32x2 b) { return f32x2{b[0] - a[1], b[1] - a[1]}; }
__device__ __forceinline__ f32x2 w32_pk_lo_ff(f32x2 a, f32x2 b) { return f32x2{b[0] - a[0], -a[1] - b[0]}; }
__device__ __forceinline__ f32x2 w32_pk_hi_fl(f32x2 a, f32x2 b) { return f32x2{a[1] - b[0], a[1] - b[1]}; }
#endif

struct ConvKArgs {
    const float* x1;
    const float* x2;
    const float* wp;
    const float* bias;
    float* y;        // final output (ksplit == 1) or split-K partials [ksplit][B,Cout,H,W]
    int B, H, W, H1, W1;
    int C1, C2, Cin, Cout, up1;
    int shift1;      // x1 is read at (y - shift1, x - shift1), zero outside its H1 x W1 extent (dgrad: the
                     // "full" correlation is a zero-padded one over a 1-pixel-extended gradient image)
    int pad_mode, act;
    float slope;
    int tiles_x, tiles_y;
    int nci4;        // padded number of 4-channel K groups in wp
    int ncot;        // number of 16-out-channel tiles in wp
    int nchunks;     // ceil(Cin / CK)
    int ksplit, chunks_per_split;
    // fused wavelet head (FUSE kernels only): second GEMM over the LeakyReLU'd block result
    const float* wp2;   // per side: packed [27 -> 32 rows, CO_T] image
    float* t;           // [B, sides*27, H*W]
    int t_ctot;
    int t_row0;   // first plane of t this launch writes
    // optional multiplicative gate of the final output (data-gradient path): y *= gate_act'(gate), gate laid out like y
    const float* gate;
    int gate_act;
    float gate_slope;
    // block-sparse execution (threshold-gated sparse decoder on the dense kernels): a block whose TH x TW pixel tile holds
    // no pixel of out_mask [B,H,W] returns at once (it tests the tile's mask bytes itself: no tile list, no counter, no
    // extra launch); a padded input position outside in_mask [B,H,W] reads 0 (the mask test follows the coordinate padding,
    // layers.py:439-453) and outputs outside out_mask are written as 0
    const uint8_t* in_mask;
    const uint8_t* out_mask;
    int in_mask_2x2;   // in_mask is constant on aligned 2x2 blocks (wmd_conv_args.in_mask_2x2)
    // out-channel slabs per pixel tile when the grid is 1-D (0: the slab is blockIdx.y -- the fused-head launches)
    int cob;
    // work-list form of the block-sparse execution (wmd_conv_args.out_tiles; LIST instantiations of conv_wino32_kernel): item i
    // of the launch = (tile_list[i / cob], slab i % cob) for i < *tile_count * cob; the K split is chosen on the device
    // (list_ksplit) up to ksmax = gridDim.z slices; ksplit slices > 1 write partial sums to `y` (workspace), one slice the final
    // result to y_final
    const int* tile_list;    // [B][tiles_y * tiles_x]: frame f's active tiles are the first tile_count[f] entries of its segment
    const int* tile_count;   // [B]
    int no_x4;               // development switch (WMD_X4=0): dword pieces everywhere
    int ksmax;
    int list_slots;          // workgroup slots of the machine for this kernel (blocks per CU x CUs): the device's K-split target
    float* y_final;
    // split-K finished inside the convolution (round 6, 32x32x2 kernels): one zero-at-rest counter per (pixel tile, out-channel
    // slab); null = the partial planes are summed by the second-stage kernels (conv_splitk_reduce[_list]_kernel)
    int* tickets;
    int xcd_slab;   // 2-D grid launches: slab = linear workgroup id % slabs (one slab's weights per XCD) instead of slab = blockIdx.y
    // conv_wino32_kernel (round 5): output rows transposed through LDS into whole 128-byte lines (WMD_W32_COALESCE=0: 16-byte
    // pieces of 64 different lines per store instruction)
    int st_coalesce;
    // development builds only (-DWMD_STAMPS, tools/probes/stamps_probe.py): per-block cycle stamps [blocks][12]; dbg_mode bit 0:
    // no output stores, bit 1: no activation
    unsigned long long* dbg;
    int dbg_mode;
};

#ifdef WMD_STAMPS
#define WMD_STAMP(k) do { asm volatile("" ::: "memory"); stamp_[k] = __builtin_readcyclecounter(); asm volatile("" ::: "memory"); } while (0)
// ... after the scalar loads / LDS traffic issued so far have returned and the scalar `dep` has been computed
#define WMD_STAMP_AFTER(k, dep) do { asm volatile("s_waitcnt lgkmcnt(0)" :: "s"(dep) : "memory"); stamp_[k] = __builtin_readcyclecounter(); asm volatile("" ::: "memory"); } while (0)
#else
#define WMD_STAMP(k) do { } while (0)
#define WMD_STAMP_AFTER(k, dep) do { } while (0)
#endif

// ---- split-K finished inside the convolution (round 6) --------------------------------------------------------------------
// Every K-slice block of a (pixel tile, slab) stores its partial tile WRITE-THROUGH at agent scope (`sc1`: the eight XCD L2s are
// not coherent with each other; a plain store may sit dirty in the writer's L2), waits for the stores' acknowledgement, then one
// thread draws a ticket from the item's counter (relaxed agent-scope atomic: executed at the memory side).  The block that draws
// the LAST ticket re-reads all slices of the tile with agent-scope loads in slice order s = 0 .. ks-1 -- the same order, the same
// bits as conv_splitk_reduce_kernel, whichever block happens to be last --, applies bias + activation + the out-mask select,
// stores the final tile and re-arms the counter (zero at rest: no fill, no second launch).  No fence: a release fence would
// write back the whole L2 of the XCD (58 us in round 4's mask kernel); the data path is coherent by construction instead.
// (raw buffer instructions with the sc1 cache-policy bit -- aux = 16 on gfx950: compiler-visible, so its hazard recognizer and
//  wait-count pass cover them.  An inline-asm global_store_dwordx4 did not survive: the compiler reused the store's data registers
//  for the next address one instruction later, and one s_nop -- the documented wait state -- was not enough on this part.)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kAuxAgent = 16;   // sc1
__device__ __forceinline__ __amdgpu_buffer_rsrc_t agent_rsrc(const float* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void st16_agent(__amdgpu_buffer_rsrc_t r, unsigned off, float4 v) {
    const f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), r, off, 0, kAuxAgent);
}
__device__ __forceinline__ void st4_agent(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, kAuxAgent);
}
__device__ __forceinline__ f32x4 ld16_agent(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, kAuxAgent));
}
__device__ __forceinline__ float ld4_agent(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, kAuxAgent));
}

// c0: first out channel of the block's slab (NCH of them), (y0, x0): its TH x TW pixel tile of frame b; flag: one int of LDS.
// Called by EVERY thread of the block after its partial stores (contains barriers).
template <int NTHREADS, int TH, int TW, int NCH, bool MASKED>
__device__ __forceinline__ void splitk_ticket_finish(const ConvKArgs& a, int* flag, int tick, int ks_n, int b, int y0, int x0, int c0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's write-through stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) *flag = atomicAdd(a.tickets + tick, 1);
    __syncthreads();
    if (*flag != ks_n - 1) return;                      // (uniform)
    if (threadIdx.x == 0) atomicExch(a.tickets + tick, 0);
    const int H = a.H, W = a.W;
    const size_t plane = (size_t)H * W, fr = (size_t)a.Cout * plane;     // floats of one frame of one slice (< 2^29: host check)
    const bool vec = (W & 3) == 0;
    constexpr int PPR = (TW + 3) / 4, NP = NCH * TH * PPR;
    for (int e = threadIdx.x; e < NP; e += NTHREADS) {
        const int ch = e / (TH * PPR), rem = e - ch * (TH * PPR), r = rem / PPR, c4 = rem - r * PPR;
        const int cg = c0 + ch, oy = y0 + r, ox = x0 + 4 * c4;
        if (cg >= a.Cout || oy >= H || ox >= W || 4 * c4 >= TW) continue;
        const size_t off = (size_t)cg * plane + (size_t)oy * W + ox;   // inside the frame
        const unsigned ob = (unsigned)(off * 4);
        const int nv = min(min(4, W - ox), TW - 4 * c4);
        const bool v16 = vec && nv == 4;
        auto slice = [&](int s) {       // this piece of slice s (agent-scope loads: another XCD wrote it)
            const __amdgpu_buffer_rsrc_t rs = agent_rsrc(a.y + ((size_t)s * a.B + b) * fr, fr * 4);
            if (v16) return ld16_agent(rs, ob);
            f32x4 q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nv) q[k] = ld4_agent(rs, ob + 4 * k);
            return q;
        };
        f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 4 <= ks_n; s += 4) {                 // four slices in flight, summed in slice order
            const f32x4 p0 = slice(s), p1 = slice(s + 1), p2 = slice(s + 2), p3 = slice(s + 3);
            acc4 = (((acc4 + p0) + p1) + p2) + p3;
        }
        for (; s < ks_n; ++s) acc4 = acc4 + slice(s);
        float v[4] = {acc4[0], acc4[1], acc4[2], acc4[3]};
        const float bv = a.bias ? a.bias[cg] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // the kernels' own epilogue arithmetic (w32_act): same bits as their unsplit stores
            const float z = v[k] + bv;
            v[k] = a.act == WMD_ACT_ELU ? w32_act<WMD_ACT_ELU>(z, a.slope) : a.act == WMD_ACT_LEAKY ? w32_act<WMD_ACT_LEAKY>(z, a.slope)
                 : a.act == WMD_ACT_SIGMOID ? w32_act<WMD_ACT_SIGMOID>(z, a.slope) : z;
        }
        if (MASKED && a.out_mask) {
            const uint8_t* mp = a.out_mask + (size_t)b * plane + (size_t)oy * W;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = mp[min(ox + k, W - 1)] ? v[k] : 0.f;
        }
        float* dst = a.y_final + (size_t)b * fr + off;
        if (v16) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nv) dst[k] = v[k];
        }
    }
}

// Device-chosen split of the input-channel reduction of a work-list launch: the same function in the convolution and in its
// second pass.  Fills the machine (`slots` workgroup slots) when few tiles are active, at least two chunks per slice.
__host__ __device__ inline void list_ksplit(int n_items, int nchunks, int ksmax, int slots, int& ks, int& cps) {
    int want = n_items > 0 ? slots / n_items : 1;
    want = want < 1 ? 1 : (want > ksmax ? ksmax : want);
    cps = (nchunks + want - 1) / want;
    const int floor_cps = nchunks < 2 ? nchunks : 2;
    if (cps < floor_cps) cps = floor_cps;
    ks = (nchunks + cps - 1) / cps;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Workgroups are dealt to the 8 XCDs round-robin (blockIdx.x % 8) and every XCD has its own L2.  Spatially adjacent
// tiles share halo rows and 128-byte lines, so give each XCD one contiguous run of the tile sequence instead of every
// eighth tile (speed only: correctness never depends on the placement).
__device__ __forceinline__ int xcd_contiguous(int bid, int n) {
    const int q = n >> 3, r = n & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// LDS-DMA wrappers.  They are deliberately NOT templates: inside a dependent context hipcc's host pass rejects
// the 16-byte form (a gfx950 feature check against the host target) and silently drops the kernel's host stub.
__device__ __forceinline__ void lds_dma4(__amdgpu_buffer_rsrc_t r, lds_ptr_t dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 4, voff, soff, 0, 0);
}
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, lds_ptr_t dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voff, soff, 0, 0);
}

// conv_wino32_kernel (wmd_conv_wino32.hip): tile geometry, shared with the configuration table in wmd_conv_fwd.hip.
// Block = WN tile groups (32 Winograd tiles = 128 pixels each) x 2 position halves, one 32-out-channel slab.
template <int TH, int TW, int WN, int CK>
struct W32Tile {
    static constexpr int NW = WN * 2, NT = NW * 64;
    static constexpr int TXB = TW / 2, TYB = TH / 2, NTILES = TXB * TYB;
    // Patch row strides are whole 16-byte groups (round 4): a tile whose patch columns all lie inside the image stages a chunk
    // with 16-byte LDS-DMA pieces -- a quarter of the staging instructions (gfx950 accepts 4-byte-aligned global addresses for
    // them: tools/probes/dma16_probe.hip) --, the columns beyond TW + 2 / TW/2 + 2 are padding nobody reads
    // (32- / 64- / 16-wide tiles only: the 40-wide ones lose more to the longer rows -- 6x40 measured +12 % -- than their rare
    //  interior tiles could gain)
    static constexpr bool X4OK = TW % 16 == 0;
    static constexpr int PH = TH + 2, PWS = X4OK ? ((TW + 2 + 3) / 4) * 4 : TW + 2;          // full-resolution patch: rows x row stride (even: 8-byte reads)
    static constexpr int PSF = PH * PWS;
    static constexpr int PHL = TH / 2 + 2, PWL = X4OK ? ((TW / 2 + 2 + 3) / 4) * 4 : TW / 2 + 2, PSL = PHL * PWL;   // low-resolution patch of the upsampled operand
    static constexpr int GF = PWS / 4, GL = PWL / 4;       // 16-byte groups per patch row (X4OK)
    static constexpr int NPOSF = (PSF + NT - 1) / NT, NPOSL = (PSL + NT - 1) / NT;
    static constexpr int RUN = CK * 256;       // one 16-out-channel run of a chunk: (CK/4) K-steps x 16 positions x 64 floats
    static constexpr int RUN_LDS = RUN + 16;   // the two runs a 32-lane read group touches fall on disjoint bank halves
    static constexpr int A_FLOATS = 2 * RUN_LDS;
    static constexpr int B_FLOATS = X4OK ? ((CK * PSF + 255) / 256) * 256 : ((CK * PSF + 63) / 64) * 64;   // whole LDS-DMA runs (256 dwords of 16-byte pieces / 64 dwords; tail = padding)
    static constexpr int NAV = (2 * CK * 64 + NT - 1) / NT;   // 16-byte weight pieces per thread and chunk
    static constexpr int BUF_FLOATS = B_FLOATS + A_FLOATS;
    static constexpr int KW = CK / 2;          // 2-channel K-steps per chunk
    static constexpr int XCH_FLOATS = WN * 2 * 32 * 64;   // the two halves of a group trade 32 partial outputs per lane
    static constexpr int LDS_FLOATS = 2 * BUF_FLOATS > XCH_FLOATS ? 2 * BUF_FLOATS : XCH_FLOATS;
    static constexpr int TAB_FLOATS = ((PH + PWS + PHL + PWL + 3) / 4) * 4;   // folded row / column offsets of the patch
    static_assert(TH % 2 == 0 && TW % 8 == 0, "whole 2x2 tiles; a lane's four consecutive tiles stay in one tile row");
    static_assert(WN * 32 >= NTILES, "more tiles than MFMA rows");
    static_assert(CK % 4 == 0, "chunks are whole 4-channel weight fragments");
    static_assert((LDS_FLOATS + TAB_FLOATS) * 4 <= 160 * 1024, "LDS");
};

// conv_wino32q_kernel (wmd_conv_wino32q.hip, round 5): one tile group (32 Winograd tiles = 128 pixels) x four quarter-position
// waves, one 32-out-channel slab; same patch / weight staging geometry as W32Tile<TH, TW, 1, CK> with 256 threads.
template <int TH, int TW, int CK>
struct W32QTile {
    static constexpr int NW = 4, NT = 256;
    static constexpr int TXB = TW / 2, TYB = TH / 2, NTILES = TXB * TYB;
    static constexpr bool X4OK = TW % 16 == 0;
    static constexpr int PH = TH + 2, PWS = X4OK ? ((TW + 2 + 3) / 4) * 4 : TW + 2;
    static constexpr int PSF = PH * PWS;
    static constexpr int PHL = TH / 2 + 2, PWL = X4OK ? ((TW / 2 + 2 + 3) / 4) * 4 : TW / 2 + 2, PSL = PHL * PWL;
    static constexpr int GF = PWS / 4, GL = PWL / 4;
    static constexpr int RUN = CK * 256;
    static constexpr int RUN_LDS = RUN + 16;
    static constexpr int A_FLOATS = 2 * RUN_LDS;
    static constexpr int B_FLOATS = ((CK * PSF + 255) / 256) * 256;   // whole LDS-DMA runs of either piece size (tail = padding)
    static constexpr int NAV = (2 * CK * 64 + NT - 1) / NT;
    static constexpr int BUF_FLOATS = B_FLOATS + A_FLOATS;
    static constexpr int KW = CK / 2;
    static constexpr int XCH_FLOATS = 4 * 32 * 64;   // the four quarters trade 32 partial outputs per lane
    static constexpr int LDS_FLOATS = 2 * BUF_FLOATS > XCH_FLOATS ? 2 * BUF_FLOATS : XCH_FLOATS;
    static constexpr int TAB_FLOATS = ((PH + PWS + PHL + PWL + 3) / 4) * 4;
    static_assert(TH % 2 == 0 && TW % 8 == 0, "whole 2x2 tiles; a lane's four consecutive tiles stay in one tile row");
    static_assert(NTILES <= 32 && NTILES % 2 == 0, "one tile group");
    static_assert(CK % 4 == 0 && PWS % 2 == 0 && PSF % 2 == 0, "8-byte patch reads");
    static_assert((LDS_FLOATS + TAB_FLOATS) * 4 <= 53 * 1024, "three blocks per CU");
};
template <int TH, int TW, int CK>
void launch_wino32q(const ConvKArgs& a, dim3 grid, hipStream_t s);   // explicit instantiations: wmd_conv_wino32q_table.inc
// quarter-position tile shapes with a LIST instantiation: the 8 x 16 list tile of wmd_mask_level_lists
constexpr bool wino32q_has_list(int TH, int TW, int CK) { return CK == 8 && TH == 8 && TW == 16; }

// conv_wino32_kernel's flattened-staging instantiation needs every chunk inside one source tensor, one full-resolution
// geometry; an input mask must live on that geometry too (same-size x1, or an upsampled x1 under a 2x2-constant mask, whose
// low-resolution patch reads the mask at (2y, 2x)); everything else runs the GENERIC instantiation
inline bool wino32_pure(const ConvKArgs& a, int CK) {
    const bool mask_ok = !a.in_mask || (a.up1 == 2 ? a.in_mask_2x2 != 0 : (a.shift1 == 0 && a.H1 == a.H && a.W1 == a.W));
    return mask_ok && (a.Cin % CK) == 0 && (a.C2 == 0 || (a.C1 % CK) == 0) && (a.C2 == 0 || a.shift1 == 0);
}

template <int TH, int TW, int WN, int CK>
void launch_wino32(const ConvKArgs& a, dim3 grid, hipStream_t s);   // explicit instantiations: wmd_conv_wino32_table.inc
// Position `ti` of the concatenated per-frame tile lists -> the tile id; total = the number of listed tiles (sum of the counts).
__device__ __forceinline__ int list_total(const int* __restrict__ tile_count, int B) {
    int n = 0;
    for (int f = 0; f < B; ++f) n += tile_count[f];
    return n;
}
__device__ __forceinline__ int list_entry(const int* __restrict__ tile_list, const int* __restrict__ tile_count, int B, int cap, int ti) {
    int f = 0, base = 0;
    for (; f < B - 1; ++f) {
        const int c = tile_count[f];
        if (ti < base + c) break;
        base += c;
    }
    return tile_list[(size_t)f * cap + (ti - base)];
}

// tile shapes with a LIST instantiation (work-list form): small tiles that nest in wmd_mask_level_lists' regions
constexpr bool wino32_has_list(int TH, int TW, int WN, int CK) {
    return CK == 8 && ((TH == 8 && TW == 16 && WN == 1) || (TH == 16 && TW == 16 && WN == 2));
}

// weight-gradient kernels (wmd_conv_bwd.hip, wmd_conv_wgrad32.hip)
struct WgradKArgs {
    const float* x1;
    const float* x2;
    const float* dz;
    float* partial;  // [nsplit][Cout*Cin*taps + Cout]  (weights, then the bias partial sums)
    int B, H, W, H1, W1, C1, C2, Cin, Cout, up1, pad_mode;
    int tiles_x, tiles_y, ntiles;  // pixel tiles per image / total (B * tiles_x * tiles_y)
    int nsplit;
    int want_bias;
};

// conv_wgrad_wino32_kernel (wmd_conv_wgrad32.hip): block = WCO x WCI slabs of 32 out / 32 in channels, two position halves each
template <int TH, int TW, int WCO, int WCI>
void launch_wgrad_wino32(const WgradKArgs& a, dim3 grid, hipStream_t s);

}  // namespace wmd
