// kernels.hip -- gfx950 (CDNA4 / MI355X) kernels of the RealSR x4 hot path.
//
//   conv3x3_mfma<NT,UPS>   the 351 3x3 convolutions of models-DF2K*/x4.param as MFMA implicit GEMM
//                          (v_mfma_f32_32x32x16_f16, fp32 accumulate), with bias / LeakyReLU /
//                          residual-axpy epilogues fused (the graph's Eltwise, BinaryOp, Concat,
//                          Split and Interp layers never exist as kernels).
//   preproc_tiles          realsr_preproc{,_tta}.comp equivalent, writes the network input planes
//   postproc_tiles         realsr_postproc{,_tta}.comp equivalent, writes the uint8 HWC image
//   *_shader               the same arithmetic in the shaders' own memory layout (parity tests)
//
// Written for gfx950 only: 64-wide wavefronts, 160 KiB LDS, MFMA 32x32x16 f16.
#include "kernels.h"

namespace rsr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// =============================================================================================
// conv3x3 (stride 1, zero pad 1) as implicit GEMM on the matrix cores
// =============================================================================================
//
//   D[cout][pixel] += W[cout][k] * X[k][pixel],   k = (tap, cin)
//
// MFMA operand roles: A = weights (32 cout x 16 k), B = activations (16 k x 32 pixels), so that the
// accumulator of a lane holds 16 output channels of ONE pixel (col = lane&31 = pixel,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = cout): the epilogue packs 4 consecutive channels into
// one 8-byte fp16 store into the [H][W][32] plane.
//
// Workgroup = 4 waves, output block 16 rows x 32 cols.  Wave w owns rows 4w..4w+3 (R = 4 M-tiles of
// 32 pixels) for all NT*32 output channels.  K is walked one 32-channel plane ("chunk") at a time:
//   stage   the (16+2) x (32+2) pixel patch of the plane (64 B/pixel) and the chunk's weight image
//           [9 taps][NT*32 cout][32 cin] into LDS,
//   compute for dx in 0..2, for 16-channel half cb: read the 6 patch rows a wave needs ONCE and
//           reuse each row fragment for the three dy taps (row r+dy of tap dy == row of tap 0):
//           6 X + 3*NT W ds_read_b128 feed 12*NT MFMAs.
// LDS layout: 64-B rows (pixel or weight row), the four 16-B slots of row i XOR-swizzled with
// (i>>2)&3 -- ds_read_b128 lane groups (16 lanes) then touch 16 distinct 16-B bank groups for any
// run of consecutive rows, i.e. conflict-free for every tap shift.
//
// Nearest-x2 upsampling (ncnn Interp type 1, x4.param:996,998) is folded into the staging address:
// UPS kernels stage patch pixel (y,x) from source pixel (y>>1, x>>1).

constexpr int kThreads = 256;
constexpr int kPatchItems = kPatchPx * 4; // 16-byte items in the patch (2448)
constexpr int kPatchIters = (kPatchItems + kThreads - 1) / kThreads; // 10

__device__ __forceinline__ const char* plane_ptr(const PlaneSrc& s, int slot, int plane)
{
    return static_cast<const char*>(s.base) + (long long)slot * s.slot_stride + (long long)plane * s.plane_stride;
}

// Fused epilogue (bias, LeakyReLU, residual axpy stages, fp16/fp32/planar stores) for one output block.
// acc[rr] = rows 4*wrow + rr of the block, output channels nt*32 .. nt*32+31.
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[4], const f32x4 (&bq)[4], int nt, int slot, int y0, int x0,
                                              int H, int W, int wrow, int l32, int hi)
{
    const int x = x0 + l32;
    {
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
        {
            const int y = y0 + wrow * 4 + rr;
            if (y >= H || x >= W) continue;
            const long long pix = (long long)y * W + x;
#pragma unroll
            for (int q = 0; q < 4; q++)
            {
                const int c0 = q * 8 + hi * 4; // channel within the 32-plane
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++)
                {
                    float t = acc[rr][q * 4 + e] + bq[q][e];
                    if (a.lrelu) t = t > 0.f ? t : t * 0.2f;
                    v[e] = t;
                }
                if (a.res1_kind == 2)
                {
                    const f32x4 r = *reinterpret_cast<const f32x4*>(plane_ptr(a.res1, slot, nt) + (pix * 32 + c0) * 4);
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = v[e] * a.s1 + r[e];
                }
                else if (a.res1_kind == 1)
                {
                    const half4 r = *reinterpret_cast<const half4*>(plane_ptr(a.res1, slot, nt) + (pix * 32 + c0) * 2);
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = v[e] * a.s1 + (float)r[e];
                }
                if (a.res2_kind == 2)
                {
                    const f32x4 r = *reinterpret_cast<const f32x4*>(plane_ptr(a.res2, slot, nt) + (pix * 32 + c0) * 4);
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = v[e] * a.s2 + r[e];
                }
                else if (a.res2_kind == 1)
                {
                    const half4 r = *reinterpret_cast<const half4*>(plane_ptr(a.res2, slot, nt) + (pix * 32 + c0) * 2);
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = v[e] * a.s2 + (float)r[e];
                }
                if (a.out16.base)
                {
                    half4 o;
#pragma unroll
                    for (int e = 0; e < 4; e++) o[e] = (_Float16)v[e];
                    *reinterpret_cast<half4*>(const_cast<char*>(plane_ptr(a.out16, slot, nt)) + (pix * 32 + c0) * 2) = o;
                }
                if (a.out32a.base)
                    *reinterpret_cast<f32x4*>(const_cast<char*>(plane_ptr(a.out32a, slot, nt)) + (pix * 32 + c0) * 4) = v;
                if (a.out32b.base)
                    *reinterpret_cast<f32x4*>(const_cast<char*>(plane_ptr(a.out32b, slot, nt)) + (pix * 32 + c0) * 4) = v;
                if (a.out_planar3 && nt == 0 && q == 0 && hi == 0)
                {
                    _Float16* o = reinterpret_cast<_Float16*>(static_cast<char*>(a.out_planar3) + (long long)slot * a.planar3_slot_stride);
                    const long long hw = (long long)H * W;
                    o[pix] = (_Float16)v[0];
                    o[hw + pix] = (_Float16)v[1];
                    o[2 * hw + pix] = (_Float16)v[2];
                }
            }
        }
    }
}

// Specialised, branch-free epilogues of conv3x3_pipe (channel-major accumulators: lane (p = lane&31, hi) holds the
// channels q*8 + hi*4 .. +3, q = 0..3, of pixel p of each of the wave's 4 rows).  The generic conv_epilogue above
// decides everything per element with wave-uniform branches on ConvArgs fields; measured on MI355X that scalar
// control flow made the epilogue the largest single cost of a launch (~4 us per work item, ~16 us for the
// residual variant).  Here the variant is a template parameter, interior blocks skip all bounds tests, and the
// residual planes are fetched for the whole block before the first use.
//   EPI 1: v = act(acc + b)                       -> fp16 plane          (dense-block convs 1-4, up/HR convs)
//   EPI 2: v = (acc + b)*s1 + r1 [, v = v*s2 + r2] -> fp16 plane          (dense-block conv 5, trunk conv; fp16 residuals)
template <int EPI, bool CHECK>
__device__ __forceinline__ void conv_epilogue_t(const ConvArgs& a, f32x16 (&acc)[4], const f32x4 (&bq)[4], int nt, int slot, int y0,
                                                int x0, int H, int W, int wrow, int l32, int hi)
{
    const int x = x0 + l32;
    const bool xin = !CHECK || x < W;
    char* op = const_cast<char*>(plane_ptr(a.out16, slot, nt)) + hi * 8;
    const float slope = a.lrelu ? 0.2f : 1.f;
    long long off[4];
#pragma unroll
    for (int rr = 0; rr < 4; rr++) off[rr] = ((long long)(y0 + wrow * 4 + rr) * W + x) * 64;
    half4 r1[4][4];
    const bool has2 = (EPI == 2) && a.res2_kind == 1;
    const char* r2p = nullptr;
    if (EPI == 2)
    {
        const char* r1p = plane_ptr(a.res1, slot, nt) + hi * 8;
        r2p = has2 ? plane_ptr(a.res2, slot, nt) + hi * 8 : r1p;
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
        {
            const bool ok = xin && (!CHECK || y0 + wrow * 4 + rr < H);
#pragma unroll
            for (int q = 0; q < 4; q++) r1[rr][q] = ok ? *reinterpret_cast<const half4*>(r1p + off[rr] + q * 16) : half4{0, 0, 0, 0};
        }
    }
#pragma unroll
    for (int rr = 0; rr < 4; rr++)
    {
        const bool ok = xin && (!CHECK || y0 + wrow * 4 + rr < H);
        half4 r2[4]; // second residual (every third dense block only): fetched per row, 8 registers instead of 32
        if (has2)
        {
#pragma unroll
            for (int q = 0; q < 4; q++) r2[q] = ok ? *reinterpret_cast<const half4*>(r2p + off[rr] + q * 16) : half4{0, 0, 0, 0};
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            half4 o;
#pragma unroll
            for (int e = 0; e < 4; e++)
            {
                float v = acc[rr][q * 4 + e] + bq[q][e];
                if (EPI == 1) v = fmaxf(v, v * slope); // LeakyReLU(0.2) or identity, branch-free
                if (EPI == 2)
                {
                    v = v * a.s1 + (float)r1[rr][q][e];
                    if (has2) v = v * a.s2 + (float)r2[q][e];
                }
                o[e] = (_Float16)v;
            }
            if (ok) *reinterpret_cast<half4*>(op + off[rr] + q * 16) = o;
        }
    }
}

// EPI 1 through a private LDS transpose (conv3x3_ring).  s_memtime stamps showed the direct epilogue above costing
// ~5,500 cycles per block although it is only ~250 instructions: its 16 stores per wave each scatter 64 x 8 B over 32
// different 64-byte segments, i.e. 2,048 partial-line write requests per block leave the CU.  Here the finished
// rows go through LDS so that every store instruction writes 16 whole pixels (1 KiB contiguous, 4 lanes x 16 B
// per pixel): 512 requests per block, 8 store instructions per wave.
// Scratch: the two patch rows of the stage just consumed that ONLY this wave reads (rows 4*wrow+2, +3 of the
// 18-row patch; neighbours' halos stop at +1 / start at +4) -- free until the loaders refill the slot, which needs
// the barrier this wave has not reached yet.  DS operations of one wave execute in order, so no barrier is needed.
template <bool CHECK>
__device__ __forceinline__ void conv_epilogue_lds(const ConvArgs& a, f32x16 (&acc)[4], const f32x4 (&bq)[4], int slot, int y0, int x0,
                                                  int H, int W, int wrow, int lane, char* patch)
{
    const int l32 = lane & 31, hi = lane >> 5;
    char* scratch = patch + (4 * wrow + 2) * (kPatchW * 64);
    char* wr = scratch + l32 * 64 + hi * 8;
    const int wswz = (l32 >> 1) & 3;
    // read side: 16 pixels per instruction, lane -> (pixel = lane>>2, 16-byte piece = lane&3)
    const int rpx = lane >> 2, rpiece = lane & 3;
    const float slope = a.lrelu ? 0.2f : 1.f;
    char* op = const_cast<char*>(plane_ptr(a.out16, slot, 0));
#pragma unroll
    for (int half = 0; half < 2; half++)
    {
#pragma unroll
        for (int rl = 0; rl < 2; rl++)
#pragma unroll
            for (int q = 0; q < 4; q++)
            {
                half4 o;
#pragma unroll
                for (int e = 0; e < 4; e++)
                {
                    const float v = acc[half * 2 + rl][q * 4 + e] + bq[q][e];
                    o[e] = (_Float16)fmaxf(v, v * slope);
                }
                *reinterpret_cast<half4*>(wr + rl * (32 * 64) + ((q ^ wswz) << 4)) = o;
            }
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const int px = k * 16 + rpx, rl = px >> 5, col = px & 31;
            const uint4 v = *reinterpret_cast<const uint4*>(scratch + px * 64 + ((rpiece ^ ((col >> 1) & 3)) << 4));
            const int y = y0 + wrow * 4 + half * 2 + rl, x = x0 + col;
            if (!CHECK || (y < H && x < W))
            {
                *reinterpret_cast<uint4*>(op + ((long long)y * W + x) * 64 + rpiece * 16) = v;
            }
        }
    }
}

// The same LDS transpose for workgroups with two MFMA waves per block row (NT = 2: waves (wrow, ntw=0/1) share the
// patch rows, so each owns ONE private row: 4*wrow+2+ntw, 2176 B >= one 32-pixel x 32-channel row) and for the
// residual epilogue EPI 2: the value staged in LDS is t = fp16(s1*(acc+b)); in the transposed domain every lane owns
// 8 consecutive channels of a pixel, loads the residual(s) as one coalesced 16-byte piece and stores
// fp16((t + r1)*s2 + r2).  (t is rounded once more than in the direct epilogue: <= 2^-11 * |0.2*x5|, a tenth of the
// output's own fp16 ulp; the parity tests bound it.)  Residual planes are fetched before the LDS round trips.
typedef __attribute__((address_space(3))) int lds_int_t;
template <int EPI, bool CHECK>
__device__ __forceinline__ void conv_epilogue_lds_row(const ConvArgs& a, f32x16 (&acc)[4], const f32x4 (&bq)[4], int nt, int slot,
                                                      int y0, int x0, int H, int W, int wrow, int lane, char* patch,
                                                      char* my_flag, char* partner_flag, int epoch, bool skip_r1,
                                                      unsigned long long* ts = nullptr)
{
#ifdef RSR_EXP_OVLTRACE
#define RSR_ETS(K) if (ts) ts[K] = __builtin_amdgcn_s_memtime();
#else
#define RSR_ETS(K)
#endif
    RSR_ETS(0)
    // The scratch row is private with respect to the other row groups, but the partner wave (same rows, other 32
    // output channels) reads it as MFMA operand: tell the partner that this wave's operand reads of the item are
    // done (DS operations of a wave complete in order, so the flag write follows them), then wait for the partner.
    // The flags are accessed through explicit LDS pointers: a volatile access through a generic pointer becomes a
    // FLAT load behind "s_waitcnt vmcnt(0)", i.e. it waits for the previous block's stores to be acknowledged.
    if (partner_flag)
    {
        if (lane == 0) *(volatile lds_int_t*)my_flag = epoch;
        while (*(volatile lds_int_t*)partner_flag < epoch) __builtin_amdgcn_s_sleep(1);
    }
    RSR_ETS(1)
    // The lane-derived addresses below are loop invariants of the kernel; hipcc hoists them out of the block loop and
    // then SPILLS them (the MFMA loop needs every register; one scratch reload costs ~2.5 us per block here) --
    // recompute them from an opaque copy of the lane id.
    asm volatile("" : "+v"(lane));
    const int l32 = lane & 31, hi = lane >> 5;
    char* scratch = patch + (4 * wrow + 2 + nt) * (kPatchW * 64);
    char* wr = scratch + l32 * 64 + hi * 8;
    const int wswz = (l32 >> 1) & 3;
    const int rpx = lane >> 2, rpiece = lane & 3; // read side: 16 pixels per instruction
    const float slope = a.lrelu ? 0.2f : 1.f;
    char* op = const_cast<char*>(plane_ptr(a.out16, slot, nt));
    const bool has2 = (EPI == 2) && a.res2_kind == 1;
    const char* r1p = nullptr;
    const char* r2p = nullptr;
    half8 zero8;
#pragma unroll
    for (int e = 0; e < 8; e++) zero8[e] = (_Float16)0.f;
    if (EPI == 2)
    {
        r1p = plane_ptr(a.res1, slot, nt);
        r2p = has2 ? plane_ptr(a.res2, slot, nt) : r1p;
    }
    // residual fetch runs one row ahead of its use (16 + 8 registers instead of 64)
    auto fetch = [&](const char* p, int rr, int k) -> half8 {
        const int y = y0 + wrow * 4 + rr, x = x0 + k * 16 + rpx;
        return (!CHECK || (y < H && x < W)) ? *reinterpret_cast<const half8*>(p + ((long long)y * W + x) * 64 + rpiece * 16) : zero8;
    };
    // skip_r1: the first residual is already inside the accumulator (identity tap, see conv3x3_pipe) -- no fetch, no add
    const bool has1 = (EPI == 2) && !skip_r1;
    half8 r1n[2] = {zero8, zero8};
    if (has1)
    {
        r1n[0] = fetch(r1p, 0, 0);
        r1n[1] = fetch(r1p, 0, 1);
    }
#pragma unroll
    for (int rr = 0; rr < 4; rr++)
    {
        half8 r1c[2] = {r1n[0], r1n[1]};
        half8 r2[2] = {zero8, zero8};
        if (EPI == 2)
        {
            if (has1 && rr < 3)
            {
                r1n[0] = fetch(r1p, rr + 1, 0);
                r1n[1] = fetch(r1p, rr + 1, 1);
            }
            if (has2)
            {
                r2[0] = fetch(r2p, rr, 0);
                r2[1] = fetch(r2p, rr, 1);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            half4 o;
#pragma unroll
            for (int e = 0; e < 4; e++)
            {
                float v = acc[rr][q * 4 + e] + bq[q][e];
                if (EPI == 1) v = fmaxf(v, v * slope);
                else v = v * a.s1;
                o[e] = (_Float16)v;
            }
            *reinterpret_cast<half4*>(wr + ((q ^ wswz) << 4)) = o;
        }
#pragma unroll
        for (int k = 0; k < 2; k++)
        {
            const int col = k * 16 + rpx;
            half8 t = *reinterpret_cast<const half8*>(scratch + col * 64 + ((rpiece ^ ((col >> 1) & 3)) << 4));
            if (EPI == 2 && (has1 || has2))
            {
#pragma unroll
                for (int e = 0; e < 8; e++)
                {
                    float v = (float)t[e];
                    if (has1) v += (float)r1c[k][e];
                    if (has2) v = v * a.s2 + (float)r2[k][e];
                    t[e] = (_Float16)v;
                }
            }
            const int y = y0 + wrow * 4 + rr, x = x0 + col;
            if (!CHECK || (y < H && x < W))
            {
                *reinterpret_cast<half8*>(op + ((long long)y * W + x) * 64 + rpiece * 16) = t;
            }
        }
        RSR_ETS(2 + rr)
    }
#undef RSR_ETS
}

// DMA = true: both LDS images are filled by LDS-DMA (global_load_lds_dwordx4: per-lane global source,
// wave-uniform LDS base + lane*16 destination), so staging costs no VGPRs and no ds_write; the XOR
// swizzle is applied on the SOURCE address (LDS item i receives logical slot (i&3)^((i>>4)&3) of
// pixel i>>2) and out-of-image pixels read a 16-byte zero page.  DMA = false: the same image built
// through registers (global_load_dwordx4 + ds_write_b128).
template <int NT, bool UPS, bool DMA>
__global__ __launch_bounds__(kThreads, 2) void conv3x3_mfma(const ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WROWS = 9 * NT * 32;          // weight rows per chunk
    constexpr int WITEMS = WROWS * 4;           // 16-byte items
    constexpr int WITERS = (WITEMS + kThreads - 1) / kThreads;
    constexpr int WOFF = kPatchLds;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, l32 = lane & 31, hi = lane >> 5;

    // XCD-aware work mapping: workgroup b runs on XCD b%8 (observed dispatch rule); give every XCD a
    // contiguous run of work items (= neighbouring blocks of the same tiles) so halo rows and the
    // weight images are re-read from that XCD's own L2.
    const int per = (a.nitems + 7) >> 3;
    const int item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (item >= a.nitems) return;
    const WorkItem it = a.items[item];
    const int slot = it.slot, y0 = it.y0, x0 = it.x0;
    const int H = it.H, W = it.W;        // output dims
    const int Wi = UPS ? (W >> 1) : W;   // input row pitch in pixels

    // ---- staging addresses (chunk invariant) ----
    int srcoff[kPatchIters]; // byte offset inside a plane, -1 = zero fill
    int dstoff[kPatchIters];
#pragma unroll
    for (int i = 0; i < kPatchIters; i++)
    {
        const int j = tid + i * kThreads;
        const int px = j >> 2, sl = j & 3;
        const int r = px / kPatchW, c = px - r * kPatchW;
        const int gy = y0 - 1 + r, gx = x0 - 1 + c;
        const bool ok = (j < kPatchItems) && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const int sy = UPS ? (gy >> 1) : gy, sx = UPS ? (gx >> 1) : gx;
        if (DMA)
        { // LDS item j is written linearly; it must receive logical slot sl ^ swz
            srcoff[i] = ok ? ((sy * Wi + sx) * 64 + ((sl ^ ((px >> 2) & 3)) << 4)) : -1;
            dstoff[i] = 0;
        }
        else
        {
            srcoff[i] = ok ? ((sy * Wi + sx) * 64 + sl * 16) : -1;
            dstoff[i] = (j < kPatchItems) ? (px * 64 + ((sl ^ ((px >> 2) & 3)) << 4)) : -1;
        }
    }

    f32x16 acc[4][NT];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[r][n][e] = 0.f;

    const int nplanes = a.n0 + a.n1;
    const char* wbase = static_cast<const char*>(a.wpk);

    for (int ck = 0; ck < nplanes; ck++)
    {
        const char* plane = (ck < a.n0) ? plane_ptr(a.src0, slot, ck) : plane_ptr(a.src1, slot, ck - a.n0);
        const char* wsrc = wbase + (long long)ck * (WROWS * 64);

        if (DMA)
        {
            __syncthreads(); // previous chunk's LDS reads are done
            const char* zp = static_cast<const char*>(a.zeros);
#pragma unroll
            for (int i = 0; i < kPatchIters; i++)
            {
                const char* src = srcoff[i] >= 0 ? plane + srcoff[i] : zp;
                if ((i * kThreads + wave * 64) * 16 < kPatchLds) // wave-uniform: the 40th 1-KiB piece does not exist
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(smem + (i * kThreads + wave * 64) * 16), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < WITERS; i++)
            {
                const int jw = i * kThreads + wave * 64; // wave-uniform first item
                if (jw < WITEMS)                         // WITEMS is a multiple of 64
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (jw + lane) * 16),
                                                     (__attribute__((address_space(3))) void*)(smem + WOFF + jw * 16), 16, 0, 0);
            }
            __syncthreads(); // hipcc drains vmcnt(0) (pending LDS-DMA) before the barrier
        }
        else
        {
            // global -> registers
            uint4 pv[kPatchIters];
#pragma unroll
            for (int i = 0; i < kPatchIters; i++)
            {
                pv[i] = make_uint4(0u, 0u, 0u, 0u);
                if (srcoff[i] >= 0) pv[i] = *reinterpret_cast<const uint4*>(plane + srcoff[i]);
            }
            uint4 wv[WITERS];
#pragma unroll
            for (int i = 0; i < WITERS; i++)
            {
                const int j = tid + i * kThreads;
                wv[i] = make_uint4(0u, 0u, 0u, 0u);
                if (j < WITEMS) wv[i] = *reinterpret_cast<const uint4*>(wsrc + j * 16);
            }
            __syncthreads(); // previous chunk's LDS reads are done
#pragma unroll
            for (int i = 0; i < kPatchIters; i++)
                if (dstoff[i] >= 0) *reinterpret_cast<uint4*>(smem + dstoff[i]) = pv[i];
#pragma unroll
            for (int i = 0; i < WITERS; i++)
            {
                const int j = tid + i * kThreads;
                if (j < WITEMS) *reinterpret_cast<uint4*>(smem + WOFF + j * 16) = wv[i];
            }
            __syncthreads();
        }

        // ---- MFMA ----
#pragma unroll
        for (int dx = 0; dx < 3; dx++)
        {
#pragma unroll
            for (int cb = 0; cb < 2; cb++)
            {
                const int ks = cb * 2 + hi;
                half8 X[6];
#pragma unroll
                for (int rr = 0; rr < 6; rr++)
                {
                    const int p = (wave * 4 + rr) * kPatchW + l32 + dx;
                    X[rr] = *reinterpret_cast<const half8*>(smem + p * 64 + ((ks ^ ((p >> 2) & 3)) << 4));
                }
#pragma unroll
                for (int dy = 0; dy < 3; dy++)
                {
#pragma unroll
                    for (int nt = 0; nt < NT; nt++)
                    {
                        const int row = (dy * 3 + dx) * (NT * 32) + nt * 32 + l32;
                        const half8 Wf = *reinterpret_cast<const half8*>(smem + WOFF + row * 64 + ((ks ^ ((row >> 2) & 3)) << 4));
#pragma unroll
                        for (int rr = 0; rr < 4; rr++)
                            acc[rr][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf, X[rr + dy], acc[rr][nt], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue ----
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
    {
        f32x4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; q++) bq[q] = *reinterpret_cast<const f32x4*>(a.bias + nt * 32 + q * 8 + hi * 4);
        f32x16 accn[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) accn[rr] = acc[rr][nt];
        conv_epilogue(a, accn, bq, nt, slot, y0, x0, H, W, wave, l32, hi);
    }
}

// ---------------------------------------------------------------------------------------------
// conv3x3_pipe: the same arithmetic as conv3x3_mfma, restructured for latency:
//   * persistent: gridDim = #CUs; each workgroup walks a strided list of work items, so the chunk
//     stream never drains between items (a 64->32 conv is only 2 chunks long),
//   * wave-specialised: waves 0-3 only ds_read + MFMA, waves 4-7 only issue LDS-DMA (they share the
//     SIMDs pairwise, so DMA issue slots never sit between two MFMAs of the same wave),
//   * double-buffered LDS: the loaders fill stage s+1 while stage s is being multiplied; one
//     barrier per stage,
//   * MFMA operand fragments are software-pipelined one (dx, 16-channel) step ahead.
// ---------------------------------------------------------------------------------------------
template <int NT, bool UPS, int EPI>
__global__ __launch_bounds__((4 * NT + 4) * 64, NT + 1) void conv3x3_pipe(const ConvArgs a)
{
    // waves [0, 4*NT): compute -- wave w owns rows 4*(w&3).. of the block and output channels 32*(w>>2)..;
    // waves [4*NT, 4*NT+4): loaders.  NT=1: 8 waves (2 per SIMD), NT=2: 12 waves (3 per SIMD, <= 168 VGPRs).
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NCW = 4 * NT;
    constexpr int WROWS = 9 * NT * 32;
    constexpr int WITEMS = WROWS * 4;
    constexpr int WPASS = (WITEMS + 255) / 256;
    constexpr int STAGE = kPatchLds + WROWS * 64;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nplanes = a.n0 + a.n1;

    // work items of this workgroup: XCD x gets the contiguous range [x*per, (x+1)*per); its nj workgroups stride it
    const int per = (a.nitems + 7) >> 3;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, nj = gridDim.x >> 3;
    const int first = xcd * per + j;
    const int end = min((xcd + 1) * per, a.nitems);
    const int nmine = first < end ? (end - first + nj - 1) / nj : 0;
    if (nmine == 0) return;
    const int S = nmine * nplanes;

    if (wave >= NCW)
    {
        // ================= loader waves =================
        const int lw = wave - NCW, ltid = lw * 64 + lane;
                const char* wbase = static_cast<const char*>(a.wpk);
        int t = 0;
        WorkItem nxt = a.items[first];
        for (int r = 0; r < nmine; r++)
        {
            // the descriptor of item r was fetched one item ago; item r+1's is requested now and lands while
            // this item's stages stream (no dependent global load sits in front of a DMA issue)
            const WorkItem it = nxt;
            if (r + 1 < nmine) nxt = a.items[first + (r + 1) * nj];
            const int H = it.H, W = it.W, Wi = UPS ? (W >> 1) : W;
            // byte offset of every patch item of this lane from (plane pixel 0 - kGuard); 0 = the plane's zero guard
            unsigned srcoff[kPatchIters];
#pragma unroll
            for (int i = 0; i < kPatchIters; i++)
            {
                const int jj = ltid + i * 256;
                const int px = jj >> 2, sl = jj & 3;
                const int rr = px / kPatchW, cc = px - rr * kPatchW;
                const int gy = it.y0 - 1 + rr, gx = it.x0 - 1 + cc;
                const bool ok = (jj < kPatchItems) && gy >= 0 && gy < H && gx >= 0 && gx < W;
                const int sy = UPS ? (gy >> 1) : gy, sx = UPS ? (gx >> 1) : gx;
                // LDS item jj receives logical 16-B slot sl ^ ((column >> 2) & 3) of its pixel (column swizzle)
                srcoff[i] = ok ? unsigned(kGuard + (sy * Wi + sx) * 64 + ((sl ^ ((cc >> 2) & 3)) << 4)) : 0u;
            }
            for (int ck = 0; ck < nplanes; ck++, t++)
            {
                if (t > 0)
                {
                    // Raw barrier (not __syncthreads): drain OUR LDS-DMA explicitly (hipcc does not carry it across
                    // this loop's back edge), then B_{t-1} publishes stage t-1 to the compute waves and tells us
                    // that buffer t&1 is free.
                    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                }
                char* buf = smem + (t & 1) * STAGE + lw * 1024;
                const char* gbase = ((ck < a.n0) ? plane_ptr(a.src0, it.slot, ck) : plane_ptr(a.src1, it.slot, ck - a.n0)) - kGuard;
                const char* wsrc = wbase + (long long)ck * (WROWS * 64) + (lw * 64 + lane) * 16;
                if (!(a.dbg & 1))
                {
                    {
#pragma unroll
                    for (int i = 0; i < kPatchIters - 1; i++)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + srcoff[i]),
                                                             (__attribute__((address_space(3))) void*)(buf + i * 4096), 16, 0, 0);
                    if (lw < 3) // the patch region is 39 one-KiB pieces: the last pass has only three
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + srcoff[kPatchIters - 1]),
                                                             (__attribute__((address_space(3))) void*)(buf + (kPatchIters - 1) * 4096), 16, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < WPASS; i++)
                        if (i * 256 + lw * 64 < WITEMS)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + i * 4096),
                                                             (__attribute__((address_space(3))) void*)(buf + kPatchLds + i * 4096), 16, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); // barrier B_{S-1}
        return;
    }

    // ================= compute waves =================
    const int l32 = lane & 31, hi = lane >> 5;
    const int wrow = wave & 3, ntw = wave >> 2;
    // LDS byte offsets (stage relative).  X: patch pixel (row, col) lives at (row*34 + col)*64 with its four 16-B
    // slots XOR-swizzled by (col>>2)&3 (a 16-lane ds_read_b128 group covers 16 consecutive columns of one row:
    // col&3 and (col>>2)&3 enumerate all 16 bank groups -> conflict-free for every tap shift).  The swizzle depends
    // on the column only, so one VGPR per dx + immediate row offsets address all 18 fragments; cb flips bit 5.
    // W: row = tap*NT*32 + nt*32 + l32 -> its swizzle ((row>>2)&3) depends on l32 only.
    int xcol[3];
#pragma unroll
    for (int dx = 0; dx < 3; dx++)
    {
        const int c = l32 + dx;
        xcol[dx] = wrow * 4 * kPatchW * 64 + c * 64 + ((hi ^ ((c >> 2) & 3)) << 4);
    }
    const int woff = kPatchLds + (ntw * 32 + l32) * 64 + ((hi ^ ((l32 >> 2) & 3)) << 4);

    f32x16 acc[4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[r][e] = 0.f;

    // bias: copied to LDS once per launch (behind the two stage buffers); every epilogue re-reads its 16 values
    // with 4 ds_read_b128 instead of pinning 16 VGPRs or paying a global-load latency per item
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * STAGE);
    if (tid < NT * 32) bias_lds[tid] = a.bias[tid];
    if (tid < 16) reinterpret_cast<int*>(smem + 2 * STAGE + 256)[tid] = 0; // epilogue hand-shake flags (NT = 2)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the ds_write has landed before this wave's first barrier

    // identity tap (see below): the first residual of an EPI-2 conv is this conv's own input planes 0,1
    const bool idt = (EPI == 2 && NT == 2 && a.res1_in_acc && !(a.dbg & 64));
    int r = 0, ck = 0;
    WorkItem it = a.items[first];
    WorkItem nxt = a.items[first + (nmine > 1 ? nj : 0)];
    for (int s = 0; s < S; s++)
    {
        // Raw s_barrier, NOT __syncthreads(): the fence in __syncthreads() makes the wave wait (vmcnt(0)) for the
        // acknowledgement of the previous item's epilogue stores -- measured ~5 us of dead time per work item.
        // All this barrier must order is LDS: the DMA writes of stage s (drained by the loaders before they
        // arrive) against our ds_reads below, and our reads of stage s-1 (consumed by MFMAs already) against
        // the loaders' next fill.
        unsigned long long t_arrive = 0;
        const bool tracing = a.trace && blockIdx.x == 0 && wave == 0;
        if (tracing) t_arrive = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // barrier B_s: stage s is in LDS
        if (tracing && s < 512 && lane == 0)
        {
            a.trace[2 * s] = t_arrive;
            a.trace[2 * s + 1] = __builtin_amdgcn_s_memtime();
        }
        const char* buf = smem + (s & 1) * STAGE;
        if (!(a.dbg & 2))
        {
        half8 X0[6], X1[6], W0[3], W1[3];
#define RSR_LOAD_STEP(X, Wf, T)                                                                                     \
    {                                                                                                                \
        constexpr int dx_ = (T) >> 1, cb_ = (T)&1;                                                                   \
        _Pragma("unroll") for (int dy = 0; dy < 3; dy++)                                                             \
        {                                                                                                            \
            X[dy] = *reinterpret_cast<const half8*>(buf + ((xcol[dx_] ^ (cb_ << 5)) + dy * (kPatchW * 64)));         \
            Wf[dy] = *reinterpret_cast<const half8*>(buf + ((woff ^ (cb_ << 5)) + (dy * 3 + dx_) * (NT * 32 * 64))); \
        }                                                                                                            \
        _Pragma("unroll") for (int rr = 3; rr < 6; rr++)                                                             \
            X[rr] = *reinterpret_cast<const half8*>(buf + ((xcol[dx_] ^ (cb_ << 5)) + rr * (kPatchW * 64)));         \
    }
#define RSR_MFMA_STEP(X, Wf)                                                                                         \
    {                                                                                                                \
        _Pragma("unroll") for (int dy = 0; dy < 3; dy++) _Pragma("unroll") for (int rr = 0; rr < 4; rr++)            \
            acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[dy], X[rr + dy], acc[rr], 0, 0, 0);                  \
    }
        RSR_LOAD_STEP(X0, W0, 0)
        RSR_LOAD_STEP(X1, W1, 1)
        RSR_MFMA_STEP(X0, W0)
        RSR_LOAD_STEP(X0, W0, 2)
        RSR_MFMA_STEP(X1, W1)
        RSR_LOAD_STEP(X1, W1, 3)
        RSR_MFMA_STEP(X0, W0)
        RSR_LOAD_STEP(X0, W0, 4)
        RSR_MFMA_STEP(X1, W1)
        RSR_LOAD_STEP(X1, W1, 5)
        RSR_MFMA_STEP(X0, W0)
        RSR_MFMA_STEP(X1, W1)
#undef RSR_LOAD_STEP
#undef RSR_MFMA_STEP
        // Pin the issue order (hipcc otherwise sinks every ds_read next to its MFMA behind lgkmcnt(0)).  Fragments of
        // steps 0,1 first; then, per step t, the loads of step t+2 are issued INTO THE REGISTERS STEP t FREES, in the
        // order it frees them: after the dy=0 MFMAs {X[0], W[0]} are dead, after dy=1 {X[1], W[1]}, after dy=2 the
        // rest -- two fragment sets stay live, not three (at 3 waves/SIMD the NT=2 kernel has 168 VGPRs).
        // Masks: 0x8 MFMA, 0x100 DS read.
        __builtin_amdgcn_sched_group_barrier(0x100, 18, 0);
#pragma unroll
        for (int t = 0; t < 4; t++)
        {
            __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x8, 24, 0);
        // Identity tap: out = s1*(conv + b) + x with x = input planes 0,1 of this very conv (RDB conv5, fp16 trunk).  While
        // plane ntw (= this wave's 32 output channels) is in LDS the residual is added on the matrix pipe as one more
        // "tap": A = (1/s1) * I (exact in fp16 for s1 = 0.2), B = the centre-tap pixel fragments -> acc += 5*x, exactly,
        // so the epilogue has no residual to fetch (its global loads sat, un-hidden, in front of every row's stores).
        if (idt && ck == ntw)
        {
            half8 idf[2], xc[2][4];
#pragma unroll
            for (int cb = 0; cb < 2; cb++)
            {
#pragma unroll
                for (int jj = 0; jj < 8; jj++) idf[cb][jj] = (cb * 16 + hi * 8 + jj == l32) ? (_Float16)a.res1_coef : (_Float16)0.f;
#pragma unroll
                for (int rr = 0; rr < 4; rr++)
                    xc[cb][rr] = *reinterpret_cast<const half8*>(buf + ((xcol[1] ^ (cb << 5)) + (rr + 1) * (kPatchW * 64)));
            }
#pragma unroll
            for (int cb = 0; cb < 2; cb++)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(idf[cb], xc[cb][rr], acc[rr], 0, 0, 0);
        }
        }
        if (++ck == nplanes)
        {
            if (!(a.dbg & 4))
            {
                f32x4 bq[4];
#pragma unroll
                for (int q = 0; q < 4; q++) bq[q] = *reinterpret_cast<const f32x4*>(bias_lds + ntw * 32 + q * 8 + hi * 4);
                if (EPI == 0) conv_epilogue(a, acc, bq, ntw, it.slot, it.y0, it.x0, it.H, it.W, wrow, l32, hi);
                else if ((EPI == 1 || EPI == 2) && !(a.dbg & 64))
                { // coalesced stores through a private LDS row of the patch that was just consumed
                    char* flags = smem + 2 * STAGE + 256;
                    char* mine = flags + wave * 4;
                    char* partner = (NT == 2) ? flags + (wave ^ 4) * 4 : nullptr;
                    char* pb = const_cast<char*>(buf);
                    unsigned long long* ets = (tracing && s < 512) ? a.trace + 1024 + 8 * s : nullptr;
                    if (it.y0 + kBlkH <= it.H && it.x0 + kBlkW <= it.W)
                        conv_epilogue_lds_row<(EPI == 2 ? 2 : 1), false>(a, acc, bq, ntw, it.slot, it.y0, it.x0, it.H, it.W, wrow, lane, pb, mine, partner, r + 1, idt, ets);
                    else
                        conv_epilogue_lds_row<(EPI == 2 ? 2 : 1), true>(a, acc, bq, ntw, it.slot, it.y0, it.x0, it.H, it.W, wrow, lane, pb, mine, partner, r + 1, idt, ets);
                }
                else if (it.y0 + kBlkH <= it.H && it.x0 + kBlkW <= it.W)
                    conv_epilogue_t<EPI, false>(a, acc, bq, ntw, it.slot, it.y0, it.x0, it.H, it.W, wrow, l32, hi);
                else
                    conv_epilogue_t<EPI, true>(a, acc, bq, ntw, it.slot, it.y0, it.x0, it.H, it.W, wrow, l32, hi);
            }
            ck = 0;
            r++;
            it = nxt;
            if (r + 1 < nmine) nxt = a.items[first + (r + 1) * nj];
#pragma unroll
            for (int rr = 0; rr < 4; rr++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[rr][e] = 0.f;
        }
    }
}

constexpr size_t pipe_lds(int NT) { return 2 * (size_t(kPatchLds) + size_t(9 * NT * 32 * 64)) + 256 + 64; } // two stages + bias + flags

template <int NT, bool UPS, int EPI>
static void launch_conv_pipe_t(const ConvArgs& a, int ncu, hipStream_t st)
{
    int grid = ncu & ~7; // multiple of 8 (XCD mapping)
    const int per = (a.nitems + 7) / 8;
    if (per * 8 < grid) grid = per * 8;
    hipLaunchKernelGGL((conv3x3_pipe<NT, UPS, EPI>), dim3(grid), dim3((4 * NT + 4) * 64), pipe_lds(NT), st, a);
}

void launch_conv_pipe(const ConvArgs& a, int nt, int ncu, hipStream_t st)
{
    if (a.nitems <= 0) return;
    const bool ups = a.lvl_out != a.lvl_in;
    // epilogue variant: fp16 plane out only -> 1 (no residual) / 2 (fp16 residuals); anything else -> generic
    int epi = 0;
    if (a.out16.base && !a.out32a.base && !a.out32b.base && !a.out_planar3 && !(a.dbg & 16))
    {
        if (a.res1_kind == 0 && a.res2_kind == 0) epi = 1;
        else if (a.res1_kind == 1 && (a.res2_kind == 0 || a.res2_kind == 1)) epi = 2;
    }
#define RSR_PIPE(NT_, UPS_)                                                   \
    do                                                                        \
    {                                                                         \
        if (epi == 1) launch_conv_pipe_t<NT_, UPS_, 1>(a, ncu, st);           \
        else if (epi == 2) launch_conv_pipe_t<NT_, UPS_, 2>(a, ncu, st);      \
        else launch_conv_pipe_t<NT_, UPS_, 0>(a, ncu, st);                    \
    } while (0)
    if (nt == 1)
    {
        if (ups) RSR_PIPE(1, true);
        else RSR_PIPE(1, false);
    }
    else
    {
        if (ups) RSR_PIPE(2, true);
        else RSR_PIPE(2, false);
    }
#undef RSR_PIPE
}

// ---------------------------------------------------------------------------------------------
// conv3x3_ring: conv3x3_pipe with the activation patches in a THREE-deep LDS ring.
// Measured on conv3x3_pipe (C2 frame): LDS-DMA fill alone ~65 ms, MFMA stream alone ~62 ms, together ~88 ms and
// 113 ms with the epilogues -- one barrier per stage couples a loader with a single stage of lookahead to the MFMA
// waves, so every epilogue (and every late DMA) stalls the other side.  Here the loaders run TWO patch stages
// ahead with counted waits ("s_waitcnt vmcnt(P)": everything but my newest patch stage has landed); the weight
// images stay double-buffered (they are small and always L2-hot).  Work-item descriptors are staged in LDS once per
// launch so that no other vector load disturbs the loaders' counts.
//   LDS: 3 x 39 KiB patches + 2 x 18 KiB weights (NT = 1) + bias + descriptors = ~157 KiB.  With 64 output
//   channels the weight images are 36 KiB each and the ring no longer fits: launch_conv_ring() then returns false
//   and the engine uses conv3x3_pipe.  (Reading the weight fragments straight from L2 instead -- tried -- costs
//   4x the vector-memory traffic per workgroup and slowed the MFMA waves by 25-40 %.)
// ---------------------------------------------------------------------------------------------
constexpr int kRingDepth = 3;

template <int NT, bool UPS, int EPI>
__global__ __launch_bounds__((4 * NT + 4) * 64, NT + 1) void conv3x3_ring(const ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NCW = 4 * NT;
    constexpr int NTHREADS = (4 * NT + 4) * 64;
    constexpr int WROWS = 9 * NT * 32;
    constexpr int WBYTES = WROWS * 64;
    constexpr int WITEMS = WROWS * 4;
    constexpr int WPASS = (WITEMS + 255) / 256;
    constexpr int WOFF = kRingDepth * kPatchLds;      // two weight images behind the patch ring
    constexpr int BIAS_OFF = WOFF + 2 * WBYTES;
    constexpr int ITEMS_OFF = BIAS_OFF + 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nplanes = a.n0 + a.n1;

    const int per = (a.nitems + 7) >> 3;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, nj = gridDim.x >> 3;
    const int first = xcd * per + j;
    const int end = min((xcd + 1) * per, a.nitems);
    const int nmine = first < end ? (end - first + nj - 1) / nj : 0;
    if (nmine == 0) return;
    const int S = nmine * nplanes;

    // stage this workgroup's work-item descriptors and the bias in LDS
    {
        uint4* dst = reinterpret_cast<uint4*>(smem + ITEMS_OFF);
        const uint4* src = reinterpret_cast<const uint4*>(a.items);
        for (int i = tid; i < nmine * 2; i += NTHREADS) dst[i] = src[(long long)(first + (i >> 1) * nj) * 2 + (i & 1)];
        float* bl = reinterpret_cast<float*>(smem + BIAS_OFF);
        if (tid < NT * 32) bl[tid] = a.bias[tid];
    }
    __syncthreads();
    const WorkItem* items = reinterpret_cast<const WorkItem*>(smem + ITEMS_OFF);

    if (wave >= NCW)
    {
        // ================= loader waves =================
        // Stage t = patch P(t) in ring slot t%3 + weight image W(t) in weight buffer t&1.  P(t) may be issued once the
        // MFMA waves are past stage t-3 (barrier B_{t-2}), W(t) once they are past stage t-2 (barrier B_{t-1}).  Issue
        // order per loader wave:   W(0) P(0) P(1) | B_0 | W(1) P(2) | B_1 | W(2) P(3) | ...
        // so before barrier B_s the newest NP instructions are exactly P(s+1) and "s_waitcnt vmcnt(NP)" means
        // "P(s), W(s) and everything older have landed" (vector loads retire in order).
        const int lw = wave - NCW, ltid = lw * 64 + lane;
        const char* wbase = static_cast<const char*>(a.wpk) + (lw * 64 + lane) * 16;
        auto issue_w = [&](int wck, int wsel) {
            if (a.dbg & 1) return; // ablation: no LDS-DMA
            const char* wsrc = wbase + (long long)wck * WBYTES;
            char* wb = smem + WOFF + wsel * WBYTES + lw * 1024;
#pragma unroll
            for (int i = 0; i < WPASS; i++)
                if (i * 256 + lw * 64 < WITEMS)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + i * 4096),
                                                     (__attribute__((address_space(3))) void*)(wb + i * 4096), 16, 0, 0);
        };
        auto wait_newest_patch_only_then_barrier = [&]() {
            // 39 one-KiB patch pieces: loader waves 0-2 own 10 each, wave 3 owns 9
            if (lw < 3) asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(9)\n\ts_barrier" ::: "memory");
        };
        // Two input chunks (the 64->32 convs): chunk ck always lands in weight buffer ck, so after the first block the two
        // images ARE the conv's whole weight set -- keep them, and stream patches only (-31 % LDS-DMA bytes per stage).
        const bool w_resident = (nplanes == 2) && !(a.dbg & 32768);
        issue_w(0, 0);
        int t = 0, slot3 = 0, wck = 1 % nplanes; // wck = chunk of the next weight image to issue, W(t-1) at loop index t
        for (int r = 0; r < nmine; r++)
        {
            const WorkItem it = items[r];
            const int H = it.H, W = it.W, Wi = UPS ? (W >> 1) : W;
            unsigned srcoff[kPatchIters]; // byte offset from (plane pixel 0 - kGuard); 0 = the plane's zero guard
#pragma unroll
            for (int i = 0; i < kPatchIters; i++)
            {
                const int jj = ltid + i * 256;
                const int px = jj >> 2, sl = jj & 3;
                const int rr = px / kPatchW, cc = px - rr * kPatchW;
                const int gy = it.y0 - 1 + rr, gx = it.x0 - 1 + cc;
                const bool ok = (jj < kPatchItems) && gy >= 0 && gy < H && gx >= 0 && gx < W;
                const int sy = UPS ? (gy >> 1) : gy, sx = UPS ? (gx >> 1) : gx;
                srcoff[i] = ok ? unsigned(kGuard + (sy * Wi + sx) * 64 + ((sl ^ ((cc >> 2) & 3)) << 4)) : 0u;
            }
            for (int ck = 0; ck < nplanes; ck++, t++)
            {
                if (t >= 2)
                {
                    wait_newest_patch_only_then_barrier(); // B_{t-2}
                    if (!(w_resident && t >= 3)) issue_w(wck, (t - 1) & 1); // W(t-1)
                    wck = (wck + 1 == nplanes) ? 0 : wck + 1;
                }
                char* buf = smem + slot3 * kPatchLds + lw * 1024;
                slot3 = slot3 == kRingDepth - 1 ? 0 : slot3 + 1;
                const char* gbase = ((ck < a.n0) ? plane_ptr(a.src0, it.slot, ck) : plane_ptr(a.src1, it.slot, ck - a.n0)) - kGuard;
                if (a.dbg & 1) continue; // ablation: no LDS-DMA
#pragma unroll
                for (int i = 0; i < kPatchIters - 1; i++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + srcoff[i]),
                                                     (__attribute__((address_space(3))) void*)(buf + i * 4096), 16, 0, 0);
                if (lw < 3)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + srcoff[kPatchIters - 1]),
                                                     (__attribute__((address_space(3))) void*)(buf + (kPatchIters - 1) * 4096), 16, 0, 0);
            }
        }
        if (S >= 2)
        {
            wait_newest_patch_only_then_barrier(); // B_{S-2}
            if (!(w_resident && S >= 3)) issue_w(wck, (S - 1) & 1); // W(S-1)
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); // B_{S-1}
        return;
    }

    // ================= MFMA waves (identical arithmetic to conv3x3_pipe) =================
    const int l32 = lane & 31, hi = lane >> 5;
    const int wrow = wave & 3, ntw = wave >> 2;
    int xcol[3];
#pragma unroll
    for (int dx = 0; dx < 3; dx++)
    {
        const int c = l32 + dx;
        xcol[dx] = wrow * 4 * kPatchW * 64 + c * 64 + ((hi ^ ((c >> 2) & 3)) << 4);
    }
    const int woff = WOFF + (ntw * 32 + l32) * 64 + ((hi ^ ((l32 >> 2) & 3)) << 4);
    const float* bias_lds = reinterpret_cast<const float*>(smem + BIAS_OFF);

    // OVL: the act-only epilogue of the NT = 1 convs (276 of the 351) is folded into the LAST stage of its block.  The
    // final 24 MFMAs run row-major (rows of the wave's 4x32-pixel tile finish 6 MFMAs apart) and each finished row goes
    // bias -> LeakyReLU -> fp16 -> private LDS row (transpose) -> one coalesced 16-B/lane buffer store per 16 pixels
    // while the matrix pipe works on the following rows; everything is branch-free (out-of-image lanes / rows are
    // dropped by the buffer resource's range check) so the whole stage stays one scheduling region.  s_memtime stamps
    // had the in-line epilogue at ~2,700 of the ~8,500 cycles of a 64->32 block.
    constexpr bool OVL = (EPI == 1 && NT == 1);
    f32x16 acc[4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[r][e] = 0.f;
    f32x4 bqv[4]; // bias in accumulator layout, resident for the whole kernel in the OVL variant
#pragma unroll
    for (int q = 0; q < 4; q++) bqv[q] = *reinterpret_cast<const f32x4*>(bias_lds + ntw * 32 + q * 8 + hi * 4);
    // transpose scratch = the two patch rows only this wave reads (4*wrow+2, +3), see conv_epilogue_lds
    const int rpx = lane >> 2, rpiece = lane & 3;
    const int scr_w = (4 * wrow + 2) * (kPatchW * 64) + l32 * 64 + hi * 8;
    const int wswz = (l32 >> 1) & 3;
    int scr_r[2];
#pragma unroll
    for (int k = 0; k < 2; k++)
    {
        const int col = k * 16 + rpx;
        scr_r[k] = (4 * wrow + 2) * (kPatchW * 64) + col * 64 + ((rpiece ^ ((col >> 1) & 3)) << 4);
    }
    const float slope = a.lrelu ? 0.2f : 1.f;
    const bool ovl = OVL && nplanes >= 2 && !(a.dbg & 64);
    // +inf the compiler cannot see through: med3(v, slope*v, +inf) = max(v, slope*v) stays ONE v_med3_f32; with a literal it
    // is folded to maxnum, which canonicalises the MFMA result first (one more v_max per value)
    float pinf = __builtin_inff();
    asm volatile("" : "+s"(pinf));
    half8 zfrag;
    {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        u32x4 z = {0u, 0u, 0u, 0u};
        asm volatile("" : "+v"(z));
        zfrag = __builtin_bit_cast(half8, z);
    }
    f32x16 bias16; // the bias in accumulator layout: C operand of the re-initialising MFMA
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int e = 0; e < 4; e++) bias16[q * 4 + e] = bqv[q][e];
    if (ovl)
    {
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = bias16;
    }

    int r = 0, ck = 0, slot3 = 0;
    WorkItem it = items[0];
    const bool tracing = a.trace && blockIdx.x == 0 && wave == 0;
#define RSR_LOAD_STEP(X, Wf, T)                                                                                     \
    {                                                                                                                \
        constexpr int dx_ = (T) >> 1, cb_ = (T)&1;                                                                   \
        _Pragma("unroll") for (int dy = 0; dy < 3; dy++)                                                             \
        {                                                                                                            \
            X[dy] = *reinterpret_cast<const half8*>(buf + ((xcol[dx_] ^ (cb_ << 5)) + dy * (kPatchW * 64)));         \
            Wf[dy] = *reinterpret_cast<const half8*>(wb + ((woff ^ (cb_ << 5)) + (dy * 3 + dx_) * (NT * 32 * 64)));  \
        }                                                                                                            \
        _Pragma("unroll") for (int rr = 3; rr < 6; rr++)                                                             \
            X[rr] = *reinterpret_cast<const half8*>(buf + ((xcol[dx_] ^ (cb_ << 5)) + rr * (kPatchW * 64)));         \
    }
#define RSR_MFMA_STEP(X, Wf)                                                                                         \
    {                                                                                                                \
        _Pragma("unroll") for (int dy = 0; dy < 3; dy++) _Pragma("unroll") for (int rr = 0; rr < 4; rr++)            \
            acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[dy], X[rr + dy], acc[rr], 0, 0, 0);                  \
    }
#ifdef RSR_EXP_OVLTRACE // experiment build: s_memtime stamps at the window boundaries of the overlapped-epilogue stage
    unsigned long long ovl_ts[6];
#define RSR_STAMP(K)                                                                                                 \
    ovl_ts[K] = __builtin_amdgcn_s_memtime();                                                                        \
    __builtin_amdgcn_sched_barrier(0);
#else
#define RSR_STAMP(K)
#endif
#define RSR_MAIN_SCHED()                                                                                             \
    {                                                                                                                \
        __builtin_amdgcn_sched_group_barrier(0x100, 18, 0);                                                          \
        _Pragma("unroll") for (int tt = 0; tt < 4; tt++)                                                             \
        {                                                                                                            \
            __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);                                                         \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                       \
            __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);                                                         \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                       \
            __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);                                                         \
            __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);                                                       \
        }                                                                                                            \
    }
// one MFMA then VALU work, six times: the interleave of a finishing row with the previous row's epilogue arithmetic
#define RSR_ROW_SCHED()                                                                                              \
    {                                                                                                                \
        _Pragma("unroll") for (int m = 0; m < 6; m++)                                                                \
        {                                                                                                            \
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                                         \
            __builtin_amdgcn_sched_group_barrier(0x2, 6, 0);                                                         \
        }                                                                                                            \
    }
// the last two (dx, cb) steps of one output row: 6 dependent MFMAs
#define RSR_MFMA_ROW(RR)                                                                                             \
    {                                                                                                                \
        _Pragma("unroll") for (int dy = 0; dy < 3; dy++)                                                             \
            acc[RR] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W0[dy], X0[(RR) + dy], acc[RR], 0, 0, 0);               \
        _Pragma("unroll") for (int dy = 0; dy < 3; dy++)                                                             \
            acc[RR] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W1[dy], X1[(RR) + dy], acc[RR], 0, 0, 0);               \
    }
// finished row RR (the bias is already in the accumulator, see RSR_ROW_CLEAR): LeakyReLU = max(v, slope*v), fp16,
// transpose-write into scratch row (RR & 1)
#define RSR_ROW_TO_LDS(RR)                                                                                           \
    {                                                                                                                \
        char* wr_ = sbuf + scr_w + ((RR)&1) * (kPatchW * 64);                                                        \
        _Pragma("unroll") for (int q = 0; q < 4; q++)                                                                \
        {                                                                                                            \
            typedef float f32x2 __attribute__((ext_vector_type(2)));                                                 \
            const f32x2 v01 = {acc[RR][q * 4 + 0], acc[RR][q * 4 + 1]}, v23 = {acc[RR][q * 4 + 2], acc[RR][q * 4 + 3]};         \
            const f32x2 s01 = v01 * slope, s23 = v23 * slope;                                                        \
            half4 o;                                                                                                 \
            o[0] = (_Float16)__builtin_amdgcn_fmed3f(v01[0], s01[0], pinf);                              \
            o[1] = (_Float16)__builtin_amdgcn_fmed3f(v01[1], s01[1], pinf);                              \
            o[2] = (_Float16)__builtin_amdgcn_fmed3f(v23[0], s23[0], pinf);                              \
            o[3] = (_Float16)__builtin_amdgcn_fmed3f(v23[1], s23[1], pinf);                              \
            *reinterpret_cast<half4*>(wr_ + ((q ^ wswz) << 4)) = o;                                                  \
        }                                                                                                            \
    }
// re-initialise a consumed accumulator row to the BIAS on the matrix pipe (0 x 0 + bias; the zero fragment is opaque to
// the compiler): 64 v_mov per block would otherwise sit, unhidden, at the end of the stage, and the bias add leaves the
// epilogue arithmetic
#define RSR_ROW_CLEAR(RR)                                                                                            \
    {                                                                                                                \
        asm volatile("" : "+v"(zfrag)); /* a 'new' value each time, or the four clears are CSE'd into one + 48 v_mov */ \
        acc[RR] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zfrag, zfrag, bias16, 0, 0, 0);                             \
    }
#define RSR_ROW_FROM_LDS(RR, T)                                                                                      \
    {                                                                                                                \
        _Pragma("unroll") for (int k = 0; k < 2; k++)                                                                \
            T[k] = *reinterpret_cast<const u32x4*>(sbuf + scr_r[k] + ((RR)&1) * (kPatchW * 64));                     \
    }
#define RSR_ROW_STORE(RR, T)                                                                                         \
    {                                                                                                                \
        const int y_ = it.y0 + wrow * 4 + (RR);                                                                      \
        _Pragma("unroll") for (int k = 0; k < 2; k++)                                                                \
            __builtin_amdgcn_raw_buffer_store_b128(T[k], y_ < it.H ? rs : rs0, voff[k] + y_ * rowb, 0, 0); /* soffset 0: see conv_flow.hip row_store */ \
    }
    for (int s = 0; s < S; s++)
    {
        unsigned long long t_arrive = 0;
        if (tracing) t_arrive = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // B_s: patch s and weights s are in LDS
        if (tracing && s < 512 && lane == 0)
        {
            a.trace[2 * s] = t_arrive;                         // arrival at barrier B_s
            a.trace[2 * s + 1] = __builtin_amdgcn_s_memtime(); // release from barrier B_s
        }
        const char* buf = smem + slot3 * kPatchLds;
        const char* wb = smem + (s & 1) * WBYTES;
        slot3 = slot3 == kRingDepth - 1 ? 0 : slot3 + 1;
        if (!(a.dbg & 2))
        {
            half8 X0[6], X1[6], W0[3], W1[3];
            if (ovl && ck == nplanes - 1)
            {
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                char* sbuf = const_cast<char*>(buf);
                char* obase = const_cast<char*>(plane_ptr(a.out16, it.slot, 0));
                const int rowb = it.W * 64;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(obase, 0, (a.dbg & 4) ? 0 : it.H * rowb, 0x00020000);
                const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(obase, 0, 0, 0x00020000);
                int voff[2];
#pragma unroll
                for (int k = 0; k < 2; k++)
                {
                    const int x = it.x0 + k * 16 + rpx;
                    voff[k] = x < it.W ? x * 64 + rpiece * 16 : int(0x80000000u);
                }
                u32x4 t0[2], t1[2], t2[2];
                RSR_LOAD_STEP(X0, W0, 0)
                RSR_LOAD_STEP(X1, W1, 1)
                RSR_MFMA_STEP(X0, W0)
                RSR_LOAD_STEP(X0, W0, 2)
                RSR_MFMA_STEP(X1, W1)
                RSR_LOAD_STEP(X1, W1, 3)
                RSR_MFMA_STEP(X0, W0)
                RSR_LOAD_STEP(X0, W0, 4)
                RSR_MFMA_STEP(X1, W1)
                RSR_LOAD_STEP(X1, W1, 5)
                RSR_MAIN_SCHED()
                __builtin_amdgcn_sched_barrier(0);
                RSR_STAMP(0)
                RSR_MFMA_ROW(0)
                __builtin_amdgcn_sched_barrier(0);
                RSR_STAMP(1)
                RSR_MFMA_ROW(1) // || row 0: arithmetic, transpose-write
                RSR_ROW_TO_LDS(0)
                RSR_ROW_SCHED()
                __builtin_amdgcn_sched_barrier(0);
                RSR_STAMP(2)
                RSR_MFMA_ROW(2) // || row 0: transpose-read; row 1: arithmetic, transpose-write
                RSR_ROW_CLEAR(0)
                RSR_ROW_FROM_LDS(0, t0)
                RSR_ROW_TO_LDS(1)
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 9, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                _Pragma("unroll") for (int m = 0; m < 6; m++)
                {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x2, 7, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                RSR_STAMP(3)
                RSR_MFMA_ROW(3) // || row 0: store; row 1: transpose-read; row 2: arithmetic, transpose-write
                RSR_ROW_CLEAR(1)
                RSR_ROW_STORE(0, t0)
                RSR_ROW_FROM_LDS(1, t1)
                RSR_ROW_TO_LDS(2)
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x40, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 9, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                _Pragma("unroll") for (int m = 0; m < 6; m++)
                {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x2, 6, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                RSR_STAMP(4)
                // exposed tail: the operand fragments are dead here, so there are registers for two more rows in flight
                RSR_ROW_FROM_LDS(2, t2)
                RSR_ROW_TO_LDS(3)
                RSR_ROW_FROM_LDS(3, t0)
                RSR_ROW_STORE(1, t1)
                RSR_ROW_CLEAR(2)
                RSR_ROW_CLEAR(3)
                RSR_ROW_STORE(2, t2)
                RSR_ROW_STORE(3, t0)
#ifdef RSR_EXP_OVLTRACE
                __builtin_amdgcn_sched_barrier(0);
                RSR_STAMP(5)
                if (tracing && s < 512 && lane == 0)
                {
#pragma unroll
                    for (int k = 0; k < 6; k++) a.trace[1024 + 8 * s + k] = ovl_ts[k];
                }
#endif
            }
            else
            {
                RSR_LOAD_STEP(X0, W0, 0)
                RSR_LOAD_STEP(X1, W1, 1)
                RSR_MFMA_STEP(X0, W0)
                RSR_LOAD_STEP(X0, W0, 2)
                RSR_MFMA_STEP(X1, W1)
                RSR_LOAD_STEP(X1, W1, 3)
                RSR_MFMA_STEP(X0, W0)
                RSR_LOAD_STEP(X0, W0, 4)
                RSR_MFMA_STEP(X1, W1)
                RSR_LOAD_STEP(X1, W1, 5)
                RSR_MFMA_STEP(X0, W0)
                RSR_MFMA_STEP(X1, W1)
                RSR_MAIN_SCHED()
                __builtin_amdgcn_sched_group_barrier(0x8, 24, 0);
            }
        }
        if (++ck == nplanes)
        {
            if (!ovl)
            {
                if (!(a.dbg & 4))
                {
                    const bool interior = it.y0 + kBlkH <= it.H && it.x0 + kBlkW <= it.W;
                    if (EPI == 0) conv_epilogue(a, acc, bqv, ntw, it.slot, it.y0, it.x0, it.H, it.W, wrow, l32, hi);
                    else if (EPI == 1 && NT == 1)
                    {
                        if (interior) conv_epilogue_lds<false>(a, acc, bqv, it.slot, it.y0, it.x0, it.H, it.W, wrow, lane, const_cast<char*>(buf));
                        else conv_epilogue_lds<true>(a, acc, bqv, it.slot, it.y0, it.x0, it.H, it.W, wrow, lane, const_cast<char*>(buf));
                    }
                    else if (interior)
                        conv_epilogue_t<EPI, false>(a, acc, bqv, ntw, it.slot, it.y0, it.x0, it.H, it.W, wrow, l32, hi);
                    else
                        conv_epilogue_t<EPI, true>(a, acc, bqv, ntw, it.slot, it.y0, it.x0, it.H, it.W, wrow, l32, hi);
                }
#pragma unroll
                for (int rr = 0; rr < 4; rr++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[rr][e] = 0.f;
            }
            ck = 0;
            r++;
            if (r < nmine) it = items[r];
        }
    }
#undef RSR_LOAD_STEP
#undef RSR_MFMA_STEP
#undef RSR_MFMA_ROW
#undef RSR_MAIN_SCHED
#undef RSR_STAMP
#undef RSR_ROW_SCHED
#undef RSR_ROW_TO_LDS
#undef RSR_ROW_CLEAR
#undef RSR_ROW_FROM_LDS
#undef RSR_ROW_STORE
}

template <int NT, bool UPS, int EPI>
static bool launch_conv_ring_t(const ConvArgs& a, int ncu, hipStream_t st)
{
    int grid = ncu & ~7;
    const int per = (a.nitems + 7) / 8;
    if (per * 8 < grid) grid = per * 8;
    const int nj = grid / 8, nmine_max = (per + nj - 1) / nj;
    const size_t fixed = size_t(kRingDepth) * kPatchLds + 2 * size_t(9 * NT * 32 * 64) + 256;
    if (fixed + sizeof(WorkItem) > 160 * 1024) return false; // 64 output channels: two weight images do not fit, use conv3x3_pipe
    // The workgroup keeps its work-item descriptors in LDS; a list that does not fit (the 4x level has 16x the blocks) is
    // walked in several launches -- they read the same finished input and write disjoint blocks, stream order suffices.
    const int cap_items = int((160 * 1024 - fixed) / sizeof(WorkItem));
    const int nsub = (nmine_max + cap_items - 1) / cap_items;
    const int chunk = (a.nitems + nsub - 1) / nsub;
    for (int off = 0; off < a.nitems; off += chunk)
    {
        ConvArgs s = a;
        s.items = a.items + off;
        s.nitems = a.nitems - off < chunk ? a.nitems - off : chunk;
        int g = ncu & ~7;
        const int p = (s.nitems + 7) / 8;
        if (p * 8 < g) g = p * 8;
        const int njs = g / 8, nm = (p + njs - 1) / njs;
        const size_t lds = fixed + size_t(nm) * sizeof(WorkItem);
        hipLaunchKernelGGL((conv3x3_ring<NT, UPS, EPI>), dim3(g), dim3((4 * NT + 4) * 64), lds, st, s);
    }
    return true;
}

bool launch_conv_ring(const ConvArgs& a, int nt, int ncu, hipStream_t st)
{
    if (a.nitems <= 0) return true;
    if (nt != 1) return false;
    const bool ups = a.lvl_out != a.lvl_in;
    int epi = 0;
    if (a.out16.base && !a.out32a.base && !a.out32b.base && !a.out_planar3 && !(a.dbg & 16))
    {
        if (a.res1_kind == 0 && a.res2_kind == 0) epi = 1;
        else if (a.res1_kind == 1 && (a.res2_kind == 0 || a.res2_kind == 1)) epi = 2;
    }
#define RSR_RING(UPS_) (epi == 1 ? launch_conv_ring_t<1, UPS_, 1>(a, ncu, st) : epi == 2 ? launch_conv_ring_t<1, UPS_, 2>(a, ncu, st) : launch_conv_ring_t<1, UPS_, 0>(a, ncu, st))
    return ups ? RSR_RING(true) : RSR_RING(false);
#undef RSR_RING
}

template <int NT, bool UPS, bool DMA>
static void launch_conv_t(const ConvArgs& a, hipStream_t st)
{
    const int per = (a.nitems + 7) / 8;
    const size_t lds = size_t(kPatchLds) + size_t(9 * NT * 32 * 64);
    hipLaunchKernelGGL((conv3x3_mfma<NT, UPS, DMA>), dim3(per * 8), dim3(kThreads), lds, st, a);
}

void launch_conv(const ConvArgs& a, int nt, bool dma, hipStream_t st)
{
    if (a.nitems <= 0) return;
    const bool ups = a.lvl_out != a.lvl_in;
    const int key = (nt == 2 ? 4 : 0) | (ups ? 2 : 0) | (dma ? 1 : 0);
    switch (key)
    {
    case 0: launch_conv_t<1, false, false>(a, st); break;
    case 1: launch_conv_t<1, false, true>(a, st); break;
    case 2: launch_conv_t<1, true, false>(a, st); break;
    case 3: launch_conv_t<1, true, true>(a, st); break;
    case 4: launch_conv_t<2, false, false>(a, st); break;
    case 5: launch_conv_t<2, false, true>(a, st); break;
    case 6: launch_conv_t<2, true, false>(a, st); break;
    default: launch_conv_t<2, true, true>(a, st); break;
    }
}

// > 64 KiB of dynamic LDS needs an opt-in per kernel AND per device (a process-wide "done" flag would leave the
// second GPU of a multi-device process without it): the engine calls this once per context, on its own device.
hipError_t kernels_init_device()
{
    hipError_t e = hipSuccess;
#define RSR_ATTR(K, BYTES)                                                                                           \
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, int(BYTES));
#define RSR_COMMA ,
#define RSR_PIPE_ATTRS(NT, UPS)                                                                                      \
    RSR_ATTR(conv3x3_pipe<NT RSR_COMMA UPS RSR_COMMA 0>, pipe_lds(NT))                                               \
    RSR_ATTR(conv3x3_pipe<NT RSR_COMMA UPS RSR_COMMA 1>, pipe_lds(NT))                                               \
    RSR_ATTR(conv3x3_pipe<NT RSR_COMMA UPS RSR_COMMA 2>, pipe_lds(NT))
    RSR_PIPE_ATTRS(1, false) RSR_PIPE_ATTRS(1, true) RSR_PIPE_ATTRS(2, false) RSR_PIPE_ATTRS(2, true)
#define RSR_RING_ATTRS(UPS)                                                                                          \
    RSR_ATTR(conv3x3_ring<1 RSR_COMMA UPS RSR_COMMA 0>, 160 * 1024)                                                  \
    RSR_ATTR(conv3x3_ring<1 RSR_COMMA UPS RSR_COMMA 1>, 160 * 1024)                                                  \
    RSR_ATTR(conv3x3_ring<1 RSR_COMMA UPS RSR_COMMA 2>, 160 * 1024)
    RSR_RING_ATTRS(false) RSR_RING_ATTRS(true)
#define RSR_MFMA_ATTRS(NT, UPS)                                                                                      \
    RSR_ATTR(conv3x3_mfma<NT RSR_COMMA UPS RSR_COMMA false>, size_t(kPatchLds) + size_t(9 * NT * 32 * 64))           \
    RSR_ATTR(conv3x3_mfma<NT RSR_COMMA UPS RSR_COMMA true>, size_t(kPatchLds) + size_t(9 * NT * 32 * 64))
    RSR_MFMA_ATTRS(1, false) RSR_MFMA_ATTRS(1, true) RSR_MFMA_ATTRS(2, false) RSR_MFMA_ATTRS(2, true)
#undef RSR_MFMA_ATTRS
#undef RSR_RING_ATTRS
#undef RSR_PIPE_ATTRS
#undef RSR_COMMA
#undef RSR_ATTR
    if (e == hipSuccess) e = flow_init_device();
    return e;
}

// =============================================================================================
// pre / post processing
// =============================================================================================

// reflect-101 exactly as realsr_preproc.comp:59-62.  The shader's single reflection leaves the image
// when the halo exceeds the image (n <= prepadding): the reference then reads out of bounds; here
// the index is clamped (as the oracle does) so tiny images stay memory-safe and deterministic.
__device__ __forceinline__ int reflect101(int v, int n)
{
    v = abs(v);
    v = (n - 1) - abs(v - (n - 1));
    return min(max(v, 0), n - 1);
}

// realsr_preproc.comp:47-95 and realsr_preproc_tta.comp:54-113, for a batch of tiles.
// One thread per padded-tile pixel; writes the 32-channel fp16 input plane(s) (channels 3..31 = 0).
// The band-relative coordinates of the shader (crop_x/crop_y/pad) are folded into x_org/y_org:
// reflecting against the band equals reflecting against the image (engine.cpp explains why).
__global__ __launch_bounds__(256) void preproc_tiles(const PreArgs a)
{
    const BaseTile t = a.tiles[blockIdx.z];
    const int gx = blockIdx.x * 32 + (threadIdx.x & 31);
    const int gy = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (gx >= t.tw || gy >= t.th) return;
    const int x = reflect101(gx + t.x_org, a.w);
    const int y = reflect101(gy + t.y_org, a.h);
    const uint8_t* p = a.img + ((long long)y * a.w + x) * a.c;
    const float norm_val = 1 / 255.f;
    const int i0 = a.bgr ? 2 : 0, i2 = a.bgr ? 0 : 2;
    half8 v0;
#pragma unroll
    for (int e = 0; e < 8; e++) v0[e] = (_Float16)0.f;
    v0[0] = (_Float16)((float)p[i0] * norm_val);
    v0[1] = (_Float16)((float)p[1] * norm_val);
    v0[2] = (_Float16)((float)p[i2] * norm_val);
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    const int nv = a.tta ? 8 : 1;
    for (int k = 0; k < nv; k++)
    {
        int oy, ox, ow;
        switch (k)
        { // realsr_preproc_tta.comp:104-111
        default: oy = gy; ox = gx; ow = t.tw; break;
        case 1: oy = gy; ox = t.tw - 1 - gx; ow = t.tw; break;
        case 2: oy = t.th - 1 - gy; ox = t.tw - 1 - gx; ow = t.tw; break;
        case 3: oy = t.th - 1 - gy; ox = gx; ow = t.tw; break;
        case 4: oy = gx; ox = gy; ow = t.th; break;
        case 5: oy = gx; ox = t.th - 1 - gy; ow = t.th; break;
        case 6: oy = t.tw - 1 - gx; ox = t.th - 1 - gy; ow = t.th; break;
        case 7: oy = t.tw - 1 - gx; ox = gy; ow = t.th; break;
        }
        char* dst = static_cast<char*>(a.in_plane) + (long long)(t.slot0 + k) * a.slot_stride + ((long long)oy * ow + ox) * (a.plane_ch * 2);
        *reinterpret_cast<half8*>(dst) = v0;
        *reinterpret_cast<uint4*>(dst + 16) = z;
        if (a.plane_ch == 32)
        {
            *reinterpret_cast<uint4*>(dst + 32) = z;
            *reinterpret_cast<uint4*>(dst + 48) = z;
        }
    }
}

void launch_preproc_tiles(const PreArgs& a, int max_tw, int max_th, hipStream_t st)
{
    if (a.ntiles <= 0) return;
    const dim3 grid((max_tw + 31) / 32, (max_th + 7) / 8, a.ntiles), block(256);
    hipLaunchKernelGGL(preproc_tiles, grid, block, 0, st, a);
}

// store conversion of realsr_postproc.comp:71-78 (v + 0.5, floor, clamp 0..255); negative values
// saturate to 0 as on the reference CPU path (GLSL uint(floor(v)) is undefined there).
__device__ __forceinline__ uint8_t post_store(float v)
{
    v = floorf(v + 0.5f);
    v = fminf(fmaxf(v, 0.f), 255.f);
    return (uint8_t)v;
}

// ncnn Interp bicubic coefficients (alpha channel only; realsr.cpp:128-140)
__device__ __forceinline__ void cubic_coeffs(int w, int outw, int dx, int& sx, float c[4])
{
    const float scale = (float)w / (float)outw;
    float fx = ((float)dx + 0.5f) * scale - 0.5f;
    sx = (int)floorf(fx);
    fx -= (float)sx;
    const float A = -0.75f;
    const float fx0 = fx + 1.f, fx1 = fx, fx2 = 1.f - fx;
    c[0] = A * fx0 * fx0 * fx0 - 5.f * A * fx0 * fx0 + 8.f * A * fx0 - 4.f * A;
    c[1] = (A + 2.f) * fx1 * fx1 * fx1 - (A + 3.f) * fx1 * fx1 + 1.f;
    c[2] = (A + 2.f) * fx2 * fx2 * fx2 - (A + 3.f) * fx2 * fx2 + 1.f;
    c[3] = 1.f - c[0] - c[1] - c[2];
    if (sx <= -1) { sx = 1; c[0] = 1.f - c[3]; c[1] = c[3]; c[2] = 0.f; c[3] = 0.f; }
    if (sx == 0) { sx = 1; c[0] = c[0] + c[1]; c[1] = c[2]; c[2] = c[3]; c[3] = 0.f; }
    if (sx == w - 2) { sx = w - 3; c[3] = c[2] + c[3]; c[2] = c[1]; c[1] = c[0]; c[0] = 0.f; }
    if (sx >= w - 1) { sx = w - 3; c[3] = 1.f - c[0]; c[2] = c[0]; c[1] = 0.f; c[0] = 0.f; }
}

__device__ __forceinline__ int clampi(int v, int n) { return v < 0 ? 0 : (v > n - 1 ? n - 1 : v); }

// realsr_postproc.comp:47-89 and realsr_postproc_tta.comp:54-110 for a batch of tiles.
// One thread per output pixel of the tile's un-padded x4 rectangle.
__global__ __launch_bounds__(256) void postproc_tiles(const PostArgs a)
{
    const BaseTile t = a.tiles[blockIdx.z];
    const int gx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int gy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (gx >= t.out_w || gy >= t.out_h) return;
    const int w = t.tw * 4, h = t.th * 4;
    const long long cstep = (long long)w * h;
    const int sx = gx + a.crop, sy = gy + a.crop;
    uint8_t* o = a.out + ((long long)(t.out_y + gy) * a.out_w + t.out_x + gx) * a.c;
    const _Float16* b0 = reinterpret_cast<const _Float16*>(static_cast<const char*>(a.planar3) + (long long)t.slot0 * a.slot_stride);
    float v[3];
    if (!a.tta)
    {
#pragma unroll
        for (int q = 0; q < 3; q++) v[q] = (float)b0[q * cstep + (long long)sy * w + sx];
    }
    else
    {
        const long long ss = a.slot_stride / 2;
#pragma unroll
        for (int q = 0; q < 3; q++)
        {
            const _Float16* b = b0 + q * cstep;
            // realsr_postproc_tta.comp:76-85
            const float v0 = (float)b[(long long)sy * w + sx];
            const float v1 = (float)b[ss + (long long)sy * w + (w - 1 - sx)];
            const float v2 = (float)b[2 * ss + (long long)(h - 1 - sy) * w + (w - 1 - sx)];
            const float v3 = (float)b[3 * ss + (long long)(h - 1 - sy) * w + sx];
            const float v4 = (float)b[4 * ss + (long long)sx * h + sy];
            const float v5 = (float)b[5 * ss + (long long)sx * h + (h - 1 - sy)];
            const float v6 = (float)b[6 * ss + (long long)(w - 1 - sx) * h + (h - 1 - sy)];
            const float v7 = (float)b[7 * ss + (long long)(w - 1 - sx) * h + sy];
            v[q] = (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7) * 0.125f;
        }
    }
    const uint8_t r = post_store(v[0] * 255.f), g = post_store(v[1] * 255.f), bl = post_store(v[2] * 255.f);
    o[a.bgr ? 2 : 0] = r;
    o[1] = g;
    o[a.bgr ? 0 : 2] = bl;
    if (a.c == 4)
    {
        // alpha: bicubic x4 of the un-padded tile's alpha (0..255 units), realsr.cpp:431-442,
        // realsr_preproc.comp:79-88 (crop), realsr_postproc.comp:58-61
        const int aw = t.out_w / 4, ah = t.out_h / 4;
        const int ax0 = t.out_x / 4, ay0 = t.out_y / 4;
        int bx, by;
        float cx[4], cy[4];
        cubic_coeffs(aw, t.out_w, gx, bx, cx);
        cubic_coeffs(ah, t.out_h, gy, by, cy);
        float rows[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            const int yy = ay0 + clampi(by - 1 + j, ah);
            const uint8_t* rp = a.in_img + ((long long)yy * a.in_w + ax0) * 4 + 3;
            rows[j] = (float)rp[clampi(bx - 1, aw) * 4] * cx[0] + (float)rp[clampi(bx, aw) * 4] * cx[1] +
                      (float)rp[clampi(bx + 1, aw) * 4] * cx[2] + (float)rp[clampi(bx + 2, aw) * 4] * cx[3];
        }
        const float av = rows[0] * cy[0] + rows[1] * cy[1] + rows[2] * cy[2] + rows[3] * cy[3];
        o[3] = post_store(av);
    }
}

void launch_postproc_tiles(const PostArgs& a, int max_ow, int max_oh, hipStream_t st)
{
    if (a.ntiles <= 0) return;
    const dim3 grid((max_ow + 63) / 64, (max_oh + 3) / 4, a.ntiles), block(256);
    hipLaunchKernelGGL(postproc_tiles, grid, block, 0, st, a);
}

// ---- shader-shaped kernels: same arithmetic, the shaders' own buffer layouts -----------------
struct Ptr8
{
    uint16_t* p[8];
};
struct CPtr8
{
    const uint16_t* p[8];
};

__global__ __launch_bounds__(256) void preproc_shader(const uint8_t* bottom, int w, int h, int channels, Ptr8 top, int ntop,
                                                      int outw, int outh, int outcstep, int pad_top, int pad_left,
                                                      int crop_x, int crop_y, uint16_t* alpha, int alphaw, int alphah, int bgr)
{
    int gx = blockIdx.x * 32 + (threadIdx.x & 31);
    int gy = blockIdx.y * 8 + (threadIdx.x >> 5);
    const int gz = blockIdx.z;
    if (gx >= outw || gy >= outh || gz >= channels) return;
    int x = gx + crop_x - pad_left;
    int y = gy + crop_y - pad_top;
    x = reflect101(x, w);
    y = reflect101(y, h);
    const int v_offset = y * w + x;
    float v;
    if (bgr == 1 && gz != 3) v = (float)bottom[v_offset * channels + 2 - gz];
    else v = (float)bottom[v_offset * channels + gz];
    if (gz == 3)
    {
        gx -= pad_left;
        gy -= pad_top;
        if (alpha && gx >= 0 && gx < alphaw && gy >= 0 && gy < alphah)
            reinterpret_cast<_Float16*>(alpha)[gy * alphaw + gx] = (_Float16)v;
        return;
    }
    const float norm_val = 1 / 255.f;
    const _Float16 hv = (_Float16)(v * norm_val);
    const int gzi = gz * outcstep;
    _Float16* const* T = reinterpret_cast<_Float16* const*>(top.p);
    T[0][gzi + gy * outw + gx] = hv;
    if (ntop == 8)
    {
        T[1][gzi + gy * outw + (outw - 1 - gx)] = hv;
        T[2][gzi + (outh - 1 - gy) * outw + (outw - 1 - gx)] = hv;
        T[3][gzi + (outh - 1 - gy) * outw + gx] = hv;
        T[4][gzi + gx * outh + gy] = hv;
        T[5][gzi + gx * outh + (outh - 1 - gy)] = hv;
        T[6][gzi + (outw - 1 - gx) * outh + (outh - 1 - gy)] = hv;
        T[7][gzi + (outw - 1 - gx) * outh + gy] = hv;
    }
}

void launch_preproc_shader(const uint8_t* bottom, int w, int h, int channels, uint16_t* const top[8], int ntop, int outw,
                           int outh, int outcstep, int pad_top, int pad_left, int crop_x, int crop_y, uint16_t* alpha,
                           int alphaw, int alphah, int bgr, hipStream_t st)
{
    Ptr8 t;
    for (int i = 0; i < 8; i++) t.p[i] = i < ntop ? top[i] : nullptr;
    const dim3 grid((outw + 31) / 32, (outh + 7) / 8, channels), block(256);
    hipLaunchKernelGGL(preproc_shader, grid, block, 0, st, bottom, w, h, channels, t, ntop, outw, outh, outcstep, pad_top,
                       pad_left, crop_x, crop_y, alpha, alphaw, alphah, bgr);
}

__global__ __launch_bounds__(256) void postproc_shader(CPtr8 bottom, int nbottom, int w, int h, int cstep, const uint16_t* alpha,
                                                       int alphaw, int alphah, uint8_t* top, int outw, int outh, int offset_x,
                                                       int gx_max, int crop_x, int crop_y, int channels, int bgr)
{
    const int gx = blockIdx.x * 32 + (threadIdx.x & 31);
    const int gy = blockIdx.y * 8 + (threadIdx.x >> 5);
    const int gz = blockIdx.z;
    if (gx >= gx_max || gy >= outh || gz >= channels) return;
    const _Float16* const* B = reinterpret_cast<const _Float16* const*>(bottom.p);
    float v;
    if (gz == 3) v = (float)reinterpret_cast<const _Float16*>(alpha)[gy * alphaw + gx];
    else
    {
        const int gzi = gz * cstep;
        const int sy = gy + crop_y, sx = gx + crop_x;
        if (nbottom == 1) v = (float)B[0][gzi + sy * w + sx];
        else
        {
            const float v0 = (float)B[0][gzi + sy * w + sx];
            const float v1 = (float)B[1][gzi + sy * w + (w - 1 - sx)];
            const float v2 = (float)B[2][gzi + (h - 1 - sy) * w + (w - 1 - sx)];
            const float v3 = (float)B[3][gzi + (h - 1 - sy) * w + sx];
            const float v4 = (float)B[4][gzi + sx * h + sy];
            const float v5 = (float)B[5][gzi + sx * h + (h - 1 - sy)];
            const float v6 = (float)B[6][gzi + (w - 1 - sx) * h + (h - 1 - sy)];
            const float v7 = (float)B[7][gzi + (w - 1 - sx) * h + sy];
            v = (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7) * 0.125f;
        }
        v = v * 255.f;
    }
    const int v_offset = gy * outw + gx + offset_x;
    const uint8_t u = post_store(v);
    if (bgr == 1 && gz != 3) top[v_offset * channels + 2 - gz] = u;
    else top[v_offset * channels + gz] = u;
}

void launch_postproc_shader(const uint16_t* const bottom[8], int nbottom, int w, int h, int cstep, const uint16_t* alpha,
                            int alphaw, int alphah, uint8_t* top, int outw, int outh, int offset_x, int gx_max, int crop_x,
                            int crop_y, int channels, int bgr, hipStream_t st)
{
    CPtr8 b;
    for (int i = 0; i < 8; i++) b.p[i] = i < nbottom ? bottom[i] : nullptr;
    const dim3 grid((gx_max + 31) / 32, (outh + 7) / 8, channels), block(256);
    hipLaunchKernelGGL(postproc_shader, grid, block, 0, st, b, nbottom, w, h, cstep, alpha, alphaw, alphah, top, outw, outh,
                       offset_x, gx_max, crop_x, crop_y, channels, bgr);
}

__global__ __launch_bounds__(256) void planar3_to_plane(const uint16_t* planar, int w, int h, void* plane, int plane_ch)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)w * h) return;
    const long long hw = (long long)w * h;
    uint4 v0 = make_uint4(0u, 0u, 0u, 0u);
    v0.x = (uint32_t)planar[i] | ((uint32_t)planar[hw + i] << 16);
    v0.y = (uint32_t)planar[2 * hw + i];
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    uint4* d = reinterpret_cast<uint4*>(static_cast<char*>(plane) + i * (plane_ch * 2));
    d[0] = v0;
    d[1] = z;
    if (plane_ch == 32)
    {
        d[2] = z;
        d[3] = z;
    }
}

void launch_planar3_to_plane(const uint16_t* planar, int w, int h, void* plane, int plane_ch, hipStream_t st)
{
    const long long n = (long long)w * h;
    hipLaunchKernelGGL(planar3_to_plane, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, planar, w, h, plane, plane_ch);
}

// Zero the 64-byte guard in front of `count` planes spaced `stride` bytes apart (see Engine::ensure_workspace).
__global__ __launch_bounds__(256) void zero_guards(char* first_guard, long long stride, long long count)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x; // one 16-byte piece per thread
    if (i >= count * 4) return;
    *reinterpret_cast<uint4*>(first_guard + (i >> 2) * stride + (i & 3) * 16) = make_uint4(0u, 0u, 0u, 0u);
}

void launch_zero_guards(void* first_guard, long long stride, long long count, hipStream_t st)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(zero_guards, dim3((unsigned)((count * 4 + 255) / 256)), dim3(256), 0, st, static_cast<char*>(first_guard), stride, count);
}

} // namespace rsr
