// kernels.hip -- gfx950 (CDNA4 / MI355X) pre/post kernels of the RealSR x4 hot path.
//
//   preproc_tiles[_lds]    realsr_preproc{,_tta}.comp equivalent, writes the network input planes
//   postproc_tiles[_lds]   realsr_postproc{,_tta}.comp equivalent, writes the uint8 HWC image
//                          (_lds: rows staged in LDS, dword loads / 1-KiB stores, transposed TTA variants through an LDS tile;
//                          chosen per launch by measurement: launch_*_tiles)
//   *_shader               the same arithmetic in the shaders' own memory layout (parity tests)
//
// The 351 convolutions live in conv_flow.hip (conv3x3_flow).  Written for gfx950 only.
#include "kernels.h"

namespace rsr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// > 64 KiB of dynamic LDS needs an opt-in per kernel AND per device (a process-wide "done" flag would leave the
// second GPU of a multi-device process without it): the engine calls this once per context, on its own device.
hipError_t kernels_init_device() { return flow_init_device(); }

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
    const int im = __builtin_amdgcn_readfirstlane(t.img), iw = a.ws[im], ih = a.hs[im];
    const int x = reflect101(gx + t.x_org, iw);
    const int y = reflect101(gy + t.y_org, ih);
    const uint8_t* p = a.imgs[im] + ((long long)y * iw + x) * a.c;
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

// The same arithmetic with the memory traffic of realsr_preproc{,_tta}.comp reorganised for HBM (round 4; byte-identical to the
// one-thread-per-pixel kernel above, which stays the default -- see launch_preproc_tiles):
//   * one workgroup = a 32 x 32 block of padded-tile pixels.  Its source rows are staged in LDS with aligned DWORD loads, one
//     wave per row (reflection folds the 32 columns of a block onto at most 32 source columns, i.e. <= 132 contiguous bytes);
//   * the pixels are converted once (uint8 -> fp16 / 255) into an LDS tile [32][33] of (r, g, b) halfs;
//   * every TTA variant is then written row by row of ITS OWN orientation: a wave stores 32 pixels x 32 B = 1 KiB of whole
//     cache lines per instruction (lane = (pixel, 16-byte half)) -- also for the four transposed variants, whose rows run down
//     the tile's columns (column reads of the LDS tile are conflict-free through the 33-pixel pitch).
__global__ __launch_bounds__(256) void preproc_tiles_lds(const PreArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char raw[32][144];
    __shared__ uint2 px[32][33];
    const BaseTile t = a.tiles[blockIdx.z];
    const int gx0 = blockIdx.x * 32, gy0 = blockIdx.y * 32;
    if (gx0 >= t.tw || gy0 >= t.th) return;
    const int im = __builtin_amdgcn_readfirstlane(t.img), iw = a.ws[im], ih = a.hs[im];
    const uint8_t* const img = a.imgs[im];
    const int nx = min(32, t.tw - gx0), ny = min(32, t.th - gy0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // source column span [lo, hi] of the block's columns (every wave computes it for itself)
    int xs = reflect101(gx0 + min(lane & 31, nx - 1) + t.x_org, iw);
    int lo = xs, hi = xs;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1)
    {
        lo = min(lo, __shfl_xor(lo, d));
        hi = max(hi, __shfl_xor(hi, d));
    }
    const long long total = (long long)iw * ih * a.c;
    for (int r = wave; r < ny; r += 4)
    {
        const int y = reflect101(gy0 + r + t.y_org, ih);
        const long long b0 = ((long long)y * iw + lo) * a.c, b1 = ((long long)y * iw + hi + 1) * a.c;
        const long long a0 = b0 & ~3ll;
        const int nd = int((b1 - a0 + 3) >> 2); // <= 33 dwords
        if (lane < nd)
        {
            const long long ad = a0 + 4 * lane;
            uint32_t v;
            if (ad + 4 <= total) v = *reinterpret_cast<const uint32_t*>(img + ad);
            else
            { // the last bytes of the image: never read past its end
                v = 0;
                for (int e = 0; e < 4; e++)
                    if (ad + e < total) v |= (uint32_t)img[ad + e] << (8 * e);
            }
            *reinterpret_cast<uint32_t*>(&raw[r][4 * lane]) = v;
        }
    }
    __syncthreads();
    const float norm_val = 1 / 255.f;
    const int i0 = a.bgr ? 2 : 0, i2 = a.bgr ? 0 : 2;
#pragma unroll
    for (int m = 0; m < 4; m++)
    {
        const int r = (tid >> 5) + 8 * m, cx = tid & 31;
        if (r < ny && cx < nx)
        {
            const int x = reflect101(gx0 + cx + t.x_org, iw);
            const int y = reflect101(gy0 + r + t.y_org, ih);
            const int shift = int((((long long)y * iw + lo) * a.c) & 3);
            const unsigned char* p = &raw[r][shift + (x - lo) * a.c];
            const _Float16 hr = (_Float16)((float)p[i0] * norm_val), hg = (_Float16)((float)p[1] * norm_val), hb = (_Float16)((float)p[i2] * norm_val);
            uint2 v;
            v.x = (uint32_t)__builtin_bit_cast(unsigned short, hr) | ((uint32_t)__builtin_bit_cast(unsigned short, hg) << 16);
            v.y = (uint32_t)__builtin_bit_cast(unsigned short, hb);
            px[r][cx] = v;
        }
    }
    __syncthreads();
    const int nv = a.tta ? 8 : 1;
    const int pi = lane >> 1, half = lane & 1; // lane = (pixel of the output row, 16-byte half of its 32 B)
    for (int k = 0; k < nv; k++)
    {
        const bool tr = k >= 4;                 // transposed variants: an output row runs down a tile column
        const int nrow = tr ? nx : ny, ncol = tr ? ny : nx;
        char* const base = static_cast<char*>(a.in_plane) + (long long)(t.slot0 + k) * a.slot_stride;
        for (int ro = wave; ro < nrow; ro += 4)
        {
            if (pi >= ncol) continue;
            const int r = tr ? pi : ro, cx = tr ? ro : pi;
            const int gx = gx0 + cx, gy = gy0 + r;
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
            const uint2 v = px[r][cx];
            const uint4 o = half ? make_uint4(0u, 0u, 0u, 0u) : make_uint4(v.x, v.y, 0u, 0u);
            *reinterpret_cast<uint4*>(base + ((long long)oy * ow + ox) * 32 + half * 16) = o;
        }
    }
}

void launch_preproc_tiles(const PreArgs& a, int max_tw, int max_th, hipStream_t st)
{
    if (a.ntiles <= 0) return;
    // Measured (tools/prepost_perf.py, profiles/r04_prepost.txt): 0.042 ms LDS-staged vs 0.032 ms per-pixel on a 1080p frame, 0.145 vs
    // 0.150 ms with the 8 TTA scatters -- the byte loads of the plain kernel are served by the caches, its 32-byte stores are whole
    // sectors: staging buys nothing here.  Default = the plain kernel; variant 2 forces the staged one (tests, A/B).
    bool aligned = true; // (the staged kernel reads the images in dwords)
    for (int i = 0; i < a.nimgs; i++) aligned = aligned && !(reinterpret_cast<uintptr_t>(a.imgs[i]) & 3);
    if (a.variant != 2 || a.plane_ch != 16 || !aligned)
    {
        const dim3 grid((max_tw + 31) / 32, (max_th + 7) / 8, a.ntiles), block(256);
        hipLaunchKernelGGL(preproc_tiles, grid, block, 0, st, a);
        return;
    }
    const dim3 grid((max_tw + 31) / 32, (max_th + 31) / 32, a.ntiles), block(256);
    hipLaunchKernelGGL(preproc_tiles_lds, grid, block, 0, st, a);
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
template <typename TP> // element type of the planar blob: _Float16 (the reference's `output` blob) or float (precise mode)
__global__ __launch_bounds__(256) void postproc_tiles(const PostArgs a)
{
    const BaseTile t = a.tiles[blockIdx.z];
    const int im = __builtin_amdgcn_readfirstlane(t.img);
    const int gx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int gy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (gx >= t.out_w || gy >= t.out_h) return;
    const int w = t.tw * 4, h = t.th * 4;
    const long long cstep = (long long)w * h;
    const int sx = gx + a.crop, sy = gy + a.crop;
    uint8_t* o = a.outs[im] + ((long long)(t.out_y - a.out_row0 + gy) * a.out_ws[im] + t.out_x + gx) * a.c;
    const TP* b0 = reinterpret_cast<const TP*>(static_cast<const char*>(a.planar3) + (long long)t.slot0 * a.slot_stride);
    float v[3];
    if (!a.tta)
    {
#pragma unroll
        for (int q = 0; q < 3; q++) v[q] = (float)b0[q * cstep + (long long)sy * w + sx];
    }
    else
    {
        const long long ss = a.slot_stride / (long long)sizeof(TP);
#pragma unroll
        for (int q = 0; q < 3; q++)
        {
            const TP* b = b0 + q * cstep;
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
            const uint8_t* rp = a.in_imgs[im] + ((long long)yy * a.in_ws[im] + ax0) * 4 + 3;
            rows[j] = (float)rp[clampi(bx - 1, aw) * 4] * cx[0] + (float)rp[clampi(bx, aw) * 4] * cx[1] +
                      (float)rp[clampi(bx + 1, aw) * 4] * cx[2] + (float)rp[clampi(bx + 2, aw) * 4] * cx[3];
        }
        const float av = rows[0] * cy[0] + rows[1] * cy[1] + rows[2] * cy[2] + rows[3] * cy[3];
        o[3] = post_store(av);
    }
}

// realsr_postproc{,_tta}.comp with coalesced traffic (round 4; byte-identical to the kernel above; the default under TTA):
// one workgroup = a 32 x 32 block of the tile's kept output rectangle, 4 pixels per thread.  The four plain TTA variants are
// read along their rows (lanes along x, forwards or backwards: contiguous either way).  The four TRANSPOSED variants store pixel
// (sx, sy) at [sx][sy]: read naively, a wave touches 64 different cache lines for 128 useful bytes -- here each (variant,
// channel) block is read along ITS rows into an LDS tile and picked up transposed (34-half pitch: conflict-free).  The merge keeps
// the shader's summation order (v0 + v1 + ... + v7) * 0.125.  The uint8 pixels are collected in LDS and leave as aligned dwords
// (a 32-pixel row segment of the HWC image = 96 or 128 contiguous bytes).
template <typename TP>
__global__ __launch_bounds__(256) void postproc_tiles_lds(const PostArgs a)
{
    __shared__ TP T[32][sizeof(TP) == 2 ? 34 : 33]; // pitch: conflict-free column reads for 2- and 4-byte elements
    __shared__ __attribute__((aligned(16))) unsigned char ob[32][128];
    const BaseTile t = a.tiles[blockIdx.z];
    const int im = __builtin_amdgcn_readfirstlane(t.img);
    const int gx0 = blockIdx.x * 32, gy0 = blockIdx.y * 32;
    if (gx0 >= t.out_w || gy0 >= t.out_h) return;
    const int nx = min(32, t.out_w - gx0), ny = min(32, t.out_h - gy0);
    const int tid = threadIdx.x, lx = tid & 31, ly = tid >> 5;
    const int w = t.tw * 4, h = t.th * 4;
    const long long cstep = (long long)w * h;
    const TP* b0 = reinterpret_cast<const TP*>(static_cast<const char*>(a.planar3) + (long long)t.slot0 * a.slot_stride);
    const long long ss = a.slot_stride / (long long)sizeof(TP);
    float acc[4][3];
    const int sx = gx0 + lx + a.crop;
#pragma unroll
    for (int m = 0; m < 4; m++)
    {
        const int gyl = ly + 8 * m, sy = gy0 + gyl + a.crop;
        const bool ok = lx < nx && gyl < ny;
#pragma unroll
        for (int q = 0; q < 3; q++)
        {
            float v = 0.f;
            if (ok)
            {
                const TP* b = b0 + q * cstep;
                v = (float)b[(long long)sy * w + sx];
                if (a.tta)
                { // realsr_postproc_tta.comp:76-79
                    v += (float)b[ss + (long long)sy * w + (w - 1 - sx)];
                    v += (float)b[2 * ss + (long long)(h - 1 - sy) * w + (w - 1 - sx)];
                    v += (float)b[3 * ss + (long long)(h - 1 - sy) * w + sx];
                }
            }
            acc[m][q] = v;
        }
    }
    if (a.tta)
    {
        for (int k = 4; k < 8; k++)
            for (int q = 0; q < 3; q++)
            {
                __syncthreads(); // the previous tile has been consumed
                const TP* b = b0 + k * ss + q * cstep;
#pragma unroll
                for (int m = 0; m < 4; m++)
                {
                    const int ai = ly + 8 * m, bi = lx; // local (sx, sy) of the element this thread fetches: lanes along sy = along the row
                    if (ai < nx && bi < ny)
                    {
                        const int ex = gx0 + ai + a.crop, ey = gy0 + bi + a.crop;
                        long long off; // realsr_postproc_tta.comp:80-83
                        if (k == 4) off = (long long)ex * h + ey;
                        else if (k == 5) off = (long long)ex * h + (h - 1 - ey);
                        else if (k == 6) off = (long long)(w - 1 - ex) * h + (h - 1 - ey);
                        else off = (long long)(w - 1 - ex) * h + ey;
                        T[ai][bi] = b[off];
                    }
                }
                __syncthreads();
#pragma unroll
                for (int m = 0; m < 4; m++)
                {
                    const int gyl = ly + 8 * m;
                    if (lx < nx && gyl < ny) acc[m][q] += (float)T[lx][gyl];
                }
            }
    }
#pragma unroll
    for (int m = 0; m < 4; m++)
    {
        const int gyl = ly + 8 * m;
        if (!(lx < nx && gyl < ny)) continue;
        float v[3];
#pragma unroll
        for (int q = 0; q < 3; q++) v[q] = a.tta ? acc[m][q] * 0.125f : acc[m][q];
        unsigned char* o = &ob[gyl][lx * a.c];
        o[a.bgr ? 2 : 0] = post_store(v[0] * 255.f);
        o[1] = post_store(v[1] * 255.f);
        o[a.bgr ? 0 : 2] = post_store(v[2] * 255.f);
        if (a.c == 4)
        { // alpha: bicubic x4 of the un-padded tile's alpha, exactly as in postproc_tiles
            const int gx = gx0 + lx, gy = gy0 + gyl;
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
                const uint8_t* rp = a.in_imgs[im] + ((long long)yy * a.in_ws[im] + ax0) * 4 + 3;
                rows[j] = (float)rp[clampi(bx - 1, aw) * 4] * cx[0] + (float)rp[clampi(bx, aw) * 4] * cx[1] +
                          (float)rp[clampi(bx + 1, aw) * 4] * cx[2] + (float)rp[clampi(bx + 2, aw) * 4] * cx[3];
            }
            o[3] = post_store(rows[0] * cy[0] + rows[1] * cy[1] + rows[2] * cy[2] + rows[3] * cy[3]);
        }
    }
    __syncthreads();
    const int nd = nx * a.c / 4; // out_w, gx0 are multiples of 4: a row segment is whole dwords, 4-byte aligned in the image
    for (int i = tid; i < ny * nd; i += 256)
    {
        const int r = i / nd, d = i - r * nd;
        uint8_t* o = a.outs[im] + ((long long)(t.out_y - a.out_row0 + gy0 + r) * a.out_ws[im] + t.out_x + gx0) * a.c;
        reinterpret_cast<uint32_t*>(o)[d] = reinterpret_cast<const uint32_t*>(&ob[r][0])[d];
    }
}

void launch_postproc_tiles(const PostArgs& a, int max_ow, int max_oh, hipStream_t st)
{
    if (a.ntiles <= 0) return;
    // Measured (profiles/r04_prepost.txt): the TTA gather 0.92 ms staged vs 2.42 ms per-pixel on the C5 frame (2.26 vs 0.86 TB/s: the
    // transposed variants), but 0.19 vs 0.12 ms for the plain single-variant conversion.  Default: staged under TTA, plain otherwise.
    const bool staged = a.variant == 2 || (a.variant == 0 && a.tta);
    bool aligned = true; // (the staged kernel stores the image in dwords)
    for (int i = 0; i < a.nimgs; i++) aligned = aligned && !(reinterpret_cast<uintptr_t>(a.outs[i]) & 3);
    if (!staged || !aligned)
    {
        const dim3 grid((max_ow + 63) / 64, (max_oh + 3) / 4, a.ntiles), block(256);
        if (a.f32) hipLaunchKernelGGL(postproc_tiles<float>, grid, block, 0, st, a);
        else hipLaunchKernelGGL(postproc_tiles<_Float16>, grid, block, 0, st, a);
        return;
    }
    const dim3 grid((max_ow + 31) / 32, (max_oh + 31) / 32, a.ntiles), block(256);
    if (a.f32) hipLaunchKernelGGL(postproc_tiles_lds<float>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(postproc_tiles_lds<_Float16>, grid, block, 0, st, a);
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
