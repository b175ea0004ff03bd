// kernels.h -- launch interface of the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rsr {

// ---- activation storage ("planes") -----------------------------------------------------------
// Every feature tensor lives in HBM as a set of 16-channel planes: plane = [guard 64 B | [H][W][16] fp16]
// (32 B per pixel, pixels row-major, no border).  A 64-channel tensor = 4 planes, the 192-channel
// dense-block working set = 12 planes.  Tiles of one batch occupy "slots" at a fixed stride.
//
// Work decomposition of the conv kernel: one workgroup computes a 16-row x 32-column block of output
// pixels for all output channels; work items (slot, y0, x0) come from a device table.

constexpr int kBlkH = 16, kBlkW = 32;          // output block of one workgroup
constexpr int kPatchH = kBlkH + 2, kPatchW = kBlkW + 2;
constexpr int kPatchPx = kPatchH * kPatchW;     // 612
constexpr int kGuard = 64;                      // zero bytes in front of every conv-input plane (LDS-DMA source for padding)
constexpr int kPlaneCh = 16;                    // channels per plane

// FOLDED blocks.  A tile whose width leaves a remainder of 1..kFoldMaxW columns behind its last full 32-column block would spend a
// whole block (32-pixel MFMA rows) on those few columns: C3's 420-wide tiles 14 block columns for 13.1, +6.7 % executed matrix
// work.  Such a last column is covered by folded work items instead: ONE block = that column over TWO block rows (rows y0..y0+15
// in lanes / patch columns 0..15, rows y0+16..y0+31 in 16..31).  The matrix loop is unchanged -- lanes never look across pixels --
// only the loaders' gather and the epilogue's scatter know (conv_flow.hip).  WorkItem::x0 carries the flag in bit 30.
constexpr int kFoldBit = 1 << 30;
constexpr int kFoldMaxW = 14; // strip = left halo + <= 14 pixels + right halo = 16 patch columns

// MERGED batches: the tiles of up to kMaxMerge small images of one geometry, handed in by concurrent rsr_process calls, walk the
// network as ONE batch (engine.cpp: Combiner) -- a 256 x 256 image is 200 blocks for 256 CUs and costs as much per launch as four of
// them.  Every image keeps its own device buffers: the kernels that touch images get the pointers by value and the tile's image index.
constexpr int kMaxMerge = 16;

struct WorkItem // 32 bytes: one aligned 2 x 16-byte fetch gives a workgroup everything about its block
{
    int slot, y0, x0; // tile slot and block origin at this table's resolution level (x0 | kFoldBit: a folded block)
    int H, W;         // dims of the slot's tile at this level (output dims of the conv)
    // 4x-level items of a non-TTA batch (conv_last writes the image itself): image coordinates of tile pixel (0,0) (= out_x - crop,
    // out_y - crop; may be negative) and pad2 = item_pad2(out_w, out_h, image): the size of the tile's un-padded output rectangle
    // (13 bits each) and the index of the tile's image in a merged batch (6 bits)
    int pad0, pad1, pad2;
};
__host__ __device__ inline int item_pad2(int out_w, int out_h, int img) { return out_w | ((img & 7) << 13) | (out_h << 16) | int((unsigned)(img >> 3) << 29); }
__host__ __device__ inline int pad2_w(int p) { return p & 0x1fff; }
__host__ __device__ inline int pad2_h(int p) { return (p >> 16) & 0x1fff; }
__host__ __device__ inline int pad2_img(int p) { return ((p >> 13) & 7) | int(((unsigned)p >> 29) << 3); }

struct TileDim
{
    int h, w; // LR dims of the (padded) tile living in this slot
};

struct PlaneSrc
{
    const void* base;     // plane 0 of slot 0
    long long slot_stride;  // bytes between slots
    long long plane_stride; // bytes between planes
};

struct ConvArgs
{
    // input planes: the first n0 from src0, then n1 from src1 (dense-block x planes + x1..x4 planes)
    PlaneSrc src0, src1;
    int n0, n1;
    int lvl_in;  // input resolution = LR << lvl_in
    int lvl_out; // output resolution = LR << lvl_out (lvl_out == lvl_in + 1 for the nearest-x2 fused convs)
    // weights: packed LDS images [plane of 16 cin][9 taps][NT*32 cout][16 cin]; bias fp32 [NT*32]
    const void* wpk16;
    const void* waux; // conv_last only: [plane of 16 cin][dx 0..2][row = dy*8 + cout][16 cin] (PackedConv::aux_off), or null
    int wpieces;      // 1-KiB pieces per plane of the resident image (set by launch_conv_flow)
    const float* bias;
    int lrelu; // LeakyReLU(0.2) on (acc + bias)
    // Output pixels closer than `margin` to the tile border (at this conv's output level) are never read by anything that
    // reaches the cropped output rectangle (engine.cpp: tail_margins): MFMA waves whose four rows lie wholly inside that frame
    // skip their block like rows below the tile.  0 = every pixel is needed.
    int margin;
    int pr; // patch ring depth of a resident-weight launch (set by launch_conv_flow)
    // residual stages: v = v*s + r  (r = fp16 planes)
    PlaneSrc res1, res2;
    int res1_kind, res2_kind; // 0 none, 1 fp16 planes
    float s1, s2;
    int res1_in_acc;  // res1 == the first planes of this conv's own input: added as an identity tap on the matrix pipe, coefficient
    float res1_coef;  // 1/s1 (must be exact in fp16)
    // outputs (any may be null)
    PlaneSrc out16;  // fp16 planes (2*NT planes)
    // conv_last: planar fp16 [3][H][W] per slot
    void* out_planar3;
    long long planar3_slot_stride; // bytes
    // conv_last fused with realsr_postproc.comp (non-TTA RGB, conv3x3_flow): uint8 HWC image, row pitch out_u8_w pixels; the
    // work items of the launch carry the tile's placement (WorkItem::pad0..2, see engine.cpp make_items)
    uint8_t* out_u8;
    int out_u8_w, out_u8_crop, out_u8_bgr; // crop = prepadding * scale; bgr: channel 0 <-> 2 on store
    // work
    const WorkItem* items;
    int nitems;
    const TileDim* dims;
    const void* zeros; // >= 16 zero bytes in device memory (LDS-DMA source for out-of-image pixels)
    unsigned long long* trace; // optional: block 0 / wave 0 writes per-stage s_memtime stamps (profiling aid), 2 x u64 per stage
    int dbg;           // ablation switches for profiling: 1 skip DMA, 2 skip MFMA, 4 skip epilogue stores, ...  (realsr_hip.h)
    // PRECISE residual stream (engine option "precise"; EPI 4 / 5 of conv3x3_flow).  The reference's Vulkan path rounds the 64-channel
    // trunk to fp16 after every RDB / RRDB (fp16 storage, realsr.cpp:44-46); the bar is its fp32 CPU path (realsr.cpp:525-838).  Here
    // a trunk value v lives as hi = fp16(v) -- the plane every conv reads, unchanged -- and lo = bf8((v - hi) * 2048): the rounding
    // residue as ONE byte (e5m2, i.e. the upper byte of an fp16; scaled so that it never becomes subnormal), 5 more bits of v than
    // hi alone -- as good as an fp16 residue for this network (profiles/r06_storage_emulation.txt).  The residual adds of the
    // epilogue use hi + lo / 2048.  The lo bytes of a tensor sit in the same allocation as its hi planes, a fixed number of bytes behind
    // plane 0 (same slot stride): only that distance travels; 0 = the tensor has no lo planes.  Layout: the two planes of an n-tile (32
    // channels) share one "pair plane" of the hi geometry (32 B per pixel: [half of the 16 channels][plane][8 channels]), so that a lane
    // of the epilogue reads / writes its 16 lo bytes of a row with ONE 1-KiB-per-wave access at the very offsets of its hi stores
    // (conv_flow.hip lo_row).  A 64-channel tensor: two pair planes = the room of two hi planes.
    int precise;          // 1: residual forms run the precise epilogue (out16 single-rounded from the fp32 value)
    long long lo1_off;    // lo planes of res1 = res1 + lo1_off bytes
    long long lo2_off;    // lo planes of res2
    long long out_lo_off; // lo planes of the output = out16 + out_lo_off (0: not kept)
    // fused conv_last: the uint8 image of every image of a (merged) batch and its row pitch in pixels (the images of a merged batch may
    // differ in size); out_u8 == out_u8s[0] doubles as the mode flag
    uint8_t* out_u8s[kMaxMerge];
    int out_u8_ws[kMaxMerge];
};

// conv_flow.hip: half-stage ring on 16-channel planes.  flags: 1 = two n-tiles per MFMA wave for 64-cout convs, 2 = no deferred epilogue,
// 4 = weights always streamed (never LDS-resident), 8 = conv_last through the generic path.
// false = this combination of outputs / residuals is not covered (the engine then reports an error)
bool launch_conv_flow(const ConvArgs& a, int nt, int ncu, int flags, hipStream_t st);
// Opt-in to > 64 KiB of dynamic LDS for every kernel instantiation, on the CURRENT device (call once per device).
hipError_t kernels_init_device();
hipError_t flow_init_device();

// ---- pre / post ------------------------------------------------------------------------------
struct BaseTile
{
    int x_org, y_org; // image coords of padded-tile pixel (0,0)  (= xi*T - P, yi*T - P)
    int tw, th;       // padded tile size
    int slot0;        // first slot (TTA: 8 consecutive slots)
    int out_x, out_y; // top-left of this tile's output rectangle in the x4 image
    int out_w, out_h; // size of that rectangle (tile_w_nopad*4, tile_h_nopad*4)
    int img;          // image of a merged batch this tile belongs to (index into PreArgs::imgs / PostArgs::outs; 0 otherwise)
};

struct PreArgs
{
    const uint8_t* imgs[kMaxMerge]; // HWC u8, one per image of the batch (ws[i] x hs[i] x c); BaseTile::img selects
    int ws[kMaxMerge], hs[kMaxMerge];
    int nimgs;
    int c;
    const BaseTile* tiles;
    int ntiles;
    int tta;
    void* in_plane; // fp16 plane [th][tw][plane_ch] per slot, channels 0..2 = RGB/255, rest 0
    long long slot_stride;
    int bgr;
    int plane_ch;   // 16
    int variant;    // 0 default (launch_preproc_tiles picks), 1 one thread per pixel (engine dbg 32768), 2 LDS-staged (dbg 65536)
};
void launch_preproc_tiles(const PreArgs& a, int max_tw, int max_th, hipStream_t st);

struct PostArgs
{
    const void* planar3; // fp16 (f32: fp32) [3][4th][4tw] per slot
    int f32;             // 1: precise mode, conv_last left its fp32 result (ConvArgs::precise)
    long long slot_stride;
    const BaseTile* tiles;
    int ntiles;
    int tta;
    int crop;   // prepadding*scale
    uint8_t* outs[kMaxMerge]; // HWC u8 (4w x 4h x c), one per image of the batch
    int out_ws[kMaxMerge];    // their row pitches in pixels (4w)
    int in_ws[kMaxMerge];     // ... and those of the source images (w)
    int nimgs;
    int c;
    int out_row0; // the outs point at output row out_row0 of the x4 image (a tile range's device buffer holds only its rows)
    const uint8_t* in_imgs[kMaxMerge]; // for alpha (c==4): the source images (w x h x 4)
    int tilesize;
    int bgr;
    int variant; // 0 default (LDS-staged under TTA, else one thread per pixel), 1 per-pixel (engine dbg 32768), 2 LDS-staged (dbg 65536)
};
void launch_postproc_tiles(const PostArgs& a, int max_ow, int max_oh, hipStream_t st);

// shader-shaped standalone kernels (parity tests): device pointers
void launch_preproc_shader(const uint8_t* bottom, int w, int h, int channels, uint16_t* const top[8], int ntop, int outw,
                           int outh, int outcstep, int pad_top, int pad_left, int crop_x, int crop_y, uint16_t* alpha,
                           int alphaw, int alphah, int bgr, hipStream_t st);
void launch_postproc_shader(const uint16_t* const bottom[8], int nbottom, int w, int h, int cstep, const uint16_t* alpha,
                            int alphaw, int alphah, uint8_t* top, int outw, int outh, int offset_x, int gx_max, int crop_x,
                            int crop_y, int channels, int bgr, hipStream_t st);

// planar fp16 [3][h][w]  ->  plane [h][w][plane_ch] (net_forward test hook)
void launch_planar3_to_plane(const uint16_t* planar, int w, int h, void* plane, int plane_ch, hipStream_t st);
// zero the 64-byte guards of `count` planes, `stride` bytes apart
void launch_zero_guards(void* first_guard, long long stride, long long count, hipStream_t st);

} // namespace rsr
