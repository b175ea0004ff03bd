/*
 * realsr_hip.h -- C-ABI of the MI355X-native RealSR x4 tiled-inference engine (librealsr_hip.so).
 *
 * This is the drop-in boundary for the reference's `class RealSR`
 * (/root/reference/src/realsr.h:13-42), the only seam between the CLI/orchestration
 * (/root/reference/src/main.cpp) and the compute path.  Plain pointers and sizes only; no C++,
 * ncnn or torch types.  Every entry point cites the reference interface it replaces.
 *
 * Conventions
 *   - return 0 = ok, negative = error (RSR_E_*); rsr_last_error() gives a message.  (The reference
 *     returns 0 unconditionally and its callers ignore the value -- main.cpp:786,325 -- so this is a
 *     strict superset.)
 *   - images are HWC uint8, tightly packed, c in {3,4}, RGB(A) order (main.cpp:275-276); the output is
 *     caller-allocated (w*scale) x (h*scale) x c.  The engine never retains either pointer.
 *   - one context per GPU (main.cpp:778-791); rsr_process* are thread-safe on a shared context
 *     (the reference calls process() concurrently from jobs_proc threads, main.cpp:811-828): every
 *     call owns private device image buffers, pinned staging and a copy stream ("lane", up to
 *     max_lanes in flight, further callers wait); the network kernels of all calls are queued on one
 *     compute stream, so upload(k+1) | kernels(k) | download(k-1) overlap; SMALL images of concurrent
 *     calls are merged into one tile batch (option "merge"; the reference's "-j 4:4:4 for many small
 *     images", README.md:61) -- same bytes, shared launches.
 *   - rsr_last_error() reports the CALLING THREAD's most recent failure.
 *   - there is NO CPU fallback: gpuid must name a HIP device; the reference's "-g -1" CPU path
 *     (RealSR::process_cpu) lives only in oracle/ as the parity checker.
 */
#ifndef REALSR_HIP_H
#define REALSR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSR_OK 0
#define RSR_E_ARG (-1)      /* bad argument */
#define RSR_E_IO (-2)       /* cannot open / truncated file */
#define RSR_E_FORMAT (-3)   /* .param/.bin not parseable */
#define RSR_E_GRAPH (-4)    /* parsed graph is not the RRDBNet(3,3,64,23,32) x4.param describes */
#define RSR_E_DEVICE (-5)   /* HIP error / no such device */
#define RSR_E_STATE (-6)    /* call order (e.g. process before load) */
#define RSR_E_NOMEM (-7)

typedef struct rsr_ctx rsr_ctx;

/* ---- lifecycle -------------------------------------------------------------------------- */

/* RealSR::RealSR(int gpuid, bool tta_mode, int num_threads)   realsr.h:16, realsr.cpp:13-23.
 * gpuid: HIP device ordinal (>= 0).  num_threads is accepted for signature parity and ignored
 * (it only sizes ncnn's CPU thread pool in the reference, realsr.cpp:17). */
int rsr_create(rsr_ctx** out, int gpuid, int tta_mode, int num_threads);

/* RealSR::~RealSR   realsr.cpp:25-35 */
void rsr_destroy(rsr_ctx* ctx);

/* RealSR::load(parampath, modelpath)   realsr.h:19-23, realsr.cpp:37-143.
 * Parses the ncnn text graph and tagged weight stream (fp16-tagged or raw fp32), checks that the
 * DAG is exactly the x4.param RRDBNet, packs the weights for the MFMA kernels and uploads them. */
int rsr_load(rsr_ctx* ctx, const char* parampath, const char* modelpath);

/* The public mutable fields `scale`, `tilesize`, `prepadding` assigned after load()
 * (realsr.h:31-33, main.cpp:788-790).  scale must be 4 (main.cpp:533-537); tilesize >= 32
 * (main.cpp:539-552; smaller values are accepted for tests, >= 1); prepadding >= 0 (10 for
 * models-DF2K*, main.cpp:663-667). */
int rsr_set_params(rsr_ctx* ctx, int scale, int tilesize, int prepadding);

/* ---- the hot path ----------------------------------------------------------------------- */

/* RealSR::process(const ncnn::Mat& in, ncnn::Mat& out) const   realsr.h:25, realsr.cpp:145-523.
 * `in`/`out` are HOST pointers (what ncnn::Mat::data is at main.cpp:275-276).  H2D, all tiles,
 * D2H; returns when `out` is complete. */
int rsr_process(rsr_ctx* ctx, const uint8_t* in, int w, int h, int c, uint8_t* out);

/* n images in one call (no reference counterpart: main.cpp's proc threads call process() once per image, :311-331).  Every image is
 * what rsr_process would make of it; up to max_lanes of them are in flight at once on helper threads of the call, so that SMALL
 * images share tile batches (option "merge") without the host having to be multi-threaded.  rcs (may be NULL) receives the code of
 * every image; the return value is the first failure (0 = all ok). */
int rsr_process_many(rsr_ctx* ctx, int n, const uint8_t* const* in, const int* w, const int* h, const int* c, uint8_t* const* out, int* rcs);

/* Same computation with both images already resident in this context's device memory
 * (what the reference keeps in VkMat in_gpu/out_gpu, realsr.cpp:211-233, minus the PCIe hops).
 * `stream` is a hipStream_t (NULL = the context's own stream).  Asynchronous when a stream is
 * given: the caller synchronises.  When nothing else is in flight on the context the kernels are
 * enqueued on `stream` itself; otherwise they run on the context's compute stream, ordered behind
 * the work already on `stream` and in front of what the caller enqueues on it next. */
int rsr_process_device(rsr_ctx* ctx, const void* d_in, int w, int h, int c, void* d_out, void* stream);

/* Pinned host memory for images.  rsr_process copies pinned buffers (these, hipHostMalloc'd or hipHostRegister'ed
 * memory) to / from the GPU directly; pageable memory goes through the call's pinned staging (the download in chunks,
 * the CPU copy of one chunk under the PCIe transfer of the next).  The reference gets the same from ncnn's Vulkan
 * staging allocator (realsr.cpp:161-167, 208-220). */
void* rsr_host_alloc(size_t bytes);
void rsr_host_free(void* p);

/* Free / total device memory of HIP device `gpuid` in MiB: what ncnn's VulkanDevice::get_heap_budget() is to the
 * reference's automatic tile-size policy (main.cpp:761-774). */
int rsr_device_memory(int gpuid, long long* free_mb, long long* total_mb);

/* The reference prints one line per tile to stderr (realsr.cpp:481).  Here all tiles of a batch run together: `cb` is
 * called once per TILE (tiles_done = 1 .. tiles_total, from the calling thread) when the tile's batch has been enqueued --
 * kernels not necessarily finished; the calls of one batch arrive back to back. */
int rsr_set_progress_callback(rsr_ctx* ctx, void (*cb)(int tiles_done, int tiles_total, void* user), void* user);

/* ---- weights as one relocatable blob (multi-GPU load path) -------------------------------- */
/* The reference re-reads x4.bin once per GPU (main.cpp:784-786).  Here rank 0 parses and packs once,
 * the blob travels by a single RCCL broadcast over xGMI (done by the caller, e.g.
 * torch.distributed.broadcast with backend nccl), and every rank loads it with rsr_load_packed. */

/* Host-only: parse + validate + pack into `dst` (capacity `cap` bytes).  *need receives the blob
 * size (33.5 MB for x4.param); call with dst=NULL to query.  No GPU required. */
int rsr_model_pack(const char* parampath, const char* modelpath, void* dst, size_t cap, size_t* need);

/* Load a blob produced by rsr_model_pack.  `blob` may be a host pointer (is_device=0) or a device
 * pointer on this context's GPU (is_device=1). */
int rsr_load_packed(rsr_ctx* ctx, const void* blob, size_t bytes, int is_device);

/* ---- several GPUs of one node ------------------------------------------------ */
/* What main.cpp:778-791 does per GPU (construct, load), for n GPUs at once: the model is parsed and packed once, uploaded to
 * gpuids[0] and broadcast to the other devices with ONE RCCL collective over xGMI (librccl is dlopen'ed; when it is not
 * available the blob is uploaded to every device from the host instead -- rsr_group_transport() tells which).  out[0..n-1]
 * receive the contexts (all NULL on failure).  Images are then dealt to the contexts by the caller's work queue
 * (main.cpp:811-828), or ONE large image is split with rsr_process_group. */
int rsr_create_group(rsr_ctx** out, const int* gpuids, int n, int tta_mode, const char* parampath, const char* modelpath);
const char* rsr_group_transport(void); /* "rccl" | "host ..." for the calling thread's last rsr_create_group */
/* With RSR_GROUP_FORCE_RCCL=1 in the environment rsr_create_group takes the RCCL branch also for n == 1 (a communicator of one
 * rank, an in-place broadcast, the model loaded from the device copy): the whole collective path can be exercised on a
 * single-GPU machine.  rsr_rccl_probe is host-only (no GPU, no communicator): 0 when librccl can be dlopen'ed and exports every
 * entry point that branch calls (ncclCommInitAll, ncclCommDestroy, ncclGroupStart, ncclGroupEnd, ncclBroadcast,
 * ncclGetErrorString), RSR_E_DEVICE + rsr_last_error() otherwise.  Reference semantics: one RealSR + one load per GPU id,
 * /root/reference/src/main.cpp:778-791. */
int rsr_rccl_probe(void);

/* RealSR::process restricted to the tiles [tile_begin, tile_end) of the image's tile grid, counted row-major (tile (yi, xi) =
 * yi * ceil(w / tilesize) + xi; tiles are independent: realsr.cpp:377-380,458-459,490).  `in` is the whole image, `out` the
 * whole (4w x 4h x c) output; only the output rectangles of those tiles are written.  Disjoint ranges may run concurrently
 * on different contexts into the same `out`.  rsr_process_rows: the same for whole tile rows [tile_row_begin, tile_row_end). */
int rsr_process_tiles(rsr_ctx* ctx, const uint8_t* in, int w, int h, int c, uint8_t* out, int tile_begin, int tile_end);
int rsr_process_rows(rsr_ctx* ctx, const uint8_t* in, int w, int h, int c, uint8_t* out, int tile_row_begin, int tile_row_end);

/* One image over n contexts (normally one per GPU): the TILES are dealt in contiguous row-major ranges of equal padded-pixel
 * load (within one tile; a 1080p frame at tile 200 = 60 tiles = 7-8 per GPU on 8 GPUs), every context runs its range on its
 * own thread and fetches the rectangles of its tiles into `out`; the call returns when `out` is complete.  All contexts must
 * carry the same tilesize / prepadding / scale / tta (checked: RSR_E_ARG otherwise). */
int rsr_process_group(rsr_ctx* const* ctx, int n, const uint8_t* in, int w, int h, int c, uint8_t* out);
/* (rsr_process_group keeps its worker threads between calls: a process that has called it must not dlclose() the library.) */

/* Host-only (no GPU): the tile ranges rsr_process_group deals to `parts` shares of a w x h image: share i runs the tiles
 * [bounds[i], bounds[i+1]) of the row-major grid; bounds has parts + 1 entries.  Balanced by padded tile area to within one
 * tile.  Returns the number of shares used (min(parts, number of tiles)) or a negative RSR_E_*. */
int rsr_tile_partition(int w, int h, int tilesize, int prepadding, int parts, int* bounds);

/* Host-only model introspection (no GPU): conv count, weight/bias counts and the .bin encoding
 * (1 = fp16-tagged, 0 = raw fp32, 2 = mixed/table).  Any pointer may be NULL. */
int rsr_model_info(const char* parampath, const char* modelpath, int* n_layers, int* n_convs,
                   long long* n_weights, long long* n_biases, int* bin_encoding);

/* ---- the four shader-equivalent kernels, callable alone (bit-exact parity tests) ----------- */
/* Arguments mirror the shaders' push constants; all pointers are HOST pointers, the call copies
 * in, runs the HIP kernel, copies out, synchronises.
 *
 * rsr_preproc:  realsr_preproc.comp:47-95 / dispatch realsr.cpp:372-415
 *   band  u8 HWC (w x h x channels) -> top: planar fp16 [3][outh][outw] (cstep = outw*outh),
 *   alpha: fp16 [alphah][alphaw] un-normalised (only when channels == 4, may be NULL otherwise).
 * rsr_preproc_tta: realsr_preproc_tta.comp:54-113; top[0..3] are outw x outh, top[4..7] outh x outw. */
int rsr_preproc(rsr_ctx* ctx, const uint8_t* band, int w, int h, int channels, uint16_t* top, int outw,
                int outh, int pad_top, int pad_left, int crop_x, int crop_y, uint16_t* alpha, int alphaw,
                int alphah);
int rsr_preproc_tta(rsr_ctx* ctx, const uint8_t* band, int w, int h, int channels, uint16_t* const top[8],
                    int outw, int outh, int pad_top, int pad_left, int crop_x, int crop_y);

/* rsr_postproc: realsr_postproc.comp:47-89 / dispatch realsr.cpp:444-472
 *   bottom planar fp16 [3][h][w] -> top u8 HWC band (outw x outh x channels), written in place
 *   for gx < gx_max at column offset offset_x (the band is copied in first, so untouched pixels
 *   keep their value).  alpha (channels==4): fp16 [alphah][alphaw], 0..255 units.
 * rsr_postproc_tta: realsr_postproc_tta.comp:54-110. */
int rsr_postproc(rsr_ctx* ctx, const uint16_t* bottom, int w, int h, const uint16_t* alpha, int alphaw,
                 int alphah, uint8_t* top, int outw, int outh, int offset_x, int gx_max, int crop_x, int crop_y,
                 int channels);
int rsr_postproc_tta(rsr_ctx* ctx, const uint16_t* const bottom[8], int w, int h, uint8_t* top, int outw,
                     int outh, int offset_x, int gx_max, int crop_x, int crop_y, int channels);

/* ---- network on one tile (layer-level parity; replaces ncnn::Extractor input/extract,
 *      realsr.cpp:420-428) -------------------------------------------------------------------- */
/* in: planar fp16 [3][h][w] in [0,1]; out: planar fp16 [3][4h][4w].  Host pointers. */
int rsr_net_forward(rsr_ctx* ctx, const uint16_t* in, int w, int h, uint16_t* out);

/* The same with the network's result in fp32 (planar [3][4h][4w]), as conv_last leaves it in precise mode (option "precise" = 1;
 * RSR_E_STATE otherwise): what the uint8 conversion sees there.  rsr_net_forward in precise mode returns this blob rounded to fp16. */
int rsr_net_forward_f32(rsr_ctx* ctx, const uint16_t* in, int w, int h, float* out);

/* One 3x3/s1/p1 convolution through the MFMA kernel with caller-supplied weights (layer-level parity;
 * the arithmetic ncnn::Convolution [+ Interp nearest x2 in front when upsample2x] performs for each
 * Convolution line of x4.param).  in: planar fp16 [cin][h][w]; weight: fp32 OIHW [cout][cin][3][3]
 * (rounded to fp16 by the packer); bias fp32 [cout]; cout <= 64; out: planar fp16 [cout][H][W],
 * H,W = h,w (or 2h,2w).  Host pointers. */
int rsr_conv3x3(rsr_ctx* ctx, const uint16_t* in, int cin, int h, int w, int upsample2x, const float* weight,
                const float* bias, int cout, int lrelu, uint16_t* out);

/* The residual epilogues of the graph on one convolution (what Eltwise / BinaryOp do behind a Convolution in x4.param):
 *   v = s1*(conv + b)  [+ in[0:cout] when own_input_residual: RDB conv5, x4.param:17-18]
 *   [v = s2*v + res when own_input_residual and res: every third RDB, x4.param:47]
 *   [v = v + res when !own_input_residual and res: trunk_conv + global skip, x4.param:994-995]
 * res: planar fp16 [cout][h][w] or NULL; cout 32 or 64; no upsampling, no activation.  Host pointers. */
int rsr_conv3x3_res(rsr_ctx* ctx, const uint16_t* in, int cin, int h, int w, const float* weight, const float* bias, int cout,
                    float s1, int own_input_residual, const uint16_t* res, float s2, uint16_t* out);

/* The same residual forms as the engine runs them in precise mode (option "precise"; 64 output channels): every tensor of the
 * residual stream is hi + lo / 2048 -- hi = fp16(v) as ever, lo = the rounding residue (v - hi) * 2048 as ONE byte (bf8 / e5m2: the
 * upper byte of an IEEE fp16); the adds are done in fp32 and rounded once.  in_lo: lo of in[0:64] (own_input_residual only),
 * res_lo: lo of res, out_lo: receives the lo of the result; any of the three may be NULL (= zero / not wanted).  Planar [64][h][w],
 * hi blobs fp16, lo blobs bytes. */
int rsr_conv3x3_res_precise(rsr_ctx* ctx, const uint16_t* in, const uint8_t* in_lo, int cin, int h, int w, const float* weight,
                            const float* bias, float s1, int own_input_residual, const uint16_t* res, const uint8_t* res_lo, float s2,
                            uint16_t* out, uint8_t* out_lo);

/* ---- measurement ------------------------------------------------------------------------- */
typedef struct rsr_profile
{
    double conv_ms;        /* summed HIP-event time of the conv3x3 MFMA kernel launches */
    double conv_flops;     /* algorithmic FLOPs of those launches: 2*9*Cin*Cout*H*W per tile (true Cin/Cout) */
    long long conv_launches;
    double pre_ms, post_ms; /* preproc / postproc kernels */
    double pre_bytes, post_bytes;
    double total_ms;       /* first launch -> last launch of the profiled rsr_process* calls */
    long long tiles;       /* network tile evaluations (x8 under TTA) */
    long long calls;
} rsr_profile;

/* enable=1: every kernel launch is bracketed by hipEvents on the launch stream; accumulate until
 * rsr_get_profile(reset=1).  The events add ~1 us per launch. */
int rsr_set_profiling(rsr_ctx* ctx, int enable);
int rsr_get_profile(rsr_ctx* ctx, rsr_profile* out, int reset);

/* Accumulated HIP-event time per convolution of x4.param (index = order in the file, 351 entries),
 * over the profiled calls since the last reset. */
int rsr_get_conv_times(rsr_ctx* ctx, double* ms, int n, int reset);

/* Profiling aid: after rsr_set_option("trace_conv", i) the launch of convolution i records, for workgroup 0 /
 * MFMA wave 0, the s_memtime stamps (arrival at, release from) every stage barrier in out[0..1023]; builds with
 * -DRSR_EXP_OVLTRACE add per-stage epilogue stamps at out[1024 + 8*stage ..]; n <= 8192 values. */
int rsr_get_trace(rsr_ctx* ctx, unsigned long long* out, int n);

/* Engine knobs (optional).  key/value:
 *   "max_workspace_mb"  tile-batch memory budget (default 65536).  The effective budget is additionally bounded by 90 % of the
 *                       device memory that is actually free when a plan is built, and a batch whose workspace cannot be
 *                       allocated is halved and re-planned (down to one tile) before RSR_E_NOMEM is reported
 *   "flow_flags"        bit 0 = 64-output-channel convs with 4 MFMA waves x 64 channels instead of 8 x 32,
 *                       bit 1 = no deferred epilogue for the 32-output-channel convs, bit 2 = weights re-streamed from L2 for every
 *                       block even where a conv's weight images fit in LDS for the whole launch (default: resident where they fit),
 *                       bit 3 = conv_last (64 -> 3) through the generic 32-output-channel path instead of the variant that puts
 *                       (dy, cout) into the MFMA's M dimension (half the matrix work; same arithmetic, other summation order)
 *   "trim"              1 [default]: blocks / rows of the convs behind the trunk whose output only feeds cropped (halo) pixels of
 *                       the tile are not computed (dead-output elimination, engine.cpp: tail_margin; output bytes unchanged);
 *                       0: every padded-tile pixel is computed at every layer
 *   "tail_group"        slots (tiles; x8 under TTA) per launch group of the 2x / 4x convs (default 0 = the whole batch at once).  Small
 *                       groups keep the 4x intermediates in the Infinity Cache between upconv2 -> HRconv -> conv_last; measured
 *                       worth <= 1.5 % of those launches on MI355X and a loss at the 2x level, hence off
 *   "precise"           1: the 64-channel residual stream of the network is kept as THREE bytes per element (the fp16 the convs read +
 *                       one byte of its rounding residue) and conv_last's fp32 result is converted to uint8 without an fp16 blob in
 *                       between: half the distance to the reference's fp32 CPU path (realsr.cpp:525-838) that fp16 storage -- the
 *                       reference's own GPU path, realsr.cpp:44-46, and this engine's default (0) -- has; costs ~6 % more workspace
 *                       and the extra traffic of the residue planes in 71 of the 351 convolutions
 *   "merge"             small images of concurrent rsr_process / synchronous rsr_process_device calls walk the network as ONE tile batch, up
 *                       to this many per batch (default 16 = the most; 1 = off: the calls queue up on the compute stream).  An image is
 *                       small when its tiles are fewer than a quarter of "merge_target_items" (default 4096) 16 x 32 blocks -- 256 x 256
 *                       at tile 128 is 200; a 1080p frame, 5,900, fills the chip by itself.  The images of a batch may differ in
 *                       size ("merge_mixed" 0: only images of one size share a batch).  Output bytes unchanged
 *   "bgr"               1: the caller's images are BGR(A) (the reference's Windows build: WIC decodes to BGR, realsr.cpp:188-206,
 *                       497-515, realsr_preproc.comp:17-21); the network always sees RGB.  Default 0 = RGB(A)
 *   "max_lanes"         rsr_process calls in flight per context (default 16: small images of concurrent calls are merged, the more in
 *                       flight the fuller the launches); "chunk_mb": download chunk for pageable
 *                       destinations (default 16); "copy_threads": CPU threads per staging copy of a pageable image
 *                       (default 4; 1 = the calling thread alone)
 *   "num_cu"            persistent grid size (profiling aid)
 *   "test_repeat"       rsr_conv3x3 / rsr_conv3x3_res: the work items N times in ONE launch (an L2-resident workload; stat "last_test_us")
 *   "ws_clamp_mb"       test hook: plant the workspace bound a failed allocation leaves behind (< 0 clears it; see rsr_get_stat)
 *   "ws_fail_above_mb"  test hook: workspaces above this size fail to allocate, every time (a persistently fragmented device; < 0 off);
 *                       stat "ws_failures" counts the refusals, "clamp_backoff" the calls between two attempts at the full size
 *   "trace_conv"        conv index whose launch records s_memtime stamps (rsr_get_trace; -DRSR_EXPERIMENT -DRSR_FLOW_TRACE builds), -1 off
 *   "alternate_order"   1 [default]: every second conv walks its work items backwards (starts on the tiles the previous conv
 *                       touched last -> Infinity Cache hits); 0: always forwards
 *   "xcd_order"         1 [default]: the backward tables of "alternate_order" are reversed inside each XCD's share of the list, so an XCD
 *                       starts on the blocks IT wrote last (its own 4 MB L2); 0: the list is reversed as a whole
 *   "fold"              1 [default]: a tile's last block column, when it is only 1..14 pixels wide, is computed by FOLDED work items
 *                       (two block rows of that narrow column per 16 x 32 block: C3's 420-wide tiles 13.5 instead of 14 block
 *                       columns); 0: one plain block column more.  Output bytes unchanged (kernels.h: kFoldBit)
 *   "flow_flags" bit 4  size the patch ring as if the rounds 1-3 LDS transpose scratch were still reserved (A/B aid)
 *   "dbg"               A-B bits of ConvArgs::dbg, profiling only (bits 1 / 4 / 64 / 128 act only in -DRSR_EXPERIMENT builds of conv_flow.hip):
 *                         1 skip LDS-DMA (64: weights only, 128: patches only), 4 skip epilogue stores, 32 MFMA waves do not skip rows
 *                         outside the tile / inside the unread frame, 8192 conv_last never writes the uint8 image itself, 16384 no split tail for early download,
 *                         32768 / 65536 force the one-thread-per-pixel / the LDS-staged pre and post kernels (default: chosen per launch) */
int rsr_set_option(rsr_ctx* ctx, const char* key, long long value);

/* Engine state, read-only (tests and measurement scripts; no reference counterpart).  key:
 *   "plan_batches" / "plan_slots_per_batch"  tile batches / slots per batch of the most recently used plan (a 4K frame at tile 400
 *                       fits ONE batch of 60 slots under the default 64 GiB budget; max_workspace_mb splits it), "plans" cached plans
 *   "plan_items_lr" / "plan_items_2x" / "plan_items_4x"  work items (16 x 32 pixel blocks) of that plan per resolution level and image: what the
 *                       conv launches of a frame actually walk (blocks that only feed cropped pixels are left out, narrow last columns folded)
 *   "merged_batches" / "merged_images" / "merged_widest" / "merged_mixed"  tile batches that merged small images of concurrent calls, the
 *                       images they carried, the widest one, those whose images differed in size; "device_direct": rsr_process_device calls that ran on the caller's own stream
 *   "workspace_mb"      device memory the workspace holds, "ws_clamp_mb" the bound a failed allocation left behind (-1 = none)
 *   "lanes", "lane_in_mb", "lane_out_mb"   rsr_process lanes created so far and the device image buffers they hold (a member of
 *                       rsr_process_group allocates only the output rows of its tile range)
 *   "last_test_us"      HIP-event time of the last rsr_conv3x3 / rsr_conv3x3_res launch (with option "test_repeat" = N the
 *                       work items are repeated N times in that one launch: an L2-resident workload) */
int rsr_get_stat(rsr_ctx* ctx, const char* key, double* value);

const char* rsr_last_error(const rsr_ctx* ctx); /* ctx may be NULL: last global (create/pack) error */
const char* rsr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* REALSR_HIP_H */
