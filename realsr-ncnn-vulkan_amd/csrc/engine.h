// engine.h -- the tiled super-resolution engine behind the C-ABI (include/realsr_hip.h).
//
// Replaces RealSR::process (/root/reference/src/realsr.cpp:145-523): instead of a row-band loop with
// one Vulkan submit_and_wait per tile and >= 722 dispatches per tile, ALL tiles of an image (x8 under
// TTA) are laid out as "slots" of one batch and walk the 351 convolutions together: one kernel launch
// per network layer per batch, no host synchronisation inside a call.
//
// Concurrency (the reference calls RealSR::process from several proc threads on one object, main.cpp:811-828,
// with per-call allocators, realsr.cpp:161-167):
//   * every rsr_process call owns a LANE for its duration: private device image buffers, private pinned staging,
//     a private copy stream and events -- nothing of a call's data path is shared with another call;
//   * the network itself runs on ONE compute stream over ONE workspace; calls enqueue their kernels under `mu`
//     (a few hundred microseconds of host time) and the GPU executes them in that order.  A lane's upload runs
//     ahead of, and its download behind, the other lanes' kernels: H2D(k+1) | kernels(k) | D2H(k-1) overlap;
//   * SMALL images of concurrent calls are MERGED: a 256 x 256 image is 200 blocks for 256 CUs -- every one of the 352
//     launches costs its fixed ~14 us whatever it carries -- so calls whose image is small (any small size; same channel
//     count) are combined into ONE tile batch (up to kMaxMerge images, Engine::submit_merged): the caller that finds no leader
//     becomes one, waits until the previous merged batch is half way through the network (while it waits, further calls
//     queue up; its own launches are then enqueued underneath the second half), takes the queued calls of its
//     geometry and enqueues them as one batch; the others sleep until
//     their batch is enqueued and then fetch their own output.  The reference runs such calls side by side on the
//     device ("-j 4:4:4 for many small images", README.md:61, main.cpp:811-828); here they share the launches.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <list>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/realsr_hip.h"
#include "kernels.h"
#include "model.h"

namespace rsr {

struct DevBuf
{
    void* p = nullptr;
    size_t bytes = 0;
};

// geometry of one rsr_process call; cached per (w,h,c,T,P,tta,budget)
struct Plan
{
    int w = 0, h = 0, c = 0, T = 0, P = 0, tta = 0;
    int nimg = 1;         // images of this geometry merged into one tile batch (kernels.h kMaxMerge); their tiles follow each other image by image
    bool precise = false; // Engine::precise when the plan was built (the slots are larger: part of the cache key)
    bool ntw2 = false;    // flow_flags bit 0 when the plan was built (an MFMA wave then reaches 4 planes from one base: the tile-size bound halves)
    int tile0 = 0, tile1 = 0; // tiles [tile0, tile1) of the image's tile grid, row-major (multi-GPU tile sharding)
    long long budget_mb = 0;
    bool trim = true, xcd_order = true, fold = true;
    int out_row0 = 0; // output rows are addressed relative to this one: the first row of the tile range (the device buffer of a range holds only its rows)
    long long clamp = -1; // Engine::ws_clamp_bytes when the plan was built (part of the cache key)
    long long cap_px = 0; // slot capacity in LR pixels
    int max_tw = 0, max_th = 0;
    struct Batch
    {
        int tile0 = 0, ntiles = 0, nslots = 0;
        std::vector<BaseTile> tiles; // slot0 relative to the batch
        std::vector<TileDim> dims;   // per slot
        std::vector<WorkItem> items[3];
        std::vector<int> item_start[3]; // first item of every slot (+ end): items are sorted by slot
        double px[3] = {0, 0, 0}; // sum over slots of H*W at each level (for FLOP accounting)
        int trim4 = 0;            // prepadding * scale when the tables were trimmed to the kept rectangle (make_items), else 0
        // device copies
        BaseTile* d_tiles = nullptr;
        TileDim* d_dims = nullptr;
        WorkItem* d_items[3] = {nullptr, nullptr, nullptr};
        WorkItem* d_items_rev[3] = {nullptr, nullptr, nullptr}; // the same blocks in reverse order (see run_network)
    };
    std::vector<Batch> batches;
    int slots_per_batch = 0;
    void* d_tables = nullptr; // one allocation backing all device tables
};

// Helper threads for the staging copies of pageable images (pinned chunk <-> the caller's malloc'ed buffer): one core
// moves ~10 GB/s, a 100 MB output would cost as much as a tenth of the network.  Started on first use.
struct CopyPool
{
    struct Job
    {
        char* dst;
        const char* src;
        size_t n;
        int* pending; // under m
    };
    std::mutex m;
    std::condition_variable cv, done;
    std::deque<Job> q;
    std::vector<std::thread> workers;
    bool stop = false;
    ~CopyPool();
    void copy(void* dst, const void* src, size_t n, int threads); // returns when all of [dst, dst+n) is written
};

// one image waiting to be merged into a tile batch (Engine::submit_merged); lives on its caller's stack
struct MergeReq
{
    const void* d_in = nullptr;
    void* d_out = nullptr;
    int w = 0, h = 0, c = 0, T = 0;
    long long items = 0;          // LR-level work items of the image (Engine::image_items)
    int width = 1;                // Engine::merge_width of its geometry when the call came in
    hipEvent_t ev_in = nullptr;   // the input is complete behind this event (null: it already is)
    hipEvent_t ev_done = nullptr; // recorded on the compute stream behind the batch (null: the leader takes one from the pool -> ev_done_pool)
    bool pool_event = false;      // ev_done came from Engine::take_event: the caller gives it back
    int rc = RSR_OK;
    std::string err;
    bool done = false, lead = false; // under Engine::cq_mu
};

// one in-flight rsr_process call (host API)
struct Lane
{
    hipStream_t copy = nullptr;
    hipEvent_t ev_in = nullptr, ev_done = nullptr, ev_half = nullptr, ev_chunk[2] = {nullptr, nullptr};
    DevBuf d_in, d_out;
    std::atomic<size_t> in_bytes{0}, out_bytes{0}; // sizes of d_in / d_out for readers that do not own the lane (device_avail, rsr_get_stat)
    void* h_in = nullptr;
    size_t h_in_bytes = 0;
    void* h_out = nullptr; // two download chunks
    size_t h_out_bytes = 0;
    bool busy = false;
};

struct Engine
{
    int device = -1;
    int tta = 0;
    int scale = 4, tilesize = 200, prepadding = 10;
    bool loaded = false;
    bool bgr = false; // pixel order of the caller's images: BGR(A) like the reference's Windows/WIC path (realsr.cpp:188-206,497-515)
    // Precise residual stream (option "precise"; kernels.h ConvArgs::precise): the 64-channel trunk is kept as an fp16 plane + a byte of rounding residue and
    // conv_last's fp32 result goes to the uint8 conversion unrounded.  Default off = the storage of the reference's Vulkan path
    // (fp16 everywhere, realsr.cpp:44-46); on = half the distance to its fp32 CPU path (realsr.cpp:525-838), the bar of the parity tests.
    bool precise = false;
    long long bytes_per_px() const; // workspace bytes per padded LR pixel of a slot
    int flow_flags = 0; // launch_conv_flow flags
    int num_cu = 256;
    int dbg = 0; // ConvArgs::dbg ablation bits (profiling only)
    bool trim_tail = true; // leave out the blocks / rows behind the trunk that only feed cropped output pixels (engine.cpp: tail_margin)
    bool xcd_order = true; // backward work-item tables reversed per XCD share (each XCD re-reads what IT wrote last: L2 hits), else as a whole
    bool fold_cols = true; // a tile's last column of <= 14 pixels as folded work items (kernels.h: kFoldBit); off = one plain block column more
    bool alternate_order = true; // odd convs walk the work items backwards: they start on the data the previous conv touched last
    int test_repeat = 1;        // conv_test: work items repeated N times in one launch (measurement aid)
    double last_test_us = 0.0;  // HIP-event time of the last conv_test launch
    int trace_conv = -1; // conv index whose launch records s_memtime stamps into trace_buf (profiling only)
    DevBuf trace_buf;
    long long max_workspace_mb = 65536;
    long long ws_clamp_bytes = -1; // set after a workspace allocation failed: the next plans stay below it (-1 = none); dropped
                                   // again when the device can give twice that much (get_plan)
    long long clamp_fail_avail = -1; // device_avail() at the moment of the last failed workspace allocation (-1: the clamp was planted by the test hook)
    long long clamp_saved = -1;      // the clamp an un-clamp attempt set aside (restored when the attempt fails)
    int clamp_backoff = 4;           // calls between two un-clamp attempts while the device does not report clearly more room; doubles per failed attempt
    int clamp_calls_left = 0;
    long long ws_fail_above_bytes = -1; // test hook: workspaces above this size fail with RSR_E_NOMEM (a persistently fragmented device)
    long long ws_failures = 0;          // ... how often it fired (stat "ws_failures")
    int tail_group_slots = 0; // slots per launch group of the 2x / 4x convs (0 = the whole batch at once), see run_network
    int max_lanes = 16; // (small images are merged across calls: the more callers in flight, the fuller the launches)
    size_t chunk_bytes = size_t(16) << 20; // download chunk for pageable destinations
    int copy_threads = 4;                  // CPU threads per staging copy (1 = the calling thread alone)
    CopyPool pool;
    hipStream_t stream = nullptr;          // the compute stream

    // model
    std::vector<PackedConv> convs;
    DevBuf blob; // packed weights on device
    DevBuf zeros;

    // workspace (one allocation per buffer kind), layout = slot capacity
    long long ws_cap_px = 0;
    bool ws_precise = false; // Engine::precise the workspace was last laid out for
    DevBuf b_in, b_fea, b_rdb[3], b_up1, b_up2, b_hr, b_out3;

    // plans, most recently used first
    std::list<Plan> plans;

    // lanes
    std::mutex lane_mu;
    std::condition_variable lane_cv;
    std::vector<std::unique_ptr<Lane>> lanes;

    // merging small images across calls (see the top of this file)
    std::atomic<int> merge_max{kMaxMerge};      // option "merge": images per merged batch at most (1 = off)
    std::atomic<int> merge_target_items{4096};  // LR-level work items a merged batch aims at (16 per CU); an image with more than a quarter of it is not merged
    std::mutex cq_mu;
    std::condition_variable cq_cv;
    std::deque<MergeReq*> cq;        // FIFO of waiting calls, under cq_mu
    bool cq_leader = false;          // some caller is forming / enqueuing a batch
    hipEvent_t merge_mid = nullptr;  // recorded half way through the network of the last merged batch (the throttle of the next leader)
    bool merge_mid_used = false;
    hipEvent_t merge_done = nullptr; // ... and behind it
    int merge_last_n = 0;            // images in that batch
    std::atomic<long long> merged_batches{0}, merged_images{0}, merged_widest{0}; // stats
    std::atomic<int> merge_inbound{0}; // calls with a small image that are on their way to submit_merged (uploading): a leader waits a moment for them
    long long device_direct = 0; // rsr_process_device calls that ran on the caller's own stream (the engine was idle), under mu
    std::atomic<bool> merge_mixed{true}; // option "merge_mixed": a merged batch may hold images of different sizes (0: of one geometry only)
    DevBuf mix_tab[3];               // rotating device tables of such batches
    hipEvent_t mix_ev[3] = {nullptr, nullptr, nullptr};
    unsigned long long mix_seq = 0;
    std::atomic<long long> merged_mixed{0}; // stat: merged batches whose images differed in size
    long long image_items(int w, int h, long long limit) const;
    int merge_width(int w, int h, int c) const; // images of this geometry one batch may take (1: not a small image / merging off)
    int submit_merged(MergeReq& r);             // returns when r's batch has been ENQUEUED (r.ev_done recorded) or failed
    int run_group(MergeReq* const* g, int n);   // mu inside

    std::mutex mu; // compute section: plan cache, workspace, kernel enqueue, profiling state
    std::vector<hipEvent_t> sync_events; // free list, under mu

    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;
    struct Seg
    {
        int cls; // 0 pre, 1 conv, 2 post
        double flops, bytes;
        int conv_index;
    };
    double conv_ms_by_index[kNumConvs] = {0};
    std::vector<Seg> segs;
    size_t ev_used = 0;
    rsr_profile prof{};
    void (*progress)(int done, int total, void* user) = nullptr; // per tile batch (the reference prints per tile, realsr.cpp:481)
    void* progress_user = nullptr;

    ~Engine();
    int init(int gpuid, int tta_mode);
    int load_files(const char* param, const char* bin);
    int load_blob_host(const void* blob, size_t bytes);
    int load_blob_device(const void* blob, size_t bytes);
    // d_in/d_out on this device.  user_stream == nullptr: returns when the output is complete (sync) or enqueued (!sync);
    // otherwise ordered after / before the work of user_stream, asynchronous.
    int process_device(const void* d_in, int w, int h, int c, void* d_out, hipStream_t user_stream, bool sync);
    // tile0/tile1: tiles [tile0, tile1) of the row-major tile grid only (tile1 < 0: all); `out` is always the full (4w x 4h x c) image,
    // only the output rectangles of
    // those tiles are written
    int process_host(const uint8_t* in, int w, int h, int c, uint8_t* out, int tile0 = 0, int tile1 = -1);
    int net_forward(const uint16_t* in, int w, int h, uint16_t* out, float* out32 = nullptr); // out32: the fp32 result (precise mode only)
    // out = act(conv + b); with s1 != 0: v = s1*(conv + b) [+ in[0:cout] when own_res] [, v = s2*v + res when res]
    // precise form (in_lo / res_lo / out_lo, any may be null): the hi + lo / 2048 residual stream of ConvArgs::precise
    int conv_test(const uint16_t* in, int cin, int h, int w, int ups, const float* weight, const float* bias, int cout, int lrelu,
                  uint16_t* out, float s1 = 0.f, int own_res = 0, const uint16_t* res = nullptr, float s2 = 1.f, bool prec = false,
                  const uint8_t* in_lo = nullptr, const uint8_t* res_lo = nullptr, uint8_t* out_lo = nullptr);

    // ---- internals (call with `mu` held unless noted) ----
    static constexpr int plane_ch() { return kPlaneCh; }
    int ensure(DevBuf& b, size_t bytes);
    int ensure_planes(DevBuf& b, size_t bytes, long long plane_bytes, bool layout_changed, bool zero_all, hipStream_t st);
    int get_plan(int w, int h, int c, int tile0, int tile1, int nimg, Plan*& out);
    long long device_avail(int w, int h, int c);
    void free_workspace(hipStream_t st);
    int ensure_workspace(int nslots, long long cap_px, hipStream_t st);
    // fused_outs: non-null = conv_last writes the uint8 images itself (one pointer per image of the batch)
    // ev_mid: recorded behind the middle RDB (a merged batch's throttle event)
    // mid_rdb >= 0: ev_mid is recorded behind that RDB.  nslots_used < b.nslots: only the first slots of the batch (a merged batch narrower than its plan)
    int run_network(const Plan::Batch& b, hipStream_t st, uint8_t* const* fused_outs = nullptr, int nimg = 1, const int* fused_out_ws = nullptr, int split_slot = 0,
                    hipEvent_t ev_half = nullptr, hipEvent_t ev_mid = nullptr, int mid_rdb = -1, int nslots_used = -1);
    int launch_batch(const Plan::Batch& b, long long cap_px, int max_tw, int max_th, int out_row0, const void* const* d_in, void* const* d_out, const int* ws,
                     const int* hs, int nimg, int c, int ntiles, hipStream_t st, int split_slot, hipEvent_t ev_half, hipEvent_t ev_mid);
    int enqueue_mixed(MergeReq* const* g, int n, hipStream_t st, hipEvent_t ev_mid); // a merged batch of images of different sizes: tables built on the fly
    int launch(ConvArgs& a, int ci, const Plan::Batch& b, hipStream_t st);
    int enqueue_image(const void* d_in, int w, int h, int c, void* d_out, hipStream_t st, int tile0 = 0, int tile1 = -1,
                      hipEvent_t ev_half = nullptr, size_t* half_rows = nullptr);
    // nimg images of one geometry as ONE tile batch (nimg <= kMaxMerge; whole images only when nimg > 1)
    // plan_nimg >= nimg: the plan is the one of plan_nimg images and only the first nimg of them are launched (every width of a merged
    // batch shares ONE plan: slots, tiles and work items of an image are contiguous, so a narrower batch is a prefix of the tables)
    int enqueue_images(const void* const* d_in, void* const* d_out, int nimg, int w, int h, int c, hipStream_t st, int tile0 = 0, int tile1 = -1,
                       hipEvent_t ev_half = nullptr, size_t* half_rows = nullptr, hipEvent_t ev_mid = nullptr, int plan_nimg = 0);
    void mark_begin(hipStream_t st);
    void mark(int cls, double flops, double bytes, hipStream_t st, int conv_index = -1);
    void collect_profile(hipStream_t st);
    void free_plans();
    hipEvent_t take_event();
    hipEvent_t take_event_timed(); // caller destroys
    void give_event(hipEvent_t e);
    Lane* acquire_lane(); // lane_mu inside
    void release_lane(Lane* l);
    int adopt_table(const unsigned char* head, size_t bytes); // validated header -> convs
    static int fail(int code, const std::string& msg);
};

const char* last_error(); // message of the calling thread's last failure
long long share_pool_stat(int what); // group.cpp: 0 = worker threads of rsr_process_group's pool, 1 = shares run inline because no worker could be started

} // namespace rsr

// The opaque context of the C-ABI (include/realsr_hip.h): ONE definition for every translation unit.
struct rsr_ctx
{
    rsr::Engine e;
};
