// engine.h -- the tiled super-resolution engine behind the C-ABI (include/realsr_hip.h).
//
// Replaces RealSR::process (/root/reference/src/realsr.cpp:145-523): instead of a row-band loop with
// one Vulkan submit_and_wait per tile and >= 722 dispatches per tile, ALL tiles of an image (x8 under
// TTA) are laid out as "slots" of one batch and walk the 351 convolutions together: one kernel launch
// per network layer per batch, no host synchronisation inside a call.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
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

// geometry of one rsr_process call; cached between calls with identical (w,h,c,T,P,tta)
struct Plan
{
    int w = 0, h = 0, c = 0, T = 0, P = 0, tta = 0;
    long long cap_px = 0; // slot capacity in LR pixels
    int max_tw = 0, max_th = 0;
    struct Batch
    {
        int tile0 = 0, ntiles = 0, nslots = 0;
        std::vector<BaseTile> tiles; // slot0 relative to the batch
        std::vector<TileDim> dims;   // per slot
        std::vector<WorkItem> items[3];
        double px[3] = {0, 0, 0}; // sum over slots of H*W at each level (for FLOP accounting)
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

struct Engine
{
    int device = -1;
    int tta = 0;
    int scale = 4, tilesize = 200, prepadding = 10;
    bool loaded = false;
    bool trunk_fp32 = false; // residual trunk storage: fp16 like the reference Vulkan path (realsr.cpp:45); true = extra fp32 copy
    bool use_dma = true;
    int kernel_version = 3; // 3: conv3x3_ring, 2: conv3x3_pipe, 1: conv3x3_mfma
    int num_cu = 256;
    bool ring_nt2 = false; // use conv3x3_ring also for 64-output-channel convs (slower there: 168-VGPR budget)
    int dbg = 0; // ConvArgs::dbg ablation bits (profiling only)
    int stagger_unit = 0; // s_sleep units (64 cycles) per (chunk + 2) of start delay between workgroup phases
    bool alternate_order = true; // odd convs walk the work items backwards: they start on the data the previous conv touched last
    int trace_conv = -1; // conv index whose launch records s_memtime stamps into trace_buf (profiling only)
    DevBuf trace_buf;
    long long max_workspace_mb = 65536;
    hipStream_t stream = nullptr;

    // model
    std::vector<PackedConv> convs;
    DevBuf blob; // packed weights on device
    DevBuf zeros;

    // workspace (one allocation per buffer kind, nslots each)
    int ws_slots = 0;
    long long ws_cap_px = 0;
    DevBuf b_in, b_fea, b_rdb[3], b_t32, b_r32, b_up1, b_up2, b_hr, b_out3;

    // image staging for the host API
    DevBuf d_img_in, d_img_out;

    Plan plan;
    std::mutex mu;
    std::string err;

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

    ~Engine();
    int init(int gpuid, int tta_mode);
    int load_files(const char* param, const char* bin);
    int load_blob_host(const void* blob, size_t bytes);
    int load_blob_device(const void* blob, size_t bytes);
    int process_device(const void* d_in, int w, int h, int c, void* d_out, hipStream_t st, bool sync);
    int process_host(const uint8_t* in, int w, int h, int c, uint8_t* out);
    int net_forward(const uint16_t* in, int w, int h, uint16_t* out);
    int conv_test(const uint16_t* in, int cin, int h, int w, int ups, const float* weight, const float* bias, int cout, int lrelu,
                  uint16_t* out);

    int ensure(DevBuf& b, size_t bytes);
    int ensure_zeroed(DevBuf& b, size_t bytes, bool layout_changed, hipStream_t st);
    int build_plan(int w, int h, int c);
    int ensure_workspace(int nslots, long long cap_px, hipStream_t st);
    void run_network(const Plan::Batch& b, hipStream_t st);
    void mark_begin(hipStream_t st);
    void mark(int cls, double flops, double bytes, hipStream_t st, int conv_index = -1);
    void collect_profile(hipStream_t st);
    void free_plan();
    int fail(int code, const std::string& msg);
};

} // namespace rsr
