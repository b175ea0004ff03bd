// capi.cpp -- extern "C" entry points declared in include/realsr_hip.h.
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "engine.h"

using namespace rsr;

// Errors raised before a context exists (rsr_create, rsr_model_pack) and errors of a context share one per-thread
// message: rsr_last_error() always reports the calling thread's most recent failure.
#define g_err_set(msg) ((void)Engine::fail(0, (msg)))

#define CK(expr)                                                          \
    do                                                                    \
    {                                                                     \
        hipError_t e_ = (expr);                                           \
        if (e_ != hipSuccess)                                             \
        {                                                                 \
            rc = ctx->e.fail(RSR_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
            goto done;                                                    \
        }                                                                 \
    } while (0)

#pragma GCC visibility push(default)
extern "C" {

const char* rsr_version(void) { return "realsr-hip 0.3 (gfx950)"; }

// Pinned host memory: images allocated here are copied to / from the GPU without the staging copy (the reference's
// Vulkan path gets the same effect from its staging allocator, realsr.cpp:161-167).
void* rsr_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess)
    {
        (void)hipGetLastError();
        Engine::fail(RSR_E_NOMEM, "hipHostMalloc failed");
        return nullptr;
    }
    return p;
}

void rsr_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

int rsr_device_memory(int gpuid, long long* free_mb, long long* total_mb)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || gpuid < 0 || gpuid >= n) return Engine::fail(RSR_E_DEVICE, "invalid gpu device " + std::to_string(gpuid));
    int prev = 0;
    (void)hipGetDevice(&prev);
    size_t f = 0, t = 0;
    hipError_t e = hipSetDevice(gpuid);
    if (e == hipSuccess) e = hipMemGetInfo(&f, &t);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) return Engine::fail(RSR_E_DEVICE, std::string("hipMemGetInfo: ") + hipGetErrorString(e));
    if (free_mb) *free_mb = (long long)(f >> 20);
    if (total_mb) *total_mb = (long long)(t >> 20);
    return RSR_OK;
}

int rsr_set_progress_callback(rsr_ctx* ctx, void (*cb)(int done, int total, void* user), void* user)
{
    if (!ctx) return RSR_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->e.mu);
    ctx->e.progress = cb;
    ctx->e.progress_user = user;
    return RSR_OK;
}

const char* rsr_last_error(const rsr_ctx* ctx)
{
    (void)ctx;
    return rsr::last_error();
}

int rsr_create(rsr_ctx** out, int gpuid, int tta_mode, int num_threads)
{
    (void)num_threads;
    if (!out) return Engine::fail(RSR_E_ARG, "null out pointer");
    *out = nullptr;
    rsr_ctx* c = new (std::nothrow) rsr_ctx;
    if (!c) return Engine::fail(RSR_E_NOMEM, "out of memory");
    const int rc = c->e.init(gpuid, tta_mode);
    if (rc != RSR_OK)
    {
        delete c;
        return rc;
    }
    *out = c;
    return RSR_OK;
}

void rsr_destroy(rsr_ctx* ctx) { delete ctx; }

int rsr_load(rsr_ctx* ctx, const char* parampath, const char* modelpath)
{
    if (!ctx) return RSR_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->e.mu);
    return ctx->e.load_files(parampath, modelpath);
}

int rsr_set_params(rsr_ctx* ctx, int scale, int tilesize, int prepadding)
{
    if (!ctx) return RSR_E_ARG;
    if (scale != 4) return ctx->e.fail(RSR_E_ARG, "scale must be 4 (main.cpp:533-537)");
    if (tilesize < 1) return ctx->e.fail(RSR_E_ARG, "tilesize must be >= 1");
    if (prepadding < 0 || prepadding > 64) return ctx->e.fail(RSR_E_ARG, "prepadding out of range");
    std::lock_guard<std::mutex> lk(ctx->e.mu);
    ctx->e.scale = scale;
    ctx->e.tilesize = tilesize;
    ctx->e.prepadding = prepadding;
    return RSR_OK;
}

int rsr_process(rsr_ctx* ctx, const uint8_t* in, int w, int h, int c, uint8_t* out)
{
    if (!ctx) return RSR_E_ARG;
    return ctx->e.process_host(in, w, h, c, out);
}

int rsr_process_many(rsr_ctx* ctx, int n, const uint8_t* const* in, const int* w, const int* h, const int* c, uint8_t* const* out, int* rcs)
{
    if (!ctx || n < 0 || (n > 0 && (!in || !w || !h || !c || !out))) return RSR_E_ARG;
    if (n == 0) return RSR_OK;
    try
    {
    // One caller thread, many images: each image is an ordinary rsr_process call on a helper thread (as many in flight as the context has
    // lanes), so that small images meet in merged tile batches exactly as the calls of a multi-threaded host do (engine.h).
    int nthreads = 1;
    {
        std::lock_guard<std::mutex> ll(ctx->e.lane_mu);
        nthreads = std::max(1, std::min(n, ctx->e.max_lanes));
    }
    std::vector<int> codes(size_t(n), RSR_OK);
    std::vector<std::string> msgs{size_t(n)};
    std::atomic<int> next{0};
    auto work = [&]() {
        for (int i = next++; i < n; i = next++)
        {
            codes[size_t(i)] = ctx->e.process_host(in[i], w[i], h[i], c[i], out[i]);
            if (codes[size_t(i)] != RSR_OK) msgs[size_t(i)] = rsr::last_error();
        }
    };
    std::vector<std::thread> helpers;
    try
    {
        for (int t = 1; t < nthreads; t++) helpers.emplace_back(work);
    }
    catch (...) // thread limit / no memory: the images the missing helpers would have taken are done by the threads that exist
    {
    }
    work(); // the calling thread takes its share
    for (std::thread& t : helpers) t.join();
    int first = RSR_OK;
    for (int i = 0; i < n; i++)
    {
        if (rcs) rcs[i] = codes[size_t(i)];
        if (first == RSR_OK && codes[size_t(i)] != RSR_OK) first = Engine::fail(codes[size_t(i)], "image " + std::to_string(i) + ": " + msgs[size_t(i)]);
    }
    return first;
    }
    catch (const std::bad_alloc&) // (the bookkeeping vectors; helper threads that could not be started are handled above) -- nothing crosses extern "C"
    {
        return Engine::fail(RSR_E_NOMEM, "rsr_process_many: out of host memory");
    }
}

int rsr_process_device(rsr_ctx* ctx, const void* d_in, int w, int h, int c, void* d_out, void* stream)
{
    if (!ctx) return RSR_E_ARG;
    return ctx->e.process_device(d_in, w, h, c, d_out, static_cast<hipStream_t>(stream), stream == nullptr);
}

int rsr_model_pack(const char* parampath, const char* modelpath, void* dst, size_t cap, size_t* need)
{
    if (!parampath || !modelpath) return Engine::fail(RSR_E_ARG, "null path");
    Model m;
    std::string err;
    int rc = load_model(parampath, modelpath, m, err);
    if (rc != RSR_OK) return Engine::fail(rc, err);
    const size_t n = packed_size(m);
    if (need) *need = n;
    if (!dst) return RSR_OK;
    rc = pack_model(m, dst, cap, err);
    return rc == RSR_OK ? RSR_OK : Engine::fail(rc, err);
}

int rsr_load_packed(rsr_ctx* ctx, const void* blob, size_t bytes, int is_device)
{
    if (!ctx || !blob) return RSR_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->e.mu);
    return is_device ? ctx->e.load_blob_device(blob, bytes) : ctx->e.load_blob_host(blob, bytes);
}

int rsr_model_info(const char* parampath, const char* modelpath, int* n_layers, int* n_convs, long long* n_weights,
                   long long* n_biases, int* bin_encoding)
{
    if (!parampath || !modelpath) return Engine::fail(RSR_E_ARG, "null path");
    Model m;
    std::string err;
    const int rc = load_model(parampath, modelpath, m, err);
    if (rc != RSR_OK) return Engine::fail(rc, err);
    if (n_layers) *n_layers = m.n_layers;
    if (n_convs) *n_convs = int(m.convs.size());
    if (n_weights) *n_weights = m.n_weights;
    if (n_biases) *n_biases = m.n_biases;
    if (bin_encoding) *bin_encoding = m.bin_encoding;
    return RSR_OK;
}

// ---- shader-equivalent kernels, host pointers in/out --------------------------------------------
static int preproc_common(rsr_ctx* ctx, const uint8_t* band, int w, int h, int channels, uint16_t* const top[8], int ntop,
                          int outw, int outh, int pad_top, int pad_left, int crop_x, int crop_y, uint16_t* alpha, int alphaw,
                          int alphah)
{
    if (!ctx || !band || !top || w < 1 || h < 1 || (channels != 3 && channels != 4) || outw < 1 || outh < 1) return RSR_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->e.mu);
    int rc = RSR_OK;
    const size_t nin = size_t(w) * h * channels, ntile = size_t(outw) * outh * 3 * 2;
    const size_t nalpha = (channels == 4 && alpha) ? size_t(alphaw) * alphah * 2 : 0;
    uint8_t* d_in = nullptr;
    uint16_t* d_top[8] = {nullptr};
    uint16_t* d_alpha = nullptr;
    hipStream_t st = ctx->e.stream;
    CK(hipSetDevice(ctx->e.device));
    CK(hipMalloc(&d_in, nin));
    CK(hipMemcpy(d_in, band, nin, hipMemcpyHostToDevice));
    for (int i = 0; i < ntop; i++)
    {
        CK(hipMalloc(&d_top[i], ntile));
        CK(hipMemsetAsync(d_top[i], 0, ntile, st)); // same stream as the kernel: the null stream does not order with it
    }
    if (nalpha)
    {
        CK(hipMalloc(&d_alpha, nalpha));
        CK(hipMemsetAsync(d_alpha, 0, nalpha, st));
    }
    launch_preproc_shader(d_in, w, h, channels, d_top, ntop, outw, outh, outw * outh, pad_top, pad_left, crop_x, crop_y, d_alpha,
                          alphaw, alphah, 0, st);
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    for (int i = 0; i < ntop; i++) CK(hipMemcpy(top[i], d_top[i], ntile, hipMemcpyDeviceToHost));
    if (nalpha) CK(hipMemcpy(alpha, d_alpha, nalpha, hipMemcpyDeviceToHost));
done:
    if (d_in) (void)hipFree(d_in);
    for (int i = 0; i < 8; i++)
        if (d_top[i]) (void)hipFree(d_top[i]);
    if (d_alpha) (void)hipFree(d_alpha);
    return rc;
}

int rsr_preproc(rsr_ctx* ctx, const uint8_t* band, int w, int h, int channels, uint16_t* top, int outw, int outh, int pad_top,
                int pad_left, int crop_x, int crop_y, uint16_t* alpha, int alphaw, int alphah)
{
    uint16_t* tops[8] = {top, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (!top) return RSR_E_ARG;
    return preproc_common(ctx, band, w, h, channels, tops, 1, outw, outh, pad_top, pad_left, crop_x, crop_y, alpha, alphaw, alphah);
}

int rsr_preproc_tta(rsr_ctx* ctx, const uint8_t* band, int w, int h, int channels, uint16_t* const top[8], int outw, int outh,
                    int pad_top, int pad_left, int crop_x, int crop_y)
{
    if (!top) return RSR_E_ARG;
    for (int i = 0; i < 8; i++)
        if (!top[i]) return RSR_E_ARG;
    return preproc_common(ctx, band, w, h, channels, top, 8, outw, outh, pad_top, pad_left, crop_x, crop_y, nullptr, 0, 0);
}

static int postproc_common(rsr_ctx* ctx, const uint16_t* const bottom[8], int nb, int w, int h, const uint16_t* alpha, int alphaw,
                           int alphah, uint8_t* top, int outw, int outh, int offset_x, int gx_max, int crop_x, int crop_y,
                           int channels)
{
    if (!ctx || !bottom || !top || w < 1 || h < 1 || (channels != 3 && channels != 4) || outw < 1 || outh < 1 || gx_max < 0)
        return RSR_E_ARG;
    if (channels == 4 && !alpha) return RSR_E_ARG;
    if (gx_max == 0) return RSR_OK;
    std::lock_guard<std::mutex> lk(ctx->e.mu);
    int rc = RSR_OK;
    const size_t ntile = size_t(w) * h * 3 * 2, nout = size_t(outw) * outh * channels;
    const size_t nalpha = channels == 4 ? size_t(alphaw) * alphah * 2 : 0;
    uint16_t* d_b[8] = {nullptr};
    uint16_t* d_alpha = nullptr;
    uint8_t* d_top = nullptr;
    hipStream_t st = ctx->e.stream;
    CK(hipSetDevice(ctx->e.device));
    for (int i = 0; i < nb; i++)
    {
        CK(hipMalloc(&d_b[i], ntile));
        CK(hipMemcpy(d_b[i], bottom[i], ntile, hipMemcpyHostToDevice));
    }
    if (nalpha)
    {
        CK(hipMalloc(&d_alpha, nalpha));
        CK(hipMemcpy(d_alpha, alpha, nalpha, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&d_top, nout));
    CK(hipMemcpy(d_top, top, nout, hipMemcpyHostToDevice));
    launch_postproc_shader(d_b, nb, w, h, w * h, d_alpha, alphaw, alphah, d_top, outw, outh, offset_x, gx_max, crop_x, crop_y,
                           channels, 0, st);
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    CK(hipMemcpy(top, d_top, nout, hipMemcpyDeviceToHost));
done:
    for (int i = 0; i < 8; i++)
        if (d_b[i]) (void)hipFree(d_b[i]);
    if (d_alpha) (void)hipFree(d_alpha);
    if (d_top) (void)hipFree(d_top);
    return rc;
}

int rsr_postproc(rsr_ctx* ctx, const uint16_t* bottom, int w, int h, const uint16_t* alpha, int alphaw, int alphah, uint8_t* top,
                 int outw, int outh, int offset_x, int gx_max, int crop_x, int crop_y, int channels)
{
    const uint16_t* b[8] = {bottom, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (!bottom) return RSR_E_ARG;
    return postproc_common(ctx, b, 1, w, h, alpha, alphaw, alphah, top, outw, outh, offset_x, gx_max, crop_x, crop_y, channels);
}

int rsr_postproc_tta(rsr_ctx* ctx, const uint16_t* const bottom[8], int w, int h, uint8_t* top, int outw, int outh, int offset_x,
                     int gx_max, int crop_x, int crop_y, int channels)
{
    if (!bottom) return RSR_E_ARG;
    for (int i = 0; i < 8; i++)
        if (!bottom[i]) return RSR_E_ARG;
    if (channels != 3) return RSR_E_ARG; // alpha goes through rsr_postproc's path
    return postproc_common(ctx, bottom, 8, w, h, nullptr, 0, 0, top, outw, outh, offset_x, gx_max, crop_x, crop_y, channels);
}

int rsr_net_forward(rsr_ctx* ctx, const uint16_t* in, int w, int h, uint16_t* out)
{
    if (!ctx) return RSR_E_ARG;
    return ctx->e.net_forward(in, w, h, out);
}

int rsr_net_forward_f32(rsr_ctx* ctx, const uint16_t* in, int w, int h, float* out)
{
    if (!ctx) return RSR_E_ARG;
    return ctx->e.net_forward(in, w, h, nullptr, out);
}

int rsr_conv3x3(rsr_ctx* ctx, const uint16_t* in, int cin, int h, int w, int upsample2x, const float* weight, const float* bias,
                int cout, int lrelu, uint16_t* out)
{
    if (!ctx) return RSR_E_ARG;
    return ctx->e.conv_test(in, cin, h, w, upsample2x, weight, bias, cout, lrelu, out);
}

int rsr_conv3x3_res(rsr_ctx* ctx, const uint16_t* in, int cin, int h, int w, const float* weight, const float* bias, int cout,
                    float s1, int own_input_residual, const uint16_t* res, float s2, uint16_t* out)
{
    if (!ctx) return RSR_E_ARG;
    if (s1 == 0.f) return Engine::fail(RSR_E_ARG, "s1 must be non-zero");
    return ctx->e.conv_test(in, cin, h, w, 0, weight, bias, cout, 0, out, s1, own_input_residual, res, s2);
}

int rsr_conv3x3_res_precise(rsr_ctx* ctx, const uint16_t* in, const uint8_t* in_lo, int cin, int h, int w, const float* weight, const float* bias,
                            float s1, int own_input_residual, const uint16_t* res, const uint8_t* res_lo, float s2, uint16_t* out, uint8_t* out_lo)
{
    if (!ctx) return RSR_E_ARG;
    if (s1 == 0.f) return Engine::fail(RSR_E_ARG, "s1 must be non-zero");
    return ctx->e.conv_test(in, cin, h, w, 0, weight, bias, 64, 0, out, s1, own_input_residual, res, s2, true, in_lo, res_lo, out_lo);
}

int rsr_set_profiling(rsr_ctx* ctx, int enable)
{
    if (!ctx) return RSR_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->e.mu);
    ctx->e.profiling = enable != 0;
    return RSR_OK;
}

int rsr_get_profile(rsr_ctx* ctx, rsr_profile* out, int reset)
{
    if (!ctx || !out) return RSR_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->e.mu);
    *out = ctx->e.prof;
    if (reset) std::memset(&ctx->e.prof, 0, sizeof(rsr_profile));
    return RSR_OK;
}

int rsr_get_conv_times(rsr_ctx* ctx, double* ms, int n, int reset)
{
    if (!ctx || !ms || n < 0) return RSR_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->e.mu);
    for (int i = 0; i < n && i < kNumConvs; i++) ms[i] = ctx->e.conv_ms_by_index[i];
    if (reset) std::memset(ctx->e.conv_ms_by_index, 0, sizeof ctx->e.conv_ms_by_index);
    return RSR_OK;
}

int rsr_get_trace(rsr_ctx* ctx, unsigned long long* out, int n)
{
    if (!ctx || !out || n < 0 || n > 8192) return RSR_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->e.mu);
    if (!ctx->e.trace_buf.p) return ctx->e.fail(RSR_E_STATE, "tracing was never enabled");
    if (hipSetDevice(ctx->e.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(out, ctx->e.trace_buf.p, size_t(n) * 8, hipMemcpyDeviceToHost) != hipSuccess)
        return ctx->e.fail(RSR_E_DEVICE, "trace readback failed");
    return RSR_OK;
}

int rsr_get_stat(rsr_ctx* ctx, const char* key, double* value)
{
    if (!ctx || !key || !value) return RSR_E_ARG;
    rsr::Engine& e = ctx->e;
    std::lock_guard<std::mutex> lk(e.mu);
    const std::string k(key);
    auto lanes_bytes = [&](bool out) {
        std::lock_guard<std::mutex> ll(e.lane_mu);
        double n = 0;
        for (const auto& l : e.lanes) n += double(out ? l->out_bytes.load(std::memory_order_relaxed) : l->in_bytes.load(std::memory_order_relaxed));
        return n;
    };
    if (k == "plan_batches") *value = e.plans.empty() ? 0.0 : double(e.plans.front().batches.size());
    else if (k == "plan_slots_per_batch") *value = e.plans.empty() ? 0.0 : double(e.plans.front().slots_per_batch);
    else if (k == "plans") *value = double(e.plans.size());
    else if (k == "plan_items_lr" || k == "plan_items_2x" || k == "plan_items_4x")
    { // work items (16 x 32 blocks; a folded item = one block) of the most recently used plan at one resolution level, all batches
        const int lvl = k == "plan_items_lr" ? 0 : (k == "plan_items_2x" ? 1 : 2);
        double n = 0;
        if (!e.plans.empty())
            for (const auto& b : e.plans.front().batches) n += double(b.items[lvl].size());
        *value = e.plans.empty() ? 0.0 : n / double(e.plans.front().nimg); // (a merged plan holds the tables of `nimg` images: per image)
    }
    else if (k == "ws_clamp_mb") *value = e.ws_clamp_bytes < 0 ? -1.0 : double(e.ws_clamp_bytes) / 1048576.0;
    else if (k == "ws_failures") *value = double(e.ws_failures);
    else if (k == "pool_workers") *value = double(share_pool_stat(0));
    else if (k == "pool_inline_runs") *value = double(share_pool_stat(1));
    else if (k == "clamp_backoff") *value = double(e.clamp_backoff);
    else if (k == "workspace_mb")
    {
        double n = 0;
        for (const rsr::DevBuf* wb : {&e.b_in, &e.b_fea, &e.b_rdb[0], &e.b_rdb[1], &e.b_rdb[2], &e.b_up1, &e.b_up2, &e.b_hr, &e.b_out3}) n += double(wb->bytes);
        *value = n / 1048576.0;
    }
    else if (k == "lanes") { std::lock_guard<std::mutex> ll(e.lane_mu); *value = double(e.lanes.size()); }
    else if (k == "lane_out_mb") *value = lanes_bytes(true) / 1048576.0;
    else if (k == "lane_in_mb") *value = lanes_bytes(false) / 1048576.0;
    else if (k == "last_test_us") *value = e.last_test_us;
    else if (k == "device_direct") *value = double(e.device_direct);
    else if (k == "merged_batches") *value = double(e.merged_batches.load());
    else if (k == "merged_images") *value = double(e.merged_images.load());
    else if (k == "merged_widest") *value = double(e.merged_widest.load());
    else if (k == "merged_mixed") *value = double(e.merged_mixed.load());
    else return e.fail(RSR_E_ARG, "unknown stat " + k);
    return RSR_OK;
}

int rsr_set_option(rsr_ctx* ctx, const char* key, long long value)
{
    if (!ctx || !key) return RSR_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->e.mu);
    const std::string k(key);
    if (k == "max_workspace_mb")
    {
        if (value < 1) return ctx->e.fail(RSR_E_ARG, "max_workspace_mb must be >= 1");
        ctx->e.max_workspace_mb = value; // plans are keyed by the budget: the next call builds a new one
        ctx->e.ws_clamp_bytes = -1;
    }
    else if (k == "tail_group")
    {
        if (value < 0) return ctx->e.fail(RSR_E_ARG, "tail_group must be >= 0");
        ctx->e.tail_group_slots = int(value);
    }
    else if (k == "bgr")
        ctx->e.bgr = value != 0;
    else if (k == "merge")
    {
        if (value < 1 || value > kMaxMerge) return ctx->e.fail(RSR_E_ARG, "merge out of range (1 .. 16)");
        ctx->e.merge_max = int(value);
    }
    else if (k == "merge_mixed")
        ctx->e.merge_mixed = value != 0;
    else if (k == "merge_target_items")
    {
        if (value < 1 || value > (1 << 20)) return ctx->e.fail(RSR_E_ARG, "merge_target_items out of range");
        ctx->e.merge_target_items = int(value);
    }
    else if (k == "precise")
        ctx->e.precise = value != 0; // plans are keyed by it (larger slots); the workspace grows on the next call
    else if (k == "flow_flags")
        ctx->e.flow_flags = int(value); // plans are keyed by bit 0 (it halves the largest tile the 32-bit plane offsets can address)
    else if (k == "max_lanes")
    {
        if (value < 1 || value > 64) return ctx->e.fail(RSR_E_ARG, "max_lanes out of range");
        std::lock_guard<std::mutex> ll(ctx->e.lane_mu);
        ctx->e.max_lanes = int(value);
    }
    else if (k == "chunk_mb")
    {
        if (value < 1 || value > 4096) return ctx->e.fail(RSR_E_ARG, "chunk_mb out of range");
        ctx->e.chunk_bytes = size_t(value) << 20;
    }
    else if (k == "copy_threads")
    {
        if (value < 1 || value > 64) return ctx->e.fail(RSR_E_ARG, "copy_threads out of range");
        ctx->e.copy_threads = int(value);
    }
    else if (k == "trace_conv")
    {
        ctx->e.trace_conv = int(value);
        if (value >= 0 && !ctx->e.trace_buf.p)
        {
            if (hipSetDevice(ctx->e.device) != hipSuccess || hipMalloc(&ctx->e.trace_buf.p, 65536) != hipSuccess)
                return ctx->e.fail(RSR_E_NOMEM, "trace buffer");
            ctx->e.trace_buf.bytes = 65536;
        }
        if (value >= 0)
        { // stamps of an earlier traced conv must not survive into this one's read-out
            if (hipSetDevice(ctx->e.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
                hipMemset(ctx->e.trace_buf.p, 0, 65536) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
                return ctx->e.fail(RSR_E_DEVICE, "trace buffer clear");
        }
    }
    else if (k == "trim")
        ctx->e.trim_tail = value != 0; // plans are keyed by it
    else if (k == "xcd_order")
        ctx->e.xcd_order = value != 0; // plans are keyed by it
    else if (k == "alternate_order")
        ctx->e.alternate_order = value != 0;
    else if (k == "fold")
        ctx->e.fold_cols = value != 0; // plans are keyed by it
    else if (k == "dbg")
        ctx->e.dbg = int(value);
    else if (k == "test_repeat")
    {
        if (value < 1 || value > 100000) return ctx->e.fail(RSR_E_ARG, "test_repeat out of range");
        ctx->e.test_repeat = int(value);
    }
    else if (k == "ws_clamp_mb") // test hook: what a failed workspace allocation leaves behind (enqueue_image's retry); < 0 clears it
    {
        ctx->e.ws_clamp_bytes = value < 0 ? -1 : value * 1048576;
        ctx->e.clamp_fail_avail = -1;
        ctx->e.clamp_calls_left = 0;
        ctx->e.clamp_backoff = 4;
    }
    else if (k == "ws_fail_above_mb") // test hook: workspaces above this size fail to allocate, persistently (< 0: off)
        ctx->e.ws_fail_above_bytes = value < 0 ? -1 : value * 1048576;
    else if (k == "num_cu")
    {
        if (value < 8 || value > 1024) return ctx->e.fail(RSR_E_ARG, "num_cu out of range");
        ctx->e.num_cu = int(value);
    }
    else
        return ctx->e.fail(RSR_E_ARG, "unknown option " + k);
    return RSR_OK;
}

} // extern "C"
#pragma GCC visibility pop
