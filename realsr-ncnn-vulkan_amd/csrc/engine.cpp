// engine.cpp -- host side of the tiled engine (compiled by hipcc as HIP host code).  See engine.h.
#include "engine.h"

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace rsr {

#define HIP_TRY(expr)                                                                              \
    do                                                                                             \
    {                                                                                              \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(RSR_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

int Engine::fail(int code, const std::string& msg)
{
    err = msg;
    return code;
}

Engine::~Engine()
{
    if (device >= 0) (void)hipSetDevice(device);
    free_plan();
    DevBuf* all[] = {&blob, &zeros, &b_in, &b_fea, &b_rdb[0], &b_rdb[1], &b_rdb[2], &b_t32, &b_r32,
                     &b_up1, &b_up2, &b_hr, &b_out3, &d_img_in, &d_img_out};
    for (DevBuf* b : all)
        if (b->p) (void)hipFree(b->p);
    for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
}

int Engine::init(int gpuid, int tta_mode)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(RSR_E_DEVICE, "no HIP device available (this engine has no CPU fallback)");
    if (gpuid < 0 || gpuid >= n) return fail(RSR_E_DEVICE, "invalid gpu device " + std::to_string(gpuid));
    device = gpuid;
    tta = tta_mode ? 1 : 0;
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RSR_E_DEVICE, std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);
    num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(&zeros.p, 256));
    zeros.bytes = 256;
    HIP_TRY(hipMemsetAsync(zeros.p, 0, 256, stream)); // on OUR stream: it is non-blocking, the null stream does not order with it
    HIP_TRY(hipStreamSynchronize(stream));
    return RSR_OK;
}

int Engine::ensure(DevBuf& b, size_t bytes)
{
    if (b.bytes >= bytes && b.p) return RSR_OK;
    if (b.p)
    {
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(b.p);
        b.p = nullptr;
        b.bytes = 0;
    }
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess)
    {
        b.p = nullptr;
        return fail(RSR_E_NOMEM, "hipMalloc(" + std::to_string(bytes) + ") failed: " + hipGetErrorString(e));
    }
    b.bytes = bytes;
    return RSR_OK;
}

// ---- model --------------------------------------------------------------------------------
int Engine::load_blob_host(const void* data, size_t bytes)
{
    std::string e;
    int rc = check_packed(data, bytes, e);
    if (rc != RSR_OK) return fail(rc, e);
    HIP_TRY(hipSetDevice(device));
    rc = ensure(blob, bytes);
    if (rc != RSR_OK) return rc;
    HIP_TRY(hipMemcpy(blob.p, data, bytes, hipMemcpyHostToDevice));
    const PackedHeader* H = static_cast<const PackedHeader*>(data);
    const PackedConv* T = reinterpret_cast<const PackedConv*>(static_cast<const unsigned char*>(data) + sizeof(PackedHeader));
    convs.assign(T, T + H->nconv);
    loaded = true;
    return RSR_OK;
}

int Engine::load_blob_device(const void* data, size_t bytes)
{
    // header + table are small: fetch them to the host for validation, keep the payload on device
    if (bytes < sizeof(PackedHeader)) return fail(RSR_E_FORMAT, "packed blob too small");
    HIP_TRY(hipSetDevice(device));
    std::vector<unsigned char> head(sizeof(PackedHeader) + size_t(kNumConvs) * sizeof(PackedConv));
    if (bytes < head.size()) return fail(RSR_E_FORMAT, "packed blob too small");
    HIP_TRY(hipMemcpy(head.data(), data, head.size(), hipMemcpyDeviceToHost));
    const PackedHeader* H = reinterpret_cast<const PackedHeader*>(head.data());
    if (H->magic != kPackedMagic || H->version != 2 || H->nconv != uint32_t(kNumConvs) || H->total_bytes != bytes)
        return fail(RSR_E_FORMAT, "packed blob header mismatch");
    int rc = ensure(blob, bytes);
    if (rc != RSR_OK) return rc;
    HIP_TRY(hipMemcpy(blob.p, data, bytes, hipMemcpyDeviceToDevice));
    const PackedConv* T = reinterpret_cast<const PackedConv*>(head.data() + sizeof(PackedHeader));
    convs.assign(T, T + H->nconv);
    for (const PackedConv& c : convs)
        if (c.w_off >= bytes || c.b_off >= bytes || c.nt == 0 || c.nt > 2 || c.nplanes == 0)
            return fail(RSR_E_FORMAT, "packed blob conv table corrupt");
    loaded = true;
    return RSR_OK;
}

int Engine::load_files(const char* param, const char* bin)
{
    if (!param || !bin) return fail(RSR_E_ARG, "null path");
    Model m;
    std::string e;
    int rc = load_model(param, bin, m, e);
    if (rc != RSR_OK) return fail(rc, e);
    std::vector<unsigned char> buf(packed_size(m));
    rc = pack_model(m, buf.data(), buf.size(), e);
    if (rc != RSR_OK) return fail(rc, e);
    return load_blob_host(buf.data(), buf.size());
}

// ---- plan ---------------------------------------------------------------------------------
void Engine::free_plan()
{
    if (plan.d_tables)
    {
        (void)hipDeviceSynchronize();
        (void)hipFree(plan.d_tables);
    }
    plan = Plan();
}

static void make_items(Plan::Batch& b)
{
    for (int lvl = 0; lvl < 3; lvl++)
    {
        b.items[lvl].clear();
        b.px[lvl] = 0;
        for (int s = 0; s < b.nslots; s++)
        {
            const int H = b.dims[size_t(s)].h << lvl, W = b.dims[size_t(s)].w << lvl;
            b.px[lvl] += double(H) * W;
            for (int y0 = 0; y0 < H; y0 += kBlkH)
                for (int x0 = 0; x0 < W; x0 += kBlkW) b.items[lvl].push_back(WorkItem{s, y0, x0, H, W, 0, 0, 0});
        }
    }
}

static size_t al256(size_t v) { return (v + 255) & ~size_t(255); }

static size_t batch_table_bytes(const Plan::Batch& b)
{
    size_t n = al256(b.tiles.size() * sizeof(BaseTile)) + al256(b.dims.size() * sizeof(TileDim));
    for (int l = 0; l < 3; l++) n += 2 * al256(b.items[l].size() * sizeof(WorkItem));
    return n;
}

// per-slot workspace bytes per LR pixel: IN 64, FEA 128, 3 x RDB 384, T32 256, R32 256, UP1 4*128,
// UP2 16*128, HR 16*128, OUT3 16*6
static constexpr long long kBytesPerPx = 64 + 128 + 3 * 384 + 256 + 256 + 512 + 2048 + 2048 + 96;

int Engine::build_plan(int w, int h, int c)
{
    if (plan.w == w && plan.h == h && plan.c == c && plan.T == tilesize && plan.P == prepadding && plan.tta == tta &&
        !plan.batches.empty())
        return RSR_OK;
    free_plan();
    const int T = tilesize, P = prepadding;
    // tile grid: realsr.cpp:170-171; tile geometry: realsr.cpp:178-181,237,246-249
    const int xtiles = (w + T - 1) / T, ytiles = (h + T - 1) / T;
    std::vector<BaseTile> all;
    long long cap = 0;
    int mtw = 0, mth = 0;
    for (int yi = 0; yi < ytiles; yi++)
        for (int xi = 0; xi < xtiles; xi++)
        {
            const int twn = std::min((xi + 1) * T, w) - xi * T;
            const int thn = std::min((yi + 1) * T, h) - yi * T;
            BaseTile t;
            // Padded tile pixel (gx,gy) samples image pixel reflect101(gx + x_org), reflect101(gy + y_org).
            // The reference reflects against the uploaded row BAND (realsr_preproc.comp:56-62 with
            // crop_y = min(yi*T, P), realsr.cpp:404); the band starts at max(yi*T-P,0) and ends at
            // min((yi+1)*T+P, h), so reflection only ever triggers where the band edge IS the image edge:
            // reflecting against the image gives the same pixel, and the whole image can stay resident.
            t.x_org = xi * T - P;
            t.y_org = yi * T - P;
            t.tw = twn + 2 * P;
            t.th = thn + 2 * P;
            t.slot0 = 0;
            t.out_x = xi * T * scale;
            t.out_y = yi * T * scale;
            t.out_w = twn * scale;
            t.out_h = thn * scale;
            all.push_back(t);
            cap = std::max(cap, (long long)t.tw * t.th);
            mtw = std::max(mtw, t.tw);
            mth = std::max(mth, t.th);
        }
    // The kernels address a plane through 32-bit byte offsets (raw buffer resources, out-of-range sentinel 2^31): the
    // largest plane is the 4x level, 16 * cap pixels * 64 B.  Tiles beyond that (~1,400 px) must be split by the caller.
    if (cap * 16 * 64 >= (1ll << 31))
        return fail(RSR_E_ARG, "tilesize too large: a padded tile may have at most 2,097,151 pixels (e.g. -t 1400)");
    const int per = tta ? 8 : 1;
    const long long per_slot = cap * kBytesPerPx;
    long long budget_slots = (max_workspace_mb * 1024 * 1024) / std::max<long long>(per_slot, 1);
    budget_slots = std::max<long long>(per, budget_slots / per * per);
    const long long total_slots = (long long)all.size() * per;
    const int spb = int(std::min<long long>(total_slots, budget_slots));
    const int tiles_per_batch = spb / per;

    plan.w = w; plan.h = h; plan.c = c; plan.T = T; plan.P = P; plan.tta = tta;
    plan.cap_px = cap;
    plan.max_tw = mtw;
    plan.max_th = mth;
    plan.slots_per_batch = spb;
    size_t table_bytes = 0;
    for (size_t t0 = 0; t0 < all.size(); t0 += size_t(tiles_per_batch))
    {
        Plan::Batch b;
        b.tile0 = int(t0);
        b.ntiles = int(std::min(all.size() - t0, size_t(tiles_per_batch)));
        b.nslots = b.ntiles * per;
        for (int i = 0; i < b.ntiles; i++)
        {
            BaseTile t = all[t0 + size_t(i)];
            t.slot0 = i * per;
            b.tiles.push_back(t);
            for (int k = 0; k < per; k++)
                b.dims.push_back(k < 4 ? TileDim{t.th, t.tw} : TileDim{t.tw, t.th}); // realsr.cpp:251-258
        }
        make_items(b);
        table_bytes += batch_table_bytes(b);
        plan.batches.push_back(std::move(b));
    }
    HIP_TRY(hipMalloc(&plan.d_tables, table_bytes));
    char* d = static_cast<char*>(plan.d_tables);
    for (Plan::Batch& b : plan.batches)
    {
        auto put = [&](const void* src, size_t bytes) -> void* {
            void* at = d;
            if (bytes) (void)hipMemcpy(at, src, bytes, hipMemcpyHostToDevice);
            d += al256(bytes);
            return at;
        };
        b.d_tiles = static_cast<BaseTile*>(put(b.tiles.data(), b.tiles.size() * sizeof(BaseTile)));
        b.d_dims = static_cast<TileDim*>(put(b.dims.data(), b.dims.size() * sizeof(TileDim)));
        for (int l = 0; l < 3; l++)
        {
            b.d_items[l] = static_cast<WorkItem*>(put(b.items[l].data(), b.items[l].size() * sizeof(WorkItem)));
            std::vector<WorkItem> rev(b.items[l].rbegin(), b.items[l].rend());
            b.d_items_rev[l] = static_cast<WorkItem*>(put(rev.data(), rev.size() * sizeof(WorkItem)));
        }
    }
    HIP_TRY(hipGetLastError());
    return RSR_OK;
}

// Every fp16 plane that a convolution may read carries a 64-byte zero GUARD in front of pixel 0 (plane = [guard |
// H*W*64 B]).  The LDS-DMA loaders address a plane as (uniform base, 32-bit lane offset); lanes whose patch pixel
// lies outside the image use offset 0 = the guard, i.e. conv zero padding without a second base pointer.  Guards
// are zeroed when a buffer is (re)allocated and never written afterwards.
int Engine::ensure_zeroed(DevBuf& b, size_t bytes, bool layout_changed, hipStream_t st)
{
    const void* before = b.p;
    const size_t had = b.bytes;
    const int rc = ensure(b, bytes);
    if (rc != RSR_OK) return rc;
    // a new plane stride moves the guards onto bytes that used to hold pixels: zero the used extent again
    if (b.p != before || b.bytes != had || layout_changed) HIP_TRY(hipMemsetAsync(b.p, 0, bytes, st)); // same stream as the kernels that follow
    return RSR_OK;
}

int Engine::ensure_workspace(int nslots, long long cap, hipStream_t st)
{
    const size_t n = size_t(nslots), c = size_t(cap), G = size_t(kGuard);
    const bool lc = (cap != ws_cap_px); // plane stride (hence guard positions) depends on the slot capacity
    int rc;
    if ((rc = ensure_zeroed(b_in, n * (c * 64 + G), lc, st)) != RSR_OK) return rc;
    if ((rc = ensure_zeroed(b_fea, n * 2 * (c * 64 + G), lc, st)) != RSR_OK) return rc;
    for (int i = 0; i < 3; i++)
        if ((rc = ensure_zeroed(b_rdb[i], n * 6 * (c * 64 + G), lc, st)) != RSR_OK) return rc;
    if (trunk_fp32)
    {
        if ((rc = ensure(b_t32, n * c * 256)) != RSR_OK) return rc;
        if ((rc = ensure(b_r32, n * c * 256)) != RSR_OK) return rc;
    }
    if ((rc = ensure_zeroed(b_up1, n * 2 * (c * 256 + G), lc, st)) != RSR_OK) return rc;
    if ((rc = ensure_zeroed(b_up2, n * 2 * (c * 1024 + G), lc, st)) != RSR_OK) return rc;
    if ((rc = ensure_zeroed(b_hr, n * 2 * (c * 1024 + G), lc, st)) != RSR_OK) return rc;
    if ((rc = ensure(b_out3, n * c * 96)) != RSR_OK) return rc;
    ws_slots = nslots;
    ws_cap_px = cap;
    return RSR_OK;
}

// ---- profiling ------------------------------------------------------------------------------
// mark_begin() records the opening event of a call; mark() records an event AFTER a launch and labels
// the segment [previous event, this event] with the launch's class.  Events sit on the launch stream.
static hipEvent_t next_event(Engine& e)
{
    if (e.ev_used == e.ev_pool.size())
    {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return nullptr;
        e.ev_pool.push_back(ev);
    }
    return e.ev_pool[e.ev_used++];
}

void Engine::mark_begin(hipStream_t st)
{
    if (!profiling) return;
    ev_used = 0;
    segs.clear();
    if (hipEvent_t ev = next_event(*this)) (void)hipEventRecord(ev, st);
}

void Engine::mark(int cls, double flops, double bytes, hipStream_t st, int conv_index)
{
    if (!profiling || ev_used == 0) return;
    if (hipEvent_t ev = next_event(*this))
    {
        (void)hipEventRecord(ev, st);
        segs.push_back(Seg{cls, flops, bytes, conv_index});
    }
}

void Engine::collect_profile(hipStream_t st)
{
    if (!profiling || ev_used < 2) { ev_used = 0; segs.clear(); return; }
    (void)hipStreamSynchronize(st);
    for (size_t i = 0; i < segs.size() && i + 1 < ev_used; i++)
    {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev_pool[i], ev_pool[i + 1]) != hipSuccess) continue;
        const Seg& s = segs[i];
        if (s.cls == 1)
        {
            prof.conv_ms += ms;
            prof.conv_flops += s.flops;
            prof.conv_launches++;
            if (s.conv_index >= 0 && s.conv_index < kNumConvs) conv_ms_by_index[s.conv_index] += ms;
        }
        else if (s.cls == 0) { prof.pre_ms += ms; prof.pre_bytes += s.bytes; }
        else if (s.cls == 2) { prof.post_ms += ms; prof.post_bytes += s.bytes; }
    }
    float tot = 0.f;
    if (hipEventElapsedTime(&tot, ev_pool[0], ev_pool[ev_used - 1]) == hipSuccess) prof.total_ms += tot;
    ev_used = 0;
    segs.clear();
}

// ---- the network schedule ---------------------------------------------------------------------
// x4.param as a fused schedule.  Buffers (per slot, 32-channel planes):
//   IN(1)  FEA(2)  RDB[3] x {x(2) + dense x1..x4 (4)}  T32/R32 (fp32 trunk / RRDB input, 2 planes each)
//   UP1(2 @2x)  UP2(2 @4x)  HR(2 @4x)  OUT3 (planar [3][4H][4W] fp16)
// Concat never happens: conv k of a dense block reads planes [x | x1..x_{k-1}] in place and writes its
// 32 channels as plane x_k.  Eltwise/BinaryOp/Interp are epilogue or staging-address variants.
void Engine::run_network(const Plan::Batch& b, hipStream_t st)
{
    const long long cap = ws_cap_px;
    const long long pb16 = cap * 64 + kGuard, pb32 = cap * 128; // fp16 planes are guarded (see ensure_workspace)
    auto PS = [](const DevBuf& buf, long long planes_per_slot, long long plane_bytes, int plane_off) {
        PlaneSrc s; // base = pixel 0 of plane `plane_off` of slot 0
        s.base = static_cast<const char*>(buf.p) + (long long)plane_off * plane_bytes + kGuard;
        s.slot_stride = planes_per_slot * plane_bytes;
        s.plane_stride = plane_bytes;
        return s;
    };
    auto PS32 = [](const DevBuf& buf, long long planes_per_slot, long long plane_bytes, int plane_off) {
        PlaneSrc s; // fp32 trunk planes: never a DMA source, no guard
        s.base = static_cast<const char*>(buf.p) + (long long)plane_off * plane_bytes;
        s.slot_stride = planes_per_slot * plane_bytes;
        s.plane_stride = plane_bytes;
        return s;
    };
    const PlaneSrc none{nullptr, 0, 0};
    const char* blobp = static_cast<const char*>(blob.p);
    int ci = 0;
    auto base_args = [&](int lvl_in, int lvl_out) {
        ConvArgs a;
        std::memset(&a, 0, sizeof a);
        const PackedConv& c = convs[size_t(ci)];
        a.wpk = blobp + c.w_off;
        a.wfrag = blobp + c.wf_off;
        a.bias = reinterpret_cast<const float*>(blobp + c.b_off);
        a.lrelu = (c.act == 2);
        a.lvl_in = lvl_in;
        a.lvl_out = lvl_out;
        // Boustrophedon: every second conv walks the blocks backwards, so it starts on the tiles the previous conv
        // produced (and read) last -- those are still in the 256 MB Infinity Cache; walking forwards again would start
        // on the least recently used data.
        a.items = (alternate_order && (ci & 1) && b.d_items_rev[lvl_out]) ? b.d_items_rev[lvl_out] : b.d_items[lvl_out];
        a.nitems = int(b.items[lvl_out].size());
        a.dims = b.d_dims;
        a.zeros = zeros.p;
        a.dbg = dbg;
        a.stagger = stagger_unit * int(c.nplanes + 2);
        a.trace = (trace_conv == ci) ? static_cast<unsigned long long*>(trace_buf.p) : nullptr;
        a.s1 = a.s2 = 1.f;
        return a;
    };
    auto go = [&](ConvArgs& a) {
        const PackedConv& c = convs[size_t(ci)];
        if (kernel_version == 3 && (c.nt == 1 || ring_nt2) && launch_conv_ring(a, int(c.nt), num_cu, st)) {}
        else if (kernel_version >= 2) launch_conv_pipe(a, int(c.nt), num_cu, st);
        else launch_conv(a, int(c.nt), use_dma, st);
        mark(1, 2.0 * 9.0 * c.cin * c.cout * b.px[a.lvl_out], 0, st, ci);
        ci++;
    };
    const PlaneSrc fea = PS(b_fea, 2, pb16, 0);
    const PlaneSrc t32 = PS32(b_t32, 2, pb32, 0), r32 = PS32(b_r32, 2, pb32, 0);
    auto rdb_x = [&](int i) { return PS(b_rdb[i], 6, pb16, 0); };
    auto rdb_d = [&](int i, int k) { return PS(b_rdb[i], 6, pb16, 2 + k); };

    { // conv_first (x4.param:4): IN -> FEA (+ fp32 trunk copies)
        ConvArgs a = base_args(0, 0);
        a.src0 = PS(b_in, 1, pb16, 0);
        a.n0 = 1;
        a.out16 = fea;
        if (trunk_fp32) { a.out32a = t32; a.out32b = r32; }
        go(a);
    }
    for (int j = 0; j < kNumRDB; j++)
    {
        const int bi = j % 3;
        const PlaneSrc xs = (j == 0) ? fea : rdb_x(bi);
        for (int k = 0; k < 4; k++)
        { // x_{k+1} = lrelu(conv([x, x1..xk]))   (x4.param:6,9,12,15)
            ConvArgs a = base_args(0, 0);
            a.src0 = xs; a.n0 = 2;
            a.src1 = rdb_d(bi, 0); a.n1 = k;
            a.out16 = rdb_d(bi, k);
            go(a);
        }
        // x5 = conv([x, x1..x4]);  out = 0.2*x5 + x   (x4.param:17-18);  every third block additionally
        // out = 0.2*out + rrdb_in   (x4.param:47, Eltwise 0=1 -23301=2,0.2,1.0)
        ConvArgs a = base_args(0, 0);
        a.src0 = xs; a.n0 = 2;
        a.src1 = rdb_d(bi, 0); a.n1 = 4;
        a.s1 = 0.2f;
        if (trunk_fp32) { a.res1 = t32; a.res1_kind = 2; a.out32a = t32; }
        else { a.res1 = xs; a.res1_kind = 1; a.res1_in_acc = !(dbg & 4096); a.res1_coef = 5.f; } // 1/0.2, exact in fp16
        if (bi == 2)
        {
            a.s2 = 0.2f;
            if (trunk_fp32) { a.res2 = r32; a.res2_kind = 2; a.out32b = r32; }
            else { a.res2 = (j == 2) ? fea : rdb_x(0); a.res2_kind = 1; }
        }
        a.out16 = rdb_x((j + 1) % 3);
        go(a);
    }
    { // trunk_conv + global skip: fea + conv(trunk)   (x4.param:994-995)
        ConvArgs a = base_args(0, 0);
        a.src0 = rdb_x(kNumRDB % 3); a.n0 = 2;
        a.res1 = fea; a.res1_kind = 1; a.s1 = 1.f;
        a.out16 = rdb_x(1);
        go(a);
    }
    const PlaneSrc up1 = PS(b_up1, 2, cap * 256 + kGuard, 0), up2 = PS(b_up2, 2, cap * 1024 + kGuard, 0), hr = PS(b_hr, 2, cap * 1024 + kGuard, 0);
    { // nearest x2 + upconv1 + lrelu   (x4.param:996-997)
        ConvArgs a = base_args(0, 1);
        a.src0 = rdb_x(1); a.n0 = 2;
        a.out16 = up1;
        go(a);
    }
    { // nearest x2 + upconv2 + lrelu   (x4.param:998-999)
        ConvArgs a = base_args(1, 2);
        a.src0 = up1; a.n0 = 2;
        a.out16 = up2;
        go(a);
    }
    { // HRconv + lrelu   (x4.param:1000)
        ConvArgs a = base_args(2, 2);
        a.src0 = up2; a.n0 = 2;
        a.out16 = hr;
        go(a);
    }
    { // conv_last 64 -> 3   (x4.param:1001), planar fp16 output = the reference's `output` blob
        ConvArgs a = base_args(2, 2);
        a.src0 = hr; a.n0 = 2;
        a.out_planar3 = b_out3.p;
        a.planar3_slot_stride = cap * 96;
        go(a);
    }
    (void)none;
}

// ---- process ----------------------------------------------------------------------------------
int Engine::process_device(const void* d_in, int w, int h, int c, void* d_out, hipStream_t st, bool sync)
{
    if (!loaded) return fail(RSR_E_STATE, "process before load");
    if (!d_in || !d_out || w < 1 || h < 1 || (c != 3 && c != 4)) return fail(RSR_E_ARG, "bad image arguments");
    if (scale != 4) return fail(RSR_E_ARG, "only scale 4 is supported (main.cpp:533-537)");
    std::lock_guard<std::mutex> lk(mu);
    HIP_TRY(hipSetDevice(device));
    if (!st) st = stream;
    int rc = build_plan(w, h, c);
    if (rc != RSR_OK) return rc;
    rc = ensure_workspace(plan.slots_per_batch, plan.cap_px, st);
    if (rc != RSR_OK) return rc;
    mark_begin(st);
    for (const Plan::Batch& b : plan.batches)
    {
        PreArgs pa;
        pa.img = static_cast<const uint8_t*>(d_in);
        pa.w = w; pa.h = h; pa.c = c;
        pa.tiles = b.d_tiles;
        pa.ntiles = b.ntiles;
        pa.tta = tta;
        pa.in_plane = static_cast<char*>(b_in.p) + kGuard;
        pa.slot_stride = plan.cap_px * 64 + kGuard;
        pa.bgr = 0;
        launch_preproc_tiles(pa, plan.max_tw, plan.max_th, st);
        mark(0, 0, b.px[0] / (tta ? 8 : 1) * c + b.px[0] * 64, st);
        run_network(b, st);
        PostArgs po;
        po.planar3 = b_out3.p;
        po.slot_stride = plan.cap_px * 96;
        po.tiles = b.d_tiles;
        po.ntiles = b.ntiles;
        po.tta = tta;
        po.crop = prepadding * scale;
        po.out = static_cast<uint8_t*>(d_out);
        po.out_w = w * scale; po.out_h = h * scale; po.c = c;
        po.in_img = static_cast<const uint8_t*>(d_in);
        po.in_w = w; po.in_h = h;
        po.tilesize = tilesize;
        po.bgr = 0;
        launch_postproc_tiles(po, (plan.max_tw - 2 * prepadding) * scale, (plan.max_th - 2 * prepadding) * scale, st);
        mark(2, 0, b.px[2] / (tta ? 8 : 1) * (6.0 * (tta ? 8 : 1) + c), st);
    }
    HIP_TRY(hipGetLastError());
    if (profiling)
    {
        collect_profile(st);
        prof.calls++;
        for (const Plan::Batch& b : plan.batches) prof.tiles += b.nslots;
    }
    else if (sync) HIP_TRY(hipStreamSynchronize(st));
    return RSR_OK;
}

int Engine::process_host(const uint8_t* in, int w, int h, int c, uint8_t* out)
{
    if (!in || !out || w < 1 || h < 1 || (c != 3 && c != 4)) return fail(RSR_E_ARG, "bad image arguments");
    const size_t nin = size_t(w) * h * c, nout = nin * size_t(scale) * scale;
    {
        std::lock_guard<std::mutex> lk(mu);
        HIP_TRY(hipSetDevice(device));
        int rc;
        if ((rc = ensure(d_img_in, nin)) != RSR_OK) return rc;
        if ((rc = ensure(d_img_out, nout)) != RSR_OK) return rc;
        HIP_TRY(hipMemcpyAsync(d_img_in.p, in, nin, hipMemcpyHostToDevice, stream));
    }
    int rc = process_device(d_img_in.p, w, h, c, d_img_out.p, stream, false);
    if (rc != RSR_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out, d_img_out.p, nout, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return RSR_OK;
}

// one tile through the network only (layer-level parity hook)
int Engine::net_forward(const uint16_t* in, int w, int h, uint16_t* out)
{
    if (!loaded) return fail(RSR_E_STATE, "net_forward before load");
    if (!in || !out || w < 1 || h < 1) return fail(RSR_E_ARG, "bad arguments");
    std::lock_guard<std::mutex> lk(mu);
    HIP_TRY(hipSetDevice(device));
    free_plan();
    Plan::Batch b;
    b.ntiles = 1;
    b.nslots = 1;
    b.dims.push_back(TileDim{h, w});
    make_items(b);
    plan.cap_px = (long long)w * h;
    plan.batches.clear();
    size_t tb = batch_table_bytes(b);
    HIP_TRY(hipMalloc(&plan.d_tables, tb));
    char* d = static_cast<char*>(plan.d_tables);
    b.d_dims = reinterpret_cast<TileDim*>(d);
    HIP_TRY(hipMemcpy(d, b.dims.data(), sizeof(TileDim), hipMemcpyHostToDevice));
    d += 256;
    for (int l = 0; l < 3; l++)
    {
        b.d_items[l] = reinterpret_cast<WorkItem*>(d);
        HIP_TRY(hipMemcpy(d, b.items[l].data(), b.items[l].size() * sizeof(WorkItem), hipMemcpyHostToDevice));
        d += al256(b.items[l].size() * sizeof(WorkItem));
    }
    int rc = ensure_workspace(1, plan.cap_px, stream);
    if (rc != RSR_OK) return rc;
    const size_t npx = size_t(w) * h;
    DevBuf tmp;
    if ((rc = ensure(tmp, npx * 6)) != RSR_OK) return rc;
    HIP_TRY(hipMemcpy(tmp.p, in, npx * 6, hipMemcpyHostToDevice));
    launch_planar3_to_plane(static_cast<const uint16_t*>(tmp.p), w, h, static_cast<char*>(b_in.p) + kGuard, stream);
    const bool was = profiling;
    profiling = false;
    run_network(b, stream);
    profiling = was;
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, b_out3.p, npx * 16 * 6, hipMemcpyDeviceToHost));
    (void)hipFree(tmp.p);
    free_plan();
    return RSR_OK;
}

// one convolution with caller-supplied weights (layer-level parity hook, include/realsr_hip.h rsr_conv3x3)
int Engine::conv_test(const uint16_t* in, int cin, int h, int w, int ups, const float* weight, const float* bias, int cout,
                      int lrelu, uint16_t* out)
{
    if (!in || !weight || !bias || !out || cin < 1 || cout < 1 || cout > 64 || h < 1 || w < 1) return fail(RSR_E_ARG, "bad arguments");
    std::lock_guard<std::mutex> lk(mu);
    HIP_TRY(hipSetDevice(device));
    Model m;
    ConvRec c;
    c.cin = cin; c.cout = cout; c.act = lrelu ? 2 : 0; c.slope = 0.2f;
    c.weight.assign(weight, weight + size_t(cin) * cout * 9);
    c.bias.assign(bias, bias + cout);
    m.convs.push_back(c);
    std::vector<unsigned char> pk(packed_size(m));
    std::string e;
    int rc = pack_model(m, pk.data(), pk.size(), e);
    if (rc != RSR_OK) return fail(rc, e);
    const PackedConv pc = *reinterpret_cast<const PackedConv*>(pk.data() + sizeof(PackedHeader));
    const int np = int(pc.nplanes), nt = int(pc.nt);
    const int H = ups ? 2 * h : h, W = ups ? 2 * w : w;
    const size_t ipx = size_t(h) * w, opx = size_t(H) * W;
    // planar [cin][h][w] -> planes [np][h][w][32]
    const size_t ipl = ipx * 32 + kGuard / 2; // halfs per guarded input plane
    std::vector<uint16_t> hin(size_t(np) * ipl, 0), hout(size_t(nt) * opx * 32, 0);
    for (int ch = 0; ch < cin; ch++)
        for (size_t p = 0; p < ipx; p++) hin[size_t(ch / 32) * ipl + kGuard / 2 + p * 32 + size_t(ch % 32)] = in[size_t(ch) * ipx + p];
    DevBuf d_w, d_in, d_out, d_tab;
    std::vector<WorkItem> items;
    for (int y0 = 0; y0 < H; y0 += kBlkH)
        for (int x0 = 0; x0 < W; x0 += kBlkW) items.push_back(WorkItem{0, y0, x0, H, W, 0, 0, 0});
    const TileDim td{h, w};
    auto cleanup = [&]() {
        for (DevBuf* b : {&d_w, &d_in, &d_out, &d_tab})
            if (b->p) (void)hipFree(b->p);
    };
    if ((rc = ensure(d_w, pk.size())) != RSR_OK || (rc = ensure(d_in, hin.size() * 2)) != RSR_OK ||
        (rc = ensure(d_out, hout.size() * 2)) != RSR_OK || (rc = ensure(d_tab, 256 + items.size() * sizeof(WorkItem))) != RSR_OK)
    {
        cleanup();
        return rc;
    }
    (void)hipMemcpy(d_w.p, pk.data(), pk.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_in.p, hin.data(), hin.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemsetAsync(d_out.p, 0, hout.size() * 2, stream);
    (void)hipMemcpy(d_tab.p, &td, sizeof td, hipMemcpyHostToDevice);
    (void)hipMemcpy(static_cast<char*>(d_tab.p) + 256, items.data(), items.size() * sizeof(WorkItem), hipMemcpyHostToDevice);
    ConvArgs a;
    std::memset(&a, 0, sizeof a);
    a.src0 = PlaneSrc{static_cast<char*>(d_in.p) + kGuard, 0, (long long)ipx * 64 + kGuard};
    a.n0 = np;
    a.lvl_in = 0;
    a.lvl_out = ups ? 1 : 0;
    a.wpk = static_cast<const char*>(d_w.p) + pc.w_off;
    a.wfrag = static_cast<const char*>(d_w.p) + pc.wf_off;
    a.bias = reinterpret_cast<const float*>(static_cast<const char*>(d_w.p) + pc.b_off);
    a.lrelu = lrelu;
    a.s1 = a.s2 = 1.f;
    a.out16 = PlaneSrc{d_out.p, 0, (long long)opx * 64};
    a.items = reinterpret_cast<const WorkItem*>(static_cast<const char*>(d_tab.p) + 256);
    a.nitems = int(items.size());
    a.dims = static_cast<const TileDim*>(d_tab.p);
    a.zeros = zeros.p;
    if (kernel_version == 3 && (nt == 1 || ring_nt2) && launch_conv_ring(a, nt, num_cu, stream)) {}
    else if (kernel_version >= 2) launch_conv_pipe(a, nt, num_cu, stream);
    else launch_conv(a, nt, use_dma, stream);
    hipError_t he = hipStreamSynchronize(stream);
    if (he == hipSuccess) he = hipGetLastError();
    if (he == hipSuccess) he = hipMemcpy(hout.data(), d_out.p, hout.size() * 2, hipMemcpyDeviceToHost);
    cleanup();
    if (he != hipSuccess) return fail(RSR_E_DEVICE, std::string("conv_test: ") + hipGetErrorString(he));
    for (int ch = 0; ch < cout; ch++)
        for (size_t p = 0; p < opx; p++) out[size_t(ch) * opx + p] = hout[(size_t(ch / 32) * opx + p) * 32 + size_t(ch % 32)];
    return RSR_OK;
}

} // namespace rsr
