// engine.cpp -- host side of the tiled engine (compiled by hipcc as HIP host code).  See engine.h.
#include "engine.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rsr {

static thread_local std::string tl_err;
const char* last_error() { return tl_err.c_str(); }

#define HIP_TRY(expr)                                                                              \
    do                                                                                             \
    {                                                                                              \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(RSR_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

int Engine::fail(int code, const std::string& msg)
{
    tl_err = msg;
    return code;
}

CopyPool::~CopyPool()
{
    {
        std::lock_guard<std::mutex> lk(m);
        stop = true;
    }
    cv.notify_all();
    for (std::thread& t : workers) t.join();
}

void CopyPool::copy(void* dst, const void* src, size_t n, int threads)
{
    const size_t kMin = size_t(1) << 20; // pieces below 1 MiB are not worth a hand-off
    int parts = int(std::min<size_t>(size_t(std::max(threads, 1)), n / kMin));
    if (parts <= 1)
    {
        std::memcpy(dst, src, n);
        return;
    }
    const size_t piece = ((n + parts - 1) / parts + 4095) & ~size_t(4095);
    int pending = 0;
    {
        std::lock_guard<std::mutex> lk(m);
        while (int(workers.size()) < threads - 1)
            workers.emplace_back([this]() {
                std::unique_lock<std::mutex> lk(m);
                for (;;)
                {
                    cv.wait(lk, [this]() { return stop || !q.empty(); });
                    if (q.empty()) return; // stop
                    const Job j = q.front();
                    q.pop_front();
                    lk.unlock();
                    std::memcpy(j.dst, j.src, j.n);
                    lk.lock();
                    if (--*j.pending == 0) done.notify_all();
                }
            });
        for (size_t off = piece; off < n; off += piece)
        {
            q.push_back(Job{static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, std::min(piece, n - off), &pending});
            pending++;
        }
    }
    cv.notify_all();
    std::memcpy(dst, src, std::min(piece, n));
    std::unique_lock<std::mutex> lk(m);
    done.wait(lk, [&]() { return pending == 0; });
}

Engine::~Engine()
{
    if (device >= 0) (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    for (auto& l : lanes)
    {
        if (l->copy) (void)hipStreamSynchronize(l->copy);
        if (l->d_in.p) (void)hipFree(l->d_in.p);
        if (l->d_out.p) (void)hipFree(l->d_out.p);
        if (l->h_in) (void)hipHostFree(l->h_in);
        if (l->h_out) (void)hipHostFree(l->h_out);
        for (hipEvent_t e : {l->ev_in, l->ev_done, l->ev_half, l->ev_chunk[0], l->ev_chunk[1]})
            if (e) (void)hipEventDestroy(e);
        if (l->copy) (void)hipStreamDestroy(l->copy);
    }
    free_plans();
    DevBuf* all[] = {&blob, &zeros, &b_in, &b_fea, &b_rdb[0], &b_rdb[1], &b_rdb[2], &b_up1, &b_up2, &b_hr, &b_out3, &trace_buf};
    for (DevBuf* b : all)
        if (b->p) (void)hipFree(b->p);
    for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : sync_events) (void)hipEventDestroy(e);
    if (merge_mid) (void)hipEventDestroy(merge_mid);
    if (merge_done) (void)hipEventDestroy(merge_done);
    for (int k = 0; k < 3; k++)
    {
        if (mix_ev[k]) (void)hipEventDestroy(mix_ev[k]);
        if (mix_tab[k].p) (void)hipFree(mix_tab[k].p);
    }
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
    HIP_TRY(kernels_init_device()); // per-device opt-in to > 64 KiB dynamic LDS for every kernel instantiation
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(&zeros.p, 256));
    zeros.bytes = 256;
    HIP_TRY(hipMemsetAsync(zeros.p, 0, 256, stream)); // on OUR stream: it is non-blocking, the null stream does not order with it
    HIP_TRY(hipStreamSynchronize(stream));
    return RSR_OK;
}

// Grow-only device buffer.  Growing frees the old allocation: the caller guarantees nothing in flight uses it
// (workspace: the compute stream is drained first; lane buffers: the lane is owned by the caller).
int Engine::ensure(DevBuf& b, size_t bytes)
{
    if (b.bytes >= bytes && b.p) return RSR_OK;
    if (b.p)
    {
        (void)hipFree(b.p);
        b.p = nullptr;
        b.bytes = 0;
    }
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess)
    {
        b.p = nullptr;
        (void)hipGetLastError();
        return fail(RSR_E_NOMEM, "hipMalloc(" + std::to_string(bytes) + ") failed: " + hipGetErrorString(e));
    }
    b.bytes = bytes;
    return RSR_OK;
}

hipEvent_t Engine::take_event()
{
    if (!sync_events.empty())
    {
        hipEvent_t e = sync_events.back();
        sync_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
}

hipEvent_t Engine::take_event_timed()
{
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void Engine::give_event(hipEvent_t e)
{
    if (e) sync_events.push_back(e);
}

// ---- model --------------------------------------------------------------------------------
int Engine::adopt_table(const unsigned char* head, size_t bytes)
{
    std::string e;
    const int rc = check_packed(head, packed_table_bytes(), bytes, e);
    if (rc != RSR_OK) return fail(rc, e);
    return RSR_OK;
}

int Engine::load_blob_host(const void* data, size_t bytes)
{
    if (!data || bytes < packed_table_bytes()) return fail(RSR_E_FORMAT, "packed blob too small");
    // validate BEFORE touching engine state: a failed load leaves a previously loaded model intact
    int rc = adopt_table(static_cast<const unsigned char*>(data), bytes);
    if (rc != RSR_OK) return rc;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamSynchronize(stream));
    DevBuf fresh;
    if ((rc = ensure(fresh, bytes)) != RSR_OK) return rc;
    hipError_t he = hipMemcpy(fresh.p, data, bytes, hipMemcpyHostToDevice);
    if (he != hipSuccess)
    {
        (void)hipFree(fresh.p);
        return fail(RSR_E_DEVICE, std::string("blob upload: ") + hipGetErrorString(he));
    }
    if (blob.p) (void)hipFree(blob.p);
    blob = fresh;
    const PackedHeader* H = static_cast<const PackedHeader*>(data);
    const PackedConv* T = reinterpret_cast<const PackedConv*>(static_cast<const unsigned char*>(data) + sizeof(PackedHeader));
    convs.assign(T, T + H->nconv);
    loaded = true;
    return RSR_OK;
}

int Engine::load_blob_device(const void* data, size_t bytes)
{
    // header + table are small: fetch them to the host for validation, keep the payload on device
    if (!data || bytes < packed_table_bytes()) return fail(RSR_E_FORMAT, "packed blob too small");
    HIP_TRY(hipSetDevice(device));
    std::vector<unsigned char> head(packed_table_bytes());
    HIP_TRY(hipMemcpy(head.data(), data, head.size(), hipMemcpyDeviceToHost));
    int rc = adopt_table(head.data(), bytes);
    if (rc != RSR_OK) return rc;
    HIP_TRY(hipStreamSynchronize(stream));
    DevBuf fresh;
    if ((rc = ensure(fresh, bytes)) != RSR_OK) return rc;
    hipError_t he = hipMemcpy(fresh.p, data, bytes, hipMemcpyDeviceToDevice);
    if (he != hipSuccess)
    {
        (void)hipFree(fresh.p);
        return fail(RSR_E_DEVICE, std::string("blob copy: ") + hipGetErrorString(he));
    }
    if (blob.p) (void)hipFree(blob.p);
    blob = fresh;
    const PackedHeader* H = reinterpret_cast<const PackedHeader*>(head.data());
    const PackedConv* T = reinterpret_cast<const PackedConv*>(head.data() + sizeof(PackedHeader));
    convs.assign(T, T + H->nconv);
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
void Engine::free_plans()
{
    if (plans.empty()) return;
    if (stream) (void)hipStreamSynchronize(stream); // kernels in flight read the tables
    for (Plan& p : plans)
        if (p.d_tables) (void)hipFree(p.d_tables);
    plans.clear();
}

// Dead-output elimination behind the trunk.  The caller keeps only the un-padded rectangle of a tile's output: pixels closer
// than crop4 = prepadding * scale to the tile border are cropped (realsr_postproc.comp:62-69, realsr.cpp:460-461).  Walking
// back from there, every 3x3 conv needs one more ring of its input: conv_last's output is read from margin crop4 inwards,
// HRconv's from crop4 - 1, upconv2's from crop4 - 2; upconv2 reads the nearest-x2 of upconv1's output (2x level) from
// (crop4 - 3) >> 1 = crop4 / 2 - 2, and so on through the last LR convs.  Output pixels outside those regions influence
// nothing that is kept, so their blocks are left out of the work-item tables (block granularity, here) and their 4-row groups
// are skipped by the MFMA waves (ConvArgs::margin).  The bytes of the kept rectangle are unchanged -- this is what a
// compiler calls dead-code elimination; the algorithmic FLOP count still counts the halo densely.
static int tail_margin(int crop4, int which) // which: 0 conv_last, 1 HRconv, 2 upconv2 (4x level), 3 upconv1 (2x level), 4.. LR convs backwards
{
    if (crop4 <= 0) return 0;
    int m;
    if (which <= 2) m = crop4 - which;
    else
    {
        m = crop4 / 2 - 2; // upconv1: (crop4 - 3) >> 1 below, the same above for the even crops prepadding * 4 gives
        if (which >= 4)
        {
            m = (m - 1) >> 1;  // trunk_conv: read by upconv1 through the nearest-x2 gather
            m -= (which - 4);  // RDB 69 conv5, conv4, ... each one ring further out
        }
    }
    return m > 0 ? m : 0;
}

// The work items of one H x W tile plane: 16 x 32 blocks row-major; blocks wholly inside the margin `m` (dead-output elimination)
// are left out.  With `fold` a last column of 1..kFoldMaxW pixels is covered by FOLDED items (kernels.h), one per pair of block rows,
// placed behind the second row of the pair: C3's 420-wide tiles 13 columns + 14 folded items instead of 14 x 27 blocks (-3.4 %),
// the 140-wide edge tiles of C2 4 columns + 7 folded items instead of 5 x 14 (-10 %).  Tile geometry: realsr.cpp:170-181,246-249.
// The kernel decides "these four rows need no matrix work" (wave_is_dead) from a folded block's FIRST strip alone; that is exact for
// rows below the tile and for the bottom margin, and it would be wrong where the first strip lies in the TOP margin and the second
// does not -- so block rows that start above `mtop` (the largest margin any conv of this level is launched with) stay plain blocks and
// the pairing begins below them.
void append_block_items(std::vector<WorkItem>& out, int slot, int H, int W, int m, int mtop, int p0, int p1, int p2, bool fold)
{
    const int rem = W % kBlkW;
    const bool fold_last = fold && rem >= 1 && rem <= kFoldMaxW;
    const int wfull = fold_last ? W - rem : W; // columns covered by plain blocks in every row
    auto row_skipped = [&](int y0) { return y0 + kBlkH <= m || y0 >= H - m; };
    auto plain_row = [&](int y0, int x_from, int x_to) {
        for (int x0 = x_from; x0 < x_to; x0 += kBlkW)
        {
            if (row_skipped(y0) || x0 + kBlkW <= m || x0 >= W - m) continue; // nothing kept depends on it
            out.push_back(WorkItem{slot, y0, x0, H, W, p0, p1, p2});
        }
    };
    // block rows above the first one that starts at or below `mtop` stay plain; pairs are counted from there
    const int first_pair = fold_last ? std::min(H, (mtop + kBlkH - 1) / kBlkH * kBlkH) : H;
    for (int y0 = 0; y0 < first_pair; y0 += kBlkH) plain_row(y0, 0, W);
    for (int y0 = first_pair; y0 < H; y0 += 2 * kBlkH) // pairs of block rows
    {
        const bool has2 = y0 + kBlkH < H;
        plain_row(y0, 0, wfull);
        if (has2) plain_row(y0 + kBlkH, 0, wfull);
        const bool dead = (row_skipped(y0) && (!has2 || row_skipped(y0 + kBlkH))) || wfull >= W - m;
        if (!dead) out.push_back(WorkItem{slot, y0, wfull | kFoldBit, H, W, p0, p1, p2});
    }
}

// place4 = prepadding * scale when the slots ARE the tiles (non-TTA: conv_last may write the image itself), < 0 otherwise
// (TTA: 8 slots per tile, net_forward); trim4 = prepadding * scale when only the cropped rectangle is kept, 0 = keep all
static void make_items(Plan::Batch& b, bool fold, int place4 = -1, int trim4 = 0, int out_row0 = 0)
{
    const int margin[3] = {0, tail_margin(trim4, 3), tail_margin(trim4, 2)}; // per level: the smallest margin of its convs
    const int mtop[3] = {tail_margin(trim4, 4), tail_margin(trim4, 3), tail_margin(trim4, 0)}; // ... and the largest (append_block_items)
    for (int lvl = 0; lvl < 3; lvl++)
    {
        b.items[lvl].clear();
        b.item_start[lvl].assign(size_t(b.nslots) + 1, 0);
        b.px[lvl] = 0;
        const int m = margin[lvl];
        for (int s = 0; s < b.nslots; s++)
        {
            b.item_start[lvl][size_t(s)] = int(b.items[lvl].size());
            const int H = b.dims[size_t(s)].h << lvl, W = b.dims[size_t(s)].w << lvl;
            b.px[lvl] += double(H) * W;
            // 4x-level items carry the tile's placement in the output image: conv_last can write the uint8 image itself
            int p0 = 0, p1 = 0, p2 = 0;
            if (lvl == 2 && place4 >= 0 && size_t(s) < b.tiles.size())
            {
                const BaseTile& t = b.tiles[size_t(s)];
                p0 = t.out_x - place4;
                p1 = t.out_y - out_row0 - place4; // rows relative to the first output row the caller's device buffer holds
                p2 = item_pad2(t.out_w, t.out_h, t.img);
            }
            append_block_items(b.items[lvl], s, H, W, m, mtop[lvl], p0, p1, p2, fold);
        }
        b.item_start[lvl][size_t(b.nslots)] = int(b.items[lvl].size());
    }
}

static size_t al256(size_t v) { return (v + 255) & ~size_t(255); }

static size_t batch_table_bytes(const Plan::Batch& b)
{
    size_t n = al256(b.tiles.size() * sizeof(BaseTile)) + al256(b.dims.size() * sizeof(TileDim));
    for (int l = 0; l < 3; l++) n += 2 * al256(b.items[l].size() * sizeof(WorkItem));
    return n;
}

// upload one batch's tables behind `d`; every copy is checked
static hipError_t upload_batch(Plan::Batch& b, char*& d, bool xcd_order = true)
{
    hipError_t err = hipSuccess;
    auto put = [&](const void* src, size_t bytes) -> void* {
        void* at = d;
        if (bytes && err == hipSuccess) err = hipMemcpy(at, src, bytes, hipMemcpyHostToDevice);
        d += al256(bytes);
        return at;
    };
    b.d_tiles = static_cast<BaseTile*>(put(b.tiles.data(), b.tiles.size() * sizeof(BaseTile)));
    b.d_dims = static_cast<TileDim*>(put(b.dims.data(), b.dims.size() * sizeof(TileDim)));
    for (int l = 0; l < 3; l++)
    {
        b.d_items[l] = static_cast<WorkItem*>(put(b.items[l].data(), b.items[l].size() * sizeof(WorkItem)));
        // The backward table.  LR level (never launched in sub-ranges): reversed INSIDE each XCD's share of the list -- the kernel
        // gives XCD x the items [x * per, (x + 1) * per), per = ceil(n / 8) -- so that a conv starts, on every XCD, with the blocks
        // that XCD itself produced last: those sit in ITS 4 MB L2.  (A whole-list reversal hands XCD 0 the blocks XCD 7 wrote:
        // Infinity-Cache hits at best, and the L2-miss path is what bounds the dense-block convs)  The 2x / 4x
        // tables are launched in slot sub-ranges (run_network) and keep the plain reversal.
        std::vector<WorkItem> rev(b.items[l].rbegin(), b.items[l].rend());
        if (l == 0 && xcd_order)
        {
            const size_t n = b.items[l].size(), per = (n + 7) / 8;
            for (size_t x = 0; x < 8; x++)
            {
                const size_t i0 = std::min(n, x * per), i1 = std::min(n, (x + 1) * per);
                for (size_t i = i0; i < i1; i++) rev[i] = b.items[l][i0 + (i1 - 1 - i)];
            }
        }
        b.d_items_rev[l] = static_cast<WorkItem*>(put(rev.data(), rev.size() * sizeof(WorkItem)));
    }
    return err;
}

// per-slot workspace bytes per LR pixel: IN 64, FEA 128, 3 x RDB 384, UP1 4*128, UP2 16*128, HR 16*128, OUT3 16*6;
// precise mode: + the one-byte lo planes of FEA and of the three RDB x tensors (4 x 64), OUT3 in fp32 (16*12)
long long Engine::bytes_per_px() const { return 64 + 128 + 3 * 384 + 512 + 2048 + 2048 + 96 + (precise ? 4 * 64 + 96 : 0); }
static constexpr size_t kMaxPlans = 8;

// What the device can give this engine's workspace right now: 90 % of (free memory + the workspace it already holds) minus the
// image buffers its lanes may still allocate (in + 16x out per lane; what they hold already is not free any more).  < 0: unknown.
long long Engine::device_avail(int w, int h, int c)
{
    size_t f = 0, t = 0;
    if (hipMemGetInfo(&f, &t) != hipSuccess)
    {
        (void)hipGetLastError();
        return -1;
    }
    long long held = 0;
    for (const DevBuf* wb : {&b_in, &b_fea, &b_rdb[0], &b_rdb[1], &b_rdb[2], &b_up1, &b_up2, &b_hr, &b_out3}) held += (long long)wb->bytes;
    long long lanes_have = 0;
    {
        std::lock_guard<std::mutex> ll(lane_mu);
        for (const auto& l : lanes) lanes_have += (long long)(l->in_bytes.load(std::memory_order_relaxed) + l->out_bytes.load(std::memory_order_relaxed));
    }
    const long long lanes_need = std::max<long long>(0, (long long)max_lanes * 17 * w * h * c - lanes_have);
    return std::max<long long>(0, ((long long)f + held) / 10 * 9 - lanes_need);
}

// The padded tiles [tile0, tile1) of a w x h image (tile grid: realsr.cpp:170-171; tile geometry: realsr.cpp:178-181,237,246-249),
// appended to `all` as tiles of image `img` of a batch; cap / mtw / mth: running maxima of tile pixels / width / height.
static void image_tiles(int w, int h, int T, int P, int scale, int tile0, int tile1, int img, std::vector<BaseTile>& all, long long& cap, int& mtw, int& mth)
{
    const int xtiles = (w + T - 1) / T;
    for (int ti = tile0; ti < tile1; ti++)
    {
        const int yi = ti / xtiles, xi = ti - yi * xtiles;
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
        t.img = img;
        all.push_back(t);
        cap = std::max(cap, (long long)t.tw * t.th);
        mtw = std::max(mtw, t.tw);
        mth = std::max(mth, t.th);
    }
}

int Engine::get_plan(int w, int h, int c, int tile0, int tile1, int nimg, Plan*& out)
{
    const long long kBytesPerPx = bytes_per_px();
    for (auto it = plans.begin(); it != plans.end(); ++it)
        if (it->w == w && it->h == h && it->c == c && it->T == tilesize && it->P == prepadding && it->tta == tta && it->nimg == nimg && it->precise == precise && it->ntw2 == ((flow_flags & 1) != 0) &&
            it->tile0 == tile0 && it->tile1 == tile1 && it->budget_mb == max_workspace_mb && it->trim == trim_tail && it->xcd_order == xcd_order && it->fold == fold_cols &&
            it->clamp == ws_clamp_bytes)
        {
            plans.splice(plans.begin(), plans, it); // most recently used first
            out = &plans.front();
            return RSR_OK;
        }
    const int T = tilesize, P = prepadding;
    // tile grid: realsr.cpp:170-171; tile geometry: realsr.cpp:178-181,237,246-249
    const int xtiles = (w + T - 1) / T, ytiles = (h + T - 1) / T;
    std::vector<BaseTile> all;
    long long cap = 0;
    int mtw = 0, mth = 0;
    if (tile0 < 0 || tile1 > xtiles * ytiles || tile0 >= tile1) return fail(RSR_E_ARG, "tile range outside the image");
    if (nimg < 1 || nimg > kMaxMerge || (nimg > 1 && (tile0 != 0 || tile1 != xtiles * ytiles))) return fail(RSR_E_ARG, "merged batches take whole images");
    for (int im = 0; im < nimg; im++) image_tiles(w, h, T, P, scale, tile0, tile1, im, all, cap, mtw, mth);
    // every merged batch -- of one geometry (this plan) or of several (enqueue_mixed) -- gives its slots the capacity of a full tile: the
    // workspace layout, hence its guards, then stays put from batch to batch whatever small images come
    if (nimg > 1) cap = std::max(cap, (long long)(T + 2 * P) * (T + 2 * P));
    // The kernels address a plane through 32-bit byte offsets (raw buffer resources, out-of-range sentinel 2^31): the
    // largest plane is the 4x level, 16 * cap pixels * 32 B, and an MFMA wave reaches the 2 planes of its n-tile (4 with two n-tiles
    // per wave, flow_flags bit 0) from one base; the stores' range limit is "end of the plane + the plane's offset" (conv_flow.hip
    // make_out), which must stay below the sentinel too.  Tiles beyond that (~1,400 px) must be split by the caller.
    {
        const long long reach = (flow_flags & 1) ? 4 : 2; // planes an MFMA wave addresses from one base
        if (cap * 16 * 32 * reach + 4 * kGuard >= (1ll << 31))
        {
            const long long max_px = ((1ll << 31) - 4 * kGuard - 1) / (16 * 32 * reach);
            return fail(RSR_E_ARG, "tilesize too large: a padded tile may have at most " + std::to_string(max_px) + " pixels" +
                                       ((flow_flags & 1) ? " with flow_flags bit 0 (e.g. -t 1000)" : " (e.g. -t 1400)"));
        }
    }
    const int per = tta ? 8 : 1;
    const long long per_slot = cap * kBytesPerPx;
    // Memory policy (the reference bounds device memory through the tile size alone, main.cpp:761-774; here ALL tiles of an image
    // form one batch, so the batch is what must be bounded): the budget is max_workspace_mb, but never more than 90 % of what
    // the device can actually give this engine right now -- free memory + the workspace it already holds - the image buffers
    // of its lanes -- and never more than a size that has already failed to allocate (ws_clamp_bytes, enqueue_image's retry).
    long long budget = max_workspace_mb * 1024 * 1024;
    {
        const long long avail = device_avail(w, h, c);
        if (avail >= 0) budget = std::min(budget, std::max<long long>(avail, per_slot * per));
    }
    if (ws_clamp_bytes >= 0) budget = std::min(budget, std::max<long long>(ws_clamp_bytes, per_slot * per));
    long long budget_slots = budget / std::max<long long>(per_slot, 1);
    budget_slots = std::max<long long>(per, budget_slots / per * per);
    const long long total_slots = (long long)all.size() * per;
    const int spb = int(std::min<long long>(total_slots, budget_slots));
    const int tiles_per_batch = spb / per;

    Plan plan;
    plan.w = w; plan.h = h; plan.c = c; plan.T = T; plan.P = P; plan.tta = tta;
    plan.nimg = nimg;
    plan.precise = precise;
    plan.ntw2 = (flow_flags & 1) != 0;
    plan.tile0 = tile0; plan.tile1 = tile1;
    plan.budget_mb = max_workspace_mb;
    plan.trim = trim_tail;
    plan.xcd_order = xcd_order;
    plan.fold = fold_cols;
    plan.out_row0 = std::min((tile0 / xtiles) * T, h) * scale; // first output row of the range's first tile row
    plan.clamp = ws_clamp_bytes;
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
        b.trim4 = trim_tail ? P * scale : 0;
        make_items(b, fold_cols, tta ? -1 : P * scale, b.trim4, plan.out_row0);
        table_bytes += batch_table_bytes(b);
        plan.batches.push_back(std::move(b));
    }
    if (plans.size() >= kMaxPlans)
    { // evict the least recently used plan; kernels in flight may still read its tables
        HIP_TRY(hipStreamSynchronize(stream));
        if (plans.back().d_tables) (void)hipFree(plans.back().d_tables);
        plans.pop_back();
    }
    HIP_TRY(hipMalloc(&plan.d_tables, table_bytes));
    char* d = static_cast<char*>(plan.d_tables);
    for (Plan::Batch& b : plan.batches)
    {
        const hipError_t e = upload_batch(b, d, xcd_order && nimg == 1); // (a merged plan is launched in prefixes: plain reversal, whose prefixes are suffixes)
        if (e != hipSuccess)
        {
            (void)hipFree(plan.d_tables);
            return fail(RSR_E_DEVICE, std::string("plan table upload: ") + hipGetErrorString(e));
        }
    }
    plans.push_front(std::move(plan));
    out = &plans.front();
    return RSR_OK;
}

// Every fp16 plane that a convolution may read carries a 64-byte zero GUARD in front of pixel 0 (plane = [guard |
// H*W*ppx B]).  The LDS-DMA loaders address a plane as (uniform base, 32-bit lane offset); lanes whose patch pixel
// lies outside the image use offset 0 = the guard, i.e. conv zero padding without a second base pointer.
// Invariant: every guard of the CURRENT layout inside the whole allocation is zero.  A fresh allocation is zeroed
// completely; when the layout (plane stride) changes inside an existing allocation, the guards of the new layout are
// zeroed over the WHOLE allocation (not just the slots this call uses: a later call with more slots of the same layout
// must not find stale pixels where its guards are).  Guards are never written afterwards.
int Engine::ensure_planes(DevBuf& b, size_t bytes, long long plane_bytes, bool layout_changed, bool zero_all, hipStream_t st)
{
    const bool grow = !(b.bytes >= bytes && b.p);
    if (grow)
    {
        const int rc = ensure(b, bytes);
        if (rc != RSR_OK) return rc;
        HIP_TRY(hipMemsetAsync(b.p, 0, b.bytes, st)); // same stream as the kernels that follow
    }
    else if (layout_changed)
    {
        if (zero_all) HIP_TRY(hipMemsetAsync(b.p, 0, b.bytes, st));
        else launch_zero_guards(b.p, plane_bytes, (long long)(b.bytes / size_t(plane_bytes)), st);
    }
    return RSR_OK;
}

int Engine::ensure_workspace(int nslots, long long cap, hipStream_t st)
{
    constexpr int pc = plane_ch();
    const size_t n = size_t(nslots), c = size_t(cap), G = size_t(kGuard), ppx = size_t(pc) * 2;
    const size_t p32 = size_t(32 / pc), p64 = size_t(64 / pc);
    // the plane stride (hence the guard positions) depends on the slot capacity; precise mode moves the planes of a slot (its one-byte
    // lo planes sit at half strides and run across plane boundaries: where they lay, the other layout has guards)
    const bool lc = cap != ws_cap_px || precise != ws_precise;
    // precise mode: FEA and the RDB buffers carry the lo planes of their 64-channel trunk tensor behind the hi / dense planes
    // (one byte per element: the four lo planes of a tensor take the room of two hi planes)
    const size_t fea_pps = p64 + (precise ? p64 / 2 : 0), rdb_pps = 3 * p64 + (precise ? p64 / 2 : 0);
    const size_t need[] = {n * p32 * (c * ppx + G), n * fea_pps * (c * ppx + G), n * rdb_pps * (c * ppx + G), n * p64 * (c * 4 * ppx + G),
                           n * p64 * (c * 16 * ppx + G), n * c * (precise ? 192 : 96)};
    DevBuf* const bufs[] = {&b_in, &b_fea, &b_rdb[0], &b_up1, &b_up2, &b_out3};
    bool any_grow = false;
    for (int i = 0; i < 6; i++)
        if (need[i] && !(bufs[i]->bytes >= need[i] && bufs[i]->p)) any_grow = true;
    if (ws_fail_above_bytes >= 0)
    { // test hook (option "ws_fail_above_mb"): a device on which workspaces above this size persistently fail to allocate
        const long long total = (long long)nslots * cap * bytes_per_px();
        if (total > ws_fail_above_bytes)
        {
            ws_failures++;
            return fail(RSR_E_NOMEM, "workspace of " + std::to_string(total >> 20) + " MiB refused (ws_fail_above_mb test hook)");
        }
    }
    if (any_grow || lc) HIP_TRY(hipStreamSynchronize(st)); // nothing in flight may use a buffer that is freed / re-laid-out
    int rc;
    // b_in: its second plane (channels 16..31 of the padded 3-channel input) must stay zero
    if ((rc = ensure_planes(b_in, need[0], long(c * ppx + G), lc, true, st)) != RSR_OK) return rc;
    if ((rc = ensure_planes(b_fea, need[1], long(c * ppx + G), lc, false, st)) != RSR_OK) return rc;
    for (int i = 0; i < 3; i++)
        if ((rc = ensure_planes(b_rdb[i], need[2], long(c * ppx + G), lc, false, st)) != RSR_OK) return rc;
    if ((rc = ensure_planes(b_up1, need[3], long(c * 4 * ppx + G), lc, false, st)) != RSR_OK) return rc;
    if ((rc = ensure_planes(b_up2, need[4], long(c * 16 * ppx + G), lc, false, st)) != RSR_OK) return rc;
    if ((rc = ensure_planes(b_hr, need[4], long(c * 16 * ppx + G), lc, false, st)) != RSR_OK) return rc;
    if ((rc = ensure(b_out3, need[5])) != RSR_OK) return rc;
    HIP_TRY(hipGetLastError());
    ws_cap_px = cap;
    ws_precise = precise;
    return RSR_OK;
}

void Engine::free_workspace(hipStream_t st)
{
    (void)hipStreamSynchronize(st);
    for (DevBuf* wb : {&b_in, &b_fea, &b_rdb[0], &b_rdb[1], &b_rdb[2], &b_up1, &b_up2, &b_hr, &b_out3})
    {
        if (wb->p) (void)hipFree(wb->p);
        wb->p = nullptr;
        wb->bytes = 0;
    }
    ws_cap_px = 0;
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
// x4.param as a fused schedule.  Buffers per slot, in planes of plane_ch() channels (P32 = planes
// per 32 channels, P64 per 64):
//   IN(P32; channels 3.. zero)  FEA(P64)  RDB[3] x {x(P64) + dense x1..x4 (4 x P32)}  T32/R32 (fp32 trunk, kernels 1-3)
//   UP1(P64 @2x)  UP2(P64 @4x)  HR(P64 @4x)  OUT3 (planar [3][4H][4W] fp16)
// Concat never happens: conv k of a dense block reads planes [x | x1..x_{k-1}] in place and writes its
// 32 channels as plane(s) x_k.  Eltwise/BinaryOp/Interp are epilogue or staging-address variants.
int Engine::launch(ConvArgs& a, int ci, const Plan::Batch& b, hipStream_t st)
{
    const PackedConv& c = convs[size_t(ci)];
    if (!launch_conv_flow(a, int(c.nt), num_cu, flow_flags, st))
        return fail(RSR_E_STATE, "conv3x3_flow has no variant for convolution " + std::to_string(ci));
    const double frac = b.items[a.lvl_out].empty() ? 1.0 : double(a.nitems) / double(b.items[a.lvl_out].size()); // split launches
    mark(1, 2.0 * 9.0 * c.cin * c.cout * b.px[a.lvl_out] * frac, 0, st, ci);
    return RSR_OK;
}

int Engine::run_network(const Plan::Batch& b, hipStream_t st, uint8_t* const* fused_outs, int nimg, const int* fused_out_ws, int split_slot, hipEvent_t ev_half,
                        hipEvent_t ev_mid, int mid_rdb, int nslots_used)
{
    const int nslots = (nslots_used > 0 && nslots_used < b.nslots) ? nslots_used : b.nslots;
    const long long cap = ws_cap_px;
    const int pc = plane_ch(), P32 = 32 / pc, P64 = 64 / pc;
    const long long ppx = pc * 2;
    const long long pb16 = cap * ppx + kGuard; // planes are guarded (see ensure_workspace)
    // precise mode (ConvArgs::precise): the lo planes of a trunk tensor sit behind the hi / dense planes of its slot
    const int fea_pps = P64 + (precise ? P64 / 2 : 0), rdb_pps = 3 * P64 + (precise ? P64 / 2 : 0);
    const long long fea_lo = precise ? P64 * pb16 : 0, rdb_lo = precise ? 3 * P64 * pb16 : 0;
    auto PS = [](const DevBuf& buf, long long planes_per_slot, long long plane_bytes, int plane_off) {
        PlaneSrc s; // base = pixel 0 of plane `plane_off` of slot 0
        s.base = static_cast<const char*>(buf.p) + (long long)plane_off * plane_bytes + kGuard;
        s.slot_stride = planes_per_slot * plane_bytes;
        s.plane_stride = plane_bytes;
        return s;
    };
    const char* blobp = static_cast<const char*>(blob.p);
    int ci = 0, rc = RSR_OK;
    auto base_args = [&](int lvl_in, int lvl_out) {
        ConvArgs a;
        std::memset(&a, 0, sizeof a);
        const PackedConv& c = convs[size_t(ci)];
        a.wpk16 = blobp + c.w16_off;
        a.waux = c.aux_off ? blobp + c.aux_off : nullptr;
        a.bias = reinterpret_cast<const float*>(blobp + c.b_off);
        a.lrelu = (c.act == 2);
        a.lvl_in = lvl_in;
        a.lvl_out = lvl_out;
        // Boustrophedon: every second conv walks the blocks backwards, so it starts on the tiles the previous conv
        // produced (and read) last -- those are still in the 256 MB Infinity Cache; walking forwards again would start
        // on the least recently used data.
        a.items = (alternate_order && (ci & 1) && b.d_items_rev[lvl_out]) ? b.d_items_rev[lvl_out] : b.d_items[lvl_out];
        a.nitems = int(b.items[lvl_out].size());
        if (nslots < b.nslots)
        { // the first `nslots` slots only: a prefix of the forward table = a suffix of the (plainly) reversed one
            const int n = a.nitems, cnt = b.item_start[lvl_out][size_t(nslots)];
            if (a.items == b.d_items_rev[lvl_out]) a.items = b.d_items_rev[lvl_out] + (n - cnt);
            a.nitems = cnt;
        }
        a.dims = b.d_dims;
        a.zeros = zeros.p;
        a.dbg = dbg;
        a.trace = (trace_conv == ci) ? static_cast<unsigned long long*>(trace_buf.p) : nullptr;
        a.s1 = a.s2 = 1.f;
        a.precise = precise ? 1 : 0;
        // x4.param order: ... RDB 69 | trunk_conv kNumConvs-5 | upconv1 -4 | upconv2 -3 | HRconv -2 | conv_last kNumConvs-1
        constexpr int kLast = kNumConvs - 1, kUp1 = kNumConvs - 4;
        a.margin = tail_margin(b.trim4, ci >= kUp1 ? kLast - ci : 4 + (kUp1 - 1 - ci));
        return a;
    };
    auto go = [&](ConvArgs& a) {
        if (rc == RSR_OK) rc = launch(a, ci, b, st);
        ci++;
    };
    const PlaneSrc fea = PS(b_fea, fea_pps, pb16, 0);
    auto rdb_x = [&](int i) { return PS(b_rdb[i], rdb_pps, pb16, 0); };
    auto rdb_d = [&](int i, int k) { return PS(b_rdb[i], rdb_pps, pb16, P64 + k * P32); };

    { // conv_first (x4.param:4): IN -> FEA
        ConvArgs a = base_args(0, 0);
        a.src0 = PS(b_in, P32, pb16, 0);
        a.n0 = P32;
        a.out16 = fea;
        a.out_lo_off = fea_lo; // the stream starts here: fea keeps its rounding residue too
        go(a);
    }
    for (int j = 0; j < kNumRDB; j++)
    {
        const int bi = j % 3;
        const PlaneSrc xs = (j == 0) ? fea : rdb_x(bi);
        for (int k = 0; k < 4; k++)
        { // x_{k+1} = lrelu(conv([x, x1..xk]))   (x4.param:6,9,12,15)
            ConvArgs a = base_args(0, 0);
            a.src0 = xs; a.n0 = P64;
            a.src1 = rdb_d(bi, 0); a.n1 = k * P32;
            a.out16 = rdb_d(bi, k);
            go(a);
        }
        // x5 = conv([x, x1..x4]);  out = 0.2*x5 + x   (x4.param:17-18);  every third block additionally
        // out = 0.2*out + rrdb_in   (x4.param:47, Eltwise 0=1 -23301=2,0.2,1.0)
        ConvArgs a = base_args(0, 0);
        a.src0 = xs; a.n0 = P64;
        a.src1 = rdb_d(bi, 0); a.n1 = 4 * P32;
        a.s1 = 0.2f;
        a.res1 = xs; a.res1_kind = 1; a.res1_in_acc = 1; a.res1_coef = 5.f; // 1/0.2, exact in fp16
        a.lo1_off = (j == 0) ? fea_lo : rdb_lo;
        if (bi == 2)
        {
            a.s2 = 0.2f;
            a.res2 = (j == 2) ? fea : rdb_x(0); a.res2_kind = 1;
            a.lo2_off = (j == 2) ? fea_lo : rdb_lo;
        }
        a.out16 = rdb_x((j + 1) % 3);
        a.out_lo_off = rdb_lo;
        go(a);
        if (j == mid_rdb && ev_mid && rc == RSR_OK && hipEventRecord(ev_mid, st) != hipSuccess) rc = fail(RSR_E_DEVICE, "hipEventRecord failed");
    }
    { // trunk_conv + global skip: fea + conv(trunk)   (x4.param:994-995)
        ConvArgs a = base_args(0, 0);
        a.src0 = rdb_x(kNumRDB % 3); a.n0 = P64;
        a.res1 = fea; a.res1_kind = 1; a.s1 = 1.f;
        a.lo1_off = fea_lo;
        a.out16 = rdb_x(1); // (only upconv1 reads it: no lo planes)
        go(a);
    }
    const PlaneSrc up1 = PS(b_up1, P64, cap * 4 * ppx + kGuard, 0), up2 = PS(b_up2, P64, cap * 16 * ppx + kGuard, 0),
                   hr = PS(b_hr, P64, cap * 16 * ppx + kGuard, 0);
    // The convs behind the trunk (upconv1 @2x, upconv2 / HRconv / conv_last @4x: ~11 % of the frame) can run per GROUP of
    // `tail_group_slots` slots (0 = the whole batch): a tile has 4x / 16x the blocks up there, so a few tiles fill the chip for
    // > 100 us per launch, and their 2 x 99 MB per tile of 4x intermediates are then read back from the 256 MB Infinity Cache
    // instead of HBM.  Measured on MI355X: <= 1.5 % on the 4x launches, a loss at 2x -- HBM bytes are not what bounds this
    // workload, so the default is off.  With split_slot > 0 a group boundary is forced there and `ev_half` is
    // recorded behind it: the caller starts downloading the finished output rows while the remaining groups are computed
    // (tiles are independent).  Items are sorted by slot, so a group is a contiguous range of the item tables (mirrored in
    // the reversed tables).
    const int ci_tail = ci;
    const int gsz = tail_group_slots > 0 ? tail_group_slots : nslots;
    for (int s0 = 0; s0 < nslots && rc == RSR_OK;)
    {
        int s1 = std::min(nslots, s0 + gsz);
        if (split_slot > s0 && split_slot < s1) s1 = split_slot;
        ci = ci_tail;
        auto sub = [&](ConvArgs& a, int lvl) {
            const int n = int(b.items[lvl].size());
            const int i0 = b.item_start[lvl][size_t(s0)], cnt = b.item_start[lvl][size_t(s1)] - i0;
            const bool rev = a.items >= b.d_items_rev[lvl] && a.items < b.d_items_rev[lvl] + n; // (base_args may have moved it to a suffix already)
            a.items = rev ? b.d_items_rev[lvl] + (n - i0 - cnt) : b.d_items[lvl] + i0;
            a.nitems = cnt;
        };
        { // nearest x2 + upconv1 + lrelu   (x4.param:996-997)
            ConvArgs a = base_args(0, 1);
            a.src0 = rdb_x(1); a.n0 = P64;
            a.out16 = up1;
            sub(a, 1);
            go(a);
        }
        { // nearest x2 + upconv2 + lrelu   (x4.param:998-999)
            ConvArgs a = base_args(1, 2);
            a.src0 = up1; a.n0 = P64;
            a.out16 = up2;
            sub(a, 2);
            go(a);
        }
        { // HRconv + lrelu   (x4.param:1000)
            ConvArgs a = base_args(2, 2);
            a.src0 = up2; a.n0 = P64;
            a.out16 = hr;
            sub(a, 2);
            go(a);
        }
        { // conv_last 64 -> 3   (x4.param:1001), planar fp16 output = the reference's `output` blob
            ConvArgs a = base_args(2, 2);
            a.src0 = hr; a.n0 = P64;
            if (fused_outs)
            { // non-TTA RGB with conv3x3_flow: conv_last applies realsr_postproc.comp itself and writes the image(s)
                a.out_u8 = fused_outs[0];
                for (int i = 0; i < nimg && i < kMaxMerge; i++)
                {
                    a.out_u8s[i] = fused_outs[i];
                    a.out_u8_ws[i] = fused_out_ws[i];
                }
                a.out_u8_w = fused_out_ws[0];
                a.out_u8_crop = prepadding * scale;
                a.out_u8_bgr = bgr ? 1 : 0;
            }
            else
            {
                a.out_planar3 = b_out3.p;
                a.planar3_slot_stride = cap * (precise ? 192 : 96);
            }
            sub(a, 2);
            go(a);
        }
        if (s1 == split_slot && ev_half && rc == RSR_OK && hipEventRecord(ev_half, st) != hipSuccess)
            rc = fail(RSR_E_DEVICE, "hipEventRecord failed");
        s0 = s1;
    }
    return rc;
}

// ---- process ----------------------------------------------------------------------------------
// enqueue preproc -> network -> postproc for every tile batch of one image on `st` (mu held)
int Engine::enqueue_image(const void* d_in, int w, int h, int c, void* d_out, hipStream_t st, int tile0, int tile1, hipEvent_t ev_half,
                          size_t* half_rows)
{
    return enqueue_images(&d_in, &d_out, 1, w, h, c, st, tile0, tile1, ev_half, half_rows);
}

int Engine::enqueue_images(const void* const* d_in, void* const* d_out, int nimg, int w, int h, int c, hipStream_t st, int tile0, int tile1,
                           hipEvent_t ev_half, size_t* half_rows, hipEvent_t ev_mid, int plan_nimg)
{
    const bool merged = plan_nimg > 0; // (every caller of a merged batch reports the progress of its own image: process_host)
    if (plan_nimg < nimg) plan_nimg = nimg;
    const long long kBytesPerPx = bytes_per_px();
    if (half_rows) *half_rows = 0;
    Plan* planp = nullptr;
    const int xtiles = (w + tilesize - 1) / tilesize;
    if (tile1 < 0) tile1 = xtiles * ((h + tilesize - 1) / tilesize);
    int rc;
    // A clamp left behind by a transient allocation failure (another process held the memory for a moment) must not halve the
    // batches for ever: a NEW call that finds the device able to give twice the clamped size again plans without it (never inside
    // the retry loop below, which would then oscillate).  But a PERSISTENT cause (fragmentation, hipMemGetInfo overstating what one
    // hipMalloc can get) makes that test true on every call -- each frame would re-plan at full size, fail, drain the stream, free
    // and rebuild the workspace.  So an un-clamp attempt is made at once only when the device reports clearly more room
    // than it did at the moment of the failure (x 1.25), and otherwise at most once per `clamp_backoff` calls, the back-off doubling
    // with every attempt that fails again.  Plans are keyed by the clamp.
    bool unclamp_attempt = false;
    if (ws_clamp_bytes >= 0)
    {
        const long long avail = device_avail(w, h, c);
        if (avail >= 2 * ws_clamp_bytes)
        {
            const bool grown = clamp_fail_avail < 0 || avail >= clamp_fail_avail + clamp_fail_avail / 4;
            if (grown || --clamp_calls_left <= 0)
            {
                clamp_saved = ws_clamp_bytes;
                ws_clamp_bytes = -1;
                unclamp_attempt = true;
            }
        }
    }
    for (;;)
    {
        rc = get_plan(w, h, c, tile0, tile1, plan_nimg, planp);
        if (rc != RSR_OK) return rc;
        // (a merged batch narrower than its plan needs the slots of its own images only)
        // (a merged batch narrower than its plan still gets the plan's workspace: growing it image by image as wider batches form
        // would re-allocate and clear gigabytes a dozen times)
        rc = ensure_workspace(planp->slots_per_batch, planp->cap_px, st);
        if (rc != RSR_E_NOMEM) break;
        // The batch does not fit (a shared or partly used GPU): halve it and plan again, down to one tile (x8 under TTA).
        const int per = tta ? 8 : 1;
        if (planp->slots_per_batch <= per) return rc;
        const std::string why = last_error();
        long long half_slots = std::max<long long>(per, (planp->slots_per_batch / 2) / per * per);
        const long long failed_bytes = (long long)planp->slots_per_batch * planp->cap_px * kBytesPerPx;
        clamp_fail_avail = device_avail(w, h, c);
        if (unclamp_attempt && clamp_saved > 0 && clamp_saved < failed_bytes)
        { // the clamp was right after all: back to it (not to half of the full size), and wait longer before the next try
            half_slots = std::max<long long>(per, clamp_saved / (planp->cap_px * kBytesPerPx) / per * per);
            clamp_backoff = std::min(clamp_backoff * 2, 1024);
        }
        unclamp_attempt = false;
        clamp_calls_left = clamp_backoff;
        ws_clamp_bytes = half_slots * planp->cap_px * kBytesPerPx;
        std::fprintf(stderr, "realsr-hip: workspace of %d tile slots (%.1f GB) does not fit device %d (%s); batches of %lld slots, next attempt at full size "
                             "after %d calls or when the device reports > %.1f GB available\n",
                     planp->slots_per_batch, double(failed_bytes) / 1e9, device, why.c_str(), half_slots, clamp_backoff,
                     double(clamp_fail_avail) * 1.25 / 1e9);
        free_workspace(st); // partly grown buffers go back first; the stream is drained, so the plan's tables are idle too
        if (planp->d_tables) (void)hipFree(planp->d_tables);
        plans.pop_front(); // get_plan put it in front
    }
    if (rc == RSR_OK && unclamp_attempt) { clamp_backoff = 4; clamp_fail_avail = -1; } // the full size fits again: forget the history
    if (rc != RSR_OK) return rc;
    const Plan& plan = *planp;
    mark_begin(st);
    int done = 0, total = 0;
    for (const Plan::Batch& b : plan.batches) total += b.ntiles;
    const int tiles_wanted = (tile1 - tile0) * nimg; // the tiles of the first nimg images (all of the plan's unless the batch is narrower)
    int ws[kMaxMerge], hs[kMaxMerge];
    for (int i = 0; i < nimg; i++) { ws[i] = w; hs[i] = h; }
    for (const Plan::Batch& b : plan.batches)
    {
        const int ntiles = std::min(b.ntiles, tiles_wanted - b.tile0); // of this batch
        if (ntiles <= 0) break;
        // host calls: split the 4x tail at a tile-row boundary so that the first output rows can travel while the rest is computed
        const bool fused = !tta && c == 3 && !(dbg & 8192);
        int split_slot = 0;
        if (fused && nimg == 1 && ev_half && half_rows && plan.batches.size() == 1 && !profiling && !(dbg & 16384))
        {
            const int xt = xtiles, yt = b.ntiles / xt;
            if (yt >= 2 && b.ntiles == xt * yt && plan.tile0 % xt == 0)
            {
                split_slot = xt * (yt / 2);
                *half_rows = size_t(b.tiles[size_t(split_slot)].out_y - b.tiles[0].out_y); // output rows finished at ev_half
            }
        }
        const bool last_batch = b.tile0 + b.ntiles >= tiles_wanted;
        rc = launch_batch(b, plan.cap_px, plan.max_tw, plan.max_th, plan.out_row0, d_in, d_out, ws, hs, nimg, c, ntiles, st, split_slot, ev_half,
                          last_batch ? ev_mid : nullptr);
        if (rc != RSR_OK) return rc;
        if (progress && !merged) // one call per TILE, like the reference's line per tile (realsr.cpp:481), issued when the tile's batch is enqueued
            for (int i = 1; i <= b.ntiles; i++) progress(done + i, total, progress_user); // (a merged batch: every caller reports its own image)
        done += b.ntiles;
    }
    HIP_TRY(hipGetLastError());
    if (profiling)
    {
        collect_profile(st);
        prof.calls++;
        for (const Plan::Batch& b : plan.batches) prof.tiles += b.nslots;
    }
    return RSR_OK;
}

// preproc -> network -> postproc of the first `ntiles` tiles of one tile batch on `st` (mu held; the workspace is laid out for cap_px).
// The images of the batch may differ in size (ws / hs): every tile carries the index of its image.
int Engine::launch_batch(const Plan::Batch& b, long long cap_px, int max_tw, int max_th, int out_row0, const void* const* d_in, void* const* d_out,
                         const int* ws, const int* hs, int nimg, int c, int ntiles, hipStream_t st, int split_slot, hipEvent_t ev_half, hipEvent_t ev_mid)
{
    constexpr int pc = plane_ch();
    const int per = tta ? 8 : 1, nslots_used = ntiles * per;
    PreArgs pa;
    std::memset(&pa, 0, sizeof pa);
    for (int i = 0; i < nimg; i++)
    {
        pa.imgs[i] = static_cast<const uint8_t*>(d_in[i]);
        pa.ws[i] = ws[i];
        pa.hs[i] = hs[i];
    }
    pa.nimgs = nimg;
    pa.c = c;
    pa.tiles = b.d_tiles;
    pa.ntiles = ntiles;
    pa.tta = tta;
    pa.in_plane = static_cast<char*>(b_in.p) + kGuard;
    pa.slot_stride = (32 / pc) * (cap_px * pc * 2 + kGuard);
    pa.bgr = bgr ? 1 : 0;
    pa.plane_ch = pc;
    pa.variant = (dbg & 32768) ? 1 : ((dbg & 65536) ? 2 : 0);
    launch_preproc_tiles(pa, max_tw, max_th, st);
    mark(0, 0, b.px[0] / per * c + b.px[0] * 64, st);
    // conv_last writes the uint8 image directly when no TTA merge / alpha channel needs the fp16 blob (dbg 8192: off)
    const bool fused = !tta && c == 3 && !(dbg & 8192);
    uint8_t* outs[kMaxMerge];
    int out_ws[kMaxMerge];
    for (int i = 0; i < nimg; i++)
    {
        outs[i] = static_cast<uint8_t*>(d_out[i]);
        out_ws[i] = ws[i] * scale;
    }
    // the throttle event of a merged batch (Engine::submit_merged): behind the RDB that leaves about half an image's worth of network
    // ahead -- the time the next batch's launches take to enqueue
    int rc = run_network(b, st, fused ? outs : nullptr, nimg, out_ws, split_slot, ev_half, ev_mid, kNumRDB - 1 - std::max(2, kNumRDB / (2 * std::max(1, nimg))),
                         nslots_used);
    if (rc != RSR_OK || fused) return rc;
    PostArgs po;
    std::memset(&po, 0, sizeof po);
    po.planar3 = b_out3.p;
    po.f32 = precise ? 1 : 0;
    po.slot_stride = cap_px * (precise ? 192 : 96);
    po.tiles = b.d_tiles;
    po.ntiles = ntiles;
    po.tta = tta;
    po.crop = prepadding * scale;
    for (int i = 0; i < nimg; i++)
    {
        po.outs[i] = outs[i];
        po.out_ws[i] = out_ws[i];
        po.in_imgs[i] = static_cast<const uint8_t*>(d_in[i]);
        po.in_ws[i] = ws[i];
    }
    po.nimgs = nimg;
    po.c = c;
    po.out_row0 = out_row0;
    po.tilesize = tilesize;
    po.bgr = bgr ? 1 : 0;
    po.variant = (dbg & 32768) ? 1 : ((dbg & 65536) ? 2 : 0);
    launch_postproc_tiles(po, (max_tw - 2 * prepadding) * scale, (max_th - 2 * prepadding) * scale, st);
    mark(2, 0, b.px[2] / per * (6.0 * per + c), st);
    return RSR_OK;
}

// A merged batch of images that DIFFER in size (a directory of thumbnails, sprites, crops ...): no cached plan can serve it -- the
// tile and work-item tables are built here, for exactly these images, and uploaded into one of three rotating device buffers (the
// throttle of submit_merged leaves at most the previous batch in flight when the next one is formed; the buffer's event makes sure).
// All slots get the capacity of a full tile, (T + 2P)^2, whatever the images: the workspace layout (hence its guards) then does not
// change from batch to batch.  mu held.
int Engine::enqueue_mixed(MergeReq* const* g, int n, hipStream_t st, hipEvent_t ev_mid)
{
    const int T = tilesize, P = prepadding, per = tta ? 8 : 1, c = g[0]->c;
    Plan::Batch b;
    long long cap = 0;
    int mtw = 0, mth = 0;
    for (int i = 0; i < n; i++)
    {
        const int xt = (g[i]->w + T - 1) / T, yt = (g[i]->h + T - 1) / T;
        image_tiles(g[i]->w, g[i]->h, T, P, scale, 0, xt * yt, i, b.tiles, cap, mtw, mth);
    }
    cap = (long long)(T + 2 * P) * (T + 2 * P);
    b.tile0 = 0;
    b.ntiles = int(b.tiles.size());
    b.nslots = b.ntiles * per;
    for (int i = 0; i < b.ntiles; i++)
    {
        BaseTile& t = b.tiles[size_t(i)];
        t.slot0 = i * per;
        for (int k = 0; k < per; k++) b.dims.push_back(k < 4 ? TileDim{t.th, t.tw} : TileDim{t.tw, t.th}); // realsr.cpp:251-258
    }
    if (cap * 16 * 32 * ((flow_flags & 1) ? 4 : 2) + 4 * kGuard >= (1ll << 31)) return fail(RSR_E_ARG, "tilesize too large");
    const long long need = (long long)b.nslots * cap * bytes_per_px();
    const long long avail = device_avail(g[0]->w, g[0]->h, c);
    if (need > max_workspace_mb * 1024 * 1024 || (avail >= 0 && need > avail)) return fail(RSR_E_NOMEM, "merged batch exceeds the workspace budget");
    b.trim4 = trim_tail ? P * scale : 0;
    make_items(b, fold_cols, tta ? -1 : P * scale, b.trim4, 0);
    HIP_TRY(hipSetDevice(device));
    const int k = int(mix_seq++ % 3);
    if (mix_ev[k]) HIP_TRY(hipEventSynchronize(mix_ev[k])); // the batch that read this buffer last has finished
    int rc;
    if ((rc = ensure(mix_tab[k], batch_table_bytes(b))) != RSR_OK) return rc;
    char* d = static_cast<char*>(mix_tab[k].p);
    const hipError_t e = upload_batch(b, d, false);
    if (e != hipSuccess) return fail(RSR_E_DEVICE, std::string("merged batch table upload: ") + hipGetErrorString(e));
    if ((rc = ensure_workspace(b.nslots, cap, st)) != RSR_OK) return rc;
    const void* ins[kMaxMerge];
    void* outs[kMaxMerge];
    int ws[kMaxMerge], hs[kMaxMerge];
    for (int i = 0; i < n; i++)
    {
        ins[i] = g[i]->d_in;
        outs[i] = g[i]->d_out;
        ws[i] = g[i]->w;
        hs[i] = g[i]->h;
    }
    rc = launch_batch(b, cap, mtw, mth, 0, ins, outs, ws, hs, n, c, b.ntiles, st, 0, nullptr, ev_mid);
    if (rc != RSR_OK) return rc;
    if (!mix_ev[k] && hipEventCreateWithFlags(&mix_ev[k], hipEventDisableTiming) != hipSuccess) mix_ev[k] = nullptr;
    if (mix_ev[k]) HIP_TRY(hipEventRecord(mix_ev[k], st));
    else HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    return RSR_OK;
}

int Engine::process_device(const void* d_in, int w, int h, int c, void* d_out, hipStream_t user_stream, bool sync)
{
    if (!d_in || !d_out || w < 1 || h < 1 || (c != 3 && c != 4)) return fail(RSR_E_ARG, "bad image arguments");
    hipEvent_t done = nullptr;
    if (!user_stream && sync)
    { // a small image: merged with whatever other calls hand in meanwhile (Engine::submit_merged)
        int T = 0, width = 1;
        long long items = 0;
        { // (the engine's settings are read under its lock: another thread may be in rsr_set_params / rsr_set_option)
            std::lock_guard<std::mutex> lk(mu);
            if (!loaded) return fail(RSR_E_STATE, "process before load");
            if (scale != 4) return fail(RSR_E_ARG, "only scale 4 is supported (main.cpp:533-537)");
            T = tilesize;
            width = merge_width(w, h, c);
            if (width > 1) items = image_items(w, h, merge_target_items);
        }
        if (width > 1)
        {
            MergeReq r;
            r.d_in = d_in; r.d_out = d_out; r.w = w; r.h = h; r.c = c; r.T = T;
            r.items = items;
            r.width = width;
            const int rc = submit_merged(r);
            if (rc != RSR_OK) return rc;
            const hipError_t e = hipEventSynchronize(r.ev_done);
            {
                std::lock_guard<std::mutex> lk(mu);
                give_event(r.ev_done);
            }
            if (e != hipSuccess) return fail(RSR_E_DEVICE, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
            return RSR_OK;
        }
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!loaded) return fail(RSR_E_STATE, "process before load");
        if (scale != 4) return fail(RSR_E_ARG, "only scale 4 is supported (main.cpp:533-537)");
        HIP_TRY(hipSetDevice(device));
        if (user_stream && hipStreamQuery(stream) == hipSuccess)
        {
            // The engine is idle -- nothing of any earlier call is pending on the compute stream (work an earlier call of this kind
            // put on ITS caller's stream included: the compute stream was made to wait for it, below) -- so this call's kernels go
            // straight onto the caller's stream: no event hop into the compute stream and back.  Whatever is enqueued on the compute
            // stream later uses the same workspace and must come behind: it waits for this call's last kernel.
            const int rc = enqueue_image(d_in, w, h, c, d_out, user_stream);
            hipEvent_t e = take_event();
            if (!e || hipEventRecord(e, user_stream) != hipSuccess || hipStreamWaitEvent(stream, e, 0) != hipSuccess)
            { // cannot order the compute stream behind it: fall back to waiting here
                (void)hipGetLastError();
                (void)hipStreamSynchronize(user_stream);
            }
            give_event(e);
            if (rc == RSR_OK) device_direct++;
            return rc;
        }
        (void)hipGetLastError(); // (hipErrorNotReady is not an error)
        // All network kernels run on the engine's compute stream (one workspace); a caller stream is ordered around them.
        if (user_stream)
        {
            hipEvent_t e = take_event();
            if (!e) return fail(RSR_E_DEVICE, "hipEventCreate failed");
            HIP_TRY(hipEventRecord(e, user_stream));
            HIP_TRY(hipStreamWaitEvent(stream, e, 0));
            give_event(e);
        }
        const int rc = enqueue_image(d_in, w, h, c, d_out, stream);
        if (rc != RSR_OK) return rc;
        if (user_stream || sync)
        {
            done = take_event();
            if (!done) return fail(RSR_E_DEVICE, "hipEventCreate failed");
            HIP_TRY(hipEventRecord(done, stream));
            if (user_stream)
            {
                HIP_TRY(hipStreamWaitEvent(user_stream, done, 0));
                give_event(done);
                done = nullptr;
            }
        }
    }
    if (done)
    { // wait outside the lock: other calls may enqueue behind this one meanwhile
        const hipError_t e = hipEventSynchronize(done);
        std::lock_guard<std::mutex> lk(mu);
        give_event(done);
        if (e != hipSuccess) return fail(RSR_E_DEVICE, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
    }
    return RSR_OK;
}

// ---- merging small images across calls (engine.h) ---------------------------------------------
// LR-level work items (16 x 32 blocks; x8 under TTA) of a w x h image at the current tile size; stops counting beyond `limit`
long long Engine::image_items(int w, int h, long long limit) const
{
    const int T = tilesize, P = prepadding;
    if (T < 1) return limit + 1;
    long long items = 0;
    for (int y0 = 0; y0 < h; y0 += T)
        for (int x0 = 0; x0 < w; x0 += T)
        {
            const long long th = std::min(y0 + T, h) - y0 + 2 * P, tw = std::min(x0 + T, w) - x0 + 2 * P;
            items += ((th + kBlkH - 1) / kBlkH) * ((tw + kBlkW - 1) / kBlkW) * (tta ? 8 : 1);
            if (items > limit) return items;
        }
    return items;
}

// How many images of this geometry one merged batch may take: 1 = the image is not small (more than a quarter of the work items a
// batch aims at: a few launches' worth of blocks per CU by itself) or merging is off; else as many as keep the batch at
// `merge_target_items` LR-level work items, at most merge_max.  ONE cached plan of that width serves every narrower batch of this
// geometry (enqueue_images: plan_nimg).
int Engine::merge_width(int w, int h, int c) const
{
    (void)c;
    const int mmax = merge_max, target = merge_target_items;
    if (mmax <= 1 || profiling) return 1;
    const long long items = image_items(w, h, target / 4);
    if (items * 4 > target) return 1;
    return int(std::max<long long>(1, std::min<long long>(std::min(mmax, kMaxMerge), target / std::max<long long>(items, 1))));
}

// Enqueue the images of g[0..n) as ONE tile batch (one geometry: the cached plan; several: tables built on the fly); records every
// request's ev_done behind it.
int Engine::run_group(MergeReq* const* g, int n)
{
    std::lock_guard<std::mutex> lk(mu);
    if (!loaded || scale != 4) return fail(RSR_E_STATE, "context parameters changed while the call was in flight");
    for (int i = 0; i < n; i++)
        if (g[i]->T != tilesize) return fail(RSR_E_STATE, "context parameters changed while the call was in flight");
    HIP_TRY(hipSetDevice(device));
    const void* ins[kMaxMerge];
    void* outs[kMaxMerge];
    for (int i = 0; i < n; i++)
    {
        ins[i] = g[i]->d_in;
        outs[i] = g[i]->d_out;
        if (g[i]->ev_in) HIP_TRY(hipStreamWaitEvent(stream, g[i]->ev_in, 0));
    }
    if (!merge_mid && hipEventCreateWithFlags(&merge_mid, hipEventDisableTiming) != hipSuccess) merge_mid = nullptr;
    merge_mid_used = false;
    bool same = true;
    for (int i = 1; i < n; i++) same = same && g[i]->w == g[0]->w && g[i]->h == g[0]->h;
    int rc;
    if (same) // one geometry: the cached plan of merge_width images, launched in a prefix
        rc = enqueue_images(ins, outs, n, g[0]->w, g[0]->h, g[0]->c, stream, 0, -1, nullptr, nullptr, merge_mid, merge_width(g[0]->w, g[0]->h, g[0]->c));
    else
    {
        rc = enqueue_mixed(g, n, stream, merge_mid);
        if (rc == RSR_OK) merged_mixed++;
        else if (rc == RSR_E_NOMEM)
        { // the batch does not fit the workspace budget / the device right now: one image at a time (each call bounds and halves its own batch)
            (void)hipStreamSynchronize(stream);
            for (int i = 0; i < n; i++)
            {
                rc = enqueue_images(&ins[i], &outs[i], 1, g[i]->w, g[i]->h, g[i]->c, stream, 0, -1, nullptr, nullptr, i == n - 1 ? merge_mid : nullptr,
                                    merge_width(g[i]->w, g[i]->h, g[i]->c));
                if (rc != RSR_OK) break;
            }
        }
    }
    if (rc == RSR_OK)
        for (int i = 0; i < n && rc == RSR_OK; i++)
        {
            if (!g[i]->ev_done)
            {
                g[i]->ev_done = take_event();
                g[i]->pool_event = true;
                if (!g[i]->ev_done) rc = fail(RSR_E_DEVICE, "hipEventCreate failed");
            }
            if (rc == RSR_OK && hipEventRecord(g[i]->ev_done, stream) != hipSuccess) rc = fail(RSR_E_DEVICE, "hipEventRecord failed");
        }
    if (rc == RSR_OK)
    {
        merge_mid_used = merge_mid != nullptr;
        if (!merge_done && hipEventCreateWithFlags(&merge_done, hipEventDisableTiming) != hipSuccess) merge_done = nullptr;
        merge_last_n = (merge_done && hipEventRecord(merge_done, stream) == hipSuccess) ? n : 0;
        merged_batches++;
        merged_images += n;
        if (n > merged_widest.load()) merged_widest = n;
    }
    else
    {
        const std::string why = last_error();
        (void)hipStreamSynchronize(stream); // kernels of the batch that did get enqueued use the callers' buffers
        for (int i = 0; i < n; i++)
            if (g[i]->pool_event)
            {
                give_event(g[i]->ev_done);
                g[i]->ev_done = nullptr;
                g[i]->pool_event = false;
            }
        return fail(rc, why);
    }
    return RSR_OK;
}

int Engine::submit_merged(MergeReq& r)
{
    std::unique_lock<std::mutex> lk(cq_mu);
    cq.push_back(&r);
    cq_cv.notify_all(); // (a leader may be waiting for the calls it knows to be inbound)
    if (cq_leader)
    {
        cq_cv.wait(lk, [&]() { return r.done || r.lead; });
        if (r.done)
        {
            if (r.rc != RSR_OK) return fail(r.rc, r.err);
            return RSR_OK;
        }
    }
    // This caller leads: its own request is at the head of the queue (it found the queue without a leader -- i.e. empty -- or the
    // previous leader woke the head).
    cq_leader = true;
    r.lead = false;
    while (!r.done)
    {
        // Throttle: the next batch is formed when the previous one is NEARLY through the network (launch_batch: an event behind the RDB
        // that leaves about half an image's worth of network ahead) -- its ~350 launches (1 - 2 ms of host time) are then enqueued
        // underneath the rest, the GPU never waits, and while this thread waits further calls queue up behind it: that is what fills a
        // batch (measured in profiles/r06_small_images.txt).
        hipEvent_t wait_ev = merge_mid_used ? merge_mid : nullptr;
        lk.unlock();
        if (wait_ev) (void)hipEventSynchronize(wait_ev);
        lk.lock();
        // Calls that are known to be on their way (their image is still being uploaded) are worth a moment: a batch costs milliseconds,
        // they arrive within microseconds -- 16 callers that start together would otherwise open with batches of 1, 1, 7, 7.
        for (const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(400);
             merge_inbound.load() > 0 && int(cq.size()) < kMaxMerge && std::chrono::steady_clock::now() < deadline;)
            cq_cv.wait_until(lk, deadline); // (every arrival notifies)
        // The batch: the head of the queue and, in order of arrival, the calls behind it that fit -- same channel count and tile size (one
        // preproc / conv_last / postproc launch serves all), images of ANY small size (option merge_mixed 0: of the head's size only) --
        // while the batch stays within merge_target_items work items and merge_max images.
        MergeReq* g[kMaxMerge];
        int n = 0;
        const MergeReq* head = cq.front();
        const int width = merge_mixed ? std::min(merge_max.load(), kMaxMerge) : head->width;
        const long long target = merge_target_items;
        long long items = 0;
        for (MergeReq* q : cq)
        {
            if (n >= width) break;
            if (q->c != head->c || q->T != head->T) continue;
            if (!merge_mixed && (q->w != head->w || q->h != head->h)) continue;
            if (n > 0 && items + q->items > target) continue;
            g[n++] = q;
            items += q->items;
        }
        // Every width up to merge_width shares one plan (enqueue_images: plan_nimg), so any number can be taken.  All of them when the GPU
        // has run dry.  While the previous batch is still running, though, taking everything that waits makes the batch sizes
        // ALTERNATE for ever (the callers of batch n - 1 are exactly what waits when batch n + 1 is formed: 1, 15, 1, 15 ... is as stable
        // as 8, 8, 8, 8 and much slower): take half of (waiting + in flight), the rest leads the next batch.
        int take = n;
        if (merge_last_n > 0 && merge_done && hipEventQuery(merge_done) == hipErrorNotReady) take = std::min(n, std::max(1, (n + merge_last_n + 1) / 2));
        (void)hipGetLastError(); // (hipErrorNotReady is not an error)
        for (int i = 0; i < take; i++) cq.erase(std::find(cq.begin(), cq.end(), g[i]));
        static const bool trace = std::getenv("RSR_MERGE_TRACE") != nullptr; // (debug aid: one line per merged batch)
        if (trace)
            std::fprintf(stderr, "merge: t=%.3f ms take %d of %d waiting (width %d, previous batch of %d %s, %d inbound)\n",
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(), take, n, width, merge_last_n,
                         (merge_done && hipEventQuery(merge_done) == hipErrorNotReady) ? "running" : "done", merge_inbound.load());
        (void)hipGetLastError();
        lk.unlock();
        const int rc = run_group(g, take);
        const std::string why = rc == RSR_OK ? std::string() : std::string(last_error());
        lk.lock();
        for (int i = 0; i < take; i++)
        {
            g[i]->rc = rc;
            g[i]->err = why;
            g[i]->done = true;
        }
        cq_cv.notify_all();
    }
    if (!cq.empty()) cq.front()->lead = true; // hand over: the head of the queue forms the next batch
    else cq_leader = false;
    cq_cv.notify_all();
    if (r.rc != RSR_OK) return fail(r.rc, r.err);
    return RSR_OK;
}

Lane* Engine::acquire_lane()
{
    std::unique_lock<std::mutex> lk(lane_mu);
    for (;;)
    {
        for (auto& l : lanes)
            if (!l->busy)
            {
                l->busy = true;
                return l.get();
            }
        if (int(lanes.size()) < max_lanes)
        {
            lanes.emplace_back(new Lane());
            lanes.back()->busy = true;
            return lanes.back().get();
        }
        lane_cv.wait(lk);
    }
}

void Engine::release_lane(Lane* l)
{
    {
        std::lock_guard<std::mutex> lk(lane_mu);
        l->busy = false;
    }
    lane_cv.notify_one();
}

static bool is_pinned_host(const void* p)
{
    hipPointerAttribute_t at;
    std::memset(&at, 0, sizeof at);
    if (hipPointerGetAttributes(&at, p) != hipSuccess)
    {
        (void)hipGetLastError(); // plain malloc'd memory: "invalid value", not an error of ours
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

static int ensure_pinned(void*& p, size_t& have, size_t need)
{
    if (have >= need && p) return RSR_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    have = 0;
    const hipError_t e = hipHostMalloc(&p, need, hipHostMallocDefault);
    if (e != hipSuccess)
    {
        p = nullptr;
        (void)hipGetLastError();
        return Engine::fail(RSR_E_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    }
    have = need;
    return RSR_OK;
}

// RealSR::process for host images.  The call owns a lane: upload on the lane's copy stream, kernels on the compute
// stream (ordered by events), download on the copy stream again.  Pinned caller memory (rsr_host_alloc, hipHostMalloc,
// hipHostRegister) is copied directly; pageable memory goes through the lane's pinned staging, the download in chunks so
// that the CPU copy of chunk i overlaps the PCIe transfer of chunk i+1.
// tile0 / tile1: only tiles [tile0, tile1) of the row-major tile grid (tiles are independent: realsr.cpp:377-380,458-459,490).
// A range of whole tile rows is a contiguous byte range of the output; a general range is up to three rectangles -- the
// tail of its first tile row, whole rows, the head of its last row -- fetched with 2-D copies.
int Engine::process_host(const uint8_t* in, int w, int h, int c, uint8_t* out, int tile0, int tile1)
{
    if (!in || !out || w < 1 || h < 1 || (c != 3 && c != 4)) return fail(RSR_E_ARG, "bad image arguments");
    const size_t nin = size_t(w) * h * c, nout_full = nin * size_t(scale) * scale;
    int T = 0, mwidth = 1;
    long long mitems = 0;
    { // state checks BEFORE anything is enqueued on behalf of this call (and the engine's settings read under its lock)
        std::lock_guard<std::mutex> lk(mu);
        if (!loaded) return fail(RSR_E_STATE, "process before load");
        if (scale != 4) return fail(RSR_E_ARG, "only scale 4 is supported (main.cpp:533-537)");
        T = tilesize;
        mwidth = merge_width(w, h, c);
        if (mwidth > 1) mitems = image_items(w, h, merge_target_items);
    }
    const int xtiles = (w + T - 1) / T, ytiles = (h + T - 1) / T;
    if (tile1 < 0) tile1 = xtiles * ytiles;
    if (tile0 < 0 || tile1 > xtiles * ytiles || tile0 >= tile1) return fail(RSR_E_ARG, "tile range outside the image");
    // a small whole image is merged with the images other callers hand in meanwhile (Engine::submit_merged); a leader that is
    // forming a batch gives the calls counted here a moment to finish their upload
    const bool mergeable = tile0 == 0 && tile1 == xtiles * ytiles && mwidth > 1;
    struct Inbound
    {
        std::atomic<int>* n;
        ~Inbound() { release(); }
        void release()
        {
            if (n) --*n;
            n = nullptr;
        }
    } inbound{nullptr};
    Lane* L = acquire_lane();
    if (mergeable)
    { // counted from here: a call that is still waiting for a lane is not "on its way" (lanes free up when batches complete)
        ++merge_inbound;
        inbound.n = &merge_inbound;
    }
    // Whatever way this call ends, nothing of it may still be in flight when the lane -- and with it the caller's `in` / `out`
    // -- is handed back: a HIP failure half way must not turn into a use-after-free of the lane buffers by the next caller.
    struct Release
    {
        Engine* e;
        Lane* l;
        bool done_recorded = false;
        ~Release()
        {
            if (l->copy) (void)hipStreamSynchronize(l->copy);
            if (done_recorded) (void)hipEventSynchronize(l->ev_done);
            e->release_lane(l);
        }
    } guard{this, L};
    HIP_TRY(hipSetDevice(device));
    if (!L->copy)
    {
        HIP_TRY(hipStreamCreateWithFlags(&L->copy, hipStreamNonBlocking));
        for (hipEvent_t* e : {&L->ev_in, &L->ev_done, &L->ev_half, &L->ev_chunk[0], &L->ev_chunk[1]}) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    int rc;
    if ((rc = ensure(L->d_in, nin)) != RSR_OK) return rc;  // lane-private: nothing else can be using the old allocation
    // The device output holds only the output ROWS of this call's tile range (a member of rsr_process_group running an eighth
    // of a 4K frame does not allocate 398 MB for it): the kernels get the address row 0 would have and only ever write
    // inside the rectangles of the range's tiles (conv_last / postproc_tiles place by tile), i.e. inside the allocation.
    const size_t rowbytes = size_t(w) * scale * c; // one output row
    auto yof = [&](int tr) { return size_t(std::min(tr * T, h)) * scale; };  // first output row of tile row tr
    auto xof = [&](int tc) { return size_t(std::min(tc * T, w)) * scale * c; }; // byte column of tile column tc
    const int r0 = tile0 / xtiles, c0 = tile0 % xtiles, r1 = (tile1 - 1) / xtiles, c1 = (tile1 - 1) % xtiles + 1; // last tile = (r1, c1 - 1)
    const size_t base_off = yof(r0) * rowbytes;
    if ((rc = ensure(L->d_out, (yof(r1 + 1) - yof(r0)) * rowbytes)) != RSR_OK) return rc;
    L->in_bytes.store(L->d_in.bytes, std::memory_order_relaxed); // what device_avail() / rsr_get_stat read (they do not hold this lane)
    L->out_bytes.store(L->d_out.bytes, std::memory_order_relaxed);
    // the kernels get the REAL base of the allocation: the plan of a tile range places its tiles relative to the range's first
    // output row (Plan::out_row0 == yof(r0), get_plan); dev(off) = device address of byte `off` of the full-size image
    char* const dbase = static_cast<char*>(L->d_out.p);
    auto dev = [&](size_t off) { return dbase + (off - base_off); };

    // ---- upload ----
    const void* src = in;
    if (!is_pinned_host(in))
    {
        if ((rc = ensure_pinned(L->h_in, L->h_in_bytes, nin)) != RSR_OK) return rc;
        pool.copy(L->h_in, in, nin, copy_threads);
        src = L->h_in;
    }
    HIP_TRY(hipMemcpyAsync(L->d_in.p, src, nin, hipMemcpyHostToDevice, L->copy));
    HIP_TRY(hipEventRecord(L->ev_in, L->copy));

    // ---- network ----
    size_t half_rows = 0;
    if (mergeable)
    {
        MergeReq r;
        r.d_in = L->d_in.p; r.d_out = dbase; r.w = w; r.h = h; r.c = c; r.T = T;
        r.items = mitems;
        r.width = mwidth;
        r.ev_in = L->ev_in;
        r.ev_done = L->ev_done;
        inbound.release(); // (it is in the queue the moment submit_merged has the lock: the leader's wait ends either way)
        rc = submit_merged(r);
        if (rc != RSR_OK) return rc; // (a batch that failed half way has been drained by its leader: nothing of it is in flight)
        guard.done_recorded = true;
        void (*cb)(int, int, void*) = nullptr;
        void* cb_user = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu);
            cb = progress;
            cb_user = progress_user;
        }
        if (cb)
            for (int i = 1; i <= xtiles * ytiles; i++) cb(i, xtiles * ytiles, cb_user);
    }
    else
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!loaded || scale != 4 || tilesize != T)
            return fail(RSR_E_STATE, "context parameters changed while the call was in flight"); // (the guard drains the upload)
        HIP_TRY(hipStreamWaitEvent(stream, L->ev_in, 0));
        rc = enqueue_image(L->d_in.p, w, h, c, dbase, stream, tile0, tile1, L->ev_half, &half_rows);
        if (rc != RSR_OK)
        {
            (void)hipStreamSynchronize(stream); // kernels of this call that did get enqueued use the lane buffers
            return rc;
        }
        HIP_TRY(hipEventRecord(L->ev_done, stream));
        guard.done_recorded = true;
    }

    // ---- download ----
    if (c0 != 0 || c1 != xtiles)
    { // general tile range: up to three rectangles.  Pinned destinations receive them directly; pageable ones through the
      // lane's pinned staging, in row chunks (the CPU copy of chunk i under the PCIe transfer of chunk i+1), like whole rows below
        HIP_TRY(hipStreamWaitEvent(L->copy, L->ev_done, 0));
        const bool pinned = is_pinned_host(out);
        const size_t CH = std::max<size_t>(chunk_bytes, 1 << 20);
        if (!pinned && (rc = ensure_pinned(L->h_out, L->h_out_bytes, 2 * std::min(CH, nout_full))) != RSR_OK) return rc;
        auto rect = [&](size_t y_a, size_t y_b, size_t x_a, size_t x_b) -> int {
            if (y_b <= y_a || x_b <= x_a) return RSR_OK;
            const size_t wb = x_b - x_a;
            if (pinned)
            {
                const size_t off = y_a * rowbytes + x_a;
                HIP_TRY(hipMemcpy2DAsync(out + off, rowbytes, dev(off), rowbytes, wb, y_b - y_a, hipMemcpyDeviceToHost, L->copy));
                return RSR_OK;
            }
            const size_t half = L->h_out_bytes / 2, rows_per = std::max<size_t>(1, half / wb);
            if (wb > half) return fail(RSR_E_ARG, "chunk_mb smaller than one output row");
            const size_t nch = (y_b - y_a + rows_per - 1) / rows_per;
            for (size_t i = 0; i <= nch; i++)
            {
                if (i < nch)
                {
                    const size_t ya = y_a + i * rows_per, nr = std::min(rows_per, y_b - ya);
                    HIP_TRY(hipMemcpy2DAsync(static_cast<char*>(L->h_out) + (i & 1) * half, wb, dev(ya * rowbytes + x_a), rowbytes, wb, nr,
                                             hipMemcpyDeviceToHost, L->copy));
                    HIP_TRY(hipEventRecord(L->ev_chunk[i & 1], L->copy));
                }
                if (i >= 1)
                {
                    const size_t k = i - 1, ya = y_a + k * rows_per, nr = std::min(rows_per, y_b - ya);
                    HIP_TRY(hipEventSynchronize(L->ev_chunk[k & 1]));
                    const char* sp = static_cast<const char*>(L->h_out) + (k & 1) * half;
                    for (size_t y = 0; y < nr; y++) std::memcpy(out + (ya + y) * rowbytes + x_a, sp + y * wb, wb);
                }
            }
            return RSR_OK;
        };
        if (r0 == r1) { if ((rc = rect(yof(r0), yof(r0 + 1), xof(c0), xof(c1))) != RSR_OK) return rc; }
        else
        {
            int full0 = r0, full1 = r1 + 1; // whole tile rows [full0, full1)
            if (c0 != 0)
            {
                if ((rc = rect(yof(r0), yof(r0 + 1), xof(c0), rowbytes)) != RSR_OK) return rc;
                full0 = r0 + 1;
            }
            if (c1 != xtiles)
            {
                if ((rc = rect(yof(r1), yof(r1 + 1), 0, xof(c1))) != RSR_OK) return rc;
                full1 = r1;
            }
            if ((rc = rect(yof(full0), yof(full1), 0, rowbytes)) != RSR_OK) return rc;
        }
        HIP_TRY(hipStreamSynchronize(L->copy));
        return RSR_OK;
    }
    // whole tile rows [r0, r1]: a contiguous byte range of the HWC image.  When the engine split the 4x tail, the first
    // `half_rows` output rows are complete at ev_half: they travel while the second part is still being computed.
    const size_t out_off = yof(r0) * rowbytes, nout = (yof(r1 + 1) - yof(r0)) * rowbytes;
    out += out_off;
    const size_t first = std::min(nout, half_rows * rowbytes);
    const bool pinned_out = is_pinned_host(out);
    if (pinned_out)
    {
        const char* dsrc = dev(out_off);
        if (first)
        {
            HIP_TRY(hipStreamWaitEvent(L->copy, L->ev_half, 0));
            HIP_TRY(hipMemcpyAsync(out, dsrc, first, hipMemcpyDeviceToHost, L->copy));
        }
        HIP_TRY(hipStreamWaitEvent(L->copy, L->ev_done, 0));
        HIP_TRY(hipMemcpyAsync(out + first, dsrc + first, nout - first, hipMemcpyDeviceToHost, L->copy));
        HIP_TRY(hipStreamSynchronize(L->copy));
        return RSR_OK;
    }
    HIP_TRY(hipStreamWaitEvent(L->copy, first ? L->ev_half : L->ev_done, 0));
    const size_t CH = std::max<size_t>(chunk_bytes, 1 << 20);
    if ((rc = ensure_pinned(L->h_out, L->h_out_bytes, 2 * std::min(CH, nout))) != RSR_OK) return rc;
    const size_t half = L->h_out_bytes / 2;
    const size_t nchunks = (nout + half - 1) / half;
    const char* dsrc = dev(out_off);
    bool waited_done = false;
    for (size_t i = 0; i <= nchunks; i++)
    {
        if (i < nchunks)
        {
            const size_t off = i * half, n = std::min(half, nout - off);
            if (first && !waited_done && off + n > first)
            { // this chunk reaches into the rows of the second part
                HIP_TRY(hipStreamWaitEvent(L->copy, L->ev_done, 0));
                waited_done = true;
            }
            // slot i&1 was drained by the CPU copy of chunk i-2 in the previous iteration
            HIP_TRY(hipMemcpyAsync(static_cast<char*>(L->h_out) + (i & 1) * half, dsrc + off, n, hipMemcpyDeviceToHost, L->copy));
            HIP_TRY(hipEventRecord(L->ev_chunk[i & 1], L->copy));
        }
        if (i >= 1)
        {
            const size_t k = i - 1, off = k * half, n = std::min(half, nout - off);
            HIP_TRY(hipEventSynchronize(L->ev_chunk[k & 1]));
            pool.copy(out + off, static_cast<const char*>(L->h_out) + (k & 1) * half, n, copy_threads);
        }
    }
    return RSR_OK;
}

// one tile through the network only (layer-level parity hook)
int Engine::net_forward(const uint16_t* in, int w, int h, uint16_t* out, float* out32)
{
    if (!in || (!out && !out32) || w < 1 || h < 1) return fail(RSR_E_ARG, "bad arguments");
    std::lock_guard<std::mutex> lk(mu);
    if (!loaded) return fail(RSR_E_STATE, "net_forward before load");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamSynchronize(stream));
    Plan::Batch b;
    b.ntiles = 1;
    b.nslots = 1;
    b.dims.push_back(TileDim{h, w});
    make_items(b, fold_cols);
    const long long cap = (long long)w * h;
    if (cap * 16 * 32 * ((flow_flags & 1) ? 4 : 2) + 4 * kGuard >= (1ll << 31)) return fail(RSR_E_ARG, "tile too large");
    DevBuf tab, tmp;
    auto cleanup = [&]() {
        if (tab.p) (void)hipFree(tab.p);
        if (tmp.p) (void)hipFree(tmp.p);
    };
    int rc;
    if ((rc = ensure(tab, batch_table_bytes(b))) != RSR_OK) return rc;
    char* d = static_cast<char*>(tab.p);
    hipError_t he = upload_batch(b, d);
    const size_t npx = size_t(w) * h;
    if (he == hipSuccess && (rc = ensure_workspace(1, cap, stream)) != RSR_OK) { cleanup(); return rc; }
    if (he == hipSuccess && (rc = ensure(tmp, npx * 6)) != RSR_OK) { cleanup(); return rc; }
    if (he == hipSuccess) he = hipMemcpy(tmp.p, in, npx * 6, hipMemcpyHostToDevice);
    if (he == hipSuccess)
    {
        launch_planar3_to_plane(static_cast<const uint16_t*>(tmp.p), w, h, static_cast<char*>(b_in.p) + kGuard, plane_ch(), stream);
        const bool was = profiling;
        profiling = false;
        rc = run_network(b, stream);
        profiling = was;
        he = hipStreamSynchronize(stream);
        if (he == hipSuccess) he = hipGetLastError();
        if (he == hipSuccess && rc == RSR_OK && !precise)
        {
            if (out) he = hipMemcpy(out, b_out3.p, npx * 16 * 6, hipMemcpyDeviceToHost);
            if (out32) rc = fail(RSR_E_STATE, "the fp32 output blob exists in precise mode only (rsr_set_option precise 1)");
        }
        else if (he == hipSuccess && rc == RSR_OK)
        { // precise mode: conv_last left fp32; the fp16 view of it is rounded here, on the host
            std::vector<float> tmp32(out32 ? 0 : npx * 16 * 3);
            float* dst = out32 ? out32 : tmp32.data();
            he = hipMemcpy(dst, b_out3.p, npx * 16 * 12, hipMemcpyDeviceToHost);
            if (he == hipSuccess && out)
                for (size_t i = 0; i < npx * 16 * 3; i++)
                {
                    const _Float16 hv = (_Float16)dst[i];
                    std::memcpy(&out[i], &hv, 2);
                }
        }
    }
    cleanup();
    if (he != hipSuccess) return fail(RSR_E_DEVICE, std::string("net_forward: ") + hipGetErrorString(he));
    return rc;
}

// one convolution with caller-supplied weights (layer-level parity hook, include/realsr_hip.h rsr_conv3x3)
int Engine::conv_test(const uint16_t* in, int cin, int h, int w, int ups, const float* weight, const float* bias, int cout,
                      int lrelu, uint16_t* out, float s1, int own_res, const uint16_t* res, float s2, bool prec, const uint8_t* in_lo,
                      const uint8_t* res_lo, uint8_t* out_lo)
{
    if (!in || !weight || !bias || !out || cin < 1 || cout < 1 || cout > 64 || h < 1 || w < 1) return fail(RSR_E_ARG, "bad arguments");
    const bool residual = s1 != 0.f;
    if (residual && (ups || lrelu || (own_res && cin < cout) || (cout % 32))) return fail(RSR_E_ARG, "residual form: no upsampling / activation, cout 32 or 64");
    if (prec && (!residual || cout != 64 || (in_lo && !own_res) || (res_lo && !res))) return fail(RSR_E_ARG, "precise form: a residual form with 64 output channels");
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
    constexpr int pch = plane_ch();
    const int np = int(pc.nplanes) * (32 / pch), nt = int(pc.nt); // input planes (cin padded to a multiple of 32)
    const int npo = nt * (32 / pch);                              // output planes
    const int H = ups ? 2 * h : h, W = ups ? 2 * w : w;
    const size_t ipx = size_t(h) * w, opx = size_t(H) * W;
    // planar [cin][h][w] -> guarded planes [np][h][w][pch]
    const size_t ipl = ipx * size_t(pch) + kGuard / 2; // halfs per guarded input plane
    // precise form: the one-byte lo planes of a tensor (plane stride / 2) follow its hi planes in the same allocation, `lo_off` bytes
    // behind pixel 0 of plane 0 (kernels.h ConvArgs::lo1_off)
    const size_t ipl_b = ipl * 2, opl_b = opx * size_t(pch) * 2; // bytes per (guarded) input plane / per output plane
    std::vector<uint16_t> hin(size_t(np + (in_lo ? npo : 0)) * ipl, 0), hout(size_t(npo) * (out_lo ? 2 : 1) * opx * size_t(pch), 0);
    for (int ch = 0; ch < cin; ch++)
        for (size_t p = 0; p < ipx; p++) hin[size_t(ch / pch) * ipl + kGuard / 2 + p * size_t(pch) + size_t(ch % pch)] = in[size_t(ch) * ipx + p];
    // (the lo bytes of an n-tile's two planes share one "pair plane" of the hi geometry: [pixel][half][plane][8 channels], conv_flow.hip lo_row)
    auto lo_index = [&](int ch, size_t p, size_t pair_stride) {
        return size_t(ch / 32) * pair_stride + p * 32 + size_t((ch % 16) / 8) * 16 + size_t((ch % 32) / 16) * 8 + size_t(ch % 8);
    };
    auto put_lo = [&](std::vector<uint16_t>& buf, size_t lo_off, const uint8_t* lo) {
        unsigned char* b = reinterpret_cast<unsigned char*>(buf.data()) + kGuard + lo_off;
        for (int ch = 0; ch < cout; ch++)
            for (size_t p = 0; p < ipx; p++) b[lo_index(ch, p, ipl_b)] = lo[size_t(ch) * ipx + p];
    };
    if (in_lo) put_lo(hin, size_t(np) * ipl_b, in_lo);
    std::vector<uint16_t> hres;
    if (residual && res)
    { // planar [cout][h][w] -> guarded planes like the input
        hres.assign(size_t(npo) * (res_lo ? 2 : 1) * ipl, 0);
        for (int ch = 0; ch < cout; ch++)
            for (size_t p = 0; p < ipx; p++) hres[size_t(ch / pch) * ipl + kGuard / 2 + p * size_t(pch) + size_t(ch % pch)] = res[size_t(ch) * ipx + p];
        if (res_lo) put_lo(hres, size_t(npo) * ipl_b, res_lo);
    }
    DevBuf d_w, d_in, d_out, d_tab, d_res;
    std::vector<WorkItem> items;
    // test_repeat > 1 (measurement aid): the same blocks N times in ONE launch -- after the first pass every patch and every
    // output line is in the L2s, i.e. the launch shows what this conv costs when nothing goes to HBM (tools/l2_bound_probe.py)
    for (int rep = 0; rep < std::max(1, test_repeat); rep++)
        append_block_items(items, 0, H, W, 0, 0, 0, 0, 0, fold_cols);
    const TileDim td{h, w};
    auto cleanup = [&]() {
        for (DevBuf* b : {&d_w, &d_in, &d_out, &d_tab, &d_res})
            if (b->p) (void)hipFree(b->p);
    };
    if ((rc = ensure(d_w, pk.size())) != RSR_OK || (rc = ensure(d_in, hin.size() * 2)) != RSR_OK ||
        (rc = ensure(d_out, hout.size() * 2)) != RSR_OK || (rc = ensure(d_tab, 256 + items.size() * sizeof(WorkItem))) != RSR_OK ||
        (!hres.empty() && (rc = ensure(d_res, hres.size() * 2)) != RSR_OK))
    {
        cleanup();
        return rc;
    }
    hipError_t he = hipMemcpy(d_w.p, pk.data(), pk.size(), hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemcpy(d_in.p, hin.data(), hin.size() * 2, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemsetAsync(d_out.p, 0, hout.size() * 2, stream);
    if (he == hipSuccess) he = hipMemcpy(d_tab.p, &td, sizeof td, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemcpy(static_cast<char*>(d_tab.p) + 256, items.data(), items.size() * sizeof(WorkItem), hipMemcpyHostToDevice);
    if (he == hipSuccess && !hres.empty()) he = hipMemcpy(d_res.p, hres.data(), hres.size() * 2, hipMemcpyHostToDevice);
    if (he == hipSuccess)
    {
        ConvArgs a;
        std::memset(&a, 0, sizeof a);
        a.src0 = PlaneSrc{static_cast<char*>(d_in.p) + kGuard, 0, (long long)ipx * pch * 2 + kGuard};
        a.n0 = np;
        a.lvl_in = 0;
        a.lvl_out = ups ? 1 : 0;
        a.wpk16 = static_cast<const char*>(d_w.p) + pc.w16_off;
        a.waux = pc.aux_off ? static_cast<const char*>(d_w.p) + pc.aux_off : nullptr;
        a.bias = reinterpret_cast<const float*>(static_cast<const char*>(d_w.p) + pc.b_off);
        a.lrelu = lrelu;
        a.s1 = a.s2 = 1.f;
        if (residual)
        { // the epilogue forms of RDB conv5 / trunk_conv (run_network)
            a.s1 = s1;
            a.precise = prec ? 1 : 0;
            const long long res_lo_off = res_lo ? (long long)npo * (long long)ipl * 2 : 0;
            if (own_res)
            {
                a.res1 = a.src0;
                a.res1_kind = 1;
                a.res1_in_acc = 1;
                a.res1_coef = 1.f / s1;
                a.lo1_off = in_lo ? (long long)np * (long long)ipl * 2 : 0;
            }
            if (res)
            {
                const PlaneSrc rp{static_cast<char*>(d_res.p) + kGuard, 0, (long long)ipx * pch * 2 + kGuard};
                if (own_res) { a.res2 = rp; a.res2_kind = 1; a.s2 = s2; a.lo2_off = res_lo_off; }
                else { a.res1 = rp; a.res1_kind = 1; a.lo1_off = res_lo_off; } // trunk_conv form: v = s1*(conv+b) + res  (s2 unused)
            }
            if (out_lo) a.out_lo_off = (long long)npo * (long long)opx * pch * 2;
        }
        a.out16 = PlaneSrc{d_out.p, 0, (long long)opx * pch * 2};
        a.items = reinterpret_cast<const WorkItem*>(static_cast<const char*>(d_tab.p) + 256);
        a.nitems = int(items.size());
        a.dims = static_cast<const TileDim*>(d_tab.p);
        a.zeros = zeros.p;
        a.dbg = dbg;
        hipEvent_t e0 = take_event_timed(), e1 = take_event_timed();
        if (e0) (void)hipEventRecord(e0, stream);
        if (!launch_conv_flow(a, nt, num_cu, flow_flags, stream)) rc = fail(RSR_E_STATE, "conv3x3_flow: no variant");
        if (e1) (void)hipEventRecord(e1, stream);
        he = hipStreamSynchronize(stream);
        float ms = 0.f;
        if (e0 && e1 && he == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) last_test_us = double(ms) * 1e3;
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (he == hipSuccess) he = hipGetLastError();
        if (he == hipSuccess) he = hipMemcpy(hout.data(), d_out.p, hout.size() * 2, hipMemcpyDeviceToHost);
    }
    cleanup();
    if (he != hipSuccess) return fail(RSR_E_DEVICE, std::string("conv_test: ") + hipGetErrorString(he));
    if (rc != RSR_OK) return rc;
    for (int ch = 0; ch < cout; ch++)
        for (size_t p = 0; p < opx; p++)
        {
            out[size_t(ch) * opx + p] = hout[(size_t(ch / pch) * opx + p) * size_t(pch) + size_t(ch % pch)];
            if (out_lo) out_lo[size_t(ch) * opx + p] = reinterpret_cast<const unsigned char*>(hout.data())[size_t(npo) * opl_b + lo_index(ch, p, opl_b)];
        }
    return RSR_OK;
}

} // namespace rsr
