// group.cpp -- several GPUs of one node behind the C-ABI: rsr_create_group / rsr_process_tiles / rsr_process_rows / rsr_process_group.
//
// The reference creates one RealSR per GPU and lets every one of them re-read x4.bin (main.cpp:778-791).  Here the model is
// parsed, validated and packed ONCE; the blob is uploaded to the first GPU and reaches the others through ONE RCCL
// broadcast over xGMI (the only collective of this workload -- tiles and images are independent).
// RCCL is resolved with dlopen at run time: the library has no link-time dependency on it, a process that already
// carries an RCCL (PyTorch) shares that copy, and a single-GPU user never loads it.
#include <dlfcn.h>
#include <pthread.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "engine.h"

using namespace rsr;

namespace {

// the six entry points we use, as declared by rccl/rccl.h (ncclResult_t = int, ncclComm_t = opaque pointer,
// ncclDataType_t ncclUint8 = 1)
struct Rccl
{
    void* lib = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void* send, void* recv, size_t count, int dtype, int root, void* comm, hipStream_t stream) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;

    bool open(std::string& why)
    {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"})
            if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
        if (!lib)
        {
            why = std::string("dlopen(librccl): ") + dlerror();
            return false;
        }
        auto sym = [&](const char* n) { return dlsym(lib, n); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Broadcast = reinterpret_cast<decltype(Broadcast)>(sym("ncclBroadcast"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Broadcast)
        {
            why = "librccl lacks a required symbol";
            return false;
        }
        return true;
    }
    std::string err(int rc) const { return GetErrorString ? GetErrorString(rc) : ("nccl error " + std::to_string(rc)); }
};

thread_local std::string g_transport = "none";

// Worker threads of rsr_process_group, kept between calls (one image = one call: starting `parts` fresh threads per image
// costs as much as a few tiles).  A worker is added only when a job finds none idle; the calling thread always runs share 0 itself.
// Robustness: a worker that cannot be created (std::system_error: thread limit, no memory) must not throw across the
// extern "C" boundary with the job already queued -- the job is taken back and run on the CALLING thread.  The pool lives on the heap
// and is never destroyed: a process exits without joining workers that sleep on a condition variable, and a fork()ed child (whose
// copy of the pool would name threads that do not exist there, so run() would wait for ever and a destructor would join nothing)
// gets a fresh, empty pool from a pthread_atfork child handler.  Consequence (include/realsr_hip.h says so): once rsr_process_group
// has run, the library must stay loaded -- dlclose() would unmap the code its sleeping workers return into.
struct SharePool
{
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    std::vector<std::thread> workers;
    int idle = 0;
    int inline_runs = 0; // jobs run on the caller's thread because no worker could be started (statistics / tests)
    void run(std::function<void()> f)
    {
        bool run_here = false;
        {
            std::lock_guard<std::mutex> lk(m);
            q.push_back(std::move(f));
            if (int(q.size()) > idle && workers.size() < 64) // nobody free for this job: one more worker (64 = more GPUs than a node has)
            {
                try
                {
                    static const bool no_threads = std::getenv("RSR_POOL_NO_THREADS") != nullptr; // test hook, read once per process
                    if (no_threads) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again));
                    workers.emplace_back([this] {
                        std::unique_lock<std::mutex> lk(m);
                        for (;;)
                        {
                            idle++;
                            cv.wait(lk, [this] { return !q.empty(); });
                            idle--;
                            std::function<void()> job = std::move(q.front());
                            q.pop_front();
                            lk.unlock();
                            job();
                            lk.lock();
                        }
                    });
                }
                catch (...) // (std::system_error: thread limit; std::bad_alloc from the vector or the std::function: nothing may cross extern "C")
                {
                    if (int(q.size()) > idle) // nobody will pick it up soon: take the job back (it is the newest) and run it here
                    {
                        f = std::move(q.back());
                        q.pop_back();
                        run_here = true;
                        inline_runs++;
                    }
                }
            }
        }
        if (run_here) f();
        else cv.notify_one();
    }
};
SharePool*& share_pool_ptr()
{
    static SharePool* p = nullptr;
    return p;
}
SharePool& share_pool()
{
    static std::once_flag once;
    std::call_once(once, [] {
        share_pool_ptr() = new SharePool;
        // the parent's workers do not exist in a forked child: start over with an empty pool (the old object is leaked on purpose --
        // its mutex may have been held by a thread that is gone)
        (void)pthread_atfork(nullptr, nullptr, [] { share_pool_ptr() = new SharePool; });
    });
    return *share_pool_ptr();
}

} // namespace

namespace rsr {
// rsr_get_stat "pool_workers" / "pool_inline_runs" (process-wide)
long long share_pool_stat(int what)
{
    SharePool& p = share_pool();
    std::lock_guard<std::mutex> lk(p.m);
    return what == 0 ? (long long)p.workers.size() : (long long)p.inline_runs;
}
} // namespace rsr

namespace {

// dst[i] (device i, `bytes` each) <- src on device gpuids[0], one broadcast.  false + why on any failure.
bool rccl_broadcast(const int* gpuids, int n, const void* src, void* const* dst, size_t bytes, std::string& why)
{
    Rccl r;
    if (!r.open(why)) return false;
    std::vector<void*> comms(size_t(n), nullptr);
    int rc = r.CommInitAll(comms.data(), n, gpuids);
    if (rc != 0)
    {
        why = "ncclCommInitAll: " + r.err(rc);
        return false;
    }
    std::vector<hipStream_t> streams(size_t(n), nullptr);
    bool ok = true;
    for (int i = 0; i < n && ok; i++)
        ok = hipSetDevice(gpuids[i]) == hipSuccess && hipStreamCreateWithFlags(&streams[size_t(i)], hipStreamNonBlocking) == hipSuccess;
    if (ok)
    {
        rc = r.GroupStart();
        for (int i = 0; i < n && rc == 0; i++)
        {
            (void)hipSetDevice(gpuids[i]);
            rc = r.Broadcast(i == 0 ? src : dst[i], dst[i], bytes, /*ncclUint8*/ 1, /*root*/ 0, comms[size_t(i)], streams[size_t(i)]);
        }
        const int rc2 = r.GroupEnd();
        if (rc == 0) rc = rc2;
        if (rc != 0)
        {
            why = "ncclBroadcast: " + r.err(rc);
            ok = false;
        }
    }
    else
        why = "stream creation failed";
    for (int i = 0; i < n; i++)
        if (streams[size_t(i)])
        {
            (void)hipSetDevice(gpuids[i]);
            if (hipStreamSynchronize(streams[size_t(i)]) != hipSuccess && ok)
            {
                ok = false;
                why = "broadcast stream failed";
            }
            (void)hipStreamDestroy(streams[size_t(i)]);
        }
    for (void* c : comms)
        if (c) (void)r.CommDestroy(c);
    return ok;
}

} // namespace

#pragma GCC visibility push(default)
extern "C" {

int rsr_create_group(rsr_ctx** out, const int* gpuids, int n, int tta_mode, const char* parampath, const char* modelpath)
{
    if (!out || !gpuids || n < 1 || !parampath || !modelpath) return Engine::fail(RSR_E_ARG, "bad arguments");
    // the calling thread's current device is the caller's business (a host application such as PyTorch relies on it)
    struct RestoreDevice
    {
        int prev = -1;
        RestoreDevice() { if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); } }
        ~RestoreDevice() { if (prev >= 0) (void)hipSetDevice(prev); }
    } restore_device;
    for (int i = 0; i < n; i++) out[i] = nullptr;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++)
            if (gpuids[i] == gpuids[j]) return Engine::fail(RSR_E_ARG, "duplicate gpu id in group");
    // parse + validate + pack once on the host (the slim blob: no round-1 weight images)
    Model m;
    std::string err;
    int rc = load_model(parampath, modelpath, m, err);
    if (rc != RSR_OK) return Engine::fail(rc, err);
    std::vector<unsigned char> blob(packed_size(m));
    rc = pack_model(m, blob.data(), blob.size(), err);
    if (rc != RSR_OK) return Engine::fail(rc, err);
    auto destroy_all = [&]() {
        for (int i = 0; i < n; i++)
        {
            delete out[i];
            out[i] = nullptr;
        }
    };
    for (int i = 0; i < n; i++)
    {
        rc = rsr_create(&out[i], gpuids[i], tta_mode, 1);
        if (rc != RSR_OK)
        {
            destroy_all();
            return rc;
        }
    }
    // First contact with a node of several GPUs: a device another job already fills is SAID here, once, at creation -- not found out
    // as a failed allocation inside the first frame.  The budget itself is left alone (a cap planted here would outlive a
    // neighbour that held the memory only for a moment): every plan is bounded by what the device can give when it is built, and a
    // batch whose workspace still cannot be allocated is halved and re-planned (engine.cpp get_plan / enqueue_images).
    for (int i = 0; i < n; i++)
    {
        size_t f = 0, t = 0;
        if (hipSetDevice(gpuids[i]) != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess) { (void)hipGetLastError(); continue; }
        Engine& e = out[i]->e;
        std::lock_guard<std::mutex> lk(e.mu);
        const long long room_mb = (long long)(f / 10 * 9 >> 20) - 2048;
        if (room_mb < e.max_workspace_mb || std::getenv("RSR_VERBOSE"))
            std::fprintf(stderr, "realsr-hip: group member %d on device %d: %.1f of %.1f GiB free, workspace budget %.1f GiB%s\n", i, gpuids[i],
                         double(f) / 1073741824.0, double(t) / 1073741824.0, double(e.max_workspace_mb) / 1024.0,
                         room_mb < e.max_workspace_mb ? " -- more than the device can give right now: tile batches will be sized by its free memory, plan by plan" : "");
    }
    g_transport = "host";
    bool done = false;
    // One GPU needs no broadcast -- unless RSR_GROUP_FORCE_RCCL=1 asks for the collective anyway (a communicator of one rank, an
    // in-place broadcast): the way to execute the whole RCCL branch -- dlopen, ncclCommInitAll, grouped ncclBroadcast, stream
    // sync, ncclCommDestroy, load from the device copy -- on a single-GPU box.
    const char* force = std::getenv("RSR_GROUP_FORCE_RCCL");
    if (n > 1 || (force && force[0] == '1'))
    {
        // device staging buffers: blob on GPU 0, receive buffers on the others
        std::vector<void*> dbuf(size_t(n), nullptr);
        bool ok = true;
        for (int i = 0; i < n && ok; i++)
            ok = hipSetDevice(gpuids[i]) == hipSuccess && hipMalloc(&dbuf[size_t(i)], blob.size()) == hipSuccess;
        ok = ok && hipSetDevice(gpuids[0]) == hipSuccess && hipMemcpy(dbuf[0], blob.data(), blob.size(), hipMemcpyHostToDevice) == hipSuccess;
        std::string why = "device staging failed";
        if (ok && rccl_broadcast(gpuids, n, dbuf[0], dbuf.data(), blob.size(), why))
        {
            done = true;
            for (int i = 0; i < n && rc == RSR_OK; i++)
            {
                std::lock_guard<std::mutex> lk(out[i]->e.mu);
                rc = out[i]->e.load_blob_device(dbuf[size_t(i)], blob.size());
            }
            g_transport = "rccl";
        }
        else
        {
            (void)hipGetLastError();
            g_transport = "host (rccl unavailable: " + why + ")";
        }
        for (int i = 0; i < n; i++)
            if (dbuf[size_t(i)])
            {
                (void)hipSetDevice(gpuids[i]);
                (void)hipFree(dbuf[size_t(i)]);
            }
    }
    if (!done)
        for (int i = 0; i < n && rc == RSR_OK; i++)
        {
            std::lock_guard<std::mutex> lk(out[i]->e.mu);
            rc = out[i]->e.load_blob_host(blob.data(), blob.size());
        }
    if (rc != RSR_OK)
    {
        const std::string keep = rsr::last_error();
        destroy_all();
        return Engine::fail(rc, keep);
    }
    return RSR_OK;
}

const char* rsr_group_transport(void) { return g_transport.c_str(); }

// Host-only: can the RCCL branch of rsr_create_group be taken at all?  dlopen + the entry points it calls, nothing else (no
// communicator, no device).  0 = yes; RSR_E_DEVICE + rsr_last_error() = which step failed.
int rsr_rccl_probe(void)
{
    Rccl r;
    std::string why;
    const bool ok = r.open(why);
    if (ok && !r.GetErrorString) why = "librccl lacks ncclGetErrorString";
    if (r.lib) dlclose(r.lib);
    if (!ok || !why.empty()) return Engine::fail(RSR_E_DEVICE, why);
    return RSR_OK;
}

int rsr_process_tiles(rsr_ctx* ctx, const uint8_t* in, int w, int h, int c, uint8_t* out, int tile_begin, int tile_end)
{
    if (!ctx) return RSR_E_ARG;
    return ctx->e.process_host(in, w, h, c, out, tile_begin, tile_end);
}

int rsr_process_rows(rsr_ctx* ctx, const uint8_t* in, int w, int h, int c, uint8_t* out, int tile_row_begin, int tile_row_end)
{
    if (!ctx) return RSR_E_ARG;
    if (w < 1 || h < 1 || tile_row_begin < 0 || tile_row_end <= tile_row_begin) return Engine::fail(RSR_E_ARG, "tile row range outside the image");
    int T;
    {
        std::lock_guard<std::mutex> lk(ctx->e.mu);
        T = ctx->e.tilesize;
    }
    const int xtiles = (w + T - 1) / T, ytiles = (h + T - 1) / T;
    if (tile_row_end > ytiles) return Engine::fail(RSR_E_ARG, "tile row range outside the image");
    return ctx->e.process_host(in, w, h, c, out, tile_row_begin * xtiles, tile_row_end * xtiles);
}

// Host-only: the contiguous row-major tile ranges rsr_process_group gives its shares.  bounds[0..parts]: share i runs tiles
// [bounds[i], bounds[i+1]).  Boundaries by padded tile area: share i ends at the first tile where the running load reaches
// (i+1)/parts of the total, every share gets at least one tile.  Returns the number of shares used (min(parts, tiles)).
int rsr_tile_partition(int w, int h, int tilesize, int prepadding, int parts, int* bounds)
{
    if (w < 1 || h < 1 || tilesize < 1 || prepadding < 0 || parts < 1 || !bounds) return Engine::fail(RSR_E_ARG, "bad arguments");
    const int T = tilesize, P = prepadding;
    const int xtiles = (w + T - 1) / T, ytiles = (h + T - 1) / T, ntiles = xtiles * ytiles;
    parts = std::min(parts, ntiles);
    std::vector<long long> cum(size_t(ntiles) + 1, 0);
    for (int t = 0; t < ntiles; t++)
    {
        const int yi = t / xtiles, xi = t % xtiles;
        const long long tw = std::min((xi + 1) * T, w) - xi * T + 2 * P, th = std::min((yi + 1) * T, h) - yi * T + 2 * P;
        cum[size_t(t) + 1] = cum[size_t(t)] + tw * th;
    }
    bounds[0] = 0;
    bounds[parts] = ntiles;
    for (int i = 1; i < parts; i++)
    {
        const long long target = cum[size_t(ntiles)] * i / parts;
        int b = int(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
        // the boundary tile goes to the side that leaves the smaller deviation from the target
        if (b > 0 && target - cum[size_t(b) - 1] < cum[size_t(b)] - target) b--;
        b = std::max(b, bounds[i - 1] + 1);     // every share gets at least one tile
        b = std::min(b, ntiles - (parts - i));  // ... also the ones behind it
        bounds[i] = b;
    }
    return parts;
}

// One image over n contexts.  The TILES of the grid (not its rows: a 1080p frame at tile 200 has 6 tile rows, the last one 80
// pixels high -- split by rows it could use 6 of 8 GPUs and wait for the slowest) are dealt in contiguous row-major ranges of
// equal padded-pixel load (within one tile), each context uploads the image, runs its range and fetches exactly the output
// rectangles of its tiles into the caller's buffer (realsr.cpp:377-380,458-459,490: tiles read a halo'd window of the
// immutable input and write disjoint rectangles).
int rsr_process_group(rsr_ctx* const* ctx, int n, const uint8_t* in, int w, int h, int c, uint8_t* out)
{
    if (!ctx || n < 1 || !in || !out || w < 1 || h < 1) return Engine::fail(RSR_E_ARG, "bad arguments");
    for (int i = 0; i < n; i++)
        if (!ctx[i]) return Engine::fail(RSR_E_ARG, "null context in group");
    // every member maps a tile range to pixels with its OWN parameters: they must agree, or rectangles are written twice / never
    int T = 0, P = 0, S = 0, tta = 0;
    for (int i = 0; i < n; i++)
    {
        std::lock_guard<std::mutex> lk(ctx[i]->e.mu);
        const Engine& e = ctx[i]->e;
        if (i == 0) { T = e.tilesize; P = e.prepadding; S = e.scale; tta = e.tta; }
        else if (e.tilesize != T || e.prepadding != P || e.scale != S || e.tta != tta)
            return Engine::fail(RSR_E_ARG, "group members carry different parameters (tilesize / prepadding / scale / tta)");
    }
    const int xtiles = (w + T - 1) / T, ytiles = (h + T - 1) / T, ntiles = xtiles * ytiles;
    const int parts = std::min(n, ntiles);
    if (parts == 1) return ctx[0]->e.process_host(in, w, h, c, out);
    std::vector<int> bound(size_t(parts) + 1, 0);
    const int prc = rsr_tile_partition(w, h, T, P, parts, bound.data());
    if (prc < 0) return prc;
    std::vector<int> rcs(size_t(parts), RSR_OK);
    std::vector<std::string> errs{size_t(parts)};
    auto share = [&](int i) {
        rcs[size_t(i)] = ctx[i]->e.process_host(in, w, h, c, out, bound[size_t(i)], bound[size_t(i) + 1]);
        if (rcs[size_t(i)] != RSR_OK) errs[size_t(i)] = rsr::last_error();
    };
    // shares 1.. on the pool's threads, share 0 on the caller's; the call returns when all have reported
    std::mutex dm;
    std::condition_variable dcv;
    int pending = parts - 1;
    for (int i = 1; i < parts; i++)
        share_pool().run([&, i] {
            share(i);
            std::lock_guard<std::mutex> lk(dm);
            if (--pending == 0) dcv.notify_one();
        });
    share(0);
    {
        std::unique_lock<std::mutex> lk(dm);
        dcv.wait(lk, [&] { return pending == 0; });
    }
    for (int i = 0; i < parts; i++)
        if (rcs[size_t(i)] != RSR_OK) return Engine::fail(rcs[size_t(i)], "gpu share " + std::to_string(i) + ": " + errs[size_t(i)]);
    return RSR_OK;
}

} // extern "C"
#pragma GCC visibility pop
