// realsr.h -- C++ host-side mirror of the reference's `class RealSR` (/root/reference/src/realsr.h:13-42) on top
// of the C-ABI (include/realsr_hip.h).  Same constructor, same method names, same public fields, so that a
// main.cpp-shaped caller reads the same; `ncnn::Mat` is replaced by the 4-field `Image` below (the reference only
// ever reads .data/.w/.h/.elempack from it, realsr.cpp:153-156).  Header-only; link against librealsr_hip.so.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/realsr_hip.h"

struct Image // what the reference passes as ncnn::Mat(w, h, data, elemsize=c, elempack=c)  (main.cpp:275-276)
{
    int w = 0, h = 0, elempack = 0; // elempack = channels (3 RGB / 4 RGBA), HWC uint8, tightly packed

    Image() = default;
    Image(const Image&) = delete;
    Image& operator=(const Image&) = delete;
    Image(Image&& o) noexcept { *this = static_cast<Image&&>(o); }
    Image& operator=(Image&& o) noexcept
    {
        if (this != &o)
        {
            release();
            w = o.w; h = o.h; elempack = o.elempack; ptr = o.ptr; bytes = o.bytes; pinned = o.pinned;
            o.ptr = nullptr;
            o.bytes = 0;
            o.w = o.h = o.elempack = 0;
        }
        return *this;
    }
    ~Image() { release(); }

    uint8_t* data() { return ptr; }
    const uint8_t* data() const { return ptr; }
    // pinned_memory: allocate with rsr_host_alloc, so that rsr_process moves the pixels over PCIe without a staging copy
    // (what ncnn's Vulkan staging allocator does for the reference, realsr.cpp:161-167)
    void create(int w_, int h_, int c_, bool pinned_memory = false)
    {
        release();
        w = w_;
        h = h_;
        elempack = c_;
        bytes = size_t(w_) * h_ * c_;
        pinned = pinned_memory;
        ptr = pinned ? static_cast<uint8_t*>(rsr_host_alloc(bytes)) : static_cast<uint8_t*>(std::malloc(bytes ? bytes : 1));
        if (!ptr && pinned)
        { // no device / pinning refused: plain memory still works, just slower
            pinned = false;
            ptr = static_cast<uint8_t*>(std::malloc(bytes ? bytes : 1));
        }
    }
    bool empty() const { return ptr == nullptr; }

private:
    void release()
    {
        if (ptr)
        {
            if (pinned) rsr_host_free(ptr);
            else std::free(ptr);
        }
        ptr = nullptr;
        bytes = 0;
    }
    uint8_t* ptr = nullptr;
    size_t bytes = 0;
    bool pinned = false;
};

class RealSR
{
public:
    RealSR(int gpuid, bool tta_mode = false, int num_threads = 1) : scale(4), tilesize(200), prepadding(10), ctx(nullptr)
    {
        const int rc = rsr_create(&ctx, gpuid, tta_mode ? 1 : 0, num_threads);
        if (rc != RSR_OK) std::fprintf(stderr, "RealSR: %s\n", rsr_last_error(nullptr));
    }
    // adopt a context created elsewhere (rsr_create_group: the model arrives by one RCCL broadcast instead of load())
    explicit RealSR(rsr_ctx* adopted) : scale(4), tilesize(200), prepadding(10), ctx(adopted) {}
    ~RealSR() { rsr_destroy(ctx); }
    RealSR(const RealSR&) = delete;
    RealSR& operator=(const RealSR&) = delete;

    bool ok() const { return ctx != nullptr; }

    int load(const std::string& parampath, const std::string& modelpath)
    {
        if (!ctx) return RSR_E_STATE;
        const int rc = rsr_load(ctx, parampath.c_str(), modelpath.c_str());
        if (rc != RSR_OK) std::fprintf(stderr, "RealSR::load: %s\n", rsr_last_error(ctx));
        return rc;
    }
    // multi-GPU: parse + pack once, hand the same blob to every context (the reference re-reads the file per GPU)
    int load_packed(const void* blob, size_t bytes)
    {
        if (!ctx) return RSR_E_STATE;
        const int rc = rsr_load_packed(ctx, blob, bytes, 0);
        if (rc != RSR_OK) std::fprintf(stderr, "RealSR::load_packed: %s\n", rsr_last_error(ctx));
        return rc;
    }

    int process(const Image& inimage, Image& outimage) const
    {
        if (!ctx) return RSR_E_STATE;
        int rc = rsr_set_params(ctx, scale, tilesize, prepadding);
        if (rc == RSR_OK)
        {
            if (outimage.w != inimage.w * scale || outimage.h != inimage.h * scale || outimage.elempack != inimage.elempack)
                outimage.create(inimage.w * scale, inimage.h * scale, inimage.elempack, true);
            rc = rsr_process(ctx, inimage.data(), inimage.w, inimage.h, inimage.elempack, outimage.data());
        }
        if (rc != RSR_OK) std::fprintf(stderr, "RealSR::process: %s\n", rsr_last_error(ctx));
        return rc;
    }

    // ONE image over several GPUs: the tile rows are split over the group; all members run with the first
    // member's parameters
    static int process_group(const std::vector<RealSR*>& group, const Image& inimage, Image& outimage)
    {
        if (group.empty()) return RSR_E_ARG;
        const RealSR& r0 = *group[0];
        std::vector<rsr_ctx*> ctxs;
        int rc = RSR_OK;
        for (RealSR* r : group)
        {
            if (!r->ctx) return RSR_E_STATE;
            if (rc == RSR_OK) rc = rsr_set_params(r->ctx, r0.scale, r0.tilesize, r0.prepadding);
            ctxs.push_back(r->ctx);
        }
        if (rc == RSR_OK)
        {
            if (outimage.w != inimage.w * r0.scale || outimage.h != inimage.h * r0.scale || outimage.elempack != inimage.elempack)
                outimage.create(inimage.w * r0.scale, inimage.h * r0.scale, inimage.elempack, true);
            rc = rsr_process_group(ctxs.data(), int(ctxs.size()), inimage.data(), inimage.w, inimage.h, inimage.elempack, outimage.data());
        }
        if (rc != RSR_OK) std::fprintf(stderr, "RealSR::process_group: %s\n", rsr_last_error(nullptr));
        return rc;
    }

public:
    // realsr parameters (assigned after load(), main.cpp:788-790)
    int scale;
    int tilesize;
    int prepadding;

private:
    rsr_ctx* ctx;
};
