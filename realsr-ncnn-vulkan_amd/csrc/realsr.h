// realsr.h -- C++ host-side mirror of the reference's `class RealSR` (/root/reference/src/realsr.h:13-42) on top
// of the C-ABI (include/realsr_hip.h).  Same constructor, same method names, same public fields, so that a
// main.cpp-shaped caller reads the same; `ncnn::Mat` is replaced by the 4-field `Image` below (the reference only
// ever reads .data/.w/.h/.elempack from it, realsr.cpp:153-156).  Header-only; link against librealsr_hip.so.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/realsr_hip.h"

struct Image // what the reference passes as ncnn::Mat(w, h, data, elemsize=c, elempack=c)  (main.cpp:275-276)
{
    int w = 0, h = 0, elempack = 0; // elempack = channels (3 RGB / 4 RGBA), HWC uint8, tightly packed
    std::vector<uint8_t> pixels;
    uint8_t* data() { return pixels.data(); }
    const uint8_t* data() const { return pixels.data(); }
    void create(int w_, int h_, int c_)
    {
        w = w_;
        h = h_;
        elempack = c_;
        pixels.assign(size_t(w_) * h_ * c_, 0);
    }
    bool empty() const { return pixels.empty(); }
};

class RealSR
{
public:
    RealSR(int gpuid, bool tta_mode = false, int num_threads = 1) : scale(4), tilesize(200), prepadding(10), ctx(nullptr)
    {
        const int rc = rsr_create(&ctx, gpuid, tta_mode ? 1 : 0, num_threads);
        if (rc != RSR_OK) std::fprintf(stderr, "RealSR: %s\n", rsr_last_error(nullptr));
    }
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
                outimage.create(inimage.w * scale, inimage.h * scale, inimage.elempack);
            rc = rsr_process(ctx, inimage.data(), inimage.w, inimage.h, inimage.elempack, outimage.data());
        }
        if (rc != RSR_OK) std::fprintf(stderr, "RealSR::process: %s\n", rsr_last_error(ctx));
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
