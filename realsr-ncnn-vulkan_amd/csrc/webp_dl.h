// webp_dl.h -- webp decode / lossless encode through the system's libwebp, bound at RUN time.
//
// The reference links libwebp (git submodule src/libwebp, empty in the checkout) and calls WebPGetFeatures + WebPDecode into
// RGB(A) and WebPEncodeLosslessRGB(A) (/root/reference/src/webp_image.h:10-48, 50-98).  This toolchain ships the library's
// runtime (libwebp.so.7) but not its headers, so the handful of simple-API entry points used here are declared below from
// libwebp's public, ABI-stable interface (webp/decode.h, webp/encode.h) and resolved with dlopen -- the same way
// group.cpp binds RCCL.  WebPDecodeRGB(A) runs the same decoder with the same default options (fancy upsampling, no
// dithering) as the reference's WebPDecode call, so pixels are those the reference would hand to RealSR::process.
// No libwebp at run time: webp files are refused with a message (a failed decode / encode in the reference's terms).
#pragma once
#include <dlfcn.h>

#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <string>

namespace webpdl {

struct BitstreamFeatures // WebPBitstreamFeatures, decoder ABI 0x02xx
{
    int width, height, has_alpha, has_animation, format;
    uint32_t pad[5];
};

struct Api
{
    void* so = nullptr;
    int (*get_features)(const uint8_t*, size_t, BitstreamFeatures*, int) = nullptr; // WebPGetFeaturesInternal; 0 = VP8_STATUS_OK
    uint8_t* (*decode_rgb)(const uint8_t*, size_t, int*, int*) = nullptr;
    uint8_t* (*decode_rgba)(const uint8_t*, size_t, int*, int*) = nullptr;
    size_t (*encode_rgb)(const uint8_t*, int, int, int, uint8_t**) = nullptr; // WebPEncodeLosslessRGB(rgb, w, h, stride, &out)
    size_t (*encode_rgba)(const uint8_t*, int, int, int, uint8_t**) = nullptr;
    void (*free_)(void*) = nullptr;
    std::string error;
};

inline const Api& api()
{
    static Api a;
    static std::once_flag once;
    std::call_once(once, []() {
        const char* env = std::getenv("RSR_LIBWEBP"); // when set, the only candidate
        const char* names[] = {"libwebp.so.7", "libwebp.so", "libwebp.so.6"};
        if (env && *env) a.so = dlopen(env, RTLD_NOW | RTLD_LOCAL);
        else
            for (const char* n : names)
                if ((a.so = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!a.so)
        {
            a.error = "webp support needs libwebp.so at run time (not found; RSR_LIBWEBP=<path> names it explicitly)";
            return;
        }
        auto sym = [&](const char* n) { return dlsym(a.so, n); };
        a.get_features = reinterpret_cast<decltype(a.get_features)>(sym("WebPGetFeaturesInternal"));
        a.decode_rgb = reinterpret_cast<decltype(a.decode_rgb)>(sym("WebPDecodeRGB"));
        a.decode_rgba = reinterpret_cast<decltype(a.decode_rgba)>(sym("WebPDecodeRGBA"));
        a.encode_rgb = reinterpret_cast<decltype(a.encode_rgb)>(sym("WebPEncodeLosslessRGB"));
        a.encode_rgba = reinterpret_cast<decltype(a.encode_rgba)>(sym("WebPEncodeLosslessRGBA"));
        a.free_ = reinterpret_cast<decltype(a.free_)>(sym("WebPFree"));
        if (!a.free_) a.free_ = &std::free; // libwebp < 0.5 handed out plain malloc memory
        if (!a.get_features || !a.decode_rgb || !a.decode_rgba || !a.encode_rgb || !a.encode_rgba)
            a.error = "the libwebp found at run time lacks the simple decode / lossless encode API";
    });
    return a;
}

// RGB (c = 3) or RGBA (c = 4, when the bitstream says it has alpha) like webp_image.h:21-35; pixels are malloc-compatible
// memory to release with release().  Returns nullptr + message on failure.
inline uint8_t* decode(const uint8_t* data, size_t len, int* w, int* h, int* c, std::string& err)
{
    const Api& a = api();
    if (!a.error.empty()) { err = a.error; return nullptr; }
    BitstreamFeatures f{};
    if (a.get_features(data, len, &f, 0x0200) != 0) { err = "not a decodable webp bitstream"; return nullptr; }
    if (f.has_animation) { err = "animated webp is not supported (WebPDecode refuses it too)"; return nullptr; }
    *c = f.has_alpha ? 4 : 3;
    uint8_t* px = (f.has_alpha ? a.decode_rgba : a.decode_rgb)(data, len, w, h);
    if (!px) err = "webp decode failed";
    return px;
}

inline void release(void* p)
{
    if (p) api().free_(p);
}

// lossless, like webp_image.h:66-85; returns the encoded size (0 + message on failure), *out to release()
inline size_t encode_lossless(const uint8_t* px, int w, int h, int c, uint8_t** out, std::string& err)
{
    const Api& a = api();
    if (!a.error.empty()) { err = a.error; return 0; }
    if (c != 3 && c != 4) { err = "webp needs an RGB or RGBA image"; return 0; }
    if (w > 16383 || h > 16383) { err = "webp holds at most 16383 x 16383 pixels"; return 0; }
    const size_t n = (c == 3 ? a.encode_rgb : a.encode_rgba)(px, w, h, w * c, out);
    if (!n) err = "webp encoder failed";
    return n;
}

} // namespace webpdl
