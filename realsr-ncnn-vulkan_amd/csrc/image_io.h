// image_io.h -- image codecs of the CLI (CPU codecs are outside the accelerated path).
// Decode: like the reference (main.cpp:208-267) through the vendored stb_image (jpg / png incl. 16-bit, palette and tRNS /
// bmp / tga / gif / psd), plus binary PNM (P5/P6) for raw test data; gray -> RGB and gray+alpha -> RGBA (main.cpp:247-260), so
// the engine only sees c in {3,4}.  Encode: png (own zlib writer: level-2 deflate is ~5x faster than stb's encoder and the
// save stage is the throughput limit of a directory run) and jpg at quality 100 through stb_image_write (main.cpp:396-404).
// webp (decode, lossless encode: webp_image.h) goes through the system's libwebp, bound at run time (webp_dl.h).
#pragma once
#include <zlib.h>

#include <cstdint>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "realsr.h"
#include "webp_dl.h"

#if defined(__GNUC__)
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wsign-compare"
#pragma GCC diagnostic ignored "-Wunused-but-set-variable"
#pragma GCC diagnostic ignored "-Wunused-function"
#endif
#define STB_IMAGE_IMPLEMENTATION
#define STB_IMAGE_STATIC
#define STBI_NO_HDR
#define STBI_NO_LINEAR
#include "../../third_party/stb/stb_image.h"
#define STB_IMAGE_WRITE_IMPLEMENTATION
#define STB_IMAGE_WRITE_STATIC
#include "../../third_party/stb/stb_image_write.h"
#if defined(__GNUC__)
#pragma GCC diagnostic pop
#endif

namespace imgio {

inline bool read_file(const std::string& path, std::vector<uint8_t>& buf)
{
    FILE* fp = std::fopen(path.c_str(), "rb");
    if (!fp) return false;
    std::fseek(fp, 0, SEEK_END);
    const long n = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    buf.resize(n > 0 ? size_t(n) : 0);
    const bool ok = n >= 0 && std::fread(buf.data(), 1, buf.size(), fp) == buf.size();
    std::fclose(fp);
    return ok;
}

inline std::string decode_pnm(const std::vector<uint8_t>& f, Image& out)
{
    if (f.size() < 7 || f[0] != 'P' || (f[1] != '5' && f[1] != '6')) return "not a binary PNM file";
    size_t pos = 2;
    int vals[3], n = 0;
    while (n < 3 && pos < f.size())
    {
        while (pos < f.size() && (f[pos] == ' ' || f[pos] == '\n' || f[pos] == '\r' || f[pos] == '\t')) pos++;
        if (pos < f.size() && f[pos] == '#') { while (pos < f.size() && f[pos] != '\n') pos++; continue; }
        int v = 0;
        bool any = false;
        while (pos < f.size() && f[pos] >= '0' && f[pos] <= '9') { v = v * 10 + (f[pos++] - '0'); any = true; }
        if (!any) return "bad PNM header";
        vals[n++] = v;
    }
    pos++; // single whitespace after maxval
    const int w = vals[0], h = vals[1], ch = f[1] == '6' ? 3 : 1;
    if (n < 3 || vals[2] != 255 || w < 1 || h < 1 || pos + size_t(w) * h * ch > f.size()) return "unsupported or truncated PNM";
    out.create(w, h, 3);
    if (out.empty()) return "out of memory";
    for (size_t i = 0; i < size_t(w) * h; i++)
        for (int q = 0; q < 3; q++) out.data()[i * 3 + q] = f[pos + i * ch + (ch == 3 ? q : 0)];
    return "";
}

inline std::string lower_ext(const std::string& path)
{
    const size_t dot = path.find_last_of('.');
    std::string e = dot == std::string::npos ? "" : path.substr(dot + 1);
    for (char& c : e) c = char(std::tolower((unsigned char)c));
    return e;
}

inline bool is_webp(const std::vector<uint8_t>& f) { return f.size() >= 12 && !std::memcmp(f.data(), "RIFF", 4) && !std::memcmp(f.data() + 8, "WEBP", 4); }

// returns "" on success, else an error message (the caller prints "decode image %s failed", main.cpp:292-299)
inline std::string load_image(const std::string& path, Image& out)
{
    std::vector<uint8_t> f;
    if (!read_file(path, f)) return "cannot read file";
    if (is_webp(f)) // the reference tries webp first and falls through to stb for everything else (main.cpp:231-242)
    {
        int w = 0, h = 0, c = 0;
        std::string err;
        uint8_t* px = webpdl::decode(f.data(), f.size(), &w, &h, &c, err);
        if (!px) return err;
        out.create(w, h, c);
        if (!out.empty()) std::memcpy(out.data(), px, size_t(w) * h * c);
        webpdl::release(px);
        return out.empty() ? "out of memory" : "";
    }
    if (f.size() >= 2 && f[0] == 'P' && (f[1] == '5' || f[1] == '6')) return decode_pnm(f, out);
    int w = 0, h = 0, c = 0;
    if (f.size() > size_t(0x7fffffff)) return "file too large";
    // stb_image with automatic channel count first (a tRNS colour key turns RGB into RGBA there), then the reference's
    // expansion rule: gray -> RGB, gray+alpha -> RGBA (main.cpp:239-260)
    unsigned char* px = stbi_load_from_memory(f.data(), int(f.size()), &w, &h, &c, 0);
    if (!px) return std::string("decode failed: ") + stbi_failure_reason();
    int want = c;
    if (c == 1 || c == 2)
    {
        want = c == 1 ? 3 : 4;
        stbi_image_free(px);
        px = stbi_load_from_memory(f.data(), int(f.size()), &w, &h, &c, want);
        if (!px) return std::string("decode failed: ") + stbi_failure_reason();
    }
    if (want != 3 && want != 4)
    {
        stbi_image_free(px);
        return "unsupported channel count";
    }
    out.create(w, h, want);
    if (!out.empty()) std::memcpy(out.data(), px, size_t(w) * h * want);
    stbi_image_free(px);
    return out.empty() ? "out of memory" : "";
}

inline void put_chunk(std::vector<uint8_t>& o, const char* type, const uint8_t* d, size_t len)
{
    auto be = [&](uint32_t v) { o.push_back(uint8_t(v >> 24)); o.push_back(uint8_t(v >> 16)); o.push_back(uint8_t(v >> 8)); o.push_back(uint8_t(v)); };
    be(uint32_t(len));
    const size_t start = o.size();
    o.insert(o.end(), type, type + 4);
    if (len) o.insert(o.end(), d, d + len);
    uLong crc = crc32(0L, &o[start], 4); // zlib's length argument is 32 bits: feed long chunks in pieces
    for (size_t off = 0; off < len;)
    {
        const size_t n = std::min(len - off, size_t(1) << 30);
        crc = crc32(crc, &o[start + 4 + off], uInt(n));
        off += n;
    }
    be(uint32_t(crc));
}

inline std::string save_png(const std::string& path, const Image& im, int level = 2)
{
    const size_t stride = size_t(im.w) * im.elempack;
    std::vector<uint8_t> raw((stride + 1) * im.h);
    for (int y = 0; y < im.h; y++) // filter 1 (Sub) compresses upscaled images well and costs nothing to compute
    {
        uint8_t* d = &raw[size_t(y) * (stride + 1)];
        const uint8_t* s = im.data() + size_t(y) * stride;
        d[0] = 1;
        for (size_t i = 0; i < stride; i++) d[i + 1] = uint8_t(s[i] - (i >= size_t(im.elempack) ? s[i - im.elempack] : 0));
    }
    uLongf clen = compressBound(uLong(raw.size()));
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), uLong(raw.size()), level) != Z_OK) return "deflate failed";
    std::vector<uint8_t> o = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    uint8_t ihdr[13] = {uint8_t(im.w >> 24), uint8_t(im.w >> 16), uint8_t(im.w >> 8), uint8_t(im.w), uint8_t(im.h >> 24), uint8_t(im.h >> 16),
                        uint8_t(im.h >> 8), uint8_t(im.h), 8, uint8_t(im.elempack == 4 ? 6 : 2), 0, 0, 0};
    put_chunk(o, "IHDR", ihdr, 13);
    // a chunk length is a 31-bit field (PNG 5.3): compressed data beyond that goes into further IDAT chunks
    const size_t kMaxChunk = size_t(0x7fffffff);
    for (size_t off = 0; off < size_t(clen) || off == 0; off += kMaxChunk)
    {
        put_chunk(o, "IDAT", comp.data() + off, std::min(kMaxChunk, size_t(clen) - off));
        if (size_t(clen) == 0) break;
    }
    put_chunk(o, "IEND", nullptr, 0);
    FILE* fp = std::fopen(path.c_str(), "wb");
    if (!fp) return "cannot open output file";
    const bool ok = std::fwrite(o.data(), 1, o.size(), fp) == o.size();
    std::fclose(fp);
    return ok ? "" : "short write";
}

// jpg at quality 100 (main.cpp:400).  An RGBA image cannot be a jpg: the caller renames the output to .png before it gets here
// (main.cpp:278-288).
inline std::string save_jpg(const std::string& path, const Image& im)
{
    if (im.elempack != 3) return "jpg needs an RGB image";
    return stbi_write_jpg(path.c_str(), im.w, im.h, 3, im.data(), 100) ? "" : "jpg encoder failed";
}

// lossless webp (webp_image.h:50-98)
inline std::string save_webp(const std::string& path, const Image& im)
{
    uint8_t* enc = nullptr;
    std::string err;
    const size_t n = webpdl::encode_lossless(im.data(), im.w, im.h, im.elempack, &enc, err);
    if (!n) return err;
    FILE* fp = std::fopen(path.c_str(), "wb");
    bool ok = fp != nullptr;
    if (fp)
    {
        ok = std::fwrite(enc, 1, n, fp) == n;
        std::fclose(fp);
    }
    webpdl::release(enc);
    return ok ? "" : "cannot write output file";
}

inline std::string save_image(const std::string& path, const Image& im)
{
    const std::string ext = lower_ext(path);
    if (ext == "jpg" || ext == "jpeg") return save_jpg(path, im);
    if (ext == "png") return save_png(path, im);
    if (ext == "webp") return save_webp(path, im);
    return "no encoder for ." + ext;
}

} // namespace imgio
