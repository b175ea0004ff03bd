// image_io.h -- minimal image codecs for the CLI (CPU codecs are outside the accelerated path).
// PNG (8/16-bit, gray / gray+alpha / RGB / RGBA / palette, non-interlaced) on zlib, and binary PNM (P5/P6).
// The reference decodes jpg/png/webp through stb_image + libwebp (main.cpp:208-267); neither libjpeg/libwebp
// headers nor stb are available to this build, so jpg/webp are reported as unsupported instead of silently skipped.
// Channel policy follows main.cpp:247-260: gray -> RGB, gray+alpha -> RGBA, so the engine only sees c in {3,4}.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "realsr.h"

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

inline uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

inline int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// returns "" on success, else an error message
inline std::string decode_png(const std::vector<uint8_t>& f, Image& out)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (f.size() < 8 + 25 || std::memcmp(f.data(), sig, 8) != 0) return "not a PNG file";
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    size_t pos = 8;
    bool end = false;
    while (!end && pos + 12 <= f.size())
    {
        const uint32_t len = be32(&f[pos]);
        const char* type = reinterpret_cast<const char*>(&f[pos + 4]);
        if (pos + 12 + len > f.size()) return "truncated PNG chunk";
        const uint8_t* d = &f[pos + 8];
        if (!std::memcmp(type, "IHDR", 4))
        {
            if (len < 13) return "bad IHDR";
            w = be32(d);
            h = be32(d + 4);
            depth = d[8];
            ctype = d[9];
            interlace = d[12];
        }
        else if (!std::memcmp(type, "PLTE", 4)) plte.assign(d, d + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(d, d + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!std::memcmp(type, "IEND", 4)) end = true;
        pos += 12 + len;
    }
    if (!w || !h || w > 65535 || h > 65535) return "bad PNG dimensions";
    if (interlace) return "interlaced PNG is not supported";
    if (!(depth == 8 || depth == 16 || (ctype == 3 && (depth == 1 || depth == 2 || depth == 4)) || (ctype == 0 && depth < 8)))
        return "unsupported PNG bit depth";
    const int nch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!nch) return "unsupported PNG colour type";
    const size_t bpp_bits = size_t(nch) * depth, stride = (size_t(w) * bpp_bits + 7) / 8, bpp = (bpp_bits + 7) / 8;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf rawlen = uLongf(raw.size());
    if (uncompress(raw.data(), &rawlen, idat.data(), uLong(idat.size())) != Z_OK || rawlen != raw.size()) return "PNG inflate failed";
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; y++)
    {
        const uint8_t ft = raw[y * (stride + 1)];
        const uint8_t* src = &raw[y * (stride + 1) + 1];
        uint8_t* cur = &img[y * stride];
        const uint8_t* up = y ? &img[(y - 1) * stride] : nullptr;
        for (size_t i = 0; i < stride; i++)
        {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int v = src[i];
            switch (ft)
            {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: return "bad PNG filter type";
            }
            cur[i] = uint8_t(v);
        }
    }
    const bool has_alpha = ctype == 4 || ctype == 6 || (ctype == 3 && !trns.empty());
    const int oc = has_alpha ? 4 : 3;
    out.create(int(w), int(h), oc);
    auto sample = [&](const uint8_t* row, size_t idx) -> int { // idx-th sample of the row, scaled to 8 bit
        if (depth == 8) return row[idx];
        if (depth == 16) return row[idx * 2];
        const int per = 8 / depth, sh = (per - 1 - int(idx % per)) * depth, v = (row[idx / per] >> sh) & ((1 << depth) - 1);
        return ctype == 3 ? v : v * 255 / ((1 << depth) - 1);
    };
    for (uint32_t y = 0; y < h; y++)
    {
        const uint8_t* row = &img[y * stride];
        uint8_t* o = &out.pixels[size_t(y) * w * oc];
        for (uint32_t x = 0; x < w; x++, o += oc)
        {
            int r, g, b, a = 255;
            if (ctype == 0) r = g = b = sample(row, x);
            else if (ctype == 4) { r = g = b = sample(row, x * 2); a = sample(row, x * 2 + 1); }
            else if (ctype == 2) { r = sample(row, x * 3); g = sample(row, x * 3 + 1); b = sample(row, x * 3 + 2); }
            else if (ctype == 6) { r = sample(row, x * 4); g = sample(row, x * 4 + 1); b = sample(row, x * 4 + 2); a = sample(row, x * 4 + 3); }
            else
            {
                const size_t pi = size_t(sample(row, x));
                if (pi * 3 + 2 >= plte.size()) return "PNG palette index out of range";
                r = plte[pi * 3]; g = plte[pi * 3 + 1]; b = plte[pi * 3 + 2];
                a = pi < trns.size() ? trns[pi] : 255;
            }
            o[0] = uint8_t(r); o[1] = uint8_t(g); o[2] = uint8_t(b);
            if (oc == 4) o[3] = uint8_t(a);
        }
    }
    return "";
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
    for (size_t i = 0; i < size_t(w) * h; i++)
        for (int q = 0; q < 3; q++) out.pixels[i * 3 + q] = f[pos + i * ch + (ch == 3 ? q : 0)];
    return "";
}

inline std::string lower_ext(const std::string& path)
{
    const size_t dot = path.find_last_of('.');
    std::string e = dot == std::string::npos ? "" : path.substr(dot + 1);
    for (char& c : e) c = char(std::tolower((unsigned char)c));
    return e;
}

inline std::string load_image(const std::string& path, Image& out)
{
    std::vector<uint8_t> f;
    if (!read_file(path, f)) return "cannot read file";
    if (f.size() >= 8 && f[0] == 0x89 && f[1] == 'P') return decode_png(f, out);
    if (f.size() >= 2 && f[0] == 'P' && (f[1] == '5' || f[1] == '6')) return decode_pnm(f, out);
    if (f.size() >= 3 && f[0] == 0xff && f[1] == 0xd8) return "jpeg decoding is not built in (no libjpeg/stb in this toolchain)";
    if (f.size() >= 12 && !std::memcmp(&f[0], "RIFF", 4) && !std::memcmp(&f[8], "WEBP", 4)) return "webp decoding is not built in (no libwebp headers)";
    return "unknown image format";
}

inline void put_chunk(std::vector<uint8_t>& o, const char* type, const uint8_t* d, size_t len)
{
    auto be = [&](uint32_t v) { o.push_back(uint8_t(v >> 24)); o.push_back(uint8_t(v >> 16)); o.push_back(uint8_t(v >> 8)); o.push_back(uint8_t(v)); };
    be(uint32_t(len));
    const size_t start = o.size();
    o.insert(o.end(), type, type + 4);
    if (len) o.insert(o.end(), d, d + len);
    be(uint32_t(crc32(0L, &o[start], uInt(4 + len))));
}

inline std::string save_png(const std::string& path, const Image& im, int level = 2)
{
    const size_t stride = size_t(im.w) * im.elempack;
    std::vector<uint8_t> raw((stride + 1) * im.h);
    for (int y = 0; y < im.h; y++) // filter 1 (Sub) compresses upscaled images well and costs nothing to compute
    {
        uint8_t* d = &raw[size_t(y) * (stride + 1)];
        const uint8_t* s = &im.pixels[size_t(y) * stride];
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
    put_chunk(o, "IDAT", comp.data(), clen);
    put_chunk(o, "IEND", nullptr, 0);
    FILE* fp = std::fopen(path.c_str(), "wb");
    if (!fp) return "cannot open output file";
    const bool ok = std::fwrite(o.data(), 1, o.size(), fp) == o.size();
    std::fclose(fp);
    return ok ? "" : "short write";
}

} // namespace imgio
