// model.cpp -- see model.h.  Host-only C++ (no HIP).
#include "model.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

#include "../../include/realsr_hip.h"

namespace rsr {

// ------------------------------------------------------------------------------------------
// fp16 helpers (round-to-nearest-even), used when packing weights for the MFMA kernels
// ------------------------------------------------------------------------------------------
uint16_t f32_to_f16(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t mant = x & 0x7fffffu;
    const int exp8 = int((x >> 23) & 0xff);
    if (exp8 == 0xff) return uint16_t(sign | 0x7c00u | (mant ? (0x200u | (mant >> 13)) : 0u));
    const int e = exp8 - 112; // re-biased
    if (e >= 31) return uint16_t(sign | 0x7c00u);
    if (e <= 0)
    {
        if (e < -10) return uint16_t(sign);
        mant |= 0x800000u;
        const int shift = 14 - e;
        uint32_t r = mant >> shift;
        const uint32_t rem = mant & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (r & 1u))) ++r;
        return uint16_t(sign | r);
    }
    uint32_t r = (uint32_t(e) << 10) | (mant >> 13);
    const uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
    return uint16_t(sign | r);
}

float f16_to_f32(uint16_t h)
{
    const uint32_t sign = (uint32_t(h) & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    uint32_t x;
    if (e == 0)
    {
        if (m == 0) x = sign;
        else
        {
            int k = 0;
            while (!(m & 0x400u)) { m <<= 1; ++k; }
            x = sign | (uint32_t(113 - k) << 23) | ((m & 0x3ffu) << 13);
        }
    }
    else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}

// ------------------------------------------------------------------------------------------
// .param (text): "7767517" / "layer_count blob_count" / one layer per line:
//   type name nbottom ntop bottoms... tops... key=value...      
// ------------------------------------------------------------------------------------------
int ParamLayer::geti(int id, int def) const
{
    for (auto& kv : scalars)
        if (kv.first == id) return std::atoi(kv.second.c_str());
    return def;
}
float ParamLayer::getf(int id, float def) const
{
    for (auto& kv : scalars)
        if (kv.first == id) return float(std::atof(kv.second.c_str()));
    return def;
}
const std::vector<float>* ParamLayer::geta(int id) const
{
    for (auto& kv : arrays)
        if (kv.first == id) return &kv.second;
    return nullptr;
}

int parse_param(const std::string& path, Model& m, std::string& err)
{
    std::ifstream f(path);
    if (!f)
    {
        err = "cannot open param file " + path;
        return RSR_E_IO;
    }
    std::string line;
    if (!std::getline(f, line) || std::atoi(line.c_str()) != 7767517)
    {
        err = "param magic mismatch (expected 7767517)";
        return RSR_E_FORMAT;
    }
    if (!std::getline(f, line))
    {
        err = "param truncated";
        return RSR_E_FORMAT;
    }
    {
        std::istringstream ss(line);
        if (!(ss >> m.n_layers >> m.n_blobs) || m.n_layers <= 0 || m.n_blobs <= 0)
        {
            err = "bad layer/blob counts";
            return RSR_E_FORMAT;
        }
    }
    m.layers.clear();
    while (std::getline(f, line))
    {
        std::istringstream ss(line);
        ParamLayer L;
        int nb = 0, nt = 0;
        if (!(ss >> L.type)) continue; // blank line
        if (!(ss >> L.name >> nb >> nt) || nb < 0 || nt < 0 || nb > 64 || nt > 64)
        {
            err = "malformed layer line: " + line.substr(0, 60);
            return RSR_E_FORMAT;
        }
        std::string tok;
        for (int i = 0; i < nb; i++)
        {
            if (!(ss >> tok)) { err = "layer " + L.name + ": missing bottom"; return RSR_E_FORMAT; }
            L.bottoms.push_back(tok);
        }
        for (int i = 0; i < nt; i++)
        {
            if (!(ss >> tok)) { err = "layer " + L.name + ": missing top"; return RSR_E_FORMAT; }
            L.tops.push_back(tok);
        }
        while (ss >> tok)
        {
            const size_t eq = tok.find('=');
            if (eq == std::string::npos) { err = "layer " + L.name + ": bad param token " + tok; return RSR_E_FORMAT; }
            const int key = std::atoi(tok.substr(0, eq).c_str());
            const std::string val = tok.substr(eq + 1);
            if (key <= -23300)
            {
                std::vector<float> arr;
                std::istringstream vs(val);
                std::string item;
                bool first = true;
                size_t count = 0;
                while (std::getline(vs, item, ','))
                {
                    if (first) { count = size_t(std::atoi(item.c_str())); first = false; }
                    else arr.push_back(float(std::atof(item.c_str())));
                }
                if (arr.size() != count) { err = "layer " + L.name + ": array param length mismatch"; return RSR_E_FORMAT; }
                L.arrays.emplace_back(-23300 - key, arr);
            }
            else
                L.scalars.emplace_back(key, val);
        }
        m.layers.push_back(std::move(L));
    }
    if (int(m.layers.size()) != m.n_layers)
    {
        err = "layer count does not match header";
        return RSR_E_FORMAT;
    }
    return RSR_OK;
}

// ------------------------------------------------------------------------------------------
// Graph validation: the engine runs a FUSED, hard-wired RRDBNet schedule, so it must refuse any
// .param that is not that network.  Both the parsed graph and a canonical construction are
// flattened to "value graphs" (Split resolved to aliases, blob names abstracted away) and compared
// node by node: type, parameters, and producer indices of every input.
// ------------------------------------------------------------------------------------------
namespace {
struct VNode
{
    std::string type, sig;
    std::vector<int> ins;
    bool operator==(const VNode& o) const { return type == o.type && sig == o.sig && ins == o.ins; }
};

std::string fnum(float v)
{
    char b[32];
    std::snprintf(b, sizeof b, "%.6g", double(v));
    return b;
}

int flatten(const Model& m, std::vector<VNode>& out, std::string& err)
{
    std::map<std::string, int> val;
    for (const ParamLayer& L : m.layers)
    {
        std::vector<int> ins;
        for (auto& b : L.bottoms)
        {
            auto it = val.find(b);
            if (it == val.end()) { err = "layer " + L.name + " consumes undefined blob " + b; return RSR_E_GRAPH; }
            ins.push_back(it->second);
        }
        if (L.type == "Split")
        {
            if (ins.size() != 1) { err = "Split with != 1 input"; return RSR_E_GRAPH; }
            for (auto& t : L.tops) val[t] = ins[0];
            continue;
        }
        if (L.tops.size() != 1) { err = "layer " + L.name + ": expected one output"; return RSR_E_GRAPH; }
        VNode n;
        n.type = L.type;
        n.ins = ins;
        std::ostringstream s;
        if (L.type == "Input")
        {
            if (L.tops[0] != "data") { err = "input blob must be named 'data' (realsr.cpp:426)"; return RSR_E_GRAPH; }
        }
        else if (L.type == "Convolution")
        {
            const std::vector<float>* ap = L.geta(10);
            s << L.geti(0, 0) << ',' << L.geti(1, 0) << ',' << L.geti(11, L.geti(1, 0)) << ',' << L.geti(2, 1) << ','
              << L.geti(3, 1) << ',' << L.geti(4, 0) << ',' << L.geti(5, 0) << ',' << L.geti(6, 0) << ',' << L.geti(9, 0)
              << ',' << (ap && !ap->empty() ? fnum((*ap)[0]) : "0");
        }
        else if (L.type == "Eltwise")
        {
            s << L.geti(0, 0);
            if (const std::vector<float>* c = L.geta(1))
                for (float v : *c) s << ',' << fnum(v);
        }
        else if (L.type == "BinaryOp")
            s << L.geti(0, 0) << ',' << L.geti(1, 0);
        else if (L.type == "Interp")
            s << L.geti(0, 0) << ',' << fnum(L.getf(1, 1.f)) << ',' << fnum(L.getf(2, 1.f));
        else if (L.type == "Concat")
            s << L.geti(0, 0);
        else
        {
            err = "unsupported layer type " + L.type;
            return RSR_E_GRAPH;
        }
        n.sig = s.str();
        val[L.tops[0]] = int(out.size());
        out.push_back(std::move(n));
    }
    auto it = val.find("output");
    if (it == val.end() || it->second != int(out.size()) - 1)
    {
        err = "last layer must produce blob 'output' (realsr.cpp:428)";
        return RSR_E_GRAPH;
    }
    return RSR_OK;
}

void canonical(std::vector<VNode>& g)
{
    auto add = [&](const std::string& type, const std::string& sig, std::vector<int> ins) {
        g.push_back(VNode{type, sig, std::move(ins)});
        return int(g.size()) - 1;
    };
    auto conv = [&](int src, int cin, int cout, bool lrelu) {
        std::ostringstream s;
        s << cout << ",3,3,1,1,1,1," << cin * cout * 9 << ',' << (lrelu ? 2 : 0) << ',' << (lrelu ? "0.2" : "0");
        return add("Convolution", s.str(), {src});
    };
    auto axpy = [&](int a, int b) { return add("Eltwise", "1,0.2,1", {a, b}); };
    const int data = add("Input", "", {});
    const int fea = conv(data, 3, kNF, false);
    int cur = fea;
    for (int i = 0; i < kNumRRDB; i++)
    {
        const int rin = cur;
        for (int j = 0; j < 3; j++)
        {
            const int x = cur;
            std::vector<int> feats{x};
            for (int k = 0; k < 4; k++)
            {
                const int src = (k == 0) ? x : add("Concat", "0", feats);
                feats.push_back(conv(src, kNF + k * kGC, kGC, true));
            }
            const int x5 = conv(add("Concat", "0", feats), kNF + 4 * kGC, kNF, false);
            cur = axpy(x5, x);
        }
        cur = axpy(cur, rin);
    }
    const int trunk = conv(cur, kNF, kNF, false);
    int s = add("BinaryOp", "0,0", {fea, trunk});
    for (int i = 0; i < 2; i++)
    {
        const int up = add("Interp", "1,2,2", {s});
        s = conv(up, kNF, kNF, true);
    }
    s = conv(s, kNF, kNF, true);
    conv(s, kNF, 3, false);
}
} // namespace

int validate_graph(const Model& m, std::string& err)
{
    std::vector<VNode> got, want;
    int rc = flatten(m, got, err);
    if (rc != RSR_OK) return rc;
    canonical(want);
    if (got.size() != want.size())
    {
        err = "graph is not RRDBNet(3,3,64,23,32): " + std::to_string(got.size()) + " compute nodes, expected " +
              std::to_string(want.size());
        return RSR_E_GRAPH;
    }
    for (size_t i = 0; i < got.size(); i++)
        if (!(got[i] == want[i]))
        {
            err = "graph differs from RRDBNet(3,3,64,23,32) at compute node " + std::to_string(i) + " (" + got[i].type +
                  " " + got[i].sig + " vs " + want[i].type + " " + want[i].sig + ")";
            return RSR_E_GRAPH;
        }
    return RSR_OK;
}

// ------------------------------------------------------------------------------------------
// .bin: per Convolution in .param order: weight blob ("type 0": 4-byte tag + payload) then bias
// blob ("type 1": raw fp32)                                         
// ------------------------------------------------------------------------------------------
namespace {
bool read_exact(std::FILE* fp, void* dst, size_t n) { return std::fread(dst, 1, n, fp) == n; }
} // namespace

int read_bin(const std::string& path, Model& m, std::string& err)
{
    std::FILE* fp = std::fopen(path.c_str(), "rb");
    if (!fp)
    {
        err = "cannot open model file " + path;
        return RSR_E_IO;
    }
    m.convs.clear();
    m.n_weights = m.n_biases = 0;
    int enc_all = -1;
    int rc = RSR_OK;
    for (const ParamLayer& L : m.layers)
    {
        if (L.type != "Convolution") continue;
        ConvRec c;
        c.cout = L.geti(0, 0);
        const int wsize = L.geti(6, 0);
        const int k = L.geti(1, 0);
        if (c.cout <= 0 || k != 3 || wsize <= 0 || wsize % (c.cout * 9) != 0 || L.geti(5, 0) != 1)
        {
            err = "layer " + L.name + ": not a 3x3 biased convolution";
            rc = RSR_E_GRAPH;
            break;
        }
        c.cin = wsize / (c.cout * 9);
        c.act = L.geti(9, 0);
        if (const std::vector<float>* ap = L.geta(10)) c.slope = ap->empty() ? 0.f : (*ap)[0];
        c.weight.resize(size_t(wsize));
        c.bias.resize(size_t(c.cout));
        uint32_t tag = 0;
        if (!read_exact(fp, &tag, 4)) { err = "model file truncated at " + L.name; rc = RSR_E_IO; break; }
        int enc;
        if (tag == 0x01306B47u)
        {
            const size_t nbytes = (size_t(wsize) * 2 + 3) & ~size_t(3);
            std::vector<uint16_t> tmp(nbytes / 2);
            if (!read_exact(fp, tmp.data(), nbytes)) { err = "model file truncated at " + L.name; rc = RSR_E_IO; break; }
            for (int i = 0; i < wsize; i++) c.weight[size_t(i)] = f16_to_f32(tmp[size_t(i)]);
            enc = 1;
        }
        else if (tag == 0u || tag == 0x0002C056u)
        {
            if (!read_exact(fp, c.weight.data(), size_t(wsize) * 4)) { err = "model file truncated at " + L.name; rc = RSR_E_IO; break; }
            enc = 0;
        }
        else if (tag == 0x000D4B38u)
        {
            err = "int8 weight blobs are not supported (layer " + L.name + ")";
            rc = RSR_E_FORMAT;
            break;
        }
        else
        {
            float table[256];
            const size_t nbytes = (size_t(wsize) + 3) & ~size_t(3);
            std::vector<uint8_t> idx(nbytes);
            if (!read_exact(fp, table, sizeof table) || !read_exact(fp, idx.data(), nbytes))
            {
                err = "model file truncated at " + L.name;
                rc = RSR_E_IO;
                break;
            }
            for (int i = 0; i < wsize; i++) c.weight[size_t(i)] = table[idx[size_t(i)]];
            enc = 2;
        }
        enc_all = (enc_all == -1) ? enc : (enc_all == enc ? enc : 2);
        if (!read_exact(fp, c.bias.data(), size_t(c.cout) * 4)) { err = "model file truncated at bias of " + L.name; rc = RSR_E_IO; break; }
        m.n_weights += wsize;
        m.n_biases += c.cout;
        m.convs.push_back(std::move(c));
    }
    if (rc == RSR_OK)
    {
        unsigned char extra;
        if (std::fread(&extra, 1, 1, fp) == 1)
        {
            err = "model file has trailing bytes after the last convolution";
            rc = RSR_E_FORMAT;
        }
    }
    std::fclose(fp);
    m.bin_encoding = enc_all;
    return rc;
}

int load_model(const std::string& param, const std::string& bin, Model& m, std::string& err)
{
    int rc = parse_param(param, m, err);
    if (rc != RSR_OK) return rc;
    rc = validate_graph(m, err);
    if (rc != RSR_OK) return rc;
    rc = read_bin(bin, m, err);
    if (rc != RSR_OK) return rc;
    if (int(m.convs.size()) != kNumConvs)
    {
        err = "expected 351 convolutions";
        return RSR_E_GRAPH;
    }
    return RSR_OK;
}

// ------------------------------------------------------------------------------------------
// packing
// ------------------------------------------------------------------------------------------
namespace {
size_t align256(size_t v) { return (v + 255) & ~size_t(255); }
} // namespace

size_t packed_size(const Model& m)
{
    size_t off = align256(sizeof(PackedHeader) + m.convs.size() * sizeof(PackedConv));
    for (const ConvRec& c : m.convs)
    {
        const size_t np = size_t((c.cin + 31) / 32), nt = size_t((c.cout + 31) / 32);
        off = align256(off + nt * 32 * 4);
        off = align256(off + 2 * np * 9 * nt * 32 * 32);
        if (c.cout <= 4) off = align256(off + 2 * np * 3 * 32 * 32);
    }
    return off;
}

int pack_model(const Model& m, void* dst, size_t cap, std::string& err)
{
    const size_t need = packed_size(m);
    if (cap < need)
    {
        err = "packed blob buffer too small";
        return RSR_E_ARG;
    }
    unsigned char* base = static_cast<unsigned char*>(dst);
    std::memset(base, 0, need);
    PackedHeader* H = reinterpret_cast<PackedHeader*>(base);
    H->magic = kPackedMagic;
    H->version = kPackedVersion;
    H->nconv = uint32_t(m.convs.size());
    H->flags = 0u;
    H->total_bytes = need;
    PackedConv* T = reinterpret_cast<PackedConv*>(base + sizeof(PackedHeader));
    size_t off = align256(sizeof(PackedHeader) + m.convs.size() * sizeof(PackedConv));
    for (size_t ci = 0; ci < m.convs.size(); ci++)
    {
        const ConvRec& c = m.convs[ci];
        const int np = (c.cin + 31) / 32, nt = (c.cout + 31) / 32;
        PackedConv& P = T[ci];
        P.cin = uint32_t(c.cin);
        P.cout = uint32_t(c.cout);
        P.act = uint32_t(c.act);
        P.nplanes = uint32_t(np);
        P.nt = uint32_t(nt);
        P.slope = c.slope;
        auto wt = [&](int n, int ic, int tap) -> uint16_t {
            float v = 0.f;
            if (n < c.cout && ic < c.cin) v = c.weight[(size_t(n) * c.cin + ic) * 9 + tap];
            return f32_to_f16(v);
        };
        const int rows = 9 * nt * 32;
        // image row n (and bias entry n) carries output channel co(n): see row_cout() in model.h
        auto co = [](int n) { return (n & ~31) + row_cout(n & 31); };
        P.b_off = off;
        float* B = reinterpret_cast<float*>(base + off);
        for (int n = 0; n < nt * 32; n++) B[n] = co(n) < c.cout ? c.bias[size_t(co(n))] : 0.f;
        off = align256(off + size_t(nt) * 32 * 4);
        P.w16_off = off;
        uint16_t* W16 = reinterpret_cast<uint16_t*>(base + off);
        for (int pl = 0; pl < 2 * np; pl++)
            for (int tap = 0; tap < 9; tap++)
                for (int n = 0; n < nt * 32; n++)
                {
                    uint16_t* R = W16 + (size_t(pl) * rows + size_t(tap) * nt * 32 + n) * 16;
                    for (int slot = 0; slot < 2; slot++)
                    {
                        const int pslot = slot ^ ((n >> 3) & 1);
                        for (int e = 0; e < 8; e++) R[pslot * 8 + e] = wt(co(n), pl * 16 + slot * 8 + e, tap);
                    }
                }
        off = align256(off + size_t(2 * np) * rows * 32);
        P.aux_off = 0;
        if (c.cout <= 4)
        { // (dy, cout) in the M dimension: [plane][dx][row = dy*8 + n][16 cin], same slot swizzle by row
            P.aux_off = off;
            uint16_t* A = reinterpret_cast<uint16_t*>(base + off);
            for (int pl = 0; pl < 2 * np; pl++)
                for (int dx = 0; dx < 3; dx++)
                    for (int dy = 0; dy < 3; dy++)
                        for (int n = 0; n < c.cout; n++)
                        {
                            const int row = dy * 8 + n;
                            uint16_t* R = A + (size_t(pl) * 96 + size_t(dx) * 32 + row) * 16;
                            for (int slot = 0; slot < 2; slot++)
                            {
                                const int pslot = slot ^ ((row >> 3) & 1);
                                for (int e = 0; e < 8; e++) R[pslot * 8 + e] = wt(n, pl * 16 + slot * 8 + e, dy * 3 + dx);
                            }
                        }
            off = align256(off + size_t(2 * np) * 96 * 32);
        }
    }
    return RSR_OK;
}

void canonical_conv(int index, int& cin, int& cout, int& act)
{
    // x4.param order: conv_first | 69 x {64->32, 96->32, 128->32, 160->32 (LeakyReLU), 192->64} | trunk_conv |
    // upconv1, upconv2, HRconv (LeakyReLU) | conv_last
    act = 0;
    if (index == 0) { cin = 3; cout = kNF; return; }
    if (index <= kNumRDB * 5)
    {
        const int k = (index - 1) % 5;
        cin = kNF + k * kGC;
        cout = k < 4 ? kGC : kNF;
        act = k < 4 ? 2 : 0;
        return;
    }
    const int t = index - 1 - kNumRDB * 5; // 0 trunk, 1 up1, 2 up2, 3 HR, 4 last
    cin = kNF;
    cout = t == 4 ? 3 : kNF;
    act = (t >= 1 && t <= 3) ? 2 : 0;
}

size_t packed_table_bytes() { return sizeof(PackedHeader) + size_t(kNumConvs) * sizeof(PackedConv); }

int check_packed(const void* head, size_t head_bytes, size_t total_bytes, std::string& err)
{
    if (!head || head_bytes < packed_table_bytes() || total_bytes < head_bytes)
    {
        err = "packed blob too small";
        return RSR_E_FORMAT;
    }
    const PackedHeader* H = static_cast<const PackedHeader*>(head);
    if (H->magic != kPackedMagic || H->version != kPackedVersion || H->nconv != uint32_t(kNumConvs) || H->total_bytes != total_bytes || H->flags != 0u)
    {
        err = "packed blob header mismatch (magic / version / conv count / size)";
        return RSR_E_FORMAT;
    }
    const PackedConv* T = reinterpret_cast<const PackedConv*>(static_cast<const unsigned char*>(head) + sizeof(PackedHeader));
    for (int i = 0; i < kNumConvs; i++)
    {
        const PackedConv& c = T[i];
        int cin, cout, act;
        canonical_conv(i, cin, cout, act);
        const uint64_t np = uint64_t((cin + 31) / 32), nt = uint64_t((cout + 31) / 32);
        bool ok = int(c.cin) == cin && int(c.cout) == cout && int(c.act) == act && c.nplanes == np && c.nt == nt;
        auto inside = [&](uint64_t off, uint64_t size) { return off >= packed_table_bytes() && (off & 255) == 0 && off <= total_bytes && size <= total_bytes - off; };
        ok = ok && inside(c.b_off, nt * 32 * 4) && inside(c.w16_off, 2 * np * 9 * nt * 32 * 32);
        ok = ok && (cout <= 4 ? inside(c.aux_off, 2 * np * 3 * 32 * 32) : c.aux_off == 0);
        if (!ok)
        {
            err = "packed blob conv table corrupt at convolution " + std::to_string(i);
            return RSR_E_FORMAT;
        }
    }
    return RSR_OK;
}

} // namespace rsr
