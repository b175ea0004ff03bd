// model.h -- ncnn .param/.bin reader, graph validation and weight packing (host only, no HIP).
//
// Replaces what the reference gets from ncnn::Net::load_param / load_model
// (/root/reference/src/realsr.cpp:75-76) for the one graph it ships:
// models/models-DF2K*/x4.param = ESRGAN RRDBNet(in 3, out 3, nf 64, nb 23, gc 32), 999 layers.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace rsr {

struct ParamLayer
{
    std::string type, name;
    std::vector<std::string> bottoms, tops;
    // scalar params by id, array params by id (id = -23300 - key)
    std::vector<std::pair<int, std::string>> scalars;
    std::vector<std::pair<int, std::vector<float>>> arrays;
    int geti(int id, int def) const;
    float getf(int id, float def) const;
    const std::vector<float>* geta(int id) const;
};

struct ConvRec
{
    int cin = 0, cout = 0;
    int act = 0; // 0 none, 2 leakyrelu
    float slope = 0.f;
    std::vector<float> weight; // OIHW
    std::vector<float> bias;
};

struct Model
{
    int n_layers = 0, n_blobs = 0;
    std::vector<ParamLayer> layers;
    std::vector<ConvRec> convs; // file order == canonical RRDBNet order (validated)
    int bin_encoding = -1;      // 1 fp16-tagged, 0 raw fp32, 2 mixed/table
    long long n_weights = 0, n_biases = 0;
};

// Network constants of the validated graph
constexpr int kNumRRDB = 23;
constexpr int kNumRDB = kNumRRDB * 3; // 69
constexpr int kNumConvs = 1 + kNumRDB * 5 + 5; // 351
constexpr int kNF = 64, kGC = 32;

// Error codes match include/realsr_hip.h
int parse_param(const std::string& path, Model& m, std::string& err);
int validate_graph(const Model& m, std::string& err); // checks DAG == canonical RRDBNet, fills nothing
int read_bin(const std::string& path, Model& m, std::string& err);
int load_model(const std::string& param, const std::string& bin, Model& m, std::string& err);

// ---- packed blob ---------------------------------------------------------------------------
// One relocatable byte blob holding every conv's weights in the exact LDS images the MFMA kernel stages (conv3x3_flow,
// 16-channel planes): per conv, per 16-input-channel plane
//     [tap 0..8][cout row 0..NT*32-1][16 cin] fp16, 32 B per row, the two 16-B slots of a row swapped when (cout>>3)&1
// followed by NT*32 fp32 biases.  Cin is zero-padded to a multiple of 32 (conv_first 3 -> 32: two 16-channel planes,
// the second all zero), Cout to a multiple of 32 (conv_last 3 -> 32).  33.5 MB for x4.param: what rsr_load uploads and
// what the multi-GPU broadcast carries.  A conv with <= 4 output channels (conv_last, 64 -> 3) additionally carries the aux image
// of conv3x3_flow's (dy, cout)-in-M variant: per plane [dx][32 rows][16 cin], row dy*8 + c = tap (dy, dx) of output channel c.
struct PackedHeader
{
    uint32_t magic;   // 'RSRP'
    uint32_t version; // 5
    uint32_t nconv;
    uint32_t flags;   // 0 (version 3 carried the 32-channel images of the round-1 kernels behind bit 0)
    uint64_t total_bytes;
};
struct PackedConv
{
    uint32_t cin, cout;     // true channel counts
    uint32_t act;           // 0 / 2
    uint32_t nplanes;       // ceil(cin/32): 32-channel chunks (= half the number of 16-channel planes)
    uint32_t nt;            // ceil(cout/32)
    float slope;
    uint64_t b_off;         // byte offsets from blob start (256-B aligned)
    uint64_t w16_off;       // 16-channel-plane images
    uint64_t aux_off;       // convs with <= 4 output channels (conv_last): [plane][dx 0..2][row = dy*8 + cout][16 cin], 3 KB per plane; else 0
};
constexpr uint32_t kPackedVersion = 6;

// Which output channel sits in row i (0..31) of a 32-row weight image (and of the bias block): the v_mfma_f32_32x32x16_f16 result
// leaves row q*8 + hi*4 + e in register q*4 + e of lane (pixel, hi).  With the rows ordered like this a lane's registers 0..7 are
// the 8 CONSECUTIVE channels hi*8 .. hi*8+7 of the first output plane and 8..15 those of the second: 16 bytes of fp16 that go to
// memory as they are -- no transpose through LDS in the epilogue (conv_flow.hip).  A bijection on 0..31 (swaps 4-7 <-> 8-11 and
// 20-23 <-> 24-27); rows 0..3 stay channels 0..3 (conv_last's three outputs).
constexpr int row_cout(int i) { return ((i >> 4) & 1) * 16 + ((i >> 2) & 1) * 8 + ((i >> 3) & 1) * 4 + (i & 3); }
constexpr uint32_t kPackedMagic = 0x50525352u; // "RSRP"

size_t packed_size(const Model& m);
int pack_model(const Model& m, void* dst, size_t cap, std::string& err);
// Full validation of a blob that may come from anywhere (a broadcast, a file): header, every offset + extent inside the
// blob, and the conv table equal to the canonical RRDBNet schedule (cin/cout/act per index) the engine hard-wires.
// Only the header + table (first packed_table_bytes() bytes) are dereferenced.
int check_packed(const void* head, size_t head_bytes, size_t total_bytes, std::string& err);
size_t packed_table_bytes();
// cin / cout of convolution `index` (0..350) in x4.param order
void canonical_conv(int index, int& cin, int& cout, int& act);

uint16_t f32_to_f16(float f);
float f16_to_f32(uint16_t h);

} // namespace rsr
