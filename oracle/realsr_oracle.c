/*
 * realsr_oracle.c -- CPU ORACLE for the RealSR x4 tiled-inference hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (librealsr_hip.so) never
 * links, loads or calls anything in oracle/.
 *
 * PARITY STATUS: **parity unpinned**.  The reference (nihui/realsr-ncnn-vulkan) has no tests,
 * no golden vectors, its NN runtime (Tencent/ncnn, git submodule src/ncnn, pinned SHA
 * unrecoverable because /root/reference has no .git) is not vendored, and the model weights
 * (models/x4.bin) are absent.  This file is therefore a plain-C *restatement* of
 *   (1) RealSR::process_cpu            /root/reference/src/realsr.cpp:525-838
 *   (2) the four GLSL compute shaders  /root/reference/src/realsr_{pre,post}proc{,_tta}.comp
 *   (3) the ncnn layer semantics the graph models/models-DF2K/x4.param needs, restated from
 *       ncnn's published behaviour: .param text format, .bin tagged
 *       blobs, Convolution(3x3,s1,p1,bias,leakyrelu), Split, Concat, Eltwise(sum,coeffs),
 *       BinaryOp(add), Interp(nearest / bicubic), copy_make_border(reflect-101),
 *       from_pixels / to_pixels.
 * It is cross-checked in tests/ against an independent PyTorch-CPU construction of the
 * same graph (tests/test_oracle.py).  The graph is *interpreted* from the .param text (generic
 * DAG executor, like ncnn::Extractor), NOT hard-coded, so it is independent of the fused
 * topology the HIP engine uses.
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * fp16 <-> fp32 (IEEE binary16, round-to-nearest-even).  Needed because the reference GPU
 * path stores tiles as fp16 (realsr.cpp:45 use_fp16_storage, shaders' `sfp`), and ncnn's
 * fp16-tagged .bin blobs are binary16.
 * ---------------------------------------------------------------------------------------- */
ORC_API uint16_t orc_f32_to_f16(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t mant = x & 0x007fffffu;
    int32_t exp = (int32_t)((x >> 23) & 0xff);
    if (exp == 0xff) /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | (mant ? 0x200u | (mant >> 13) : 0));
    int32_t e = exp - 127 + 15;
    if (e >= 0x1f) return (uint16_t)(sign | 0x7c00u); /* overflow -> inf */
    if (e <= 0)
    {
        if (e < -10) return (uint16_t)sign; /* underflow -> signed zero */
        mant |= 0x00800000u;
        uint32_t shift = (uint32_t)(14 - e);
        uint32_t half = mant >> shift;
        uint32_t rem = mant & ((1u << shift) - 1);
        uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (half & 1))) half++;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((uint32_t)e << 10) | (mant >> 13);
    uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) half++; /* may carry into exp: ok */
    return (uint16_t)(sign | half);
}

ORC_API float orc_f16_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f;
    uint32_t mant = h & 0x3ffu;
    uint32_t x;
    if (exp == 0)
    {
        if (mant == 0)
            x = sign;
        else
        {
            int e = -1;
            do
            {
                e++;
                mant <<= 1;
            } while (!(mant & 0x400u));
            x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((mant & 0x3ffu) << 13);
        }
    }
    else if (exp == 0x1f)
        x = sign | 0x7f800000u | (mant << 13);
    else
        x = sign | ((exp - 15 + 127) << 23) | (mant << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}

/* ------------------------------------------------------------------------------------------
 * ncnn .param text graph  (file: models/models-DF2K/x4.param)
 * ---------------------------------------------------------------------------------------- */
enum
{
    L_INPUT = 0,
    L_CONV,
    L_SPLIT,
    L_CONCAT,
    L_ELTWISE,
    L_BINARYOP,
    L_INTERP
};

#define ORC_MAX_IO 16

typedef struct
{
    int type;
    char name[64];
    int nin, nout;
    int in[ORC_MAX_IO];
    int out[ORC_MAX_IO];
    /* Convolution (A.3): 0=num_output 1=kernel_w 2=dilation 3=stride 4=pad 5=bias_term
       6=weight_data_size 9=activation_type -23310=activation_params */
    int num_output, kernel, dilation, stride, pad, bias_term, weight_data_size, act_type;
    float act_p0;
    int cin; /* derived = weight_data_size / (num_output*k*k) */
    float* weight; /* OIHW fp32 */
    float* bias;
    /* Eltwise: 0=op_type (1=sum) -23301=coeffs */
    int op_type;
    int ncoeff;
    float coeff[4];
    /* Interp: 0=resize_type (1 nearest, 3 bicubic) 1=height_scale 2=width_scale */
    int resize_type;
    float hscale, wscale;
} orc_layer;

typedef struct orc_net
{
    int nlayers, nblobs;
    orc_layer* layers;
    char (*blob_names)[64];
    int nblob_names;
    int* consumers; /* per blob: how many layer inputs reference it */
    int in_blob, out_blob;
    int nconv;
    int bin_encoding; /* 1 = all weight blobs fp16-tagged, 0 = raw fp32, 2 = mixed/other */
} orc_net;

static int blob_index(orc_net* n, const char* name)
{
    for (int i = 0; i < n->nblob_names; i++)
        if (strcmp(n->blob_names[i], name) == 0) return i;
    if (n->nblob_names >= n->nblobs) return -1;
    snprintf(n->blob_names[n->nblob_names], 64, "%s", name);
    return n->nblob_names++;
}

static void seterr(char* err, int errlen, const char* msg)
{
    if (err && errlen > 0)
    {
        strncpy(err, msg, (size_t)errlen - 1);
        err[errlen - 1] = 0;
    }
}

/* one "key=value" token; array keys are -23300-id with value "n,v0,v1,..." (A.1) */
static void parse_kv(orc_layer* L, const char* tok)
{
    int key = 0;
    const char* eq = strchr(tok, '=');
    if (!eq) return;
    key = atoi(tok);
    const char* val = eq + 1;
    if (key <= -23300)
    {
        int id = -23300 - key;
        int n = atoi(val);
        float v[8] = {0};
        const char* p = strchr(val, ',');
        int i = 0;
        while (p && i < n && i < 8)
        {
            v[i++] = (float)atof(p + 1);
            p = strchr(p + 1, ',');
        }
        if (L->type == L_CONV && id == 10)
            L->act_p0 = v[0];
        if (L->type == L_ELTWISE && id == 1)
        {
            L->ncoeff = n < 4 ? n : 4;
            for (int k = 0; k < L->ncoeff; k++) L->coeff[k] = v[k];
        }
        return;
    }
    int iv = atoi(val);
    float fv = (float)atof(val);
    switch (L->type)
    {
    case L_CONV:
        if (key == 0) L->num_output = iv;
        if (key == 1) L->kernel = iv;
        if (key == 2) L->dilation = iv;
        if (key == 3) L->stride = iv;
        if (key == 4) L->pad = iv;
        if (key == 5) L->bias_term = iv;
        if (key == 6) L->weight_data_size = iv;
        if (key == 9) L->act_type = iv;
        break;
    case L_ELTWISE:
    case L_BINARYOP:
        if (key == 0) L->op_type = iv;
        break;
    case L_INTERP:
        if (key == 0) L->resize_type = iv;
        if (key == 1) L->hscale = fv;
        if (key == 2) L->wscale = fv;
        break;
    default:
        break;
    }
}

static int load_param(orc_net* n, const char* path, char* err, int errlen)
{
    FILE* fp = fopen(path, "rb");
    if (!fp)
    {
        seterr(err, errlen, "cannot open .param");
        return -1;
    }
    int magic = 0;
    if (fscanf(fp, "%d", &magic) != 1 || magic != 7767517)
    {
        seterr(err, errlen, "bad .param magic (want 7767517)");
        fclose(fp);
        return -1;
    }
    if (fscanf(fp, "%d %d", &n->nlayers, &n->nblobs) != 2 || n->nlayers <= 0 || n->nblobs <= 0)
    {
        seterr(err, errlen, "bad layer/blob count");
        fclose(fp);
        return -1;
    }
    n->layers = (orc_layer*)calloc((size_t)n->nlayers, sizeof(orc_layer));
    n->blob_names = (char(*)[64])calloc((size_t)n->nblobs, 64);
    n->consumers = (int*)calloc((size_t)n->nblobs, sizeof(int));
    n->in_blob = n->out_blob = -1;
    char tok[256];
    for (int li = 0; li < n->nlayers; li++)
    {
        orc_layer* L = &n->layers[li];
        char type[64];
        if (fscanf(fp, "%63s %63s %d %d", type, L->name, &L->nin, &L->nout) != 4)
        {
            seterr(err, errlen, "truncated .param");
            fclose(fp);
            return -1;
        }
        if (L->nin > ORC_MAX_IO || L->nout > ORC_MAX_IO)
        {
            seterr(err, errlen, "too many layer inputs/outputs");
            fclose(fp);
            return -1;
        }
        if (!strcmp(type, "Input")) L->type = L_INPUT;
        else if (!strcmp(type, "Convolution")) L->type = L_CONV;
        else if (!strcmp(type, "Split")) L->type = L_SPLIT;
        else if (!strcmp(type, "Concat")) L->type = L_CONCAT;
        else if (!strcmp(type, "Eltwise")) L->type = L_ELTWISE;
        else if (!strcmp(type, "BinaryOp")) L->type = L_BINARYOP;
        else if (!strcmp(type, "Interp")) L->type = L_INTERP;
        else
        {
            char m[128];
            snprintf(m, sizeof m, "unsupported layer type %s", type);
            seterr(err, errlen, m);
            fclose(fp);
            return -1;
        }
        /* defaults (A.3) */
        L->kernel = 0; L->dilation = 1; L->stride = 1; L->pad = 0; L->bias_term = 0;
        L->act_type = 0; L->act_p0 = 0.f; L->op_type = 0; L->ncoeff = 0;
        L->resize_type = 0; L->hscale = 1.f; L->wscale = 1.f;
        for (int i = 0; i < L->nin; i++)
        {
            if (fscanf(fp, "%255s", tok) != 1) { seterr(err, errlen, "truncated"); fclose(fp); return -1; }
            L->in[i] = blob_index(n, tok);
            if (L->in[i] < 0) { seterr(err, errlen, "blob count overflow"); fclose(fp); return -1; }
            n->consumers[L->in[i]]++;
        }
        for (int i = 0; i < L->nout; i++)
        {
            if (fscanf(fp, "%255s", tok) != 1) { seterr(err, errlen, "truncated"); fclose(fp); return -1; }
            L->out[i] = blob_index(n, tok);
            if (L->out[i] < 0) { seterr(err, errlen, "blob count overflow"); fclose(fp); return -1; }
            if (L->type == L_INPUT && !strcmp(tok, "data")) n->in_blob = L->out[i];
            if (!strcmp(tok, "output")) n->out_blob = L->out[i];
        }
        /* rest of the line: key=value tokens */
        int ch;
        for (;;)
        {
            /* skip spaces, stop at newline */
            while ((ch = fgetc(fp)) == ' ' || ch == '\t' || ch == '\r') {}
            if (ch == '\n' || ch == EOF) break;
            int k = 0;
            tok[k++] = (char)ch;
            while ((ch = fgetc(fp)) != EOF && ch != ' ' && ch != '\t' && ch != '\n' && ch != '\r')
                if (k < 255) tok[k++] = (char)ch;
            tok[k] = 0;
            parse_kv(L, tok);
            if (ch == '\n' || ch == EOF) break;
        }
        if (L->type == L_CONV)
        {
            if (L->kernel != 3 || L->stride != 1 || L->dilation != 1 || L->pad != 1 || !L->bias_term ||
                L->num_output <= 0 || L->weight_data_size % (L->num_output * 9) != 0)
            {
                seterr(err, errlen, "Convolution outside the 3x3/s1/p1/bias subset used by x4.param");
                fclose(fp);
                return -1;
            }
            L->cin = L->weight_data_size / (L->num_output * 9);
            n->nconv++;
        }
    }
    fclose(fp);
    if (n->in_blob < 0 || n->out_blob < 0)
    {
        seterr(err, errlen, "graph lacks blob 'data' or 'output' (realsr.cpp:772,775)");
        return -1;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * ncnn .bin (A.2).  Per Convolution in .param order: weights = "type 0" blob (4-byte tag then
 * payload), bias = "type 1" blob (raw fp32).
 * ---------------------------------------------------------------------------------------- */
static int read_weight_blob(FILE* fp, float* dst, int n, int* enc)
{
    uint32_t tag;
    if (fread(&tag, 4, 1, fp) != 1) return -1;
    if (tag == 0x01306B47u) /* fp16 */
    {
        size_t nbytes = ((size_t)n * 2 + 3) & ~(size_t)3; /* payload padded to 4 bytes */
        uint16_t* tmp = (uint16_t*)malloc(nbytes);
        if (fread(tmp, 1, nbytes, fp) != nbytes) { free(tmp); return -1; }
        for (int i = 0; i < n; i++) dst[i] = orc_f16_to_f32(tmp[i]);
        free(tmp);
        *enc = 1;
        return 0;
    }
    if (tag == 0 || tag == 0x0002C056u) /* raw fp32 */
    {
        if (fread(dst, 4, (size_t)n, fp) != (size_t)n) return -1;
        *enc = 0;
        return 0;
    }
    if (tag == 0x000D4B38u) return -2; /* int8: never produced for this model */
    /* any other tag with non-zero byte sum: 256-entry fp32 table + n uint8 indices (pad 4) */
    {
        float table[256];
        if (fread(table, 4, 256, fp) != 256) return -1;
        size_t nbytes = ((size_t)n + 3) & ~(size_t)3;
        uint8_t* idx = (uint8_t*)malloc(nbytes);
        if (fread(idx, 1, nbytes, fp) != nbytes) { free(idx); return -1; }
        for (int i = 0; i < n; i++) dst[i] = table[idx[i]];
        free(idx);
        *enc = 2;
        return 0;
    }
}

static int load_bin(orc_net* n, const char* path, char* err, int errlen)
{
    FILE* fp = fopen(path, "rb");
    if (!fp)
    {
        seterr(err, errlen, "cannot open .bin");
        return -1;
    }
    int enc_all = -1;
    for (int li = 0; li < n->nlayers; li++)
    {
        orc_layer* L = &n->layers[li];
        if (L->type != L_CONV) continue;
        L->weight = (float*)malloc(sizeof(float) * (size_t)L->weight_data_size);
        L->bias = (float*)malloc(sizeof(float) * (size_t)L->num_output);
        int enc = 0;
        int r = read_weight_blob(fp, L->weight, L->weight_data_size, &enc);
        if (r != 0)
        {
            seterr(err, errlen, r == -2 ? "int8 weight blobs unsupported" : "truncated .bin (weights)");
            fclose(fp);
            return -1;
        }
        if (enc_all == -1) enc_all = enc;
        else if (enc_all != enc) enc_all = 2;
        if (fread(L->bias, 4, (size_t)L->num_output, fp) != (size_t)L->num_output)
        {
            seterr(err, errlen, "truncated .bin (bias)");
            fclose(fp);
            return -1;
        }
    }
    /* must be at EOF */
    uint8_t extra;
    if (fread(&extra, 1, 1, fp) == 1)
    {
        seterr(err, errlen, ".bin has trailing bytes");
        fclose(fp);
        return -1;
    }
    fclose(fp);
    n->bin_encoding = enc_all;
    return 0;
}

ORC_API void orc_net_free(orc_net* n)
{
    if (!n) return;
    if (n->layers)
        for (int i = 0; i < n->nlayers; i++)
        {
            free(n->layers[i].weight);
            free(n->layers[i].bias);
        }
    free(n->layers);
    free(n->blob_names);
    free(n->consumers);
    free(n);
}

/* restates ncnn::Net::load_param + load_model as called at realsr.cpp:75-76 */
ORC_API orc_net* orc_net_load(const char* param_path, const char* bin_path, char* err, int errlen)
{
    orc_net* n = (orc_net*)calloc(1, sizeof(orc_net));
    if (load_param(n, param_path, err, errlen) != 0 || load_bin(n, bin_path, err, errlen) != 0)
    {
        orc_net_free(n);
        return NULL;
    }
    return n;
}

ORC_API int orc_net_num_layers(const orc_net* n) { return n->nlayers; }
ORC_API int orc_net_num_convs(const orc_net* n) { return n->nconv; }
ORC_API int orc_net_bin_encoding(const orc_net* n) { return n->bin_encoding; }

/* i-th Convolution in file order: returns cin,cout,act (0 none / 2 leakyrelu), slope, pointers */
ORC_API int orc_net_conv_info(const orc_net* n, int idx, int* cin, int* cout, int* act, float* slope,
                              const float** weight, const float** bias)
{
    int k = 0;
    for (int li = 0; li < n->nlayers; li++)
    {
        const orc_layer* L = &n->layers[li];
        if (L->type != L_CONV) continue;
        if (k++ == idx)
        {
            if (cin) *cin = L->cin;
            if (cout) *cout = L->num_output;
            if (act) *act = L->act_type;
            if (slope) *slope = L->act_p0;
            if (weight) *weight = L->weight;
            if (bias) *bias = L->bias;
            return 0;
        }
    }
    return -1;
}

/* ------------------------------------------------------------------------------------------
 * Layer arithmetic (planar CHW fp32, as ncnn::Mat on the CPU path with fp16 storage off,
 * realsr.cpp:45-46).
 * ---------------------------------------------------------------------------------------- */

/* Convolution 3x3 s1 p1 (zero pad) + bias + activation (A.3).  Direct, fp32.  ncnn x86 may use
   winograd/sgemm; summation order differs at the 1e-6 level (A.7). */
ORC_API void orc_conv3x3(const float* in, int cin, int h, int w, const float* weight, const float* bias,
                         int cout, int act_type, float slope, float* out)
{
    const int pw = w + 2, ph = h + 2;
    float* pad = (float*)calloc((size_t)cin * ph * pw, sizeof(float));
#pragma omp parallel for schedule(static)
    for (int c = 0; c < cin; c++)
        for (int y = 0; y < h; y++)
            memcpy(pad + ((size_t)c * ph + y + 1) * pw + 1, in + ((size_t)c * h + y) * w, sizeof(float) * (size_t)w);

    /* Register blocking for speed only (the test suite spends most of its time here): OB output channels x XB pixels of one row
       accumulate over all input channels in a small local array, so every loaded input value feeds OB outputs.  Per output the
       operations and their order are the ones of the plain triple loop: bias, then for ic ascending the three tap rows. */
    enum { OB = 4, XB = 64 };
    const int RB = 8; /* row block */
    const int nrb = (h + RB - 1) / RB, nob = (cout + OB - 1) / OB;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int ob = 0; ob < nob; ob++)
        for (int rb = 0; rb < nrb; rb++)
        {
            const int oc0 = ob * OB, noc = (oc0 + OB <= cout) ? OB : cout - oc0;
            const int y0 = rb * RB, y1 = (y0 + RB < h) ? y0 + RB : h;
            for (int y = y0; y < y1; y++)
                for (int x0 = 0; x0 < w; x0 += XB)
                {
                    const int n = (x0 + XB <= w) ? XB : w - x0;
                    float acc[OB][XB];
                    for (int j = 0; j < OB; j++)
                    {
                        const float b = (bias && j < noc) ? bias[oc0 + j] : 0.f;
                        for (int x = 0; x < n; x++) acc[j][x] = b;
                    }
                    for (int ic = 0; ic < cin; ic++)
                    {
                        const float* r0 = pad + ((size_t)ic * ph + y) * pw + x0;
                        const float* r1 = r0 + pw;
                        const float* r2 = r1 + pw;
                        float k[OB][9];
                        for (int j = 0; j < OB; j++)
                            for (int t = 0; t < 9; t++) k[j][t] = j < noc ? weight[((size_t)(oc0 + j) * cin + ic) * 9 + t] : 0.f;
                        for (int x = 0; x < n; x++)
                        {
                            const float a0 = r0[x], a1 = r0[x + 1], a2 = r0[x + 2];
                            const float a3 = r1[x], a4 = r1[x + 1], a5 = r1[x + 2];
                            const float a6 = r2[x], a7 = r2[x + 1], a8 = r2[x + 2];
#define ORC_ACC(J)                                                                     \
    {                                                                                  \
        float s = acc[J][x];                                                           \
        s += k[J][0] * a0 + k[J][1] * a1 + k[J][2] * a2;                               \
        s += k[J][3] * a3 + k[J][4] * a4 + k[J][5] * a5;                               \
        s += k[J][6] * a6 + k[J][7] * a7 + k[J][8] * a8;                               \
        acc[J][x] = s;                                                                 \
    }
                            ORC_ACC(0) ORC_ACC(1) ORC_ACC(2) ORC_ACC(3)
#undef ORC_ACC
                        }
                    }
                    for (int j = 0; j < noc; j++)
                    {
                        float* orow = out + ((size_t)(oc0 + j) * h + y) * w + x0;
                        for (int x = 0; x < n; x++)
                        {
                            float v = acc[j][x];
                            if (act_type == 2) v = v < 0.f ? v * slope : v; /* leakyrelu */
                            else if (act_type == 1) v = v < 0.f ? 0.f : v;
                            orow[x] = v;
                        }
                    }
                }
        }
    free(pad);
}

/* Interp nearest (A.4): in_y = min((int)(y * (1/scale)), h-1) */
static void interp_nearest(const float* in, int c, int h, int w, int oh, int ow, float hs, float ws, float* out)
{
#pragma omp parallel for schedule(static)
    for (int q = 0; q < c; q++)
        for (int y = 0; y < oh; y++)
        {
            int iy = (int)(y * hs);
            if (iy > h - 1) iy = h - 1;
            const float* ir = in + ((size_t)q * h + iy) * w;
            float* orow = out + ((size_t)q * oh + y) * ow;
            for (int x = 0; x < ow; x++)
            {
                int ix = (int)(x * ws);
                if (ix > w - 1) ix = w - 1;
                orow[x] = ir[ix];
            }
        }
}

/* ncnn Interp bicubic (resize_type 3), A.5: half-pixel mapping, Keys cubic A=-0.75, border
   taps folded by re-weighting (not index clamping), separable.  Used for the alpha channel only
   (realsr.cpp:128-140). */
static void cubic_coeffs(int w, int outw, int* xofs, float* alpha)
{
    double scale = (double)w / outw;
    for (int dx = 0; dx < outw; dx++)
    {
        float fx = (float)((dx + 0.5) * scale - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        const float A = -0.75f;
        float fx0 = fx + 1, fx1 = fx, fx2 = 1 - fx;
        float c0 = A * fx0 * fx0 * fx0 - 5 * A * fx0 * fx0 + 8 * A * fx0 - 4 * A;
        float c1 = (A + 2) * fx1 * fx1 * fx1 - (A + 3) * fx1 * fx1 + 1;
        float c2 = (A + 2) * fx2 * fx2 * fx2 - (A + 3) * fx2 * fx2 + 1;
        float c3 = 1.f - c0 - c1 - c2;
        if (sx <= -1)
        {
            sx = 1;
            c0 = 1.f - c3;
            c1 = c3;
            c2 = 0.f;
            c3 = 0.f;
        }
        if (sx == 0)
        {
            sx = 1;
            c0 = c0 + c1;
            c1 = c2;
            c2 = c3;
            c3 = 0.f;
        }
        if (sx == w - 2)
        {
            sx = w - 3;
            c3 = c2 + c3;
            c2 = c1;
            c1 = c0;
            c0 = 0.f;
        }
        if (sx >= w - 1)
        {
            sx = w - 3;
            c3 = 1.f - c0;
            c2 = c0;
            c1 = 0.f;
            c0 = 0.f;
        }
        xofs[dx] = sx;
        alpha[dx * 4 + 0] = c0;
        alpha[dx * 4 + 1] = c1;
        alpha[dx * 4 + 2] = c2;
        alpha[dx * 4 + 3] = c3;
    }
}

ORC_API void orc_bicubic(const float* in, int h, int w, int oh, int ow, float* out)
{
    int* xofs = (int*)malloc(sizeof(int) * (size_t)ow);
    int* yofs = (int*)malloc(sizeof(int) * (size_t)oh);
    float* xa = (float*)malloc(sizeof(float) * 4 * (size_t)ow);
    float* ya = (float*)malloc(sizeof(float) * 4 * (size_t)oh);
    cubic_coeffs(w, ow, xofs, xa);
    cubic_coeffs(h, oh, yofs, ya);
    /* rows first (horizontal pass per needed source row), then columns: same result as ncnn's
       rolling-row implementation up to fp32 association order inside each 4-tap sum */
    float* rows = (float*)malloc(sizeof(float) * (size_t)h * ow);
    for (int y = 0; y < h; y++)
        for (int dx = 0; dx < ow; dx++)
        {
            const float* S = in + (size_t)y * w + xofs[dx];
            const float* a = xa + dx * 4;
            /* taps sx-1 .. sx+2 ; degenerate tiny widths are clamped */
            int i0 = xofs[dx] - 1, i1 = xofs[dx], i2 = xofs[dx] + 1, i3 = xofs[dx] + 2;
            (void)S;
#define CLAMPI(i, n) ((i) < 0 ? 0 : ((i) > (n)-1 ? (n)-1 : (i)))
            const float* R = in + (size_t)y * w;
            rows[(size_t)y * ow + dx] = R[CLAMPI(i0, w)] * a[0] + R[CLAMPI(i1, w)] * a[1] + R[CLAMPI(i2, w)] * a[2] + R[CLAMPI(i3, w)] * a[3];
        }
    for (int dy = 0; dy < oh; dy++)
    {
        int sy = yofs[dy];
        const float* b = ya + dy * 4;
        const float* r0 = rows + (size_t)CLAMPI(sy - 1, h) * ow;
        const float* r1 = rows + (size_t)CLAMPI(sy, h) * ow;
        const float* r2 = rows + (size_t)CLAMPI(sy + 1, h) * ow;
        const float* r3 = rows + (size_t)CLAMPI(sy + 2, h) * ow;
        for (int dx = 0; dx < ow; dx++)
            out[(size_t)dy * ow + dx] = r0[dx] * b[0] + r1[dx] * b[1] + r2[dx] * b[2] + r3[dx] * b[3];
    }
#undef CLAMPI
    free(rows);
    free(xofs);
    free(yofs);
    free(xa);
    free(ya);
}

/* refcounted buffers so Split can alias (ncnn Split is a refcount artefact, A.3) */
typedef struct
{
    float* p;
    int rc;
} orc_buf;

typedef struct
{
    orc_buf* buf;
    int c, h, w;
    int pending; /* consumers left */
} orc_blob;

static void blob_release(orc_blob* b)
{
    if (b->buf && --b->buf->rc == 0)
    {
        free(b->buf->p);
        free(b->buf);
    }
    b->buf = NULL;
}

static orc_buf* buf_new(size_t n)
{
    orc_buf* b = (orc_buf*)malloc(sizeof(orc_buf));
    b->p = (float*)malloc(sizeof(float) * n);
    b->rc = 1;
    return b;
}

/* Restates `ex.input("data", in); ex.extract("output", out)` (realsr.cpp:771-775): runs the
   whole DAG on one CHW fp32 tile.  Output buffer must hold out_c*out_h*out_w floats; the shape is
   returned through oc/oh/ow.  Returns 0 on success. */
ORC_API int orc_net_forward(const orc_net* n, const float* in, int c, int h, int w, float* out, int* oc, int* oh, int* ow)
{
    orc_blob* B = (orc_blob*)calloc((size_t)n->nblobs, sizeof(orc_blob));
    int rc = 0;
    for (int li = 0; li < n->nlayers && rc == 0; li++)
    {
        const orc_layer* L = &n->layers[li];
        switch (L->type)
        {
        case L_INPUT:
        {
            orc_blob* o = &B[L->out[0]];
            o->buf = buf_new((size_t)c * h * w);
            memcpy(o->buf->p, in, sizeof(float) * (size_t)c * h * w);
            o->c = c; o->h = h; o->w = w;
            break;
        }
        case L_CONV:
        {
            orc_blob* i0 = &B[L->in[0]];
            if (!i0->buf || i0->c != L->cin) { rc = -10; break; }
            orc_blob* o = &B[L->out[0]];
            o->buf = buf_new((size_t)L->num_output * i0->h * i0->w);
            o->c = L->num_output; o->h = i0->h; o->w = i0->w;
            orc_conv3x3(i0->buf->p, L->cin, i0->h, i0->w, L->weight, L->bias, L->num_output, L->act_type, L->act_p0, o->buf->p);
            break;
        }
        case L_SPLIT:
        {
            orc_blob* i0 = &B[L->in[0]];
            if (!i0->buf) { rc = -11; break; }
            for (int k = 0; k < L->nout; k++)
            {
                orc_blob* o = &B[L->out[k]];
                o->buf = i0->buf;
                i0->buf->rc++;
                o->c = i0->c; o->h = i0->h; o->w = i0->w;
            }
            break;
        }
        case L_CONCAT:
        {
            int ctot = 0;
            for (int k = 0; k < L->nin; k++)
            {
                if (!B[L->in[k]].buf) { rc = -12; break; }
                ctot += B[L->in[k]].c;
            }
            if (rc) break;
            orc_blob* i0 = &B[L->in[0]];
            orc_blob* o = &B[L->out[0]];
            o->buf = buf_new((size_t)ctot * i0->h * i0->w);
            o->c = ctot; o->h = i0->h; o->w = i0->w;
            size_t off = 0;
            for (int k = 0; k < L->nin; k++) /* axis 0 = channel, inputs in listed order */
            {
                orc_blob* ik = &B[L->in[k]];
                size_t sz = (size_t)ik->c * ik->h * ik->w;
                memcpy(o->buf->p + off, ik->buf->p, sizeof(float) * sz);
                off += sz;
            }
            break;
        }
        case L_ELTWISE: /* op_type 1 = sum with coeffs: out = c0*in0 + c1*in1 (A.3) */
        {
            orc_blob* a = &B[L->in[0]];
            orc_blob* b = &B[L->in[1]];
            if (!a->buf || !b->buf || L->nin != 2 || L->op_type != 1) { rc = -13; break; }
            orc_blob* o = &B[L->out[0]];
            size_t sz = (size_t)a->c * a->h * a->w;
            o->buf = buf_new(sz);
            o->c = a->c; o->h = a->h; o->w = a->w;
            const float c0 = L->ncoeff >= 2 ? L->coeff[0] : 1.f, c1 = L->ncoeff >= 2 ? L->coeff[1] : 1.f;
            const float* pa = a->buf->p;
            const float* pb = b->buf->p;
            float* po = o->buf->p;
            /* ncnn Eltwise SUM with coeffs: first two inputs  out = a*c0 + b*c1 */
#pragma omp parallel for schedule(static)
            for (long i = 0; i < (long)sz; i++) po[i] = pa[i] * c0 + pb[i] * c1;
            break;
        }
        case L_BINARYOP: /* op_type 0 = add, no scalar */
        {
            orc_blob* a = &B[L->in[0]];
            orc_blob* b = &B[L->in[1]];
            if (!a->buf || !b->buf || L->op_type != 0) { rc = -14; break; }
            orc_blob* o = &B[L->out[0]];
            size_t sz = (size_t)a->c * a->h * a->w;
            o->buf = buf_new(sz);
            o->c = a->c; o->h = a->h; o->w = a->w;
            const float* pa = a->buf->p;
            const float* pb = b->buf->p;
            float* po = o->buf->p;
#pragma omp parallel for schedule(static)
            for (long i = 0; i < (long)sz; i++) po[i] = pa[i] + pb[i];
            break;
        }
        case L_INTERP:
        {
            orc_blob* i0 = &B[L->in[0]];
            if (!i0->buf || L->resize_type != 1) { rc = -15; break; }
            orc_blob* o = &B[L->out[0]];
            int oh_ = (int)(i0->h * L->hscale), ow_ = (int)(i0->w * L->wscale);
            o->buf = buf_new((size_t)i0->c * oh_ * ow_);
            o->c = i0->c; o->h = oh_; o->w = ow_;
            interp_nearest(i0->buf->p, i0->c, i0->h, i0->w, oh_, ow_, 1.f / L->hscale, 1.f / L->wscale, o->buf->p);
            break;
        }
        }
        if (rc) break;
        /* set consumer counts on outputs, release consumed inputs */
        for (int k = 0; k < L->nout; k++) B[L->out[k]].pending = n->consumers[L->out[k]];
        for (int k = 0; k < L->nin; k++)
        {
            orc_blob* ib = &B[L->in[k]];
            if (--ib->pending <= 0 && L->in[k] != n->out_blob) blob_release(ib);
        }
    }
    if (rc == 0)
    {
        orc_blob* o = &B[n->out_blob];
        if (!o->buf) rc = -20;
        else
        {
            if (oc) *oc = o->c;
            if (oh) *oh = o->h;
            if (ow) *ow = o->w;
            memcpy(out, o->buf->p, sizeof(float) * (size_t)o->c * o->h * o->w);
        }
    }
    for (int i = 0; i < n->nblobs; i++)
        if (B[i].buf) blob_release(&B[i]);
    free(B);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * RealSR::process_cpu restatement  (/root/reference/src/realsr.cpp:525-838)
 * ---------------------------------------------------------------------------------------- */

/* ncnn::copy_make_border(type=2) = reflect-101 (edge pixel not repeated), A.6; realsr.cpp:613,764 */
static void reflect_pad(const float* in, int c, int h, int w, int top, int bottom, int left, int right, float* out)
{
    const int oh = h + top + bottom, ow = w + left + right;
    for (int q = 0; q < c; q++)
        for (int y = 0; y < oh; y++)
        {
            int sy = y - top;
            if (sy < 0) sy = -sy;
            if (sy > h - 1) sy = (h - 1) - (sy - (h - 1));
            /* like the shaders (realsr_preproc.comp:59-62) a single reflection; clamp for degenerate sizes */
            if (sy < 0) sy = 0;
            if (sy > h - 1) sy = h - 1;
            for (int x = 0; x < ow; x++)
            {
                int sx = x - left;
                if (sx < 0) sx = -sx;
                if (sx > w - 1) sx = (w - 1) - (sx - (w - 1));
                if (sx < 0) sx = 0;
                if (sx > w - 1) sx = w - 1;
                out[((size_t)q * oh + y) * ow + x] = in[((size_t)q * h + sy) * w + sx];
            }
        }
}

/* ncnn to_pixels saturating cast (A.6): (uchar) min(max((int)v, 0), 255), realsr.cpp:820-831 */
static inline uint8_t sat_u8(float v)
{
    int i = (int)v;
    return (uint8_t)(i < 0 ? 0 : (i > 255 ? 255 : i));
}

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/*
 * in : HWC uint8, w*h*c, c in {3,4}      (main.cpp:275)
 * out: HWC uint8, (w*scale)*(h*scale)*c  (main.cpp:276)
 * out_f32 (optional, may be NULL): HWC float, the pre-quantise value in [0,1] units (i.e. the network
 *      `output` blob, TTA-averaged), same geometry as out, RGB channels only meaningful.
 * Alpha (c==4) follows the GPU shaders' semantics (crop to the un-padded tile, bicubic x4 of the
 * un-normalised alpha, +0.5, saturate) -- SURVEY.md 8(c) quirk (i): the CPU reference's alpha path is
 * broken for multi-tile images, the shaders are authoritative.
 * Returns 0, or <0 on error.
 */
ORC_API int orc_process(const orc_net* net, const uint8_t* in, int w, int h, int c, uint8_t* out, int scale,
                        int tilesize, int prepadding, int tta, float* out_f32)
{
    if (!net || !in || !out || (c != 3 && c != 4) || scale != 4 || tilesize < 1 || w < 1 || h < 1) return -1;
    const int T = tilesize, P = prepadding;
    const int xtiles = (w + T - 1) / T; /* realsr.cpp:538-539 */
    const int ytiles = (h + T - 1) / T;
    int rc = 0;
    for (int yi = 0; yi < ytiles && rc == 0; yi++)
    {
        const int tile_h_nopad = imin((yi + 1) * T, h) - yi * T; /* :543 */
        const int in_tile_y0 = imax(yi * T - P, 0);               /* :545 */
        const int in_tile_y1 = imin((yi + 1) * T + P, h);         /* :546 */
        for (int xi = 0; xi < xtiles && rc == 0; xi++)
        {
            const int tile_w_nopad = imin((xi + 1) * T, w) - xi * T; /* :550 */
            const int in_tile_x0 = imax(xi * T - P, 0);               /* :552 */
            const int in_tile_x1 = imin((xi + 1) * T + P, w);         /* :553 */
            const int cw = in_tile_x1 - in_tile_x0, chh = in_tile_y1 - in_tile_y0;

            /* from_pixels_roi (A.6) then * 1/255 (:738-748) */
            float* crop = (float*)malloc(sizeof(float) * 3 * (size_t)cw * chh);
            for (int q = 0; q < 3; q++)
                for (int y = 0; y < chh; y++)
                    for (int x = 0; x < cw; x++)
                    {
                        float v = (float)in[((size_t)(in_tile_y0 + y) * w + in_tile_x0 + x) * c + q];
                        crop[((size_t)q * chh + y) * cw + x] = v * (1 / 255.f);
                    }
            /* border padding (:756-766 / :606-615) */
            const int pad_top = imax(P - yi * T, 0);
            const int pad_bottom = imax(imin((yi + 1) * T + P - h, P), 0);
            const int pad_left = imax(P - xi * T, 0);
            const int pad_right = imax(imin((xi + 1) * T + P - w, P), 0);
            const int tw = cw + pad_left + pad_right, th = chh + pad_top + pad_bottom;
            float* tile0 = (float*)malloc(sizeof(float) * 3 * (size_t)tw * th);
            reflect_pad(crop, 3, chh, cw, pad_top, pad_bottom, pad_left, pad_right, tile0);
            free(crop);

            const int OW = tw * scale, OH = th * scale;
            const int ow_nopad = tile_w_nopad * scale, oh_nopad = tile_h_nopad * scale;
            float* res = (float*)malloc(sizeof(float) * 3 * (size_t)ow_nopad * oh_nopad); /* in [0,1] units */

            if (!tta)
            {
                float* o = (float*)malloc(sizeof(float) * 3 * (size_t)OW * OH);
                int oc_, oh_, ow_;
                rc = orc_net_forward(net, tile0, 3, th, tw, o, &oc_, &oh_, &ow_); /* :768-776 */
                if (rc == 0 && (oc_ != 3 || oh_ != OH || ow_ != OW)) rc = -30;
                if (rc == 0)
                    for (int q = 0; q < 3; q++) /* :794-807 crop prepadding*scale */
                        for (int i = 0; i < oh_nopad; i++)
                            for (int j = 0; j < ow_nopad; j++)
                                res[((size_t)q * oh_nopad + i) * ow_nopad + j] = o[((size_t)q * OH + i + P * scale) * OW + j + P * scale];
                free(o);
            }
            else
            {
                /* the other 7 directions (:617-664): tiles 0-3 are tw x th, 4-7 are th x tw */
                float* it[8];
                float* ot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int k = 0; k < 8; k++) it[k] = (float*)malloc(sizeof(float) * 3 * (size_t)tw * th);
                memcpy(it[0], tile0, sizeof(float) * 3 * (size_t)tw * th);
                for (int q = 0; q < 3; q++)
                    for (int i = 0; i < th; i++)
                        for (int j = 0; j < tw; j++)
                        {
                            float v = tile0[((size_t)q * th + i) * tw + j];
                            it[1][((size_t)q * th + (th - 1 - i)) * tw + j] = v;            /* :641,654 */
                            it[2][((size_t)q * th + i) * tw + (tw - 1 - j)] = v;            /* :642,655 */
                            it[3][((size_t)q * th + (th - 1 - i)) * tw + (tw - 1 - j)] = v; /* :643,656 */
                            it[4][((size_t)q * tw + j) * th + i] = v;                       /* :647 */
                            it[5][((size_t)q * tw + (tw - 1 - j)) * th + i] = v;            /* :648 */
                            it[6][((size_t)q * tw + j) * th + (th - 1 - i)] = v;            /* :649 */
                            it[7][((size_t)q * tw + (tw - 1 - j)) * th + (th - 1 - i)] = v; /* :650 */
                        }
                for (int k = 0; k < 8 && rc == 0; k++) /* :666-675 */
                {
                    ot[k] = (float*)malloc(sizeof(float) * 3 * (size_t)OW * OH);
                    int oc_, oh_, ow_;
                    int kh = k < 4 ? th : tw, kw = k < 4 ? tw : th;
                    rc = orc_net_forward(net, it[k], 3, kh, kw, ot[k], &oc_, &oh_, &ow_);
                    if (rc == 0 && (oc_ != 3 || oh_ != kh * scale || ow_ != kw * scale)) rc = -30;
                    free(it[k]);
                    it[k] = NULL;
                }
                if (rc == 0)
                {
                    const int PS = P * scale;
                    for (int q = 0; q < 3; q++) /* :693-724 */
                        for (int i = 0; i < oh_nopad; i++)
                            for (int j = 0; j < ow_nopad; j++)
                            {
                                float v0 = ot[0][((size_t)q * OH + i + PS) * OW + j + PS];
                                float v1 = ot[1][((size_t)q * OH + (OH - 1 - i - PS)) * OW + j + PS];
                                float v2 = ot[2][((size_t)q * OH + i + PS) * OW + (OW - 1 - PS - j)];
                                float v3 = ot[3][((size_t)q * OH + (OH - 1 - i - PS)) * OW + (OW - 1 - PS - j)];
                                /* 4-7 are transposed: OW rows of OH */
                                float v4 = ot[4][((size_t)q * OW + j + PS) * OH + i + PS];
                                float v5 = ot[5][((size_t)q * OW + (OW - 1 - j - PS)) * OH + i + PS];
                                float v6 = ot[6][((size_t)q * OW + j + PS) * OH + (OH - 1 - i - PS)];
                                float v7 = ot[7][((size_t)q * OW + (OW - 1 - j - PS)) * OH + (OH - 1 - i - PS)];
                                res[((size_t)q * oh_nopad + i) * ow_nopad + j] = (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7) / 8; /* :719 */
                            }
                }
                for (int k = 0; k < 8; k++)
                {
                    free(it[k]);
                    free(ot[k]);
                }
            }
            free(tile0);

            /* alpha: GPU-shader semantics (realsr_preproc.comp:79-88, realsr.cpp:431-442, realsr_postproc.comp:58-61) */
            float* alpha_out = NULL;
            if (rc == 0 && c == 4)
            {
                float* a = (float*)malloc(sizeof(float) * (size_t)tile_w_nopad * tile_h_nopad);
                for (int y = 0; y < tile_h_nopad; y++)
                    for (int x = 0; x < tile_w_nopad; x++)
                        a[(size_t)y * tile_w_nopad + x] = (float)in[((size_t)(yi * T + y) * w + xi * T + x) * 4 + 3];
                alpha_out = (float*)malloc(sizeof(float) * (size_t)ow_nopad * oh_nopad);
                orc_bicubic(a, tile_h_nopad, tile_w_nopad, oh_nopad, ow_nopad, alpha_out);
                free(a);
            }

            if (rc == 0)
            {
                /* `* 255.f + 0.5f` (:804) then to_pixels at the tile offset with full-image stride (:816-833) */
                const size_t ostride = (size_t)w * scale * c;
                uint8_t* obase = out + (size_t)yi * scale * T * ostride + (size_t)xi * scale * T * c;
                for (int i = 0; i < oh_nopad; i++)
                    for (int j = 0; j < ow_nopad; j++)
                    {
                        for (int q = 0; q < 3; q++)
                        {
                            float r = res[((size_t)q * oh_nopad + i) * ow_nopad + j];
                            obase[(size_t)i * ostride + (size_t)j * c + q] = sat_u8(r * 255.f + 0.5f);
                            if (out_f32)
                                out_f32[((size_t)(yi * scale * T + i) * w * scale + xi * scale * T + j) * c + q] = r;
                        }
                        if (c == 4)
                        {
                            float av = alpha_out[(size_t)i * ow_nopad + j];
                            obase[(size_t)i * ostride + (size_t)j * c + 3] = sat_u8(av + 0.5f);
                            if (out_f32) out_f32[((size_t)(yi * scale * T + i) * w * scale + xi * scale * T + j) * c + 3] = av / 255.f;
                        }
                    }
            }
            free(alpha_out);
            free(res);
        }
    }
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Scalar restatements of the four compute shaders (uint8 storage + fp16 storage variant,
 * i.e. NCNN_int8_storage && NCNN_fp16_storage, which is what a modern GPU runs: realsr.cpp:44-47).
 * One C loop iteration == one shader invocation (gx,gy,gz).
 * ---------------------------------------------------------------------------------------- */

/* realsr_preproc.comp:47-95.  `bottom` = band u8 HWC (w x h x channels).  `top` = planar fp16,
   3 * outcstep halfs.  alpha (may be NULL) = alphaw*alphah halfs, un-normalised. */
ORC_API void orc_preproc(const uint8_t* bottom, int w, int h, int channels, uint16_t* top, int outw, int outh,
                         int outcstep, int pad_top, int pad_left, int crop_x, int crop_y, uint16_t* alpha,
                         int alphaw, int alphah, int bgr)
{
    for (int gz = 0; gz < channels; gz++)
        for (int gy = 0; gy < outh; gy++)
            for (int gx = 0; gx < outw; gx++)
            {
                int x = gx + crop_x - pad_left; /* :56-57 */
                int y = gy + crop_y - pad_top;
                x = abs(x); /* :59-62 */
                y = abs(y);
                x = (w - 1) - abs(x - (w - 1));
                y = (h - 1) - abs(y - (h - 1));
                /* halo larger than the band (tiny images): the shader would index out of bounds; clamp */
                x = x < 0 ? 0 : x;
                y = y < 0 ? 0 : y;
                int v_offset = y * w + x;
                float v;
                if (bgr == 1 && gz != 3) v = (float)bottom[v_offset * channels + 2 - gz]; /* :69-72 */
                else v = (float)bottom[v_offset * channels + gz];
                if (gz == 3) /* :79-88 */
                {
                    int ax = gx - pad_left, ay = gy - pad_top;
                    if (alpha && ax >= 0 && ax < alphaw && ay >= 0 && ay < alphah) alpha[ay * alphaw + ax] = orc_f32_to_f16(v);
                }
                else
                {
                    const float norm_val = 1 / 255.f; /* :91-93 */
                    top[gz * outcstep + gy * outw + gx] = orc_f32_to_f16(v * norm_val);
                }
            }
}

/* store conversion shared by both postproc shaders (realsr_postproc.comp:71-78): v+0.5, floor,
   clamp to 0..255.  For negative v the GLSL uint(floor(v)) is undefined; follow the CPU path
   (saturate to 0) as SURVEY.md 8(a-4) prescribes. */
static inline uint8_t post_store(float v)
{
    v = v + 0.5f;
    float f = floorf(v);
    if (!(f > 0.f)) return 0;
    if (f > 255.f) return 255;
    return (uint8_t)f;
}

/* realsr_postproc.comp:47-89.  bottom = planar fp16 (3 * cstep), w x h tile (x4 scaled, with halo);
   top = u8 HWC band outw x outh x channels. */
ORC_API void orc_postproc(const uint16_t* bottom, int w, int h, int cstep, const uint16_t* alpha, int alphaw,
                          int alphah, uint8_t* top, int outw, int outh, int offset_x, int gx_max, int crop_x,
                          int crop_y, int channels, int bgr)
{
    (void)h;
    (void)alphah;
    for (int gz = 0; gz < channels; gz++)
        for (int gy = 0; gy < outh; gy++)
            for (int gx = 0; gx < gx_max; gx++)
            {
                float v;
                if (gz == 3) v = orc_f16_to_f32(alpha[gy * alphaw + gx]); /* :58-61 */
                else v = orc_f16_to_f32(bottom[gz * cstep + (gy + crop_y) * w + gx + crop_x]) * 255.f; /* :62-69 */
                int v_offset = gy * outw + gx + offset_x;
                uint8_t u = post_store(v);
                if (bgr == 1 && gz != 3) top[v_offset * channels + 2 - gz] = u;
                else top[v_offset * channels + gz] = u;
            }
}

/* realsr_preproc_tta.comp:54-113: same sample, scattered to 8 dihedral blobs.
   top[0..3] are outw x outh, top[4..7] are outh x outw (row stride outh). */
ORC_API void orc_preproc_tta(const uint8_t* bottom, int w, int h, int channels, uint16_t* const top[8], int outw,
                             int outh, int outcstep, int pad_top, int pad_left, int crop_x, int crop_y,
                             uint16_t* alpha, int alphaw, int alphah, int bgr)
{
    for (int gz = 0; gz < channels; gz++)
        for (int gy = 0; gy < outh; gy++)
            for (int gx = 0; gx < outw; gx++)
            {
                int x = gx + crop_x - pad_left;
                int y = gy + crop_y - pad_top;
                x = abs(x);
                y = abs(y);
                x = (w - 1) - abs(x - (w - 1));
                y = (h - 1) - abs(y - (h - 1));
                /* halo larger than the band (tiny images): the shader would index out of bounds; clamp */
                x = x < 0 ? 0 : x;
                y = y < 0 ? 0 : y;
                int v_offset = y * w + x;
                float v;
                if (bgr == 1 && gz != 3) v = (float)bottom[v_offset * channels + 2 - gz];
                else v = (float)bottom[v_offset * channels + gz];
                if (gz == 3)
                {
                    int ax = gx - pad_left, ay = gy - pad_top;
                    if (alpha && ax >= 0 && ax < alphaw && ay >= 0 && ay < alphah) alpha[ay * alphaw + ax] = orc_f32_to_f16(v);
                }
                else
                {
                    uint16_t hv = orc_f32_to_f16(v * (1 / 255.f));
                    int gzi = gz * outcstep;
                    top[0][gzi + gy * outw + gx] = hv;                                  /* :104 */
                    top[1][gzi + gy * outw + (outw - 1 - gx)] = hv;                     /* :105 */
                    top[2][gzi + (outh - 1 - gy) * outw + (outw - 1 - gx)] = hv;        /* :106 */
                    top[3][gzi + (outh - 1 - gy) * outw + gx] = hv;                     /* :107 */
                    top[4][gzi + gx * outh + gy] = hv;                                  /* :108 */
                    top[5][gzi + gx * outh + (outh - 1 - gy)] = hv;                     /* :109 */
                    top[6][gzi + (outw - 1 - gx) * outh + (outh - 1 - gy)] = hv;        /* :110 */
                    top[7][gzi + (outw - 1 - gx) * outh + gy] = hv;                     /* :111 */
                }
            }
}

/* realsr_postproc_tta.comp:54-110 */
ORC_API void orc_postproc_tta(const uint16_t* const bottom[8], int w, int h, int cstep, const uint16_t* alpha,
                              int alphaw, int alphah, uint8_t* top, int outw, int outh, int offset_x, int gx_max,
                              int crop_x, int crop_y, int channels, int bgr)
{
    (void)alphah;
    for (int gz = 0; gz < channels; gz++)
        for (int gy = 0; gy < outh; gy++)
            for (int gx = 0; gx < gx_max; gx++)
            {
                float v;
                if (gz == 3) v = orc_f16_to_f32(alpha[gy * alphaw + gx]);
                else
                {
                    int gzi = gz * cstep;
                    int sy = gy + crop_y, sx = gx + crop_x;
                    float v0 = orc_f16_to_f32(bottom[0][gzi + sy * w + sx]);                     /* :76 */
                    float v1 = orc_f16_to_f32(bottom[1][gzi + sy * w + (w - 1 - sx)]);           /* :77 */
                    float v2 = orc_f16_to_f32(bottom[2][gzi + (h - 1 - sy) * w + (w - 1 - sx)]); /* :78 */
                    float v3 = orc_f16_to_f32(bottom[3][gzi + (h - 1 - sy) * w + sx]);           /* :79 */
                    float v4 = orc_f16_to_f32(bottom[4][gzi + sx * h + sy]);                     /* :80 */
                    float v5 = orc_f16_to_f32(bottom[5][gzi + sx * h + (h - 1 - sy)]);           /* :81 */
                    float v6 = orc_f16_to_f32(bottom[6][gzi + (w - 1 - sx) * h + (h - 1 - sy)]); /* :82 */
                    float v7 = orc_f16_to_f32(bottom[7][gzi + (w - 1 - sx) * h + sy]);           /* :83 */
                    v = (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7) * 0.125f;                        /* :85 */
                    v = v * 255.f;
                }
                int v_offset = gy * outw + gx + offset_x;
                uint8_t u = post_store(v);
                if (bgr == 1 && gz != 3) top[v_offset * channels + 2 - gz] = u;
                else top[v_offset * channels + gz] = u;
            }
}

ORC_API int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORC_API void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
