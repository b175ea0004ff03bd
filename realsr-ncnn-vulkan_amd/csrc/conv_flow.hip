// conv_flow.hip -- conv3x3_flow: the 3x3 convolutions of x4.param as MFMA implicit GEMM on 16-channel planes,
// streamed through LDS in 16-channel HALF-STAGES (gfx950 / MI355X only).
//
// Why a second generation (round-1 kernels: conv3x3_ring / conv3x3_pipe in kernels.hip): s_memtime traces of those
// showed the matrix pipe idle for (a) the fragment refill after every stage barrier, (b) LDS-DMA data that had not
// landed (64-output-channel convs: one 77-KB stage of look-ahead < the ~6,000-cycle DMA latency under load) and
// (c) the epilogue, which is bound by the ~7-14 B/clk/CU at which a CU can issue global stores and used the patch
// slot as transpose scratch (so the slot could not be refilled meanwhile).  This kernel removes all three:
//
//   * K is walked in units of ONE 16-channel plane (a "half-stage": 36 MFMAs per 32 output channels).  A half-stage
//     is a 20-KB patch + a 9-KB (x NT) weight image, so 6 (NT=1) / 4 (NT=2) patches and 3 weight images ride in a
//     ring: 5 / 3 half-stages of look-ahead instead of 2 / 1 stages.
//   * activations live in HBM as 16-channel planes [H][W][16] fp16 (32 B per pixel), so every LDS-DMA piece and
//     every epilogue store is 1 KiB of whole cache lines; one v_mfma_f32_32x32x16_f16 consumes exactly one plane.
//   * the MFMA waves never drain: operand fragments are reloaded IN PLACE, one (dx) step ahead, across half-stage
//     and block boundaries; the 12 (dy, row) cells of a step run in anti-diagonal order so that fragment X[j] is dead
//     after diagonal j and its reload has >= 10 MFMA slots to land.  The single s_barrier per half-stage sits in the
//     middle of the MFMA stream, after the wave's last LDS read of the half-stage ("early barrier": the slot is
//     handed back to the loaders a third of a half-stage before its MFMAs finish).
//   * epilogue: bias enters as the C operand of the block's first MFMAs; finished rows go fp32 -> (scale / LeakyReLU)
//     -> fp16 -> 1-KiB buffer stores (rounds 1-3: through a private LDS transpose scratch; round 4: straight from the
//     registers, see below).  With 32 output channels (4 MFMA waves, 256 VGPRs) the accumulators are double-buffered and
//     block r is drained row by row underneath block r+1's MFMAs.
//
// Round 4: NO TRANSPOSE in the epilogue.  The rows of the weight images are ordered (model.h row_cout) so that the registers of
// an MFMA result are runs of 8 consecutive output channels: a lane's 16 bytes of a plane row go to memory as they are (one
// buffer_store_b128 per plane), the 4 ds_write_b64 + 2 ds_read_b128 per row and their waits are gone, and the deferred drain
// (40 pieces instead of 60) rides behind every THIRD MFMA cell: -1.3 % frame time (profiles/r04_ab_epilogue.txt).
//
// Round 5: FOLDED blocks (kernels.h kFoldBit) -- a tile's last block column, when only 1..14 pixels wide (420 = 13 x 32 + 4), runs as one
// 16 x 32 block per TWO block rows: strip 1 in patch columns / lanes 0..15, strip 2 (16 rows further down) in 16..31.  The matrix loop
// is untouched; the loaders' gather (block_offsets) and the epilogue's scatter (make_out, store3) are what know.  The experiment
// scaffolding of rounds 2-4 (ablation bits, s_memtime instrumentation, the Winograd instruction-mix study) now compiles only with
// -DRSR_EXPERIMENT, and the LDS scratch the round 1-3 epilogue transposed through is gone from the LDS map.
//
// Round 3: weight images LDS-RESIDENT for the whole launch where they fit (WRES: every conv but the 192 -> 64
// ones), MFMA waves SKIP blocks whose four rows lie below the tile or inside the frame of output pixels nothing kept depends on
// (wave_is_dead, ConvArgs::margin), and conv_last runs with (dy, cout) in the MFMA's M dimension (EPI 3).
//
// Work decomposition as before: one workgroup = a 16 x 32 pixel block, MFMA wave w owns rows 4*(w&3)..+3, persistent
// grid (one workgroup per CU) walking an XCD-contiguous, strided list of work items; 4 loader waves only issue
// LDS-DMA (global_load_lds_dwordx4).  Work-item descriptors are fetched with SCALAR loads (they neither touch the
// loaders' vmcnt bookkeeping nor need LDS), so a launch never has to be split.
#include "kernels.h"
#include "model.h" // row_cout(): the order of the output channels in the rows of a weight image

namespace rsr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef f32x4 __attribute__((may_alias)) f32x4_lds;

namespace {

constexpr int kFPx = 32;                     // bytes per pixel of a 16-channel plane
constexpr int kFRow = kPatchW * kFPx;        // 1088 B per patch row
constexpr int kFPatch = 20 * 1024;           // LDS bytes per patch slot: 20 one-KiB DMA pieces (612 px * 32 B = 19,584 used)
constexpr int kFPieces = 20;
constexpr int kFItems = kPatchPx * 2;        // 16-byte items in a patch (1224)

template <int NT>
struct FlowCfg
{
    static constexpr int PR = NT == 1 ? 6 : 4; // patch ring depth (half-stages)
#ifndef RSR_STREAM_WR
#define RSR_STREAM_WR 3 // (4 = the LDS the transpose scratch used to hold as a fourth weight slot for conv5: A/B'ed in round 5, profiles/r05_ab_lds.txt)
#endif
    static constexpr int WR = RSR_STREAM_WR;   // weight ring depth of a streamed-weight launch
    static constexpr int WB = NT * 9 * 32 * 32; // bytes of one weight image [9 taps][NT*32 cout][16 cin]
    static constexpr int WPIECES = NT * 9;
    static constexpr int W_OFF = PR * kFPatch;
    static constexpr int BIAS_OFF = W_OFF + WR * WB; // (rounds 1-3 kept an 8 / 16 KB epilogue transpose scratch in front of the bias: gone, round 5)
    static constexpr int TOTAL = BIAS_OFF + NT * 128;
};
static_assert(FlowCfg<1>::TOTAL <= 160 * 1024 && FlowCfg<2>::TOTAL <= 160 * 1024, "LDS budget");

__device__ __forceinline__ const char* plane_ptr(const PlaneSrc& s, int slot, int plane)
{
    return static_cast<const char*>(s.base) + (long long)slot * s.slot_stride + (long long)plane * s.plane_stride;
}

// Work-item descriptor through the scalar cache (constant address space => s_load_dwordx8 for a uniform index).
__device__ __forceinline__ WorkItem load_item(const WorkItem* items, int idx)
{
    typedef const __attribute__((address_space(4))) i32x8* cptr_t;
    const i32x8 v = *(cptr_t)(unsigned long long)(items + idx);
    WorkItem it;
    it.slot = __builtin_amdgcn_readfirstlane(v[0]);
    it.y0 = __builtin_amdgcn_readfirstlane(v[1]);
    it.x0 = __builtin_amdgcn_readfirstlane(v[2]);
    it.H = __builtin_amdgcn_readfirstlane(v[3]);
    it.W = __builtin_amdgcn_readfirstlane(v[4]);
    it.pad0 = __builtin_amdgcn_readfirstlane(v[5]);
    it.pad1 = __builtin_amdgcn_readfirstlane(v[6]);
    it.pad2 = __builtin_amdgcn_readfirstlane(v[7]);
    return it;
}

// A pointer the compiler can PROVE wave-uniform (buffer resources built from it need no waterfall loop).  readfirstlane
// returns a signed int: widen through unsigned, or a low half with bit 31 set sign-extends into the high half.
__device__ __forceinline__ char* uniform_ptr(const void* p)
{
    const unsigned long long b = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return reinterpret_cast<char*>((unsigned long long)lo | ((unsigned long long)hi << 32));
}

// two bf8 (e5m2) bytes of a dword -> fp32 / two fp32 -> two bf8 bytes of a dword (the word selector of v_cvt_pk_* is an immediate)
__device__ __forceinline__ f32x2 bf8_pair(unsigned w, int hiword)
{
    return hiword ? __builtin_amdgcn_cvt_pk_f32_bf8(int(w), true) : __builtin_amdgcn_cvt_pk_f32_bf8(int(w), false);
}
__device__ __forceinline__ int pack_bf8_pair(float x, float y, int old, int hiword)
{
    return hiword ? __builtin_amdgcn_cvt_pk_bf8_f32(x, y, old, true) : __builtin_amdgcn_cvt_pk_bf8_f32(x, y, old, false);
}

#define RSR_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define RSR_GLB(p) ((const __attribute__((address_space(1))) void*)(p))
#define RSR_MFMA 0x8
#define RSR_DSR 0x100
#define RSR_SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

// Experiment scaffolding is compiled OUT of the product build.  -DRSR_EXPERIMENT (tools/build_variant.sh) brings back
// the run-time ablation bits of ConvArgs::dbg -- 1 no LDS-DMA at all, 4 no epilogue stores, 64 no weight DMA, 128 no patch DMA (stale
// LDS is multiplied: wrong results by construction) -- and admits the instrumented builds RSR_FLOW_LIFE / RSR_FLOW_TRACE (s_memtime
// stamps) and RSR_EXP_THIN (the Winograd instruction-mix study of DESIGN.md 4.1).  dbg bit 32 ("compute the rows a wave would skip")
// is a TEST hook with correct results and stays in every build.
#ifdef RSR_EXPERIMENT
#define RSR_ABL(bits) (a.dbg & (bits))
#else
#define RSR_ABL(bits) 0
#if defined(RSR_FLOW_LIFE) || defined(RSR_FLOW_TRACE) || defined(RSR_EXP_THIN)
#error "RSR_FLOW_LIFE / RSR_FLOW_TRACE / RSR_EXP_THIN are experiment builds: add -DRSR_EXPERIMENT"
#endif
#endif

} // namespace

// EPI: 0 generic (conv_last: planar fp16 [3][H][W] output)      1 v = act(acc)  -> fp16 planes      3 conv_last with (dy, cout) in M
//      2 v = s1*acc (the conv's own input rides in the accumulator as an identity tap) [, v = s2*v + r2] -> fp16 planes
//      4 / 5 the PRECISE forms of 2 without / with the second residual (engine option "precise"; 64 output channels): the residual
//        stream is kept as hi + lo / 2048, hi an fp16 plane as ever and lo ONE BYTE per element (bf8 = e5m2: the upper byte of an
//        fp16; kernels.h ConvArgs::precise) -- v = s1*acc + lo1/2048 [, v = s2*v + r2 + lo2/2048] in fp32, then ONE rounding:
//        hi = fp16(v) -> out16 (what the next convs read), lo = bf8((v - hi) * 2048) -> the output's lo planes (16 B per pixel).  The reference's GPU path rounds the trunk 92 times on its way
//        through the 23 RRDBs (fp16 storage, realsr.cpp:44-46); this halves the engine's distance to the fp32 CPU path
//        (realsr.cpp:525-838; profiles/r06_storage_emulation.txt).  Absent lo planes are read through a null buffer resource
//        (zeros, no memory access) and written into one: the epilogue is branch-free.
//      6 / 7 = 3 / 0 in precise mode: conv_last's fp32 result goes to the uint8 conversion (or a planar fp32 blob) without the fp16
//        rounding of the reference's `output` blob in between.
// NTW: n-tiles (32 output channels) per MFMA wave; the workgroup has 4*NT/NTW MFMA waves + 4 loader waves.
// DEFER: double-buffered accumulators, block r drained underneath block r+1 (NT == 1 only).
// WRES: the conv's weight images stay RESIDENT in LDS for the whole launch (loaded once per workgroup; slot = plane index, the
//       "ring" is never refilled) instead of being re-streamed from L2 for every block.  Why: the dense-block convs run at a
//       constant ~25 GB/s per CU of vector-memory traffic (LDS-DMA loads + epilogue stores: 148 / 206 / 264 / 322 KiB per block
//       for cin 64 / 96 / 128 / 160 in 121 / 159 / 209 / 253 us per launch, round 3) whatever their FLOP count, and a build with
//       2.25x fewer MFMAs gains only 13 % there -- they are bound by the CU's vector-memory path, of which the re-streamed
//       weights are 24-28 % (47 % of the loads of a 64 -> 64 conv).  The patch ring gets what LDS is left (ConvArgs::pr >= 3).
template <int NT, int NTW, bool UPS, int EPI, bool DEFER, bool WRES>
__global__ __launch_bounds__((4 * NT / NTW + 4) * 64, (4 * NT / NTW + 4) / 4) void conv3x3_flow(const ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef FlowCfg<NT> C;
    constexpr int MW = 4 * NT / NTW;
    constexpr int WB = C::WB;
    constexpr bool WDB = (NTW == 1); // double-buffered weight fragments (registers to spare with one n-tile per wave)
    static_assert(!DEFER || (NT == 1 && NTW == 1), "deferred epilogue needs the 256-VGPR budget of the 8-wave workgroup");
    constexpr bool LAST3 = (EPI == 3 || EPI == 6); // conv_last with (dy, cout) in M
    constexpr bool LASTG = (EPI == 0 || EPI == 7); // conv_last through the generic path
    constexpr bool OUT32 = (EPI == 6 || EPI == 7); // ... in precise mode: its fp32 result is what is converted to uint8 / stored (planar fp32)
    static_assert(!LAST3 || (NT == 1 && NTW == 1 && !UPS && !DEFER && WRES), "conv_last's (dy, cout) layout: 32 rows, resident aux image");
    static_assert((EPI != 4 && EPI != 5) || (NT == 2 && NTW == 1 && !UPS && !DEFER), "the precise residual epilogue exists for the 64-output-channel convs of the trunk");
    constexpr bool RESID = (EPI == 2 || EPI == 4 || EPI == 5); // residual forms
    constexpr bool PREC = (EPI == 4 || EPI == 5);              // ... with the hi + lo residual stream

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef RSR_FLOW_LIFE // experiment builds only (tools/flow_life.py): per-workgroup s_memtime stamps -- entry, first half-stage in LDS, exit
    const unsigned long long life_t0 = __builtin_amdgcn_s_memtime();
#define RSR_LIFE(SLOT) if (a.trace && tid == 0) a.trace[blockIdx.x * 4 + (SLOT)] = __builtin_amdgcn_s_memtime();
#else
#define RSR_LIFE(SLOT)
#endif
    const int nst = a.n0 + a.n1; // half-stages per block (16-channel planes), even by construction (engine pads)
    // ring depths and LDS offsets: compile-time constants when the weights are streamed
    const int PR = WRES ? a.pr : C::PR, WR = WRES ? nst : C::WR;
    const int kWOff = PR * kFPatch, kBiasOff = kWOff + WR * (LAST3 ? a.wpieces * 1024 : WB);

    const int per = (a.nitems + 7) >> 3;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, nj = gridDim.x >> 3;
    const int first = xcd * per + j;
    const int end = min((xcd + 1) * per, a.nitems);
    const int nmine = first < end ? (end - first + nj - 1) / nj : 0;
    if (nmine == 0) return;
    const int S = nmine * nst;

    // Launch prologue (measured, round 4: tools/flow_life.py, profiles/r04_ab_prologue.txt): 5 - 7 us from kernel entry to the first
    // half-stage in LDS -- this bias round trip, a cold descriptor load, then every weight image ahead of the first patch.  A build
    // that brought the bias in by LDS-DMA (no barrier here), issued W(0) ahead of the descriptor and interleaved the other images
    // with the first block's patches cut 1.5 - 2.6 us of it per launch and did NOT shorten the frame (+0.3 %): under the 1,400 W cap
    // the frame is bound by energy, idle microseconds are not the resource.  Kept as it was.
    if (tid < NT * 32) reinterpret_cast<float*>(smem + kBiasOff)[tid] = a.bias[tid];
    __syncthreads();

    if (wave >= MW)
    {
        // ================================ loader waves ================================
        // Half-stage u = patch P(u) in slot u % PR + weight image W(u) in slot u % WR.  After barrier E_t the MFMA waves
        // have finished READING half-stage t-1, so the loaders issue W(t+WR-1) and P(t+PR-1), in that order; before
        // E_t they wait until P(t) and W(t) have landed.  Vector loads retire in order, so with
        //     ... W(t) P(t+PR-WR) | W(t+1) P(t+PR-WR+1) | ... | W(t+WR-2) P(t+PR-2)      <- issue order, newest right
        // "s_waitcnt vmcnt(nP + (WR-2)*(nW+nP))" is exactly "W(t) and everything older has landed" (nP, nW = this
        // wave's pieces per patch / weight image).  Near the end of the work list the tail is shorter: drain fully.
        const int lw = wave - MW;
        constexpr int NP = kFPieces / 4; // 5 patch pieces per loader wave
        const int nW = (C::WPIECES - lw + 3) / 4;
        int pr[NP], pc[NP], pxor[NP];
        bool pvalid[NP];
#pragma unroll
        for (int i = 0; i < NP; i++)
        {
            const int jj = (lw + 4 * i) * 64 + lane; // 16-byte item of the patch image, LDS-linear
            const int px = jj >> 1;
            pr[i] = px / kPatchW;
            pc[i] = px - pr[i] * kPatchW;
            // LDS item jj must receive logical 16-B slot (jj & 1) ^ f(column), f(c) = (c >> 3) & 1 (read-side swizzle)
            pxor[i] = ((jj & 1) ^ ((pc[i] >> 3) & 1)) << 4;
            pvalid[i] = jj < kFItems;
        }
        const char* wsrc_lane = static_cast<const char*>(a.wpk16) + lw * 1024 + lane * 16;

        int uP = 0, uW = 0, sP = 0, sW = 0, ckP = 0, ckW = 0, rP = 0;
        WorkItem itP = load_item(a.items, first);
        unsigned srcoff[NP];
        // A FOLDED block (WorkItem::x0 & kFoldBit; kernels.h): the last, <= 14 pixel wide column of a tile, two block rows at once.
        // Patch columns 0..15 hold the halo'd strip of rows y0.. (left halo, <= 14 pixels, right halo), columns 16..31 the strip
        // of rows y0 + 16..; the MFMA lanes read the patch exactly as they always do (pixel l32 + dx), so lanes 0..13 compute the
        // first strip and lanes 16..29 the second -- only the loaders' source addresses and the epilogue's destinations differ.
        auto block_offsets = [&]() {
            const int H = itP.H, W = itP.W, Wi = UPS ? (W >> 1) : W;
            const bool fold = (itP.x0 & kFoldBit) != 0;
            const int bx0 = itP.x0 & (kFoldBit - 1);
#pragma unroll
            for (int i = 0; i < NP; i++)
            {
                const int half = (fold && pc[i] >= 16) ? 16 : 0;
                const int gy = itP.y0 - 1 + pr[i] + half, gx = bx0 - 1 + pc[i] - half;
                const bool ok = pvalid[i] && gy >= 0 && gy < H && gx >= 0 && gx < W;
                const int sy = UPS ? (gy >> 1) : gy, sx = UPS ? (gx >> 1) : gx;
                srcoff[i] = ok ? unsigned(kGuard + (sy * Wi + sx) * kFPx + pxor[i]) : 0u; // 0 = the plane's zero guard
            }
        };
        block_offsets();
        auto issueP = [&]() {
            if (uP >= S) return;
            const char* gbase = ((ckP < a.n0) ? plane_ptr(a.src0, itP.slot, ckP) : plane_ptr(a.src1, itP.slot, ckP - a.n0)) - kGuard;
            char* dst = smem + sP * kFPatch + lw * 1024;
            if (!RSR_ABL(1 | 128)) // ablation (experiment builds): 1 = no LDS-DMA at all, 128 = no patch DMA, 64 = no weight DMA (stale LDS is multiplied)
            {
#pragma unroll
                for (int i = 0; i < NP; i++)
                    __builtin_amdgcn_global_load_lds(RSR_GLB(gbase + srcoff[i]), RSR_LDS(dst + i * 4096), 16, 0, 0);
            }
            sP = sP == PR - 1 ? 0 : sP + 1;
            uP++;
            if (++ckP == nst)
            {
                ckP = 0;
                if (++rP < nmine)
                {
                    itP = load_item(a.items, first + rP * nj);
                    block_offsets();
                }
            }
        };
        auto issueW = [&]() {
            if (uW >= S) return;
            const char* src = wsrc_lane + (long long)ckW * WB;
            char* dst = smem + kWOff + sW * WB + lw * 1024;
            if (!RSR_ABL(1 | 64))
            {
#pragma unroll
                for (int i = 0; i < (C::WPIECES + 3) / 4; i++)
                    if (i < nW)
                        __builtin_amdgcn_global_load_lds(RSR_GLB(src + i * 4096), RSR_LDS(dst + i * 4096), 16, 0, 0);
            }
            sW = sW == WR - 1 ? 0 : sW + 1;
            uW++;
            ckW = ckW + 1 == nst ? 0 : ckW + 1;
        };
        if (WRES)
        {
            // every weight image of the conv, once, ahead of the first patch: vector loads retire in order, so "P(0) has landed"
            // implies "all weights have landed"
            const int npieces = nst * (LAST3 ? a.wpieces : C::WPIECES);
            if (!RSR_ABL(1 | 64))
                for (int p_ = lw; p_ < npieces; p_ += 4)
                    __builtin_amdgcn_global_load_lds(RSR_GLB(static_cast<const char*>(a.wpk16) + p_ * 1024 + lane * 16), RSR_LDS(smem + kWOff + p_ * 1024), 16, 0, 0);
            for (int tt = -(PR - 1); tt < 0; tt++) issueP();
            // before E_t: P(t) and everything older has landed <=> at most the (PR - 2) newer patches are in flight
            for (int t = 0; t < S; t++)
            {
                if (RSR_ABL(1) || t + PR - 2 >= S) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                else if (PR == 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(1 * NP) : "memory");
                else if (PR == 4) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * NP) : "memory");
                else if (PR == 5) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(3 * NP) : "memory");
                else if (PR == 6) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * NP) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                issueP();
            }
        }
        else
        {
            for (int tt = -(PR - 1); tt < 0; tt++)
            {
                if (tt + WR - 1 >= 0) issueW();
                issueP();
            }
            // counted wait immediates: nP + (WR-2)*(nW+nP) with nP = 5
            constexpr int WRm2 = C::WR - 2;
            constexpr int NW_HI = (C::WPIECES + 3) / 4, NW_LO = C::WPIECES / 4; // loader waves own NW_HI or NW_LO weight pieces
            for (int t = 0; t < S; t++)
            {
                if (RSR_ABL(1) || t + PR - 2 >= S) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                else if (RSR_ABL(64)) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((WRm2 + 1) * NP) : "memory"); // ablations: the same look-ahead
                else if (RSR_ABL(128) && nW == NW_HI) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(WRm2 * NW_HI) : "memory");
                else if (RSR_ABL(128)) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(WRm2 * NW_LO) : "memory");
                else if (nW == NW_HI) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((WRm2 + 1) * NP + WRm2 * NW_HI) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((WRm2 + 1) * NP + WRm2 * NW_LO) : "memory");
                issueW();
                issueP();
            }
        }
        asm volatile("s_barrier" ::: "memory"); // E_S: the MFMA waves pass one barrier per half-stage, also in the last one
        return;
    }

    // ================================ MFMA waves ================================
    const int l32 = lane & 31, hi = lane >> 5;
    const int wrow = wave & 3, ntw0 = (wave >> 2) * NTW;
    // LDS fragment addresses.  X: patch pixel (r, c) at r*1088 + c*32, its two 16-B slots (k = 0..7 | 8..15) swapped when
    // (c >> 3) & 1: the 16 lanes of a ds_read_b128 group then cover 16 distinct 16-B bank groups for every tap shift.
    // One VGPR per dx + immediate row offsets address all fragments.  W: row = tap*NT*32 + n, slots swapped when (n>>3)&1.
    int xcol[3];
#pragma unroll
    for (int dx = 0; dx < 3; dx++)
    {
        const int c = l32 + dx;
        xcol[dx] = wrow * 4 * kFRow + c * kFPx + ((hi ^ ((c >> 3) & 1)) << 4);
    }
    const int woff = kWOff + (ntw0 * 32 + l32) * 32 + ((hi ^ ((l32 >> 3) & 1)) << 4);

    // bias in accumulator layout (lane (px, hi), reg q*4+e -> cout q*8 + hi*4 + e): C operand of a block's first MFMAs
    // Without the deferred epilogue the registers are re-read from LDS at the end of every epilogue (live only up to the
    // next block's first step): the 3-waves-per-SIMD workgroup has 168 VGPRs and the epilogue needs them.
    f32x16 bias16[NTW];
    auto load_bias = [&]() {
#pragma unroll
        for (int n = 0; n < NTW; n++)
#pragma unroll
            for (int q = 0; q < 4; q++)
            {
                const f32x4 b4 = *reinterpret_cast<const f32x4_lds*>(smem + kBiasOff + ((ntw0 + n) * 32 + q * 8 + hi * 4) * 4);
#pragma unroll
                for (int e = 0; e < 4; e++) bias16[n][q * 4 + e] = b4[e];
            }
    };
    load_bias();

    // Epilogue lane map.  The weight images carry their rows in the order row_cout() (model.h): register q*4+e of lane (px, hi)
    // = row q*8 + hi*4 + e of the MFMA result = output channel (q>>1)*16 + hi*8 + (q&1)*4 + e, i.e. registers 0..7 are the 8
    // consecutive channels hi*8.. of the n-tile's first plane, 8..15 those of its second: converted to fp16 they are the 16
    // bytes at pixel*32 + hi*16 of the plane row -- one buffer_store_b128 per plane, no transpose through LDS (rounds 1-3 went
    // accumulator -> ds_write_b64 x4 -> ds_read_b128 x2 -> store per row; the 64 lanes of a store cover the same 1 KiB either
    // way, and the (pixel, half) lane order stores as fast as the pixel-pair order: tools/ubench/store_pattern.hip).
    const float slope = a.lrelu ? 0.2f : 1.f;
    const bool idt = RESID && a.res1_in_acc; // a fetched first residual arrives as res2 with s2 = 1 (launch_conv_flow)
    const bool has2 = EPI == 5 || ((EPI == 2) && a.res2_kind == 1);
#ifdef RSR_FLOW_TRACE // experiment builds only: an s_memtime in the loop forces every lgkmcnt wait to 0 (SMEM returns out of order)
    const bool tracing = a.trace && blockIdx.x == 0 && wave == 0;
#else
    constexpr bool tracing = false;
#endif

    float pinf0 = __builtin_inff();
    asm volatile("" : "+s"(pinf0)); // opaque +inf: see the deferred epilogue

    // ---- epilogue pieces ----
    struct OutDesc
    {
        char* base; // plane 2*ntw0 of the output slot
        int live;   // 0: stores are dropped (null resource)
        int voff;   // lane's byte offset inside a plane row (+ 16 rows for the second strip of a folded block), or the out-of-range sentinel
        int y0, H, W, slot;
        unsigned lim; // bytes of one plane of this tile: H * W * 32
    };
    auto make_out = [&](const WorkItem& it, bool live) {
        OutDesc o;
        o.base = const_cast<char*>(plane_ptr(a.out16, it.slot, ntw0 * 2));
        o.live = (live && !RSR_ABL(4)) ? 1 : 0;
        // lane -> pixel: column l32 of the block; in a FOLDED block column l32 & 15 of the strip l32 >> 4 (16 rows further down)
        const bool fold = (it.x0 & kFoldBit) != 0;
        const int x = (it.x0 & (kFoldBit - 1)) + (fold ? (l32 & 15) : l32);
        o.voff = x < it.W ? x * kFPx + hi * 16 + ((fold && l32 >= 16) ? 16 * it.W * kFPx : 0) : int(0x80000000u);
        o.y0 = it.y0 + wrow * 4;
        o.H = it.H;
        o.W = it.W;
        o.slot = it.slot;
        // every store / residual load of the block is range-checked against the END OF ITS PLANE (+ the plane's offset, which rides
        // in the VGPR offset): rows below the tile -- of either strip of a folded block -- fall out by themselves
        o.lim = unsigned(it.H) * unsigned(it.W * kFPx);
        return o;
    };
    // one finished row (32 px x 32 cout of an n-tile): scale / LeakyReLU, fp16; tq[p] = this lane's 16 bytes of output plane p
    auto row_pack = [&](const f32x16& acc, u32x4 (&tq)[2]) {
#pragma unroll
        for (int p = 0; p < 2; p++)
        {
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; e++)
            {
                float v = acc[p * 8 + e];
                if (EPI == 2)
                {
                    v = v * a.s1;
                    asm volatile("" : "+v"(v)); // see row_emit
                }
                else v = __builtin_amdgcn_fmed3f(v, v * slope, pinf0); // = max(v, slope*v), one instruction
                o[e] = (_Float16)v;
            }
            tq[p] = __builtin_bit_cast(u32x4, o);
        }
    };
    // residual stage (a lane holds 8 consecutive channels of one pixel, like the residual's plane rows): v = v*s2 + r2
    // The second residual (the RRDB input, every third conv5) is fetched ahead of its use when the registers allow it (one
    // n-tile per wave) -- rows 0-1 one half-stage before the epilogue, rows 2-3 into the same registers once the
    // epilogue is through with rows 0-1: four dependent HBM round trips per block otherwise.
    constexpr bool PRE2 = (EPI == 2 || EPI == 4) && NTW == 1; // (EPI 5 fetches lo rows at the start of its epilogue anyway: nothing to gain, 16 VGPRs to lose)
    u32x4 r2q[PRE2 ? 2 : 1][NTW][2]; // rows 0-1, then rows 2-3
#ifndef RSR_PREC_PRE
#define RSR_PREC_PRE 1 // EPI 4: lo rows of residual 1 fetched one half-stage ahead of the epilogue (the rest at its start).  1: no spill; 2 spills two dwords and measures the same, 3 / 4 spill a row (profiles/r06_precise_cost.txt)
#endif
    u32x4 l1q[EPI == 4 ? 4 : 1];     // EPI 4: the four lo rows of residual 1
    // Out-of-image lanes / rows read zeros through the buffer range check, like row_store drops them.
    auto res2_row = [&](u32x4 (&dst)[2], const OutDesc& o, int rr, int n) {
        const int y = o.y0 + rr;
        char* ub = uniform_ptr(const_cast<char*>(plane_ptr(a.res2, o.slot, ntw0 * 2)));
        const unsigned pstride = unsigned(a.res2.plane_stride);
#pragma unroll
        for (int p = 0; p < 2; p++)
        {
            const unsigned poff = unsigned(n * 2 + p) * pstride;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ub, 0, int(o.lim + poff), 0x00020000);
            dst[p] = __builtin_amdgcn_raw_buffer_load_b128(rs, o.voff, int(unsigned(y) * unsigned(o.W * kFPx) + poff), 0);
        }
    };
    // the same fetch from the planes `off` bytes behind those of `s` (EPI 4 / 5: the lo planes of a residual; off == 0: there are none,
    // the null resource returns zeros)
    // The lo bytes of an n-tile's TWO planes share one "pair plane" with the geometry of a hi plane (32 B per pixel): lane (pixel, half)
    // owns the 16 bytes at pixel * 32 + half * 16 = [its 8 channels of plane 0 | its 8 channels of plane 1] -- the hi addressing with
    // the n-tile in the place of the plane, one 1-KiB load / store per row and n-tile (the layout is private to these epilogues).
    auto lo_row = [&](const PlaneSrc& s, long long off, u32x4& dst, const OutDesc& o, int rr, int n) {
        const int y = o.y0 + rr;
        const unsigned pstride = unsigned(s.plane_stride);
        char* ub = uniform_ptr(const_cast<char*>(plane_ptr(s, o.slot, 0)) + off + (long long)(ntw0 + n) * pstride);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ub, 0, off ? int(o.lim) : 0, 0x00020000);
        dst = __builtin_amdgcn_raw_buffer_load_b128(rs, o.voff, int(unsigned(y) * unsigned(o.W * kFPx)), 0);
    };
    auto res2_prefetch = [&](const WorkItem& w, int row0) { // rows row0, row0 + 1
        if (!PRE2) return;
        if (EPI == 4)
        { // no second residual: the prefetch brings ALL FOUR rows of the first residual's lo planes (16 VGPRs, what rows 0-1 of a second
          // residual cost EPI 2)
            if (row0 != 0) return;
            const OutDesc o = make_out(w, true);
#pragma unroll
            for (int rr = 0; rr < RSR_PREC_PRE; rr++) lo_row(a.res1, a.lo1_off, l1q[EPI == 4 ? rr : 0], o, rr, 0);
            __builtin_amdgcn_sched_barrier(0);
            return;
        }
        if (!has2) return;
        const OutDesc o = make_out(w, true);
#pragma unroll
        for (int rr = 0; rr < 2; rr++)
#pragma unroll
            for (int n = 0; n < NTW; n++) res2_row(r2q[PRE2 ? rr : 0][n], o, row0 + rr, n);
        __builtin_amdgcn_sched_barrier(0);
    };
    // 1 KiB per store instruction: 32 pixels x 32 B of one 16-channel plane row; rows / columns outside the image are
    // dropped by the buffer range check (null resource / out-of-range offset), so the epilogue is branch-free
    auto row_store = [&](const u32x4 (&tq)[2], const OutDesc& o, int rr, int n) {
        const int y = o.y0 + rr;
        char* ub = uniform_ptr(o.base);
        const unsigned pstride = unsigned(a.out16.plane_stride);
        // The row / plane offset travels in the VGPR offset, soffset = 0: gfx950 needs 2 wait states between a > 64-bit
        // buffer store and a VALU write of its data registers ALSO when soffset is an SGPR, but hipcc only pads the
        // soffset-less form (seen as garbage in dword 0 of lanes 12-15 of every 16 -- the lanes whose data is read last --
        // whenever the next row's arithmetic reused the registers right behind the store).  Out-of-image lanes keep bit 31.
#pragma unroll
        for (int p = 0; p < 2; p++)
        {
            const unsigned poff = unsigned(n * 2 + p) * pstride;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ub, 0, o.live ? int(o.lim + poff) : 0, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(tq[p], rs, o.voff + int(unsigned(y) * unsigned(o.W * kFPx) + poff), 0, 0);
        }
    };
    // the inline epilogue of one row: plane by plane (pack -> residual -> store), so that only one 16-byte value is live at a time
    // (the 168-VGPR kernels have no room for a whole packed row next to the prefetched residual)
    auto row_emit = [&](const f32x16& acc, const OutDesc& o, int rr, int n) {
        const int y = o.y0 + rr;
        char* ub = uniform_ptr(o.base);
        const unsigned pstride = unsigned(a.out16.plane_stride);
        u32x4 r2x[2];
        if (EPI == 2 && has2 && !PRE2) res2_row(r2x, o, rr, n);
#pragma unroll
        for (int p = 0; p < 2; p++)
        {
            half8 v;
#pragma unroll
            for (int e = 0; e < 8; e++)
            {
                float f = acc[p * 8 + e];
                if (EPI == 2)
                {
                    f = f * a.s1;
                    // the product is rounded to fp32 BEFORE it becomes an fp16 (as in rounds 1-3): left visible, the compiler fuses
                    // multiply and conversion into one v_fma_mix with a single rounding in some instantiations and not in others
                    // (1 ulp on 1e-4 of the values, seen as a changed frame checksum) -- the kernel variants must agree bit for bit
                    asm volatile("" : "+v"(f));
                }
                else f = __builtin_amdgcn_fmed3f(f, f * slope, pinf0);
                v[e] = (_Float16)f;
            }
            if (EPI == 2 && has2)
            {
                const half8 r2 = __builtin_bit_cast(half8, PRE2 ? r2q[PRE2 ? (rr & 1) : 0][n][p] : r2x[p]);
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = (_Float16)((float)v[e] * a.s2 + (float)r2[e]);
            }
            const unsigned poff = unsigned(n * 2 + p) * pstride;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ub, 0, o.live ? int(o.lim + poff) : 0, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, o.voff + int(unsigned(y) * unsigned(o.W * kFPx) + poff), 0, 0);
            __builtin_amdgcn_sched_barrier(0); // one plane at a time: interleaved, the two planes' temporaries do not fit the 168-VGPR kernels
        }
    };
    // EPI 4 / 5, one row of one n-tile: everything in fp32, one rounding (see the template comment).  l1 / r2h / r2l: this lane's 16
    // bytes per plane of residual 1's lo planes, of residual 2 and of residual 2's lo planes (the latter two: EPI 5 only).
    constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;
    auto row_emit4 = [&](const f32x16& acc, const OutDesc& o, int rr, int n, const u32x4& l1, const u32x4 (&r2h)[2], const u32x4& r2l) {
        const int y = o.y0 + rr;
        char* ub = uniform_ptr(o.base);
        const unsigned pstride = unsigned(a.out16.plane_stride);
        char* ul = uniform_ptr(const_cast<char*>(plane_ptr(a.out16, o.slot, 0)) + a.out_lo_off + (long long)(ntw0 + n) * pstride);
        const unsigned rowoff = unsigned(y) * unsigned(o.W * kFPx);
        u32x4 vl;
#pragma unroll
        for (int p = 0; p < 2; p++)
        {
            // Four values at a time (one dword of the lo bytes: the 168-VGPR budget has no room for a whole decoded row), in PAIRS: this
            // epilogue is bound by its VALU issue slots (8 instructions per value against 2 in EPI 2), and packed fp32 arithmetic
            // (v_pk_mul / v_pk_fma / v_pk_add), v_cvt_pk_f32_bf8 and v_cvt_pk_bf8_f32 halve them.  Rounding points as before: the product
            // acc * s1 is rounded to fp32 before anything is added (an explicit fma takes it as its addend: nothing can be contracted).
            half8 vh;
            const half8 r = __builtin_bit_cast(half8, r2h[p]);
#pragma unroll
            for (int k = 0; k < 2; k++)
            {
                int wq = 0;
#pragma unroll
                for (int e2 = 0; e2 < 2; e2++)
                {
                    const int c0 = p * 8 + k * 4 + e2 * 2; // accumulator register of the pair's first value
                    const f32x2 av = {acc[c0], acc[c0 + 1]};
                    const f32x2 kinv = {kLoInv, kLoInv}, s1v = {a.s1, a.s1};
                    f32x2 f = __builtin_elementwise_fma(bf8_pair(l1[p * 2 + k], e2), kinv, av * s1v);
                    if (EPI == 5)
                    {
                        const f32x2 rv = {(float)r[k * 4 + e2 * 2], (float)r[k * 4 + e2 * 2 + 1]}, s2v = {a.s2, a.s2};
                        f = __builtin_elementwise_fma(bf8_pair(r2l[p * 2 + k], e2), kinv, __builtin_elementwise_fma(f, s2v, rv));
                    }
                    asm volatile("" : "+v"(f)); // f is an fp32 VALUE (hi rounds it, lo keeps its residue): the compiler must not fuse the fma above into the fp16 conversion
                    const half2v hv = __builtin_convertvector(f, half2v);
                    vh[k * 4 + e2 * 2] = hv[0];
                    vh[k * 4 + e2 * 2 + 1] = hv[1];
                    const f32x2 ksc = {kLoScale, kLoScale};
                    const f32x2 t = (f - __builtin_convertvector(hv, f32x2)) * ksc; // exact: hv is f rounded to 11 bits; <= half an ulp x 2048, never beyond 32768
                    wq = pack_bf8_pair(t[0], t[1], wq, e2);
                }
                vl[p * 2 + k] = unsigned(wq);
            }
            const unsigned poff = unsigned(n * 2 + p) * pstride;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ub, 0, o.live ? int(o.lim + poff) : 0, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vh), rs, o.voff + int(rowoff + poff), 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(ul, 0, (o.live && a.out_lo_off) ? int(o.lim) : 0, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(vl, rl, o.voff + int(rowoff), 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // conv_last (EPI 0): channels 0..2 of n-tile 0 -> planar fp16 [3][H][W] (the reference's `output` blob, consumed by
    // postproc_tiles), or -- non-TTA RGB -- straight into the uint8 image: realsr_postproc.comp:62-83 on the value rounded to
    // fp16 exactly as the planar path stores it (v*255 + 0.5, floor, clamp), at the tile's place minus the halo crop.
    auto store3 = [&](const float (&val)[4][3], const WorkItem& it) {
        if (hi != 0 || ntw0 != 0 || RSR_ABL(4)) return;
        const bool fold = (it.x0 & kFoldBit) != 0; // lane -> pixel as in make_out
        const int x = (it.x0 & (kFoldBit - 1)) + (fold ? (l32 & 15) : l32);
        const int yb = it.y0 + wrow * 4 + ((fold && l32 >= 16) ? 16 : 0);
        if (a.out_u8)
        {
            const int ox = it.pad0 + x, ow = pad2_w(it.pad2), oh = pad2_h(it.pad2);
            const int oim = pad2_img(it.pad2); // (a merged batch: the tile's own image and its row pitch)
            uint8_t* const oimg = a.out_u8s[oim];
            const int opitch = a.out_u8_ws[oim];
#pragma unroll
            for (int rr = 0; rr < 4; rr++)
            {
                const int y = yb + rr;
                // (x - crop, y - crop) inside the un-padded rectangle  <=>  image pixel (pad0 + x, pad1 + y) inside the tile's box
                const int rx = x - a.out_u8_crop, ry = y - a.out_u8_crop;
                if (rx >= 0 && rx < ow && ry >= 0 && ry < oh)
                {
                    uint8_t* o = oimg + ((long long)(it.pad1 + y) * opitch + ox) * 3;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        // the reference's `output` blob is fp16 (realsr.cpp:44-46); precise mode converts the fp32 value itself
                        float v = OUT32 ? val[rr][ch] : (float)(_Float16)val[rr][ch];
                        v = floorf(v * 255.f + 0.5f);
                        v = fminf(fmaxf(v, 0.f), 255.f);
                        o[a.out_u8_bgr ? 2 - ch : ch] = (uint8_t)v;
                    }
                }
            }
            return;
        }
        _Float16* o = reinterpret_cast<_Float16*>(static_cast<char*>(a.out_planar3) + (long long)it.slot * a.planar3_slot_stride);
        const long long hw = (long long)it.H * it.W;
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
        {
            const int y = yb + rr;
            if (y < it.H && x < it.W)
            {
                const long long pix = (long long)y * it.W + x;
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                {
                    float v = val[rr][ch];
                    if (a.lrelu) v = fmaxf(v, v * 0.2f);
                    if (OUT32) reinterpret_cast<float*>(o)[ch * hw + pix] = v; // (the slot stride is given in bytes)
                    else o[ch * hw + pix] = (_Float16)v;
                }
            }
        }
    };
    auto planar_store = [&](const f32x16 (&acc)[4][NTW], const WorkItem& it) {
        float val[4][3];
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) val[rr][ch] = acc[rr][0][ch];
        store3(val, it);
    };

    // ---- operand fragments, accumulators ----
    half8 X[6], Wa[3][NTW], Wb[WDB ? 3 : 1][NTW];
    f32x16 accA[4][NTW], accB[DEFER ? 4 : 1][NTW];

    int sP = 0, sW = 0; // ring slots of the half-stage being multiplied
    int r = 0, ck = 0, t = 0;
    auto xbase = [&](int slot, int dx) { return smem + slot * kFPatch + xcol[dx]; };
    auto wbase = [&](int slot) { return smem + slot * WB + woff; };
#define RSR_LDX(R, XB) X[R] = *reinterpret_cast<const half8*>((XB) + (R)*kFRow);
#define RSR_LDW(WS, DY, DX, WBASE)                                                                                   \
    _Pragma("unroll") for (int n_ = 0; n_ < NTW; n_++) WS[DY][n_] =                                                  \
        *reinterpret_cast<const half8*>((WBASE) + ((DY)*3 + (DX)) * (NT * 1024) + n_ * 1024);
#ifndef RSR_EXP_THIN
#define RSR_CELL(ACC, WS, DY, RR, FIRST)                                                                             \
    _Pragma("unroll") for (int n_ = 0; n_ < NTW; n_++) ACC[RR][n_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(         \
        WS[DY][n_], X[(RR) + (DY)], ((FIRST) && (DY) == 0) ? bias16[n_] : ACC[RR][n_], 0, 0, 0);
#else
    // EXPERIMENT BUILD ONLY (tools/build_variant.sh ... -DRSR_EXP_THIN=16|24; wrong results by construction): the instruction mix
    // of a Winograd convolution on this kernel's data path -- F(2x2,3x3) keeps 16 of the 36 MFMAs of a half-stage and pays 8
    // packed-fp16 adds per kept MFMA for the input transform, F(2,3) along x keeps 24 and pays ~1.5 -- with the same LDS-DMA,
    // fragment reads, barriers and epilogue, on data that stays full-entropy (the kept taps give a sane partial sum).  What it
    // measures is the BEST case of such a kernel (no output transform, no extra accumulators): DESIGN.md 4.1.
#define RSR_KEEP(DY, RR)                                                                                             \
    ((DY) == 0 || (RSR_EXP_THIN == 24 && (DY) == 1) ||                                                               \
     (RSR_EXP_THIN == 16 && ((curdx_ == 0 && (DY) == 1 && (RR) == 0) || (curdx_ == 1 && (DY) == 1 && (RR) == 1) || (curdx_ == 2 && (DY) == 2 && (RR) < 2))))
#define RSR_CELL(ACC, WS, DY, RR, FIRST)                                                                             \
    if (RSR_KEEP(DY, RR))                                                                                            \
    {                                                                                                                \
        _Pragma("unroll") for (int n_ = 0; n_ < NTW; n_++) ACC[RR][n_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(     \
            WS[DY][n_], X[(RR) + (DY)], ((FIRST) && (DY) == 0) ? bias16[n_] : ACC[RR][n_], 0, 0, 0);                 \
        if (RSR_EXP_THIN == 16 || ((DY) == 0 && (RR) < 3))                                                           \
        {                                                                                                            \
            half8 y_ = X[((RR) + (DY)) % 6] + X[((RR) + 2 * (DY) + 1) % 6]; /* distinct per cell: no CSE */         \
            if (RSR_EXP_THIN == 16) y_ = y_ + X[((RR) + (DY) + 3) % 6];                                              \
            asm volatile("" ::"v"(y_));                                                                              \
        }                                                                                                            \
    }
#endif
#define RSR_NOHK(c)

    // One (dx) step = 12 cells (dy, rr): rows 0-1 through the three dy taps, then rows 2-3 -- (0,0) (0,1) (1,0) (1,1) (2,0)
    // (2,1) | (0,2) (0,3) (1,2) (1,3) (2,2) (2,3).  Consecutive MFMAs share the weight fragment in pairs and the pixel
    // fragment along the short diagonals (14 operand changes per step; 18 in plain anti-diagonal order), a row's
    // accumulator comes back every second MFMA, and fragment X[j] is dead after cells 0 / 2 / 6 / 8 / 10 / 11: the
    // fragments of the NEXT step (NXB / NWB: their LDS bases, NDX its dx) are reloaded in place right there, >= 8 MFMA
    // slots before their first use.  BAR: the
    // dx = 2 step of a half-stage -- once the first three cells are issued every LDS read of the half-stage has had > 100
    // cycles to return, the wave passes the barrier ("my reads of this half-stage are done" / "the next one has landed")
    // and only then touches the next half-stage's slots.  WCUR / WNXT: weight fragment sets (the same array when !WDB).
    // HK(c): work of the deferred epilogue that rides behind cell c (fenced, so it stays there).
    // ITEMQ: also fetch the descriptor of the block after next -- a scalar load the COMPILER must not see (a visible SMEM
    // load turns every counted lgkmcnt wait behind it into lgkmcnt(0)); it lands under the three cells in front of the
    // barrier and is covered by the barrier's own lgkmcnt(0).  (The compiler's counted DS waits in between stay safe: the
    // hidden load can only make them over-wait by one.)
#define RSR_STEP(ACC, WCUR, WNXT, NXB, NWB, NDX, FIRST, BAR, ITEMQ, HK)                                              \
    {                                                                                                                \
        constexpr int curdx_ = ((NDX) + 2) % 3; /* the dx this step multiplies (experiment builds) */                \
        (void)curdx_;                                                                                                \
        if ((BAR) && (ITEMQ))                                                                        \
        {                                                                                                            \
            const WorkItem* ip_ = a.items + (first + min(r + 2, nmine - 1) * nj);                                    \
            asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(item_q) : "s"(ip_));                                    \
        }                                                                                                            \
        RSR_CELL(ACC, WCUR, 0, 0, FIRST) HK(0)                                                                       \
        if (!(BAR)) { RSR_LDX(0, NXB) if (WDB) { RSR_LDW(WNXT, 0, NDX, NWB) } }                                      \
        RSR_CELL(ACC, WCUR, 0, 1, FIRST) HK(1) RSR_CELL(ACC, WCUR, 1, 0, FIRST) HK(2)                                \
        if (BAR)                                                                                                     \
        {                                                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            if (tracing) t_arr = __builtin_amdgcn_s_memtime();                                                       \
            if (ITEMQ) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+s"(item_q)::"memory");                   \
            else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                     \
            if (tracing) t_rel = __builtin_amdgcn_s_memtime();                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            RSR_LDX(0, NXB) if (WDB) { RSR_LDW(WNXT, 0, NDX, NWB) }                                                  \
        }                                                                                                            \
        RSR_LDX(1, NXB) if (WDB) { RSR_LDW(WNXT, 1, NDX, NWB) }                                                      \
        RSR_CELL(ACC, WCUR, 1, 1, FIRST) HK(3) RSR_CELL(ACC, WCUR, 2, 0, FIRST) HK(4) RSR_CELL(ACC, WCUR, 2, 1, FIRST) HK(5) \
        if (WDB) { RSR_LDW(WNXT, 2, NDX, NWB) }                                                                      \
        RSR_CELL(ACC, WCUR, 0, 2, FIRST) HK(6)                                                                       \
        RSR_LDX(2, NXB)                                                                                              \
        RSR_CELL(ACC, WCUR, 0, 3, FIRST) HK(7)                                                                       \
        if (!WDB) { RSR_LDW(WNXT, 0, NDX, NWB) }                                                                     \
        RSR_CELL(ACC, WCUR, 1, 2, FIRST) HK(8)                                                                       \
        RSR_LDX(3, NXB)                                                                                              \
        RSR_CELL(ACC, WCUR, 1, 3, FIRST) HK(9)                                                                       \
        if (!WDB) { RSR_LDW(WNXT, 1, NDX, NWB) }                                                                     \
        RSR_CELL(ACC, WCUR, 2, 2, FIRST) HK(10)                                                                      \
        RSR_LDX(4, NXB)                                                                                              \
        RSR_CELL(ACC, WCUR, 2, 3, FIRST) HK(11)                                                                      \
        RSR_LDX(5, NXB)                                                                                              \
        if (!WDB) { RSR_LDW(WNXT, 2, NDX, NWB) }                                                                     \
    }
    // Issue-order pins for a step without hooks (MFMA / DS-read group sizes follow RSR_STEP; a BAR step is split by the
    // barrier).  Steps with hooks are pinned by the hooks' fences instead (PIN = 0).
#define RSR_STEP_SCHED(BAR, PIN)                                                                                     \
    if (PIN)                                                                                                         \
    {                                                                                                                \
        if (BAR)                                                                                                     \
        {                                                                                                            \
            RSR_SG(RSR_DSR, WDB ? 4 : 2);                                                                            \
            RSR_SG(RSR_MFMA, 3 * NTW);                                                                               \
        }                                                                                                            \
        else                                                                                                         \
        {                                                                                                            \
            RSR_SG(RSR_MFMA, 1 * NTW); RSR_SG(RSR_DSR, WDB ? 2 : 1);                                                 \
            RSR_SG(RSR_MFMA, 2 * NTW); RSR_SG(RSR_DSR, WDB ? 2 : 1);                                                 \
            RSR_SG(RSR_MFMA, 3 * NTW);                                                                               \
        }                                                                                                            \
        if (WDB)                                                                                                     \
        {                                                                                                            \
            RSR_SG(RSR_DSR, 1);                                                                                      \
            RSR_SG(RSR_MFMA, 1); RSR_SG(RSR_DSR, 1);                                                                 \
            RSR_SG(RSR_MFMA, 2); RSR_SG(RSR_DSR, 1);                                                                 \
            RSR_SG(RSR_MFMA, 2); RSR_SG(RSR_DSR, 1);                                                                 \
            RSR_SG(RSR_MFMA, 1); RSR_SG(RSR_DSR, 1);                                                                 \
        }                                                                                                            \
        else                                                                                                         \
        {                                                                                                            \
            RSR_SG(RSR_MFMA, 1 * NTW); RSR_SG(RSR_DSR, 1);                                                           \
            RSR_SG(RSR_MFMA, 1 * NTW); RSR_SG(RSR_DSR, NTW);                                                         \
            RSR_SG(RSR_MFMA, 1 * NTW); RSR_SG(RSR_DSR, 1);                                                           \
            RSR_SG(RSR_MFMA, 1 * NTW); RSR_SG(RSR_DSR, NTW);                                                         \
            RSR_SG(RSR_MFMA, 1 * NTW); RSR_SG(RSR_DSR, 1);                                                           \
            RSR_SG(RSR_MFMA, 1 * NTW); RSR_SG(RSR_DSR, 1 + NTW);                                                     \
        }                                                                                                            \
    }
    // Identity tap: out = s1*(conv + b) + x with x = this conv's own input planes 2*nt, 2*nt+1 (RDB conv5, fp16 trunk):
    // while that plane is the current half-stage, acc += (1/s1)*x on the matrix pipe (A = coef*I on the plane's 16
    // output channels, B = the centre-tap pixels) -- the epilogue has no residual to fetch.  Runs after the half-stage's
    // first step (every accumulator row is initialised by then) and before its dx = 2 step hands the slot back.
#define RSR_IDTAP(ACC)                                                                                               \
    if (RESID && idt)                                                                                                \
    {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        _Pragma("unroll") for (int n_ = 0; n_ < NTW; n_++) if ((ck >> 1) == ntw0 + n_)                               \
        {                                                                                                            \
            half8 idf;                                                                                               \
            int l32i = l32;                                                                                          \
            if (PREC) asm volatile("" : "+v"(l32i)); /* recomputed per use: hoisted, the two fragments cost the precise kernels 8 VGPRs they need */ \
            _Pragma("unroll") for (int e = 0; e < 8; e++) idf[e] =                                                   \
                ((ck & 1) * 16 + hi * 8 + e == row_cout(l32i)) ? (_Float16)a.res1_coef : (_Float16)0.f;              \
            const char* cb_ = xbase(sP, 1);                                                                          \
            _Pragma("unroll") for (int rr = 0; rr < 4; rr++)                                                         \
            {                                                                                                        \
                const half8 xc = *reinterpret_cast<const half8*>(cb_ + (rr + 1) * kFRow);                            \
                ACC[rr][n_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(idf, xc, ACC[rr][n_], 0, 0, 0);                 \
            }                                                                                                        \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    }
    // one half-stage (3 steps) on accumulators ACC; W0 holds the dx = 0 fragments (sets alternate per step when WDB);
    // HK0..2: hook macros of the three steps (RSR_NOHK = none, the step is then pinned by RSR_STEP_SCHED)
#define RSR_HALF(ACC, W0, W1, FIRST, HK0, P0, HK1, P1, HK2, P2)                                                      \
    {                                                                                                                \
        const int nsP = sP == PR - 1 ? 0 : sP + 1, nsW = sW == WR - 1 ? 0 : sW + 1;                                  \
        {                                                                                                            \
            const char* nxb = xbase(sP, 1);                                                                          \
            const char* nwb = wbase(sW);                                                                             \
            RSR_STEP(ACC, W0, W1, nxb, nwb, 1, FIRST, false, false, HK0)                                             \
            RSR_STEP_SCHED(false, P0)                                                                                \
        }                                                                                                            \
        RSR_IDTAP(ACC)                                                                                               \
        {                                                                                                            \
            const char* nxb = xbase(sP, 2);                                                                          \
            const char* nwb = wbase(sW);                                                                             \
            RSR_STEP(ACC, W1, W0, nxb, nwb, 2, false, false, false, HK1)                                             \
            RSR_STEP_SCHED(false, P1)                                                                                \
        }                                                                                                            \
        {                                                                                                            \
            const char* nxb = xbase(nsP, 0);                                                                         \
            const char* nwb = wbase(nsW);                                                                            \
            RSR_STEP(ACC, W0, W1, nxb, nwb, 0, false, true, FIRST, HK2)                                              \
            RSR_STEP_SCHED(true, P2)                                                                                 \
        }                                                                                                            \
        if (tracing && lane == 0 && t + 1 < 512)                                                                     \
        {                                                                                                            \
            a.trace[2 * (t + 1)] = t_arr;                                                                            \
            a.trace[2 * (t + 1) + 1] = t_rel;                                                                        \
        }                                                                                                            \
        sP = nsP;                                                                                                    \
        sW = nsW;                                                                                                    \
        t++;                                                                                                         \
    }
#define RSR_HALF_PLAIN(ACC, W0, W1, FIRST) RSR_HALF(ACC, W0, W1, FIRST, RSR_NOHK, 1, RSR_NOHK, 1, RSR_NOHK, 1)

    WorkItem it = load_item(a.items, first);
    WorkItem nxt = load_item(a.items, first + (nmine > 1 ? nj : 0));
    i32x8 item_q = {0, 0, 0, 0, 0, 0, 0, 0}; // descriptor of block r+2, fetched during block r's first half-stage
    auto item_from_q = [&]() {
        WorkItem w;
        w.slot = item_q[0]; w.y0 = item_q[1]; w.x0 = item_q[2]; w.H = item_q[3]; w.W = item_q[4];
        w.pad0 = item_q[5]; w.pad1 = item_q[6]; w.pad2 = item_q[7];
        return w;
    };

    // A block whose four rows of THIS wave lie wholly below the tile (blocks of 16 rows: a 220-row tile ends in its last block's
    // third wave, a 100-row tile in the first; any height works -- a 101-row tile keeps one row of the second wave alive and
    // row_store drops the other three) -- or wholly inside the frame of output
    // pixels nothing downstream reads (ConvArgs::margin) -- does no matrix work here: the wave only
    // keeps the workgroup's barrier cadence -- one per half-stage -- fetches the descriptor the live path would fetch, and
    // reloads the operands of the next block's first step.  16 x 32 block quantisation otherwise costs 5.4 % of the MFMA
    // work of a 1080p frame at tile 200 (PMC, round 2), 2.8 points of it in such rows; under the board's power cap skipped
    // MFMAs + fragment reads are what buys clock.
    auto wave_is_dead = [&](const WorkItem& w) {
        const int y = w.y0 + wrow * 4;
        // A folded block (kFoldBit) is judged by its FIRST strip alone: rows below the tile / inside the bottom margin hit the second
        // strip whenever they hit the first, and the tables never fold a pair of block rows that reaches into the top margin
        // (engine.cpp append_block_items) -- so this expression, whose exact shape the register allocation of every instantiation
        // hangs on (+90 VGPRs and spills for a second pair of compares here, round 5), stays as it was.
        return (y >= w.H - a.margin || y + 4 <= a.margin) && !(a.dbg & 32);
    };
    auto skip_block = [&]() {
        {
            const WorkItem* ip_ = a.items + (first + min(r + 2, nmine - 1) * nj);
            asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(item_q) : "s"(ip_));
        }
        for (int h_ = 0; h_ < nst; h_++)
        {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+s"(item_q)::"memory");
            sP = sP == PR - 1 ? 0 : sP + 1;
            sW = sW == WR - 1 ? 0 : sW + 1;
            t++;
        }
        __builtin_amdgcn_sched_barrier(0);
        const char* xb = xbase(sP, 0);
        const char* wb = wbase(sW);
#pragma unroll
        for (int q = 0; q < 6; q++) { RSR_LDX(q, xb) }
#pragma unroll
        for (int dy = 0; dy < 3; dy++) { RSR_LDW(Wa, dy, 0, wb) }
        if (!DEFER) load_bias();
        __builtin_amdgcn_sched_barrier(0);
    };

    unsigned long long t_arr = 0, t_rel = 0;
    (void)t_rel;
    if (tracing) t_arr = __builtin_amdgcn_s_memtime();
    asm volatile("s_barrier" ::: "memory"); // E_0: half-stage 0 is in LDS
#ifdef RSR_FLOW_LIFE
    if (a.trace && tid == 0) a.trace[blockIdx.x * 4 + 0] = life_t0;
    RSR_LIFE(1)
#endif
    if (tracing && lane == 0)
    {
        a.trace[0] = t_arr;
        a.trace[1] = __builtin_amdgcn_s_memtime();
    }
    if constexpr (LAST3)
    {
        // ---- conv_last, 64 -> 3: (dy, cout) in the M dimension -------------------------------------------------------------
        // With 3 real output channels a 32 x 32 x 16 MFMA has room for the three dy taps at once: A = the aux image
        // [dx][row = dy*8 + c][16 cin] (rows of other (dy, c) zero), B = patch row R  =>  D[dy*8 + c][px] is tap (dy, dx)'s
        // contribution of patch row R to OUTPUT row R - dy.  One accumulator per patch row (6 per wave for its 4 output rows),
        // 3 MFMAs per patch row and plane instead of 9 per output row: 72 instead of 144 per block.  Output row rr =
        // acc[rr][0*4 + c] + acc[rr + 1][1*4 + c] + acc[rr + 2][2*4 + c] in the lanes with hi == 0 (accumulator row q*8 + hi*4 + e).
        // Fragments are double-buffered across the dx steps; the half-stage barrier sits behind the wave's last LDS read.
        f32x16 acc[6];
        half8 Xa[6], Xb[6], Wfa, Wfb;
        const int woff3 = kWOff + l32 * 32 + ((hi ^ ((l32 >> 3) & 1)) << 4);
        const int WB3 = a.wpieces * 1024;
        auto ldx = [&](half8(&Xf)[6], int slot, int dx) {
            const char* xb = xbase(slot, dx);
#pragma unroll
            for (int q = 0; q < 6; q++) Xf[q] = *reinterpret_cast<const half8*>(xb + q * kFRow);
        };
        auto ldw = [&](int plane, int dx) { return *reinterpret_cast<const half8*>(smem + woff3 + plane * WB3 + dx * 1024); };
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define RSR_M6(XF, WF, FIRST)                                                                                        \
    _Pragma("unroll") for (int q = 0; q < 6; q++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WF, XF[q], (FIRST) ? zero16 : acc[q], 0, 0, 0);
        // one half-stage on fragments (XC, WC) = its dx 0 operands, already in registers; leaves the next half-stage's in (XN, WN)
#define RSR_HALF3(XC, WC, XN, WN, FIRST)                                                                             \
    {                                                                                                                \
        const int nsP = sP == PR - 1 ? 0 : sP + 1, nck = ck + 1 == nst ? 0 : ck + 1;                                 \
        if (FIRST) /* descriptor of the block after next: a scalar load hidden from the compiler (see RSR_STEP), early */ \
        {                                                                                                            \
            const WorkItem* ip_ = a.items + (first + min(r + 2, nmine - 1) * nj);                                    \
            asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(item_q) : "s"(ip_));                                    \
        }                                                                                                            \
        ldx(XN, sP, 1); WN = ldw(ck, 1);                                                                             \
        RSR_M6(XC, WC, FIRST)                                                                                        \
        ldx(XC, sP, 2); WC = ldw(ck, 2);                                                                             \
        RSR_M6(XN, WN, false)                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+s"(item_q)::"memory");                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ldx(XN, nsP, 0); WN = ldw(nck, 0);                                                                           \
        RSR_M6(XC, WC, false)                                                                                        \
        sP = nsP;                                                                                                    \
        t++;                                                                                                         \
    }
        ldx(Xa, 0, 0);
        Wfa = ldw(0, 0);
        for (r = 0; r < nmine; r++)
        {
            while (wave_is_dead(it))
            {
                {
                    const WorkItem* ip_ = a.items + (first + min(r + 2, nmine - 1) * nj);
                    asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(item_q) : "s"(ip_));
                }
                for (int h_ = 0; h_ < nst; h_++)
                {
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+s"(item_q)::"memory");
                    sP = sP == PR - 1 ? 0 : sP + 1;
                    t++;
                }
                ldx(Xa, sP, 0);
                Wfa = ldw(0, 0);
                it = nxt;
                nxt = item_from_q();
                if (++r >= nmine) break;
            }
            if (r >= nmine) break;
            for (int cp = 0; cp < nst; cp += 2)
            {
                ck = cp;
                if (cp == 0) { RSR_HALF3(Xa, Wfa, Xb, Wfb, true) }
                else { RSR_HALF3(Xa, Wfa, Xb, Wfb, false) }
                ck = cp + 1;
                RSR_HALF3(Xb, Wfb, Xa, Wfa, false)
            }
            __builtin_amdgcn_sched_barrier(0);
            float val[4][3];
            const float* bl = reinterpret_cast<const float*>(smem + kBiasOff);
#pragma unroll
            for (int rr = 0; rr < 4; rr++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) val[rr][ch] = acc[rr][ch] + acc[rr + 1][4 + ch] + acc[rr + 2][8 + ch] + bl[ch];
            store3(val, it);
            __builtin_amdgcn_sched_barrier(0);
            it = nxt;
            nxt = item_from_q();
        }
#undef RSR_HALF3
#undef RSR_M6
        RSR_LIFE(2)
        return;
    }
    {
        const char* xb = xbase(0, 0);
        const char* wb = wbase(0);
#pragma unroll
        for (int rr = 0; rr < 6; rr++) { RSR_LDX(rr, xb) }
#pragma unroll
        for (int dy = 0; dy < 3; dy++) { RSR_LDW(Wa, dy, 0, wb) }
    }

    if (!DEFER)
    {
        for (r = 0; r < nmine; r++)
        {
            // dead blocks are consumed in a loop of their own IN FRONT of a live block (an if / else around the block body
            // makes the untouched accumulators meet the live path's fresh ones in phis the register coalescer does not
            // fold: +64 VGPRs, spills in the 168-VGPR kernels)
            while (wave_is_dead(it))
            {
                skip_block();
                it = nxt;
                nxt = item_from_q();
                if (++r >= nmine) break;
            }
            if (r >= nmine) break;
            for (int cp = 0; cp < nst; cp += 2)
            {
                ck = cp;
                if (WDB)
                {
                    if (cp == 0) { RSR_HALF_PLAIN(accA, Wa, Wb, true) }
                    else { RSR_HALF_PLAIN(accA, Wa, Wb, false) }
                    ck = cp + 1;
                    if (cp + 2 >= nst) res2_prefetch(it, 0);
                    RSR_HALF_PLAIN(accA, Wb, Wa, false)
                }
                else
                {
                    if (cp == 0) { RSR_HALF_PLAIN(accA, Wa, Wa, true) }
                    else { RSR_HALF_PLAIN(accA, Wa, Wa, false) }
                    ck = cp + 1;
                    if (PRE2 && cp + 2 >= nst) res2_prefetch(it, 0); // (the precise kernels: one n-tile per wave without double-buffered weights)
                    RSR_HALF_PLAIN(accA, Wa, Wa, false)
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- epilogue of block r (the fragments of block r+1's first step are already in registers) ----
            // The operands of block r+1's first step and the bias registers are (re)loaded while the last row is converted
            // -- three of the four accumulator rows are dead by then -- so that their registers are the epilogue's in
            // between (the block's last step has fetched the same fragments already; that copy simply dies here).
            auto refill = [&]() {
                __builtin_amdgcn_sched_barrier(0);
                const char* xb = xbase(sP, 0);
                const char* wb = wbase(sW);
#pragma unroll
                for (int q = 0; q < 6; q++) { RSR_LDX(q, xb) }
#pragma unroll
                for (int dy = 0; dy < 3; dy++) { RSR_LDW(Wa, dy, 0, wb) }
                load_bias();
                __builtin_amdgcn_sched_barrier(0);
            };
            if (LASTG)
            {
                planar_store(accA, it);
                refill();
            }
            else if constexpr (PREC && NTW == 1)
            {
                // Register budget: 168 VGPRs, all four accumulator rows live at the start.  EPI 4: the lo rows of residual 1 came in with
                // the prefetch, one half-stage ahead -- nothing is fetched here.  EPI 5 (every third RDB, trunk_conv): rows 0-1 of
                // residual 2 and of both residuals' lo planes are fetched here, rows 2-3 into the same registers once rows 0-1 are
                // through: two exposed round trips per block, in 24 of the 352 launches.
                const OutDesc o = make_out(it, true);
                if (EPI == 4)
                {
                    const u32x4 (&nor)[2] = r2q[0][0]; // (not read)
#pragma unroll
                    for (int rr = RSR_PREC_PRE; rr < 4; rr++) lo_row(a.res1, a.lo1_off, l1q[EPI == 4 ? rr : 0], o, rr, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int rr = 0; rr < 4; rr++)
                    {
                        if (rr == 3) refill();
                        row_emit4(accA[rr][0], o, rr, 0, l1q[EPI == 4 ? rr : 0], nor, l1q[0]);
                    }
                }
                else
                {
                    u32x4 rh[2][2];
                    u32x4 la[2], ra[2];
#pragma unroll
                    for (int h2 = 0; h2 < 2; h2++)
                    {
#pragma unroll
                        for (int k = 0; k < 2; k++)
                        {
                            res2_row(rh[k], o, 2 * h2 + k, 0);
                            lo_row(a.res1, a.lo1_off, la[k], o, 2 * h2 + k, 0);
                            lo_row(a.res2, a.lo2_off, ra[k], o, 2 * h2 + k, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        row_emit4(accA[2 * h2][0], o, 2 * h2, 0, la[0], rh[0], ra[0]);
                        if (h2) refill();
                        row_emit4(accA[2 * h2 + 1][0], o, 2 * h2 + 1, 0, la[1], rh[1], ra[1]);
                    }
                }
            }
            else
            {
                const OutDesc o = make_out(it, true);
#pragma unroll
                for (int rr = 0; rr < 4; rr++)
                {
                    if (rr == 3) refill();
#pragma unroll
                    for (int n = 0; n < NTW; n++) row_emit(accA[rr][n], o, rr, n);
                    if (EPI == 2 && rr == 1) res2_prefetch(it, 2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            it = nxt;
            nxt = item_from_q();
        }
    }
    else
    {
        // ---- deferred epilogue (32 output channels, 4 MFMA waves): block r accumulates into one accumulator set while
        // the other set, holding block r-1, is drained during block r's first ten steps.  The work is cut into 40 pieces
        // of <= 5 instructions that ride behind every THIRD MFMA cell (fenced in place; generated by
        // tools/gen_flow_hooks.py): per row eight conversions of a value pair (LeakyReLU = med3(v, slope*v, +inf), fp16
        // pack) and the two 1-KiB stores of the row's planes, straight from the packed pairs (double-buffered by row).
        // Measured: a piece behind every cell 89.9 ms per C2 frame, every second 89.4-89.6, every third 88.9, 3.5 no better;
        // the round-3 drain (60 pieces incl. the LDS transpose, every second cell) 90.1.  The first block drains its
        // uninitialised partner set into a null resource.
        OutDesc od = make_out(it, false);
        u32x4 tq[2];
        half2v pk[2][4][2]; // value pairs of the row being converted, two rows in flight (a row's stores read set r & 1)
        float pinf = __builtin_inff();
        asm volatile("" : "+s"(pinf)); // opaque: med3(v, s*v, +inf) stays ONE instruction (a literal folds to maxnum + canonicalize)
        auto row_store1 = [&](const u32x4& v, const OutDesc& o, int rr, int p) {
            const int y = o.y0 + rr;
            char* ub = uniform_ptr(o.base);
            const unsigned poff = unsigned(p) * unsigned(a.out16.plane_stride);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ub, 0, o.live ? int(o.lim + poff) : 0, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, o.voff + int(unsigned(y) * unsigned(o.W * kFPx) + poff), 0, 0);
        };
#include "conv_flow_hooks.inc" // RSR_HK_S<step>_<cell>: generated at build time by tools/gen_flow_hooks.py (Makefile)
    // A dead block first stores the pending set (the one a live block would drain underneath its MFMAs) -- once: `od` is dead
    // afterwards -- and "defines" its own set with an empty asm: without that the untouched accumulators meet the live path's
    // fresh ones in phis the register coalescer does not fold (+64 VGPRs = spills).
#define RSR_BLOCK(ACC)                                                                                               \
    if (wave_is_dead(it))                                                                                            \
    {                                                                                                                \
        if (od.live)                                                                                                 \
        {                                                                                                            \
            _Pragma("unroll") for (int rr = 0; rr < 4; rr++)                                                         \
            {                                                                                                        \
                row_pack(RSR_OLD[rr][0], tq);                                                                        \
                row_store(tq, od, rr, 0);                                                                            \
            }                                                                                                        \
        }                                                                                                            \
        skip_block();                                                                                                \
        _Pragma("unroll") for (int rr = 0; rr < 4; rr++) asm volatile("" : "=v"(ACC[rr][0]));                        \
        od = make_out(it, false);                                                                                    \
        it = nxt;                                                                                                    \
        nxt = item_from_q();                                                                                         \
    }                                                                                                                \
    else                                                                                                             \
    {                                                                                                                \
        ck = 0;                                                                                                      \
        RSR_HALF(ACC, Wa, Wb, true, RSR_HK_S0, RSR_PIN_S0, RSR_HK_S1, RSR_PIN_S1, RSR_HK_S2, RSR_PIN_S2)             \
        ck = 1;                                                                                                      \
        RSR_HALF(ACC, Wb, Wa, false, RSR_HK_S3, RSR_PIN_S3, RSR_HK_S4, RSR_PIN_S4, RSR_HK_S5, RSR_PIN_S5)            \
        ck = 2;                                                                                                      \
        RSR_HALF(ACC, Wa, Wb, false, RSR_HK_S6, RSR_PIN_S6, RSR_HK_S7, RSR_PIN_S7, RSR_HK_S8, RSR_PIN_S8)            \
        ck = 3;                                                                                                      \
        RSR_HALF(ACC, Wb, Wa, false, RSR_HK_S9, RSR_PIN_S9, RSR_HK_S10, RSR_PIN_S10, RSR_HK_S11, RSR_PIN_S11)        \
        for (int cp = 4; cp < nst; cp += 2)                                                                          \
        {                                                                                                            \
            ck = cp;                                                                                                 \
            RSR_HALF_PLAIN(ACC, Wa, Wb, false)                                                                       \
            ck = cp + 1;                                                                                             \
            RSR_HALF_PLAIN(ACC, Wb, Wa, false)                                                                       \
        }                                                                                                            \
        od = make_out(it, true);                                                                                     \
        it = nxt;                                                                                                    \
        nxt = item_from_q();                                                                                         \
    }
        for (r = 0; r < nmine; r += 2)
        {
#define RSR_OLD accB
            RSR_BLOCK(accA)
#undef RSR_OLD
            if (r + 1 < nmine)
            {
                r++;
#define RSR_OLD accA
                RSR_BLOCK(accB)
#undef RSR_OLD
                r--;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (nmine & 1)
        {
#pragma unroll
            for (int rr = 0; rr < 4; rr++)
            {
                row_pack(accA[rr][0], tq);
                row_store(tq, od, rr, 0);
            }
        }
        else
        {
#pragma unroll
            for (int rr = 0; rr < 4; rr++)
            {
                row_pack(accB[rr][0], tq);
                row_store(tq, od, rr, 0);
            }
        }
#undef RSR_BLOCK
#include "conv_flow_hooks_undef.inc"
    }
    RSR_LIFE(2)
#undef RSR_LIFE
#undef RSR_HALF_PLAIN
#undef RSR_HALF
#undef RSR_IDTAP
#undef RSR_STEP
#undef RSR_STEP_SCHED
#undef RSR_CELL
#undef RSR_LDX
#undef RSR_LDW
#undef RSR_NOHK
}

// ---- launch ------------------------------------------------------------------------------------
constexpr int kLdsMax = 160 * 1024;

template <int NT, int NTW, bool UPS, int EPI, bool DEFER>
static hipError_t flow_attr()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_flow<NT, NTW, UPS, EPI, DEFER, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, FlowCfg<NT>::TOTAL);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_flow<NT, NTW, UPS, EPI, DEFER, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsMax);
    return e;
}

// patch ring depth a resident-weight launch can afford (0: the weights do not fit next to 3 patches)
template <int NT>
static int resident_ring(int nst, int reserve)
{
    const int rest = kLdsMax - nst * FlowCfg<NT>::WB - reserve - NT * 128;
    const int pr = rest / kFPatch;
    return pr < 3 ? 0 : (pr > 6 ? 6 : pr);
}

template <int NT, int NTW, bool UPS, int EPI, bool DEFER>
static void flow_launch(const ConvArgs& a_in, int ncu, bool resident, int reserve, hipStream_t st)
{
    int grid = ncu & ~7;
    const int per = (a_in.nitems + 7) / 8;
    if (per * 8 < grid) grid = per * 8;
    const int nst = a_in.n0 + a_in.n1;
    const int pr = resident ? resident_ring<NT>(nst, reserve * NT) : 0; // (measured: a ring of 3 is as fast as one of 5 -- depth is not a limit)
    if (pr)
    {
        ConvArgs a = a_in;
        a.pr = pr;
        const int lds = pr * kFPatch + nst * FlowCfg<NT>::WB + NT * 128;
        hipLaunchKernelGGL((conv3x3_flow<NT, NTW, UPS, EPI, DEFER, true>), dim3(grid), dim3((4 * NT / NTW + 4) * 64), lds, st, a);
    }
    else
        hipLaunchKernelGGL((conv3x3_flow<NT, NTW, UPS, EPI, DEFER, false>), dim3(grid), dim3((4 * NT / NTW + 4) * 64), FlowCfg<NT>::TOTAL, st, a_in);
}

// every instantiation the engine can reach, for the per-device opt-in to > 64 KiB of dynamic LDS
#define RSR_FLOW_VARIANTS(F)                                                                                         \
    F(1, 1, false, 0, false) F(1, 1, false, 1, false) F(1, 1, false, 1, true) F(1, 1, false, 2, false) F(1, 1, false, 7, false) \
    F(1, 1, true, 1, false)                                                                                          \
    F(2, 1, false, 1, false) F(2, 1, false, 2, false) F(2, 1, true, 1, false) F(2, 1, false, 4, false) F(2, 1, false, 5, false) \
    F(2, 2, false, 1, false) F(2, 2, false, 2, false) F(2, 2, true, 1, false)

hipError_t flow_init_device()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_flow<1, 1, false, 3, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsMax);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_flow<1, 1, false, 6, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsMax);
#define RSR_F(NT, NTW, UPS, EPI, DEFER)                                                                              \
    if (e == hipSuccess) e = flow_attr<NT, NTW, UPS, EPI, DEFER>();
    RSR_FLOW_VARIANTS(RSR_F)
#undef RSR_F
    return e;
}

// flags: bit 0 = two n-tiles per MFMA wave for the 64-output-channel convs (4 MFMA waves), bit 1 = no deferred epilogue,
//        bit 2 = weights re-streamed per block even where they would fit in LDS for the whole launch,
//        bit 3 = conv_last through the generic 32-output-channel path (9 taps x 32 padded couts) instead of (dy, cout) in M
bool launch_conv_flow(const ConvArgs& a_in, int nt, int ncu, int flags, hipStream_t st)
{
    if (a_in.nitems <= 0) return true;
    ConvArgs a = a_in;
    if (a.res1_kind == 1 && !a.res1_in_acc)
    {
        // a first residual that is not this conv's own input (trunk_conv: fea + trunk) is fetched through the second
        // residual's slot: fp16(s1*acc)*1 + r rounds exactly like fp16(s1*acc) + r
        if (a.res2_kind) return false;
        a.res2 = a.res1;
        a.lo2_off = a.lo1_off;
        a.lo1_off = 0;
        a.res2_kind = 1;
        a.s2 = 1.f;
    }
    if (((a.n0 + a.n1) & 1) || !a.wpk16) return false;
    const bool ups = a.lvl_out != a.lvl_in;
    int epi = 0;
    if (a.out16.base && !a.out_planar3 && !a.out_u8)
    {
        if (a.res1_kind == 0 && a.res2_kind == 0) epi = 1;
        else if (a.res1_kind == 1 && (a.res2_kind == 0 || a.res2_kind == 1)) epi = 2;
        else return false;
        // the precise residual stream: every residual form, and a plain conv that starts the stream (conv_first: its lo planes are kept)
        if (a.precise && (epi == 2 || (epi == 1 && a.out_lo_off && !a.lrelu && !ups))) epi = a.res2_kind ? 5 : 4;
        if (epi >= 4 && !a.res1_kind) a.res1 = a.out16; // (conv_first: lo1_off == 0, the null resource is built on a valid base)
    }
    else if ((!a.out_planar3 && !a.out_u8) || a.out16.base || a.res1_kind || a.res2_kind) return false;
    const bool ntw2 = (flags & 1) != 0, defer = !(flags & 2), res = !(flags & 4);
    const int reserve = (flags & 16) ? 8192 : 0; // A/B aid: size the patch ring as if the rounds 1-3 transpose scratch (8 KB per n-tile) were still there
    if (nt == 1)
    {
        if (epi == 0 && !ups && a.waux && !(flags & 8))
        { // conv_last with (dy, cout) in the MFMA's M dimension: the 3-KB-per-plane aux image, always resident
            const int nst = a.n0 + a.n1;
            int pr = (kLdsMax - nst * 3072 - 128) / kFPatch;
            pr = pr > 6 ? 6 : pr;
            if (pr < 3) return false;
            a.wpk16 = a.waux;
            a.wpieces = 3;
            a.pr = pr;
            int grid = ncu & ~7;
            const int per = (a.nitems + 7) / 8;
            if (per * 8 < grid) grid = per * 8;
            if (a.precise) hipLaunchKernelGGL((conv3x3_flow<1, 1, false, 6, false, true>), dim3(grid), dim3(512), pr * kFPatch + nst * 3072 + 128, st, a);
            else hipLaunchKernelGGL((conv3x3_flow<1, 1, false, 3, false, true>), dim3(grid), dim3(512), pr * kFPatch + nst * 3072 + 128, st, a);
        }
        else if (epi == 0 && !ups && a.precise) flow_launch<1, 1, false, 7, false>(a, ncu, res, reserve, st);
        else if (epi == 0 && !ups) flow_launch<1, 1, false, 0, false>(a, ncu, res, reserve, st);
        else if (epi == 1 && !ups && defer) flow_launch<1, 1, false, 1, true>(a, ncu, res, reserve, st);
        else if (epi == 1 && !ups) flow_launch<1, 1, false, 1, false>(a, ncu, res, reserve, st);
        else if (epi == 1) flow_launch<1, 1, true, 1, false>(a, ncu, res, reserve, st);
        else if (epi == 2 && !ups) flow_launch<1, 1, false, 2, false>(a, ncu, res, reserve, st);
        else return false;
        return true;
    }
    if (nt != 2 || epi == 0) return false;
    if (ntw2 && epi < 4) // (the precise residual forms exist in the 8 x 32 wave layout only: their epilogue needs its SGPRs)
    {
        if (epi == 1 && !ups) flow_launch<2, 2, false, 1, false>(a, ncu, res, reserve, st);
        else if (epi == 1) flow_launch<2, 2, true, 1, false>(a, ncu, res, reserve, st);
        else if (epi == 2 && !ups) flow_launch<2, 2, false, 2, false>(a, ncu, res, reserve, st);
        else return false;
    }
    else
    {
        if (epi == 1 && !ups) flow_launch<2, 1, false, 1, false>(a, ncu, res, reserve, st);
        else if (epi == 1) flow_launch<2, 1, true, 1, false>(a, ncu, res, reserve, st);
        else if (epi == 4 && !ups) flow_launch<2, 1, false, 4, false>(a, ncu, res, reserve, st);
        else if (epi == 5 && !ups) flow_launch<2, 1, false, 5, false>(a, ncu, res, reserve, st);
        else if (epi == 2 && !ups) flow_launch<2, 1, false, 2, false>(a, ncu, res, reserve, st);
        else return false;
    }
    return true;
}

} // namespace rsr
