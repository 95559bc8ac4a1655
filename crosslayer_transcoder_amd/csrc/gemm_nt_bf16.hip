// Fast path of clt_gemm_bf16 for the "NT" form  C[M][N] (+)= A[M][K] * B[N][K]^T + bias[N]  (both operands
// k-contiguous): the encoder forward of the step (reference model/clt.py:157-165 under bf16-mixed; A = standardized
// activations [tokens][d], B = feature-major encoder weight [features][d]) and every other NT-shaped contraction.
//
// gfx950 design: 256 (B rows) x 256 (A rows) x 64 block tile, 512 threads = 8 waves as 2 x 4, each wave 128 x 64 of
// v_mfma_f32_16x16x32_bf16 tiles.  B rows ride the MFMA "M" index and A rows the "N" index, so a lane's four
// accumulator registers are four CONSECUTIVE output columns of one C row: the epilogue is one 16-byte store per
// lane per tile.  Operand tiles go global -> LDS by DMA (buffer_load_dwordx4 ... lds, no VGPR round trip), double
// buffered (2 x 64 KB); the LDS image is lane-linear 128-byte rows whose 16-byte chunks are XOR-swizzled with
// bits 1..3 of the row — applied to the per-lane SOURCE address and again on the ds_read_b128 fragment reads —
// which spreads the 16 rows of a fragment over all 16 slots of the 256-byte bank row.  Blocks are remapped so
// that each XCD (own L2) works on a contiguous range of tiles.  Either operand may also be K-MAJOR ([K][rows]: template
// flags KMA / KMB, see the kernel) — the same kernel then serves all four transpose forms of clt_gemm_bf16.
#include <type_traits>
#include <stdlib.h>
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// (kernel experiments are built as variants of this file — `make variant V=n` defines GEMM_VARIANT — and compared in one
// process by scripts/gemm_ab.py; tried and rejected in round 2, all slower or equal on the shapes of the step: DMA pieces
// spread over the MFMA groups instead of issued together, s_setprio around the MFMA groups, the bias as the initial
// accumulator, an integer maximum-and-compare scan in the select epilogue, a staggered start of the persistent blocks: profiles/r02_gemm_ab.md)
#ifndef GEMM_VARIANT
#define GEMM_VARIANT 0
#endif

// GEMM_NT_STORE: the C tiles leave as non-temporal stores (written once, not read again by the kernel: they do not displace the
// operand panels from L2): PLT JumpReLU dense 20.3 -> 20.0 ms, TopK + aux 28.9 -> 28.5 ms, scripts/archive/gpu_gemmnt_ab.sh
#ifndef GEMM_NT_STORE
#define GEMM_NT_STORE 1
#endif
namespace {
// 16-bit conversions of the epilogues: an fp16 overflow is collected in `bad` and raises the loss scaler's flag once per tile
template <bool F16> __device__ __forceinline__ uint32_t pack16t(float lo, float hi, uint32_t& bad) {
    if constexpr (F16) return pack_f16x2_track(lo, hi, bad); else return pack_bf16x2(lo, hi);
}
template <bool F16> __device__ __forceinline__ f32x4 round16v(f32x4 v, uint32_t& bad) {
    const uint32_t u0 = pack16t<F16>(v[0], v[1], bad), u1 = pack16t<F16>(v[2], v[3], bad);
    if constexpr (F16) return f32x4{f16_to_f32((uint16_t)(u0 & 0xffffu)), f16_to_f32((uint16_t)(u0 >> 16)), f16_to_f32((uint16_t)(u1 & 0xffffu)), f16_to_f32((uint16_t)(u1 >> 16))};
    else return f32x4{__uint_as_float(u0 << 16), __uint_as_float(u0 & 0xffff0000u), __uint_as_float(u1 << 16), __uint_as_float(u1 & 0xffff0000u)};
}
constexpr int TB = 256, TA = 256, TK = 64;              // tile: B rows (C columns), A rows (C rows), k
constexpr int OP_BYTES = 256 * TK * 2;                  // one operand tile in LDS (32 KiB)
constexpr int STAGE_BYTES = 2 * OP_BYTES;               // B tile | A tile
constexpr int EROW = 512 + 16;                          // epilogue transposition: padded row of 128 floats
constexpr int ESTG = 2048;                              // select epilogue: block-level candidate staging entries
// LDS map: [0, 64 KiB) staging buffer 0 | [64, 128 KiB) staging buffer 1 | epilogue scratch.  The epilogue of a tile runs
// while buffer 0 already receives the next tile's first k-tile, so its scratch starts at 64 KiB (store epilogue:
// 8 waves x 16 rows x 528 B = 66 KiB) or above both buffers (select epilogue: key/index staging, 16 KiB + counters).
constexpr int EPI_BASE_STORE = STAGE_BYTES, EPI_BASE_SELECT = 2 * STAGE_BYTES;
constexpr int BIAS_BASE = EPI_BASE_SELECT + ESTG * 8 + 64;   // two 1 KiB slots: the tile's 256 bias values (tile parity)
constexpr int LDS_BYTES = BIAS_BASE + 2 * 1024;
static_assert(EPI_BASE_STORE + 8 * 16 * EROW <= BIAS_BASE && LDS_BYTES <= 160 * 1024, "LDS");

extern __shared__ __attribute__((aligned(1024))) unsigned char nt_smem[];


// block barrier that orders LDS traffic only (no wait for global loads / stores / DMA pieces in flight)
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// Epilogue 1 ("select"): instead of storing C, every element >= the segment's lower-bound key is appended as
// (order-preserving key, flat index batch*sC + row*ldc + col) to the candidate list of the TopK selection
// (select_lists.hip) — the 6.4 GB pre-activation tensor is then neither written nor re-read.
// one problem of a grouped launch (clt_gemm_bf16_nt_grouped): same M, N, K and strides, own operand / result pointers
struct GemmGroup { long aoff; const uint16_t* b; float* c; };

struct SelEpi {
    const uint32_t* ctl;          // SelCtl records of 4 words: [lb key, candidate count, -, -]
    uint32_t* cand_key; uint32_t* cand_idx; uint32_t capS;
    int seg_per_batch;            // 1: segment = batch index (per-layer selection); 0: one segment
    // Epilogue 3 ("row select", per-row TopK): the lower bound is PER ROW of C (row_lb[row * nbatch + batch], a float) and a
    // candidate goes into its row's slot block: slot_key / slot_idx [rows][slots] (idx = the column), position by a global
    // atomic on slot_cnt[row]; rows = M * nbatch, row index = row * nbatch + batch (the [B][L] order of the pre-activations)
    const float* row_lb; uint32_t* slot_cnt; int slots, nbatch;
};
constexpr bool is_select(int epi) { return epi == 1 || epi == 3; }

// F16: fp16 operands (clt_set_loss_scaling) instead of bf16 — the same kernel around v_mfma_f32_16x16x32_f16
// SPEC (round 3, the store epilogues): wave roles.  vmcnt retires in order, so a wave that has just issued its C stores cannot
// learn that a LATER operand DMA piece has landed without waiting for those stores.  With SPEC the waves 0-3 ("loaders")
// issue ALL operand DMA pieces and never store; the waves 4-7 ("storers") issue ALL C stores — their own accumulators' and,
// handed over through the LDS transposition slabs, their partner wave's — and never wait on vmcnt in the main loop.  All
// eight waves run the same MFMA loop.  Measured (profiles/r03_gemm_ab.md): +3..7 % where the main loop is long (K 2048 ..
// 8192: no wave's DMA wait sits behind a store any more, and the loaders' 16 pieces per k-tile are in flight longer), -2..3 %
// on the K = 768 encoder shapes, whose cost is the store traffic itself (the storers issue twice the stores while the loaders
// idle at the slab barriers).  The host picks the form by K.  (The loaders' piece offsets: one per-lane offset per operand,
// the pieces' row stride in the scalar offset; per-piece clamped offsets only on edge tiles — sixteen per-lane offsets spill.)
// KMA / KMB (round 3, fp32 store epilogue, every wave loading): the A / B operand is K-MAJOR — stored [K][rows], the layout a
// contraction over the outer index of a row-major tensor has (weight gradients: K = tokens; decoder forward: K = features of a
// [F][d] weight) — so those contractions need no transposed 16-bit copy of their operands.  Its tile is staged as 64 k-rows of
// 512 B (a DMA piece = two k-rows), 32-byte column pairs XOR-swizzled with km_f(k) (bits 0, 1 and 3 of k: the eight k-rows a
// 32-lane half of a transposed read touches land in eight different 32-byte bank groups), and the MFMA fragments come out of it
// with ds_read_b64_tr_b16: per 16-lane group a block of 4 k-rows x 16 columns delivered column-major, two reads per fragment
// (k + 0..3, k + 4..7) — the same bytes per fragment as the one b128 read of the k-contiguous image.
// (the compiler puts s_waitcnt vmcnt(0) in front of every ds_read_b64_tr_b16 — the builtin's memory operand carries no alias scope —
// while an LDS-DMA load IT knows of is in flight, which would serialise the operand ring; the k-major instantiations therefore
// issue their DMA pieces through inline asm (m0 = LDS address, as the builtin does) and wait for them with the explicit
// s_waitcnt vmcnt the kernel uses anyway.  The compiler's own vmcnt accounting stays conservative: operations it does not
// know of only make its waits longer.)
__device__ __forceinline__ int km_f(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 rsrc_words(const void* p, int records) {
    const unsigned long a = (unsigned long)p;
    return i32x4{__builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)), __builtin_amdgcn_readfirstlane((int)(a >> 32) & 0xffff), records, 0x00020000};
}
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned char* lds, int voff, int soff) {
    typedef __attribute__((address_space(3))) void lds_void_t;
    const unsigned at = (unsigned)(unsigned long)(lds_void_t*)lds;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(at), "v"(voff), "s"(rsrc), "s"(soff));
}

template <int EPI, bool F16 = false, bool SPEC = false, bool KMA = false, bool KMB = false>
__global__ __launch_bounds__(512) void gemm_nt_bf16_kernel(const uint16_t* __restrict__ A0, const uint16_t* __restrict__ B0,
                                                           float* __restrict__ C0, int M, int N, int K, long lda, long ldb,
                                                           long ldc, long sA, long sB, long sC, const float* __restrict__ bias0,
                                                           long sBias, int gflags, int tilesA, int tilesB, int total, int diag_arg,
                                                           SelEpi se, const uint16_t* const* __restrict__ Bseg, int kts,
                                                           const GemmGroup* __restrict__ grp) {
#ifdef CLT_DIAG
    const int diag = diag_arg;                          // ablation build only (make DIAG=1): see gemm_diag()
#else
    constexpr int diag = 0;
#endif
    // gflags: CLT_GEMM_ACCUMULATE (C +=), CLT_GEMM_ROUND_ACC (the product is rounded to the operand format before the bias is
    // added: the encoder forward of the 16-bit modes, whose einsum result is a 16-bit tensor under the reference's autocast)
    static_assert(!(KMA || KMB) || (EPI == 0 && !SPEC), "k-major operands: fp32 store epilogue, every wave loading");
    const int accumulate = EPI == 0 ? (gflags & CLT_GEMM_ACCUMULATE) : 0;
    const bool racc = EPI != 2 && (gflags & CLT_GEMM_ROUND_ACC) != 0;      // (16-bit C: rounded by the store itself)
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index: scalar
    const int wm = w >> 2, wn = w & 3;                  // wave position: 2 along B rows (128 each), 4 along A rows (64 each)
    const int nwg = tilesA * tilesB;
    const int fr = lane & 15, fq = lane >> 4;
    // ---- virtual block -> (batch, tile) with an XCD-aware, bijective remap -----------------------------
    // The grid is persistent (one block per CU, a multiple of 8): block p walks the virtual blocks p, p + G, ...,
    // so vb & 7 is the block's XCD for every tile it processes.
    struct Tile { int batch, a0, b0; };
    auto decode = [&](int vb) {
        Tile t;
        t.batch = vb / nwg;
        int lid = vb - t.batch * nwg;
        {
            const int xcd = lid & 7, q = nwg >> 3, r = nwg & 7;
            lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lid >> 3);
        }
        // tile order inside an XCD's contiguous range: panels of GA = 4 A-tiles, B-tile index next, A-tile fastest — the
        // 32 blocks an XCD runs together form a 4 x 8 patch (12 distinct operand tiles per k-step instead of 18), and
        // the panel's A rows (4 x 256 rows x K bf16, 1.5 MB at K = 768) stay in the XCD's 4 MB L2 while the B tiles
        // stream (8192^3: 927 -> 1006 TFLOP/s)
        int ta, tb;
        constexpr int GA = 4;
        const int full = tilesA / GA, rem = tilesA - full * GA, per = GA * tilesB;
        if (lid < full * per) { const int pn = lid / per, r = lid - pn * per; tb = r / GA; ta = pn * GA + (r - tb * GA); }
        else { const int r = lid - full * per; tb = r / rem; ta = full * GA + (r - tb * rem); }
        t.a0 = ta * TA; t.b0 = tb * TB;
        return t;
    };

    // ---- staging addresses: per DMA piece a wave fills 8 rows x 128 B ---------------------------
    // B may be K-SEGMENTED (Bseg != nullptr): k-tiles [s*kts, (s+1)*kts) of the contraction come from the matrix at
    // Bseg[s] (same ldb) — the cross-layer decoder's feature gradient sums over target layers, i.e. contracts over
    // (target layer, d_acts) with one weight tensor per target layer, and is then written once instead of accumulated
    // The DMA pieces are MUBUF (buffer_load ... lds), not global_load_lds: with FLAT-encoded LDS-DMA in flight the
    // compiler turns every LDS wait into lgkmcnt(0), which would serialise the pipelined fragment reads below.  Buffer
    // descriptors are rebased per tile (base = first row of the tile), so the per-lane 32-bit offsets stay small.
    int vB[4], vA[4];
    __amdgpu_buffer_rsrc_t rB, rA;
    long brow0 = 0;
    const uint16_t *kbaseA = nullptr, *kbaseB = nullptr;     // k-major operands: first element of the tile's k-row 0
    const bool loader = !SPEC || w < 4;
    bool edge = false;                                  // SPEC: the tile touches the matrix edge (rows clamped piece by piece)
    int eb0 = 0, ea0 = 0;
    // (extent 2 GiB: per-lane offsets stay below it, and an offset with bit 31 set reads as zero — the k tail below)
    auto rsrc = [](const uint16_t* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p), 0, (int)0x80000000u, 0x00020000); };
    // K need only be a multiple of 8: in the last k-tile the lanes whose 16-byte chunk starts at k >= K aim out of the
    // descriptor's range, so zeros land in LDS (a lane's logical chunk is the same for its four rows of a piece)
    const int nk = (diag & 2) ? 0 : (K + TK - 1) / TK;
    const int my_chunk = (lane & 7) ^ ((w * 4 + (lane >> 4)) & 7);
    const int tail_oob = (K % TK != 0 && my_chunk * 8 >= K % TK) ? (int)0x80000000u : 0;
    auto aim = [&](const Tile& t) {
        if constexpr (SPEC) {
            // piece i of loader w covers rows (i * 4 + w) * 8 .. + 7 of both operand tiles: vB[0] / vA[0] hold piece 0's
            // per-lane offset, piece i adds i * 32 rows in the scalar offset
            const int row = w * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            vB[0] = (int)(((long)row * ldb + chunk * 8) * 2);
            vA[0] = (int)(((long)row * lda + chunk * 8) * 2);
            vB[1] = chunk * 16;                             // (edge tiles: the chunk's byte offset alone)
            edge = t.a0 + TA > M || t.b0 + TB > N;
            eb0 = t.b0; ea0 = t.a0;
        } else
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (i * 8 + w) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);       // logical k-chunk stored at this physical slot
            int rb = t.b0 + row; rb = (rb < N ? rb : N - 1) - t.b0; // clamp: rows past the edge are loaded, never stored
            int ra = t.a0 + row; ra = (ra < M ? ra : M - 1) - t.a0;
            // k-major: the lane's slot of piece i * 8 + w is (k-row 2 * piece + lane / 32, physical chunk lane % 32); columns past
            // the edge read the tile's first chunk instead (M / N % 8 == 0: a chunk is inside or outside as a whole)
            const int kin = 2 * (i * 8 + w) + (lane >> 5);
            const int kchunk = (lane & 31) ^ (km_f(kin) << 1);
            if constexpr (KMB) { const int c = t.b0 + kchunk * 8 < N ? kchunk * 8 : 0; vB[i] = (int)(((long)kin * ldb + c) * 2); }
            else vB[i] = (int)(((long)rb * ldb + chunk * 8) * 2);
            if constexpr (KMA) { const int c = t.a0 + kchunk * 8 < M ? kchunk * 8 : 0; vA[i] = (int)(((long)kin * lda + c) * 2); }
            else vA[i] = (int)(((long)ra * lda + chunk * 8) * 2);
        }
        brow0 = KMB ? (long)t.b0 : (long)t.b0 * ldb;
        const uint16_t* ap = A0 + ((!is_select(EPI) && grp) ? grp[t.batch].aoff : (long)t.batch * sA) + (KMA ? (long)t.a0 : (long)t.a0 * lda);
        const uint16_t* bp = ((!is_select(EPI) && grp) ? grp[t.batch].b : B0 + (long)t.batch * sB) + brow0;
        if constexpr (KMA || KMB) { kbaseA = ap; kbaseB = bp; }
        else { rA = rsrc(ap); rB = rsrc(bp); }
    };
    typedef __attribute__((address_space(3))) void lds_void;
    auto stage = [&](int buf, int kt) {
        unsigned char* base = nt_smem + buf * STAGE_BYTES;
        int sob = kt * (TK * 2);
        const int oob = kt == nk - 1 ? tail_oob : 0;
        __amdgpu_buffer_rsrc_t rb = rB;
        if (!is_select(EPI) && Bseg) { const int sg = kt / kts; rb = rsrc(Bseg[sg] + brow0); sob = (kt - sg * kts) * (TK * 2); }
        if constexpr (SPEC) {
            if (!loader) return;
            const int koa = kt * (TK * 2);
            if (!edge) {
                const int sbs = (int)(ldb * 64), sas = (int)(lda * 64);     // 32 rows, in bytes
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(base + (i * 4 + w) * 1024), 16, vB[0] | oob, sob + i * sbs, 0, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void*)(base + OP_BYTES + (i * 4 + w) * 1024), 16, vA[0] | oob, koa + i * sas, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    int r = eb0 + (i * 4 + w) * 8 + (lane >> 3); r = (r < N ? r : N - 1) - eb0;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(base + (i * 4 + w) * 1024), 16, (r * (int)ldb * 2 + vB[1]) | oob, sob, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    int r = ea0 + (i * 4 + w) * 8 + (lane >> 3); r = (r < M ? r : M - 1) - ea0;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void*)(base + OP_BYTES + (i * 4 + w) * 1024), 16, (r * (int)lda * 2 + vB[1]) | oob, koa, 0, 0);
                }
            }
            return;
        }
        if constexpr (KMA || KMB) {
            // k-major operand: the descriptor is rebased to the k-tile's first k-row (a scalar offset of kt * 64 rows can pass
            // 2^31 bytes); k-rows at or past K (last k-tile, K % 64 != 0) aim out of range and land as zeros
            const int kend = (kt == nk - 1 && K % TK != 0) ? K % TK : TK;
            const int kl = lane >> 5;
            const i32x4 wb = rsrc_words(KMB ? kbaseB + (long)kt * TK * ldb : kbaseB, (int)0x80000000u);
            const i32x4 wa = rsrc_words(KMA ? kbaseA + (long)kt * TK * lda : kbaseA, (int)0x80000000u);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = KMB ? (kl >= kend - 2 * (i * 8 + w) ? (int)0x80000000u : 0) : oob;
                dma16(wb, base + (i * 8 + w) * 1024, vB[i] | o, KMB ? 0 : kt * (TK * 2));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = KMA ? (kl >= kend - 2 * (i * 8 + w) ? (int)0x80000000u : 0) : oob;
                dma16(wa, base + OP_BYTES + (i * 8 + w) * 1024, vA[i] | o, KMA ? 0 : kt * (TK * 2));
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(base + (i * 8 + w) * 1024), 16, vB[i] | oob, sob, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void*)(base + OP_BYTES + (i * 8 + w) * 1024), 16, vA[i] | oob, kt * (TK * 2), 0, 0);
    };

    // The tile's bias slice rides along as a ninth DMA piece of wave 0 (columns past N read as zero through the
    // descriptor's bound): the epilogues then read it from LDS, so no global load is pending anywhere near the stores —
    // a pending load makes the compiler put a full vmcnt(0) in front of every store of the epilogue.
    auto stage_bias = [&](const Tile& t, int slot) {
        if (bias0 && w == 0) {
            if constexpr (KMA || KMB) {
                dma16(rsrc_words(bias0 + (long)t.batch * sBias, N * 4), nt_smem + BIAS_BASE + slot * 1024, lane * 16, t.b0 * 4);
                return;
            }
            __amdgpu_buffer_rsrc_t r =     // (not const: a const-qualified resource makes the host pass drop the kernel stub silently)
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias0 + (long)t.batch * sBias), 0, N * 4, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(nt_smem + BIAS_BASE + slot * 1024), 16, (int)((t.b0 + lane * 4) * 4), 0, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes inside an operand tile) ------------------------------------------
    int offB[2], offA[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int phys = (ks * 4 + fq) ^ (fr >> 1);
        offB[ks] = (wm * 128 + fr) * 128 + phys * 16;
        offA[ks] = OP_BYTES + (wn * 64 + fr) * 128 + phys * 16;
    }

    // k-major operand: this lane's address for the transposed reads — lane 4q + p of a 16-lane group supplies k-row
    // fq * 8 + q, columns 4p .. 4p + 3 of the fragment's 16; fragment j of the wave = 32-byte pair (wave's first pair + j), XORed in
    // by the loaders below; + ks * 16 KiB for the k-half, + 2 KiB for the second four k-rows
    int trB = 0, trA = 0;
    {
        const int q = fr >> 2, p = fr & 3, f = q | ((fq & 1) << 2);
        const int in_pair = ((p >> 1) << 4) + ((p & 1) << 3);
        trB = (fq * 8 + q) * 512 + (((wm * 8) ^ f) << 5) + in_pair;
        trA = OP_BYTES + (fq * 8 + q) * 512 + (((wn * 4) ^ f) << 5) + in_pair;
    }
    auto tr8 = [](const unsigned char* p) {
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 2048));
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    bool stores_after_prefetch = false;
    int vb = blockIdx.x;
    Tile t = decode(vb);
    aim(t);
    stage(0, 0);
    int slot = 0;
    stage_bias(t, slot);
    for (;;) {
        f32x4 acc[8][4];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        // first k-tile (staged before the previous tile's epilogue).  Vector-memory operations retire in issue order, so
        // after an interior tile's store epilogue (exactly 32 stores per lane, no loads) the prefetch is complete once all
        // but the 32 youngest operations are: the stores keep draining under the main loop
        if (SPEC) { if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }     // (loaders hold no stores; storers wait for nothing)
        else if (stores_after_prefetch) { if (EPI == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (bare s_barrier: __syncthreads() carries a release fence, i.e. a compiler-inserted vmcnt(0) that would wait for the
        // stores just issued; what the barriers of this kernel order is LDS traffic, waited for explicitly)
        asm volatile("s_barrier" ::: "memory");
        if (nk > 0) {
            // Software-pipelined k loop.  A k-tile is 16 groups of 4 MFMAs (group s: B fragment s & 7 of k-half s >> 3
            // against the 4 A fragments of that half).  Fragment reads run 4 groups ahead of their MFMAs through a ring
            // of 8 B registers and two sets of A registers, and they cross the k-tile boundary: at group 12 every read of
            // the current buffer has been issued, so the wave drains them, the block meets (the next buffer's DMA pieces
            // are complete, this buffer is free), the k-tile after next is staged into this buffer, and the reads of the
            // next k-tile's first groups go out under the last 16 MFMAs of this one.
            bf16x8 fa[2][4], fb[8];
            // (k-major: the fragment's pair index and the staging buffer's offset are XORed into the lane's address as ONE scalar the
            // compiler cannot take apart — left alone it keeps the twelve lane addresses `tr ^ (j << 5)` in registers over the loop,
            // and the instantiation spills; a spill reload in the tile transition waits, in order, behind the DMA prefetch)
            auto km_at = [&](const unsigned char* b, int tr, int j, int ks) {
                int u = (int)(b - nt_smem) | (j << 5);
                asm volatile("" : "+s"(u));
                return nt_smem + ((tr ^ u) + ks * 16384);
            };
            auto ldA = [&](const unsigned char* b, int ks, int n) {
                if constexpr (KMA) return tr8(km_at(b, trA, n, ks));
                else return *reinterpret_cast<const bf16x8*>(b + offA[ks] + n * 2048);
            };
            auto ldB = [&](const unsigned char* b, int ks, int m) {
                if constexpr (KMB) return tr8(km_at(b, trB, m, ks));
                else return *reinterpret_cast<const bf16x8*>(b + offB[ks] + m * 2048);
            };
            if (nk > 1) stage(1, 1);
#pragma unroll
            for (int n = 0; n < 4; ++n) fa[0][n] = ldA(nt_smem, 0, n);
#pragma unroll
            for (int m = 0; m < 4; ++m) fb[m] = ldB(nt_smem, 0, m);
            auto ktile = [&](auto last_tag, int kt) {
                constexpr bool LAST = decltype(last_tag)::value;
                const unsigned char* base = nt_smem + (kt & 1) * STAGE_BYTES;
                const unsigned char* nbase = nt_smem + ((kt & 1) ^ 1) * STAGE_BYTES;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int ks = s >> 3, m = s & 7, r = s + 4;
                    if (r < 16) {
                        if (r == 8) {
#pragma unroll
                            for (int n = 0; n < 4; ++n) fa[1][n] = ldA(base, 1, n);
                        }
                        // (k-major instantiations: the compiler drains every LDS read at the head of the k-tile loop — lgkmcnt(0)
                        // where the k-contiguous form gets the exact count —, so group 0 issues no read in front of that wait and
                        // group 1 issues two fragments)
                        if (!((KMA || KMB) && s == 0)) {
                            if ((KMA || KMB) && s == 1) fb[(r - 1) & 7] = ldB(base, (r - 1) >> 3, (r - 1) & 7);
                            fb[r & 7] = ldB(base, r >> 3, r & 7);
                        }
                    } else if (!LAST) {
                        if (r == 16) {
                            if ((diag & 4) || !loader) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (storers hold no DMA piece; ablation 4: no wait for the DMA — wrong results)
                            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                            if (kt + 2 < nk) stage(kt & 1, kt + 2);
#pragma unroll
                            for (int n = 0; n < 4; ++n) fa[0][n] = ldA(nbase, 0, n);
                        }
                        fb[r & 7] = ldB(nbase, 0, r & 7);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        if constexpr (F16)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fb[m]), __builtin_bit_cast(f16x8, fa[ks][n]), acc[m][n], 0, 0, 0);
                        else
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[m], fa[ks][n], acc[m][n], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            for (int kt = 0; kt + 1 < nk; ++kt) ktile(std::false_type{}, kt);
            ktile(std::true_type{}, nk - 1);
            LDS_BARRIER();                                 // every wave is done reading both buffers
        }
        // Both staging buffers are idle now (the trailing barrier retired every read).  The next tile's first k-tile
        // goes into buffer 0 while this tile's epilogue runs out of the LDS above it.
        const Tile cur = t;
        const int nvb = vb + (int)gridDim.x;
        const bool more = nvb < total;
        if (more) { t = decode(nvb); aim(t); stage(0, 0); stage_bias(t, slot ^ 1); }
        const int batch = cur.batch, a0 = cur.a0, b0 = cur.b0;
        const float* bias = bias0 ? reinterpret_cast<const float*>(nt_smem + BIAS_BASE + slot * 1024) : nullptr;   // [256] in LDS
        slot ^= 1;
        uint32_t bad16 = 0;                              // fp16 mode: some conversion of this tile's epilogue overflowed

        if (is_select(EPI)) {
            // ---- select epilogues: block-level staging in LDS (slot allocation by LDS atomics), one global atomic per block
            // (1: one lower bound per segment) or one per candidate on its row's slot counter (3: per-row lower bounds)
            constexpr bool ROWS = EPI == 3;
            uint32_t* skey = reinterpret_cast<uint32_t*>(nt_smem + EPI_BASE_SELECT);
            uint32_t* sidx = skey + ESTG;
            uint32_t* scnt = sidx + ESTG;                     // [0] staged count, [1] global base
            const int seg = (!ROWS && se.seg_per_batch) ? batch : 0;
            uint32_t* gcnt = ROWS ? nullptr : const_cast<uint32_t*>(se.ctl) + 4 * seg + 1;
            const uint32_t lb = ROWS ? 0u : se.ctl[4 * seg];
            uint32_t* ck = ROWS ? se.cand_key : se.cand_key + (size_t)seg * se.capS;
            uint32_t* ci = ROWS ? se.cand_idx : se.cand_idx + (size_t)seg * se.capS;
            // rows form: a candidate's place = its row's slot block (row index = flat index / N)
            auto to_slot = [&](uint32_t key, uint32_t fi) {
                const uint32_t r = fi / (uint32_t)N, f = fi - r * (uint32_t)N;
                const uint32_t pos = atomicAdd(&se.slot_cnt[r], 1u);
                if (pos < (uint32_t)se.slots) { ck[(size_t)r * se.slots + pos] = key; ci[(size_t)r * se.slots + pos] = f; }
            };
            if (tid == 0) scnt[0] = 0;
            LDS_BARRIER();
            float lbf = key_f32(lb);                          // fast reject in float: !(v < lbf) also lets NaNs through
            if (diag & 32) lbf = INFINITY;                    // (ablation: nothing passes the fast reject)
            if (diag & 16) {                                  // (ablation: no epilogue work at all)
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) asm volatile("" ::"v"(acc[m][n]));
                if (!more) break;
                vb = nvb;
                continue;
            }
            // (B-fragment index outermost: one bias vector live at a time — the accumulators leave few registers)
            uint32_t flat_row[4];
            float rlb[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int row = a0 + wn * 64 + n * 16 + fr;
                flat_row[n] = (uint32_t)((long)batch * sC + (long)row * ldc);
                rlb[n] = ROWS ? (row < M ? se.row_lb[(long)row * se.nbatch + batch] : INFINITY) : lbf;
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int col = b0 + wm * 128 + m * 16 + fq * 4;
                const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + wm * 128 + m * 16 + fq * 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int row = a0 + wn * 64 + n * 16 + fr;
                    f32x4 v = acc[m][n];
                    if (racc) v = round16v<F16>(v, bad16);
                    v += bv;
                    const float lbn = ROWS ? rlb[n] : lbf;
                    const bool in = row < M && col < N;
                    const bool maybe = in && (!(v[0] < lbn) || !(v[1] < lbn) || !(v[2] < lbn) || !(v[3] < lbn));
                    if (__ballot(maybe) == 0ull) continue;      // ~70 % of the 64 x 4 groups hold no candidate (one bound per segment)
                    // rare (about 0.1 % of the elements): each lane appends its own candidates, one LDS atomic apiece
                    if (maybe) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const uint32_t key = f32_key(v[c]);
                            if (ROWS ? !(v[c] < lbn) : key >= lb) {
                                const uint32_t fi = flat_row[n] + (uint32_t)(col + c);
                                const uint32_t off = atomicAdd(&scnt[0], 1u);
                                if (off < (uint32_t)ESTG) { skey[off] = key; sidx[off] = fi; }
                                else if (ROWS) to_slot(key, fi);   // staging full (pathological tile): straight to the global lists
                                else {
                                    const uint32_t pos = atomicAdd(gcnt, 1u);
                                    if (pos < se.capS) { ck[pos] = key; ci[pos] = fi; }
                                }
                            }
                        }
                    }
                }
            }
            LDS_BARRIER();
            const uint32_t staged = min(scnt[0], (uint32_t)ESTG);
            if (ROWS) {
                for (uint32_t i = tid; i < staged; i += 512) to_slot(skey[i], sidx[i]);
            } else {
                if (tid == 0 && staged) scnt[1] = atomicAdd(gcnt, staged);
                LDS_BARRIER();
                const uint32_t gbase = scnt[1];
                for (uint32_t i = tid; i < staged; i += 512) {
                    const uint32_t pos = gbase + i;
                    if (pos < se.capS) { ck[pos] = skey[i]; ci[pos] = sidx[i]; }
                }
            }
        } else if (SPEC && EPI == 2) {
            // ---- 16-bit store epilogue, loader / storer waves: every wave transposes its accumulators through its LDS slab,
            // the storers push their partner's slab and their own out (2 x 4 stores per lane and pass)
            uint16_t* C = reinterpret_cast<uint16_t*>(C0) + (long)batch * sC;
            unsigned char* ebuf = nt_smem + EPI_BASE_STORE + w * (16 * EROW);
            const int er = lane >> 4, ec = (lane & 15) * 8;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    *reinterpret_cast<f32x4*>(ebuf + fr * EROW + (m * 16 + fq * 4) * 4) = acc[m][n];
                LDS_BARRIER();
                if (!loader) {
#pragma unroll
                    for (int half = 0; half < 2; ++half) {        // half 0: the partner loader's slab (B rows 0..127), 1: this wave's own
                        const unsigned char* sbuf = ebuf - (1 - half) * 4 * (16 * EROW);
                        const int col = b0 + half * 128 + ec;
                        f32x4 b0v = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, b1v = b0v;
                        if (bias) { b0v = *reinterpret_cast<const f32x4*>(bias + half * 128 + ec); b1v = *reinterpret_cast<const f32x4*>(bias + half * 128 + ec + 4); }
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int lr = it * 4 + er;
                            const int row = a0 + wn * 64 + n * 16 + lr;
                            const f32x4 v0 = *reinterpret_cast<const f32x4*>(sbuf + lr * EROW + ec * 4) + b0v;
                            const f32x4 v1 = *reinterpret_cast<const f32x4*>(sbuf + lr * EROW + ec * 4 + 16) + b1v;
                            if (row < M && col < N) {
                                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                                typedef __attribute__((address_space(1))) u32x4 gvec;
                                u32x4 o;
                                o[0] = pack16t<F16>(v0[0], v0[1], bad16); o[1] = pack16t<F16>(v0[2], v0[3], bad16);
                                o[2] = pack16t<F16>(v1[0], v1[1], bad16); o[3] = pack16t<F16>(v1[2], v1[3], bad16);
                                __builtin_nontemporal_store(o, (gvec*)(C + (long)row * ldc + col));
                            }
                        }
                    }
                }
                if (n < 3) LDS_BARRIER();                     // (after the last pass: the next tile's first barrier)
            }
        } else if (SPEC && EPI == 0) {
            // ---- fp32 store epilogue, loader / storer waves (2 x 8 stores per storer lane and pass)
            float* C = grp ? grp[batch].c : C0 + (long)batch * sC;
            unsigned char* ebuf = nt_smem + EPI_BASE_STORE + w * (16 * EROW);
            const int er = lane >> 5, ec = (lane & 31) * 4;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    *reinterpret_cast<f32x4*>(ebuf + fr * EROW + (m * 16 + fq * 4) * 4) = acc[m][n];
                LDS_BARRIER();
                if (!loader) {
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const unsigned char* sbuf = ebuf - (1 - half) * 4 * (16 * EROW);
                        const int col = b0 + half * 128 + ec;
                        f32x4 bv = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                        if (bias) bv = *reinterpret_cast<const f32x4*>(bias + half * 128 + ec);
#pragma unroll
                        for (int it = 0; it < 8; ++it) {
                            const int lr = it * 2 + er;
                            const int row = a0 + wn * 64 + n * 16 + lr;
                            f32x4 v = *reinterpret_cast<const f32x4*>(sbuf + lr * EROW + ec * 4);
                            if (row < M && col < N) {
                                typedef __attribute__((address_space(1))) f32x4 gvec;
                                gvec* dst = (gvec*)(C + (long)row * ldc + col);
                                if (racc) v = round16v<F16>(v, bad16);
                                v += bv;
                                if (accumulate) { const f32x4 old = *dst; v += old; }
                                __builtin_nontemporal_store(v, dst);
                            }
                        }
                    }
                }
                if (n < 3) LDS_BARRIER();
            }
        } else if (EPI == 2) {
            // ---- bf16 store epilogue: C leaves as bf16 (half the bytes AND half the store instructions of the fp32 form:
            // the K = 768 contractions of the step are bound by pushing the tile out of the CU).  Same LDS transposition;
            // a lane then takes 8 consecutive columns of one row (two b128 reads), adds the bias, rounds to nearest even
            // and stores 16 bytes: 256-byte row segments, 4 rows per wave-instruction, exactly 16 stores per lane and tile.
            uint16_t* C = reinterpret_cast<uint16_t*>(C0) + (long)batch * sC;
            unsigned char* ebuf = nt_smem + EPI_BASE_STORE + w * (16 * EROW);
            const int er = lane >> 4, ec = (lane & 15) * 8;
            const int col = b0 + wm * 128 + ec;
            f32x4 b0v = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, b1v = b0v;
            if (bias) { b0v = *reinterpret_cast<const f32x4*>(bias + wm * 128 + ec); b1v = *reinterpret_cast<const f32x4*>(bias + wm * 128 + ec + 4); }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    *reinterpret_cast<f32x4*>(ebuf + fr * EROW + (m * 16 + fq * 4) * 4) = acc[m][n];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int lr = it * 4 + er;
                    const int row = a0 + wn * 64 + n * 16 + lr;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(ebuf + lr * EROW + ec * 4) + b0v;
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(ebuf + lr * EROW + ec * 4 + 16) + b1v;
                    if (row < M && col < N) {
                        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                        typedef __attribute__((address_space(1))) u32x4 gvec;
                        u32x4 o;
                        o[0] = pack16t<F16>(v0[0], v0[1], bad16); o[1] = pack16t<F16>(v0[2], v0[3], bad16);
                        o[2] = pack16t<F16>(v1[0], v1[1], bad16); o[3] = pack16t<F16>(v1[2], v1[3], bad16);
#if GEMM_NT_STORE
                        __builtin_nontemporal_store(o, (gvec*)(C + (long)row * ldc + col));
#else
                        *(gvec*)(C + (long)row * ldc + col) = o;
#endif
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            // ---- store epilogue -------------------------------------------------------------------------------
            // A lane holds C[a-row fr][4 consecutive columns] of each 16x16 tile, i.e. 64-byte runs per row.  Each wave
            // transposes its 64 x 128 sub-tile through a private LDS region (four passes of 16 rows, 528-byte padded
            // rows: conflict-free b128 writes) and stores 512-byte contiguous row segments: 2 rows per wave-instruction.
            float* C = grp ? grp[batch].c : C0 + (long)batch * sC;
            unsigned char* ebuf = nt_smem + EPI_BASE_STORE + w * (16 * EROW);
            const int er = lane >> 5, ec = (lane & 31) * 4;
            const int col = b0 + wm * 128 + ec;
            f32x4 bv = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (bias) bv = *reinterpret_cast<const f32x4*>(bias + wm * 128 + ec);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    *reinterpret_cast<f32x4*>(ebuf + fr * EROW + (m * 16 + fq * 4) * 4) = acc[m][n];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int lr = it * 2 + er;
                    const int row = a0 + wn * 64 + n * 16 + lr;
                    f32x4 v = *reinterpret_cast<const f32x4*>(ebuf + lr * EROW + ec * 4);
                    if (row < M && col < N) {
                        // (global address space spelled out: C may come from the group table, and generic-pointer stores
                        // would be flat_store — slower, and outside the in-order vmcnt accounting used at the loop top)
                        typedef __attribute__((address_space(1))) f32x4 gvec;
                        gvec* dst = (gvec*)(C + (long)row * ldc + col);
                        if (racc) v = round16v<F16>(v, bad16);
                        v += bv;
                        if (accumulate) { const f32x4 old = *dst; v += old; }
#if GEMM_NT_STORE
                        if (diag & 1) { asm volatile("" ::"v"(v)); } else __builtin_nontemporal_store(v, dst);
#else
                        if (diag & 1) { asm volatile("" ::"v"(v)); } else *dst = v;
#endif
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (F16) clt_raise_inf_flag(bad16);
        if (!more) break;
        stores_after_prefetch = !is_select(EPI) && !accumulate && !(diag & 1) && a0 + TA <= M && b0 + TB <= N;
        vb = nvb;
    }
}

// Ablation switches of the kernel: compiled in only by `make DIAG=1` (-DCLT_DIAG), where CLT_GEMM_DIAG is read once from the
// environment: 1 no stores, 2 no k loop, 4 no wait for the DMA pieces (wrong results), 16 no select epilogue, 32 nothing
// passes the select epilogue's fast reject.  The shipped library ignores the variable (diag is the constant 0).
int gemm_diag() {
#ifdef CLT_DIAG
    static const int d = getenv("CLT_GEMM_DIAG") ? atoi(getenv("CLT_GEMM_DIAG")) : 0;
    return d;
#else
    return 0;
#endif
}

// per-lane DMA offsets are 32-bit byte offsets from the tile's first row (+ the k offset in the scalar offset)
bool dma_offsets_fit(long lda, long ldb, long K) { return (255 * (lda > ldb ? lda : ldb) + K) * 2 < 0x7fffffffL; }

// > 0: the persistent grid takes at most this many blocks (a multiple of 8): a launch that leaves the other CUs to a
// bandwidth-bound kernel running beside it on another stream (clt_gemm_nt_grid_limit)
int g_grid_limit = 0;
// persistent grid: one block per CU (a multiple of 8, so that a block keeps its XCD over all its tiles)
int persistent_grid(long nblk) {
    static int cus = 0;
    if (!cus) {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) cus = 256;
        else cus = pr.multiProcessorCount >= 8 ? pr.multiProcessorCount / 8 * 8 : 8;
    }
    const int lim = g_grid_limit > 0 && g_grid_limit < cus ? g_grid_limit : cus;
    return (int)(nblk < lim ? nblk : lim);
}
// which form of the kernel the store epilogues launch: 1 = every wave loads and stores (rounds 1-2), 2 = loader / storer
// waves, 0 = by shape: the loader / storer form for 2048 <= K <= 8192 (measured, profiles/r03_gemm_ab.md: 8192^3 1372 ->
// 1464 TFLOP/s, 4096^3 1283 -> 1319, the K = 4096 weight gradient 1194 -> 1234; the K = 768 encoder shapes 840 -> 819 and
// the K = 32768 decoder forward 960 -> 933 are better off with every wave doing both)
int g_nt_variant = 0;
}  // namespace

// test / A-B hook: see include/clt_hip.h
extern "C" int clt_gemm_nt_variant(int variant, int reserved) {
    const int old = g_nt_variant;
    if (variant >= 0 && variant <= 2) g_nt_variant = variant;
    (void)reserved;
    return old;
}

// see include/clt_hip.h
extern "C" int clt_gemm_nt_grid_limit(int blocks) {
    const int old = g_grid_limit;
    if (blocks >= 0) g_grid_limit = blocks / 8 * 8;
    return old;
}

// (the fp16 store / rounding epilogues raise the loss scaler's overflow flag: this unit needs the state too)
CLT_DEFINE_TU_STATE_SETTER(clt_internal_set_state_gemm_nt)

// Returns CLT_OK when the fast kernel was launched, 1 when the shape/alignment does not qualify (caller falls back
// to the generic kernel), or a negative error.
int clt_internal_gemm_nt_bf16(const uint16_t* A, const uint16_t* B, float* C, int M, int N, int K, long lda, long ldb, long ldc,
                              int batch, long sA, long sB, long sC, const float* bias, long sBias, int accumulate,
                              hipStream_t st, const uint16_t* const* Bseg, int kts, const void* groups, bool out16, int km) {
    const GemmGroup* grp = static_cast<const GemmGroup*>(groups);
    // km: bit 0 = A is k-major ([K][M], row stride lda), bit 1 = B is k-major ([K][N], row stride ldb)
    const bool kma = km & 1, kmb = km & 2;
    if (km && (out16 || Bseg || (kma && M % 8) || (kmb && N % 8))) return 1;
    // bf16 result (C is then a bf16 array; ldc / sC in bf16 elements): 8-column alignment, no accumulate, no groups
    if (out16 && ((accumulate & CLT_GEMM_ACCUMULATE) || grp || N % 8 || ldc % 8 || sC % 8)) return 1;
    const bool ok = K % 8 == 0 && (!Bseg || (K % TK == 0 && kts > 0)) && N % 4 == 0 && ldc % 4 == 0 && lda % 8 == 0 && ldb % 8 == 0 && sA % 8 == 0 && sB % 8 == 0 &&
                    sC % 4 == 0 && sBias % 4 == 0 && aligned16(A) && (Bseg || grp || aligned16(B)) && (grp || aligned16(C)) && (!bias || aligned16(bias)) &&
                    (long)M * N >= 256L * 256L && dma_offsets_fit(kma ? 0 : lda, kmb ? 0 : ldb, K) && lda < (1L << 23) && ldb < (1L << 23);
    if (!ok) return 1;
    const int tilesA = (M + TA - 1) / TA, tilesB = (N + TB - 1) / TB;
    const long nblk = (long)tilesA * tilesB * batch;
    if (nblk > 0x7fffffffL) return 1;
    const unsigned grid = (unsigned)persistent_grid(nblk);
    const int diag = gemm_diag();
#define NT_GO(EPI_, F16_, SPEC_)                                                                                              \
    do {                                                                                                                        \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_bf16_kernel<EPI_, F16_, SPEC_>),                          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return CLT_ERR_LAUNCH;    \
        hipLaunchKernelGGL((gemm_nt_bf16_kernel<EPI_, F16_, SPEC_>), dim3(grid), dim3(512), LDS_BYTES, st, A, B, C, M, N, K, lda, ldb, ldc, \
                           sA, sB, sC, bias, sBias, EPI_ == 2 ? 0 : accumulate, tilesA, tilesB, (int)nblk, diag, SelEpi{}, Bseg, kts,       \
                           EPI_ == 2 ? (const GemmGroup*)nullptr : grp);                                                        \
    } while (0)
    const bool spec = g_nt_variant == 2 || (g_nt_variant == 0 && K >= 2048 && K <= 8192);
    const bool f16 = clt_host_state.fp16 != 0;
#define NT_GOK(F16_, KMA_, KMB_)                                                                                               \
    do {                                                                                                                        \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_bf16_kernel<0, F16_, false, KMA_, KMB_>),                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return CLT_ERR_LAUNCH;    \
        hipLaunchKernelGGL((gemm_nt_bf16_kernel<0, F16_, false, KMA_, KMB_>), dim3(grid), dim3(512), LDS_BYTES, st, A, B, C, M, N, K, lda, ldb, \
                           ldc, sA, sB, sC, bias, sBias, accumulate, tilesA, tilesB, (int)nblk, diag, SelEpi{}, Bseg, kts, grp);  \
    } while (0)
    if (km) {
        if (f16) { if (kma && kmb) NT_GOK(true, true, true); else if (kmb) NT_GOK(true, false, true); else NT_GOK(true, true, false); }
        else { if (kma && kmb) NT_GOK(false, true, true); else if (kmb) NT_GOK(false, false, true); else NT_GOK(false, true, false); }
    } else
    if (out16) {
        if (f16) { if (spec) NT_GO(2, true, true); else NT_GO(2, true, false); }
        else { if (spec) NT_GO(2, false, true); else NT_GO(2, false, false); }
    } else {
        if (f16) { if (spec) NT_GO(0, true, true); else NT_GO(0, true, false); }
        else { if (spec) NT_GO(0, false, true); else NT_GO(0, false, false); }
    }
#undef NT_GO
#undef NT_GOK
    CLT_LAUNCH_CHECK();
    return CLT_OK;
}

// NT GEMM whose B operand is K-segmented: C[M][N] (+)= sum_s A[:, s*kseg:(s+1)*kseg] * Bseg[s][N][kseg]^T.  Bseg is a DEVICE array
// of nseg device pointers (each 16-byte aligned, row stride ldb).  Fast kernel only: returns 1 when the shape does not qualify.
extern "C" int clt_gemm_bf16_nt_kseg(const uint16_t* A, const uint16_t* const* Bseg, float* C, int M, int N, int nseg, int kseg, long lda,
                                     long ldb, long ldc, const float* bias, int accumulate, void* stream) {
    if (M <= 0 || N <= 0 || nseg <= 0 || kseg <= 0 || !A || !Bseg || !C) return CLT_ERR_SHAPE;
    if (kseg % TK) return 1;
    return clt_internal_gemm_nt_bf16(A, nullptr, C, M, N, nseg * kseg, lda, ldb, ldc, 1, 0, 0, 0, bias, 0, accumulate, as_stream(stream),
                                     Bseg, kseg / TK);
}

// NT GEMMs whose result leaves as bf16 (C: bf16 [M][N], row stride ldc, batch stride sC, in bf16 elements): the dense bf16
// regime's pre-activations and feature gradients are consumed once, elementwise, by kernels that round to bf16 anyway, and at
// K = 768 the fp32 store is what bounds the kernel.  Fast kernel only: 1 when the shape does not qualify (N % 8, ldc % 8, ...).
extern "C" int clt_gemm_bf16_nt_out16(const uint16_t* A, const uint16_t* B, uint16_t* C, int M, int N, int K, long lda, long ldb, long ldc,
                                      int batch, long sA, long sB, long sC, const float* bias, long sBias, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || !A || !B || !C) return CLT_ERR_SHAPE;
    return clt_internal_gemm_nt_bf16(A, B, reinterpret_cast<float*>(C), M, N, K, lda, ldb, ldc, batch, sA, sB, sC, bias, sBias, 0,
                                     as_stream(stream), nullptr, 0, nullptr, true);
}
extern "C" int clt_gemm_bf16_nt_kseg_out16(const uint16_t* A, const uint16_t* const* Bseg, uint16_t* C, int M, int N, int nseg, int kseg,
                                           long lda, long ldb, long ldc, void* stream) {
    if (M <= 0 || N <= 0 || nseg <= 0 || kseg <= 0 || !A || !Bseg || !C) return CLT_ERR_SHAPE;
    if (kseg % TK) return 1;
    return clt_internal_gemm_nt_bf16(A, nullptr, reinterpret_cast<float*>(C), M, N, nseg * kseg, lda, ldb, ldc, 1, 0, 0, 0, nullptr, 0, 0,
                                     as_stream(stream), Bseg, kseg / TK, nullptr, true);
}

// Grouped NT GEMM: ngroups problems of one shape in ONE persistent launch, C_g[M][N] = (A + aoff_g)[M][K] * B_g[N][K]^T.
// groups: DEVICE array of {long aoff (elements); const uint16_t* B; float* C} records (all 16-byte aligned).
extern "C" int clt_gemm_bf16_nt_grouped(const uint16_t* A, const void* groups, int ngroups, int M, int N, int K, long lda, long ldb,
                                        long ldc, int kmajor, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || ngroups <= 0 || !A || !groups) return CLT_ERR_SHAPE;
    return clt_internal_gemm_nt_bf16(A, nullptr, nullptr, M, N, K, lda, ldb, ldc, ngroups, 0, 0, 0, nullptr, 0, 0, as_stream(stream), nullptr, 0,
                                     groups, false, kmajor & 3);
}

// C is never written: candidates (key >= ctl[seg].lb) go to the selection workspace.  ldc / sC only define the flat
// index of an element.  Returns 1 when the shape does not qualify for the fast kernel.
extern "C" int clt_gemm_bf16_select(const uint16_t* A, const uint16_t* B, int M, int N, int K, long lda, long ldb, long ldc, int batch,
                                    long sA, long sB, long sC, const float* bias, long sBias, const uint32_t* ctl,
                                    uint32_t* cand_key, uint32_t* cand_idx, uint32_t capS, int seg_per_batch, int flags, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || !ctl || !cand_key || !cand_idx || capS == 0) return CLT_ERR_SHAPE;
    const bool ok = K % 8 == 0 && N % 4 == 0 && lda % 8 == 0 && ldb % 8 == 0 && sA % 8 == 0 && sB % 8 == 0 && sBias % 4 == 0 &&
                    aligned16(A) && aligned16(B) && (!bias || aligned16(bias)) && dma_offsets_fit(lda, ldb, K);
    if (!ok) return 1;
    if ((unsigned long long)(batch - 1) * sC + (unsigned long long)(M - 1) * ldc + N > 0xffffffffull) return CLT_ERR_RANGE;
    const int tilesA = (M + TA - 1) / TA, tilesB = (N + TB - 1) / TB;
    const long nblk = (long)tilesA * tilesB * batch;
    if (nblk > 0x7fffffffL) return 1;
    if (clt_host_state.fp16) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_bf16_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS_BYTES) != hipSuccess) return CLT_ERR_LAUNCH;
        hipLaunchKernelGGL((gemm_nt_bf16_kernel<1, true>), dim3((unsigned)persistent_grid(nblk)), dim3(512), LDS_BYTES, as_stream(stream), A, B,
                           (float*)nullptr, M, N, K, lda, ldb, ldc, sA, sB, sC, bias, sBias, flags & CLT_GEMM_ROUND_ACC, tilesA, tilesB, (int)nblk, gemm_diag(),
                           SelEpi{ctl, cand_key, cand_idx, capS, seg_per_batch, nullptr, nullptr, 0, 0}, (const uint16_t* const*)nullptr, 0, (const GemmGroup*)nullptr);
        CLT_LAUNCH_CHECK();
        return CLT_OK;
    }
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_bf16_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_BYTES) != hipSuccess) return CLT_ERR_LAUNCH;
    hipLaunchKernelGGL(gemm_nt_bf16_kernel<1>, dim3((unsigned)persistent_grid(nblk)), dim3(512), LDS_BYTES, as_stream(stream), A, B,
                       (float*)nullptr, M, N, K, lda, ldb, ldc, sA, sB, sC, bias, sBias, flags & CLT_GEMM_ROUND_ACC, tilesA, tilesB, (int)nblk,
                       gemm_diag(),
                       SelEpi{ctl, cand_key, cand_idx, capS, seg_per_batch, nullptr, nullptr, 0, 0}, (const uint16_t* const*)nullptr, 0,
                       (const GemmGroup*)nullptr);
    CLT_LAUNCH_CHECK();
    return CLT_OK;
}

// Per-row form: C is never written; element (batch, row, col) is a candidate when it is >= row_lb[row * batch_count + batch]
// and goes into the slot block of row index row * batch_count + batch.  Requires sC == N and ldc == batch * N (the [B][L][F]
// layout: flat index / N = the row index).  Returns 1 when the shape does not qualify for the fast kernel.
extern "C" int clt_gemm_bf16_select_rows(const uint16_t* A, const uint16_t* B, int M, int N, int K, long lda, long ldb, int batch,
                                         long sA, long sB, const float* bias, long sBias, const float* row_lb, uint32_t* slot_cnt,
                                         uint32_t* slot_key, uint32_t* slot_idx, int slots, int flags, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || !row_lb || !slot_cnt || !slot_key || !slot_idx || slots <= 0) return CLT_ERR_SHAPE;
    const bool ok = K % 8 == 0 && N % 4 == 0 && lda % 8 == 0 && ldb % 8 == 0 && sA % 8 == 0 && sB % 8 == 0 && sBias % 4 == 0 &&
                    aligned16(A) && aligned16(B) && (!bias || aligned16(bias)) && dma_offsets_fit(lda, ldb, K);
    if (!ok) return 1;
    const long ldc = (long)batch * N, sC = N;
    if ((unsigned long long)M * batch * N > 0xffffffffull) return CLT_ERR_RANGE;
    const int tilesA = (M + TA - 1) / TA, tilesB = (N + TB - 1) / TB;
    const long nblk = (long)tilesA * tilesB * batch;
    if (nblk > 0x7fffffffL) return 1;
    const SelEpi se{nullptr, slot_key, slot_idx, 0u, 0, row_lb, slot_cnt, slots, batch};
#define NT_ROWS(F16_)                                                                                                           \
    do {                                                                                                                        \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_bf16_kernel<3, F16_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                LDS_BYTES) != hipSuccess) return CLT_ERR_LAUNCH;                                                \
        hipLaunchKernelGGL((gemm_nt_bf16_kernel<3, F16_>), dim3((unsigned)persistent_grid(nblk)), dim3(512), LDS_BYTES, as_stream(stream), A, B, \
                           (float*)nullptr, M, N, K, lda, ldb, ldc, sA, sB, sC, bias, sBias, flags & CLT_GEMM_ROUND_ACC, tilesA, tilesB, (int)nblk, \
                           gemm_diag(), se, (const uint16_t* const*)nullptr, 0, (const GemmGroup*)nullptr);                     \
    } while (0)
    if (clt_host_state.fp16) NT_ROWS(true); else NT_ROWS(false);
#undef NT_ROWS
    CLT_LAUNCH_CHECK();
    return CLT_OK;
}
