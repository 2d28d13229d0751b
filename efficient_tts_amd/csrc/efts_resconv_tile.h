// efts_resconv_tile.h -- device side of efts_resconv5 (see efts_resconv.hip for the design): launch structures, the tile function of the
// 8-wave ping-pong kernel with its epilogue variants, and the kernel template.  Included by efts_resconv.hip (forward / plain layers:
// MODE 0) and efts_resconv_bwd.hip (the training backward's dgrad layer with the fused activation backward: MODE 1) -- two translation
// units, because every variant inlined into one kernel pushed the register allocator into scratch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "efts_mma.h"

// lab builds only (-DRC_EXP=...): ablation bits 1 no LDS-DMA in the loop, 4 no MFMA + fragment reads, 8 no epilogue loads / stores,
// 16 no epilogue at all (tools/rc_ab.sh)
#ifndef RC_EXP
#define RC_EXP 0
#endif
// lab builds only (-DRC_STAMP=1): every workgroup stamps its start and end with the 100 MHz constant clock into the buffer whose
// address EFTS_RC_STAMP (hex) names, 64 launches deep: dispatch skew, kernel span and the idle gap between dependent launches.
// -DRC_STAMP=2: every wave sums the shader-clock cycles of its read phases, MFMA phases and the waits at their barriers
#ifndef RC_STAMP
#define RC_STAMP 0
#endif
#ifndef RC_LAB_MIN
#define RC_LAB_MIN 0
#endif

namespace efts {

constexpr int RC_BN = 256;                                   // output columns per tile
constexpr int RC_WIN_BYTES = 256 * 128;                      // one window buffer (256 rows x 128 B)
constexpr int RC_W_BYTES = RC_BN * 128;                      // one weight tile
constexpr int RC_RING = 2 * RC_WIN_BYTES;                    // LDS offset of the weight ring
constexpr int RC_LDS = 2 * RC_WIN_BYTES + 3 * RC_W_BYTES;    // 163840 = all of a CU's LDS
constexpr int RC_MAXCLS = 4, RC_MAXTILES = 8;

struct RcSched {
    int ncls;                                // workgroup group g belongs to class g % ncls
    int rows[RC_MAXCLS];                     // output rows a group of this class owns
    int ntile[RC_MAXCLS];
    unsigned char ni[RC_MAXCLS][RC_MAXTILES];   // tile heights in half units of 32 window rows, 2..8 (a tile yields 32 * h - 4 rows)
};

// one residual layer (a "problem" of a launch): every pointer and stride that differs between the layers of a grouped launch
struct RcProb {
    const char* a;          // operand plane of x (split 1: bf16 hi; split 2: [32 hi | 32 lo] chunks), row 0
    const char* a_lo;       // split 1: lo plane of x (same layout) or null
    const float* resid;     // fp32 x (takes precedence over the planes as the residual) or null
    const char* w;
    const float* bias;
    const float* rowmask;
    float* out_f32;
    char* ob;
    char* ob_lo;
    long lda, ldw, w_tap_stride, ldr, ldo, ldob;
    int m;                  // rows of this layer
    int out_split;
    float slope;
    int taps;               // 5 | 3
    int no_resid;           // 1: y = act(conv + bias) * mask (no residual term)
    int ldsg;               // bytes per row of `sign` (n / 8)
    char* sign;             // training: sign bits of the activated conv output (bit j of byte c = column 8 c + j), or null
    // training backward, RC_EPI_DGRAD_ACT: this layer is the dgrad G' = (G + conv_T(dZ)) * mask of stack layer l, and its epilogue also runs the
    // activation backward of layer l - 1 on G': dZ' = G' * (sign_in ? 1 : slope_bwd) -> the operand plane `ob`, column sums -> bias_part
    const char* sign_in;    // sign bits layer l - 1's forward wrote (same layout as `sign`)
    float* bias_part;       // [groups * bp_tiles * 2][n] partial column sums of dZ' (plain stores; rows of tiles that do not exist stay zero)
    float slope_bwd;
};
constexpr int RC_EPI_GENERAL = 0, RC_EPI_FAST = 1, RC_EPI_DGRAD_ACT = 2;

constexpr int RC_MAXPROB = 2;

// A launch covers the rows of up to RC_MAXPROB layers laid end to end ("virtual rows": layer 0 first); the schedule cuts the
// virtual rows into groups and tiles, and a tile never crosses from one layer into the next.
struct RcArgs {
    RcProb pr[RC_MAXPROB];
    int nprob;
    int m;                  // virtual rows = sum of pr[i].m
    int nchunk, ntn;
    RcSched s;
    unsigned long long* stamp;
    int bp_tiles;           // RC_EPI_DGRAD_ACT: rows of `bias_part` per group = 2 * bp_tiles (tile t of group g, wave row wm: row (g * bp_tiles + t) * 2 + wm)
};

__device__ __forceinline__ void rc_wait(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}

template <int V> struct RcRow { static constexpr int value = V; };

struct RcCtx {
    char* smem;
    unsigned lds0;
    int lane, wave, wm, wn, lrow, lhalf;
    int n0;
    unsigned vow[4];        // per-lane source offsets of this wave's 4 weight pieces (fixed per workgroup)
    const char* w_base;       // weights of the current tile's layer, this workgroup's columns
    const char* w_next;       // ... of the next tile's layer (the weight requests two steps ahead wrap into the next tile)
    long wts, wts_next;       // tap strides of the two
    float bv[2];            // bias of this lane's two accumulator columns
    int ws;                 // ring slot of the next step
    int wpar;               // window buffer of the next tile's chunk 0
    unsigned long long tph[4];   // RC_STAMP 2: shader-clock cycles in read phases / their barrier / MFMA phases / their barrier
    int nst;                // RC_STAMP 3: marks written so far
    int bp_row;             // RC_EPI_DGRAD_ACT: this wave row's row of `bias_part` for the current tile
};
// -DRC_STAMP=3: wave 0 of every workgroup marks kernel start, end of tile 0's main loop, end of tile 0's epilogue, kernel end (100 MHz clock)
#define RC_MARK(p, c) do { if (RC_STAMP == 3 && (p).stamp && (c).wave == 0 && (c).nst < 4) (p).stamp[blockIdx.x * 4 + (c).nst++] = __builtin_amdgcn_s_memrealtime(); } while (0)

// window pieces of one tile of h half units: piece P = q * 8 + wave, P < 4 h, covers window rows 8P .. 8P+7 (lane l: row 8P + l/8,
// physical slot l%8); rows past the guard band after the matrix are clamped (their outputs are never stored)
__device__ __forceinline__ int rc_pieces(int h, int wave) { return (4 * h - wave + 7) >> 3; }

// One (32 * h) x 256 tile at output row m0, as seen by one wave: NI = the 32-row blocks of its wave row (ceil(h/2) for wm = 0,
// floor(h/2) for wm = 1; waves of the two rows run different instantiations with the same barrier sequence).  Main loop over
// (chunk, tap) steps, then the fused epilogue.
// The operand streams are CONTINUOUS across the tiles of a workgroup: its weights do not depend on the tile (n0 is fixed), so
// the weight requests two steps ahead simply wrap into the next tile's steps 0 and 1, and the next tile's first window
// (rows m1, height h1; 0 = no next tile) is requested at the first tap of this tile's last chunk.  On entry the window of
// chunk 0 and the weights of steps 0 and 1 are therefore in flight or landed.
// FAST: a layer in the middle of a stack on planes -- residual from the planes (split 1: hi + lo planes; split 2: the [hi | lo] chunks),
// row mask, planes out in the same format, nothing else -- with every per-layer switch of the epilogue a compile-time constant.  Read from
// the argument segment inside the sweep those switches cost a unit (32 rows x 64 columns) ~850 instructions incl. 24 scalar loads and ~50
// uniform branches; as constants ~380 (round 4; the sweep is bound by instruction issue, not by its bytes).
template <int SPLIT, int NI, int TAPS, int EPI>
__device__ __forceinline__ void rc_tile(const RcArgs& p, const RcProb& pq, const RcProb& pn, RcCtx& c, int m0, int h, int rows_out, int m1, int h1) {
    // TAPS 5 or 3: a k3 layer keeps the k5 geometry (window from row m0 - 2, 32 h - 4 output rows per tile) and simply reads window
    // rows r + k + 1 for its three taps; fewer steps per chunk, everything else -- streams, barriers, epilogue -- is the same code
    char* const smem = c.smem;
    const int lane = c.lane, wave = c.wave, lrow = c.lrow, lhalf = c.lhalf, wm = c.wm, wn = c.wn;
    const int row0w = wm ? 32 * ((h + 1) >> 1) : 0;         // first tile row of this wave row
    const int nq = rc_pieces(h, wave), nq1 = rc_pieces(h1, wave);      // window pieces this wave requests (this tile / the next)

    unsigned voa[4];
    {
        const int rmax = pq.m + 143 - (m0 - 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = (q * 8 + wave) * 8 + (lane >> 3);
            const int sl = (lane & 7) ^ ((r >> 1) & 7);
            voa[q] = (unsigned)((r < rmax ? r : rmax) * (int)pq.lda + (sl << 4));
        }
    }
    const char* a_base = pq.a + (long)(m0 - 2) * pq.lda;
    auto issue_a = [&](int cn, int buf) {
        const char* sb = a_base + (long)cn * 128;
        const unsigned l = c.lds0 + buf * RC_WIN_BYTES + wave * 1024;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < nq) dma16(l + q * 8192, voa[q], sb);
    };
    auto issue_a_next = [&](int buf) {                     // chunk 0 of the next tile (its own height and rows)
        const int rmax = pn.m + 143 - (m1 - 2);
        const char* sb = pn.a + (long)(m1 - 2) * pn.lda;
        const unsigned l = c.lds0 + buf * RC_WIN_BYTES + wave * 1024;
        for (int q = 0; q < nq1; ++q) {
            const int r = (q * 8 + wave) * 8 + (lane >> 3);
            const int sl = (lane & 7) ^ ((r >> 1) & 7);
            dma16(l + q * 8192, (unsigned)((r < rmax ? r : rmax) * (int)pn.lda + (sl << 4)), sb);
        }
    };

    f32x16 acc[NI][2];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- the tile's first operands were requested earlier (kernel start / previous tile); the previous epilogue's loads
    // and stores share the counter, so everything is waited for once here
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int ws = c.ws;
    // ---- ping-pong main loop.  Row 0: [L0] B [M0] B [L1] B [M1, wait] B.  Row 1: B [L0] B [M0] B [L1, wait] B [M1].  (L = fragment reads
    // of a half step + its LDS-DMA pieces, M = its MFMAs, B = s_barrier of all 8 waves.)  Between two barriers exactly one wave of
    // every SIMD issues MFMAs.  Ordering: every wave counts its own requests down (rc_wait) before the step's 4th barrier, so
    // the next step's weights / window are readable right behind it (row 0) or one barrier later (row 1); every fragment read of a
    // step is retired (lgkmcnt 0) before that same barrier, so the requests issued behind it may overwrite what the step read.
    bf16x8 fa[2][NI], fa2[SPLIT == 2 ? NI : 1], fb[2][2], fb2[SPLIT == 2 ? 2 : 1];
    (void)fa2; (void)fb2;
    const int brow_pp = wn * 64 + lrow;
    auto loadH = [&](const char* at, const char* wt, int arow, int hs) {
        if (RC_EXP & 4) return;
        if constexpr (SPLIT == 1) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int slot16 = (2 * hs + s2) * 2 + lhalf;
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[s2][j] = *(const bf16x8*)(wt + lds_off(brow_pp + j * 32, slot16));
#pragma unroll
                for (int i = 0; i < NI; ++i) fa[s2][i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot16));
            }
        } else {
            const int slot16 = hs * 2 + lhalf;              // fa[0] / fb[0] = hi, fa2 / fb2 = lo of k-slice hs
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                fb[0][j] = *(const bf16x8*)(wt + lds_off(brow_pp + j * 32, slot16));
                fb2[j] = *(const bf16x8*)(wt + lds_off(brow_pp + j * 32, slot16 + 4));
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                fa[0][i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot16));
                fa2[i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot16 + 4));
            }
        }
    };
    auto mmaH = [&]() {
        if (RC_EXP & 4) return;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (SPLIT == 1) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s2][i], fb[s2][j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa2[i], fb[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb2[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
                }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    unsigned long long tmark = RC_STAMP == 2 ? __builtin_readcyclecounter() : 0ull;
    auto lap = [&](int which) {                            // RC_STAMP 2: cycles since the last mark go to phase counter `which`
        if (RC_STAMP == 2) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t = __builtin_readcyclecounter();
            c.tph[which] += t - tmark;
            tmark = t;
        }
    };
    auto bar = [&]() {                                      // ends an MFMA phase
        __builtin_amdgcn_sched_barrier(0);
        lap(2);
        __builtin_amdgcn_s_barrier();
        lap(3);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto bar_reads = [&]() {                                // ends a read phase: the fragment reads issued so far are retired first
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        lap(0);
        __builtin_amdgcn_s_barrier();
        lap(1);
        __builtin_amdgcn_sched_barrier(0);
    };
    // (the two rows are two separate loop nests: one nest with a row branch inside every step made the register allocator spill
    // ~1 000 VGPRs)
    auto steps = [&](auto rowc) {
    constexpr int ROW = decltype(rowc)::value;
    for (int ch = 0; ch < p.nchunk; ++ch) {
        const int wbuf = (c.wpar + ch) & 1;
        const bool lastc = ch + 1 == p.nchunk;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) {
            const int kn = (k + 2) % TAPS;
            int cn = ch + (k + 2) / TAPS;
            cn = cn == p.nchunk ? 0 : cn;
            const bool into_next = lastc && (k + 2) / TAPS == 1;          // the last two steps request the NEXT tile's steps 0 and 1
            const char* wsrc = (into_next ? c.w_next + (long)kn * c.wts_next : c.w_base + (long)kn * c.wts) + (long)cn * 128;
            const unsigned wdst = c.lds0 + RC_RING + (ws == 0 ? 2 : ws - 1) * RC_W_BYTES + wave * 1024;
            int nwin = 0;
            auto window = [&]() {                          // next window into the idle buffer: next chunk, or the next tile's chunk 0
                if (k == 0 && !(RC_EXP & 1)) {
                    if (!lastc) { issue_a(ch + 1, wbuf ^ 1); nwin = nq; }
                    else if (h1 > 0) { issue_a_next(wbuf ^ 1); nwin = nq1; }
                }
            };
            auto dma_w = [&](int g) { if (!(RC_EXP & 1)) dma16(wdst + g * 8192, c.vow[g], wsrc); };
            int kv = k;
            asm volatile("" : "+s"(kv));
            const char* at = smem + wbuf * RC_WIN_BYTES;
            const char* wt = smem + RC_RING + ws * RC_W_BYTES;
            const int arow = row0w + lrow + kv + (5 - TAPS) / 2;
            if constexpr (ROW == 0) {
                window();
                loadH(at, wt, arow, 0); dma_w(0); dma_w(1);
                bar_reads();
                mmaH();
                bar();
                loadH(at, wt, arow, 1); dma_w(2); dma_w(3);
                bar_reads();
                mmaH();
                if (RC_EXP & 1) rc_wait(0); else rc_wait(4 + nwin);
                bar();
            } else {
                bar();
                window();
                loadH(at, wt, arow, 0); dma_w(0); dma_w(1);
                bar_reads();
                mmaH();
                bar();
                loadH(at, wt, arow, 1); dma_w(2); dma_w(3);
                if (RC_EXP & 1) rc_wait(0); else rc_wait(4 + nwin);
                bar_reads();
                mmaH();
            }
            ws = (ws == 2) ? 0 : ws + 1;
        }
    }
    };
    if (wm == 0) steps(RcRow<0>{}); else steps(RcRow<1>{});
    RC_MARK(p, c);
    c.ws = ws;
    c.wpar = (c.wpar + p.nchunk) & 1;

    if (RC_EXP & 16) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    // ---- epilogue, wave-private, 64 columns at a time: the wave's two 32 x 32 accumulator blocks of a block row are staged in
    // 2 x 4 KiB of LDS that is free between two tiles -- its share of the ring slot the last step read (block j = 0) and of the
    // window buffer the last chunk used (block j = 1); the other window buffer and the other two ring slots already hold the next
    // tile's first operands -- and swept out 8 rows per pass with 8 lanes per row: every buffer instruction moves whole 128-byte
    // lines of the separate hi / lo planes (64-byte segments of the [32 hi | 32 lo] chunks of a split-2 plane).
    char* const st0 = smem + RC_RING + (ws == 0 ? 2 : ws - 1) * RC_W_BYTES + wave * 4096;
    char* const st1 = smem + ((c.wpar ^ 1) & 1) * RC_WIN_BYTES + wave * 4096;       // c.wpar now names the NEXT tile's chunk-0 buffer
    const __amdgpu_buffer_rsrc_t r_a = make_rsrc(pq.a + (long)m0 * pq.lda, (long)rows_out * pq.lda);
    const __amdgpu_buffer_rsrc_t r_al = make_rsrc(pq.a_lo ? pq.a_lo + (long)m0 * pq.lda : nullptr, pq.a_lo ? (long)rows_out * pq.lda : 0);
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(pq.resid ? pq.resid + (long)m0 * pq.ldr : nullptr, pq.resid ? (long)rows_out * pq.ldr * 4 : 0);
    const __amdgpu_buffer_rsrc_t r_m = make_rsrc(pq.rowmask ? pq.rowmask + m0 : nullptr, pq.rowmask ? (long)rows_out * 4 : 0);
    const __amdgpu_buffer_rsrc_t r_of = make_rsrc(pq.out_f32 ? pq.out_f32 + (long)m0 * pq.ldo : nullptr, pq.out_f32 ? (long)rows_out * pq.ldo * 4 : 0);
    const __amdgpu_buffer_rsrc_t r_ob = make_rsrc(pq.ob ? pq.ob + (long)m0 * pq.ldob : nullptr, pq.ob ? (long)rows_out * pq.ldob : 0);
    const __amdgpu_buffer_rsrc_t r_ol = make_rsrc(pq.ob_lo ? pq.ob_lo + (long)m0 * pq.ldob : nullptr, pq.ob_lo ? (long)rows_out * pq.ldob : 0);
    const __amdgpu_buffer_rsrc_t r_sg = make_rsrc(pq.sign ? pq.sign + (long)m0 * pq.ldsg : nullptr, pq.sign ? (long)rows_out * pq.ldsg : 0);
    constexpr bool FAST = EPI == RC_EPI_FAST, DACT = EPI == RC_EPI_DGRAD_ACT;
    const bool res_f32 = FAST ? false : (DACT ? true : pq.resid != nullptr);
    const bool has_mask = (FAST || DACT) ? !(RC_EXP & 8) : (pq.rowmask != nullptr && !(RC_EXP & 8));
    const bool f_noresid = (FAST || DACT) ? false : pq.no_resid != 0;
    const bool f_sign = (FAST || DACT) ? false : pq.sign != nullptr;
    const bool f_of32 = FAST ? false : (DACT ? true : pq.out_f32 != nullptr);
    const bool f_ob = (FAST || DACT) ? true : pq.ob != nullptr;
    const bool f_os2 = (FAST || DACT) ? SPLIT == 2 : pq.out_split == 2;
    const bool f_oblo = FAST ? SPLIT == 1 : (DACT ? false : pq.ob_lo != nullptr);
    const float slope = pq.slope;
    const float slope_bwd = pq.slope_bwd;
    const __amdgpu_buffer_rsrc_t r_si = make_rsrc(DACT ? pq.sign_in + (long)m0 * pq.ldsg : nullptr, DACT ? (long)rows_out * pq.ldsg : 0);
    float csum[8];                                    // DACT: column sums of dZ' over this lane's rows of the tile
#pragma unroll
    for (int e = 0; e < 8; ++e) csum[e] = 0.f;
    const unsigned ldsg = (unsigned)pq.ldsg;
    const int srow = lane >> 3;                       // row of the 8-row pass this lane handles
    const int c8 = lane & 7;                          // its 8 columns inside the 64-column pair of blocks
    const char* const stl = (c8 < 4 ? st0 : st1);     // the block those columns were staged in

    // Addressing: one per-lane byte offset per stream (this lane's row of the pass, its 8 columns); the block row i and the pass
    // go into the scalar offset of the buffer instruction.
    const unsigned lrow0 = row0w + srow;                           // tile row of this lane in pass (i = 0, pass = 0)
    const unsigned col0 = c.n0 + wn * 64 + c8 * 8;                 // its first column
    const unsigned vx = res_f32 ? lrow0 * (unsigned)pq.ldr * 4 + col0 * 4
                                : lrow0 * (unsigned)pq.lda + (SPLIT == 1 ? col0 * 2 : (col0 >> 5) * 128 + (col0 & 31) * 2);
    const unsigned sx_row = res_f32 ? (unsigned)pq.ldr * 4 : (unsigned)pq.lda;       // bytes per row
    const unsigned vm = lrow0 * 4;
    const unsigned vof = lrow0 * (unsigned)pq.ldo * 4 + col0 * 4, sof_row = (unsigned)pq.ldo * 4;
    const unsigned vob = lrow0 * (unsigned)pq.ldob + (!f_os2 ? col0 * 2 : (col0 >> 5) * 128 + (col0 & 31) * 2);
    const unsigned sob_row = (unsigned)pq.ldob;
    const unsigned vsg = lrow0 * ldsg + (col0 >> 3);

    // (split-2 planes keep 64-byte hi / lo halves per instruction.  Measured and dropped: lanes 0-3 of a row on the hi slots and
    // lanes 4-7 on the lo slots of the same 32 columns, words swapped through ds_bpermute, both lanes computing the same outputs --
    // whole lines per instruction, but 292.6 -> 295.9 us per B = 64 launch: the doubled VALU / LDS work of the sweep costs more
    // than the half-line requests.)
    // operands of pass P = 4 i + ps (block row i, 8-row pass ps): 8 residual values (fp32, or bf16 hi + lo) and the row mask.  Plain layers
    // request a unit (two passes) at a time into slots P & 3, one unit ahead of the one being written out; the dgrad + activation-backward
    // variant requests pass by pass into a ring of three (two passes ahead): 16 registers less at the peak, which it needs for its column sums
    u32x4 xa[4], xb[4];
    float rmv[4];
    unsigned sgv[4];                                  // DACT: the sign byte of this lane's 8 columns
    auto request_pass = [&](int P, int sl) {
        const unsigned rofs = (P >> 2) * 32 + (P & 3) * 8;
        const unsigned so = rofs * sx_row;
        if ((RC_EXP & 8) || f_noresid) { xa[sl] = u32x4{0, 0, 0, 0}; xb[sl] = xa[sl]; }
        else if (res_f32) {
            xa[sl] = __builtin_amdgcn_raw_buffer_load_b128(r_x, vx, so, 0);
            xb[sl] = __builtin_amdgcn_raw_buffer_load_b128(r_x, vx, so + 16, 0);
        } else if (SPLIT == 1) {
            xa[sl] = __builtin_amdgcn_raw_buffer_load_b128(r_a, vx, so, 0);
            xb[sl] = __builtin_amdgcn_raw_buffer_load_b128(r_al, vx, so, 0);      // null plane: zeros
        } else {
            xa[sl] = __builtin_amdgcn_raw_buffer_load_b128(r_a, vx, so, 0);
            xb[sl] = __builtin_amdgcn_raw_buffer_load_b128(r_a, vx, so + 64, 0);
        }
        rmv[sl] = has_mask ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_m, vm, rofs * 4, 0)) : 1.f;
        if (DACT) sgv[sl] = __builtin_amdgcn_raw_buffer_load_b8(r_si, vsg, rofs * ldsg, 0);
    };
    request_pass(0, 0);
    request_pass(1, 1);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        // accumulator blocks -> LDS.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
        // 16-byte slots of a row are XORed with (row >> 1) & 1: the row-major read-back is bank-conflict free
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            char* const stj = j ? st1 : st0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                float v = acc[i][j][r];
                if (!DACT) {                          // (the dgrad layer has no bias and slope 1)
                    v += c.bv[j];
                    v = v > 0.f ? v : v * slope;
                }
                *(float*)(stj + rl * 128 + ((((lrow >> 2) ^ ((rl >> 1) & 1))) << 4) + (lrow & 3) * 4) = v;
            }
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int P = 4 * i + ps, sl = DACT ? P % 3 : (P & 3);
            if (DACT) {
                if (P + 2 < 4 * NI) request_pass(P + 2, (P + 2) % 3);
            } else if ((P & 1) == 0 && P + 2 < 4 * NI) {
                request_pass(P + 2, (P + 2) & 3);
                request_pass(P + 3, (P + 3) & 3);
            }
            const int row = ps * 8 + srow;
            const int sw = (row >> 1) & 1;
            const float4 d0 = *(const float4*)(stl + row * 128 + ((((c8 & 3) * 2) ^ sw) << 4));
            const float4 d1 = *(const float4*)(stl + row * 128 + ((((c8 & 3) * 2 + 1) ^ sw) << 4));
            float x[8];
            const u32x4 qa = xa[sl], qb = xb[sl];
            if (res_f32) {
                x[0] = __uint_as_float(qa.x); x[1] = __uint_as_float(qa.y); x[2] = __uint_as_float(qa.z); x[3] = __uint_as_float(qa.w);
                x[4] = __uint_as_float(qb.x); x[5] = __uint_as_float(qb.y); x[6] = __uint_as_float(qb.z); x[7] = __uint_as_float(qb.w);
            } else {
                const unsigned ha[4] = {qa.x, qa.y, qa.z, qa.w}, lo[4] = {qb.x, qb.y, qb.z, qb.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    x[2 * u] = __uint_as_float(ha[u] << 16) + __uint_as_float(lo[u] << 16);
                    x[2 * u + 1] = __uint_as_float(ha[u] & 0xffff0000u) + __uint_as_float(lo[u] & 0xffff0000u);
                }
            }
            const float rm = rmv[sl];
            float y[8] = {(x[0] + d0.x) * rm, (x[1] + d0.y) * rm, (x[2] + d0.z) * rm, (x[3] + d0.w) * rm,
                          (x[4] + d1.x) * rm, (x[5] + d1.y) * rm, (x[6] + d1.z) * rm, (x[7] + d1.w) * rm};
            const unsigned brow = i * 32 + ps * 8;
            // (rows of the next tile / past the matrix: every output descriptor ends at this tile's last row, so their stores are dropped)
            if (RC_EXP & 8) { asm volatile("" ::"v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7])); continue; }
            if (f_sign) {                                       // training: one byte of sign bits per lane (its 8 columns)
                const unsigned sb = (d0.x > 0.f ? 1u : 0u) | (d0.y > 0.f ? 2u : 0u) | (d0.z > 0.f ? 4u : 0u) | (d0.w > 0.f ? 8u : 0u) |
                                    (d1.x > 0.f ? 16u : 0u) | (d1.y > 0.f ? 32u : 0u) | (d1.z > 0.f ? 64u : 0u) | (d1.w > 0.f ? 128u : 0u);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)sb, r_sg, vsg, brow * ldsg, 0);
            }
            if (f_of32) {
                const u32x4 o0 = {__float_as_uint(y[0]), __float_as_uint(y[1]), __float_as_uint(y[2]), __float_as_uint(y[3])};
                const u32x4 o1 = {__float_as_uint(y[4]), __float_as_uint(y[5]), __float_as_uint(y[6]), __float_as_uint(y[7])};
                store_b128(o0, r_of, vof, brow * sof_row);               // constant displacements go into the scalar
                store_b128(o1, r_of, vof, brow * sof_row + 16);          // offset: no VALU address math between stores
            }
            if (DACT) {                                         // activation backward of the layer below on the values just produced
                const unsigned sb = sgv[sl];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    y[e] = (sb >> e) & 1u ? y[e] : y[e] * slope_bwd;
                    // (the tile's last 4 MFMA rows and the rows past rows_out are discarded rows: their accumulators saw window rows nobody
                    //  loaded and may hold anything, NaN included -- 0 * NaN must not reach the sums; their stores are dropped as always)
                    csum[e] += rm != 0.f ? y[e] : 0.f;
                }
            }
            if (f_ob) {
                float rr[8];
                const u32x4 hi = {pack_bf16x2(y[0], y[1], &rr[0], &rr[1]), pack_bf16x2(y[2], y[3], &rr[2], &rr[3]),
                                  pack_bf16x2(y[4], y[5], &rr[4], &rr[5]), pack_bf16x2(y[6], y[7], &rr[6], &rr[7])};
                float d0_, d1_;
                const u32x4 lo = {pack_bf16x2(rr[0], rr[1], &d0_, &d1_), pack_bf16x2(rr[2], rr[3], &d0_, &d1_),
                                  pack_bf16x2(rr[4], rr[5], &d0_, &d1_), pack_bf16x2(rr[6], rr[7], &d0_, &d1_)};
                const unsigned so = brow * sob_row;
                store_b128(hi, r_ob, vob, so);
                if (f_os2) store_b128(lo, r_ob, vob, so + 64);
                else if (f_oblo) store_b128(lo, r_ol, vob, so);
            }
        }
    }
    if (DACT) {
        // column sums of this wave's rows: the 8 lanes of a column octet (same c8, rows srow = 0..7) are added up, lane srow = 0 stores
        // its 8 columns.  Rows past rows_out belong to the next tile (or lie past the matrix): the mask descriptor reads 0 for them and they
        // were left out above
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = csum[e];
            t += __shfl_xor(t, 8); t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
            csum[e] = t;
        }
        // (no divergent branch: lanes of rows srow > 0 point past the end of the descriptor, which drops their stores)
        const __amdgpu_buffer_rsrc_t r_bp = make_rsrc(pq.bias_part ? pq.bias_part + (long)c.bp_row * (p.ntn * RC_BN) : nullptr, pq.bias_part ? (long)p.ntn * RC_BN * 4 : 0);
        const unsigned vbp = srow == 0 ? col0 * 4 : 0x7fffff00u;
        store_b128(u32x4{__float_as_uint(csum[0]), __float_as_uint(csum[1]), __float_as_uint(csum[2]), __float_as_uint(csum[3])}, r_bp, vbp, 0);
        store_b128(u32x4{__float_as_uint(csum[4]), __float_as_uint(csum[5]), __float_as_uint(csum[6]), __float_as_uint(csum[7])}, r_bp, vbp, 16);
    }
}



// MODE 0: every layer form of the forward (and the plain dgrad layer).  MODE 1: layers with the fused activation backward only
// (RC_EPI_DGRAD_ACT, 5 taps, one problem per launch) -- its own kernel, compiled in its own translation unit.
template <int SPLIT, int MODE>
__global__ __launch_bounds__(512, 2) void resconv5_kernel(RcArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RcCtx c;
    c.smem = smem;
    c.lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem));
    const int tid = threadIdx.x;
    if (RC_STAMP == 1 && p.stamp && tid == 0) p.stamp[blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime();
    c.tph[0] = c.tph[1] = c.tph[2] = c.tph[3] = 0;
    c.nst = 0;
    c.lane = tid & 63;
    c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    RC_MARK(p, c);
    c.wm = c.wave >> 2; c.wn = c.wave & 3;
    c.lrow = c.lane & 31; c.lhalf = c.lane >> 5;

    // XCD-aware order: block b runs on XCD b % 8; each XCD gets a contiguous range of (group, column tile) pairs, so the
    // column tiles of a group (which read the same windows) share an L2
    int v = blockIdx.x;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7;
        const int xcd = v & 7, loc = v >> 3;
        v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int g = v / p.ntn, nt = v - g * p.ntn;
    c.n0 = nt * RC_BN;
    const int cls = g % p.s.ncls;
    int sum_rows = 0, pre = 0;
    for (int i = 0; i < p.s.ncls; ++i) { if (i < cls) pre += p.s.rows[i]; sum_rows += p.s.rows[i]; }
    int vrow = (g / p.s.ncls) * sum_rows + pre;                       // this group's first VIRTUAL row (layer 0's rows, then layer 1's)
    const int vend = vrow + p.s.rows[cls] < p.m ? vrow + p.s.rows[cls] : p.m;
    const int ntile = p.s.ntile[cls];
    const int mfirst = p.nprob > 1 ? p.pr[0].m : p.m;                 // virtual row where layer 1 starts
    // tile t starting at virtual row vr: its layer, its first row inside that layer, its height (half units: the scheduled one, cut
    // to what is left of this group's rows IN THIS LAYER -- a tile never crosses into the next layer; a group cut by the layer
    // boundary runs one tile more than scheduled) and the rows it yields
    auto locate = [&](int t, int vr, int& pi, int& ml, int& hh, int& rows) {
        pi = 0; ml = 0; hh = 0; rows = 0;
        if (vr >= vend) return;
        pi = vr >= mfirst ? 1 : 0;
        const int lend = (pi == 0 && mfirst < vend) ? mfirst : vend;
        int need = (lend - vr + 4 + 31) >> 5;
        need = need < 2 ? 2 : need;
        const int hs = t < ntile ? p.s.ni[cls][t] : 8;
        hh = hs < need ? hs : need;
        rows = lend - vr < 32 * hh - 4 ? lend - vr : 32 * hh - 4;
        ml = vr - (pi ? mfirst : 0);
    };
    int pi, m0, h, rows_out;
    locate(0, vrow, pi, m0, h, rows_out);
    auto zero_bias_rows = [&](int t_from) {
        // rows of `bias_part` this workgroup's group owns but runs no tile for (a class with fewer tiles than bp_tiles, a group beyond the end of
        // the row space): zeroed here, so that the table needs no caller-side clearing whatever m / plan it was last used with
        if constexpr (MODE == 1) {
            if (p.pr[0].bias_part)
                for (int tt = t_from; tt < p.bp_tiles; ++tt)
                    p.pr[0].bias_part[(long)(((g * p.bp_tiles) + tt) * 2 + (tid >> 8)) * (p.ntn * RC_BN) + c.n0 + (tid & 255)] = 0.f;
        }
    };
    if (h == 0) { zero_bias_rows(0); return; }

#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (q * 8 + c.wave) * 8 + (c.lane >> 3);
        const int sl = (c.lane & 7) ^ ((r >> 1) & 7);
        c.vow[q] = (unsigned)(r * (int)p.pr[0].ldw + (sl << 4));     // (every layer of a launch has the same weight row stride)
    }
    auto bias_of = [&](const RcProb& q) {
#pragma unroll
        for (int j = 0; j < 2; ++j) c.bv[j] = q.bias ? q.bias[c.n0 + c.wn * 64 + j * 32 + c.lrow] : 0.f;
    };
    {
        const RcProb& q0 = p.pr[pi];
        c.w_base = q0.w + (long)c.n0 * q0.ldw;
        c.wts = q0.w_tap_stride;
        bias_of(q0);
        c.ws = 0; c.wpar = 0;
        // first tile: window of chunk 0, weights of steps 0 and 1
        const int rmax = q0.m + 143 - (m0 - 2);
        const char* sb = q0.a + (long)(m0 - 2) * q0.lda;
        const int nq = rc_pieces(h, c.wave);
        for (int q = 0; q < nq; ++q) {
            const int r = (q * 8 + c.wave) * 8 + (c.lane >> 3);
            const int sl = (c.lane & 7) ^ ((r >> 1) & 7);
            dma16(c.lds0 + c.wave * 1024 + q * 8192, (unsigned)((r < rmax ? r : rmax) * (int)q0.lda + (sl << 4)), sb);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                dma16(c.lds0 + RC_RING + s * RC_W_BYTES + c.wave * 1024 + q * 8192, c.vow[q], c.w_base + (long)s * c.wts);
    }
    int t = 0;
    for (; h > 0; ++t) {
        const int vnext = vrow + rows_out;
        int pi1, m1, h1, rows1;
        locate(t + 1, vnext, pi1, m1, h1, rows1);
        const RcProb& pq = p.pr[pi];
        const RcProb& pn = p.pr[h1 > 0 ? pi1 : pi];
        c.w_next = pn.w + (long)c.n0 * pn.ldw;
        c.wts_next = pn.w_tap_stride;
        const int nblk = c.wm ? h >> 1 : (h + 1) >> 1;      // 32-row blocks of this wave's row (wave-uniform)
        const bool fast = !pq.resid && pq.rowmask && pq.ob && !pq.out_f32 && !pq.sign && !pq.no_resid &&
                          (SPLIT == 1 ? (pq.a_lo && pq.ob_lo && pq.out_split == 1) : pq.out_split == 2);
        c.bp_row = ((g * p.bp_tiles) + t) * 2 + c.wm;
        if constexpr (MODE == 1) {
            (void)fast;
            switch (nblk) {
                case 1: rc_tile<SPLIT, 1, 5, RC_EPI_DGRAD_ACT>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
                case 2: rc_tile<SPLIT, 2, 5, RC_EPI_DGRAD_ACT>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
                case 3: rc_tile<SPLIT, 3, 5, RC_EPI_DGRAD_ACT>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
                default: rc_tile<SPLIT, 4, 5, RC_EPI_DGRAD_ACT>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
            }
        } else {
#if RC_LAB_MIN
        // lab builds (tools/lab_build.sh): only the variant the probes launch -- 5 taps, planes in and out -- is instantiated.  With all 24
        // tile variants AND the stamping code inlined into one kernel the register allocator runs out, the wave-uniform operands of the
        // LDS-DMA statements come back from spill slots in VGPRs and the build fails ("s" constraint) or, forced through readfirstlane,
        // spills 60-100 VGPRs into the main loop; the product build (no stamps) has 223 VGPRs and no scratch.
        (void)fast;
        switch (nblk) {
            case 1: rc_tile<SPLIT, 1, 5, RC_EPI_FAST>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
            case 2: rc_tile<SPLIT, 2, 5, RC_EPI_FAST>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
            case 3: rc_tile<SPLIT, 3, 5, RC_EPI_FAST>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
            default: rc_tile<SPLIT, 4, 5, RC_EPI_FAST>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
        }
#else
        if (pq.taps == 3) {
            switch (nblk) {
                case 1: rc_tile<SPLIT, 1, 3, RC_EPI_GENERAL>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
                case 2: rc_tile<SPLIT, 2, 3, RC_EPI_GENERAL>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
                case 3: rc_tile<SPLIT, 3, 3, RC_EPI_GENERAL>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
                default: rc_tile<SPLIT, 4, 3, RC_EPI_GENERAL>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
            }
        } else if (fast) {
            switch (nblk) {
                case 1: rc_tile<SPLIT, 1, 5, RC_EPI_FAST>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
                case 2: rc_tile<SPLIT, 2, 5, RC_EPI_FAST>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
                case 3: rc_tile<SPLIT, 3, 5, RC_EPI_FAST>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
                default: rc_tile<SPLIT, 4, 5, RC_EPI_FAST>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
            }
        } else {
            switch (nblk) {
                case 1: rc_tile<SPLIT, 1, 5, RC_EPI_GENERAL>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
                case 2: rc_tile<SPLIT, 2, 5, RC_EPI_GENERAL>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
                case 3: rc_tile<SPLIT, 3, 5, RC_EPI_GENERAL>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
                default: rc_tile<SPLIT, 4, 5, RC_EPI_GENERAL>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
            }
        }
#endif
        }
        RC_MARK(p, c);
        if (h1 > 0 && pi1 != pi) bias_of(pn);
        c.w_base = c.w_next; c.wts = c.wts_next;
        vrow = vnext; pi = pi1; m0 = m1; h = h1; rows_out = rows1;
    }
    zero_bias_rows(t);
    // the last tile's wrapped weight requests may still be landing: LDS must not be handed to another workgroup under them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (RC_STAMP == 1 && p.stamp && tid == 0) p.stamp[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
    if (RC_STAMP == 3 && p.stamp && tid == 0) p.stamp[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    if (RC_STAMP == 2 && p.stamp && blockIdx.x < 32 && c.lane == 0) {          // [workgroup < 32][wave][4 phase counters]
#pragma unroll
        for (int q = 0; q < 4; ++q) p.stamp[(blockIdx.x * 8 + c.wave) * 4 + q] = c.tph[q];
    }
}

}  // namespace efts

