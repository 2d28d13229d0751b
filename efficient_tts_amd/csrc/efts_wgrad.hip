// efts_wgrad.hip -- weight gradient of the k5 residual convolutions (and, since round 2, of the k3 convolutions and the
// Linears with 128 | cout, 64 | cin: TAPS = 3, 1) straight from the ROW-MAJOR operand planes the forward / dgrad
// contractions already use (no transposed copies):
//
//   part[segment][k][co][ci] = sum_{t in segment} dZ[t][co] * X[t + k - pad][ci]       (k = 0..taps-1, pad = (taps - 1) / 2)
//
// Replaces, for the (512, 512, 5) layers, efts_pack_t (one transposed plane of dZ + FIVE shifted
// transposed planes of X per layer, 183 MB of HBM traffic) + the split-K efts_gemm.
// (reference: autograd of F.conv1d in nntts/layers/efts_modules.py:48-51)
//
// The contraction runs over t, i.e. over the ROWS of both planes.  MFMA wants each lane to hold 8
// consecutive k (= t) values of one output row/column, which in a row-major LDS image are strided:
// gfx950's ds_read_b64_tr_b16 (transposing LDS read) delivers exactly that, 4 t-values per read.
//   * workgroup tile: 128 co x 64 ci x all 5 taps, one K-split of the rows; 2x2 waves of 64 co x 32 ci,
//     accumulators 5 taps x 2 blocks x 16 = 160 VGPRs.
//   * per step 64 rows (bf16x3 planes: 32 rows, twice the chunks): dZ tile [2 chunks][64 t][128 B] + X window
//     [72 t][128 B] (rows t0-2 .. t0+69) = 25 KiB by LDS-DMA (asm, counted vmcnt), 3-stage ring = 75 KiB, two workgroups per CU.
//     The tap shift is a ROW offset into the X window (as in the forward kernel), so one window
//     serves all 5 taps: 40 MFMAs per wave and step against 25 KiB staged (the transposed-plane GEMM:
//     16 MFMAs per 19 KiB).
//   * LDS image: row-major, 16-byte slots XORed with ((row >> 1) & 1) << 2, which makes the 4-row x
//     64-byte footprint of a 32-lane transposing read cover all 64 banks exactly once.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "efts_internal.h"

namespace efts {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int WG_NST = 3;

// Tile constants.  SPLIT 1 (bf16 planes, 64 channels per 128-byte chunk): 64 rows per step, dZ = 2 chunks,
// X = 1 chunk.  SPLIT 2 (bf16x3 planes, 32 hi + 32 lo channels per chunk): 32 rows per step, dZ = 4 chunks,
// X = 2 chunks -- the same 128 co x 64 ci tile and about the same LDS per stage.
template <int SPLIT>
struct WgCfg {
    static constexpr int ROWS = SPLIT == 1 ? 64 : 32;            // t rows per step
    static constexpr int CA = SPLIT == 1 ? 2 : 4;                // dZ chunks per tile
    static constexpr int CB = SPLIT == 1 ? 1 : 2;                // X chunks per tile
    static constexpr int BROWS = ROWS + 8;                       // window rows (4 halo rows, rounded up to a DMA piece)
    static constexpr int A_BYTES = CA * ROWS * 128;              // 16384
    static constexpr int B_BYTES = CB * BROWS * 128;             // 9216 / 10240
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int LDS = WG_NST * STAGE;                   // 76800 / 79872
    static constexpr int APP = ROWS / 8;                         // DMA pieces per dZ chunk
    static constexpr int BPP = BROWS / 8;                        // DMA pieces per X chunk
    static constexpr int NBP = CB * BPP;                         // 9 / 10 window pieces per step
    static constexpr int KK = ROWS / 16;                         // MFMA k-steps per step
};

__device__ __forceinline__ int tn_off(int row, int byte_in_row) {      // swizzled byte offset inside a [rows][128 B] image
    return row * 128 + ((((byte_in_row >> 4) ^ (((row >> 1) & 1) << 2)) << 4) | (byte_in_row & 15));
}

__device__ __forceinline__ void tn_dma16(unsigned lds_addr, unsigned voff, const char* sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}

__device__ __forceinline__ bf16x8 tr_frag(const char* p0, const char* p1) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}

// One segment of one output tile as seen by one workgroup: rows [t_begin, t_begin + nsteps * ROWS) of the item's planes, tile (mt, nt);
// leaves the partial sums in the accumulators `acc` (zeroed here).  All LDS traffic of the segment is retired on return.
template <int SPLIT, int TAPS>
__device__ __forceinline__ void wg_run(char* smem, unsigned lds0, const char* __restrict__ dz, long ldz, const char* __restrict__ x, long ldx,
                                       int mt, int nt, int t_begin, int nsteps, f32x16 (&acc)[TAPS][2]) {
    using C = WgCfg<SPLIT>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- DMA plan of this wave.  dZ: 16 pieces of 8 rows (chunk = piece / APP), wave w owns 4w..4w+3.
    // X window: NBP pieces (chunk = piece / BPP), wave w owns the pieces p with p % 4 == w (2 or 3 of them).
    unsigned voa[4], vob[3], ldb[3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pc = wave * 4 + q;
        const int row = (pc % C::APP) * 8 + (lane >> 3);
        const int sl = (lane & 7) ^ (((row >> 1) & 1) << 2);
        voa[q] = (unsigned)(row * (int)ldz + (pc / C::APP) * 128 + (sl << 4));
    }
    const int nb = (C::NBP - 1 - wave) / 4 + 1;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int pc = wave + 4 * q;
        const int pcc = pc < C::NBP ? pc : 0;
        const int row = (pcc % C::BPP) * 8 + (lane >> 3);
        const int sl = (lane & 7) ^ (((row >> 1) & 1) << 2);
        vob[q] = (unsigned)(row * (int)ldx + (pcc / C::BPP) * 128 + (sl << 4));
        ldb[q] = (unsigned)(C::A_BYTES + pcc * 1024);
    }
    const char* a_src = dz + (long)t_begin * ldz + (long)(mt * C::CA) * 128;
    const char* b_src = x + (long)(t_begin - (TAPS - 1) / 2) * ldx + (long)(nt * C::CB) * 128;
    auto issue = [&](int st) {                      // operands of step st -> ring slot st % 3
        const unsigned base = lds0 + (st % WG_NST) * C::STAGE;
        const char* sa = a_src + (long)st * C::ROWS * ldz;
        const char* sb = b_src + (long)st * C::ROWS * ldx;
#pragma unroll
        for (int q = 0; q < 4; ++q) tn_dma16(base + (wave * 4 + q) * 1024, voa[q], sa);
#pragma unroll
        for (int q = 0; q < 2; ++q) tn_dma16(base + __builtin_amdgcn_readfirstlane(ldb[q]), vob[q], sb);
        if (nb == 3) tn_dma16(base + __builtin_amdgcn_readfirstlane(ldb[2]), vob[2], sb);
    };
    auto wait_next = [&](bool issued) {             // the operands of the next step have landed; this step's issue may stay in flight
        if (!issued) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (nb == 3) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    };

#pragma unroll
    for (int k = 0; k < TAPS; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][i][r] = 0.f;

    // per-lane geometry of the transposing reads: 16-lane group g = lane >> 4 reads a [4 t][16 ch] block;
    // lane q of the group supplies the address of row (q >> 2), channels 4 * (q & 3) .. +3 of that block
    const int q16 = lane & 15;
    const int t_lane = (lane >> 5) * 8 + (q16 >> 2);             // + kk * 16 + h * 4
    const int ch_lane = ((lane >> 4) & 1) * 16 + (q16 & 3) * 4;  // channel inside the 32-channel block of this lane group pair

    // (a stream-K workgroup enters with the previous segment's slab stores in flight: stores and loads retire out of order with
    //  respect to each other, so the counted waits below must start from an empty queue)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (nsteps > 0) issue(0);
    if (nsteps > 1) issue(1);
    wait_next(nsteps > 1);
    __builtin_amdgcn_s_barrier();

    for (int st = 0; st < nsteps; ++st) {
        const bool more = st + 2 < nsteps;
        if (more) issue(st + 2);
        const char* stage = smem + (st % WG_NST) * C::STAGE;
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) {
            const int ra = kk * 16 + t_lane;
            if constexpr (SPLIT == 1) {
                const char* at = stage + wm * (C::ROWS * 128);                 // dZ chunk wm = this wave's 64 co
                const char* bt = stage + C::A_BYTES;
                bf16x8 af[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = tr_frag(at + tn_off(ra, (ch_lane + i * 32) * 2), at + tn_off(ra + 4, (ch_lane + i * 32) * 2));
#pragma unroll
                for (int k = 0; k < TAPS; ++k) {
                    const int rb = ra + k;                                     // window row of t at tap k: (t - t0) + pad + (k - pad)
                    const bf16x8 bfr = tr_frag(bt + tn_off(rb, (wn * 32 + ch_lane) * 2), bt + tn_off(rb + 4, (wn * 32 + ch_lane) * 2));
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[k][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr, acc[k][i], 0, 0, 0);
                }
            } else {
                const char* bt = stage + C::A_BYTES + wn * (C::BROWS * 128);   // X chunk wn = this wave's 32 ci
                bf16x8 ah[2], al[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const char* at = stage + (wm * 2 + i) * (C::ROWS * 128);   // dZ chunk of this 32-co block: 32 hi | 32 lo
                    ah[i] = tr_frag(at + tn_off(ra, ch_lane * 2), at + tn_off(ra + 4, ch_lane * 2));
                    al[i] = tr_frag(at + tn_off(ra, 64 + ch_lane * 2), at + tn_off(ra + 4, 64 + ch_lane * 2));
                }
#pragma unroll
                for (int k = 0; k < TAPS; ++k) {
                    const int rb = ra + k;
                    const bf16x8 bh = tr_frag(bt + tn_off(rb, ch_lane * 2), bt + tn_off(rb + 4, ch_lane * 2));
                    const bf16x8 bl = tr_frag(bt + tn_off(rb, 64 + ch_lane * 2), bt + tn_off(rb + 4, 64 + ch_lane * 2));
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        acc[k][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[k][i], 0, 0, 0);
                        acc[k][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[k][i], 0, 0, 0);
                        acc[k][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[k][i], 0, 0, 0);
                    }
                }
            }
        }
        if (st + 1 < nsteps) wait_next(more);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

}

// ---------------------------------------------------------------------------------------------------------------
// Grouped form (round 5): the weight gradients of ALL layers of a residual stack in one launch, the work dealt out
// stream-K style.  The (item, tile, step) space of an XCD -- the tiles T = 8 j + xcd of the launch, 401 steps each at
// B = 32 x 800 frames -- is cut into equal contiguous ranges, one per workgroup; a workgroup runs the segments of the
// (at most `maxseg`) tiles its range touches and leaves one [taps][128][64] fp32 slab per segment.  Against one launch
// per layer with 8 K-splits: 2.3 slabs per tile instead of 8 (12 instead of 42 MB of partials per layer written and
// read back), no partial last round, one launch boundary per stack instead of one per layer; the reduction
// (wgrad_reduce_sk_kernel, efts_train.hip) derives the slabs of a tile from the same geometry, in a fixed order.
// ---------------------------------------------------------------------------------------------------------------
struct WgItem { const char* dz; const char* x; long ldz, ldx; };
struct WgSkArgs {
    WgItem it[EFTS_WGRAD_MAX_ITEMS];
    float* part;
    int tiles_item, ntn;          // tiles per item, ci tiles per co tile
    int nx;                       // 8: tile T belongs to XCD T % 8 (workgroup w runs on XCD w % 8); 1: one list
    int steps_tile, q, total;     // steps per tile; steps per workgroup; steps of one list
    int maxseg;
    int stamp[8];                 // written behind the slabs by workgroup 0: what the reduction checks (efts_internal.h)
    int* stamp_dst;
};

template <int SPLIT, int TAPS>
__global__ __launch_bounds__(256, 2) void wgrad_sk_kernel(WgSkArgs p) {
    using C = WgCfg<SPLIT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (blockIdx.x == 0 && threadIdx.x < 8) p.stamp_dst[threadIdx.x] = p.stamp[threadIdx.x];
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int w = blockIdx.x;
    const int xcd = w % p.nx, loc = w / p.nx;
    int g = loc * p.q;
    const int g1 = g + p.q < p.total ? g + p.q : p.total;
    for (int seg = 0; g < g1; ++seg) {
        const int j = g / p.steps_tile, s0 = g - j * p.steps_tile;
        const int n = p.steps_tile - s0 < g1 - g ? p.steps_tile - s0 : g1 - g;
        const int T = j * p.nx + xcd;
        const int item = T / p.tiles_item, tt = T - item * p.tiles_item;
        const int mt = tt / p.ntn, nt = tt - mt * p.ntn;
        const WgItem& q = p.it[item];
        f32x16 acc[TAPS][2];
        wg_run<SPLIT, TAPS>(smem, lds0, q.dz, q.ldz, q.x, q.ldx, mt, nt, s0 * C::ROWS, n, acc);
        // slab [k][128 co][64 ci]: lanes along ci (128-byte runs), the two lane halves on rows 4 apart; one per-lane offset, the
        // (tap, block, register) displacement in the scalar offset: no address registers beside the 160 accumulators
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.part + (long)(w * p.maxseg + seg) * (TAPS * 128 * 64)), 0,
                                                                            TAPS * 128 * 64 * 4, 0x00020000);
        const unsigned vo = (unsigned)(((wm * 64 + 4 * (lane >> 5)) * 64 + wn * 32 + (lane & 31)) * 4);
#pragma unroll
        for (int k = 0; k < TAPS; ++k)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[k][i][r]), rs, vo, ((k * 128 + i * 32 + (r & 3) + 8 * (r >> 2)) * 64) * 4, 0);
        g += n;
    }
}

}  // namespace efts

using namespace efts;

// geometry shared by the grouped launch and its reduction (efts_train.hip): steps per tile, lists, steps per workgroup, slabs per workgroup
int efts_wgrad_sk_geometry(int count, int rows, int cout, int cin, int split, int workgroups, efts_wgrad_sk_geom* gm) {
    if (count < 1 || count > EFTS_WGRAD_MAX_ITEMS) return efts_fail(EFTS_EINVAL, "efts_wgrad_tn_grouped: 1..%d items", EFTS_WGRAD_MAX_ITEMS);
    if (!(split == 1 || split == 2)) return efts_fail(EFTS_EINVAL, "efts_wgrad_tn_grouped: split must be 1 or 2");
    if (rows <= 0 || cout <= 0 || cin <= 0 || (cout & 127) || (cin & 63))
        return efts_fail(EFTS_ESHAPE, "efts_wgrad_tn_grouped: cout must be a multiple of 128, cin of 64");
    const int rps = split == 1 ? WgCfg<1>::ROWS : WgCfg<2>::ROWS;
    gm->steps_tile = (rows + rps - 1) / rps;
    gm->tiles_item = (cout / 128) * (cin / 64);
    gm->ntn = cin / 64;
    const int ntiles = count * gm->tiles_item;
    int G = workgroups > 0 ? workgroups : 2 * efts_num_cus();
    gm->nx = (ntiles % 8 == 0 && G >= 8) ? 8 : 1;
    G -= G % gm->nx;
    const long total = (long)(ntiles / gm->nx) * gm->steps_tile;          // steps of one list
    int per_list = G / gm->nx;
    if (per_list > total) per_list = (int)total;
    gm->q = (int)((total + per_list - 1) / per_list);
    per_list = (int)((total + gm->q - 1) / gm->q);                        // (no empty workgroups)
    gm->total = (int)total;
    gm->workgroups = per_list * gm->nx;
    gm->maxseg = (gm->q + gm->steps_tile - 1) / gm->steps_tile + 1;
    return 0;
}

extern "C" int64_t efts_wgrad_grouped_part_bytes(int32_t count, int32_t rows, int32_t cout, int32_t cin, int32_t taps, int32_t split, int32_t workgroups) {
    efts_wgrad_sk_geom gm;
    if (efts_wgrad_sk_geometry(count, rows, cout, cin, split, workgroups, &gm)) return -1;
    return (int64_t)gm.workgroups * gm.maxseg * taps * 128 * 64 * 4 + 64;        // (+ the geometry stamp)
}

extern "C" int efts_wgrad_tn_grouped(const efts_wgrad_item* items, int32_t count, float* part, int32_t rows, int32_t cout, int32_t cin,
                                     int32_t taps, int32_t split, int32_t workgroups, void* stream) {
    if (!items || !part) return efts_fail(EFTS_EINVAL, "efts_wgrad_tn_grouped: null pointer");
    if (!(taps == 1 || taps == 3 || taps == 5)) return efts_fail(EFTS_EINVAL, "efts_wgrad_tn_grouped: implemented for taps 1, 3, 5");
    efts_wgrad_sk_geom gm;
    const int rc = efts_wgrad_sk_geometry(count, rows, cout, cin, split, workgroups, &gm);
    if (rc) return rc;
    WgSkArgs k;
    for (int i = 0; i < EFTS_WGRAD_MAX_ITEMS; ++i) {
        const efts_wgrad_item* q = items + (i < count ? i : 0);
        if (!q->dz_plane || !q->x_plane) return efts_fail(EFTS_EINVAL, "efts_wgrad_tn_grouped: null operand plane");
        if ((q->ldz & 15) || (q->ldx & 15) || q->ldz < (int64_t)cout * 2 * split || q->ldx < (int64_t)cin * 2 * split || q->ldz > (1 << 20) || q->ldx > (1 << 20) ||
            ((uintptr_t)q->dz_plane & 15) || ((uintptr_t)q->x_plane & 15))
            return efts_fail(EFTS_EALIGN, "efts_wgrad_tn_grouped: plane strides / alignment");
        k.it[i].dz = (const char*)q->dz_plane; k.it[i].x = (const char*)q->x_plane; k.it[i].ldz = q->ldz; k.it[i].ldx = q->ldx;
    }
    k.part = part; k.tiles_item = gm.tiles_item; k.ntn = gm.ntn; k.nx = gm.nx; k.steps_tile = gm.steps_tile; k.q = gm.q; k.total = gm.total;
    k.maxseg = gm.maxseg;
    efts_wgrad_stamp(k.stamp, count, rows, cout, cin, taps, split, gm);
    k.stamp_dst = (int*)(part + (size_t)gm.workgroups * gm.maxseg * taps * 128 * 64);
    const dim3 grid(gm.workgroups);
#define EFTS_WGSK(S, T)                                                                                                                  \
    do {                                                                                                                                 \
        static bool attr = false;                                                                                                        \
        if (!attr) { (void)hipFuncSetAttribute((const void*)wgrad_sk_kernel<S, T>, hipFuncAttributeMaxDynamicSharedMemorySize, WgCfg<S>::LDS); attr = true; } \
        hipLaunchKernelGGL((wgrad_sk_kernel<S, T>), grid, dim3(256), WgCfg<S>::LDS, (hipStream_t)stream, k);                              \
    } while (0)
    if (split == 1) { if (taps == 5) EFTS_WGSK(1, 5); else if (taps == 3) EFTS_WGSK(1, 3); else EFTS_WGSK(1, 1); }
    else { if (taps == 5) EFTS_WGSK(2, 5); else if (taps == 3) EFTS_WGSK(2, 3); else EFTS_WGSK(2, 1); }
#undef EFTS_WGSK
    return efts_check_launch("efts_wgrad_tn_grouped");
}
