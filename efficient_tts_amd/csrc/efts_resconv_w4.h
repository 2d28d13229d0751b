// efts_resconv_w4.h -- LAB VARIANT of efts_resconv5's kernel (included by efts_resconv.hip under -DRC_W4=1 | 2 only; never part of the
// product library): one wave per SIMD.  Bit-identical to the product kernel on every shape tried, and SLOWER.  Where its time goes
// (tools/gpu_probe_rc_marks.py on -DRC_STAMP=3 builds of both kernels, one tile of 7 | 6 half units per workgroup = B = 32 x 800 frames,
// bf16, us per workgroup, same box):
//                                        main loop     epilogue     workgroup
//   product kernel (8 waves, ping-pong)     61.7         13.9          75.8
//   this one                                65.8         30.0          96.3
//   this one without LDS-DMA in the loop    48.9         28.9          78.2      (-DRC_EXP=1; without the step barrier: no change)
// i.e. the MFMA stream alone runs at ~70 % (34-36 us would be the issue rate), the LDS-DMA of the same 4 waves costs another 17-20 us
// (nothing overlaps its issue: in the product kernel the partner wave's MFMA phase does), and four waves sweep a tile out in twice the
// time of eight, however deep the prefetch (2 units ahead: no change).  The steady-state steps are free of spills (tools/rc_w4_loops.sh
// counts scratch / accumulator-move / lane-spill instructions per step from the ISA); a form with the loop's reads and MFMAs as volatile
// asm in a fixed interleaved order produced the intended ISA and the same time.  Other forms tried: wave rows of 4 + 3 blocks without
// the shared odd row block (153 us per B = 64 launch against 129: the SIMDs of the taller row set the pace), each wave all h blocks x
// 64 columns (247 us: spills), not-inlined tile functions (472 us: callees do not get the accumulator file).  Split-2 planes: 2 x slower
// than the product kernel (the shared row block's fragment selects).
// Build: tools/rc_w4_build.sh [1 | 2]  (2 = k5 layers only, half the compile time);  marks: add -DRC_STAMP=3.
// =================================================================================================================
// The same layer with ONE wave per SIMD (lab variant, -DRC_W4=1): 2 x 2 waves of up to 128 x 128 outputs on the same (32 h) x 256 tile.
// A wave tile twice as wide needs 8 fragment reads per 16 MFMAs instead of 12, so LDS bandwidth stops being the co-limit of the
// main loop, and a wave that has the matrix pipe to itself needs no ping-pong barriers: the fragments of the next k-slice are read
// under the MFMAs of the current one, ONE barrier per (chunk, tap) step, in front of its last k-slice -- behind it every wave has
// finished reading the step's weight tile, so the tile THREE steps ahead is requested into that slot right there (all three ring
// slots in use), and the next step's first fragments are read under the last slice's MFMAs.
// Rows: wave row wm takes the NF = h / 2 row blocks wm NF .. wm NF + NF - 1 over its 128 columns; an odd height leaves ONE more row
// block (XT = 1), which the four waves share by columns -- wave (wm, wn) takes its 32-column blocks 2 wm, 2 wm + 1 of it -- so every
// SIMD issues the same number of MFMAs at every height.  Up to 16 accumulator blocks = 256 registers per lane.
// =================================================================================================================
__device__ __forceinline__ void dma16u(unsigned lds_addr, unsigned voff, const char* sbase) {
    const unsigned long long a = (unsigned long long)sbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    dma16(__builtin_amdgcn_readfirstlane(lds_addr), voff, (const char*)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ void rc4_wait(int n) {           // s_waitcnt vmcnt(n), n = 0..24 at run time
#define RC4_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        RC4_W(1) RC4_W(2) RC4_W(3) RC4_W(4) RC4_W(5) RC4_W(6) RC4_W(7) RC4_W(8) RC4_W(9) RC4_W(10) RC4_W(11) RC4_W(12) RC4_W(13) RC4_W(14) RC4_W(15) RC4_W(16)
        RC4_W(17) RC4_W(18) RC4_W(19) RC4_W(20) RC4_W(21) RC4_W(22) RC4_W(23) RC4_W(24)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef RC4_W
}

struct Rc4Ctx {
    char* smem;
    unsigned lds0;
    int lane, wave, wm, wn, lrow, lhalf;
    int n0;
    unsigned vow[8];        // per-lane source offsets of this wave's 8 weight pieces (fixed per workgroup)
    const char* w_base;
    const char* w_next;
    long wts, wts_next;
    float bv[4];
    int ws;                 // ring slot of the next step
    int wpar;               // window buffer of the next tile's chunk 0
    int nst;
};

__device__ __forceinline__ int rc4_pieces(int h, int wave) { return (4 * h - wave + 3) >> 2; }   // pieces P = 4 q + wave < 4 h

template <int SPLIT, int NF, int XT, int TAPS>
__device__ __forceinline__ void rc4_tile(const RcArgs& p, const RcProb& pq, const RcProb& pn, Rc4Ctx& c, int m0, int h, int rows_out, int m1, int h1) {
    char* const smem = c.smem;
    const int lane = c.lane, wave = c.wave, lrow = c.lrow, lhalf = c.lhalf, wm = c.wm, wn = c.wn;
    const int row0w = wm * NF * 32;                         // first tile row of this wave's own row blocks
    constexpr int rowx = 2 * NF * 32;                       // ... of the shared one (XT)
    const int nq = rc4_pieces(h, wave), nq1 = rc4_pieces(h1, wave);
    constexpr int KS = SPLIT == 1 ? 4 : 2;                  // k-slices (fragment groups) per step
    constexpr int NP = SPLIT == 1 ? 1 : 2;                  // fragments per operand block and slice (hi [, lo])
    constexpr int NA = NF + XT;                             // A fragment blocks per slice

    // window pieces: piece P = 4 q + wave covers window rows 8 P .. 8 P + 7 (lane l: row 8 P + l / 8, physical slot (l & 7) ^ ((row >> 1) & 7));
    // rows past the guard band after the matrix are clamped (their outputs are never stored).  Per-lane offsets, ONE scalar base
    // per request group: (scalar per-piece bases cost more SGPRs than the kernel has -- their spills were reloaded from scratch inside
    // the loop, each reload behind an s_waitcnt vmcnt(0) that drained the LDS-DMA queue)
    unsigned voa[8];
    {
        const int rmax = pq.m + 143 - (m0 - 2);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = (q * 4 + wave) * 8 + (lane >> 3);
            const int sl = (lane & 7) ^ ((r >> 1) & 7);
            voa[q] = (unsigned)((r < rmax ? r : rmax) * (int)pq.lda + (sl << 4));
        }
    }
    const char* a_base = pq.a + (long)(m0 - 2) * pq.lda;
    auto issue_a = [&](int cn, int buf) {
        const char* sb = a_base + (long)cn * 128;
        const unsigned l = c.lds0 + buf * RC_WIN_BYTES + wave * 1024;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q < nq) dma16u(l + q * 4096, voa[q], sb);
    };
    auto issue_a_next = [&](int buf) {
        const int rmax = pn.m + 143 - (m1 - 2);
        const char* sb = pn.a + (long)(m1 - 2) * pn.lda;
        const unsigned l = c.lds0 + buf * RC_WIN_BYTES + wave * 1024;
        for (int q = 0; q < nq1; ++q) {
            const int r = (q * 4 + wave) * 8 + (lane >> 3);
            const int sl = (lane & 7) ^ ((r >> 1) & 7);
            dma16u(l + q * 4096, (unsigned)((r < rmax ? r : rmax) * (int)pn.lda + (sl << 4)), sb);
        }
    };

    f32x16 acc[NF][4], accx[XT ? 2 : 1];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int j = 0; j < (XT ? 2 : 1); ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accx[j][r] = 0.f;

    int a2 = 0, a1 = 0, g1 = 0;                            // window pieces of the groups issued two / one barrier ago, size of the latter
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // chunk 1's window (chunk 0 came with the previous tile's last chunks / the kernel prologue; the buffer was the epilogue's staging)
    if (p.nchunk > 1 && !(RC_EXP & 1)) { issue_a(1, (c.wpar ^ 1) & 1); a1 = nq; g1 = nq; }

    bf16x8 fa[2][NP][NA], fb[2][NP][4];
    const int bcol = wn * 128 + lrow;
    auto load = [&](int buf, const char* at, const char* wt, int tapoff, int s) {
        const int slot16 = s * 2 + lhalf;                   // split 1: 16 k per slice; split 2: hi at slot, lo at slot + 4
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            fb[buf][0][j] = *(const bf16x8*)(wt + lds_off(bcol + j * 32, slot16));
            if constexpr (SPLIT == 2) fb[buf][1][j] = *(const bf16x8*)(wt + lds_off(bcol + j * 32, slot16 + 4));
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int arow = (i < NF ? row0w + i * 32 : rowx) + lrow + tapoff;
            fa[buf][0][i] = *(const bf16x8*)(at + lds_off(arow, slot16));
            if constexpr (SPLIT == 2) fa[buf][1][i] = *(const bf16x8*)(at + lds_off(arow, slot16 + 4));
        }
    };
    auto mma3 = [&](f32x16& a, int buf, int i, int j) {
        if constexpr (SPLIT == 2) {                         // (the order of every other kernel inside a k-slice: lo*hi, hi*lo, hi*hi)
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][1][i], fb[buf][0][j], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][0][i], fb[buf][1][j], a, 0, 0, 0);
        }
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][0][i], fb[buf][0][j], a, 0, 0, 0);
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mma3(acc[i][j], buf, i, j);
        if constexpr (XT) {
            // this wave's two 32-column blocks of the shared row block: 2 wm, 2 wm + 1 of its own 128 columns.  The B fragments are
            // SELECTED (8 v_cndmask per slice), not branched on: with a branch the accumulators of the two arms met in phi nodes and
            // were copied between the accumulator and the vector file every step (160-190 v_accvgpr moves per step)
#pragma unroll
            for (int jx = 0; jx < 2; ++jx) {
                bf16x8 b0, b1;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    b0[e] = wm ? fb[buf][0][2 + jx][e] : fb[buf][0][jx][e];
                    if constexpr (SPLIT == 2) b1[e] = wm ? fb[buf][1][2 + jx][e] : fb[buf][1][jx][e];
                }
                if constexpr (SPLIT == 2) {
                    accx[jx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][1][NF], b0, accx[jx], 0, 0, 0);
                    accx[jx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][0][NF], b1, accx[jx], 0, 0, 0);
                }
                accx[jx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][0][NF], b0, accx[jx], 0, 0, 0);
            }
        }
    };
    auto interleave = [&]() {                              // the DS reads of the next slice between the MFMAs of this one
        constexpr int nread = NP * (NA + 4), nmma = (NF * 4 + 2 * XT) * (SPLIT == 1 ? 1 : 3);
        constexpr int per = nmma / nread > 0 ? nmma / nread : 1;
#pragma unroll
        for (int q = 0; q < nread; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, per, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, nmma - per * nread > 0 ? nmma - per * nread : 0, 0);
    };

    int ws = c.ws;
    load(0, smem + (c.wpar & 1) * RC_WIN_BYTES, smem + RC_RING + ws * RC_W_BYTES, (5 - TAPS) / 2, 0);
    int cur = 0;
    for (int ch = 0; ch < p.nchunk; ++ch) {
        const int wbuf = (c.wpar + ch) & 1;
        const bool lastc = ch + 1 == p.nchunk;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) {
            int kv = k;
            asm volatile("" : "+s"(kv));
            const char* at = smem + wbuf * RC_WIN_BYTES;
            const char* wt = smem + RC_RING + ws * RC_W_BYTES;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the fragments of this slice (read under the previous MFMAs)
                if (s == KS - 1) {
                    // Every read of this step's weight tile and (last tap) window is retired.  Behind the barrier the weight tile three
                    // steps ahead goes into this step's slot, and at a chunk's last tap the window TWO chunks ahead goes into this
                    // chunk's buffer (a whole chunk of slack for the one stream that comes from HBM).  The counter retires in order --
                    // ..., W(t-2), [A(t-2)], W(t-1), [A(t-1)] -- and the next step needs W(t-2): weights are issued in front of
                    // the window of the same group and the wait leaves A(t-2) and the whole group t-1 outstanding.
                    rc4_wait(a2 + g1);
                    if (!(RC_EXP & 2)) __builtin_amdgcn_s_barrier();
                    const int kn = (k + 3) % TAPS;
                    int cn = ch + (k + 3) / TAPS;
                    const bool into_next = cn >= p.nchunk;
                    cn = into_next ? cn - p.nchunk : cn;
                    const char* wsrc = (into_next ? c.w_next + (long)kn * c.wts_next : c.w_base + (long)kn * c.wts) + (long)cn * 128;
                    const unsigned wdst = c.lds0 + RC_RING + ws * RC_W_BYTES + wave * 1024;
                    int na = 0;
                    if (!(RC_EXP & 1)) {
#pragma unroll
                        for (int g = 0; g < 8; ++g) dma16u(wdst + g * 4096, c.vow[g], wsrc);
                        if (k == TAPS - 1) {
                            if (ch + 2 < p.nchunk) { issue_a(ch + 2, wbuf); na = nq; }
                            else if (ch + 2 == p.nchunk && h1 > 0) { issue_a_next(wbuf); na = nq1; }
                        }
                    }
                    a2 = a1; a1 = na; g1 = (RC_EXP & 1) ? 0 : 8 + na;
                    // first fragments of the next step under this slice's MFMAs (not across the epilogue: the next tile has other rows)
                    const bool more = !(lastc && k == TAPS - 1);
                    if (more) {
                        const int wsn = ws == 2 ? 0 : ws + 1;
                        const bool nextc = k == TAPS - 1;
                        int kn1 = nextc ? 0 : k + 1;
                        asm volatile("" : "+s"(kn1));
                        load(cur ^ 1, smem + (nextc ? wbuf ^ 1 : wbuf) * RC_WIN_BYTES, smem + RC_RING + wsn * RC_W_BYTES, kn1 + (5 - TAPS) / 2, 0);
                    }
                } else {
                    load(cur ^ 1, at, wt, kv + (5 - TAPS) / 2, s + 1);
                }
                mma(cur);
                interleave();
                cur ^= 1;
            }
            ws = (ws == 2) ? 0 : ws + 1;
        }
    }
    RC_MARK(p, c);
    c.ws = ws;
    c.wpar = (c.wpar + p.nchunk) & 1;

    // ---- epilogue, wave-private, 64 columns x 32 rows at a time (see rc_tile): staging in this wave's 8 KiB of the window buffer the
    // last chunk used (all three ring slots and the other window buffer hold the next tile's first operands)
    char* const st0 = smem + ((c.wpar ^ 1) & 1) * RC_WIN_BYTES + wave * 8192;
    char* const st1 = st0 + 4096;
    const __amdgpu_buffer_rsrc_t r_a = make_rsrc(pq.a + (long)m0 * pq.lda, (long)rows_out * pq.lda);
    const __amdgpu_buffer_rsrc_t r_al = make_rsrc(pq.a_lo ? pq.a_lo + (long)m0 * pq.lda : nullptr, pq.a_lo ? (long)rows_out * pq.lda : 0);
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(pq.resid ? pq.resid + (long)m0 * pq.ldr : nullptr, pq.resid ? (long)rows_out * pq.ldr * 4 : 0);
    const __amdgpu_buffer_rsrc_t r_m = make_rsrc(pq.rowmask ? pq.rowmask + m0 : nullptr, pq.rowmask ? (long)rows_out * 4 : 0);
    const __amdgpu_buffer_rsrc_t r_of = make_rsrc(pq.out_f32 ? pq.out_f32 + (long)m0 * pq.ldo : nullptr, pq.out_f32 ? (long)rows_out * pq.ldo * 4 : 0);
    const __amdgpu_buffer_rsrc_t r_ob = make_rsrc(pq.ob ? pq.ob + (long)m0 * pq.ldob : nullptr, pq.ob ? (long)rows_out * pq.ldob : 0);
    const __amdgpu_buffer_rsrc_t r_ol = make_rsrc(pq.ob_lo ? pq.ob_lo + (long)m0 * pq.ldob : nullptr, pq.ob_lo ? (long)rows_out * pq.ldob : 0);
    const __amdgpu_buffer_rsrc_t r_sg = make_rsrc(pq.sign ? pq.sign + (long)m0 * pq.ldsg : nullptr, pq.sign ? (long)rows_out * pq.ldsg : 0);
    const bool res_f32 = pq.resid != nullptr;
    const bool has_mask = pq.rowmask != nullptr;
    const int srow = lane >> 3, c8 = lane & 7;
    const char* const stl = (c8 < 4) ? st0 : st1;
    const unsigned sx_row = res_f32 ? (unsigned)pq.ldr * 4 : (unsigned)pq.lda;
    const unsigned sof_row = (unsigned)pq.ldo * 4, sob_row = (unsigned)pq.ldob;
    // one unit = one 32-row block x 64 columns: (first tile row, first column, the two accumulator blocks, their biases)
    constexpr int NU = NF * 2 + XT;
    u32x4 xa[3][4], xb[3][4];                               // residual values of three units: two in flight ahead of the one being swept out
    float rmv[3][4];
    auto unit_row = [&](int u) { return u < NF * 2 ? row0w + (u >> 1) * 32 : rowx; };
    auto unit_col = [&](int u) { return c.n0 + wn * 128 + (u < NF * 2 ? (u & 1) * 64 : wm * 64); };
    auto request = [&](int u, int bsel) {                  // the residual values and row masks of unit u: 4 passes of 8 rows
        const unsigned lrow0 = unit_row(u) + srow, col0 = unit_col(u) + c8 * 8;
        const unsigned vx = res_f32 ? lrow0 * (unsigned)pq.ldr * 4 + col0 * 4
                                    : lrow0 * (unsigned)pq.lda + (SPLIT == 1 ? col0 * 2 : (col0 >> 5) * 128 + (col0 & 31) * 2);
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const unsigned so = pp * 8 * sx_row;
            if (pq.no_resid) { xa[bsel][pp] = u32x4{0, 0, 0, 0}; xb[bsel][pp] = xa[bsel][pp]; }
            else if (res_f32) {
                xa[bsel][pp] = __builtin_amdgcn_raw_buffer_load_b128(r_x, vx, so, 0);
                xb[bsel][pp] = __builtin_amdgcn_raw_buffer_load_b128(r_x, vx, so + 16, 0);
            } else if (SPLIT == 1) {
                xa[bsel][pp] = __builtin_amdgcn_raw_buffer_load_b128(r_a, vx, so, 0);
                xb[bsel][pp] = __builtin_amdgcn_raw_buffer_load_b128(r_al, vx, so, 0);
            } else {
                xa[bsel][pp] = __builtin_amdgcn_raw_buffer_load_b128(r_a, vx, so, 0);
                xb[bsel][pp] = __builtin_amdgcn_raw_buffer_load_b128(r_a, vx, so + 64, 0);
            }
            rmv[bsel][pp] = has_mask ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_m, lrow0 * 4, pp * 32, 0)) : 1.f;
        }
    };
    request(0, 0);
    if (NU > 1) request(1, 1);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int bsel = u % 3;
        const unsigned lrow0 = unit_row(u) + srow, col0 = unit_col(u) + c8 * 8;
        const unsigned vof = lrow0 * (unsigned)pq.ldo * 4 + col0 * 4;
        const unsigned vob = lrow0 * (unsigned)pq.ldob + (pq.out_split == 1 ? col0 * 2 : (col0 >> 5) * 128 + (col0 & 31) * 2);
        const unsigned vsg = lrow0 * (unsigned)pq.ldsg + (col0 >> 3);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            char* const stj = j ? st1 : st0;
            const f32x16& a = u < NF * 2 ? acc[u >> 1][(u & 1) * 2 + j] : accx[XT ? j : 0];
            const float bj = u < NF * 2 ? c.bv[(u & 1) * 2 + j] : (wm == 0 ? c.bv[j] : c.bv[2 + j]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                float v = a[r] + bj;
                v = v > 0.f ? v : v * pq.slope;
                *(float*)(stj + rl * 128 + ((((lrow >> 2) ^ ((rl >> 1) & 1))) << 4) + (lrow & 3) * 4) = v;
            }
        }
        if (u + 2 < NU) request(u + 2, (u + 2) % 3);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = ps * 8 + srow;
            const int sw = (row >> 1) & 1;
            const float4 d0 = *(const float4*)(stl + row * 128 + ((((c8 & 3) * 2) ^ sw) << 4));
            const float4 d1 = *(const float4*)(stl + row * 128 + ((((c8 & 3) * 2 + 1) ^ sw) << 4));
            float x[8];
            const u32x4 qa = xa[bsel][ps], qb = xb[bsel][ps];
            if (res_f32) {
                x[0] = __uint_as_float(qa.x); x[1] = __uint_as_float(qa.y); x[2] = __uint_as_float(qa.z); x[3] = __uint_as_float(qa.w);
                x[4] = __uint_as_float(qb.x); x[5] = __uint_as_float(qb.y); x[6] = __uint_as_float(qb.z); x[7] = __uint_as_float(qb.w);
            } else {
                const unsigned ha[4] = {qa.x, qa.y, qa.z, qa.w}, lo[4] = {qb.x, qb.y, qb.z, qb.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[2 * e] = __uint_as_float(ha[e] << 16) + __uint_as_float(lo[e] << 16);
                    x[2 * e + 1] = __uint_as_float(ha[e] & 0xffff0000u) + __uint_as_float(lo[e] & 0xffff0000u);
                }
            }
            const float rm = rmv[bsel][ps];
            float y[8] = {(x[0] + d0.x) * rm, (x[1] + d0.y) * rm, (x[2] + d0.z) * rm, (x[3] + d0.w) * rm,
                          (x[4] + d1.x) * rm, (x[5] + d1.y) * rm, (x[6] + d1.z) * rm, (x[7] + d1.w) * rm};
            const unsigned brow = ps * 8;
            if ((int)(lrow0 + brow) >= rows_out) continue;
            if (pq.sign) {
                const unsigned sb = (d0.x > 0.f ? 1u : 0u) | (d0.y > 0.f ? 2u : 0u) | (d0.z > 0.f ? 4u : 0u) | (d0.w > 0.f ? 8u : 0u) |
                                    (d1.x > 0.f ? 16u : 0u) | (d1.y > 0.f ? 32u : 0u) | (d1.z > 0.f ? 64u : 0u) | (d1.w > 0.f ? 128u : 0u);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)sb, r_sg, vsg, brow * (unsigned)pq.ldsg, 0);
            }
            if (pq.out_f32) {
                const u32x4 o0 = {__float_as_uint(y[0]), __float_as_uint(y[1]), __float_as_uint(y[2]), __float_as_uint(y[3])};
                const u32x4 o1 = {__float_as_uint(y[4]), __float_as_uint(y[5]), __float_as_uint(y[6]), __float_as_uint(y[7])};
                store_b128(o0, r_of, vof, brow * sof_row);
                store_b128(o1, r_of, vof, brow * sof_row + 16);
            }
            if (pq.ob) {
                float rr[8];
                const u32x4 hi = {pack_bf16x2(y[0], y[1], &rr[0], &rr[1]), pack_bf16x2(y[2], y[3], &rr[2], &rr[3]),
                                  pack_bf16x2(y[4], y[5], &rr[4], &rr[5]), pack_bf16x2(y[6], y[7], &rr[6], &rr[7])};
                float d0_, d1_;
                const u32x4 lo = {pack_bf16x2(rr[0], rr[1], &d0_, &d1_), pack_bf16x2(rr[2], rr[3], &d0_, &d1_),
                                  pack_bf16x2(rr[4], rr[5], &d0_, &d1_), pack_bf16x2(rr[6], rr[7], &d0_, &d1_)};
                const unsigned so = brow * sob_row;
                store_b128(hi, r_ob, vob, so);
                if (pq.out_split == 2) store_b128(lo, r_ob, vob, so + 64);
                else if (pq.ob_lo) store_b128(lo, r_ol, vob, so);
            }
        }
    }
}

template <int SPLIT>
__global__ __launch_bounds__(256, 1) void resconv5w4_kernel(RcArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Rc4Ctx c;
    c.smem = smem;
    c.lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem));
    const int tid = threadIdx.x;
    c.lane = tid & 63;
    c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    c.wm = c.wave >> 1; c.wn = c.wave & 1;
    c.lrow = c.lane & 31; c.lhalf = c.lane >> 5;
    c.nst = 0;
    RC_MARK(p, c);
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
    int vrow = (g / p.s.ncls) * sum_rows + pre;
    const int vend = vrow + p.s.rows[cls] < p.m ? vrow + p.s.rows[cls] : p.m;
    const int ntile = p.s.ntile[cls];
    const int mfirst = p.nprob > 1 ? p.pr[0].m : p.m;
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
    if (h == 0) return;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = (q * 4 + c.wave) * 8 + (c.lane >> 3);
        const int sl = (c.lane & 7) ^ ((r >> 1) & 7);
        c.vow[q] = (unsigned)(r * (int)p.pr[0].ldw + (sl << 4));     // (every layer of a launch has the same weight row stride)
    }
    auto bias_of = [&](const RcProb& q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c.bv[j] = q.bias ? q.bias[c.n0 + c.wn * 128 + j * 32 + c.lrow] : 0.f;
    };
    {
        const RcProb& q0 = p.pr[pi];
        c.w_base = q0.w + (long)c.n0 * q0.ldw;
        c.wts = q0.w_tap_stride;
        bias_of(q0);
        c.ws = 0; c.wpar = 0;
        const int rmax = q0.m + 143 - (m0 - 2);
        const char* sb = q0.a + (long)(m0 - 2) * q0.lda;
        const int nq = rc4_pieces(h, c.wave);
        for (int q = 0; q < nq; ++q) {
            const int r = (q * 4 + c.wave) * 8 + (c.lane >> 3);
            const int sl = (c.lane & 7) ^ ((r >> 1) & 7);
            dma16u(c.lds0 + c.wave * 1024 + q * 4096, (unsigned)((r < rmax ? r : rmax) * (int)q0.lda + (sl << 4)), sb);
        }
        // weights of steps 0, 1, 2: (chunk 0, taps 0, 1, 2) for 3 or 5 taps
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int q = 0; q < 8; ++q)
                dma16u(c.lds0 + RC_RING + s * RC_W_BYTES + c.wave * 1024 + q * 4096, c.vow[q], c.w_base + (long)s * c.wts);
    }
    for (int t = 0; h > 0; ++t) {
        const int vnext = vrow + rows_out;
        int pi1, m1, h1, rows1;
        locate(t + 1, vnext, pi1, m1, h1, rows1);
        const RcProb& pq = p.pr[pi];
        const RcProb& pn = p.pr[h1 > 0 ? pi1 : pi];
        c.w_next = pn.w + (long)c.n0 * pn.ldw;
        c.wts_next = pn.w_tap_stride;
#define RC4_CASE(H, TP) case H: rc4_tile<SPLIT, H / 2, H & 1, TP>(p, pq, pn, c, m0, h, rows_out, m1, h1); break;
#if RC_W4 == 2                                  /* development build: k5 layers only (half the instantiations) */
        { switch (h) { RC4_CASE(2, 5) RC4_CASE(3, 5) RC4_CASE(4, 5) RC4_CASE(5, 5) RC4_CASE(6, 5) RC4_CASE(7, 5) default: rc4_tile<SPLIT, 4, 0, 5>(p, pq, pn, c, m0, h, rows_out, m1, h1); break; } }
#else
        if (pq.taps == 3) {
            switch (h) { RC4_CASE(2, 3) RC4_CASE(3, 3) RC4_CASE(4, 3) RC4_CASE(5, 3) RC4_CASE(6, 3) RC4_CASE(7, 3) default: rc4_tile<SPLIT, 4, 0, 3>(p, pq, pn, c, m0, h, rows_out, m1, h1); break; }
        } else {
            switch (h) { RC4_CASE(2, 5) RC4_CASE(3, 5) RC4_CASE(4, 5) RC4_CASE(5, 5) RC4_CASE(6, 5) RC4_CASE(7, 5) default: rc4_tile<SPLIT, 4, 0, 5>(p, pq, pn, c, m0, h, rows_out, m1, h1); break; }
        }
#endif
#undef RC4_CASE
        RC_MARK(p, c);
        if (h1 > 0 && pi1 != pi) bias_of(pn);
        c.w_base = c.w_next; c.wts = c.wts_next;
        vrow = vnext; pi = pi1; m0 = m1; h = h1; rows_out = rows1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (RC_STAMP == 3 && p.stamp && tid == 0) p.stamp[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
}

