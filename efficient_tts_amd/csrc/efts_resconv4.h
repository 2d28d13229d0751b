// efts_resconv4.h -- efts_resconv5's ONE-WAVE-PER-SIMD kernel (included by efts_resconv.hip): the same residual k5 layer
//
//   y[row, :] = ( x[row, :] + LeakyReLU( sum_{tap<5} x[row + tap - 2, :] . W[tap] + bias ) ) * rowmask[row]
//
// (nntts/layers/efts_modules.py:48-51, :32-35), same tiles, schedule, planes and results as resconv5_kernel, for bf16 planes (split 1)
// and 5 taps.  What differs (DESIGN.md 4a'):
//   * 4 waves of 256 threads, one per SIMD, 512 registers each: wave w owns the tile's columns 64 w .. 64 w + 63 and ALL its h row blocks
//     (up to 16 accumulator blocks = a[0:255]), so every height h = 2..8 loads the four matrix pipes equally and nothing ping-pongs.
//   * the main loop is ONE generated inline-asm statement per tile height (efts_rc4_loop.inc, tools/gen_rc4_asm.py): all chunks x taps,
//     8 h MFMAs per step with the fragment reads, the operand staging (buffer_load -> VGPR -> ds_write_b128, spread evenly over the
//     step: no LDS-DMA, whose issue cost nothing hides on a one-wave SIMD) and one barrier per step placed by hand into the MFMAs'
//     shadows: 2 210 cycles per step at h = 8 for 2 048 of MFMA issue (tools/micro/rc4_loop_test.hip).
//   * LDS: two window buffers + TWO weight slots (a weight tile waits in registers for its slot) + 32 KiB of epilogue staging of its
//     own, so a tile's first operands stay put while the previous tile is swept out.
//   * the accumulators are held transposed (weight fragment = the MFMA's A operand), so the epilogue stages a 32 x 64 unit with eight
//     ds_write_b128 straight from the accumulator file; bias and activation are applied behind the transposition, in the row-major sweep.
// The sweep itself (residual hi + lo, mask, hi / lo / fp32 / sign outputs) computes exactly what rc_tile's does, value for value.
#pragma once

#include "efts_rc4_loop.inc"

namespace efts {

constexpr int RC4_STAGE = 4 * 32768;                 // LDS offset of the epilogue staging (4 waves x 8 KiB)

struct Rc4Ctx {
    char* smem;
    int lds0;
    int lane, wave;
    int n0;
    int state;              // bit 0: weight slot of the next step, bit 1: window buffer of the next tile's chunk 0
    int nst;
};

// plain staging of one 8-row piece set (kernel prologue only): pieces P = 4 q + wave, q < n, of rows [8 P, 8 P + 8) from `src` (row stride
// ld, rows clamped at rmax) into the piece-linear LDS image at `dst`
__device__ __forceinline__ void rc4_prime(char* dst, const char* src, long ld, int n, int rmax, int wave, int lane) {
    for (int q = 0; q < n; ++q) {
        const int P = 4 * q + wave, r = 8 * P + (lane >> 3);
        const int sl = (lane & 7) ^ ((r >> 1) & 7);
        const u32x4 d = *(const u32x4*)(src + (long)(r < rmax ? r : rmax) * ld + sl * 16);
        *(u32x4*)(dst + P * 1024 + lane * 16) = d;
    }
}

// the epilogue of one tile of H row blocks, general form: accumulators (a[0:255], transposed) -> staging -> row-major sweep, every
// per-layer switch read at run time (first / last layers of a stack, fp32 residual, sign bytes for training, [32 hi|32 lo] planes out).
// The layer in the middle of a stack on bf16 planes takes the generated sweep instead (RC4_EPI_H*, rc4_tile below).
template <int H>
__device__ __forceinline__ void rc4_epilogue(const RcArgs& p, const RcProb& pq, Rc4Ctx& c, int m0, int rows_out) {
    const int lane = c.lane, wave = c.wave;
    const bool f_noresid = pq.no_resid != 0;
    const bool f_sign = pq.sign != nullptr;
    const bool f_of32 = pq.out_f32 != nullptr;
    const bool f_ob = pq.ob != nullptr;
    const bool f_os2 = pq.out_split == 2;
    const bool f_oblo = pq.ob_lo != nullptr;
    const float slope = pq.slope;
    float bv[8];                                                   // bias of this lane's 8 sweep columns
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = pq.bias ? pq.bias[c.n0 + wave * 64 + (lane & 7) * 8 + e] : 0.f;
    const unsigned ldsg = (unsigned)pq.ldsg;
    char* const st = c.smem + RC4_STAGE + wave * 8192;             // [32 rows][64 channels] fp32, 16-byte slots XORed with (row & 7) << 1
    // staging write addresses: lane = time row (lane & 31); register quad (j, g) = channels 32 j + 8 g + 4 (lane >> 5) + 0..3 = slot j 8 + g 2 + (lane >> 5)
    unsigned wv[8];
    {
        const unsigned lrow = lane & 31, lhalf = lane >> 5;
        const unsigned base = (unsigned)c.lds0 + RC4_STAGE + wave * 8192 + lrow * 256;
#pragma unroll
        for (int q = 0; q < 8; ++q) wv[q] = base + ((((unsigned)q * 2 + lhalf) ^ ((lrow & 7) << 1)) << 4);
    }
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
    const int srow = lane >> 3;                       // row of the 8-row pass this lane handles
    const int c8 = lane & 7;                          // its 8 columns inside the wave's 64
    const char* const rd = st + srow * 256 + ((c8 ^ srow) << 5);       // this lane's 32 bytes of pass 0 (passes are 8 rows = 2 KiB apart)
    const unsigned col0 = c.n0 + wave * 64 + c8 * 8;
    const unsigned vx = res_f32 ? (unsigned)srow * (unsigned)pq.ldr * 4 + col0 * 4 : (unsigned)srow * (unsigned)pq.lda + col0 * 2;
    const unsigned sx_row = res_f32 ? (unsigned)pq.ldr * 4 : (unsigned)pq.lda;
    const unsigned vm = srow * 4;
    const unsigned vof = (unsigned)srow * (unsigned)pq.ldo * 4 + col0 * 4, sof_row = (unsigned)pq.ldo * 4;
    const unsigned vob = (unsigned)srow * (unsigned)pq.ldob + (!f_os2 ? col0 * 2 : (col0 >> 5) * 128 + (col0 & 31) * 2);
    const unsigned sob_row = (unsigned)pq.ldob;
    const unsigned vsg = (unsigned)srow * ldsg + (col0 >> 3);

    // operands of unit u (row block u): four 8-row passes -- 8 residual values (fp32, or bf16 hi + lo) and the row mask each; two units
    // are in flight ahead of the one being swept out (a wave alone on its SIMD has nobody to hide a load's latency behind)
    u32x4 xa[3][4], xb[3][4];
    float rmv[3][4];
    auto request = [&](int u, int bsel) {
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const unsigned rofs = u * 32 + pp * 8;
            const unsigned so = rofs * sx_row;
            if (f_noresid) { xa[bsel][pp] = u32x4{0, 0, 0, 0}; xb[bsel][pp] = xa[bsel][pp]; }
            else if (res_f32) {
                xa[bsel][pp] = __builtin_amdgcn_raw_buffer_load_b128(r_x, vx, so, 0);
                xb[bsel][pp] = __builtin_amdgcn_raw_buffer_load_b128(r_x, vx, so + 16, 0);
            } else {
                xa[bsel][pp] = __builtin_amdgcn_raw_buffer_load_b128(r_a, vx, so, 0);
                xb[bsel][pp] = __builtin_amdgcn_raw_buffer_load_b128(r_al, vx, so, 0);      // null plane: zeros
            }
            rmv[bsel][pp] = has_mask ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_m, vm, rofs * 4, 0)) : 1.f;
        }
    };
    // LDS serves one wave's accesses in issue order, and the staging block is wave-private: the read-back of unit u, the staging of
    // unit u + 1 into the same block and the next read-back need no waits between them -- only the values' own lgkmcnt.  So per unit:
    // all eight read-backs first (32 registers), then unit u + 1 goes into the block while unit u is worked on from registers.
    auto dump = [&](int u) {
        switch (u) {
            case 0: RC4_DUMP_UNIT0(wv[0], wv[1], wv[2], wv[3], wv[4], wv[5], wv[6], wv[7]); break;
            case 1: RC4_DUMP_UNIT1(wv[0], wv[1], wv[2], wv[3], wv[4], wv[5], wv[6], wv[7]); break;
            case 2: RC4_DUMP_UNIT2(wv[0], wv[1], wv[2], wv[3], wv[4], wv[5], wv[6], wv[7]); break;
            case 3: RC4_DUMP_UNIT3(wv[0], wv[1], wv[2], wv[3], wv[4], wv[5], wv[6], wv[7]); break;
            case 4: RC4_DUMP_UNIT4(wv[0], wv[1], wv[2], wv[3], wv[4], wv[5], wv[6], wv[7]); break;
            case 5: RC4_DUMP_UNIT5(wv[0], wv[1], wv[2], wv[3], wv[4], wv[5], wv[6], wv[7]); break;
            case 6: RC4_DUMP_UNIT6(wv[0], wv[1], wv[2], wv[3], wv[4], wv[5], wv[6], wv[7]); break;
            default: RC4_DUMP_UNIT7(wv[0], wv[1], wv[2], wv[3], wv[4], wv[5], wv[6], wv[7]); break;
        }
    };
    dump(0);
    request(0, 0);
    if (H > 1) request(1, 1);
#pragma unroll
    for (int u = 0; u < H; ++u) {
        const int bsel = u % 3;
        float4 q0[4], q1[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            q0[ps] = *(const float4*)(rd + ps * 2048);
            q1[ps] = *(const float4*)(rd + ps * 2048 + 16);
        }
        if (u + 1 < H) dump(u + 1);                                 // (its "memory" clobber keeps the read-backs above it)
        if (u + 2 < H) request(u + 2, (u + 2) % 3);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            float d[8] = {q0[ps].x, q0[ps].y, q0[ps].z, q0[ps].w, q1[ps].x, q1[ps].y, q1[ps].z, q1[ps].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {                          // bias + activation, as rc_tile applies them in front of its staging
                float v = d[e] + bv[e];
                d[e] = v > 0.f ? v : v * slope;
            }
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
            float y[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = (x[e] + d[e]) * rm;
            const unsigned brow = u * 32 + ps * 8;
            // (rows of the next tile / past the matrix: every output descriptor ends at this tile's last row, their stores are dropped)
            if (f_sign) {                                           // training: one byte of sign bits per lane (its 8 columns)
                unsigned sb = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) sb |= d[e] > 0.f ? (1u << e) : 0u;
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)sb, r_sg, vsg, brow * ldsg, 0);
            }
            if (f_of32) {
                const u32x4 o0 = {__float_as_uint(y[0]), __float_as_uint(y[1]), __float_as_uint(y[2]), __float_as_uint(y[3])};
                const u32x4 o1 = {__float_as_uint(y[4]), __float_as_uint(y[5]), __float_as_uint(y[6]), __float_as_uint(y[7])};
                store_b128(o0, r_of, vof, brow * sof_row);
                store_b128(o1, r_of, vof, brow * sof_row + 16);
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
}

// the main loop of one tile as one asm statement per height + the epilogue
template <int H>
__device__ __forceinline__ void rc4_tile(const RcArgs& p, const RcProb& pq, const RcProb& pn, Rc4Ctx& c, int m0, int rows_out, int m1, int h1) {
    const char* abase = pq.a + (long)(m0 - 2) * pq.lda;
    const char* a1base = h1 > 0 ? pn.a + (long)(m1 - 2) * pn.lda : abase;
    const __amdgpu_buffer_rsrc_t ars = make_rsrc(abase, 0x7fffffffL), a1rs = make_rsrc(a1base, 0x7fffffffL);
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(pq.w + (long)c.n0 * pq.ldw, 0x7fffffffL);
    const __amdgpu_buffer_rsrc_t w1rs = make_rsrc(pn.w + (long)c.n0 * pn.ldw, 0x7fffffffL);
    const int lda = __builtin_amdgcn_readfirstlane((int)pq.lda), lda1 = __builtin_amdgcn_readfirstlane((int)(h1 > 0 ? pn.lda : pq.lda));
    const int rmax = __builtin_amdgcn_readfirstlane(pq.m + 143 - (m0 - 2));
    const int rmax1 = __builtin_amdgcn_readfirstlane(h1 > 0 ? pn.m + 143 - (m1 - 2) : rmax);
    const int ldw = __builtin_amdgcn_readfirstlane((int)pq.ldw);
    const int wts = __builtin_amdgcn_readfirstlane((int)pq.w_tap_stride), wts1 = __builtin_amdgcn_readfirstlane((int)pn.w_tap_stride);
    const int nch = __builtin_amdgcn_readfirstlane(p.nchunk);
    const int wave = c.wave, state = __builtin_amdgcn_readfirstlane(c.state), lds0 = c.lds0;
    if constexpr (H == 2) RC4_LOOP_H2(ars, a1rs, wrs, w1rs, lda, rmax, lda1, rmax1, ldw, wts, wts1, nch, wave, state, lds0);
    if constexpr (H == 3) RC4_LOOP_H3(ars, a1rs, wrs, w1rs, lda, rmax, lda1, rmax1, ldw, wts, wts1, nch, wave, state, lds0);
    if constexpr (H == 4) RC4_LOOP_H4(ars, a1rs, wrs, w1rs, lda, rmax, lda1, rmax1, ldw, wts, wts1, nch, wave, state, lds0);
    if constexpr (H == 5) RC4_LOOP_H5(ars, a1rs, wrs, w1rs, lda, rmax, lda1, rmax1, ldw, wts, wts1, nch, wave, state, lds0);
    if constexpr (H == 6) RC4_LOOP_H6(ars, a1rs, wrs, w1rs, lda, rmax, lda1, rmax1, ldw, wts, wts1, nch, wave, state, lds0);
    if constexpr (H == 7) RC4_LOOP_H7(ars, a1rs, wrs, w1rs, lda, rmax, lda1, rmax1, ldw, wts, wts1, nch, wave, state, lds0);
    if constexpr (H == 8) RC4_LOOP_H8(ars, a1rs, wrs, w1rs, lda, rmax, lda1, rmax1, ldw, wts, wts1, nch, wave, state, lds0);
    // 5 nchunk steps later: the weight slot and the window buffer the NEXT tile starts on
    c.state = __builtin_amdgcn_readfirstlane(state ^ (((5 * nch) & 1) | ((nch & 1) << 1)));
    RC_MARK(p, c);
    // the layer in the middle of a stack on bf16 planes (residual from the hi + lo planes, row mask, hi + lo planes out, 0 <= slope <= 1 so
    // that LeakyReLU is max(v, v slope)): the generated, stage-major sweep
    const bool fast = pq.a_lo && !pq.resid && pq.rowmask && pq.ob && pq.ob_lo && pq.out_split == 1 && !pq.out_f32 && !pq.sign && !pq.no_resid &&
                      pq.slope >= 0.f && pq.slope <= 1.f;
    if (!fast) { rc4_epilogue<H>(p, pq, c, m0, rows_out); return; }
    const __amdgpu_buffer_rsrc_t r_a = make_rsrc(pq.a + (long)m0 * pq.lda, (long)rows_out * pq.lda);
    const __amdgpu_buffer_rsrc_t r_al = make_rsrc(pq.a_lo + (long)m0 * pq.lda, (long)rows_out * pq.lda);
    const __amdgpu_buffer_rsrc_t r_m = make_rsrc(pq.rowmask + m0, (long)rows_out * 4);
    const __amdgpu_buffer_rsrc_t r_ob = make_rsrc(pq.ob + (long)m0 * pq.ldob, (long)rows_out * pq.ldob);
    const __amdgpu_buffer_rsrc_t r_ol = make_rsrc(pq.ob_lo + (long)m0 * pq.ldob, (long)rows_out * pq.ldob);
    const __amdgpu_buffer_rsrc_t r_b = make_rsrc(pq.bias, pq.bias ? (long)p.ntn * RC_BN * 4 : 0);      // no bias: the loads return zeros
    const int ldob = __builtin_amdgcn_readfirstlane((int)pq.ldob), n0 = __builtin_amdgcn_readfirstlane(c.n0);
    const unsigned long slope2 = (unsigned long)__builtin_amdgcn_readfirstlane((int)__float_as_uint(pq.slope));
    if constexpr (H == 2) RC4_EPI_H2(r_a, r_al, r_m, r_ob, r_ol, r_b, lda, ldob, slope2, n0, wave, lds0);
    if constexpr (H == 3) RC4_EPI_H3(r_a, r_al, r_m, r_ob, r_ol, r_b, lda, ldob, slope2, n0, wave, lds0);
    if constexpr (H == 4) RC4_EPI_H4(r_a, r_al, r_m, r_ob, r_ol, r_b, lda, ldob, slope2, n0, wave, lds0);
    if constexpr (H == 5) RC4_EPI_H5(r_a, r_al, r_m, r_ob, r_ol, r_b, lda, ldob, slope2, n0, wave, lds0);
    if constexpr (H == 6) RC4_EPI_H6(r_a, r_al, r_m, r_ob, r_ol, r_b, lda, ldob, slope2, n0, wave, lds0);
    if constexpr (H == 7) RC4_EPI_H7(r_a, r_al, r_m, r_ob, r_ol, r_b, lda, ldob, slope2, n0, wave, lds0);
    if constexpr (H == 8) RC4_EPI_H8(r_a, r_al, r_m, r_ob, r_ol, r_b, lda, ldob, slope2, n0, wave, lds0);
}

__global__ __launch_bounds__(256, 1) void resconv5w4_kernel(RcArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Rc4Ctx c;
    c.smem = smem;
    c.lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem));
    const int tid = threadIdx.x;
    if (RC_STAMP == 1 && p.stamp && tid == 0) p.stamp[blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime();
    c.lane = tid & 63;
    c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    c.nst = 0;
    RC_MARK(p, c);
    // XCD-aware order, groups, classes and tiles: exactly resconv5_kernel's (same plans, same rows per workgroup)
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
    {
        // kernel prologue: the first tile's window chunk 0 into buffer 0 and its weight tile (tap 0, chunk 0) into slot 0
        const RcProb& q0 = p.pr[pi];
        c.state = 0;
        rc4_prime(smem, q0.a + (long)(m0 - 2) * q0.lda, q0.lda, h, q0.m + 143 - (m0 - 2), c.wave, c.lane);
        rc4_prime(smem + 2 * 32768, q0.w + (long)c.n0 * q0.ldw, q0.ldw, 8, 0x7fffffff, c.wave, c.lane);
    }
    for (int t = 0; h > 0; ++t) {
        const int vnext = vrow + rows_out;
        int pi1, m1, h1, rows1;
        locate(t + 1, vnext, pi1, m1, h1, rows1);
        const RcProb& pq = p.pr[pi];
        const RcProb& pn = p.pr[h1 > 0 ? pi1 : pi];
        switch (h) {
            case 2: rc4_tile<2>(p, pq, pn, c, m0, rows_out, m1, h1); break;
            case 3: rc4_tile<3>(p, pq, pn, c, m0, rows_out, m1, h1); break;
            case 4: rc4_tile<4>(p, pq, pn, c, m0, rows_out, m1, h1); break;
            case 5: rc4_tile<5>(p, pq, pn, c, m0, rows_out, m1, h1); break;
            case 6: rc4_tile<6>(p, pq, pn, c, m0, rows_out, m1, h1); break;
            case 7: rc4_tile<7>(p, pq, pn, c, m0, rows_out, m1, h1); break;
            default: rc4_tile<8>(p, pq, pn, c, m0, rows_out, m1, h1); break;
        }
        RC_MARK(p, c);
        vrow = vnext; pi = pi1; m0 = m1; h = h1; rows_out = rows1;
    }
    if (RC_STAMP == 1 && p.stamp && tid == 0) p.stamp[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
    if (RC_STAMP == 3 && p.stamp && tid == 0) p.stamp[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
}

}  // namespace efts
