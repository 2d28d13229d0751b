// efts_resconv.hip -- the residual k5 convolution layer of the EFTS-CNN stacks at mel length on gfx950 (CDNA4):
//
//   y[row, :] = ( x[row, :] + LeakyReLU( sum_{tap<5} x[row + tap - 2, :] . W[tap] + bias ) ) * rowmask[row]
//
// i.e. one `ResConv1d` layer of the reference (nntts/layers/efts_modules.py:48-51 forward, :32-35 the Conv1d,
// :77-79 the stack loop) over the padded row space of include/efts_abi.h.
//
// Why a kernel of its own: at 64 x 800 frames this layer is 134 GFLOP and ~80 % of the forward.
// The general contraction (efts_gemm.hip) runs two independent 4-wave workgroups per CU which collide on the matrix
// pipe and pay their HBM-bound epilogues in lock-step.  Here:
//   * ONE persistent 8-wave workgroup per CU, one barrier domain: 2 x 4 waves on a (32*h) x 256 tile, h = 2..8 half units
//     picked per tile; the upper wave row takes ceil(h/2) 32-row blocks, the lower one floor(h/2) (each SIMD hosts one wave
//     of either row, so odd heights still load the four matrix pipes equally).  A 256-column weight tile (32 KiB) feeds 8
//     waves, so the LDS-DMA line requests per MFMA are half those of a 128-column tile.
//   * LDS = two 256-row windows (double-buffered: the next K chunk's window lands while the current one is read at its
//     5 tap shifts) + a 3-stage ring of 32 KiB weight tiles = exactly 160 KiB.  All operands arrive by LDS-DMA issued
//     from inline asm with counted s_waitcnt vmcnt(N).
//   * ping-pong main loop: the two waves of a SIMD (one of each wave row) alternate -- while one issues the MFMAs of a half
//     (chunk, tap) step from fragments it already holds, the other reads its next fragments and issues its LDS-DMA pieces;
//     four s_barrier per step keep the rows a quarter step apart.
//   * the residual stream travels as bf16 hi/lo planes (x = hi + lo, 16 mantissa bits) instead of an fp32 copy
//     beside the bf16 operand plane: the epilogue reads 4 B and writes 4 B per element instead of 4 + 6..8 B.
//   * the epilogue is wave-private (no barriers): each wave drops one 32 x 32 accumulator block at a time into its own
//     4 KiB of LDS and sweeps it out as 16-byte row segments; the next tile's first operands are already in flight.
//   * a STATIC tile schedule (host side, `rc_schedule`): every CU gets its share of rows as a short list of tiles of
//     different heights, with two classes of workgroups whose epilogues (HBM bursts) fall at different times, and no
//     partial last round.
// Results are bit-identical to efts_gemm on the same operands (same per-element summation order: chunk, tap, k-slice).
#include "efts_resconv_tile.h"
#include "efts_resconv4.h"     // the one-wave-per-SIMD kernel (bf16 planes, 5 taps): hand-scheduled main loop

using namespace efts;

// the MODE 1 kernels (dgrad layer + fused activation backward) live in efts_resconv_bwd.hip
void efts_rc_launch_dgrad_act(int split, unsigned grid, void* stream, const efts::RcArgs& k);

// ---------------------------------------------------------------------------------------------------------------
// The static tile schedule.  `groups` workgroup groups (one workgroup per column tile each) share the m rows; a group's
// rows are cut into tiles of 32 * h - 4 rows, h = 2..8 half units.  Two classes of groups alternate (g % 2): when the rows allow it
// class 1 gets one 32-row half unit less than class 0, so that its last epilogue -- an HBM burst no MFMA work of the same CU
// can hide -- runs while class 0 still computes, and the epilogues in between fall at different times too.
// ---------------------------------------------------------------------------------------------------------------
static int rc_cover(int units) { return 32 * units - 4 * ((units + 7) / 8); }    // rows `units` half units yield in ceil(units / 8) tiles

static void rc_split(int units, bool descending, unsigned char* ni, int* ntile) {
    const int t = (units + 7) / 8;
    for (int i = 0; i < t; ++i) {
        const int v = units / t + (i < units % t ? 1 : 0);      // as even as possible, larger first
        ni[descending ? i : t - 1 - i] = (unsigned char)v;
    }
    *ntile = t;
}

static void rc_schedule(int m, int slots, RcSched* s, int* groups) {
    // slots = workgroup groups that can run at once (CUs / column tiles)
    int units = 2;                                               // half units; the smallest tile has two
    while (rc_cover(units) * (long)slots < m) ++units;           // every group `units` half units: covers m
    if (units > 8 * RC_MAXTILES) units = 8 * RC_MAXTILES;        // (beyond: more groups than slots, several rounds)
    s->ncls = 2;
    // (short tiles run at the LDS-DMA rate, where a class of even shorter ones only adds workgroups that stream the weights:
    //  measured 47.6 vs 49.3 us at 16 x 800 frames for heights 4|4 against 4|3, a tie from 7|6 upwards)
    const bool uneven = units >= 6 && slots >= 2 && (long)(rc_cover(units) + rc_cover(units - 1)) * (slots / 2) >= m;
    s->rows[0] = rc_cover(units);
    rc_split(units, true, s->ni[0], &s->ntile[0]);
    if (uneven) {
        s->rows[1] = rc_cover(units - 1);
        rc_split(units - 1, true, s->ni[1], &s->ntile[1]);
    } else {
        s->rows[1] = rc_cover(units);
        rc_split(units, false, s->ni[1], &s->ntile[1]);
    }
    const long pair = s->rows[0] + s->rows[1];
    long g = (m / pair) * 2;
    const long rem = m - (m / pair) * pair;
    if (rem > 0) g += rem > s->rows[0] ? 2 : 1;
    *groups = (int)g;
}

// plan <-> RcSched.  A plan is an int32 array: [groups, classes, then per class: rows, ntile, ni[0..7]]
constexpr int RC_PLAN_INTS = 2 + RC_MAXCLS * (2 + RC_MAXTILES);

static void rc_plan_write(const RcSched& s, int groups, int32_t* plan) {
    plan[0] = groups; plan[1] = s.ncls;
    for (int c = 0; c < RC_MAXCLS; ++c) {
        int32_t* q = plan + 2 + c * (2 + RC_MAXTILES);
        q[0] = c < s.ncls ? s.rows[c] : 0;
        q[1] = c < s.ncls ? s.ntile[c] : 0;
        for (int t = 0; t < RC_MAXTILES; ++t) q[2 + t] = (c < s.ncls && t < s.ntile[c]) ? s.ni[c][t] : 0;
    }
}

// a caller-supplied plan: every class's tiles must add up to its rows, and the groups must cover m
static const char* rc_plan_read(const int32_t* plan, int m, RcSched* s, int* groups) {
    *groups = plan[0];
    s->ncls = plan[1];
    if (s->ncls < 1 || s->ncls > RC_MAXCLS || *groups < 1) return "plan: classes must be 1..4 and groups positive";
    long sum = 0;
    for (int c = 0; c < s->ncls; ++c) {
        const int32_t* q = plan + 2 + c * (2 + RC_MAXTILES);
        s->rows[c] = q[0]; s->ntile[c] = q[1];
        if (q[1] < 1 || q[1] > RC_MAXTILES) return "plan: 1..8 tiles per class";
        int rows = 0;
        for (int t = 0; t < q[1]; ++t) {
            if (q[2 + t] < 2 || q[2 + t] > 8) return "plan: tile heights must be 2..8 (half units of 32 window rows)";
            s->ni[c][t] = (unsigned char)q[2 + t];
            rows += 32 * q[2 + t] - 4;
        }
        if (rows != q[0]) return "plan: a class's rows must equal the sum of 32 * h - 4 over its tiles";
        sum += rows;
    }
    long covered = (*groups / s->ncls) * sum;
    for (int c = 0; c < *groups % s->ncls; ++c) covered += s->rows[c];
    if (covered < m) return "plan: the groups do not cover m rows";
    return nullptr;
}

extern "C" int efts_resconv5_plan(int32_t m, int32_t n, int32_t cus, int32_t* plan, int32_t cap) {
    if (m <= 0 || n <= 0 || n % RC_BN) return efts_fail(EFTS_ESHAPE, "efts_resconv5_plan: m must be positive and n a positive multiple of 256");
    if (!plan || cap < RC_PLAN_INTS) return efts_fail(EFTS_EINVAL, "efts_resconv5_plan: plan must hold %d int32", RC_PLAN_INTS);
    if (cus <= 0) cus = efts_num_cus();
    const int ntn = n / RC_BN;
    RcSched s;
    int groups = 0;
    rc_schedule(m, cus / ntn > 0 ? cus / ntn : 1, &s, &groups);
    rc_plan_write(s, groups, plan);
    return RC_PLAN_INTS;
}

static int rc_check(const efts_resconv5_args* a, const char* who) {
    if (!(a->split == 1 || a->split == 2)) return efts_fail(EFTS_EINVAL, "%s: split must be 1 or 2", who);
    if (a->m <= 0 || a->n <= 0 || a->nchunk <= 0) return efts_fail(EFTS_ESHAPE, "%s: m, n, nchunk must be positive", who);
    if (a->n % RC_BN) return efts_fail(EFTS_ESHAPE, "%s: n must be a multiple of 256", who);
    if (!a->x || !a->w) return efts_fail(EFTS_EINVAL, "%s: null operand", who);
    if (!a->y && !a->y_f32) return efts_fail(EFTS_EINVAL, "%s: no output", who);
    if (((uintptr_t)a->x & 15) || ((uintptr_t)a->x_lo & 15) || ((uintptr_t)a->w & 15) || (a->ldx & 15) || (a->ldw & 15) || (a->w_tap_stride & 15) ||
        ((uintptr_t)a->y & 15) || ((uintptr_t)a->y_lo & 15) || (a->ldy & 15) || ((uintptr_t)a->x_f32 & 15) || (a->ldr & 3) ||
        ((uintptr_t)a->y_f32 & 15) || (a->ldo & 3))
        return efts_fail(EFTS_EALIGN, "%s: planes and fp32 streams must be 16-byte aligned (pointers and row strides)", who);
    if (a->ldx < (int64_t)a->nchunk * 128 || a->ldw < (int64_t)a->nchunk * 128)
        return efts_fail(EFTS_ESHAPE, "%s: row stride smaller than nchunk*128 bytes", who);
    if (a->ldx > (1 << 23) || a->ldw > (1 << 23)) return efts_fail(EFTS_ESHAPE, "%s: row stride above 8 MiB", who);
    if (a->y && !(a->y_split == 1 || a->y_split == 2)) return efts_fail(EFTS_EINVAL, "%s: y_split must be 1 or 2", who);
    if (a->split == 2 && a->x_lo) return efts_fail(EFTS_EINVAL, "%s: x_lo is for split-1 planes (split 2 carries lo inside x)", who);
    if (a->y_split == 2 && a->y_lo) return efts_fail(EFTS_EINVAL, "%s: y_lo is for split-1 output planes", who);
    if (!(a->taps == 0 || a->taps == 3 || a->taps == 5)) return efts_fail(EFTS_EINVAL, "%s: taps must be 5 (0 = default) or 3", who);
    return 0;
}

extern "C" int efts_resconv5_multi(const efts_resconv5_args* a, int32_t count, void* stream) {
    if (!a) return efts_fail(EFTS_EINVAL, "efts_resconv5: null args");
    if (count < 1 || count > RC_MAXPROB) return efts_fail(EFTS_EINVAL, "efts_resconv5_multi: 1..%d layers per launch", RC_MAXPROB);
    RcArgs k;
    long mtot = 0;
    for (int i = 0; i < count; ++i) {
        const efts_resconv5_args* q = a + i;
        const int rc = rc_check(q, count > 1 ? "efts_resconv5_multi" : "efts_resconv5");
        if (rc) return rc;
        if (q->split != a->split || q->n != a->n || q->nchunk != a->nchunk || q->ldw != a->ldw)
            return efts_fail(EFTS_ESHAPE, "efts_resconv5_multi: the layers of a launch must agree in split, n, nchunk and ldw");
        if (q->kernel != a->kernel) return efts_fail(EFTS_EINVAL, "efts_resconv5_multi: the layers of a launch must name the same kernel");
        RcProb& r = k.pr[i];
        r.a = (const char*)q->x; r.a_lo = (const char*)q->x_lo; r.resid = q->x_f32; r.w = (const char*)q->w;
        r.bias = q->bias; r.rowmask = q->rowmask; r.out_f32 = q->y_f32; r.ob = (char*)q->y; r.ob_lo = (char*)q->y_lo;
        r.lda = q->ldx; r.ldw = q->ldw; r.w_tap_stride = q->w_tap_stride; r.ldr = q->ldr; r.ldo = q->ldo; r.ldob = q->ldy;
        r.m = q->m; r.out_split = q->y_split; r.slope = q->slope; r.taps = q->taps == 3 ? 3 : 5; r.no_resid = q->no_residual ? 1 : 0; r.ldsg = q->n >> 3; r.sign = (char*)q->sign_bits;
        r.sign_in = (const char*)q->act_bwd_sign; r.bias_part = q->act_bwd_bias_part; r.slope_bwd = q->act_bwd_slope;
        if (q->act_bwd_sign) {
            if (count != 1) return efts_fail(EFTS_EINVAL, "efts_resconv5: a layer with the fused activation backward is launched alone");
            if (!q->x_f32 || !q->y_f32 || !q->y || !q->rowmask || q->bias || q->no_residual || q->sign_bits || q->y_lo || r.taps != 5 || q->y_split != q->split ||
                q->slope != 1.f)
                return efts_fail(EFTS_EINVAL, "efts_resconv5: act_bwd_sign needs the dgrad form (x_f32, y_f32, y in the operand format, rowmask, no bias, slope 1, 5 taps)");
        }
        mtot += q->m;
    }
    for (int i = count; i < RC_MAXPROB; ++i) k.pr[i] = k.pr[0];
    if (mtot > 0x3fffffffL) return efts_fail(EFTS_ESHAPE, "efts_resconv5: too many rows");
    k.nprob = count; k.m = (int)mtot; k.nchunk = a->nchunk; k.ntn = a->n / RC_BN;
    int groups = 0;
    if (a->plan) {
        const char* bad = rc_plan_read(a->plan, k.m, &k.s, &groups);
        if (bad) return efts_fail(EFTS_EINVAL, "efts_resconv5: %s", bad);
    } else {
        rc_schedule(k.m, efts_num_cus() / k.ntn > 0 ? efts_num_cus() / k.ntn : 1, &k.s, &groups);
    }
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)resconv5_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS);
        (void)hipFuncSetAttribute((const void*)resconv5_kernel<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS);
        attr = true;
    }
    const dim3 grid(groups * k.ntn);
    k.stamp = nullptr;
    k.bp_tiles = 0;
    for (int c = 0; c < k.s.ncls; ++c) k.bp_tiles = k.s.ntile[c] > k.bp_tiles ? k.s.ntile[c] : k.bp_tiles;
    if (a->act_bwd_sign) {
        if (a->act_bwd_bias_part && a->act_bwd_bias_rows < groups * k.bp_tiles * 2)
            return efts_fail(EFTS_ESHAPE, "efts_resconv5: act_bwd_bias_part needs %d rows (efts_resconv5_bias_rows)", groups * k.bp_tiles * 2);
        efts_rc_launch_dgrad_act(a->split, grid.x, stream, k);
        return efts_check_launch("efts_resconv5");
    }
#if RC_STAMP
    {
        static unsigned long long* base = nullptr;
        static int launch = 0;
        if (!base) { const char* e = getenv("EFTS_RC_STAMP"); if (e) base = (unsigned long long*)strtoull(e, nullptr, 16); }
        if (base) { k.stamp = base + (size_t)(launch % 64) * 1024; ++launch; }
    }
#endif
    // The one-wave-per-SIMD kernel with the generated main loop (efts_resconv4.h; bf16 planes, 5 taps) is launched on request only
    // (efts_resconv5_args.kernel = 2): its main loop needs 8 % fewer cycles than the ping-pong kernel's (2 210 vs ~2 400 per full step), but on
    // MI355X both run at the clock the power budget leaves (1.4-1.5 GHz with every CU issuing MFMAs on random operands) and take the same
    // time -- measured in one process: forward 1.596 vs 1.561 ms, training step 3.80 vs 3.70 ms, the 8-wave kernel ahead (DESIGN.md 4a').
    if (!(a->kernel == 0 || a->kernel == 2)) return efts_fail(EFTS_EINVAL, "efts_resconv5: kernel must be 0 (the 8-wave kernel) or 2");
    bool w4 = a->kernel == 2 && a->split == 1 && k.nchunk >= 2;
    for (int i = 0; i < count; ++i) w4 = w4 && k.pr[i].taps == 5;
    if (w4) {
        static bool attr4 = false;
        if (!attr4) { (void)hipFuncSetAttribute((const void*)resconv5w4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS); attr4 = true; }
        hipLaunchKernelGGL(resconv5w4_kernel, grid, dim3(256), RC_LDS, (hipStream_t)stream, k);
        return efts_check_launch("efts_resconv5");
    }
    if (a->split == 1) hipLaunchKernelGGL((resconv5_kernel<1, 0>), grid, dim3(512), RC_LDS, (hipStream_t)stream, k);
    else hipLaunchKernelGGL((resconv5_kernel<2, 0>), grid, dim3(512), RC_LDS, (hipStream_t)stream, k);
    return efts_check_launch("efts_resconv5");
}

extern "C" int efts_resconv5(const efts_resconv5_args* a, void* stream) { return efts_resconv5_multi(a, 1, stream); }

extern "C" int efts_resconv5_bias_rows(int32_t m, int32_t n) {
    if (m <= 0 || n <= 0 || n % RC_BN) return efts_fail(EFTS_ESHAPE, "efts_resconv5_bias_rows: m must be positive and n a positive multiple of 256");
    const int ntn = n / RC_BN;
    RcSched s;
    int groups = 0, tiles = 0;
    rc_schedule(m, efts_num_cus() / ntn > 0 ? efts_num_cus() / ntn : 1, &s, &groups);
    for (int c = 0; c < s.ncls; ++c) tiles = s.ntile[c] > tiles ? s.ntile[c] : tiles;
    return groups * tiles * 2;
}
