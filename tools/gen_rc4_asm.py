#!/usr/bin/env python3
"""Generator of the hand-scheduled main loop of efts_resconv5's one-wave-per-SIMD kernel (csrc/efts_rc4_loop.inc).

    python tools/gen_rc4_asm.py            # rewrites efficient_tts_amd/csrc/efts_rc4_loop.inc

What is generated: for every tile height h = 2..8 (half units of 32 window rows) ONE inline-asm string holding the whole main loop of a
(32 h) x 256 tile of the residual k5 convolution (reference op: nntts/layers/efts_modules.py:48-51 -- conv1d k5 over 512 channels):
all K chunks x 5 taps, operand staging, barriers and 8 h MFMAs per wave and (chunk, tap) step, as a fixed instruction stream.  Why a
generated stream and not C++: three rounds of compiler-scheduled variants converged at 0.38-0.39 of the MFMA peak; the one-wave-per-SIMD
form needs its 256 accumulators pinned in the accumulator file and every other instruction placed into the 32-cycle shadow of an MFMA
(MI355X_MICROARCH.md: <= 5 single-issue fillers per v_mfma_f32_32x32x16_bf16), which hipcc neither guarantees nor keeps free of spills.

Geometry (DESIGN.md 4a'): 4 waves = one per SIMD; wave w owns output columns 64 w .. 64 w + 63 of the 256-column tile and ALL h row
blocks: acc block (i, j) = a[(2 i + j) 16 .. + 15], i < h, j < 2, held TRANSPOSED (the weight fragment is the MFMA's A operand: lane = time
row, register quad g = channels 32 j + 8 g + 4 (lane >> 5) + 0..3), so the epilogue stages it with ds_write_b128 from the accumulator
file.  LDS: two window buffers [256 rows][128 B] at 0 / 32 KiB, two weight
tiles [256 cols][128 B] at 64 / 96 KiB (16-byte slots XOR-swizzled by (row >> 1) & 7, as everywhere in this library), 32 KiB of
epilogue staging behind them.  Operands do NOT come by LDS-DMA: a `buffer_load ... lds` costs its wave 60-185 cycles of issue, which
nothing hides when the wave is the only one on its SIMD (the lab variant of round 3 lost 17-20 us per tile to exactly that); they are
loaded into VGPRs (`buffer_load_dwordx4`, a few cycles of issue) and dropped into LDS by `ds_write_b128` a step later, both in the
shadow of MFMAs.  Per step s = (chunk, tap), four k-slices of 2 h MFMAs each:
   slice 0: MFMAs on the fragments read earlier | reads of slice 1's fragments | vmcnt wait, 8 ds_write: weight tile of step s + 1
   slice 1:                                      | reads of slice 2            | 8 buffer_load: weight tile of step s + 2 (tap 0: + the
                                                                                 next chunk's window; tap 2: that window's ds_writes)
   slice 2:                                      | reads of slice 3            |
   slice 3: lgkmcnt(0), 2 MFMAs, s_barrier       | reads of step s + 1's slice 0 (other weight slot; next tap / next window buffer)
One barrier per step; a weight tile is written one barrier before it is read and overwritten one barrier after its last read.
The chunk loop is a real loop (5 taps unrolled); the LAST chunk has a body of its own: its window load fetches the NEXT tile's chunk 0
(other rows, possibly another layer of a grouped launch) and its last two weight loads the next tile's step 0 -- so a tile starts with
its first operands in LDS -- and it ends without loads in flight (registers do not survive the epilogue's compiler code).

The C++ side (efts_resconv.hip) passes 15 wave-uniform operands; everything per lane is computed inside from v_mbcnt.  Registers used
inside are fixed (register map below) and declared as clobbers; the accumulators stay in a[0:255] for the epilogue's dump statements.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "efficient_tts_amd", "csrc", "efts_rc4_loop.inc")

# ---------------------------------------------------------------------------------------------------------------- register map
V0 = 48                       # first VGPR owned by the asm block
FA = V0                       # FA[2][8][4]   A (window) fragments, double-buffered by slice parity
FB = FA + 64                  # FB[2][2][4]   B (weight) fragments
SW = FB + 16                  # SW[8][4]      staged weight pieces (one tile)
SA = SW + 32                  # SA[8][4]      staged window pieces
VOW = SA + 32                 # vow[8]        per-lane source offsets of this wave's weight pieces
VOA = VOW + 8                 # voa[8]        ... window pieces of this tile (rows clamped at rmax)
VOA1 = VOA + 8                # voa1[8]       ... of the next tile
OA0 = VOA1 + 8                # OA0[5]        A fragment read address per tap (slice 0, current window buffer)
OBR = OA0 + 5                 # B fragment read address (slice 0, current weight slot)
TA = OBR + 1                  # per-slice A read address
TB = TA + 1                   # per-slice B read address
WRA = TB + 1                  # window piece write address (the buffer being filled)
WRW = WRA + 1                 # weight piece write address (the slot being filled)
T0 = WRW + 1                  # temporaries T0 .. T0 + 5
VEND = T0 + 6
assert VEND <= 256, VEND

S0 = 76                       # first SGPR owned by the asm block
S_CH = S0                     # chunks left for the loop body
S_WOFF = S0 + 1               # soffset of the weight loads being issued
S_AOFF = S0 + 2               # soffset ((chunk + 1) * 128) of the next window load
S_TAP = S0 + 3                # S_TAP[5]: tap k * wts
S_WCH = S0 + 8                # chunk * 128
S_T = S0 + 9                  # scalar temporaries S_T, S_T + 1
SEND = S0 + 11
assert SEND <= 100

# operands of the asm statement (all "s"): four buffer descriptors (128-bit), then 32-bit scalars
OPS = ["arsrc", "a1rsrc", "wrsrc", "w1rsrc", "lda", "rmax", "lda1", "rmax1", "ldw", "wts", "wts1", "nchunk", "wave", "state", "lds0"]
OP = {n: i for i, n in enumerate(OPS)}


def v(i, n=1):
    return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"


def s(i, n=1):
    return f"s{i}" if n == 1 else f"s[{i}:{i + n - 1}]"


def a(i, n=1):
    return f"a{i}" if n == 1 else f"a[{i}:{i + n - 1}]"


def op(name):
    return f"%{OP[name]}"


class Gen:
    """ablate: set of {"mfma", "loads", "writes", "reads", "barrier"} left OUT (timing experiments only)"""

    def __init__(self, h, ablate=()):
        self.h, self.out, self.ab = h, [], set(ablate)
        self.label = 0

    def e(self, line):
        self.out.append(line)

    # ---- fragment reads of one k-slice into buffer p: B blocks 0..1 at TB, A blocks 0..h-1 at TA
    def reads(self, p):
        if "reads" in self.ab:
            return []
        r = [f"ds_read_b128 {v(FB + (p * 2 + j) * 4, 4)}, {v(TB)} offset:{j * 4096}" for j in range(2)]
        r += [f"ds_read_b128 {v(FA + (p * 8 + i) * 4, 4)}, {v(TA)} offset:{i * 4096}" for i in range(self.h)]
        return r

    def mfmas(self, p, first):
        m = []
        for i in range(self.h):
            for j in range(2):
                acc = a((2 * i + j) * 16, 16)
                src = "0" if first else acc
                m.append(None if "mfma" in self.ab else
                         f"v_mfma_f32_32x32x16_bf16 {acc}, {v(FB + (p * 2 + j) * 4, 4)}, {v(FA + (p * 8 + i) * 4, 4)}, {src}")
        return m

    def weave(self, mf, fillers, lead=0):
        """MFMAs with the fillers spread over the gaps behind them (`lead` MFMAs first without any)"""
        n = len(mf)
        slots = max(n - lead, 1)
        per = [[] for _ in range(n)]
        for q, f in enumerate(fillers):
            per[lead + min(q * slots // max(len(fillers), 1), slots - 1)].append(f)
        for t in range(n):
            if mf[t] is not None:
                self.e(mf[t])
            for f in per[t]:
                self.e(f)

    def step(self, k, last_chunk, first=False):
        """one (chunk, tap) step.  first: the tile's first step (accumulators start from 0); last_chunk: body of the last chunk.
        Staging traffic is SPREAD over slices 0-2 (three / three / two [wait, ds_write piece g, buffer_load piece g] groups): with all
        eight of a kind in one slice the four waves of the CU -- which run this stream in lock-step -- hit the one address path /
        the two LDS store paths of the CU at the same moment and every instruction cost 22-24 cycles of MFMA issue (micro benchmark:
        2 575 cycles per step against 2 110 without staging, 2 048 of MFMA issue)."""
        h = self.h
        final = last_chunk and k == 4
        wr, ld = "writes" not in self.ab, "loads" not in self.ab
        spread = "bunch" not in self.ab
        kn2, wrap = (k + 2) % 5, k + 2 >= 5
        # ---- the weight pieces: piece g of the NEXT step's tile goes from its staging registers into LDS, then the same registers
        # are re-loaded with piece g of the tile two steps ahead (not in the tile's final step)
        pre = []                    # scalar set-up of the loads' soffset / descriptor
        if not wrap:
            pre.append(f"s_add_u32 {s(S_WOFF)}, {s(S_TAP + kn2)}, {s(S_WCH)}")
            rs = op("wrsrc")
        elif not last_chunk:
            pre += [f"s_add_u32 {s(S_WOFF)}, {s(S_TAP + kn2)}, {s(S_WCH)}", f"s_add_u32 {s(S_WOFF)}, {s(S_WOFF)}, 0x80"]
            rs = op("wrsrc")
        else:                       # the next tile's step kn2 (its layer's descriptor and tap stride, chunk 0)
            pre.append(f"s_mul_i32 {s(S_WOFF)}, {op('wts1')}, {kn2}")
            rs = op("w1rsrc")
        # loads younger than piece g's when it is written: the other 7 pieces (+ at tap 1 the window pieces issued behind tap 0's)
        vm = 7 + (h if k == 1 else 0)
        groups = []
        for g in range(8):
            grp = []
            if wr:
                if "vwait" not in self.ab:
                    grp.append(f"s_waitcnt vmcnt({vm if (ld and not final) or g == 0 else max(vm - g, 0)})" if ld else "s_nop 0")
                grp.append(f"ds_write_b128 {v(WRW)}, {v(SW + g * 4, 4)} offset:{g * 4096}")
            if ld and not final:
                grp.append(f"buffer_load_dwordx4 {v(SW + g * 4, 4)}, {v(VOW + g)}, {rs}, {s(S_WOFF)} offen")
            groups.append(grp)
        if final and wr and ld and "vwait" not in self.ab:
            # no re-loads behind the writes: piece g is waited for with 7 - g younger loads in flight
            for g in range(8):
                groups[g][0] = f"s_waitcnt vmcnt({7 - g})"
        if spread:
            part = [groups[0] + groups[1] + groups[2], groups[3] + groups[4] + groups[5], groups[6] + groups[7]]
        else:
            part = [[x for g in groups for x in g if x.startswith(("s_waitcnt", "ds_write", "s_nop"))],
                    [x for g in groups for x in g if x.startswith("buffer_load")], []]
            if part[0]:
                part[0] = [f"s_waitcnt vmcnt({h if k == 1 else 0})"] + [x for x in part[0] if x.startswith("ds_write")]
        nwr = [sum(1 for x in pp if x.startswith("ds_write")) for pp in part]
        # ---- the window pieces: loaded at tap 0 (behind the weight loads), written at tap 2 (by then older than every load waited for)
        awin, awr = [], []
        if k == 0 and ld:
            if not last_chunk:      # the next chunk's window: h pieces per wave
                awin = [f"buffer_load_dwordx4 {v(SA + q * 4, 4)}, {v(VOA + q)}, {op('arsrc')}, {s(S_AOFF)} offen" for q in range(h)]
            else:                   # the next tile's chunk 0 (any height: all 8 pieces)
                awin = [f"s_mov_b32 {s(S_T)}, 0"] + [f"buffer_load_dwordx4 {v(SA + q * 4, 4)}, {v(VOA1 + q)}, {op('a1rsrc')}, {s(S_T)} offen" for q in range(8)]
        if k == 2 and wr:
            awr = [f"ds_write_b128 {v(WRA)}, {v(SA + q * 4, 4)} offset:{q * 4096}" for q in range(8 if last_chunk else h)]
        # ---------------- slice 0
        self.e("s_waitcnt lgkmcnt(0)")
        f0 = [f"v_xor_b32 {v(TA)}, 0x20, {v(OA0 + k)}", f"v_xor_b32 {v(TB)}, 0x20, {v(OBR)}"] + self.reads(1) + (pre if ld and not final else []) + part[0]
        self.weave(self.mfmas(0, first), f0)
        # ---------------- slice 1
        self.e(f"s_waitcnt lgkmcnt({nwr[0]})")
        f1 = [f"v_xor_b32 {v(TA)}, 0x40, {v(OA0 + k)}", f"v_xor_b32 {v(TB)}, 0x40, {v(OBR)}"] + self.reads(0) + part[1]
        if not spread:
            f1 = f1[:12] + (pre if ld and not final else []) + f1[12:] + awin + awr
        self.weave(self.mfmas(1, False), f1)
        n1 = nwr[1] + (len(awr) if not spread else 0)
        # ---------------- slice 2: + the precomputed fragment addresses of the next step's slice 0
        self.e(f"s_waitcnt lgkmcnt({n1})")
        kn = (k + 1) % 5
        f2 = [f"v_xor_b32 {v(TA)}, 0x60, {v(OA0 + k)}", f"v_xor_b32 {v(TB)}, 0x60, {v(OBR)}"] + self.reads(1) + part[2]
        if spread:
            f2 += awin + awr
        if not final:
            f2 += [f"v_xor_b32 {v(T0)}, 0x8000, {v(OBR)}",
                   (f"v_xor_b32 {v(T0 + 1)}, 0x8000, {v(OA0 + kn)}" if k == 4 else f"v_mov_b32 {v(T0 + 1)}, {v(OA0 + kn)}")]
        self.weave(self.mfmas(0, False), f2)
        # ---------------- slice 3: barrier, then at once the first fragments of the next step; the toggles behind them
        self.e("s_waitcnt lgkmcnt(0)")
        mf = self.mfmas(1, False)
        if final:
            self.weave(mf, [])
            return
        f3 = [] if "barrier" in self.ab else ["s_barrier"]
        f3 += [f"v_mov_b32 {v(TB)}, {v(T0)}", f"v_mov_b32 {v(TA)}, {v(T0 + 1)}"] + self.reads(0)
        f3 += [f"v_xor_b32 {v(OBR)}, 0x8000, {v(OBR)}", f"v_xor_b32 {v(WRW)}, 0x8000, {v(WRW)}"]
        if k == 4:      # the next step is tap 0 of the next chunk: the other window buffer; the chunk offsets advance
            f3 += [f"v_xor_b32 {v(OA0 + t)}, 0x8000, {v(OA0 + t)}" for t in range(5)] + [f"v_xor_b32 {v(WRA)}, 0x8000, {v(WRA)}"]
            f3 += [f"s_add_u32 {s(S_WCH)}, {s(S_WCH)}, 0x80", f"s_add_u32 {s(S_AOFF)}, {s(S_AOFF)}, 0x80"]
        self.weave(mf, f3, lead=min(2, len(mf) - 1))

    def setup(self):
        e = self.e
        e("; ---- lane id, fragment read addresses")
        e(f"v_mbcnt_lo_u32_b32 {v(T0)}, -1, 0")
        e(f"v_mbcnt_hi_u32_b32 {v(T0)}, -1, {v(T0)}")                      # T0 = lane
        e(f"v_and_b32 {v(T0 + 1)}, 31, {v(T0)}")                           # lrow
        e(f"v_lshrrev_b32 {v(T0 + 2)}, 5, {v(T0)}")                        # lhalf
        # OA0[k] = lds0 + wpar * 32768 + (lrow + k) * 128 + ((lhalf ^ (((lrow + k) >> 1) & 7)) << 4)       (state: bit 0 ws, bit 1 wpar)
        e(f"s_and_b32 {s(S_T)}, {op('state')}, 2")
        e(f"s_lshl_b32 {s(S_T)}, {s(S_T)}, 14")
        e(f"s_add_u32 {s(S_T)}, {s(S_T)}, {op('lds0')}")
        for k in range(5):
            e(f"v_add_u32 {v(T0 + 3)}, {k}, {v(T0 + 1)}")
            e(f"v_lshrrev_b32 {v(T0 + 4)}, 1, {v(T0 + 3)}")
            e(f"v_and_b32 {v(T0 + 4)}, 7, {v(T0 + 4)}")
            e(f"v_xor_b32 {v(T0 + 4)}, {v(T0 + 4)}, {v(T0 + 2)}")
            e(f"v_lshlrev_b32 {v(T0 + 4)}, 4, {v(T0 + 4)}")
            e(f"v_lshl_add_u32 {v(OA0 + k)}, {v(T0 + 3)}, 7, {v(T0 + 4)}")
            e(f"v_add_u32 {v(OA0 + k)}, {s(S_T)}, {v(OA0 + k)}")
        # OBR = lds0 + 65536 + ws * 32768 + (wave * 64 + lrow) * 128 + ((lhalf ^ ((lrow >> 1) & 7)) << 4)
        e(f"s_and_b32 {s(S_T)}, {op('state')}, 1")
        e(f"s_lshl_b32 {s(S_T)}, {s(S_T)}, 15")
        e(f"s_add_u32 {s(S_T)}, {s(S_T)}, {op('lds0')}")
        e(f"s_add_u32 {s(S_T)}, {s(S_T)}, 0x10000")
        e(f"s_lshl_b32 {s(S_T + 1)}, {op('wave')}, 13")
        e(f"s_add_u32 {s(S_T + 1)}, {s(S_T + 1)}, {s(S_T)}")
        e(f"v_lshrrev_b32 {v(T0 + 4)}, 1, {v(T0 + 1)}")
        e(f"v_and_b32 {v(T0 + 4)}, 7, {v(T0 + 4)}")
        e(f"v_xor_b32 {v(T0 + 4)}, {v(T0 + 4)}, {v(T0 + 2)}")
        e(f"v_lshlrev_b32 {v(T0 + 4)}, 4, {v(T0 + 4)}")
        e(f"v_lshl_add_u32 {v(OBR)}, {v(T0 + 1)}, 7, {v(T0 + 4)}")
        e(f"v_add_u32 {v(OBR)}, {s(S_T + 1)}, {v(OBR)}")
        e("; ---- piece write addresses: lane * 16 + wave * 1024 + region (the weight slot / window buffer NOT in use)")
        e(f"v_lshlrev_b32 {v(T0 + 3)}, 4, {v(T0)}")
        e(f"s_lshl_b32 {s(S_T + 1)}, {op('wave')}, 10")
        e(f"s_add_u32 {s(S_T + 1)}, {s(S_T + 1)}, {op('lds0')}")
        e(f"v_add_u32 {v(T0 + 3)}, {s(S_T + 1)}, {v(T0 + 3)}")
        e(f"s_and_b32 {s(S_T)}, {op('state')}, 1")
        e(f"s_xor_b32 {s(S_T)}, {s(S_T)}, 1")
        e(f"s_lshl_b32 {s(S_T)}, {s(S_T)}, 15")
        e(f"s_add_u32 {s(S_T)}, {s(S_T)}, 0x10000")
        e(f"v_add_u32 {v(WRW)}, {s(S_T)}, {v(T0 + 3)}")
        e(f"s_and_b32 {s(S_T)}, {op('state')}, 2")
        e(f"s_xor_b32 {s(S_T)}, {s(S_T)}, 2")
        e(f"s_lshl_b32 {s(S_T)}, {s(S_T)}, 14")
        e(f"v_add_u32 {v(WRA)}, {s(S_T)}, {v(T0 + 3)}")
        e("; ---- piece source offsets: piece P = 4 q + wave = rows 8 P .. 8 P + 7; lane: row r = 8 P + (lane >> 3), logical slot (lane & 7) ^ ((r >> 1) & 7)")
        e(f"v_lshrrev_b32 {v(T0 + 1)}, 3, {v(T0)}")
        e(f"v_and_b32 {v(T0 + 2)}, 7, {v(T0)}")
        e(f"s_lshl_b32 {s(S_T)}, {op('wave')}, 3")
        for q in range(8):
            e(f"v_add_u32 {v(T0 + 3)}, {q * 32}, {v(T0 + 1)}")
            e(f"v_add_u32 {v(T0 + 3)}, {s(S_T)}, {v(T0 + 3)}")             # r
            e(f"v_lshrrev_b32 {v(T0 + 4)}, 1, {v(T0 + 3)}")
            e(f"v_and_b32 {v(T0 + 4)}, 7, {v(T0 + 4)}")
            e(f"v_xor_b32 {v(T0 + 4)}, {v(T0 + 4)}, {v(T0 + 2)}")
            e(f"v_lshlrev_b32 {v(T0 + 4)}, 4, {v(T0 + 4)}")                # slot * 16
            e(f"v_mul_lo_u32 {v(VOW + q)}, {v(T0 + 3)}, {op('ldw')}")
            e(f"v_add_u32 {v(VOW + q)}, {v(VOW + q)}, {v(T0 + 4)}")
            e(f"v_min_u32 {v(T0 + 5)}, {op('rmax')}, {v(T0 + 3)}")
            e(f"v_mul_lo_u32 {v(VOA + q)}, {v(T0 + 5)}, {op('lda')}")
            e(f"v_add_u32 {v(VOA + q)}, {v(VOA + q)}, {v(T0 + 4)}")
            e(f"v_min_u32 {v(T0 + 5)}, {op('rmax1')}, {v(T0 + 3)}")
            e(f"v_mul_lo_u32 {v(VOA1 + q)}, {v(T0 + 5)}, {op('lda1')}")
            e(f"v_add_u32 {v(VOA1 + q)}, {v(VOA1 + q)}, {v(T0 + 4)}")
        e("; ---- scalar state")
        for k in range(5):
            e(f"s_mul_i32 {s(S_TAP + k)}, {op('wts')}, {k}")
        e(f"s_mov_b32 {s(S_WCH)}, 0")
        e(f"s_mov_b32 {s(S_AOFF)}, 0x80")
        e(f"s_sub_u32 {s(S_CH)}, {op('nchunk')}, 1")                        # passes of the loop body (the last chunk has its own)
        e("; ---- prologue: weight tile of step 1 into the staging registers; everything the previous tile / the kernel prologue left in LDS is visible")
        if "loads" not in self.ab:
            for g in range(8):
                e(f"buffer_load_dwordx4 {v(SW + g * 4, 4)}, {v(VOW + g)}, {op('wrsrc')}, {s(S_TAP + 1)} offen")
        e("s_waitcnt lgkmcnt(0)")
        if "barrier" not in self.ab:
            e("s_barrier")
        e(f"v_mov_b32 {v(TA)}, {v(OA0)}")
        e(f"v_mov_b32 {v(TB)}, {v(OBR)}")
        for r in self.reads(0):
            e(r)

    def build(self):
        e = self.e
        self.setup()
        # the first chunk's first step starts the accumulators from 0: peel it when there is a loop body to peel it from
        e("; ==== first step of the tile")
        e(f"s_cmp_eq_u32 {s(S_CH)}, 0")
        e("s_cbranch_scc1 9f")                       # (one chunk only is not supported: the C++ side never asks for it)
        self.step(0, False, first=True)
        for k in range(1, 5):
            self.step(k, False)
        e(f"s_sub_u32 {s(S_CH)}, {s(S_CH)}, 1")
        e(f"s_cmp_eq_u32 {s(S_CH)}, 0")
        e("s_cbranch_scc1 2f")
        e("1:")
        for k in range(5):
            self.step(k, False)
        e(f"s_sub_u32 {s(S_CH)}, {s(S_CH)}, 1")
        e(f"s_cmp_lg_u32 {s(S_CH)}, 0")
        e("s_cbranch_scc1 1b")
        e("2:")
        e("; ==== last chunk")
        for k in range(5):
            self.step(k, True)
        e("9:")
        e("s_waitcnt lgkmcnt(0)")
        e("s_nop 15")                                # the last MFMAs' results before anything reads the accumulators (16 passes: 18 states)
        e("s_nop 7")
        return self.out

    def clobbers(self):
        return [f"v{i}" for i in range(V0, VEND)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(S0, SEND)] + ["scc", "memory"]


# ---------------------------------------------------------------------------------------------------------------- the FAST epilogue
# One asm statement per tile height for the layer in the middle of a stack on bf16 planes (residual from the hi + lo planes, row mask,
# hi + lo planes out).  Why asm here too: a wave alone on its SIMD has no partner to fill the gaps behind dependent VALU results, and hipcc
# emits the sweep chain by chain (shift, and, add, add, mul, cvt, shift, and, fma: ~7.5 cycles per instruction, 12.8 us per tile against
# 8.1 us for the 8-wave kernel's two interleaved waves).  Here the 16 value pairs of two 8-row passes go through every stage together, so
# a result is needed 8 instructions after it was issued.  The operations and their order per value are exactly the compiler's (h + l, + d,
# * m, cvt, fma(s, m, -g): the remainder keeps the contraction the C++ epilogues compile to), hence the same bits.
E0 = 16                       # first VGPR of the epilogue statement (v0 .. v15 stay with the compiler)
E_WV = E0                     # WV[8]     staging (dump) addresses
E_RD = E_WV + 8               # read-back address of pass 0
E_VX = E_RD + 1               # residual offset (both planes), E_VM mask offset, E_VOB output offset (both planes)
E_VM = E_VX + 1
E_VOB = E_VM + 1
E_T = E_VOB + 1               # 4 temporaries
E_BIAS = 32                   # BIAS[8] (pairs, even-aligned)
E_Q = E_BIAS + 8              # Q[32]     read-back of the unit: pass ps element e at Q + 8 ps + e
E_XA = E_Q + 32               # XA[3][16] hi words of three units in flight (pass ps: 4 registers)
E_XB = E_XA + 48              # XB[3][16] lo words
E_RM = E_XB + 48              # RM[3][8]  row masks (one even-aligned pair slot per pass)
E_T1 = E_RM + 24              # T1[16], T2[16]  stage temporaries of one half unit (2 passes x 4 pairs)
E_T2 = E_T1 + 16
E_OUT = E_T2 + 16             # OUT[2][16] packed hi (8) + lo (8) words of a half unit, double-buffered (store data must stay put)
E_END = E_OUT + 32
assert E_END <= 256, E_END
ES0 = 76                      # SGPRs: ES0 .. ES0 + 7 temporaries
EOPS = ["ra", "ral", "rm", "rob", "rol", "rbias", "lda", "ldob", "slope2", "n0", "wave", "lds0"]
EOP = {n: i for i, n in enumerate(EOPS)}


def eop(name):
    return f"%{EOP[name]}"


def gen_epilogue(h):
    out = []
    e = out.append
    vm_ops = []                                   # VMEM ops in issue order: (kind, unit)

    def loads(u):                                 # residual hi / lo words and row masks of unit u -> set u % 3
        st = u % 3
        for ps in range(4):
            rofs = u * 32 + ps * 8
            e(f"s_mul_i32 {s(ES0)}, {eop('lda')}, {rofs}")
            e(f"buffer_load_dwordx4 {v(E_XA + st * 16 + ps * 4, 4)}, {v(E_VX)}, {eop('ra')}, {s(ES0)} offen")
            e(f"buffer_load_dwordx4 {v(E_XB + st * 16 + ps * 4, 4)}, {v(E_VX)}, {eop('ral')}, {s(ES0)} offen")
            e(f"buffer_load_dword {v(E_RM + st * 8 + ps * 2)}, {v(E_VM)}, {eop('rm')}, 0 offen offset:{rofs * 4}")
            vm_ops.extend([("ld", u)] * 3)

    def dump(u):
        for j in range(2):
            for g in range(4):
                base = (2 * u + j) * 16 + 4 * g
                e(f"ds_write_b128 {v(E_WV + j * 4 + g)}, {a(base, 4)}")

    # ---- per-lane constants
    e("v_mbcnt_lo_u32_b32 " + v(E_T) + ", -1, 0")
    e(f"v_mbcnt_hi_u32_b32 {v(E_T)}, -1, {v(E_T)}")                       # lane
    e(f"v_and_b32 {v(E_T + 1)}, 31, {v(E_T)}")                            # lrow (time row of the accumulator layout)
    e(f"v_lshrrev_b32 {v(E_T + 2)}, 5, {v(E_T)}")                         # lhalf
    e(f"s_lshl_b32 {s(ES0)}, {eop('wave')}, 13")
    e(f"s_add_u32 {s(ES0)}, {s(ES0)}, {eop('lds0')}")
    e(f"s_add_u32 {s(ES0)}, {s(ES0)}, 0x20000")                           # staging block of this wave (RC4_STAGE = 128 KiB)
    e(f"v_and_b32 {v(E_T + 3)}, 7, {v(E_T + 1)}")
    e(f"v_lshlrev_b32 {v(E_T + 3)}, 1, {v(E_T + 3)}")                     # (lrow & 7) << 1
    e(f"v_lshlrev_b32 {v(E_RD)}, 8, {v(E_T + 1)}")                        # lrow * 256
    e(f"v_add_u32 {v(E_RD)}, {s(ES0)}, {v(E_RD)}")
    for q in range(8):
        e(f"v_or_b32 {v(E_WV + q)}, {2 * q}, {v(E_T + 2)}")               # slot = 2 q + lhalf
        e(f"v_xor_b32 {v(E_WV + q)}, {v(E_WV + q)}, {v(E_T + 3)}")
        e(f"v_lshl_add_u32 {v(E_WV + q)}, {v(E_WV + q)}, 4, {v(E_RD)}")
    e(f"v_lshrrev_b32 {v(E_T + 1)}, 3, {v(E_T)}")                         # srow
    e(f"v_and_b32 {v(E_T + 2)}, 7, {v(E_T)}")                             # c8
    e(f"v_xor_b32 {v(E_T + 3)}, {v(E_T + 1)}, {v(E_T + 2)}")
    e(f"v_lshlrev_b32 {v(E_T + 3)}, 5, {v(E_T + 3)}")                     # (c8 ^ srow) * 32
    e(f"v_lshl_add_u32 {v(E_RD)}, {v(E_T + 1)}, 8, {v(E_T + 3)}")         # srow * 256 + ...
    e(f"v_add_u32 {v(E_RD)}, {s(ES0)}, {v(E_RD)}")
    e(f"s_lshl_b32 {s(ES0 + 1)}, {eop('wave')}, 6")
    e(f"s_add_u32 {s(ES0 + 1)}, {s(ES0 + 1)}, {eop('n0')}")               # n0 + wave * 64
    e(f"v_lshl_add_u32 {v(E_T + 3)}, {v(E_T + 2)}, 3, {s(ES0 + 1)}")      # col0 = n0 + wave 64 + c8 8
    e(f"v_mul_lo_u32 {v(E_VX)}, {v(E_T + 1)}, {eop('lda')}")
    e(f"v_lshl_add_u32 {v(E_VX)}, {v(E_T + 3)}, 1, {v(E_VX)}")            # srow * lda + col0 * 2
    e(f"v_mul_lo_u32 {v(E_VOB)}, {v(E_T + 1)}, {eop('ldob')}")
    e(f"v_lshl_add_u32 {v(E_VOB)}, {v(E_T + 3)}, 1, {v(E_VOB)}")
    e(f"v_lshlrev_b32 {v(E_VM)}, 2, {v(E_T + 1)}")                        # srow * 4
    e(f"v_lshlrev_b32 {v(E_T + 3)}, 2, {v(E_T + 3)}")                     # col0 * 4
    e(f"buffer_load_dwordx4 {v(E_BIAS, 4)}, {v(E_T + 3)}, {eop('rbias')}, 0 offen")
    e(f"buffer_load_dwordx4 {v(E_BIAS + 4, 4)}, {v(E_T + 3)}, {eop('rbias')}, 0 offen offset:16")
    vm_ops.extend([("bias", -1)] * 2)
    loads(0)
    if h > 1:
        loads(1)
    dump(0)
    for u in range(h):
        st = u % 3
        for ps in range(4):
            e(f"ds_read_b128 {v(E_Q + ps * 8, 4)}, {v(E_RD)} offset:{ps * 2048}")
            e(f"ds_read_b128 {v(E_Q + ps * 8 + 4, 4)}, {v(E_RD)} offset:{ps * 2048 + 16}")
        nd = 0
        if u + 1 < h:
            dump(u + 1)
            nd = 8
        if u + 2 < h:
            loads(u + 2)
        last = max(i for i, (k, uu) in enumerate(vm_ops) if k == "ld" and uu == u)
        e(f"s_waitcnt vmcnt({len(vm_ops) - 1 - last})")
        e(f"s_waitcnt lgkmcnt({nd})")
        for half in range(2):
            ob = E_OUT + ((2 * u + half) % 2) * 16
            P = [2 * half, 2 * half + 1]
            chains = [(ps, k) for ps in P for k in range(4)]            # 8 pair chains: pass ps, pair k (elements 2 k, 2 k + 1)
            def Q2(ps, k): return v(E_Q + ps * 8 + 2 * k, 2)
            def T1(i): return E_T1 + 2 * i
            def T2(i): return E_T2 + 2 * i
            # 1. d += bias      2. t = d * slope      3. d = max(d, t)
            for i, (ps, k) in enumerate(chains):
                e(f"v_pk_add_f32 {Q2(ps, k)}, {Q2(ps, k)}, {v(E_BIAS + 2 * k, 2)}")
            for i, (ps, k) in enumerate(chains):
                e(f"v_pk_mul_f32 {v(T1(i), 2)}, {eop('slope2')}, {Q2(ps, k)} op_sel_hi:[0,1]")
            for i, (ps, k) in enumerate(chains):
                e(f"v_max_f32 {v(E_Q + ps * 8 + 2 * k)}, {v(E_Q + ps * 8 + 2 * k)}, {v(T1(i))}")
                e(f"v_max_f32 {v(E_Q + ps * 8 + 2 * k + 1)}, {v(E_Q + ps * 8 + 2 * k + 1)}, {v(T1(i) + 1)}")
            # 4. unpack hi -> T1, lo -> T2     5. x = h + l     6. s = x + d     7. y = s * m
            for i, (ps, k) in enumerate(chains):
                H, L = E_XA + st * 16 + ps * 4 + k, E_XB + st * 16 + ps * 4 + k
                e(f"v_lshlrev_b32 {v(T1(i))}, 16, {v(H)}")
                e(f"v_and_b32 {v(T1(i) + 1)}, 0xffff0000, {v(H)}")
                e(f"v_lshlrev_b32 {v(T2(i))}, 16, {v(L)}")
                e(f"v_and_b32 {v(T2(i) + 1)}, 0xffff0000, {v(L)}")
            for i, (ps, k) in enumerate(chains):
                e(f"v_pk_add_f32 {v(T1(i), 2)}, {v(T1(i), 2)}, {v(T2(i), 2)}")
            for i, (ps, k) in enumerate(chains):
                e(f"v_pk_add_f32 {v(T1(i), 2)}, {v(T1(i), 2)}, {Q2(ps, k)}")
            for i, (ps, k) in enumerate(chains):
                e(f"v_pk_mul_f32 {v(T2(i), 2)}, {v(T1(i), 2)}, {v(E_RM + st * 8 + ps * 2, 2)} op_sel_hi:[1,0]")
            # 8. hi' = cvt(y)      9. g = unpack(hi')      10. rem = fma(s, m, -g)      11. lo' = cvt(rem)
            for i, (ps, k) in enumerate(chains):
                e(f"v_cvt_pk_bf16_f32 {v(ob + (ps - P[0]) * 4 + k)}, {v(T2(i))}, {v(T2(i) + 1)}")
            for i, (ps, k) in enumerate(chains):
                hp = ob + (ps - P[0]) * 4 + k
                e(f"v_lshlrev_b32 {v(T2(i))}, 16, {v(hp)}")
                e(f"v_and_b32 {v(T2(i) + 1)}, 0xffff0000, {v(hp)}")
            for i, (ps, k) in enumerate(chains):
                e(f"v_pk_fma_f32 {v(T1(i), 2)}, {v(T1(i), 2)}, {v(E_RM + st * 8 + ps * 2, 2)}, {v(T2(i), 2)} op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]")
            for i, (ps, k) in enumerate(chains):
                e(f"v_cvt_pk_bf16_f32 {v(ob + 8 + (ps - P[0]) * 4 + k)}, {v(T1(i))}, {v(T1(i) + 1)}")
            for ps in P:
                rofs = u * 32 + ps * 8
                e(f"s_mul_i32 {s(ES0)}, {eop('ldob')}, {rofs}")
                e(f"buffer_store_dwordx4 {v(ob + (ps - P[0]) * 4, 4)}, {v(E_VOB)}, {eop('rob')}, {s(ES0)} offen")
                e(f"buffer_store_dwordx4 {v(ob + 8 + (ps - P[0]) * 4, 4)}, {v(E_VOB)}, {eop('rol')}, {s(ES0)} offen")
                vm_ops.extend([("st", u)] * 2)
    e("s_nop 1")
    clob = [f"v{i}" for i in range(E0, E_END)] + [f"s{i}" for i in range(ES0, ES0 + 8)] + ["scc", "memory"]
    return out, clob


def render(h, ablate=()):
    g = Gen(h, ablate)
    lines = g.build()
    body = "\n".join(f'    "{ln}\\n\\t"' for ln in lines)
    clob = ", ".join(f'"{c}"' for c in g.clobbers())
    return lines, body, clob


def main():
    out = ["// GENERATED by tools/gen_rc4_asm.py -- do not edit.  The main loop of efts_resconv5's one-wave-per-SIMD kernel as one inline-asm",
           "// statement per tile height (see the generator's docstring for the schedule and the register map).",
           "// Operands (all \"s\"): " + ", ".join(f"%{i} {n}" for i, n in enumerate(OPS)),
           f"#define RC4_V0 {V0}", f"#define RC4_VEND {VEND}", ""]
    stats = []
    for h in range(2, 9):
        lines, body, clob = render(h)
        n_mfma = sum(1 for ln in lines if ln.startswith("v_mfma"))
        stats.append((h, len(lines), n_mfma))
        out.append(f"#define RC4_LOOP_H{h}(ARS, A1RS, WRS, W1RS, LDA, RMAX, LDA1, RMAX1, LDW, WTS, WTS1, NCH, WAVE, STATE, LDS0) \\")
        out.append("  asm volatile( \\")
        out.append(body.replace("\n", " \\\n") + " \\")
        out.append("    : : \"s\"(ARS), \"s\"(A1RS), \"s\"(WRS), \"s\"(W1RS), \"s\"(LDA), \"s\"(RMAX), \"s\"(LDA1), \"s\"(RMAX1), \"s\"(LDW), \"s\"(WTS), \"s\"(WTS1), \"s\"(NCH), \"s\"(WAVE), \"s\"(STATE), \"s\"(LDS0) \\")
        out.append(f"    : {clob})")
        out.append("")
    # the epilogue's staging statements: unit i (row block i, both column blocks) of the accumulators -> 8 ds_write_b128 straight from the
    # accumulator file (the MFMAs are issued with the WEIGHT fragment as the A operand, so a lane holds, for its time row, 4 consecutive
    # channels per register quad: block (i, j) registers 4 g .. 4 g + 3 = channels 32 j + 8 g + 4 (lane >> 5) + 0..3)
    for i in range(8):
        body = " \\\n".join(f'    "ds_write_b128 %{j * 4 + g}, a[{(2 * i + j) * 16 + 4 * g}:{(2 * i + j) * 16 + 4 * g + 3}]\\n\\t"' for j in range(2) for g in range(4))
        out.append(f"#define RC4_DUMP_UNIT{i}(W0, W1, W2, W3, W4, W5, W6, W7) \\")
        out.append("  asm volatile( \\")
        out.append(body + " \\")
        out.append('    : : "v"(W0), "v"(W1), "v"(W2), "v"(W3), "v"(W4), "v"(W5), "v"(W6), "v"(W7) : "memory")')
        out.append("")
    # the FAST epilogue, one statement per tile height.  Operands (all "s"): " + ", ".join(f"%{i} {n}" for i, n in enumerate(EOPS)) + "
    for h in range(2, 9):
        lines, clob = gen_epilogue(h)
        body = " \\\n".join(f'    "{ln}\\n\\t"' for ln in lines)
        out.append(f"#define RC4_EPI_H{h}(RA, RAL, RM, ROB, ROL, RBIAS, LDA, LDOB, SLOPE2, N0, WAVE, LDS0) \\")
        out.append("  asm volatile( \\")
        out.append(body + " \\")
        out.append('    : : "s"(RA), "s"(RAL), "s"(RM), "s"(ROB), "s"(ROL), "s"(RBIAS), "s"(LDA), "s"(LDOB), "s"(SLOPE2), "s"(N0), "s"(WAVE), "s"(LDS0) \\')
        out.append("    : " + ", ".join(f'"{c}"' for c in clob) + ")")
        out.append("")
    # ablation variants of the full-height loop for the micro benchmark (tools/micro/rc4_loop_test.hip)
    # NOSTAGING: no operand loads / LDS writes; WRITES_ONLY / LOADS_ONLY: one half of the staging; BUNCHED: the first schedule (all eight
    # ds_writes in k-slice 0, all eight loads in k-slice 1), kept to show what spreading them bought
    for name, ab in (("NOSTAGING", ("loads", "writes")), ("WRITES_ONLY", ("loads",)), ("LOADS_ONLY", ("writes",)), ("BUNCHED", ("bunch",))):
        lines, body, clob = render(8, ab)
        out.append(f"#define RC4_LOOP_H8_{name}(ARS, A1RS, WRS, W1RS, LDA, RMAX, LDA1, RMAX1, LDW, WTS, WTS1, NCH, WAVE, STATE, LDS0) \\")
        out.append("  asm volatile( \\")
        out.append(body.replace("\n", " \\\n") + " \\")
        out.append("    : : \"s\"(ARS), \"s\"(A1RS), \"s\"(WRS), \"s\"(W1RS), \"s\"(LDA), \"s\"(RMAX), \"s\"(LDA1), \"s\"(RMAX1), \"s\"(LDW), \"s\"(WTS), \"s\"(WTS1), \"s\"(NCH), \"s\"(WAVE), \"s\"(STATE), \"s\"(LDS0) \\")
        out.append(f"    : {clob})")
        out.append("")
    open(OUT, "w").write("\n".join(out))
    for h, n, m in stats:
        print(f"h={h}: {n} asm lines, {m} MFMAs (3 unrolled chunk bodies x 5 steps x {8 * h})")


if __name__ == "__main__":
    main()
