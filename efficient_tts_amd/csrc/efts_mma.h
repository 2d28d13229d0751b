// efts_mma.h -- device helpers shared by the MFMA contraction kernels (efts_gemm.hip, efts_resconv.hip): operand tile
// geometry, the swizzled LDS addressing, LDS-DMA issue / counted waits, raw buffer descriptors, bf16 hi/lo packing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "efts_internal.h"

namespace efts {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int BN = 128;                   // output columns per tile
constexpr int WIN = 128;                  // window rows = MFMA rows per tile
constexpr int TILE_BYTES = 128 * 128;     // one operand tile: 128 rows x 128 B
constexpr int NST = 3;                    // weight ring stages
constexpr int GEMM_LDS = 2 * TILE_BYTES + NST * TILE_BYTES;   // 81920 = 160 KiB / 2

// Swizzled LDS byte offset of (row, 16-byte slot) inside a [rows][128 B] tile.  A ds_read_b128
// lane group holds 16 different rows at one logical slot; rows r and r+2 share banks, so the
// physical slot is XORed with (r >> 1) & 7 (conflict-free for every tap shift).
__device__ __forceinline__ int lds_off(int row, int slot) {
    return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
}



// One LDS-DMA piece = one wave instruction = 8 tile rows x 128 B.  Lane l lands at LDS
// m0 + 16*l, i.e. tile row 8*piece + l/8, physical slot l%8; its source is sbase + voff.
__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, const char* sbase) {
    // s_nop 4 first: hipcc does not pad hazards across an asm boundary, and `sbase` (or `voff`) may have been written by a VALU
    // instruction (v_readlane of a spilled SGPR, v_readfirstlane) right before this statement; a VMEM instruction reading
    // an SGPR a VALU just wrote needs 5 wait states (cdna_hip_programming.md 5.7 item 2)
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase)
                 : "memory");
}

__device__ __forceinline__ void wait_vmcnt(int n) {   // n = LDS-DMA pieces allowed to stay in flight (multiple of 4)
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    }
}

// barrier that orders LDS traffic only: global loads / stores in flight stay in flight
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// raw buffer descriptor over `bytes` bytes at p: loads past the end return 0, stores past it are dropped
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, long bytes) {
    const unsigned n = bytes <= 0 ? 0u : (bytes > 0xffffffffL ? 0xffffffffu : (unsigned)bytes);
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, n, 0x00020000);
}
// 16-byte buffer store whose data registers stay untouched for a few more issue slots.  hipcc pads a VALU write to the data
// VGPRs of a dwordx4 store with `s_nop 1`; on gfx950 that was observed NOT to be enough: with the register file full the
// compiler re-used data register 0 for the next store's address (v_add right behind the s_nop), and lanes 12-15 of every
// 16-lane group stored the address instead of the value (tests/test_resconv_gpu.py caught it as 1e-40-sized outputs).
// Naming the data as an input of a trailing asm keeps it live -- and unmodified -- across 5 more wait states.
__device__ __forceinline__ void store_b128(u32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
    asm volatile("s_nop 4" ::"v"(v));
}

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b, float* ra, float* rb) {
    const unsigned h = cvt_pk_bf16(a, b);
    *ra = a - __uint_as_float(h << 16);
    *rb = b - __uint_as_float(h & 0xffff0000u);
    return h;
}

}  // namespace efts
