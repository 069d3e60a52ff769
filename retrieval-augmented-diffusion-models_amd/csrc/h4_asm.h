// Inline-asm vocabulary of the one-wave-per-SIMD kernels (conv_halo4.hip, lin4.hip): literal AGPR accumulators, fragment loads,
// LDS reads / writes and stores as asm volatile statements whose program order hipcc keeps (it only allocates the registers).
#pragma once
#include "common.h"

// The 4 x FN accumulator fragments live in a[64 : 64 + 64 FN) and are OWNED by the asm statements below: they are named literally
// (fragment index IDX -> a[64 + 16 IDX : 64 + 16 IDX + 15]), never bound to a C++ variable -- a "+a" operand made hipcc copy all
// 192 accumulators to VGPRs (and on to scratch) at the loop exit.  hipcc allocates AGPRs of its own from a0 upwards (values that
// only travel between memory instructions, parked VGPRs); a[0:63] are left to it, and the build fails (check_agpr.py) if any
// compiler-generated instruction of these kernels names an AGPR at or above a64.
#define H4_ACC0 64
#define H4_ACC_CLOBBERS "a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127","a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159","a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191","a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207","a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223","a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239","a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255"
#define H4_MFMA(IDX, a, b)  asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(a), "v"(b), "i"(H4_ACC0 + (IDX) * 16), "i"(H4_ACC0 + (IDX) * 16 + 15))
#define H4_MFMA0(IDX, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, 0" :: "v"(a), "v"(b), "i"(H4_ACC0 + (IDX) * 16), "i"(H4_ACC0 + (IDX) * 16 + 15))
#define H4_ACCZERO(IDX, R) asm volatile("v_accvgpr_write_b32 a%c0, 0" :: "i"(H4_ACC0 + (IDX) * 16 + (R)))
#define H4_ACCREAD(dst, IDX, R) asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(dst) : "i"(H4_ACC0 + (IDX) * 16 + (R)))
#define H4_LDSR(dst, addr, off)  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define H4_LDSW(addr, src)  asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(src) : "memory")
#define H4_GLOADB(dst, voff, sbase, imm) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "i"(imm) : "memory")
#define H4_GLOADH(dst, vaddr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(vaddr) : "memory")
// (the loop-carried fragment variables are plain "=v" outputs re-defined in straight-line code -- no branch merges between their
//  definitions and uses: a register copy of a fragment whose load is still in flight would read stale data; the loop is checked
//  for v_mov of fragment registers in the assembly)
// epilogue stores as asm: invisible to hipcc's wait-count model (it guarded the registers of its own stores with vmcnt waits inside
// the hand-counted stream), counted by the first step of the next work item instead of drained
typedef __attribute__((ext_vector_type(4))) unsigned h4_u32x4;
#define H4_GSTORE(vaddr, data) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(vaddr), "v"(data) : "memory")
#define H4_PIN1(a) asm volatile("" : "+v"(a))
#define H4_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define H4_PIN3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
#define H4_GSTOREO(vaddr, data, off) asm volatile("global_store_dwordx4 %0, %1, off offset:%2\n\ts_nop 1" :: "v"(vaddr), "v"(data), "i"(off) : "memory")
#define H4_GSTORES(voff, data, sbase) asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(voff), "v"(data), "s"(sbase) : "memory")
// non-temporal form: a write-once stream bigger than the L2 (the GEGLU hidden tensor: 400 MB per launch) should not evict the operands
#define H4_GSTORES_NT(voff, data, sbase) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" :: "v"(voff), "v"(data), "s"(sbase) : "memory")
#define H4_GSTORESO(voff, data, sbase, off) asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" :: "v"(voff), "v"(data), "s"(sbase), "i"(off) : "memory")
__device__ __forceinline__ unsigned long long h4_uni64(unsigned long long v) {    // uniform value -> SGPR pair
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// the low AGPRs (below H4_ACC0) as landing registers for loads the VGPR file has no room for: 16 bytes into a[AIDX : AIDX + 3]
#define H4_GLOADB_A(AIDX, voff, sbase) asm volatile("global_load_dwordx4 a[%c0:%c1], %2, %3" :: "i"(AIDX), "i"((AIDX) + 3), "v"(voff), "s"(sbase) : "memory")
#define H4_AREAD(dst, AIDX) asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(dst) : "i"(AIDX))
// accumulator group G (4 registers) of fragment IDX straight from the AGPRs to LDS
#define H4_LDSW_ACC(addr, IDX, G, off) asm volatile("ds_write_b128 %0, a[%c1:%c2] offset:%3" :: "v"(addr), "i"(H4_ACC0 + (IDX) * 16 + (G) * 4), "i"(H4_ACC0 + (IDX) * 16 + (G) * 4 + 3), "i"(off) : "memory")
#define H4_ACCWRITE(IDX, R, src) asm volatile("v_accvgpr_write_b32 a%c0, %1" :: "i"(H4_ACC0 + (IDX) * 16 + (R)), "v"(src))
#define H4_LDSWO(addr, src, off)  asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(src), "i"(off) : "memory")
// lanes [32:63] of a <-> lanes [0:31] of b
#define H4_PERMSWAP(a, b) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b))

