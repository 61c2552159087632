// The gfx950 instructions the kernels of this library issue BY HAND -- inline asm or builtins hipcc must not schedule, count or allocate its own way -- each
// next to the portable statement the CPU test build (tests/emu: lanes as fibers, LDS-DMA synchronous) compiles in its place.  This file holds the ONE
// NNC_HIP_EMULATOR switch of the kernel sources; the kernels use the names below and carry no conditional of their own.
#pragma once
#include <hip/hip_runtime.h>

namespace nnc {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half_t;
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));
typedef int wf_rsrc_t __attribute__((ext_vector_type(4)));
typedef unsigned int bf16x8_t __attribute__((ext_vector_type(4))); // eight bf16 values, two per dword (element 2i in the low half of dword i)

// Pin a value to its position in the instruction stream: an empty volatile asm that "rewrites" x is ordered against the
// sched_barrier fences, so arithmetic that consumes x cannot be hoisted above the fence in front of it (pure address
// arithmetic otherwise floats to the top of the loop body, in front of the first MFMA).  No instruction is emitted.
// s_waitcnt with only vmcnt counted (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14)
// One MFMA, accumulating in place.  hipcc's builtin cannot be used here: the wave owns 72 accumulator tiles (288 registers)
// while the accumulator file holds 256, and with the builtin hipcc keeps EVERY tile in VGPRs and copies it through a[0:7]
// around each MFMA (measured: 554 v_accvgpr_write + 308 v_accvgpr_read per chunk).  As an asm statement the register
// class is part of the operand: 64 tiles live in AGPRs ("+a"), the last 8 in VGPRs ("+v"), all in place.  What hipcc does not
// do for an asm MFMA (cdna guide 5.7) is handled by construction: its A operand was computed at least a whole slot earlier
// (VALU -> MFMA operand needs 2 wait states), its B operand comes from a ds_read hipcc waits for, and the epilogue's first
// read of an accumulator sits behind explicit s_nops.
// WF_MFMA0: the same with C = 0 -- a work item's first MFMA into each accumulator, so that the 288 registers are not zeroed per
// item.  (Declared read-write all the same: as a pure output it would be a NEW value per item, and hipcc then spills every
// accumulator to scratch around the loop to merge the two -- measured, 1.1 KB of scratch per lane.)
// One LDS-DMA piece: every lane copies 16 bytes from (descriptor base + VOFF + SOFF) to LDS byte address LDS_ADDR + 16 * lane; a
// lane whose offset is out of the descriptor's range writes zeros.  As an asm statement for a different reason than the MFMA:
// hipcc counts a builtin LDS-DMA as a pending LDS write and puts s_waitcnt vmcnt(0) in front of the next ds_read of ANY part
// of the array (measured: one per iteration) -- the pipeline would be synchronous.  Invisible to hipcc, the pieces are
// waited for by the kernel's own counted WF_WAIT_VMCNT.  M0 (the LDS destination base) is written in the statement that
// uses it and restored (cdna guide 5.7).
// Column `i` (= lane & 15) of the [4][16] block of halves whose rows start at blk, blk + pitch, ...: out[j] = blk[j * pitch + i].  On the device this is
// ONE ds_read_b64_tr_b16: each 16-lane group reads the block, lane i supplying the address of four consecutive halves of row i >> 2 and receiving column i.
// NNC_ASM_NOPS("s_nop 4"): wait states hipcc's hazard pass cannot know about (operands of asm MFMAs, descriptor words fresh from v_readfirstlane).
// NNC_WAIT_LGKM0(): the asm LDS reads (tr_read4) are not counted by hipcc either.  NNC_WAIT_VM0_ONLY(): every vector-memory operation so far, nothing else.
#ifdef NNC_HIP_EMULATOR

#define NNC_PIN_V(x) ((void)0)
#define NNC_PIN_S(x) ((void)0)
#define WF_WAIT_VMCNT(n) __builtin_amdgcn_wave_barrier()
#define WF_MFMA(ACC, A, B, IN_AGPR) (ACC) = __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (ACC), 0, 0, 0)
#define WF_MFMA0(ACC, A, B, IN_AGPR) (ACC) = __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (floatx4{ 0.f, 0.f, 0.f, 0.f }), 0, 0, 0)
__device__ __forceinline__ wf_rsrc_t wf_make_rsrc(const void* base, unsigned bytes)
{
	const unsigned long long b = (unsigned long long)base;
	return wf_rsrc_t{ (int)(unsigned)b, (int)(unsigned)(b >> 32), (int)bytes, 0 };
}
__device__ __forceinline__ unsigned wf_lds_addr(float* p) { return (unsigned)(unsigned long long)p; } // emulator: low half of the host pointer; wf_dma16 gets the base again
__device__ __forceinline__ void wf_dma16(const wf_rsrc_t r, float* lds_base, unsigned lds_addr, unsigned voff, unsigned soff)
{
	const unsigned long long b = (unsigned long long)(unsigned)r[0] | (unsigned long long)(unsigned)r[1] << 32;
	float* const dst = (float*)((char*)lds_base + (lds_addr - wf_lds_addr(lds_base)));
	__builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void*)b, 0, (unsigned)r[2], 0), (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}
// 16-byte store through a raw buffer descriptor: per-lane byte offset voff (range-checked against the descriptor: an out-of-range lane stores nothing) plus a
// wave-uniform byte offset soff (NOT part of the range check, as on the device)
template <int POLICY = 0>
static inline void wf_store16(const wf_rsrc_t r, unsigned voff, unsigned soff, float x, float y, float z, float w)
{
	if ((unsigned long long)voff + 16 > (unsigned long long)(unsigned)r[2]) return;
	const unsigned long long b = (unsigned long long)(unsigned)r[0] | (unsigned long long)(unsigned)r[1] << 32;
	float* const p = (float*)(b + voff + soff);
	p[0] = x; p[1] = y; p[2] = z; p[3] = w;
}
static inline float wf_max(float a, float b) { return a > b ? a : b; }
static inline halfx4 tr_read4(const half_t* const blk, const int pitch, const int i) { return halfx4{ blk[i], blk[pitch + i], blk[2 * pitch + i], blk[3 * pitch + i] }; }
static inline floatx16 nnc_mfma_f16(const halfx8 a, const halfx8 b, const floatx16 c) { return emu_mfma_f32_32x32x16_f16(a, b, c); }
// v_mfma_f32_32x32x16_bf16 on packed bf16 operands; nnc_pack_hi16(b, a) = the HIGH halves of a and b in one dword, a's in the low half (v_perm_b32):
// two fp32 values truncated to bf16 and packed (mfma_gemm_bf16x3.h)
static inline floatx16 nnc_mfma_bf16(const bf16x8_t a, const bf16x8_t b, const floatx16 c) { return emu_mfma_f32_32x32x16_bf16(a, b, c); }
static inline unsigned nnc_pack_hi16(const unsigned b, const unsigned a) { return (a >> 16) | (b & 0xffff0000u); }
#define NNC_PIN_VEC(v) ((void)0)
#define NNC_ASM_NOPS(text) ((void)0)
#define NNC_WAIT_LGKM0() ((void)0)
#define NNC_LDS_BARRIER() __syncthreads()
// acc (a pair) += h (a pair) * the LOW / HIGH half of the pair rp: v_pk_fma_f32 with op_sel, so that a per-lane coefficient costs ONE register, not a splat pair
#define NNC_PK_FMA_LO(acc, h, rp) do { (acc)[0] += (h)[0] * (rp)[0]; (acc)[1] += (h)[1] * (rp)[0]; } while (0)
#define NNC_PK_FMA_HI(acc, h, rp) do { (acc)[0] += (h)[0] * (rp)[1]; (acc)[1] += (h)[1] * (rp)[1]; } while (0)
#define NNC_WAIT_VM0_ONLY() ((void)0)
// hardware transcendentals (v_exp_f32: 2^x, v_rcp_f32: 1 / x; both within 1 ulp): the emulator takes the C library's
static inline float nnc_fast_exp2(const float x) { return exp2f(x); }
static inline float nnc_fast_rcp(const float x) { return 1.f / x; }
// -- workgroups of ONE launch that hand each other a few words (cmd_norm.cpp's cluster kernels): agent-scope accesses of the words themselves, no fences.
//    A granule is one naturally aligned 8-byte {tag, value} written by ONE store: the data is the flag.  The emulator keeps a window of workgroups resident
//    and runs them interleaved (tests/emu: launch_concurrent); a polling loop yields to them.
#define NNC_LAUNCH_CONCURRENT(kernel, grid, block, shmem, stream, ...) emuLaunchConcurrentKernel(kernel, grid, block, shmem, stream, __VA_ARGS__)
static inline unsigned nnc_fetch_add_agent(unsigned* p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }
static inline void nnc_store_agent(unsigned* p, unsigned v) { *p = v; }
static inline void nnc_store_granule(unsigned long long* p, unsigned tag, float value) { unsigned u; memcpy(&u, &value, 4); *p = ((unsigned long long)tag << 32) | u; }
static inline unsigned long long nnc_load_granule(const unsigned long long* p) { return *(const volatile unsigned long long*)p; }
#define NNC_SPIN_SLEEP() emu::spin_yield()
// 24 x 24 -> low 32 bits of the product (v_mul_u32_u24 / v_mad_u32_u24: full rate, where the 32-bit v_mul_lo_u32 is quarter rate)
static inline unsigned nnc_mul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }

#else

__device__ __forceinline__ unsigned nnc_mul24(unsigned a, unsigned b) { return __umul24(a, b); }

#define NNC_PIN_V(x) asm volatile("" : "+v"(x))
#define NNC_PIN_S(x) asm volatile("" : "+s"(x))
#define WF_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (15 << 8) | (((n) >> 4) << 14))
#define WF_MFMA0(ACC, A, B, IN_AGPR) do { \
		if (IN_AGPR) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "+a"(ACC) : "v"(A), "v"(B)); \
		else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "+v"(ACC) : "v"(A), "v"(B)); \
	} while (0)
#define WF_MFMA(ACC, A, B, IN_AGPR) do { \
		if (IN_AGPR) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B)); \
		else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B)); \
	} while (0)
__device__ __forceinline__ wf_rsrc_t wf_make_rsrc(const void* base, unsigned bytes)
{ // raw buffer descriptor (stride 0), word 3 = the gfx90a / gfx94x / gfx950 raw-buffer format word; all four words wave-uniform
	const unsigned long long b = (unsigned long long)base;
	return wf_rsrc_t{ __builtin_amdgcn_readfirstlane((int)(unsigned)b), __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32) & 0xffff), __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000 };
}
__device__ __forceinline__ unsigned wf_lds_addr(float* p) { return (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)p; } // LDS byte address
__device__ __forceinline__ void wf_dma16(const wf_rsrc_t r, float*, unsigned lds_addr, unsigned voff, unsigned soff)
{ // lds_addr / soff: wave-uniform integers the caller keeps in SGPRs (plain integer arithmetic on the array's base address -- a
  // pointer cast per piece costs hipcc's null check, four SALU).  M0 is not restored: nothing else in this kernel uses it
  // (checked in the ISA: hipcc's LDS instructions do not read M0 on gfx950, the kernel has no other LDS-DMA and no s_movrel).
	asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_addr), "v"(voff), "s"(r), "s"(soff) : "memory");
}
// 16-byte store through a raw buffer descriptor kept in SGPRs: voff per lane (range-checked: an out-of-range lane stores nothing), soff wave-uniform
// POLICY: 0 the default cache policy, 1 nt (streaming: the line need not stay in the XCD's L2), 2 sc1, 3 sc1 nt, 4 sc0 sc1
template <int POLICY = 0>
__device__ __forceinline__ void wf_store16(const wf_rsrc_t r, unsigned voff, unsigned soff, float x, float y, float z, float w)
{
	typedef float f4 __attribute__((ext_vector_type(4)));
	const f4 v = { x, y, z, w };
	if (POLICY == 1) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt" :: "v"(v), "v"(voff), "s"(r), "s"(soff) : "memory");
	else if (POLICY == 2) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc1" :: "v"(v), "v"(voff), "s"(r), "s"(soff) : "memory");
	else if (POLICY == 3) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc1 nt" :: "v"(v), "v"(voff), "s"(r), "s"(soff) : "memory");
	else if (POLICY == 4) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc0 sc1" :: "v"(v), "v"(voff), "s"(r), "s"(soff) : "memory");
	else asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" :: "v"(v), "v"(voff), "s"(r), "s"(soff) : "memory");
}
// max without the canonicalising v_max x, x hipcc puts in front of fmaxf (the operands here are results of arithmetic: already canonical)
__device__ __forceinline__ float wf_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ halfx4 tr_read4(const half_t* const blk, const int pitch, const int i)
{
	halfx4 v;
	const unsigned addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)(blk + (i >> 2) * pitch + (i & 3) * 4);
	asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
	return v;
}
__device__ __forceinline__ floatx16 nnc_mfma_f16(const halfx8 a, const halfx8 b, const floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ floatx16 nnc_mfma_bf16(const bf16x8_t a, const bf16x8_t b, const floatx16 c)
{
	typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
	return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned nnc_pack_hi16(const unsigned b, const unsigned a) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
#define NNC_LAUNCH_CONCURRENT(kernel, grid, block, shmem, stream, ...) hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__)
// (global_atomic / global_load / global_store with sc1: served by the memory side of the L2s, never by a CU's L1 -- MI355X guide, inter-workgroup visibility)
__device__ __forceinline__ unsigned nnc_fetch_add_agent(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void nnc_store_agent(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void nnc_store_granule(unsigned long long* p, unsigned tag, float value) { __hip_atomic_store(p, ((unsigned long long)tag << 32) | __float_as_uint(value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long nnc_load_granule(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#define NNC_SPIN_SLEEP() __builtin_amdgcn_s_sleep(4)
#define NNC_PIN_VEC(v) asm volatile("" : "+v"(v))
#define NNC_ASM_NOPS(text) asm volatile(text)
#define NNC_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// a workgroup barrier that orders LDS traffic ONLY: __syncthreads() also waits for every global load and store the wave has in flight (vmcnt(0)) -- in a loop whose
// steps hand data through LDS and stream results to HBM that wait is the stores' round trip, twice per step (lstm_rows_*_kernel: 4.5 -> ~1.5 us per step)
#define NNC_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
// acc (a register pair) += h (a pair) * the LOW / HIGH half of the pair rp (hipcc splats a per-lane scalar into a pair of its own for v_pk_fma_f32: 2 x the registers)
#define NNC_PK_FMA_LO(acc, h, rp) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(h), "v"(rp))
#define NNC_PK_FMA_HI(acc, h, rp) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(h), "v"(rp))
#define NNC_WAIT_VM0_ONLY() __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8)) // vmcnt(0), expcnt / lgkmcnt not waited for (gfx9 encoding)
// hardware transcendentals: v_exp_f32 (2^x) and v_rcp_f32 (1 / x), one instruction each, within 1 ulp -- where expf() and an IEEE division are chains of ten and
// more dependent instructions each (the LSTM's per-step latency chain: cmd_lstm.cpp lstm_fast_sigmoid / lstm_fast_tanh)
__device__ __forceinline__ float nnc_fast_exp2(const float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float nnc_fast_rcp(const float x) { return __builtin_amdgcn_rcpf(x); }

#endif

} // namespace nnc
