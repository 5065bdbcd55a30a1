// Runtime shim for the HRFuser gfx950 kernels.
//
// Product build (hipcc --offload-arch=gfx950): plain HIP, nothing else.
// Test build (-DHRF_EMUL, g++): the SAME kernel sources run on a CPU fiber emulator
// (tests/emul/emul_rt.h) so kernel index math / barriers / shuffles / MFMA fragment layouts can
// be debugged and sanitised (ASan/UBSan) without a GPU.  The emulator is test infrastructure:
// the product package never loads the emulation library.
#pragma once

#ifdef HRF_EMUL
#include "emul_rt.h"
#else
#include <hip/hip_runtime.h>
#define HRF_DYN_SMEM(T, name)                                                   \
  extern __shared__ __attribute__((aligned(16))) unsigned char hrf_dyn_smem_[]; \
  T* name = reinterpret_cast<T*>(hrf_dyn_smem_)
// every launch of the library goes through here
#define HRF_LAUNCH(kern, grid, block, smem, stream, ...) \
  hipLaunchKernelGGL(kern, grid, block, smem, (hipStream_t)(stream), __VA_ARGS__)
// LDS hand-off between lanes of ONE wave (a lane reads what another lane of its wave stored): the hardware executes a
// wave's LDS operations in program order, so only the compiler must be kept from moving code across the point
#define HRF_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
typedef float hrf_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ hrf_f4 hrf_mfma16(float a, float b, hrf_f4 c) {
  // v_mfma_f32_16x16x4_f32: exact fp32 (fmaf chain), A[l&15][l>>4], B[l>>4][l&15],
  // C/D: col = lane&15, row = (lane>>4)*4 + reg.
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// 16-byte global access on 4-byte aligned addresses (gfx950 global_load/store_dwordx4 do not need
// natural alignment; NHWC rows of C = 18, 54, 78 ... floats are only 8-byte aligned)
typedef float hrf_f4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ hrf_f4 hrf_ld4(const float* p) { return *reinterpret_cast<const hrf_f4u*>(p); }
__device__ __forceinline__ void hrf_st4(float* p, hrf_f4 v) { *reinterpret_cast<hrf_f4u*>(p) = v; }
// sum over each 16-lane row (lanes sharing lane>>4) with four DPP adds - no LDS, no ds_bpermute
__device__ __forceinline__ float hrf_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}
__device__ __forceinline__ void hrf_atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void hrf_atomic_add(double* p, double v) { unsafeAtomicAdd(p, v); }
#endif

#define HRF_OK 0
#define HRF_ERR_ARG 1
#define HRF_ERR_LAUNCH 2

static inline int hrf_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

#ifndef HRF_EMUL
static inline int hrf_check_launch() {
  return hipGetLastError() == hipSuccess ? HRF_OK : HRF_ERR_LAUNCH;
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) of a kernel, once per DEVICE and safe under concurrent callers (forwards from two
// Python threads, the autograd thread): `mask` = one bit per device ordinal; racing threads both set the attribute (idempotent)
// before either publishes its bit.  (a plain `static bool once` was neither.)
#include <atomic>
static inline int hrf_dyn_lds_once(std::atomic<unsigned>& mask, const void* fn, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return HRF_ERR_LAUNCH;
  const unsigned bit = 1u << (dev & 31);
  if (mask.load(std::memory_order_acquire) & bit) return HRF_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return HRF_ERR_LAUNCH;
  mask.fetch_or(bit, std::memory_order_release);
  return HRF_OK;
}
#else
static inline int hrf_check_launch() { return HRF_OK; }
#endif
