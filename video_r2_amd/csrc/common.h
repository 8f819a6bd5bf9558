// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the GRPO rollout path.
// gfx950 only: no CUDA / dual-platform paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define VR2_WAVE 64

// error plumbing (api.cpp)
extern "C" const char* vr2_last_error(void);
void vr2_set_error(const char* fmt, ...);
int vr2_check_launch(const char* what);

// tuning knobs (api.hip: vr2_tune_set / vr2_tune_get); read on the host at launch time only
enum { VR2_TUNE_GEMV_VARIANT = 0, VR2_TUNE_GEMV_ABLATE = 1, VR2_TUNE_GEMV_XCHUNK = 2, VR2_TUNE_ATTN_SUFFIX_DEPTH = 3,
       VR2_TUNE_ATTN_PREFIX_DEPTH = 4, VR2_TUNE_COUNT = 8 };
int vr2_tune_get(int key);

// attention_fwd.hip: the head_dim-128 forward behind vr2_attn_tok
int vr2_launch_attn_fwd128(const void* work_i32x8, int64_t n_work, const void* Q, int64_t ldq, const void* KA, const void* VA,
                           const void* KB, const void* VB, int64_t ldk, void* O, int64_t ldo, int64_t n_heads, int64_t group,
                           float scale, void* lse_out, void* stream);

#define VR2_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      vr2_set_error(__VA_ARGS__);         \
      return -1;                          \
    }                                     \
  } while (0)

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x = 256 (4 waves); scratch: >= 4 floats of LDS
__device__ __forceinline__ float block_sum_256(float v, float* scratch) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[w] = v;
  __syncthreads();
  return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}
__device__ __forceinline__ float block_max_256(float v, float* scratch) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[w] = v;
  __syncthreads();
  return fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
}

// Direct-to-LDS loads as inline asm: hipcc's wait-count pass treats the builtin form as an LDS store that every later LDS read
// may alias and puts s_waitcnt vmcnt(0) in front of each ds_read_b64_tr_b16, which would drain the ring at every tile.  Written
// this way the pass does not see them; the kernels wait for them by count (vmcnt) in front of the tile's barrier.
__device__ __forceinline__ void vr2_dma16(const void* g, const void* l) {
  const unsigned la = (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)l;
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(la) : "memory", "m0");
}
__device__ __forceinline__ void vr2_dma4(const void* g, const void* l) {
  const unsigned la = (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)l;
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(la) : "memory", "m0");
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
