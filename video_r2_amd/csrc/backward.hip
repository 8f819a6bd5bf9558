// Row-wise backward kernels of the LM update path: SwiGLU, RMSNorm, rotary, log-prob/softmax, embedding
// scatter-add, column sums (bias / norm-weight gradients).  HBM-bound, fp32 math, 8/16-byte vector accesses.
// Autograd semantics of the reference's torch graph (HF modeling_qwen2_5_vl.py forward ops; grpo_trainer.py:723-741 for the
// log-prob head); the engine's training forward saves exactly the tensors these kernels read.
#include "common.h"

namespace {

// ------------------------------------------------------------------ SwiGLU backward on the interleaved [gate tile | up tile] layout
// gu [T][2*I16] bf16 (pre-activations, tile p: cols 32p..32p+15 gate, 32p+16..32p+31 up), dact [T][ldd] bf16 (cols 16p..)
// -> dgu [T][2*I16] bf16 in the same interleaved layout.   act = silu(g) * u
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16* __restrict__ gu, long ldg, const bf16* __restrict__ dact, long ldd,
                                                         bf16* __restrict__ dgu, long ldo, int I16) {
  const int t = blockIdx.y;
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;   // output-column index (0..I16)
  if (i >= I16) return;
  const int p = i >> 4, c = i & 15;
  const bf16x4 g4 = *reinterpret_cast<const bf16x4*>(gu + (long)t * ldg + 32 * p + c);
  const bf16x4 u4 = *reinterpret_cast<const bf16x4*>(gu + (long)t * ldg + 32 * p + 16 + c);
  const bf16x4 d4 = *reinterpret_cast<const bf16x4*>(dact + (long)t * ldd + i);
  bf16x4 og, ou;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float g = bf2f(g4[j]), u = bf2f(u4[j]), d = bf2f(d4[j]);
    const float sig = 1.0f / (1.0f + expf(-g));
    og[j] = f2bf(d * u * sig * (1.0f + g * (1.0f - sig)));
    ou[j] = f2bf(d * g * sig);
  }
  *reinterpret_cast<bf16x4*>(dgu + (long)t * ldo + 32 * p + c) = og;
  *reinterpret_cast<bf16x4*>(dgu + (long)t * ldo + 32 * p + 16 + c) = ou;
}

// act = silu(g) * u from saved interleaved pre-activations (training forward)
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16* __restrict__ gu, long ldg, bf16* __restrict__ act, long lda, int I16) {
  const int t = blockIdx.y;
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= I16) return;
  const int p = i >> 4, c = i & 15;
  const bf16x4 g4 = *reinterpret_cast<const bf16x4*>(gu + (long)t * ldg + 32 * p + c);
  const bf16x4 u4 = *reinterpret_cast<const bf16x4*>(gu + (long)t * ldg + 32 * p + 16 + c);
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float g = bf2f(g4[j]);
    o[j] = f2bf(g / (1.0f + expf(-g)) * bf2f(u4[j]));
  }
  *reinterpret_cast<bf16x4*>(act + (long)t * lda + i) = o;
}

// ------------------------------------------------------------------ GELU (erf form) on saved fp32 pre-activations and its backward
// (the merger's fc1 when the merger trains: the fused GEMM epilogue keeps no pre-activation)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ z, long ldz, bf16* __restrict__ h, long ldh, int N) {
  const int t = blockIdx.y;
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= N) return;
  const float4 v = *reinterpret_cast<const float4*>(z + (long)t * ldz + i);
  bf16x4 o = {f2bf(gelu_erf_f(v.x)), f2bf(gelu_erf_f(v.y)), f2bf(gelu_erf_f(v.z)), f2bf(gelu_erf_f(v.w))};
  *reinterpret_cast<bf16x4*>(h + (long)t * ldh + i) = o;
}

__device__ __forceinline__ float gelu_erf_grad_f(float x) {   // d/dx 0.5 x (1 + erf(x / sqrt 2)) = Phi(x) + x phi(x)
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ z, long ldz, const bf16* __restrict__ dh, long ldd,
                                                       bf16* __restrict__ dz, long ldo, int N) {
  const int t = blockIdx.y;
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= N) return;
  const float4 v = *reinterpret_cast<const float4*>(z + (long)t * ldz + i);
  const bf16x4 d = *reinterpret_cast<const bf16x4*>(dh + (long)t * ldd + i);
  bf16x4 o = {f2bf(bf2f(d[0]) * gelu_erf_grad_f(v.x)), f2bf(bf2f(d[1]) * gelu_erf_grad_f(v.y)),
              f2bf(bf2f(d[2]) * gelu_erf_grad_f(v.z)), f2bf(bf2f(d[3]) * gelu_erf_grad_f(v.w))};
  *reinterpret_cast<bf16x4*>(dz + (long)t * ldo + i) = o;
}

// ------------------------------------------------------------------ dropout on the LoRA branch's input (peft lora.Linear: lora_dropout(x))
// keep(t, k) is a pure function of (seed, salt, t, k) -- a 32-bit mix of the element's coordinates, compared with p * 2^32 -- so the
// backward pass redraws the forward's mask from the same four numbers instead of keeping a T x K mask per adapted layer.
__device__ __forceinline__ unsigned drop_hash(unsigned seed, unsigned salt, unsigned t, unsigned k) {
  unsigned h = seed ^ (salt * 0x9E3779B9u);
  h ^= t + 0x7F4A7C15u + (h << 6) + (h >> 2);
  h *= 0x85EBCA6Bu; h ^= h >> 13;
  h ^= k + 0x165667B1u + (h << 6) + (h >> 2);
  h *= 0xC2B2AE35u; h ^= h >> 16;
  h *= 0x27D4EB2Fu; h ^= h >> 15;
  return h;
}

__global__ __launch_bounds__(256) void dropout_bf16_kernel(const bf16* __restrict__ x, long ldx, bf16* __restrict__ out, long ldo, int K,
                                                           unsigned thresh, float inv_keep, unsigned seed, unsigned salt) {
  const int t = blockIdx.y;
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= K) return;
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(x + (long)t * ldx + i);
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (drop_hash(seed, salt, (unsigned)t, (unsigned)(i + j)) >= thresh) ? f2bf(bf2f(v[j]) * inv_keep) : f2bf(0.f);
  *reinterpret_cast<bf16x4*>(out + (long)t * ldo + i) = o;
}

// dst (+)= keep ? g / (1 - p) : 0  -- the adapter branch's share of the input gradient, through the same mask; dst fp32 or bf16
template <bool DST_BF16>
__global__ __launch_bounds__(256) void dropout_add_kernel(const float* __restrict__ g, long ldg, void* __restrict__ dst, long ldd, int K,
                                                          unsigned thresh, float inv_keep, unsigned seed, unsigned salt) {
  const int t = blockIdx.y;
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= K) return;
  const float4 v = *reinterpret_cast<const float4*>(g + (long)t * ldg + i);
  const float a[4] = {v.x, v.y, v.z, v.w};
  float add[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) add[j] = (drop_hash(seed, salt, (unsigned)t, (unsigned)(i + j)) >= thresh) ? a[j] * inv_keep : 0.f;
  if constexpr (DST_BF16) {
    bf16* d = reinterpret_cast<bf16*>(dst) + (long)t * ldd + i;
    const bf16x4 old = *reinterpret_cast<const bf16x4*>(d);
    bf16x4 o = {f2bf(bf2f(old[0]) + add[0]), f2bf(bf2f(old[1]) + add[1]), f2bf(bf2f(old[2]) + add[2]), f2bf(bf2f(old[3]) + add[3])};
    *reinterpret_cast<bf16x4*>(d) = o;
  } else {
    float* d = reinterpret_cast<float*>(dst) + (long)t * ldd + i;
    float4 old = *reinterpret_cast<const float4*>(d);
    old.x += add[0]; old.y += add[1]; old.z += add[2]; old.w += add[3];
    *reinterpret_cast<float4*>(d) = old;
  }
}

// ------------------------------------------------------------------ RMSNorm backward (row part)
// y = x * inv * w, inv = rsqrt(mean(x^2) + eps).  dx_out[m][:] (+)= inv * (w o dy) - x * inv^3 / H * sum(x o w o dy)
// rinv[m] = inv is also written (the weight-gradient column pass needs it).
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                          const float* __restrict__ dy, long ldy, float* __restrict__ dx, long lddx,
                                                          float* __restrict__ rinv, int H, float eps, int accumulate,
                                                          bf16* __restrict__ dxb, long lddb) {
  __shared__ float scratch[4];
  const int m = blockIdx.x;
  const float* xr = x + (long)m * ldx;
  const float* dyr = dy + (long)m * ldy;
  float ss = 0.f, dot = 0.f;
  for (int i = threadIdx.x * 4; i < H; i += 1024) {
    const float4 xv = *reinterpret_cast<const float4*>(xr + i);
    const float4 gv = *reinterpret_cast<const float4*>(w + i);
    const float4 dv = *reinterpret_cast<const float4*>(dyr + i);
    ss += xv.x * xv.x + xv.y * xv.y + xv.z * xv.z + xv.w * xv.w;
    dot += xv.x * gv.x * dv.x + xv.y * gv.y * dv.y + xv.z * gv.z * dv.z + xv.w * gv.w * dv.w;
  }
  ss = block_sum_256(ss, scratch);
  dot = block_sum_256(dot, scratch);
  const float inv = rsqrtf(ss / (float)H + eps);
  const float coef = inv * inv * inv * dot / (float)H;
  if (rinv && threadIdx.x == 0) rinv[m] = inv;
  float* dxr = dx + (long)m * lddx;
  for (int i = threadIdx.x * 4; i < H; i += 1024) {
    const float4 xv = *reinterpret_cast<const float4*>(xr + i);
    const float4 gv = *reinterpret_cast<const float4*>(w + i);
    const float4 dv = *reinterpret_cast<const float4*>(dyr + i);
    float4 o = make_float4(inv * gv.x * dv.x - xv.x * coef, inv * gv.y * dv.y - xv.y * coef, inv * gv.z * dv.z - xv.z * coef,
                           inv * gv.w * dv.w - xv.w * coef);
    if (accumulate) {
      const float4 old = *reinterpret_cast<const float4*>(dxr + i);
      o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
    }
    *reinterpret_cast<float4*>(dxr + i) = o;
    if (dxb) {
      bf16x4 ob = {f2bf(o.x), f2bf(o.y), f2bf(o.z), f2bf(o.w)};
      *reinterpret_cast<bf16x4*>(dxb + (long)m * lddb + i) = ob;
    }
  }
}

// Row part and weight-gradient part in ONE pass for H <= 4096: a workgroup walks `rows_per_block` rows, keeps each row's x and
// dy in registers between the two block reductions and the dx store (read once instead of twice + once more by the column
// pass), adds dy * x * inv into per-thread column sums, and at the end adds them into dw through LDS so that consecutive lanes
// add consecutive floats (whole 256-byte segments per wave instruction: the shape float atomics run at full rate with).
template <int NG>
__global__ __launch_bounds__(256) void rmsnorm_bwd_dw_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                             const float* __restrict__ dy, long ldy, float* __restrict__ dx, long lddx,
                                                             float* __restrict__ rinv, float* __restrict__ dw, int M, int H, float eps,
                                                             int accumulate, int rows_per_block, bf16* __restrict__ dxb, long lddb) {
  __shared__ float scratch[8];
  __shared__ __attribute__((aligned(16))) float cols[NG * 1024];
  const int tid = threadIdx.x, wave = tid >> 6;
  float4 wv[NG], acc[NG];
  bool ok[NG];
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    const int i = tid * 4 + j * 1024;
    ok[j] = i < H;
    wv[j] = ok[j] ? *reinterpret_cast<const float4*>(w + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
  for (int m = r0; m < r1; ++m) {
    const float* xr = x + (long)m * ldx;
    const float* dyr = dy + (long)m * ldy;
    float4 xv[NG], dv[NG];
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int i = tid * 4 + j * 1024;
      xv[j] = ok[j] ? *reinterpret_cast<const float4*>(xr + i) : make_float4(0.f, 0.f, 0.f, 0.f);
      dv[j] = ok[j] ? *reinterpret_cast<const float4*>(dyr + i) : make_float4(0.f, 0.f, 0.f, 0.f);
      ss += xv[j].x * xv[j].x + xv[j].y * xv[j].y + xv[j].z * xv[j].z + xv[j].w * xv[j].w;
      dot += xv[j].x * wv[j].x * dv[j].x + xv[j].y * wv[j].y * dv[j].y + xv[j].z * wv[j].z * dv[j].z + xv[j].w * wv[j].w * dv[j].w;
    }
    ss = wave_sum(ss);
    dot = wave_sum(dot);
    __syncthreads();
    if ((tid & 63) == 0) { scratch[wave] = ss; scratch[4 + wave] = dot; }
    __syncthreads();
    ss = scratch[0] + scratch[1] + scratch[2] + scratch[3];
    dot = scratch[4] + scratch[5] + scratch[6] + scratch[7];
    const float inv = rsqrtf(ss / (float)H + eps);
    const float coef = inv * inv * inv * dot / (float)H;
    if (rinv && tid == 0) rinv[m] = inv;
    float* dxr = dx + (long)m * lddx;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      if (!ok[j]) continue;
      const int i = tid * 4 + j * 1024;
      float4 o = make_float4(inv * wv[j].x * dv[j].x - xv[j].x * coef, inv * wv[j].y * dv[j].y - xv[j].y * coef,
                             inv * wv[j].z * dv[j].z - xv[j].z * coef, inv * wv[j].w * dv[j].w - xv[j].w * coef);
      if (accumulate) {
        const float4 old = *reinterpret_cast<const float4*>(dxr + i);
        o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
      }
      *reinterpret_cast<float4*>(dxr + i) = o;
      if (dxb) {   // the bf16 copy the next dgrad / wgrad GEMMs read (saves a separate cast pass over dx)
        bf16x4 ob = {f2bf(o.x), f2bf(o.y), f2bf(o.z), f2bf(o.w)};
        *reinterpret_cast<bf16x4*>(dxb + (long)m * lddb + i) = ob;
      }
      acc[j].x += dv[j].x * xv[j].x * inv; acc[j].y += dv[j].y * xv[j].y * inv;
      acc[j].z += dv[j].z * xv[j].z * inv; acc[j].w += dv[j].w * xv[j].w * inv;
    }
  }
#pragma unroll
  for (int j = 0; j < NG; ++j) *reinterpret_cast<float4*>(cols + tid * 4 + j * 1024) = acc[j];
  __syncthreads();
  for (int i = tid; i < H; i += 256) unsafeAtomicAdd(dw + i, cols[i]);
}

// dw[h] += sum_m dy[m][h] * x[m][h] * rinv[m]      (column pass; grid over 256-column blocks x row slabs, fp32 atomics per slab)
__global__ __launch_bounds__(256) void rmsnorm_dw_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ dy, long ldy,
                                                         const float* __restrict__ rinv, float* __restrict__ dw, int M, int H,
                                                         int rows_per_block) {
  const int h = blockIdx.x * 256 + threadIdx.x;
  if (h >= H) return;
  const int r0 = blockIdx.y * rows_per_block;
  int r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
  float acc = 0.f;
  for (int m = r0; m < r1; ++m) acc += dy[(long)m * ldy + h] * x[(long)m * ldx + h] * rinv[m];
  atomicAdd(dw + h, acc);
}

// out[n] += sum_m a[m][n]   (bias gradients; bf16 input)
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16* __restrict__ a, long lda, float* __restrict__ out, int M, int N,
                                                          int rows_per_block) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int r0 = blockIdx.y * rows_per_block;
  int r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
  float acc = 0.f;
  for (int m = r0; m < r1; ++m) acc += bf2f(a[(long)m * lda + n]);
  atomicAdd(out + n, acc);
}

// ------------------------------------------------------------------ rotary backward: d(q,k) rotated by -theta, v passthrough
// dq [T][ldq] bf16, dk/dv [T][ldkv] FP32 (sums of the key-centric work items) -> dqkv [T][ld] bf16 packed [q | k | v heads]
__global__ __launch_bounds__(256) void rope_bwd_kernel(const bf16* __restrict__ dq, long ldq, const float* __restrict__ dk,
                                                       const float* __restrict__ dv, long ldkv, const float* __restrict__ cs,
                                                       const float* __restrict__ sn, int nq, int nkv, int hd,
                                                       bf16* __restrict__ dqkv, long ld) {
  const int t = blockIdx.x;
  const int half = hd / 2;
  const float* c = cs + (long)t * half;
  const float* s = sn + (long)t * half;
  bf16* out = dqkv + (long)t * ld;
  for (int i = threadIdx.x; i < (nq + nkv) * half; i += 256) {
    const int h = i / half, j = i % half;
    float y1, y2;
    if (h < nq) { const bf16* src = dq + (long)t * ldq + h * hd; y1 = bf2f(src[j]); y2 = bf2f(src[j + half]); }
    else { const float* src = dk + (long)t * ldkv + (h - nq) * hd; y1 = src[j]; y2 = src[j + half]; }
    out[h * hd + j] = f2bf(y1 * c[j] + y2 * s[j]);
    out[h * hd + j + half] = f2bf(y2 * c[j] - y1 * s[j]);
  }
  const float* v = dv + (long)t * ldkv;
  bf16* vo = out + (long)(nq + nkv) * hd;
  for (int i = threadIdx.x * 4; i < nkv * hd; i += 1024) {
    const float4 x = *reinterpret_cast<const float4*>(v + i);
    bf16x4 o = {f2bf(x.x), f2bf(x.y), f2bf(x.z), f2bf(x.w)};
    *reinterpret_cast<bf16x4*>(vo + i) = o;
  }
}

// ------------------------------------------------------------------ log-prob head backward
// logp[r] = z[tok]/T - lse(z/T)  =>  dz[r][v] = dlogp[r]/T * (1[v==tok] - exp(z[v]/T - lse[r]))     (bf16 out, zero-padded to ldo)
__global__ __launch_bounds__(256) void dlogits_kernel(const float* __restrict__ logits, long ldl, int V, const int* __restrict__ tok,
                                                      const float* __restrict__ lse, const float* __restrict__ dlogp, float inv_temp,
                                                      bf16* __restrict__ out, long ldo) {
  const int r = blockIdx.y;
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= ldo) return;
  const float gscale = dlogp[r] * inv_temp, l = lse[r];
  const int tk = tok[r];
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int v = i + j;
    float d = 0.f;
    if (v < V) d = gscale * ((v == tk ? 1.0f : 0.0f) - expf(logits[(long)r * ldl + v] * inv_temp - l));
    o[j] = f2bf(d);
  }
  *reinterpret_cast<bf16x4*>(out + (long)r * ldo + i) = o;
}

// ------------------------------------------------------------------ embedding backward: dtable[ids[t]][:] += dx[t][:] for selected rows
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int* __restrict__ ids, const int* __restrict__ skip_id, const float* __restrict__ dx,
                                                        long ldx, float* __restrict__ dtable, long ldt, int H) {
  const int t = blockIdx.x;
  const int id = ids[t];
  if (skip_id && id == skip_id[0]) return;   // video placeholder rows were overwritten by (frozen) vision features
  const float* s = dx + (long)t * ldx;
  float* d = dtable + (long)id * ldt;
  for (int i = threadIdx.x; i < H; i += 256) atomicAdd(d + i, s[i]);
}

}  // namespace

#define STREAM ((hipStream_t)stream)

extern "C" int vr2_swiglu_bwd(const void* gu_bf16, int64_t ldg, const void* dact_bf16, int64_t ldd, void* dgu_bf16, int64_t ldo,
                              int64_t T, int64_t I16, void* stream) {
  VR2_REQUIRE(T >= 1 && I16 % 16 == 0 && ldg % 4 == 0 && ldd % 4 == 0 && ldo % 4 == 0, "vr2_swiglu_bwd: bad shape");
  dim3 grid((unsigned)((I16 / 4 + 255) / 256), (unsigned)T);
  hipLaunchKernelGGL(swiglu_bwd_kernel, grid, dim3(256), 0, STREAM, (const bf16*)gu_bf16, (long)ldg, (const bf16*)dact_bf16,
                     (long)ldd, (bf16*)dgu_bf16, (long)ldo, (int)I16);
  return vr2_check_launch("vr2_swiglu_bwd");
}

extern "C" int vr2_swiglu_fwd(const void* gu_bf16, int64_t ldg, void* act_bf16, int64_t lda, int64_t T, int64_t I16, void* stream) {
  VR2_REQUIRE(T >= 1 && I16 % 16 == 0 && ldg % 4 == 0 && lda % 4 == 0, "vr2_swiglu_fwd: bad shape");
  dim3 grid((unsigned)((I16 / 4 + 255) / 256), (unsigned)T);
  hipLaunchKernelGGL(swiglu_fwd_kernel, grid, dim3(256), 0, STREAM, (const bf16*)gu_bf16, (long)ldg, (bf16*)act_bf16, (long)lda, (int)I16);
  return vr2_check_launch("vr2_swiglu_fwd");
}

extern "C" int vr2_gelu_fwd(const void* z_f32, int64_t ldz, void* h_bf16, int64_t ldh, int64_t T, int64_t N, void* stream) {
  VR2_REQUIRE(T >= 1 && N % 4 == 0 && ldz % 4 == 0 && ldh % 4 == 0 && ldz >= N && ldh >= N, "vr2_gelu_fwd: bad shape");
  dim3 grid((unsigned)((N / 4 + 255) / 256), (unsigned)T);
  hipLaunchKernelGGL(gelu_fwd_kernel, grid, dim3(256), 0, STREAM, (const float*)z_f32, (long)ldz, (bf16*)h_bf16, (long)ldh, (int)N);
  return vr2_check_launch("vr2_gelu_fwd");
}

extern "C" int vr2_gelu_bwd(const void* z_f32, int64_t ldz, const void* dh_bf16, int64_t ldd, void* dz_bf16, int64_t ldo, int64_t T,
                            int64_t N, void* stream) {
  VR2_REQUIRE(T >= 1 && N % 4 == 0 && ldz % 4 == 0 && ldd % 4 == 0 && ldo % 4 == 0 && ldz >= N && ldd >= N && ldo >= N, "vr2_gelu_bwd: bad shape");
  dim3 grid((unsigned)((N / 4 + 255) / 256), (unsigned)T);
  hipLaunchKernelGGL(gelu_bwd_kernel, grid, dim3(256), 0, STREAM, (const float*)z_f32, (long)ldz, (const bf16*)dh_bf16, (long)ldd,
                     (bf16*)dz_bf16, (long)ldo, (int)N);
  return vr2_check_launch("vr2_gelu_bwd");
}

static unsigned drop_threshold(float p) {   // keep iff hash >= p * 2^32
  const double t = (double)p * 4294967296.0;
  return t <= 0.0 ? 0u : (t >= 4294967295.0 ? 4294967295u : (unsigned)t);
}

extern "C" int vr2_dropout_bf16(const void* x_bf16, int64_t ldx, void* out_bf16, int64_t ldo, int64_t T, int64_t K, float p,
                                uint32_t seed, uint32_t salt, void* stream) {
  VR2_REQUIRE(T >= 1 && K % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldx >= K && ldo >= K && p >= 0.f && p < 1.f, "vr2_dropout_bf16: bad shape");
  dim3 grid((unsigned)((K / 4 + 255) / 256), (unsigned)T);
  hipLaunchKernelGGL(dropout_bf16_kernel, grid, dim3(256), 0, STREAM, (const bf16*)x_bf16, (long)ldx, (bf16*)out_bf16, (long)ldo, (int)K,
                     drop_threshold(p), 1.0f / (1.0f - p), (unsigned)seed, (unsigned)salt);
  return vr2_check_launch("vr2_dropout_bf16");
}

extern "C" int vr2_dropout_add(const void* g_f32, int64_t ldg, void* dst, int64_t ldd, int dst_is_bf16, int64_t T, int64_t K, float p,
                               uint32_t seed, uint32_t salt, void* stream) {
  VR2_REQUIRE(T >= 1 && K % 4 == 0 && ldg % 4 == 0 && ldd % 4 == 0 && ldg >= K && ldd >= K && p >= 0.f && p < 1.f, "vr2_dropout_add: bad shape");
  dim3 grid((unsigned)((K / 4 + 255) / 256), (unsigned)T);
  if (dst_is_bf16)
    hipLaunchKernelGGL(dropout_add_kernel<true>, grid, dim3(256), 0, STREAM, (const float*)g_f32, (long)ldg, dst, (long)ldd, (int)K,
                       drop_threshold(p), 1.0f / (1.0f - p), (unsigned)seed, (unsigned)salt);
  else
    hipLaunchKernelGGL(dropout_add_kernel<false>, grid, dim3(256), 0, STREAM, (const float*)g_f32, (long)ldg, dst, (long)ldd, (int)K,
                       drop_threshold(p), 1.0f / (1.0f - p), (unsigned)seed, (unsigned)salt);
  return vr2_check_launch("vr2_dropout_add");
}

extern "C" int vr2_rmsnorm_bwd(const void* x_f32, int64_t ldx, const void* w_f32, const void* dy_f32, int64_t ldy, void* dx_f32,
                               int64_t lddx, void* rinv_f32, void* dw_f32, int64_t M, int64_t H, float eps, int accumulate,
                               void* dx_bf16, int64_t lddb, void* stream) {
  VR2_REQUIRE(M >= 1 && H % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && lddx % 4 == 0, "vr2_rmsnorm_bwd: bad shape");
  VR2_REQUIRE(dx_bf16 == nullptr || (lddb % 4 == 0 && lddb >= H), "vr2_rmsnorm_bwd: bad bf16 output stride");
  VR2_REQUIRE(dw_f32 == nullptr || rinv_f32 != nullptr, "vr2_rmsnorm_bwd: dw needs the rinv scratch");
  if (dw_f32 && H <= 4096) {   // one pass: rows and the weight gradient
    const int rpb = M >= 4096 ? 8 : (M >= 1024 ? 4 : 1);
    const unsigned nb = (unsigned)((M + rpb - 1) / rpb);
    if (H <= 2048)
      hipLaunchKernelGGL(rmsnorm_bwd_dw_kernel<2>, dim3(nb), dim3(256), 0, STREAM, (const float*)x_f32, (long)ldx, (const float*)w_f32,
                         (const float*)dy_f32, (long)ldy, (float*)dx_f32, (long)lddx, (float*)rinv_f32, (float*)dw_f32, (int)M, (int)H, eps,
                         accumulate, rpb, (bf16*)dx_bf16, (long)lddb);
    else
      hipLaunchKernelGGL(rmsnorm_bwd_dw_kernel<4>, dim3(nb), dim3(256), 0, STREAM, (const float*)x_f32, (long)ldx, (const float*)w_f32,
                         (const float*)dy_f32, (long)ldy, (float*)dx_f32, (long)lddx, (float*)rinv_f32, (float*)dw_f32, (int)M, (int)H, eps,
                         accumulate, rpb, (bf16*)dx_bf16, (long)lddb);
    return vr2_check_launch("vr2_rmsnorm_bwd(rows + dw)");
  }
  hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3((unsigned)M), dim3(256), 0, STREAM, (const float*)x_f32, (long)ldx,
                     (const float*)w_f32, (const float*)dy_f32, (long)ldy, (float*)dx_f32, (long)lddx, (float*)rinv_f32, (int)H, eps,
                     accumulate, (bf16*)dx_bf16, (long)lddb);
  if (vr2_check_launch("vr2_rmsnorm_bwd")) return -2;
  if (dw_f32) {
    const int rpb = 256;
    dim3 grid((unsigned)((H + 255) / 256), (unsigned)((M + rpb - 1) / rpb));
    hipLaunchKernelGGL(rmsnorm_dw_kernel, grid, dim3(256), 0, STREAM, (const float*)x_f32, (long)ldx, (const float*)dy_f32, (long)ldy,
                       (const float*)rinv_f32, (float*)dw_f32, (int)M, (int)H, rpb);
    return vr2_check_launch("vr2_rmsnorm_bwd(dw)");
  }
  return 0;
}

extern "C" int vr2_colsum_bf16(const void* a_bf16, int64_t lda, void* out_f32, int64_t M, int64_t N, void* stream) {
  VR2_REQUIRE(M >= 1 && N >= 1, "vr2_colsum_bf16: bad shape");
  const int rpb = 256;
  dim3 grid((unsigned)((N + 255) / 256), (unsigned)((M + rpb - 1) / rpb));
  hipLaunchKernelGGL(colsum_bf16_kernel, grid, dim3(256), 0, STREAM, (const bf16*)a_bf16, (long)lda, (float*)out_f32, (int)M, (int)N, rpb);
  return vr2_check_launch("vr2_colsum_bf16");
}

extern "C" int vr2_rope_bwd(const void* dq_bf16, int64_t ldq, const void* dk_f32, const void* dv_f32, int64_t ldkv,
                            const void* cos_f32, const void* sin_f32, int64_t T, int64_t nq, int64_t nkv, int64_t head_dim,
                            void* dqkv_bf16, int64_t ld, void* stream) {
  VR2_REQUIRE(T >= 1 && head_dim % 8 == 0 && ld >= (nq + 2 * nkv) * head_dim && ldkv % 4 == 0 && ld % 4 == 0, "vr2_rope_bwd: bad shape");
  hipLaunchKernelGGL(rope_bwd_kernel, dim3((unsigned)T), dim3(256), 0, STREAM, (const bf16*)dq_bf16, (long)ldq, (const float*)dk_f32,
                     (const float*)dv_f32, (long)ldkv, (const float*)cos_f32, (const float*)sin_f32, (int)nq, (int)nkv, (int)head_dim,
                     (bf16*)dqkv_bf16, (long)ld);
  return vr2_check_launch("vr2_rope_bwd");
}

extern "C" int vr2_dlogits(const void* logits_f32, int64_t ldl, int64_t R, int64_t V, const void* tok_i32, const void* lse_f32,
                           const void* dlogp_f32, float temperature, void* out_bf16, int64_t ldo, void* stream) {
  VR2_REQUIRE(R >= 1 && V >= 1 && ldo >= V && ldo % 4 == 0 && temperature > 0.f, "vr2_dlogits: bad shape");
  dim3 grid((unsigned)((ldo / 4 + 255) / 256), (unsigned)R);
  hipLaunchKernelGGL(dlogits_kernel, grid, dim3(256), 0, STREAM, (const float*)logits_f32, (long)ldl, (int)V, (const int*)tok_i32,
                     (const float*)lse_f32, (const float*)dlogp_f32, 1.0f / temperature, (bf16*)out_bf16, (long)ldo);
  return vr2_check_launch("vr2_dlogits");
}

extern "C" int vr2_embed_bwd(const void* ids_i32, const void* skip_id_i32, const void* dx_f32, int64_t ldx, void* dtable_f32,
                             int64_t ldt, int64_t T, int64_t H, void* stream) {
  VR2_REQUIRE(T >= 1 && H >= 1, "vr2_embed_bwd: bad shape");
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)T), dim3(256), 0, STREAM, (const int*)ids_i32, (const int*)skip_id_i32,
                     (const float*)dx_f32, (long)ldx, (float*)dtable_f32, (long)ldt, (int)H);
  return vr2_check_launch("vr2_embed_bwd");
}
