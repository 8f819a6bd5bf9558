// HBM-bound row kernels of the rollout path: residual-add + RMSNorm, casts/gathers, SiLU*mul, 2-D vision RoPE,
// M-RoPE tables and application (+ KV-cache append), embedding lookup / row scatter.
// All are one pass over their operands with 8/16-byte vector accesses; fp32 math, bf16 only on the GEMM side.
#include "common.h"

namespace {

// ------------------------------------------------------------------ add + RMSNorm
// x[m][:] (+)= bias + sum_s partial[s][m][:]   (optional)   ;  y = bf16( x * rsqrt(mean(x^2)+eps) * w )
// HF modeling_qwen2_5_vl.py:65-79 computes the norm in fp32; the residual stream x stays fp32 here.
// REG: the row (H <= 4096) is held in registers between the sum-of-squares and the scaled store: one pass over x.
__device__ __forceinline__ float4 row_update(float* xr, int i, const float* partial, long split_stride, long prow, int nsplit,
                                              const float* bias) {
  float4 v = *reinterpret_cast<const float4*>(xr + i);
  if (partial) {
    for (int s = 0; s < nsplit; ++s) {
      const float4 p = *reinterpret_cast<const float4*>(partial + (long)s * split_stride + prow + i);
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    if (bias) {
      const float4 b = *reinterpret_cast<const float4*>(bias + i);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    *reinterpret_cast<float4*>(xr + i) = v;
  }
  return v;
}

// SB = split-K slabs read per batch of loads (4 column groups each): 4 for the general kernel; 8 for the decode rows behind an
// 8-part projection (down): all 32 slab loads of a thread are then in flight at once -- one memory round trip instead of two for
// a launch that is a latency chain -- at 128 more registers, which the M-row training launches should not pay (own instantiation)
template <bool REG, int SB = 4>
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(float* __restrict__ x, long ldx, const float* __restrict__ partial,
                                                          long split_stride, long ldp, int nsplit,
                                                          const float* __restrict__ bias, const float* __restrict__ w,
                                                          bf16* __restrict__ y, long ldy, int H, float eps,
                                                          unsigned char* __restrict__ q8, long ldq, float* __restrict__ qscale) {
  __shared__ float scratch[4];
  const int m = blockIdx.x;
  float* xr = x + (long)m * ldx;
  const long prow = (long)m * ldp;
  float ss = 0.f;
  if constexpr (REG) {
    float4 keep[4], gw[4];
    bool on[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {   // norm weights first: their latency hides under the row reduction
      const int i = threadIdx.x * 4 + it * 1024;
      on[it] = i < H;
      gw[it] = (y && on[it]) ? *reinterpret_cast<const float4*>(w + i) : make_float4(0.f, 0.f, 0.f, 0.f);
      keep[it] = on[it] ? *reinterpret_cast<const float4*>(xr + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (partial) {
      // split-K slabs: 4 slabs x 4 column groups = 16 independent 16-byte loads in flight per thread (at 64 decode rows and
      // 16 slabs a one-load-at-a-time loop was 15 us per call); summed in slab order, so the result does not depend on timing
      for (int s0 = 0; s0 < nsplit; s0 += SB) {
        float4 pv[SB][4];
#pragma unroll
        for (int j = 0; j < SB; ++j)
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int i = threadIdx.x * 4 + it * 1024;
            pv[j][it] = (on[it] && s0 + j < nsplit) ? *reinterpret_cast<const float4*>(partial + (long)(s0 + j) * split_stride + prow + i)
                                                    : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
        for (int j = 0; j < SB; ++j)
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            keep[it].x += pv[j][it].x; keep[it].y += pv[j][it].y; keep[it].z += pv[j][it].z; keep[it].w += pv[j][it].w;
          }
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int i = threadIdx.x * 4 + it * 1024;
        if (!on[it]) continue;
        if (bias) {
          const float4 b = *reinterpret_cast<const float4*>(bias + i);
          keep[it].x += b.x; keep[it].y += b.y; keep[it].z += b.z; keep[it].w += b.w;
        }
        *reinterpret_cast<float4*>(xr + i) = keep[it];
      }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it)
      ss += keep[it].x * keep[it].x + keep[it].y * keep[it].y + keep[it].z * keep[it].z + keep[it].w * keep[it].w;
    ss = block_sum_256(ss, scratch);
    if (!y) return;
    const float inv = rsqrtf(ss / (float)H + eps);
    bf16* yr = y + (long)m * ldy;
    float am = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = threadIdx.x * 4 + it * 1024;
      if (i < H) {
        const float4 v = keep[it];
        const float4 g = gw[it];
        bf16x4 o = {f2bf(v.x * inv * g.x), f2bf(v.y * inv * g.y), f2bf(v.z * inv * g.z), f2bf(v.w * inv * g.w)};
        *reinterpret_cast<bf16x4*>(yr + i) = o;
        if (q8) {   // the quantiser sees the bf16 values the bf16 path would hand to the GEMV
          keep[it] = make_float4(bf2f(o[0]), bf2f(o[1]), bf2f(o[2]), bf2f(o[3]));
          am = fmaxf(fmaxf(am, fmaxf(fabsf(keep[it].x), fabsf(keep[it].y))), fmaxf(fabsf(keep[it].z), fabsf(keep[it].w)));
        }
      }
    }
    if (q8) {
      // fused vr2_quant_rows_fp8 (gemv_w8.hip): e4m3 bytes in the two-k-step fragment order + the row's scale.  H % 64 == 0 (host)
      am = block_max_256(am, scratch);
      const float qi = am > 0.f ? 448.0f / am : 1.0f;
      if (threadIdx.x == 0) qscale[m] = am > 0.f ? am / 448.0f : 1.0f;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int i = threadIdx.x * 4 + it * 1024;
        if (i < H) {
          int d = 0;
          d = __builtin_amdgcn_cvt_pk_fp8_f32(keep[it].x * qi, keep[it].y * qi, d, false);
          d = __builtin_amdgcn_cvt_pk_fp8_f32(keep[it].z * qi, keep[it].w * qi, d, true);
          const int kk = i & 63;                                         // byte of the row: block, slot (kk % 32) / 8, half kk / 32
          *reinterpret_cast<int*>(q8 + (long)m * ldq + (i & ~63) + ((kk & 31) >> 3) * 16 + (kk >> 5) * 8 + (kk & 7)) = d;
        }
      }
    }
  } else {
    for (int i = threadIdx.x * 4; i < H; i += 1024) {
      const float4 v = row_update(xr, i, partial, split_stride, prow, nsplit, bias);
      ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss = block_sum_256(ss, scratch);
    if (!y) return;
    const float inv = rsqrtf(ss / (float)H + eps);
    bf16* yr = y + (long)m * ldy;
    for (int i = threadIdx.x * 4; i < H; i += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(xr + i);
      const float4 g = *reinterpret_cast<const float4*>(w + i);
      bf16x4 o = {f2bf(v.x * inv * g.x), f2bf(v.y * inv * g.y), f2bf(v.z * inv * g.z), f2bf(v.w * inv * g.w)};
      *reinterpret_cast<bf16x4*>(yr + i) = o;
    }
  }
}

// ------------------------------------------------------------------ fp32 rows -> bf16 rows (optional gather, zero pad)
__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ src, long lds_, const int* __restrict__ row_idx,
                                                        bf16* __restrict__ dst, long ldd, int K, int Kpad) {
  const int m = blockIdx.x;
  const long sr = row_idx ? row_idx[m] : m;
  const float* s = src + sr * lds_;
  bf16* d = dst + (long)m * ldd;
  for (int i = threadIdx.x * 4; i < Kpad; i += 1024) {
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = f2bf((i + j) < K ? s[i + j] : 0.f);
    *reinterpret_cast<bf16x4*>(d + i) = o;
  }
}

// ------------------------------------------------------------------ antialiased bicubic resize of uint8 frames, one axis
// out[r][i] = clamp((sum_j in[r][first[i] + j] * w[i][j] + 2^(bits-1)) >> bits, 0, 255) along the contiguous axis
// (stride 1) or along rows (stride = row length): the integer arithmetic of torch's uint8 antialias kernel (the op behind
// the reference's torchvision resize, vision_process.py:802-807); the tap tables come from host/frames.py resize_taps.
// Bit-exact by construction (int32 accumulation of uint8 x int16 products).
__global__ __launch_bounds__(256) void resize_axis_u8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                             const int* __restrict__ first, const int* __restrict__ count,
                                                             const short* __restrict__ wt, int kmax, int bits, long planes,
                                                             int in_rows, int in_cols, int out_rows, int out_cols, int vertical) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = planes * out_rows * out_cols;
  if (idx >= total) return;
  const int ox = (int)(idx % out_cols);
  const int oy = (int)((idx / out_cols) % out_rows);
  const long pl = idx / ((long)out_cols * out_rows);
  const unsigned char* sp = src + pl * in_rows * in_cols;
  const int i = vertical ? oy : ox;
  const int x0 = first[i], n = count[i];
  const short* w = wt + (long)i * kmax;
  int acc = 1 << (bits - 1);
  if (vertical) {
    for (int j = 0; j < n; ++j) acc += (int)sp[(long)(x0 + j) * in_cols + ox] * (int)w[j];
  } else {
    const unsigned char* row = sp + (long)oy * in_cols + x0;
    for (int j = 0; j < n; ++j) acc += (int)row[j] * (int)w[j];
  }
  acc >>= bits;
  dst[idx] = (unsigned char)(acc < 0 ? 0 : acc > 255 ? 255 : acc);
}

// ------------------------------------------------------------------ uint8 frames -> normalised, patchified bf16 GEMM rows
// One workgroup per OUTPUT row m of the patch-embed GEMM input: source patch p = row_idx[m] in the processor's row order
// (t, h/merge, w/merge, merge_h, merge_w); columns (channel, temporal, ph, pw).  (x - mean[c]) / std[c] in fp32 with the
// already rescaled constants (HF fuses rescale into normalise the same way), ONE bf16 rounding; a clip with an odd frame
// count repeats its last frame (HF video processor padding).  Replaces the host normalise + permute + H2D + cast.
__global__ __launch_bounds__(256) void patchify_u8_kernel(const unsigned char* __restrict__ frames, int n_frames, int C, int Himg,
                                                          int Wimg, const int* __restrict__ row_idx, float3 mean, float3 stdv,
                                                          bf16* __restrict__ dst, long ldd, int ps, int tp, int merge, int K,
                                                          int Kpad) {
  const int m = blockIdx.x;
  const int p = row_idx ? row_idx[m] : m;
  const int gh = Himg / ps, gw = Wimg / ps;
  const int per_t = gh * gw;
  const int t = p / per_t, r = p % per_t;
  const int unit = merge * merge, wb_n = gw / merge;
  const int blk = r / unit, in = r % unit;
  const int hb = blk / wb_n, wb = blk % wb_n, mh = in / merge, mw = in % merge;
  const int y0 = (hb * merge + mh) * ps, x0 = (wb * merge + mw) * ps;
  bf16* d = dst + (long)m * ldd;
  for (int k = threadIdx.x; k < Kpad; k += 256) {
    float v = 0.f;
    if (k < K) {
      const int pw = k % ps, ph = (k / ps) % ps, ti = (k / (ps * ps)) % tp, c = k / (ps * ps * tp);
      int f = t * tp + ti; if (f > n_frames - 1) f = n_frames - 1;
      const float px = (float)frames[(((long)f * C + c) * Himg + (y0 + ph)) * Wimg + (x0 + pw)];
      const float mu = c == 0 ? mean.x : c == 1 ? mean.y : mean.z, sd = c == 0 ? stdv.x : c == 1 ? stdv.y : stdv.z;
      v = (px - mu) / sd;
    }
    d[k] = f2bf(v);
  }
}

// ------------------------------------------------------------------ rotary tables
// vision: pos [P][2] (h,w) int32, inv_freq [hd/4] -> cos,sin [P][hd/2]   (HF :125-134, :441-446)
__global__ void vision_rope_table_kernel(const int* __restrict__ pos, const float* __restrict__ inv_freq, float* __restrict__ cs,
                                         float* __restrict__ sn, int P, int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P * half) return;
  const int p = i / half, j = i % half;
  const int q = half / 2;
  const float ang = (float)pos[p * 2 + (j >= q ? 1 : 0)] * inv_freq[j % q];
  cs[i] = cosf(ang);
  sn[i] = sinf(ang);
}

// M-RoPE: pos [3][T] int32, inv_freq [hd/2], section id per frequency sec[hd/2] in {0,1,2} -> cos,sin [T][hd/2]
// (HF :525-538 + :579-585: frequency j takes the position of axis sec[j]; both halves of the head share it)
__global__ void mrope_table_kernel(const int* __restrict__ pos, long T, const float* __restrict__ inv_freq, const int* __restrict__ sec,
                                   float* __restrict__ cs, float* __restrict__ sn, int half) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * half) return;
  const long t = i / half;
  const int j = (int)(i % half);
  const float ang = (float)pos[(long)sec[j] * T + t] * inv_freq[j];
  cs[i] = cosf(ang);
  sn[i] = sinf(ang);
}

// ------------------------------------------------------------------ rotate-half RoPE on packed QKV rows
// qkv [T][ld] FP32 (straight from the QKV GEMM, so q/k are rounded to bf16 exactly once, after the rotation):
// q heads -> q_out rows t ; k heads (roped) -> k_out rows kv_row0 + t ; v heads -> v_out rows kv_row0 + t.
// x' = x*cos + rotate_half(x)*sin in fp32.  One block per token.  k_out / v_out are KV-cache rows for the LM, or the
// k / v column blocks of a packed bf16 [q|k|v] buffer for the ViT.
__global__ __launch_bounds__(256) void rope_qkv_kernel(const float* __restrict__ qkv, long ld, const float* __restrict__ cs,
                                                       const float* __restrict__ sn, int nq, int nkv, int hd,
                                                       bf16* __restrict__ q_out, long ldq, bf16* __restrict__ kdst,
                                                       bf16* __restrict__ vdst, long ldkv, long kv_row0, int seg_rows,
                                                       long seg_stride) {
  const int t = blockIdx.x;
  // K/V destination row: t, or -- batched prefill of several prompts into per-prompt cache slots -- row t % seg_rows of
  // the slot t / seg_rows, slots seg_stride rows apart
  const long drow = kv_row0 + (seg_rows > 0 ? (long)(t / seg_rows) * seg_stride + (t % seg_rows) : (long)t);
  const int half = hd / 2;
  const float* row = qkv + (long)t * ld;
  const float* c = cs + (long)t * half;
  const float* s = sn + (long)t * half;
  const int nrot = nq + nkv;
  for (int i = threadIdx.x; i < nrot * half; i += 256) {
    const int h = i / half, j = i % half;
    const float* p = row + h * hd;
    const float x1 = p[j], x2 = p[j + half];
    const float o1 = x1 * c[j] - x2 * s[j];
    const float o2 = x2 * c[j] + x1 * s[j];
    bf16* d = (h < nq) ? (q_out + (long)t * ldq + h * hd) : (kdst + drow * ldkv + (h - nq) * hd);
    d[j] = f2bf(o1);
    d[j + half] = f2bf(o2);
  }
  const float* v = row + (long)(nq + nkv) * hd;
  bf16* vd = vdst + drow * ldkv;
  for (int i = threadIdx.x * 4; i < nkv * hd; i += 1024) {
    const float4 x = *reinterpret_cast<const float4*>(v + i);
    bf16x4 o = {f2bf(x.x), f2bf(x.y), f2bf(x.z), f2bf(x.w)};
    *reinterpret_cast<bf16x4*>(vd + i) = o;
  }
}

// decode: qkv = bias + sum_s partial[s][g][:] (fp32, from the split-K GEMV) ; position p = pos[g] + step (1-D: t=h=w)
// q -> q_out [G][nq*hd] bf16 (roped) ; k (roped), v -> cache row (g*rows_per_seq + step[0]).  float4 accesses.
__global__ __launch_bounds__(256) void rope_append_decode_kernel(const float* __restrict__ partial, long split_stride, long ldp,
                                                                 int nsplit, const float* __restrict__ bias,
                                                                 const int* __restrict__ pos, const int* __restrict__ step,
                                                                 const float* __restrict__ inv_freq, int nq, int nkv, int hd,
                                                                 bf16* __restrict__ q_out, long ldq, bf16* __restrict__ kc,
                                                                 bf16* __restrict__ vc, long ldkv, long rows_per_seq) {
  __shared__ float cs_s[128], sn_s[128];
  const int g = blockIdx.x;
  const int half = hd / 2;
  const float p = (float)(pos[g] + step[0]);
  const long crow = (long)g * rows_per_seq + step[0];
  const int nh = nq + 2 * nkv;
  if (threadIdx.x < half) sincosf(p * inv_freq[threadIdx.x], &sn_s[threadIdx.x], &cs_s[threadIdx.x]);
  __syncthreads();
  const int q4 = half / 4;                       // float4 groups per half head
  const float* prow = partial + (long)g * ldp;
  for (int i = threadIdx.x + 256 * blockIdx.y; i < nh * q4; i += 256 * gridDim.y) {   // grid.y spreads one row over CUs
    const int h = i / q4, j = (i % q4) * 4;
    const long c1 = (long)h * hd + j, c2 = c1 + half;
    float4 x1 = bias ? *reinterpret_cast<const float4*>(bias + c1) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 x2 = bias ? *reinterpret_cast<const float4*>(bias + c2) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = 0; s0 < nsplit; s0 += 4) {      // 8 independent loads in flight, summed in slab order
      float4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool ok = s0 + u < nsplit;
        a[u] = ok ? *reinterpret_cast<const float4*>(prow + (long)(s0 + u) * split_stride + c1) : make_float4(0.f, 0.f, 0.f, 0.f);
        b[u] = ok ? *reinterpret_cast<const float4*>(prow + (long)(s0 + u) * split_stride + c2) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        x1.x += a[u].x; x1.y += a[u].y; x1.z += a[u].z; x1.w += a[u].w;
        x2.x += b[u].x; x2.y += b[u].y; x2.z += b[u].z; x2.w += b[u].w;
      }
    }
    bf16x4 o1, o2;
    if (h < nq + nkv) {
      const float a1[4] = {x1.x, x1.y, x1.z, x1.w}, a2[4] = {x2.x, x2.y, x2.z, x2.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float cv = cs_s[j + e], sv = sn_s[j + e];
        o1[e] = f2bf(a1[e] * cv - a2[e] * sv);
        o2[e] = f2bf(a2[e] * cv + a1[e] * sv);
      }
    } else {
      o1 = bf16x4{f2bf(x1.x), f2bf(x1.y), f2bf(x1.z), f2bf(x1.w)};
      o2 = bf16x4{f2bf(x2.x), f2bf(x2.y), f2bf(x2.z), f2bf(x2.w)};
    }
    bf16* d;
    if (h < nq) d = q_out + (long)g * ldq + (long)h * hd;
    else if (h < nq + nkv) d = kc + crow * ldkv + (long)(h - nq) * hd;
    else d = vc + crow * ldkv + (long)(h - nq - nkv) * hd;
    *reinterpret_cast<bf16x4*>(d + j) = o1;
    *reinterpret_cast<bf16x4*>(d + j + half) = o2;
  }
}

// ------------------------------------------------------------------ embedding lookup / row scatter
__global__ __launch_bounds__(256) void embed_rows_kernel(const int* __restrict__ ids, const bf16* __restrict__ table, long ldt,
                                                         float* __restrict__ out, long ldo, int H) {
  const int t = blockIdx.x;
  const bf16* r = table + (long)ids[t] * ldt;
  float* o = out + (long)t * ldo;
  for (int i = threadIdx.x * 4; i < H; i += 1024) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(r + i);
    *reinterpret_cast<float4*>(o + i) = make_float4(bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3]));
  }
}

// dst[idx[i]][:] = src[i][:]  (fp32 rows; masked_scatter of merged video tokens, HF :1206-1223)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, long lds_, const int* __restrict__ idx,
                                                           float* __restrict__ dst, long ldd, int H) {
  const int t = blockIdx.x;
  const float* s = src + (long)t * lds_;
  float* d = dst + (long)idx[t] * ldd;
  for (int i = threadIdx.x * 4; i < H; i += 1024) *reinterpret_cast<float4*>(d + i) = *reinterpret_cast<const float4*>(s + i);
}

__global__ void add_int_kernel(int* p, int v) {
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += v;
}

}  // namespace

#define STREAM ((hipStream_t)stream)

extern "C" int vr2_add_rmsnorm(void* x_f32, int64_t ldx, const void* partial_f32, int64_t split_stride, int64_t ldp,
                               int nsplit, const void* bias_f32, const void* w_f32, void* y_bf16, int64_t ldy, int64_t M,
                               int64_t H, float eps, void* stream) {
  VR2_REQUIRE(M >= 1 && H >= 4 && H % 4 == 0, "vr2_add_rmsnorm: bad shape M=%ld H=%ld", (long)M, (long)H);
  VR2_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ldp % 4 == 0 && split_stride % 4 == 0, "vr2_add_rmsnorm: strides must be multiples of 4");
  VR2_REQUIRE(y_bf16 == nullptr || w_f32 != nullptr, "vr2_add_rmsnorm: norm weight missing");
  if (H <= 4096 && partial_f32 != nullptr && nsplit > 4 && M <= 256)
    hipLaunchKernelGGL((add_rmsnorm_kernel<true, 8>), dim3((unsigned)M), dim3(256), 0, STREAM, (float*)x_f32, (long)ldx,
                       (const float*)partial_f32, (long)split_stride, (long)ldp, nsplit, (const float*)bias_f32,
                       (const float*)w_f32, (bf16*)y_bf16, (long)ldy, (int)H, eps, (unsigned char*)nullptr, 0L, (float*)nullptr);
  else if (H <= 4096)
    hipLaunchKernelGGL(add_rmsnorm_kernel<true>, dim3((unsigned)M), dim3(256), 0, STREAM, (float*)x_f32, (long)ldx,
                       (const float*)partial_f32, (long)split_stride, (long)ldp, nsplit, (const float*)bias_f32,
                       (const float*)w_f32, (bf16*)y_bf16, (long)ldy, (int)H, eps, (unsigned char*)nullptr, 0L, (float*)nullptr);
  else
    hipLaunchKernelGGL(add_rmsnorm_kernel<false>, dim3((unsigned)M), dim3(256), 0, STREAM, (float*)x_f32, (long)ldx,
                       (const float*)partial_f32, (long)split_stride, (long)ldp, nsplit, (const float*)bias_f32,
                       (const float*)w_f32, (bf16*)y_bf16, (long)ldy, (int)H, eps, (unsigned char*)nullptr, 0L, (float*)nullptr);
  return vr2_check_launch("vr2_add_rmsnorm");
}

extern "C" int vr2_add_rmsnorm_q8(void* x_f32, int64_t ldx, const void* partial_f32, int64_t split_stride, int64_t ldp, int nsplit,
                                  const void* bias_f32, const void* w_f32, void* y_bf16, int64_t ldy, int64_t M, int64_t H,
                                  float eps, void* q_fp8, int64_t ldq, void* qscale_f32, void* stream) {
  VR2_REQUIRE(M >= 1 && H >= 64 && H % 64 == 0 && H <= 4096, "vr2_add_rmsnorm_q8: H=%ld must be a multiple of 64, <= 4096", (long)H);
  VR2_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ldp % 4 == 0 && split_stride % 4 == 0 && ldq % 16 == 0 && ldq >= H, "vr2_add_rmsnorm_q8: strides");
  VR2_REQUIRE(y_bf16 != nullptr && w_f32 != nullptr && q_fp8 != nullptr && qscale_f32 != nullptr, "vr2_add_rmsnorm_q8: outputs missing");
  hipLaunchKernelGGL(add_rmsnorm_kernel<true>, dim3((unsigned)M), dim3(256), 0, STREAM, (float*)x_f32, (long)ldx,
                     (const float*)partial_f32, (long)split_stride, (long)ldp, nsplit, (const float*)bias_f32,
                     (const float*)w_f32, (bf16*)y_bf16, (long)ldy, (int)H, eps, (unsigned char*)q_fp8, (long)ldq, (float*)qscale_f32);
  return vr2_check_launch("vr2_add_rmsnorm_q8");
}

extern "C" int vr2_cast_rows_f32_bf16(const void* src_f32, int64_t lds_, const void* row_idx_i32, void* dst_bf16,
                                      int64_t ldd, int64_t M, int64_t K, int64_t Kpad, void* stream) {
  VR2_REQUIRE(M >= 1 && K >= 1 && Kpad >= K && Kpad % 4 == 0 && ldd >= Kpad && ldd % 4 == 0, "vr2_cast_rows_f32_bf16: bad shape");
  hipLaunchKernelGGL(cast_rows_kernel, dim3((unsigned)M), dim3(256), 0, STREAM, (const float*)src_f32, (long)lds_,
                     (const int*)row_idx_i32, (bf16*)dst_bf16, (long)ldd, (int)K, (int)Kpad);
  return vr2_check_launch("vr2_cast_rows_f32_bf16");
}

extern "C" int vr2_resize_axis_u8(const void* src_u8, void* dst_u8, const void* first_i32, const void* count_i32,
                                  const void* weights_i16, int64_t kmax, int64_t bits, int64_t planes, int64_t in_rows,
                                  int64_t in_cols, int64_t out_rows, int64_t out_cols, int vertical, void* stream) {
  VR2_REQUIRE(planes >= 1 && in_rows >= 1 && in_cols >= 1 && out_rows >= 1 && out_cols >= 1 && kmax >= 1 && bits >= 1 && bits <= 22,
              "vr2_resize_axis_u8: bad shape");
  VR2_REQUIRE(vertical ? in_cols == out_cols : in_rows == out_rows, "vr2_resize_axis_u8: one axis per call");
  const long total = (long)planes * out_rows * out_cols;
  hipLaunchKernelGGL(resize_axis_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, STREAM, (const unsigned char*)src_u8,
                     (unsigned char*)dst_u8, (const int*)first_i32, (const int*)count_i32, (const short*)weights_i16, (int)kmax,
                     (int)bits, (long)planes, (int)in_rows, (int)in_cols, (int)out_rows, (int)out_cols, vertical);
  return vr2_check_launch("vr2_resize_axis_u8");
}

extern "C" int vr2_patchify_u8(const void* frames_u8, int64_t n_frames, int64_t H, int64_t W, const void* row_idx_i32,
                               const float* mean3, const float* std3, void* dst_bf16, int64_t ldd, int64_t P,
                               int64_t patch_size, int64_t temporal_patch_size, int64_t merge_size, int64_t Kpad, void* stream) {
  const int64_t K = 3 * temporal_patch_size * patch_size * patch_size;
  VR2_REQUIRE(n_frames >= 1 && P >= 1 && H % (patch_size * merge_size) == 0 && W % (patch_size * merge_size) == 0,
              "vr2_patchify_u8: H=%ld W=%ld must be multiples of patch_size*merge_size", (long)H, (long)W);
  VR2_REQUIRE(Kpad >= K && ldd >= Kpad && mean3 && std3, "vr2_patchify_u8: bad row length");
  const int64_t t = (n_frames + temporal_patch_size - 1) / temporal_patch_size;
  VR2_REQUIRE(P == t * (H / patch_size) * (W / patch_size), "vr2_patchify_u8: P=%ld does not match the frame grid", (long)P);
  hipLaunchKernelGGL(patchify_u8_kernel, dim3((unsigned)P), dim3(256), 0, STREAM, (const unsigned char*)frames_u8, (int)n_frames, 3,
                     (int)H, (int)W, (const int*)row_idx_i32, make_float3(mean3[0], mean3[1], mean3[2]),
                     make_float3(std3[0], std3[1], std3[2]), (bf16*)dst_bf16, (long)ldd, (int)patch_size,
                     (int)temporal_patch_size, (int)merge_size, (int)K, (int)Kpad);
  return vr2_check_launch("vr2_patchify_u8");
}

extern "C" int vr2_vision_rope_table(const void* pos_hw_i32, const void* inv_freq_f32, void* cos_f32, void* sin_f32,
                                     int64_t P, int64_t head_dim, void* stream) {
  VR2_REQUIRE(P >= 1 && head_dim % 4 == 0, "vr2_vision_rope_table: bad shape");
  const int half = (int)head_dim / 2;
  const long n = P * half;
  hipLaunchKernelGGL(vision_rope_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, STREAM,
                     (const int*)pos_hw_i32, (const float*)inv_freq_f32, (float*)cos_f32, (float*)sin_f32, (int)P, half);
  return vr2_check_launch("vr2_vision_rope_table");
}

extern "C" int vr2_mrope_table(const void* pos3_i32, int64_t T, const void* inv_freq_f32, const void* section_i32,
                               void* cos_f32, void* sin_f32, int64_t head_dim, void* stream) {
  VR2_REQUIRE(T >= 1 && head_dim % 2 == 0, "vr2_mrope_table: bad shape");
  const int half = (int)head_dim / 2;
  const long n = T * half;
  hipLaunchKernelGGL(mrope_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, STREAM, (const int*)pos3_i32,
                     (long)T, (const float*)inv_freq_f32, (const int*)section_i32, (float*)cos_f32, (float*)sin_f32, half);
  return vr2_check_launch("vr2_mrope_table");
}

extern "C" int vr2_rope_qkv(const void* qkv_f32, int64_t ld, const void* cos_f32, const void* sin_f32, int64_t T, int64_t nq,
                            int64_t nkv, int64_t head_dim, void* q_out_bf16, int64_t ldq, void* k_out_bf16, void* v_out_bf16,
                            int64_t ldkv, int64_t kv_row0, void* stream) {
  VR2_REQUIRE(T >= 1 && head_dim % 8 == 0 && ld >= (nq + 2 * nkv) * head_dim && ld % 4 == 0, "vr2_rope_qkv: bad shape");
  VR2_REQUIRE(q_out_bf16 && k_out_bf16 && v_out_bf16 && ldkv % 4 == 0, "vr2_rope_qkv: outputs missing");
  hipLaunchKernelGGL(rope_qkv_kernel, dim3((unsigned)T), dim3(256), 0, STREAM, (const float*)qkv_f32, (long)ld,
                     (const float*)cos_f32, (const float*)sin_f32, (int)nq, (int)nkv, (int)head_dim, (bf16*)q_out_bf16,
                     (long)ldq, (bf16*)k_out_bf16, (bf16*)v_out_bf16, (long)ldkv, (long)kv_row0, 0, 0L);
  return vr2_check_launch("vr2_rope_qkv");
}

extern "C" int vr2_rope_qkv_seg(const void* qkv_f32, int64_t ld, const void* cos_f32, const void* sin_f32, int64_t T, int64_t nq,
                                int64_t nkv, int64_t head_dim, void* q_out_bf16, int64_t ldq, void* k_out_bf16,
                                void* v_out_bf16, int64_t ldkv, int64_t seg_rows, int64_t seg_stride, void* stream) {
  VR2_REQUIRE(T >= 1 && head_dim % 8 == 0 && ld >= (nq + 2 * nkv) * head_dim && ld % 4 == 0, "vr2_rope_qkv_seg: bad shape");
  VR2_REQUIRE(q_out_bf16 && k_out_bf16 && v_out_bf16 && ldkv % 4 == 0, "vr2_rope_qkv_seg: outputs missing");
  VR2_REQUIRE(seg_rows >= 1 && seg_stride >= seg_rows, "vr2_rope_qkv_seg: seg_rows=%ld seg_stride=%ld", (long)seg_rows, (long)seg_stride);
  hipLaunchKernelGGL(rope_qkv_kernel, dim3((unsigned)T), dim3(256), 0, STREAM, (const float*)qkv_f32, (long)ld,
                     (const float*)cos_f32, (const float*)sin_f32, (int)nq, (int)nkv, (int)head_dim, (bf16*)q_out_bf16,
                     (long)ldq, (bf16*)k_out_bf16, (bf16*)v_out_bf16, (long)ldkv, 0L, (int)seg_rows, (long)seg_stride);
  return vr2_check_launch("vr2_rope_qkv_seg");
}

extern "C" int vr2_rope_append_decode(const void* partial_f32, int64_t split_stride, int64_t ldp, int nsplit,
                                      const void* bias_f32, const void* pos_i32, const void* step_i32,
                                      const void* inv_freq_f32, int64_t G, int64_t nq, int64_t nkv, int64_t head_dim,
                                      void* q_out_bf16, int64_t ldq, void* kcache_bf16, void* vcache_bf16, int64_t ldkv,
                                      int64_t rows_per_seq, void* stream) {
  VR2_REQUIRE(G >= 1 && head_dim % 8 == 0 && head_dim <= 256 && nsplit >= 1 && ldp % 4 == 0 && split_stride % 4 == 0 &&
                  ldq % 4 == 0 && ldkv % 4 == 0, "vr2_rope_append_decode: bad shape");
  const long items = (nq + 2 * nkv) * (head_dim / 8);
  const unsigned gy = (unsigned)(items >= 1024 ? 4 : items >= 512 ? 2 : 1);
  hipLaunchKernelGGL(rope_append_decode_kernel, dim3((unsigned)G, gy), dim3(256), 0, STREAM, (const float*)partial_f32,
                     (long)split_stride, (long)ldp, nsplit, (const float*)bias_f32, (const int*)pos_i32,
                     (const int*)step_i32, (const float*)inv_freq_f32, (int)nq, (int)nkv, (int)head_dim,
                     (bf16*)q_out_bf16, (long)ldq, (bf16*)kcache_bf16, (bf16*)vcache_bf16, (long)ldkv, (long)rows_per_seq);
  return vr2_check_launch("vr2_rope_append_decode");
}

extern "C" int vr2_embed_rows(const void* ids_i32, const void* table_bf16, int64_t ldt, void* out_f32, int64_t ldo,
                              int64_t T, int64_t H, void* stream) {
  VR2_REQUIRE(T >= 1 && H % 4 == 0 && ldt % 4 == 0 && ldo % 4 == 0, "vr2_embed_rows: bad shape");
  hipLaunchKernelGGL(embed_rows_kernel, dim3((unsigned)T), dim3(256), 0, STREAM, (const int*)ids_i32,
                     (const bf16*)table_bf16, (long)ldt, (float*)out_f32, (long)ldo, (int)H);
  return vr2_check_launch("vr2_embed_rows");
}

extern "C" int vr2_scatter_rows_f32(const void* src_f32, int64_t lds_, const void* idx_i32, void* dst_f32, int64_t ldd,
                                    int64_t T, int64_t H, void* stream) {
  VR2_REQUIRE(T >= 1 && H % 4 == 0 && lds_ % 4 == 0 && ldd % 4 == 0, "vr2_scatter_rows_f32: bad shape");
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)T), dim3(256), 0, STREAM, (const float*)src_f32, (long)lds_,
                     (const int*)idx_i32, (float*)dst_f32, (long)ldd, (int)H);
  return vr2_check_launch("vr2_scatter_rows_f32");
}

extern "C" int vr2_add_int(void* p_i32, int v, void* stream) {
  hipLaunchKernelGGL(add_int_kernel, dim3(1), dim3(64), 0, STREAM, (int*)p_i32, v);
  return vr2_check_launch("vr2_add_int");
}
