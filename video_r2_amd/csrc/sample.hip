// Token-level kernels at the lm_head boundary:
//   * vr2_sample_topk  : temperature -> top-k (ties kept, HF logits_process.py:542) -> multinomial, EOS/pad
//                        bookkeeping of HF generation/utils.py:2894-2937.  Exact k-th-largest by 4x8-bit radix
//                        select on order-preserving uint keys; the multinomial draw is a Gumbel-max over the kept
//                        set with a counter-based Philox4x32-10 stream keyed by (seed; token, row, step), so the
//                        result does not depend on thread scheduling and is reproducible per (seed, step, row).
//   * vr2_lse_gather   : per-token log-prob = logit[tok]/T - logsumexp(logits/T)   (grpo_trainer.py:738-739)
//   * vr2_grpo_loss    : the clipped-ratio GRPO loss, all six aggregations, its gradient wrt the log-probs and the
//                        kl / clip metrics (grpo_trainer.py:1099-1177)
#include "common.h"

namespace {

__device__ __forceinline__ unsigned f2key(float f) {
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0, unsigned k1) {
  const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
  const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ __forceinline__ float philox_uniform(unsigned long long seed, unsigned a, unsigned b, unsigned c) {
  unsigned c0 = a, c1 = b, c2 = c, c3 = 0x9E3779B9u;
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return ((float)(c0 >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
}

__global__ __launch_bounds__(1024) void sample_topk_kernel(const float* __restrict__ logits, long ldl, int V, float inv_temp,
                                                           int top_k, unsigned long long seed_arg,
                                                           const unsigned long long* __restrict__ seed_dev, const int* __restrict__ step,
                                                           int* __restrict__ finished, int eos_id, int pad_id,
                                                           int* __restrict__ next_ids, int* __restrict__ tokens, long ldt) {
  __shared__ unsigned hist[256];
  __shared__ unsigned sh_prefix, sh_remaining;
  __shared__ float red_v[16];
  __shared__ int red_i[16];
  const int g = blockIdx.x;
  const float* row = logits + (long)g * ldl;
  const int tid = threadIdx.x;
  const unsigned long long seed = seed_dev ? seed_dev[0] : seed_arg;  // device seed: graph replays stay re-seedable
  unsigned thr_key = 0u;
  if (top_k > 0 && top_k < V) {
    unsigned prefix = 0u, mask = 0u;
    unsigned remaining = (unsigned)top_k;
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      for (int i = tid; i < 256; i += 1024) hist[i] = 0u;
      __syncthreads();
      for (int i = tid; i < V; i += 1024) {
        const unsigned k = f2key(row[i]);
        if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        unsigned cum = 0u;
        int b = 255;
        for (; b > 0; --b) {
          if (cum + hist[b] >= remaining) break;
          cum += hist[b];
        }
        sh_prefix = prefix | ((unsigned)b << shift);
        sh_remaining = remaining - cum;
      }
      __syncthreads();
      prefix = sh_prefix;
      remaining = sh_remaining;
      mask |= 255u << shift;
      __syncthreads();
    }
    thr_key = prefix;  // key of the k-th largest logit
  }
  // Gumbel-max over the kept set {key >= thr_key}
  const unsigned st = (unsigned)step[0];
  float best = -INFINITY;
  int besti = 0x7fffffff;
  for (int i = tid; i < V; i += 1024) {
    const float v = row[i];
    if (f2key(v) >= thr_key) {
      const float u = philox_uniform(seed, (unsigned)i, (unsigned)g, st);
      const float sc = v * inv_temp - __logf(-__logf(u));
      if (sc > best || (sc == best && i < besti)) { best = sc; besti = i; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
  }
  if ((tid & 63) == 0) { red_v[tid >> 6] = best; red_i[tid >> 6] = besti; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w)
      if (red_v[w] > best || (red_v[w] == best && red_i[w] < besti)) { best = red_v[w]; besti = red_i[w]; }
    int tok = besti;
    if (finished) {
      if (finished[g]) tok = pad_id;                 // HF :2928-2929
      else if (tok == eos_id) finished[g] = 1;       // HF :2931-2932
    }
    next_ids[g] = tok;
    if (tokens) tokens[(long)g * ldt + st] = tok;
  }
}

// ---------------------------------------------------------------- two-stage top-k (the decode fast path)
// Stage 1: NSLICE workgroups per row each find the top-k of their vocabulary slice (keys held in registers, 4-pass radix
// select with a parallel suffix scan of the 256-bin histogram) and publish <= CAND candidates.  Stage 2: one workgroup per
// row selects the exact global k-th largest among the NSLICE*CAND candidates and draws the token (Gumbel-max, Philox).
// The global top-k set is a subset of the union of per-slice top-k sets, so the result equals the single-pass kernel's
// unless one slice holds more than CAND values tied at its threshold.
constexpr int NSLICE = 32, CAND = 64, NJ = 24;  // slice length <= 256*NJ

// k-th largest of the keys held by this workgroup (key==0 means "absent").  hist: 256 uints of LDS.
template <int NK>
__device__ __forceinline__ unsigned radix_select_256(const unsigned (&key)[NK], unsigned k, unsigned* hist, unsigned* bc) {
  const int tid = threadIdx.x;
  unsigned prefix = 0u, mask = 0u, remaining = k;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[tid] = 0u;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NK; ++j)
      if (key[j] != 0u && (key[j] & mask) == prefix) atomicAdd(&hist[(key[j] >> shift) & 255u], 1u);
    __syncthreads();
    // suffix sums: incl(b) = sum_{j>=b} hist[j]  (Hillis-Steele on the reversed array)
    const unsigned own = hist[tid];
    unsigned v = own;
    for (int off = 1; off < 256; off <<= 1) {
      const unsigned add = (tid + off < 256) ? hist[tid + off] : 0u;
      __syncthreads();
      v += add;
      hist[tid] = v;
      __syncthreads();
    }
    const unsigned incl = v, excl = v - own;
    if (excl < remaining && incl >= remaining) { bc[0] = prefix | ((unsigned)tid << shift); bc[1] = remaining - excl; }
    __syncthreads();
    prefix = bc[0];
    remaining = bc[1];
    mask |= 255u << shift;
    __syncthreads();
  }
  return prefix;
}

// The slice's k-th largest key without walking the radix passes over all of its keys.  Each thread's largest key is one of 256
// DISTINCT elements of the slice, so the k-th largest of those maxima (k <= 256) is a lower bound T0 of the slice's k-th largest
// key: every top-k key is >= T0, and on anything but degenerate data only a few times k keys are.  Those go to a short LDS list
// (LCAP entries) on which the exact k-th largest is found by counting ranks; a list that overflows (more than LCAP keys >= T0:
// long runs of equal logits) falls back to the 4-pass radix select over the registers.  The kept set {key >= thr} is the same
// either way.  (The radix passes cost 4 x (19 same-address-heavy LDS atomics per thread + a 16-barrier scan): 70 us per token at
// 128 rows; this form: a maximum, 256 + |list| comparisons per thread and 6 barriers.)
constexpr int LCAP_SEL = 768;

// Slot of this thread's element in the list of all elements of the workgroup whose predicate holds, in (round-major, thread-minor)
// order -- the order of the token indices when round j of thread t holds index base + t + 256 j.  Replaces atomicAdd slot
// counters: with those, WHICH elements fall off a full list and the order of the survivors (hence the rounding of sums over them)
// depended on the arrival order of the atomics, i.e. a draw was not reproducible for a fixed seed once a list overflowed or a sum
// ran over it.  `running` = elements listed in earlier rounds (same value in every thread); wc: 4 ints of LDS.  Two barriers.
__device__ __forceinline__ int ordered_slot_256(bool pred, int& running, int* wc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long bal = __ballot(pred);
  const int below = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wc[wave] = __popcll(bal);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) { const int c = wc[w]; off += w < wave ? c : 0; tot += c; }
  const int slot = running + off + below;
  running += tot;
  __syncthreads();
  return slot;
}

__global__ __launch_bounds__(256) void topk_candidates_kernel(const float* __restrict__ logits, long ldl, int V, int top_k,
                                                              float* __restrict__ cand_val, int* __restrict__ cand_idx) {
  __shared__ unsigned hist[256];
  __shared__ unsigned bc[2];
  __shared__ int count;
  __shared__ unsigned lkey[LCAP_SEL];
  __shared__ int lidx[LCAP_SEL];
  __shared__ int lcount;
  const int s = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
  const int L = (V + NSLICE - 1) / NSLICE;
  const int lo = s * L;
  const int n = (lo + L <= V) ? L : (V - lo > 0 ? V - lo : 0);
  const float* row = logits + (long)g * ldl;
  unsigned key[NJ];
  unsigned mx = 0u;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = tid + 256 * j;
    key[j] = (i < n) ? f2key(row[lo + i]) : 0u;   // f2key never returns 0 for a finite float or +-inf
    mx = key[j] > mx ? key[j] : mx;
  }
  float* cv = cand_val + ((long)g * NSLICE + s) * CAND;
  int* ci = cand_idx + ((long)g * NSLICE + s) * CAND;
  if (tid < CAND) { cv[tid] = -INFINITY; ci[tid] = 0x7fffffff; }
  if (tid == 0) { count = 0; lcount = 0; }
  if (n == 0) return;
  const unsigned k = (unsigned)(top_k < n ? top_k : n);
  unsigned thr;
  bool listed = false;
  if (k <= 256u && n >= 256) {
    // T0 = the k-th largest of the 256 thread maxima: the maximum whose rank interval [#greater, #greater-or-equal) holds k - 1
    hist[tid] = mx;
    __syncthreads();
    unsigned gt = 0u, ge = 0u;
#pragma unroll 8
    for (int o = 0; o < 256; ++o) {
      const unsigned m = hist[o];
      gt += m > mx;
      ge += m >= mx;
    }
    if (gt < k && k <= ge) bc[0] = mx;             // every thread that satisfies it holds the same value
    __syncthreads();
    const unsigned t0 = bc[0];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (key[j] != 0u && key[j] >= t0) {
        const int slot = atomicAdd(&lcount, 1);
        if (slot < LCAP_SEL) { lkey[slot] = key[j]; lidx[slot] = lo + tid + 256 * j; }
      }
    __syncthreads();
    const int ln = lcount;
    listed = ln <= LCAP_SEL;                        // (ln >= k: the k maxima that are >= T0 are in the list)
    if (listed) {
      // exact k-th largest of the list by rank counting; LCAP_SEL / 256 entries per thread
      for (int e = tid; e < ln; e += 256) {
        const unsigned me = lkey[e];
        unsigned a = 0u, b = 0u;
        for (int o = 0; o < ln; ++o) {
          const unsigned m = lkey[o];
          a += m > me;
          b += m >= me;
        }
        if (a < k && k <= b) bc[1] = me;
      }
      __syncthreads();
      thr = bc[1];
      // the kept keys leave in token-index order (slot = number of kept keys with a smaller index): the candidate layout, and
      // which keys survive when more than CAND are tied at the threshold (the lowest indices), do not depend on scheduling
      for (int e = tid; e < ln; e += 256)
        if (lkey[e] >= thr) {
          const int i = lidx[e];
          int slot = 0;
          for (int o = 0; o < ln; ++o) slot += (lkey[o] >= thr && lidx[o] < i) ? 1 : 0;
          if (slot < CAND) { cv[slot] = row[i]; ci[slot] = i; }
        }
      return;
    }
    __syncthreads();                                // overflow: the radix path re-uses hist / bc
  }
  thr = radix_select_256<NJ>(key, k, hist, bc);
  int running = 0;
  int* wc = reinterpret_cast<int*>(hist);             // (the select is done with hist)
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    if (running >= CAND) break;                       // uniform: every thread holds the same running count
    const bool on = key[j] != 0u && key[j] >= thr;
    const int slot = ordered_slot_256(on, running, wc);
    if (on && slot < CAND) {
      const int i = lo + tid + 256 * j;
      cv[slot] = row[i];
      ci[slot] = i;
    }
  }
}

// top_p < 1 / min_p > 0: the nucleus and min-p warpers of HF generate, applied in its order AFTER top-k on the temperature-scaled
// scores (generation/logits_process.py TopPLogitsWarper / MinPLogitsWarper; the reference forwards both flags to generate,
// grpo_trainer.py:559-570).  With p = softmax over the top-k set: a token stays iff the mass of the STRICTLY better tokens is
// below top_p (HF removes ascending-sorted tokens whose cumulative mass is <= 1 - top_p: the same set; tokens tied in score are
// kept or dropped together here) and iff p >= min_p * p_max (the best token survives top-p, so the renormalisation cancels).
__global__ __launch_bounds__(256) void sample_candidates_kernel(const float* __restrict__ cand_val, const int* __restrict__ cand_idx,
                                                                float inv_temp, int top_k, float top_p, float min_p,
                                                                unsigned long long seed_arg,
                                                                const unsigned long long* __restrict__ seed_dev,
                                                                const int* __restrict__ step, int* __restrict__ finished,
                                                                int eos_id, int pad_id, int* __restrict__ next_ids,
                                                                int* __restrict__ tokens, long ldt) {
  __shared__ unsigned hist[256];
  __shared__ unsigned bc[2];
  __shared__ float red_v[4];
  __shared__ int red_i[4];
  constexpr int NK = NSLICE * CAND / 256;
  const int g = blockIdx.x, tid = threadIdx.x;
  const unsigned long long seed = seed_dev ? seed_dev[0] : seed_arg;
  const float* cv = cand_val + (long)g * NSLICE * CAND;
  const int* ci = cand_idx + (long)g * NSLICE * CAND;
  unsigned key[NK];
  float val[NK];
  int idx[NK];
  unsigned nvalid = 0;
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    val[j] = cv[tid + 256 * j];
    idx[j] = ci[tid + 256 * j];
    key[j] = (idx[j] != 0x7fffffff) ? f2key(val[j]) : 0u;
    nvalid += key[j] != 0u;
  }
  // total valid candidates (>= top_k whenever V >= top_k)
  hist[tid] = nvalid;
  __syncthreads();
  if (tid == 0) { unsigned t = 0; for (int i = 0; i < 256; ++i) t += hist[i]; bc[0] = t; }
  __syncthreads();
  const unsigned total = bc[0];
  __syncthreads();
  const unsigned k = (unsigned)top_k < total ? (unsigned)top_k : total;
  const unsigned thr = radix_select_256<NK>(key, k, hist, bc);
  const unsigned st = (unsigned)step[0];
  unsigned drop = 0u;                       // bit j: candidate j of this thread was removed by a warper (a mask, not key[j] = 0: hipcc
                                            // 7.2 lost that store for j = 0 when only min_p applied -- seen in the ISA and in the draws)
  if (top_p < 1.0f || min_p > 0.0f) {       // wave-uniform: kernel arguments
    constexpr int LCAP = 512;                // kept scores (top-k + ties): beyond this many the warpers see the first LCAP only
    __shared__ float ls[LCAP];
    __shared__ int ln;
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NK; ++j)
      if (key[j] != 0u && key[j] >= thr) m = fmaxf(m, val[j] * inv_temp);
    m = block_max_256(m, red_v);
    __shared__ int wc4[4];
    int running = 0;
#pragma unroll
    for (int j = 0; j < NK; ++j) {                  // list slots in candidate order (ordered_slot_256): the sums below run over the
      const bool on = key[j] != 0u && key[j] >= thr; // same scores in the same order on every run
      const int slot = ordered_slot_256(on, running, wc4);
      if (on && slot < LCAP) ls[slot] = val[j] * inv_temp;
    }
    if (tid == 0) ln = running;
    __syncthreads();
    const int n = running < LCAP ? running : LCAP;
    float Z = 0.f;
    for (int i = 0; i < n; ++i) Z += __expf(ls[i] - m);        // every thread the same order: identical Z
    const float lim = top_p * Z;
#pragma unroll
    for (int j = 0; j < NK; ++j)
      if (key[j] != 0u && key[j] >= thr) {
        const float sj = val[j] * inv_temp;
        float better = 0.f;
        for (int i = 0; i < n; ++i) better += ls[i] > sj ? __expf(ls[i] - m) : 0.f;
        const bool keep = (top_p >= 1.0f || better < lim) && (__expf(sj - m) >= min_p);
        if (!keep) drop |= 1u << j;                           // dropped by a warper: out of the draw
      }
    __syncthreads();
  }
  float best = -INFINITY;
  int besti = 0x7fffffff;
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    if (key[j] != 0u && key[j] >= thr && !((drop >> j) & 1u)) {
      const float u = philox_uniform(seed, (unsigned)idx[j], (unsigned)g, st);
      const float sc = val[j] * inv_temp - __logf(-__logf(u));
      if (sc > best || (sc == best && idx[j] < besti)) { best = sc; besti = idx[j]; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
  }
  if ((tid & 63) == 0) { red_v[tid >> 6] = best; red_i[tid >> 6] = besti; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (red_v[w] > best || (red_v[w] == best && red_i[w] < besti)) { best = red_v[w]; besti = red_i[w]; }
    int tok = besti;
    if (finished) {
      if (finished[g]) tok = pad_id;
      else if (tok == eos_id) finished[g] = 1;
    }
    next_ids[g] = tok;
    if (tokens) tokens[(long)g * ldt + st] = tok;
  }
}

__global__ __launch_bounds__(256) void lse_gather_kernel(const float* __restrict__ logits, long ldl, int V, const int* __restrict__ tok,
                                                         float inv_temp, float* __restrict__ logp, float* __restrict__ lse_out) {
  __shared__ float scratch[4];
  const int r = blockIdx.x;
  const float* row = logits + (long)r * ldl;
  float m = -INFINITY, s = 0.f;
  for (int i = threadIdx.x * 4; i < V; i += 1024) {
    const float4 v4 = *reinterpret_cast<const float4*>(row + i);
    const float v[4] = {v4.x * inv_temp, v4.y * inv_temp, v4.z * inv_temp, v4.w * inv_temp};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (v[j] > m) { s = s * expf(m - v[j]) + 1.0f; m = v[j]; }
      else s += expf(v[j] - m);
    }
  }
  const float mm = block_max_256(m, scratch);
  const float ss = block_sum_256(m == -INFINITY ? 0.f : s * expf(m - mm), scratch);
  if (threadIdx.x == 0) {
    const float lse = mm + logf(ss);
    if (lse_out) lse_out[r] = lse;
    logp[r] = row[tok[r]] * inv_temp - lse;
  }
}

// loss types: 0 grpo, 1 bnpo, 2 dr_grpo, 3 gmpo, 4 mapo (= grpo aggregation), 5 mapo_bnpo (= bnpo aggregation)
// unit = rows per normalisation block of the token-mean aggregations (bnpo, mapo_bnpo) and of the kl / clip metrics: the
// reference computes them over ONE RANK's micro-batch (per_device_train_batch_size completions, 1 in train_grpo.sh) and then
// averages over ranks and accumulation steps (grpo_trainer.py:1148-1149, :1159-1177 + DDP / Trainer means); here a rank owns
// whole groups, so the B rows are cut into blocks of `unit` consecutive rows, each normalised by its own token count, and the
// blocks are averaged (blocks without tokens count as 0 in the loss, as clamp(min=1) does, and are skipped in the metrics,
// as nanmean does).  unit <= 0 or >= B: one block = the whole batch.
__global__ __launch_bounds__(256) void grpo_loss_kernel(const float* __restrict__ lp, const float* __restrict__ ref,
                                                        const float* __restrict__ old, const int* __restrict__ mask,
                                                        const float* __restrict__ adv, int B, int C, float beta, float eps_lo,
                                                        float eps_hi, int loss_type, float max_completion_length, int unit,
                                                        float* __restrict__ out /*[8]*/, float* __restrict__ dlogp) {
  __shared__ float scratch[4];
  __shared__ float cnt_s[256], raw_s[256], seq_s[256], gsum_s[256], blk_s[256], kl_s[256], low_s[256], high_s[256], reg_s[256];
  const int tid = threadIdx.x;
  const bool gm = loss_type == 3;
  const bool per_seq = (loss_type == 0 || loss_type == 3 || loss_type == 4);
  if (unit <= 0 || unit > B) unit = B;
  const int nblk = (B + unit - 1) / unit;
  // per-row token counts (and GMPO sequence log-ratio)
  for (int b = 0; b < B; ++b) {
    float c = 0.f, d = 0.f;
    for (int t = tid; t < C; t += 256) {
      const float mk = (float)mask[(long)b * C + t];
      c += mk;
      if (gm) d += (lp[(long)b * C + t] - (old ? old[(long)b * C + t] : lp[(long)b * C + t])) * mk;
    }
    c = block_sum_256(c, scratch);
    d = block_sum_256(d, scratch);
    if (tid == 0) { raw_s[b] = c; cnt_s[b] = fmaxf(c, 1.0f); seq_s[b] = d / fmaxf(c, 1.0f); }
  }
  __syncthreads();
  if (tid < nblk) {                                  // token count of every block
    float c = 0.f;
    for (int b = tid * unit; b < B && b < (tid + 1) * unit; ++b) c += raw_s[b];
    blk_s[tid] = c;
  }
  __syncthreads();
  const float lo = 1.0f - eps_lo, hi = 1.0f + eps_hi;
  float loss_acc = 0.f, tot_mask = 0.f;
  for (int b = 0; b < B; ++b) {
    const float A = adv[b];
    const float blk = fmaxf(blk_s[b / unit], 1.0f);
    float gs = 0.f, kl_acc = 0.f, low_acc = 0.f, high_acc = 0.f, reg_acc = 0.f;
    for (int t = tid; t < C; t += 256) {
      const long i = (long)b * C + t;
      const float mk = (float)mask[i];
      const float l = lp[i];
      const float o = old ? old[i] : l;
      float kl = 0.f, dkl = 0.f;
      if (beta != 0.f) {
        const float d = ref[i] - l;
        const float e = expf(d);
        kl = e - d - 1.0f;
        dkl = 1.0f - e;
      }
      const float coef1 = gm ? expf(seq_s[b]) : expf(l - o);
      const float coef2 = fminf(fmaxf(coef1, lo), hi);
      const float l1 = coef1 * A, l2 = coef2 * A;
      float ptl = -fminf(l1, l2) + beta * kl;
      const float inside = (coef1 >= lo && coef1 <= hi) ? 1.f : 0.f;
      const float g1 = l1 < l2 ? 1.f : (l1 == l2 ? 0.5f : 0.f);
      const float g2 = l2 < l1 ? 1.f : (l1 == l2 ? 0.5f : 0.f);
      const float dptl_dc1 = -(g1 * A + g2 * A * inside);
      float wgt;
      if (per_seq) wgt = mk / cnt_s[b] / (float)B;
      else if (loss_type == 2) wgt = mk / ((float)B * max_completion_length);
      else wgt = mk / blk / (float)nblk;
      loss_acc += ptl * wgt;
      kl_acc += kl * mk;
      const float isl = (coef1 < lo && A < 0.f) ? 1.f : 0.f;
      const float ish = (coef1 > hi && A > 0.f) ? 1.f : 0.f;
      low_acc += isl * mk; high_acc += ish * mk; reg_acc += fmaxf(isl, ish) * mk;
      if (dlogp) {
        if (gm) { gs += wgt * dptl_dc1; dlogp[i] = wgt * beta * dkl; }
        else dlogp[i] = wgt * (dptl_dc1 * coef1 + beta * dkl);
      }
    }
    if (gm && dlogp) gs = block_sum_256(gs, scratch);
    kl_acc = block_sum_256(kl_acc, scratch);
    low_acc = block_sum_256(low_acc, scratch);
    high_acc = block_sum_256(high_acc, scratch);
    reg_acc = block_sum_256(reg_acc, scratch);
    if (tid == 0) { gsum_s[b] = gs; kl_s[b] = kl_acc; low_s[b] = low_acc; high_s[b] = high_acc; reg_s[b] = reg_acc; tot_mask += raw_s[b]; }
  }
  __syncthreads();
  if (gm && dlogp) {
    for (int b = 0; b < B; ++b) {
      const float c1 = expf(seq_s[b]);
      for (int t = tid; t < C; t += 256) {
        const long i = (long)b * C + t;
        dlogp[i] += gsum_s[b] * c1 * ((float)mask[i] / cnt_s[b]);
      }
    }
  }
  loss_acc = block_sum_256(loss_acc, scratch);
  if (tid == 0) {
    // metrics: mean over the blocks that have tokens of (masked sum / token count)
    float kl = 0.f, low = 0.f, high = 0.f, reg = 0.f, live = 0.f;
    for (int k = 0; k < nblk; ++k) {
      if (blk_s[k] <= 0.f) continue;
      float a = 0.f, l = 0.f, h = 0.f, r = 0.f;
      for (int b = k * unit; b < B && b < (k + 1) * unit; ++b) { a += kl_s[b]; l += low_s[b]; h += high_s[b]; r += reg_s[b]; }
      kl += a / blk_s[k]; low += l / blk_s[k]; high += h / blk_s[k]; reg += r / blk_s[k]; live += 1.f;
    }
    out[0] = loss_acc;
    out[1] = kl / live;                              // 0/0 = NaN when no block has tokens, as the reference's nanmean of nothing
    out[2] = low / live;
    out[3] = high / live;
    out[4] = reg / live;
    out[5] = tot_mask;
  }
}

}  // namespace

// HF RepetitionPenaltyLogitsProcessor (generation/logits_process.py; grpo_trainer.py:559-570 forwards the flag): for every token id
// that occurs in the sequence so far (prompt + generated), logit = logit < 0 ? logit * penalty : logit / penalty -- once per id.
// seen: one bit per vocabulary entry and row, kept across the decode steps; new_ids[g] (the token appended since the last call) is
// folded in here by the thread that owns its word.
__global__ __launch_bounds__(256) void repetition_penalty_kernel(float* __restrict__ logits, long ldl, int V, float penalty,
                                                                 unsigned* __restrict__ seen, int words, const int* __restrict__ new_ids) {
  const int g = blockIdx.y;
  unsigned* sb = seen + (long)g * words;
  int tw = -1;
  unsigned tb = 0u;
  if (new_ids) {
    const int t = new_ids[g];
    if (t >= 0 && t < V) { tw = t >> 5; tb = 1u << (t & 31); }
  }
  float* row = logits + (long)g * ldl;
  for (int w = blockIdx.x * 256 + threadIdx.x; w < words; w += gridDim.x * 256) {
    unsigned bits = sb[w];
    if (w == tw) { bits |= tb; sb[w] = bits; }     // (the grid-stride loop hands every word to exactly one thread)
    while (bits) {
      const int b = __ffs(bits) - 1;
      bits &= bits - 1;
      const int i = w * 32 + b;
      if (i < V) { const float l = row[i]; row[i] = l < 0.f ? l * penalty : l / penalty; }
    }
  }
}

extern "C" int vr2_repetition_penalty(void* logits_f32, int64_t ldl, int64_t G, int64_t V, float penalty, void* seen_bits_u32,
                                      int64_t words_per_row, const void* new_ids_i32, void* stream) {
  VR2_REQUIRE(G >= 1 && V >= 1 && penalty > 0.f && words_per_row * 32 >= V, "vr2_repetition_penalty: bad arguments (penalty > 0, words * 32 >= V)");
  const unsigned bx = (unsigned)((words_per_row + 255) / 256);
  hipLaunchKernelGGL(repetition_penalty_kernel, dim3(bx, (unsigned)G), dim3(256), 0, (hipStream_t)stream, (float*)logits_f32, (long)ldl,
                     (int)V, penalty, (unsigned*)seen_bits_u32, (int)words_per_row, (const int*)new_ids_i32);
  return vr2_check_launch("vr2_repetition_penalty");
}

extern "C" int vr2_sample_topk_p(const void* logits_f32, int64_t ldl, int64_t G, int64_t V, float temperature, int top_k,
                                 float top_p, float min_p, uint64_t seed, const void* seed_dev_u64, const void* step_i32,
                                 void* finished_i32, int eos_id, int pad_id, void* next_ids_i32, void* tokens_i32, int64_t ldt,
                                 void* cand_ws, void* stream) {
  VR2_REQUIRE(G >= 1 && V >= 1 && temperature > 0.f, "vr2_sample_topk: bad arguments (temperature must be > 0)");
  VR2_REQUIRE(top_p > 0.f && top_p <= 1.f && min_p >= 0.f && min_p <= 1.f, "vr2_sample_topk_p: top_p=%g must be in (0,1], min_p=%g in [0,1]", top_p, min_p);
  const bool warp = top_p < 1.f || min_p > 0.f;
  VR2_REQUIRE(!warp || (cand_ws != nullptr && top_k >= 1 && top_k <= CAND && V <= NSLICE * 256 * NJ),
              "vr2_sample_topk_p: top_p / min_p are built on the two-stage top-k path: they need 1 <= top_k <= %d (got %d) and the "
              "candidate workspace", CAND, top_k);
  if (cand_ws != nullptr && top_k >= 1 && top_k <= CAND && V <= NSLICE * 256 * NJ) {
    // two-stage path; workspace: G*NSLICE*CAND floats followed by as many ints (vr2_sample_ws_bytes)
    float* cv = (float*)cand_ws;
    int* ci = (int*)(cv + (long)G * NSLICE * CAND);
    hipLaunchKernelGGL(topk_candidates_kernel, dim3(NSLICE, (unsigned)G), dim3(256), 0, (hipStream_t)stream,
                       (const float*)logits_f32, (long)ldl, (int)V, top_k, cv, ci);
    if (vr2_check_launch("vr2_sample_topk(candidates)")) return -2;
    hipLaunchKernelGGL(sample_candidates_kernel, dim3((unsigned)G), dim3(256), 0, (hipStream_t)stream, (const float*)cv,
                       (const int*)ci, 1.0f / temperature, top_k, top_p, min_p, (unsigned long long)seed,
                       (const unsigned long long*)seed_dev_u64, (const int*)step_i32, (int*)finished_i32, eos_id, pad_id,
                       (int*)next_ids_i32, (int*)tokens_i32, (long)ldt);
    return vr2_check_launch("vr2_sample_topk(draw)");
  }
  hipLaunchKernelGGL(sample_topk_kernel, dim3((unsigned)G), dim3(1024), 0, (hipStream_t)stream, (const float*)logits_f32,
                     (long)ldl, (int)V, 1.0f / temperature, top_k, (unsigned long long)seed,
                     (const unsigned long long*)seed_dev_u64, (const int*)step_i32,
                     (int*)finished_i32, eos_id, pad_id, (int*)next_ids_i32, (int*)tokens_i32, (long)ldt);
  return vr2_check_launch("vr2_sample_topk");
}

extern "C" int vr2_sample_topk(const void* logits_f32, int64_t ldl, int64_t G, int64_t V, float temperature, int top_k,
                               uint64_t seed, const void* seed_dev_u64, const void* step_i32, void* finished_i32,
                               int eos_id, int pad_id, void* next_ids_i32, void* tokens_i32, int64_t ldt, void* cand_ws,
                               void* stream) {
  return vr2_sample_topk_p(logits_f32, ldl, G, V, temperature, top_k, 1.0f, 0.0f, seed, seed_dev_u64, step_i32, finished_i32, eos_id,
                           pad_id, next_ids_i32, tokens_i32, ldt, cand_ws, stream);
}

extern "C" int64_t vr2_sample_ws_bytes(int64_t G) { return G * NSLICE * CAND * 8; }

extern "C" int vr2_lse_gather(const void* logits_f32, int64_t ldl, int64_t R, int64_t V, const void* tok_i32,
                              float temperature, void* logp_f32, void* lse_f32, void* stream) {
  VR2_REQUIRE(R >= 1 && V % 4 == 0 && ldl % 4 == 0 && temperature > 0.f, "vr2_lse_gather: bad arguments");
  hipLaunchKernelGGL(lse_gather_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, (const float*)logits_f32,
                     (long)ldl, (int)V, (const int*)tok_i32, 1.0f / temperature, (float*)logp_f32, (float*)lse_f32);
  return vr2_check_launch("vr2_lse_gather");
}

extern "C" int vr2_grpo_loss(const void* logp_f32, const void* ref_logp_f32, const void* old_logp_f32, const void* mask_i32,
                             const void* adv_f32, int64_t B, int64_t C, float beta, float eps_low, float eps_high,
                             int loss_type, float max_completion_length, int64_t norm_rows, void* out8_f32, void* dlogp_f32,
                             void* stream) {
  VR2_REQUIRE(B >= 1 && B <= 256 && C >= 1, "vr2_grpo_loss: B=%ld must be in [1,256]", (long)B);
  VR2_REQUIRE(loss_type >= 0 && loss_type <= 5, "vr2_grpo_loss: unknown loss type %d", loss_type);
  VR2_REQUIRE(beta == 0.f || ref_logp_f32 != nullptr, "vr2_grpo_loss: beta != 0 needs reference log-probs");
  hipLaunchKernelGGL(grpo_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)logp_f32,
                     (const float*)ref_logp_f32, (const float*)old_logp_f32, (const int*)mask_i32, (const float*)adv_f32,
                     (int)B, (int)C, beta, eps_low, eps_high, loss_type, max_completion_length, (int)norm_rows,
                     (float*)out8_f32, (float*)dlogp_f32);
  return vr2_check_launch("vr2_grpo_loss");
}
