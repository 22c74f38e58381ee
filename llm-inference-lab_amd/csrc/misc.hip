// Small kernels around the forward: embedding gather, argmax finalisation, and the
// device-side control of the draft-then-verify loop (next draft token, accept scan,
// in-place state advance).

#include "kernels.h"
#include "engine.h"

namespace sd {

// x[t][:] = tok_emb[clamp(token[t])][:] (+ pos_emb[pos][:] for GPT-2)
// clamp = validate_and_clamp_tokens, /root/reference/src/specdec/utils/token_validation.py:15-78
__global__ __launch_bounds__(256) void embed_kernel(const EmbedArgs a) {
  SD_SKIP_IF_INACTIVE(a.skip_k, a.skip_i);
  const int t = blockIdx.x;
  const int b = t / a.M, m = t - b * a.M;
  int tok = a.tokens[b * a.tok_stride + m];
  tok = tok < 0 ? 0 : (tok >= a.vocab ? a.vocab - 1 : tok);
  const uint4* src = reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(a.tok_emb) + static_cast<size_t>(tok) * a.d);
  uint4* dst = reinterpret_cast<uint4*>(static_cast<uint16_t*>(a.x) + static_cast<size_t>(t) * a.d);
  const int nvec = a.d >> 3;
  if (!a.pos_emb) {
    for (int v = threadIdx.x; v < nvec; v += blockDim.x) dst[v] = src[v];
    return;
  }
  int pos = a.pos_base[b] + a.pos_off + m;
  pos = pos < 0 ? 0 : (pos >= a.max_pos ? a.max_pos - 1 : pos);
  const uint4* ps = reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(a.pos_emb) + static_cast<size_t>(pos) * a.d);
  for (int v = threadIdx.x; v < nvec; v += blockDim.x) {
    const uint4 e = src[v], p = ps[v];
    const uint32_t ew[4] = {e.x, e.y, e.z, e.w}, pw[4] = {p.x, p.y, p.z, p.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = __uint_as_float(ew[j] << 16) + __uint_as_float(pw[j] << 16);
      const float hi = __uint_as_float(ew[j] & 0xffff0000u) + __uint_as_float(pw[j] & 0xffff0000u);
      o[j] = static_cast<uint32_t>(float_to_bf16_bits(lo)) | (static_cast<uint32_t>(float_to_bf16_bits(hi)) << 16);
    }
    dst[v] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

int launch_embed(const EmbedArgs& a, hipStream_t st) {
  SD_REQUIRE(a.d % 8 == 0, "embed: d=%d must be a multiple of 8", a.d);
  hipLaunchKernelGGL(embed_kernel, dim3(a.T), dim3(256), 0, st, a);
  SD_LAUNCH_CHECK();
  return 0;
}

// wave-wide fold of one token's per-workgroup partials
__device__ __forceinline__ int fold_partials(const float* pv, const int* pi, int grid, int lane) {
  float v = -INFINITY;
  int i = 0x7fffffff;
  for (int s = lane; s < grid; s += kWave) {
    const float sv = pv[s];
    const int si = pi[s];
    if (argmax_better(sv, si, v, i)) { v = sv; i = si; }
  }
  wave_reduce_argmax(v, i);
  return i;
}

// ids[b*ids_stride + m] = argmax over the lm_head partials of token t = b*M + m
__global__ __launch_bounds__(kWave) void argmax_finalize_kernel(const float* part_val, const int* part_idx,
                                                                int grid, int M, int ids_stride,
                                                                int32_t* ids, const int32_t* skip_k, int skip_i) {
  SD_SKIP_IF_INACTIVE(skip_k, skip_i);
  const int t = blockIdx.x, lane = threadIdx.x;
  const int i = fold_partials(part_val + static_cast<size_t>(t) * grid, part_idx + static_cast<size_t>(t) * grid, grid, lane);
  if (lane == 0) {
    // M > 0: token t = b * M + m. M < 0 (the batched head launch): t = j * B + b with B = -M rows per matrix j -> ids[b][j]
    if (M > 0) {
      const int b = t / M, m = t - b * M;
      ids[b * ids_stride + m] = i;
    } else {
      const int j = t / -M, b = t + j * M;
      ids[b * ids_stride + j] = i;
    }
  }
}

int launch_argmax_finalize(const float* part_val, const int* part_idx, int T, int grid, int M,
                           int ids_stride, int32_t* ids_out, hipStream_t st, const int32_t* skip_k, int skip_i) {
  hipLaunchKernelGGL(argmax_finalize_kernel, dim3(T), dim3(kWave), 0, st, part_val, part_idx, grid, M,
                     ids_stride, ids_out, skip_k, skip_i);
  SD_LAUNCH_CHECK();
  return 0;
}

// ---- draft: token i of the proposal ------------------------------------------------
// The draft forward produced M tokens per row; the last one's argmax is d_{i+1}.
__global__ void draft_next_kernel(int M, int i, SpecState s) {
  if (s.adaptive && i >= 1) SD_SKIP_IF_INACTIVE(s.k_active, i);   // forward i did not run: d_{i+1} counts for no row
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= s.B) return;
  const int d = s.draft_ids[b * 2 + (M - 1)];
  s.draft_tok[b * s.K + i] = d;
  s.verify_tok[b * (s.K + 1) + i + 1] = d;
  s.next_tok[b] = d;
}

// The two above as one launch (one wave per token of the draft pass): the lm_head partials of token (b, m) -> draft_ids[b][m],
// and the row's last token is d_{i+1}. `ids` is draft_ids, or draft_ids + 1 for the 1-token form of forward 0 (whose id lands
// where the 2-token form leaves its second one). skip_k / skip_i: the words the forward itself was gated on.
__global__ __launch_bounds__(kWave) void draft_finalize_kernel(const float* part_val, const int* part_idx, int grid, int M, int i,
                                                               int32_t* ids, SpecState s, const int32_t* skip_k, int skip_i) {
  // (the gate word is loaded next to the partials, not in front of them: one memory round trip instead of two in a launch
  //  that is nothing but latency; a gated-off launch folds stale partials and stores nothing)
  const int gate = skip_k ? *skip_k : 0x7fffffff;
  const int t = blockIdx.x, lane = threadIdx.x;
  const int d = fold_partials(part_val + static_cast<size_t>(t) * grid, part_idx + static_cast<size_t>(t) * grid, grid, lane);
  if (lane != 0 || gate <= skip_i) return;
  const int b = t / M, m = t - b * M;
  ids[b * 2 + m] = d;
  if (m == M - 1) {
    s.draft_tok[b * s.K + i] = d;
    s.verify_tok[b * (s.K + 1) + i + 1] = d;
    s.next_tok[b] = d;
  }
}

int launch_draft_finalize(const float* part_val, const int* part_idx, int grid, int M, int i, int32_t* ids, const SpecState& s,
                          const int32_t* skip_k, int skip_i, hipStream_t st) {
  SD_REQUIRE(M >= 1 && M <= 2 && i >= 0 && i < s.K, "draft_finalize: M=%d i=%d", M, i);
  hipLaunchKernelGGL(draft_finalize_kernel, dim3(s.B * M), dim3(kWave), 0, st, part_val, part_idx, grid, M, i, ids, s, skip_k, skip_i);
  SD_LAUNCH_CHECK();
  return 0;
}

// Medusa-lite with heads tied to (or copied from) the lm_head, greedy (modes/medusa.py:71-186): every head
// is the lm_head and the draftor re-uses head 0 on the SAME hidden state for all K proposals, so the draft is
// K copies of the target's own next token (the argmax of the M = 1 forward that precedes this kernel).
__global__ void medusa_fill_kernel(SpecState s) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= s.B) return;
  const int d = s.draft_ids[b * 2];
  for (int i = 0; i < s.K; ++i) {
    s.draft_tok[b * s.K + i] = d;
    s.verify_tok[b * (s.K + 1) + i + 1] = d;
  }
  s.next_tok[b] = d;
}

// Persistent Medusa heads: row of the target's residual stream that predicted the last emitted token
// (position accept_len of the verify pass), and the hand-over of the heads' tokens to the next step.
__global__ void medusa_rows_kernel(SpecState s, int32_t* row_idx) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < s.B) row_idx[b] = b * (s.K + 1) + s.accept_len[b];
}

// more rows than one GEMV pass takes (> 9): the rows the heads read, gathered into a contiguous block for the
// multi-token kernels (which have no row indirection)
__global__ __launch_bounds__(256) void medusa_gather_kernel(const uint16_t* x, const int32_t* rows, uint16_t* out, int d) {
  const uint4* src = reinterpret_cast<const uint4*>(x + static_cast<size_t>(rows[blockIdx.x]) * d);
  uint4* dst = reinterpret_cast<uint4*>(out + static_cast<size_t>(blockIdx.x) * d);
  for (int i = threadIdx.x; i < d / 8; i += blockDim.x) dst[i] = src[i];
}

int launch_medusa_gather(const void* x, const int32_t* rows, void* out, int B, int d, hipStream_t st) {
  SD_REQUIRE(d % 8 == 0, "medusa_gather: d=%d", d);
  hipLaunchKernelGGL(medusa_gather_kernel, dim3(B), dim3(256), 0, st, static_cast<const uint16_t*>(x), rows, static_cast<uint16_t*>(out), d);
  SD_LAUNCH_CHECK();
  return 0;
}

__global__ void medusa_commit_kernel(SpecState s) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= s.B) return;
  for (int i = 0; i < s.K; ++i) s.draft_tok[b * s.K + i] = s.verify_tok[b * (s.K + 1) + i + 1];
}

// EAGLE-lite (the reference's _run_eagle_hf, pipeline.py:765-889, greedy): the draft is read off EXTRAPOLATED hidden
// states of the target, no draft model. h_t = final_norm(residual row of the last token); with the state the previous
// step left behind (its LAST extrapolated row; none on a row's first step, where h_t stands in for it)
//   h_1 = h_t + alpha (h_t - E),  h_2 = h_1 + alpha (h_1 - h_t), ...   d_i = argmax lm_head(h_i)
// Every operation rounds to bf16 as the bf16 tensors of the reference do (difference, alpha * difference in fp32 with
// alpha as a float, sum). The h_i do not depend on the tokens, so all K rows are produced here and ONE lm_head launch
// scores them. One workgroup per batch row; the row's state E <- h_K.
struct EagleArgs {
  const uint16_t* x;   // [B][d] residual rows (forward with skip_head)
  uint16_t* H;         // [B*K][d] extrapolated rows
  uint16_t* prev;      // [B][d] state E
  int32_t* has_prev;   // [B]
  const uint16_t* norm_w;
  const uint16_t* norm_b;
  float eps, alpha;
  int d, K, rms;
};

__global__ __launch_bounds__(256) void eagle_extrapolate_kernel(EagleArgs a) {
  __shared__ float red[2][4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const uint16_t* x = a.x + static_cast<size_t>(b) * a.d;
  float s1 = 0.f, s2 = 0.f;
  for (int i = tid; i < a.d; i += 256) {
    const float v = bf16_bits_to_float(x[i]);
    s1 += v;
    s2 += v * v;
  }
  s1 = wave_reduce_sum(s1);
  s2 = wave_reduce_sum(s2);
  if ((tid & 63) == 0) { red[0][tid >> 6] = s1; red[1][tid >> 6] = s2; }
  __syncthreads();
  const float sum = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  const float sq = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  const float invd = 1.0f / static_cast<float>(a.d);
  const float mean = sum * invd;
  const float rs = a.rms ? rsqrtf(sq * invd + a.eps) : rsqrtf(fmaxf(sq * invd - mean * mean, 0.f) + a.eps);
  const bool has = a.has_prev[b] != 0;
  uint16_t* prev = a.prev + static_cast<size_t>(b) * a.d;
  for (int i = tid; i < a.d; i += 256) {
    const float v = bf16_bits_to_float(x[i]);
    float ht;
    if (a.rms) {   // HF LlamaRMSNorm: weight * (x * rsqrt(var + eps)).to(bf16)
      ht = bf16_bits_to_float(float_to_bf16_bits(bf16_bits_to_float(float_to_bf16_bits(v * rs)) * bf16_bits_to_float(a.norm_w[i])));
    } else {       // LayerNorm in fp32, rounded once
      ht = bf16_bits_to_float(float_to_bf16_bits((v - mean) * rs * bf16_bits_to_float(a.norm_w[i]) + bf16_bits_to_float(a.norm_b[i])));
    }
    float prv = has ? bf16_bits_to_float(prev[i]) : ht, cur = ht;
    for (int k = 0; k < a.K; ++k) {
      const float diff = bf16_bits_to_float(float_to_bf16_bits(cur - prv));
      const float sc = bf16_bits_to_float(float_to_bf16_bits(a.alpha * diff));
      const uint16_t nb = float_to_bf16_bits(cur + sc);
      a.H[(static_cast<size_t>(b) * a.K + k) * a.d + i] = nb;
      prv = cur;
      cur = bf16_bits_to_float(nb);
    }
    prev[i] = float_to_bf16_bits(cur);
  }
  if (tid == 0) a.has_prev[b] = 1;
}

int launch_eagle_extrapolate(const void* x, void* H, void* prev, int32_t* has_prev, const void* norm_w, const void* norm_b,
                             float eps, float alpha, int d, int B, int K, int rms, hipStream_t st) {
  EagleArgs a{};
  a.x = static_cast<const uint16_t*>(x);
  a.H = static_cast<uint16_t*>(H);
  a.prev = static_cast<uint16_t*>(prev);
  a.has_prev = has_prev;
  a.norm_w = static_cast<const uint16_t*>(norm_w);
  a.norm_b = static_cast<const uint16_t*>(norm_b);
  a.eps = eps;
  a.alpha = alpha;
  a.d = d;
  a.K = K;
  a.rms = rms;
  hipLaunchKernelGGL(eagle_extrapolate_kernel, dim3(B), dim3(256), 0, st, a);
  SD_LAUNCH_CHECK();
  return 0;
}

int launch_medusa_rows(const SpecState& s, int32_t* row_idx, hipStream_t st) {
  hipLaunchKernelGGL(medusa_rows_kernel, dim3((s.B + 63) / 64), dim3(64), 0, st, s, row_idx);
  SD_LAUNCH_CHECK();
  return 0;
}

int launch_medusa_commit(const SpecState& s, hipStream_t st) {
  hipLaunchKernelGGL(medusa_commit_kernel, dim3((s.B + 63) / 64), dim3(64), 0, st, s);
  SD_LAUNCH_CHECK();
  return 0;
}

int launch_medusa_fill(const SpecState& s, hipStream_t st) {
  hipLaunchKernelGGL(medusa_fill_kernel, dim3((s.B + 63) / 64), dim3(64), 0, st, s);
  SD_LAUNCH_CHECK();
  return 0;
}

int launch_draft_next(int M, int i, const SpecState& s, hipStream_t st) {
  hipLaunchKernelGGL(draft_next_kernel, dim3((s.B + 63) / 64), dim3(64), 0, st, M, i, s);
  SD_LAUNCH_CHECK();
  return 0;
}

// ---- verify: accept scan + state advance ----------------------------------------------
// One wave per batch row. t_m = argmax of the target at position m of (last, d_1..d_K);
// lane k holds match[k] = (t_k == d_{k+1}); __ballot + ctz gives the longest accepted
// prefix (the contract of verify_prefix_ref, reference.py:36-53, and of
// LongestPrefixPolicy.accept_tokens, policies.py:156-180, in one instruction).
// The row's state is advanced on the device for the common case so that the next step
// can be launched without a host round trip; the host re-synchronises a row whenever
// the reference's host-side rules (EOS cut, de-duplication, budget) say otherwise.
__device__ __forceinline__ int accept_scan(const SpecState& s, int b, int lane, int my_t) {
  const int K = s.K;
  const bool match = (lane < K) && (my_t == s.draft_tok[b * K + lane]);
  const unsigned long long m64 = __ballot(match);
  const unsigned long long valid = (K >= 64) ? ~0ull : ((1ull << K) - 1ull);
  const unsigned long long miss = (~m64) & valid;
  const int a = miss ? __builtin_ctzll(miss) : K;
  return s.adaptive ? min(a, s.k_row[b]) : a;   // per-row K: proposals past k_row[b] do not count
}

// AdaptiveKController.get_k (controllers.py:100-126) for one row, fed with the row's own cumulative acceptance rate
// (strict: accepted draft tokens / proposals that counted): history of the last four rates, their mean against the
// band target +- 0.1, K moved by step_size inside [min_k, max_k]. Double arithmetic in the reference's order
// (sum(h[-4:]) / 4 adds oldest first), so the host mirror (the same class, per row) agrees bit for bit.
__device__ __forceinline__ void adaptive_update(const SpecState& s, int b, int a) {
  int32_t* c = s.ctl + 4 * b;
  double* h = s.ctl_hist + 4 * b;
  const int k = s.k_row[b];
  c[3] = k;
  if (!s.active[b]) return;
  c[0] += a;
  c[1] += k;
  const double rate = static_cast<double>(c[0]) / static_cast<double>(c[1] > 0 ? c[1] : 1);
  int n = c[2];
  if (n < 4) {
    h[n++] = rate;
  } else {
    h[0] = h[1];
    h[1] = h[2];
    h[2] = h[3];
    h[3] = rate;
  }
  c[2] = n;
  if (n >= 4) {
    const double recent = (((h[0] + h[1]) + h[2]) + h[3]) / 4.0;
    int kn = k;
    if (recent > s.a_hi) kn = min(k + s.a_step, s.a_max);
    else if (recent < s.a_lo) kn = max(k - s.a_step, s.a_min);
    s.k_row[b] = kn;
  }
}

// sampling mode: the accept length alone, so that the sampler knows which logits row to draw from
__global__ __launch_bounds__(kWave) void accept_len_kernel(SpecState s) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int my_t = (lane <= s.K) ? s.target_ids[b * (s.K + 1) + lane] : -1;
  const int a = accept_scan(s, b, lane, my_t);
  if (lane == 0) s.accept_len[b] = a;
}

// one wave, one row: my_t = the target's id at verify position `lane` (lanes 0..K)
__device__ __forceinline__ void accept_row(const SpecState& s, int b, int lane, int my_t, int mode, int use_sampled) {
  const int K = s.K, M = K + 1;
  const int a = accept_scan(s, b, lane, my_t);
  // sampled bonus token (pipeline.py:3140-3160 / :3351-3361): the token after the accepted prefix
  // is drawn from the target distribution at that position instead of its argmax
  if (use_sampled && lane == a && s.active[b]) my_t = s.sampled[b];

  // tokens emitted by this step
  //   mode 0 (generate_batch, pipeline.py:3059-3292): base tokens t_0..t_{a-1} + bonus t_a
  //   mode 1 (generate, pipeline.py:1190-1235): draft tokens d_1..d_a, or t_0 when a == 0
  const int n_new = (mode == 0) ? a + 1 : (a > 0 ? a : 1);
  if (lane < M) s.new_tok[b * M + lane] = (lane < n_new) ? my_t : -1;
  const int last_new = __shfl(my_t, n_new - 1, 64);
  const int prev_new = __shfl(my_t, n_new >= 2 ? n_new - 2 : 0, 64);
  if (lane == 0) {
    s.accept_len[b] = a;
    s.n_new[b] = n_new;
    if (s.fwd0_w && b == 0) {
      // Next step's draft forward 0 (engine.hip enqueue_step). New `prev` sits at old cur_len + n_new - 1. Bonus mode: that is
      // d_a (a >= 1; an input of draft forward a, which ran iff a < k) or the old `last` (a == 0; forward 0's input): its K/V
      // are in the draft cache unless a == k. Draft-emit mode: prev is d_{a-1} or the old `last`: always there.
      const int k_eff = s.adaptive ? s.k_row[b] : K;   // (before this step's controller update)
      const bool two = !s.active[b] || (mode == 0 && a >= k_eff);
      s.fwd0_w[0] = two ? 2 : 1;
      s.fwd0_w[1] = two ? 1 : 2;
    }
    if (s.adaptive) adaptive_update(s, b, a);
    if (s.active[b]) {
      const int old_last = s.tok2[b * 2 + 1];
      s.tok2[b * 2 + 0] = (n_new >= 2) ? prev_new : old_last;
      s.tok2[b * 2 + 1] = last_new;
      s.verify_tok[b * M + 0] = last_new;
      s.cur_len[b] += n_new;
    }
  }
}

__global__ __launch_bounds__(kWave) void accept_kernel(SpecState s, int mode, int use_sampled) {
  const int b = blockIdx.x, lane = threadIdx.x;
  accept_row(s, b, lane, (lane <= s.K) ? s.target_ids[b * (s.K + 1) + lane] : -1, mode, use_sampled);
}

int launch_accept(const SpecState& s, int mode, int use_sampled, hipStream_t st) {
  SD_REQUIRE(s.K >= 1 && s.K <= 63, "accept: K=%d out of range 1..63", s.K);
  SD_REQUIRE(!use_sampled || (s.sampled && mode == 0), "accept: sampled bonus needs the bonus emit mode");
  hipLaunchKernelGGL(accept_kernel, dim3(s.B), dim3(kWave), 0, st, s, mode, use_sampled);
  SD_LAUNCH_CHECK();
  return 0;
}

// The step record, written straight into pinned host memory (device-accessible). Two slots, selected by the parity
// of a device-resident step counter, so that the host can still read step s while step s+1 (launched ahead) writes
// its own. One wave walks the rows, then advances the counter.
struct RecordArgs {
  int32_t* rec_slots;            // pinned host memory: 2 slots x [B][rec_ints]
  int rec_ints;
  int32_t* step_counter;         // device
  const unsigned* draft_status;  // status words of the persistent launches (null: none)
  const unsigned* target_status;
};

__device__ __forceinline__ void record_rows(const SpecState& s, const RecordArgs& ra, int lane) {
  int32_t* const rec_slots = ra.rec_slots;
  const int rec_ints = ra.rec_ints;
  int32_t* const step_counter = ra.step_counter;
  const unsigned* const draft_status = ra.draft_status;
  const unsigned* const target_status = ra.target_status;
  const int K = s.K;
  const int slot = *step_counter & 1;
  int32_t* rec = rec_slots + static_cast<size_t>(slot) * s.B * rec_ints;
  for (int b = 0; b < s.B; ++b) {
    int32_t* r = rec + static_cast<size_t>(b) * rec_ints;
    if (lane == 0) {
      r[0] = s.accept_len[b];
      r[1] = s.n_new[b];
      r[2] = s.cur_len[b];
    }
    if (lane <= K) {
      r[3 + lane] = s.new_tok[b * (K + 1) + lane];
      r[4 + 2 * K + lane] = s.target_ids[b * (K + 1) + lane];
    }
    if (lane < K) r[4 + K + lane] = s.draft_tok[b * K + lane];
    if (lane == 0) r[5 + 3 * K] = s.adaptive ? s.ctl[4 * b + 3] : K;   // proposals that counted for the row in this step
    // health of the persistent launches (sd_model_engine_status): non-zero = a launch of this or an earlier step gave up
    if (lane == 0) r[6 + 3 * K] = static_cast<int32_t>((draft_status ? *draft_status : 0u) | (target_status ? *target_status : 0u));
  }
  if (s.adaptive) {   // widest row of the next step
    int ka = 0;
    for (int b = lane; b < s.B; b += kWave) ka = max(ka, s.active[b] ? s.k_row[b] : 0);
    for (int off = 32; off > 0; off >>= 1) ka = max(ka, __shfl_xor(ka, off, 64));
    if (lane == 0) *s.k_active = ka;
  }
  if (lane == 0) *step_counter = *step_counter + 1;
}

__global__ __launch_bounds__(kWave) void pack_record_kernel(SpecState s, RecordArgs ra) { record_rows(s, ra, threadIdx.x); }

int launch_pack_record(const SpecState& s, int32_t* rec_slots, int rec_ints, int32_t* step_counter, const unsigned* draft_status,
                       const unsigned* target_status, hipStream_t st) {
  hipLaunchKernelGGL(pack_record_kernel, dim3(1), dim3(kWave), 0, st, s, RecordArgs{rec_slots, rec_ints, step_counter, draft_status, target_status});
  SD_LAUNCH_CHECK();
  return 0;
}

// ---- verify: the three above as ONE launch (greedy steps) ----------------------------------------------------------------
// The lm_head's per-workgroup partials of the B x (K+1) verify positions -> target_ids, the accept scan and state advance of
// every row, the step record. One workgroup of up to 16 waves: wave w folds the partials of positions w, w + n_waves, ...
// (ids through LDS), then waves take rows (accept_row), then wave 0 writes the record — 3 launches of ~4.5 us become one.
constexpr int kTailMaxTokens = 144;   // B x (K+1) positions the one-workgroup tail takes (16 rows at K = 8); more: the three launches
__global__ __launch_bounds__(1024) void verify_tail_kernel(SpecState s, const float* part_val, const int* part_idx, int grid, int mode,
                                                           RecordArgs ra) {
  __shared__ int ids[kTailMaxTokens];
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave, nw = blockDim.x / kWave;
  const int M = s.K + 1, T = s.B * M;
  for (int t = w; t < T; t += nw) {
    const int id = fold_partials(part_val + static_cast<size_t>(t) * grid, part_idx + static_cast<size_t>(t) * grid, grid, lane);
    if (lane == 0) {
      ids[t] = id;
      s.target_ids[t] = id;
    }
  }
  __syncthreads();
  for (int b = w; b < s.B; b += nw) accept_row(s, b, lane, (lane < M) ? ids[b * M + lane] : -1, mode, 0);
  __syncthreads();   // (the rows' state is in global memory: written and read by this one workgroup)
  if (w == 0) record_rows(s, ra, lane);
}

bool verify_tail_fits(const SpecState& s) { return s.B * (s.K + 1) <= kTailMaxTokens && s.K <= 63; }

int launch_verify_tail(const SpecState& s, const float* part_val, const int* part_idx, int grid, int mode, int32_t* rec_slots,
                       int rec_ints, int32_t* step_counter, const unsigned* draft_status, const unsigned* target_status, hipStream_t st) {
  SD_REQUIRE(verify_tail_fits(s), "verify_tail: B=%d K=%d", s.B, s.K);
  const int T = s.B * (s.K + 1);
  const int nw = T < 16 ? T : 16;
  hipLaunchKernelGGL(verify_tail_kernel, dim3(1), dim3(nw * kWave), 0, st, s, part_val, part_idx, grid, mode,
                     RecordArgs{rec_slots, rec_ints, step_counter, draft_status, target_status});
  SD_LAUNCH_CHECK();
  return 0;
}

int launch_accept_len(const SpecState& s, hipStream_t st) {
  SD_REQUIRE(s.K >= 1 && s.K <= 63, "accept: K=%d out of range 1..63", s.K);
  hipLaunchKernelGGL(accept_len_kernel, dim3(s.B), dim3(kWave), 0, st, s);
  SD_LAUNCH_CHECK();
  return 0;
}

}  // namespace sd
