// Engine-internal types: model instance and the device-side state of the
// draft-then-verify loop. The C-ABI view of these is in include/specdec_hip.h.
#pragma once

#include <vector>

#include "kernels.h"

namespace sd {

// device-resident loop state (all int32, B rows, K draft tokens per step)
struct SpecState {
  int B, K;
  int32_t* cur_len;     // [B]   tokens whose KV is final in the target cache (= position of `last`)
  int32_t* active;      // [B]   1 = row advances
  int32_t* tok2;        // [B][2]   (prev, last): input of draft forward 0
  int32_t* next_tok;    // [B]      input of draft forwards 1..K-1
  int32_t* draft_ids;   // [B][2]   argmax ids of the current draft forward
  int32_t* draft_tok;   // [B][K]   d_1..d_K
  int32_t* verify_tok;  // [B][K+1] (last, d_1..d_K): input of the verify forward
  int32_t* target_ids;  // [B][K+1] target argmax at each verify position
  int32_t* accept_len;  // [B]
  int32_t* n_new;       // [B]
  int32_t* new_tok;     // [B][K+1] emitted tokens, -1 padded
  int32_t* sampled;     // [B]   sampled token for position accept_len (sampling mode), else unused
  // per-row adaptive K (sd_specdec_set_adaptive): the rule of AdaptiveKController (controllers.py:63-141) per row, on
  // the device, so that the next step never waits for the host. K above stays the SHAPE of the step (max_k).
  int adaptive;         // 0 = every row proposes K
  int a_min, a_max, a_step;
  double a_hi, a_lo;    // target_acceptance_rate + 0.1 / - 0.1 (computed by the host in double, as the reference does)
  int32_t* k_row;       // [B]    proposals that count for the row in the NEXT step
  int32_t* ctl;         // [B][4] accepted so far, proposed so far, history length (<= 4), k of the step just done
  double* ctl_hist;     // [B][4] the last four reported acceptance rates, oldest first
  int32_t* k_active;    // [1]    max k_row over the active rows
  // one-sequence loops with a persistent draft: which form of draft forward 0 the NEXT step needs. {2, 1}: the 2-token
  // pass over (prev, last) — prev's K/V are not in the draft cache yet (all k proposals were accepted: d_k was never an
  // input, or the host has just set the row); {1, 2}: prev's K/V are there, the 1-token pass over `last` is enough.
  // Both passes are in the captured step; each returns at entry unless its word is > 1 (SD_SKIP_IF_INACTIVE).
  int32_t* fwd0_w;      // [2]    null: always the 2-token pass
};

int launch_draft_next(int M, int i, const SpecState& s, hipStream_t st);
int launch_draft_finalize(const float* part_val, const int* part_idx, int grid, int M, int i, int32_t* ids, const SpecState& s,
                          const int32_t* skip_k, int skip_i, hipStream_t st);
int launch_medusa_fill(const SpecState& s, hipStream_t st);
int launch_medusa_rows(const SpecState& s, int32_t* row_idx, hipStream_t st);
int launch_medusa_commit(const SpecState& s, hipStream_t st);
int launch_medusa_gather(const void* x, const int32_t* rows, void* out, int B, int d, hipStream_t st);
int launch_eagle_extrapolate(const void* x, void* H, void* prev, int32_t* has_prev, const void* norm_w, const void* norm_b,
                             float eps, float alpha, int d, int B, int K, int rms, hipStream_t st);
int launch_accept(const SpecState& s, int mode, int use_sampled, hipStream_t st);
int launch_accept_len(const SpecState& s, hipStream_t st);
int launch_pack_record(const SpecState& s, int32_t* rec_slots, int rec_ints, int32_t* step_counter, const unsigned* draft_status,
                       const unsigned* target_status, hipStream_t st);
bool verify_tail_fits(const SpecState& s);
int launch_verify_tail(const SpecState& s, const float* part_val, const int* part_idx, int grid, int mode, int32_t* rec_slots,
                       int rec_ints, int32_t* step_counter, const unsigned* draft_status, const unsigned* target_status, hipStream_t st);

// csrc/sample.hip
struct SampleArgs;
int launch_sample_step(const SpecState& s, const void* logits, int V, float temperature, int top_k, float top_p,
                       uint64_t seed, uint32_t* draw, const int32_t* stream_id, hipStream_t st);

}  // namespace sd
