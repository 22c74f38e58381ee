// Every environment variable the LIBRARY reads, in one place. None of them is part of the product's interface
// (include/specdec_hip.h): they are test hooks and measurement switches, read once (static) or at bind / create time,
// and the shipped defaults are what the measurements under profiles/ kept. Python-side knobs (SPECDEC_HIP_LIB,
// SPECDEC_NO_PACK, SPECDEC_PAGED_KV, SPECDEC_MODEL_DIR, ...) are documented where they are read (specdec_hip/*.py).
#pragma once

namespace sd {
namespace debug_env {

// ---- persistent forward (csrc/persist.hip, engine.hip) ------------------------------------------------------------------
constexpr const char* kNoPersist = "SPECDEC_NO_PERSIST";            // set: no model is eligible (every pass on the launch path). For a
                                                                     // GPU shared with other processes: the launch needs all 256 CUs.
constexpr const char* kPersistMaxT = "SPECDEC_PERSIST_MAX_T";       // tokens per persistent pass at bind time (default 2 for d_model <= 2048, else 0)
constexpr const char* kPersistTaps = "SPECDEC_PERSIST_TAPS";        // set at sd_specdec_create: the loop's draft also stores its stage rows (sd_model_debug_rows)
constexpr const char* kPersistDropWg = "SPECDEC_PERSIST_TEST_DROP_WG";  // test hook: launch one workgroup short, so that every bounded wait expires
constexpr const char* kNoFwd0Select = "SPECDEC_NO_FWD0_SELECT";     // set at sd_specdec_create: draft forward 0 is always the 2-token pass
// ---- launch path ----------------------------------------------------------------------------------------------------------
constexpr const char* kMaxPassTokens = "SPECDEC_MAX_PASS_TOKENS";   // sd_model_create: cap on tokens per pass (9 forces gemv.hip everywhere; tests)
constexpr const char* kNoDirect = "SPECDEC_NO_DIRECT";              // multi-token family: never take gemm_direct_kernel (tests of the other bodies)
constexpr const char* kNoPipe = "SPECDEC_NO_PIPE";                  //                     never take gemm_pipe_kernel
constexpr const char* kNoAttnSplit = "SPECDEC_NO_ATTN_SPLIT";       // attention: never split a row's keys over workgroups (tests of the single-workgroup form)
constexpr const char* kNoGemmPrefill = "SPECDEC_NO_GEMM_PREFILL";   // prompts always go through the 128-token passes (tests: the two prefill paths against each other)
constexpr const char* kPrefillMinTokens = "SPECDEC_PREFILL_MIN_TOKENS";   // shortest pass that takes the GEMM prefill path (default kPrefillMinTokens; read once)
constexpr const char* kMedusaPerHead = "SPECDEC_MEDUSA_PER_HEAD";   // Medusa heads: one launch per head even when they sit at a constant stride
// ---- measurement hooks (sd_model_probe_gemv) --------------------------------------------------------------------------------
constexpr const char* kGemvTimeline = "SPECDEC_GEMV_TIMELINE";      // in-kernel 100 MHz stamps of the probed launch, printed to stderr ("2": per workgroup too)
constexpr const char* kProbeHot = "SPECDEC_PROBE_HOT";              // cycle over n layers only (1: cache-resident weights)
constexpr const char* kProbeNoXstat = "SPECDEC_PROBE_NO_XSTAT";     // probed launches compute their own row statistics

}  // namespace debug_env
}  // namespace sd
