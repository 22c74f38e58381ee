// Persistent decode forward for gfx950: embedding -> n_layers x (norm+QKV+RoPE+KV append, attention, out-projection,
// norm+gate/up+SwiGLU, down-projection) -> final norm + lm_head + argmax partials as ONE launch, for passes of
// <= 8 tokens (the draft forwards of a speculative step: 1 or 2 tokens).
//
// What it replaces: the 5-launches-per-layer forward of engine.hip (itself the replacement of the reference's k-step
// draft loop, /root/reference/src/specdec/models/hf_wrappers.py:417-539, driven from src/specdec/core/pipeline.py:2397).
// With one launch per operator every kernel pays its own head (arguments, first weights in flight) and tail (reduce,
// epilogue, drain) with HBM idle, and a boundary between them: 38 us per Llama-3.2-1B layer for 121.6 MB (3.2 TB/s);
// this launch runs the same layer in ~32 us (profiles/round3_persist_ab.md).
// Here the weight stream never waits for a dependency:
//
//   * one 256-thread workgroup per CU (256 of them, all resident: each declares the CU's whole LDS), four waves with
//     fixed roles, one per SIMD:
//       wave 0  LOADER   walks the CU's share of every matrix in stream order (the packed tile streams of csrc/pack.hip:
//                        workgroup c owns a contiguous byte range of each matrix) and copies it into a byte RING in LDS
//                        with LDS-DMA (global_load_lds_dwordx4 ... nt, 1 KiB per wave-instruction, 16-piece slots, two
//                        slots in flight behind a counted vmcnt, one while this CU sweeps). It depends on nothing but
//                        ring space, so it runs AHEAD of every dependency of the layer: while the consumers wait for
//                        activations, the next ~100 KiB of weights land.
//       waves 1-3        CONSUMERS: v_mfma_f32_16x16x32_bf16 on 1-KiB A fragments read from the ring (ds_read_b128) with the
//                        staged activations as B, in chunks of 16 fragments (all LDS reads of a chunk issued before its
//                        first MFMA); the chunks of a tile go round-robin over the waves that multiply in the op and
//                        their 16x16 partials reach the leader through LDS. Besides that,
//       wave 1  LEADER   (consumer 0) folds the partial tiles, runs the fused epilogue (RoPE + KV append, residual, SwiGLU,
//                        argmax) and PUBLISHES the results to the other CUs; it multiplies only in long single-tile ops;
//       wave 2  GATHERER (consumer 1) sweeps the NEXT op's input rows from the other CUs (its loads are in flight while the
//                        leader still runs the epilogue of this op), normalises them and stages them in LDS;
//       wave 3           (consumer 2) sweeps the second half of wide rows, and every other row of a multi-token pass.
//   * activations cross CUs as 8-byte {tag, value} GRANULES (one relaxed agent-scope store each: the data is the flag,
//     no fence, no barrier): every CU publishes the <= 64 values it owns, every CU's gatherer sweeps the whole vector
//     (16 dwordx2 loads per lane per 8 KiB, all in flight) until all tags match. tag = (launch << 9) | (layer, edge), so
//     no buffer is ever re-initialised; buffers alternate with the layer parity.
//   * attention of (row, q head) units runs on B x Hq of the CUs (3 waves split the cached keys, MFMA QK^T / PV as in
//     attention_device.h, the new positions' K/V come from the granules), the other CUs wait for its output with their
//     loaders still prefetching.
//   * nothing spins unbounded: every wait checks a 50 ms deadline on the 100 MHz clock and an abort word in LDS;
//     on expiry the workgroup sets a status bit in sync[1] and leaves (the host reads it with the step record and raises).
//   * every control value a wave derives from an LDS word is made wave-uniform explicitly (readfirstlane): one wave per SIMD
//     issues an instruction every ~5 cycles, and exec-masked control flow around the wait loops cost 4 % of the forward.
//
// Numerics: the same bf16 rounding points as the launch-per-operator forward (normalised rows, q/k/v, attention rows,
// residual stream, activations and logits are bf16; sums are fp32); the order of the fp32 sums differs (3 interleaved
// K slices instead of 16/n_tiles contiguous ones), so logits agree to fp32 summation order, not bit for bit.

#include <stdlib.h>

#include "gemv_device.h"
#include "persist.h"

// ---- fixed tuning values: what the same-box A/B runs of round 3 kept (profiles/round3_persist_ab.md has every alternative that
// was measured: chunk 8 / 32, two sweep passes in flight, loader paused instead of thinned, three slots in flight, leader in the
// QKV multiply, ...). Only the diagnostic build is still a compile-time switch.
#ifndef SD_P_DIAG
#define SD_P_DIAG 0        // diagnostic build: the timeline instance times the third consumer's MFMA part in shader cycles (slots 8-11)
#endif

namespace sd {
namespace {

constexpr unsigned kPiece = 1024;                        // bytes per LDS-DMA wave-instruction
constexpr int kChunk = 16;            // MFMAs per chunk (8: +14 us per 1B forward, 32: +50 us)
constexpr int kBackoff = 2;           // s_sleep units after an incomplete sweep pass (0 / 2 / 8: no difference)
constexpr int kSplitSweep = 2048;     // plain rows wider than this many granules are swept by two waves (off: +15 us per forward)
constexpr int kLeadInUnits = 96;      // the leader multiplies in single-tile ops of at least this many MFMAs per tile (down-projection)
// Wait loops whose give-up test is forced wave-uniform, one bit per site (see expired()): 1 wait_word, 2 loader, 4 chunk loop,
// 8 granule sweeps, 16 attention sweep. All 32 combinations were measured on one box (us per 1B forward; 0: 613): 5 -> 601,
// 7 -> 592, 13 -> 587, 15 -> 592; every combination of 1 with 16 -> 718-754 (profiles/round3_persist_ab.md).
constexpr unsigned kUniformSites = 13;
constexpr unsigned kTimeoutTicks = 50u * 1000u * 100u;   // 50 ms of the 100 MHz constant clock
constexpr int kPartT = kPersistMaxT;                     // token columns kept of a partial tile

// status bits left in sync[1] by a wave that gave up
enum : unsigned { ST_LOADER = 1u, ST_LANDED = 2u, ST_USEQ = 4u, ST_PART = 8u, ST_GRANULE = 16u, ST_ATTN = 32u };

struct PCtl {   // LDS control words (all written with relaxed workgroup-scope atomics; LDS executes in order)
  unsigned landed;       // pieces of the stream that have landed in the ring
  unsigned consumed[3];  // first piece consumer w still needs
  unsigned abort_;
  unsigned u_seq;        // ops whose input rows the leader has staged
  unsigned done[3];      // tiles whose partial consumer w has written
  unsigned lead_done;    // tiles the leader has folded
  unsigned a_seq;        // attention units whose q / new k / new v are staged
  unsigned a_done[3];    // attention partials written
  unsigned a_merged;     // attention units merged
  unsigned g2_seq;       // ops whose second half of the input rows the third consumer has staged (wide rows only)
  unsigned a2_seq;       // attention units whose odd new positions the third consumer has staged (M > 1)
  unsigned nsum[2];      // gather_norm_half: sum of squares of each half row (float bits)
  unsigned nseq[2];      //              op whose half-row sum is in nsum
  unsigned gathering;    // the gatherer is sweeping: the loader keeps one slot in flight (its bursts queue in front of the sweep's loads)
};

typedef const PersistOp __attribute__((address_space(4)))* cops_t;   // uniform reads through the scalar cache

// (control words are wave-uniform by construction: readfirstlane keeps everything derived from them — loop bounds, ring
//  positions, branches — on the scalar unit; left as a per-lane value, every wait loop and the whole chunk loop of the
//  consumers became exec-masked vector control flow, ~110 instructions per chunk for a wave that issues one per ~5 cycles)
__device__ __forceinline__ unsigned lds_ld(const unsigned* p) {
  const unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
  return static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(v)));
}
__device__ __forceinline__ void lds_st(unsigned* p, unsigned v) {
  asm volatile("" ::: "memory");
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// one piece: 64 lanes x 16 bytes from sbase + voff to LDS byte address lds_dst (wave-uniform) + 16 * lane
__device__ __forceinline__ void dma_piece(const char* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// STAMPS: the diagnostic instance of the kernel (sd_model_probe_forward's timeline). In the product instance the stamps do not
// exist: as a run-time check of debug_ts at ~40 sites they cost 24 us per 1B forward (same-box A/B).
// TAPS: the instance that also leaves the last layer's stage rows in global memory (x after every residual add, q after RoPE,
// attention rows, MLP activations: sd_model_debug_rows, and the hidden rows a caller reads after a pass that skips the head).
// The draft of a speculative loop runs the instance without them: nobody reads a draft's hidden rows, and the stores with their
// 64-bit address arithmetic cost 5 us per 1B forward on the leader wave (six same-box repeats: 594.2 -> 589.2).
template <bool STAMPS, bool TAPS>
struct PCtxT {
  static constexpr bool kStamps = STAMPS;
  static constexpr bool kTaps = TAPS;
  const PersistArgs* a;
  unsigned char* smem;
  PCtl* ctl;
  unsigned long long t_start;
  int lane, cu, T;
  unsigned tag0;
  cops_t ops;
};

// (cold paths, out of line: they are referenced from every wait loop of a kernel that is one big inlined function)
__device__ __attribute__((noinline)) bool expired_slow(const unsigned* abort_word, unsigned long long t_start) {
  return lds_ld(abort_word) != 0u || static_cast<unsigned>(__builtin_amdgcn_s_memrealtime() - t_start) > kTimeoutTicks;
}
// SITE: which wait loop asks (experiment switch kUniformSites, one bit per site: 0 wait_word, 1 loader, 2 chunk loop, 3 sweeps, 4 attention sweep)
template <int SITE = 0, class C>
__device__ __forceinline__ bool expired(const C& c) {
  // (the callee's result comes back in a VGPR: without the readfirstlane every exit of every wait loop is a divergent branch,
  //  and every value carried around such a loop — ring positions, piece counts, tile numbers — leaves the scalar unit)
  const bool e = expired_slow(&c.ctl->abort_, c.t_start);
  if constexpr ((kUniformSites >> SITE) & 1) return __builtin_amdgcn_readfirstlane(static_cast<int>(e)) != 0;
  else return e;
}
__device__ __attribute__((noinline)) void give_up_slow(unsigned* abort_word, unsigned* status, unsigned* host_status, unsigned code, int lane) {
  lds_st(abort_word, 1u);
  if (lane == 0) {
    atomicOr(status, code);
    // the host's copy (pinned memory): readable without a stream synchronisation by whoever consumes the pass's outputs.
    // A plain system-scope store (PCIe atomics are not a given): any non-zero value says "invalid", the device word has the OR.
    if (host_status) __hip_atomic_store(host_status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
template <class C>
__device__ __forceinline__ void give_up(const C& c, unsigned code) { give_up_slow(&c.ctl->abort_, c.a->sync + 1, c.a->host_status, code, c.lane); }
template <class C>
__device__ __forceinline__ void stamp(const C& c, int slot) {
  if constexpr (!C::kStamps) return;
  if (c.a->debug_ts && c.lane == 0) c.a->debug_ts[static_cast<size_t>(c.cu) * (12 * c.a->n_ops + 4) + slot] = __builtin_amdgcn_s_memrealtime();
}

// wait until an LDS word reaches `need`
// (no s_sleep: every wave of the workgroup owns its SIMD, so a spinning wave costs the others nothing but LDS read slots,
//  and each hop of the intra-CU hand-overs — input staged, partial in, tile folded — sits on the layer's critical path)
template <int NAP, class C>
__device__ __forceinline__ bool wait_word(const C& c, const unsigned* p, unsigned need, unsigned code) {
  for (unsigned spins = 1; lds_ld(p) < need; ++spins) {
    if ((spins & 1023u) == 0u && expired<0>(c)) { give_up(c, code); return false; }
  }
  return true;
}

// geometry of one op for this workgroup
struct OpView {
  int kind, layer, K, steps, n_pairs, ppw, tile_pairs, p_lo, my_pairs, n_tiles;
  unsigned pair_bytes, pieces;
  const void* norm_w;
  const char* src;
  unsigned bytes;
};
template <class C>
__device__ __forceinline__ OpView load_op(const C& c, int i) {
  OpView o;
  o.kind = c.ops[i].kind;
  o.layer = c.ops[i].layer;
  o.K = c.ops[i].K;
  o.steps = o.K >> 5;
  o.n_pairs = c.ops[i].n_pairs;
  o.ppw = c.ops[i].ppw;
  o.tile_pairs = c.ops[i].tile_pairs;
  o.pair_bytes = c.ops[i].pair_bytes;
  o.norm_w = c.ops[i].norm_w;
  o.p_lo = c.cu * o.ppw;
  int my = o.n_pairs - o.p_lo;
  my = my < 0 ? 0 : (my > o.ppw ? o.ppw : my);
  o.my_pairs = my;
  o.n_tiles = (my + o.tile_pairs - 1) / o.tile_pairs;
  o.bytes = static_cast<unsigned>(my) * o.pair_bytes;
  o.pieces = (o.bytes + kPiece - 1) / kPiece;
  o.src = static_cast<const char*>(c.ops[i].W) + static_cast<size_t>(o.p_lo) * o.pair_bytes;
  return o;
}

// ------------------------------------------------------------------------------------------------ loader (wave 0)
template <class C>
__device__ __forceinline__ void loader_role(const C& c) {
  const PersistArgs& a = *c.a;
  const unsigned ring_pieces = a.ring_bytes / kPiece;
  const unsigned voff = c.lane * 16;
  unsigned issued = 0, rpos = 0;   // pieces issued; ring position (pieces)
  int op = 0;
  const char* src = nullptr;
  unsigned left = 0, tail = 0;     // pieces left in the segment; bytes of its last piece
  auto next_seg = [&]() {
    while (left == 0 && op < a.n_ops) {
      const OpView o = load_op(c, op);
      if (op > 0) stamp(c, 12 * (op - 1) + 7);   // every piece of the previous op has been issued
      ++op;
      src = o.src;
      left = o.pieces;
      tail = o.bytes - (o.pieces ? (o.pieces - 1) * kPiece : 0);
    }
  };
  next_seg();
  unsigned pub = 0;   // landed, as published
  while (left) {
    // ring space for a whole slot (the last slot of the stream may be shorter; it is drained with vmcnt(0) below)
    for (unsigned spins = 1;; ++spins) {
      const unsigned cmin = min(lds_ld(&c.ctl->consumed[0]), min(lds_ld(&c.ctl->consumed[1]), lds_ld(&c.ctl->consumed[2])));
      // (a consumer's "first piece still needed" may lie beyond what has been issued: signed distance)
      if (static_cast<int>(issued + 16u - cmin) <= static_cast<int>(ring_pieces)) break;
      if (pub != issued) {   // blocked: everything issued so far may as well land and be published
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pub = issued;
        lds_st(&c.ctl->landed, pub);
      }
      __builtin_amdgcn_s_sleep(2);
      if ((spins & 255u) == 0u && expired<1>(c)) { give_up(c, ST_LOADER); return; }
    }
    unsigned n = 0;
    if (left > 16 || (left == 16 && tail == kPiece)) {
      // a whole slot of full pieces inside one segment: the tight path (the loader must issue a slot in well under its
      // 0.64 us landing cadence; the general path below costs a segment check per piece)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        dma_piece(src, voff, a.lds_ring + rpos * kPiece);
        src += kPiece;
        rpos = (rpos + 1 == ring_pieces) ? 0u : rpos + 1;
      }
      left -= 16;
      n = 16;
      if (left == 0) next_seg();
    } else {
      for (int j = 0; j < 16 && left; ++j) {
        const unsigned dst = a.lds_ring + rpos * kPiece;
        if (left == 1 && tail < kPiece) {
          if (voff < tail) dma_piece(src, voff, dst);
        } else {
          dma_piece(src, voff, dst);
        }
        src += kPiece;
        --left;
        rpos = (rpos + 1 == ring_pieces) ? 0u : rpos + 1;
        ++n;
        if (left == 0) next_seg();
      }
    }
    issued += n;
    if (lds_ld(&c.ctl->gathering)) {   // this CU sweeps: one slot in flight (its bursts queue in front of the sweep's loads; -10 us per forward)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      pub = issued;
      lds_st(&c.ctl->landed, pub);
    } else if (n == 16 && left) {
      // every slot before this one was a full one: all but the 32 newest instructions have landed
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // two slots in flight (three: no faster, longer queues in front of the sweeps)
      constexpr unsigned kBehind = 16u;
      if (issued >= kBehind && issued - kBehind > pub) {
        pub = issued - kBehind;
        lds_st(&c.ctl->landed, pub);
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      pub = issued;
      lds_st(&c.ctl->landed, pub);
    }
  }
  stamp(c, 12 * (a.n_ops - 1) + 7);
}

// ------------------------------------------------------------------------------------------------ consumers (waves 1..3)
struct ConsState {
  unsigned piece0;      // first piece of the current op in the stream
  unsigned rbase;       // largest multiple of ring_bytes <= the byte offsets this wave is at
  unsigned landed;      // cached copy of ctl->landed
  unsigned tile_no;     // tiles done so far (all ops)
  unsigned att_no;      // attention units done on this CU
  // leader only
  float best_v;
  int best_i;
};

struct LeadLane {   // per-lane constants of the leader's epilogue item (token t = lane >> 3, pair slot jp = lane & 7)
  int t, jp, b, m, pos;
  bool tok_ok;
};

// storage offset of granule idx of an edge (see PersistArgs::gran_unit)
__device__ __forceinline__ unsigned granule_slot(const PersistArgs& a, unsigned idx) { return (idx >> 4) * a.gran_unit + (idx & 15u); }
// base of an edge's buffer for a layer; element idx lives at base[granule_slot(idx)]
__device__ __forceinline__ unsigned long long* edge_base(const PersistArgs& a, int layer, int edge) {
  return a.gran + static_cast<size_t>(layer & 1) * a.gran_parity + a.off_edge[edge];
}
__device__ __forceinline__ unsigned long long* granule_ptr(const PersistArgs& a, int layer, int edge, unsigned idx) {
  return edge_base(a, layer, edge) + granule_slot(a, idx);
}
template <class C>
__device__ __forceinline__ unsigned edge_tag(const C& c, int layer, int edge) { return c.tag0 | static_cast<unsigned>(layer * 8 + edge + 1); }
__device__ __forceinline__ void store_granule(unsigned long long* g, unsigned tag, unsigned value) {
  __hip_atomic_store(g, (static_cast<unsigned long long>(tag) << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- fused epilogue of one tile (leader): lane = (token t, pair slot jp); y0 / y1 = the two rows of the pair ---------
// Split in two: everything that does not depend on the sums (addresses, RoPE factors, the old residual) is computed BEFORE
// the leader waits for the partial tiles — a wave alone on its SIMD runs ~5 cycles per instruction, and the epilogue sits
// on the layer's critical path at every op.
struct EpiPrep {
  bool valid;
  int p;
  unsigned long long* gran;   // granule this lane publishes
  unsigned tag;
  // QKV
  bool rope;
  float2 cs;
  uint16_t* d0;               // q row / K row / V^T column of the pair's first element (null: nothing to store)
  size_t d1;                  // element offset of the pair's second element from d0
  // residual ops
  unsigned* res;
  unsigned old;
  unsigned* xtap;
  // gate/up
  unsigned* acttap;
};

template <class C>
__device__ __forceinline__ EpiPrep epilogue_prep(const C& c, const OpView& o, int tile, int np, const LeadLane& L) {
  const PersistArgs& a = *c.a;
  EpiPrep e{};
  e.valid = L.tok_ok && L.jp < np;
  e.p = o.p_lo + tile * o.tile_pairs + L.jp;
  const int p = e.valid ? e.p : o.p_lo;   // keep the address arithmetic of idle lanes in range
  if (o.kind == POP_QKV) {
    const int half = a.head_dim >> 1, Dh = a.head_dim;
    const int h = p / half, i = p - h * half;
    e.rope = h < a.n_q_heads + a.n_kv_heads && L.pos >= 0 && L.pos < a.max_pos;
    e.cs = reinterpret_cast<const float2*>(c.smem + a.lds_rope)[tile * 64 + c.lane];
    e.gran = granule_ptr(a, o.layer, PE_QKV, static_cast<unsigned>(L.t) * o.n_pairs + p);
    e.tag = edge_tag(c, o.layer, PE_QKV);
    if (h < a.n_q_heads) {
      if (C::kTaps) e.d0 = static_cast<uint16_t*>(a.q) + static_cast<size_t>(L.t) * a.n_q_heads * Dh + h * Dh + i;
      e.d1 = half;
    } else if (L.pos >= 0 && L.pos < a.l_max) {
      // in-place KV append, as epilogue<EPI_QKV_ROPE> (gemv_device.h): K rows [l_max][D], V transposed [D][l_max]
      const size_t lbase = static_cast<size_t>(o.layer) * a.layer_kv;
      if (h < a.n_q_heads + a.n_kv_heads) {
        const int kvh = h - a.n_q_heads;
        e.d0 = static_cast<uint16_t*>(a.k_cache) + lbase + ((static_cast<size_t>(L.b) * a.n_kv_heads + kvh) * a.l_max + L.pos) * Dh + i;
        e.d1 = half;
      } else {
        const int kvh = h - a.n_q_heads - a.n_kv_heads;
        e.d0 = static_cast<uint16_t*>(a.v_cache) + lbase + ((static_cast<size_t>(L.b) * a.n_kv_heads + kvh) * Dh + i) * a.l_max + L.pos;
        e.d1 = static_cast<size_t>(half) * a.l_max;
      }
    }
  } else if (o.kind == POP_OUT || o.kind == POP_DOWN) {
    e.res = reinterpret_cast<unsigned*>(c.smem + a.lds_resid) + (e.valid ? L.t * a.resid_ppw + tile * o.tile_pairs + L.jp : 0u);
    e.old = *e.res;
    const int edge = (o.kind == POP_OUT) ? PE_X2 : PE_X;
    const int lay = (o.kind == POP_OUT) ? o.layer : o.layer + 1;   // the down-projection feeds the NEXT layer (or the head)
    e.gran = granule_ptr(a, lay, edge, static_cast<unsigned>(L.t) * o.n_pairs + p);
    e.tag = edge_tag(c, lay, edge);
    e.xtap = reinterpret_cast<unsigned*>(static_cast<uint16_t*>(a.x) + static_cast<size_t>(L.t) * a.d_model) + p;
  } else if (o.kind == POP_GATEUP) {
    e.gran = granule_ptr(a, o.layer, PE_ACT, static_cast<unsigned>(L.t) * (o.n_pairs >> 1) + (p >> 1));
    e.tag = edge_tag(c, o.layer, PE_ACT);
    e.acttap = reinterpret_cast<unsigned*>(static_cast<uint16_t*>(a.act) + static_cast<size_t>(L.t) * a.d_ff) + (p >> 1);
  }
  return e;
}

// dbl: the tile was multiplied two steps at a time (consume_op): real row r of the tile = MFMA rows r (even k groups, token
// column 2 t, stored as partial row r) + r + 8 (odd k groups, column 2 t + 1, stored as partial row r + 8)
template <int W_FIRST, bool DBL, class C>
__device__ __forceinline__ void epilogue_finish(const C& c, const OpView& o, const EpiPrep& e, const float* part, const LeadLane& L, ConsState& st) {
  const PersistArgs& a = *c.a;
  float y0 = 0.f, y1 = 0.f;
  if (e.valid) {
#pragma unroll
    for (int w = W_FIRST; w < 3; ++w) {   // fixed order: deterministic
      const float* pw = part + w * 16 * kPartT + L.t;
      if constexpr (DBL) {
        y0 += pw[L.jp * kPartT] + pw[(L.jp + 8) * kPartT];
        y1 += pw[(L.jp + 4) * kPartT] + pw[(L.jp + 12) * kPartT];
      } else {
        y0 += pw[L.jp * kPartT];
        y1 += pw[(L.jp + 8) * kPartT];
      }
    }
  }
  if (o.kind == POP_QKV) {
    if (e.valid) {
      float o0 = y0, o1 = y1;
      if (e.rope) {
        o0 = y0 * e.cs.x - y1 * e.cs.y;
        o1 = y1 * e.cs.x + y0 * e.cs.y;
      }
      const uint16_t u0 = float_to_bf16_bits(o0), u1 = float_to_bf16_bits(o1);
      store_granule(e.gran, e.tag, static_cast<unsigned>(u0) | (static_cast<unsigned>(u1) << 16));
      if (e.d0) {
        e.d0[0] = u0;
        e.d0[e.d1] = u1;
      }
    }
  } else if (o.kind == POP_OUT || o.kind == POP_DOWN) {
    if (e.valid) {
      const float n0 = __uint_as_float(e.old << 16) + y0;
      const float n1 = __uint_as_float(e.old & 0xffff0000u) + y1;
      const unsigned nv = static_cast<unsigned>(float_to_bf16_bits(n0)) | (static_cast<unsigned>(float_to_bf16_bits(n1)) << 16);
      store_granule(e.gran, e.tag, nv);
      *e.res = nv;
      if (C::kTaps) *e.xtap = nv;
    }
  } else if (o.kind == POP_GATEUP) {
    unsigned u = 0;
    if (e.valid) u = float_to_bf16_bits(y0 / (1.0f + __expf(-y0)) * y1);
    const unsigned partner = __shfl_xor(u, 1, 64);
    if (e.valid && (L.jp & 1) == 0) {
      const unsigned v = u | (partner << 16);
      store_granule(e.gran, e.tag, v);
      if (C::kTaps) *e.acttap = v;
    }
  } else {   // POP_HEAD: logits are the bf16-rounded products (epilogue<EPI_ARGMAX>)
    if (e.valid) {
      const int r0 = 2 * e.p, r1 = 2 * e.p + 1;
      const uint16_t u0 = float_to_bf16_bits(y0), u1 = float_to_bf16_bits(y1);
      const float f0 = bf16_bits_to_float(u0), f1 = bf16_bits_to_float(u1);
      if (argmax_better(f0, r0, st.best_v, st.best_i)) { st.best_v = f0; st.best_i = r0; }
      if (r1 < a.vocab && argmax_better(f1, r1, st.best_v, st.best_i)) { st.best_v = f1; st.best_i = r1; }
      if (a.logits) {
        if (a.logits_dtype == SD_F32) {
          float* lo = static_cast<float*>(a.logits) + static_cast<size_t>(L.t) * a.logits_stride;
          lo[r0] = f0;
          if (r1 < a.vocab) lo[r1] = f1;
        } else {
          uint16_t* lo = static_cast<uint16_t*>(a.logits) + static_cast<size_t>(L.t) * a.logits_stride;
          lo[r0] = u0;
          if (r1 < a.vocab) lo[r1] = u1;
        }
      }
    }
  }
}

// The MFMA part of one op, for consumer cw (0 = leader, 1 = gatherer, 2 = plain). Tiles of <= 8 pairs. A wave alone on its
// SIMD hides nothing: every wait, branch and address computation is paid in full (4-step groups with a wait each ran at
// 650-1000 cycles per group, measured). So the unit of work is a CHUNK: 16 LDS reads issued back to back, then 8 MFMAs on
// two alternating accumulators, and one flag update; the chunks of a tile go round-robin over the participating waves.
//   * Tiles of 4 pairs (the d_model-wide matrices at 256 workgroups: out- and down-projection) fill only 8 of the 16 MFMA
//     rows. They are multiplied TWO steps per MFMA: a lane's 16 bytes are chunk `lane` of the 1-KiB double step, which puts
//     (step 2j + g/2, k group 2 (g % 2)) of row n under MFMA row n < 8 and k group 2 (g % 2) + 1 of row n - 8 under MFMA
//     row n >= 8; token t supplies its even-group activations as B column 2 t and the odd groups as column 2 t + 1, so
//     D[r][2t] + D[r + 8][2t + 1] is row r's sum (the cross terms are never read). Half the MFMAs and LDS reads per byte.
//   * The leader multiplies only in long single-tile ops (>= 96 MFMAs: the down-projection); in short ops it is worth more
//     with its epilogue operands ready when the partials arrive, in ops of several tiles (gate/up, lm_head) its epilogue of
//     tile i runs while the other two are in tile i + 1.
// Returns false when the wave gave up.

// one chunk: UB / XS = byte strides of the weight fragments / activation fragments (0: run-time values ub / xs)
struct ChunkDiag { unsigned long long reads, mfma, wait, spins, chunks, other; };

// WRAP: the chunk may run past the end of the ring: every lane wraps its own address (min(a, a - ring) on unsigned values)
template <unsigned UB, unsigned XS, bool WRAP, bool DIAG = false>
__device__ __forceinline__ void chunk_mfma(const unsigned char* ringp, unsigned woff, unsigned ring, const unsigned char* xb, unsigned ub, unsigned xs,
                                           f32x4_t& acc0, f32x4_t& acc1, ChunkDiag* dg = nullptr) {
  u32x4 wf[kChunk], xf[kChunk];
  unsigned long long t0 = 0;
  if constexpr (DIAG) t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int j = 0; j < kChunk; ++j) {
    unsigned ra = woff + static_cast<unsigned>(j) * (UB ? UB : ub);
    if constexpr (WRAP) ra = min(ra, ra - ring);
    wf[j] = *reinterpret_cast<const u32x4*>(ringp + ra);
    xf[j] = *reinterpret_cast<const u32x4*>(xb + static_cast<unsigned>(j) * (XS ? XS : xs));
  }
  // all 16 reads are issued before the first MFMA (hipcc otherwise keeps two pairs in flight and exposes the LDS latency
  // eight times per chunk); the waits become counted lgkmcnt(N)
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t1 = 0;
  if constexpr (DIAG) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int j = 0; j < kChunk; j += 2) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[j]), __builtin_bit_cast(bf16x8_t, xf[j]), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[j + 1]), __builtin_bit_cast(bf16x8_t, xf[j + 1]), acc1, 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (DIAG) {
    asm volatile("s_nop 0" :: "v"(acc0[0]), "v"(acc1[0]));   // the sums have to be there
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    dg->reads += t1 - t0;
    dg->mfma += t2 - t1;
    dg->chunks += 1;
  }
}

// NP > 0: every tile of the op has exactly NP pairs (8: gate/up, lm_head; 6: Llama-3.2-1B QKV; 4: out- / down-projection,
// double-stepped) and a whole number of chunks — the tile geometry is then a compile-time constant and the prologue of a wave's
// MFMA part (1000 shader cycles from entry to the first LDS read in the generic form, measured; it sits on the critical path of
// every op) shrinks to the ring position. NP = 0: any geometry (partial last tiles, other models).
template <int NP = 0, bool DIAG = false, class C>
__device__ __forceinline__ bool consume_op(const C& c, int cw, const OpView& o, ConsState& st, const LeadLane& L, int ts_mfma, unsigned useq_need) {
  // ONE copy of this routine for the three consumers (the leader's parts under a wave-uniform branch): the launch's code must
  // stay inside the 64 KiB instruction cache two CUs share — with a copy per role (and per call site of the gathers) the
  // kernel was ~150 KiB and every phase of every wave started on instructions fetched from memory
  const bool LEAD = cw == 0;
  const PersistArgs& a = *c.a;
  ChunkDiag dg{};
  unsigned long long d_t = 0, d_loop0 = 0, d_first = 0, d_end = 0;
  if constexpr (DIAG) d_loop0 = __builtin_amdgcn_s_memtime();
  const int lane = c.lane, g = lane >> 4, n = lane & 15;
  const unsigned ring = a.ring_bytes;
  const unsigned char* ringp = c.smem + a.lds_ring;
  const unsigned char* ubase = c.smem + a.lds_u;
  const unsigned char* xrow_std = ubase + static_cast<unsigned>(n < c.T ? n : 0) * a.u_stride + g * 16;
  const unsigned char* xrow_dbl = ubase + static_cast<unsigned>((n >> 1) < c.T ? (n >> 1) : 0) * a.u_stride + g * 32 + (n & 1) * 16;
  float* part_all = reinterpret_cast<float*>(c.smem + a.lds_part);
  const unsigned op_abs0 = st.piece0 * kPiece;
  const bool multi = o.n_tiles > 1;
  const bool dbl0 = NP ? NP == 4 : (!multi && o.my_pairs == 4 && (o.steps & 1) == 0);
  const int units0 = dbl0 ? o.steps >> 1 : o.steps;
  const bool lead_in = !multi && units0 >= kLeadInUnits;
  const int share = lead_in ? 3 : 2;
  // (when it does multiply, the leader takes the short share: chunk 2, 5, ... of a tile)
  const int first_chunk = lead_in ? (cw + 2) % 3 : cw - 1;   // -1: this wave (the leader) does not multiply in this op
  if (LEAD && !lead_in) lds_st(&c.ctl->consumed[0], st.piece0 + o.pieces);   // never reads this op's weights
  for (int tile = 0; tile < o.n_tiles; ++tile) {
    const int np = NP ? NP : min(o.tile_pairs, o.my_pairs - tile * o.tile_pairs);
    const bool dbl = NP ? NP == 4 : (np == 4 && (o.steps & 1) == 0);
    const unsigned sb = static_cast<unsigned>(np) * 128u;   // bytes of one 32-k step of the tile
    const unsigned ub = dbl ? 1024u : sb;                   // bytes of one MFMA's worth (a "unit")
    const int units = dbl ? o.steps >> 1 : o.steps;
    const int n_chunk = (units + kChunk - 1) / kChunk;
    const unsigned tile_off = static_cast<unsigned>(tile * o.tile_pairs) * o.pair_bytes;
    int jp = n & 7, second = n >> 3;
    if (jp >= np) { jp = 0; second = 0; }   // lanes without a pair alias the first row (their outputs are never read)
    const unsigned lane_off = dbl ? static_cast<unsigned>(lane) * 16u : static_cast<unsigned>(g * 2 * np + second * np + jp) * 16u;   // + 16 <= ub
    const unsigned char* xrow = dbl ? xrow_dbl : xrow_std;
    const unsigned xstep = dbl ? 128u : 64u;
    EpiPrep prep{};
    if (LEAD) {
      prep = epilogue_prep(c, o, tile, np, L);
      // The leader enters the op without waiting for the staged rows: its epilogue operands (addresses, RoPE factors, the old
      // residual) do not depend on them, so they are computed while the gatherer still sweeps; only when it multiplies itself
      // (long single-tile ops) does it need the rows — the other consumers waited for them before they came here.
      if (tile == 0 && lead_in && !wait_word<1>(c, &c.ctl->u_seq, useq_need, ST_USEQ)) return false;
    }
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int ch = first_chunk; ch >= 0 && ch < n_chunk; ch += share) {
      const unsigned off = tile_off + static_cast<unsigned>(ch * kChunk) * ub;
      const int ns = NP ? kChunk : min(kChunk, units - ch * kChunk);
      const unsigned need = st.piece0 + ((off + static_cast<unsigned>(ns) * ub + kPiece - 1) / kPiece);
      if constexpr (DIAG) { d_t = __builtin_amdgcn_s_memtime(); if (!d_first) d_first = d_t; }
      if (st.landed < need) {
        for (unsigned spins = 1;; ++spins) {
          st.landed = lds_ld(&c.ctl->landed);
          if constexpr (DIAG) dg.spins += 1;
          if (st.landed >= need) break;
          if ((spins & 1023u) == 0u && expired<2>(c)) { give_up(c, ST_LANDED); return false; }
        }
      }
      if constexpr (DIAG) dg.wait += __builtin_amdgcn_s_memtime() - d_t;
      const unsigned abs0 = op_abs0 + off;
      while (abs0 - st.rbase >= ring) st.rbase += ring;
      const unsigned pos = abs0 - st.rbase;
      const unsigned char* xb = xrow + static_cast<unsigned>(ch * kChunk) * xstep;
      if (ns == kChunk) {
        const unsigned woff = pos + lane_off;
        const bool wraps = pos + kChunk * ub > ring;
        // 1-KiB units (8-pair tiles and double-stepped 4-pair tiles: gate/up, lm_head, out- and down-projection = 90 % of the
        // bytes): compile-time strides, so the reads carry immediate offsets and cost no address arithmetic. A chunk that runs
        // past the end of the ring (one in eight at 16 KiB) wraps per lane — still all reads first (as a serial loop of
        // read, wait, MFMA such a chunk cost 3200 cycles against 700, and doubled the time of every op; measured)
        if (ub == 1024u && !dbl && !wraps) chunk_mfma<1024, 64, false, DIAG>(ringp, woff, ring, xb, 0u, 0u, acc0, acc1, &dg);
        else if (ub == 1024u && dbl && !wraps) chunk_mfma<1024, 128, false, DIAG>(ringp, woff, ring, xb, 0u, 0u, acc0, acc1, &dg);
        else if (ub == 768u && !wraps) chunk_mfma<768, 64, false, DIAG>(ringp, woff, ring, xb, 0u, 0u, acc0, acc1, &dg);   // 6-pair tiles: Llama-3.2-1B QKV
        else chunk_mfma<0, 0, true, DIAG>(ringp, woff, ring, xb, ub, xstep, acc0, acc1, &dg);
      } else {   // the short last chunk of a tile whose steps are not a multiple of the chunk
        for (int j = 0; j < ns; ++j) {
          unsigned ra = pos + static_cast<unsigned>(j) * ub + lane_off;
          ra = (ra >= ring) ? ra - ring : ra;
          const u32x4 wf = *reinterpret_cast<const u32x4*>(ringp + ra);
          const u32x4 xf = *reinterpret_cast<const u32x4*>(xb + static_cast<unsigned>(j) * xstep);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, xf), acc0, 0, 0, 0);
        }
      }
      // first piece this wave still needs: its next chunk of this tile, or the end of the tile (the next tile / op starts there)
      const unsigned nxt_off = (ch + share < n_chunk) ? off + static_cast<unsigned>(share * kChunk) * ub : tile_off + static_cast<unsigned>(o.steps) * sb;
      lds_st(&c.ctl->consumed[cw], st.piece0 + nxt_off / kPiece);
    }
    if (tile == o.n_tiles - 1 && ts_mfma >= 0) stamp(c, ts_mfma);   // diagnostic: this wave's MFMA part of the op is done
    if constexpr (DIAG) d_end = __builtin_amdgcn_s_memtime();
    // ---- hand the partial to the leader (double-buffered by tile parity)
    if (st.tile_no >= 2 && !wait_word<1>(c, &c.ctl->lead_done, st.tile_no - 1, ST_PART)) return false;
    float* part = part_all + (st.tile_no & 1u) * (3 * 16 * kPartT);
    if (first_chunk >= 0) {
      if (dbl) {   // MFMA rows < 8 count in the even token columns, rows >= 8 in the odd ones
        if (((g < 2) == ((n & 1) == 0)) && (n >> 1) < c.T) {
#pragma unroll
          for (int q = 0; q < 4; ++q) part[(cw * 16 + 4 * g + q) * kPartT + (n >> 1)] = acc0[q] + acc1[q];
        }
      } else if (n < c.T) {
#pragma unroll
        for (int q = 0; q < 4; ++q) part[(cw * 16 + 4 * g + q) * kPartT + n] = acc0[q] + acc1[q];
      }
    }
    ++st.tile_no;
    lds_st(&c.ctl->done[cw], st.tile_no);
    if constexpr (DIAG) {
      if (tile == o.n_tiles - 1 && ts_mfma >= 0 && c.lane == 0) {   // slots 8..11 of the op, shader cycles
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
        unsigned long long* d = c.a->debug_ts + static_cast<size_t>(c.cu) * (12 * c.a->n_ops + 4) + ts_mfma + 3;
        if (!d_first) d_first = d_end;
        d[0] = d_first - d_loop0;                                   // prologue: entry -> first chunk
        d[1] = dg.reads + dg.mfma;                                  // LDS reads + MFMAs
        d[2] = dg.wait;                                             // waiting for weights to land
        d[3] = (t3 - d_first) - (dg.reads + dg.mfma + dg.wait);     // everything else inside the chunk loop + partial tile + done flag
      }
    }
    if (LEAD) {
      if (!wait_word<1>(c, &c.ctl->done[1], st.tile_no, ST_PART) || !wait_word<1>(c, &c.ctl->done[2], st.tile_no, ST_PART)) return false;
      if (!SD_P_DIAG && tile == o.n_tiles - 1 && ts_mfma >= 0) stamp(c, ts_mfma + 3);   // diagnostic (slot 9): the last tile's partials are in
      if (lead_in) {
        if (dbl) epilogue_finish<0, true>(c, o, prep, part, L, st);
        else epilogue_finish<0, false>(c, o, prep, part, L, st);
      } else {
        if (dbl) epilogue_finish<1, true>(c, o, prep, part, L, st);
        else epilogue_finish<1, false>(c, o, prep, part, L, st);
      }
      lds_st(&c.ctl->lead_done, st.tile_no);
    }
  }
  st.piece0 += o.pieces;
  lds_st(&c.ctl->consumed[cw], st.piece0);
  return true;
}

// ---- gatherer (consumer 1): sweep granules ----------------------------------------------------------------------------
// The staged rows of the previous op may still be read by the other two consumers' MFMAs: they are free once both have
// handed in the partial of the op's last tile. The sweep's loads are issued BEFORE this wait.
template <class C>
__device__ __forceinline__ bool wait_rows_free(const C& c, const ConsState& st, int me = 1) {
  const int o1 = (me + 1) % 3, o2 = (me + 2) % 3;
  return wait_word<1>(c, &c.ctl->done[o1], st.tile_no, ST_PART) && wait_word<1>(c, &c.ctl->done[o2], st.tile_no, ST_PART);
}

// NC chunks of 1024 granules (index k * 1024 + j * 64 + lane), all 16 * NC loads of a lane in flight: ONE round trip per
// pass; re-read until every tag matches. The loads are UNCONDITIONAL (lanes past `count` re-read the last granule): a load
// under `if (idx < count)` is compiled into its own exec-masked block with a vmcnt(0) behind it, i.e. 16 * NC serial
// round trips of 0.4 us (measured: 7 us per 8 KiB vector, 26 us per 32 KiB).
// TWO passes are kept in flight, half a round trip apart: a pass that was issued just before the last producer's granule
// became visible comes back incomplete, and with one pass at a time the next one only starts then (a full round trip, ~1.2 us,
// lost on most edges; measured 3.7 us from the last publish to the staged vector for an 8 KiB edge).
template <int NC, int LPC = 16, class C>
__device__ __forceinline__ bool sweep(const C& c, const unsigned long long* base, unsigned first, int count, unsigned tag, unsigned (&v)[NC][LPC]) {
  const PersistArgs& a = *c.a;
  const unsigned long long* p[NC][LPC];
#pragma unroll
  for (int k = 0; k < NC; ++k)
#pragma unroll
    for (int j = 0; j < LPC; ++j) {
      const int idx = k * 1024 + j * 64 + c.lane;
      p[k][j] = base + granule_slot(a, first + static_cast<unsigned>(idx < count ? idx : count - 1));
    }
  unsigned long long xa[NC][LPC];
  auto issue = [&](unsigned long long (&x)[NC][LPC]) {
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
      for (int j = 0; j < LPC; ++j) x[k][j] = __hip_atomic_load(p[k][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto complete = [&](const unsigned long long (&x)[NC][LPC]) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
      for (int j = 0; j < LPC; ++j) {
        v[k][j] = static_cast<unsigned>(x[k][j]);
        ok &= static_cast<unsigned>(x[k][j] >> 32) == tag;
      }
    return __all(ok) != 0;
  };
  for (unsigned spins = 1;; ++spins) {
    issue(xa);
    if (complete(xa)) return true;
    __builtin_amdgcn_s_sleep(kBackoff);
    if ((spins & 255u) == 0u && expired<3>(c)) { give_up(c, ST_GRANULE); return false; }
  }
}

// input rows of a norm-fused op (QKV, GATEUP, HEAD): gather the d_model-wide rows (granules of edge `edge`, or the
// embedding rows for layer 0), RMSNorm them (HF LlamaRMSNorm: weight * (x * rsqrt(mean(x^2) + eps)).to(bf16)), stage as bf16
template <int HC, class C>
__device__ __forceinline__ bool gather_norm_rows(const C& c, const OpView& o, int edge, bool from_embedding, const ConsState& st, int ts, int t0, int tstep,
                                                 int me) {
  const PersistArgs& a = *c.a;
  const int npt = a.d_model >> 1;   // dwords (pairs) per row
  const unsigned* nw = static_cast<const unsigned*>(o.norm_w);
  unsigned wv[HC][16];
#pragma unroll
  for (int hc = 0; hc < HC; ++hc)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int idx = hc * 1024 + j * 64 + c.lane;
      wv[hc][j] = nw[idx < npt ? idx : npt - 1];   // unconditional (clamped): the loads stay back to back
    }
  for (int t = t0; t < c.T; t += tstep) {   // rows t0, t0 + tstep, ...: with several tokens two waves share the rows
    unsigned v[HC][16];
    if (from_embedding) {
      const int b = t / a.M, m = t - b * a.M;
      int tok = a.tokens[b * a.tok_stride + m];
      tok = tok < 0 ? 0 : (tok >= a.vocab ? a.vocab - 1 : tok);   // validate_and_clamp_tokens (token_validation.py:15-78)
      const unsigned* row = reinterpret_cast<const unsigned*>(static_cast<const uint16_t*>(a.tok_emb) + static_cast<size_t>(tok) * a.d_model);
#pragma unroll
      for (int hc = 0; hc < HC; ++hc)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int idx = hc * 1024 + j * 64 + c.lane;
          v[hc][j] = row[idx < npt ? idx : npt - 1];
        }
#pragma unroll
      for (int hc = 0; hc < HC; ++hc)
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (hc * 1024 + j * 64 + c.lane >= npt) v[hc][j] = 0u;
    } else {
      if (!sweep<HC>(c, edge_base(a, o.layer, edge), static_cast<unsigned>(t) * npt, npt, edge_tag(c, o.layer, edge), v)) return false;
#pragma unroll
      for (int hc = 0; hc < HC; ++hc)
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (hc * 1024 + j * 64 + c.lane >= npt) v[hc][j] = 0u;   // clamped duplicates do not count in the statistic
    }
    f32x2_t s2 = {0.f, 0.f};
#pragma unroll
    for (int hc = 0; hc < HC; ++hc)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const f32x2_t f = bf16x2_unpack(v[hc][j]);
        s2 += f * f;
      }
    const float sq = wave_reduce_sum(s2.x + s2.y);
    const float rs = rsqrtf(sq / static_cast<float>(a.d_model) + a.norm_eps);
    if (t == t0) { if (!SD_P_DIAG && ts >= 0) stamp(c, ts + 8); if (!wait_rows_free(c, st, me)) return false; }
    unsigned* dst = reinterpret_cast<unsigned*>(c.smem + a.lds_u + static_cast<unsigned>(t) * a.u_stride);
#pragma unroll
    for (int hc = 0; hc < HC; ++hc)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int idx = hc * 1024 + j * 64 + c.lane;
        if (idx < npt) dst[idx] = rmsnorm_pair(v[hc][j], rs, wv[hc][j]);
      }
  }
  return true;
}

// 1-token passes: half of the row per wave (me = 1: first half, 2: second half). The post-sweep arithmetic of a
// whole row (statistic, two roundings per element, 16 LDS stores per lane) is ~1.1 us on one wave and sits on the critical
// path of three edges per layer; the two halves exchange their sums of squares through LDS (first half + second half in both
// waves, so both normalise with the same factor).
template <int HC, class C>
__device__ __forceinline__ bool gather_norm_half(const C& c, const OpView& o, int edge, bool from_embedding, const ConsState& st, int ts, int me, unsigned seq) {
  const PersistArgs& a = *c.a;
  constexpr int LP = HC == 1 ? 8 : 16;   // loads per lane for half a row (d_model <= 2048: <= 512 granules)
  const int npt = a.d_model >> 1, half = npt >> 1, c0 = (me == 2) ? half : 0;
  const unsigned* nw = static_cast<const unsigned*>(o.norm_w) + c0;
  unsigned wv[LP];
#pragma unroll
  for (int j = 0; j < LP; ++j) {
    const int idx = j * 64 + c.lane;
    wv[j] = nw[idx < half ? idx : half - 1];
  }
  unsigned v[1][LP];
  if (from_embedding) {
    int tok = a.tokens[0];
    tok = tok < 0 ? 0 : (tok >= a.vocab ? a.vocab - 1 : tok);   // validate_and_clamp_tokens (token_validation.py:15-78)
    const unsigned* row = reinterpret_cast<const unsigned*>(static_cast<const uint16_t*>(a.tok_emb) + static_cast<size_t>(tok) * a.d_model) + c0;
#pragma unroll
    for (int j = 0; j < LP; ++j) {
      const int idx = j * 64 + c.lane;
      v[0][j] = row[idx < half ? idx : half - 1];
    }
  } else {
    if (!sweep<1, LP>(c, edge_base(a, o.layer, edge), static_cast<unsigned>(c0), half, edge_tag(c, o.layer, edge), v)) return false;
  }
  f32x2_t s2 = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < LP; ++j) {
    if (j * 64 + c.lane >= half) v[0][j] = 0u;   // clamped duplicates do not count in the statistic
    const f32x2_t f = bf16x2_unpack(v[0][j]);
    s2 += f * f;
  }
  const float mine = wave_reduce_sum(s2.x + s2.y);
  lds_st(&c.ctl->nsum[me - 1], __float_as_uint(mine));
  lds_st(&c.ctl->nseq[me - 1], seq);
  if (!wait_word<1>(c, &c.ctl->nseq[2 - me], seq, ST_USEQ)) return false;
  const float other = __uint_as_float(lds_ld(&c.ctl->nsum[2 - me]));
  const float sq = (me == 1) ? mine + other : other + mine;
  const float rs = rsqrtf(sq / static_cast<float>(a.d_model) + a.norm_eps);
  if (!SD_P_DIAG && ts >= 0) stamp(c, ts + 8);
  if (!wait_rows_free(c, st, me)) return false;
  unsigned* dst = reinterpret_cast<unsigned*>(c.smem + a.lds_u) + c0;
#pragma unroll
  for (int j = 0; j < LP; ++j) {
    const int idx = j * 64 + c.lane;
    if (idx < half) dst[idx] = rmsnorm_pair(v[0][j], rs, wv[j]);
  }
  return true;
}

// input rows taken as they are (attention rows for the out-projection, activations for the down-projection)
template <int NC, class C>
__device__ __forceinline__ bool gather_plain_chunks(const C& c, const unsigned long long* g, unsigned first, int count, unsigned tag, unsigned* dst,
                                                    int dst0, bool& first_write, const ConsState& st, int me) {
  unsigned v[NC][16];
  if (!sweep<NC>(c, g, first, count, tag, v)) return false;
  if (first_write && !wait_rows_free(c, st, me)) return false;
  first_write = false;
#pragma unroll
  for (int k = 0; k < NC; ++k)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int idx = k * 1024 + j * 64 + c.lane;
      if (idx < count) dst[dst0 + idx] = v[k][j];
    }
  return true;
}

// granules [lo, hi) of every token row (npt per row); wave `me` of the three consumers does the sweeping
template <class C>
__device__ __forceinline__ bool gather_plain_rows(const C& c, int layer, int edge, int npt, int lo, int hi, const ConsState& st, int me, int t0, int tstep) {
  const PersistArgs& a = *c.a;
  const unsigned tag = edge_tag(c, layer, edge);
  const unsigned long long* g = edge_base(a, layer, edge);
  bool first_write = true;
  for (int t = t0; t < c.T; t += tstep) {
    unsigned* dst = reinterpret_cast<unsigned*>(c.smem + a.lds_u + static_cast<unsigned>(t) * a.u_stride);
    for (int c0 = lo; c0 < hi;) {
      const int left = hi - c0;
      const unsigned first = static_cast<unsigned>(t * npt + c0);
      bool ok;
      if (left > 1024) { ok = gather_plain_chunks<2>(c, g, first, left < 2048 ? left : 2048, tag, dst, c0, first_write, st, me); c0 += 2048; }
      else { ok = gather_plain_chunks<1>(c, g, first, left, tag, dst, c0, first_write, st, me); c0 += 1024; }
      if (!ok) return false;
    }
  }
  return true;
}
// rows wider than this many granules are swept by two waves (the gatherer and the third consumer, half each): a 64-load
// pass of one wave took 3 us, and a pass that starts before the last producer has published is a pass lost

// ---- attention of one (row b, q head h) unit, by the three consumer waves of its CU ---------------------------------
// LDS scratch at lds_attn: q_s [M][D] bf16 | k_s [M][D] | vT_s [D][8] (V of the new positions, transposed, zero beyond M) |
//                          o_s [3][M][D] f32 | m_s [3][8] | l_s [3][8]
// The gatherer sweeps q / new k / new v of the unit; every wave has the K / V operands of its first cached block in flight
// before it waits for them; the cached 32-key blocks go round-robin over the waves, the block of the M new positions (from
// LDS) to the wave whose turn it is; the leader merges the three partials and publishes the rows.
template <int D, class C>
__device__ __forceinline__ bool attention_unit(const C& c, int cw, int layer, int b, int h, ConsState& st, int ts) {
  const PersistArgs& a = *c.a;
  constexpr int NKS = D / 32, NDT = D / 16;
  const int lane = c.lane, g = lane >> 4, n = lane & 15;
  const int M = a.M, Hq = a.n_q_heads, Hkv = a.n_kv_heads, G = Hq / Hkv, kvh = h / G, half = D / 2;
  uint16_t* q_s = reinterpret_cast<uint16_t*>(c.smem + a.lds_attn);
  uint16_t* k_s = q_s + M * D;
  uint16_t* vT_s = k_s + M * D;
  float* o_s = reinterpret_cast<float*>(vT_s + 8 * D);
  float* m_s = o_s + 3 * M * D;
  float* l_s = m_s + 3 * 8;
  const int pos0 = max(0, min(a.pos_base[b] + a.pos_off, a.l_max));   // cached keys [0, pos0); new keys pos0 + m
  const unsigned unit_no = st.att_no + 1;
  const int nb_old = (pos0 + 31) >> 5;

  const size_t lbase = static_cast<size_t>(layer) * a.layer_kv;
  const uint16_t* kc = static_cast<const uint16_t*>(a.k_cache) + lbase + (static_cast<size_t>(b) * Hkv + kvh) * a.l_max * D;
  const uint16_t* vt = static_cast<const uint16_t*>(a.v_cache) + lbase + (static_cast<size_t>(b) * Hkv + kvh) * D * a.l_max;
  auto load_block = [&](int blk, u32x4 (&kf)[2][NKS], u32x4 (&vf)[NDT]) {
    const int key0 = blk * 32;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int key = key0 + 8 * (n >> 2) + 4 * u + (n & 3);
      if (key >= a.l_max) key = a.l_max - 1;
      const uint16_t* krow = kc + static_cast<size_t>(key) * D + g * 8;
#pragma unroll
      for (int s = 0; s < NKS; ++s) kf[u][s] = *reinterpret_cast<const u32x4*>(krow + s * 32);
    }
    int kofs = key0 + 8 * g;
    if (kofs + 8 > a.l_max) kofs = a.l_max - 8;
    const uint16_t* vrow = vt + static_cast<size_t>(n) * a.l_max + kofs;
#pragma unroll
    for (int i = 0; i < NDT; ++i) vf[i] = *reinterpret_cast<const u32x4*>(vrow + static_cast<size_t>(16 * i) * a.l_max);
  };
  u32x4 kf0[2][NKS], vf0[NDT];
  if (cw < nb_old) load_block(cw, kf0, vf0);   // in flight while q is on its way
  u32x4 kf1[2][NKS], vf1[NDT];
  const bool second = cw + 3 < nb_old;   // a wave's second cached block is in flight before it computes its first
  if (second) load_block(cw + 3, kf1, vf1);
  // the scratch is rewritten (this unit's sweep and partials) only after the leader has merged the previous unit
  if (cw != 0 && st.att_no > 0 && !wait_word<1>(c, &c.ctl->a_merged, st.att_no, ST_ATTN)) return false;

  const bool pos_shared = M > 1;   // several new positions: even ones swept by the gatherer, odd ones by the third consumer
  if (cw == 1 || (cw == 2 && pos_shared)) {
    if (cw == 1) lds_st(&c.ctl->gathering, 1u);
    // sweep q_h, k_kvh, v_kvh of the M new positions: granule (t, pair p) holds rows (i, i + half) of head p / half.
    // 3 * half granules per position = NL loads per lane, unconditional (clamped) and all in flight (see sweep)
    constexpr int NL = (3 * (D / 2) + 63) / 64;
    const int n_pairs = (Hq + 2 * Hkv) * half;
    const unsigned tag = edge_tag(c, layer, PE_QKV);
    const unsigned long long* gq = edge_base(a, layer, PE_QKV);
    for (int m = cw - 1; m < M; m += pos_shared ? 2 : 1) {
      const unsigned g0 = static_cast<unsigned>(b * M + m) * n_pairs;
      const unsigned long long* p[NL];
      int sel[NL], ii[NL];
      bool act[NL];
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const int i = l * 64 + lane;            // 0..half-1: q, half..2half-1: k, 2half..: v
        act[l] = i < 3 * half;
        const int ic = act[l] ? i : 3 * half - 1;
        sel[l] = ic / half;
        ii[l] = ic - sel[l] * half;
        const int head = (sel[l] == 0) ? h : (sel[l] == 1 ? Hq + kvh : Hq + Hkv + kvh);
        p[l] = gq + granule_slot(a, g0 + head * half + ii[l]);
      }
      unsigned val[NL];
      unsigned long long xa[NL], xb[NL];
      auto issue = [&](unsigned long long (&x)[NL]) {
#pragma unroll
        for (int l = 0; l < NL; ++l) x[l] = __hip_atomic_load(p[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      auto complete = [&](const unsigned long long (&x)[NL]) {
        bool ok = true;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          val[l] = static_cast<unsigned>(x[l]);
          ok &= static_cast<unsigned>(x[l] >> 32) == tag;
        }
        return __all(ok) != 0;
      };
      issue(xa);   // two passes in flight (see sweep)
      __builtin_amdgcn_s_sleep(8);
      for (unsigned spins = 1;; ++spins) {
        issue(xb);
        __builtin_amdgcn_sched_barrier(0);
        if (complete(xa)) break;
        issue(xa);
        __builtin_amdgcn_sched_barrier(0);
        if (complete(xb)) break;
        if ((spins & 127u) == 0u && expired<4>(c)) { give_up(c, ST_GRANULE); return false; }
      }
#pragma unroll
      for (int l = 0; l < NL; ++l)
        if (act[l]) {
          if (sel[l] == 2) {   // V: channel-major, position m in slot m
            vT_s[ii[l] * 8 + m] = static_cast<uint16_t>(val[l]);
            vT_s[(ii[l] + half) * 8 + m] = static_cast<uint16_t>(val[l] >> 16);
          } else {
            uint16_t* dst = (sel[l] == 0 ? q_s : k_s) + m * D + ii[l];
            dst[0] = static_cast<uint16_t>(val[l]);
            dst[half] = static_cast<uint16_t>(val[l] >> 16);
          }
        }
    }
    if (cw == 2) {
      lds_st(&c.ctl->a2_seq, unit_no);
      if (!wait_word<1>(c, &c.ctl->a_seq, unit_no, ST_ATTN)) return false;
    } else {
      if (pos_shared && !wait_word<1>(c, &c.ctl->a2_seq, unit_no, ST_ATTN)) return false;
      lds_st(&c.ctl->gathering, 0u);
      lds_st(&c.ctl->a_seq, unit_no);
      if (!SD_P_DIAG) stamp(c, ts + 10);   // diagnostic: q / new k / new v staged
    }
  } else {
    if (!wait_word<1>(c, &c.ctl->a_seq, unit_no, ST_ATTN)) return false;
  }

  // Q fragments: lane (g, n) holds Q[row n][32 s + 8 g .. +8]; rows >= M are zero
  u32x4 qf[NKS];
#pragma unroll
  for (int s = 0; s < NKS; ++s) qf[s] = (n < M) ? *reinterpret_cast<const u32x4*>(q_s + n * D + s * 32 + g * 8) : u32x4{0u, 0u, 0u, 0u};

  f32x4_t oacc[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i) oacc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  // one 32-key block: kf / vf are its operands. Cached block: keys below pos0 count (nvis = pos0 - key0);
  // block of the new positions: key j = position pos0 + j, seen by query rows n >= j
  auto block = [&](const u32x4 (&kf)[2][NKS], const u32x4 (&vf)[NDT], int nvis, bool causal_new) {
    f32x4_t stt[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      stt[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NKS; ++s)
        stt[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[u][s]), __builtin_bit_cast(bf16x8_t, qf[s]), stt[u], 0, 0, 0);
    }
    float sc[8], mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jk = 8 * g + 4 * u + r;   // key index inside the block
        const bool vis = causal_new ? (jk < M && jk <= n) : (jk < nvis);
        const float v = vis ? stt[u][r] * a.attn_scale : -INFINITY;
        sc[4 * u + r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    float alpha = 1.f, psum = 0.f, p[8];
    if (m_new > -INFINITY) {
      alpha = (m_run > -INFINITY) ? __expf(m_run - m_new) : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        p[j] = (sc[j] > -INFINITY) ? __expf(sc[j] - m_new) : 0.f;
        psum += p[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) p[j] = 0.f;
    }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    const u32x4 pf = {bf16x2_pack(f32x2_t{p[0], p[1]}), bf16x2_pack(f32x2_t{p[2], p[3]}), bf16x2_pack(f32x2_t{p[4], p[5]}), bf16x2_pack(f32x2_t{p[6], p[7]})};
    float al[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) al[r] = __shfl(alpha, 4 * g + r, 64);
#pragma unroll
    for (int i = 0; i < NDT; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[i][r] *= al[r];
      oacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pf), __builtin_bit_cast(bf16x8_t, vf[i]), oacc[i], 0, 0, 0);
    }
  };

  if (cw < nb_old) block(kf0, vf0, pos0 - cw * 32, false);
  if (second) block(kf1, vf1, pos0 - (cw + 3) * 32, false);
  for (int blk = cw + 6; blk < nb_old; blk += 3) {
    u32x4 kf[2][NKS], vf[NDT];
    load_block(blk, kf, vf);
    block(kf, vf, pos0 - blk * 32, false);
  }
  if (cw == nb_old % 3) {
    // the M new positions, from LDS: key j = row j of k_s / v_s
    u32x4 kf[2][NKS], vf[NDT];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = 8 * (n >> 2) + 4 * u + (n & 3);
#pragma unroll
      for (int s = 0; s < NKS; ++s) kf[u][s] = (j < M) ? *reinterpret_cast<const u32x4*>(k_s + j * D + s * 32 + g * 8) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < NDT; ++i) {   // keys 8 g + jj < M <= 8 only exist for g = 0; slots >= M of a channel are zero
      const u32x4 w = *reinterpret_cast<const u32x4*>(vT_s + (16 * i + n) * 8);
      vf[i] = (g == 0) ? w : u32x4{0u, 0u, 0u, 0u};
    }
    block(kf, vf, 0, true);
  }

  // ---- partial (max, sum, O) of this wave -> LDS; the leader merges
  if (g == 0 && n < M) {
    m_s[cw * 8 + n] = m_run;
    l_s[cw * 8 + n] = l_run;
  }
#pragma unroll
  for (int i = 0; i < NDT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * g + r < M) o_s[(cw * M + 4 * g + r) * D + 16 * i + n] = oacc[i][r];
  lds_st(&c.ctl->a_done[cw], unit_no);
  if (cw == 0) {
    if (!wait_word<1>(c, &c.ctl->a_done[1], unit_no, ST_ATTN) || !wait_word<1>(c, &c.ctl->a_done[2], unit_no, ST_ATTN)) return false;
    if (!SD_P_DIAG) stamp(c, ts + 11);   // diagnostic: the three partials are in
    const unsigned tag = edge_tag(c, layer, PE_ATTN);
    for (int it = lane; it < M * half; it += 64) {
      const int r = it / half, dj = it - r * half;
      float mm = -INFINITY;
#pragma unroll
      for (int w = 0; w < 3; ++w) mm = fmaxf(mm, m_s[w * 8 + r]);
      float n0 = 0.f, n1 = 0.f, den = 0.f;
#pragma unroll
      for (int w = 0; w < 3; ++w) {
        const float mw = m_s[w * 8 + r];
        const float f = (mw > -INFINITY) ? __expf(mw - mm) : 0.f;
        n0 += f * o_s[(w * M + r) * D + 2 * dj];
        n1 += f * o_s[(w * M + r) * D + 2 * dj + 1];
        den += f * l_s[w * 8 + r];
      }
      const float v0 = (den > 0.f) ? n0 / den : 0.f, v1 = (den > 0.f) ? n1 / den : 0.f;
      const unsigned val = static_cast<unsigned>(float_to_bf16_bits(v0)) | (static_cast<unsigned>(float_to_bf16_bits(v1)) << 16);
      const int t = b * M + r;
      store_granule(granule_ptr(a, layer, PE_ATTN, static_cast<unsigned>(t) * (Hq * half) + h * half + dj), tag, val);
      if (C::kTaps) reinterpret_cast<unsigned*>(static_cast<uint16_t*>(a.attn) + static_cast<size_t>(t) * Hq * D + h * D)[dj] = val;
    }
    lds_st(&c.ctl->a_merged, unit_no);
  }
  st.att_no = unit_no;
  return true;
}

// attention units (b, h) -> CUs: spread over the chip and over the XCDs (workgroup id mod 8)
__device__ __forceinline__ int unit_of_cu(int cu, int n_units) {
  const int stride = kPersistCUs / n_units;   // >= 1 (checked by the host)
  const int u = cu / stride;
  const int home = u * stride + (stride >= 8 ? (u & 7) : 0);
  return (u < n_units && home == cu) ? u : -1;
}

template <int D, int HC, class C>
__device__ __forceinline__ void consumer_role(const C& c, int cw) {
  const PersistArgs& a = *c.a;
  const int lane = c.lane;
  ConsState st{};
  st.best_v = -INFINITY;
  st.best_i = 0x7fffffff;
  LeadLane L{};
  L.t = lane >> 3;
  L.jp = lane & 7;
  L.tok_ok = L.t < c.T;
  L.b = L.tok_ok ? L.t / a.M : 0;
  L.m = L.tok_ok ? L.t - L.b * a.M : 0;
  L.pos = a.pos_base[L.b] + a.pos_off + L.m;
  const int n_units = a.B * a.n_q_heads;
  const int unit = unit_of_cu(c.cu, n_units);

  if (cw == 0) {
    // ---- one-time staging by the leader: RoPE factors of its QKV items and its share of the residual stream (embedding rows)
    const OpView o0 = load_op(c, 0);   // QKV of layer 0: geometry of every QKV op
    const int half = a.head_dim >> 1;
    float2* rope = reinterpret_cast<float2*>(c.smem + a.lds_rope);
    for (int tile = 0; tile < o0.n_tiles; ++tile) {
      const int p = o0.p_lo + tile * o0.tile_pairs + L.jp;
      float2 cs = {1.f, 0.f};
      if (L.tok_ok && L.jp < o0.tile_pairs && p < o0.n_pairs && L.pos >= 0 && L.pos < a.max_pos) {
        const int i = p % half;
        cs.x = a.rope_cos[static_cast<size_t>(L.pos) * half + i];
        cs.y = a.rope_sin[static_cast<size_t>(L.pos) * half + i];
      }
      rope[tile * 64 + lane] = cs;
    }
    // residual: pairs [cu * resid_ppw, +resid_ppw) of every token row
    unsigned* res = reinterpret_cast<unsigned*>(c.smem + a.lds_resid);
    const int npt = a.d_model >> 1;
    for (int it = lane; it < c.T * static_cast<int>(a.resid_ppw); it += 64) {
      const int t = it / a.resid_ppw, j = it - t * a.resid_ppw;
      const int p = c.cu * a.resid_ppw + j;
      const int b = t / a.M, m = t - b * a.M;
      int tok = a.tokens[b * a.tok_stride + m];
      tok = tok < 0 ? 0 : (tok >= a.vocab ? a.vocab - 1 : tok);
      res[it] = (p < npt) ? reinterpret_cast<const unsigned*>(static_cast<const uint16_t*>(a.tok_emb) + static_cast<size_t>(tok) * a.d_model)[p] : 0u;
    }
  }

  for (int i = 0; i < a.n_ops; ++i) {
    const OpView o = load_op(c, i);
    // ---- the op's input rows: the gatherer stages them (the third consumer sweeps the second half of wide plain rows).
    //      One call site per routine: see consume_op on code size.
    const bool normed = o.kind == POP_QKV || o.kind == POP_GATEUP || o.kind == POP_HEAD;
    if (cw == 1) stamp(c, 12 * i + 0);
    if (cw == 1) lds_st(&c.ctl->gathering, 1u);
    bool ok = true;
    // One token: the gatherer stages the row (the third consumer sweeps the second half of wide plain rows). Several tokens:
    // the rows go alternately to the gatherer and the third consumer, so a second token costs no second round of sweeps
    // (measured before: every gather of a 2-token pass took 1.6-2x the 1-token time, +10 us per layer).
    const bool rows_shared = c.T > 1;
    if (normed) {
      const bool halves = !rows_shared && (a.d_model & 255) == 0;   // (half a row in whole 64-granule sweeps)
      if (cw == 1 || (cw == 2 && (rows_shared || halves))) {
        const int edge = (o.kind == POP_GATEUP) ? PE_X2 : PE_X;   // the head reads the rows the last down-projection left (layer index n_layers)
        const bool emb = (o.kind == POP_QKV && o.layer == 0) || (o.kind == POP_HEAD && a.n_layers == 0);
        if (halves) ok = gather_norm_half<HC>(c, o, edge, emb, st, cw == 1 ? 12 * i : -1, cw, static_cast<unsigned>(i + 1));
        else ok = gather_norm_rows<HC>(c, o, edge, emb, st, cw == 1 ? 12 * i : -1, cw - 1, rows_shared ? 2 : 1, cw);
        if (ok && (rows_shared || halves)) {
          if (cw == 2) lds_st(&c.ctl->g2_seq, static_cast<unsigned>(i + 1));
          else ok = wait_word<1>(c, &c.ctl->g2_seq, static_cast<unsigned>(i + 1), ST_USEQ);
        }
      }
    } else if (cw != 0) {
      const int edge = (o.kind == POP_OUT) ? PE_ATTN : PE_ACT;
      const int npt = (o.kind == POP_OUT) ? (a.n_q_heads * a.head_dim) >> 1 : a.d_ff >> 1;
      const int mid = rows_shared ? npt : ((npt > kSplitSweep) ? ((npt >> 1) + 1023) & ~1023 : npt);
      const int lo = (cw == 1 || rows_shared) ? 0 : mid, hi = (cw == 1 || rows_shared) ? mid : npt;
      if (lo < hi) ok = gather_plain_rows(c, o.layer, edge, npt, lo, hi, st, cw, rows_shared ? cw - 1 : 0, rows_shared ? 2 : 1);
      if (ok && (mid < npt || rows_shared)) {
        if (cw == 2) lds_st(&c.ctl->g2_seq, static_cast<unsigned>(i + 1));
        else ok = wait_word<1>(c, &c.ctl->g2_seq, static_cast<unsigned>(i + 1), ST_USEQ);   // the third consumer's part
      }
    }
    if (!ok) return;
    if (cw == 1) {
      lds_st(&c.ctl->gathering, 0u);
      lds_st(&c.ctl->u_seq, static_cast<unsigned>(i + 1));
      stamp(c, 12 * i + 1);
    } else if (cw == 2) {
      if (!wait_word<1>(c, &c.ctl->u_seq, static_cast<unsigned>(i + 1), ST_USEQ)) return;
      stamp(c, 12 * i + 4);
    }   // (the leader: see consume_op)
    {
      const int ts = cw == 0 ? 12 * i + 6 : (cw == 2 ? 12 * i + 5 : -1);
      // regular geometry: whole tiles of 8 / 6 / 4 pairs and whole chunks (see consume_op)
      const bool whole = o.my_pairs > 0 && o.my_pairs % o.tile_pairs == 0;
      const int units = (o.tile_pairs == 4) ? o.steps >> 1 : o.steps;
      const bool reg = whole && (o.tile_pairs != 4 || (o.steps & 1) == 0) && units % kChunk == 0;
      bool ok2;
      if (SD_P_DIAG && C::kStamps && cw == 2) {
        if (reg && o.tile_pairs == 8) ok2 = consume_op<8, true>(c, cw, o, st, L, ts, static_cast<unsigned>(i + 1));
        else if (reg && o.tile_pairs == 4) ok2 = consume_op<4, true>(c, cw, o, st, L, ts, static_cast<unsigned>(i + 1));
        else if (reg && o.tile_pairs == 6) ok2 = consume_op<6, true>(c, cw, o, st, L, ts, static_cast<unsigned>(i + 1));
        else ok2 = consume_op<0, true>(c, cw, o, st, L, ts, static_cast<unsigned>(i + 1));
      } else {
        if (reg && o.tile_pairs == 8) ok2 = consume_op<8>(c, cw, o, st, L, ts, static_cast<unsigned>(i + 1));
        else if (reg && o.tile_pairs == 4) ok2 = consume_op<4>(c, cw, o, st, L, ts, static_cast<unsigned>(i + 1));
        else if (reg && o.tile_pairs == 6) ok2 = consume_op<6>(c, cw, o, st, L, ts, static_cast<unsigned>(i + 1));
        else ok2 = consume_op<0>(c, cw, o, st, L, ts, static_cast<unsigned>(i + 1));
      }
      if (!ok2) return;
    }
    if (cw == 0) stamp(c, 12 * i + 2);
    if (o.kind == POP_QKV && unit >= 0) {
      if (!attention_unit<D>(c, cw, o.layer, unit / a.n_q_heads, unit % a.n_q_heads, st, 12 * i)) return;
      if (cw == 0) stamp(c, 12 * i + 3);
    }
  }
  if (cw == 0 && a.n_ops > 0 && c.ops[a.n_ops - 1].kind == POP_HEAD) {
    // per-workgroup argmax partial of every token: fold the 8 pair slots of a token (lanes t * 8 .. t * 8 + 7)
    float bv = st.best_v;
    int bi = st.best_i;
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (argmax_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (L.tok_ok && L.jp == 0) {
      a.part_val[static_cast<size_t>(L.t) * kPersistCUs + c.cu] = bv;
      a.part_idx[static_cast<size_t>(L.t) * kPersistCUs + c.cu] = bi;
    }
  }
}

// SEL: the same code under a second name, for launches that carry a skip word (a pass of the captured step that may return at
// entry: the two forms of draft forward 0, adaptive-K forwards) — so that profiles keep the launches that always run (the
// kernel the bench's roofline line names) apart from those that sometimes return after a few hundred cycles.
template <int D, int HC, bool STAMPS, bool SEL, bool TAPS>
__global__ __launch_bounds__(256) void persist_forward_kernel(const PersistArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  SD_SKIP_IF_INACTIVE(a.skip_k, a.skip_i);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < 64) reinterpret_cast<unsigned*>(smem)[tid] = 0u;
  // V^T scratch of the new positions (attention_unit): key slots >= M are never written and must read as zero
  for (int i = tid; i < 4 * D; i += 256) reinterpret_cast<unsigned*>(smem + a.lds_attn + 2 * a.M * D * 2)[i] = 0u;
  const unsigned launch = __hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  PCtxT<STAMPS, TAPS> c;
  c.a = &a;
  c.smem = smem;
  c.ctl = reinterpret_cast<PCtl*>(smem);
  c.t_start = __builtin_amdgcn_s_memrealtime();
  c.lane = tid & 63;
  c.cu = blockIdx.x;
  c.T = a.B * a.M;
  c.tag0 = launch << 9;
  c.ops = (cops_t)a.ops;
  // diagnostic: the shader clock this launch ran at = delta s_memtime / delta s_memrealtime x 100 MHz (leader of every CU)
  unsigned long long clk0 = 0;
  if (STAMPS && a.debug_ts && wave == 1) clk0 = __builtin_amdgcn_s_memtime();
  if (wave == 0) loader_role(c);
  else consumer_role<D, HC>(c, wave - 1);
  if (STAMPS && a.debug_ts && wave == 1 && c.lane == 0) {
    unsigned long long* d = a.debug_ts + static_cast<size_t>(c.cu) * (12 * a.n_ops + 4) + 12 * a.n_ops;
    d[0] = c.t_start;
    d[1] = clk0;
    d[2] = __builtin_amdgcn_s_memrealtime();
    d[3] = __builtin_amdgcn_s_memtime();
  }
  // The next launch's tags: advanced by workgroup 0's leader when the launch COMPLETED. Its leader can only get here after
  // every workgroup has published its share of the last edge, i.e. after every workgroup has read the counter at entry. A launch
  // that gave up leaves the counter alone: a workgroup of it that is scheduled late (the CUs were shared with another kernel)
  // must not pick up the next launch's tags; sd_model_engine_status_clear moves the counter past the failed launch from the
  // host, after the stream has drained.
  if (blockIdx.x == 0 && wave == 1 && c.lane == 0 && lds_ld(&c.ctl->abort_) == 0u)
    __hip_atomic_store(a.sync, (launch + 1u) & 0x7fffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr size_t kLdsBytes = 160 * 1024;
static unsigned align16(unsigned v) { return (v + 15u) & ~15u; }

struct LdsPlan {
  unsigned rope, resid, part, attn, u, ring, ring_bytes, u_stride;
  bool ok;
};
static LdsPlan plan_lds(int d_model, int HqD, int d_ff, int head_dim, int T, int M) {
  LdsPlan p{};
  int kmax = d_model > HqD ? d_model : HqD;
  if (d_ff > kmax) kmax = d_ff;
  unsigned off = 256;
  p.rope = off;
  off += kPersistMaxQkvTiles * 64 * 8;
  p.resid = off;
  const int resid_ppw = gemv_geometry(d_model / 2, HqD).ppw;
  off += align16(static_cast<unsigned>(T) * resid_ppw * 4);
  p.part = off;
  off += 2 * 3 * 16 * kPartT * 4;
  p.attn = off;
  off += align16(static_cast<unsigned>((2 * M + 8) * head_dim * 2 + 3 * M * head_dim * 4 + 2 * 3 * 8 * 4));
  p.u = off;
  p.u_stride = static_cast<unsigned>(kmax + kXPad) * 2;
  off += align16(static_cast<unsigned>(T) * p.u_stride);
  p.ring = off;
  p.ok = off + 64 * 1024 <= kLdsBytes;   // at least 64 KiB of ring
  p.ring_bytes = p.ok ? ((static_cast<unsigned>(kLdsBytes) - off) / kPiece) * kPiece : 0;
  return p;
}

}  // namespace

bool persist_model_ok(const sd_model_config& c, bool packed, bool w8, int n_cus) {
  if (getenv(debug_env::kNoPersist)) return false;
  if (c.arch != SD_ARCH_LLAMA || !packed || w8 || n_cus != kPersistCUs) return false;
  if (c.head_dim != 64 && c.head_dim != 128) return false;
  const int HqD = c.n_heads * c.head_dim;
  if (c.d_model % 128 || HqD % 128 || c.d_ff % 128 || c.d_model > 4096) return false;   // rows: whole 64-granule sweeps, <= 2 chunks of 1024
  if (c.n_layers < 1 || c.n_layers > 60) return false;                                    // tags: layer * 8 + edge + 1 < 512
  for (int l = 0; l < c.n_layers; ++l) {
    const sd_layer_weights& w = c.layers[l];
    if (w.bqkv || w.bo || w.b_up || w.b_down) return false;
  }
  // QKV tiles (RoPE factors are staged per tile), even SwiGLU tiles (an activation granule holds two neighbouring pairs)
  const GemvGeom gq = gemv_geometry((c.n_heads + 2 * c.n_kv_heads) * c.head_dim / 2, c.d_model);
  if ((gq.ppw + gq.tile_pairs - 1) / gq.tile_pairs > kPersistMaxQkvTiles) return false;
  const GemvGeom gu = gemv_geometry(c.d_ff, c.d_model);
  if ((gu.ppw & 1) || (gu.tile_pairs & 1) || (c.d_ff & 1)) return false;
  // every matrix cut for <= 256 workgroups
  const GemvGeom go = gemv_geometry(c.d_model / 2, HqD), gd = gemv_geometry(c.d_model / 2, c.d_ff), gh = gemv_geometry((c.vocab + 1) / 2, c.d_model);
  if (gq.grid > kPersistCUs || gu.grid > kPersistCUs || go.grid > kPersistCUs || gd.grid > kPersistCUs || gh.grid > kPersistCUs) return false;
  return persist_max_tokens(c) >= 1;
}

int persist_max_tokens(const sd_model_config& c) {
  int best = 0;
  for (int T = 1; T <= kPersistMaxT; ++T)
    if (plan_lds(c.d_model, c.n_heads * c.head_dim, c.d_ff, c.head_dim, T, T).ok) best = T;
  return best;
}

// storage granules of a vector of n granules per token at the widest unit stride (n is a multiple of 16)
static size_t edge_storage(size_t n_per_tok, unsigned unit) { return (n_per_tok * kPersistMaxT / 16) * unit; }
static size_t persist_gran_parity(const sd_model_config& c) {
  const size_t per_tok = static_cast<size_t>(c.d_model / 2) * 2 + (c.n_heads + 2 * c.n_kv_heads) * c.head_dim / 2 + c.n_heads * c.head_dim / 2 + c.d_ff / 2;
  return edge_storage(per_tok, kGranUnitMax) + 5 * kGranUnitMax;
}

size_t persist_workspace_bytes(const sd_model_config& c) {
  size_t n = 256;                                                           // sync words
  n += (static_cast<size_t>(4 * c.n_layers + 1) * sizeof(PersistOp) + 255) & ~static_cast<size_t>(255);
  n += (2 * persist_gran_parity(c) * 8 + 255) & ~static_cast<size_t>(255);
  return n;
}

template <int D, int HC, bool STAMPS, bool SEL, bool TAPS>
static int launch_inst(const PersistArgs& a, size_t smem, hipStream_t st) {
  // (the attribute is per device: a process that drives several GPUs sets it on each)
  static unsigned long long attr_set = 0;
  if (int rc = opt_in_dynamic_lds(reinterpret_cast<const void*>(&persist_forward_kernel<D, HC, STAMPS, SEL, TAPS>), static_cast<int>(kLdsBytes), attr_set)) return rc;
  // test hook (tests/test_hip_persist_gpu.py): one workgroup short, so granules are missing and every bounded wait has to expire
  const int grid = getenv(debug_env::kPersistDropWg) ? kPersistCUs - 1 : kPersistCUs;
  hipLaunchKernelGGL((persist_forward_kernel<D, HC, STAMPS, SEL, TAPS>), dim3(grid), dim3(256), smem, st, a);
  SD_LAUNCH_CHECK();
  return 0;
}
template <int D, int HC>
static int launch_one(const PersistArgs& a, size_t smem, hipStream_t st) {
  if (a.debug_ts) return launch_inst<D, HC, true, false, true>(a, smem, st);
  if (a.taps) return a.skip_k ? launch_inst<D, HC, false, true, true>(a, smem, st) : launch_inst<D, HC, false, false, true>(a, smem, st);
  return a.skip_k ? launch_inst<D, HC, false, true, false>(a, smem, st) : launch_inst<D, HC, false, false, false>(a, smem, st);
}

int launch_persist_forward(PersistArgs a, hipStream_t st) {
  const int T = a.B * a.M;
  SD_REQUIRE(T >= 1 && T <= kPersistMaxT && a.M <= 8, "persist: T=%d M=%d out of range", T, a.M);
  SD_REQUIRE(a.B * a.n_q_heads <= kPersistCUs, "persist: %d attention units exceed the CUs", a.B * a.n_q_heads);
  // attention_unit reads V^T in 16-byte vectors of 8 keys at n * l_max + key0 and clamps to l_max - 8
  SD_REQUIRE(a.l_max >= 8 && a.l_max % 8 == 0, "persist: cache rows of %d positions (need a multiple of 8)", a.l_max);
  const int HqD = a.n_q_heads * a.head_dim;
  const LdsPlan p = plan_lds(a.d_model, HqD, a.d_ff, a.head_dim, T, a.M);
  SD_REQUIRE(p.ok, "persist: T=%d rows do not fit the LDS next to a 64 KiB ring", T);
  a.lds_rope = p.rope;
  a.lds_resid = p.resid;
  a.lds_part = p.part;
  a.lds_attn = p.attn;
  a.lds_u = p.u;
  a.lds_ring = p.ring;
  a.ring_bytes = p.ring_bytes;
  a.u_stride = p.u_stride;
  a.resid_ppw = gemv_geometry(a.d_model / 2, HqD).ppw;
  // granule buffers inside a parity (kPersistMaxT rows each), 16-granule units at a stride of gran_unit storage granules
  a.gran_unit = kGranUnitMax;   // (16 / 64 / 144 / 528 measured: no difference; every 128-byte unit in a page of its own)
  size_t off = 0;
  a.off_edge[PE_X] = static_cast<unsigned>(off); off += edge_storage(a.d_model / 2, a.gran_unit);
  a.off_edge[PE_QKV] = static_cast<unsigned>(off); off += edge_storage((a.n_q_heads + 2 * a.n_kv_heads) * a.head_dim / 2, a.gran_unit);
  a.off_edge[PE_ATTN] = static_cast<unsigned>(off); off += edge_storage(HqD / 2, a.gran_unit);
  a.off_edge[PE_X2] = static_cast<unsigned>(off); off += edge_storage(a.d_model / 2, a.gran_unit);
  a.off_edge[PE_ACT] = static_cast<unsigned>(off); off += edge_storage(a.d_ff / 2, a.gran_unit);
  SD_REQUIRE(off <= a.gran_parity, "persist: granule buffer too small");
  const size_t smem = kLdsBytes;
  const bool two = a.d_model > 2048;
  if (a.head_dim == 64) return two ? launch_one<64, 2>(a, smem, st) : launch_one<64, 1>(a, smem, st);
  return two ? launch_one<128, 2>(a, smem, st) : launch_one<128, 1>(a, smem, st);
}

}  // namespace sd
