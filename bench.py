#!/usr/bin/env python3
"""bench.py — accepted tokens/sec + acceptance rate of the draft-then-verify loop.

    python bench.py --gpus 1 --steps 40 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        (no launcher: bench.py starts the N ranks itself, see launch_ranks)

Workload (BASELINE.json configs[1]): Llama-3.2-3B target + Llama-3.2-1B draft shapes, K = 4,
batch 1 per GPU, bf16, synthetic prompts (32 ids, generator seed 1234+i) and architecture-
exact synthetic weights (specdec_hip/weights.py: no checkpoints or datasets offline).
A "step" is one draft-then-verify step of the product pipeline (one hipGraph replay +
the host-side reference rules). EXACTLY --steps steps are timed between barrier +
torch.cuda.synchronize() pairs; value = tokens emitted by all ranks / max-over-ranks time.
With N > 1 every rank decodes its own prompts on its own GPU (data parallel over prompts,
weak scaling, no collective in the data path) and the per-rank stats are all-gathered
once over RCCL.

One JSON line is printed by rank 0. Besides the contract keys it carries
  roofline     — the dominant kernel (fused norm + gate/up + SwiGLU GEMV of the target at
                 T = K+1 tokens) priced against HBM: algorithmic bytes N*K*2 per launch over
                 its average launch duration, timed here with HIP events on the launch stream
  cpu_baseline — the oracle's reference-faithful loop (2K full-prefix forwards per step, as
                 the reference with KV append off) on the host cores, bounded sample
  step_roofline_frac — (K*W_draft + W_target bytes) / step time / 8 TB/s for the whole step
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "llm-inference-lab_amd"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

HBM_PEAK_BPS = 8.0e12  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
PROMPT_LEN = 32


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--k", type=int, default=4)
    ap.add_argument("--batch", type=int, default=1, help="rows per GPU")
    ap.add_argument("--target", default="llama-3.2-3b")
    ap.add_argument("--draft", default="llama-3.2-1b")
    ap.add_argument("--flip", type=float, default=0.2, help="fraction of tokens whose draft successor differs")
    ap.add_argument("--cpu-baseline-steps", type=int, default=4,
                    help="steps per timed repeat of the CPU baseline leg (1 warm-up step + 3 repeats); 0 disables the leg")
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--weight-dtype", choices=["bf16", "fp8"], default="bf16",
                    help="fp8: both models stream an OCP e4m3 copy of their Linear weights (per-row scales, bf16 activations and "
                         "MFMA); not the headline configuration (BASELINE config 2 is bf16), CPU parity leg uses the dequantised weights")
    ap.add_argument("--draft-mode", choices=["vanilla", "medusa", "medusa-heads", "eagle"], default="vanilla",
                    help="medusa: BASELINE config 5's draft — Medusa-lite heads tied to the lm_head (the reference draftor's "
                         "semantics: K copies of the target's next token), single-prompt generate() loop, no draft model; "
                         "medusa-heads: K persistent heads over the target's last hidden state (not in the reference; synthetic heads, "
                         "--flip of their rows wrong), generate_batch loop")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch plumbing only (CPU tests of --gpus N): every rank initialises its process group, contributes a "
                         "fixed stats struct to the all-gather and rank 0 prints n_gpus; nothing is decoded or timed")
    ap.add_argument("--do-sample", action="store_true",
                    help="sampled bonus token (T=0.7, top_k=50, top_p=0.9: the reference's default sampler) instead of greedy; "
                         "not the headline configuration (SPECDEC_DETERMINISTIC is greedy), no CPU parity leg")
    return ap.parse_args()


def build_models(args, device):
    from specdec_hip import weights as W

    presets = {"llama-3.2-1b": W.LLAMA_3_2_1B, "llama-3.2-3b": W.LLAMA_3_2_3B, "llama-3-8b": W.LLAMA_3_8B}
    ckpt = os.environ.get("SPECDEC_MODEL_DIR")
    if ckpt and os.path.isdir(os.path.join(ckpt, args.target)) and os.path.isdir(os.path.join(ckpt, args.draft)):
        tgt = W.load_checkpoint_dir(os.path.join(ckpt, args.target), device=device)
        drf = W.load_checkpoint_dir(os.path.join(ckpt, args.draft), device=device)
        return drf, tgt, "checkpoint"
    tgt = W.synthetic_llama(presets[args.target], seed=0, device=device)
    drf = W.synthetic_llama(presets[args.draft], seed=1, device=device, embed_from=tgt, flip_fraction=args.flip)
    return drf, tgt, "synthetic"


def prompts_for(rank, batch, vocab):
    out = []
    for i in range(batch):
        g = torch.Generator().manual_seed(1234 + rank * batch + i)
        out.append(torch.randint(4, vocab, (PROMPT_LEN,), generator=g, dtype=torch.int64).tolist())
    return out


def cpu_model_string() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform

    return platform.processor() or platform.machine()


def cpu_threads():
    """(threads to use, how that number was found): the cgroup CPU quota when the box enforces one (a 1-GPU share of a
    multi-GPU host), else the affinity mask; SPECDEC_CPU_THREADS overrides."""
    if os.environ.get("SPECDEC_CPU_THREADS"):
        return int(os.environ["SPECDEC_CPU_THREADS"]), "SPECDEC_CPU_THREADS"
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            return max(1, min(visible, int(int(quota) / int(period)))), f"cgroup cpu.max {quota}/{period} (of {visible} visible)"
    except (OSError, ValueError):
        pass
    # no quota visible: a 1-GPU box of this pool is given a 16-CPU share of the host whatever the mask says
    return min(visible, 16), f"min(affinity mask {visible}, 16-CPU share of a 1-GPU box)"


def cpu_baseline(drf, tgt, prompts, k, gpu_rows, weight_dtype="bf16", seg_steps=4, repeats=3, gpu_counters=None):
    """Oracle (CPU restatement of the reference loop) on the host cores, bounded sample (SURVEY section 8d protocol:
    one warm-up, >= 3 timed repeats, median; CPU model and thread count stated).
      L0 = 32 (the GPU run's prompt 0): ONE run of 1 + repeats * seg_steps steps; step 1 is the warm-up, every following
        segment of `seg_steps` steps is one timed repeat (tokens of the segment / its wall time); all of its tokens are
        compared with what the GPU emitted for that prompt;
      L0 = 128 (a second synthetic prompt, CPU only): 1 warm-up step + `repeats` single-step repeats.
    Returns the baseline dict and whether the GPU emitted the same tokens on the L0 = 32 steps."""
    from oracle.model_ref import OracleLM
    from oracle.pipeline_ref import OraclePipeline

    threads, how = cpu_threads()
    torch.set_num_threads(threads)
    log(f"cpu_baseline: {threads} threads [{how}] on {cpu_model_string()}; copying weights to the host")
    d_cpu, t_cpu = drf.to("cpu"), tgt.to("cpu")
    if weight_dtype == "fp8":
        from oracle import fp8_ref

        d_cpu, t_cpu = fp8_ref.dequantized(d_cpu), fp8_ref.dequantized(t_cpu)
    base, draft = OracleLM(t_cpu, "bf16"), OracleLM(d_cpu, "bf16")
    pipe = OraclePipeline(base, draft, k=k, eos_token_id=tgt.config.eos_token_id, reprefill=True)
    # materialise the fp32 weight copies outside the timed region
    base.forward(torch.tensor([prompts[0][:2]]))
    draft.forward(torch.tensor([prompts[0][:2]]))

    def run(prompt, seg):
        n_steps = 1 + repeats * seg
        marks = [(time.time(), 0)]
        res = pipe.generate_batch([prompt], max_tokens=10 ** 6, max_steps=n_steps, step_log=marks)[0]
        rates = []
        for r in range(repeats):
            (t0, n0), (t1, n1) = marks[1 + r * seg], marks[1 + (r + 1) * seg]
            rates.append((n1 - n0) / (t1 - t0) if t1 > t0 else 0.0)
        return res, sorted(rates)[len(rates) // 2], rates, marks[-1][0] - marks[0][0]

    log(f"cpu_baseline: L0={PROMPT_LEN}: 1 warm-up step + {repeats} x {seg_steps} timed steps of the reference-faithful loop")
    res, med, rates, wall = run(prompts[0], seg_steps)
    n = len(res["generated_tokens"])
    same = gpu_rows[0][:n] == res["generated_tokens"]
    # ... and the draft's side of the same steps: proposed / accepted after the CPU sample's last step against the GPU row's
    # counters after as many of ITS steps (a persistent draft that proposed other tokens would leave the output — the target's
    # greedy continuation either way — alone and only move these)
    counters = None
    if gpu_counters is not None and len(gpu_counters) >= res["steps"] >= 1:
        gp, ga, gn, _ = gpu_counters[res["steps"] - 1]
        counters = {"steps": res["steps"], "cpu": {"proposed": res["proposed"], "accepted": res["accepted"], "tokens": n},
                    "gpu": {"proposed": gp, "accepted": ga, "tokens": gn}}
        counters["equal"] = counters["cpu"] == counters["gpu"]
        same = same and counters["equal"]
        log(f"cpu_baseline: counters after {res['steps']} steps: cpu {counters['cpu']} gpu {counters['gpu']}")
    divergence = None
    if not same:
        # where the GPU's tokens leave the CPU loop's, and how close the call was for the target ON THE CPU: the oracle's logits
        # at that position, its top-2 margin and the gap of the token the GPU emitted (a flip between near-ties is bf16
        # arithmetic under another summation order; a flip at a clear margin is a defect) — SURVEY section 7, "hard parts"
        ref_t = res["generated_tokens"]
        i = next((j for j, (x, y) in enumerate(zip(gpu_rows[0], ref_t)) if x != y), min(len(gpu_rows[0]), len(ref_t)))
        if i < min(len(gpu_rows[0]), len(ref_t)):
            lg, _ = base.forward(torch.tensor([list(prompts[0]) + ref_t[:i]]))
            row = lg[0, -1].double()
            rms = row.pow(2).mean().sqrt().item()
            top2 = row.topk(2).values
            divergence = {"token_index": i, "gpu_token": int(gpu_rows[0][i]), "cpu_token": int(ref_t[i]),
                          "cpu_target_argmax": int(row.argmax()),
                          "cpu_top2_margin_over_rms": float((top2[0] - top2[1]) / rms),
                          "gpu_token_gap_over_rms": float((row.max() - row[gpu_rows[0][i]]) / rms)}
            log(f"cpu_baseline: tokens diverge at {i}: gpu {divergence['gpu_token']} cpu {divergence['cpu_token']}, "
                f"cpu top-2 margin {divergence['cpu_top2_margin_over_rms']:.4f} x RMS(logits)")
    g = torch.Generator().manual_seed(1234 + 10007)
    long_prompt = torch.randint(4, tgt.config.vocab, (128,), generator=g, dtype=torch.int64).tolist()
    log(f"cpu_baseline: L0=128: 1 warm-up step + {repeats} x 1 timed step")
    res_l, med_l, rates_l, wall_l = run(long_prompt, 1)
    return {
        "value": med, "unit": "tokens/s", "cores": threads, "kind": "port",
        "min": min(rates), "median": med, "max": max(rates),
        "cpu_model": cpu_model_string(), "threads_source": how,
        "sample": f"prompt 0 (L0={PROMPT_LEN}), reference-faithful loop (2K full-prefix forwards per step, bf16-rounded weights / activations, fp32 "
                  f"accumulation): 1 warm-up step, then {repeats} repeats of {seg_steps} steps each, median of the per-repeat rates "
                  f"{[round(r, 3) for r in rates]} tokens/s ({n} tokens, {wall:.1f} s in all)",
        "acceptance_rate": res["acceptance_rate"],
        "same_steps_counters": counters,   # proposed / accepted / tokens of the CPU sample's steps against the GPU's first as many steps
        "divergence": divergence,     # null when the GPU emitted the CPU sample's tokens
        "L0_128": {"value": med_l, "unit": "tokens/s", "rates": [round(r, 3) for r in rates_l],
                   "sample": f"a 128-token synthetic prompt, 1 warm-up step + {repeats} x 1 step ({wall_l:.1f} s)"},
    }, bool(same)


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run as a CHILD
    process (one rank per GPU, rendezvous on 127.0.0.1, a free port) and return its exit code; rank 0's JSON line goes
    to this process's stdout unchanged. Called before anything in this process has touched the GPU — the ranks are
    fresh processes, nothing is re-exec'ed."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    log(f"--gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd[1:])}")
    return subprocess.run(cmd, env=env).returncode


def bind_to_gpu_numa(local: int, ranks_on_host: int) -> dict:
    """Pin this rank's host threads to the CPUs of its GPU's NUMA node (the step loop is one Python
    thread launching a graph and reading a pinned record per step, so it should sit next to its GPU's PCIe root):
    /sys/bus/pci/devices/<bdf of cuda:local>/{numa_node,local_cpulist}, intersected with the affinity mask the job was
    given. Leaves the mask alone when sysfs has no answer (numa_node -1, container without the files)."""
    info = {"numa_node": None, "cpus": None}
    try:
        pr = torch.cuda.get_device_properties(local)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        base = f"/sys/bus/pci/devices/{bdf}"
        with open(f"{base}/numa_node") as f:
            node = int(f.read().strip())
        info["numa_node"] = node
        if node < 0:
            return info
        with open(f"{base}/local_cpulist") as f:
            spec = f.read().strip()
        cpus = set()
        for part in spec.split(","):
            if part:
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        cpus &= set(os.sched_getaffinity(0))
        if cpus:
            os.sched_setaffinity(0, cpus)
            info["cpus"] = len(cpus)
    except (AttributeError, OSError, ValueError):
        pass
    return info


def lib_sha256() -> str:
    import hashlib

    from specdec_hip import _abi

    return hashlib.sha256(_abi.lib_path().read_bytes()).hexdigest()


def roofline_leg(sess, B, K, wd, self_draft):
    """Every weight-streaming GEMV of the step timed live (sd_model_probe_gemv: HIP events on the launch stream, round-robin
    over the layers so the weights come from HBM); `roofline` describes the one with the most time per step —
    launches per step x average launch time — priced at its algorithmic bytes (N*K*sizeof(weight)) over that time."""
    st = torch.cuda.Stream()
    tm = sess.rt["target"]
    dm = None if self_draft else sess.rt["draft"]
    Tt = min(B * (K + 1), tm.pass_tokens)
    cand = []   # (label, model, which, T, launches per step)
    names = {tm.PROBE_O: ("o_proj", "EPI_RESID"), tm.PROBE_GATE_UP: ("norm+gate/up+SwiGLU", "EPI_SWIGLU"), tm.PROBE_DOWN: ("down", "EPI_RESID"),
             tm.PROBE_LM_HEAD: ("norm+lm_head+argmax", "EPI_ARGMAX")}
    # passes of <= persist_t tokens of the draft run as ONE persistent launch (csrc/persist.hip): their GEMV launches do not exist
    persist_t = dm.persist_tokens if dm is not None else 0
    for which in (tm.PROBE_O, tm.PROBE_GATE_UP, tm.PROBE_DOWN, tm.PROBE_LM_HEAD):
        n = 1 if which == tm.PROBE_LM_HEAD else tm.cfg.n_layers
        cand.append(("target", tm, which, Tt, n))
        if dm is not None:
            nd = (1 if which == dm.PROBE_LM_HEAD else dm.cfg.n_layers)
            if B > persist_t:
                cand.append(("draft", dm, which, min(B, dm.pass_tokens), nd * (K - 1)))       # draft forwards 2..K: one token per row
            if 2 * B > persist_t:
                cand.append(("draft", dm, which, min(2 * B, dm.pass_tokens), nd))             # draft forward 1: (prev, last)
    rows = []
    for who, m, which, T, n in cand:
        u, nb = m.probe_gemv(which, T=T, iters=200 if which != m.PROBE_LM_HEAD else 60, stream=st)
        # (> 9 tokens: launch_gemm_skinny picks one of four bodies per shape — pipe, slice, direct or the chunked fallback)
        kname = "gemv_mfma_kernel" if T <= 9 else "gemm_pipe/slice/direct/skinny_kernel"
        rows.append({"kernel": f"{kname}<{names[which][1]}> ({who} {names[which][0]}, {T} token{'s' if T > 1 else ''})", "who": who, "which": which,
                     "T": T, "launches_per_step": n, "avg_launch_us": u, "bytes_per_launch": nb, "GBps": nb / (u * 1e-6) / 1e9,
                     "us_per_step": n * u})
    if dm is not None and persist_t >= B and B == 1:
        # the whole 1-token draft forward (embedding, layers, lm_head, argmax partials) is one kernel; timed at a context of
        # prompt + 32 positions; its algorithmic bytes = every matmul weight of the draft once
        # draft forwards 1..K-1 of a step. (Forward 0 is in the captured step twice under the kernel's second name <..., true, false>:
        # a 2-token pass and a 1-token pass, of which the device runs one and the other returns at entry; not counted here.)
        u, nb, _ = dm.probe_forward(M=1, iters=40, pos0=PROMPT_LEN + 32, stream=st)
        # Forward 0 of a step ALSO runs the kernel (under its second name, <..., true, false>: the captured step holds a 2-token
        # and a 1-token form, the device runs one, the other returns at entry): a 2-token pass after a fully accepted step or a
        # repaired row, a 1-token pass otherwise — weighted by what this run observed.
        n_steps = max(sess.stats.get("steps", 0), 1)
        f_full = min(1.0, sess.stats.get("full_accepts", 0) / n_steps)
        fwd0 = None
        if 2 * B <= persist_t:
            u2, _, _ = dm.probe_forward(M=2, iters=40, pos0=PROMPT_LEN + 32, stream=st)
            select = not os.environ.get("SPECDEC_NO_FWD0_SELECT")
            fwd0 = {"two_token_us": u2, "one_token_us": u, "fully_accepted_fraction": f_full if select else 1.0,
                    "us_per_step": (f_full * u2 + (1.0 - f_full) * u) if select else u2}
        c = dm.cfg
        us_step = (K - 1) * u + (fwd0["us_per_step"] if fwd0 else 0.0)
        rows.append({"kernel": f"persist_forward_kernel<{c.head_dim}, {1 if c.d_model <= 2048 else 2}, false, false, false> (draft forward, 1 token: {c.n_layers} layers + "
                               "lm_head as ONE launch)", "who": "draft", "which": -1, "T": 1, "launches_per_step": (K - 1) + (1 if fwd0 else 0), "avg_launch_us": u,
                     "bytes_per_launch": nb, "GBps": nb / (u * 1e-6) / 1e9, "us_per_step": us_step, "forward0": fwd0})
    top = max(rows, key=lambda r: r["us_per_step"])
    ach = top["GBps"]
    roof = {"bound": "hbm", "kernel": top["kernel"], "achieved": ach, "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s",
            "frac": ach * 1e9 / HBM_PEAK_BPS, "traffic": None, "bytes_per_launch": top["bytes_per_launch"],
            "avg_launch_us": top["avg_launch_us"], "launches_per_step": top["launches_per_step"], "us_per_step": top["us_per_step"],
            "forward0": top.get("forward0"),
            "selection": "the weight-streaming kernel with the largest (launches per step x average launch time), all timed in this run"}
    # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in their own runs, FETCH_SIZE x2 on
    # gfx950; profiles/summarize.py): counters cannot be read from inside this process, so the figure comes from the
    # committed summary — and only when that summary was taken with THIS build of the library (content hash)
    pmc = os.path.join(ROOT, "profiles", "round4_pmc_traffic.json")
    epi = {tm.PROBE_O: 1, tm.PROBE_GATE_UP: 2, tm.PROBE_DOWN: 1, tm.PROBE_LM_HEAD: 4, -1: -1}[top["which"]]
    tt = next(t for t in (1, 2, 3, 5, 9) if top["T"] <= t) if top["T"] <= 9 else None
    if os.path.exists(pmc) and tt is not None and wd == "bf16":
        with open(pmc) as f:
            tj = json.load(f)
        meta = tj.get("_meta", {})
        if meta.get("lib_sha256") != lib_sha256():
            roof["traffic_note"] = "profiles/round4_pmc_traffic.json was taken with another build of libspecdec_hip.so: not quoted"
        elif meta.get("workload") != f"{sess.pipe.base_lm.model_name}+{'' if self_draft else sess.pipe.draft_lm.model_name} K={K} B={B}":
            roof["traffic_note"] = f"profiles/round4_pmc_traffic.json is for {meta.get('workload')!r}: not quoted"
        else:
            # gate/up and down differ in bytes per launch: pick the entry of this epilogue and token bucket whose bytes are nearest
            prefix = "persist_forward_kernel<" if epi == -1 else f"gemv_mfma_kernel<{epi}, false, {tt}, false"
            ks = [(k, v) for k, v in tj.items() if k.startswith(prefix)]
            if ks:
                k_, t = min(ks, key=lambda kv: abs(kv[1]["hbm_bytes_per_launch"] - top["bytes_per_launch"]))
                roof["traffic"] = t["hbm_bytes_per_launch"]
                if t.get("avg_us_in_graph"):
                    # the probe above times the kernel in a loop of its own; inside the step's hipGraph (draft forwards 1..K-1, the
                    # acceptance pattern's cache lengths) rocprofv3's kernel trace of the same library has it slightly slower
                    roof["avg_launch_us_in_graph"] = t["avg_us_in_graph"]
                    roof["frac_in_graph"] = top["bytes_per_launch"] / (t["avg_us_in_graph"] * 1e-6) / HBM_PEAK_BPS
                roof["traffic_source"] = f"profiles/round4_pmc_traffic.json [{k_}] (rocprofv3 --pmc passes of bench.py, same library build)"
    roof["other_kernels"] = {r["kernel"]: {k: r[k] for k in ("avg_launch_us", "GBps", "launches_per_step", "us_per_step")} for r in rows if r is not top}
    return roof


def dry_run(args, world, rank):
    """Launch plumbing without a GPU (tests/test_bench_launch.py): process group over gloo, the one all-gather, rank 0
    prints how many ranks contributed. Nothing is decoded, timed or reported as a measurement."""
    from specdec_hip.dist_stats import gather_stats

    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("gloo")
        dist.barrier()
    job = gather_stats({"tokens": rank + 1, "proposed": 4, "accepted": 1, "accepted_strict": 0, "wall_ns": 10 ** 9 + rank, "steps": 1,
                        "numa_node": rank % 2, "cpus": len(os.sched_getaffinity(0))}, torch.device("cpu"))
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": int(job.per_rank.shape[0]), "tokens": job.total("tokens"), "value": None,
                          "ranks": {"ms_per_step_min": min(job.column("wall_ns")) / 1e6, "ms_per_step_max": max(job.column("wall_ns")) / 1e6,
                                    "numa_node": job.column("numa_node"), "cpus": job.column("cpus")}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback to time)")
    # SPECDEC_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: every rank
    # shares cuda:0 and the one collective runs over gloo on host tensors (the real job is RCCL)
    rehearsal = os.environ.get("SPECDEC_BENCH_BACKEND", "nccl") == "gloo"
    if rehearsal:
        local = 0
        # the persistent draft forward needs all 256 CUs to itself (one workgroup per CU, spinning on the others' hand-offs):
        # ranks that share a GPU run the launch path (include/specdec_hip.h, sd_model_engine_status_clear)
        os.environ["SPECDEC_NO_PERSIST"] = "1"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    stats_device = torch.device("cpu") if rehearsal else device
    dist = None
    if world > 1:
        import torch.distributed as dist

        # one process per GPU on one host: keep each rank's host-side torch pool to its share of the cores (the step
        # loop itself is one Python thread launching a graph and reading a 68-byte record per step)
        try:
            share = max(1, len(os.sched_getaffinity(0)) // world)
        except AttributeError:
            share = max(1, (os.cpu_count() or world) // world)
        torch.set_num_threads(share)

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)  # RCCL over xGMI
    numa = bind_to_gpu_numa(local, world) if not rehearsal else {"numa_node": None, "cpus": None}

    from src.specdec import HipLM, SpeculativePipeline
    from specdec_hip.engine import HipSpecDec

    log(f"rank {rank}/{world}: building {args.target} + {args.draft} weights on {device}")
    drf, tgt, source = build_models(args, device)
    wd = args.weight_dtype
    medusa = args.draft_mode in ("medusa", "eagle")   # self-drafting generate() loops of the reference (no draft model, no bonus token)
    heads = None
    if args.draft_mode == "medusa-heads":
        from specdec_hip import weights as W

        heads = W.synthetic_medusa_heads(tgt, args.k, flip_fraction=args.flip)
    pipe = SpeculativePipeline(base_lm=HipLM(tgt, weight_dtype=wd), draft_lm=None if heads is not None else HipLM(drf, weight_dtype=wd),
                               draft_model="none", controller="fixed", controller_params={"k": args.k}, seed=1234,
                               draft_mode="medusa" if heads is not None else args.draft_mode, medusa_heads=heads)
    if medusa or heads is not None:
        args.cpu_baseline_steps = 0
    K, B = args.k, args.batch
    prompts = prompts_for(rank, B, tgt.config.vocab)
    total_steps = args.warmup + args.steps
    sampling = {"temperature": 0.7, "top_k": 50, "top_p": 0.9, "seed": 1234} if args.do_sample else None
    if sampling:
        args.cpu_baseline_steps = 0
    sess = pipe.start_session(prompts, max_tokens=total_steps * (K + 1) + 1,
                              emit_mode=HipSpecDec.EMIT_DRAFT if medusa else HipSpecDec.EMIT_BONUS,
                              sampling=sampling, self_draft=medusa or heads is not None)
    K = sess.k   # eagle: min(k, eagle.max_draft) proposals per step (reference default max_draft = 2)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    log("prefill done, warming up")
    for _ in range(args.warmup):
        sess.advance()
    n0 = sum(len(r.generated) for r in sess.rows)
    p0, a0 = sess.stats["proposed"], sum(r.accepted for r in sess.rows)
    import gc

    # PCIe-inclusive figure (contract: inputs are resident when the timed region starts; what a caller with HOST prompt
    # buffers pays on top is this one copy per request): B x PROMPT_LEN int32 ids, pinned host -> device
    h_ids = torch.tensor(prompts, dtype=torch.int32).pin_memory()
    torch.cuda.synchronize()
    th = time.perf_counter()
    for _ in range(20):
        d_ids = h_ids.to(device, non_blocking=True)
        torch.cuda.synchronize()
    h2d_s = (time.perf_counter() - th) / 20
    del d_ids
    gc.collect()
    gc.disable()            # no collector pauses between graph launches inside the timed region
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sess.advance()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    gc.enable()
    barrier()
    sess.finish()
    n_tok = sum(len(r.generated) for r in sess.rows) - n0
    proposed = sess.stats["proposed"] - p0
    accepted_ref = sum(r.accepted for r in sess.rows) - a0          # reference definition (bonus counted)
    accepted_strict = accepted_ref - (0 if medusa else args.steps * B)   # draft tokens accepted only (generate() counts no bonus)
    from specdec_hip.dist_stats import gather_stats

    # the one collective of the job: a 48-byte struct per rank, all-gathered over RCCL/xGMI
    job = gather_stats({"tokens": n_tok, "proposed": proposed, "accepted": accepted_ref,
                        "accepted_strict": accepted_strict, "wall_ns": int(dt_local * 1e9), "steps": args.steps,
                        "numa_node": -1 if numa.get("numa_node") is None else numa["numa_node"],
                        "cpus": len(os.sched_getaffinity(0))}, stats_device)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    t_max = job.max_wall_s()
    tokens = job.total("tokens")
    prop = job.total("proposed")
    value = job.tokens_per_s()
    ms_per_step = t_max / args.steps * 1e3
    bytes_step = (2 * tgt.matmul_bytes()) if medusa else (K * drf.matmul_bytes() + tgt.matmul_bytes())
    if heads is not None:
        bytes_step = tgt.matmul_bytes() + K * tgt.config.vocab * tgt.config.d_model * 2
    if wd == "fp8":
        bytes_step //= 2      # one byte per weight (+ 4 bytes per output row of scales: < 0.1 %)
    out = {
        "metric": "accepted tokens/sec + acceptance-rate, Llama-3.2-3B/1B K=4 @1/2/4/8 GPU",
        "value": value, "unit": "tokens/s", "n_gpus": int(job.per_rank.shape[0]),   # ranks that contributed to the all-gather
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if wd == "bf16" else "bf16 MFMA over fp8-e4m3 weight storage", "data": "synthetic",
        "config": {"workload": f"{args.target} target + {args.draft} draft, K={K}, batch {B}/GPU, {'sampled bonus token T=0.7 top_k=50 top_p=0.9' if args.do_sample else 'greedy'}{(', EAGLE-lite hidden-state extrapolation (self-draft), generate() loop' if args.draft_mode == 'eagle' else ', Medusa-lite tied heads (self-draft), generate() loop') if medusa else ''}{', persistent Medusa heads (synthetic)' if heads is not None else ''}, "
                               f"prompt {PROMPT_LEN} ids, hipGraph step, weights: {source}"
                               + (" (BASELINE config 4's per-GPU shape: 32 rows over 8 GPUs)" if args.target == "llama-3-8b" and B == 4 and not medusa else ""),
                   "K": K, "batch_per_gpu": B, "global_batch": B * world, "prompt_len": PROMPT_LEN,
                   "parallelism": f"dp{world}"},
        "acceptance_rate": job.acceptance(),                 # reference definition: bonus token counted
        "acceptance_rate_strict": job.acceptance(True),      # draft tokens accepted / proposed
        "tokens_per_step": tokens / (args.steps * world * B),
        "step_bytes": bytes_step,
        "step_roofline_frac": bytes_step / (ms_per_step / 1e3) / HBM_PEAK_BPS,
        "resyncs": sess.stats["resyncs"],
        # per rank: time per step (the job's figure is the slowest rank's), the NUMA node its threads were pinned to (-1: the
        # box gave no answer, mask left alone) and the CPUs left in its affinity mask (>= 1 by construction)
        "ranks": {"ms_per_step_min": min(job.column("wall_ns")) / 1e6 / args.steps, "ms_per_step_max": max(job.column("wall_ns")) / 1e6 / args.steps,
                  "numa_node": job.column("numa_node"), "cpus": job.column("cpus")},
        "rank0_numa": numa, "collective_backend": ("gloo (rehearsal: all ranks on cuda:0)" if rehearsal else "nccl (RCCL)") if world > 1 else None,
        # one-off H2D of the prompt ids of this rank's rows, measured, and the rate with it added to the timed region
        "h2d_prompt_us": h2d_s * 1e6,
        "value_pcie_inclusive": tokens / (t_max + h2d_s),
    }
    log(f"timed region: {ms_per_step:.3f} ms/step, {value:.1f} tok/s")
    # ---- roofline of the dominant kernel, timed live with HIP events --------------------
    if not args.no_probe:
        out["roofline"] = roofline_leg(sess, B, K, wd, medusa or heads is not None)
    # ---- CPU baseline (rank 0, N = 1 only) ---------------------------------------------------
    if world == 1 and args.cpu_baseline_steps > 0:
        gpu_rows = [list(r.generated) for r in sess.rows]
        cb, same = cpu_baseline(drf, tgt, prompts, K, gpu_rows, wd, seg_steps=args.cpu_baseline_steps, gpu_counters=list(sess.rows[0].counters))
        out["cpu_baseline"] = cb
        out["parity_with_cpu_sample"] = same
        out["speedup_vs_cpu_baseline"] = value / cb["value"] if cb["value"] > 0 else None
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
