#!/usr/bin/env python3
"""Headline benchmark: SO(3) reverse-diffusion sample-steps/s (BASELINE.json metric).

Workload (BASELINE config 3): B = 2^20 rotations per GPU, T = 1000-step reverse chain with
the RotPredict score network (bf16 MFMA operands, fp32 accumulate, fp32 rotation state),
synthetic random-quaternion inputs, seed-0 default-init weights, in-kernel Philox noise.
One "step" = one p_sample application to the whole batch; K steps run as consecutive
timesteps T-1, T-2, ... (wrapping after t = 0) inside the chain-resident kernel.

  python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment), and as the plain command above -- then this process starts the N ranks itself as
child processes (before anything here imports torch or touches a GPU; it never does), passes rank 0's JSON line through and
exits with the worst child's code.

Prints ONE JSON line on rank 0.  Multi-GPU = the batch axis sharded (weak scaling, 2^20 per
GPU), no data-path collective (samples are independent; Philox streams keyed by the global
sample index); the training leg all-reduces its flat gradient once per step (RCCL).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _self_launch(n_ranks, argv):
    """`python bench.py --gpus N` without a launcher: N children of this same command with the rendezvous in their environment
    (one rank per GPU, LOCAL_RANK = RANK; on a box with fewer GPUs than ranks the children share devices round-robin --
    parallel.init() -- which with SO3X_DIST_BACKEND=gloo exercises this path on a one-GPU box).  stdout of rank 0 is the job's
    stdout; the other ranks' output goes to stderr.  The parent stays a pure-Python supervisor: no torch import, no HIP call."""
    import socket
    with socket.socket() as sk:   # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SO3X_BENCH_CHILD="1")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n_ranks):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr, stderr=None))
    # poll ALL the ranks: the first one to fail ends the job at once (a rank that died in its imports would otherwise leave its
    # peers in the rendezvous until torch's timeout, minutes later) and its code is the job's
    rcs = [None] * n_ranks
    failed_at = None
    try:
        while any(rc is None for rc in rcs):
            for i, pr in enumerate(procs):
                if rcs[i] is None:
                    rcs[i] = pr.poll()
            if failed_at is None and any(rc not in (None, 0) for rc in rcs):
                failed_at = time.monotonic()
            # peers of a failed rank get two seconds to fail by themselves (they usually hit the same wall, and say so); then they
            # are ended -- in a collective whose partner is gone they would sit until torch's timeout
            if failed_at is not None and time.monotonic() - failed_at > 2.0:
                for pr in procs:   # exactly the processes started here
                    if pr.poll() is None:
                        pr.terminate()
                for pr in procs:
                    try:
                        pr.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        pr.kill()
                break
            time.sleep(0.05)
        rcs = [pr.poll() for pr in procs]
    except BaseException:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        raise
    return max((abs(rc) if rc is not None else 1 for rc in rcs), default=0)


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _ap = argparse.ArgumentParser(add_help=False)
    _ap.add_argument("--gpus", type=int, default=1)
    _n = _ap.parse_known_args()[0].gpus
    if _n > 1:
        sys.exit(_self_launch(_n, sys.argv[1:]))

import numpy as np
import torch

MLP_FLOP_PER_SAMPLE = 34190          # 2*(4*65*65 + 65*3), SURVEY.md 8d
BF16_MFMA_PEAK_TFLOPS = 2500.0       # dense, MI355X_MICROARCH.md chip table
HBM_PEAK_GBS = 8000.0                # spec, same table
IGSO3_BYTES_PER_EVAL = 56            # 36 R + 4 eps in, 4 logp + 12 score out, SURVEY.md 8d


def run_steps(B, params, sched, trap_p, x, T, nsteps, seed, index_base, precision, rng_offset=0, per_launch=100, guide_p=None, prepared=None):
    """nsteps consecutive reverse steps starting at t = T-1, wrapping; `per_launch` steps per kernel launch so
    that every launch (warmup and timed alike) does the same work and rocprof's per-kernel average duration is
    directly comparable with the number reported here.  prepared: the workspace of so3x_p_sample_prepare for these weights
    (built once by the caller: it depends on the weights and tables only) -- every launch then is the chain kernel alone."""
    done = 0
    launches = 0
    t = T - 1
    while done < nsteps:
        seg = min(nsteps - done, t + 1, per_launch)
        if prepared is not None:
            B.p_sample_prepared(prepared, sched, trap_p, x, t, seg, seed=seed, rng_offset=rng_offset + done, index_base=index_base,
                                precision=precision, out=x, guide_p=guide_p)
        else:
            B.p_sample_chain(params, sched, trap_p, x, t, seg, seed=seed, rng_offset=rng_offset + done, index_base=index_base,
                             precision=precision, out=x, guide_p=guide_p)
        done += seg
        launches += 1
        t = t - seg
        if t < 0:
            t = T - 1
    return launches


def cpu_baseline(T, params_np, betas, budget_s=6.0):
    """The oracle (C restatement, OpenMP over the batch, all host threads) timed on this box's host cores on a bounded sample of
    the same workload: 2^17 rotations x as many reverse steps around the middle of the chain as ~6 s of CPU work hold.  The timed
    build is the oracle's source compiled -O3 -march=native -fopenmp ON THIS HOST (oracle/Makefile: libso3_oracle_fast.so; the
    checker build stays -O2 -ffp-contract=off) and is called on preallocated arrays: only the C call sits in the timed region."""
    from oracle import oracle as O
    sched = np.ascontiguousarray(O.schedule_from_betas(betas), np.float32)
    trap_p = np.ascontiguousarray(O.igso3_build_tables(np.exp(np.float32(0.5) * sched[9])), np.float32)
    params_np = np.ascontiguousarray(params_np, np.float32).ravel()
    rng = np.random.default_rng(0)
    n = 1 << 17
    x = np.ascontiguousarray(O.quat_to_rmat(rng.standard_normal((n, 4)).astype(np.float32)), np.float32).reshape(n, 9)
    y = np.empty_like(x)
    axes = rng.standard_normal((n, 3)).astype(np.float32)
    unif = rng.random(n, dtype=np.float32)
    O.p_sample_step_timed(params_np, sched, trap_p, x, 500, axes, unif, y)  # builds the timed library on this host; warms the threads
    t0 = time.perf_counter()
    steps = 0
    tt = T // 2
    while True:
        O.p_sample_step_timed(params_np, sched, trap_p, x, tt, axes, unif, y)
        x, y = y, x
        steps += 1
        tt = tt - 1 if tt > 0 else T - 1
        el = time.perf_counter() - t0
        if el > budget_s or steps >= 1000:
            break
    return {"value": n * steps / el, "unit": "sample-steps/s", "cores": O.omp_threads(), "kind": "port",
            "host_cpus": os.cpu_count(), "torch_num_threads": torch.get_num_threads(), "build": O.FAST_FLAGS,
            "sample": f"{n} rotations x {steps} reverse steps (t from {T // 2} down), {el:.1f} s of CPU work, {O.omp_threads()} OpenMP threads, "
                      f"oracle source built {O.FAST_FLAGS} on this host"}


class Pmc:
    """The committed rocprofv3 --pmc passes (profiles/pmc_traffic.json, written by tools/summarize_profiles.py from a
    tools/profile_round.sh run).  PMC counters cannot be read from inside this process, so every PMC-derived field of the line is a
    QUOTE from that file -- and says so: `pmc_source` names the profile (round tag, the git commit it was summarised at) and
    whether the kernel sources are still the ones it was taken on (sha256 over csrc/ + include/so3x.h, recorded on the GPU box at
    profile time).  When they are not, nothing is quoted: a stale profile must not price a new kernel."""

    def __init__(self):
        self.data, self.meta, self.fresh, self.fresh_units = {}, {}, False, {}
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                self.data = json.load(f)
            self.meta = self.data.get("_meta", {})
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from csrc_digest import digest, unit_digests, kernel_unit
            self._kernel_unit = kernel_unit
            self.fresh = bool(self.meta.get("csrc_sha256")) and self.meta["csrc_sha256"] == digest(ROOT)
            # per translation unit: a kernel's counters stay quotable while ITS unit (and the headers it includes) is unchanged
            now, then = unit_digests(ROOT), self.meta.get("csrc_sha256_by_unit") or {}
            self.fresh_units = {u: then.get(u) == h for u, h in now.items()}
        except (OSError, ValueError, ImportError):
            pass

    def is_fresh(self, kernel):
        if self.fresh:
            return True
        unit = self._kernel_unit(kernel) if self.data else None
        return bool(unit) and self.fresh_units.get(unit, False)

    @property
    def source(self):
        if not self.data:
            return "none: profiles/pmc_traffic.json missing"
        tag = f"committed profile {self.meta.get('tag', '?')} (profiles/pmc_traffic.json, summarised at git {self.meta.get('git', '?')})"
        if self.fresh:
            return tag + ": kernel sources unchanged since it was taken"
        stale = sorted(u for u, ok in self.fresh_units.items() if not ok)
        return tag + (": units changed since it was taken: " + ", ".join(stale) + " -- PMC-derived fields of THEIR kernels withheld"
                      if self.fresh_units and len(stale) < len(self.fresh_units) else
                      ": STALE -- the kernel sources changed since it was taken; PMC-derived fields withheld")

    def traffic(self, kernel, **match):
        """HBM bytes per launch (FETCH_SIZE doubled per the gfx950 rule of MI355X_MICROARCH.md + WRITE_SIZE), only when the
        profiled configuration matches this run"""
        rec = self.data.get(kernel) if self.is_fresh(kernel) else None
        if rec and all(rec.get("config", {}).get(k) == v for k, v in match.items()):
            return rec["hbm_bytes_per_launch"]
        return None

    def counter(self, kernel, name):
        try:
            return self.data["mfma_utilisation"][kernel][name] if self.is_fresh(kernel) else None
        except KeyError:
            return None


PMC = Pmc()


def pmc_traffic(kernel, **match):
    return PMC.traffic(kernel, **match)


def pmc_mfma_busy(kernel):
    return PMC.counter(kernel, "mfma_pipe_busy_frac")


def pmc_counter(kernel, name):
    return PMC.counter(kernel, name)


def isa_mix(name):
    """instruction mix of a kernel's hot loop (profiles/<tag>_<name>_isa_mix.json: tools/count_isa.py on the committed sources,
    written by tools/summarize_profiles.py), or None when the sources changed since"""
    try:
        with open(os.path.join(ROOT, "profiles", f"{PMC.meta.get('tag', 'r04')}_{name}_isa_mix.json")) as f:
            return json.load(f) if PMC.is_fresh("k_p_sample_chain") else None
    except (OSError, ValueError):
        return None


class RawChain:
    """so3x_p_sample_chain through the raw C ABI with a workspace this process owns, so that the two clock words every launch
    leaves there (so3x_p_sample_clock_offset: shader-clock ticks and 100 MHz reference ticks of wave 0) can be read back: the
    clock the timed launches actually ran at.  Same entry point, same arguments as the operator the headline loop calls."""

    def __init__(self, B, torch, T, prec):
        import ctypes as C
        self.C, self.torch, self.lib, self.T, self.prec = C, torch, B.lib(), T, prec
        dev = torch.device("cuda", torch.cuda.current_device())
        self.nb = int(self.lib.so3x_p_sample_workspace_bytes(C.c_int(T), C.c_int(prec)))
        self.ws = torch.zeros(self.nb, dtype=torch.uint8, device=dev)
        self.off = int(self.lib.so3x_p_sample_clock_offset(C.c_int(T), C.c_int(prec)))

    def __call__(self, params, sched, trap_p, guide_p, x, t_start, n_steps, seed, rng_offset, index_base):
        C, P = self.C, (lambda t: self.C.c_void_p(t.data_ptr()))
        rc = self.lib.so3x_p_sample_chain(C.c_void_p(self.torch.cuda.current_stream().cuda_stream), P(params), P(sched), C.c_int(self.T), P(trap_p),
                                          P(guide_p), P(x), P(x), C.c_int(t_start), C.c_int(n_steps), None, None, C.c_uint64(seed),
                                          C.c_uint64(rng_offset), C.c_int64(index_base), C.c_int64(x.numel() // 9), C.c_int(self.prec),
                                          P(self.ws), C.c_size_t(self.nb))
        assert rc == 0, rc

    def clock_ghz(self):
        ticks, ref = self.ws[self.off:self.off + 16].view(self.torch.int64).tolist()
        return ticks / ref * 0.1 if ref > 0 else None


def pmc_valu_busy(kernel):
    return PMC.counter(kernel, "valu_busy_frac")


def train_leg(B, torch, ctx, T, n=1 << 19, reps=50, warm=20):
    """BASELINE config 4 on every rank: one training step of so3_train.py (noising + RotPredict + MSE + backward + gradient
    all-reduce + Adam) on this GPU's shard of 2^19 rotations (global batch 2^19 x N; 2^22 at N = 8), bf16 MLP operands,
    replayed as a captured hipGraph (so3x.graphs.TrainStepGraph: the RCCL all-reduce of the flat 69 KB gradient inside the
    graph, or between two graphs where the stack cannot capture it).  Timed like the headline: barrier, `reps` replays,
    synchronize, max over ranks.  `allreduce_us` = the collective alone, timed on the same flat buffer."""
    import torch.distributed as dist
    from so3x import optim as so3x_optim
    from so3x.graphs import TrainStepGraph
    from so3x.diffusion import SO3Diffusion
    from so3x.so3_train import RotPredict
    from so3x import parallel
    dev = ctx.device
    torch.manual_seed(0)
    net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
    parallel.broadcast_parameters(net, ctx)
    proc = SO3Diffusion(net, timesteps=T).to(dev)
    proc.index_base = ctx.rank * n
    opt = so3x_optim.Adam(net, lr=3e-4)
    x0 = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + ctx.rank)))

    def barrier():
        torch.cuda.synchronize()
        if ctx.world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def wall(fn, reps):
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        barrier()
        tm = torch.tensor([el, -el], device=dev, dtype=torch.float64)
        if ctx.world_size > 1:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        wall.fastest_rank = -float(tm[1].item()) / reps
        return float(tm[0].item()) / reps

    out = {"batch_per_gpu": n, "global_batch": n * ctx.world_size, "ranks": ctx.world_size, "mlp_operands": "bf16",
           "optimizer": "so3x.optim.Adam (one launch on the flat buffers; torch.optim.Adam's update rule)",
           "collective": None if ctx.world_size == 1 else f"{dist.get_backend()} all_reduce of {net.flat_data().numel()} fp32, once per step"}
    if ctx.world_size == 1:  # the eager Python loop beside it (host-bound)
        def eager():
            loss = proc(x0)
            opt.zero_grad()
            loss.backward()
            opt.step()
        for _ in range(3):
            eager()
        out["eager_python_loop_ms_per_step"] = wall(eager, 10) * 1e3
    other_ms = staged_ms = None
    if ctx.world_size == 1:  # round 3's forms beside it (three kernels with the pre-activation stash in HBM): the pipelined stages and
        for form in (True, "staged"):  # the stages in one stream with reduction + Adam as one launch (five launches)
            ts = TrainStepGraph(proc, opt, x0.shape, ctx=ctx, n_global=n * ctx.world_size, pipeline=form)
            for _ in range(warm):
                ts.replay()
            ms_ = wall(ts.replay, reps) * 1e3
            ts.flush()
            del ts
            if form is True:
                other_ms = ms_
            else:
                staged_ms = ms_
    tg = TrainStepGraph(proc, opt, x0.shape, ctx=ctx, n_global=n * ctx.world_size)
    for _ in range(warm):
        tg.replay()
    sec = wall(tg.replay, reps)
    tg.flush()
    loss = parallel.mean_scalar(tg.loss.clone(), ctx)
    out.update({"ms_per_step": sec * 1e3, "ms_per_step_fastest_rank": wall.fastest_rank * 1e3, "samples_per_s": n * ctx.world_size / sec,
                "mode": {"in_graph": "one captured hipGraph per step" + ("" if ctx.world_size == 1 else ", all-reduce inside"),
                         "split": "captured hipGraphs with the eager all-reduce between them"}[tg.mode],
                "pipelined": tg.pipelined, "one_kernel": tg.fused,
                "form": ("one-kernel step: prep -> k_train_fused (noising + forward + loss + backward) -> " +
                         ("[slab reduction + Adam]" if ctx.world_size == 1 else "slab reduction -> all-reduce -> Adam")) if tg.fused
                        else ("pipelined stages" if tg.pipelined else "serial graph of six launches"),
                "round3_staged_five_launch_ms_per_step_at_one_gpu": staged_ms,
                "round3_pipelined_ms_per_step_at_one_gpu": other_ms,
                "algorithmic_TFLOPs_per_gpu": 94120 * n / sec / 1e12, "loss": loss, "finite": bool(loss == loss), "steps_timed": reps})
    try:
        out["kernel"] = fused_kernel_leg(B, torch, proc, net, x0, n, T)
    except Exception as e:  # report, never hide
        out["kernel"] = {"error": repr(e)}
    if ctx.world_size > 1:
        flat = net.gather_flat_grad()
        def ar():
            parallel.allreduce_flat(flat, ctx)
        for _ in range(5):
            ar()
        out["allreduce_us"] = wall(ar, 50) * 1e6
    return out


TRAIN_FLOP_PER_SAMPLE = 94120   # 34,190 forward + 59,930 backward (dW + dX of layers 2-5), SURVEY.md 8d


def fused_kernel_leg(B, torch, proc, net, x0, n, T, reps=20):
    """so3x_train_fused alone (the prep launch + k_train_fused) through the raw C ABI, HIP events around `reps` back-to-back calls:
    the dominant kernel of the training step against the bf16 MFMA peak on its ALGORITHMIC flops (94,120 per sample) and its
    algorithmic HBM bytes (36 B of x_0 per sample in; one 70 KB partial slab per workgroup out)."""
    import ctypes as C
    lib = B.lib()
    trap_q, _ = proc._tables()
    params = net.flat_data()
    loss = torch.zeros(1, device=x0.device)
    ctr = torch.zeros(1, dtype=torch.int64, device=x0.device)
    ws = torch.empty(int(lib.so3x_train_workspace_bytes(C.c_int64(n), C.c_int(T))), dtype=torch.uint8, device=x0.device)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    def call():
        rc = lib.so3x_train_fused(C.c_void_p(torch.cuda.current_stream().cuda_stream), P(params), P(proc._sched), C.c_int(T), P(trap_q),
                                  P(proc._guide_q), P(x0), None, None, C.c_int(1), None, None, C.c_uint64(1), C.c_uint64(0), P(ctr), C.c_int64(0),
                                  C.c_int64(n), P(loss), None, None, P(ws), C.c_size_t(ws.numel()))
        assert rc == 0, rc
    for _ in range(5):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tf = TRAIN_FLOP_PER_SAMPLE * n / (ms * 1e-3) / 1e12
    slab_bytes = 256 * 17556 * 4
    return {"kernel": "k_train_fused (+ k_prep: weight images and per-timestep tables)", "ms_per_call": ms, "bound": "mfma (nominal); LDS bandwidth + one wave's issue rate in fact (DESIGN.md section 4)",
            "achieved": tf, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / BF16_MFMA_PEAK_TFLOPS, "flop_per_sample": TRAIN_FLOP_PER_SAMPLE,
            "algorithmic_hbm_bytes_per_call": 36 * n + slab_bytes, "traffic": pmc_traffic("k_train_fused", n=n),
            "mfma_pipe_busy_frac_pmc": pmc_mfma_busy("k_train_fused"), "pmc_source": PMC.source, "timing": "HIP events around back-to-back raw C-ABI calls",
            "finite": bool(torch.isfinite(loss).item())}


def wide_net_extra(B, torch, sched, trap_p, n=1 << 18, steps=50, reps=3):
    """the 255-wide residual score network of so3_lock_train.py as the chain's denoiser (the shape
    where the matrix cores are the bound) and one full training step with it.  Algorithmic flops: 781,830 per sample."""
    from so3x.so3_lock_train import RotPredict as WideNet
    from so3x.diffusion import SO3Diffusion
    dev = torch.device("cuda", torch.cuda.current_device())
    flop = 6 * 2 * 255 * 255 + 2 * 255 * 3  # multiply-adds of the 7 linear layers, as MLP_FLOP_PER_SAMPLE counts the 65-wide net
    torch.manual_seed(0)
    wnet = WideNet(out_type="skewvec", precision="bf16").to(dev)
    params = wnet.flat_params_nograd()
    x = B.quat_to_rmat(torch.randn(n, 4, device=dev))

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    guide_p = B.igso3_build_guide(trap_p)
    ms = timed(lambda: B.resnet_p_sample_chain(params, sched, trap_p, x, 600, steps, seed=1, precision=B.PREC_BF16,
                                               guide_p=guide_p), reps)
    tf = flop * n * steps / (ms * 1e-3) / 1e12
    out = {"chain": {"kernel": "k_resnet_chain", "batch": n, "steps_per_launch": steps, "ms_per_launch": ms,
                     "sample_steps_per_s": n * steps / (ms * 1e-3), "bound": "mfma", "achieved": tf,
                     "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / BF16_MFMA_PEAK_TFLOPS,
                     "mfma_pipe_busy_frac_pmc": pmc_mfma_busy("k_resnet_chain"), "flop_per_sample_step": flop}}
    nt = 1 << 19
    proc = SO3Diffusion(wnet, timesteps=sched.shape[1]).to(dev)
    x0 = B.quat_to_rmat(torch.randn(nt, 4, device=dev))
    from so3x import optim as so3x_optim
    from so3x.graphs import TrainStepGraph
    opt = so3x_optim.Adam(wnet, lr=3e-4)   # the product's optimizer, as the 65-wide leg: one launch on the flat 392,448-float buffers
    tg = TrainStepGraph(proc, opt, x0.shape)   # serial form (the pipelined stages are the 65-wide network's)
    for _ in range(10):   # the clocks settle over the first ~20 ms under this load (the step had been timed on its replays 5..9)
        tg.replay()
    ms = min(timed(tg.replay, 10) for _ in range(3))
    out["train_step"] = {"batch": nt, "ms_per_step": ms, "timing": "best of three passes of ten replays behind ten untimed ones", "samples_per_s": nt / (ms * 1e-3), "operands": "bf16",
                         "algorithmic_TFLOPs": 3 * flop * nt / (ms * 1e-3) / 1e12, "optimizer": "so3x.optim.Adam",
                         "frac_of_bf16_mfma_peak": 3 * flop * nt / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                         "mode": "one captured hipGraph per step (so3x.graphs.TrainStepGraph)", "finite": bool(torch.isfinite(tg.loss).item())}
    # the step's two device calls timed apart (HIP events around the operators): what bounds each is the HBM traffic of the
    # register dumps -- 512 B per sample, layer and stream (X_l: 7, Y_l: 6, dZ_l: 7) -- not the matrix pipe
    tt = torch.randint(0, sched.shape[1], (nt,), device=dev)
    dout = torch.randn(nt, 3, device=dev)
    _, stash = B.resnet_fwd_stash(params, x0, tt, sched.shape[1], B.PREC_BF16)
    # (best of three passes of ten calls: the operators allocate their outputs, and the first calls after a switch of kernels run
    #  with cold instruction and memory-side caches)
    ms_f = min(timed(lambda: B.resnet_fwd_stash(params, x0, tt, sched.shape[1], B.PREC_BF16), 10) for _ in range(3))
    ms_b = min(timed(lambda: B.resnet_bwd(params, x0, tt, dout, sched.shape[1], B.PREC_BF16, stash=stash), 10) for _ in range(3))
    kb = 512 * nt / 1e9   # GB per dumped stream
    out["train_step"]["stages"] = {
        "forward_with_dumps (k_resnet_fwd<bf16, stash>)": {
            "ms": ms_f, "frac_of_bf16_mfma_peak": flop * nt / (ms_f * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS,
            "hbm_GB_written": 13 * kb, "hbm_TBps": 13 * kb / ms_f, "frac_of_hbm_peak": 13 * kb / ms_f / (HBM_PEAK_GBS / 1e3)},
        "backward (k_resnet_bwd: dX chain, reads Y, writes dZ; k_resnet_dw: dW GEMM over samples, reads X and dZ; reduce)": {
            "ms": ms_b, "frac_of_bf16_mfma_peak": 2 * flop * nt / (ms_b * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS,
            "hbm_GB_moved": 27 * kb, "hbm_TBps": 27 * kb / ms_b, "frac_of_hbm_peak": 27 * kb / ms_b / (HBM_PEAK_GBS / 1e3)},
        "per_kernel_durations": "profiles/r04_bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats of this script): k_resnet_fwd<1, true>, k_resnet_bwd, k_resnet_dw",
        "hbm_bytes_per_launch_pmc": {k: PMC.traffic(k, n=nt) for k in ("k_resnet_fwd", "k_resnet_bwd", "k_resnet_dw")},
        "pmc_source": PMC.source,
        "streaming_rates_of_this_hardware_TBps": "fill 6.9, read 5.3-6.3, copy 5.4 (profiles/r04_hbm_rates.json, tools/ab/read_rate.hip)"}
    return out


def igso3_eval_roofline(B, torch, n=1 << 20, reps=50, eps_input="schedule", sched=None):
    """BASELINE config 2: IGSO(3) log-density + score, HBM-bound kernel, on the inputs BASELINE.md section 4 states:
    eps_input "scalar" = (2a) one eps = 0.5 for the whole batch; "schedule" = (2b) eps_i = sqrt(1 - abar_{t_i}) from the
    schedule, t_i = randint(0, T) seed 0 -- the eps the training loss sees, including the small-eps rows where most
    rotations sit far out in the density's tail; "uniform" = eps ~ U(0.1, 1), round 2's easier input, kept for continuity
    and labelled as NOT a BASELINE input.  The C ABI is called directly with preallocated outputs, captured once into a HIP
    graph and replayed, so the events bracket back-to-back kernel launches (the Python wrapper's per-call allocation
    otherwise leaves the GPU idle between 15-us kernels)."""
    import ctypes as C
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev).manual_seed(0)
    R = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
    if eps_input == "scalar":
        eps, eps_stride = torch.full((1,), 0.5, device=dev), 0
    elif eps_input == "schedule":
        ti = torch.randint(0, sched.shape[1], (n,), device=dev, generator=g)
        eps, eps_stride = sched[4][ti].contiguous(), 1      # sqrt_one_minus_alphas_cumprod[t_i]
    else:
        eps, eps_stride = torch.rand(n, device=dev, generator=g) * 0.9 + 0.1, 1
    bytes_per_eval = IGSO3_BYTES_PER_EVAL - (4 if eps_stride == 0 else 0)
    logp = torch.empty(n, device=dev)
    score = torch.empty(n, 3, device=dev)
    lib = B.lib()

    def launch():
        rc = lib.so3x_igso3_logprob_score(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(R.data_ptr()),
                                          C.c_void_p(eps.data_ptr()), C.c_int64(eps_stride), C.c_void_p(logp.data_ptr()),
                                          C.c_void_p(score.data_ptr()), None, C.c_int64(n))
        assert rc == 0, rc

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            launch()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            launch()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gbs = bytes_per_eval * n / (ms * 1e-3) / 1e9
    finite = float(torch.isfinite(logp).float().mean().item())
    return {"kernel": "k_logprob_score", "eps_input": {"scalar": "2a: eps = 0.5 for every sample", "uniform": "eps ~ U(0.1, 1) (round 2's input; NOT a BASELINE input)",
                                                        "schedule": "2b: eps_i = sqrt(1 - abar_t_i), t_i = randint(0, T), seed 0"}[eps_input],
            "evals_per_s": n / (ms * 1e-3), "bound": "hbm", "achieved": gbs,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "n": n, "ms": ms,
            "bytes_per_eval": bytes_per_eval, "launches": reps, "timing": "HIP events around a graph replay",
            "finite_logp_frac": finite, "traffic": pmc_traffic("k_logprob_score:" + eps_input, n=n, eps_input=eps_input)}


def se3_legs(B, torch, reps=20):
    """BASELINE config 5's device work per GPU (prot_train.py's SE(3) path; reference prot_util.py:73-81, diffusion.py:432-522):
    `k_rigid_move` -- every residue of 4096 structures x 256 residues moved by its structure's rigid transform (positions
    AND frames: 48 B in + 48 B out per residue) -- and `se3_q_sample_target` (IGSO3 x R^3 noising + both regression targets,
    2^20 frames: 36 + 12 + 8 B in, 36 + 12 + 12 + 12 B out).  HIP events around `reps` back-to-back launches."""
    from so3x.se3 import SE3Diffusion, AffineGrad
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev).manual_seed(0)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    S, L = 4096, 256
    rot = B.quat_to_rmat(torch.randn(S, 4, device=dev, generator=g))
    shift = torch.randn(S, 3, device=dev, generator=g)
    pos = torch.randn(S, L, 3, device=dev, generator=g)
    frames = B.quat_to_rmat(torch.randn(S, L, 4, device=dev, generator=g))
    ms = timed(lambda: B.rigid_move(rot, shift, pos, frames))
    gbs = 96 * S * L / (ms * 1e-3) / 1e9
    out = {"rigid_move": {"kernel": "k_rigid_move", "structures": S, "residues_per_structure": L, "bytes_per_residue": 96, "ms": ms,
                          "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                          "timing": "HIP events around back-to-back operator calls (includes their output allocation)",
                          "traffic": pmc_traffic("k_rigid_move", structures=S, residues=L)}}
    n = 1 << 20
    proc3 = SE3Diffusion(lambda x, t: AffineGrad(x.rot[..., 0], x.shift), timesteps=1000).to(dev)
    tq, _ = proc3._tables()
    xr = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
    xs = torch.randn(n, 3, device=dev, generator=g)
    tt = torch.randint(0, 1000, (n,), device=dev, generator=g)
    # the C ABI directly with preallocated outputs, `reps` launches captured into one hipGraph (as the igso3_eval leg): through the
    # operator each call allocates its four outputs, and that host time -- ~30 us, about the kernel's own -- was inside the events
    # (rounds 2-3 reported 67 us = 24.6 % for a 40-us kernel)
    import ctypes as C
    lib = B.lib()
    o = [torch.empty(n, 3, 3, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)]
    P = lambda a: C.c_void_p(a.data_ptr())  # noqa: E731

    def launch():
        rc = lib.so3x_se3_q_sample_target(C.c_void_p(torch.cuda.current_stream().cuda_stream), P(proc3._sched), C.c_int(1000), P(tq), P(proc3._guide_q),
                                          C.c_float(75.0), P(xr), P(xs), P(tt), C.c_int(1), None, None, None, C.c_uint64(1), C.c_uint64(0),
                                          C.c_int64(0), P(o[0]), P(o[1]), P(o[2]), P(o[3]), C.c_int64(n))
        assert rc == 0, rc
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            launch()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            launch()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ms_op = timed(lambda: B.se3_q_sample_target(proc3._sched, tq, 75.0, xr, xs, tt, seed=1))
    nb = 36 + 12 + 8 + 36 + 12 + 12 + 12
    gbs = nb * n / (ms * 1e-3) / 1e9
    out["se3_q_sample_target"] = {"kernel": "k_se3_q_sample_target", "n": n, "bytes_per_frame": nb, "ms": ms, "bound": "hbm (nominal; per-sample CDF-row gathers + VALU in fact)",
                                  "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                  "timing": "HIP events around a graph replay of raw C-ABI launches on preallocated outputs",
                                  "ms_through_the_operator_with_output_allocation": ms_op,
                                  "finite": bool(torch.isfinite(o[0]).all().item()),
                                  "traffic": pmc_traffic("k_se3_q_sample_target", n=n)}
    return out


def external_loop_leg(torch, proc, x, T, calls=1000):
    """The reference's own calling pattern (so3_test.py:24-31): ONE p_sample call per reverse step from a Python loop, t a
    (1,)-shaped device tensor -- here every call is one kernel launch from the prepared state (so3x_p_sample_prepared), with the
    timestep read on the device, so the loop never synchronises.  Wall clock around `calls` calls + one synchronize."""
    dev = x.device
    y = x.clone()
    for i in reversed(range(T - 20, T)):   # builds the prepared state, warms the allocator
        y = proc.p_sample(y, torch.full((1,), i, device=dev, dtype=torch.long))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(calls):
        y = proc.p_sample(y, torch.full((1,), T - 1 - (k % T), device=dev, dtype=torch.long))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    n = x.numel() // 9
    return {"what": "for i in reversed(range(T)): x = process.p_sample(x, torch.full((1,), i)) -- one launch per call from the prepared state, t read on the device",
            "batch": n, "calls": calls, "ms_per_call": el / calls * 1e3, "sample_steps_per_s": n * calls / el, "finite": bool(torch.isfinite(y).all().item())}


def planenet_leg(torch, reps=10):
    """the PlaneNet denoiser (reference models.py:185-210; aircraft_rotate.py:17-47: batch 32, 256 points,
    dim 512, 4 heads, 4 layers) on this package's kernels, bf16 form: the forward (what ProjectedSO3Diffusion.p_sample and the
    validation pass run) and one training evaluation (forward with stash + backward), at the reference's default shape and at
    32 x 2048 points.  Algorithmic flops: every multiply-add of the Linear layers and of attention's two products, x 2 (backward:
    x 2 more for the Linear layers, x 2.5 more for attention, whose probabilities are recomputed)."""
    from so3x.models import PlaneNet
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(0)
    net = PlaneNet(precision="bf16", dropout=0.0).to(dev).eval()

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    out = {}
    for Bn, P in ((32, 256), (32, 2048)):
        x = torch.randn(Bn, P, 3, device=dev) * 0.5
        t = torch.randint(0, 1000, (Bn,), device=dev)
        n = Bn * P
        per_tok = 2 * 256 * 256 + 4 * (2 * 512 * 1536 + 4 * P * 512 + 2 * 512 * 512 + 4 * 512 * 2048)
        flop = n * per_tok + Bn * (2 * 512 * 512 + 6 * 512)
        attn = n * 4 * 4 * P * 512
        with torch.no_grad():
            for _ in range(3):
                net(x, t)
            ms = min(timed(lambda: net(x, t), reps) for _ in range(3))
        tf = flop / ms / 1e9
        key = f"{Bn}x{P}"
        out["forward_" + key] = {"kernel": "so3x_planenet_fwd (bf16): k_gemm256_bf16 / k_gemm_bf16, k_attn_fwd, k_ln_bf16, ...", "bound": "mfma", "clouds": Bn,
                                 "points": P, "ms": ms, "flop": flop, "achieved": tf, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": tf / BF16_MFMA_PEAK_TFLOPS, "timing": "HIP events around back-to-back operator calls",
                                 "mfma_pipe_busy_frac_pmc": {k: pmc_mfma_busy(k) for k in ("k_gemm256_bf16", "k_gemm_bf16", "k_attn_fwd")},
                                 "pmc_source": PMC.source}
        net.train()
        dout = torch.randn(Bn, 3, device=dev)

        def step():
            net.zero_grad(set_to_none=True)
            (net(x, t) * dout).sum().backward()
        for _ in range(2):
            step()
        ms_t = min(timed(step, max(2, reps // 2)) for _ in range(2))
        ms_d = None
        if P == 2048:   # the same evaluation as the reference trains it: nn.TransformerEncoderLayer's default dropout 0.1 (aircraft_rotate.py:66)
            net.dropout = 0.1
            for _ in range(2):
                step()
            ms_d = min(timed(step, max(2, reps // 2)) for _ in range(2))
            net.dropout = 0.0
        net.eval()
        flop_t = 3 * (flop - attn) + 3.5 * attn
        out["train_eval_" + key] = {"kernel": "so3x_planenet_fwd (stash) + so3x_planenet_bwd (bf16)", "bound": "mfma", "clouds": Bn, "points": P, "ms": ms_t,
                                    "flop": flop_t, "achieved": flop_t / ms_t / 1e9, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": flop_t / ms_t / 1e9 / BF16_MFMA_PEAK_TFLOPS, "timing": "HIP events; through autograd (zero_grad, forward, backward)"}
        if ms_d is not None:
            out["train_eval_" + key]["ms_with_dropout_0.1"] = ms_d
    return out


def protnet_leg(torch, complexes=4096, rec_len=198, lig_len=58, reps=10):
    """BASELINE config 5's denoiser: ProtNet (reference models.py:212-319, prot_train.py:75-104) at 4096 complexes x 256 residues (a
    198-residue receptor + a 58-residue ligand each: the BPTI docking set's typical split), class-default widths.  `forward`: the bf16
    matrix-core form (what ProjectedSE3Diffusion's reverse chain runs per step).  `train_eval`: forward with stash + backward in the
    exact-fp32 form (the form that has a backward) on 256 complexes, priced against the exact-fp32 MFMA peak (v_mfma_f32_32x32x2_f32:
    256 flop / clk / CU = 157 TFLOP/s).  Algorithmic flops: every multiply-add of the Conv1d / Linear layers and of attention's two
    products over each chain's OWN length, x 2 (backward: x 2 more; attention x 2.5: the exact form keeps its probabilities)."""
    from so3x import backend as B
    from so3x.models import ProtNet
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev).manual_seed(1)

    def make(n):
        def chains(L):
            res = torch.zeros(n * L, 21, device=dev)
            res[torch.arange(n * L, device=dev), torch.randint(0, 21, (n * L,), device=dev, generator=g)] = 1.0
            pos = torch.randn(n * L, 3, device=dev, generator=g) * 8.0
            ang = B.quat_to_rmat(torch.randn(n * L, 4, device=dev, generator=g)).reshape(n * L, 9)
            return (res, pos, ang), torch.arange(0, n * L + 1, L, device=dev, dtype=torch.int64)
        rec, roff = chains(rec_len)
        lig, loff = chains(lig_len)
        return B.ProtBatch(rec, lig, roff, loff, max(rec_len, lig_len), [(rec_len, lig_len)] * n), torch.randint(0, 1000, (n,), device=dev, generator=g)

    def flops(n, dim=64, t_depth=4, c_depth=3, ffn=2048):
        lin = attn = 0
        for L in (rec_len, lig_len):
            conv = 21 * dim * 3 + (c_depth - 2) * dim * dim * 3 + dim * (dim - dim // 2 - dim // 4) * 3
            siren = 3 * (dim // 2) + (dim // 2) ** 2 + 9 * (dim // 4) + (dim // 4) ** 2
            lin += 2 * n * L * (conv + siren + t_depth * (4 * dim * dim + 2 * dim * ffn) + 2 * dim)
            attn += 2 * n * L * t_depth * 2 * L * dim
        return lin, attn

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    torch.manual_seed(0)
    net = ProtNet(precision="bf16").to(dev).eval()
    batch, t = make(complexes)
    with torch.no_grad():
        for _ in range(2):
            out = net(batch, t)
        ms = min(timed(lambda: net(batch, t), reps) for _ in range(3))
    lin, attn = flops(complexes)
    tf = (lin + attn) / ms / 1e9
    res = {"forward": {"kernel": "so3x_protnet_fwd (bf16): k_embed, 4 x (k_attn, k_ffn), k_poolb, head", "bound": "mfma", "complexes": complexes,
                       "residues": complexes * (rec_len + lig_len), "chain_lengths": [rec_len, lig_len], "ms": ms, "flop": lin + attn, "achieved": tf,
                       "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / BF16_MFMA_PEAK_TFLOPS, "finite": bool(torch.isfinite(out.rot_g).all()),
                       "timing": "HIP events around back-to-back module calls"}}
    n_t = 256
    tb, tt = make(n_t)
    net.train()
    dout = torch.randn(n_t, 6, device=dev)

    def step():
        net.zero_grad(set_to_none=True)
        o = net(tb, tt)
        (torch.cat((o.rot_g, o.shift_g), -1) * dout).sum().backward()
    step()
    ms_t = min(timed(step, 3) for _ in range(2))          # net.train() with the reference's dropout 0.1 (prot_train.py:75)
    net.dropout = 0.0
    step()
    ms_plain = min(timed(step, 3) for _ in range(2))
    net.dropout = 0.1
    lin_t, attn_t = flops(n_t)
    fl_t = 3 * lin_t + 3.5 * attn_t
    FP32_MFMA_PEAK = 157.0
    res["train_eval"] = {"kernel": "so3x_protnet_fwd (stash) + so3x_protnet_bwd, exact-fp32 form", "bound": "mfma (fp32)", "complexes": n_t,
                         "residues": n_t * (rec_len + lig_len), "ms": ms_t, "ms_without_dropout": ms_plain, "dropout": 0.1, "flop": fl_t, "achieved": fl_t / ms_t / 1e9, "peak": FP32_MFMA_PEAK,
                         "unit": "TFLOP/s", "frac": fl_t / ms_t / 1e9 / FP32_MFMA_PEAK,
                         "note": "the exact-fp32 form pads every chain to the longest one and keeps [chains][heads][L][L] probabilities: the parity "
                                 "form, not a throughput form; a bf16 backward is not built",
                         "timing": "HIP events; through autograd (zero_grad, forward, backward)"}
    # the batch prot_train.py itself defaults to (--batch 4): 1,584 padded residues -- a regime of ~470 small launches, not of flops
    n_s = 4
    tb, tt = make(n_s)
    dout = torch.randn(n_s, 6, device=dev)
    step()
    res["train_eval_reference_batch"] = {"complexes": n_s, "ms": min(timed(step, 10) for _ in range(3)), "dropout": 0.1,
                                         "note": "prot_train.py:22 default batch; bound by launches and per-kernel latency (tools/ab/protnet_small_batch.py)",
                                         "timing": "HIP events; through autograd (zero_grad, forward, backward)"}
    return res


def secondary_rooflines(line):
    """The other kernels' rooflines in ONE place under `roofline` (the driver's record keeps `roofline` and `cpu_baseline` whole and
    reduces every other key to its name): per leg the kernel, its bound, achieved / peak / frac, the launch time and the
    algorithmic work per launch the rate is computed from -- enough to recompute every fraction from this object alone."""
    def pick(d, *keys, **extra):
        if not isinstance(d, dict) or "error" in d:
            return {"error": (d or {}).get("error", "leg did not run") if isinstance(d, dict) else "leg did not run"}
        out = {k: d[k] for k in keys if k in d}
        out.update(extra)
        return out
    sec = {}
    ig = line.get("igso3_eval")
    if isinstance(ig, dict) and "error" not in ig:
        sec["igso3_eval (BASELINE config 2, per-sample eps from the schedule)"] = pick(
            ig, "kernel", "bound", "achieved", "peak", "unit", "frac", "ms", "n", "bytes_per_eval", "traffic",
            algorithmic_bytes_per_launch=ig["bytes_per_eval"] * ig["n"])
        if "scalar_eps_0p5" in ig:
            sc = ig["scalar_eps_0p5"]
            sec["igso3_eval (config 2a, scalar eps = 0.5)"] = pick(sc, "kernel", "bound", "achieved", "peak", "unit", "frac", "ms", "n", "bytes_per_eval")
    elif ig is not None:
        sec["igso3_eval"] = pick(ig)
    tr = line.get("train_step")
    if isinstance(tr, dict) and "error" not in tr:
        sec["train_step (BASELINE config 4 per GPU), whole step"] = {
            "bound": "mfma", "ms_per_step": tr.get("ms_per_step"), "batch_per_gpu": tr.get("batch_per_gpu"), "flop_per_sample": TRAIN_FLOP_PER_SAMPLE,
            "achieved": tr.get("algorithmic_TFLOPs_per_gpu"), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": None if tr.get("algorithmic_TFLOPs_per_gpu") is None else tr["algorithmic_TFLOPs_per_gpu"] / BF16_MFMA_PEAK_TFLOPS,
            "eager_python_loop_ms_per_step": tr.get("eager_python_loop_ms_per_step")}
        sec["train_step kernel (k_train_fused)"] = pick(tr.get("kernel"), "kernel", "bound", "achieved", "peak", "unit", "frac", "ms_per_call",
                                                         "flop_per_sample", "algorithmic_hbm_bytes_per_call", "traffic", "mfma_pipe_busy_frac_pmc")
    elif tr is not None:
        sec["train_step"] = pick(tr)
    se = line.get("se3")
    if isinstance(se, dict) and "error" not in se:
        for k, v in se.items():
            sec["se3." + k + " (BASELINE config 5 per GPU)"] = pick(v, "kernel", "bound", "achieved", "peak", "unit", "frac", "ms", "traffic",
                                                                      "bytes_per_residue", "bytes_per_frame", "n", "structures", "residues_per_structure")
    elif se is not None:
        sec["se3"] = pick(se)
    wn = line.get("wide_net")
    if isinstance(wn, dict) and "error" not in wn:
        sec["wide_net.chain (255-wide residual network, reverse chain)"] = pick(wn.get("chain"), "kernel", "bound", "achieved", "peak", "unit", "frac",
                                                                                "ms_per_launch", "batch", "steps_per_launch", "flop_per_sample_step")
        ts = wn.get("train_step", {})
        sec["wide_net.train_step"] = pick(ts, "batch", "ms_per_step", "algorithmic_TFLOPs", "frac_of_bf16_mfma_peak", bound="hbm (the design's register dumps) / mfma",
                                          peak=BF16_MFMA_PEAK_TFLOPS, unit="TFLOP/s")
    elif wn is not None:
        sec["wide_net"] = pick(wn)
    pn = line.get("planenet")
    if isinstance(pn, dict) and "error" not in pn:
        for k, v in pn.items():
            if isinstance(v, dict) and "frac" in v:
                sec["planenet." + k] = pick(v, "kernel", "bound", "achieved", "peak", "unit", "frac", "ms", "flop", "clouds", "points", "ms_with_dropout_0.1")
    elif pn is not None:
        sec["planenet"] = pick(pn)
    pt = line.get("protnet")
    if isinstance(pt, dict) and "error" not in pt:
        for k, v in pt.items():
            sec["protnet." + k + " (BASELINE config 5's denoiser)"] = pick(v, "kernel", "bound", "achieved", "peak", "unit", "frac", "ms", "flop", "complexes", "residues")
    elif pt is not None:
        sec["protnet"] = pick(pt)
    el = line.get("external_loop")
    if el is not None:
        sec["external_loop (one p_sample call per reverse step, so3_test.py:28-31)"] = pick(el, "sample_steps_per_s", "ms_per_call", "calls", "batch",
                                                                                          "vs_chain_kernel_rate", "at_batch_4M")
    return sec


def flat_roofline_keys(line):
    """The other legs' headline numbers as FLAT SCALAR members of `roofline` (the driver's record keeps only scalar members of that
    object; the nested `secondary` repeats them with everything needed to recompute each one).  A leg that did not run gives None."""
    def get(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d if isinstance(d, (int, float)) and not isinstance(d, bool) else None
    ig, tr, se, wn, pn, el, pt = (line.get(k) for k in ("igso3_eval", "train_step", "se3", "wide_net", "planenet", "external_loop", "protnet"))
    flat = {
        "cfg2_hbm_frac": get(ig, "frac"), "cfg2_ms": get(ig, "ms"), "cfg2_hbm_frac_at_2p24": get(ig, "at_n_2p24", "frac"),
        "cfg2_traffic_bytes": get(ig, "traffic"),
        "cfg4_step_ms": get(tr, "ms_per_step"),
        "cfg4_step_frac": None if get(tr, "algorithmic_TFLOPs_per_gpu") is None else get(tr, "algorithmic_TFLOPs_per_gpu") / BF16_MFMA_PEAK_TFLOPS,
        "cfg4_kernel_frac": get(tr, "kernel", "frac"), "cfg4_kernel_ms": get(tr, "kernel", "ms_per_call"),
        "cfg4_kernel_traffic_bytes": get(tr, "kernel", "traffic"), "cfg4_eager_loop_ms": get(tr, "eager_python_loop_ms_per_step"),
        "cfg5_rigid_frac": get(se, "rigid_move", "frac"), "cfg5_noise_frac": get(se, "se3_q_sample_target", "frac"),
        "cfg5_protnet_fwd_frac": get(pt, "forward", "frac"), "cfg5_protnet_fwd_ms": get(pt, "forward", "ms"),
        "cfg5_protnet_train_frac": get(pt, "train_eval", "frac"), "cfg5_protnet_train_ms": get(pt, "train_eval", "ms"),
        "cfg5_protnet_train_batch4_ms": get(pt, "train_eval_reference_batch", "ms"),
        "wide_chain_frac": get(wn, "chain", "frac"), "wide_train_step_ms": get(wn, "train_step", "ms_per_step"),
        "wide_train_step_frac": get(wn, "train_step", "frac_of_bf16_mfma_peak"),
        "planenet_fwd_256_frac": get(pn, "forward_32x256", "frac"), "planenet_fwd_256_ms": get(pn, "forward_32x256", "ms"),
        "planenet_train_256_frac": get(pn, "train_eval_32x256", "frac"), "planenet_train_256_ms": get(pn, "train_eval_32x256", "ms"),
        "planenet_fwd_2048_frac": get(pn, "forward_32x2048", "frac"), "planenet_fwd_2048_ms": get(pn, "forward_32x2048", "ms"),
        "planenet_train_2048_frac": get(pn, "train_eval_32x2048", "frac"), "planenet_train_2048_ms": get(pn, "train_eval_32x2048", "ms"),
        "planenet_train_2048_dropout_ms": get(pn, "train_eval_32x2048", "ms_with_dropout_0.1"),
        "external_loop_ratio": get(el, "vs_chain_kernel_rate"), "external_loop_ms_per_call": get(el, "ms_per_call"),
        "full_chain_sample_steps_per_s": get(line.get("full_chain"), "sample_steps_per_s"),
    }
    return flat


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch-log2", type=int, default=20, help="rotations per GPU = 2^this (BASELINE: 20)")
    ap.add_argument("--timesteps", type=int, default=1000)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--steps-per-launch", type=int, default=100, help="reverse steps fused into one chain-kernel launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--train-timeout", type=float, default=240.0, help="N > 1: seconds the training leg may take before the line is printed without it")
    args = ap.parse_args()

    from so3x import backend as B
    from so3x import parallel
    from so3x.so3_train import RotPredict
    from so3x.diffusion import SO3Diffusion

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and args.gpus > 1:   # (a plain `--gpus N` never gets here: _self_launch above started the ranks)
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}, or with no launcher at all")
    ctx = parallel.init()
    dev = ctx.device
    if dev.type != "cuda":
        raise SystemExit("bench.py needs an MI355X (no CPU path)")

    T = args.timesteps
    n = 1 << args.batch_log2
    prec = B.PREC_BF16 if args.precision == "bf16" else B.PREC_F32
    torch.manual_seed(0)
    net = RotPredict(out_type="skewvec", precision=args.precision)          # seed-0 default init (CPU generator)
    q = torch.randn(n, 4, generator=torch.Generator().manual_seed(ctx.rank))  # synthetic random quaternions
    # the CPU baseline FIRST (rank 0 at N = 1 only; ~6 s of host work): every second after it is GPU work, so a driver that samples
    # GPU utilisation during the run sees the GPU legs, not the host leg
    cpu_leg = None
    if not args.no_cpu_baseline and ctx.world_size == 1 and ctx.rank == 0:
        try:
            cpu_leg = cpu_baseline(T, net.flat_params_nograd().detach().cpu().numpy(), B.cosine_beta_schedule(T))
        except Exception as e:  # report, never hide -- and never lose the GPU line to a host-side build problem
            cpu_leg = {"error": repr(e)}
    t_gpu_legs = time.perf_counter()
    net = net.to(dev)
    proc = SO3Diffusion(net, timesteps=T).to(dev)
    _, trap_p = proc._tables()
    params = net.flat_params_nograd()
    x = B.quat_to_rmat(q.to(dev))
    index_base = ctx.rank * n

    def barrier():
        torch.cuda.synchronize()
        if ctx.world_size > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    RAMP = 800  # untimed clock ramp before anything is timed, whatever --warmup says: the chip takes ~40 ms under load to settle
                # (profiles/r02_chain_dispatches.json: 7.8, 7.2, 6.9, 6.7 ... 6.5 ms for consecutive identical launches)
    # the chain kernel's prepared state (weight image, per-timestep rows, CDF records): a function of the weights and the tables,
    # built once, outside the timed region -- as SO3Diffusion caches it across p_sample calls
    prep = B.p_sample_prepare(params, proc._sched, trap_p, prec, guide_p=proc._guide_p)
    run_steps(B, params, proc._sched, trap_p, x, T, RAMP, 0, index_base, prec, per_launch=100, guide_p=proc._guide_p, prepared=prep)
    run_steps(B, params, proc._sched, trap_p, x, T, args.warmup, 0, index_base, prec, rng_offset=RAMP, per_launch=args.steps_per_launch,
              guide_p=proc._guide_p, prepared=prep)
    barrier()
    t0 = time.perf_counter()
    launches = run_steps(B, params, proc._sched, trap_p, x, T, args.steps, 0, index_base, prec, rng_offset=RAMP + args.warmup,
                         per_launch=args.steps_per_launch, guide_p=proc._guide_p, prepared=prep)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    barrier()
    tmax = torch.tensor([el, -el], device=dev, dtype=torch.float64)
    if ctx.world_size > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    el, el_fastest = float(tmax[0].item()), -float(tmax[1].item())
    ok = bool(torch.isfinite(x).all().item())

    # ---- roofline of the dominant kernel on a FIXED shape (100 steps per launch, 5 launches, HIP events on the launch
    #      stream): the shape the rocprofv3 summaries under profiles/ were taken on, whatever --steps was
    RL_STEPS, RL_LAUNCHES = 100, 5
    raw = RawChain(B, torch, T, prec)
    for i in range(4):  # back to steady clocks after the host-side pause above
        raw(params, proc._sched, trap_p, proc._guide_p, x, T - 1, RL_STEPS, 1, 9_000 + 100 * i, index_base)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()                        # same (current) stream the C ABI launches on
    for i in range(RL_LAUNCHES):
        raw(params, proc._sched, trap_p, proc._guide_p, x, T - 1 - 100 * i, RL_STEPS, 1, 10_000 + 100 * i, index_base)
    ev1.record()
    torch.cuda.synchronize()
    ms_per_launch = ev0.elapsed_time(ev1) / RL_LAUNCHES
    clock_ghz = raw.clock_ghz()         # of the last timed launch (wave 0 of workgroup 0: shader ticks / 100 MHz reference ticks)
    # vector-issue-port accounting: per 64-sample wave-step the kernel issues N_valu vector instructions
    # (PMC SQ_INSTS_VALU of the committed profile, MFMAs included) of which N_trans are 8-cycle ones and 106 are MFMAs that hold
    # the port for 8 of their 32 cycles; everything else costs 4 (MI355X_MICROARCH.md, row 'vector-instruction ISSUE cost').
    # A SIMD serves two waves: port cycles of one wave-step / SIMD cycles per wave-step (live time x live clock).
    port = None
    iv = pmc_counter("k_p_sample_chain", "SQ_INSTS_VALU")
    # sanity gate on the quoted profile: a 100-step launch of this kernel issues ~1,100 vector instructions per 64-sample wave-step.
    # A value outside [500, 3000] means the profile's counters belong to another launch shape (round 5: one-step launches summarised
    # as 100-step ones gave 14) -- then NOTHING PMC-derived is quoted for this kernel.
    n_valu_profile = None if not iv else iv / ((1 << 20) // 64 * 100)              # the profile's shape: 2^20 samples, 100 steps per launch
    pmc_chain_ok = n_valu_profile is not None and 500.0 <= n_valu_profile <= 3000.0
    pmc_chain_note = ("ok: %.0f vector instructions per wave-step in the profile" % n_valu_profile) if pmc_chain_ok else (
        "withheld: no fresh profile" if n_valu_profile is None else
        "withheld: the profile implies %.1f vector instructions per wave-step, outside [500, 3000] -- its counters are not a 100-step launch's" % n_valu_profile)
    if prec == B.PREC_BF16 and clock_ghz and pmc_chain_ok and isa_mix("chain"):
        wave_steps = ((n + 63) // 64) * RL_STEPS
        n_valu = n_valu_profile
        mix = isa_mix("chain") or {}
        n_mfma, n_trans = mix.get("mfma_per_wave_step"), mix.get("trans_per_wave_step")   # tools/count_isa.py on the step loop
        port_cycles = (n_valu - n_mfma - n_trans) * 4 + n_trans * 8 + n_mfma * 8
        simd_cycles = ms_per_launch * 1e-3 * clock_ghz * 1e9 / (wave_steps / 1024.0)
        port = {"frac": port_cycles / simd_cycles, "port_cycles_per_wave_step": port_cycles, "simd_cycles_per_wave_step": simd_cycles,
                "vector_instructions_per_wave_step_pmc": n_valu, "mfma": n_mfma, "transcendental": n_trans,
                "source": "SQ_INSTS_VALU: " + PMC.source + "; mfma / transcendental counts: profiles/" + f"{PMC.meta.get('tag', 'r04')}_chain_isa_mix.json"
                          " (tools/count_isa.py); time and clock: this run"}
    flop_per_launch = MLP_FLOP_PER_SAMPLE * n * RL_STEPS
    tflops = flop_per_launch / (ms_per_launch * 1e-3) / 1e12
    # ISSUED matrix work: the PMC count of bf16 MFMA "MOPS" per launch of the profiled shape (x 512 flop each: the padded 96-row
    # tiles and the fifth k-step included) over THIS run's launch time -- the other way to read "40 % MFMA utilisation"
    mops = pmc_counter("k_p_sample_chain", "SQ_INSTS_VALU_MFMA_MOPS_BF16")
    issued_tflops = None if (not mops or not pmc_chain_ok or n != (1 << 20) or prec != B.PREC_BF16) else mops * 512.0 / (ms_per_launch * 1e-3) / 1e12

    # ---- the f16-operand LEG of the same kernel (SO3X_PREC_F16): a labelled extra beside the bf16 headline -- what
    #      "config 3 names bf16" costs or buys -- same shape, three launches, HIP events.  Never the headline.
    f16_leg = None
    if prec == B.PREC_BF16 and not args.no_extras:
        try:
            xf16 = x.clone()
            B.p_sample_chain(params, proc._sched, trap_p, xf16, T - 1, RL_STEPS, seed=1, rng_offset=20_000, index_base=index_base,
                             precision=B.PREC_F16, out=xf16, guide_p=proc._guide_p)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(3):
                B.p_sample_chain(params, proc._sched, trap_p, xf16, T - 1 - 100 * i, RL_STEPS, seed=1, rng_offset=20_100 + 100 * i,
                                 index_base=index_base, precision=B.PREC_F16, out=xf16, guide_p=proc._guide_p)
            e1.record()
            torch.cuda.synchronize()
            ms16 = e0.elapsed_time(e1) / 3
            f16_leg = {"what": "the same kernel with IEEE half operand bits (SO3X_PREC_F16): labelled extra, NOT the headline (BASELINE config 3 names bf16)",
                       "ms_per_launch": ms16, "sample_steps_per_s": n * RL_STEPS / (ms16 * 1e-3),
                       "frac": MLP_FLOP_PER_SAMPLE * n * RL_STEPS / (ms16 * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                       "vs_bf16_time": ms16 / ms_per_launch, "finite": bool(torch.isfinite(xf16).all().item()),
                       "note": "2.5x closer to the fp32 chain per step (median 6e-7 vs 1.5e-6) and ~8 % SLOWER: v_fma_mixlo / mixhi_f16 fuse an "
                               "activation's multiply-add, conversion and pack but issue at the packed-op rate, and the kernel needs 256 "
                               "registers (profiles/r04_ab_chain_f16_operands.json)"}
            del xf16
        except Exception as e:  # report, never hide
            f16_leg = {"error": repr(e)}

    # ---- the metric verbatim: one complete p_sample_loop, B rotations through all T reverse steps (diffusion.py:328-337)
    proc.p_sample_loop((256,))  # untimed: first-use initialisation of the start distribution (its CDF table; ~25 ms of host work)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    xf = proc.p_sample_loop((n,))
    torch.cuda.synchronize()
    full_s = time.perf_counter() - t1
    tfull = torch.tensor([full_s], device=dev, dtype=torch.float64)
    if ctx.world_size > 1:
        torch.distributed.all_reduce(tfull, op=torch.distributed.ReduceOp.MAX)
    full_s = float(tfull.item())
    full_ok = bool(torch.isfinite(xf).all().item())
    del xf

    devices = [f"{dev} ({torch.cuda.get_device_name(dev)})"]
    if ctx.world_size > 1:  # what each rank actually ran on
        devices = [None] * ctx.world_size
        torch.distributed.all_gather_object(devices, f"rank {ctx.rank}: {dev} ({torch.cuda.get_device_name(dev)})")

    line = None
    if ctx.rank == 0:
        total = ctx.world_size * n * args.steps
        line = {
            "metric": "SO(3) sample-steps/sec, reverse p_sample chain with score MLP", "value": total / el,
            "unit": "sample-steps/s", "n_gpus": ctx.world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE config 3: full reverse p_sample chain with RotPredict score MLP",
                       "batch_per_gpu": n, "global_batch": ctx.world_size * n, "timesteps": T,
                       "mlp_operands": args.precision, "rotation_state": "fp32", "noise": "in-kernel Philox4x32-10",
                       "parallelism": f"batch-sharded x{ctx.world_size}, no collective",
                       "launches_timed": launches, "clock_ramp_steps_untimed": RAMP,
                       "prepared_state": "so3x_p_sample_prepare once, untimed (weights and tables only); timed launches are so3x_p_sample_prepared"},
            "finite": ok,
            "ranks_seen": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
            "backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
            "launcher": "self (bench.py started the ranks)" if os.environ.get("SO3X_BENCH_CHILD") else
                        ("torch.distributed.run / external" if ctx.world_size > 1 else "single process"),
            "rank_ms_per_step": {"slowest": el / args.steps * 1e3, "fastest": el_fastest / args.steps * 1e3},
            "devices": devices,
            "full_chain": {"what": "one complete p_sample_loop: IGSO3(1) start, T reverse steps, ONE kernel launch", "batch_per_gpu": n,
                           "timesteps": T, "seconds": full_s, "sample_steps_per_s": ctx.world_size * n * T / full_s, "finite": full_ok},
            "roofline": {"kernel": "k_p_sample_chain", "bound": "mfma", "achieved": tflops, "peak": BF16_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": tflops / BF16_MFMA_PEAK_TFLOPS,
                         "issued_frac": None if issued_tflops is None else issued_tflops / BF16_MFMA_PEAK_TFLOPS,
                         "issued_TFLOPs": issued_tflops,
                         "pmc_source": PMC.source, "silu": "256-entry secant table (max abs error 1.3e-4), bf16 chain kernel",
                         "f16_operand_leg": f16_leg,
                         "pmc_sanity": pmc_chain_note,
                         "traffic": pmc_traffic("k_p_sample_chain", batch=n, steps_per_launch=RL_STEPS, precision=args.precision) if pmc_chain_ok else None,
                         "mfma_pipe_busy_frac_pmc": pmc_mfma_busy("k_p_sample_chain") if pmc_chain_ok else None,
                         "valu_busy_frac_pmc": pmc_valu_busy("k_p_sample_chain") if pmc_chain_ok else None,
                         "in_kernel_clock_ghz": clock_ghz,
                         "valu_port_bound_frac": None if port is None else port["frac"], "valu_port_accounting": port,
                         "launches": RL_LAUNCHES, "steps_per_launch": RL_STEPS, "ms_per_launch": ms_per_launch,
                         "sample_steps_per_s": n * RL_STEPS / (ms_per_launch * 1e-3),
                         "flop_per_sample_step": MLP_FLOP_PER_SAMPLE,
                         "measured_here": ["achieved", "frac", "ms_per_launch", "sample_steps_per_s", "in_kernel_clock_ghz"],
                         "quoted_from_the_committed_profile": ["traffic", "mfma_pipe_busy_frac_pmc", "valu_busy_frac_pmc", "issued_frac (count)",
                                                               "valu_port_bound_frac (instruction counts)"],
                         "note": "algorithmic MLP flops vs the dense bf16 MFMA peak, on a fixed 100-step launch timed with HIP events "
                                 "(the shape profiled under profiles/); the kernel's real bound is the VALU issue port "
                                 "(valu_busy_frac_pmc; DESIGN.md section 4); algorithmic HBM traffic is 72 B/sample per launch"},
        }

    # ---- BASELINE config 4: the training step, on every rank (gradient all-reduce when N > 1).  The headline above is complete at
    #      this point; with more than one rank the leg runs under a watchdog, so that a collective that never returns (this leg is
    #      the only part of the run with a data-path collective, and no multi-GPU node was available to any round) costs the
    #      training numbers, not the line: rank 0 prints what it has with the reason, every rank exits.
    train = None
    if not args.no_extras:
        watchdog = None
        if ctx.world_size > 1:
            import threading

            def give_up():
                if ctx.rank == 0 and line is not None:
                    line["train_step"] = {"error": f"no result after {args.train_timeout} s (a collective that did not return?); the headline above is unaffected"}
                    line["exit_code"] = 3
                    print(json.dumps(line), flush=True)
                # NON-ZERO, on every rank: a collective that never returned is a failed run (kernels or collectives may still be in
                # flight); the line above carries the headline and the reason for whoever reads stdout
                os._exit(3)

            watchdog = threading.Timer(args.train_timeout, give_up)
            watchdog.daemon = True
            watchdog.start()
        try:
            train = train_leg(B, torch, ctx, T)
        except Exception as e:  # report, never hide
            train = {"error": repr(e)}
        if watchdog is not None:
            watchdog.cancel()

    if ctx.rank == 0:
        if train is not None:
            line["train_step"] = train
        if not args.no_extras and ctx.world_size == 1:
            try:  # BASELINE config 2 on its stated inputs: (2b) per-sample eps from the schedule is the headline of this leg
                leg = igso3_eval_roofline(B, torch, eps_input="schedule", sched=proc._sched)
                big = igso3_eval_roofline(B, torch, n=1 << 24, reps=10, eps_input="schedule", sched=proc._sched)
                leg["at_n_2p24"] = {k: big[k] for k in ("achieved", "frac", "ms", "evals_per_s")}
                leg["scalar_eps_0p5"] = igso3_eval_roofline(B, torch, eps_input="scalar")
                leg["uniform_eps_not_a_baseline_input"] = {k: v for k, v in igso3_eval_roofline(B, torch, eps_input="uniform").items()
                                                           if k in ("eps_input", "achieved", "frac", "ms", "finite_logp_frac")}
                line["igso3_eval"] = leg
            except Exception as e:  # report, never hide
                line["igso3_eval"] = {"error": repr(e)}
            try:  # BASELINE config 5's per-GPU device work
                line["se3"] = se3_legs(B, torch)
            except Exception as e:
                line["se3"] = {"error": repr(e)}
            try:
                line["wide_net"] = wide_net_extra(B, torch, proc._sched, trap_p)
            except Exception as e:
                line["wide_net"] = {"error": repr(e)}
            try:
                line["planenet"] = planenet_leg(torch)
            except Exception as e:
                line["planenet"] = {"error": repr(e)}
            try:
                line["protnet"] = protnet_leg(torch)
            except Exception as e:
                line["protnet"] = {"error": repr(e)}
            try:
                el_leg = external_loop_leg(torch, proc, x, T)
                el_leg["vs_chain_kernel_rate"] = el_leg["sample_steps_per_s"] / line["roofline"]["sample_steps_per_s"]
                # the same loop on four times the batch: a launch's fixed costs (20-30 us: LDS image and table fill, the state's
                # HBM round trip, first-chunk warm-up, tail; tools/ab/README.md round 5) against four times the work
                x4 = torch.cat([x] * 4)
                big = external_loop_leg(torch, proc, x4, T, calls=250)
                el_leg["at_batch_4M"] = {k: big[k] for k in ("batch", "calls", "ms_per_call", "sample_steps_per_s", "finite")}
                el_leg["at_batch_4M"]["vs_chain_kernel_rate"] = big["sample_steps_per_s"] / line["roofline"]["sample_steps_per_s"]
                del x4
                line["external_loop"] = el_leg
            except Exception as e:
                line["external_loop"] = {"error": repr(e)}
        if cpu_leg is not None:
            line["cpu_baseline"] = cpu_leg
        line["roofline"].update(flat_roofline_keys(line))     # flat scalars first: what the driver's record keeps
        line["roofline"]["secondary"] = secondary_rooflines(line)
        line["gpu_legs_wall_seconds"] = time.perf_counter() - t_gpu_legs
        print(json.dumps(line), flush=True)
    parallel.finalize(ctx)


if __name__ == "__main__":
    main()
