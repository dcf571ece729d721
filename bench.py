#!/usr/bin/env python3
"""bench.py -- throughput of the body->hand hot path on N MI355X of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8 --steps 20 --warmup 3         # spawns the 8 ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the fused four-layer kernel (ConvModel.forward,
body2hand/src/models/HandPoseModels.py:40-64 of the reference) over this rank's
resident shard of a synthetic keypoint stream.  Workload: BASELINE.json config 3
(bf16 MFMA path, T = 200) fed as a sustained stream -- 262 144 sequences x 200
frames per GPU per step (1024 batches of config 3's batch=256; 13.8 GB of HBM
traffic per step, far beyond the 256 MiB Infinity Cache; a step lasts ~3 ms so
that the GPU's clock/power transient at the start of sustained load does not
dominate short runs), sequence-sharded across ranks with no data-path collective
(weak scaling).  The metric "hand-crops/sec" of BASELINE.json is reported as
frames/s: the reference has no image crops, one "crop" = one frame of 12x2 body
keypoints in -> 21x2 hand keypoints out.

N > 1: one process per GPU over RCCL (torch.distributed backend "nccl").  Run from a
plain shell with --gpus N the script starts the N ranks itself (a child
`python -m torch.distributed.run`, before this process has touched the GPU) and relays
rank 0's line.  RCCL is mandatory: if it cannot initialise, or two ranks sit on the same
GPU, the run FAILS (non-zero exit, no line); `--backend gloo` exists only to rehearse the
N > 1 plumbing on a one-GPU box and says so in the line.  For N > 1 the hand-back of the
keypoints to rank 0 (north_star: "RCCL over xGMI used only to gather per-frame keypoints
back to rank 0") is timed too and reported in the `gather` object, incl. and excl. the
transfer; `value` stays the exchange-free compute rate.

Rank 0 prints ONE JSON line (see the repo's task contract), with extra objects:
"roofline" (HIP-event launch time of the headline kernel vs HBM peak), "kernels" (the same
graded figures for every precision of the path, each with its error on the CPU sample),
"cpu_baseline" (the reference's CPU execution -- four torch Conv1d calls -- timed on this
box's host cores; reported only) and, for N > 1, "gather".
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BYTES_PER_FRAME = 264      # 24 fp32 in + 42 fp32 out; weights 76 KB amortised
FLOP_PER_FRAME = 37800     # 18 900 MAC
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# dense matrix peak per precision (MI355X_MICROARCH.md); f16x3 issues three f16 MFMAs per product
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f16x3": 2500.0 / 3, "f32_mfma": 157.3, "f32_valu": 157.3, "fp32": 157.3}
DTYPE = {"bf16": "bf16", "f16": "f16", "f16x3": "f16x3 (f16 hi+lo operands, fp32-grade)", "f32_mfma": "f32", "f32_valu": "f32", "fp32": "f32"}
TRAFFIC_JSON = os.path.join("profiles", "r3_final", "traffic.json")
TRAFFIC_SOURCES = [os.path.join("hand_pose_sl_amd", "csrc", f) for f in ("kernel_mfma16.h", "kernel_mfma.h", "b2h_common.h", "b2h_api.hip")]


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--seqs", type=int, default=262144, help="sequences per GPU per step")
    ap.add_argument("--frames", type=int, default=200, help="T, frames per sequence (--max-frames default, run.py:28)")
    ap.add_argument("--precision", default="bf16", choices=sorted(MFMA_PEAK_TFLOPS))
    ap.add_argument("--precondition", type=int, default=0,
                    help="extra untimed launches BEFORE the W warmup steps (default 0: `value` follows the driver's "
                         "protocol exactly -- W warmup steps, then K timed steps; counted in `untimed_launches`)")
    ap.add_argument("--sustained", type=int, default=200,
                    help="N = 1: after `value` is measured, run this many more untimed launches and time K steps again "
                         "(reported as `value_sustained`, never as `value`; 0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL, mandatory on a real node; gloo ONLY to rehearse "
                         "N>1 on a one-GPU box)")
    ap.add_argument("--gather-seqs", type=int, default=0,
                    help="sequences per GPU in the pipelined hand-back measurement (0 = the bench shard, "
                         "reduced to what fits rank 0's memory; 2048 under gloo)")
    ap.add_argument("--no-gather", action="store_true", help="skip the hand-back measurement (N > 1)")
    ap.add_argument("--gather-budget", type=float, default=180.0,
                    help="seconds the hand-back measurement may take before the line is printed without it")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group even for one rank (smoke test of RCCL on a one-GPU box)")
    return ap.parse_args(argv)


def sources_sha256():
    """Fingerprint of the sources the committed PMC traffic figure was measured on."""
    h = hashlib.sha256()
    for rel in TRAFFIC_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def committed_traffic(S, T, kernel_name, precision, path=None):
    """HBM bytes per launch from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE, tools/profile.sh):
    counters cannot be read inside a bench run, so the committed figure for this exact workload
    is reported -- only while the kernel sources still hash to what it was measured on; null for
    any other configuration or after any edit of those sources."""
    r = {"traffic": None}
    try:
        tj = json.load(open(path or os.path.join(ROOT, TRAFFIC_JSON)))
        if (tj["seqs_per_gpu"], tj["frames_per_seq"]) == (S, T) and tj["kernel"] in kernel_name \
                and tj["precision"] == precision:
            if tj.get("sources_sha256") == sources_sha256():
                r["traffic"] = tj["traffic_bytes"]
                r["traffic_source"] = TRAFFIC_JSON
            else:
                r["traffic_note"] = f"{TRAFFIC_JSON} was measured on other kernel sources (sha256 mismatch): not reported"
    except (OSError, KeyError, ValueError):
        pass
    return r


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) from a plain shell: start the N ranks as a CHILD
    `python -m torch.distributed.run` and relay rank 0's JSON line.  This process never
    touches the GPU (no torch import at all), so nothing that has initialised HIP is ever
    re-executed."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    sys.stderr.write("[bench] self-launch: " + " ".join(cmd) + "\n")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, cwd=ROOT)
    lines = [ln for ln in r.stdout.decode("utf-8", "replace").splitlines() if ln.startswith("{")]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    if r.returncode != 0:
        sys.stderr.write(f"[bench] a rank failed (torch.distributed.run exit code {r.returncode})\n")
        if lines and '"gather": {"error"' in lines[-1]:     # `value` stands, the hand-back measurement failed
            sys.exit(EXIT_SECONDARY_FAILED)
        sys.exit(r.returncode if 0 < r.returncode < 256 else 1)
    if not lines:
        sys.stderr.write("[bench] the ranks exited cleanly but rank 0 printed no line\n")
        sys.exit(1)
    sys.exit(0)


def cpu_baseline(model, seconds):
    """Reference CPU path (torch Conv1d x4 == oracle.torch_port) on the host cores, bounded sample:
    config 2/3-sized batches (256 x 200 frames).  Fixed protocol, so that two runs can be compared:
    the process stays on the affinity set the job was given; each of 1 / 8 / 16 threads (capped at
    that set) gets 3 warm-up passes and is then timed for `seconds` / 3 (>= 3 s each at the default);
    ALL rates are reported together with the 1-minute load average before and after, and `value` is
    the best of them (`cores` = its thread count).  Returns the baseline object, the sample and the
    CPU result (for the per-kernel error)."""
    import torch

    import oracle
    # The GPU box gives one GPU's job a share of 16 host cores (cpu_count reports the whole
    # 256-thread host; running torch on all of them is 1000x slower through oversubscription).
    affinity = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    avail = len(affinity)
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    port = oracle.TorchPort(state, pos_emb=False)
    g = torch.Generator().manual_seed(0)
    x = torch.rand((256, 200, 12, 2), generator=g) - 0.5
    per = max(1.0, seconds / 3.0)
    load0 = os.getloadavg()[0]
    rates, passes, y_cpu = {}, {}, None
    for nt in sorted({1, min(8, avail), min(16, avail)}):
        torch.set_num_threads(nt)
        for _ in range(3):
            y_cpu = port(x)
        n, t0 = 0, time.perf_counter()
        while True:
            y_cpu = port(x)
            n += 1
            el = time.perf_counter() - t0
            if el >= per or n >= 200000:
                break
        rates[nt] = n * 256 * 200 / el
        passes[nt] = n
    cores = max(rates, key=rates.get)
    fps = rates[cores]
    out = {"value": fps, "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": f"(256,200,12,2) U[-0.5,0.5], torch {torch.__version__} Conv1d x4 fp32 (oracle/torch_port.py); per "
                     f"thread count 3 warm-up passes then {per:.1f} s timed ({passes[cores]} passes at {cores} threads); "
                     f"value = best of {sorted(rates)} threads",
           "rates_by_threads": {str(k): v for k, v in sorted(rates.items())},
           "host_cores_available": avail, "loadavg_1m_before": load0, "loadavg_1m_after": os.getloadavg()[0],
           "gflops": fps * FLOP_PER_FRAME / 1e9}
    return out, x, y_cpu.contiguous()


def kernel_roofline(model, x, y, prec, iters):
    """Graded figures of one precision of the path on the bench shard: HIP events on the launch
    stream (b2h_time_forward), algorithmic bytes and FLOPs per launch, both ceilings; `bound` is
    the ceiling that is lower in frames/s for that arithmetic."""
    S, T = x.shape[0], x.shape[1]
    ms = model.time_forward(x, y, iters, precision=prec)
    gbs = S * T * BYTES_PER_FRAME / (ms * 1e-3) / 1e9
    tfl = S * T * FLOP_PER_FRAME / (ms * 1e-3) / 1e12
    peak_t = MFMA_PEAK_TFLOPS[prec]
    hbm_bound = (HBM_PEAK_GBS * 1e9 / BYTES_PER_FRAME) <= (peak_t * 1e12 / FLOP_PER_FRAME)
    r = {"kernel": model.kernel_name(prec), "dtype": DTYPE[prec], "launch_ms": ms, "launches_timed": iters,
         "frames_per_s": S * T / (ms * 1e-3), "bytes_per_launch": S * T * BYTES_PER_FRAME,
         "flop_per_launch": S * T * FLOP_PER_FRAME,
         "hbm_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
         "mfma_tflops": tfl, "mfma_peak_tflops": peak_t, "mfma_frac": tfl / peak_t}
    if hbm_bound:
        r.update(bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS)
    else:
        r.update(bound="mfma", achieved=tfl, peak=peak_t, unit="TFLOP/s", frac=tfl / peak_t)
    return r


EXIT_SECONDARY_FAILED = 3   # `value` was measured and the line printed, but the N > 1 hand-back measurement failed


class SecondaryGuard:
    """Failure handling of the N > 1 hand-back measurement (a SECONDARY figure: `value` and `roofline` are
    already measured when it starts).  Whatever happens in it -- an exception in any rank, a hung peer, a phase
    that outlives its budget -- rank 0 still prints the line, with the cause in `gather.error`, and THEN every
    rank exits with EXIT_SECONDARY_FAILED, so the failure is in the exit code as well as in the line.

    The ranks talk through the rendezvous store (a TCP store, independent of RCCL / gloo collectives, which
    may be the thing that hangs): a failing rank posts `fail` = its message; a watcher thread on every rank
    polls it; rank 0's watcher emits the line and posts `ack`; the other ranks leave only after the ack (or a
    grace period), because under `torch.distributed.run` the first non-zero exit makes the agent kill the
    remaining ranks -- rank 0 must have printed by then.  Only the watcher thread touches the store."""

    def __init__(self, store, rank, world, budget, emit, log=sys.stderr.write, poll=0.25, grace=15.0, prefix="b2h/secondary/"):
        import threading
        self.store, self.rank, self.world, self.budget = store, rank, world, budget
        self.emit, self.log, self.poll, self.grace, self.prefix = emit, log, poll, grace, prefix
        self._local_fail = None         # set by the main thread when phase() raised
        self._local_done = False        # set by the main thread when phase() returned
        self._ok = threading.Event()    # every rank finished the phase
        self._thread = threading.Thread(target=self._watch, daemon=True)

    def _key(self, k):
        return self.prefix + k

    def _die(self, reason):
        """Never returns.  Rank 0: emit the line with the error, ack.  Others: wait for the ack."""
        self.log(f"[bench rank {self.rank}] hand-back measurement failed: {reason}\n")
        if self.rank == 0:
            try:
                self.emit(reason)
            finally:
                try:
                    self.store.set(self._key("ack"), "1")
                except Exception:  # noqa: BLE001 -- the store may be gone with its host rank
                    pass
        else:
            t0 = time.monotonic()
            while time.monotonic() - t0 < self.grace:
                try:
                    if self.store.check([self._key("ack")]):
                        break
                except Exception:  # noqa: BLE001
                    break
                time.sleep(self.poll)
        os._exit(EXIT_SECONDARY_FAILED)

    def _watch(self):
        t0 = time.monotonic()
        posted_done = False
        while True:
            if self._local_fail is not None:
                try:
                    self.store.set(self._key("fail"), self._local_fail)
                except Exception:  # noqa: BLE001
                    pass
                self._die(self._local_fail)
            try:
                if self.store.check([self._key("fail")]):
                    self._die(self.store.get(self._key("fail")).decode("utf-8", "replace"))
                if self._local_done and not posted_done:
                    self.store.add(self._key("done"), 1)
                    posted_done = True
                if posted_done and self.store.add(self._key("done"), 0) >= self.world:
                    self._ok.set()
                    return
            except Exception as exc:  # noqa: BLE001 -- store unreachable: its host (rank 0) is gone
                self._die(f"rendezvous store unreachable ({type(exc).__name__}: {exc})")
            if time.monotonic() - t0 > self.budget:
                reason = (f"hand-back measurement did not finish within {self.budget:.0f} s (rank {self.rank} gave up); "
                          f"`value` / `roofline` were measured before it and stand")
                try:
                    self.store.set(self._key("fail"), reason)
                except Exception:  # noqa: BLE001
                    pass
                self._die(reason)
            time.sleep(self.poll)

    def run(self, phase):
        """phase() on this rank; returns its result once EVERY rank has finished it, else never returns."""
        self._thread.start()
        try:
            res = phase()
        except Exception as exc:  # noqa: BLE001 -- any failure of the secondary measurement
            self._local_fail = f"rank {self.rank}: {type(exc).__name__}: {exc}"
            self._thread.join()      # the watcher ends the process
            os._exit(EXIT_SECONDARY_FAILED)
        self._local_done = True
        self._ok.wait()
        return res


def gpu_identity(torch, dev):
    p = torch.cuda.get_device_properties(dev)
    ident = {"index": dev.index, "name": p.name,
             "pci": f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}"}
    try:
        ident["uuid"] = str(p.uuid)
    except Exception:  # noqa: BLE001 -- older builds have no uuid field
        pass
    return ident


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        self_launch(args, sys.argv[1:])          # does not return
    # Rank 0 prints exactly ONE line on stdout: the JSON.  Libraries underneath (RCCL's version
    # banner, gloo's connection notes, hipcc when the extension is rebuilt) write to fd 1 as they
    # please, so fd 1 is pointed at stderr for the whole run and the JSON goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import datetime

    import torch
    import torch.distributed as dist

    import hand_pose_sl_amd as hps

    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"[bench rank {rank}] --gpus {args.gpus} but WORLD_SIZE={world}: reporting n_gpus={world}\n")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU path in the product)")
    ndev = torch.cuda.device_count()
    backend = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)          # only missing for a lone --force-dist rank
        backend = args.backend
        if backend == "nccl" and ndev < world:
            sys.exit(f"bench.py --gpus {world}: only {ndev} GPU(s) visible; one rank per GPU over RCCL is required "
                     f"(use --backend gloo only to rehearse the plumbing on a one-GPU box)")
    dev_index = local_rank % ndev          # one rank per GPU on a real node; wraps only in gloo rehearsals
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    gpus = [gpu_identity(torch, dev)]
    if use_dist:
        # RCCL carries the barrier, the max of the elapsed time, the GPU census and the keypoint
        # hand-back.  No fallback: an RCCL failure ends the run with a non-zero exit code.
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=10))
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                sys.exit(f"[bench rank {rank}] RCCL all_reduce probe returned {probe.item()} for world {world}")
        else:
            sys.stderr.write(f"[bench rank {rank}] --backend gloo: REHEARSAL of the {world}-rank plumbing, "
                             f"not an xGMI measurement\n")
            dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=10))
        census = [None] * world
        dist.all_gather_object(census, gpus[0])
        gpus = census
        distinct = len({g.get("uuid") or g["pci"] for g in gpus})
        if backend == "nccl" and distinct != world:
            sys.exit(f"[bench rank {rank}] {world} ranks on {distinct} distinct GPU(s): {gpus}")

    S, T = args.seqs, args.frames
    torch.manual_seed(0)
    model = hps.ConvModel(30, "ReLU", False, precision=args.precision).to(dev).eval()
    # synthetic stream shard U[-0.5, 0.5], generated on the device, resident in HBM before the
    # timed region
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.rand((S, T, 12, 2), dtype=torch.float32, device=dev, generator=g).sub_(0.5)
    y = torch.empty((S, T, 21, 2), dtype=torch.float32, device=dev)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if not use_dist:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    import ctypes

    from hand_pose_sl_amd import _lib
    lib = model._ensure_handle()
    kern = _lib.KERNELS[args.precision]
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    xp, yp = ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr())

    def step():
        _lib.check(lib.b2h_forward(model._handle, xp, yp, S, T, kern, st))

    # `value`: W untimed warmup steps, then exactly K steps between barriers -- the driver's protocol, with
    # no other launch before it unless --precondition asks for some (counted in `untimed_launches`).
    for _ in range(max(0, args.precondition)):
        step()
    for _ in range(args.warmup):
        step()

    def timed_steps():
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    el = timed_steps()

    frames_total = S * T * world * args.steps
    value = frames_total / el
    out = {
        "metric": "hand-crops/sec (1 crop = 1 frame: 12x2 body kpts -> 21x2 hand kpts; the reference has no 256x256 images)",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE[args.precision], "data": "synthetic",
        "untimed_launches": max(0, args.precondition) + args.warmup,
        "config": {"workload": f"BASELINE config 3 stream: ConvModel(30,'ReLU',pos_emb=False) {args.precision} "
                               f"path, {S} seq x {T} frames per GPU per step, inputs resident in HBM, "
                               f"sequence-sharded, no data-path collective in `value`",
                   "seqs_per_gpu": S, "frames_per_seq": T, "kernel": model.kernel_name(),
                   "preconditioning_steps": max(0, args.precondition),
                   "parallelism": f"seq-shard x{world}", "gpus": gpus},
    }
    if world == 1 and args.sustained > 0:
        # the same K steps once more after `--sustained` further untimed launches: the rate of the stream once the
        # clock / power controller has settled (tools/sustain_probe.py: a launch takes 5-15 % longer during the
        # first ~30 ms after an idle gap).  Reported beside `value`, never as it.
        for _ in range(args.sustained):
            step()
        el2 = timed_steps()
        out["value_sustained"] = {"value": S * T * args.steps / el2, "unit": "frames/s", "ms_per_step": el2 / args.steps * 1e3,
                                  "untimed_launches_before": out["untimed_launches"] + args.steps + args.sustained}
    if backend is not None:
        out["config"]["backend"] = "rccl" if backend == "nccl" else "gloo (REHEARSAL on shared GPUs, not xGMI)"

    if rank == 0:
        # roofline of the dominant (only) kernel: HIP events on the launch stream
        iters = max(5, min(args.steps, 50))
        rf = kernel_roofline(model, x, y, args.precision, iters)
        rf.update(committed_traffic(S, T, model.kernel_name(), args.precision))
        out["roofline"] = rf

    # ---- N > 1: the hand-back of the keypoints to rank 0, timed (never part of `value`) ----------
    def gather_phase():
        """The timed hand-back of the keypoints to rank 0 (never part of `value`); returns the `gather` object."""
        from hand_pose_sl_amd.stream import ShardedStream, shard_bounds
        stream = ShardedStream(model)
        gat = {"backend": "rccl" if backend == "nccl" else "gloo (rehearsal)",
               "transport": "direct peer->root send/recv (grouped per piece), no ring"}

        def timed(fn, reps, warm):
            for _ in range(warm):
                fn()
            barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            barrier()
            return max_over_ranks(time.perf_counter() - t0) / reps

        # (1) the bench shard, handed back piece by piece while the next piece computes
        Sg = args.gather_seqs or (S if backend == "nccl" else min(S, 2048))
        Sg = min(Sg, S)
        if rank == 0:   # rank 0 holds the whole gathered stream (and, briefly, a piece in flight per peer)
            free, _tot = torch.cuda.mem_get_info(dev)
            while Sg > 1024 and world * Sg * T * 168 * 1.25 > free:
                Sg //= 2
        sg_t = torch.tensor([Sg], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dist.broadcast(sg_t, 0)
        Sg = int(sg_t.item())
        pieces = 8
        chunk = max(1, (Sg + pieces - 1) // pieces)
        xs = x[:Sg]
        n_tot = Sg * world
        res = {}

        def handed_back():          # the previous result is released before the next one is allocated
            res.clear()
            res["y"] = stream.run_pipelined(xs, n_tot, chunk)

        t_incl = timed(handed_back, 3, 1)
        if rank == 0 and tuple(res["y"].shape) != (n_tot, T, 21, 2):
            sys.exit(f"[bench] gathered stream has shape {tuple(res['y'].shape)}, expected {(n_tot, T, 21, 2)}")
        res.clear()
        torch.cuda.empty_cache()
        t_excl = timed(lambda: [model.forward_into(xs[a:a + chunk], y[a:a + chunk]) for a in range(0, Sg, chunk)], 3, 1)
        into0 = (n_tot - Sg) * T * 168
        gat["pipelined"] = {"workload": f"bench shard: {Sg} seq x {T} frames per GPU in {pieces} pieces, piece k handed "
                                        f"to rank 0 while piece k+1 computes",
                            "seqs_per_gpu": Sg, "ms_incl": t_incl * 1e3, "ms_excl": t_excl * 1e3,
                            "frames_per_s_incl": n_tot * T / t_incl, "frames_per_s_excl": n_tot * T / t_excl,
                            "bytes_into_rank0": into0, "rank0_ingest_GBps": into0 / t_incl / 1e9}
        # (2) BASELINE config 4: 2 000 sequences over the ranks, one gather at the end
        n4 = 2000
        lo, hi = shard_bounds(n4, rank, world)
        x4 = x[: hi - lo]
        t4_incl = timed(lambda: stream.run(x4, n4, gather=True), 20, 3)
        t4_excl = timed(lambda: stream.run(x4, n4, gather=False), 20, 3)
        into0 = (n4 - (shard_bounds(n4, 0, world)[1])) * T * 168
        gat["config4"] = {"workload": f"BASELINE config 4: {n4} seq x {T} frames sharded over {world} GPUs, keypoints "
                                      f"gathered to rank 0",
                          "ms_incl": t4_incl * 1e3, "ms_excl": t4_excl * 1e3,
                          "frames_per_s_incl": n4 * T / t4_incl, "frames_per_s_excl": n4 * T / t4_excl,
                          "bytes_into_rank0": into0}
        return gat

    if world > 1 and not args.no_gather:
        # `value` and the roofline above are already measured.  The hand-back is a SECONDARY figure: an
        # RCCL error in it, a hung peer or a blown budget must not cost the run its line -- but must show in
        # the exit code: rank 0 prints the line with `gather.error`, then every rank exits with
        # EXIT_SECONDARY_FAILED (3), long before RCCL's own watchdog would abort the process.
        def emit_failed(reason):
            out["gather"] = {"error": reason}
            os.write(json_fd, (json.dumps(out) + "\n").encode())

        store = dist.distributed_c10d._get_default_store()
        out["gather"] = SecondaryGuard(store, rank, world, args.gather_budget, emit_failed).run(gather_phase)

    if rank == 0 and world == 1:
        # the same shard through every other precision of the path, graded the same way
        kernels = {args.precision: dict(out["roofline"])}
        for prec, it in (("bf16", 10), ("f16", 10), ("f16x3", 5), ("f32_mfma", 3)):
            if prec in kernels:
                continue
            try:
                kernels[prec] = kernel_roofline(model, x, y, prec, it)
            except RuntimeError as exc:
                kernels[prec] = {"error": str(exc)}
        out["kernels"] = kernels
        # north_star's gate is <= 1e-3 max-abs on ANY input; bf16 meets it on normalised keypoints only (1.05e-3 at
        # N(0,1), tests/test_gpu_parity.py), the f16 instantiation of the same kernel meets it everywhere (1.2e-4
        # worst case): its figure from this same run, graded like `value`'s kernel
        if "frames_per_s" in kernels.get("f16", {}):
            out["value_le_1e-3_any_input"] = {"value": kernels["f16"]["frames_per_s"], "unit": "frames/s", "precision": "f16",
                                              "kernel": kernels["f16"]["kernel"], "frac": kernels["f16"]["frac"],
                                              "timing": f"HIP events over {kernels['f16']['launches_timed']} launches, same shard"}
        if not args.no_cpu_baseline:
            cb, xs_cpu, y_cpu = cpu_baseline(model, args.cpu_seconds)
            # the same sample through the HIP path, every precision, checked against the CPU result
            xd = xs_cpu.to(dev)
            yd = torch.empty((xd.shape[0], xd.shape[1], 21, 2), dtype=torch.float32, device=dev)
            for prec, rec in kernels.items():
                if "error" in rec:
                    continue
                model.forward_into(xd, yd, precision=prec)
                rec["gpu_max_abs_err_on_sample"] = float((yd.cpu() - y_cpu).abs().max())
            cb["gpu_max_abs_err_on_sample"] = kernels[args.precision]["gpu_max_abs_err_on_sample"]
            out["roofline"]["gpu_max_abs_err_on_sample"] = cb["gpu_max_abs_err_on_sample"]
            out["cpu_baseline"] = cb

    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
